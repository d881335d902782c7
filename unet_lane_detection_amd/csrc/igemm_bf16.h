// bf16 implicit-GEMM convolution (BASELINE.json configs[2]: bf16 storage, fp32 accumulate).
//
// Same decomposition as igemm_f32.h - LDS halo tile of the input read at 9 shifted offsets, weights
// pre-packed in MFMA B-fragment order and streamed to VGPRs, per-lane tap-validity masks with a zero slot -
// on v_mfma_f32_16x16x32_bf16: lane (i = lane&15, q = lane>>4) holds k = 8q..8q+7 of a 32-channel chunk for
// both operands, so a pixel's chunk is 64 bytes in LDS exactly as in the fp32 kernel and every fragment is
// one 16-byte read.  Activations are bf16 NHWC (channel counts multiples of 32 on the input side);
// accumulation, the folded-BatchNorm scale/shift and ReLU are fp32; the result is rounded to bf16 once
// (round-to-nearest-even through the hardware convert) or kept fp32 (OUT32, used by nothing yet).
//
// This is the first, structure-sharing version of the bf16 tier: the MFMA pipe is 16x faster than in fp32
// while LDS, L1 and the epilogue are not, so this kernel is bound by operand delivery, not by MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "igemm_f32.h"

namespace unet {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgsBf {
  const uint16_t* in;   // NHWC bf16, pixel stride Cin (multiple of 32)
  const uint16_t* wt;   // packed fragments [cs][chunk(32)][tap][lane][8]
  const float* scale;
  const float* shift;
  uint16_t* out;        // NHWC bf16
  int N, H, W, Cin, Cout, CoutPad, ldo, co_off, TH, TW, tilesX, nChunks, relu;
  int coTiles, coGroup, pixTiles;
  uint16_t* pool;        // MODE 0, optional: (N,H/2,W/2,Cout) max-pooled copy (needs TH % 4 == 0, TW % 2 == 0)
  const float* headW;    // MODE 0, optional: fused 1x1 head (needs a single channel tile): weights [Cout]
  float headB, headThr;
  float* logits;         // (N,H,W) fp32 outputs of the fused head (any may be null)
  float* probs;
  uint8_t* mask;
  int storeOut;          // 0: the activation itself is not written (its only consumer is the fused head)
};

__device__ __forceinline__ uint16_t f2bf(float v) { return f32_to_bf16_rne(v); }

template <int TAPS, int MS, int NS, int NLD, int MODE>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(const ConvArgsBf a) {
  constexpr int WN = 2;
  constexpr int CK = 32;                 // channels per chunk (64 bytes per pixel)
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int BUF_VEC = NLD * 256;     // 16-byte vectors per LDS buffer
  constexpr int ZERO_VEC = 2 * BUF_VEC;  // 4 zero vectors behind the two buffers

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lq = lane >> 4;

  const int bid = blockIdx.x;
  const int cInG = bid % a.coGroup;
  const int rest = bid / a.coGroup;
  const int tile = rest % a.pixTiles;
  const int coTile = (rest / a.pixTiles) * a.coGroup + cInG;
  const int NH = a.N * a.H;
  const int tx = tile % a.tilesX;
  const int g0 = (tile / a.tilesX) * a.TH;
  const int x0 = tx * a.TW;
  const int HW2 = a.TW + 2 * HALO, HH2 = a.TH + 2 * HALO;

  const int totalVec = HH2 * HW2 * 4;
  const f32x4* src[NLD];   // 16-byte vectors: 8 bf16
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    int idx = tid + j * 256;
    idx = idx < totalVec ? idx : totalVec - 1;
    const int pix = idx >> 2, v = idx & 3;
    const int hr = pix / HW2, hc = pix - hr * HW2;
    int g = g0 - HALO + hr, x = x0 - HALO + hc;
    g = g < 0 ? 0 : (g > NH - 1 ? NH - 1 : g);
    x = x < 0 ? 0 : (x > a.W - 1 ? a.W - 1 : x);
    src[j] = reinterpret_cast<const f32x4*>(a.in + ((size_t)g * a.W + x) * (size_t)a.Cin + v * 8);
  }

  int aOff[MS];        // in 16-byte vectors
  unsigned aMask[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int p = (wm * MS + ms) * 16 + li;
    const int r = p / a.TW, c = p - r * a.TW;
    const int g = g0 + r, x = x0 + c;
    const int y = g % a.H;
    aOff[ms] = (r * HW2 + c) * 4 + lq;
    unsigned m = 0;
    if (g < NH && x < a.W) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ky = (TAPS == 9) ? t / 3 : 1, kx = (TAPS == 9) ? t % 3 : 1;
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) m |= 1u << t;
      }
    }
    aMask[ms] = m;
  }

  f32x4 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const f32x4* bPtr[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const size_t cs = ((size_t)coTile * NS + ns) * WN + wn;
    bPtr[ns] = reinterpret_cast<const f32x4*>(a.wt) + (cs * a.nChunks * TAPS) * 64 + lane;
  }

  f32x4 stage[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) stage[j] = *src[j];
  f32x4 bCur[NS], bNxt[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) bCur[ns] = bPtr[ns][0];
#pragma unroll
  for (int j = 0; j < NLD; ++j) smemv[tid + j * 256] = stage[j];
  if (tid < 4) smemv[ZERO_VEC + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int kc = 0; kc < a.nChunks; ++kc) {
    const int kn = (kc + 1) < a.nChunks ? kc + 1 : kc;
#pragma unroll
    for (int j = 0; j < NLD; ++j) stage[j] = src[j][(size_t)kn * 4];   // next chunk: +64 bytes per pixel
    __builtin_amdgcn_sched_barrier(0);
    auto readA = [&](f32x4 (&dst)[MS], int t) {
      const int ky = (TAPS == 9) ? t / 3 : 0, kx = (TAPS == 9) ? t % 3 : 0;
      const int tapOff = (ky * HW2 + kx) * 4;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        int off = (kc & 1) * BUF_VEC + aOff[ms] + tapOff;
        if (TAPS == 9 && t != 4) off = ((aMask[ms] >> t) & 1u) ? off : ZERO_VEC;
        dst[ms] = smemv[off];
      }
    };
    f32x4 af[MS], afN[MS];
    readA(af, 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int nextIdx = (kc * TAPS + t + 1) * 64;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bNxt[ns] = bPtr[ns][nextIdx];
      // A fragments one tap ahead: a bf16 MFMA is too short to hide a ds_read issued in its own tap, and
      // letting the scheduler hoist all nine taps' reads spills (9 x MS x 4 VGPRs)
      if (t + 1 < TAPS) readA(afN, t + 1);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ms]),
                                                                __builtin_bit_cast(bf16x8, bCur[ns]),
                                                                acc[ms][ns], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bCur[ns] = bNxt[ns];
      if (t + 1 < TAPS) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) af[ms] = afN[ms];
      }
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) smemv[((kc + 1) & 1) * BUF_VEC + tid + j * 256] = stage[j];
    __syncthreads();
  }

  // ---- epilogue: fp32 scale/shift/ReLU, transposed through LDS (staging buffers are free) one pixel half
  //      at a time, rounded to bf16 and written as 16-byte vectors of 8 channels ----
  constexpr int BN = 32 * NS;
  constexpr int LDB = BN + 4;
  constexpr int HALF_ROWS = MS * 16;
  constexpr int V8_PER_ROW = BN / 8;
  float* ep = reinterpret_cast<float*>(smemv);
  float sc[NS], sh[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int n = (coTile * NS + ns) * (WN * 16) + wn * 16 + li;
    sc[ns] = a.scale[n];
    sh[ns] = a.shift[n];
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[ms][ns][r] * sc[ns] + sh[ns];
            if (a.relu) v = v > 0.f ? v : 0.f;
            ep[(ms * 16 + lq * 4 + r) * LDB + (ns * WN + wn) * 16 + li] = v;
          }
    }
    __syncthreads();
    for (int idx = tid; a.storeOut && idx < HALF_ROWS * V8_PER_ROW; idx += 256) {
      const int row = idx / V8_PER_ROW, c8 = idx - row * V8_PER_ROW;
      const int p = half * HALF_ROWS + row;
      const int rr = p / a.TW, cc = p - rr * a.TW;
      const int g = g0 + rr, x = x0 + cc;
      if (g >= NH || x >= a.W) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + row * LDB + c8 * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + row * LDB + c8 * 8 + 4);
      const int n0 = coTile * BN + c8 * 8;
      size_t o;
      int cbase, climit;
      if (MODE == 0) {
        o = ((size_t)g * a.W + x) * (size_t)a.ldo + a.co_off + n0;
        cbase = n0;
        climit = a.Cout;
      } else {
        const int ab = n0 / a.CoutPad;
        cbase = n0 - ab * a.CoutPad;
        climit = ab < 4 ? a.Cout : 0;
        const size_t og = (size_t)g * 2 + (ab >> 1), ox = (size_t)x * 2 + (ab & 1);
        o = (og * (size_t)(2 * a.W) + ox) * (size_t)a.ldo + a.co_off + cbase;
      }
      if (cbase + 7 < climit) {
        uint4 pk;
        pk.x = (uint32_t)f2bf(v0[0]) | ((uint32_t)f2bf(v0[1]) << 16);
        pk.y = (uint32_t)f2bf(v0[2]) | ((uint32_t)f2bf(v0[3]) << 16);
        pk.z = (uint32_t)f2bf(v1[0]) | ((uint32_t)f2bf(v1[1]) << 16);
        pk.w = (uint32_t)f2bf(v1[2]) | ((uint32_t)f2bf(v1[3]) << 16);
        *reinterpret_cast<uint4*>(a.out + o) = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (cbase + e < climit) a.out[o + e] = f2bf(e < 4 ? v0[e] : v1[e - 4]);
      }
    }
    if (MODE == 0 && a.pool) {
      // MaxPool2d(2,2) of this half tile straight from the LDS image (a half holds TH/2 whole rows, an even
      // number, and tiles start on even rows/columns, so every 2x2 window is complete): no separate pool pass
      const int PW = a.TW >> 1, PR = (HALF_ROWS / a.TW) >> 1;
      const int halfRow0 = half * (HALF_ROWS / a.TW);
      for (int idx = tid; idx < PR * PW * V8_PER_ROW; idx += 256) {
        const int c8 = idx % V8_PER_ROW;
        const int pp = idx / V8_PER_ROW;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int g = g0 + halfRow0 + 2 * pr, x = x0 + 2 * pc;
        if (g >= NH || x >= a.W) continue;
        const int n0 = coTile * BN + c8 * 8;
        if (n0 >= a.Cout) continue;
        const float* r0 = ep + ((2 * pr) * a.TW + 2 * pc) * LDB + c8 * 8;
        const float* r1 = r0 + a.TW * LDB;
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float m0 = fmaxf(fmaxf(r0[e], r0[LDB + e]), fmaxf(r1[e], r1[LDB + e]));
          const float m1 = fmaxf(fmaxf(r0[e + 1], r0[LDB + e + 1]), fmaxf(r1[e + 1], r1[LDB + e + 1]));
          pk[e >> 1] = (uint32_t)f2bf(m0) | ((uint32_t)f2bf(m1) << 16);
        }
        *reinterpret_cast<uint4*>(a.pool + (((size_t)(g >> 1)) * (a.W >> 1) + (x >> 1)) * (size_t)a.Cout + n0) =
            make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
    }
    if (MODE == 0 && a.headW) {
      // fused 1x1 head (reference README.md:1447): one thread per pixel of the half tile, fp32 dot over the
      // Cout (= BN) values of its LDS row
      for (int row = tid; row < HALF_ROWS; row += 256) {
        const int p = half * HALF_ROWS + row;
        const int rr = p / a.TW, cc = p - rr * a.TW;
        const int g = g0 + rr, x = x0 + cc;
        if (g >= NH || x >= a.W) continue;
        float z = 0.f;
        for (int c = 0; c < a.Cout; c += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * LDB + c);
          // the unfused path rounds the activation to bf16 before the head reads it: keep that rounding
          z = fmaf(__builtin_bit_cast(float, (uint32_t)f2bf(v[0]) << 16), a.headW[c], z);
          z = fmaf(__builtin_bit_cast(float, (uint32_t)f2bf(v[1]) << 16), a.headW[c + 1], z);
          z = fmaf(__builtin_bit_cast(float, (uint32_t)f2bf(v[2]) << 16), a.headW[c + 2], z);
          z = fmaf(__builtin_bit_cast(float, (uint32_t)f2bf(v[3]) << 16), a.headW[c + 3], z);
        }
        z += a.headB;
        const size_t o = (size_t)g * a.W + x;
        if (a.logits) a.logits[o] = z;
        if (a.probs) a.probs[o] = 1.f / (1.f + __expf(-z));
        if (a.mask) a.mask[o] = z > a.headThr ? 255 : 0;
      }
    }
    __syncthreads();
  }
}

// ---- bf16 helpers of the forward path ------------------------------------------------------------------

// MaxPool2d(2,2) on bf16 NHWC (8 channels = 16 bytes per thread); max of bf16 values is exact.
__global__ __launch_bounds__(256) void maxpool2x2_bf16_kernel(const uint16_t* __restrict__ in,
                                                              uint16_t* __restrict__ out, int n, int h, int w, int c,
                                                              int ldi) {
  const int c8 = c >> 3;
  const int oh = h >> 1, ow = w >> 1;
  const size_t total = (size_t)n * oh * ow * c8;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % c8);
    size_t t = i / c8;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const size_t img = t / oh;
    const uint16_t* p = in + ((img * h + (size_t)oy * 2) * w + (size_t)ox * 2) * (size_t)ldi + cv * 8;
    bf16x8 v[4];
    v[0] = *reinterpret_cast<const bf16x8*>(p);
    v[1] = *reinterpret_cast<const bf16x8*>(p + ldi);
    v[2] = *reinterpret_cast<const bf16x8*>(p + (size_t)w * ldi);
    v[3] = *reinterpret_cast<const bf16x8*>(p + (size_t)w * ldi + ldi);
    bf16x8 m;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float best = -3.4e38f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = __builtin_bit_cast(float, (uint32_t)(uint16_t)v[k][e] << 16);
        best = f > best ? f : best;
      }
      m[e] = (short)(__builtin_bit_cast(uint32_t, best) >> 16);
    }
    *reinterpret_cast<bf16x8*>(out + i * 8) = m;
  }
}

// 1x1 head on bf16 activations: fp32 dot + bias -> fp32 logits (+ optional sigmoid / threshold).
template <int LPP>
__global__ __launch_bounds__(256) void head1x1_bf16_kernel(const uint16_t* __restrict__ in,
                                                           const float* __restrict__ w, float bias, size_t npix, int c,
                                                           float* __restrict__ logits, float* __restrict__ probs,
                                                           uint8_t* __restrict__ mask, float thr) {
  const int sub = threadIdx.x % LPP;
  const size_t pixPerBlock = 256 / LPP;
  size_t p = (size_t)blockIdx.x * pixPerBlock + threadIdx.x / LPP;
  const size_t stride = (size_t)gridDim.x * pixPerBlock;
  const size_t pEnd = (npix + pixPerBlock - 1) / pixPerBlock * pixPerBlock;
  for (; p < pEnd; p += stride) {
    float s = 0.f;
    if (p < npix) {
      const uint16_t* x = in + p * (size_t)c;
      for (int k = sub * 8; k < c; k += LPP * 8) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + k);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          s = fmaf(__builtin_bit_cast(float, (uint32_t)(uint16_t)xv[e] << 16), w[k + e], s);
      }
    }
#pragma unroll
    for (int d = LPP >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (sub == 0 && p < npix) {
      const float z = s + bias;
      if (logits) logits[p] = z;
      if (probs) probs[p] = 1.f / (1.f + __expf(-z));
      if (mask) mask[p] = z > thr ? 255 : 0;
    }
  }
}

}  // namespace unet
