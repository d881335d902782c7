// First convolution of the bf16 tier: uint8 RGB frames -> normalise -> 3x3 conv (3 -> Cout) + folded BN + ReLU ->
// bf16 NHWC, in one kernel.  (Before: pack_u8 to an fp32 NHWC4 copy, then the fp32 implicit-GEMM kernel with a
// bf16 store: 4.5 ms at batch 1024 for 6.6 GB of output, MFMA pipe 32 % busy on K = 27.)
//
// K = 27 fits ONE v_mfma_f32_16x16x32_bf16 (k = tap*3 + ci, 27..31 zero).  To keep fp32-level accuracy on this
// layer - the input has 8 significant bits and the old path computed it in fp32 - both operands are split into
// bf16 hi + lo and three MFMAs are issued per tile (w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulate): products
// are exact to ~2^-16.  The kernel is bound by its 128 B/pixel stores, not by the 12 MFMAs per 16 pixels.
//
// Block = 256 threads, tile = 8 rows x 32 columns of one image (H % 8 == 0).  Stage 1: the uint8 halo
// (10 x 34 x 3 bytes) -> normalised fp32 in LDS (zero outside the image: padding applies to the normalised tensor,
// reference README.md:1427).  Stage 2: thread p builds pixel p's im2col row as bf16 hi | lo (2 x 64 bytes in LDS).
// Stage 3: wave w multiplies pixels [64w, 64w+64): weights are the MFMA A operand with the channel permutation of
// conv_bf16_ws.h, so a lane ends up with 16 consecutive channels of one pixel and stores 32 bytes.
#pragma once
#include "conv_bf16_ws.h"

namespace unet {

struct ConvFirstArgs {
  const uint8_t* frames;   // (N,H,W,3) uint8
  const uint16_t* wt;      // [coTile(64 ch)][cs(4)][hi|lo][lane(64)][8] bf16, see pack_first_bf16x3
  const float* scale;
  const float* shift;
  uint16_t* out;           // (N,H,W,ldo) bf16, channels [0,Cout)
  int N, H, W, Cout, ldo, tilesX, relu;
  float m0, m1, m2, s0, s1, s2;   // (x - m) / s as pack_u8_nhwc4_kernel
};

__global__ __launch_bounds__(256) void conv_first_bf16x3_kernel(const ConvFirstArgs a) {
  constexpr int TH = 8, TW = 32, HR = TH + 2, HC = TW + 2;
  __shared__ float halo[HR * HC * 3 + 4];
  __shared__ __attribute__((aligned(16))) uint32_t rows[2][256][16 + 4];   // [hi|lo][pixel][32 bf16 (+16 B pad)]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int tile = blockIdx.x;
  const int g0 = (tile / a.tilesX) * TH, x0 = (tile % a.tilesX) * TW;
  const int y0 = g0 % a.H;

  // ---- stage 1: normalised halo.  All of a thread's byte loads before the first conversion (conv_first_x3.h: as a plain
  //      loop hipcc waited vmcnt(0) behind every load, four round trips to memory per block in a row) ----
  {
    constexpr int N1 = (HR * HC * 3 + 255) / 256;
    unsigned rawb[N1];
    bool inb[N1];
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int i = it * 256 + tid;
      const int px = i / 3, ci = i - px * 3;
      const int hr = px / HC, hc = px - hr * HC;
      const int y = y0 - 1 + hr, x = x0 - 1 + hc;
      inb[it] = i < HR * HC * 3 && y >= 0 && y < a.H && x >= 0 && x < a.W;
      rawb[it] = 0;
      if (inb[it]) rawb[it] = a.frames[((size_t)(g0 - 1 + hr) * a.W + x) * 3 + ci];
    }
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int i = it * 256 + tid;
      const int ci = i % 3;
      float v = 0.f;
      if (inb[it]) {
        const float m = ci == 0 ? a.m0 : (ci == 1 ? a.m1 : a.m2);
        const float s = ci == 0 ? a.s0 : (ci == 1 ? a.s1 : a.s2);
        v = ((float)rawb[it] - m) / s;
      }
      if (i < HR * HC * 3) halo[i] = v;
    }
  }
  __syncthreads();

  // ---- stage 2: im2col row of pixel tid: k = tap*3 + ci, split into bf16 hi and lo ----
  {
    const int r = tid / TW, c = tid - r * TW;
    uint32_t hi[16], lo[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * k2 + e;
        if (k < 27) {
          const int t = k / 3, ci = k - t * 3;
          v[e] = halo[((r + t / 3) * HC + c + t % 3) * 3 + ci];
        } else {
          v[e] = 0.f;
        }
      }
      hi[k2] = pk_bf16(v[0], v[1]);
      const float h0 = __builtin_bit_cast(float, hi[k2] << 16), h1 = __builtin_bit_cast(float, hi[k2] & 0xFFFF0000u);
      lo[k2] = pk_bf16(v[0] - h0, v[1] - h1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<uint4*>(&rows[0][tid][q * 4]) = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
      *reinterpret_cast<uint4*>(&rows[1][tid][q * 4]) = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
    }
  }
  __syncthreads();

  // ---- stage 3: MFMA, 64 channels at a time ----
  const int nCt = a.Cout / 64;
  for (int ct = 0; ct < nCt; ++ct) {
    f32x4 wh[4], wl[4];
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wt) + (size_t)ct * (4 * 2 * 64) + lane;
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      wh[cs] = wp[(cs * 2 + 0) * 64];
      wl[cs] = wp[(cs * 2 + 1) * 64];
    }
    const int cbase = ct * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cbase + cs * 4);
    }
    const float lo0 = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int p = wave * 64 + f * 16 + li;
      const f32x4 xh = *reinterpret_cast<const f32x4*>(&rows[0][p][lq * 4]);
      const f32x4 xl = *reinterpret_cast<const f32x4*>(&rows[1][p][lq * 4]);
      uint32_t pk[8];
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[cs]), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[cs]), __builtin_bit_cast(bf16x8, xl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[cs]), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[r], sc[cs][r], sh[cs][r]), lo0);
        pk[cs * 2] = pk_bf16(v[0], v[1]);
        pk[cs * 2 + 1] = pk_bf16(v[2], v[3]);
      }
      const int r = p / TW, c = p - r * TW;
      const int x = x0 + c;
      if (x < a.W) {
        uint4* o = reinterpret_cast<uint4*>(a.out + ((size_t)(g0 + r) * a.W + x) * (size_t)a.ldo + cbase);
        o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
  }
}

}  // namespace unet
