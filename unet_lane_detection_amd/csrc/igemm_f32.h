// Implicit-GEMM convolution on the gfx950 fp32 matrix pipe.
//
// One kernel template covers the three GEMM-shaped operators of the U-Net
// forward (reference README.md:1452, :1442):
//   TAPS = 9 : 3x3 cross-correlation, stride 1, pad 1      (M = pixels, K = 9*Cin, N = Cout)
//   TAPS = 1 : ConvTranspose2d k=2 s=2 as a 1x1 GEMM        (M = input pixels, K = Cin, N = 4*Cout)
// Arithmetic is v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate,
// k-ordered fma chain (same numerics class as the fp32 oracle).
//
// Data layout
//   activations : NHWC fp32, pixel stride = Cin (padded to a multiple of CK)
//   weights     : pre-packed on the host in MFMA B-fragment order
//                 [co_subtile(16)][k_chunk(CK)][tap][lane(64)][CK/4]
//                 so that every B fragment is ONE lane-linear 16-byte (CK=16)
//                 load per lane, 1 KiB contiguous per wave, straight to VGPRs.
//   LDS         : only the input halo tile, [(TH+2)*(TW+2) pixels][CK], double
//                 buffered; the 9 taps re-read it at shifted offsets, so each
//                 input element crosses HBM/L2 -> CU once per output-channel tile.
//
// Work decomposition
//   block = 256 threads = 4 waves as 2 (pixels) x 2 (channels);
//   block tile = BM x BN = (32*MS) pixels x (32*NS) channels; a pixel tile is
//   TH consecutive "global rows" (row index g = n*H + y over the whole batch)
//   x TW columns, so tiles may straddle images (H = 14, 28 are not multiples
//   of 8); every boundary condition (image top/bottom, left/right, ragged
//   last tile) is a per-lane 9-bit tap-validity mask applied to the A
//   fragment, never a property of the staged data.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;     // NHWC, pixel stride Cin
  const float* wt;     // packed fragments
  const float* scale;  // [n_total_pad] per output column
  const float* shift;  // [n_total_pad]
  float* out;
  int N, H, W;         // input batch and spatial size
  int Cin;             // multiple of CK
  int Cout;            // real output channels per (a,b) group
  int CoutPad;         // Cout rounded up to 16 (TAPS=1: per (a,b) group)
  int ldo;             // output pixel stride (floats)
  int co_off;          // output channel offset (concat slice)
  int TH, TW;          // pixel tile: TH global rows x TW columns, TH*TW == BM
  int tilesX;          // ceil(W / TW)
  int nChunks;         // Cin / CK
  int relu;
  int coTiles;         // number of BN-wide output-channel tiles
  int coGroup;         // channel tiles interleaved on consecutive block ids (<= 8, divides coTiles)
  int pixTiles;        // number of pixel tiles
  int out_bf16;        // 1: `out` is a bf16 (uint16) tensor - the fp32 first layer feeding the bf16 tier
};

// fp32 -> bf16, round to nearest even (activations are finite here; NaN handling is not needed)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float v) {
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int KPL> struct KFrag;
template <> struct KFrag<4> { typedef f32x4 type; };
template <> struct KFrag<1> { typedef float type; };

template <int KPL> __device__ __forceinline__ float kget(const typename KFrag<KPL>::type& v, int e);
template <> __device__ __forceinline__ float kget<4>(const f32x4& v, int e) { return v[e]; }
template <> __device__ __forceinline__ float kget<1>(const float& v, int) { return v; }

// MODE 0: conv output NHWC (pixel stride ldo, channel offset co_off)
// MODE 1: ConvTranspose 2x2 scatter: column n = (a*2+b)*CoutPad + co -> pixel (2y+a, 2x+b)
template <int CK, int TAPS, int MS, int NS, int NLD, int MODE>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const ConvArgs a) {
  constexpr int WN = 2;
  constexpr int KPL = CK / 4;          // k values per lane per chunk
  constexpr int VPP = CK / 4;          // float4 vectors per staged pixel
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int BUF_FLOATS = NLD * 256 * 4;
  constexpr int ZERO_OFF = 2 * BUF_FLOATS;  // 16 zero floats behind the two buffers
  typedef typename KFrag<KPL>::type kfrag;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int li = lane & 15;   // row (pixel) index inside a 16x16 A fragment / column of B
  const int lq = lane >> 4;   // k group

  // Block id -> (pixel tile, channel tile).  Consecutive logical ids walk `coGroup` channel tiles of ONE
  // pixel tile, then the next pixel tile; channel-tile groups are outermost.  The dispatcher deals
  // consecutive blocks round-robin over the 8 XCDs, so the logical id is remapped to put consecutive ids on
  // ONE XCD: its L2 then serves the input tile to all channel tiles (measured on the Winograd kernel: half
  // the fabric traffic, +0.6 % frames/s, against spreading a pixel tile's channel tiles over the XCDs).
  // Placement only affects speed, never results.
  int bid = blockIdx.x;
  {
    const int g8 = (int)gridDim.x & ~7;
    if (bid < g8) bid = (bid & 7) * (g8 >> 3) + (bid >> 3);
  }
  const int cInG = bid % a.coGroup;
  const int rest = bid / a.coGroup;
  const int tile = rest % a.pixTiles;
  const int coTile = (rest / a.pixTiles) * a.coGroup + cInG;
  const int NH = a.N * a.H;
  const int tx = tile % a.tilesX;
  const int g0 = (tile / a.tilesX) * a.TH;
  const int x0 = tx * a.TW;
  const int HW2 = a.TW + 2 * HALO;
  const int HH2 = a.TH + 2 * HALO;

  // ---- staging plan: which 16-byte vectors of the halo tile this thread moves ----
  const int totalVec = HH2 * HW2 * VPP;
  // Slots past the end of the tile re-load its last vector into the unused tail of the LDS buffer,
  // so neither the loads nor the LDS writes need a predicate (keeps vmcnt countable).
  size_t srcOff[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    int idx = tid + j * 256;
    idx = idx < totalVec ? idx : totalVec - 1;
    const int pix = idx / VPP;
    const int v = idx - pix * VPP;
    const int hr = pix / HW2;
    const int hc = pix - hr * HW2;
    int g = g0 - HALO + hr;
    int x = x0 - HALO + hc;
    g = g < 0 ? 0 : (g > NH - 1 ? NH - 1 : g);
    x = x < 0 ? 0 : (x > a.W - 1 ? a.W - 1 : x);
    srcOff[j] = ((size_t)g * a.W + x) * (size_t)a.Cin + v * 4;
  }

  // ---- per-lane A-fragment plan: LDS offset and tap validity of each 16-pixel subtile ----
  int aOff[MS];
  unsigned aMask[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int p = (wm * MS + ms) * 16 + li;
    const int r = p / a.TW;
    const int c = p - r * a.TW;
    const int g = g0 + r;
    const int x = x0 + c;
    const int y = g % a.H;
    aOff[ms] = (r * HW2 + c) * CK + lq * KPL;
    unsigned m = 0;
    if (g < NH && x < a.W) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ky = (TAPS == 9) ? t / 3 : 1;
        const int kx = (TAPS == 9) ? t % 3 : 1;
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

  // B fragment streams: subtile cs = (coTile*NS + ns)*WN + wn; fragments of one
  // subtile are contiguous over (chunk, tap).
  const kfrag* bPtr[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const size_t cs = ((size_t)coTile * NS + ns) * WN + wn;
    bPtr[ns] = reinterpret_cast<const kfrag*>(a.wt) + (cs * a.nChunks * TAPS) * 64 + lane;
  }

  // ---- prologue: stage chunk 0, fetch first B fragments ----
  f32x4 stage[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j)
    stage[j] = *reinterpret_cast<const f32x4*>(a.in + srcOff[j]);
  kfrag bCur[NS], bNxt[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) bCur[ns] = bPtr[ns][0];
#pragma unroll
  for (int j = 0; j < NLD; ++j)
    *reinterpret_cast<f32x4*>(smem + (tid + j * 256) * 4) = stage[j];
  if (tid < 4) *reinterpret_cast<f32x4*>(smem + ZERO_OFF + tid * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int kc = 0; kc < a.nChunks; ++kc) {
    // Stage the next chunk unconditionally (the last iteration re-stages its own chunk into the
    // idle buffer): no branch around the loads, so the compiler can count vmcnt instead of draining.
    const int kn = (kc + 1) < a.nChunks ? kc + 1 : kc;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      stage[j] = *reinterpret_cast<const f32x4*>(a.in + srcOff[j] + (size_t)kn * CK);
    // keep the loads up here: without the fence the scheduler sinks them next to the LDS writes
    // at the end of the chunk and exposes the full memory latency once per chunk
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      // next B fragment (the packed buffer carries one spare fragment per subtile stream end)
      const int nextIdx = (kc * TAPS + t + 1) * 64;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bNxt[ns] = bPtr[ns][nextIdx];

      const int ky = (TAPS == 9) ? t / 3 : 0;
      const int kx = (TAPS == 9) ? t % 3 : 0;
      const int tapOff = (ky * HW2 + kx) * CK;
      kfrag af[MS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        // an invalid tap (outside the image) reads the zero slot instead of masking the data
        int off = (kc & 1) * BUF_FLOATS + aOff[ms] + tapOff;
        if (TAPS == 9 && t != 4) off = ((aMask[ms] >> t) & 1u) ? off : ZERO_OFF;
        af[ms] = *reinterpret_cast<const kfrag*>(smem + off);
      }
#pragma unroll
      for (int e = 0; e < KPL; ++e)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(kget<KPL>(af[ms], e), kget<KPL>(bCur[ns], e),
                                                               acc[ms][ns], 0, 0, 0);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bCur[ns] = bNxt[ns];
    }
    {
      float* nbuf = smem + ((kc + 1) & 1) * BUF_FLOATS;
#pragma unroll
      for (int j = 0; j < NLD; ++j)
        *reinterpret_cast<f32x4*>(nbuf + (tid + j * 256) * 4) = stage[j];
    }
    __syncthreads();
  }

  // ---- epilogue: y = acc*scale + shift (+ReLU).  The accumulator layout gives a lane one channel of four
  //      pixels (64 contiguous bytes per 16 lanes), so the tile is first transposed through LDS - the staging
  //      buffers are free now - one pixel half (wm) at a time, then written as whole 16-byte vectors:
  //      BN*4 contiguous bytes per output pixel ----
  constexpr int BN = 32 * NS;
  constexpr int LDB = BN + 4;            // row pitch in floats (keeps 16-byte alignment, skews banks)
  constexpr int HALF_ROWS = MS * 16;
  constexpr int V4_PER_ROW = BN / 4;
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
            smem[(ms * 16 + lq * 4 + r) * LDB + (ns * WN + wn) * 16 + li] = v;
          }
    }
    __syncthreads();
    for (int idx = tid; idx < HALF_ROWS * V4_PER_ROW; idx += 256) {
      const int row = idx / V4_PER_ROW, c4 = idx - row * V4_PER_ROW;
      const int p = half * HALF_ROWS + row;
      const int rr = p / a.TW, cc = p - rr * a.TW;
      const int g = g0 + rr, x = x0 + cc;
      if (g >= NH || x >= a.W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * LDB + c4 * 4);
      const int n0 = coTile * BN + c4 * 4;
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
      if (cbase + 3 < climit) {
        if (a.out_bf16) {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16_rne(v[0]) | ((uint32_t)f32_to_bf16_rne(v[1]) << 16);
          pk.y = (uint32_t)f32_to_bf16_rne(v[2]) | ((uint32_t)f32_to_bf16_rne(v[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + o) = pk;
        } else {
          *reinterpret_cast<f32x4*>(a.out + o) = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cbase + e < climit) {
            if (a.out_bf16)
              reinterpret_cast<uint16_t*>(a.out)[o + e] = f32_to_bf16_rne(v[e]);
            else
              a.out[o + e] = v[e];
          }
      }
    }
    __syncthreads();
  }
}

}  // namespace unet
