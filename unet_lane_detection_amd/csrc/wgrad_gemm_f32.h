// Pixel-reduction GEMM on the fp32 matrix pipe: C[rows][cols] = sum over pixels k of A[k][rows] * B[k][cols].
// Used for the weight gradients that are plain GEMMs (training backward):
//   * ConvTranspose2d(2,2): rows = (a,b,co) of the space-to-depth output gradient, cols = ci;
//   * the first 3x3 convolution (Cin = 3): rows = co, cols = (ci, tap) of an im2col'd input (27 -> 32 columns),
//     where the 64x64-tile kernels would spend 16x their useful work on padding.
// wgrad_f32_kernel<1> ran these at 45 TFLOP/s: four accumulators per wave and one k-group of loads in flight.
//
// Block = 512 threads = 8 waves = 4 (rows) x 2 (cols); wave tile (16*MS) x (16*NS), block tile (64*MS) x (32*NS):
// MS=4, NS=2 (256 x 64) for the upconvs, MS=1, NS=1 (64 x 32) for the first layer.  Both operands are staged by
// LDS-DMA in stages of 32 pixels, double buffered, as [pixel][tile rows] / [pixel][tile cols]; a lane of K-step s
// (pixel 4s + q) reads its MS consecutive rows and NS consecutive cols with one ds_read each.  K is split over
// blocks; partial tiles go to the slab [split][RowsPad][ColsPad] that wgrad_reduce_kernel adds in split order.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"
#include <stdint.h>

namespace unet {

typedef float wgf4 __attribute__((ext_vector_type(4)));

struct WgradGemmArgs {
  const float* a;      // [K][lda], rows [0, rows)
  const float* b;      // [K][ldb], cols [0, cols)
  const float* zeros;  // >= 4 zero floats
  float* slab;         // [splits][RowsPad][ColsPad]
  long K;
  int lda, ldb, rows, cols;   // rows, cols multiples of 4
  int RowsPad, ColsPad;       // multiples of the block tile
  int nStages;                // ceil(K / 32)
  int stagesPerSplit;
};

template <int N>
struct WgVec;
template <>
struct WgVec<1> {
  typedef float type;
};
template <>
struct WgVec<2> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct WgVec<4> {
  typedef float type __attribute__((ext_vector_type(4)));
};
template <int N>
__device__ __forceinline__ float wg_get(const typename WgVec<N>::type& v, int e) {
  return v[e];
}
template <>
__device__ __forceinline__ float wg_get<1>(const float& v, int) {
  return v;
}

template <int MS, int NS>
__global__ __launch_bounds__(512, 1) void wgrad_gemm_f32_kernel(const WgradGemmArgs a) {
  constexpr int RT = 64 * MS, CT = 32 * NS;          // block tile
  constexpr int ABYTES = 32 * RT * 4, BBYTES = 32 * CT * 4;
  constexpr int STAGE = ABYTES + BBYTES;
  constexpr int NQA = ABYTES / 1024, NQ = STAGE / 1024;
  constexpr int NQW = (NQ + 7) / 8;
  typedef typename WgVec<MS>::type avec;
  typedef typename WgVec<NS>::type bvec;

  extern __shared__ __attribute__((aligned(16))) char wgsmem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x;
  const int col0 = blockIdx.y * CT, row0 = blockIdx.z * RT;
  const int sBegin = split * a.stagesPerSplit;
  const int sEnd = sBegin + a.stagesPerSplit < a.nStages ? sBegin + a.stagesPerSplit : a.nStages;
  const int nMine = sEnd > sBegin ? sEnd - sBegin : 0;

  wgf4 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ms][ns] = (wgf4){0.f, 0.f, 0.f, 0.f};

  const int alane = (lq * RT + wm * 16 * MS + li * MS) * 4;
  const int blane = ABYTES + (lq * CT + wn * 16 * NS + li * NS) * 4;

  // Iteration `it` issues the DMA of stage `it`, multiplies stage `it - 1`, waits for the DMA, joins the barrier.
  for (int it = 0; it <= nMine; ++it) {
    if (it < nMine) {
      const long k0 = (long)(sBegin + it) * 32;
      char* dst = wgsmem + (it & 1) * STAGE;
#pragma unroll
      for (int j = 0; j < NQW; ++j) {
        const int q = wave + 8 * j;
        if (q < NQ) {   // uniform
          const float* src;
          if (q < NQA) {
            const int f = q * 256 + lane * 4;
            const int px = f / RT, e = f - px * RT;
            const bool ok = k0 + px < a.K && row0 + e < a.rows;
            src = ok ? a.a + (size_t)(k0 + px) * (size_t)a.lda + row0 + e : a.zeros;
          } else {
            const int f = (q - NQA) * 256 + lane * 4;
            const int px = f / CT, e = f - px * CT;
            const bool ok = k0 + px < a.K && col0 + e < a.cols;
            src = ok ? a.b + (size_t)(k0 + px) * (size_t)a.ldb + col0 + e : a.zeros;
          }
          lds_dma16(src, lds_address(dst) + q * 1024);
        }
      }
    }
    if (it >= 1) {
      const char* buf = wgsmem + ((it - 1) & 1) * STAGE;
      const char* ab = buf + alane;
      const char* bb = buf + blane;
      avec av[8];
      bvec bv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        av[s] = *reinterpret_cast<const avec*>(ab + s * 4 * RT * 4);
        bv[s] = *reinterpret_cast<const bvec*>(bb + s * 4 * CT * 4);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg_get<MS>(av[s], ms), wg_get<NS>(bv[s], ns),
                                                               acc[ms][ns], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // accumulator (ms, ns): MFMA row index i' = 4*lq + r is block row wm*16*MS + MS*i' + ms, column index li is
  // block column wn*16*NS + NS*li + ns
  float* sl = a.slab + (size_t)split * a.RowsPad * a.ColsPad;
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wm * 16 * MS + MS * (4 * lq + r) + ms;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        sl[(size_t)row * a.ColsPad + col0 + wn * 16 * NS + NS * li + ns] = acc[ms][ns][r];
    }
}

// im2col of the first layer's input for its weight gradient: x (N,H,W,4) fp32 with channel 3 zero ->
// out[pixel][32], column n = ci*9 + tap (n < 27) = x[pixel + tap offset][ci] with zero padding, columns 27..31 zero.
__global__ __launch_bounds__(256) void im2col27_kernel(const float* __restrict__ x, int n, int h, int w,
                                                       float* __restrict__ out) {
  const size_t total = (size_t)n * h * w * 8;   // one float4 of the 32 columns per thread
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int g = (int)(i & 7);
    const size_t p = i >> 3;
    const int xx = (int)(p % w);
    const size_t t2 = p / w;
    const int yy = (int)(t2 % h);
    wgf4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = g * 4 + e;
      float val = 0.f;
      if (col < 27) {
        const int ci = col / 9, t = col - ci * 9;
        const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
        if (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w)
          val = x[(p + (size_t)((long)(y2 - yy) * w + (x2 - xx))) * 4 + ci];
      }
      v[e] = val;
    }
    *reinterpret_cast<wgf4*>(out + i * 4) = v;
  }
}

}  // namespace unet
