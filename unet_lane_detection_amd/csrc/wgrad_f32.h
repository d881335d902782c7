// Weight-gradient GEMM on the fp32 matrix pipe (training backward of SURVEY.md section 8 rows a1/a5):
//   dW[co][ci][t] = sum over pixels p of dZ[p][co] * X[p + tap(t)][ci]        (3x3: 9 taps, 1x1: 1 tap)
// i.e. M = Cout, N = Cin (x taps), K = all N*H*W pixels.  K is split over blocks; every block writes its
// partial tile to a slab [split][tap][CoPad][CiPad] and a second kernel adds the slabs in split order
// (deterministic, no float atomics) and transposes to the PyTorch layout (O,I,kh,kw).
//
// Block = 256 threads = 4 waves; block tile = 64 co x 64 ci x TAPS.  Wave w owns co subtile w (16 rows)
// for all 4 ci subtiles and all taps: 4*TAPS accumulators of v_mfma_f32_16x16x4_f32.
//   A operand (dZ, k = pixel): one dword per lane straight from global memory (lane (i,q): co = i,
//     pixel = 4*step + q -> 64 contiguous bytes per pixel), prefetched one group of k-steps ahead.
//   B operand (X): the zero-filled halo tile of the current pixel tile lives in LDS [halo pixel][64 ci],
//     channels interleaved (c = 16*j + i at i*4 + j) so that a lane's four ci-subtile values of one tap are
//     one 16-byte read; the 9 taps read it at shifted pixel offsets (9 ds_read_b128 per 36 MFMAs).  The next
//     tile's halo is fetched into registers while the current tile is being multiplied.
// Pixel tiles (TH x TW, <= 128 pixels) never straddle images, so zero padding is a property of the
// staged halo and no per-lane masks are needed; ragged tiles are handled by zeroing the A operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {

typedef float wf4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const float* dz;   // (N,H,W,Cout) dense
  const float* x;    // (N,H,W,*) pixel stride ldx, channels [0,Cin)
  float* slab;       // [splits][taps][CoPad][CiPad]
  int N, H, W;
  int Cout, Cin, ldx;
  int CoPad, CiPad;  // multiples of 64
  int TH, TW;        // pixel tile inside an image
  int tilesY, tilesX;
  int tilesPerSplit; // pixel tiles walked by one block
  int nTiles;        // N * tilesY * tilesX
};

constexpr int WG_XSTRIDE = 64;    // floats per halo pixel in LDS; channel c = 16*j + i is stored at i*4 + j, so the four
                                  // B values a lane needs per tap (j = 0..3) are ONE conflict-free ds_read_b128
constexpr int WG_MAX_HALO = 180;  // (TH+2)*(TW+2) upper bound
constexpr int WG_GROUP = 4;       // k-steps per A-prefetch group

template <int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const WgradArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  __shared__ __attribute__((aligned(16))) float xs[WG_MAX_HALO * WG_XSTRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 64;
  const int co0 = blockIdx.z * 64;
  const int HW2 = a.TW + 2 * HALO, HH2 = a.TH + 2 * HALO;
  const int KT = a.TH * a.TW;
  const int kSteps = (KT + 3) >> 2;
  const int coA = co0 + wave * 16 + li;     // this lane's dZ column
  const bool coOk = coA < a.Cout;

  wf4 acc[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (wf4){0.f, 0.f, 0.f, 0.f};

  const int tBeg = split * a.tilesPerSplit;
  const int tEnd = (tBeg + a.tilesPerSplit < a.nTiles) ? tBeg + a.tilesPerSplit : a.nTiles;

  // ---- staging plan of the X halo tile: thread `tid` moves float4 number tid + j*256 of the tile
  //      ([halo pixel][16 float4]); its channel group (tid & 15) and, packed per entry, the halo coordinates
  //      are the same for every tile, only the tile origin changes ----
  constexpr int NX = (WG_MAX_HALO * 16 + 255) / 256;
  const int nVec = HH2 * HW2 * 16;
  const int xChan = ci0 + (tid & 15) * 4;
  const bool xChanOk = xChan < a.Cin;
  unsigned plan[NX];   // hr | hc << 8 | (halo pixel index) << 16; hr = 255 marks an unused slot
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int idx = tid + j * 256;
    const int pix = idx >> 4;
    const int hr = pix / HW2, hc = pix - hr * HW2;
    plan[j] = idx < nVec ? (unsigned)hr | ((unsigned)hc << 8) | ((unsigned)pix << 16) : 255u;
  }
  auto loadX = [&](wf4 (&dst)[NX], int tile) {
    const int tx = tile % a.tilesX;
    const int ty = (tile / a.tilesX) % a.tilesY;
    const int img = tile / (a.tilesX * a.tilesY);
    const int yb = ty * a.TH - HALO, xb = tx * a.TW - HALO;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int hr = plan[j] & 255, hc = (plan[j] >> 8) & 255;
      const int y = yb + hr, x = xb + hc;
      dst[j] = (wf4){0.f, 0.f, 0.f, 0.f};
      if (hr != 255 && xChanOk && y >= 0 && y < a.H && x >= 0 && x < a.W)
        dst[j] = *reinterpret_cast<const wf4*>(a.x + (((size_t)img * a.H + y) * a.W + x) * (size_t)a.ldx + xChan);
    }
  };

  wf4 xst[NX];
  if (tBeg < tEnd) loadX(xst, tBeg);
  for (int tile = tBeg; tile < tEnd; ++tile) {
    const int tx = tile % a.tilesX;
    const int ty = (tile / a.tilesX) % a.tilesY;
    const int img = tile / (a.tilesX * a.tilesY);
    const int y0 = ty * a.TH, x0 = tx * a.TW;

    __syncthreads();  // previous tile's LDS reads are done
    {
      // this thread's float4 holds channels 4*(tid&15) .. +3 = ci subtile jx, rows ix .. ix+3
      const int jx = (tid & 15) >> 2, ix = ((tid & 15) & 3) * 4;
#pragma unroll
      for (int j = 0; j < NX; ++j)
        if ((plan[j] & 255) != 255) {
          float* dstp = xs + (plan[j] >> 16) * WG_XSTRIDE + ix * 4 + jx;
#pragma unroll
          for (int e = 0; e < 4; ++e) dstp[e * 4] = xst[j][e];
        }
    }
    __syncthreads();
    // next tile's halo goes into registers now and lands under this tile's MFMAs (the staging loop used to
    // issue load -> wait -> LDS store twelve times in sequence before any MFMA of the tile could start)
    if (tile + 1 < tEnd) loadX(xst, tile + 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---- K loop over the pixels of the tile, 4 per MFMA.  Pixel p = 4*step + lq walks the tile row-major;
    //      (r, c) advance incrementally (no division in the loop).  A runs one group of steps ahead. ----
    const size_t dzImg = (size_t)img * a.H;
    int ar = lq / a.TW, ac = lq - ar * a.TW, ap = lq;     // A (dZ) iterator
    int br = ar, bc = ac, bp = lq;                          // B (X) iterator
    auto loadA = [&]() -> float {
      float v = 0.f;
      const int y = y0 + ar, x = x0 + ac;
      if (coOk && ap < KT && y < a.H && x < a.W) v = a.dz[((dzImg + y) * a.W + x) * (size_t)a.Cout + coA];
      ap += 4;
      ac += 4;
      while (ac >= a.TW) {
        ac -= a.TW;
        ++ar;
      }
      return v;
    };
    float aCur[WG_GROUP], aNxt[WG_GROUP];
#pragma unroll
    for (int s = 0; s < WG_GROUP; ++s) aCur[s] = loadA();
    for (int g0 = 0; g0 < kSteps; g0 += WG_GROUP) {
#pragma unroll
      for (int s = 0; s < WG_GROUP; ++s) aNxt[s] = loadA();   // past the tile: p >= KT -> 0, no load issued
#pragma unroll
      for (int s = 0; s < WG_GROUP; ++s) {
        const int step = g0 + s;
        if (step < kSteps) {   // uniform
          // pad lanes (p >= KT) carry A == 0; keep their B address inside the tile
          const int rr = bp < KT ? br : 0, cc = bp < KT ? bc : 0;
          const wf4* bBase = reinterpret_cast<const wf4*>(xs) + (rr * HW2 + cc) * (WG_XSTRIDE / 4) + li;
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const int ky = (TAPS == 9) ? t / 3 : 0, kx = (TAPS == 9) ? t % 3 : 0;
            const wf4 bq = bBase[(ky * HW2 + kx) * (WG_XSTRIDE / 4)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aCur[s], bq[j], acc[t][j], 0, 0, 0);
          }
          bp += 4;
          bc += 4;
          while (bc >= a.TW) {
            bc -= a.TW;
            ++br;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < WG_GROUP; ++s) aCur[s] = aNxt[s];
    }
  }

  // ---- partial tile -> slab[split][tap][co][ci] (rows = co: (lane>>4)*4 + r, cols = ci: lane&15) ----
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wave * 16 + lq * 4 + r;
        const int ci = ci0 + j * 16 + li;
        a.slab[(((size_t)split * TAPS + t) * a.CoPad + co) * (size_t)a.CiPad + ci] = acc[t][j][r];
      }
}

// dW (PyTorch layout) = sum over splits of the slab, in split order.
//   mode 0: conv (O,I,3,3):      out[(co*Cin + ci)*9 + t]
//   mode 1: upconv (I,O,2,2):    GEMM rows = (ab, co) of the space-to-depth gradient (row = ab*Cgrp + co), cols = ci:
//                                out[(ci*Cgrp + co)*4 + ab]
//   mode 2: conv for the first layer with padded input channels: same as mode 0 with CinReal columns
//   mode 3: first layer through the im2col GEMM (taps = 1, 28 columns of which 27 = (ci, tap) are real):
//                                out[row*27 + col], col < 27
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int taps,
                                                           int CoPad, int CiPad, int rowsReal, int colsReal, int mode,
                                                           int Cgrp, float* __restrict__ out,
                                                           const float* __restrict__ outScale = nullptr) {
  // 64 consecutive (tap, row, ci) elements per block (ci fastest, so slab reads are coalesced), 4 threads per
  // element each adding a quarter of the splits, four loads in flight, combined in a fixed order.
  __shared__ double red[4][64];
  const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
  const size_t total = (size_t)rowsReal * colsReal * taps;
  for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
    const size_t i = base + e;
    const bool ok = i < total;
    const int ci = ok ? (int)(i % colsReal) : 0;
    const size_t t2 = ok ? i / colsReal : 0;
    const int row = (int)(t2 % rowsReal);
    const int t = (int)(t2 / rowsReal);
    double s = 0.0;
    if (ok) {
      const int per = (splits + 3) / 4;
      const int s0 = part * per, s1 = (s0 + per < splits) ? s0 + per : splits;
      const size_t stride = (size_t)taps * CoPad * CiPad;
      const float* p = slab + ((size_t)t * CoPad + row) * (size_t)CiPad + ci;
      int sp = s0;
      for (; sp + 3 < s1; sp += 4)
        s += ((double)p[sp * stride] + (double)p[(sp + 1) * stride]) +
             ((double)p[(sp + 2) * stride] + (double)p[(sp + 3) * stride]);
      for (; sp < s1; ++sp) s += (double)p[sp * stride];
    }
    red[part][e] = s;
    __syncthreads();
    if (ok && part == 0) {
      float v = (float)((red[0][e] + red[1][e]) + (red[2][e] + red[3][e]));
      if (outScale) v *= *outScale;   // a power of two: undoes the scaling of a gradient operand (wgrad_x3_ws.h)
      if (mode == 1) {
        const int ab = row / Cgrp, co = row - ab * Cgrp;
        out[((size_t)ci * Cgrp + co) * 4 + ab] = v;
      } else if (mode == 3) {
        if (ci < 27) out[(size_t)row * 27 + ci] = v;
      } else {
        out[((size_t)row * colsReal + ci) * taps + t] = v;
      }
    }
    __syncthreads();
  }
}

}  // namespace unet
