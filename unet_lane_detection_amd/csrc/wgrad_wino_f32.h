// Weight gradient of the 3x3 convolutions by Winograd F(3x3, 2x2) on the fp32 matrix pipe (training backward,
// SURVEY.md section 8 rows a1/a5; the reference computes it inside loss.backward(), README.md:2076).
//
//   dW[co][ci][ky][kx] = sum over images and pixels of dZ[y][x][co] * X[y+ky-1][x+kx-1][ci]
//
// Per 2x2 output tile this is a 2-D correlation of the tile's 4x4 input patch d with its 2x2 gradient patch g
// giving a 3x3 result - the transpose of the forward F(2x2,3x3) algorithm, with the same interpolation points:
//
//   dW = G^T [ sum over tiles (A g A^T) (.) (B^T d B) ] G          (.) = element-wise, 16 products per tile
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   A = [1 0; 1 1; 1 -1; 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//
// 16 multiplies instead of 36 per (tile, co, ci): 2.25x fewer MFMA cycles than wgrad_f32.h.  The sum over tiles
// commutes with the outer transform, so the kernel accumulates the 16 "point" matrices M_p[co][ci] and a second
// kernel adds the split slabs and applies G once.
//
// GEMM shape per point: M = 64 co, N = 64 ci per block, K = tiles (v_mfma_f32_16x16x4_f32: 4 tiles per MFMA).
// Block = 512 threads = 8 waves = 4 (co quarters) x 2 (ci halves); a wave holds 16 points x 2 ci subtiles = 32
// accumulators (128 VGPRs).  Lane (i, q) of a K-step owns tile q of four adjacent tiles, output channel 16*wc + i
// (A operand) and input channels 4*i + 2*wn + {0,1} (B operand: 8 contiguous bytes per patch pixel) and does
// both transforms in registers.  Tile groups (GH x GW tiles) are staged in LDS by LDS-DMA, double-buffered:
// [halo pixel][64 ci] and [pixel][64 co], natural channel order (ds_read_b64 / b32 run 2-way bank-conflicted:
// LDS is ~30 % busy).  Out-of-tensor pixels come from a zero page; rows that belong to a neighbouring image are
// zeroed per K-step (all four tiles of a K-step are in one tile row, so that is a uniform decision).
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"
#include <stdint.h>

namespace unet {

typedef float wwf4 __attribute__((ext_vector_type(4)));
typedef float wwf2 __attribute__((ext_vector_type(2)));

struct WgradWinoArgs {
  const float* dz;     // (N,H,W,Cout) dense
  const float* x;      // (N,H,W,*) pixel stride ldx, channels [0,Cin)
  const float* zeros;  // >= 64 zero floats
  float* slab;         // [splits][16][Cout][Cin]
  int N, H, W;
  int Cout, Cin, ldx;  // Cout, Cin multiples of 64
  int groupsX;         // ceil((W/2) / GW)
  int nGroups;         // groupsX * ceil((N*H/2) / GH)
  int groupsPerSplit;
};

template <int GH, int GW>
struct WgradWinoShape {
  static constexpr int RH = 2 * GH + 2, RW = 2 * GW + 2;   // halo of the input patch grid
  static constexpr int DH = 2 * GH, DW = 2 * GW;
  static constexpr int XPIX = RH * RW, DPIX = DH * DW;
  static constexpr int XBYTES = XPIX * 256, DBYTES = DPIX * 256;
  static constexpr int STAGE = XBYTES + DBYTES;
  static constexpr int NQX = XPIX / 4, NQD = DPIX / 4;     // 1 KiB DMA pieces (4 pixels x 64 channels)
  static constexpr int LDS_BYTES = 2 * STAGE;
  static_assert(XPIX % 4 == 0 && DPIX % 4 == 0, "whole DMA pieces");
  static_assert(GH * GW == 32, "8 K-steps of 4 tiles per stage");
};

template <int GH, int GW>
__global__ __launch_bounds__(512, 1) void wgrad_wino_f32_kernel(const WgradWinoArgs a) {
  using S = WgradWinoShape<GH, GW>;
  constexpr int RW = S::RW, DW = S::DW;
  constexpr int NQ = S::NQX + S::NQD;        // 77
  constexpr int NQW = (NQ + 7) / 8;          // pieces per wave per stage (10)

  extern __shared__ __attribute__((aligned(16))) char wwsmem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int wc = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x;
  const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const int NH = a.N * a.H;
  const int gBegin = split * a.groupsPerSplit;
  const int gEnd = gBegin + a.groupsPerSplit < a.nGroups ? gBegin + a.groupsPerSplit : a.nGroups;
  const int nMine = gEnd > gBegin ? gEnd - gBegin : 0;

  wwf4 acc[16][2];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) acc[p][ns] = (wwf4){0.f, 0.f, 0.f, 0.f};

  // per-lane byte offsets inside a stage buffer: tile column q of a K-step is 2q pixels to the right
  const int xlane = (2 * lq) * 256 + li * 16 + wn * 8;
  const int dlane = S::XBYTES + (2 * lq) * 256 + (16 * wc + li) * 4;

  // Iteration `it` issues the DMA of group `it`, multiplies group `it - 1`, then waits for the DMA and joins
  // the barrier.  One issue site and one compute site: nothing here may end up in scratch.
  for (int it = 0; it <= nMine; ++it) {
    if (it < nMine) {
      const int g = gBegin + it;
      const int tr0 = (g / a.groupsX) * GH, tc0 = (g % a.groupsX) * GW;
      char* dst = wwsmem + (it & 1) * S::STAGE;
#pragma unroll
      for (int j = 0; j < NQW; ++j) {
        const int q = wave + 8 * j;
        if (q < NQ) {   // uniform
          const float* src;
          if (q < S::NQX) {
            const int pix = q * 4 + lq;
            const int hr = pix / RW, hc = pix - hr * RW;
            const int R = 2 * tr0 - 1 + hr, C = 2 * tc0 - 1 + hc;
            const bool ok = R >= 0 && R < NH && C >= 0 && C < a.W;
            src = ok ? a.x + ((size_t)R * a.W + C) * (size_t)a.ldx + ci0 + li * 4 : a.zeros + li * 4;
          } else {
            const int pix = (q - S::NQX) * 4 + lq;
            const int dr = pix / DW, dc = pix - dr * DW;
            const int R = 2 * tr0 + dr, C = 2 * tc0 + dc;
            const bool ok = R < NH && C < a.W;
            src = ok ? a.dz + ((size_t)R * a.W + C) * (size_t)a.Cout + co0 + li * 4 : a.zeros + li * 4;
          }
          lds_dma16(src, lds_address(dst) + q * 1024);
        }
      }
    }
    if (it >= 1) {
      const int g = gBegin + it - 1;
      const int tr0 = (g / a.groupsX) * GH;
      const char* buf = wwsmem + ((it - 1) & 1) * S::STAGE;
      const char* xb = buf + xlane;
      const char* db = buf + dlane;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        // K-step s: four adjacent tiles of one tile row
        const int trl = (GW == 8) ? (s >> 1) : s;
        const int tcu = (GW == 8) ? 4 * (s & 1) : 0;
        const int y0 = (2 * (tr0 + trl)) % a.H;   // uniform: image row of the tiles' first pixel row
        const bool topOut = y0 == 0, botOut = y0 == a.H - 2;
        // ---- gradient patch -> V = A g A^T (16 values of the MFMA A operand) ----
        float gq[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            gq[i][j] = *reinterpret_cast<const float*>(db + ((2 * trl + i) * DW + 2 * tcu + j) * 256);
        float sv[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sv[0][j] = gq[0][j];
          sv[1][j] = gq[0][j] + gq[1][j];
          sv[2][j] = gq[0][j] - gq[1][j];
          sv[3][j] = -gq[1][j];
        }
        float V[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          V[p][0] = sv[p][0];
          V[p][1] = sv[p][0] + sv[p][1];
          V[p][2] = sv[p][0] - sv[p][1];
          V[p][3] = -sv[p][1];
        }
        // ---- input patch (two channels per lane) -> U = B^T d B ----
        wwf2 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            d[r][c] = *reinterpret_cast<const wwf2*>(xb + ((2 * trl + r) * RW + 2 * tcu + c) * 256);
        if (topOut) {
#pragma unroll
          for (int c = 0; c < 4; ++c) d[0][c] = (wwf2){0.f, 0.f};
        }
        if (botOut) {
#pragma unroll
          for (int c = 0; c < 4; ++c) d[3][c] = (wwf2){0.f, 0.f};
        }
        wwf2 t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          t[0][c] = d[0][c] - d[2][c];
          t[1][c] = d[1][c] + d[2][c];
          t[2][c] = d[2][c] - d[1][c];
          t[3][c] = d[1][c] - d[3][c];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          wwf2 U[4];
          U[0] = t[p][0] - t[p][2];
          U[1] = t[p][1] + t[p][2];
          U[2] = t[p][2] - t[p][1];
          U[3] = t[p][1] - t[p][3];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ns = 0; ns < 2; ++ns)
              acc[p * 4 + q][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[p][q], U[q][ns], acc[p * 4 + q][ns], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- accumulators -> slab[split][point][co][ci]: rows = co0 + 16*wc + 4*lq + r, cols = ci0 + 4*li + 2*wn + ns.
  //      One base pointer per accumulator row; the 16 points are a uniform stride apart. ----
  float* sl = a.slab + (size_t)split * 16 * a.Cout * a.Cin;
  const size_t pointStride = (size_t)a.Cout * (size_t)a.Cin;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = co0 + 16 * wc + 4 * lq + r;
    float* rowp = sl + (size_t)co * (size_t)a.Cin + ci0 + 4 * li + 2 * wn;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      wwf2 v = {acc[p][0][r], acc[p][1][r]};
      *reinterpret_cast<wwf2*>(rowp + p * pointStride) = v;
    }
  }
}

// dW[co][ci][ky][kx] = sum_{p,q} G[p][ky] G[q][kx] M[p][q][co][ci], M = the split slabs added in split order.
// 64 consecutive ci per block, 4 threads per (co, ci) each adding a quarter of the splits (doubles), combined in
// a fixed order: deterministic.
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ slab, int splits, int Cout,
                                                                int Cin, float* __restrict__ out) {
  __shared__ double red[3][64][16];
  const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
  const size_t pairs = (size_t)Cout * Cin;
  const size_t i = (size_t)blockIdx.x * 64 + e;   // Cin % 64 == 0: always in range
  const size_t stride = 16 * pairs;
  double m[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) m[p] = 0.0;
  const int per = (splits + 3) / 4;
  const int s0 = part * per, s1 = s0 + per < splits ? s0 + per : splits;
  for (int sp = s0; sp < s1; ++sp) {
    const float* ptr = slab + (size_t)sp * stride + i;
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] += (double)ptr[(size_t)p * pairs];
  }
  if (part > 0) {
#pragma unroll
    for (int p = 0; p < 16; ++p) red[part - 1][e][p] = m[p];
  }
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] = (m[p] + red[0][e][p]) + (red[1][e][p] + red[2][e][p]);
    // rows: G^T M (3x4), then columns
    double h[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h[0][q] = m[0 * 4 + q] + 0.5 * (m[1 * 4 + q] + m[2 * 4 + q]);
      h[1][q] = 0.5 * (m[1 * 4 + q] - m[2 * 4 + q]);
      h[2][q] = 0.5 * (m[1 * 4 + q] + m[2 * 4 + q]) + m[3 * 4 + q];
    }
    float* o = out + i * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      o[ky * 3 + 0] = (float)(h[ky][0] + 0.5 * (h[ky][1] + h[ky][2]));
      o[ky * 3 + 1] = (float)(0.5 * (h[ky][1] - h[ky][2]));
      o[ky * 3 + 2] = (float)(0.5 * (h[ky][1] + h[ky][2]) + h[ky][3]);
    }
  }
}

}  // namespace unet
