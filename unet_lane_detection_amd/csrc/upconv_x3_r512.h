// ConvTranspose2d(kernel 2, stride 2) / plain GEMM of the split-operand (f16x3) tier on the one-wave-per-SIMD structure.
//
// Same arithmetic, operands, packed weights and accumulation order as upconv_x3_ws.h (read that header first; the two
// kernels give bit-identical results), arranged as conv_x3_r512.h arranges the 3x3 convolution.  upconv_x3_ws.h stages
// 48 KiB (16 KiB of pixels, 32 KiB of weights) per 4 x 96 MFMAs and is bound by that staging (MFMA pipe busy 0.35,
// profiles/r03/r03z_summary.md); here
//  * a block is 4 waves with up to 512 registers, no loader waves; a work item is 224 consecutive input pixels
//    (flattened n, y, x) x one tile of 64 output channels x all four (a,b): wave w owns (a,b) = w - a column group of 64 -
//    and all 14 pixel fragments: 56 accumulator tiles in AGPRs;
//  * a weight fragment is needed by ONE wave and goes straight from L2 into registers (16 fragments per 64-channel
//    stage and wave, fetched one stage ahead);
//  * only the pixels are staged in LDS (LDS-DMA issued by the four waves): 64 channels x 224 pixels x 2 planes = 56 KiB
//    per stage, double buffered; one s_barrier per stage of 336 MFMAs per wave; every LDS read address is one lane
//    constant plus an immediate;
//  * per stage and CU 56 KiB of pixels + 64 KiB of weights for 4 x 336 MFMAs: 0.37 of upconv_x3_ws.h's bytes per MFMA.
// MODE 0: y[2i+a][2j+b][co] = sum_ci x[i][j][ci] W[ci][co][a][b] + bias[co], both planes.  MODE 1: plain GEMM with fp32
// output, a channel tile = 256 columns = four groups of 64 (UpconvX3Args).  Needs Cin % 128 == 0 (two stages per
// loop trip) and, MODE 0, w >= 4.
#pragma once
#include "conv_q8_r512.h"
#include "upconv_x3_ws.h"

namespace unet {

struct UpconvX3RShape {
  static constexpr int TP = 224, NPF = 14;
  static constexpr int XPL = TP * 64;            // one plane of one 32-channel sub-chunk: 14 pieces of 1 KiB
  static constexpr int XST = 4 * XPL;            // [sub-chunk][plane]
  static constexpr int LDS_BYTES = 2 * XST;      // 114,688
};

// OUTQ (MODE 0, f16q8 tier): the output's q plane is written in its lo plane's place (conv_q8_r512.h: the one consumer
// of the upper channel half of the concat buffer is a convolution of that tier)
template <int MODE, bool OUTQ = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void upconv2x2_x3_r512_kernel(
    const UpconvX3Args a) {
  using S = UpconvX3RShape;
  constexpr int NF = S::NPF;
  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;   // consecutive items: the channel tiles of one pixel tile
  if (lb >= numWork) return;
  const unsigned ldsBase = lds_address(smemv);
  const char* lds = reinterpret_cast<const char*>(smemv);
  const int nStages = a.nChunks >> 1;           // 64 channels per stage

  // ---- LDS-DMA: wave k stages plane k & 1 of sub-chunk k >> 1: 14 pieces of 16 pixels; a lane's 16 bytes: pixel
  //      q * 16 + lane / 4, part (lane & 3) ^ 2 * bit 2 of the pixel = a lane constant ----
  const int dPix = lane >> 2;
  const int dPart = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const char* srcLane = reinterpret_cast<const char*>(a.in) + ((size_t)dPix * (size_t)a.Cin + (wave >> 1) * 32 + dPart * 8) * 2 +
                        ((wave & 1) ? a.inLo * 2 : (size_t)0);
  const size_t pieceStep = (size_t)a.Cin * 32;   // 16 pixels further on, bytes
  const unsigned dstWave = ldsBase + wave * S::XPL;
  auto issue_piece = [&](long p0, int kc64, int q, int buf) __attribute__((always_inline)) {
    const bool ok = p0 + q * 16 + dPix < a.npix;
    const char* src = srcLane + (size_t)p0 * (size_t)a.Cin * 2 + (size_t)q * pieceStep + kc64 * 128;
    lds_dma16(ok ? src : zp, dstWave + buf * S::XST + q * 1024);
  };

  // ---- LDS read side: this lane's 16 bytes of pixel li of fragment 0 (+ f * 1024 + (sub-chunk * 2 + plane) * XPL) ----
  const int xa0 = li * 64 + ((lq ^ (((li >> 2) & 1) << 1)) << 4);

  // ---- weights: packed [coTile][chunk(32)][plane(2)][ab(4)][cs(4)][lane][8 halfs]; wave w reads group (a,b) = w ----
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wt), 0, a.coTiles * a.nChunks * (2 * 16 * 1024), 0x00020000);
  const int laneW = lane * 16 + wave * 4096;
  auto w_load = [&](int coTile, int kc32, int plane, int cs) __attribute__((always_inline)) -> f32x4 {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, laneW + plane * 16384 + cs * 1024,
                                                          (coTile * a.nChunks + kc32) * (2 * 16 * 1024), 0);
    return __builtin_bit_cast(f32x4, v);
  };

  // ---- prologue: stage 0 of the first item ----
  int tileCur = lb / a.coTiles, ctCur = lb - tileCur * a.coTiles;
  f32x4 wreg[2][2][2][4];   // [ring][sub-chunk][plane][cs]
  {
#pragma unroll
    for (int j = 0; j < 14; ++j) issue_piece((long)tileCur * S::TP, 0, j, 0);
#pragma unroll
    for (int sc = 0; sc < 2; ++sc)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) wreg[0][sc][p][cs] = w_load(ctCur, sc, p, cs);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const float dyn = (MODE == 1 && a.dynScale) ? *a.dynScale : 1.f;
  int cc = 0;   // stages this block has gone through: buffer and ring parity
  float amax = 0.f;
  for (int w = lb; w < numWork; w += G) {
    const bool lastItem = w + G >= numWork;
    int tileNext = tileCur, ctNext = ctCur;
    if (!lastItem) {
      tileNext = (w + G) / a.coTiles;
      ctNext = (w + G) - tileNext * a.coTiles;
    }
    // MODE 1: column groups of this tile that exist (wave-uniform; the packed weights of the others are zero, their
    // products are not stored)
    const int nAb = MODE == 0 ? 4 : ((a.Cout - ctCur * 256) >= 256 ? 4 : (a.Cout - ctCur * 256) / 64);
    const bool active = wave < nAb;

    f32x4 acc[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[f][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // two stages per trip so that buffer and ring parities are compile-time constants (nStages is even: Cin % 128 == 0)
    for (int ks = 0; ks < nStages; ks += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par, ++cc) {
        const int kcur = ks + par;
        const bool lastStage = kcur + 1 == nStages;
        const bool haveNext = !(lastStage && lastItem);
        const long p0Iss = (long)(lastStage ? tileNext : tileCur) * S::TP;
        const int kIss = lastStage ? (lastItem ? kcur : 0) : kcur + 1;
        const int ctIss = lastStage ? ctNext : ctCur;
        const int xbuf = par * S::XST;   // this stage's buffer; the next stage goes to the other one
        const int nbuf = par ^ 1;
        f32x4 xh[3], xl[3];
#pragma unroll
        for (int L = 0; L < 2; ++L) {
          const int sc = L / NF, f = L - sc * NF;
          xh[L] = *reinterpret_cast<const f32x4*>(lds + xa0 + xbuf + (sc * 2) * S::XPL + f * 1024);
          xl[L] = *reinterpret_cast<const f32x4*>(lds + xa0 + xbuf + (sc * 2 + 1) * S::XPL + f * 1024);
        }
#define UR_GAP __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int L = 0; L < 2 * NF; ++L) {
          const int sc = L / NF, f = L - sc * NF;
          const bool pre = L + 2 < 2 * NF;
          const int pL = L + 2, psc = pL / NF, pf = pL - psc * NF, ps = pL % 3;
          auto M = [&](int m) __attribute__((always_inline)) {
            const int cs = m / 3, k = m - cs * 3;
            mfma_x3_acc(acc[f][cs], wreg[par][sc][k == 0 ? 1 : 0][cs], k == 1 ? xl[L % 3] : xh[L % 3]);
          };
          M(0);
          UR_GAP;
          if (pre) xh[ps] = *reinterpret_cast<const f32x4*>(lds + xa0 + xbuf + (psc * 2) * S::XPL + pf * 1024);
          UR_GAP;
          M(1);
          M(2);
          UR_GAP;
          if (pre) xl[ps] = *reinterpret_cast<const f32x4*>(lds + xa0 + xbuf + (psc * 2 + 1) * S::XPL + pf * 1024);
          UR_GAP;
          M(3);
          M(4);
          UR_GAP;
          // the next stage's operands: its 14 DMA pieces in the first 14 steps, its 16 weight fragments behind them
          if (L < 14) issue_piece(p0Iss, kIss, L, nbuf);
          UR_GAP;
          M(5);
          M(6);
          UR_GAP;
          if (L >= 6 && L < 22 && haveNext) {
            const int i = L - 6;   // (sub-chunk, plane, cs)
            wreg[par ^ 1][i >> 3][(i >> 2) & 1][i & 3] = w_load(ctIss, kIss * 2 + (i >> 3), (i >> 2) & 1, i & 3);
          }
          UR_GAP;
          M(7);
          M(8);
          M(9);
          M(10);
          M(11);
          UR_GAP;
        }
#undef UR_GAP
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }

    // ---- epilogue: lane (li, lq) holds columns 16 * lq + [0, 16) of its group for pixel li of each fragment ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int liE = li;
    asm volatile("" : "+v"(liE));
    const long pBase = (long)tileCur * S::TP + liE;
    if (MODE == 1) {
      if (active) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const long p = pBase + 16 * f;
          if (p < a.npix) {
            float* orow = a.outF + (size_t)p * (size_t)a.ldo + a.co_off + ctCur * 256 + wave * 64 + lq * 16;
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
              f32x4 v = acc[f][cs];
              v *= dyn;
              *reinterpret_cast<f32x4*>(orow + cs * 4) = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      const int cbase = ctCur * 64 + lq * 16;
      f32x4 sc4[4], bi[4];
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
        sc4[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
        bi[cs] = *reinterpret_cast<const f32x4*>(a.bias + cbase + cs * 4);
      }
      // (row, x) of the lane's pixel of fragment 0; the fragments' pixels follow 16 apart
      long pc = pBase < a.npix ? pBase : 0;
      long row = pc / a.w;   // n * h + y
      int x = (int)(pc - row * a.w);
      const int oa = wave >> 1, ob = wave & 1;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const bool ok = pBase + 16 * f < a.npix;
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cs = e >> 1, r = (2 * e) & 3;
          const float v0 = fmaf(acc[f][cs][r], sc4[cs][r], bi[cs][r]), v1 = fmaf(acc[f][cs][r + 1], sc4[cs][r + 1], bi[cs][r + 1]);
          amax3(amax, v0, v1);
          split_pk_f16_mix(v0, v1, ph[e], pl[e]);
        }
        uint32_t qh[4], ql[4];
        if (OUTQ) {   // the q plane's bytes as planes_to_q8_kernel makes them from (hi, lo)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          const float hs = __builtin_ldexpf(1.f, kQ8HiShift), ls = __builtin_ldexpf(1.f, kQ8LoShift);
          float hf[16], lf[16];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const f32x2 h2 = __builtin_convertvector(__builtin_bit_cast(f16x2, ph[e]), f32x2);
            const f32x2 l2 = __builtin_convertvector(__builtin_bit_cast(f16x2, pl[e]), f32x2);
            hf[2 * e] = h2[0] * hs;
            hf[2 * e + 1] = h2[1] * hs;
            lf[2 * e] = l2[0] * ls;
            lf[2 * e + 1] = l2[1] * ls;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            qh[e] = q8_pack4(hf[4 * e], hf[4 * e + 1], hf[4 * e + 2], hf[4 * e + 3]);
            ql[e] = q8_pack4(lf[4 * e], lf[4 * e + 1], lf[4 * e + 2], lf[4 * e + 3]);
          }
        }
        // 64 contiguous bytes per pixel and store instruction (conv_x3_r512.h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          auto r1 = __builtin_amdgcn_permlane16_swap(ph[k], ph[4 + k], false, false);
          auto q1 = __builtin_amdgcn_permlane32_swap(r1[0], r1[1], false, false);
          ph[k] = q1[0];
          ph[4 + k] = q1[1];
          if (!OUTQ) {
            auto rl = __builtin_amdgcn_permlane16_swap(pl[k], pl[4 + k], false, false);
            auto q2 = __builtin_amdgcn_permlane32_swap(rl[0], rl[1], false, false);
            pl[k] = q2[0];
            pl[4 + k] = q2[1];
          } else {   // block 0 of the wave's 64 channels complete in qh, block 1 in ql (conv_q8_r512.h, EPI 4)
            auto sw = __builtin_amdgcn_permlane32_swap(qh[k], ql[k], false, false);
            qh[k] = sw[0];
            ql[k] = sw[1];
          }
        }
        uint16_t* op = a.out + (((size_t)(2 * row + oa) * (size_t)(2 * a.w)) + 2 * x + ob) * (size_t)a.ldo + a.co_off +
                       ctCur * 64 + lq * 8;
        if (ok) {
          *reinterpret_cast<uint4*>(op) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          *reinterpret_cast<uint4*>(op + 32) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
          if (!OUTQ) {
            *reinterpret_cast<uint4*>(op + a.outLo) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            *reinterpret_cast<uint4*>(op + a.outLo + 32) = make_uint4(pl[4], pl[5], pl[6], pl[7]);
          } else {
            *reinterpret_cast<uint4*>(op + a.outLo) = make_uint4(qh[0], qh[1], qh[2], qh[3]);
            *reinterpret_cast<uint4*>(op + a.outLo + 32) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
          }
        }
        x += 16;
#pragma unroll
        for (int k = 0; k < 4; ++k)   // 16 pixels cross up to four row ends (w >= 4; the host checks)
          if (x >= a.w) {
            x -= a.w;
            ++row;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tileCur = tileNext;
    ctCur = ctNext;
  }
  if (MODE == 0) x3_report_range(amax, a.err);
}

}  // namespace unet
