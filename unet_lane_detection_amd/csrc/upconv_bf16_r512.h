// ConvTranspose2d(kernel 2, stride 2) of the bf16 tier on the one-wave-per-SIMD structure (upconv_x3_r512.h with one
// bf16 plane and one MFMA per product).  Same operands, packed weights and accumulation order as upconv_bf16_ws.h (read
// that header first; bit-identical results).  upconv_bf16_ws.h stages 48 KiB (16 KiB of pixels, 32 KiB of weights) per
// 4 x 64 MFMAs and is bound by that staging (MFMA pipe busy 0.23, 1.6 - 3 TB/s of stores: profiles/r03/r03z_summary.md);
// here
//  * a block is 4 waves with up to 512 registers; a work item is 224 consecutive input pixels x one 64-channel tile x
//    all four (a,b); wave w owns (a,b) = w and all 14 pixel fragments: 56 accumulator tiles;
//  * a weight fragment is needed by ONE wave and goes straight from L2 into registers (8 fragments per 64-channel stage
//    and wave, one stage ahead); only the pixels are staged (28 KiB per stage, double buffered, LDS-DMA issued by the
//    four waves); one s_barrier per stage of 112 MFMAs per wave; every LDS read address is a lane constant plus an
//    immediate;
//  * per stage and CU 28 KiB of pixels + 32 KiB of weights for 4 x 112 MFMAs: 0.71 of upconv_bf16_ws.h's bytes per MFMA;
//  * epilogue: 64 contiguous bytes per pixel and store instruction (two lane-row swaps per register, conv_x3_r512.h).
// Needs Cin % 128 == 0 (two stages per loop trip) and w >= 4.
#pragma once
#include "conv_bf16_r512.h"
#include "upconv_bf16_ws.h"

namespace unet {

struct UpconvBfRShape {
  static constexpr int TP = 224, NPF = 14;
  static constexpr int XST = TP * 128;           // one stage: 64 channels x 224 pixels, 28 pieces of 1 KiB
  static constexpr int LDS_BYTES = 2 * XST;      // 57,344
  static constexpr int NJ = 7;                   // pieces per wave and stage
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void upconv2x2_bf16_r512_kernel(
    const UpconvWsArgs a) {
  using S = UpconvBfRShape;
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
  const int nStages = a.nChunks;                // 64 channels per stage

  // ---- LDS-DMA: wave k issues pieces q = k + 4 j (8 pixels x 128 bytes each); a lane's 16 bytes: pixel q * 8 + lane / 8,
  //      part (lane & 7) ^ ((pixel >> 1) & 7), and (pixel >> 1) & 7 = 4 (q & 1) + lane / 16 with q & 1 = k & 1 ----
  const int dPix = lane >> 3;
  const int dPart = (lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7);
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 7) * 16;
  const char* srcLane = reinterpret_cast<const char*>(a.in) + ((size_t)(wave * 8 + dPix) * (size_t)a.Cin + dPart * 8) * 2;
  const size_t pieceStep = (size_t)a.Cin * 64;   // 4 pieces = 32 pixels further on, bytes
  const unsigned dstWave = ldsBase + wave * 1024;
  auto issue_piece = [&](long p0, int kc, int j, int buf) __attribute__((always_inline)) {
    const bool ok = p0 + (wave + 4 * j) * 8 + dPix < a.npix;
    const char* src = srcLane + (size_t)p0 * (size_t)a.Cin * 2 + (size_t)j * pieceStep + kc * 128;
    lds_dma16(ok ? src : zp, dstWave + buf * S::XST + j * 4096);
  };

  // ---- LDS read side: this lane's 16 bytes of pixel li of fragment 0 at k-step j (+ f * 2048) ----
  int xa[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) xa[j] = li * 128 + (((j * 4 + lq) ^ ((li >> 1) & 7)) << 4);

  // ---- weights: packed [coTile][chunk(64)][kstep(2)][ab(4)][cs(4)][lane][8]; wave w reads (a,b) = w ----
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wt), 0, a.coTiles * a.nChunks * (32 * 1024), 0x00020000);
  const int laneW = lane * 16 + wave * 4096;
  auto w_load = [&](int coTile, int kc, int ks, int cs) __attribute__((always_inline)) -> f32x4 {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, laneW + ks * 16384 + cs * 1024,
                                                          (coTile * a.nChunks + kc) * (32 * 1024), 0);
    return __builtin_bit_cast(f32x4, v);
  };

  // ---- prologue: stage 0 of the first item ----
  int tileCur = lb / a.coTiles, ctCur = lb - tileCur * a.coTiles;
  f32x4 wreg[2][2][4];   // [ring][k-step][cs]
  {
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) issue_piece((long)tileCur * S::TP, 0, j, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) wreg[0][ks][cs] = w_load(ctCur, 0, ks, cs);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  for (int w = lb; w < numWork; w += G) {
    const bool lastItem = w + G >= numWork;
    int tileNext = tileCur, ctNext = ctCur;
    if (!lastItem) {
      tileNext = (w + G) / a.coTiles;
      ctNext = (w + G) - tileNext * a.coTiles;
    }
    f32x4 acc[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[f][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // two stages per trip: buffer and ring parities are compile-time constants (nStages is even)
    for (int ks2 = 0; ks2 < nStages; ks2 += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int kcur = ks2 + par;
        const bool lastStage = kcur + 1 == nStages;
        const bool haveNext = !(lastStage && lastItem);
        const long p0Iss = (long)(lastStage ? tileNext : tileCur) * S::TP;
        const int kIss = lastStage ? (lastItem ? kcur : 0) : kcur + 1;
        const int ctIss = lastStage ? ctNext : ctCur;
        const int xbuf = par * S::XST;
        const int nbuf = par ^ 1;
        f32x4 xf[3];
#pragma unroll
        for (int L = 0; L < 2; ++L) xf[L] = *reinterpret_cast<const f32x4*>(lds + xa[L / NF] + xbuf + (L % NF) * 2048);
#define UB_GAP __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int L = 0; L < 2 * NF; ++L) {   // k-step major: all fragments of k-step 0, then of k-step 1
          const int ks = L / NF, f = L - ks * NF;
          const bool pre = L + 2 < 2 * NF;
          const int pL = L + 2, pks = pL / NF, pf = pL - pks * NF;
          mfma_bf16_acc(acc[f][0], wreg[par][ks][0], xf[L % 3]);
          UB_GAP;
          if (pre) xf[pL % 3] = *reinterpret_cast<const f32x4*>(lds + xa[pks] + xbuf + pf * 2048);
          UB_GAP;
          mfma_bf16_acc(acc[f][1], wreg[par][ks][1], xf[L % 3]);
          UB_GAP;
          // the next stage's operands: its 7 DMA pieces first, its 8 weight fragments behind them
          if (L < S::NJ) issue_piece(p0Iss, kIss, L, nbuf);
          if (L >= 8 && L < 16 && haveNext) {
            const int i = L - 8;
            wreg[par ^ 1][i >> 2][i & 3] = w_load(ctIss, kIss, i >> 2, i & 3);
          }
          UB_GAP;
          mfma_bf16_acc(acc[f][2], wreg[par][ks][2], xf[L % 3]);
          mfma_bf16_acc(acc[f][3], wreg[par][ks][3], xf[L % 3]);
          UB_GAP;
        }
#undef UB_GAP
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }

    // ---- epilogue: lane (li, lq) holds channels 64 * coTile + 16 * lq + [0, 16) of pixel li of each fragment ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int liE = li;
    asm volatile("" : "+v"(liE));
    const long pBase = (long)tileCur * S::TP + liE;
    const int cbase = ctCur * 64 + lq * 16;
    f32x4 bi[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) bi[cs] = *reinterpret_cast<const f32x4*>(a.bias + cbase + cs * 4);
    long pc = pBase < a.npix ? pBase : 0;
    long row = pc / a.w;   // n * h + y
    int x = (int)(pc - row * a.w);
    const int oa = wave >> 1, ob = wave & 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const bool ok = pBase + 16 * f < a.npix;
      uint32_t pk[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cs = e >> 1, r = (2 * e) & 3;
        pk[e] = pk_bf16(acc[f][cs][r] + bi[cs][r], acc[f][cs][r + 1] + bi[cs][r + 1]);
      }
      // the four lanes of a pixel hold 32 bytes each as two 16-byte halves; after two lane-row swaps per register lane row
      // q holds bytes [16 q, +16) of the pixel's first 64 bytes in pk[0..3] and of its second 64 in pk[4..7]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        auto r1 = __builtin_amdgcn_permlane16_swap(pk[k], pk[4 + k], false, false);
        auto q1 = __builtin_amdgcn_permlane32_swap(r1[0], r1[1], false, false);
        pk[k] = q1[0];
        pk[4 + k] = q1[1];
      }
      uint16_t* op = a.out + (((size_t)(2 * row + oa) * (size_t)(2 * a.w)) + 2 * x + ob) * (size_t)a.ldo + a.co_off +
                     ctCur * 64 + lq * 8;
      if (ok) {
        *reinterpret_cast<uint4*>(op) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(op + 32) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
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
    tileCur = tileNext;
    ctCur = ctNext;
  }
}

}  // namespace unet
