// bf16 tier: 3x3 convolution + folded BatchNorm + ReLU for the 56 x 56 ... 14 x 14 levels on the structure of
// conv_x3_r512.h (read that header first) - one wave per SIMD with the whole 512-register file, weights straight from
// L2 into registers two taps ahead, only the input halo tile staged in LDS (by LDS-DMA issued from the four waves
// themselves), 224-pixel tiles of 8 x 28 / 16 x 14 whose 14 fragments of 16 pixels straddle tile rows, one s_barrier
// per 32-channel chunk - with ONE bf16 plane and ONE v_mfma_f32_16x16x32_bf16 per product:
//
//  * per tap a wave issues 14 fragments x 4 channel subtiles = 56 MFMAs (WPX = 1) against 14 ds_read_b128, 4 weight
//    loads and one LDS-DMA: three times the operand instructions per MFMA of the split-operand kernel, so the 63 LDS
//    read addresses of a chunk (7 fragment classes x 9 taps, swizzle applied) are computed once per kernel and kept
//    in registers - a read costs one v_add (the buffer's parity);
//  * same accumulation order as igemm_bf16.h / conv_bf16_ws.h (chunk by chunk, tap by tap), same packed weights as
//    conv_bf16_ws.h ([coTile(64)][chunk(32)][tap][cs][lane][8]), bf16 NHWC in and out: results bit-identical to
//    both (tests/test_bf16_gpu.py).
//
// Needs Cin % 32 == 0, Cout % (256 / WPX) == 0, W % TWX == 0.
#pragma once
#include "conv_bf16_ws.h"
#include "conv_x3_r512.h"

namespace unet {

struct ConvBfRArgs {
  const uint16_t* in;     // NHWC bf16, pixel stride Cin
  const uint16_t* wt;     // packed [coTile(64)][chunk(32)][tap(9)][cs(4)][lane(64)][8] (pack_fragments_ws)
  const uint16_t* zeros;  // >= 64 zero halfs
  const float* scale;
  const float* shift;
  uint16_t* out;          // NHWC bf16, pixel stride ldo, channel offset co_off
  int N, H, W, Cin, Cout, ldo, co_off, tilesX, tilesY, nChunks, relu;
  int coTiles, coGroup, pixTiles;   // coTiles: groups of 64 * (4 / WPX) output channels
  int imgH;                         // FLAT: the batch is one image of N * imgH rows (H = N * imgH, N = 1)
};

template <int TWX_>
struct BfRShape {
  using X = X3RShape<TWX_>;
  static constexpr int XPL = X::XPL;              // the one plane of a chunk's halo tile
  static constexpr int ZOFF = 2 * XPL;            // zero slot
  static constexpr int LDS_BYTES = ZOFF + 64;
};

__device__ __forceinline__ void mfma_bf16_acc(f32x4& c, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int TWX_, int WPX, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_bf16_r512_kernel(
    const ConvBfRArgs a) {
  using SX = X3RShape<TWX_>;
  using S = BfRShape<TWX_>;
  constexpr int TWX = SX::TWX, TH = SX::TH, P = SX::P, NQX = SX::NQX, NJ = SX::NJ;
  constexpr int WCO = 4 / WPX;
  constexpr int NF = SX::NPF / WPX;
  static_assert(SX::FP == 7, "tile widths 28 and 14");

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WCO, wc = wave - wp * WCO;
  const int li = lane & 15, lq = lane >> 4;
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;
  if (lb >= numWork) return;
  const unsigned ldsBase = lds_address(smemv);
  const char* lds = reinterpret_cast<const char*>(smemv);

  // ---- LDS-DMA: pieces q = wave + 4j of the plane ----
  int hrc[NJ];
  unsigned soff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;
    const int v = q * 64 + lane;
    const int qpix = v >> 2;
    const int part = (v & 3) ^ (((qpix >> 2) & 1) << 1);
    const int hr = qpix / P, hc = qpix - hr * P;
    hrc[j] = (hr << 8) | hc;
    soff[j] = (unsigned)(((hr * a.W + hc) * a.Cin + part * 8) * 2);
  }
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;

  struct Geo {
    const char* tb;
    int hrMin, hrSpan, hcMin, hcSpan;
    int n, y0, x0, cg;
  };
  auto geo_of = [&](int w) __attribute__((always_inline)) {
    Geo g;
    const int cInG = w % a.coGroup;
    const int rest = w / a.coGroup;
    const int tile = rest % a.pixTiles;
    g.cg = (rest / a.pixTiles) * a.coGroup + cInG;
    const int rowTile = tile / a.tilesX;
    g.x0 = (tile - rowTile * a.tilesX) * TWX;
    g.n = rowTile / a.tilesY;
    g.y0 = (rowTile - g.n * a.tilesY) * TH;
    const int hrMax = a.H - g.y0 < SX::HH2 - 1 ? a.H - g.y0 : SX::HH2 - 1;
    const int hcMax = a.W - g.x0 < SX::HW2 - 1 ? a.W - g.x0 : SX::HW2 - 1;
    g.hrMin = g.y0 == 0 ? 1 : 0;
    g.hcMin = g.x0 == 0 ? 1 : 0;
    g.hrSpan = hrMax - g.hrMin;
    g.hcSpan = hcMax - g.hcMin;
    g.tb = reinterpret_cast<const char*>(a.in) +
           ((((long)g.n * a.H + g.y0 - 1) * a.W + g.x0 - 1) * (long)a.Cin) * 2;
    return g;
  };
  auto issue_piece = [&](const Geo& g, int kc, int j, int buf) __attribute__((always_inline)) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;
    const int hr = hrc[j] >> 8, hc = hrc[j] & 255;
    const bool ok = (unsigned)(hr - g.hrMin) <= (unsigned)g.hrSpan && (unsigned)(hc - g.hcMin) <= (unsigned)g.hcSpan;
    const char* src = g.tb + soff[j] + (unsigned)(kc * 64);
    lds_dma16(ok ? src : zp, ldsBase + buf * S::XPL + q * 1024);
  };

  // ---- the 63 read addresses of a chunk in buffer 0 (class f % 7, tap t): kept, a read adds the buffer's offset ----
  int xa[7][9];
#pragma unroll
  for (int f7 = 0; f7 < 7; ++f7) {
    const int i = 16 * (NF * wp + f7) + li;
    const int r = i / TWX, c = i - r * TWX;
    const int xb = (r * P + c) * 64 + lq * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ky = t / 3, kx = t - ky * 3;
      const int b = xb + (ky * P + kx) * 64;
      xa[f7][t] = b ^ ((b >> 3) & 32);
      asm volatile("" : "+v"(xa[f7][t]));
    }
  }

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wt), 0, (a.Cout / 64) * a.nChunks * (9 * 4 * 1024), 0x00020000);
  const int laneW = lane * 16;
  auto w_block = [&](int cg, int kc) __attribute__((always_inline)) -> int {
    return ((cg * WCO + wc) * a.nChunks + kc) * (9 * 4 * 1024);
  };
  auto w_load = [&](int blk, int tap, int cs) __attribute__((always_inline)) -> f32x4 {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, laneW + cs * 1024, blk + tap * 4096, 0);
    return __builtin_bit_cast(f32x4, v);
  };

  if (tid < 4)
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smemv) + S::ZOFF + tid * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};

  Geo gCur = geo_of(lb);
  f32x4 wreg[3][4];   // ring over taps
  {
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue_piece(gCur, 0, j, 0);
    const int blk = w_block(gCur.cg, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) wreg[t][cs] = w_load(blk, t, cs);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int cc = 0;
  for (int w = lb; w < numWork; w += G) {
    const bool lastItem = w + G >= numWork;
    Geo gNext = gCur;
    if (!lastItem) gNext = geo_of(w + G);

    f32x4 acc[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[f][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cbase = (gCur.cg * WCO + wc) * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cbase + cs * 4);
      asm volatile("" : "+v"(sc[cs]), "+v"(sh[cs]));
    }

    unsigned keep = 0xFFFFFFFFu;
    if (FLAT) {
      const int y0m = gCur.y0 % a.imgH;
      keep = 0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int i = 16 * (NF * wp + f) + li;
        int yy = y0m + i / TWX;
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        keep |= (yy != 0 ? 1u : 0u) << f;
        keep |= (yy != a.imgH - 1 ? 1u : 0u) << (16 + f);
      }
    }

    for (int kc = 0; kc < a.nChunks; ++kc, ++cc) {
      const bool lastChunk = kc + 1 == a.nChunks;
      const bool haveNext = !(lastChunk && lastItem);
      const Geo& gIss = lastChunk ? gNext : gCur;
      const int kcIss = lastChunk ? (lastItem ? kc : 0) : kc + 1;
      const int wCur = w_block(gCur.cg, kc);
      const int wNxt = haveNext ? w_block(gIss.cg, kcIss) : wCur;
      int bufOff = (cc & 1) * S::XPL;
      asm volatile("" : "+v"(bufOff));   // a VGPR: the reads add it with one v_add each
      const int nbuf = (cc + 1) & 1;

      f32x4 xr[3];   // ring over (tap, fragment) in program order
      auto x_addr = [&](int t, int f) __attribute__((always_inline)) -> int {
        const int ky = t / 3;
        int addr = xa[f % 7][t] + bufOff + (f / 7) * SX::FRAGP;
        if (FLAT && ky != 1) {
          const bool kp = (keep >> ((ky == 0 ? 0 : 16) + f)) & 1u;
          addr = kp ? addr : S::ZOFF;
        }
        return addr;
      };
#pragma unroll
      for (int f = 0; f < 2; ++f) xr[f] = *reinterpret_cast<const f32x4*>(lds + x_addr(0, f));
#define R5_GAP __builtin_amdgcn_sched_barrier(0)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int L = t * NF + f;
          const bool pre = f + 2 < NF || t < 8;
          const int pt = f + 2 < NF ? t : t + 1, pf = f + 2 < NF ? f + 2 : f + 2 - NF, ps = (L + 2) % 3;
          int addr = 0;
          mfma_bf16_acc(acc[f][0], wreg[t % 3][0], xr[L % 3]);
          R5_GAP;
          if (pre) addr = x_addr(pt, pf);
          R5_GAP;
          mfma_bf16_acc(acc[f][1], wreg[t % 3][1], xr[L % 3]);
          R5_GAP;
          if (pre) xr[ps] = *reinterpret_cast<const f32x4*>(lds + addr);
          R5_GAP;
          mfma_bf16_acc(acc[f][2], wreg[t % 3][2], xr[L % 3]);
          R5_GAP;
          if (f < 4 / WPX) {   // the four weight fragments of the tap two ahead
            const int tt = t + 2;
#pragma unroll
            for (int i = f * WPX; i < (f + 1) * WPX; ++i) wreg[tt % 3][i] = w_load(tt < 9 ? wCur : wNxt, tt % 9, i);
          }
          if (f == NF - 1 && t < NJ) issue_piece(gIss, kcIss, t, nbuf);
          R5_GAP;
          mfma_bf16_acc(acc[f][3], wreg[t % 3][3], xr[L % 3]);
          R5_GAP;
        }
      }
#undef R5_GAP
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue: lane (li, lq) holds channels 16*lq + [0,16) of its pixel of each fragment ----
    const float floorV = a.relu ? 0.f : -3.4e38f;
    const size_t g0 = (size_t)gCur.n * a.H + gCur.y0;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int liE = li;
    asm volatile("" : "+v"(liE));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = 16 * (NF * wp + f) + liE;
      const int r = i / TWX, c = i - r * TWX;
      const bool ok = gCur.y0 + r < a.H;
      const size_t pix = (g0 + r) * a.W + gCur.x0 + c;
      uint32_t pk[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int e0 = 2 * e, e1 = 2 * e + 1;
        const float v0 = fmaxf(fmaf(acc[f][e0 >> 2][e0 & 3], sc[e0 >> 2][e0 & 3], sh[e0 >> 2][e0 & 3]), floorV);
        const float v1 = fmaxf(fmaf(acc[f][e1 >> 2][e1 & 3], sc[e1 >> 2][e1 & 3], sh[e1 >> 2][e1 & 3]), floorV);
        pk[e] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
      }
      uint16_t* rowp = a.out + pix * (size_t)a.ldo + a.co_off + cbase;
      if (ok) {
        uint4* o = reinterpret_cast<uint4*>(rowp);
        o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    gCur = gNext;
  }
}

}  // namespace unet
