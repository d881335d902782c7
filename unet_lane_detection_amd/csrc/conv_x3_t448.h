// Split-operand ("f16x3") 3x3 convolution, third structure: the layers with FEW output channels (Cout = 64 / 128: the
// 224 x 224 and 112 x 112 levels of the network, reference README.md:1449-1458 at features[0:2] and their decoder
// mirrors :1476-1479 - 43 % of the forward's flops).
//
// Same arithmetic, packed weights and accumulation order as conv_x3_ws.h / conv_x3_r512.h (chunk -> tap row -> tap ->
// (w_lo x_hi, w_hi x_lo, w_hi x_hi)), so all three structures give bit-identical results.  The second structure's
// recipe (one wave per SIMD, accumulators in AGPRs, weights straight from L2, only the input halo tile in LDS, one
// barrier per chunk, one instruction per MFMA gap) needs 256 (or 128) output channels per block to give every wave a
// 64-channel slice of its own.  With 64 or 128 channels the waves have to share the channels and split the PIXELS, so
// the block tile grows along the pixels instead:
//
//  * block tile = 16 rows x TWX columns (TWX = 28: 448 pixels; TWX = 32: 512 pixels, for maps whose width is not a
//    multiple of 28: the 640 x 640 configuration) x 64 * WCO output channels.  WCO = 1: the four waves own four rows each
//    and all 64 channels (7 / 8 pixel fragments x 4 channel subtiles = 112 / 128 AGPRs); WCO = 2: wave (wp, wc) owns eight
//    rows x channels [64 wc, 64 wc + 64) (14 / 16 fragments x 4 subtiles = 224 / 256 AGPRs - the per-wave shape of the
//    second structure's 256-channel form: 2 LDS reads and 2/3 of a weight load per 12 MFMAs);
//  * a pixel fragment is a 4 x 4 BLOCK of pixels, not 16 consecutive pixels of a row: lane li holds pixel
//    (li >> 2, li & 3) of the block.  Every lane addresses its own pixel anyway, and with this shape
//      - all fragments of a wave are a constant number of bytes apart, as are the nine taps: the LDS address of every
//        read of a chunk is ONE lane register plus an instruction immediate (the second structure recomputes 63
//        addresses per chunk, 4 VALU operations each: 252 of its ~2,000 non-MFMA instructions per chunk);
//      - the bank swizzle - the 16-byte part of a pixel's 64 bytes XOR-ed with 2 x (halo row & 1) - is a constant of the
//        lane per tap-row parity, and makes every ds_read_b128 conflict free for ANY even row pitch
//        (tools/lds_conflicts.py --blocks), so the pitch is TWX + 2 with no padding columns: both planes of a
//        32-channel chunk's halo tile are 68 KiB (TWX = 28), double buffered 136 KiB;
//      - the 2 x 2 max-pool windows lie inside a fragment: horizontal partner lane li ^ 1, vertical partner li ^ 4, both
//        DPP operands - the pooled epilogue needs no second fragment and no LDS;
//  * the plane stores write 64 contiguous bytes per pixel and instruction (conv_x3_r512.h).  Whole 128-byte lines - one
//    more exchange between the lanes of pixels li and li ^ 8 - were built and measured: the epilogue's 7.8k / 15.5k cycles
//    per item (7 / 14 fragments) did not move (profiles/r04/t448_experiments.md): a CU stores ~15 bytes per cycle whatever
//    the shape of the instruction;
//  * staging, weights, barrier and instruction placement as in the second structure: the halo tile of the next chunk by
//    LDS-DMA from the four waves themselves (9 / 10 piece indices per wave, spread over the first six taps), weights of
//    the tap two ahead by buffer loads
//    (every wave of a 64-channel group loads the group's 8 fragments: the block reads them WPXW times, from L1 / L2),
//    one s_barrier per chunk.
//
// EPI 0 stores the two planes, 1 the planes and their 2 x 2 max-pool, 2 runs the fused 1 x 1 head (Cout = 64; activation
// not stored), 3 stores fp32 (training; optional BatchNorm partial sums).  Needs Cin % 32 == 0, Cout % (64 WCO) == 0,
// W % TWX == 0; any H (rows past the bottom read the zero page and are not stored).
#pragma once
#include <type_traits>
#include <utility>
#include "conv_x3_r512.h"

namespace unet {

template <int TWX_, int RB_>
struct X3TShape {
  static constexpr int TWX = TWX_, RB = RB_, TH = 4 * RB_;   // 16 rows (64 / 128 channels per block), 8 rows (256)
  static constexpr int CB = TWX_ / 4;                        // 4 x 4 pixel blocks per tile row: 7 / 8
  static constexpr int P = TWX_ + 2;                         // LDS row pitch in pixels = the halo's width
  static constexpr int HH2 = TH + 2, HW2 = TWX_ + 2;
  static constexpr int NQX = (HH2 * P * 64 + 1023) / 1024;   // 1 KiB DMA pieces of one plane's halo tile: 34 / 39 (19 / 22)
  static constexpr int XPL = NQX * 1024;                     // one plane buffer
  static constexpr int XST = 2 * XPL;                        // hi + lo of one chunk
  static constexpr int NJ = (NQX + 3) / 4;                   // piece indices per wave: 9 / 10 (5 / 6)
  static constexpr int TOFF = 2 * XST;                       // the fused head's 64 weights (EPI 2; never FLAT)
  // FLAT: two zero regions, one per plane (XPL apart like the planes), each as long as the span of a row block's read
  // immediates at one tap row: (CB - 1) * 256 + 2 * 64 + 64 bytes
  static constexpr int ZOFF = 2 * XST, ZLEN = 2048;
  static constexpr int LDS_BYTES_PLAIN = TOFF + 256;
  static constexpr int LDS_BYTES_FLAT = ZOFF + XPL + ZLEN;
  static_assert(TWX_ == 28 || TWX_ == 32, "tile widths 28 and 32");
  static_assert(RB_ == 2 || RB_ == 4, "8- and 16-row tiles");
  static_assert((CB - 1) * 256 + 3 * 64 <= ZLEN, "");
  static_assert(LDS_BYTES_PLAIN <= 160 * 1024 && (RB_ == 4 || LDS_BYTES_FLAT <= 160 * 1024), "");
  // the largest read immediate: lo plane + last row block + last column block + tap (2, 2)
  static_assert(XPL + 4 * P * 64 + (CB - 1) * 256 + (2 * P + 2) * 64 < 65536, "ds_read offsets are 16 bits");
};

// the DMA piece index that goes out at fragment step L of a chunk (NF steps per tap), or -1: the NJ indices evenly
// spaced over the first `taps` taps
constexpr int x3t_piece_at(int L, int NF, int NJ, int taps) {
  for (int j = 0; j < NJ; ++j)
    if (L == ((j + 1) * taps * NF) / NJ - 1) return j;
  return -1;
}

template <int... Is, class F>
__device__ __forceinline__ void x3t_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// v (lane li) -> v (lane li + 4 of the same row of 16 lanes): the pixel one row down in a 4 x 4 block
__device__ __forceinline__ float dpp_rowshl4_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x104, 0xF, 0xF, true));
}

// rows blocks of the block tile by the waves' layout: 16 rows where the waves split the pixels, 8 where they do not
constexpr int x3t_row_blocks(int WCO) { return WCO == 4 ? 2 : 4; }

template <int TWX_, int WCO, int EPI, bool FLAT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_x3_t448_kernel(
    const ConvX3Args a) {
  using S = X3TShape<TWX_, x3t_row_blocks(WCO)>;
  constexpr int TWX = S::TWX, TH = S::TH, P = S::P, NQX = S::NQX, NJ = S::NJ, CB = S::CB;
  constexpr int WPXW = 4 / WCO;         // waves along the pixels: 4 / 2 / 1
  constexpr int RBW = S::RB / WPXW;     // row blocks (4 rows) per wave: 1 / 2 / 2
  constexpr int NF = CB * RBW;          // pixel fragments per wave: 7 / 14 / 14 (8 / 16 / 16)
  static_assert(WCO == 1 || ((WCO == 2 || WCO == 4) && NF <= 14),
                "waves along the channels: 14 fragments x 4 subtiles = 224 AGPRs is what fits");
  static_assert(EPI != 2 || WCO == 1, "the fused head reduces over the 64 channels of one wave");
  static_assert(!FLAT || (WCO == 4 && EPI != 2), "the tall-image tiling: 8-row tiles only");

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WCO, wc = wave - wp * WCO;
  const int li = lane & 15, lq = lane >> 4;
  const int G = gridDim.x;   // multiple of 8: consecutive logical blocks share an XCD (and its L2)
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;   // coTiles: groups of 64 * WCO output channels
  if (lb >= numWork) return;
  const unsigned ldsBase = lds_address(smemv);
  const char* lds = reinterpret_cast<const char*>(smemv);

  // ---- LDS-DMA: this wave issues pieces q = wave + 4j of both planes; tile independent per-lane parts ----
  int hrc[NJ];         // halo row << 8 | halo column of this lane's 16 bytes
  unsigned soff[NJ];   // byte offset of its source from the halo's top-left pixel (row y0 - 1, column x0 - 1), chunk 0
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;   // the last round only exists for some waves: duplicates rewrite the same bytes
    const int v = q * 64 + lane;
    const int qpix = v >> 2;
    const int hr = qpix / P, hc = qpix - hr * P;
    const int part = (v & 3) ^ ((hr & 1) << 1);
    hrc[j] = (hr << 8) | hc;
    soff[j] = (unsigned)(((hr * a.W + hc) * a.Cin + part * 8) * 2);
  }
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const size_t inLoB = a.inLo * 2;

  struct Geo {
    const char* tb;   // address of the halo's top-left pixel, chunk 0, hi plane (not dereferenced where out of image)
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
    const int hrMax = a.H - g.y0 < S::HH2 - 1 ? a.H - g.y0 : S::HH2 - 1;
    const int hcMax = a.W - g.x0 < S::HW2 - 1 ? a.W - g.x0 : S::HW2 - 1;
    g.hrMin = g.y0 == 0 ? 1 : 0;
    g.hcMin = g.x0 == 0 ? 1 : 0;
    g.hrSpan = hrMax - g.hrMin;
    g.hcSpan = hcMax - g.hcMin;
    g.tb = reinterpret_cast<const char*>(a.in) +
           ((((long)g.n * a.H + g.y0 - 1) * a.W + g.x0 - 1) * (long)a.Cin) * 2;
    return g;
  };
  // both planes of piece index j of (item geometry g, chunk kc) -> halo buffer `buf`
  auto issue_piece = [&](const Geo& g, int kc, int j, int buf) __attribute__((always_inline)) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;
    const int hr = hrc[j] >> 8, hc = hrc[j] & 255;
    const bool ok = (unsigned)(hr - g.hrMin) <= (unsigned)g.hrSpan && (unsigned)(hc - g.hcMin) <= (unsigned)g.hcSpan;
    const char* src = g.tb + soff[j] + (unsigned)(kc * 64);
    const unsigned dst = ldsBase + buf * S::XST + q * 1024;
    lds_dma16(ok ? src : zp, dst);
    lds_dma16(ok ? src + inLoB : zp, dst + S::XPL);
  };

  // ---- LDS read side: fragment f = row block f / CB, column block f % CB of the wave's rows; lane li holds pixel
  //      (li >> 2, li & 3) of the block.  xa[k]: byte address (buffer 0, hi plane) of the lane's 16 bytes of fragment 0
  //      at tap (0, 0) for a tap row of parity k: the swizzle depends on the halo row's parity only, and the wave's first
  //      row and the row blocks' offsets are even ----
  int xa[2];
  {
    const int pr = li >> 2, pc = li & 3;
    const int A0 = ((wp * RBW * 4 + pr) * P + pc) * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) xa[k] = A0 + ((lq ^ (((pr + k) & 1) << 1)) << 4);
  }

  // ---- weights: this wave's channel tile of 64; packed [coTile][chunk][tapRow][plane][kx][cs][lane][8 halfs] ----
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wt), 0, (a.Cout / 64) * a.chunksTotal * (9 * 2 * 4 * 1024), 0x00020000);
  const int laneW = lane * 16;
  auto w_block = [&](int cg, int kc) __attribute__((always_inline)) -> int {   // byte offset of (channel tile, chunk)
    const int ct = cg * WCO + wc;
    return (ct * a.chunksTotal + kc) * (9 * 2 * 4 * 1024);
  };
  auto w_load = [&](int blk, int tap, int plane, int cs) __attribute__((always_inline)) -> f32x4 {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int ky = tap / 3, kx = tap - ky * 3;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, laneW + cs * 1024,
                                                          blk + ((ky * 2 + plane) * 3 + kx) * 4096, 0);
    return __builtin_bit_cast(f32x4, v);
  };

  if (EPI == 2 && tid < 64) reinterpret_cast<float*>(reinterpret_cast<char*>(smemv) + S::TOFF)[tid] = a.headW[tid];
  if (FLAT) {   // the two zero regions (2 x 2 KiB = 256 x 16 bytes)
    const int z = tid >> 7, o = (tid & 127) * 16;
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smemv) + S::ZOFF + z * S::XPL + o) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // ---- prologue: chunk 0 of the first item, weights of its taps 0 and 1 ----
  Geo gCur = geo_of(lb);
  f32x4 wreg[3][2][4];   // ring over taps: tap t of a chunk sits in set t % 3
  {
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue_piece(gCur, 0, j, 0);
    const int blk = w_block(gCur.cg, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) wreg[t][p][cs] = w_load(blk, t, p, cs);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int cc = 0;   // chunks this block has gone through: halo buffer parity
  float amax = 0.f;   // largest |activation| this lane stored as planes (conv_x3_ws.h, range watch)
  float ssum[16], ssq[16];   // EPI 3 with statPartial (conv_x3_r512.h): a block keeps ONE channel group
#pragma unroll
  for (int e = 0; e < 16; ++e) ssum[e] = ssq[e] = 0.f;
  int statCbase = 0;
#if UNET_R512_STAMPS
  unsigned long long tLoop = 0, tBar = 0, tEpi = 0;
  const unsigned long long tStart = __builtin_amdgcn_s_memtime(), rStart = __builtin_amdgcn_s_memrealtime();
#endif
  for (int w = lb; w < numWork; w += G) {
    const bool lastItem = w + G >= numWork;
    Geo gNext = gCur;
    if (!lastItem) gNext = geo_of(w + G);

    f32x4 acc[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[f][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the epilogue's per-channel constants: fetched now, so that their latency is not the epilogue's
    const int cbase = (gCur.cg * WCO + wc) * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cbase + cs * 4);
      asm volatile("" : "+v"(sc[cs]), "+v"(sh[cs]));
    }

    // FLAT (the batch as one image of N * imgH rows): is this lane's pixel of row block rbw in the first / last row of
    // its image?  Then tap row 0 / 2 belongs to the neighbouring image and reads the zero region instead
    bool zTop[RBW], zBot[RBW];
    if (FLAT) {
      const int y0m = gCur.y0 % a.imgH;
#pragma unroll
      for (int rbw = 0; rbw < RBW; ++rbw) {
        int yy = y0m + (wp * RBW + rbw) * 4 + (li >> 2);
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        zTop[rbw] = yy == 0;
        zBot[rbw] = yy == a.imgH - 1;
      }
    }

    // ---- the epilogue's item constants.  The epilogue runs straight from the accumulators: lane (li, lq) holds channels
    //      16*lq + [0,16) of its pixel of each fragment: acc[f][cs][r] is channel 16*lq + 4*cs + r of the wave's tile ----
    if (a.dynScale) {
      const float ds = *a.dynScale;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) sc[cs] *= ds;
    }
    const float floorV = a.relu ? 0.f : -3.4e38f;
    const size_t g0 = (size_t)gCur.n * a.H + gCur.y0;   // global row of the tile's first row
    f32x4 hw[4];
    if (EPI == 2) {
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
        hw[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (lq * 16 + cs * 4) * 4);
    }
    auto epi_fragment = [&](int f) __attribute__((always_inline)) {
      // pixel offsets are recomputed per fragment from an opaque copy of the lane's index: hoisted out of the item loop
      // they are ~40 registers that get spilled (conv_x3_r512.h)
      int liE = li;
      asm volatile("" : "+v"(liE));
      const int prE = liE >> 2, pcE = liE & 3;
      const int r = wp * RBW * 4 + (f / CB) * 4 + prE, c = (f % CB) * 4 + pcE;
      const bool ok = gCur.y0 + r < a.H;
      const size_t pix = (g0 + r) * a.W + gCur.x0 + c;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        v[e] = fmaxf(fmaf(acc[f][e >> 2][e & 3], sc[e >> 2][e & 3], sh[e >> 2][e & 3]), floorV);
      if (EPI == 3) {
        if (a.statPartial) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float t = ok ? v[e] : 0.f;
            ssum[e] += t;
            ssq[e] = fmaf(t, t, ssq[e]);
          }
        }
        // 4 x 4 transpose of 16-byte pieces across the four lanes of a pixel: a store writes 64 contiguous bytes per pixel
        uint32_t u[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) u[e] = __builtin_bit_cast(uint32_t, v[e]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          auto r01 = __builtin_amdgcn_permlane16_swap(u[j], u[4 + j], false, false);
          auto r23 = __builtin_amdgcn_permlane16_swap(u[8 + j], u[12 + j], false, false);
          auto s02 = __builtin_amdgcn_permlane32_swap(r01[0], r23[0], false, false);
          auto s13 = __builtin_amdgcn_permlane32_swap(r01[1], r23[1], false, false);
          u[j] = s02[0];
          u[8 + j] = s02[1];
          u[4 + j] = s13[0];
          u[12 + j] = s13[1];
        }
        float* rowp = a.outF + pix * (size_t)a.ldo + a.co_off + (cbase - lq * 16) + lq * 4;
        if (ok) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4*>(rowp + 16 * k) = make_uint4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]);
        }
      } else {
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // not clamped: out-of-range values become inf and are reported (amax)
          amax3(amax, v[2 * e], v[2 * e + 1]);
          split_pk_f16_mix(v[2 * e], v[2 * e + 1], ph[e], pl[e]);
        }
        if (EPI == 2) {
          // fused 1x1 head (reference README.md:1447) on hi + lo, in conv_x3_ws.h's summation order
          float z = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float e0, e1;
            merge_pk_f16(ph[i], pl[i], e0, e1);
            z = fmaf(e0, hw[i >> 1][(2 * i) & 3], z);
            z = fmaf(e1, hw[i >> 1][(2 * i + 1) & 3], z);
          }
          z += __shfl_xor(z, 16, 64);
          z += __shfl_xor(z, 32, 64);
          z += a.headB;
          if (ok && lq == 0) {
            if (a.logits) a.logits[pix] = z;
            if (a.probs) a.probs[pix] = 1.f / (1.f + __expf(-z));
            if (a.mask) a.mask[pix] = z > a.headThr ? 255 : 0;
          }
        } else {
          if (EPI == 1) {
            // MaxPool2d(2,2) on the fp32 values (the split is monotonic): both partners are in this fragment.  Valid in
            // the lanes of even block row and even block column: li = 0, 2, 8, 10
            uint32_t qh[8], ql[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              float m0 = fmaxf(v[2 * i], dpp_xor1_f(v[2 * i])), m1 = fmaxf(v[2 * i + 1], dpp_xor1_f(v[2 * i + 1]));
              m0 = fmaxf(m0, dpp_rowshl4_f(m0));
              m1 = fmaxf(m1, dpp_rowshl4_f(m1));
              split_pk_f16_mix(m0, m1, qh[i], ql[i]);
            }
            const bool okp = gCur.y0 + r + 1 < a.H && (liE & 5) == 0;
            uint16_t* pp = a.pool + (((g0 + r) >> 1) * (size_t)(a.W >> 1) + ((gCur.x0 + c) >> 1)) * (size_t)a.Cout + cbase;
            if (okp) {
              uint4* o = reinterpret_cast<uint4*>(pp);
              o[0] = make_uint4(qh[0], qh[1], qh[2], qh[3]);
              o[1] = make_uint4(qh[4], qh[5], qh[6], qh[7]);
              uint4* ol = reinterpret_cast<uint4*>(pp + a.poolLo);
              ol[0] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
              ol[1] = make_uint4(ql[4], ql[5], ql[6], ql[7]);
            }
          }
          // two lane-row swaps per register hand lane row q bytes [16 q, 16 q + 16) of the first 64 bytes of the pixel's
          // 128 in one register set and of the second 64 in the other (conv_x3_r512.h)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            auto rr = __builtin_amdgcn_permlane16_swap(ph[k], ph[4 + k], false, false);
            auto q = __builtin_amdgcn_permlane32_swap(rr[0], rr[1], false, false);
            ph[k] = q[0];
            ph[4 + k] = q[1];
            auto rl = __builtin_amdgcn_permlane16_swap(pl[k], pl[4 + k], false, false);
            auto ql2 = __builtin_amdgcn_permlane32_swap(rl[0], rl[1], false, false);
            pl[k] = ql2[0];
            pl[4 + k] = ql2[1];
          }
          uint16_t* rowp = a.out + pix * (size_t)a.ldo + a.co_off + (cbase - lq * 16) + lq * 8;
          if (ok) {
            if (WCO == 1) {
              // 64-channel layers (the 224 x 224 level): non-temporal stores.  With plain stores the output lines push the
              // input's out of the XCD's 4 MiB L2 between the two 32-channel chunks that share each 128-byte line, and
              // every input line is fetched twice: 2 x FETCH_SIZE 10.4 -> 8.5 GB per launch of the 128 -> 64 layer (7.9
              // with no re-fetch at all), 4.12 -> 4.09 ms; 64 -> 64 pooled 2.56 -> 2.52 ms.  Not on the 128- and
              // 256-channel forms: no gain at 112 x 112, 0.9 % slower at 56 x 56 (profiles/r04/t448_experiments.md)
              typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
              __builtin_nontemporal_store((u32x4nt){ph[0], ph[1], ph[2], ph[3]}, reinterpret_cast<u32x4nt*>(rowp));
              __builtin_nontemporal_store((u32x4nt){ph[4], ph[5], ph[6], ph[7]}, reinterpret_cast<u32x4nt*>(rowp + 32));
              __builtin_nontemporal_store((u32x4nt){pl[0], pl[1], pl[2], pl[3]}, reinterpret_cast<u32x4nt*>(rowp + a.outLo));
              __builtin_nontemporal_store((u32x4nt){pl[4], pl[5], pl[6], pl[7]},
                                          reinterpret_cast<u32x4nt*>(rowp + a.outLo + 32));
            } else {
              *reinterpret_cast<uint4*>(rowp) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
              *reinterpret_cast<uint4*>(rowp + 32) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
              *reinterpret_cast<uint4*>(rowp + a.outLo) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
              *reinterpret_cast<uint4*>(rowp + a.outLo + 32) = make_uint4(pl[4], pl[5], pl[6], pl[7]);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one fragment at a time: 16 values live
    };

    for (int kc = 0; kc < a.nChunks; ++kc, ++cc) {
      const bool lastChunk = kc + 1 == a.nChunks;
      const bool haveNext = !(lastChunk && lastItem);
      // (the block's very last chunk re-stages itself into the idle buffer: no branch in the unrolled body)
      const Geo& gIss = lastChunk ? gNext : gCur;
      const int kcIss = lastChunk ? (lastItem ? kc : 0) : kc + 1;
      const int wCur = w_block(gCur.cg, kc);
      const int wNxt = haveNext ? w_block(gIss.cg, kcIss) : wCur;
      const int bufOff = (cc & 1) * S::XST;
      const int nbuf = (cc + 1) & 1;

      int xc[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xc[k] = xa[k] + bufOff;
        asm volatile("" : "+v"(xc[k]));
      }
      // FLAT: the base of a row block's reads at tap rows 0 and 2 - the tile, or the zero region minus the part of the
      // immediate that is not the column block and the tap column (4 selects per chunk instead of one per read)
      int xTop[RBW], xBot[RBW];
      if (FLAT) {
#pragma unroll
        for (int rbw = 0; rbw < RBW; ++rbw) {
          xTop[rbw] = zTop[rbw] ? S::ZOFF + (lane >> 4) * 16 - rbw * (4 * P * 64) : xc[0];
          xBot[rbw] = zBot[rbw] ? S::ZOFF + (lane >> 4) * 16 - rbw * (4 * P * 64) - 2 * P * 64 : xc[0];
          asm volatile("" : "+v"(xTop[rbw]), "+v"(xBot[rbw]));
        }
      }
      R5_STAMP(tC0);
      f32x4 xh[3], xl[3];   // ring over (tap, fragment) in program order
      // hi-plane read of fragment f at tap t: one of two lane registers + an immediate
      auto x_read = [&](int t, int f, int plane) __attribute__((always_inline)) -> f32x4 {
        const int ky = t / 3, kx = t - ky * 3;
        const int base = (FLAT && ky == 0) ? xTop[f / CB] : (FLAT && ky == 2) ? xBot[f / CB] : xc[ky & 1];
        return *reinterpret_cast<const f32x4*>(lds + base + plane * S::XPL + (f / CB) * (4 * P * 64) + (f % CB) * 256 +
                                               (ky * P + kx) * 64);
      };
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        xh[f] = x_read(0, f, 0);
        xl[f] = x_read(0, f, 1);
      }
      // The 9 * NF (tap, fragment) steps of a chunk, tap by tap (compile-time steps).
      // Tried and not kept (profiles/r04/t448_experiments.md): ending a chunk FRAGMENT by fragment over taps 7 and 8, so that
      // in an item's last chunk fragment f - 1 is converted and stored between the MFMAs of fragments f and f + 1 and the
      // stores drain under MFMAs.  Bit-identical, no spills (as ONE body for every chunk with the epilogue blocks behind a
      // uniform branch; a second body for the last chunk cost 451 v_accvgpr_mov and 66 - 340 spilled registers) - and
      // 0.98 - 1.02 x the time: these kernels are not waiting for their stores.
      constexpr int STEPS = 9 * NF;
#define T4_GAP __builtin_amdgcn_sched_barrier(0)
      x3t_static_for(std::make_integer_sequence<int, STEPS>{}, [&](auto Lc) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int t = L / NF, f = L % NF;
        // Under this fragment's 12 MFMAs (three per channel subtile, small terms first) everything else goes out one
        // instruction per MFMA gap: the operands of the step two ahead (not across the chunk's end: the other buffer is
        // published by the barrier), the weight fragments of the tap two ahead, the DMA piece indices.
        constexpr int L2 = L + 2;
        constexpr bool pre = !(UNET_R512_ABLATE & 4) && L2 < STEPS;
        constexpr int pt = L2 / NF, pf = L2 % NF, ps = L2 % 3;
        auto M = [&](int m) __attribute__((always_inline)) {
          const int cs = m / 3, k = m - cs * 3;
          mfma_x3_acc(acc[f][cs], wreg[t % 3][k == 0 ? 1 : 0][cs], k == 1 ? xl[L % 3] : xh[L % 3]);
        };
        // the eight weight fragments of the tap two ahead: one per fragment step (NF >= 8), or two (NF = 7)
        auto W = [&](int i) __attribute__((always_inline)) {
          if (i < 8 && !(UNET_R512_ABLATE & 2)) {
            const int tt = t + 2;
            wreg[tt % 3][i >> 2][i & 3] = w_load(tt < 9 ? wCur : wNxt, tt % 9, i >> 2, i & 3);
          }
        };
        M(0);
        T4_GAP;
        M(1);
        T4_GAP;
        if (pre) xh[ps] = x_read(pt, pf, 0);
        T4_GAP;
        M(2);
        M(3);
        T4_GAP;
        if (pre) xl[ps] = x_read(pt, pf, 1);
        T4_GAP;
        M(4);
        M(5);
        T4_GAP;
        W(NF >= 8 ? f : 2 * f);
        T4_GAP;
        M(6);
        M(7);
        T4_GAP;
        if (NF < 8) W(2 * f + 1);
        T4_GAP;
        M(8);
        M(9);
        T4_GAP;
        // the NJ piece indices of the next chunk's halo, evenly spaced over the first six taps: the last three taps are for
        // the last pieces to land (one piece per tap with the ninth at the very end: 670 cycles of wait at the barrier)
        if (!(UNET_R512_ABLATE & 1)) {
          constexpr int jp = x3t_piece_at(L, NF, NJ, 6);
          if (jp >= 0) issue_piece(gIss, kcIss, jp, nbuf);
        }
        T4_GAP;
        M(10);
        M(11);
        T4_GAP;
      });
#undef T4_GAP
      // this wave's pieces of the next chunk have landed, its reads of this chunk are done
      R5_ACCUM(tLoop, tC0);
      R5_STAMP(tB0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      R5_ACCUM(tBar, tB0);
    }
    R5_STAMP(tE0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the first accumulator read
#pragma unroll
    for (int f = 0; f < NF; ++f) epi_fragment(f);
    statCbase = cbase;
    gCur = gNext;
    R5_ACCUM(tEpi, tE0);
  }
  if (EPI != 3) x3_report_range(amax, a.err);
  if (EPI == 3 && a.statPartial) {   // sum over the 16 pixels of a fragment (lanes li), then one lane per 16 channels
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        ssum[e] += __shfl_xor(ssum[e], m, 64);
        ssq[e] += __shfl_xor(ssq[e], m, 64);
      }
    }
    if (li == 0) {
      float* row = a.statPartial + (size_t)(blockIdx.x * WPXW + wp) * 2 * a.Cout + statCbase;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4*>(row + 4 * q) = (f32x4){ssum[4 * q], ssum[4 * q + 1], ssum[4 * q + 2], ssum[4 * q + 3]};
        *reinterpret_cast<f32x4*>(row + a.Cout + 4 * q) = (f32x4){ssq[4 * q], ssq[4 * q + 1], ssq[4 * q + 2], ssq[4 * q + 3]};
      }
    }
  }
#if UNET_R512_STAMPS
  if (tid == 0) {
    unsigned long long* st = reinterpret_cast<unsigned long long*>(a.logits) + (size_t)blockIdx.x * 8;
    st[0] = tLoop;
    st[1] = tBar;
    st[2] = tEpi;
    st[3] = __builtin_amdgcn_s_memtime() - tStart;
    st[4] = __builtin_amdgcn_s_memrealtime() - rStart;
    st[5] = (unsigned long long)cc;
  }
#endif
}

}  // namespace unet
