// Split-operand ("f16x3") 3x3 convolution, second structure: one wave per SIMD with the whole 512-register file.
//
// Same arithmetic as conv_x3_ws.h (read that header first: hi/lo fp16 planes, three v_mfma_f32_16x16x32_f16 per
// product, fp32 accumulate, the same accumulation order chunk -> tap row -> tap -> (w_lo x_hi, w_hi x_lo, w_hi x_hi), so
// the two kernels give bit-identical results), the same packed weights and the same swizzled LDS image of the input
// halo tile.  What differs is who moves the operands (profiles/r02/x3_kernel_experiments.md: in the first structure the
// weight path through LDS costs 20 % of the run time, the LDS reads and the L2 -> LDS staging 12 % each):
//
//  * a block is 4 waves, one per SIMD, up to 512 registers each (accumulators in AGPRs); there are no loader waves;
//  * block tile = 224 pixels x 256 output channels (WPX = 1: wave w owns ALL 14 pixel fragments x channels
//    [64w, 64w + 64) = 56 accumulator tiles = 224 AGPRs) or 224 pixels x 128 channels (WPX = 2: wave (wp, wc) owns 7
//    fragments x 64 channels);
//  * a weight fragment is needed by ONE wave (WPX = 1), so weights never touch LDS: each wave loads the 8 fragments
//    (4 channel subtiles x hi, lo) of a tap straight from L2 into registers, two taps ahead, in a ring of three register
//    sets - 8 global_load_dwordx4 per 168 MFMAs;
//  * the input halo tile of a 32-channel chunk (both planes) is the only thing staged in LDS, double buffered per
//    chunk, by LDS-DMA issued from the four waves themselves: 12 - 14 pieces of 1 KiB per wave and chunk, one piece
//    index per tap, against 1512 MFMAs; one s_barrier per chunk;
//  * every wave reads every pixel fragment: 2 ds_read_b128 (hi, lo) per 12 MFMAs - half the LDS reads per MFMA of
//    the first structure, all of the saving being the weight reads;
//  * pixel tiles are 8 rows x 28 columns (maps whose width is a multiple of 28: 224, 112, 56, 28) or 16 rows x 14
//    columns (the 14 x 14 bottleneck): 224 pixels = 14 fragments of 16 consecutive pixels in row-major tile order, so
//    a fragment may straddle two tile rows and no MFMA is spent on padding columns (the first structure pads 28 and 56
//    to 32 and 64, 14 to 16).  Each lane computes the LDS address of its own pixel, so the tap shift is one add and the
//    bank swizzle one XOR per read; with LDS pitches of 36 and 22 pixels every ds_read_b128 is conflict free
//    (tools/lds_conflicts.py).  Fragments f and f + 7 are a whole number of tile rows apart (an immediate offset);
//  * FLAT instances (maps whose height is not a multiple of the tile height: 28 and 14) tile the batch as one image of
//    N * imgH rows; a tap that would cross an image boundary reads a zero slot in LDS instead (an address select per
//    read instead of masking the operands).
//
// EPI 0 stores the two planes, EPI 3 fp32 (training).  The 2x2 max-pool and 1x1 head fusions stay with the first
// structure (conv_x3_ws.h); the host falls back to it (or to the separate pooling kernel) for those layers.
//
// Tile widths 32, 16 and 8 (7 x 32, 14 x 16, 28 x 8 pixels) serve maps whose width is not a multiple of 28: the 640 x 640
// configuration's 320-, 160-, 80- and 40-wide levels; the 7 x 32 tile also in the two-wave form (Cout = 128 at 320 x 320).
//
// Needs Cin % 32 == 0, Cout % (256 / WPX) == 0, W % TWX == 0.
#pragma once
#include "conv_x3_ws.h"
#include "lds_dma.h"

// Diagnostic build (-DUNET_R512_STAMPS=1, tools/probes only): s_memtime sums of wave 0 per block - chunk loops, barrier
// waits, epilogues, whole kernel, and the kernel in 100 MHz ticks - written to ConvX3Args::logits (unused by this
// kernel) as 8 x uint64 per block.  Never on in the shipped library; no output depends on a stamp.
#ifndef UNET_R512_STAMPS
#define UNET_R512_STAMPS 0
#endif
#if UNET_R512_STAMPS
#define R5_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define R5_ACCUM(acc, t0) acc += __builtin_amdgcn_s_memtime() - (t0)
#else
#define R5_STAMP(var)
#define R5_ACCUM(acc, t0)
#endif

// Timing-only builds (wrong results; tools/probes): bit 0 = no LDS-DMA in the chunk loop, bit 1 = no weight loads in
// the chunk loop, bit 2 = no LDS reads in the chunk loop (each leaves the instruction stream otherwise unchanged)
#ifndef UNET_R512_ABLATE
#define UNET_R512_ABLATE 0
#endif
// 1 = the epilogue transposes 16-byte halves between the four lanes of a pixel so that a store writes 64 contiguous bytes
#ifndef UNET_R512_STORE64
#define UNET_R512_STORE64 1
#endif

namespace unet {

// acc += A * B with acc in the accumulator file.  Operands come from loads only (the compiler places their
// s_waitcnt in front of the statement); nothing reads acc but the next MFMA on it until the epilogue's pad.
__device__ __forceinline__ void mfma_x3_acc(f32x4& c, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// split_pk_f16 without the clamp (out-of-range values become inf; the caller watches the range), in 4 instructions per
// pair: hi = v_cvt_pk_f16_f32, lo = rn16(v - hi) as one mixed-precision FMA per value that reads hi as fp16 and writes
// its fp16 result into one half of the destination (v - hi is exact in fp32, so the single rounding is the same as in
// split_pk_f16: the two kernel structures stay bit-identical, tests/test_x3_gpu.py)
__device__ __forceinline__ void split_pk_f16_mix(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v0, v1}, f16x2));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo)
      : "v"(hi), "v"(v0), "v"(v1));
}
// amax = max(amax, |v0|, |v1|) in one instruction
__device__ __forceinline__ void amax3(float& amax, float v0, float v1) {
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v0), "v"(v1));
}

template <int TWX_>
struct X3RShape {
  static constexpr int TWX = TWX_;
  static constexpr int TH = 224 / TWX_;                      // 8 / 16 (widths 28k, 14); 7 / 14 / 28 (widths 32k, 16k, 8k)
  static constexpr int NPF = 14;                             // pixel fragments of a block tile
  // LDS row pitch in pixels: the smallest >= TWX + 2 with every ds_read_b128 conflict free (tools/lds_conflicts.py)
  static constexpr int P = TWX_ == 28 ? 36 : TWX_ == 14 ? 22 : TWX_ == 32 ? 36 : TWX_ == 16 ? 20 : 16;
  static constexpr int HH2 = TH + 2, HW2 = TWX_ + 2;
  static constexpr int NQX = (HH2 * P * 64 + 1023) / 1024;   // 1 KiB DMA pieces of one plane's halo tile: 23 / 25 / 21 / 20 / 30
  static constexpr int XPL = NQX * 1024;                     // one plane buffer
  static constexpr int XST = 2 * XPL;                        // hi + lo of one chunk
  static constexpr int NJ = (NQX + 3) / 4;                   // piece indices per wave: 6 / 7 / 6 / 5 / 8
  // The fragments' LDS positions repeat every FP fragments, FRAGP bytes further on - a whole number of tile rows and a
  // multiple of 512 bytes, so that the bank swizzle keeps its phase and fragment f is fragment f % FP plus an immediate
  static constexpr int FP = (TWX_ == 28 || TWX_ == 14) ? 7 : TWX_ == 32 ? 4 : TWX_ == 16 ? 2 : 1;
  static constexpr int FRAGP = (FP * 16 / TWX_) * P * 64;    // 9216 / 11264 / 4608 / 2560 / 2048
  static constexpr int ZOFF = 2 * XST;                       // zero slot (and a second one XPL behind it)
  static constexpr int LDS_BYTES = ZOFF + XPL + 64;
  static_assert(TWX_ == 28 || TWX_ == 14 || TWX_ == 32 || TWX_ == 16 || TWX_ == 8, "tile widths 28, 14, 32, 16, 8");
  static_assert((FP * 16) % TWX_ == 0, "a period is a whole number of tile rows");
  static_assert(NJ <= 9, "one piece index per tap");
  static_assert(FRAGP % 512 == 0, "fragment f + FP must keep the bank swizzle phase");
  static_assert(LDS_BYTES <= 160 * 1024, "");
};

// EPI: 0 = store the activation planes, 3 = store fp32
template <int TWX_, int WPX, int EPI, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_x3_r512_kernel(
    const ConvX3Args a) {
  using S = X3RShape<TWX_>;
  constexpr int TWX = S::TWX, TH = S::TH, P = S::P, NQX = S::NQX, NJ = S::NJ;
  constexpr int WCO = 4 / WPX;          // waves along the output channels
  constexpr int NF = S::NPF / WPX;      // pixel fragments per wave: 14 / 7
  constexpr int FP = S::FP;
  static_assert(WPX == 1 || (WPX == 2 && (FP == 7 || TWX_ == 32)),
                "the two-wave split: widths 28k and 14, and the 7 x 32 tile (the 320-wide level of the 640 x 640 configuration)");

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
    const int part = (v & 3) ^ (((qpix >> 2) & 1) << 1);
    const int hr = qpix / P, hc = qpix - hr * P;
    hrc[j] = (hr << 8) | hc;
    soff[j] = (unsigned)(((hr * a.W + hc) * a.Cin + part * 8) * 2);
  }
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const size_t inLoB = a.inLo * 2;

  // geometry of a work item: uniform values only
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

  // ---- LDS read side: byte position (before the tap shift and the swizzle) of this lane's 16 bytes of fragment
  //      7 * (f / 7) + f7: pixel i = 16 * (first fragment of the wave + f7) + li of the tile in row-major order ----
  int xb[FP];
#pragma unroll
  for (int f7 = 0; f7 < FP; ++f7) {
    const int i = 16 * (NF * wp + f7) + li;
    const int r = i / TWX, c = i - r * TWX;
    xb[f7] = (r * P + c) * 64 + lq * 16;
  }

  // ---- weights: this wave's channel tile of 64; packed [coTile][chunk][tapRow][plane][kx][cs][lane][8 halfs].
  //      Buffer loads: descriptor and block offset in SGPRs, the lane's 16 bytes as the only vector operand ----
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

  if (tid < 8)   // the two zero slots
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smemv) + S::ZOFF + (tid >> 2) * S::XPL + (tid & 3) * 16) =
        (f32x4){0.f, 0.f, 0.f, 0.f};

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
  // EPI 3 with statPartial: this lane's running sum / sum of squares of its 16 channels.  A block keeps ONE channel
  // group (items lb, lb + G, ... with G a multiple of 8 and the groups' count 1, 2 or 4), so the sums stay per channel.
  float ssum[16], ssq[16];
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
    // the epilogue's per-channel constants: fetched now, so that their latency is not the epilogue's first 2000 cycles
    const int cbase = (gCur.cg * WCO + wc) * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cbase + cs * 4);
      asm volatile("" : "+v"(sc[cs]), "+v"(sh[cs]));
    }

    // FLAT: bit f = "fragment f's pixel of this lane is not in the first row of an image" (tap row 0 is real),
    //       bit 16 + f = "... not in the last row" (tap row 2 is real)
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
      // the chunk that follows this one in the block's stream: its halo tile is staged and its first two taps'
      // weights are fetched while this one is multiplied
      const bool haveNext = !(lastChunk && lastItem);
      // (the block's very last chunk re-stages itself into the idle buffer: no branch in the unrolled body)
      const Geo& gIss = lastChunk ? gNext : gCur;
      const int kcIss = lastChunk ? (lastItem ? kc : 0) : kc + 1;
      const int wCur = w_block(gCur.cg, kc);
      const int wNxt = haveNext ? w_block(gIss.cg, kcIss) : wCur;
      const int bufOff = (cc & 1) * S::XST;
      const int nbuf = (cc + 1) & 1;

      // the 63 read addresses of a chunk are recomputed where they are used (4 VALU operations each): hoisted out of
      // the chunk loop they would occupy 126 registers
      int xc[FP];
#pragma unroll
      for (int f7 = 0; f7 < FP; ++f7) {
        xc[f7] = xb[f7] + bufOff;
        asm volatile("" : "+v"(xc[f7]));
      }
      R5_STAMP(tC0);
      f32x4 xh[3], xl[3];   // ring over (tap, fragment) in program order
      // LDS address of this lane's 16 bytes of fragment f at tap t (hi plane; lo plane XPL behind it)
      auto x_addr = [&](int t, int f) __attribute__((always_inline)) -> int {
        const int ky = t / 3, kx = t - ky * 3;
        const int b = xc[f % FP] + (ky * P + kx) * 64;
        int addr = (b ^ ((b >> 3) & 32)) + (f / FP) * S::FRAGP;
        if (FLAT && ky != 1) {
          const bool kp = (keep >> ((ky == 0 ? 0 : 16) + f)) & 1u;
          addr = kp ? addr : S::ZOFF;
        }
        return addr;
      };
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int addr = x_addr(0, f);
        xh[f] = *reinterpret_cast<const f32x4*>(lds + addr);
        xl[f] = *reinterpret_cast<const f32x4*>(lds + addr + S::XPL);
      }
#define R5_GAP __builtin_amdgcn_sched_barrier(0)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int L = t * NF + f;
          // Under this fragment's 12 MFMAs (three per channel subtile, small terms first: w_lo x_hi, w_hi x_lo,
          // w_hi x_hi) everything else goes out ONE instruction per MFMA gap - a cluster in front of the MFMAs
          // cost 25 cycles per fragment (profiles/r03/r512_experiments.md): the operands of the fragment two ahead
          // in program order (not across the chunk's end: the other buffer is published by the barrier), a weight
          // fragment of the tap two ahead, one DMA piece index per tap.
          const bool pre = !(UNET_R512_ABLATE & 4) && (f + 2 < NF || t < 8);
          const int pt = f + 2 < NF ? t : t + 1, pf = f + 2 < NF ? f + 2 : f + 2 - NF, ps = (L + 2) % 3;
          const int pky = pt / 3, pkx = pt - pky * 3;
          int b = 0, sw = 0, addr = 0;
          auto M = [&](int m) __attribute__((always_inline)) {
            const int cs = m / 3, k = m - cs * 3;
            mfma_x3_acc(acc[f][cs], wreg[t % 3][k == 0 ? 1 : 0][cs], k == 1 ? xl[L % 3] : xh[L % 3]);
          };
          M(0);
          R5_GAP;
          if (pre) b = xc[pf % FP] + (pky * P + pkx) * 64;
          R5_GAP;
          M(1);
          R5_GAP;
          if (pre) sw = b >> 3;
          R5_GAP;
          M(2);
          R5_GAP;
          if (pre) sw &= 32;
          R5_GAP;
          M(3);
          R5_GAP;
          if (pre) addr = (b ^ sw) + (pf / FP) * S::FRAGP;
          R5_GAP;
          M(4);
          if (FLAT && pre && pky != 1) {
            R5_GAP;
            const bool kp = (keep >> ((pky == 0 ? 0 : 16) + pf)) & 1u;
            addr = kp ? addr : S::ZOFF;
          }
          R5_GAP;
          M(5);
          R5_GAP;
          if (pre) xh[ps] = *reinterpret_cast<const f32x4*>(lds + addr);
          R5_GAP;
          M(6);
          M(7);
          R5_GAP;
          if (pre) xl[ps] = *reinterpret_cast<const f32x4*>(lds + addr + S::XPL);
          R5_GAP;
          M(8);
          R5_GAP;
          if (f < 8 / WPX && !(UNET_R512_ABLATE & 2)) {   // the eight weight fragments of the tap two ahead
            const int tt = t + 2, i = f * WPX;
            wreg[tt % 3][i >> 2][i & 3] = w_load(tt < 9 ? wCur : wNxt, tt % 9, i >> 2, i & 3);
          }
          R5_GAP;
          M(9);
          R5_GAP;
          if (WPX == 2 && f < 4 && !(UNET_R512_ABLATE & 2)) {
            const int tt = t + 2, i = f * WPX + 1;
            wreg[tt % 3][i >> 2][i & 3] = w_load(tt < 9 ? wCur : wNxt, tt % 9, i >> 2, i & 3);
          }
          if (f == NF - 1 && t < NJ && !(UNET_R512_ABLATE & 1)) issue_piece(gIss, kcIss, t, nbuf);
          R5_GAP;
          M(10);
          M(11);
          R5_GAP;
        }
      }
#undef R5_GAP
      // this wave's pieces of the next chunk have landed, its reads of this chunk are done
      R5_ACCUM(tLoop, tC0);
      R5_STAMP(tB0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      R5_ACCUM(tBar, tB0);
    }
    R5_STAMP(tE0);

    // ---- epilogue straight from the accumulators: lane (li, lq) holds channels 16*lq + [0,16) of its pixel of each
    //      fragment: acc[f][cs][r] is channel 16*lq + 4*cs + r of the wave's channel tile ----
    if (a.dynScale) {
      const float ds = *a.dynScale;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) sc[cs] *= ds;
    }
    const float floorV = a.relu ? 0.f : -3.4e38f;
    const size_t g0 = (size_t)gCur.n * a.H + gCur.y0;   // global row of the tile's first row
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the first accumulator read
    // the fragments' pixel offsets are recomputed per item from an opaque copy of the lane's index: hoisted out of the
    // item loop they are 28 registers that get spilled, and every reload waits (vmcnt is in order) for the stores of the
    // fragment before it - 1700 cycles per fragment
    int liE = li;
    asm volatile("" : "+v"(liE));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = 16 * (NF * wp + f) + liE;
      const int r = i / TWX, c = i - r * TWX;
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
#if UNET_R512_STORE64
        // 4 x 4 transpose of 16-byte pieces across the four lanes of a pixel (two swap stages): store k then writes
        // bytes [64 k + 16 lq, + 16) of the pixel's 256 - 64 contiguous bytes per pixel and instruction
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
#else
        float* rowp = a.outF + pix * (size_t)a.ldo + a.co_off + cbase;
        if (ok) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(rowp + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
#endif
      } else {
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // not clamped: out-of-range values become inf and are reported (amax)
          amax3(amax, v[2 * e], v[2 * e + 1]);
          split_pk_f16_mix(v[2 * e], v[2 * e + 1], ph[e], pl[e]);
        }
#if UNET_R512_STORE64
        // The four lanes of a pixel (lq = 0..3) hold bytes [32 lq, 32 lq + 32) of its 128 bytes per plane as two 16-byte
        // halves; stored as they are, every store instruction writes 16-byte pieces 32 bytes apart.  Two lane-row
        // swaps per register (rows of 16 lanes: odd <-> even rows, then upper <-> lower half wave) hand lane row q
        // bytes [16 q, 16 q + 16) of the first 64 bytes in one register set and of the second 64 in the other: each
        // store instruction then writes 64 contiguous bytes per pixel.
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          auto r = __builtin_amdgcn_permlane16_swap(ph[k], ph[4 + k], false, false);
          auto q = __builtin_amdgcn_permlane32_swap(r[0], r[1], false, false);
          ph[k] = q[0];
          ph[4 + k] = q[1];
          auto rl = __builtin_amdgcn_permlane16_swap(pl[k], pl[4 + k], false, false);
          auto ql = __builtin_amdgcn_permlane32_swap(rl[0], rl[1], false, false);
          pl[k] = ql[0];
          pl[4 + k] = ql[1];
        }
        uint16_t* rowp = a.out + pix * (size_t)a.ldo + a.co_off + (cbase - lq * 16) + lq * 8;
        if (ok) {
          *reinterpret_cast<uint4*>(rowp) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          *reinterpret_cast<uint4*>(rowp + 32) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
          *reinterpret_cast<uint4*>(rowp + a.outLo) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          *reinterpret_cast<uint4*>(rowp + a.outLo + 32) = make_uint4(pl[4], pl[5], pl[6], pl[7]);
        }
#else
        uint16_t* rowp = a.out + pix * (size_t)a.ldo + a.co_off + cbase;
        if (ok) {
          uint4* o = reinterpret_cast<uint4*>(rowp);
          o[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          o[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
          uint4* ol = reinterpret_cast<uint4*>(rowp + a.outLo);
          ol[0] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          ol[1] = make_uint4(pl[4], pl[5], pl[6], pl[7]);
        }
#endif
      }
      __builtin_amdgcn_sched_barrier(0);   // one fragment at a time: 16 values live, not 224
    }
    statCbase = cbase;
    gCur = gNext;
    R5_ACCUM(tEpi, tE0);
  }
  if (EPI != 3) x3_report_range(amax, a.err);
  if (EPI == 3 && a.statPartial) {   // sum over the 16 pixels of a fragment row (lanes li), then one lane per 16 channels
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        ssum[e] += __shfl_xor(ssum[e], m, 64);
        ssq[e] += __shfl_xor(ssq[e], m, 64);
      }
    }
    if (li == 0) {
      float* row = a.statPartial + (size_t)(blockIdx.x * WPX + wp) * 2 * a.Cout + statCbase;
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
