// Split-operand ("f16x3") 3x3 weight gradient:  dW[co][ci][ky][kx] = sum over pixels of dZ[y][x][co] X[y+ky-1][x+kx-1][ci].
//
// The same arithmetic as conv_x3_ws.h (every fp32 operand as fp16 hi + lo, three v_mfma_f32_16x16x32_f16 per product,
// fp32 accumulate), on a GEMM whose K dimension is the PIXEL index: both operands are stored channel-fastest (NHWC
// planes), so the MFMA operands (8 consecutive k per lane) are columns of the LDS image.  gfx950's transposed LDS
// read (ds_read_b64_tr_b16: a 4-row x 16-column block of 16-bit elements per 16-lane group, delivered column-major)
// hands them over without a transposing pass.
//
// Work decomposition
//   K-step  = 32 pixels = 2 image rows x 16 columns (every map height of the network is even; columns past the image
//             edge are zero-filled), enumerated (image, row pair, 16-column strip): `steps` in total;
//   block   = 64 output channels x 64 input channels x 9 taps, over a contiguous range of K-steps (split-K); the
//             partial sums go to slab[split][tap][Cout][Cin] and wgrad_reduce_kernel adds the splits;
//   grid    = 1-D, block b -> XCD b & 7; all blocks of one split (the same pixels, different channel tiles) sit on
//             one XCD and walk the K-steps at the same pace, so its L2 serves their operand reads.
//   waves   = 4 MFMA waves (2 x 2: 32 co x 32 ci each = 2 x 2 fragments x 9 taps = 36 accumulators) + 4 loader waves.
//
// LDS stage (one K-step, 26 KiB, ring of 4):
//   X halo  [plane 2][channel block 4][4 rows x 18 columns][16 channels]   32 B per (pixel, block): 18 KiB
//   dZ      [plane 2][channel block 4][2 rows x 16 columns][16 channels]                            :  8 KiB
// A transposed read takes 4 consecutive pixels (rows of its block) of one channel block; the two 16-lane groups of a
// 32-lane half take 8 consecutive pixels = 256 contiguous bytes = every bank once: conflict free for every tap shift.
// Lane (i, g) of an operand holds k-slots j = 0..3 -> pixel (row 0, column 4g + j), j = 4..7 -> (row 1, column 4g +
// j - 4) of the step, for dZ and (shifted by the tap) for X alike.
//
// Synchronisation: one s_barrier per K-step.  Loader iteration i issues the DMA of step i + 1, waits until step
// i - 1 has landed (vmcnt leaves the two younger steps in flight) and joins barrier i; the MFMA waves compute step s
// after barrier s + 1.  Step i + 1 overwrites the buffer of step i - 3, whose readers passed barrier i - 1 before the
// loader could issue.
//
// Needs Cout % 64 == 0, Cin % 64 == 0, H even.  Inputs must be in the fp16 range (gradients are pre-scaled by a power
// of two: split_planes_scaled_kernel; the reduction multiplies the inverse back in).
#pragma once
#include "conv_x3_ws.h"
#include "lds_dma.h"

namespace unet {

struct WgradX3Args {
  const uint16_t* dz;   // hi plane, NHWC fp16 (N,H,W,Cout); lo plane at dz + dzLo
  size_t dzLo;
  const uint16_t* x;    // hi plane (N,H,W,Cin); lo plane at x + xLo
  size_t xLo;
  const uint16_t* zeros;   // >= 8 zero halfs
  float* slab;          // [split][tap][Cout][Cin]
  int N, H, W, Cout, Cin;
  int strips;           // ceil(W / 16)
  int steps;            // N * (H / 2) * strips
  int stepsPerSplit, splits, tiles, ciTiles;   // tiles = (Cout / 64) * (Cin / 64)
};

struct WgX3 {
  static constexpr int XC = 18, XPX = 4 * XC;            // halo tile: 4 rows x 18 columns
  static constexpr int XBLK = XPX * 32;                  // one (plane, channel block): 2304 B
  static constexpr int XBYTES = 8 * XBLK;                // 18 KiB
  static constexpr int DBLK = 32 * 32;                   // 1 KiB
  static constexpr int DOFF = XBYTES, DBYTES = 8 * DBLK;
  static constexpr int STAGE = XBYTES + DBYTES;          // 26 KiB
  static constexpr int PIECES = STAGE / 1024;            // 26
  static constexpr int XPIECES = XBYTES / 1024;          // 18
  static constexpr int NBUF = 4;
  static constexpr int LDS_BYTES = NBUF * STAGE;
  static constexpr int NJ = 7;                           // pieces per loader wave (the last round duplicates piece 25)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// two transposed reads -> the 8 k-slots of one operand fragment
__device__ __forceinline__ f16x8 tr_frag(unsigned addr0, unsigned addr1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)addr0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)addr1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}

__global__ __launch_bounds__(512, 1) void wgrad3x3_x3_ws_kernel(const WgradX3Args a) {
  using S = WgX3;
  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (split, channel tile): XCD = blockIdx & 7 holds splits congruent to it mod 8
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int split = xcd + 8 * (jb / a.tiles);
  const int tile = jb % a.tiles;
  const int coTile = tile / a.ciTiles, ciTile = tile - coTile * a.ciTiles;
  const int t0 = split * a.stepsPerSplit;
  const int t1 = t0 + a.stepsPerSplit < a.steps ? t0 + a.stepsPerSplit : a.steps;
  const int nSteps = t1 > t0 ? t1 - t0 : 0;   // block-uniform; blocks past the work write a zero slab

  if (wave >= 4) {
    // ---------------- loader waves: wave 4+k issues pieces q = k + 4j ----------------
    const int k = wave - 4;
    const int rows2 = a.H >> 1;
    // per piece: offset of this lane's 16 bytes from the step's first pixel (halfs), and its (row, column) in the tile
    long off[S::NJ];
    int rc[S::NJ];   // X pieces: (halo row << 8) | halo column;  dZ pieces: 0x10000 | column
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) {
      int q = k + 4 * j;
      q = q < S::PIECES ? q : S::PIECES - 1;
      const int v = q * 64 + lane;
      if (q < S::XPIECES) {
        const int half = v & 1, p = v >> 1;
        const int blk = p / S::XPX, px = p - blk * S::XPX;
        const int plane = blk >> 2, cb = blk & 3;
        const int hr = px / S::XC, hc = px - hr * S::XC;
        off[j] = ((long)(hr - 1) * a.W + (hc - 1)) * a.Cin + ciTile * 64 + cb * 16 + half * 8 + (plane ? (long)a.xLo : 0L);
        rc[j] = (hr << 8) | hc;
      } else {
        const int v2 = v - S::XPIECES * 64;
        const int half = v2 & 1, p = v2 >> 1;
        const int blk = p >> 5, px = p & 31;
        const int plane = blk >> 2, cb = blk & 3;
        const int r = px >> 4, c = px & 15;
        off[j] = ((long)r * a.W + c) * a.Cout + coTile * 64 + cb * 16 + half * 8 + (plane ? (long)a.dzLo : 0L);
        rc[j] = 0x10000 | c;
      }
    }
    // position of the step being issued
    int strip = 0, yPair = 0, n = 0;
    {
      strip = t0 % a.strips;
      const int rest = t0 / a.strips;
      yPair = rest % rows2;
      n = rest / rows2;
    }
    auto issue = [&](int buf) __attribute__((always_inline)) {
      const int y0 = 2 * yPair, x0 = 16 * strip;
      const long pix0 = ((long)n * a.H + y0) * a.W + x0;
      const uint16_t* xb = a.x + pix0 * a.Cin;
      const uint16_t* db = a.dz + pix0 * a.Cout;
      char* dst = reinterpret_cast<char*>(smemv) + buf * S::STAGE;
#pragma unroll
      for (int j = 0; j < S::NJ; ++j) {
        int q = k + 4 * j;
        q = q < S::PIECES ? q : S::PIECES - 1;
        const uint16_t* src;
        if (q < S::XPIECES) {   // wave-uniform
          const int yy = y0 + (rc[j] >> 8) - 1, xx = x0 + (rc[j] & 0xFF) - 1;
          const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
          src = ok ? xb + off[j] : a.zeros;
        } else {
          const bool ok = x0 + (rc[j] & 0xFF) < a.W;
          src = ok ? db + off[j] : a.zeros;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      if (++strip == a.strips) {
        strip = 0;
        if (++yPair == rows2) {
          yPair = 0;
          ++n;
        }
      }
    };
    if (nSteps > 0) issue(0);
    for (int i = 0; i <= nSteps; ++i) {
      if (i + 1 < nSteps) {
        issue((i + 1) & 3);
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");   // steps i and i + 1 may still be in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // lane 4q + p of a 16-lane group addresses row q (pixel 4g + q), 16-bit columns 4p .. 4p + 3 of the block
  const unsigned laneOff = (unsigned)((4 * lg + (li >> 2)) * 32 + (li & 3) * 8);
  unsigned aBase0 = lds_address(reinterpret_cast<char*>(smemv)) + S::DOFF + (wm * 2) * S::DBLK + laneOff;
  unsigned bBase0 = lds_address(reinterpret_cast<char*>(smemv)) + (wn * 2) * S::XBLK + laneOff;

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[t][f][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  ws_barrier();   // barrier 0
  for (int s = 0; s < nSteps; ++s) {
    ws_barrier();   // barrier s + 1: step s is in LDS
    const unsigned sb = (unsigned)(s & 3) * S::STAGE;
    const unsigned aB = aBase0 + sb, bB = bBase0 + sb;
    f16x8 ah[2], al[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      ah[f] = tr_frag(aB + f * S::DBLK, aB + f * S::DBLK + 512);
      al[f] = tr_frag(aB + 4 * S::DBLK + f * S::DBLK, aB + 4 * S::DBLK + f * S::DBLK + 512);
    }
    // X operand rows: tap (ky, kx) pairs halo rows ky and ky + 1 at column shift kx, so per kx four row reads (per
    // fragment and plane) serve three taps; they are fetched rolling, one tap ahead of their first use
    s16x4 xr[4][2][2];   // [halo row][fragment e][plane]
    auto rread = [&](int row, int kx) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          xr[row][e][pl] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(size_t)(bB + pl * 4 * S::XBLK + e * S::XBLK +
                                                                  (row * S::XC + kx) * 32));
    };
    auto pair = [](s16x4 lo4, s16x4 hi4) __attribute__((always_inline)) -> f16x8 {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(f16x8, v);
    };
    rread(0, 0);
    rread(1, 0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int t = ky * 3 + kx;
        if (ky < 2) {
          rread(ky + 2, kx);
        } else if (kx < 2) {
          rread(0, kx + 1);
          rread(1, kx + 1);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f16x8 bh = pair(xr[ky][e][0], xr[ky + 1][e][0]);
          const f16x8 bl = pair(xr[ky][e][1], xr[ky + 1][e][1]);
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            acc[t][f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh, acc[t][f][e], 0, 0, 0);
            acc[t][f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl, acc[t][f][e], 0, 0, 0);
            acc[t][f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh, acc[t][f][e], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- slab store: acc[t][f][e][r] = dW[co = 64 coTile + 32 wm + 16 f + 4 lg + r][ci = 64 ciTile + 32 wn + 16 e + li]
  float* out = a.slab + (size_t)split * 9 * a.Cout * a.Cin;
  const size_t co0 = (size_t)coTile * 64 + wm * 32 + 4 * lg;
  const size_t ci0 = (size_t)ciTile * 64 + wn * 32 + li;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((size_t)t * a.Cout + co0 + f * 16 + r) * a.Cin + ci0 + e * 16] = acc[t][f][e][r];
}

// ---------------------------------------------------------------------------------------------------------------
// One-tap variant: C[row][col] = sum over pixels of A[p][row] * B[p][col], both operands NHWC hi/lo planes (the
// transposed convolution's weight gradient: A = the space-to-depth'd output gradient, 4f rows; B = its input, 2f
// columns).  Same transposed-read scheme; with one tap per accumulator the block tile is 128 x 128 (wave tile 64 x 64 =
// 4 x 4 fragments, 48 MFMAs per 32 transposed reads), a K-step is 32 consecutive pixels of the flat pixel list.
// LDS stage: [operand 2][plane 2][channel block 8][32 pixels][16 channels] = 32 KiB, ring of 4.  Slab: [split][rows][cols].
// Needs rows % 128 == 0 and cols % 128 == 0.
// ---------------------------------------------------------------------------------------------------------------
struct Wgrad1X3Args {
  const uint16_t* a;   // hi plane (P, lda halfs per pixel); lo plane at a + aLo
  size_t aLo;
  const uint16_t* b;   // hi plane (P, ldb); lo plane at b + bLo
  size_t bLo;
  const uint16_t* zeros;
  float* slab;         // [split][rows][cols]
  long P;
  int lda, ldb, rows, cols;
  int steps, stepsPerSplit, splits, tiles, colTiles;   // tiles = (rows / 128) * (cols / 128)
};

struct Wg1X3 {
  static constexpr int OPB = 16 * 1024;            // one operand, both planes
  static constexpr int STAGE = 2 * OPB;            // 32 KiB
  static constexpr int NBUF = 4;
  static constexpr int LDS_BYTES = NBUF * STAGE;
  static constexpr int NJ = 8;                     // 32 pieces per stage, 8 per loader wave
};

__global__ __launch_bounds__(512, 1) void wgrad1x1_x3_ws_kernel(const Wgrad1X3Args a) {
  using S = Wg1X3;
  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int split = xcd + 8 * (jb / a.tiles);
  const int tile = jb % a.tiles;
  const int rowTile = tile / a.colTiles, colTile = tile - rowTile * a.colTiles;
  const int t0 = split * a.stepsPerSplit;
  const int t1 = t0 + a.stepsPerSplit < a.steps ? t0 + a.stepsPerSplit : a.steps;
  const int nSteps = t1 > t0 ? t1 - t0 : 0;

  if (wave >= 4) {
    const int k = wave - 4;
    long off[S::NJ];   // this lane's 16 bytes of piece k + 4j relative to the step's first pixel (halfs)
    int pxl[S::NJ];    // its pixel within the step
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) {
      const int q = k + 4 * j;            // 0..15: operand A, 16..31: operand B
      const int v = (q & 15) * 64 + lane;
      const int half = v & 1, p = v >> 1;
      const int blk = p >> 5, px = p & 31;
      const int plane = blk >> 3, cb = blk & 7;
      pxl[j] = px;
      if (q < 16)
        off[j] = (long)px * a.lda + rowTile * 128 + cb * 16 + half * 8 + (plane ? (long)a.aLo : 0L);
      else
        off[j] = (long)px * a.ldb + colTile * 128 + cb * 16 + half * 8 + (plane ? (long)a.bLo : 0L);
    }
    int t = t0;
    auto issue = [&](int buf) __attribute__((always_inline)) {
      const long p0 = (long)t * 32;
      const uint16_t* ab = a.a + p0 * a.lda;
      const uint16_t* bb = a.b + p0 * a.ldb;
      char* dst = reinterpret_cast<char*>(smemv) + buf * S::STAGE;
#pragma unroll
      for (int j = 0; j < S::NJ; ++j) {
        const int q = k + 4 * j;
        const bool ok = p0 + pxl[j] < a.P;
        const uint16_t* src = ok ? ((q < 16 ? ab : bb) + off[j]) : a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
      ++t;
    };
    if (nSteps > 0) issue(0);
    for (int i = 0; i <= nSteps; ++i) {
      if (i + 1 < nSteps) {
        issue((i + 1) & 3);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // steps i and i + 1 may still be in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");
    }
    return;
  }

  const int li = lane & 15, lg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned laneOff = (unsigned)((4 * lg + (li >> 2)) * 32 + (li & 3) * 8);
  const unsigned base0 = lds_address(reinterpret_cast<char*>(smemv)) + laneOff;
  const unsigned aBase0 = base0 + (wm * 4) * 1024;             // row blocks 4 wm .. 4 wm + 3
  const unsigned bBase0 = base0 + S::OPB + (wn * 4) * 1024;    // column blocks 4 wn .. 4 wn + 3

  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[f][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  ws_barrier();
  for (int s = 0; s < nSteps; ++s) {
    ws_barrier();
    const unsigned sb = (unsigned)(s & 3) * S::STAGE;
    const unsigned aB = aBase0 + sb, bB = bBase0 + sb;
    f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      ah[f] = tr_frag(aB + f * 1024, aB + f * 1024 + 512);
      al[f] = tr_frag(aB + 8 * 1024 + f * 1024, aB + 8 * 1024 + f * 1024 + 512);
      bh[f] = tr_frag(bB + f * 1024, bB + f * 1024 + 512);
      bl[f] = tr_frag(bB + 8 * 1024 + f * 1024, bB + 8 * 1024 + f * 1024 + 512);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh[e], acc[f][e], 0, 0, 0);
        acc[f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl[e], acc[f][e], 0, 0, 0);
        acc[f][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh[e], acc[f][e], 0, 0, 0);
      }
  }

  float* out = a.slab + (size_t)split * a.rows * a.cols;
  const size_t r0 = (size_t)rowTile * 128 + wm * 64 + 4 * lg;
  const size_t c0 = (size_t)colTile * 128 + wn * 64 + li;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(r0 + f * 16 + r) * a.cols + c0 + e * 16] = acc[f][e][r];
}

}  // namespace unet
