// int8 tier (model B): convolutions on v_mfma_i32_16x16x64_i8, integer-exact.
//
// Quantisation scheme: unet_lane_detection_amd/quant.py (per-tensor asymmetric int8 activations, per-output-channel
// asymmetric int8 weights, as the reference configures its RKNN conversion: README.md:3106-3116, :3370-3383).  The
// operator computed here is, for every output pixel p and channel co,
//     acc[p,co] = sum_k (qx[p,k] - zx) (qw[k,co] - zw[co])          k over taps x input channels, 0 outside the image
//     y[p,co]   = clamp(rint(float(acc + bias_q[co]) * mult[co]) + zy, lo, 127)          (lo = zy for ReLU, else -128)
// bit for bit as oracle/int8_oracle.py does with 64-bit integers.  The i8 MFMA multiplies signed bytes, so the
// zero points are taken out of the sum:
//     acc = sum_k qx qw  -  zw[co] Sx[p]  -  zx Sw[co]  +  K zx zw[co],      Sx[p] = sum_k qx[p,k],  Sw[co] = sum_k qw[k,co]
// with every sum over the same PADDED K (taps x channels rounded up to 64): out-of-image pixels are staged as zx and
// padded weight rows hold zw[co], so each of them contributes exactly 0 whatever the padded activation bytes hold.
// sum_k qx qw comes from the MFMA; Sx[p] from v_dot4 on the very fragments the MFMA consumes (no extra loads); the
// rest is a per-channel constant c0[co] = bias_q + K zx zw - zx Sw folded on the host.
//
// Kernel: persistent blocks of 256 threads (2 - 3 per CU); a block serves one tile of 64 output channels and walks
// tiles of 128 pixels (TAPS = 9: 8 rows x 16 columns of one image; TAPS = 1: 128 consecutive pixels of the flattened
// tensor).  A tile's whole input halo (all channels, <= 51 KiB) is staged in LDS with a 32-byte pad per pixel
// (conv_i8_pitch: conflict-free ds_read_b128), the next tile's (the narrow layers: the next two tiles') already in
// flight into registers; weight fragments stream from L2 in MFMA A-operand order through a ring of three steps, two
// steps ahead.  Wave w owns 2 pixel fragments x 4 channel subtiles.  Accumulator lane (li, lq) ends up with 16
// consecutive channels of one pixel (channel permutation of conv_bf16_ws.h): one 16-byte store.  TAPS = 1 also
// serves the transposed convolution (column n = (a,b) * CoutPad + co scattered to pixel (2y+a, 2x+b)) and the first
// layer (on 64-byte im2col rows built by im2col27_i8_kernel).  Round 4's measurements of what a tile waits for:
// profiles/r04/t448_experiments.md (last section), profiles/r04/int8_stamps.txt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic build (-DUNET_I8_STAMPS=1, never shipped): s_memtime sums of wave 0 of every block per phase of the tile loop,
// added up in g_i8Stamps and printed per launch by unet_i8.inc
#ifndef UNET_I8_STAMPS
#define UNET_I8_STAMPS 0
#endif
#if UNET_I8_STAMPS
#define I8_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define I8_ACCUM(acc, t0) acc += __builtin_amdgcn_s_memtime() - (t0)
#else
#define I8_STAMP(var)
#define I8_ACCUM(acc, t0)
#endif

namespace unet {

#if UNET_I8_STAMPS
__device__ unsigned long long g_i8Stamps[8];
#endif

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef float f32x4i8 __attribute__((ext_vector_type(4)));

struct ConvI8Args {
  const int8_t* in;      // NHWC int8, pixel stride Cin (multiple of 64)
  const int8_t* wt;      // packed [coTile(64)][chunk(64)][tap][cs(4)][lane(64)][16]
  const int32_t* c0;     // [colsPad] bias_q + K zx zw - zx Sw
  const int32_t* wzp;    // [colsPad]
  const float* mult;     // [colsPad]
  int8_t* out;           // NHWC int8, pixel stride ldo
  int N, H, W, Cin, ldo, co_off;
  int cols;              // valid GEMM columns (TAPS = 9 / plain 1x1: Cout; scatter: 4 * coutPad)
  int coutReal, coutPad; // scatter: real / padded channels per (a,b) group
  int tilesX, tilesY, pixTiles, coTiles;
  int tileBlocks;        // persistent blocks per channel tile (grid = tileBlocks * coTiles)
  int8_t* dump;          // >= 64 KiB of scratch: conv_i8_lw_kernel's lanes that have nothing to store put their 16 bytes here
  int xzp, yzp, lo;
  int scatter;
  // TAPS = 9 with a 32-byte pixel stride (<= 32 input channels): the K = 64 of one MFMA holds TWO taps' 32 channels
  // (5 steps instead of 9; step 4's second half is padding: weight rows = zw, any staged bytes), so narrow tensors
  // are stored and staged at 32 bytes per pixel instead of being padded to 64
  int pair;
};

__device__ __forceinline__ int rint_mul(int t, float m) { return (int)rintf(__fmul_rn((float)t, m)); }

// LDS bytes per staged pixel.  ds_read_b128 is served in four groups of 16 lanes - lanes {0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31} and the same + 32 - and a group takes one cycle when its lanes hit 16 distinct 16-byte slots of the
// 256-byte bank row; lane (li, lq) reads slot (pixel * pitch / 16 + lq) mod 16 of consecutive pixels li.  A pad of 32
// bytes (pitch / 16 = 2 mod 4) gives 4 cycles per read for every channel count and both tap layouts; the 16-byte pad
// used until round 4 gave 8 (rocprofv3: SQ_LDS_BANK_CONFLICT 0.43 of the LDS cycles) - tools/lds_conflicts.py --i8.
// The 32-channel pair layout (lq & 1 picks the half, lq >> 1 the tap) is conflict free without a pad.
__host__ __device__ __forceinline__ int conv_i8_pitch(int cin) { return cin == 32 ? 32 : cin + 32; }

// ---- epilogue: zero-point corrections, requantisation, one 16-byte store per fragment ----
// Layers with <= 32 output columns (the 224 x 224 level) fill only the lanes lq < 2 of every accumulator: fragment 1's
// useful half is swapped into fragment 0's idle lanes (v_permlane32_swap) and ONE pass requantises both fragments
// (these layers spend as long in this VALU code as in their MFMAs).  zw * Sx as a 24-bit multiply: |zw| <= 128,
// |Sx| <= K * 128 < 2^23 for every K this kernel accepts (v_mul_lo_u32 runs at a quarter of the rate).
// ALWAYS: every lane issues its store(s) - lanes with nothing to store write their 16 bytes to dumpSlot - so that the number of
// store instructions per tile is a constant (conv_i8_lw_kernel counts them in its s_waitcnt).
template <int TAPS, int MS, bool ALWAYS>
__device__ __forceinline__ void i8_epilogue(const ConvI8Args& a, v4i32 (&acc)[MS][4], int (&sx)[MS], const int* ldsC0, int coTile,
                                            int wave, int li, int lq, int n, int y0, int x0, long p0, long npix,
                                            int8_t* dumpSlot) {
  const bool half = a.cols <= 32;
  int sxr[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    int s = sx[ms];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);   // Sx of pixel li over all K
    sxr[ms] = s;
  }
  if (half) {
#pragma unroll
    for (int mp = 0; mp < MS; mp += 2)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          auto sw = __builtin_amdgcn_permlane32_swap(acc[mp][cs][r], acc[mp + 1][cs][r], false, false);
          acc[mp][cs][r] = sw[0];   // lanes 0-31: the even fragment's columns 0-31; lanes 32-63: the odd one's
        }
  }
  const int lqc = half ? (lq & 1) : lq;          // which 16 columns of the channel tile this lane requantises
  const int colB = coTile * 64 + lqc * 16;
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    if (half && (ms & 1)) continue;   // uniform: the odd fragment went with the even one
    const int s = half ? (lq >= 2 ? sxr[ms | 1] : sxr[ms]) : sxr[ms];
    uint32_t pk[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      const v4i32 c0v = *reinterpret_cast<const v4i32*>(ldsC0 + lqc * 16 + cs * 4);
      const v4i32 zwv = *reinterpret_cast<const v4i32*>(ldsC0 + 64 + lqc * 16 + cs * 4);   // -zw
      const f32x4i8 mv = *reinterpret_cast<const f32x4i8*>(ldsC0 + 128 + lqc * 16 + cs * 4);
      uint32_t w = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = __mul24(zwv[r], s) + acc[ms][cs][r] + c0v[r];   // zwv = -zw: acc - zw * Sx + c0 (v_mad_i32_i24 + add)
        int q = rint_mul(t, mv[r]) + a.yzp;
        q = min(max(q, a.lo), 127);
        w |= (uint32_t)(q & 0xFF) << (8 * r);
      }
      pk[cs] = w;
    }
    const int f = wave * MS + (half ? ms + (lq >> 1) : ms);
    if (TAPS == 9) {
      const int y = y0 + f, x = x0 + li;
      const bool okS = y < a.H && x < a.W && colB < a.cols;
      int8_t* const dst = a.out + (((size_t)n * a.H + y) * a.W + x) * (size_t)a.ldo + a.co_off + colB;
      if (ALWAYS)
        *reinterpret_cast<uint4*>(okS ? dst : dumpSlot) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      else if (okS)
        *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    } else if (ALWAYS) {
      const long p = p0 + f * 16 + li;
      int8_t* dst = dumpSlot;
      if (p < npix) {
        if (!a.scatter) {
          if (colB < a.cols) dst = a.out + (size_t)p * (size_t)a.ldo + a.co_off + colB;
        } else {
          const int ab = colB / a.coutPad, co = colB - ab * a.coutPad;
          if (ab < 4 && co < a.coutReal) {
            const int x = (int)(p % a.W);
            const long row = p / a.W;   // n*H + y
            dst = a.out + ((size_t)(2 * row + (ab >> 1)) * (size_t)(2 * a.W) + 2 * x + (ab & 1)) * (size_t)a.ldo + a.co_off + co;
          }
        }
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    } else {
      const long p = p0 + f * 16 + li;
      if (p < npix) {
        if (!a.scatter) {
          if (colB < a.cols)
            *reinterpret_cast<uint4*>(a.out + (size_t)p * (size_t)a.ldo + a.co_off + colB) =
                make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
          const int ab = colB / a.coutPad, co = colB - ab * a.coutPad;
          if (ab < 4 && co < a.coutReal) {
            const int x = (int)(p % a.W);
            const long row = p / a.W;   // n*H + y
            const size_t o = ((size_t)(2 * row + (ab >> 1)) * (size_t)(2 * a.W) + 2 * x + (ab & 1)) * (size_t)a.ldo;
            *reinterpret_cast<uint4*>(a.out + o + a.co_off + co) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          }
        }
      }
    }
  }
}

template <int TAPS, bool PAIR, int NIT, int MS>
__global__ __launch_bounds__(256, (NIT == 2 && MS == 2) ? 3 : 2) void conv_i8_kernel(const ConvI8Args a) {
  // MS = fragments (16 pixels) per wave: 2 (8 x 16-pixel tiles / 128 pixels) is what the host launches.  4 (16 x 16 / 256) was
  // measured for the narrow layers, whose per-tile waits (profiles/r04/int8_stamps.txt: the in-order vmcnt makes the loop top
  // wait for the previous tile's stores) would then be paid half as often: 0 - 15 % slower (two blocks per CU instead of three).
  constexpr int TH = 4 * MS, TW = 16, HALO = TAPS == 9 ? 1 : 0;
  constexpr int HR = TH + 2 * HALO, HC = TW + 2 * HALO;
  static_assert(!PAIR || TAPS == 9, "the pair layout is a 3x3 layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  // Persistent blocks: block b serves channel tile b / tileBlocks and walks pixel tiles b % tileBlocks, + tileBlocks, ...
  // (at the 224 x 224 level a tile is 40 - 72 MFMAs per wave: one block per tile spent its time on the staging round
  // trip and the epilogue's 48 constants per lane, 33k frames/s at MFMA busy 0.19).  The next tile's input is already on
  // its way into registers while this tile computes; the constants sit in LDS behind the tile.
  const int coTile = blockIdx.x / a.tileBlocks;
  const int firstTile = blockIdx.x - coTile * a.tileBlocks;
  const int pitch = conv_i8_pitch(a.Cin);
  const int nChunks = PAIR ? 1 : a.Cin >> 6;
  const int vpp = a.Cin >> 4;                          // 16-byte vectors per pixel
  const int vshift = 31 - __builtin_clz(vpp);
  const bool vpow2 = (vpp & (vpp - 1)) == 0;           // (every width of model B; other widths divide)
  const long npix = (long)a.N * a.H * a.W;
  const int tilePx = TAPS == 9 ? HR * HC : 64 * MS;
  const int total = tilePx * vpp;                      // 16-byte vectors of one staged tile
  int* const ldsC0 = reinterpret_cast<int*>(smem8 + tilePx * pitch);   // [64] c0, [64] wzp, [64] mult of this channel tile
  if (tid < 64) {
    ldsC0[tid] = a.c0[coTile * 64 + tid];
    ldsC0[64 + tid] = -a.wzp[coTile * 64 + tid];   // negated: the epilogue's multiply-add
    reinterpret_cast<float*>(ldsC0)[128 + tid] = a.mult[coTile * 64 + tid];
  }

  // weight fragments: a ring of three steps, two ahead of the MFMAs (an L2 hit is ~700 cycles, a step's 8 MFMAs ~130);
  // the index is clamped at the end instead of guarded, so that every step issues the same loads and the compiler counts
  // vmcnt instead of draining
  constexpr int SPC = PAIR ? 5 : TAPS;   // steps per 64-byte K chunk
  const int steps = nChunks * SPC;
  const v4i32* wbase = reinterpret_cast<const v4i32*>(a.wt) + (size_t)coTile * steps * 4 * 64 + lane;
  v4i32 wf[3][4];
  auto w_load = [&](int slot, int s) __attribute__((always_inline)) {
    const int sc = s < steps ? s : steps - 1;
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) wf[slot][cs] = wbase[((size_t)sc * 4 + cs) * 64];
  };

  // ---- staging: a thread's vectors of one round (NIT x 256 vectors) are all loaded before the first is stored ----
  // NIT (2, 3 or 6, chosen by the host from the channel count): vectors per thread and round; one round covers the whole
  // tile up to 128 channels (3x3) / 192 (1x1).  The narrow layers (NIT <= 3: the 224 x 224 and 112 x 112 levels, 6 - 12 KiB
  // per tile) keep TWO tiles ahead in registers: a CU needs ~16 KiB in flight to cover the HBM latency at 2 TB/s
  constexpr bool PF2 = NIT <= 3 && MS == 2;
  const uint32_t zb = (uint32_t)(a.xzp & 0xFF) * 0x01010101u;
  const uint4 zfill = make_uint4(zb, zb, zb, zb);
  struct Geo {
    int n, y0, x0;
    long p0;
  };
  auto geo_of = [&](int tile) __attribute__((always_inline)) -> Geo {
    Geo g = {0, 0, 0, 0};
    if (TAPS == 9) {
      const int rowTile = tile / a.tilesX;
      g.x0 = (tile - rowTile * a.tilesX) * TW;
      g.n = rowTile / a.tilesY;
      g.y0 = (rowTile - g.n * a.tilesY) * TH;
    } else {
      g.p0 = (long)tile * (64 * MS);
    }
    return g;
  };
  // Address arithmetic of the staging loops: v_mul_lo_u32 / v_mul_hi_u32 run at a quarter of the VALU rate and a tile of
  // the narrow layers is only 40 - 72 MFMAs per wave, so for the power-of-two channel counts (all of model B's) the
  // products are shifts and 24-bit multiplies (W < 2^23, pixel indices < 2^31: the host checks), and px / 18 is
  // (px * 57) >> 10 (exact below 180).
  const int cshift = vshift + 4;   // log2(Cin) when vpow2
  auto px_of = [&](int i, int& v) __attribute__((always_inline)) -> int {
    if (vpow2) {
      v = i & (vpp - 1);
      return i >> vshift;
    }
    const int px = i / vpp;
    v = i - px * vpp;
    return px;
  };
  auto lds_off = [&](int px, int v) __attribute__((always_inline)) -> int {
    if (vpow2) return (PAIR ? 0 : (px << cshift)) + (px << 5) + (v << 4);   // pitch = Cin + 32, or 32 (pair layout)
    return px * pitch + v * 16;
  };
  auto src_off = [&](int pix, int v) __attribute__((always_inline)) -> size_t {
    if (vpow2) return ((size_t)(unsigned)pix << cshift) + (size_t)(v << 4);
    return (size_t)pix * (size_t)a.Cin + v * 16;
  };
  auto stage_load = [&](const Geo& g, int base, uint4 (&val)[NIT]) __attribute__((always_inline)) {
    const int rowBase = (g.n * a.H + g.y0 - 1) * a.W + g.x0 - 1;   // wave-uniform
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = base + it * 256 + tid;
      val[it] = zfill;   // out-of-image pixels hold the input zero point
      if (i < total) {
        int v;
        const int px = px_of(i, v);
        if (TAPS == 9) {
          const int hr = __mul24(px, 57) >> 10, hc = px - hr * HC;
          const int y = g.y0 - 1 + hr, x = g.x0 - 1 + hc;
          if (y >= 0 && y < a.H && x >= 0 && x < a.W)
            val[it] = *reinterpret_cast<const uint4*>(a.in + src_off(rowBase + __mul24(hr, a.W) + hc, v));
        } else {
          if (g.p0 + px < npix) val[it] = *reinterpret_cast<const uint4*>(a.in + src_off((int)g.p0 + px, v));
        }
      }
    }
  };
  auto stage_store = [&](int base, const uint4 (&val)[NIT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = base + it * 256 + tid;
      if (i < total) {
        int v;
        const int px = px_of(i, v);
        *reinterpret_cast<uint4*>(smem8 + lds_off(px, v)) = val[it];
      }
    }
  };

  // this lane's pixel in fragment ms (tile-local halo coordinates of its tap (0,0))
  int pixBase[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int f = wave * MS + ms;   // fragment = tile row (TAPS 9) or 16 consecutive pixels (TAPS 1)
    pixBase[ms] = (TAPS == 9 ? f * HC + li : f * 16 + li) * pitch + (PAIR ? (lq & 1) : lq) * 16;
  }
  // byte offset of step t of a chunk.  PAIR: this lane's 16 bytes are channels 16 (lq & 1) .. of tap 2 t + (lq >> 1); the
  // padding half of the last step reads tap 8 again (its weight rows are zw)
  auto tap_off = [&](int t) __attribute__((always_inline)) -> int {
    if (TAPS != 9) return 0;
    if (PAIR) {
      const int tp = 2 * t + (lq >> 1) > 8 ? 8 : 2 * t + (lq >> 1);
      return ((tp / 3) * HC + (tp % 3)) * pitch;
    }
    return ((t / 3) * HC + (t % 3)) * pitch;
  };

#if UNET_I8_STAMPS
  unsigned long long tStore = 0, tBar1 = 0, tPre = 0, tLoop = 0, tEpi = 0, tBar2 = 0, nTiles = 0;
  const unsigned long long tKernel = __builtin_amdgcn_s_memtime();
#endif
  uint4 val[NIT], val2[PF2 ? NIT : 1];
  Geo gCur = geo_of(firstTile);
  if (firstTile < a.pixTiles) stage_load(gCur, 0, val);
  if (PF2) {
    const int t1 = firstTile + a.tileBlocks;
    if (t1 < a.pixTiles) stage_load(geo_of(t1), 0, reinterpret_cast<uint4(&)[NIT]>(val2));
  }
  for (int tile = firstTile; tile < a.pixTiles; tile += a.tileBlocks) {
    // ---- this tile's input into LDS (rounds past the first are loaded here: > 128 / 192 channels) ----
    I8_STAMP(t0);
    stage_store(0, val);
    for (int base = NIT * 256; base < total; base += NIT * 256) {
      stage_load(gCur, base, val);
      stage_store(base, val);
    }
    w_load(0, 0);
    w_load(1, 1);
    I8_ACCUM(tStore, t0);
    I8_STAMP(t1);
    __syncthreads();
    I8_ACCUM(tBar1, t1);
    I8_STAMP(t2);
    // ---- the next tile's first round sets off now and lands under the MFMAs ----
    const int nextTile = tile + a.tileBlocks;
    const Geo gNext = geo_of(nextTile < a.pixTiles ? nextTile : tile);
    if (PF2) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) val[it] = val2[it];
      const int t2 = nextTile + a.tileBlocks;
      if (t2 < a.pixTiles) stage_load(geo_of(t2), 0, reinterpret_cast<uint4(&)[NIT]>(val2));
    } else if (nextTile < a.pixTiles) {
      stage_load(gNext, 0, val);
    }

    I8_ACCUM(tPre, t2);
    I8_STAMP(t3);
    v4i32 acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[ms][cs] = (v4i32){0, 0, 0, 0};
    int sx[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) sx[ms] = 0;
    auto step = [&](int slot, int off) __attribute__((always_inline)) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const v4i32 xf = *reinterpret_cast<const v4i32*>(smem8 + pixBase[ms] + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) sx[ms] = __builtin_amdgcn_sdot4(xf[e], 0x01010101, sx[ms], false);
#pragma unroll
        for (int cs = 0; cs < 4; ++cs)
          acc[ms][cs] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[slot][cs], xf, acc[ms][cs], 0, 0, 0);
      }
    };
    if (TAPS == 9) {
      // SPC = 9 or 5 steps per chunk, unrolled: ring slot = step % 3 is static (9 % 3 == 0; the pair layout has one chunk)
#pragma unroll(MS == 2 ? 2 : 1)
      for (int kc = 0; kc < nChunks; ++kc) {
#pragma unroll
        for (int t = 0; t < SPC; ++t) {
          w_load((t + 2) % 3, kc * SPC + t + 2);
          step(t % 3, tap_off(t) + kc * 64);
        }
      }
    } else {
      // one step per chunk: three at a time so that the ring slots are static
      for (int s = 0; s < steps; s += 3) {
        w_load(2, s + 2);
        step(0, s * 64);
        if (s + 1 < steps) {
          w_load(0, s + 3);
          step(1, (s + 1) * 64);
        }
        if (s + 2 < steps) {
          w_load(1, s + 4);
          step(2, (s + 2) * 64);
        }
      }
    }

    // ---- epilogue: zero-point corrections, requantisation, one 16-byte store per fragment ----
#if UNET_I8_STAMPS
    asm volatile("s_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][3]));   // (the MFMAs have issued; their latency goes to the epilogue)
#endif
    I8_ACCUM(tLoop, t3);
    I8_STAMP(t4);
    i8_epilogue<TAPS, MS, false>(a, acc, sx, ldsC0, coTile, wave, li, lq, gCur.n, gCur.y0, gCur.x0, gCur.p0, npix, nullptr);
    gCur = gNext;
    I8_ACCUM(tEpi, t4);
    I8_STAMP(t5);
    __syncthreads();   // every wave is done reading this tile before the next one is stored over it
    I8_ACCUM(tBar2, t5);
#if UNET_I8_STAMPS
    ++nTiles;
#endif
  }
#if UNET_I8_STAMPS
  if (tid == 0) {
    atomicAdd(&g_i8Stamps[0], tStore);
    atomicAdd(&g_i8Stamps[1], tBar1);
    atomicAdd(&g_i8Stamps[2], tPre);
    atomicAdd(&g_i8Stamps[3], tLoop);
    atomicAdd(&g_i8Stamps[4], tEpi);
    atomicAdd(&g_i8Stamps[5], tBar2);
    atomicAdd(&g_i8Stamps[6], nTiles);
    atomicAdd(&g_i8Stamps[7], __builtin_amdgcn_s_memtime() - tKernel);
  }
#endif
}

// The narrow layers' kernel (<= 64 input channels: the 224 x 224 and 112 x 112 levels of model B, half of the tier's time).
// A tile of such a layer is 40 - 72 MFMAs per wave; what bounds it is how many bytes a CU keeps in flight (three blocks x
// one 6 - 17 KiB tile against ~4 us of loaded latency is ~1.2 TB/s chip-wide, which is what conv_i8_kernel reaches on
// them).  Deeper prefetch into registers does not help there: the vector-memory counter is in order, so the first wait
// for a weight fragment also waits for every older prefetch load.  Here nothing but the input loads and the output stores
// touches vector memory inside the tile loop:
//   * the channel tile's weights (steps x 4 KiB) and constants are copied into LDS once per block and read from there;
//   * the input of the next TWO tiles is in flight in two register sets, loaded by inline assembly (hipcc then keeps no
//     score of them) and awaited with a counted s_waitcnt: behind a set's loads only the other set's loads and one tile's
//     output stores may still be outstanding - and their number is a constant, because every lane always issues its
//     loads (lanes outside the tile or the image read a valid address and their result is replaced by the zero point)
//     and its stores (lanes with nothing to store write to the scratch page a.dump).
// Same arithmetic, same packed weights, same LDS tile layout and epilogue as conv_i8_kernel: bit-identical results.
// The assembly loads' destination registers must not be copied or spilled between issue and wait (hipcc believes they
// hold their value from the asm statement on): the instances compile without scratch and without such moves at the
// register bounds below - forcing four blocks per CU (128 registers) spills them and the results are wrong at once, which
// tests/test_int8_gpu.py (every intermediate tensor, bit for bit) shows.  Rebuilding with another compiler means re-running it.
template <int TAPS, bool PAIR, int NIT>
__global__ __launch_bounds__(256, (TAPS == 9 && !PAIR) ? 2 : 3) void conv_i8_lw_kernel(const ConvI8Args a) {
  constexpr int MS = 2;
  constexpr int TH = 4 * MS, TW = 16, HALO = TAPS == 9 ? 1 : 0;
  constexpr int HR = TH + 2 * HALO, HC = TW + 2 * HALO;
  static_assert(!PAIR || TAPS == 9, "the pair layout is a 3x3 layout");
  static_assert(NIT >= 1 && NIT <= 3, "one round of at most three vectors per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int coTile = blockIdx.x / a.tileBlocks;
  const int firstTile = blockIdx.x - coTile * a.tileBlocks;
  const int pitch = conv_i8_pitch(a.Cin);
  const int vpp = a.Cin >> 4;                 // 16-byte vectors per pixel: a power of two here (the host checks)
  const int vshift = 31 - __builtin_clz(vpp);
  const int cshift = vshift + 4;
  const long npix = (long)a.N * a.H * a.W;
  constexpr int tilePx = TAPS == 9 ? HR * HC : 64 * MS;
  const int total = tilePx * vpp;
  constexpr int SPC = PAIR ? 5 : TAPS;
  const int steps = (PAIR ? 1 : a.Cin >> 6) * SPC;
  int* const ldsC0 = reinterpret_cast<int*>(smem8 + tilePx * pitch);
  unsigned char* const ldsW = smem8 + tilePx * pitch + 768;   // [step][cs][lane] 16 bytes
  if (tid < 64) {
    ldsC0[tid] = a.c0[coTile * 64 + tid];
    ldsC0[64 + tid] = -a.wzp[coTile * 64 + tid];   // negated: the epilogue's multiply-add
    reinterpret_cast<float*>(ldsC0)[128 + tid] = a.mult[coTile * 64 + tid];
  }
  {
    const v4i32* wsrc = reinterpret_cast<const v4i32*>(a.wt) + (size_t)coTile * steps * 256;
    for (int i = tid; i < steps * 256; i += 256) reinterpret_cast<v4i32*>(ldsW)[i] = wsrc[i];
  }

  const uint32_t zb = (uint32_t)(a.xzp & 0xFF) * 0x01010101u;
  struct Geo {
    int n, y0, x0;
    long p0;
  };
  auto geo_of = [&](int tile) __attribute__((always_inline)) -> Geo {
    Geo g = {0, 0, 0, 0};
    if (TAPS == 9) {
      const int rowTile = tile / a.tilesX;
      g.x0 = (tile - rowTile * a.tilesX) * TW;
      g.n = rowTile / a.tilesY;
      g.y0 = (rowTile - g.n * a.tilesY) * TH;
    } else {
      g.p0 = (long)tile * (64 * MS);
    }
    return g;
  };
  // one set of NIT loads; always issued by every lane (see above).  valid: bit it = the vector lies in the tile and the image
  auto asm_issue = [&](int tile, v4i32 (&pv)[NIT], unsigned& valid) __attribute__((always_inline)) {
    const Geo g = geo_of(tile < a.pixTiles ? tile : firstTile);
    const int rowBase = (g.n * a.H + g.y0 - 1) * a.W + g.x0 - 1;
    valid = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = it * 256 + tid;
      const int px = i >> vshift, v = i & (vpp - 1);
      bool ok = i < total && tile < a.pixTiles;
      int pix;
      if (TAPS == 9) {
        const int hr = __mul24(px, 57) >> 10, hc = px - hr * HC;   // px / 18, exact below 180
        const int y = g.y0 - 1 + hr, x = g.x0 - 1 + hc;
        ok = ok && y >= 0 && y < a.H && x >= 0 && x < a.W;
        pix = rowBase + __mul24(hr, a.W) + hc;
      } else {
        ok = ok && g.p0 + px < npix;
        pix = (int)g.p0 + px;
      }
      const int8_t* src = ok ? a.in + (((size_t)(unsigned)pix << cshift) + (size_t)(v << 4)) : a.in;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pv[it]) : "v"(src) : "memory");
      valid |= ok ? 1u << it : 0u;
    }
  };
  auto consume = [&](v4i32 (&pv)[NIT], unsigned valid) __attribute__((always_inline)) {
    const v4i32 zf = (v4i32){(int)zb, (int)zb, (int)zb, (int)zb};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      asm volatile("" : "+v"(pv[it]));   // only now is the value what the load returned
      const int i = it * 256 + tid;
      if (i < total) {
        const int px = i >> vshift, v = i & (vpp - 1);
        const int off = (PAIR ? 0 : (px << cshift)) + (px << 5) + (v << 4);   // pitch = Cin + 32, or 32 (pair layout)
        *reinterpret_cast<v4i32*>(smem8 + off) = (valid >> it & 1) ? pv[it] : zf;
      }
    }
  };

  int pixBase[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int f = wave * MS + ms;
    pixBase[ms] = (TAPS == 9 ? f * HC + li : f * 16 + li) * pitch + (PAIR ? (lq & 1) : lq) * 16;
  }
  auto tap_off = [&](int t) __attribute__((always_inline)) -> int {
    if (TAPS != 9) return 0;
    if (PAIR) {
      const int tp = 2 * t + (lq >> 1) > 8 ? 8 : 2 * t + (lq >> 1);
      return ((tp / 3) * HC + (tp % 3)) * pitch;
    }
    return ((t / 3) * HC + (t % 3)) * pitch;
  };
  int8_t* const dumpSlot = a.dump + ((size_t)(blockIdx.x & 15) * 256 + tid) * 16;
  const bool half = a.cols <= 32;   // one store per wave and tile instead of two (i8_epilogue)
  // (the epilogue's 48 constants per lane stay in LDS: resident in registers they cost a block per CU and 4 - 18 % of the time)

  // one tile: wait for its set (counted), stage it, MFMAs from LDS only, refill the set for the tile two ahead, epilogue
#if UNET_I8_STAMPS
  unsigned long long tStore = 0, tBar1 = 0, tPre = 0, tLoop = 0, tEpi = 0, tBar2 = 0, nTiles = 0;
  const unsigned long long tKernel = __builtin_amdgcn_s_memtime();
#endif
  auto tile_body = [&](int tile, v4i32 (&pv)[NIT], unsigned& valid, bool first) __attribute__((always_inline)) {
    I8_STAMP(t0);
    // younger than this set's loads: the other set's NIT loads and, unless this is the block's first tile, the previous
    // tile's stores (1 or 2 per wave)
    if (first) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIT) : "memory");
    } else if (half) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIT + 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIT + 2) : "memory");
    }
    I8_ACCUM(tPre, t0);   // (stamps: pre = the counted wait)
    I8_STAMP(t1);
    consume(pv, valid);
    I8_ACCUM(tStore, t1);
    I8_STAMP(t2);
    __syncthreads();
    I8_ACCUM(tBar1, t2);
    I8_STAMP(t3);
    const Geo g = geo_of(tile);
    v4i32 acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[ms][cs] = (v4i32){0, 0, 0, 0};
    int sx[MS] = {0, 0};
    auto step = [&](int s, int off) __attribute__((always_inline)) {
      v4i32 wv[4];
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) wv[cs] = *reinterpret_cast<const v4i32*>(ldsW + ((s * 4 + cs) * 64 + lane) * 16);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        const v4i32 xf = *reinterpret_cast<const v4i32*>(smem8 + pixBase[ms] + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) sx[ms] = __builtin_amdgcn_sdot4(xf[e], 0x01010101, sx[ms], false);
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[ms][cs] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[cs], xf, acc[ms][cs], 0, 0, 0);
      }
    };
    if (TAPS == 9) {
#pragma unroll
      for (int t = 0; t < SPC; ++t) step(t, tap_off(t));   // (<= 64 channels: one chunk)
    } else {
      step(0, 0);
    }
#if UNET_I8_STAMPS
    asm volatile("s_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][3]));
#endif
    I8_ACCUM(tLoop, t3);
    I8_STAMP(t4);
    asm_issue(tile + 2 * a.tileBlocks, pv, valid);   // the set is free again: the tile two ahead
    i8_epilogue<TAPS, MS, true>(a, acc, sx, ldsC0, coTile, wave, li, lq, g.n, g.y0, g.x0, g.p0, npix, dumpSlot);
    I8_ACCUM(tEpi, t4);
    I8_STAMP(t5);
    __syncthreads();   // every wave is done reading this tile before the next one is stored over it
    I8_ACCUM(tBar2, t5);
#if UNET_I8_STAMPS
    ++nTiles;
#endif
  };

  v4i32 pvA[NIT], pvB[NIT];
  unsigned validA = 0, validB = 0;
  __syncthreads();   // weights and constants are in LDS (hipcc waits for their loads before the stores)
  asm_issue(firstTile, pvA, validA);
  asm_issue(firstTile + a.tileBlocks, pvB, validB);
  bool first = true;
  for (int tile = firstTile; tile < a.pixTiles; tile += 2 * a.tileBlocks) {
    tile_body(tile, pvA, validA, first);
    first = false;
    if (tile + a.tileBlocks < a.pixTiles) tile_body(tile + a.tileBlocks, pvB, validB, false);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sets in flight past the last tile land before the registers are given up
#if UNET_I8_STAMPS
  if (tid == 0) {
    atomicAdd(&g_i8Stamps[0], tStore);
    atomicAdd(&g_i8Stamps[1], tBar1);
    atomicAdd(&g_i8Stamps[2], tPre);
    atomicAdd(&g_i8Stamps[3], tLoop);
    atomicAdd(&g_i8Stamps[4], tEpi);
    atomicAdd(&g_i8Stamps[5], tBar2);
    atomicAdd(&g_i8Stamps[6], nTiles);
    atomicAdd(&g_i8Stamps[7], __builtin_amdgcn_s_memtime() - tKernel);
  }
#endif
}

// uint8 RGB frame -> 64-byte int8 im2col rows of the first convolution: k = tap*3 + ci (27 used), the per-channel
// table applies (u8 - mean) / std and the input quantisation; out-of-image taps hold the input zero point
__global__ __launch_bounds__(256) void im2col27_i8_kernel(const uint8_t* __restrict__ frames, const int8_t* __restrict__ lut,
                                                          int n, int h, int w, int zin, int8_t* __restrict__ out) {
  __shared__ int8_t tab[3 * 256];
  for (int i = threadIdx.x; i < 768; i += 256) tab[i] = lut[i];
  __syncthreads();
  const size_t npix = (size_t)n * h * w;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    const int x = (int)(p % w);
    const size_t row = p / w;
    const int y = (int)(row % h);
    // all 27 bytes first, the table afterwards: with the lookup right behind each load hipcc waited vmcnt(0) 27 times per
    // pixel, one memory round trip after the other (round 4: 0.41 -> 0.2x ms at batch 256)
    unsigned raw[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int t = k / 3, ci = k - t * 3;
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      raw[k] = 0x100;   // outside the image
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) raw[k] = frames[((row + (t / 3 - 1)) * (size_t)w + xx) * 3 + ci];
    }
    uint32_t words[16];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      uint32_t wv = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k4 * 4 + e;
        int q = zin;
        if (k < 27 && raw[k < 27 ? k : 0] < 0x100) q = tab[(k % 3) * 256 + raw[k < 27 ? k : 0]];
        wv |= (uint32_t)(q & 0xFF) << (8 * e);
      }
      words[k4] = wv;
    }
    uint4* o = reinterpret_cast<uint4*>(out + p * 64);
#pragma unroll
    for (int v = 0; v < 4; ++v) o[v] = make_uint4(words[4 * v], words[4 * v + 1], words[4 * v + 2], words[4 * v + 3]);
  }
}

// MaxPool2d(2,2) on int8 NHWC (16 channels per thread): x (N,H,W,ldi) channels [0,c) -> y (N,H/2,W/2,ldo)
__global__ __launch_bounds__(256) void maxpool2x2_i8_kernel(const int8_t* __restrict__ x, int n, int h, int w, int c,
                                                            int ldi, int8_t* __restrict__ y, int ldo) {
  const int cv = c / 16;
  const size_t total = (size_t)n * (h / 2) * (w / 2) * cv;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int v = (int)(i % cv);
    size_t p = i / cv;
    const int xo = (int)(p % (w / 2));
    p /= (w / 2);
    const int yo = (int)(p % (h / 2));
    const int nn = (int)(p / (h / 2));
    int8_t m[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e] = -128;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const uint4 q = *reinterpret_cast<const uint4*>(x + (((size_t)nn * h + 2 * yo + dy) * w + 2 * xo + dx) * (size_t)ldi + v * 16);
        const int8_t* b = reinterpret_cast<const int8_t*>(&q);
#pragma unroll
        for (int e = 0; e < 16; ++e) m[e] = b[e] > m[e] ? b[e] : m[e];
      }
    *reinterpret_cast<uint4*>(y + (((size_t)nn * (h / 2) + yo) * (w / 2) + xo) * (size_t)ldo + v * 16) =
        *reinterpret_cast<const uint4*>(m);
  }
}

// 1x1 head on int8 (the blob's ConvSigmoid): logit = float(sum_c (qx - zx)(qw - zw) + bias_q) * mult.
// One thread per pixel, 16 channels per load; sum (qx - zx)(qw - zw) = sum qx qw - zw sum qx - zx sum qw + c zx zw with
// the byte sums from v_dot4 (exact in int32).  c % 16 == 0.
__global__ __launch_bounds__(256) void head_i8_kernel(const int8_t* __restrict__ x, int ldx, int c, size_t npix,
                                                      const int8_t* __restrict__ wq, int wzp, int xzp, int biasq,
                                                      float mult, float* __restrict__ logits, float* __restrict__ probs,
                                                      uint8_t* __restrict__ mask, float thr) {
  const size_t stride = (size_t)gridDim.x * 256;
  int sw = 0;
  for (int i = 0; i < c; i += 4) sw = __builtin_amdgcn_sdot4(*reinterpret_cast<const int*>(wq + i), 0x01010101, sw, false);
  const int cst = biasq - xzp * sw + c * xzp * wzp;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    int dot = 0, sx = 0;
    for (int i = 0; i < c; i += 16) {
      const v4i32 xv = *reinterpret_cast<const v4i32*>(x + p * ldx + i);
      const v4i32 wv = *reinterpret_cast<const v4i32*>(wq + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dot = __builtin_amdgcn_sdot4(xv[e], wv[e], dot, false);
        sx = __builtin_amdgcn_sdot4(xv[e], 0x01010101, sx, false);
      }
    }
    const int t = dot - wzp * sx + cst;
    const float z = __fmul_rn((float)t, mult);
    if (logits) logits[p] = z;
    if (probs) probs[p] = 1.f / (1.f + __expf(-z));
    if (mask) mask[p] = z > thr ? 255 : 0;
  }
}

// Calibration pass of the float tier: running (min, max) of one activation tensor, as order-preserving unsigned keys
// (atomicMin / atomicMax are exact and order independent).  x: npix pixels of c channels at pixel stride ld.
__device__ __forceinline__ unsigned f32_order_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void minmax_f32_kernel(const float* __restrict__ x, size_t npix, int c, int ld,
                                                         unsigned* __restrict__ keys) {
  unsigned lo = 0xFFFFFFFFu, hi = 0u;
  const size_t total = npix * (size_t)c;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const size_t p = i / c;
    const unsigned k = f32_order_key(x[p * ld + (i - p * c)]);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
  __shared__ unsigned slo[256], shi[256];
  slo[threadIdx.x] = lo;
  shi[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      slo[threadIdx.x] = min(slo[threadIdx.x], slo[threadIdx.x + s]);
      shi[threadIdx.x] = max(shi[threadIdx.x], shi[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMin(&keys[0], slo[0]);
    atomicMax(&keys[1], shi[0]);
  }
}

}  // namespace unet
