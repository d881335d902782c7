// First convolution of the split-operand (f16x3) tier: uint8 RGB frames -> normalise -> 3x3 conv (3 -> Cout) +
// folded BN + ReLU -> fp16 hi/lo planes, in one kernel (conv_first_bf16x3.h with fp16 operands and a split output).
//
// K = 27 fits ONE v_mfma_f32_16x16x32_f16 (k = tap*3 + ci, 27..31 zero); both operands are split into fp16 hi + lo
// and three MFMAs are issued per tile: the layer is exact to ~2^-22 like the rest of the tier.  Bound by its stores
// (2 x 128 B per pixel), not by the 12 MFMAs per 16 pixels.
//
// Block = 256 threads, tile = 8 rows x 32 columns of one image (tiles are per image; rows past the bottom are not
// stored).  Stage 1: the uint8 halo (10 x 34 x 3 bytes) -> normalised fp32 in LDS (zero outside the image: padding
// applies to the normalised tensor, reference README.md:1427).  Stage 2: thread p builds pixel p's im2col row as
// fp16 hi | lo (2 x 64 bytes in LDS).  Stage 3: wave w multiplies pixels [64w, 64w+64): weights are the MFMA A
// operand with the channel permutation of conv_bf16_ws.h, so a lane ends up with 16 consecutive channels of one pixel.
#pragma once
#include "conv_x3_ws.h"

namespace unet {

struct ConvFirstX3Args {
  const void* frames;      // U8: (N,H,W,3) uint8, normalised here; else (N,3,H,W) float32, already normalised
  const uint16_t* wt;      // [coTile(64 ch)][cs(4)][hi|lo][lane(64)][8] fp16 (pre-scaled per channel)
  const float* scale;      // folded BN scale / weight pre-scale
  const float* shift;
  uint16_t* out;           // hi plane (N,H,W,ldo), channels [0,Cout); lo plane at out + outLo
  size_t outLo;
  int N, H, W, Cout, ldo, tilesX, tilesY, relu;
  float m0, m1, m2, s0, s1, s2;   // (x - m) / s as pack_u8_nhwc4_kernel
  unsigned* err;                  // error block (may be null): word 1 = an activation left the fp16 range
};

template <bool U8>
__global__ __launch_bounds__(256) void conv_first_x3_kernel(const ConvFirstX3Args a) {
  constexpr int TH = 8, TW = 32, HR = TH + 2, HC = TW + 2;
  __shared__ float halo[HR * HC * 3 + 4];
  __shared__ __attribute__((aligned(16))) uint32_t rows[2][256][16 + 4];   // [hi|lo][pixel][32 fp16 (+16 B pad)]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int tile = blockIdx.x;
  const int rowTile = tile / a.tilesX;
  const int x0 = (tile - rowTile * a.tilesX) * TW;
  const int n = rowTile / a.tilesY;
  const int y0 = (rowTile - n * a.tilesY) * TH;
  const size_t g0 = (size_t)n * a.H + y0;

  // the first channel tile's weights and constants are asked for first of all: they land under stages 1 and 2
  f32x4 wh[4], wl[4], sc[4], sh[4];
  auto load_ct = [&](int ct) __attribute__((always_inline)) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wt) + (size_t)ct * (4 * 2 * 64) + lane;
    const int cb = ct * 64 + lq * 16;
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      wh[cs] = wp[(cs * 2 + 0) * 64];
      wl[cs] = wp[(cs * 2 + 1) * 64];
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cb + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cb + cs * 4);
    }
  };
  load_ct(0);

  // ---- stage 1: normalised halo.  All of a thread's loads go out before the first is used: as a plain loop hipcc issued
  //      one byte load, waited vmcnt(0), converted and stored, four round trips to memory per block in a row - most of the
  //      kernel's time (round 4, found with the int8 tier's stamps; 0.98 -> 0.7x ms at batch 256) ----
  {
    constexpr int N1 = (HR * HC * 3 + 255) / 256;
    float rawv[N1];       // fp32 input
    unsigned rawb[N1];    // uint8 input: the bytes as loaded - converting here would put the wait right behind each load
    bool inb[N1];
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int i = it * 256 + tid;
      const int px = i / 3, ci = i - px * 3;
      const int hr = px / HC, hc = px - hr * HC;
      const int y = y0 - 1 + hr, x = x0 - 1 + hc;
      inb[it] = i < HR * HC * 3 && y >= 0 && y < a.H && x >= 0 && x < a.W;
      rawv[it] = 0.f;
      rawb[it] = 0;
      if (inb[it]) {
        if (U8)
          rawb[it] = static_cast<const uint8_t*>(a.frames)[((g0 + hr - 1) * a.W + x) * 3 + ci];
        else
          rawv[it] = static_cast<const float*>(a.frames)[(((size_t)n * 3 + ci) * a.H + y) * a.W + x];
      }
    }
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int i = it * 256 + tid;
      const int ci = i % 3;
      float v = 0.f;
      if (inb[it]) {
        if (U8) {
          const float m = ci == 0 ? a.m0 : (ci == 1 ? a.m1 : a.m2);
          const float s = ci == 0 ? a.s0 : (ci == 1 ? a.s1 : a.s2);
          v = ((float)rawb[it] - m) / s;
        } else {
          v = rawv[it];
        }
      }
      if (i < HR * HC * 3) halo[i] = v;
    }
  }
  __syncthreads();

  // ---- stage 2: im2col row of pixel tid: k = tap*3 + ci, split into fp16 hi and lo ----
  {
    const int r = tid / TW, c = tid - r * TW;
    uint32_t hi[16], lo[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * k2 + e;
        if (k < 27) {
          const int t = k / 3, ci = k - t * 3;
          v[e] = halo[((r + t / 3) * HC + c + t % 3) * 3 + ci];
        } else {
          v[e] = 0.f;
        }
      }
      split_pk_f16(v[0], v[1], hi[k2], lo[k2]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<uint4*>(&rows[0][tid][q * 4]) = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
      *reinterpret_cast<uint4*>(&rows[1][tid][q * 4]) = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
    }
  }
  __syncthreads();

  // ---- stage 3: MFMA, 64 channels at a time ----
  const int nCt = a.Cout / 64;
  for (int ct = 0; ct < nCt; ++ct) {
    if (ct > 0) load_ct(ct);
    const int cbase = ct * 64 + lq * 16;
    const float lo0 = a.relu ? 0.f : -3.4e38f;
    float amax = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int p = wave * 64 + f * 16 + li;
      const f32x4 xh = *reinterpret_cast<const f32x4*>(&rows[0][p][lq * 4]);
      const f32x4 xl = *reinterpret_cast<const f32x4*>(&rows[1][p][lq * 4]);
      uint32_t ph[8], pl[8];
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[cs]), __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[cs]), __builtin_bit_cast(f16x8, xl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[cs]), __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc[r], sc[cs][r], sh[cs][r]), lo0);
        split_pk_f16(v[0], v[1], ph[cs * 2], pl[cs * 2], amax);
        split_pk_f16(v[2], v[3], ph[cs * 2 + 1], pl[cs * 2 + 1], amax);
      }
      const int r = p / TW, c = p - r * TW;
      const int x = x0 + c;
      if (x < a.W && y0 + r < a.H) {
        uint16_t* op = a.out + ((g0 + r) * a.W + x) * (size_t)a.ldo + cbase;
        // non-temporal: the 3.3 GB of planes are next read a whole kernel later, and lines kept in L2 only displace
        // others (1.023 -> 0.983 ms at batch 256; the same hint on the transposed convolutions' stores cost 5 - 23 %:
        // profiles/r04/t448_experiments.md)
        typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
        u32x4nt* o = reinterpret_cast<u32x4nt*>(op);
        __builtin_nontemporal_store((u32x4nt){ph[0], ph[1], ph[2], ph[3]}, o);
        __builtin_nontemporal_store((u32x4nt){ph[4], ph[5], ph[6], ph[7]}, o + 1);
        u32x4nt* ol = reinterpret_cast<u32x4nt*>(op + a.outLo);
        __builtin_nontemporal_store((u32x4nt){pl[0], pl[1], pl[2], pl[3]}, ol);
        __builtin_nontemporal_store((u32x4nt){pl[4], pl[5], pl[6], pl[7]}, ol + 1);
      }
    }
    x3_report_range(amax, a.err);
  }
}

}  // namespace unet
