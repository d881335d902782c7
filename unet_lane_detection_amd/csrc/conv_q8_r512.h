// 3x3 convolution of the "f16+q8" tier: the split-operand product with its two cross terms on the fp8 matrix pipe.
//
// conv_x3_r512.h forms every product w x as w_hi x_hi + w_hi x_lo + w_lo x_hi with three fp16 MFMAs.  The main term
// carries 22 bits; the two cross terms are 2^-11 of it, so they need only a few significant bits of their own.  Here
//     w x  ~=  w_hi x_hi  (v_mfma_f32_16x16x32_f16, as before)
//            + 2^-5 [ q8(2^8 w_lo) q8(2^-3 x_hi) + q8(2^-3 w_hi) q8(2^8 x_lo) ]   (v_mfma_scale_f32_16x16x128_f8f6f4)
// with q8 = round to nearest OCP fp8 e4m3 (4 significant bits): each cross term is kept to 2^-5 of itself, the product
// to ~2^-16 |w x| instead of f16x3's 2^-22 (tests/dev/mixed_precision_sim.py: max |dlogit| 5e-4 on the reference frame
// against 3e-5; BASELINE.json's tolerance is 1e-3).  One K = 128 fp8 MFMA holds both cross terms of TWO taps of a
// 32-channel chunk (k groups of 32: w_lo x_hi of tap A, w_hi x_lo of tap A, the same for tap B), so a chunk costs per
// accumulator tile 9 fp16 + 5 fp8 MFMAs (the ninth tap has no partner: half of its fp8 MFMA multiplies zeros) where
// f16x3 spends 27 fp16 MFMAs: measured on MI355X from registers (tools/probes/mfma_mix.hip) 2 fp16 + 1 scaled fp8
// take 33.9 ns against 55.9 ns for 6 fp16.  The uniform factor 2^-5 is the instruction's E8M0 scale operand.
//
// Data: the hi plane is the f16x3 tier's; the lo plane's place (same offset, same 2 bytes per element) is taken by
// the "q plane": per pixel and 32-channel block 64 bytes = fp8(x_hi / 8) of the 32 channels, then fp8(256 x_lo)
// (planes_to_q8_kernel below, or a producer's epilogue).  Geometry, LDS-DMA staging, tiles, FLAT instances and the
// epilogue are conv_x3_r512.h's (read that header first); what differs:
//  * the q plane is staged with its own bank swizzle (16-byte part ^ bit 2 of the pixel position): a lane reads 32
//    contiguous bytes of a pixel as two ds_read_b128, conflict free at the same pitches (tools/lds_conflicts.py --q8);
//  * a chunk is 5 steps (tap pairs (0,0)+(1,0), (0,1)+(1,1), (0,2)+(1,2), (2,0)+(2,1), (2,2) alone) of 14 fragments;
//    per step and fragment 4 fp8 + 8 fp16 MFMAs (small terms first, four independent accumulators between dependent
//    MFMAs - the two MFMA kinds are different opcodes and nothing in inline asm pads their dependency);
//  * weights: the fp16 hi fragments come from the f16x3 pack (its lo fragments are not read), the fp8 fragments
//    from a second pack [coTile][chunk][step][cs][half][lane][16 B]; both straight from L2 into a ring of two steps.
//
// Needs Cin % 64 == 0 (chunk pairs keep the ring parity), Cout % 256 == 0, W % 28 == 0 or W == 14.
//
// An accuracy tier of its own (logits within BASELINE.json's 1e-3, not the 2e-4 the f16x3 tier is tested to): off unless
// unet_set_x3_cross_fp8(1) / precision "f16q8" asks for it.  Measurements: profiles/r03/mx_experiments.md; probe:
// tools/probes/conv_mx_r512_probe.hip.
#pragma once
#include "conv_x3_r512.h"

// Timing-only builds (wrong results; tools/probes): bit 0 = no LDS-DMA in the chunk loop, 1 = no weight loads, 2 = no LDS
// reads, 3 = no fp8 MFMAs, 4 = no fp16 MFMAs.  UNET_MX_STAMPS: wave 0 of every block writes its s_memtime / s_memrealtime
// span to ConvX3Args::logits (the clock the chip holds under the kernel).
#ifndef UNET_MX_ABLATE
#define UNET_MX_ABLATE 0
#endif
#ifndef UNET_MX_STAMPS
#define UNET_MX_STAMPS 0
#endif

namespace unet {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// ConvX3Args + the fp8 cross-term weight fragments, [coTile(64)][chunk(32)][step(5)][cs(4)][half(2)][lane(64)][16 bytes];
// `inLo` names the input's q plane (same offset and size as the lo plane)
struct ConvQ8Args : ConvX3Args {
  const uint32_t* wq;
};

constexpr int kQ8HiShift = -3;   // q plane, first half: fp8(x_hi * 2^-3)
constexpr int kQ8LoShift = 8;    // second half: fp8(x_lo * 2^8); both products then carry 2^5
constexpr int kQ8ScaleA = 127 - 5, kQ8ScaleB = 127;   // E8M0 scale bytes of the MFMA (A x B x 2^-5)

// acc += 2^-5 A B, A and B 32 fp8 e4m3 per lane (k group = lane >> 4)
__device__ __forceinline__ void mfma_q8_acc(f32x4& c, const i32x8& a, const i32x8& b, int sa, int sb) {
  if (UNET_MX_ABLATE & 8) return;
#if UNET_MX_ABLATE & 32   // timing only: the same registers read as fp6 e2m3 (the first 24 bytes of each operand)
  typedef int i32x6 __attribute__((ext_vector_type(6)));
  const i32x6 a6 = __builtin_shufflevector(a, a, 0, 1, 2, 3, 4, 5), b6 = __builtin_shufflevector(b, b, 0, 1, 2, 3, 4, 5);
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
               : "+a"(c)
               : "v"(a6), "v"(b6), "v"(sa), "v"(sb));
#else
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
               : "+a"(c)
               : "v"(a), "v"(b), "v"(sa), "v"(sb));
#endif
}

// ---- q plane from the two fp16 planes: one thread per (pixel, 16 channels) ----
__device__ __forceinline__ uint32_t q8_pack4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
__global__ __launch_bounds__(256) void planes_to_q8_kernel(const uint16_t* __restrict__ hi, size_t loOff, size_t nPix,
                                                           int C, int ld, uint8_t* __restrict__ q) {
  // hi / lo planes: pixel stride ld halfs, C channels used (C % 32 == 0); q: pixel stride ld * 2 bytes
  const size_t per = (size_t)(C / 16);
  const float hs = __builtin_ldexpf(1.f, kQ8HiShift), ls = __builtin_ldexpf(1.f, kQ8LoShift);
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nPix * per; t += (size_t)gridDim.x * 256) {
    const size_t p = t / per;
    const int g = (int)(t - p * per);   // 16-channel group
    const uint16_t* src = hi + p * ld + g * 16;
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    const f16x8v h0 = *reinterpret_cast<const f16x8v*>(src), h1 = *reinterpret_cast<const f16x8v*>(src + 8);
    const f16x8v l0 = *reinterpret_cast<const f16x8v*>(src + loOff), l1 = *reinterpret_cast<const f16x8v*>(src + loOff + 8);
    uint4 oh, ol;
    oh.x = q8_pack4((float)h0[0] * hs, (float)h0[1] * hs, (float)h0[2] * hs, (float)h0[3] * hs);
    oh.y = q8_pack4((float)h0[4] * hs, (float)h0[5] * hs, (float)h0[6] * hs, (float)h0[7] * hs);
    oh.z = q8_pack4((float)h1[0] * hs, (float)h1[1] * hs, (float)h1[2] * hs, (float)h1[3] * hs);
    oh.w = q8_pack4((float)h1[4] * hs, (float)h1[5] * hs, (float)h1[6] * hs, (float)h1[7] * hs);
    ol.x = q8_pack4((float)l0[0] * ls, (float)l0[1] * ls, (float)l0[2] * ls, (float)l0[3] * ls);
    ol.y = q8_pack4((float)l0[4] * ls, (float)l0[5] * ls, (float)l0[6] * ls, (float)l0[7] * ls);
    ol.z = q8_pack4((float)l1[0] * ls, (float)l1[1] * ls, (float)l1[2] * ls, (float)l1[3] * ls);
    ol.w = q8_pack4((float)l1[4] * ls, (float)l1[5] * ls, (float)l1[6] * ls, (float)l1[7] * ls);
    uint8_t* dst = q + p * (size_t)ld * 2 + (g >> 1) * 64 + (g & 1) * 16;
    *reinterpret_cast<uint4*>(dst) = oh;
    *reinterpret_cast<uint4*>(dst + 32) = ol;
  }
}

// MaxPool2d(2,2) on planes for the f16q8 tier: x (N,H,W,ldi: the first `c` channels) -> y (N,H/2,W/2,c), as
// maxpool2x2_planes_kernel (same values: the maximum of hi + lo, split again), and the q planes of either tensor while
// their bytes are in registers anyway: srcQ - the input's q plane is written over its lo plane, in place (its remaining
// consumer is a convolution of this tier; a thread owns a whole 32-channel block, whose 64 lo bytes are exactly the
// block's q bytes); dstQ - the output gets its q plane instead of its lo plane.  One thread per output pixel and block.
__global__ __launch_bounds__(256) void maxpool2x2_planes_q8_kernel(uint16_t* __restrict__ src, size_t srcLo, int n, int h,
                                                                   int w, int c, int ldi, uint16_t* __restrict__ dst,
                                                                   size_t dstLo, int srcQ, int dstQ) {
  const int nb = c / 32;
  const size_t total = (size_t)n * (h / 2) * (w / 2) * nb;
  const size_t stride = (size_t)gridDim.x * 256;
  const float hs = __builtin_ldexpf(1.f, kQ8HiShift), ls = __builtin_ldexpf(1.f, kQ8LoShift);
  auto q_block = [&](const uint32_t* ph, const uint32_t* pl, uint4* out) __attribute__((always_inline)) {
    // 16 packed pairs hi, 16 packed pairs lo -> [hi8 x 32][lo8 x 32]
    uint32_t qh[8], ql[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a0, a1, a2, a3, b0, b1, b2, b3;
      merge_pk_f16(ph[2 * e], 0u, a0, a1);
      merge_pk_f16(ph[2 * e + 1], 0u, a2, a3);
      merge_pk_f16(pl[2 * e], 0u, b0, b1);
      merge_pk_f16(pl[2 * e + 1], 0u, b2, b3);
      qh[e] = q8_pack4(a0 * hs, a1 * hs, a2 * hs, a3 * hs);
      ql[e] = q8_pack4(b0 * ls, b1 * ls, b2 * ls, b3 * ls);
    }
    out[0] = make_uint4(qh[0], qh[1], qh[2], qh[3]);
    out[1] = make_uint4(qh[4], qh[5], qh[6], qh[7]);
    out[2] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
    out[3] = make_uint4(ql[4], ql[5], ql[6], ql[7]);
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int b = (int)(i % nb);
    size_t p = i / nb;
    const int xo = (int)(p % (w / 2));
    p /= (w / 2);
    const int yo = (int)(p % (h / 2));
    const int nn = (int)(p / (h / 2));
    float m[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) m[e] = -3.4e38f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint16_t* sp = src + ((((size_t)nn * h + 2 * yo + (d >> 1)) * w + 2 * xo + (d & 1)) * (size_t)ldi + b * 32);
      uint32_t ph[16], pl[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 vh = reinterpret_cast<const uint4*>(sp)[q], vl = reinterpret_cast<const uint4*>(sp + srcLo)[q];
        ph[4 * q] = vh.x, ph[4 * q + 1] = vh.y, ph[4 * q + 2] = vh.z, ph[4 * q + 3] = vh.w;
        pl[4 * q] = vl.x, pl[4 * q + 1] = vl.y, pl[4 * q + 2] = vl.z, pl[4 * q + 3] = vl.w;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float a, bb;
        merge_pk_f16(ph[e], pl[e], a, bb);
        m[2 * e] = fmaxf(m[2 * e], a);
        m[2 * e + 1] = fmaxf(m[2 * e + 1], bb);
      }
      if (srcQ) q_block(ph, pl, reinterpret_cast<uint4*>(sp + srcLo));
    }
    uint32_t oh[16], ol[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) split_pk_f16(m[2 * e], m[2 * e + 1], oh[e], ol[e]);
    uint16_t* dp = dst + ((((size_t)nn * (h / 2) + yo) * (w / 2) + xo) * (size_t)c + b * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      reinterpret_cast<uint4*>(dp)[q] = make_uint4(oh[4 * q], oh[4 * q + 1], oh[4 * q + 2], oh[4 * q + 3]);
    if (dstQ) {
      q_block(oh, ol, reinterpret_cast<uint4*>(dp + dstLo));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        reinterpret_cast<uint4*>(dp + dstLo)[q] = make_uint4(ol[4 * q], ol[4 * q + 1], ol[4 * q + 2], ol[4 * q + 3]);
    }
  }
}

// EPI: 0 = store the two fp16 planes (a.out / a.outLo); 4 = store the hi plane and, in the lo plane's place, the q plane of
// the OUTPUT (the consumer is another convolution of this kind and nothing else reads the tensor: no conversion pass)
template <int TWX_, int EPI, bool FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_q8_r512_kernel(
    const ConvQ8Args a) {
  using S = X3RShape<TWX_>;
  constexpr int TWX = S::TWX, TH = S::TH, P = S::P, NQX = S::NQX, NJ = S::NJ;
  constexpr int NF = S::NPF;   // 14 pixel fragments per wave, 64 channels per wave
  constexpr int FP = S::FP;
  static_assert(FP == 7, "tile widths 28 and 14");
  static_assert(EPI == 0 || EPI == 4, "plane output only");
  static_assert(NJ <= 10, "two piece indices per step");

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;   // coTiles: groups of 256 output channels
  if (lb >= numWork) return;
  const unsigned ldsBase = lds_address(smemv);
  const char* lds = reinterpret_cast<const char*>(smemv);

  // ---- LDS-DMA: pieces q = wave + 4j of both planes.  hrc: halo row << 8 | column; bits 16.. = 4 + (the q plane's
  //      source part - the hi plane's) ----
  int hrc[NJ];
  unsigned soff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;
    const int v = q * 64 + lane;
    const int qpix = v >> 2;
    const int part = (v & 3) ^ (((qpix >> 2) & 1) << 1);
    const int partQ = (v & 3) ^ ((qpix >> 2) & 1);
    const int hr = qpix / P, hc = qpix - hr * P;
    hrc[j] = ((4 + partQ - part) << 16) | (hr << 8) | hc;
    soff[j] = (unsigned)(((hr * a.W + hc) * a.Cin + part * 8) * 2);
  }
  const char* zp = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const size_t inLoB = a.inLo * 2;

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
  auto issue_piece = [&](const Geo& g, int kc, int j, int buf) __attribute__((always_inline)) {
    int q = wave + j * 4;
    q = q < NQX ? q : NQX - 1;
    const int hr = (hrc[j] >> 8) & 255, hc = hrc[j] & 255;
    const int dq = ((hrc[j] >> 16) - 4) * 16;
    const bool ok = (unsigned)(hr - g.hrMin) <= (unsigned)g.hrSpan && (unsigned)(hc - g.hcMin) <= (unsigned)g.hcSpan;
    const char* src = g.tb + soff[j] + (unsigned)(kc * 64);
    const unsigned dst = ldsBase + buf * S::XST + q * 1024;
    lds_dma16(ok ? src : zp, dst);
    lds_dma16(ok ? src + inLoB + dq : zp, dst + S::XPL);
  };

  // ---- LDS read side.  xb: byte position of this lane's pixel of fragment 7 * (f / 7) + f7 (no lane part).
  //      fp16 plane: + lq * 16 (8 channels per lane).  q plane: + (lq & 1) * 32 (first half: x_hi, second: x_lo);
  //      lanes lq >= 2 read the step's second tap ----
  int xb[FP];
#pragma unroll
  for (int f7 = 0; f7 < FP; ++f7) {
    const int i = 16 * f7 + li;
    const int r = i / TWX, c = i - r * TWX;
    xb[f7] = (r * P + c) * 64;
  }
  const int lq16 = lq * 16;
  const int qHalf = (lq & 1) * 32;
  const bool tapB = lq >= 2;
  const int qOffV = qHalf + (tapB ? P * 64 : 0);   // vertical pair: tap B one row down
  const int qOffH = qHalf + (tapB ? 64 : 0);       // horizontal pair: tap B one column right

  // ---- weights ----
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wt), 0, (a.Cout / 64) * a.chunksTotal * (9 * 2 * 4 * 1024), 0x00020000);
  const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(a.wq), 0, (a.Cout / 64) * a.chunksTotal * (5 * 4 * 2 * 1024), 0x00020000);
  const int laneW = lane * 16;
  auto w_block = [&](int cg, int kc) __attribute__((always_inline)) -> int {
    return ((cg * 4 + wave) * a.chunksTotal + kc);   // (channel tile of 64, chunk)
  };
  auto wh_load = [&](int blk, int tap, int cs) __attribute__((always_inline)) -> f32x4 {
    const int ky = tap / 3, kx = tap - ky * 3;
    const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  wrsrc, laneW + cs * 1024, blk * (9 * 2 * 4 * 1024) + ((ky * 2) * 3 + kx) * 4096, 0));
    return __builtin_bit_cast(f32x4, v);
  };
  auto wq_load = [&](int blk, int step, int cs, int half) __attribute__((always_inline)) -> i32x4 {
    return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         qrsrc, laneW + (cs * 2 + half) * 1024, blk * (5 * 4 * 2 * 1024) + step * 8192, 0));
  };
  // taps of a step
  auto tapA_of = [](int s) { return s < 3 ? s : (s == 3 ? 6 : 8); };
  auto tapB_of = [](int s) { return s < 3 ? 3 + s : 7; };   // s == 4: none

  if (tid < 8)
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smemv) + S::ZOFF + (tid >> 2) * S::XPL + (tid & 3) * 16) =
        (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: chunk 0 of the first item, weights of its step 0 ----
  Geo gCur = geo_of(lb);
  i32x8 wq[2][4];       // ring over steps
  f32x4 wh[2][2][4];    // [ring][tap A / B][cs]
  {
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue_piece(gCur, 0, j, 0);
    const int blk = w_block(gCur.cg, 0);
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      const i32x4 l = wq_load(blk, 0, cs, 0), h = wq_load(blk, 0, cs, 1);
      wq[0][cs] = (i32x8){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
      wh[0][0][cs] = wh_load(blk, tapA_of(0), cs);
      wh[0][1][cs] = wh_load(blk, tapB_of(0), cs);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int scA = kQ8ScaleA, scB = kQ8ScaleB;
  asm volatile("" : "+v"(scA), "+v"(scB));
  int cc = 0;
  float amax = 0.f;
#if UNET_MX_STAMPS
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
    const int cbase = (gCur.cg * 4 + wave) * 64 + lq * 16;

    // FLAT: bit f = "tap row 0 of this lane's pixel of fragment f is inside the image", bit 16 + f = "tap row 2 is"
    unsigned keep = 0xFFFFFFFFu;
    if (FLAT) {
      const int y0m = gCur.y0 % a.imgH;
      keep = 0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int i = 16 * f + li;
        int yy = y0m + i / TWX;
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        yy = yy >= a.imgH ? yy - a.imgH : yy;
        keep |= (yy != 0 ? 1u : 0u) << f;
        keep |= (yy != a.imgH - 1 ? 1u : 0u) << (16 + f);
      }
    }
    // q plane, vertical pairs: lanes of tap B read tap row 1, which is always inside
    const unsigned keepQV = tapB ? 0xFFFFFFFFu : keep;

    for (int kc = 0; kc < a.nChunks; ++kc, ++cc) {
      const bool lastChunk = kc + 1 == a.nChunks;
      const bool haveNext = !(lastChunk && lastItem);
      const Geo& gIss = lastChunk ? gNext : gCur;
      const int kcIss = lastChunk ? (lastItem ? kc : 0) : kc + 1;
      const int wCur = w_block(gCur.cg, kc);
      const int wNxt = haveNext ? w_block(gIss.cg, kcIss) : wCur;
      const int bufOff = (cc & 1) * S::XST;
      const int nbuf = (cc + 1) & 1;

      int xc[FP];
#pragma unroll
      for (int f7 = 0; f7 < FP; ++f7) {
        xc[f7] = xb[f7] + bufOff;
        asm volatile("" : "+v"(xc[f7]));
      }
      // FLAT: the row masks' bit tests are redone where they are used (hoisted out of the chunk loop as 42 lane masks
      // they are spilled scalar registers, reloaded by v_readlane in the MFMA loop)
      unsigned keepC = keep, keepQC = keepQV;
      if (FLAT) asm volatile("" : "+v"(keepC), "+v"(keepQC));
      // ---- operand addresses.  Fragments are visited in the order 0, 7, 1, 8, ...: fragments f7 and f7 + 7 sit FRAGP
      //      bytes apart (an immediate of the read), so one address computation serves two fragments ----
      auto frag_of = [](int k) { return (k >> 1) + (k & 1) * 7; };
      // fp16 plane, tap, period fragment f7 (without FRAGP, before the FLAT select)
      auto h_base = [&](int tap, int f7) __attribute__((always_inline)) -> int {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int b = xc[f7] + lq16 + (ky * P + kx) * 64;
        return b ^ ((b >> 3) & 32);
      };
      auto h_sel = [&](int base, int tap, int f) __attribute__((always_inline)) -> int {
        const int ky = tap / 3;
        int addr = base + (f / FP) * S::FRAGP;
        if (FLAT && ky != 1) {
          const bool kp = (keepC >> ((ky == 0 ? 0 : 16) + f)) & 1u;
          addr = kp ? addr : S::ZOFF;
        }
        return addr;
      };
      // q plane (relative to the fp16 plane's buffer; + XPL at the read), step s: first 16 bytes; the second 16 are
      // at ^ 16
      auto q_base = [&](int s, int f7) __attribute__((always_inline)) -> int {
        const int tA = s < 3 ? s : (s == 3 ? 6 : 8);
        const int ky = tA / 3, kx = tA - ky * 3;
        const int b = xc[f7] + (s < 3 ? qOffV : (s == 3 ? qOffH : qHalf)) + (ky * P + kx) * 64;
        return b ^ ((b >> 4) & 16);
      };
      auto q_sel = [&](int base, int s, int f) __attribute__((always_inline)) -> int {
        int addr = base + (f / FP) * S::FRAGP;
        if (FLAT) {
          const bool kp = s < 3 ? (keepQC >> f) & 1u : (keepC >> (16 + f)) & 1u;
          addr = kp ? addr : S::ZOFF;
        }
        if (s == 4) addr = tapB ? S::ZOFF : addr;   // no partner tap: A holds zeros there, B must be finite
        return addr;
      };

      i32x8 xq[2];
      f32x4 xa[2], xt[2];   // fp16 operands of tap A and tap B, ring over (step, fragment)
      int bq = q_base(0, 0), ba = h_base(tapA_of(0), 0), bb = h_base(tapB_of(0), 0);
      {
        const int qa = q_sel(bq, 0, 0);
        const i32x4 l = *reinterpret_cast<const i32x4*>(lds + S::XPL + qa);
        const i32x4 h = *reinterpret_cast<const i32x4*>(lds + S::XPL + (qa ^ 16));
        xq[0] = (i32x8){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
        xa[0] = *reinterpret_cast<const f32x4*>(lds + h_sel(ba, tapA_of(0), 0));
        xt[0] = *reinterpret_cast<const f32x4*>(lds + h_sel(bb, tapB_of(0), 0));
      }
#define MX_GAP __builtin_amdgcn_sched_barrier(0)
#define MXH(c, a, b)                    \
  if (!(UNET_MX_ABLATE & 16)) mfma_x3_acc(c, a, b)
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int rs = s & 1, rn = rs ^ 1;   // weight ring: this step, the next
        const int sN = s < 4 ? s + 1 : 0;    // the step whose weights are fetched now
        const int wBlkN = s < 4 ? wCur : wNxt;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          const int f = frag_of(k);
          const int L = s * NF + k;
          const int cur = L & 1, nxt = cur ^ 1;
          const bool pre = k + 1 < NF || s < 4;   // not across the chunk's end
          const int pk = k + 1 < NF ? k + 1 : 0, ps = k + 1 < NF ? s : s + 1;
          const int pf = frag_of(pk);
          const bool fresh = (pk & 1) == 0;   // a new period fragment: new addresses
          const bool preB = pre && ps < 4;
          int qa = 0, ha = 0, hb = 0;
          // This step's weights are all waited for in its first fragment; the LDS-DMA of the next chunk goes out in
          // the second (its loads are invisible to the compiler's vmcnt counts: issued between a weight load and its
          // first use they would be waited for in full), the next step's weights behind it, two per fragment
          // small terms first; dependent MFMAs are four apart
          mfma_q8_acc(acc[f][0], wq[rs][0], xq[cur], scA, scB);
          MX_GAP;
          if (pre && fresh) bq = q_base(ps, pf);
          MX_GAP;
          mfma_q8_acc(acc[f][1], wq[rs][1], xq[cur], scA, scB);
          MX_GAP;
          if (pre) qa = q_sel(bq, ps, pf);
          MX_GAP;
          mfma_q8_acc(acc[f][2], wq[rs][2], xq[cur], scA, scB);
          MX_GAP;
          if (pre && !(UNET_MX_ABLATE & 4)) {
            const i32x4 l = *reinterpret_cast<const i32x4*>(lds + S::XPL + qa);
            const i32x4 h = *reinterpret_cast<const i32x4*>(lds + S::XPL + (qa ^ 16));
            xq[nxt] = (i32x8){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
          }
          MX_GAP;
          mfma_q8_acc(acc[f][3], wq[rs][3], xq[cur], scA, scB);
          MX_GAP;
          if (pre && fresh) ba = h_base(tapA_of(ps), pf);
          MX_GAP;
          MXH(acc[f][0], wh[rs][0][0], xa[cur]);
          MX_GAP;
          if (pre) ha = h_sel(ba, tapA_of(ps), pf);
          if (pre && !(UNET_MX_ABLATE & 4)) xa[nxt] = *reinterpret_cast<const f32x4*>(lds + ha);
          MX_GAP;
          MXH(acc[f][1], wh[rs][0][1], xa[cur]);
          MX_GAP;
          if (preB && fresh) bb = h_base(tapB_of(ps), pf);
          MX_GAP;
          MXH(acc[f][2], wh[rs][0][2], xa[cur]);
          MX_GAP;
          if (preB) hb = h_sel(bb, tapB_of(ps), pf);
          if (preB && !(UNET_MX_ABLATE & 4)) xt[nxt] = *reinterpret_cast<const f32x4*>(lds + hb);
          MX_GAP;
          MXH(acc[f][3], wh[rs][0][3], xa[cur]);
          MX_GAP;
          if (k >= 2 && k < 10 && !(UNET_MX_ABLATE & 2)) {   // weight fragments 2 (k - 2) and 2 (k - 2) + 1 of the next step
            const int i0 = 2 * (k - 2);
            if (i0 < 8) {
              const i32x4 t = wq_load(wBlkN, sN, i0 >> 1, 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) wq[rn][i0 >> 1][e] = t[e];
            } else if (i0 < 12) {
              wh[rn][0][i0 - 8] = wh_load(wBlkN, tapA_of(sN), i0 - 8);
            } else if (sN < 4) {
              wh[rn][1][i0 - 12] = wh_load(wBlkN, tapB_of(sN), i0 - 12);
            }
          }
          if (k == 1 && 2 * s < NJ && !(UNET_MX_ABLATE & 1)) issue_piece(gIss, kcIss, 2 * s, nbuf);
          if (s < 4) {
            MX_GAP;
            MXH(acc[f][0], wh[rs][1][0], xt[cur]);
            MX_GAP;
          }
          if (k >= 2 && k < 10 && !(UNET_MX_ABLATE & 2)) {
            const int i1 = 2 * (k - 2) + 1;
            if (i1 < 8) {
              const i32x4 t = wq_load(wBlkN, sN, i1 >> 1, 1);
#pragma unroll
              for (int e = 0; e < 4; ++e) wq[rn][i1 >> 1][4 + e] = t[e];
            } else if (i1 < 12) {
              wh[rn][0][i1 - 8] = wh_load(wBlkN, tapA_of(sN), i1 - 8);
            } else if (sN < 4) {
              wh[rn][1][i1 - 12] = wh_load(wBlkN, tapB_of(sN), i1 - 12);
            }
          }
          if (k == 1 && 2 * s + 1 < NJ && !(UNET_MX_ABLATE & 1)) issue_piece(gIss, kcIss, 2 * s + 1, nbuf);
          if (s < 4) {
            MX_GAP;
            MXH(acc[f][1], wh[rs][1][1], xt[cur]);
            MXH(acc[f][2], wh[rs][1][2], xt[cur]);
            MXH(acc[f][3], wh[rs][1][3], xt[cur]);
          }
          MX_GAP;
        }
      }
#undef MX_GAP
#undef MXH
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // step 4 fetched the next chunk's step 0 into ring slot 1; step 0 reads slot 0
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
        wq[0][cs] = wq[1][cs];
        wh[0][0][cs] = wh[1][0][cs];
        wh[0][1][cs] = wh[1][1][cs];
        asm volatile("" : "+v"(wq[0][cs]), "+v"(wh[0][0][cs]), "+v"(wh[0][1][cs]));
      }
    }

    // ---- epilogue (conv_x3_r512.h, EPI 0) ----
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(a.scale + cbase + cs * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(a.shift + cbase + cs * 4);
    }
    if (a.dynScale) {
      const float ds = *a.dynScale;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) sc[cs] *= ds;
    }
    const float floorV = a.relu ? 0.f : -3.4e38f;
    const size_t g0 = (size_t)gCur.n * a.H + gCur.y0;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int liE = li;
    asm volatile("" : "+v"(liE));
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int i = 16 * f + liE;
      const int r = i / TWX, c = i - r * TWX;
      const bool ok = gCur.y0 + r < a.H;
      const size_t pix = (g0 + r) * a.W + gCur.x0 + c;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        v[e] = fmaxf(fmaf(acc[f][e >> 2][e & 3], sc[e >> 2][e & 3], sh[e >> 2][e & 3]), floorV);
      uint32_t ph[8], pl[8];
      uint32_t qh[4], ql[4];
      if (EPI == 4) {
        // hi plane as always; the q plane's bytes exactly as planes_to_q8_kernel would make them from (hi, lo):
        // fp8(hi / 8) and fp8(256 * rn16(v - hi))
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const float hs = __builtin_ldexpf(1.f, kQ8HiShift), ls = __builtin_ldexpf(1.f, kQ8LoShift);
        float hf[16], lf[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          amax3(amax, v[2 * e], v[2 * e + 1]);
          split_pk_f16_mix(v[2 * e], v[2 * e + 1], ph[e], pl[e]);
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
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          amax3(amax, v[2 * e], v[2 * e + 1]);
          split_pk_f16_mix(v[2 * e], v[2 * e + 1], ph[e], pl[e]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        auto r1 = __builtin_amdgcn_permlane16_swap(ph[k], ph[4 + k], false, false);
        auto q1 = __builtin_amdgcn_permlane32_swap(r1[0], r1[1], false, false);
        ph[k] = q1[0];
        ph[4 + k] = q1[1];
        if (EPI == 0) {
          auto rl = __builtin_amdgcn_permlane16_swap(pl[k], pl[4 + k], false, false);
          auto q2 = __builtin_amdgcn_permlane32_swap(rl[0], rl[1], false, false);
          pl[k] = q2[0];
          pl[4 + k] = q2[1];
        } else {
          // lanes lq = 0, 1 hold the two halves of the wave's first 32-channel block, lq = 2, 3 of its second; a block
          // is [x_hi h0 | x_hi h1 | x_lo h0 | x_lo h1] x 16 bytes.  Swapping the upper half wave of qh with the lower
          // half wave of ql leaves block 0 complete in qh (lane row lq = its 16-byte piece lq) and block 1 in ql
          auto sw = __builtin_amdgcn_permlane32_swap(qh[k], ql[k], false, false);
          qh[k] = sw[0];
          ql[k] = sw[1];
        }
      }
      uint16_t* rowp = a.out + pix * (size_t)a.ldo + a.co_off + (cbase - lq * 16) + lq * 8;
      if (ok) {
        *reinterpret_cast<uint4*>(rowp) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        *reinterpret_cast<uint4*>(rowp + 32) = make_uint4(ph[4], ph[5], ph[6], ph[7]);
        if (EPI == 0) {
          *reinterpret_cast<uint4*>(rowp + a.outLo) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          *reinterpret_cast<uint4*>(rowp + a.outLo + 32) = make_uint4(pl[4], pl[5], pl[6], pl[7]);
        } else {
          // q plane: same offset and pixel stride (in bytes) as the lo plane; the wave's 64 channels are 128 bytes of it
          uint16_t* qp = a.out + a.outLo + pix * (size_t)a.ldo + a.co_off + (cbase - lq * 16) + lq * 8;
          *reinterpret_cast<uint4*>(qp) = make_uint4(qh[0], qh[1], qh[2], qh[3]);
          *reinterpret_cast<uint4*>(qp + 32) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    gCur = gNext;
  }
  x3_report_range(amax, a.err);
#if UNET_MX_STAMPS
  if (tid == 0) {
    unsigned long long* st = reinterpret_cast<unsigned long long*>(a.logits) + (size_t)blockIdx.x * 2;
    st[0] = __builtin_amdgcn_s_memtime() - tStart;
    st[1] = __builtin_amdgcn_s_memrealtime() - rStart;
  }
#endif
}

}  // namespace unet
