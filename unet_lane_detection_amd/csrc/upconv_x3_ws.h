// ConvTranspose2d(kernel 2, stride 2) of the split-operand (f16x3) tier (reference README.md:1442, :1476):
// y[2i+a][2j+b][co] = sum_ci x[i][j][ci] * W[ci][co][a][b] + bias[co], written into the upper channel slice of the
// concat buffer, both fp16 planes.  Arithmetic as conv_x3_ws.h (hi/lo operands, three fp16 MFMAs per product, fp32
// accumulate); structure as upconv_bf16_ws.h with 32-channel stages so that both planes of both operands fit LDS
// double buffered: per stage 2 x 8 KiB of pixels and 2 x 16 KiB of weights.
//
//  * a work item is 128 consecutive input pixels (flattened n,y,x) x 64 output channels x all four (a,b);
//  * MFMA wave w owns pixels [32w, 32w+32) as two fragments and all 4 x 64 outputs: 32 accumulators; per stage 96
//    MFMAs for 4 pixel-fragment and 32 weight-fragment ds_read_b128;
//  * pixels are 64 bytes apart in LDS; the 16-byte part is XOR-swizzled on the DMA source side exactly as the halo
//    tile of conv_x3_ws.h (part ^= 2 * bit 2 of the pixel index).
// The layer is bound by its stores (4x the input bytes, two planes).
#pragma once
#include "conv_x3_ws.h"

namespace unet {

struct UpconvX3Args {
  const uint16_t* in;     // hi plane (N,h,w,Cin) fp16; lo plane at in + inLo
  size_t inLo;
  const uint16_t* wt;     // packed [coTile(64)][chunk(32)][plane(2)][ab(4)][cs(4)][lane][8]
  const uint16_t* zeros;  // >= 64 zero halfs
  const float* scale;     // [Cout] 2^-k of the weights' per-channel pre-scale
  const float* bias;      // [Cout]
  uint16_t* out;          // hi plane (N,2h,2w,ldo), channels [co_off, co_off + Cout); lo plane at out + outLo
  size_t outLo;
  long npix;              // N*h*w
  int h, w, Cin, Cout, ldo, co_off, nChunks;   // nChunks = Cin / 32 (even)
  int coTiles, pixTiles;
  // MODE 1 (plain 1x1 GEMM, fp32 output): outF[p][co_off + n] = dyn * sum_k in[p][k] W[k][n], n < Cout; a channel
  // tile is 256 columns = the four 64-column groups that MODE 0 scatters to the four (a,b) positions
  float* outF;
  const float* dynScale;   // optional device scalar (undoes the power-of-two scaling of a gradient input)
  // MODE 0, small batches: abSplit = 4 makes one work item per (pixel tile, channel tile, (a,b)) - four times the items,
  // each with a quarter of the weights and MFMAs (a single frame's 14x14 -> 28x28 layer is otherwise 16 items)
  int abSplit;
  unsigned* err;   // error block (may be null): word 1 = an activation left the fp16 range
};

struct UpconvX3Shape {
  static constexpr int TP = 128;
  static constexpr int XPL = TP * 64, XST = 2 * XPL;          // 8 KiB per plane
  static constexpr int WPL = 16 * 1024, WST = 2 * WPL;        // 4 (a,b) x 4 subtiles x 1 KiB per plane
  static constexpr int WOFF = 0, XOFF = 2 * WST, TOFF = XOFF + 2 * XST;
  static constexpr int MAX_COUT = 1024;
  static constexpr int LDS_BYTES = TOFF + 2 * MAX_COUT * 4;   // 106,496
};

template <int MODE>
__global__ __launch_bounds__(512, 1) void upconv2x2_x3_ws_kernel(const UpconvX3Args a) {
  using S = UpconvX3Shape;
  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles * a.abSplit;   // consecutive items: the (a,b) / channel tiles of one pixel tile
  const int tilesMine = lb < numWork ? (numWork - lb + G - 1) / G : 0;
  const int totalStages = tilesMine * a.nChunks;

  if (wave >= 4) {
    // ---------------- loader waves: wave 4+k issues pixel pieces k, k+4 of both planes and weight pieces k + 4j ----
    const int k = wave - 4;
    int wN = lb, kcN = 0;
    for (int i = 0; i <= totalStages; ++i) {
      if (i < totalStages) {
        const int abSelN = a.abSplit == 4 ? (wN & 3) : -1;
        const int wN2 = a.abSplit == 4 ? (wN >> 2) : wN;
        const int tile = wN2 / a.coTiles, coTile = wN2 - tile * a.coTiles;
        const long p0 = (long)tile * S::TP;
        char* xdst = reinterpret_cast<char*>(smemv) + S::XOFF + (i & 1) * S::XST;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = k + 4 * j;                      // 1 KiB piece: 16 pixels x 64 bytes
          const int px = q * 16 + (lane >> 2);
          const int part = (lane & 3) ^ (((px >> 2) & 1) << 1);
          const bool ok = p0 + px < a.npix;
          const uint16_t* src = ok ? a.in + (size_t)(p0 + px) * (size_t)a.Cin + kcN * 32 + part * 8 : a.zeros + part * 8;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(xdst + q * 1024), 16, 0, 0);
          const uint16_t* srcLo = ok ? src + a.inLo : src;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcLo,
                                           (__attribute__((address_space(3))) void*)(xdst + S::XPL + q * 1024), 16, 0, 0);
        }
        {
          char* wdst = reinterpret_cast<char*>(smemv) + S::WOFF + (i & 1) * S::WST;
          const uint16_t* wsrc = a.wt + ((size_t)coTile * a.nChunks + kcN) * (size_t)(S::WST / 2) + lane * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) {   // 32 pieces: [plane][ab][cs]
            if (abSelN >= 0 && (((k + 4 * j) >> 2) & 3) != abSelN) continue;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wsrc + (k + 4 * j) * 512),
                (__attribute__((address_space(3))) void*)(wdst + (k + 4 * j) * 1024), 16, 0, 0);
          }
        }
        if (++kcN == a.nChunks) {
          kcN = 0;
          wN += G;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int li = lane & 15, lq = lane >> 4;
  const char* lds = reinterpret_cast<const char*>(smemv);
  if (MODE == 0) {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smemv) + S::TOFF);
    for (int c = tid; c < a.Cout; c += 256) {
      tab[c] = a.scale[c];
      tab[S::MAX_COUT + c] = a.bias[c];
    }
  }
  const float dyn = (MODE == 1 && a.dynScale) ? *a.dynScale : 1.f;
  int xa[2];   // this lane's 16 bytes of pixel fragment ms (stage buffer 0, hi plane)
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const int px = wave * 32 + ms * 16 + li;
    xa[ms] = S::XOFF + px * 64 + ((lq ^ (((px >> 2) & 1) << 1)) << 4);
    asm volatile("" : "+v"(xa[ms]));
  }
  int wa = S::WOFF + lane * 16;
  asm volatile("" : "+v"(wa));
  ws_barrier();
  int stage = 0;
  float amax = 0.f;
  for (int w = lb; w < numWork; w += G) {
    const int abSel = (MODE == 0 && a.abSplit == 4) ? (w & 3) : -1;
    const int w2 = (MODE == 0 && a.abSplit == 4) ? (w >> 2) : w;
    const int tile = w2 / a.coTiles, coTile = w2 - tile * a.coTiles;
    // MODE 1: column groups of this tile that exist (wave-uniform; the packed weights of the others are zero)
    const int nAb = MODE == 0 ? 4 : ((a.Cout - coTile * 256) >= 256 ? 4 : (a.Cout - coTile * 256) / 64);
    f32x4 acc[2][4][4];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab)
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[ms][ab][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < a.nChunks; ++kc, ++stage) {
      const int xoff = (stage & 1) * S::XST, woff = (stage & 1) * S::WST;
      f32x4 xh[2], xl[2];
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        xh[ms] = *reinterpret_cast<const f32x4*>(lds + xa[ms] + xoff);
        xl[ms] = *reinterpret_cast<const f32x4*>(lds + xa[ms] + xoff + S::XPL);
      }
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        if (MODE == 1 && ab >= nAb) continue;
        if (MODE == 0 && abSel >= 0 && ab != abSel) continue;
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) {
          const f32x4 wh = *reinterpret_cast<const f32x4*>(lds + wa + woff + (ab * 4 + cs) * 1024);
          const f32x4 wl = *reinterpret_cast<const f32x4*>(lds + wa + woff + S::WPL + (ab * 4 + cs) * 1024);
#pragma unroll
          for (int ms = 0; ms < 2; ++ms) {
            acc[ms][ab][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl),
                                                                     __builtin_bit_cast(f16x8, xh[ms]),
                                                                     acc[ms][ab][cs], 0, 0, 0);
            acc[ms][ab][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh),
                                                                     __builtin_bit_cast(f16x8, xl[ms]),
                                                                     acc[ms][ab][cs], 0, 0, 0);
            acc[ms][ab][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh),
                                                                     __builtin_bit_cast(f16x8, xh[ms]),
                                                                     acc[ms][ab][cs], 0, 0, 0);
          }
        }
      }
      ws_barrier();
    }

    if (MODE == 1) {
      // ---- plain epilogue: lane (li, lq) holds columns 256*coTile + 64*ab + 16*lq + [0,16) of pixel li ----
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        const long p = (long)tile * S::TP + wave * 32 + ms * 16 + li;
        if (p >= a.npix) continue;
        float* orow = a.outF + (size_t)p * (size_t)a.ldo + a.co_off + coTile * 256 + lq * 16;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
          if (ab >= nAb) continue;
#pragma unroll
          for (int cs = 0; cs < 4; ++cs) {
            f32x4 v = acc[ms][ab][cs];
            v *= dyn;
            *reinterpret_cast<f32x4*>(orow + ab * 64 + cs * 4) = v;
          }
        }
      }
      continue;
    }

    // ---- epilogue: lane (li, lq) holds channels 64*coTile + 16*lq + [0,16) of input pixel li of each fragment, for
    //      each (a,b): * 2^-k + bias, split, 2 x 32-byte stores per plane to output pixel (2y+a, 2x+b) ----
    const int cbase = coTile * 64 + lq * 16;
    f32x4 sc[4], bi[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (cbase + cs * 4) * 4);
      bi[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (S::MAX_COUT + cbase + cs * 4) * 4);
    }
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const long p = (long)tile * S::TP + wave * 32 + ms * 16 + li;
      const bool ok = p < a.npix;
      const long pc = ok ? p : 0;
      const int x = (int)(pc % a.w);
      const long row = pc / a.w;   // n*h + y
      uint16_t* obase = a.out + ((size_t)(2 * row) * (size_t)(2 * a.w) + 2 * x) * (size_t)a.ldo + a.co_off + cbase;
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        if (abSel >= 0 && ab != abSel) continue;
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int cs = i >> 1, r = (2 * i) & 3;
          split_pk_f16(fmaf(acc[ms][ab][cs][r], sc[cs][r], bi[cs][r]),
                       fmaf(acc[ms][ab][cs][r + 1], sc[cs][r + 1], bi[cs][r + 1]), ph[i], pl[i], amax);
        }
        if (ok) {
          uint16_t* op = obase + ((size_t)(ab >> 1) * (size_t)(2 * a.w) + (ab & 1)) * (size_t)a.ldo;
          uint4* o = reinterpret_cast<uint4*>(op);
          o[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          o[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
          uint4* ol = reinterpret_cast<uint4*>(op + a.outLo);
          ol[0] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          ol[1] = make_uint4(pl[4], pl[5], pl[6], pl[7]);
        }
      }
    }
  }
  if (MODE == 0) x3_report_range(amax, a.err);
}

// ConvTranspose2d weight (Cin, Cout, 2, 2) fp32 -> the kernel's operand layout
// [coTile(64)][chunk(32)][plane(2)][ab(4)][cs(4)][lane][8], un-prescaled (training: re-derived on the device after every
// optimizer step; the host packer of the inference tier pre-scales per output channel)
// Operand of the transposed convolution's INPUT gradient as a MODE 1 GEMM: K = 4 * cout rows k = ab * cout + co,
// N = cin columns n = ci, W_d[k][n] = w[ci][co][ab]; layout [tile(256 columns)][chunk(32 k)][plane][group(4)][cs][lane][8]
// with zeros for columns >= cin (cin = 128 fills half a tile).
__global__ __launch_bounds__(256) void pack_upconv_dgrad_x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ out,
                                                                   int cin, int cout) {
  const int K = 4 * cout, nCh = K / 32;
  const int nT = (cin + 255) / 256;
  const size_t total = (size_t)nT * nCh * 16 * 64;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int lane = (int)(i & 63);
    size_t r0 = i >> 6;
    const int cs = (int)(r0 & 3);
    const int grp = (int)((r0 >> 2) & 3);
    r0 >>= 4;
    const int kc = (int)(r0 % nCh);
    const int ct = (int)(r0 / nCh);
    const int j = lane & 15, lq = lane >> 4;
    const int n = 256 * ct + 64 * grp + 16 * (j >> 2) + 4 * cs + (j & 3);   // = ci
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2] = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = kc * 32 + lq * 8 + e2 * 2 + e;
        const int ab = k / cout, co = k - ab * cout;
        if (n < cin) v[e] = w[((size_t)n * cout + co) * 4 + ab];
      }
      split_pk_f16(v[0], v[1], hi[e2], lo[e2]);
    }
    uint16_t* base = out + ((size_t)ct * nCh + kc) * (size_t)(2 * 16 * 64 * 8);
    *reinterpret_cast<uint4*>(base + ((size_t)0 * 16 + grp * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(base + ((size_t)1 * 16 + grp * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

__global__ __launch_bounds__(256) void pack_upconv_x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ out,
                                                             int cin, int cout) {
  const int nCh = cin / 32;
  const size_t total = (size_t)(cout / 64) * nCh * 16 * 64;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int lane = (int)(i & 63);
    size_t r0 = i >> 6;
    const int cs = (int)(r0 & 3);
    const int ab = (int)((r0 >> 2) & 3);
    r0 >>= 4;
    const int kc = (int)(r0 % nCh);
    const int ct = (int)(r0 / nCh);
    const int j = lane & 15, lq = lane >> 4;
    const int co = 64 * ct + 16 * (j >> 2) + 4 * cs + (j & 3);
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int ci = kc * 32 + lq * 8 + e2 * 2;
      split_pk_f16(w[((size_t)ci * cout + co) * 4 + ab], w[((size_t)(ci + 1) * cout + co) * 4 + ab], hi[e2], lo[e2]);
    }
    uint16_t* base = out + ((size_t)ct * nCh + kc) * (size_t)(2 * 16 * 64 * 8);
    *reinterpret_cast<uint4*>(base + ((size_t)0 * 16 + ab * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(base + ((size_t)1 * 16 + ab * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

}  // namespace unet
