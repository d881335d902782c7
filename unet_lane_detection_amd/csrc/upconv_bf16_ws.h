// ConvTranspose2d(kernel 2, stride 2) of the bf16 tier as a persistent wave-specialised kernel (reference
// README.md:1442, :1476): y[2i+a][2j+b][co] = sum_ci x[i][j][ci] * W[ci][co][a][b] + bias[co], written into the upper
// channel slice of the concat buffer.  Same construction as conv_bf16_ws.h - loader waves feed LDS by DMA, the MFMA
// waves touch no vector memory until their register epilogue - for a 1x1 GEMM with four output positions:
//
//  * a work item is 128 consecutive input pixels (flattened n,y,x: a 1x1 convolution has no halo) x 64 output
//    channels x all four (a,b); a stage is one 64-channel chunk of it: 16 KiB of pixels and 32 KiB of weights
//    (4 (a,b) x 4 subtiles x 2 k-steps of fragments), double buffered; weights stay resident when the layer has one
//    channel tile and exactly two chunks (stage parity = chunk);
//  * MFMA wave w owns pixels [32w, 32w+32) as two 16-pixel fragments and all 4 x 64 outputs: 32 accumulators
//    (128 VGPRs); per stage 64 MFMAs for 4 pixel-fragment and 32 weight-fragment ds_read_b128;
//  * weights are the MFMA A operand with the channel permutation of conv_bf16_ws.h, so a lane holds 16 consecutive
//    channels of one output pixel per (a,b) and stores 32 bytes, four times per fragment;
//  * the pixel image is XOR-swizzled on the DMA source side (16-byte part ^= (pixel >> 1) & 7): pixels are 128
//    bytes apart, and a fragment read would otherwise be an 8-way bank conflict.
//
// The layer is bound by its stores (4x the input bytes); the old kernel (igemm_bf16_kernel MODE 1) ran loads, MFMA
// and its LDS-transposed epilogue back to back at 45-55 % of the HBM rate.
#pragma once
#include "conv_bf16_ws.h"

namespace unet {

struct UpconvWsArgs {
  const uint16_t* in;     // (N,h,w,Cin) bf16
  const uint16_t* wt;     // packed [coTile(64)][chunk(64 ch)][kstep(2)][ab(4)][cs(4)][lane][8], see pack_upconv_ws
  const uint16_t* zeros;  // >= 64 zero elements
  const float* bias;      // [Cout]
  uint16_t* out;          // (N,2h,2w,ldo) bf16, channels [co_off, co_off + Cout)
  long npix;              // N*h*w
  int h, w, Cin, Cout, ldo, co_off, nChunks;   // nChunks = Cin / 64
  int coTiles, pixTiles;
};

struct UpconvWsShape {
  static constexpr int TP = 128;                       // pixels per tile
  static constexpr int XBUF = TP * 128, WBUF = 2 * 4 * 4 * 1024;   // 16 KiB, 32 KiB
  static constexpr int NQX = XBUF / 1024, NQW = WBUF / 1024;        // 16, 32 DMA pieces
  static constexpr int WOFF = 0, XOFF = 2 * WBUF;
  static constexpr int TOFF = XOFF + 2 * XBUF;          // bias table (fp32)
  static constexpr int MAX_COUT = 1024;
  static constexpr int LDS_BYTES = TOFF + MAX_COUT * 4; // 102,400
};

__global__ __launch_bounds__(512, 1) void upconv2x2_bf16_ws_kernel(const UpconvWsArgs a) {
  using S = UpconvWsShape;
  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;   // consecutive items: the channel tiles of one pixel tile
  const int tilesMine = lb < numWork ? (numWork - lb + G - 1) / G : 0;
  const int totalStages = tilesMine * a.nChunks;
  const bool resident = a.coTiles == 1 && a.nChunks == 2;   // chunk kc always lands in weight buffer kc

  if (wave >= 4) {
    // ---------------- loader waves: wave 4+k issues the pieces q = k (mod 4) ----------------
    const int k = wave - 4;
    int wN = lb, kcN = 0;
    for (int i = 0; i <= totalStages; ++i) {
      if (i < totalStages) {
        const int tile = wN / a.coTiles, coTile = wN - tile * a.coTiles;
        const long p0 = (long)tile * S::TP;
        char* xdst = reinterpret_cast<char*>(smemv) + S::XOFF + (i & 1) * S::XBUF;
#pragma unroll
        for (int j = 0; j < S::NQX / 4; ++j) {
          const int q = k + 4 * j;
          const int px = q * 8 + (lane >> 3);
          const int part = (lane & 7) ^ ((px >> 1) & 7);
          const uint16_t* src = p0 + px < a.npix ? a.in + (size_t)(p0 + px) * (size_t)a.Cin + kcN * 64 + part * 8
                                                 : a.zeros + part * 8;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(xdst + q * 1024), 16, 0, 0);
        }
        if (!resident || i < a.nChunks) {
          char* wdst = reinterpret_cast<char*>(smemv) + S::WOFF + (i & 1) * S::WBUF;
          const uint16_t* wsrc = a.wt + ((size_t)coTile * a.nChunks + kcN) * (S::NQW * 512) + lane * 8;
#pragma unroll
          for (int j = 0; j < S::NQW / 4; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wsrc + (k + 4 * j) * 512),
                (__attribute__((address_space(3))) void*)(wdst + (k + 4 * j) * 1024), 16, 0, 0);
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
  {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smemv) + S::TOFF);
    for (int c = tid; c < a.Cout; c += 256) tab[c] = a.bias[c];
  }
  // this lane's 16 bytes of pixel fragment ms at k-step j: pixel 32*wave + 16*ms + li, logical part 4*j + lq
  int xa[2][2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = wave * 32 + ms * 16 + li;
      xa[ms][j] = S::XOFF + px * 128 + (((j * 4 + lq) ^ ((px >> 1) & 7)) << 4);
      asm volatile("" : "+v"(xa[ms][j]));
    }
  int wa = lane * 16;
  asm volatile("" : "+v"(wa));
  ws_barrier();
  int stage = 0;
  for (int w = lb; w < numWork; w += G) {
    const int tile = w / a.coTiles, coTile = w - tile * a.coTiles;
    f32x4 acc[2][4][4];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab)
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[ms][ab][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < a.nChunks; ++kc, ++stage) {
      const int xoff = (stage & 1) * S::XBUF, woff = (stage & 1) * S::WBUF;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 xf[2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) xf[ms] = *reinterpret_cast<const f32x4*>(lds + xa[ms][j] + xoff);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
          for (int cs = 0; cs < 4; ++cs) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(lds + wa + woff + ((j * 4 + ab) * 4 + cs) * 1024);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
              acc[ms][ab][cs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                        __builtin_bit_cast(bf16x8, xf[ms]),
                                                                        acc[ms][ab][cs], 0, 0, 0);
          }
      }
      ws_barrier();
    }

    // ---- epilogue: lane (li, lq) holds channels 64*coTile + 16*lq + [0,16) of input pixel li of each fragment,
    //      for each (a,b); + bias, bf16, 32-byte stores to output pixel (2y+a, 2x+b) ----
    const int cbase = coTile * 64 + lq * 16;
    f32x4 bi[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) bi[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (cbase + cs * 4) * 4);
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
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int cs = i >> 1, r = (2 * i) & 3;
          pk[i] = pk_bf16(acc[ms][ab][cs][r] + bi[cs][r], acc[ms][ab][cs][r + 1] + bi[cs][r + 1]);
        }
        if (ok) {
          uint4* o = reinterpret_cast<uint4*>(obase + ((size_t)(ab >> 1) * (size_t)(2 * a.w) + (ab & 1)) * (size_t)a.ldo);
          o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
      }
    }
  }
}

}  // namespace unet
