// Persistent, wave-specialised bf16 3x3 convolution for the wide shallow layers (BASELINE.json configs[2]).
//
// igemm_bf16.h keeps 2x2 waves per block, prefetches the next K-chunk of the input tile into VGPRs and streams
// weight fragments from L2 with global loads.  The vector memory pipeline returns data in order - per wave
// (vmcnt) and, measured here, effectively per CU: an L2-hit weight load queued behind HBM-miss tile loads waits
// for them (main loop at 68 cycles per MFMA with a DMA prefetch stream beside it, 16 is the pipe rate) - so in
// that kernel tile loads, MFMA and the epilogue run back to back (enc1.conv2 at batch 1024: 3.6 + 3.4 + 1.3 ms).
// This kernel keeps the MFMA waves off the vector memory pipeline altogether:
//
//  * one block per CU walks many 16x32-pixel tiles (persistent, XCD-local order);
//  * waves 4-7 only issue LDS-DMA (global_load_lds_dwordx4): per stage (= one 32-channel chunk of one tile) the
//    halo tile and that chunk's weights (9 taps x 64 channels), into the buffer pair the MFMA waves are not
//    reading; out-of-image halo pixels come from a zero page, so no validity masks; when a layer has one
//    channel tile and two chunks the weights are loaded once and stay resident;
//  * waves 0-3 (one per SIMD, 256 VGPRs) each own 128 pixels x 64 output channels and read BOTH operands from
//    LDS: 32 MFMAs per tap for 8 + 4 ds_read_b128, ping-pong registers (no copies), no vmcnt in the loop;
//  * operands are swapped - weights are the MFMA A operand, pixels the B operand - so an accumulator lane holds
//    16 consecutive channels of ONE pixel (the packing permutes channels to make them consecutive) and the
//    epilogue stores straight from registers: no LDS transpose, no barrier; 2x2 max-pool and the 1x1 head are
//    register/DPP operations on the same values;
//  * the pixel image is XOR-swizzled (16-byte part ^= 2 * bit 2 of the LDS pixel index, applied to the DMA SOURCE
//    address) so that the hardware's ds_read_b128 lane groups hit 16 distinct slots (2-way conflict otherwise).
//
// Barrier protocol: one s_barrier at start and one at the end of every stage, executed by all eight waves.  During
// stage s the loader fills buffer pair (s+1)&1 - which the MFMA waves stopped reading before the end barrier of
// stage s-1 (they drain lgkmcnt before every barrier) - and waits for it (vmcnt(0)) before the end barrier of s.
//
// Needs: Cin % 64 == 0 (chunks are consumed in pairs), Cout % 64 == 0, H % 16 == 0 (a tile never straddles two
// images, so "outside the image" is a per-tile property the loader can decide).
#pragma once
#include "igemm_bf16.h"

// Diagnostic build: s_memtime stamps around the phases of wave 0 (MFMA) and wave 4 (loader), summed per block into
// ConvWsArgs::stamps.  Never on in the shipped library; stamps go to their own buffer and no output depends on them.
#ifndef UNET_WS_STAMPS
#define UNET_WS_STAMPS 0
#endif
#if UNET_WS_STAMPS
#define WS_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WS_ACCUM(acc, t0) acc += __builtin_amdgcn_s_memtime() - (t0)
#else
#define WS_STAMP(var)
#define WS_ACCUM(acc, t0)
#endif

namespace unet {

struct ConvWsArgs {
  const uint16_t* in;     // NHWC bf16, pixel stride Cin
  const uint16_t* wt;     // packed [coTile(64 ch)][chunk(32)][tap][cs][lane][8], see pack_fragments_ws
  const uint16_t* zeros;  // >= Cin + 32 zero elements
  const float* scale;
  const float* shift;
  uint16_t* out;          // NHWC bf16, pixel stride ldo, channel offset co_off
  int N, H, W, Cin, Cout, ldo, co_off, tilesX, nChunks, relu;
  int coTiles, coGroup, pixTiles;
  uint16_t* pool;         // optional fused MaxPool2d(2,2): (N,H/2,W/2,Cout)
  const float* headW;     // optional fused 1x1 head (Cout == 64): weights [64]
  float headB, headThr;
  float* logits;
  float* probs;
  uint8_t* mask;
  int storeOut;
  unsigned long long* stamps;   // diagnostic builds (-DUNET_WS_STAMPS=1): 8 counters per block, else unused
};

struct WsShape {
  static constexpr int TW = 32, TH = 16;
  static constexpr int HW2 = TW + 2, HH2 = TH + 2;
  static constexpr int P = 36;                          // LDS row pitch in pixels (P % 8 == 4: see xa below)
  static constexpr int NQX = (HH2 * P * 4 + 63) / 64;   // 1 KiB DMA pieces of the halo tile (41)
  static constexpr int NQW = 9 * 4;                     // ... of one chunk's weights (36)
  static constexpr int XBUF = NQX * 1024, WBUF = NQW * 1024;
  static constexpr int WOFF = 0, XOFF = 2 * WBUF;       // weight buffers first: their read offsets fit ds_read's 16-bit immediate
  static constexpr int TOFF = 2 * XBUF + 2 * WBUF;      // scale | shift | head weights table (fp32) behind the buffers
  static constexpr int MAX_COUT = 512;
  static constexpr int LDS_BYTES = TOFF + (2 * MAX_COUT + 64) * 4;   // 162,048 of 163,840
  static constexpr int NLOAD = 4;                       // loader waves (waves 4..7)
};

__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// work item -> (first global row, first column, channel tile); consecutive items walk coGroup channel tiles of one
// pixel tile (same XCD, same input tile in L2)
__device__ __forceinline__ void ws_decode(int w, int coGroup, int pixTiles, int tilesX, int& g0, int& x0,
                                          int& coTile) {
  const int cInG = w % coGroup;
  const int rest = w / coGroup;
  const int tile = rest % pixTiles;
  coTile = (rest / pixTiles) * coGroup + cInG;
  g0 = (tile / tilesX) * WsShape::TH;
  x0 = (tile % tilesX) * WsShape::TW;
}

// value of lane li ^ 1 (quad_perm [1,0,3,2])
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// element-wise max of two packed int16 pairs (v_pk_max_i16).  On bf16 bit patterns: max(x, 0) is ReLU (a negative
// bf16, -0 included, is a negative int16), and for non-negative values int16 order is bf16 order.
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}

// EPI: 0 = store the activation, 1 = store it and its 2x2 max-pool, 2 = fused 1x1 head only (activation not stored)
template <int EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_bf16_ws_kernel(const ConvWsArgs a) {
  using S = WsShape;
  constexpr int TW = S::TW, TH = S::TH, P = S::P, NQX = S::NQX, NQW = S::NQW;

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;   // multiple of 8: consecutive logical blocks share an XCD (and its L2)
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles;
  const int NH = a.N * a.H;
  const int tilesMine = lb < numWork ? (numWork - lb + G - 1) / G : 0;
  const int totalStages = tilesMine * a.nChunks;
  const bool resident = a.coTiles == 1 && a.nChunks == 2;   // chunk kc always lands in weight buffer kc

  if (wave >= 4) {
    // ---------------- loader waves: wave 4+k issues the 1 KiB DMA pieces q = k (mod 4) ----------------
    // (one wave alone needs about a whole stage just to issue ~80 pieces beside an MFMA wave on its SIMD)
    const int k = wave - 4;
    constexpr int NL = S::NLOAD;
    constexpr int NQXK = (NQX + NL - 1) / NL;   // 11
    constexpr int NQWK = NQW / NL;              // 9
    int hrc[NQXK];   // halo row << 16 | halo column << 4 | source 16-byte part; tile independent
#pragma unroll
    for (int j = 0; j < NQXK; ++j) {
      int q = k + j * NL;
      q = q < NQX ? q : NQX - 1;   // the last round only exists for some k: duplicates rewrite the same bytes
      const int v = q * 64 + lane;
      const int qpix = v >> 2;
      const int part = (v & 3) ^ (((qpix >> 2) & 1) << 1);
      const int hr = qpix / P, hc = qpix - hr * P;
      hrc[j] = (hr << 16) | (hc << 4) | part;
    }
    const uint16_t* ptr[NQXK];
    // One loop, one issue site (the arrays must stay in registers: a scratch reload would be a vmcnt operation
    // queued behind the DMA).  Iteration i issues stage i, then waits for it and joins the barrier that ends
    // stage i-1 (the start barrier for i = 0).
    int wN = lb, kcN = 0, coTileN = 0;
    unsigned long long tDma = 0, tBarL = 0;
    WS_STAMP(tStartL);
    for (int i = 0; i <= totalStages; ++i) {
      if (i < totalStages) {
        if (kcN == 0) {
          int g0, x0;
          ws_decode(wN, a.coGroup, a.pixTiles, a.tilesX, g0, x0, coTileN);
          const int y0 = g0 % a.H;
          const int hrMin = y0 == 0 ? 1 : 0;
          const int hrMax = a.H - y0 < S::HH2 - 1 ? a.H - y0 : S::HH2 - 1;
          const int hcMin = x0 == 0 ? 1 : 0;
          const int hcMax = a.W - x0 < S::HW2 - 1 ? a.W - x0 : S::HW2 - 1;
          const uint16_t* tileBase = a.in + ((size_t)g0 * a.W + x0) * (size_t)a.Cin;
#pragma unroll
          for (int j = 0; j < NQXK; ++j) {
            const int hr = hrc[j] >> 16, hc = (hrc[j] >> 4) & 0xFFF, part = hrc[j] & 3;
            const bool ok = hr >= hrMin && hr <= hrMax && hc >= hcMin && hc <= hcMax;
            const int off = ((hr - 1) * a.W + (hc - 1)) * a.Cin + part * 8;
            ptr[j] = ok ? tileBase + off : a.zeros + part * 8;
          }
        }
        char* xdst = reinterpret_cast<char*>(smemv) + S::XOFF + (i & 1) * S::XBUF;
#pragma unroll
        for (int j = 0; j < NQXK; ++j) {
          int q = k + j * NL;
          q = q < NQX ? q : NQX - 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ptr[j] + kcN * 32),
                                           (__attribute__((address_space(3))) void*)(xdst + q * 1024), 16, 0, 0);
        }
        if (!resident || i < 2) {
          char* wdst = reinterpret_cast<char*>(smemv) + S::WOFF + (i & 1) * S::WBUF;
          const uint16_t* wsrc = a.wt + ((size_t)coTileN * a.nChunks + kcN) * (NQW * 512) + lane * 8;
#pragma unroll
          for (int j = 0; j < NQWK; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wsrc + (k + j * NL) * 512),
                (__attribute__((address_space(3))) void*)(wdst + (k + j * NL) * 1024), 16, 0, 0);
        }
        if (++kcN == a.nChunks) {
          kcN = 0;
          wN += G;
        }
      }
      {
        WS_STAMP(t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_ACCUM(tDma, t0);
        WS_STAMP(t1);
        asm volatile("s_barrier" ::: "memory");
        WS_ACCUM(tBarL, t1);
      }
    }
#if UNET_WS_STAMPS
    if (a.stamps && lane == 0 && k == 0) {
      a.stamps[blockIdx.x * 8 + 4] = tDma;
      a.stamps[blockIdx.x * 8 + 5] = tBarL;
      a.stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime() - tStartL;
    }
#endif
    return;
  }

  // ---------------- MFMA waves ----------------
  const int li = lane & 15, lq = lane >> 4;
  // Fragment ms of this wave is 16 consecutive columns of one tile row: column block cb = ms & 1, tile row
  // wave*4 + ms/2.  xa[cb][kx][rho]: byte address (pixel buffer 0) of this lane's 16-byte piece at tap column kx
  // in the wave's first row, for LDS rows of parity rho (P % 8 == 4, so the swizzle bit of pixel index
  // row*P + col is bit 2 of 4*(row & 1) + (col & 7)); fragment row and tap row add the immediate
  // (ms/2 + ky) * P * 64.
  int xa[2][3][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int rho = 0; rho < 2; ++rho) {
        const int col = cb * 16 + li + kx;
        const int key = ((4 * rho + (col & 7)) >> 2) & 1;
        xa[cb][kx][rho] = S::XOFF + (wave * 4 * P + col) * 64 + ((lq ^ (key << 1)) << 4);
        // opaque to the optimiser: otherwise every (base + constant) pair becomes its own hoisted VGPR
        // (72 of them for the weights alone) instead of one base plus ds_read's immediate offset
        asm volatile("" : "+v"(xa[cb][kx][rho]));
      }
  int wa[2] = {lane * 16, S::WBUF + lane * 16};   // this lane's 16 bytes of (tap 0, cs 0) in weight buffers 0 / 1
  asm volatile("" : "+v"(wa[0]), "+v"(wa[1]));
  const char* lds = reinterpret_cast<const char*>(smemv);
  // scale | shift | head weights -> LDS once: a global load in the epilogue would queue behind the DMA stream
  {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smemv) + S::TOFF);
    for (int c = tid; c < a.Cout; c += 256) {
      tab[c] = a.scale[c];
      tab[S::MAX_COUT + c] = a.shift[c];
    }
    if (tid < 64) tab[2 * S::MAX_COUT + tid] = a.headW ? a.headW[tid] : 0.f;
  }
  // per-lane byte offsets of the epilogue stores inside a fragment row (pixel li, channels 16*lq..)
  const unsigned outLane = (unsigned)(li * a.ldo + lq * 16) * 2u;
  const unsigned poolLane = (unsigned)((li >> 1) * a.Cout + lq * 16) * 2u;
  unsigned long long tBar = 0, tEpi = 0;
  WS_STAMP(tStart);
  ws_barrier();
  for (int w = lb; w < numWork; w += G) {
    int g0, x0, coTile;
    ws_decode(w, a.coGroup, a.pixTiles, a.tilesX, g0, x0, coTile);
    f32x4 acc[8][4];
#pragma unroll
    for (int ms = 0; ms < 8; ++ms)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[ms][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 wf[2][4], xf[8];
    for (int kc = 0; kc < a.nChunks; kc += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // stage (tile, kc + h) sits in buffer pair h (stage index parity: tiles have an even number of chunks)
        auto xread = [&](int ms, int t) -> f32x4 {
          const int ky = t / 3, kx = t % 3, row = ms / 2 + ky;
          return *reinterpret_cast<const f32x4*>(lds + xa[ms & 1][kx][row & 1] + h * S::XBUF + row * (P * 64));
        };
        auto wread = [&](int cs, int t) -> f32x4 {
          return *reinterpret_cast<const f32x4*>(lds + wa[h] + t * 4096 + cs * 1024);
        };
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) wf[h & 1][cs] = wread(cs, 0);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) xf[ms] = xread(ms, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int cur = (h + t) & 1, nxt = cur ^ 1;
          // Each tap runs in two halves of 16 MFMAs; the pixel fragments are single-buffered and refilled half
          // a tap (256 cycles) ahead: fragments 4-7 of this tap during half A, fragments 0-3 and the weights of
          // the next tap during half B.  One read goes out behind every second MFMA.
#pragma unroll
          for (int ms = 4; ms < 8; ++ms) xf[ms] = xread(ms, t);
#pragma unroll
          for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int cs = 0; cs < 4; ++cs)
              acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cur][cs]),
                                                                    __builtin_bit_cast(bf16x8, xf[ms]),
                                                                    acc[ms][cs], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (t < 8) {
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) xf[ms] = xread(ms, t + 1);
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) wf[nxt][cs] = wread(cs, t + 1);
          }
#pragma unroll
          for (int ms = 4; ms < 8; ++ms)
#pragma unroll
            for (int cs = 0; cs < 4; ++cs)
              acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cur][cs]),
                                                                    __builtin_bit_cast(bf16x8, xf[ms]),
                                                                    acc[ms][cs], 0, 0, 0);
          if (t < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          WS_STAMP(t0);
          ws_barrier();
          WS_ACCUM(tBar, t0);
        }
      }
    }

    WS_STAMP(tE0);
    // ---- epilogue straight from the accumulators: lane (li, lq) holds channels 64*coTile + 16*lq + [0,16) of
    //      pixel li of each fragment: acc[ms][cs][r] is channel 16*lq + 4*cs + r.  Addresses are a uniform
    //      (SGPR) part per fragment plus a per-lane byte offset computed once per kernel. ----
    const int cbase = coTile * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (cbase + cs * 4) * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (S::MAX_COUT + cbase + cs * 4) * 4);
    }
    // ReLU on the packed bf16 pair: max with 0, or with int16 min (a no-op) when the layer has none
    const uint32_t floorPk = a.relu ? 0u : 0x80008000u;
    const bool fullW = x0 + TW <= a.W;   // uniform: only the last tile of a row can be partial
    // fragment pairs that are vertical neighbours (tile rows 2k, 2k+1): (ms, ms+2) for ms in {0,1,4,5}
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int msA = (j & 1) + (j >> 1) * 4;
      const int cb = msA & 1, rA = wave * 4 + (msA >> 1);
      const bool okx = fullW || x0 + cb * 16 + li < a.W;
      uint32_t pk[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ms = msA + 2 * u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int cs = i >> 1, r = (2 * i) & 3;
          const float v0 = fmaf(acc[ms][cs][r], sc[cs][r], sh[cs][r]);
          const float v1 = fmaf(acc[ms][cs][r + 1], sc[cs][r + 1], sh[cs][r + 1]);
          pk[u][i] = pk_max_i16(pk_bf16(v0, v1), floorPk);
        }
      }
      if (EPI != 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          char* rowp = reinterpret_cast<char*>(a.out) +
                       (((size_t)(g0 + rA + u) * a.W + x0 + cb * 16) * (size_t)a.ldo + a.co_off + coTile * 64) * 2;
          if (okx) {
            uint4* o = reinterpret_cast<uint4*>(rowp + outLane);
            o[0] = make_uint4(pk[u][0], pk[u][1], pk[u][2], pk[u][3]);
            o[1] = make_uint4(pk[u][4], pk[u][5], pk[u][6], pk[u][7]);
          }
        }
      }
      if (EPI == 1) {
        // MaxPool2d(2,2) on the rounded, non-negative (post-ReLU: the host only fuses the pool then) values, where
        // int16 order is bf16 order and max commutes with the rounding: vertical neighbour in the partner fragment,
        // horizontal neighbour in lane li ^ 1
        uint32_t pp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t m = pk_max_i16(pk[0][i], pk[1][i]);
          pp[i] = pk_max_i16(m, dpp_xor1(m));
        }
        char* rowp = reinterpret_cast<char*>(a.pool) +
                     (((size_t)((g0 + rA) >> 1) * (a.W >> 1) + ((x0 + cb * 16) >> 1)) * (size_t)a.Cout + coTile * 64) * 2;
        if (okx && (li & 1) == 0) {
          uint4* o = reinterpret_cast<uint4*>(rowp + poolLane);
          o[0] = make_uint4(pp[0], pp[1], pp[2], pp[3]);
          o[1] = make_uint4(pp[4], pp[5], pp[6], pp[7]);
        }
      }
      if (EPI == 2) {
        // fused 1x1 head (reference README.md:1447) on the bf16-rounded activation, as the unfused path reads it
        f32x4 hw[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs)
          hw[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (2 * S::MAX_COUT + lq * 16 + cs * 4) * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float z = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            z = fmaf(__builtin_bit_cast(float, pk[u][i] << 16), hw[i >> 1][(2 * i) & 3], z);
            z = fmaf(__builtin_bit_cast(float, pk[u][i] & 0xFFFF0000u), hw[i >> 1][(2 * i + 1) & 3], z);
          }
          z += __shfl_xor(z, 16, 64);
          z += __shfl_xor(z, 32, 64);
          z += a.headB;
          if (okx && lq == 0) {
            const size_t o = (size_t)(g0 + rA + u) * a.W + x0 + cb * 16 + li;
            if (a.logits) a.logits[o] = z;
            if (a.probs) a.probs[o] = 1.f / (1.f + __expf(-z));
            if (a.mask) a.mask[o] = z > a.headThr ? 255 : 0;
          }
        }
      }
    }
    WS_ACCUM(tEpi, tE0);
  }
#if UNET_WS_STAMPS
  if (a.stamps && lane == 0 && wave == 0) {
    a.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime() - tStart;
    a.stamps[blockIdx.x * 8 + 1] = tBar;
    a.stamps[blockIdx.x * 8 + 2] = tEpi;
  }
#endif
}

}  // namespace unet
