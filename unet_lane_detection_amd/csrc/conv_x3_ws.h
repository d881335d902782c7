// Split-operand ("f16x3") 3x3 convolution: fp32-level accuracy on the fp16 MFMA pipe.
//
// Every fp32 operand v is carried as two fp16 values, hi = rn(v) and lo = rn(v - hi) (v - hi is exact in fp32, so
// hi + lo holds 22 significant bits of v), and every product is formed as
//     w x  ~=  w_hi x_hi + w_hi x_lo + w_lo x_hi          (the dropped w_lo x_lo term is ~2^-22 |w x|)
// by three v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Three fp16 MFMAs cost 3/16 of the fp32 MFMA that the
// exact-fp32 kernels (igemm_f32.h, wino_f32.h) spend on the same products.  fp16 subnormals pass through the MFMA
// un-flushed (tools/probes/mfma_f16_denorm.hip, measured on MI355X), so small lo parts degrade gracefully to an
// absolute 2^-24; weights are additionally pre-scaled per output channel by a power of two (folded back into the
// BatchNorm scale) so their lo parts stay normal.  Measured on the reference frame (tests/dev/split_precision_sim.py,
// torch CPU emulation of exactly this arithmetic): max |dlogit| 3.1e-5 against the fp32 oracle - the same as the
// exact-fp32 HIP kernels - where the bf16 variant of the split gives 3.2e-4 and plain bf16 1.4e-1.
//
// Activations live in HBM as two fp16 NHWC planes (hi plane, lo plane `loOff` elements behind it): the same 4 bytes
// per element as fp32.  The producer's epilogue does the split once per element; consumers only read fp16 operands.
//
// Kernel structure = conv_bf16_ws.h (persistent, wave-specialised, both operands through LDS-DMA; read that
// header first), with these differences:
//  * block tile 256 pixels x 64 output channels: TW = 32 -> 8 rows x 32 columns, TW = 16 -> 16 x 16 (the 28x28 and
//    14x14 maps); tiles are per image (tilesY = ceil(H / TH)), rows past the image bottom read the zero page and are
//    not stored, so any H works;
//  * MFMA wave w owns 4 pixel fragments (64 pixels) x 64 channels = 16 accumulators; per tap it reads 8 pixel
//    fragments (hi, lo) and 8 weight fragments (hi, lo) for 48 MFMAs;
//  * LDS: the halo tile of a 32-channel chunk for both planes (2 x 23 KiB), double buffered per CHUNK, and the
//    weights of ONE TAP ROW (3 taps x 64 channels x 32 ci, both planes: 24 KiB), double buffered per SUB-STAGE
//    (chunk, tap row); one s_barrier per sub-stage.  Loader iteration i issues the weights of sub-stage i and a third
//    of the halo tile of the chunk that starts at sub-stage 3*ceil(i/3), waits for them, and joins the barrier
//    that ends sub-stage i-1;
//  * epilogue: scale/shift, ReLU in fp32, then the hi/lo split; 2 x 32-byte stores per plane per fragment; the
//    2x2 max-pool is taken on the fp32 values (the split is monotonic), the fused 1x1 head on hi + lo.
//
// Needs Cin % 64 == 0 (chunks are consumed in pairs so that buffer parities are compile-time constants) and
// Cout % 64 == 0.  |activation| must stay below 65504 (fp16 range; values are clamped, not wrapped).
#pragma once
#include "conv_bf16_ws.h"

namespace unet {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvX3Args {
  const uint16_t* in;     // hi plane, NHWC fp16, pixel stride Cin; lo plane at in + inLo
  size_t inLo;
  const uint16_t* wt;     // packed [coTile(64)][chunk(32)][tapRow(3)][plane(2)][kx(3)][cs(4)][lane(64)][8]
  const uint16_t* zeros;  // >= 64 zero halfs
  const float* scale;     // per channel, already divided by the weights' power-of-two pre-scale
  const float* shift;
  uint16_t* out;          // hi plane, pixel stride ldo, channel offset co_off; lo plane at out + outLo
  size_t outLo;
  int N, H, W, Cin, Cout, ldo, co_off, tilesX, tilesY, nChunks, relu;
  int coTiles, coGroup, pixTiles;
  uint16_t* pool;         // optional fused MaxPool2d(2,2): hi plane (N,H/2,W/2,Cout), lo plane at pool + poolLo
  size_t poolLo;
  const float* headW;     // optional fused 1x1 head (Cout == 64)
  float headB, headThr;
  float* logits;
  float* probs;
  uint8_t* mask;
  float* outF;            // EPI 3: fp32 output (pixel stride ldo, channel offset co_off) instead of the planes
  // split-K (small batches: a 14x14 map is one pixel tile, its layer 16 work items on 256 CUs): a work item covers
  // nChunks of the chunksTotal = Cin / 32 input-channel chunks, kSplit items per (pixel tile, channel tile); with
  // EPI 3 item ks writes its raw partial sums at outF + ks * splitStride and x3_splitk_finish_kernel adds them up
  int kSplit, chunksTotal;
  size_t splitStride;
  // FLAT instances: the batch is addressed as ONE image of N * imgH rows (H = N * imgH, N = 1), so tiles of TH rows pack
  // maps whose height is not a multiple of TH (28, 14) without padding rows; taps that would cross an image boundary
  // are zeroed in the MFMA loop (first row of an image: tap row 0, last row: tap row 2)
  int imgH;
  const float* dynScale;  // optional device scalar multiplied into every channel scale (undoes the power-of-two
                          // scaling of an input that was brought into the fp16 range: split_planes_scaled_kernel)
  unsigned* err;          // the handle's error block (may be null): word 1 = an activation left the fp16 range
  // conv_x3_r512.h, EPI 3 (training forward): when not null, every block adds up, per output channel, the sum and the
  // sum of squares of the fp32 values it stores (the BatchNorm statistics' first pass) and writes them as row
  // blockIdx.x * WPX + wp of a zero-initialised [rows][2][Cout] array that bn_finalize_kernel then sums
  float* statPartial;
};

template <int TW_>
struct X3Shape {
  static constexpr int TW = TW_, TH = 256 / TW_;
  static constexpr int HW2 = TW + 2, HH2 = TH + 2;
  static constexpr int P = TW + 4;                       // LDS row pitch in pixels: 36 / 20, P % 8 == 4
  static constexpr int NQX = (HH2 * P * 64 + 1023) / 1024;   // 1 KiB DMA pieces of one plane's halo tile (23)
  static constexpr int XPL = NQX * 1024;                 // bytes of one plane buffer
  static constexpr int XST = 2 * XPL;                    // hi + lo of one chunk
  static constexpr int WPL = 3 * 4 * 1024;               // one tap row, one plane: 3 taps x 4 subtiles x 1 KiB
  static constexpr int WST = 2 * WPL;                    // hi + lo
  static constexpr int WOFF = 0, XOFF = 2 * WST, TOFF = XOFF + 2 * XST;
  static constexpr int MAX_COUT = 1024;
  static constexpr int LDS_BYTES = TOFF + (2 * MAX_COUT + 64) * 4;
  static constexpr int ROWS_PER_WAVE = TH / 4;           // 2 / 4
  static constexpr int NJ = (NQX + 3) / 4;               // halo pieces per loader wave and plane (6)
  static_assert(P % 8 == 4, "the bank swizzle assumes P % 8 == 4");
  static_assert(HH2 * P * 64 <= XPL, "halo tile does not fit its buffer");
  static_assert(NJ % 3 == 0, "halo pieces are issued in three equal parts");
};

// fp32 -> (hi, lo) fp16 pairs for two values: hi = rn(v), lo = rn(v - hi); values clamped to the fp16 range
__device__ __forceinline__ void split_pk_f16(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  v0 = fminf(fmaxf(v0, -65504.f), 65504.f);
  v1 = fminf(fmaxf(v1, -65504.f), 65504.f);
  const f16x2 h = __builtin_convertvector((f32x2){v0, v1}, f16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector((f32x2){v0 - hf[0], v1 - hf[1]}, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// Range watch.  The fp16 planes hold |v| <= 65504; a producer that has to clamp returns results an fp32 network
// would not (reference README.md:1449-1458 is plain fp32).  Every kernel that stores ACTIVATIONS as planes keeps the
// largest |v| it split in `amax` (one v_max3_f32 per pair) and, if that left the range, sets word 1 of the handle's
// error block (word 0: a bounded wait gave up, wino_f32.h): unet_device_error then reports UNET_ERR_RANGE and the
// caller re-runs the frames on the exact-fp32 tier (py_utils/rknn_executor.py).
__device__ __forceinline__ void split_pk_f16(float v0, float v1, uint32_t& hi, uint32_t& lo, float& amax) {
  amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
  split_pk_f16(v0, v1, hi, lo);
}
__device__ __forceinline__ void x3_report_range(float amax, unsigned* err) {
  if (amax > 65504.f && err) *reinterpret_cast<volatile unsigned*>(err + 1) = 1u;
}

// (hi, lo) packed pairs -> the two fp32 values they stand for
__device__ __forceinline__ void merge_pk_f16(uint32_t hi, uint32_t lo, float& v0, float& v1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f32x2 h = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
  const f32x2 l = __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
  v0 = h[0] + l[0];
  v1 = h[1] + l[1];
}

__device__ __forceinline__ float dpp_xor1_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int TW_>
__device__ __forceinline__ void x3_decode(int w, const ConvX3Args& a, int& n, int& y0, int& x0, int& coTile, int& ks) {
  using S = X3Shape<TW_>;
  ks = w % a.kSplit;
  w /= a.kSplit;
  const int cInG = w % a.coGroup;
  const int rest = w / a.coGroup;
  const int tile = rest % a.pixTiles;
  coTile = (rest / a.pixTiles) * a.coGroup + cInG;
  const int rowTile = tile / a.tilesX;
  x0 = (tile - rowTile * a.tilesX) * S::TW;
  n = rowTile / a.tilesY;
  y0 = (rowTile - n * a.tilesY) * S::TH;
}

// EPI: 0 = store the activation, 1 = store it and its 2x2 max-pool, 2 = fused 1x1 head only (activation not stored),
//      3 = store the activation as fp32 (training: the raw convolution output feeds BatchNorm statistics)
template <int TW_, int EPI, bool FLAT = false>
__global__ __launch_bounds__(512, 1) void conv3x3_x3_ws_kernel(const ConvX3Args a) {
  using S = X3Shape<TW_>;
  constexpr int TW = S::TW, P = S::P, NQX = S::NQX, NJ = S::NJ;

  extern __shared__ __attribute__((aligned(16))) f32x4 smemv[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;   // multiple of 8: consecutive logical blocks share an XCD (and its L2)
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int numWork = a.pixTiles * a.coTiles * a.kSplit;
  const int tilesMine = lb < numWork ? (numWork - lb + G - 1) / G : 0;
  const int totalSub = tilesMine * a.nChunks * 3;   // sub-stages of this block

  if (wave >= 4) {
    // ---------------- loader waves: wave 4+k issues halo pieces q = k + 4j of both planes, weight pieces k + 4j ----
    const int k = wave - 4;
    int hrc[NJ];   // halo row << 16 | halo column << 4 | source 16-byte part; tile independent
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int q = k + j * 4;
      q = q < NQX ? q : NQX - 1;   // the last round only exists for some k: duplicates rewrite the same bytes
      const int v = q * 64 + lane;
      const int qpix = v >> 2;
      const int part = (v & 3) ^ (((qpix >> 2) & 1) << 1);
      const int hr = qpix / P, hc = qpix - hr * P;
      hrc[j] = (hr << 16) | (hc << 4) | part;
    }
    const uint16_t* ptr[NJ];   // hi-plane source of each piece for the current halo tile (lo plane: + inLo, if in-image)
    unsigned okMask = 0;       // bit j: piece j of this lane is inside the image (else the zero page)
    int wX = lb, kcX = 0;      // (work item, chunk) whose halo tile is being issued
    int wW = lb, kcW = 0, rW = 0, coTileW = 0;   // (work item, chunk, tap row) whose weights are being issued
    int kBaseX = 0, kBaseW = 0;   // first chunk of the item's K range
    auto halo_ptrs = [&]() __attribute__((always_inline)) {
      int n, y0, x0, coT, ksX;
      x3_decode<TW_>(wX, a, n, y0, x0, coT, ksX);
      kBaseX = ksX * a.nChunks;
      const int hrMax = a.H - y0 < S::HH2 - 1 ? a.H - y0 : S::HH2 - 1;
      const int hrMin = y0 == 0 ? 1 : 0;
      const int hcMin = x0 == 0 ? 1 : 0;
      const int hcMax = a.W - x0 < S::HW2 - 1 ? a.W - x0 : S::HW2 - 1;
      const uint16_t* tileBase = a.in + (((size_t)n * a.H + y0) * a.W + x0) * (size_t)a.Cin;
      okMask = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int hr = hrc[j] >> 16, hc = (hrc[j] >> 4) & 0xFFF, part = hrc[j] & 3;
        const bool ok = hr >= hrMin && hr <= hrMax && hc >= hcMin && hc <= hcMax;
        const long off = ((long)(hr - 1) * a.W + (hc - 1)) * a.Cin + part * 8;
        ptr[j] = ok ? tileBase + off : a.zeros + part * 8;
        okMask |= ok ? (1u << j) : 0u;
      }
    };
    auto issue_halo = [&](int part3, int buf) __attribute__((always_inline)) {   // a third of the halo tile of (wX, kcX), both planes
      char* xdst = reinterpret_cast<char*>(smemv) + S::XOFF + buf * S::XST;
#pragma unroll
      for (int jj = 0; jj < NJ / 3; ++jj) {
        const int j = part3 * (NJ / 3) + jj;
        int q = k + j * 4;
        q = q < NQX ? q : NQX - 1;
        const uint16_t* src = ptr[j] + (((okMask >> j) & 1u) ? (kBaseX + kcX) * 32 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xdst + q * 1024), 16, 0, 0);
        const uint16_t* srcLo = ((okMask >> j) & 1u) ? src + a.inLo : src;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcLo,
                                         (__attribute__((address_space(3))) void*)(xdst + S::XPL + q * 1024), 16, 0, 0);
      }
    };
    for (int i = 0; i <= totalSub; ++i) {
      if (i < totalSub) {
        // ---- weights of sub-stage i -> weight buffer i & 1 ----
        if (kcW == 0 && rW == 0) {
          int n, y0, x0, ksW;
          x3_decode<TW_>(wW, a, n, y0, x0, coTileW, ksW);
          kBaseW = ksW * a.nChunks;
        }
        {
          char* wdst = reinterpret_cast<char*>(smemv) + S::WOFF + (i & 1) * S::WST;
          const uint16_t* wsrc =
              a.wt + (((size_t)coTileW * a.chunksTotal + kBaseW + kcW) * 3 + rW) * (size_t)(S::WST / 2) + lane * 8;
#pragma unroll
          for (int j = 0; j < 6; ++j)   // 24 pieces of 1 KiB: [plane][kx][cs]
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wsrc + (k + j * 4) * 512),
                (__attribute__((address_space(3))) void*)(wdst + (k + j * 4) * 1024), 16, 0, 0);
        }
        if (++rW == 3) {
          rW = 0;
          if (++kcW == a.nChunks) {
            kcW = 0;
            wW += G;
          }
        }
        // ---- halo: chunk c = ceil(i / 3) is read from sub-stage 3c on; its thirds go out at i = 3c-2, 3c-1, 3c
        //      (i = 0: all of chunk 0) ----
        if (i == 0) {
          halo_ptrs();
          issue_halo(0, 0);
          issue_halo(1, 0);
          issue_halo(2, 0);
        } else {
          const int c = (i + 2) / 3, part3 = (i + 2) - 3 * c;
          if (c * 3 < totalSub) {
            if (part3 == 0) {   // first third: advance to chunk c
              if (++kcX == a.nChunks) {
                kcX = 0;
                wX += G;
              }
              if (kcX == 0) halo_ptrs();
            }
            switch (part3) {   // constant piece indices at every issue site: the pointer array stays in registers
              case 0: issue_halo(0, c & 1); break;
              case 1: issue_halo(1, c & 1); break;
              default: issue_halo(2, c & 1); break;
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int li = lane & 15, lq = lane >> 4;
  constexpr int NCB = TW / 16;   // column blocks of 16 pixels: 2 / 1
  // fragment ms of this wave: tile row wave*ROWS_PER_WAVE + ms / NCB, column block ms % NCB.
  // xa[cb][kx][rho]: byte address (chunk buffer 0, hi plane) of this lane's 16-byte piece at tap column kx in the
  // wave's first row, for LDS rows of parity rho; fragment row and tap row add (row + ky) * P * 64.
  int xa[NCB][3][2];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int rho = 0; rho < 2; ++rho) {
        const int col = cb * 16 + li + kx;
        const int key = ((4 * rho + (col & 7)) >> 2) & 1;
        xa[cb][kx][rho] = S::XOFF + (wave * S::ROWS_PER_WAVE * P + col) * 64 + ((lq ^ (key << 1)) << 4);
        asm volatile("" : "+v"(xa[cb][kx][rho]));
      }
  int wa[2] = {S::WOFF + lane * 16, S::WOFF + S::WST + lane * 16};
  asm volatile("" : "+v"(wa[0]), "+v"(wa[1]));
  const char* lds = reinterpret_cast<const char*>(smemv);
  {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smemv) + S::TOFF);
    const float ds = a.dynScale ? *a.dynScale : 1.f;
    for (int c = tid; c < a.Cout; c += 256) {
      tab[c] = a.scale[c] * ds;
      tab[S::MAX_COUT + c] = a.shift[c];
    }
    if (tid < 64) tab[2 * S::MAX_COUT + tid] = a.headW ? a.headW[tid] : 0.f;
  }
  const unsigned outLane = (unsigned)(li * a.ldo + lq * 16) * 2u;
  const unsigned poolLane = (unsigned)((li >> 1) * a.Cout + lq * 16) * 2u;
  float amax = 0.f;
  ws_barrier();
  for (int w = lb; w < numWork; w += G) {
    int n, y0, x0, coTile, ks;
    x3_decode<TW_>(w, a, n, y0, x0, coTile, ks);
    f32x4 acc[4][4];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) acc[ms][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // FLAT: AND-masks (all ones / zero) of fragment ms for the first and the last tap row
    uint32_t mTop[4], mBot[4];
    if (FLAT) {
      const int yImg = (y0 + wave * S::ROWS_PER_WAVE) % a.imgH;   // image row of this wave's first tile row
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        const int yy = (yImg + ms / NCB) % a.imgH;
        mTop[ms] = yy == 0 ? 0u : 0xFFFFFFFFu;
        mBot[ms] = yy == a.imgH - 1 ? 0u : 0xFFFFFFFFu;
      }
    }
    for (int kc = 0; kc < a.nChunks; kc += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          // sub-stage (kc + h, r): halo buffer h, weight buffer (3h + r) & 1 (tiles hold an even number of chunks,
          // so a tile starts at an even sub-stage index)
          const int wb = (3 * h + r) & 1;
          auto xread = [&](int ms, int kx, int plane) __attribute__((always_inline)) -> f32x4 {
            const int row = ms / NCB + r;
            return *reinterpret_cast<const f32x4*>(lds + xa[ms % NCB][kx][row & 1] + h * S::XST + plane * S::XPL +
                                                   row * (P * 64));
          };
          auto wread = [&](int cs, int kx, int plane) __attribute__((always_inline)) -> f32x4 {
            return *reinterpret_cast<const f32x4*>(lds + wa[wb] + plane * S::WPL + (kx * 4 + cs) * 1024);
          };
          f32x4 xh[2][4], xl[2][4], wh[2][4], wl[2][4];   // ping-pong by tap parity
#pragma unroll
          for (int cs = 0; cs < 4; ++cs) {
            wh[0][cs] = wread(cs, 0, 0);
            wl[0][cs] = wread(cs, 0, 1);
          }
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) {
            xh[0][ms] = xread(ms, 0, 0);
            xl[0][ms] = xread(ms, 0, 1);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // the first tap's 16 reads go out back to back
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int cur = kx & 1, nxt = cur ^ 1;
            if (kx < 2) {
#pragma unroll
              for (int cs = 0; cs < 4; ++cs) {
                wh[nxt][cs] = wread(cs, kx + 1, 0);
                wl[nxt][cs] = wread(cs, kx + 1, 1);
              }
#pragma unroll
              for (int ms = 0; ms < 4; ++ms) {
                xh[nxt][ms] = xread(ms, kx + 1, 0);
                xl[nxt][ms] = xread(ms, kx + 1, 1);
              }
            }
            if (FLAT && r != 1) {   // taps that would read the neighbouring image's row contribute zero
              typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
              for (int ms = 0; ms < 4; ++ms) {
                const uint32_t m = r == 0 ? mTop[ms] : mBot[ms];
                xh[cur][ms] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, xh[cur][ms]) & m);
                xl[cur][ms] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, xl[cur][ms]) & m);
              }
            }
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
#pragma unroll
              for (int cs = 0; cs < 4; ++cs) {
                // small terms first
                acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[cur][cs]),
                                                                     __builtin_bit_cast(f16x8, xh[cur][ms]),
                                                                     acc[ms][cs], 0, 0, 0);
                acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[cur][cs]),
                                                                     __builtin_bit_cast(f16x8, xl[cur][ms]),
                                                                     acc[ms][cs], 0, 0, 0);
                acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[cur][cs]),
                                                                     __builtin_bit_cast(f16x8, xh[cur][ms]),
                                                                     acc[ms][cs], 0, 0, 0);
              }
            if (kx < 2) {
              // the 16 reads of the next tap go out one behind every third MFMA of this one
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
            } else {
              __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          ws_barrier();
        }
      }
    }

    // ---- epilogue straight from the accumulators: lane (li, lq) holds channels 64*coTile + 16*lq + [0,16) of
    //      pixel li of each fragment: acc[ms][cs][r] is channel 16*lq + 4*cs + r ----
    const int cbase = coTile * 64 + lq * 16;
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      sc[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (cbase + cs * 4) * 4);
      sh[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (S::MAX_COUT + cbase + cs * 4) * 4);
    }
    const float floorV = a.relu ? 0.f : -3.4e38f;
    const size_t g0 = (size_t)n * a.H + y0;   // global row of the tile's first row
    // vertical neighbours: fragment pairs (msA, msA + NCB) with msA = (j % NCB) + (j / NCB) * 2 * NCB, j < 2
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int msA = (j % NCB) + (j / NCB) * 2 * NCB;
      const int cb = msA % NCB;
      const int rA = wave * S::ROWS_PER_WAVE + msA / NCB;
      const bool okx = x0 + cb * 16 + li < a.W;
      float v[2][16];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ms = msA + NCB * u;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int cs = c >> 2, r = c & 3;
          v[u][c] = fmaxf(fmaf(acc[ms][cs][r], sc[cs][r], sh[cs][r]), floorV);
        }
      }
      uint32_t ph[2][8], pl[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) split_pk_f16(v[u][2 * i], v[u][2 * i + 1], ph[u][i], pl[u][i], amax);
      if (EPI == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool oky = y0 + rA + u < a.H;
          float* rowp = a.outF + (size_t)ks * a.splitStride +
                        ((g0 + rA + u) * a.W + x0 + cb * 16 + li) * (size_t)a.ldo + a.co_off + coTile * 64 + lq * 16;
          if (okx && oky) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(rowp + 4 * q) = (f32x4){v[u][4 * q], v[u][4 * q + 1], v[u][4 * q + 2], v[u][4 * q + 3]};
          }
        }
      }
      if (EPI == 0 || EPI == 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool oky = y0 + rA + u < a.H;
          char* rowp = reinterpret_cast<char*>(a.out) +
                       (((g0 + rA + u) * a.W + x0 + cb * 16) * (size_t)a.ldo + a.co_off + coTile * 64) * 2;
          if (okx && oky) {
            uint4* o = reinterpret_cast<uint4*>(rowp + outLane);
            o[0] = make_uint4(ph[u][0], ph[u][1], ph[u][2], ph[u][3]);
            o[1] = make_uint4(ph[u][4], ph[u][5], ph[u][6], ph[u][7]);
            uint4* ol = reinterpret_cast<uint4*>(rowp + outLane + a.outLo * 2);
            ol[0] = make_uint4(pl[u][0], pl[u][1], pl[u][2], pl[u][3]);
            ol[1] = make_uint4(pl[u][4], pl[u][5], pl[u][6], pl[u][7]);
          }
        }
      }
      if (EPI == 1) {
        // MaxPool2d(2,2) on the fp32 values: vertical neighbour in the partner fragment, horizontal in lane li ^ 1
        uint32_t qh[8], ql[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float m0 = fmaxf(v[0][2 * i], v[1][2 * i]), m1 = fmaxf(v[0][2 * i + 1], v[1][2 * i + 1]);
          m0 = fmaxf(m0, dpp_xor1_f(m0));
          m1 = fmaxf(m1, dpp_xor1_f(m1));
          split_pk_f16(m0, m1, qh[i], ql[i]);
        }
        const bool oky = y0 + rA + 1 < a.H;
        char* rowp = reinterpret_cast<char*>(a.pool) +
                     ((((g0 + rA) >> 1) * (size_t)(a.W >> 1) + ((x0 + cb * 16) >> 1)) * (size_t)a.Cout + coTile * 64) * 2;
        if (okx && oky && (li & 1) == 0) {
          uint4* o = reinterpret_cast<uint4*>(rowp + poolLane);
          o[0] = make_uint4(qh[0], qh[1], qh[2], qh[3]);
          o[1] = make_uint4(qh[4], qh[5], qh[6], qh[7]);
          uint4* ol = reinterpret_cast<uint4*>(rowp + poolLane + a.poolLo * 2);
          ol[0] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
          ol[1] = make_uint4(ql[4], ql[5], ql[6], ql[7]);
        }
      }
      if (EPI == 2) {
        // fused 1x1 head (reference README.md:1447) on hi + lo, as a consumer of the two planes would read it
        f32x4 hw[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs)
          hw[cs] = *reinterpret_cast<const f32x4*>(lds + S::TOFF + (2 * S::MAX_COUT + lq * 16 + cs * 4) * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float z = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float e0, e1;
            merge_pk_f16(ph[u][i], pl[u][i], e0, e1);
            z = fmaf(e0, hw[i >> 1][(2 * i) & 3], z);
            z = fmaf(e1, hw[i >> 1][(2 * i + 1) & 3], z);
          }
          z += __shfl_xor(z, 16, 64);
          z += __shfl_xor(z, 32, 64);
          z += a.headB;
          const bool oky = y0 + rA + u < a.H;
          if (okx && oky && lq == 0) {
            const size_t o = (g0 + rA + u) * a.W + x0 + cb * 16 + li;
            if (a.logits) a.logits[o] = z;
            if (a.probs) a.probs[o] = 1.f / (1.f + __expf(-z));
            if (a.mask) a.mask[o] = z > a.headThr ? 255 : 0;
          }
        }
      }
    }
  }
  if (EPI != 3) x3_report_range(amax, a.err);
}

// Second half of a split-K convolution: y = act(scale * sum_ks partial[ks] + shift) written as hi / lo planes (pixel
// stride ldo halfs, channel offset co_off).  partial: [kSplit][P][C] fp32, dense.
__global__ __launch_bounds__(256) void x3_splitk_finish_kernel(const float* __restrict__ partial, int kSplit, size_t P,
                                                               int C, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu,
                                                               uint32_t* __restrict__ outHi, size_t outLo2, int ldo,
                                                               int co_off, unsigned* err) {
  float amax = 0.f;
  const int c4 = C >> 2;
  const size_t total = P * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t slab = P * (size_t)C;
  const float floorV = relu ? 0.f : -3.4e38f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    // kSplit is a power of two >= 2: pairs of independent loads in flight, added in a fixed order
    const float* src = partial + p * C + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(src) + *reinterpret_cast<const f32x4*>(src + slab);
    for (int k = 2; k < kSplit; k += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + k * slab);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (k + 1) * slab);
      f32x4 a2 = (f32x4){0.f, 0.f, 0.f, 0.f}, a3 = a2;
      if (k + 2 < kSplit) {
        a2 = *reinterpret_cast<const f32x4*>(src + (k + 2) * slab);
        a3 = *reinterpret_cast<const f32x4*>(src + (k + 3) * slab);
      }
      v += (a0 + a1) + (a2 + a3);
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), floorV);
    uint32_t h0, l0, h1, l1;
    split_pk_f16(y[0], y[1], h0, l0, amax);
    split_pk_f16(y[2], y[3], h1, l1, amax);
    const size_t o = (p * (size_t)ldo + co_off + c) >> 1;
    *reinterpret_cast<uint2*>(outHi + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(outHi + outLo2 + o) = make_uint2(l0, l1);
  }
  x3_report_range(amax, err);
}

// Device-side repack of fp32 PyTorch-layout 3x3 weights into this kernel's hi/lo fragment order (training: after every
// optimizer step).  mode 0: forward, W(co, ci, t) = w[(co*cin + ci)*9 + t] with w (cout, cin, 3, 3); mode 1: input
// gradient, a convolution with the roles of the channels swapped and the taps flipped: its "cout" is the forward
// cin and W(n, k, t) = w[(k*coutD + n)*9 + (8 - t)] with w (cin = k range, coutD = n range, 3, 3).  No pre-scaling: an
// fp16 subnormal lo part costs a few bits of the 22, far inside the training tolerance.
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout,
                                                      int cin, int mode) {
  const int nCh = cin / 32;
  const size_t total = (size_t)(cout / 64) * nCh * 9 * 4 * 64;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int lane = (int)(i & 63);
    size_t r0 = i >> 6;
    const int cs = (int)(r0 & 3);
    r0 >>= 2;
    const int t = (int)(r0 % 9);
    r0 /= 9;
    const int kc = (int)(r0 % nCh);
    const int ct = (int)(r0 / nCh);
    const int j = lane & 15, lq = lane >> 4;
    const int co = 64 * ct + 16 * (j >> 2) + 4 * cs + (j & 3);
    const int tr = t / 3, kx = t - tr * 3;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ci = kc * 32 + lq * 8 + e2 * 2 + e;
        v[e] = mode == 0 ? w[((size_t)co * cin + ci) * 9 + t] : w[((size_t)ci * cout + co) * 9 + (8 - t)];
      }
      split_pk_f16(v[0], v[1], hi[e2], lo[e2]);
    }
    uint16_t* base = out + ((((size_t)ct * nCh + kc) * 3 + tr) * (size_t)(2 * 3 * 4 * 64 * 8));
    *reinterpret_cast<uint4*>(base + (((size_t)0 * 3 + kx) * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(base + (((size_t)1 * 3 + kx) * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// The same packing, one block per (channel tile of 64, chunk of 32) with the tile's 18,432 weights brought through LDS:
// they are 64 runs of 1,152 contiguous bytes (mode 0: rows co, 32 ci x 9 taps each) or 32 runs of 2,304 bytes (mode 1,
// the input-gradient operator: rows are the forward co = this operator's ci, 64 forward ci x 9 taps each), where
// pack_x3_kernel gathers them 4 bytes at a time (0.4 ms per training step for the 34 packs; this one is bound by the
// 0.5 GB it moves).  Launch with (cout / 64) * (cin / 32) blocks of 256 threads and kPackX3LdsBytes of dynamic LDS.
constexpr int kPackX3LdsBytes = 64 * (288 + 4) * 4;   // 74,752 (mode 1: 32 x (576 + 4) x 4 = 74,240)
__device__ __forceinline__ void pack_x3_lds_tile(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin,
                                                 int mode, int tileIdx, float* tileW) {
  const int nCh = cin / 32;
  const int ct = tileIdx / nCh, kc = tileIdx - ct * nCh;
  const int tid = threadIdx.x;
  // rows x rowLen floats of the source, row pitch rowLen + 4 in LDS
  const int rows = mode == 0 ? 64 : 32, rowLen = mode == 0 ? 288 : 576, pitch = rowLen + 4;
  for (int i = tid; i < rows * (rowLen / 4); i += 256) {
    const int r = i / (rowLen / 4), q = i - r * (rowLen / 4);
    // mode 0: w[(co * cin + ci) * 9 + t], co = 64 ct + r, ci from 32 kc; mode 1: w[(ci * cout + co) * 9 + t] with this
    // operator's ci = 32 kc + r as the forward co (row) and its co from 64 ct as the forward ci
    const size_t src = mode == 0 ? ((size_t)(64 * ct + r) * cin + 32 * kc) * 9 : ((size_t)(32 * kc + r) * cout + 64 * ct) * 9;
    *reinterpret_cast<float4*>(tileW + r * pitch + 4 * q) = *reinterpret_cast<const float4*>(w + src + 4 * q);
  }
  __syncthreads();
  for (int i = tid; i < 9 * 4 * 64; i += 256) {
    const int lane = i & 63;
    const int cs = (i >> 6) & 3;
    const int t = i >> 8;
    const int j = lane & 15, lq = lane >> 4;
    const int col = 16 * (j >> 2) + 4 * cs + (j & 3);   // this operator's co within the tile
    const int tr = t / 3, kx = t - tr * 3;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = lq * 8 + e2 * 2 + e;   // this operator's ci within the chunk
        v[e] = mode == 0 ? tileW[col * pitch + k * 9 + t] : tileW[k * pitch + col * 9 + (8 - t)];
      }
      split_pk_f16(v[0], v[1], hi[e2], lo[e2]);
    }
    uint16_t* base = out + ((((size_t)ct * nCh + kc) * 3 + tr) * (size_t)(2 * 3 * 4 * 64 * 8));
    *reinterpret_cast<uint4*>(base + (((size_t)0 * 3 + kx) * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(base + (((size_t)1 * 3 + kx) * 4 + cs) * 64 * 8 + lane * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

__global__ __launch_bounds__(256) void pack_x3_lds_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout,
                                                          int cin, int mode) {
  extern __shared__ __attribute__((aligned(16))) float tileW[];
  pack_x3_lds_tile(w, out, cout, cin, mode, (int)blockIdx.x, tileW);
}

// All of a network's packs in one launch: block b belongs to the table entry e with start[e] <= b < start[e + 1] and
// is tile b - start[e] of it (the training step re-packs 34 operators after every optimizer step: as 34 launches they
// cost their launch gaps)
struct PackX3Desc {
  const float* w;
  uint16_t* out;
  int cout, cin, mode;
};
__global__ __launch_bounds__(256) void pack_x3_lds_multi_kernel(const PackX3Desc* __restrict__ descs,
                                                                const unsigned* __restrict__ start, int n) {
  extern __shared__ __attribute__((aligned(16))) float tileW[];
  int lo = 0, hi = n - 1;   // the entry whose block range holds blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const PackX3Desc d = descs[lo];
  pack_x3_lds_tile(d.w, d.out, d.cout, d.cin, d.mode, (int)(blockIdx.x - start[lo]), tileW);
}

// ---- plane helpers (test entry points and the unfused fallbacks) ----

// fp32 NHWC (pixel stride ld, `c` channels used) -> hi/lo planes with the same geometry
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, size_t nvec2,
                                                           uint32_t* __restrict__ hi, uint32_t* __restrict__ lo,
                                                           unsigned* err = nullptr) {
  float amax = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec2; i += stride) {
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    uint32_t h, l;
    split_pk_f16(v.x, v.y, h, l, amax);
    hi[i] = h;
    lo[i] = l;
  }
  x3_report_range(amax, err);
}

// The same split for a tensor whose magnitudes sit far below the fp16 range (activation gradients: ~1e-7): every
// value is first multiplied by 2^k with k chosen from the tensor's max |x| (`absmaxKey`: its float bits, produced by
// the kernel that wrote x) so that the maximum lands in [2^13, 2^14); 2^-k goes to *invOut for the consumer's epilogue.
// Power-of-two scaling is exact.
__global__ __launch_bounds__(256) void split_planes_scaled_kernel(const float* __restrict__ x, size_t nvec2,
                                                                  uint32_t* __restrict__ hi, uint32_t* __restrict__ lo,
                                                                  const unsigned* __restrict__ absmaxKey,
                                                                  float* __restrict__ invOut) {
  const unsigned key = *absmaxKey;
  int k = 0;
  if ((key >> 23) != 0 && (key >> 23) < 255) k = 13 - ((int)(key >> 23) - 127);   // normal, finite maximum
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float up = ldexpf(1.f, k);
  if (blockIdx.x == 0 && threadIdx.x == 0) *invOut = ldexpf(1.f, -k);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec2; i += stride) {
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    uint32_t h, l;
    split_pk_f16(v.x * up, v.y * up, h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

__global__ __launch_bounds__(256) void merge_planes_kernel(const uint32_t* __restrict__ hi,
                                                           const uint32_t* __restrict__ lo, size_t nvec2,
                                                           float* __restrict__ y) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec2; i += stride) {
    float a, b;
    merge_pk_f16(hi[i], lo[i], a, b);
    reinterpret_cast<float2*>(y)[i] = make_float2(a, b);
  }
}

// MaxPool2d(2,2) on planes (fallback when the producer could not fuse it): x (N,H,W,ldi) -> y (N,H/2,W/2,C)
__global__ __launch_bounds__(256) void maxpool2x2_planes_kernel(const uint32_t* __restrict__ hi, size_t inLo2, int n,
                                                                int h, int w, int c, int ldi,
                                                                uint32_t* __restrict__ out, size_t outLo2) {
  const int c2 = c / 2, ld2 = ldi / 2;
  const size_t total = (size_t)n * (h / 2) * (w / 2) * c2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cc = (int)(i % c2);
    size_t p = i / c2;
    const int xo = (int)(p % (w / 2));
    p /= (w / 2);
    const int yo = (int)(p % (h / 2));
    const int nn = (int)(p / (h / 2));
    float m0 = -3.4e38f, m1 = -3.4e38f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const size_t o = (((size_t)nn * h + 2 * yo + dy) * w + 2 * xo + dx) * ld2 + cc;
        float a, b;
        merge_pk_f16(hi[o], hi[o + inLo2], a, b);
        m0 = fmaxf(m0, a);
        m1 = fmaxf(m1, b);
      }
    uint32_t qh, ql;
    split_pk_f16(m0, m1, qh, ql);
    out[i] = qh;
    out[i + outLo2] = ql;
  }
}

// 1x1 head on planes (fallback when the last convolution could not fuse it): x (npix, C) planes -> logits
__global__ __launch_bounds__(256) void head1x1_planes_kernel(const uint32_t* __restrict__ hi, size_t inLo2,
                                                             const float* __restrict__ w, float bias, size_t npix,
                                                             int c, float* __restrict__ logits,
                                                             float* __restrict__ probs, uint8_t* __restrict__ mask,
                                                             float thr) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int c2 = c / 2;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    float z = bias;
    for (int i = 0; i < c2; ++i) {
      float a, b;
      merge_pk_f16(hi[p * c2 + i], hi[p * c2 + i + inLo2], a, b);
      z = fmaf(a, w[2 * i], z);
      z = fmaf(b, w[2 * i + 1], z);
    }
    if (logits) logits[p] = z;
    if (probs) probs[p] = 1.f / (1.f + __expf(-z));
    if (mask) mask[p] = z > thr ? 255 : 0;
  }
}

}  // namespace unet
