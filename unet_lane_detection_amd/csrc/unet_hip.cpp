// libunet_hip.so - host side of the C ABI declared in include/unet_hip.h.
//
// Owns: parameter intake in PyTorch layout, BatchNorm folding, packing of conv
// weights into MFMA fragment order, the activation workspace and the launch
// sequence of the forward pass (reference README.md:1460-1481).  All
// arithmetic on activations happens in the HIP kernels of igemm_f32.h and
// elementwise.h; there is no CPU fallback.
#include "../../include/unet_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "elementwise.h"
#include "igemm_f32.h"
#include "igemm_bf16.h"
#include "conv_bf16_ws.h"
#include "conv_first_bf16x3.h"
#include "upconv_bf16_ws.h"
#include "upconv_bf16_r512.h"
#include "conv_x3_ws.h"
#include "conv_x3_r512.h"
#include "conv_x3_t448.h"
#include "conv_q8_r512.h"
#include "conv_bf16_r512.h"
#include "upconv_x3_ws.h"
#include "upconv_x3_r512.h"
#include "conv_first_x3.h"
#include "conv_i8.h"
#include "wino_f32.h"
#include "train_kernels.h"
#include "wgrad_f32.h"
#include "wgrad_wino_f32.h"
#include "wgrad_gemm_f32.h"
#include "wgrad_x3_ws.h"
#include "camera_stage.h"

namespace {

using unet::ConvArgs;

constexpr float kBnEps = 1e-5f;  // nn.BatchNorm2d default (reference README.md:1453)

#define HIPCHK(ctx_err, expr)                                                                   \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx_err) = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
      return UNET_ERR_HIP;                                                                      \
    }                                                                                           \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
// Optional per-launch timing: a hipEvent pair around every kernel launch, recorded on the stream
// the kernel runs on.  Used by bench.py for the live roofline numbers; off by default.
// ------------------------------------------------------------------------------------------
struct ProfRecord {
  std::string name;
  double flops = 0, bytes = 0, ms = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
struct Profiler {
  bool on = false;
  std::vector<ProfRecord> recs;
  void begin(const char* name, double flops, double bytes, hipStream_t s) {
    if (!on) return;
    ProfRecord r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s);
    recs.push_back(r);
  }
  void end(hipStream_t s) {
    if (!on) return;
    hipEventRecord(recs.back().e1, s);
  }
  void resolve() {
    for (auto& r : recs)
      if (r.e0) {
        hipEventSynchronize(r.e1);
        float ms = 0;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        r.ms = ms;
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
        r.e0 = r.e1 = nullptr;
      }
  }
  void clear() {
    resolve();
    recs.clear();
  }
};
// Per-call launch context of the calling thread: the handle's profiler (or none) and its device error word.
// thread_local, so two handles driven from two threads do not see each other's scope.
thread_local Profiler* g_prof = nullptr;
// the handle's device-visible error block: word 0 = a bounded spin gave up, word 1 = an activation of the f16x3 tier left
// the fp16 range (conv_x3_ws.h, range watch)
thread_local unsigned* g_errWord = nullptr;

// Kernels that need more dynamic LDS than the default opt in once per (device, kernel): the attribute is per device.
hipError_t ensure_dyn_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, fn})) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({dev, fn});
  return e;
}

// Device error word for launches made outside a handle (the single-operator test entry points): one per device.
unsigned* op_err_word() {
  static std::mutex mu;
  static std::map<int, unsigned*> words;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  auto it = words.find(dev);
  if (it != words.end()) return it->second;
  unsigned* w = nullptr;
  if (hipMalloc((void**)&w, 2 * sizeof(unsigned)) != hipSuccess) return nullptr;
  hipMemset(w, 0, 2 * sizeof(unsigned));
  words[dev] = w;
  return w;
}

inline void prof_begin(const char* n, double fl, double by, hipStream_t s) { if (g_prof) g_prof->begin(n, fl, by, s); }
inline void prof_end(hipStream_t s) { if (g_prof) g_prof->end(s); }

// ------------------------------------------------------------------------------------------
// One GEMM-shaped operator (3x3 conv or 2x2 transposed conv) with device-resident packed weights
// ------------------------------------------------------------------------------------------
struct GemmOp {
  int taps = 9;      // 9: conv3x3, 1: upconv 2x2
  int cinReal = 0;   // channels of the PyTorch tensor
  int cin = 0;       // padded to CK
  int cout = 0;      // real channels (per (a,b) group for upconv)
  int coutPad = 0;   // rounded up to 16
  int nTotal = 0;    // GEMM N padded to BN
  int ck = 16;
  int relu = 0;
  const char* name = nullptr;  // profiler label override for the direct kernel
  const char* nameWino = nullptr;  // ... and for the Winograd kernel on the same operator
  int plain = 0;     // taps == 1 only: 1 = ordinary 1x1 GEMM output (no pixel-shuffle scatter)
  float* wt = nullptr;
  float* wtWino = nullptr;  // Winograd F(2x2,3x3) transformed weights (3x3 convs with Cin % 16 == 0)
  float* scale = nullptr;
  float* shift = nullptr;
  void free_dev() {
    if (wt) hipFree(wt);
    if (wtWino) hipFree(wtWino);
    wtWino = nullptr;
    if (scale) hipFree(scale);
    if (shift) hipFree(shift);
    wt = scale = shift = nullptr;
  }
};

struct TileChoice {
  int ms, th, tw, nld;
};

// Pick the pixel tile: BM = 32*MS pixels as TH global rows x TW columns.  Candidates are the
// factorizations of 128 (MS=4) and 224 (MS=7); cost = padded work / useful work, then halo size.
TileChoice choose_tile(int nh, int w, int ck, bool halo) {
  TileChoice best{4, 8, 16, 4};
  double bestCost = 1e30;
  for (int ms : {7, 4}) {
    const int bm = 32 * ms;
    const int nld = (ck == 16) ? (ms == 7 ? 6 : 4) : (ms == 7 ? 2 : 1);
    const int vpp = ck / 4;
    for (int tw = 2; tw <= bm; ++tw) {
      if (bm % tw) continue;
      const int th = bm / tw;
      const int hp = halo ? (th + 2) * (tw + 2) : th * tw;
      if (hp * vpp > nld * 256) continue;
      const double padW = (double)((w + tw - 1) / tw * tw) / w;
      const double padH = (double)((nh + th - 1) / th * th) / nh;
      const double haloF = (double)hp / (th * tw);
      // padded MFMA work dominates; halo traffic is a mild tie-breaker; prefer the larger tile
      const double cost = padW * padH * (1.0 + 0.02 * haloF) * (ms == 7 ? 0.97 : 1.0);
      if (cost < bestCost) {
        bestCost = cost;
        best = {ms, th, tw, nld};
      }
    }
  }
  return best;
}

template <int CK, int TAPS, int MS, int NS, int NLD, int MODE>
hipError_t launch_one(const ConvArgs& a, dim3 grid, hipStream_t s) {
  // two halo-tile buffers + zero slot, or the epilogue's transposed half tile, whichever is larger
  const size_t lds = std::max<size_t>((size_t)2 * NLD * 256 * 16 + 64, (size_t)MS * 16 * (32 * NS + 4) * 4);
  hipLaunchKernelGGL((unet::igemm_f32_kernel<CK, TAPS, MS, NS, NLD, MODE>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

template <int CK, int TAPS, int MODE>
hipError_t launch_cfg(const ConvArgs& a, dim3 grid, int ms, int ns, hipStream_t s) {
  constexpr int NLD7 = (CK == 16) ? 6 : 2;
  constexpr int NLD4 = (CK == 16) ? 4 : 1;
  if (ms == 7) return launch_one<CK, TAPS, 7, 2, NLD7, MODE>(a, grid, s);
  if (ms == 4 && ns == 4) return launch_one<CK, TAPS, 4, 4, NLD4, MODE>(a, grid, s);
  return launch_one<CK, TAPS, 4, 2, NLD4, MODE>(a, grid, s);
}

std::atomic<int> g_winoMode{-1};  // -1: read UNET_NO_WINOGRAD once; 0 / 1: forced by unet_set_winograd
bool wino_enabled() {
  if (g_winoMode < 0) {
    const char* e = getenv("UNET_NO_WINOGRAD");
    g_winoMode = (e && e[0] == '1') ? 0 : 1;
  }
  return g_winoMode == 1;
}

struct WinoTile {
  int tht, twt;
};

// Tile grid of one Winograd block: THt x TWt tiles (<= 128), raw halo (2THt+2)(2TWt+2) <= 640 pixels.
WinoTile choose_wino_tile(int gt, int wt) {
  WinoTile best{8, 16};
  double bestCost = 1e30;
  for (int twt = 1; twt <= unet::WINO_TILES; ++twt) {
    const int tht = unet::WINO_TILES / twt;
    if ((2 * tht + 2) * (2 * twt + 2) > unet::WINO_NLD * unet::WINO_THREADS / 4) continue;
    const double padW = (double)((wt + twt - 1) / twt * twt) / wt;
    const double padH = (double)((gt + tht - 1) / tht * tht) / gt;
    const double fill = (double)unet::WINO_TILES / (tht * twt);
    const double halo = (double)(2 * tht + 2) * (2 * twt + 2) / (4.0 * tht * twt);
    const double cost = padW * padH * fill * (1.0 + 0.03 * halo);
    if (cost < bestCost) {
      bestCost = cost;
      best = {tht, twt};
    }
  }
  return best;
}

bool wino_applicable(const GemmOp& op, int h, int w) {
  return op.taps == 9 && op.wtWino && (h % 2 == 0) && (w % 2 == 0) && wino_enabled();
}
// the Winograd kernel addresses its input with 32-bit float4 indices
bool wino_fits(const GemmOp& op, int n, int h, int w) { return (double)n * h * w * op.cin < 17179869184.0; }

hipError_t run_wino(const GemmOp& op, const float* in, int n, int h, int w, float* out, int ldo, int coOff,
                    float* pool, hipStream_t s) {
  unet::WinoArgs a;
  a.in = in;
  a.wt = op.wtWino;
  a.scale = op.scale;
  a.shift = op.shift;
  a.out = out;
  a.pool = pool;
  a.N = n;
  a.H = h;
  a.W = w;
  a.Cin = op.cin;
  a.Cout = op.cout;
  a.ldo = ldo;
  a.co_off = coOff;
  const int gt = n * (h / 2), wt = w / 2;
  const WinoTile t = choose_wino_tile(gt, wt);
  a.THt = t.tht;
  a.TWt = t.twt;
  a.tilesX = (wt + t.twt - 1) / t.twt;
  a.nChunks = op.cin / 16;
  a.relu = op.relu;
  a.pixTiles = a.tilesX * ((gt + t.tht - 1) / t.tht);
  a.coTiles = op.nTotal / 32;
  a.coGroup = 1;
  for (int g : {8, 4, 2})
    if (a.coTiles % g == 0) {
      a.coGroup = g;
      break;
    }
  // keep a pixel tile's channel tiles on one XCD so the input tile crosses the fabric once, not once per XCD
  // (measured +0.6 % frames/s over spreading them, on every layer; UNET_WINO_XCD=0 restores the spread)
  static const int xcdMode = [] { const char* e = getenv("UNET_WINO_XCD"); return e ? atoi(e) : -1; }();
  a.xcdLocal = xcdMode >= 0 ? xcdMode : 1;
  a.err = g_errWord ? g_errWord : op_err_word();
  {  // 144 KiB of dynamic LDS needs the opt-in
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void*>(&unet::wino_f32_kernel<2>), unet::WINO_LDS_BYTES);
    if (ea != hipSuccess) return ea;
  }
  const double px = (double)n * h * w;
  prof_begin(op.nameWino ? op.nameWino : "conv3x3_wino_f32", 2.0 * px * 9 * op.cinReal * op.cout,
             4.0 * (px * op.cinReal + px * op.cout + 9.0 * op.cinReal * op.cout), s);
  hipLaunchKernelGGL((unet::wino_f32_kernel<2>), dim3((unsigned)((size_t)a.pixTiles * a.coTiles)),
                     dim3(unet::WINO_THREADS), unet::WINO_LDS_BYTES, s, a);
  prof_end(s);
  return hipGetLastError();
}

// in: (N,H,W,op.cin) -> out with pixel stride ldo at channel offset coOff
hipError_t run_gemm_op(const GemmOp& op, const float* in, int n, int h, int w, float* out, int ldo, int coOff,
                       hipStream_t s, int outBf16 = 0) {
  if (!outBf16 && wino_applicable(op, h, w) && wino_fits(op, n, h, w))
    return run_wino(op, in, n, h, w, out, ldo, coOff, nullptr, s);
  const TileChoice t = choose_tile(n * h, w, op.ck, op.taps == 9);
  ConvArgs a;
  a.in = in;
  a.wt = op.wt;
  a.scale = op.scale;
  a.shift = op.shift;
  a.out = out;
  a.N = n;
  a.H = h;
  a.W = w;
  a.Cin = op.cin;
  a.Cout = op.cout;
  a.CoutPad = op.coutPad;
  a.ldo = ldo;
  a.co_off = coOff;
  a.TH = t.th;
  a.TW = t.tw;
  a.tilesX = (w + t.tw - 1) / t.tw;
  a.nChunks = op.cin / op.ck;
  a.relu = op.relu;
  a.out_bf16 = outBf16;
  const int tilesY = (n * h + t.th - 1) / t.th;
  // The packed fragment order does not depend on NS, so the channel tile is a launch-time choice:
  // 128 columns with the 128-pixel tile, 64 with the 224-pixel tile (224x128 does not fit 256 VGPRs).
  const int ns = (t.ms == 4 && op.nTotal % 128 == 0) ? 4 : 2;
  a.pixTiles = a.tilesX * tilesY;
  a.coTiles = op.nTotal / (32 * ns);
  a.coGroup = 1;
  for (int g : {8, 4, 2})
    if (a.coTiles % g == 0) {
      a.coGroup = g;
      break;
    }
  dim3 grid((unsigned)((size_t)a.pixTiles * a.coTiles));
  // algorithmic work of this launch: 2*MAC flops on the real (unpadded) channel counts; bytes = read the
  // input once + write the output once + the weights once
  const double px = (double)n * h * w;
  const double nOut = (op.taps == 9 || op.plain) ? op.cout : 4.0 * op.cout;
  const double flops = 2.0 * px * op.taps * op.cinReal * nOut;
  const double bytes = 4.0 * (px * op.cinReal + px * nOut + (double)op.taps * op.cinReal * nOut);
  prof_begin(op.name ? op.name : (op.taps == 9 ? "conv3x3_igemm_f32" : "upconv2x2_igemm_f32"), flops, bytes, s);
  hipError_t e;
  if (op.taps == 9) {
    e = (op.ck == 16) ? launch_cfg<16, 9, 0>(a, grid, t.ms, ns, s) : launch_cfg<4, 9, 0>(a, grid, t.ms, ns, s);
  } else {
    if (op.plain)
      e = (op.ck == 16) ? launch_cfg<16, 1, 0>(a, grid, t.ms, ns, s) : launch_cfg<4, 1, 0>(a, grid, t.ms, ns, s);
    else
      e = (op.ck == 16) ? launch_cfg<16, 1, 1>(a, grid, t.ms, ns, s) : launch_cfg<4, 1, 1>(a, grid, t.ms, ns, s);
  }
  prof_end(s);
  return e;
}

// Pack W into [subtile][chunk][tap][lane][KPL]; column -> (source fetch) is given by `col(n, ci, tap)`.
template <class F>
std::vector<float> pack_fragments(int nTotal, int cin, int ck, int taps, F&& col) {
  const int kpl = ck / 4;
  const int nSub = nTotal / 16;
  const int nChunks = cin / ck;
  std::vector<float> out(((size_t)nSub * nChunks * taps + 1) * 64 * kpl, 0.f);  // +1 spare fragment (prefetch overrun)
  for (int cs = 0; cs < nSub; ++cs)
    for (int kc = 0; kc < nChunks; ++kc)
      for (int t = 0; t < taps; ++t) {
        float* dst = out.data() + (((size_t)cs * nChunks + kc) * taps + t) * 64 * kpl;
        for (int lane = 0; lane < 64; ++lane) {
          const int n = cs * 16 + (lane & 15);
          for (int e = 0; e < kpl; ++e) {
            const int ci = kc * ck + (lane >> 4) * kpl + e;
            dst[lane * kpl + e] = col(n, ci, t);
          }
        }
      }
  return out;
}

int upload(std::string& err, float** dst, const std::vector<float>& v) {
  HIPCHK(err, hipMalloc((void**)dst, v.size() * sizeof(float)));
  HIPCHK(err, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return UNET_OK;
}

// conv3x3: w (O,I,3,3), scale/shift per O
int build_conv3x3(std::string& err, GemmOp& op, const float* w, int cout, int cinReal, const float* scale,
                  const float* shift, int relu) {
  op.free_dev();
  op.taps = 9;
  op.cinReal = cinReal;
  op.ck = (cinReal % 16 == 0) ? 16 : 4;
  op.cin = round_up(cinReal, op.ck);
  op.cout = cout;
  op.coutPad = round_up(cout, 16);
  op.nTotal = round_up(cout, cout >= 128 ? 128 : 64);
  op.relu = relu;
  auto packed = pack_fragments(op.nTotal, op.cin, op.ck, 9, [&](int n, int ci, int t) -> float {
    if (n >= cout || ci >= cinReal) return 0.f;
    return w[((size_t)n * cinReal + ci) * 9 + t];
  });
  std::vector<float> sc(op.nTotal, 0.f), sh(op.nTotal, 0.f);
  for (int i = 0; i < cout; ++i) {
    sc[i] = scale[i];
    sh[i] = shift[i];
  }
  int rc;
  if ((rc = upload(err, &op.wt, packed))) return rc;
  if (op.ck == 16) {
    // Winograd F(2x2,3x3): U = G g G^T per (co, ci), computed in double
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    auto wino = pack_fragments(op.nTotal, op.cin, 16, 16, [&](int n, int ci, int p) -> float {
      if (n >= cout || ci >= cinReal) return 0.f;
      const float* g = w + ((size_t)n * cinReal + ci) * 9;
      const int pa = p >> 2, pb = p & 3;
      double u = 0.0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) u += G[pa][a] * (double)g[a * 3 + b] * G[pb][b];
      return (float)u;
    });
    if ((rc = upload(err, &op.wtWino, wino))) return rc;
  }
  if ((rc = upload(err, &op.scale, sc))) return rc;
  return upload(err, &op.shift, sh);
}

// upconv 2x2 s2: w (I,O,2,2), bias per O. GEMM column n = (a*2+b)*coutPad + co.
int build_upconv(std::string& err, GemmOp& op, const float* w, int cinReal, int cout, const float* bias) {
  op.free_dev();
  op.taps = 1;
  op.cinReal = cinReal;
  op.ck = (cinReal % 16 == 0) ? 16 : 4;
  op.cin = round_up(cinReal, op.ck);
  op.cout = cout;
  op.coutPad = round_up(cout, 16);
  op.nTotal = round_up(4 * op.coutPad, 4 * op.coutPad >= 128 ? 128 : 64);
  op.relu = 0;
  const int cp = op.coutPad;
  auto packed = pack_fragments(op.nTotal, op.cin, op.ck, 1, [&](int n, int ci, int) -> float {
    const int ab = n / cp, co = n % cp;
    if (ab >= 4 || co >= cout || ci >= cinReal) return 0.f;
    return w[((size_t)ci * cout + co) * 4 + ab];
  });
  std::vector<float> sc(op.nTotal, 0.f), sh(op.nTotal, 0.f);
  for (int n = 0; n < 4 * cp; ++n) {
    const int co = n % cp;
    if (co < cout) {
      sc[n] = 1.f;
      sh[n] = bias[co];
    }
  }
  int rc;
  if ((rc = upload(err, &op.wt, packed))) return rc;
  if ((rc = upload(err, &op.scale, sc))) return rc;
  return upload(err, &op.shift, sh);
}

inline unsigned grid_for(size_t items, int perBlock = 256) {
  size_t b = (items + perBlock - 1) / perBlock;
  return (unsigned)std::min<size_t>(std::max<size_t>(b, 1), 256 * 8 * 4);
}

hipError_t run_maxpool(const float* in, float* out, int n, int h, int w, int c, int ldi, hipStream_t s) {
  const size_t total = (size_t)n * (h / 2) * (w / 2) * (c / 4);
  prof_begin("maxpool2x2", 0.0, 4.0 * total * 4 * 5, s);
  hipLaunchKernelGGL(unet::maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, n, h, w, c, ldi);
  prof_end(s);
  return hipGetLastError();
}

hipError_t run_head(const float* in, const float* w, float bias, size_t npix, int c, float* logits, float* probs,
                    uint8_t* mask, float thr, hipStream_t s, const float* biasPtr = nullptr) {
  int lpp = 1;
  while (lpp < 16 && lpp * 2 * 4 <= c) lpp *= 2;
  const unsigned g = grid_for(npix, 256 / lpp);
  prof_begin("head1x1", 2.0 * npix * c, 4.0 * npix * (c + 1), s);
#define HEAD(L)                                                                                              \
  hipLaunchKernelGGL((unet::head1x1_kernel<L>), dim3(g), dim3(256), 0, s, in, w, bias, biasPtr, npix, c, logits, \
                     probs, mask, thr)
  switch (lpp) {
    case 1: HEAD(1); break;
    case 2: HEAD(2); break;
    case 4: HEAD(4); break;
    case 8: HEAD(8); break;
    default: HEAD(16); break;
  }
#undef HEAD
  prof_end(s);
  return hipGetLastError();
}

struct ParamSpec {
  std::string name;
  size_t numel;
};

}  // namespace

// ------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------
struct TrainState;
struct Bf16Net;
struct X3Net;
static void train_free(unet_ctx* h);
static void bf16_free(unet_ctx* h);
static void x3_free(unet_ctx* h);

struct unet_ctx {
  TrainState* train = nullptr;
  Bf16Net* bf = nullptr;
  X3Net* x3 = nullptr;
  unet_config cfg{};
  std::vector<ParamSpec> spec;
  std::map<std::string, std::vector<float>> params;
  bool finalized = false;

  std::vector<GemmOp> enc;  // 2 per level
  std::vector<GemmOp> bott; // 2
  std::vector<GemmOp> up;   // 1 per level (decoder order: deepest first)
  std::vector<GemmOp> dec;  // 2 per level (decoder order)
  float* headW = nullptr;
  float headB = 0.f;

  // workspace
  char* ws = nullptr;
  size_t wsBytes = 0;

  std::string err;
  Profiler prof;
  // Device-visible error block (pinned host memory mapped into the device).  Word 0: a kernel whose bounded
  // wave-progress wait gives up stores a non-zero value here instead of silently continuing with stale data; the next
  // entry point that sees it returns UNET_ERR_HIP and clears it (the failing launch may be an earlier one: like an
  // asynchronous HIP error).  Word 1: an f16x3 kernel had to store an activation beyond the fp16 range
  // (conv_x3_ws.h); only unet_device_error, which synchronises first and can therefore name the call, reports it
  // (UNET_ERR_RANGE) and clears it.
  unsigned* errHost = nullptr;
  unsigned* errDev = nullptr;
  unsigned* rangeKeys = nullptr;   // calibration pass (unet_forward_u8_ranges): 2 order keys per activation tensor
  void ensure_err_word() {
    if (errHost) return;
    if (hipHostMalloc((void**)&errHost, 2 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
      errHost = nullptr;
      return;
    }
    errHost[0] = errHost[1] = 0;
    if (hipHostGetDevicePointer((void**)&errDev, errHost, 0) != hipSuccess) errDev = nullptr;
  }
  int async_error() {
    volatile unsigned* w = errHost;
    if (w && w[0]) {
      w[0] = 0;   // reported once: a transient failure must not fail every later call
      err = "a bounded wave-progress wait timed out inside a kernel: the results of that launch are invalid";
      return UNET_ERR_HIP;
    }
    return UNET_OK;
  }
  // after a device synchronisation: everything the launches so far have reported, cleared
  int take_device_status() {
    volatile unsigned* w = errHost;
    if (!w) return UNET_OK;
    const unsigned timeout = w[0], range = w[1];
    w[0] = 0;
    w[1] = 0;
    if (timeout) {
      err = "a bounded wave-progress wait timed out inside a kernel: the results of that launch are invalid";
      return UNET_ERR_HIP;
    }
    if (range) {
      err = "f16x3 tier: an activation left the fp16 range (|v| > 65504); the results are not at fp32 parity - run "
            "these frames on the fp32 tier";
      return UNET_ERR_RANGE;
    }
    return UNET_OK;
  }

  void free_all() {
    if (errHost) hipHostFree(errHost);
    errHost = errDev = nullptr;
    for (auto* v : {&enc, &bott, &up, &dec})
      for (auto& op : *v) op.free_dev();
    if (headW) hipFree(headW);
    headW = nullptr;
    if (ws) hipFree(ws);
    ws = nullptr;
    wsBytes = 0;
  }
};

namespace {

// Profiler and error word of handle `h` become the calling thread's launch context for the scope's lifetime.
struct LaunchScope {
  Profiler* prevProf;
  unsigned* prevErr;
  explicit LaunchScope(unet_ctx* h) : prevProf(g_prof), prevErr(g_errWord) {
    h->ensure_err_word();
    g_prof = h->prof.on ? &h->prof : nullptr;
    g_errWord = h->errDev;
  }
  ~LaunchScope() {
    g_prof = prevProf;
    g_errWord = prevErr;
  }
};

void add_double_conv(std::vector<ParamSpec>& spec, const std::string& prefix, int cin, int cout) {
  const int convIdx[2] = {0, 3}, bnIdx[2] = {1, 4};
  for (int k = 0; k < 2; ++k) {
    const int ci = k == 0 ? cin : cout;
    spec.push_back({prefix + "." + std::to_string(convIdx[k]) + ".weight", (size_t)cout * ci * 9});
    for (const char* leaf : {"weight", "bias", "running_mean", "running_var"})
      spec.push_back({prefix + "." + std::to_string(bnIdx[k]) + "." + leaf, (size_t)cout});
  }
}

// Same key set as the reference module's state_dict minus the integer num_batches_tracked counters.
std::vector<ParamSpec> build_spec(const unet_config& c) {
  std::vector<ParamSpec> spec;
  int cin = c.in_channels;
  for (int i = 0; i < c.depth; ++i) {
    add_double_conv(spec, "encoder_blocks." + std::to_string(i), cin, c.features[i]);
    cin = c.features[i];
  }
  for (int j = 0; j < c.depth; ++j) {
    const int f = c.features[c.depth - 1 - j];
    spec.push_back({"decoder_blocks." + std::to_string(2 * j) + ".weight", (size_t)2 * f * f * 4});
    spec.push_back({"decoder_blocks." + std::to_string(2 * j) + ".bias", (size_t)f});
    add_double_conv(spec, "decoder_blocks." + std::to_string(2 * j + 1), 2 * f, f);
  }
  add_double_conv(spec, "bottleneck", c.features[c.depth - 1], 2 * c.features[c.depth - 1]);
  spec.push_back({"output.weight", (size_t)c.out_channels * c.features[0]});
  spec.push_back({"output.bias", (size_t)c.out_channels});
  return spec;
}

// Workspace layout for (n,h,w): offsets in floats.
struct WsPlan {
  size_t x0 = 0;                // (N,H,W,4)
  std::vector<size_t> cat, pool;  // per level
  size_t tmpA = 0, tmpB = 0;
  size_t total = 0;
};

WsPlan plan_ws(const unet_config& c, int n, int h, int w) {
  WsPlan p;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += (floats + 63) / 64 * 64;  // 256-byte granules
    return o;
  };
  const size_t px0 = (size_t)n * h * w;
  p.x0 = take(px0 * 4);
  size_t maxT = 0;
  for (int l = 0; l < c.depth; ++l) {
    const size_t px = px0 >> (2 * l);
    p.cat.push_back(take(px * 2 * c.features[l]));
    p.pool.push_back(take((px >> 2) * c.features[l]));
    maxT = std::max(maxT, px * c.features[l]);
  }
  maxT = std::max(maxT, (px0 >> (2 * c.depth)) * 2 * c.features[c.depth - 1]);
  p.tmpA = take(maxT);
  p.tmpB = take(maxT);
  p.total = off;
  return p;
}

int fold_bn_and_build(unet_ctx* h, GemmOp& op, const std::string& prefix, int convIdx, int bnIdx, int cin, int cout) {
  auto& P = h->params;
  const auto& w = P[prefix + "." + std::to_string(convIdx) + ".weight"];
  const std::string bn = prefix + "." + std::to_string(bnIdx) + ".";
  const auto &g = P[bn + "weight"], &b = P[bn + "bias"], &m = P[bn + "running_mean"], &v = P[bn + "running_var"];
  std::vector<float> sc(cout), sh(cout);
  for (int i = 0; i < cout; ++i) {
    // eval BatchNorm: (x - mean) / sqrt(var + eps) * gamma + beta (reference README.md:1453)
    const float inv = 1.0f / std::sqrt(v[i] + kBnEps);
    sc[i] = g[i] * inv;
    sh[i] = b[i] - m[i] * sc[i];
  }
  return build_conv3x3(h->err, op, w.data(), cout, cin, sc.data(), sh.data(), 1);
}

int forward_common(unet_ctx* h, int n, int height, int width, float* logits, float* probs, uint8_t* mask,
                   float thr, hipStream_t s, const WsPlan& p) {
  const unet_config& c = h->cfg;
  LaunchScope scope(h);
  float* ws = reinterpret_cast<float*>(h->ws);
  const float* cur = ws + p.x0;
  int ch = height, cw = width;
  float* tmpA = ws + p.tmpA;
  float* tmpB = ws + p.tmpB;
  // calibration pass: (min, max) of tensor `idx` (order: quant.tensor_names) right after it has been produced
  auto probe = [&](int idx, const float* x, size_t npx, int cc, int ld) {
    if (!h->rangeKeys) return;
    hipLaunchKernelGGL(unet::minmax_f32_kernel, dim3(grid_for(npx * cc)), dim3(256), 0, s, x, npx, cc, ld,
                       h->rangeKeys + 2 * idx);
  };
  const int D = c.depth;
  probe(0, cur, (size_t)n * height * width, 4, 4);
  for (int l = 0; l < c.depth; ++l) {
    const int f = c.features[l];
    float* cat = ws + p.cat[l];
    float* pool = ws + p.pool[l];
    HIPCHK(h->err, run_gemm_op(h->enc[2 * l], cur, n, ch, cw, tmpA, f, 0, s));
    probe(1 + 2 * l, tmpA, (size_t)n * ch * cw, f, f);
    if (wino_applicable(h->enc[2 * l + 1], ch, cw) && wino_fits(h->enc[2 * l + 1], n, ch, cw)) {
      // the 2x2 output tile of the Winograd kernel is one pooling window: pooled copy written from registers
      HIPCHK(h->err, run_wino(h->enc[2 * l + 1], tmpA, n, ch, cw, cat, 2 * f, 0, pool, s));
    } else {
      HIPCHK(h->err, run_gemm_op(h->enc[2 * l + 1], tmpA, n, ch, cw, cat, 2 * f, 0, s));  // skip half of concat
      HIPCHK(h->err, run_maxpool(cat, pool, n, ch, cw, f, 2 * f, s));
    }
    cur = pool;
    ch /= 2;
    cw /= 2;
  }
  const int fb = 2 * c.features[c.depth - 1];
  HIPCHK(h->err, run_gemm_op(h->bott[0], cur, n, ch, cw, tmpA, fb, 0, s));
  probe(1 + 2 * D, tmpA, (size_t)n * ch * cw, fb, fb);
  HIPCHK(h->err, run_gemm_op(h->bott[1], tmpA, n, ch, cw, tmpB, fb, 0, s));
  probe(2 + 2 * D, tmpB, (size_t)n * ch * cw, fb, fb);
  cur = tmpB;
  for (int j = 0; j < c.depth; ++j) {
    const int l = c.depth - 1 - j;
    const int f = c.features[l];
    float* cat = ws + p.cat[l];
    // ConvTranspose writes the upper channel half of the concat buffer: torch.cat([skip, x]) elided
    HIPCHK(h->err, run_gemm_op(h->up[j], cur, n, ch, cw, cat, 2 * f, f, s));
    ch *= 2;
    cw *= 2;
    probe(2 + 2 * l, cat, (size_t)n * ch * cw, 2 * f, 2 * f);   // skip half and upconv half: one concat tensor
    HIPCHK(h->err, run_gemm_op(h->dec[2 * j], cat, n, ch, cw, tmpA, f, 0, s));
    probe(3 + 2 * D + 2 * j, tmpA, (size_t)n * ch * cw, f, f);
    HIPCHK(h->err, run_gemm_op(h->dec[2 * j + 1], tmpA, n, ch, cw, tmpB, f, 0, s));
    probe(4 + 2 * D + 2 * j, tmpB, (size_t)n * ch * cw, f, f);
    cur = tmpB;
  }
  HIPCHK(h->err, run_head(cur, h->headW, h->headB, (size_t)n * height * width, c.features[0], logits, probs, mask,
                          thr, s));
  return h->async_error();
}

int check_shape(unet_ctx* h, int n, int height, int width) {
  const int m = 1 << h->cfg.depth;
  if (n <= 0 || height <= 0 || width <= 0) {
    h->err = "batch and spatial sizes must be positive";
    return UNET_ERR_INVALID_ARG;
  }
  if (height % m || width % m) {
    h->err = "H and W must be multiples of " + std::to_string(m);
    return UNET_ERR_SHAPE;
  }
  return UNET_OK;
}

}  // namespace

extern "C" {

int unet_set_winograd(int on) {
  const int prev = wino_enabled() ? 1 : 0;
  g_winoMode = on ? 1 : 0;
  return prev;
}

const char* unet_version(void) { return "unet_hip 0.1 (gfx950, fp32 MFMA implicit GEMM)"; }

int unet_create(const unet_config* cfg, unet_handle_t* out) {
  if (!cfg || !out) return UNET_ERR_INVALID_ARG;
  if (cfg->depth < 1 || cfg->depth > UNET_MAX_DEPTH || cfg->out_channels != 1 || cfg->in_channels < 1 ||
      cfg->in_channels > 4)
    return UNET_ERR_INVALID_ARG;
  for (int i = 0; i < cfg->depth; ++i)
    if (cfg->features[i] <= 0 || cfg->features[i] % 4) return UNET_ERR_INVALID_ARG;
  for (int i = 0; i + 1 < cfg->depth; ++i)  // ConvTranspose2d(2f -> f) feeds on the level below: widths must double
    if (cfg->features[i + 1] != 2 * cfg->features[i]) return UNET_ERR_INVALID_ARG;
  if (cfg->device < 0) return UNET_ERR_INVALID_ARG;  // the device itself is first touched by unet_finalize
  auto* h = new unet_ctx();
  h->cfg = *cfg;
  h->spec = build_spec(*cfg);
  *out = h;
  return UNET_OK;
}

int unet_num_params(unet_handle_t h) { return h ? (int)h->spec.size() : 0; }
const char* unet_param_name(unet_handle_t h, int i) {
  return (h && i >= 0 && i < (int)h->spec.size()) ? h->spec[i].name.c_str() : nullptr;
}
size_t unet_param_numel(unet_handle_t h, int i) {
  return (h && i >= 0 && i < (int)h->spec.size()) ? h->spec[i].numel : 0;
}

int unet_load_param(unet_handle_t h, const char* name, const float* data, size_t numel) {
  if (!h || !name || !data) return UNET_ERR_INVALID_ARG;
  for (const auto& s : h->spec) {
    if (s.name == name) {
      if (s.numel != numel) {
        h->err = std::string(name) + ": expected " + std::to_string(s.numel) + " elements, got " + std::to_string(numel);
        return UNET_ERR_SHAPE;
      }
      h->params[s.name].assign(data, data + numel);
      h->finalized = false;
      return UNET_OK;
    }
  }
  h->err = std::string("unknown parameter ") + name;
  return UNET_ERR_UNKNOWN_PARAM;
}

int unet_finalize(unet_handle_t h) {
  if (!h) return UNET_ERR_INVALID_ARG;
  for (const auto& s : h->spec)
    if (!h->params.count(s.name)) {
      h->err = "missing parameter " + s.name;
      return UNET_ERR_STATE;
    }
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  const unet_config& c = h->cfg;
  // the other tiers pack their operands lazily from h->params (and the bf16 tier borrows this tier's scale/shift
  // arrays): drop them so that they are rebuilt from the tensors being finalised now
  bf16_free(h);
  x3_free(h);
  for (auto* v : {&h->enc, &h->bott, &h->up, &h->dec})
    for (auto& op : *v) op.free_dev();
  h->enc.assign(2 * c.depth, GemmOp());
  h->dec.assign(2 * c.depth, GemmOp());
  h->up.assign(c.depth, GemmOp());
  h->bott.assign(2, GemmOp());
  int rc;
  int cin = c.in_channels;
  for (int l = 0; l < c.depth; ++l) {
    const std::string p = "encoder_blocks." + std::to_string(l);
    if ((rc = fold_bn_and_build(h, h->enc[2 * l], p, 0, 1, cin, c.features[l]))) return rc;
    if ((rc = fold_bn_and_build(h, h->enc[2 * l + 1], p, 3, 4, c.features[l], c.features[l]))) return rc;
    cin = c.features[l];
  }
  const int fl = c.features[c.depth - 1];
  if ((rc = fold_bn_and_build(h, h->bott[0], "bottleneck", 0, 1, fl, 2 * fl))) return rc;
  if ((rc = fold_bn_and_build(h, h->bott[1], "bottleneck", 3, 4, 2 * fl, 2 * fl))) return rc;
  for (int j = 0; j < c.depth; ++j) {
    const int f = c.features[c.depth - 1 - j];
    const std::string pu = "decoder_blocks." + std::to_string(2 * j);
    if ((rc = build_upconv(h->err, h->up[j], h->params[pu + ".weight"].data(), 2 * f, f,
                           h->params[pu + ".bias"].data())))
      return rc;
    const std::string pd = "decoder_blocks." + std::to_string(2 * j + 1);
    if ((rc = fold_bn_and_build(h, h->dec[2 * j], pd, 0, 1, 2 * f, f))) return rc;
    if ((rc = fold_bn_and_build(h, h->dec[2 * j + 1], pd, 3, 4, f, f))) return rc;
  }
  if (h->headW) hipFree(h->headW);
  h->headW = nullptr;
  if ((rc = upload(h->err, &h->headW, h->params["output.weight"]))) return rc;
  h->headB = h->params["output.bias"][0];
  h->finalized = true;
  return UNET_OK;
}

size_t unet_workspace_bytes(unet_handle_t h, int n, int height, int width) {
  if (!h || check_shape(h, n, height, width)) return 0;
  return plan_ws(h->cfg, n, height, width).total * sizeof(float);
}

int unet_reserve(unet_handle_t h, int n, int height, int width) {
  if (!h) return UNET_ERR_INVALID_ARG;
  int rc = check_shape(h, n, height, width);
  if (rc) return rc;
  const size_t need = plan_ws(h->cfg, n, height, width).total * sizeof(float);
  if (need <= h->wsBytes) return UNET_OK;
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  if (h->ws) {
    HIPCHK(h->err, hipDeviceSynchronize());
    hipFree(h->ws);
    h->ws = nullptr;
    h->wsBytes = 0;
  }
  if (hipMalloc((void**)&h->ws, need) != hipSuccess) {
    h->err = "workspace allocation of " + std::to_string(need) + " bytes failed";
    return UNET_ERR_NOMEM;
  }
  h->wsBytes = need;
  return UNET_OK;
}

static int forward_prologue(unet_handle_t h, const void* in, int n, int height, int width) {
  if (!h || !in) return UNET_ERR_INVALID_ARG;
  if (!h->finalized) {
    h->err = "unet_finalize has not been called";
    return UNET_ERR_STATE;
  }
  int rc = check_shape(h, n, height, width);
  if (rc) return rc;
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  return unet_reserve(h, n, height, width);
}

int unet_forward_u8(unet_handle_t h, const uint8_t* frames, int n, int height, int width, float* logits,
                    float* probs, uint8_t* mask, float thr, void* stream) {
  int rc = forward_prologue(h, frames, n, height, width);
  if (rc) return rc;
  if (h->cfg.in_channels != 3) {
    h->err = "uint8 frames need in_channels == 3";
    return UNET_ERR_INVALID_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const WsPlan p = plan_ws(h->cfg, n, height, width);
  const size_t npix = (size_t)n * height * width;
  const unet_config& c = h->cfg;
  hipLaunchKernelGGL(unet::pack_u8_nhwc4_kernel, dim3(grid_for(npix)), dim3(256), 0, s, frames,
                     reinterpret_cast<float*>(h->ws) + p.x0, npix, c.input_mean[0], c.input_mean[1], c.input_mean[2],
                     c.input_std[0], c.input_std[1], c.input_std[2]);
  HIPCHK(h->err, hipGetLastError());
  return forward_common(h, n, height, width, logits, probs, mask, thr, s, p);
}

int unet_forward_f32(unet_handle_t h, const float* image, int n, int height, int width, float* logits, float* probs,
                     uint8_t* mask, float thr, void* stream) {
  int rc = forward_prologue(h, image, n, height, width);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const WsPlan p = plan_ws(h->cfg, n, height, width);
  const size_t npix = (size_t)n * height * width;
  hipLaunchKernelGGL(unet::pack_nchw_nhwc4_kernel, dim3(grid_for(npix)), dim3(256), 0, s, image,
                     reinterpret_cast<float*>(h->ws) + p.x0, n, (size_t)height * width, h->cfg.in_channels);
  HIPCHK(h->err, hipGetLastError());
  return forward_common(h, n, height, width, logits, probs, mask, thr, s, p);
}

int unet_destroy(unet_handle_t h) {
  if (!h) return UNET_OK;
  hipSetDevice(h->cfg.device);
  hipDeviceSynchronize();
  train_free(h);
  bf16_free(h);
  x3_free(h);
  h->free_all();
  delete h;
  return UNET_OK;
}

int unet_profile_enable(unet_handle_t h, int on) {
  if (!h) return UNET_ERR_INVALID_ARG;
  h->prof.clear();
  h->prof.on = on != 0;
  return UNET_OK;
}

int unet_profile_count(unet_handle_t h) {
  if (!h) return 0;
  h->prof.resolve();
  return (int)h->prof.recs.size();
}

int unet_profile_get(unet_handle_t h, int i, char* name, size_t nameCap, double* ms, double* flops, double* bytes) {
  if (!h || i < 0 || i >= (int)h->prof.recs.size()) return UNET_ERR_INVALID_ARG;
  h->prof.resolve();
  const ProfRecord& r = h->prof.recs[i];
  if (name && nameCap) {
    std::strncpy(name, r.name.c_str(), nameCap - 1);
    name[nameCap - 1] = 0;
  }
  if (ms) *ms = r.ms;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return UNET_OK;
}

const char* unet_last_error(unet_handle_t h) { return h ? h->err.c_str() : "null handle"; }

int unet_device_error(unet_handle_t h) {
  if (!h) return UNET_ERR_INVALID_ARG;
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  HIPCHK(h->err, hipDeviceSynchronize());
  return h->take_device_status();
}

int unet_device_error_on(unet_handle_t h, void* stream) {
  if (!h) return UNET_ERR_INVALID_ARG;
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  HIPCHK(h->err, hipStreamSynchronize((hipStream_t)stream));
  return h->take_device_status();
}

namespace {
__global__ void status_word_kernel(const unsigned* __restrict__ err, float* __restrict__ dst) {
  const volatile unsigned* e = err;
  *dst = (e[0] | e[1]) ? 1.f : 0.f;
}
}  // namespace

int unet_device_status_to(unet_handle_t h, float* dst, void* stream) {
  if (!h || !dst) return UNET_ERR_INVALID_ARG;
  HIPCHK(h->err, hipSetDevice(h->cfg.device));
  h->ensure_err_word();
  if (!h->errDev) return UNET_ERR_NOMEM;
  hipLaunchKernelGGL(status_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)h->errDev, dst);
  HIPCHK(h->err, hipGetLastError());
  return UNET_OK;
}

int unet_debug_set_error_block(unet_handle_t h, int word, unsigned value) {
  if (!h || word < 0 || word > 1) return UNET_ERR_INVALID_ARG;
  h->ensure_err_word();
  if (!h->errHost) return UNET_ERR_NOMEM;
  reinterpret_cast<volatile unsigned*>(h->errHost)[word] = value;
  return UNET_OK;
}

// ---- single operators (test entry points) -------------------------------------------------

static thread_local std::string g_opErr;

int unet_op_conv3x3(int device, const float* x, int n, int h, int w, int cin, const float* wHost,
                    const float* scale, const float* shift, int cout, int relu, float* y, void* stream) {
  if (!x || !wHost || !scale || !shift || !y || cin % 4 || cout % 4) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  GemmOp op;
  int rc = build_conv3x3(g_opErr, op, wHost, cout, cin, scale, shift, relu);
  if (!rc) {
    hipError_t e = run_gemm_op(op, x, n, h, w, y, cout, 0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) rc = UNET_ERR_HIP;
  }
  op.free_dev();
  return rc;
}

int unet_op_upconv2x2(int device, const float* x, int n, int h, int w, int cin, const float* wHost,
                      const float* bias, int cout, float* y, void* stream) {
  if (!x || !wHost || !bias || !y || cin % 4 || cout % 4) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  GemmOp op;
  int rc = build_upconv(g_opErr, op, wHost, cin, cout, bias);
  if (!rc) {
    hipError_t e = run_gemm_op(op, x, n, h, w, y, cout, 0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) rc = UNET_ERR_HIP;
  }
  op.free_dev();
  return rc;
}

int unet_op_conv1x1(int device, const float* x, int n, int h, int w, int cin, const float* wHost, int cout, float* y,
                    void* stream) {
  if (!x || !wHost || !y || cin % 4 || cout % 4) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  GemmOp op;
  op.taps = 1;
  op.plain = 1;
  op.cinReal = cin;
  op.ck = (cin % 16 == 0) ? 16 : 4;
  op.cin = round_up(cin, op.ck);
  op.cout = cout;
  op.coutPad = round_up(cout, 16);
  op.nTotal = round_up(cout, cout >= 128 ? 128 : 64);
  auto packed = pack_fragments(op.nTotal, op.cin, op.ck, 1, [&](int nn, int ci, int) -> float {
    return (nn < cout && ci < cin) ? wHost[(size_t)nn * cin + ci] : 0.f;
  });
  std::vector<float> sc(op.nTotal, 1.f), sh(op.nTotal, 0.f);
  int rc = upload(g_opErr, &op.wt, packed);
  if (!rc) rc = upload(g_opErr, &op.scale, sc);
  if (!rc) rc = upload(g_opErr, &op.shift, sh);
  if (!rc) {
    hipError_t e = run_gemm_op(op, x, n, h, w, y, cout, 0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) rc = UNET_ERR_HIP;
  }
  op.free_dev();
  return rc;
}

int unet_op_maxpool2x2(int device, const float* x, int n, int h, int w, int c, float* y, void* stream) {
  if (!x || !y || c % 4 || h % 2 || w % 2) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  HIPCHK(g_opErr, run_maxpool(x, y, n, h, w, c, c, (hipStream_t)stream));
  HIPCHK(g_opErr, hipStreamSynchronize((hipStream_t)stream));
  return UNET_OK;
}

int unet_op_head1x1(int device, const float* x, int n, int h, int w, int c, const float* wHost, float bias,
                    float* logits, void* stream) {
  if (!x || !wHost || !logits || c % 4) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  float* wd = nullptr;
  HIPCHK(g_opErr, hipMalloc((void**)&wd, c * sizeof(float)));
  HIPCHK(g_opErr, hipMemcpy(wd, wHost, c * sizeof(float), hipMemcpyHostToDevice));
  hipError_t e = run_head(x, wd, bias, (size_t)n * h * w, c, logits, nullptr, nullptr, 0.f, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  hipFree(wd);
  return e == hipSuccess ? UNET_OK : UNET_ERR_HIP;
}

}  // extern "C"

#include "unet_bf16.inc"
#include "unet_x3.inc"
#include "unet_train.inc"
#include "unet_i8.inc"

// ---- camera stage (camera_stage.h) ---------------------------------------------------------------------------
extern "C" {

int unet_ipm_prestage_u8(int device, const uint8_t* img, int height, int width, int step, int bgrIn,
                         const double minv[9], int warpW, int warpH, int outW, int outH, uint8_t* outRgb,
                         void* stream) {
  if (!img || !minv || !outRgb || height <= 0 || width <= 0 || step < 3 * width || warpW <= 0 || warpH <= 0 ||
      outW <= 0 || outH <= 0)
    return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  unet::CameraArgs a;
  a.img = img;
  a.out = outRgb;
  for (int i = 0; i < 9; ++i) a.minv[i] = minv[i];
  a.scale_x = 1.0 / ((double)outW / (double)warpW);
  a.scale_y = 1.0 / ((double)outH / (double)warpH);
  a.height = height;
  a.width = width;
  a.step = step;
  a.swap_rb = bgrIn ? 1 : 0;
  a.warp_w = warpW;
  a.warp_h = warpH;
  a.out_w = outW;
  a.out_h = outH;
  hipLaunchKernelGGL(unet::ipm_prestage_kernel, dim3((unsigned)(((size_t)outW * outH + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  HIPCHK(g_opErr, hipGetLastError());
  return UNET_OK;
}

int unet_resize_u8(int device, const uint8_t* src, int height, int width, int channels, int outW, int outH,
                   uint8_t* dst, void* stream) {
  if (!src || !dst || height <= 0 || width <= 0 || channels <= 0 || outW <= 0 || outH <= 0) return UNET_ERR_INVALID_ARG;
  HIPCHK(g_opErr, hipSetDevice(device));
  const size_t total = (size_t)outW * outH * channels;
  hipLaunchKernelGGL(unet::resize_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, height,
                     width, channels, 1.0 / ((double)outW / (double)width), 1.0 / ((double)outH / (double)height), outW,
                     outH, dst);
  HIPCHK(g_opErr, hipGetLastError());
  return UNET_OK;
}

}  // extern "C"
