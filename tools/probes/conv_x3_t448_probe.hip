// Probe (GPU box): the third split-operand convolution structure (csrc/conv_x3_t448.h) against the first
// (csrc/conv_x3_ws.h) and, where Cout % 128 == 0, the second's two-wave form (csrc/conv_x3_r512.h) on one layer shape:
// bitwise comparison of the output planes (EPI 0), of the pooled planes (EPI 1) and of the fused head's logits (EPI 2),
// a float64 host check of sampled outputs, and interleaved timing in one process on the same operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I unet_lane_detection_amd/csrc -o tools/probes/conv_x3_t448_probe \
//         tools/probes/conv_x3_t448_probe.hip
//   conv_x3_t448_probe N H W Cin Cout [rounds] [epi]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "conv_x3_t448.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static uint16_t f16_bits(_Float16 v) {
  uint16_t u;
  memcpy(&u, &v, 2);
  return u;
}
static float f16_val(uint16_t u) {
  _Float16 v;
  memcpy(&v, &u, 2);
  return (float)v;
}
static void host_split(float v, uint16_t& hi, uint16_t& lo) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  hi = f16_bits(h);
  lo = f16_bits(l);
}

// as pack_conv_x3 in csrc/unet_x3.inc
static std::vector<uint16_t> pack(const std::vector<float>& w, int cout, int cin, const std::vector<float>& pre) {
  const int nCt = cout / 64, nCh = cin / 32;
  std::vector<uint16_t> out((size_t)nCt * nCh * 3 * 2 * 3 * 4 * 64 * 8, 0);
  for (int ct = 0; ct < nCt; ++ct)
    for (int kc = 0; kc < nCh; ++kc)
      for (int r = 0; r < 3; ++r)
        for (int kx = 0; kx < 3; ++kx)
          for (int cs = 0; cs < 4; ++cs) {
            const size_t base = (((size_t)ct * nCh + kc) * 3 + r) * (2 * 3 * 4 * 64 * 8);
            uint16_t* dh = out.data() + base + ((size_t)(0 * 3 + kx) * 4 + cs) * 64 * 8;
            uint16_t* dl = out.data() + base + ((size_t)(1 * 3 + kx) * 4 + cs) * 64 * 8;
            for (int lane = 0; lane < 64; ++lane) {
              const int j = lane & 15, lq = lane >> 4;
              const int co = 64 * ct + 16 * (j >> 2) + 4 * cs + (j & 3);
              for (int e = 0; e < 8; ++e) {
                const int ci = kc * 32 + lq * 8 + e;
                host_split(w[((size_t)co * cin + ci) * 9 + r * 3 + kx] * pre[co], dh[lane * 8 + e], dl[lane * 8 + e]);
              }
            }
          }
  return out;
}

template <class K>
static void set_lds(K kern, int bytes) {
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2, H = argc > 2 ? atoi(argv[2]) : 112, W = argc > 3 ? atoi(argv[3]) : 112;
  const int Cin = argc > 4 ? atoi(argv[4]) : 64, Cout = argc > 5 ? atoi(argv[5]) : 64;
  const int rounds = argc > 6 ? atoi(argv[6]) : 0;
  const int epi = argc > 7 ? atoi(argv[7]) : 0;   // 0 planes, 1 planes + pool, 2 head
  const int twx = W % 28 == 0 ? 28 : 32;
  if (Cin % 64 || Cout % 64 || W % twx || (epi == 2 && Cout != 64) || (epi == 1 && (H % 2 || W % 2))) {
    printf("unsupported shape\n");
    return 1;
  }
  // waves along the channels: 1 (64 per block), 2 (128), 4 (256: 8-row tiles); argv[8] forces one
  // (16 x 32 tiles: 16 fragments per wave do not fit the registers, so widths that are no multiple of 28 run the one-wave form)
  int wco = (Cout % 256 == 0 && twx == 28) ? 4 : (Cout % 128 == 0 && twx == 28) ? 2 : 1;
  if (argc > 8) wco = atoi(argv[8]);
  if ((wco != 1 && wco != 2 && wco != 4) || Cout % (64 * wco) || (wco != 1 && twx != 28) || (wco == 4 && epi == 2)) {
    printf("unsupported wave layout\n");
    return 1;
  }
  const int thT = wco == 4 ? 8 : 16;
  const bool tflat = wco == 4 && N > 1 && H % thT != 0;   // the batch as one tall image (8-row tiles only)
  const size_t px = (size_t)N * H * W, ein = px * Cin, eout = px * Cout, epool = eout / 4;
  // the host makes min(N, 4) distinct images; the device input repeats them (a batch-256 input is 1.6 G values)
  const int nu = std::min(N, 4);
  const size_t pxU = (size_t)nu * H * W, einU = pxU * Cin;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> x(einU), w((size_t)Cout * Cin * 9), sc(Cout), sh(Cout), hw(64);
  for (auto& v : x) v = std::max(nd(rng), 0.f) + (rng() % 7 == 0 ? 0.f : 0.01f * nd(rng));
  const float ws = std::sqrt(2.f / (9.f * Cin));
  for (auto& v : w) v = ws * nd(rng);
  for (int c = 0; c < Cout; ++c) {
    sc[c] = 1.f + 0.2f * nd(rng);
    sh[c] = 0.1f * nd(rng);
  }
  for (auto& v : hw) v = 0.2f * nd(rng);
  std::vector<float> pre(Cout);
  for (int c = 0; c < Cout; ++c) {
    float m = 0.f;
    for (size_t i = 0; i < (size_t)Cin * 9; ++i) m = std::max(m, std::fabs(w[(size_t)c * Cin * 9 + i]));
    int e;
    std::frexp(m, &e);
    pre[c] = std::ldexp(1.f, 10 - e);
  }
  std::vector<float> scp(Cout);
  for (int c = 0; c < Cout; ++c) scp[c] = sc[c] / pre[c];
  std::vector<uint16_t> xp(2 * einU);
  for (size_t i = 0; i < einU; ++i) host_split(x[i], xp[i], xp[einU + i]);
  const std::vector<uint16_t> wp = pack(w, Cout, Cin, pre);

  uint16_t *dIn, *dW, *dZero, *dOut[3], *dPool[3];
  float *dSc, *dSh, *dHw, *dLog[3];
  unsigned* dErr;
  CK(hipMalloc(&dIn, 2 * ein * 2));
  CK(hipMalloc(&dW, wp.size() * 2));
  CK(hipMalloc(&dZero, 4096));
  for (int i = 0; i < 3; ++i) {
    CK(hipMalloc(&dOut[i], 2 * eout * 2));
    CK(hipMalloc(&dPool[i], 2 * epool * 2 + 64));
    CK(hipMalloc(&dLog[i], px * 4));
    CK(hipMemset(dOut[i], 0xEE - i, 2 * eout * 2));
    CK(hipMemset(dPool[i], 0xEE - i, 2 * epool * 2));
    CK(hipMemset(dLog[i], 0xEE - i, px * 4));
  }
  CK(hipMalloc(&dSc, Cout * 4));
  CK(hipMalloc(&dSh, Cout * 4));
  CK(hipMalloc(&dHw, 64 * 4));
  CK(hipMalloc(&dErr, 64));
  CK(hipMemset(dErr, 0, 64));
  for (int plane = 0; plane < 2; ++plane)
    for (size_t n0 = 0; n0 < (size_t)N; n0 += nu) {
      const size_t cnt = std::min<size_t>(nu, N - n0) * (size_t)H * W * Cin;
      CK(hipMemcpy(dIn + plane * ein + n0 * (size_t)H * W * Cin, xp.data() + plane * einU, cnt * 2, hipMemcpyHostToDevice));
    }
  CK(hipMemcpy(dW, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dZero, 0, 4096));
  CK(hipMemcpy(dSc, scp.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dSh, sh.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dHw, hw.data(), 64 * 4, hipMemcpyHostToDevice));

  unet::ConvX3Args a;
  memset(&a, 0, sizeof(a));
  a.in = dIn;
  a.inLo = ein;
  a.wt = dW;
  a.zeros = dZero;
  a.scale = dSc;
  a.shift = dSh;
  a.outLo = eout;
  a.N = N;
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.ldo = Cout;
  a.co_off = 0;
  a.nChunks = Cin / 32;
  a.chunksTotal = Cin / 32;
  a.relu = 1;
  a.kSplit = 1;
  a.imgH = H;
  a.err = dErr;
  a.poolLo = epool;
  if (epi == 2) {
    a.headW = dHw;
    a.headB = 0.37f;
    a.headThr = 0.f;
  }

  // ---- first structure: TW 32 ----
  unet::ConvX3Args o = a;
  o.out = dOut[0];
  o.pool = epi == 1 ? dPool[0] : nullptr;
  o.logits = epi == 2 ? dLog[0] : nullptr;
  o.tilesX = (W + 31) / 32;
  o.tilesY = (H + 7) / 8;
  o.pixTiles = N * o.tilesY * o.tilesX;
  o.coTiles = Cout / 64;
  o.coGroup = 1;
  for (int g : {8, 4, 2})
    if (o.coTiles % g == 0) {
      o.coGroup = g;
      break;
    }
  const long oWork = (long)o.pixTiles * o.coTiles;
  const int oGrid = (int)std::max<long>(8, std::min<long>(256, oWork / 8 * 8));
  auto launch_old = [&]() {
#define LAUNCH_OLD(E)                                                                                  \
  {                                                                                                    \
    auto k = unet::conv3x3_x3_ws_kernel<32, E, false>;                                                 \
    set_lds(k, unet::X3Shape<32>::LDS_BYTES);                                                          \
    hipLaunchKernelGGL(k, dim3(oGrid), dim3(512), (size_t)unet::X3Shape<32>::LDS_BYTES, 0, o);         \
  }
    if (epi == 0) LAUNCH_OLD(0) else if (epi == 1) LAUNCH_OLD(1) else LAUNCH_OLD(2)
  };

  // ---- second structure (EPI 0 only; widths 28k): its 256-channel form where Cout allows, else the two-wave form ----
  const bool haveR = epi == 0 && Cout % 128 == 0 && W % 28 == 0;
  const int rwpx = Cout % 256 == 0 ? 1 : 2;
  const bool rflat = N > 1 && H % 8 != 0;
  unet::ConvX3Args r = a;
  r.out = dOut[1];
  r.tilesX = W / 28;
  r.tilesY = (H + 7) / 8;
  r.pixTiles = N * r.tilesY * r.tilesX;
  if (rflat) {
    r.N = 1;
    r.H = N * H;
    r.tilesY = (N * H + 7) / 8;
    r.pixTiles = r.tilesY * r.tilesX;
  }
  r.coTiles = Cout / (256 / rwpx);
  r.coGroup = r.coTiles;
  const long rWork = (long)r.pixTiles * r.coTiles;
  const int rGrid = (int)std::max<long>(8, std::min<long>(256, rWork / 8 * 8));
  auto launch_r512 = [&]() {
#define LAUNCH_R(WPX, FL)                                                                          \
  {                                                                                                \
    auto k = unet::conv3x3_x3_r512_kernel<28, WPX, 0, FL>;                                         \
    set_lds(k, unet::X3RShape<28>::LDS_BYTES);                                                     \
    hipLaunchKernelGGL(k, dim3(rGrid), dim3(256), (size_t)unet::X3RShape<28>::LDS_BYTES, 0, r);    \
  }
    if (rwpx == 1) {
      if (rflat) LAUNCH_R(1, true) else LAUNCH_R(1, false)
    } else {
      if (rflat) LAUNCH_R(2, true) else LAUNCH_R(2, false)
    }
  };

  // ---- third structure ----
  unet::ConvX3Args b = a;
  b.out = dOut[2];
  b.pool = epi == 1 ? dPool[2] : nullptr;
  b.logits = epi == 2 ? dLog[2] : nullptr;
  b.tilesX = W / twx;
  b.tilesY = (H + thT - 1) / thT;
  b.pixTiles = N * b.tilesY * b.tilesX;
  if (tflat) {
    b.N = 1;
    b.H = N * H;
    b.tilesY = (N * H + thT - 1) / thT;
    b.pixTiles = b.tilesY * b.tilesX;
  }
  b.coTiles = Cout / (64 * wco);
  b.coGroup = b.coTiles;
  const long nWork = (long)b.pixTiles * b.coTiles;
  const int nGrid = (int)std::max<long>(8, std::min<long>(256, nWork / 8 * 8));
#if UNET_R512_STAMPS
  unsigned long long* dStamps;
  CK(hipMalloc(&dStamps, 256 * 8 * 8));
  CK(hipMemset(dStamps, 0, 256 * 8 * 8));
  if (epi != 2) b.logits = reinterpret_cast<float*>(dStamps);
  unsigned long long* dStampsR;   // the second structure's kernel is a stamps build too
  CK(hipMalloc(&dStampsR, 256 * 8 * 8));
  r.logits = reinterpret_cast<float*>(dStampsR);
  if (epi != 2) o.logits = nullptr;
#endif
  auto launch_new = [&]() {
#define LAUNCH_NEW(TWX, WCO, E, FL)                                                                              \
  {                                                                                                              \
    using SH = unet::X3TShape<TWX, unet::x3t_row_blocks(WCO)>;                                                   \
    constexpr int ldsB = FL ? SH::LDS_BYTES_FLAT : SH::LDS_BYTES_PLAIN;                                          \
    auto k = unet::conv3x3_x3_t448_kernel<TWX, WCO, E, FL>;                                                      \
    set_lds(k, ldsB);                                                                                            \
    hipLaunchKernelGGL(k, dim3(nGrid), dim3(256), (size_t)ldsB, 0, b);                                           \
  }
    if (twx == 28) {
      if (wco == 1) {
        if (epi == 0) LAUNCH_NEW(28, 1, 0, false) else if (epi == 1) LAUNCH_NEW(28, 1, 1, false) else LAUNCH_NEW(28, 1, 2, false)
      } else if (wco == 2) {
        if (epi == 0) LAUNCH_NEW(28, 2, 0, false) else LAUNCH_NEW(28, 2, 1, false)
      } else if (tflat) {
        if (epi == 0) LAUNCH_NEW(28, 4, 0, true) else LAUNCH_NEW(28, 4, 1, true)
      } else {
        if (epi == 0) LAUNCH_NEW(28, 4, 0, false) else LAUNCH_NEW(28, 4, 1, false)
      }
    } else {
      if (epi == 0) LAUNCH_NEW(32, 1, 0, false) else if (epi == 1) LAUNCH_NEW(32, 1, 1, false) else LAUNCH_NEW(32, 1, 2, false)
    }
  };
  printf("N %d H %d W %d Cin %d Cout %d epi %d: first tw 32 grid %d (%ld items); third %dx%d%s wco %d grid %d (%ld items)\n", N,
         H, W, Cin, Cout, epi, oGrid, oWork, thT, twx, tflat ? " flat" : "", wco, nGrid, nWork);

  launch_old();
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  launch_new();
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  if (haveR) {
    launch_r512();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
  }
  bool bad = false;
  std::vector<uint16_t> ya(2 * eout), yb(2 * eout);
  if (epi != 2) {
    CK(hipMemcpy(ya.data(), dOut[0], ya.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(yb.data(), dOut[2], yb.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0, first = (size_t)-1;
    for (size_t i = 0; i < ya.size(); ++i)
      if (ya[i] != yb[i]) {
        if (first == (size_t)-1) first = i;
        ++diff;
      }
    printf("third vs first structure: %zu of %zu halfs differ", diff, ya.size());
    if (diff) {
      const size_t e = first % eout, p = e / Cout;
      printf(" (first: plane %zu n %zu y %zu x %zu c %zu: %04x vs %04x)", first / eout, p / ((size_t)H * W), (p / W) % H,
             p % W, e % Cout, ya[first], yb[first]);
      bad = true;
    }
    printf("\n");
    // float64 check of sampled outputs of the new kernel (inputs as the planes hold them)
    double worst = 0, worstRef = 0;
    const int samples = 4000;
    for (int sidx = 0; sidx < samples; ++sidx) {
      const size_t p = ((size_t)sidx * 2654435761u) % pxU;   // within the distinct images (image n repeats image n % nu)
      const int co = (int)(((size_t)sidx * 40503u) % Cout);
      const int n = (int)(p / ((size_t)H * W)), y = (int)((p / W) % H), xx = (int)(p % W);
      double s = 0;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y + ky - 1, xc = xx + kx - 1;
          if (yy < 0 || yy >= H || xc < 0 || xc >= W) continue;
          const size_t ib = (((size_t)n * H + yy) * W + xc) * Cin;
          for (int ci = 0; ci < Cin; ++ci) {
            const double xv = (double)f16_val(xp[ib + ci]) + (double)f16_val(xp[einU + ib + ci]);
            s += xv * (double)w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
          }
        }
      const double ref = std::max(0.0, s * sc[co] + sh[co]);
      const double got = (double)f16_val(yb[p * Cout + co]) + (double)f16_val(yb[eout + p * Cout + co]);
      worst = std::max(worst, std::fabs(got - ref));
      worstRef = std::max(worstRef, std::fabs(ref));
    }
    printf("third structure vs float64 on %d samples: max |err| %.3e (max |ref| %.3f)\n", samples, worst, worstRef);
    if (!(worst < 1e-3 * std::max(1.0, worstRef))) bad = true;
  }
  if (epi == 1) {
    std::vector<uint16_t> pa(2 * epool), pb(2 * epool);
    CK(hipMemcpy(pa.data(), dPool[0], pa.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pb.data(), dPool[2], pb.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < pa.size(); ++i) diff += pa[i] != pb[i];
    printf("pooled planes, third vs first structure: %zu of %zu halfs differ\n", diff, pa.size());
    if (diff) bad = true;
  }
  if (epi == 2) {
    std::vector<float> la(px), lb2(px);
    CK(hipMemcpy(la.data(), dLog[0], px * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(lb2.data(), dLog[2], px * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < px; ++i) diff += memcmp(&la[i], &lb2[i], 4) != 0;
    printf("fused head logits, third vs first structure: %zu of %zu differ (logit[0] %.6f vs %.6f)\n", diff, px, la[0], lb2[0]);
    if (diff) bad = true;
  }
  if (haveR) {
    std::vector<uint16_t> yr(2 * eout);
    CK(hipMemcpy(yr.data(), dOut[1], yr.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < ya.size(); ++i) diff += ya[i] != yr[i];
    printf("second (two-wave) vs first structure: %zu halfs differ\n", diff);
  }

  if (rounds > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop = 2.0 * px * 9.0 * Cin * Cout;
    for (int it = 0; it < 20; ++it) {   // ~settle clocks
      launch_old();
      launch_new();
      if (haveR) launch_r512();
    }
    CK(hipDeviceSynchronize());
    std::vector<float> tOld, tNew, tR;
    auto timeit = [&](auto&& fn, std::vector<float>& dst) {
      float ms;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) fn();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      dst.push_back(ms / 5);
    };
    for (int rr = 0; rr < rounds; ++rr) {
      timeit(launch_old, tOld);
      timeit(launch_new, tNew);
      if (haveR) timeit(launch_r512, tR);
    }
    std::sort(tOld.begin(), tOld.end());
    std::sort(tNew.begin(), tNew.end());
    const float mo = tOld[tOld.size() / 2], mn = tNew[tNew.size() / 2];
#if UNET_R512_STAMPS
    if (epi != 2) {
      std::vector<unsigned long long> st(256 * 8);
      CK(hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> loop, bar, epiC, tot, clk;
      for (int i = 0; i < nGrid; ++i) {
        const unsigned long long* q = &st[(size_t)i * 8];
        const double chunks = (double)q[5], items = chunks / (Cin / 32);
        loop.push_back(q[0] / chunks);
        bar.push_back(q[1] / chunks);
        epiC.push_back(q[2] / items);
        tot.push_back((double)q[3]);
        clk.push_back((double)q[3] / (double)q[4] * 0.1);
      }
      auto med = [](std::vector<double> v) {
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      const int nf = (twx / 4) * (wco == 1 ? 1 : 2);   // fragments per wave
      printf("stamps (wave 0, median over blocks): chunk loop %.0f cycles (ideal %d = %d MFMAs x 16), barrier %.0f per chunk, "
             "epilogue %.0f per item, kernel %.0f cycles, clock %.3f GHz\n",
             med(loop), 9 * nf * 12 * 16, 9 * nf * 12, med(bar), med(epiC), med(tot), med(clk));
    }
#endif
    printf("first : median %.4f ms (min %.4f) = %.1f TFLOP/s algorithmic\n", mo, tOld[0], flop / mo * 1e-9);
    if (haveR) {
      std::sort(tR.begin(), tR.end());
      const float mr = tR[tR.size() / 2];
      printf("second: median %.4f ms (min %.4f) = %.1f TFLOP/s algorithmic (%s form)\n", mr, tR[0], flop / mr * 1e-9,
             rwpx == 1 ? "256-channel" : "two-wave");
    }
    printf("third : median %.4f ms (min %.4f) = %.1f TFLOP/s algorithmic, %.1f executed; ratio to first %.3f\n", mn, tNew[0],
           flop / mn * 1e-9, 3 * flop / mn * 1e-9, mo / mn);
  }
  return bad ? 3 : 0;
}
