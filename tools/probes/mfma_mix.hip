// Probe (GPU box): can the two cross terms of the split-operand product (w_lo x_hi + w_hi x_lo) run on the block-scaled
// narrow-format MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) beside the fp16 main term?
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_mix tools/probes/mfma_mix.hip
// Part 1 (numerics): the operand lane map and the scale semantics of the scaled MFMA with fp6 (e2m3) and fp8 (e4m3)
//   operands, checked with exactly representable data against a host sum: lane l = (row / column l & 15, k group
//   l >> 4) holds 32 values of its row; value j sits at bits [6j, 6j + 6) (fp6) / byte j (fp8) of its registers; the
//   lane's scale byte (E8M0, 2^(b - 127)) multiplies exactly those 32 values.  The test does not need the k index of
//   slot (group, j) - only that A and B pair slot (g, j) with slot (g, j).
// Part 2 (rate): register-only loops over 8 accumulators, per "64-channel unit" of one accumulator tile:
//   f16x3 = 6 fp16 MFMAs; mx6 = 2 fp16 + 1 scaled fp6; mx8 = 2 fp16 + 1 scaled fp8; plus the pure streams.
//   Random operands; each variant runs ~1 s to settle the clock, then 0.5 s timed.  Reports units / s and the
//   fp16-equivalent "direct convolution" TFLOP/s (2 * 16 * 16 * 64 flops per unit).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- part 1
// FMT: 0 = fp8 e4m3, 2 = fp6 e2m3
template <int FMT>
__global__ void one_mfma(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, FMT, FMT, 0, sa[l], 0, sb[l]);
  c[l] = acc;
}

static float fp6_val(int code) {   // e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}
static float fp8_val(int code) {   // e4m3fn, bias 7
  const int s = code >> 7, e = (code >> 3) & 15, m = code & 7;
  const float v = e == 0 ? m * 0.125f * ldexpf(1.f, -6) : (1.f + m * 0.125f) * ldexpf(1.f, e - 7);
  return s ? -v : v;
}

template <int FMT>
static int check_layout() {
  // codes[role][row][group][j]
  std::vector<int> ca(16 * 4 * 32), cb(16 * 4 * 32), sca(64), scb(64);
  srand(7 + FMT);
  for (auto& v : ca) v = FMT == 2 ? rand() & 63 : ((rand() & 0x80) | ((rand() % 9 + 3) << 3) | (rand() & 7));
  for (auto& v : cb) v = FMT == 2 ? rand() & 63 : ((rand() & 0x80) | ((rand() % 9 + 3) << 3) | (rand() & 7));
  for (int l = 0; l < 64; ++l) {
    sca[l] = 127 + (rand() % 7) - 3;   // byte 0 = the scale; the other bytes hold rubbish on purpose
    scb[l] = 127 + (rand() % 7) - 3;
  }
  std::vector<i32x8> ha(64), hb(64);
  for (int l = 0; l < 64; ++l) {
    uint32_t wa[8] = {0}, wb[8] = {0};
    const int r = l & 15, g = l >> 4;
    for (int j = 0; j < 32; ++j) {
      const int bits = FMT == 2 ? 6 : 8;
      const uint64_t va = ca[(r * 4 + g) * 32 + j], vb = cb[(r * 4 + g) * 32 + j];
      const int pos = j * bits, wi = pos >> 5, sh = pos & 31;
      wa[wi] |= (uint32_t)(va << sh);
      wb[wi] |= (uint32_t)(vb << sh);
      if (sh + bits > 32) {
        wa[wi + 1] |= (uint32_t)(va >> (32 - sh));
        wb[wi + 1] |= (uint32_t)(vb >> (32 - sh));
      }
    }
    for (int i = 0; i < 8; ++i) {
      ha[l][i] = (int)wa[i];
      hb[l][i] = (int)wb[i];
    }
  }
  std::vector<int> hsa(64), hsb(64);
  for (int l = 0; l < 64; ++l) {
    hsa[l] = sca[l] | (0x5A << 8) | (0x33 << 16) | (0x7F << 24);
    hsb[l] = scb[l] | (0x11 << 8) | (0x99 << 16) | (0x80 << 24);
  }
  i32x8 *da, *db;
  int *dsa, *dsb;
  f32x4* dc;
  hipMalloc(&da, 64 * 32);
  hipMalloc(&db, 64 * 32);
  hipMalloc(&dsa, 256);
  hipMalloc(&dsb, 256);
  hipMalloc(&dc, 64 * 16);
  hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(one_mfma<FMT>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  std::vector<f32x4> hc(64);
  hipMemcpy(hc.data(), dc, 64 * 16, hipMemcpyDeviceToHost);
  // C[i][n] = sum over (g, j) of A[i][g][j] * 2^(sa[i,g] - 127) * B[n][g][j] * 2^(sb[n,g] - 127); C layout: lane l holds
  // column n = l & 15, rows 4 * (l >> 4) + r
  double worst = 0.0, big = 0.0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int n = l & 15, i = 4 * (l >> 4) + r;
      double s = 0.0;
      for (int g = 0; g < 4; ++g) {
        double sg = 0.0;
        for (int j = 0; j < 32; ++j) {
          const int a = ca[(i * 4 + g) * 32 + j], b = cb[(n * 4 + g) * 32 + j];
          sg += (double)(FMT == 2 ? fp6_val(a) : fp8_val(a)) * (double)(FMT == 2 ? fp6_val(b) : fp8_val(b));
        }
        s += sg * ldexp(1.0, sca[g * 16 + i] - 127) * ldexp(1.0, scb[g * 16 + n] - 127);
      }
      worst = fmax(worst, fabs(s - (double)hc[l][r]));
      big = fmax(big, fabs(s));
    }
  printf("scaled MFMA, %s operands: max |device - host| = %.3e (largest |C| %.3e) -> %s\n", FMT == 2 ? "fp6 e2m3" : "fp8 e4m3",
         worst, big, worst <= 1e-5 * big ? "lane map + scale semantics as assumed" : "MISMATCH");
  hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc);
  return worst <= 1e-5 * big ? 0 : 1;
}

// ---------------------------------------------------------------- part 2
// MODE 0: 6 fp16 MFMAs per unit (f16x3); 1: 2 fp16 + 1 fp6; 2: 2 fp16 + 1 fp8; 3: fp6 only (1 per unit); 4: fp8 only;
// 5: 2 fp16 only (the main term alone)
template <int MODE>
__global__ __launch_bounds__(256) void burn(const f16x8* __restrict__ src, float* __restrict__ out, int iters,
                                            unsigned long long* __restrict__ stamps) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f16x8 a[4], b[4];
  i32x8 qa[2], qb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(t * 8 + i) & 0xFFFF];
    b[i] = src[(t * 8 + 4 + i) & 0xFFFF];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f16x8 u0 = src[(t * 16 + 8 + 2 * i) & 0xFFFF], u1 = src[(t * 16 + 9 + 2 * i) & 0xFFFF];
    const f16x8 v0 = src[(t * 16 + 12 + 2 * i) & 0xFFFF], v1 = src[(t * 16 + 13 + 2 * i) & 0xFFFF];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 p0 = __builtin_bit_cast(i32x4, u0), p1 = __builtin_bit_cast(i32x4, u1);
    const i32x4 r0 = __builtin_bit_cast(i32x4, v0), r1 = __builtin_bit_cast(i32x4, v1);
    // random bits; for fp8 keep exponents away from NaN (e4m3fn: 0x7F / 0xFF are NaN)
    const int msk = (MODE == 2 || MODE == 4) ? 0x77777777 : -1;
    qa[i] = (i32x8){p0[0] & msk, p0[1] & msk, p0[2] & msk, p0[3] & msk, p1[0] & msk, p1[1] & msk, p1[2] & msk, p1[3] & msk};
    qb[i] = (i32x8){r0[0] & msk, r0[1] & msk, r0[2] & msk, r0[3] & msk, r1[0] & msk, r1[1] & msk, r1[2] & msk, r1[3] & msk};
  }
  if (MODE >= 6) {
    // MODE 6: random hi and lo (as MODE 0, in the real term order); 7: half of the activations zero (ReLU), lo of a zero is
    // zero; 8 / 9 / 10: as 7 with the lo operands cut to a 4- / 6- / 8-bit significand (low mantissa bits zeroed)
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u16x8 xh = __builtin_bit_cast(u16x8, b[i]), xl = __builtin_bit_cast(u16x8, b[2 + i]);
      u16x8 wl = __builtin_bit_cast(u16x8, a[2 + i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (MODE >= 7 && ((xh[e] >> 3) & 1)) {   // a pseudo-random half of the elements
          xh[e] = 0;
          xl[e] = 0;
        }
        if (MODE >= 8) {   // 8: 3 mantissa bits kept, 9: 5, 10: 7
          const unsigned short m = MODE == 8 ? 0xFF80 : (MODE == 9 ? 0xFFE0 : 0xFFF8);
          xl[e] &= m;
          wl[e] &= m;
        }
      }
      b[i] = __builtin_bit_cast(f16x8, xh);
      b[2 + i] = __builtin_bit_cast(f16x8, xl);
      a[2 + i] = __builtin_bit_cast(f16x8, wl);
    }
  }
  const int sc = 127 - 4 + (t & 3);
  f32x4 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u) {   // 24 units per iteration, accumulators round robin
      const int c = u & 7;
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + k) & 3], b[(u + k + 1) & 3], acc[c], 0, 0, 0);
      } else if (MODE >= 6) {
        // f16x3 with operands that look like the real ones: per 32 channels (w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi);
        // a[0], a[1] = w_hi, a[2], a[3] = w_lo; b[0], b[1] = x_hi, b[2], b[3] = x_lo (prepared in front of the loop)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 + ((u + k) & 1)], b[(u + k) & 1], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + k) & 1], b[2 + ((u + k) & 1)], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + k) & 1], b[(u + k) & 1], acc[c], 0, 0, 0);
        }
      } else {
        if (MODE == 1 || MODE == 3)
          acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[u & 1], qb[(u >> 1) & 1], acc[c], 2, 2, 0, sc, 0, sc);
        if (MODE == 2 || MODE == 4)
          acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[u & 1], qb[(u >> 1) & 1], acc[c], 0, 0, 0, sc, 0, sc);
        if (MODE == 1 || MODE == 2 || MODE == 5) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u + 1) & 3], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + 1) & 3], b[(u + 2) & 3], acc[c], 0, 0, 0);
        }
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) sum += acc[c][0] + acc[c][3];
  if (sum == 123.456f) out[t] = sum;   // keeps the loop alive, never true in practice
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

// whole-kernel cycles / 100 MHz ticks of one wave: the clock the chip holds under the stream
template <int MODE>
void run(const char* name, const f16x8* src, float* out, int blocks) {
  static unsigned long long* stamps = nullptr;
  if (!stamps) hipMalloc(&stamps, 4096 * 16);
  const int iters = 1000;
  const double unitsPerLaunch = (double)blocks * 4 * iters * 24;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  hipEventRecord(e0, 0);
  do {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((burn<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, iters, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 1000.f);
  hipEventRecord(e0, 0);
  int timed = 0;
  do {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((burn<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, iters, stamps);
    timed += 10;
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 500.f);
  const double ups = unitsPerLaunch * timed / (ms * 1e-3);
  // cycles per unit and SIMD at the nominal 2.4 GHz would be 2.4e9 * 1024 SIMDs / ups
  unsigned long long st[2];
  hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost);
  printf("[%.1f cycles per unit, clock %.3f GHz] ", (double)st[0] / (iters * 24.0), (double)st[0] / (double)st[1] * 0.1);
  printf("%-40s %8.2f G units/s = %7.1f TFLOP/s direct-convolution equivalent; %.1f ns per unit and SIMD\n", name, ups / 1e9,
         ups * 2.0 * 16 * 16 * 64 / 1e12, 1024.0 / ups * 1e9);
  fflush(stdout);
}

int main() {
  int bad = check_layout<2>();
  bad += check_layout<0>();
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;   // 4 waves per CU, one per SIMD (the r512 occupancy)
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
    v = (_Float16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
  }
  f16x8* rnd;
  float* out;
  hipMalloc(&rnd, h.size() * 2);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  printf("CUs %d, blocks %d x 256 threads (one wave per SIMD)\n", prop.multiProcessorCount, blocks);
  run<0>("f16x3: 6 fp16 MFMAs per unit", rnd, out, blocks);
  run<1>("mx6: 2 fp16 + 1 scaled fp6 per unit", rnd, out, blocks);
  run<2>("mx8: 2 fp16 + 1 scaled fp8 per unit", rnd, out, blocks);
  run<5>("main term only: 2 fp16 per unit", rnd, out, blocks);
  run<3>("scaled fp6 only: 1 per unit", rnd, out, blocks);
  run<4>("scaled fp8 only: 1 per unit", rnd, out, blocks);
  run<0>("f16x3 again", rnd, out, blocks);
  run<6>("f16x3, real term order, random operands", rnd, out, blocks);
  run<7>("f16x3, half of the activations zero", rnd, out, blocks);
  run<8>("f16x3, half zero + lo operands cut to 4 bits", rnd, out, blocks);
  run<9>("f16x3, half zero + lo operands cut to 6 bits", rnd, out, blocks);
  run<10>("f16x3, half zero + lo operands cut to 8 bits", rnd, out, blocks);
  run<7>("f16x3, half of the activations zero (again)", rnd, out, blocks);
  run<0>("f16x3 once more", rnd, out, blocks);
  return bad;
}
