// Probe (GPU box): sustained fp16 MFMA rate of the whole chip from registers only (no LDS, no memory), per MFMA
// shape and per operand content - what the power management lets dense MFMA code run at.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_power tools/probes/mfma_power.hip
// Variants: shape 16x16x32 vs 32x32x16; operands random normal / zero; "run length" = how many consecutive MFMAs
// accumulate into the same registers before the next accumulator's turn (a run keeps SrcC inside the MFMA pipe).  Each variant runs ~1.5 s back to back so the clock has settled; the last 0.5 s is timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(256) void burn(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(t * 8 + i) & 0xFFFF];
    b[i] = src[(t * 8 + 4 + i) & 0xFFFF];
  }
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 72; ++u) {   // CHAINS = length of a run of consecutive MFMAs on one accumulator
        const int c = CHAINS == 0 ? 0 : (u / CHAINS) & 7;
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 3], acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += acc[c][0] + acc[c][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 36; ++u) {   // 36 x (32x32x16) = the flops of 72 x (16x16x32)
        const int c = CHAINS == 0 ? 0 : (u / CHAINS) & 3;
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 1) & 3], acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += acc[c][0] + acc[c][15];
  }
  if (sum == 123.456f) out[t] = sum;   // keeps the loop alive, never true in practice
}

template <int SHAPE, int CHAINS>
void run(const char* name, const f16x8* src, float* out, int blocks) {
  const int iters = 1000;
  const double flopsPerLaunch = (double)blocks * 4 * iters * 72 * (2.0 * 16 * 16 * 32);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // settle ~1 s, then time ~0.5 s
  float ms = 0.f;
  int launches = 0;
  hipEventRecord(e0, 0);
  do {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((burn<SHAPE, CHAINS>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    launches += 10;
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 1000.f);
  hipEventRecord(e0, 0);
  int timed = 0;
  do {
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((burn<SHAPE, CHAINS>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    timed += 10;
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  } while (ms < 500.f);
  printf("%-44s %8.1f TFLOP/s  (%d launches, %.3f ms each)\n", name, flopsPerLaunch * timed / (ms * 1e-3) / 1e12, timed,
         ms / timed);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 2;   // 8 waves per CU, 2 per SIMD
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
    v = (_Float16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
  }
  f16x8 *rnd, *zer;
  float* out;
  hipMalloc(&rnd, h.size() * 2);
  hipMalloc(&zer, h.size() * 2);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(zer, 0, h.size() * 2);
  printf("CUs %d, blocks %d x 256 threads\n", prop.multiProcessorCount, blocks);
  run<16, 1>("16x16x32 random, run length 1 (8 accs)", rnd, out, blocks);
  run<16, 2>("16x16x32 random, run length 2", rnd, out, blocks);
  run<16, 3>("16x16x32 random, run length 3", rnd, out, blocks);
  run<16, 6>("16x16x32 random, run length 6", rnd, out, blocks);
  run<16, 9>("16x16x32 random, run length 9", rnd, out, blocks);
  run<16, 18>("16x16x32 random, run length 18", rnd, out, blocks);
  run<16, 0>("16x16x32 random, one accumulator", rnd, out, blocks);
  run<32, 1>("32x32x16 random, run length 1 (4 accs)", rnd, out, blocks);
  run<32, 3>("32x32x16 random, run length 3", rnd, out, blocks);
  run<32, 9>("32x32x16 random, run length 9", rnd, out, blocks);
  run<32, 0>("32x32x16 random, one accumulator", rnd, out, blocks);
  run<16, 1>("16x16x32 zeros, run length 1", zer, out, blocks);
  run<16, 0>("16x16x32 zeros, one accumulator", zer, out, blocks);
  run<32, 1>("32x32x16 zeros, run length 1", zer, out, blocks);
  run<16, 3>("16x16x32 random, run length 3 (again)", rnd, out, blocks);
  return 0;
}
