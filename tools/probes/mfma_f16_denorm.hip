// Probe (GPU box): does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs, and how are products accumulated?
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_f16_denorm tools/probes/mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* a, const _Float16* b, float* out) {
  const int lane = threadIdx.x;
  f16x8 av, bv;
  for (int j = 0; j < 8; ++j) {
    av[j] = a[(lane & 15) * 32 + (lane >> 4) * 8 + j];   // A[row = lane&15][k]
    bv[j] = b[((lane >> 4) * 8 + j) * 16 + (lane & 15)]; // B[k][col = lane&15]
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];   // D[row][col]
}

int main() {
  _Float16 ha[16 * 32], hb[32 * 16];
  // row 0 of A: subnormal 2^-20 in k=0 only; row 1: 2^-24 (smallest subnormal) in every k; row 2: 1.0 + tiny
  for (int i = 0; i < 16 * 32; ++i) ha[i] = (_Float16)0.f;
  for (int i = 0; i < 32 * 16; ++i) hb[i] = (_Float16)0.f;
  ha[0 * 32 + 0] = (_Float16)ldexpf(1.f, -20);
  for (int k = 0; k < 32; ++k) ha[1 * 32 + k] = (_Float16)ldexpf(1.f, -24);
  // row 3: exact-sum test: 2048 in k=0, 1 in k=1..31 with B = 1  -> 2079 exactly if accumulated in >= fp32
  ha[3 * 32 + 0] = (_Float16)2048.f;
  for (int k = 1; k < 32; ++k) ha[3 * 32 + k] = (_Float16)1.f;
  // row 4: cancellation: 60000 * 1 + (-60000) * 1 + 2^-14 * 1  (fp32-exact accumulation gives 2^-14)
  ha[4 * 32 + 0] = (_Float16)60000.f;
  ha[4 * 32 + 1] = (_Float16)-60000.f;
  ha[4 * 32 + 2] = (_Float16)ldexpf(1.f, -14);
  for (int k = 0; k < 32; ++k) hb[k * 16 + 0] = (_Float16)1.f;        // column 0: ones
  for (int k = 0; k < 32; ++k) hb[k * 16 + 1] = (_Float16)1024.f;     // column 1: 2^10
  for (int k = 0; k < 32; ++k) hb[k * 16 + 2] = (_Float16)ldexpf(1.f, -20);   // column 2: subnormal B
  _Float16 *da, *db;
  float* dout;
  hipMalloc(&da, sizeof(ha));
  hipMalloc(&db, sizeof(hb));
  hipMalloc(&dout, 256 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
  float o[256];
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("A subnormal 2^-20 x B 1       : got %.10e  expect %.10e\n", o[0 * 16 + 0], ldexp(1.0, -20));
  printf("A subnormal 2^-20 x B 1024    : got %.10e  expect %.10e\n", o[0 * 16 + 1], ldexp(1.0, -10));
  printf("32 x (2^-24 x 1)              : got %.10e  expect %.10e\n", o[1 * 16 + 0], 32 * ldexp(1.0, -24));
  printf("A 1 (k>=1) x B subnormal 2^-20: got %.10e  expect %.10e\n", o[3 * 16 + 2], (2048 + 31) * ldexp(1.0, -20));
  printf("2048 + 31 x 1                 : got %.10e  expect 2079\n", o[3 * 16 + 0]);
  printf("60000 - 60000 + 2^-14         : got %.10e  expect %.10e\n", o[4 * 16 + 0], ldexp(1.0, -14));
  const bool keeps = o[0] == (float)ldexp(1.0, -20) && o[1 * 16 + 0] == (float)(32 * ldexp(1.0, -24)) &&
                     o[3 * 16 + 2] == (float)((2048 + 31) * ldexp(1.0, -20));
  printf("RESULT fp16 subnormal MFMA inputs %s\n", keeps ? "KEPT" : "FLUSHED (or rounded)");
  return 0;
}
