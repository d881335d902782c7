// Probe (GPU box): what a split-operand (fp16 hi + lo, 3 MFMAs per product) GEMM sustains with the kernel structure the
// convolution does NOT have yet: no loader waves, 4 waves per CU with up to 512 registers each, a 256 x 256 block tile
// (wave tile 128 x 128 = 64 accumulators), operands staged by LDS-DMA issued from the MFMA waves themselves, 32-deep
// K stages double buffered.  Per stage a wave reads 32 fragments from LDS for 192 MFMAs (the convolution kernel: 16 per
// 48) and the block stages 64 KiB for 768 MFMAs (the convolution kernel: 40 KiB per 576).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/gemm_x3_512reg tools/probes/gemm_x3_512reg.hip
// C[M][N] fp32 = A[M][K] * B[K][N];  A: two fp16 planes, row major; B: two planes in MFMA fragment order
// [K/32][N/16][lane 64][8].  Checks a few entries against a float64 host sum and prints TFLOP/s (executed = 3 x).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int A_PLANE = BM * 64;          // 256 rows x 64 bytes = 16 KiB
constexpr int B_PLANE = (BN / 16) * 1024; // 16 fragments x 1 KiB = 16 KiB
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;   // 64 KiB
constexpr int LDS_BYTES = 2 * STAGE;

struct Args {
  const _Float16 *aHi, *aLo;   // [M][K]
  const _Float16 *bHi, *bLo;   // [K/32][N/16][64][8]
  float* c;                    // [M][N]
  int M, N, K;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_x3(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int tilesN = a.N / BN;
  // Persistent: gridDim.x blocks (a multiple of 8) walk the tiles; XCD-aware order: block b runs on XCD b & 7 and takes
  // logical slot lb, so the column tiles of one row tile are consecutive on one XCD and share its L2 for the A rows.
  const int G = gridDim.x;
  const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int tiles = (a.M / BM) * tilesN;
  const int wm = wave >> 1, wn = wave & 1;
  const int nStages = a.K / BK;
  if (lb >= tiles) return;

  // ---- staging: 64 pieces of 1 KiB per stage, 16 per wave.  A piece = 16 rows x 64 B (swizzled 16-byte parts),
  //      B piece = one fragment.  Per-piece source pointers of a tile's stage 0, advanced by a constant per stage ----
  auto tile_src = [&](int t, const _Float16* (&src)[16]) __attribute__((always_inline)) {
    const int tm = t / tilesN, tn = t - tm * tilesN;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = wave * 16 + j;   // 0..31: A (plane = q >> 4, 16-row group q & 15); 32..63: B (plane, fragment)
      const int planeA = q >> 4, grp = q & 15;
      const int row = grp * 16 + (lane >> 2);
      const int part = (lane & 3) ^ (((row >> 2) & 1) << 1);
      const _Float16* pa = (planeA ? a.aLo : a.aHi) + (size_t)(tm * BM + row) * a.K + part * 8;
      const int planeB = (q - 32) >> 4, frag = (q - 32) & 15;
      const _Float16* pb = (planeB ? a.bLo : a.bHi) + ((size_t)tn * (BN / 16) + frag) * 512 + lane * 8;
      src[j] = q < 32 ? pa : pb;
    }
  };
  const size_t stepMine = wave < 2 ? (size_t)BK : (size_t)(a.N / 16) * 512;   // waves 0,1 stage A, waves 2,3 stage B
  // issued as asm: with the builtin the compiler assumes the in-flight LDS write may alias every later ds_read of the
  // same wave and waits lgkmcnt(0) in front of every use of LDS data (csrc/lds_dma.h)
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto dma = [&](const _Float16* src, int buf, int j) __attribute__((always_inline)) {
    const unsigned dst = ldsBase + buf * STAGE + (wave * 16 + j) * 1024;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
  };

  // A fragment ms of this wave: rows wm*128 + ms*16 + li
  int aoff[8];
#pragma unroll
  for (int ms = 0; ms < 8; ++ms) {
    const int row = wm * 128 + ms * 16 + li;
    aoff[ms] = row * 64 + ((lq ^ (((row >> 2) & 1) << 1)) << 4);
  }
  const int boff = 2 * A_PLANE + (wn * 8) * 1024 + lane * 16;

  const _Float16 *srcCur[16], *srcNext[16];
  tile_src(lb, srcCur);
#pragma unroll
  for (int j = 0; j < 16; ++j) dma(srcCur[j], 0, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int par = 0;   // LDS buffer of the stage being multiplied
  for (int t = lb; t < tiles; t += G) {
    const bool hasNext = t + G < tiles;
    tile_src(hasNext ? t + G : t, srcNext);
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nStages; ++s, par ^= 1) {
      const unsigned char* cur = smem + par * STAGE;
      const bool last = s + 1 == nStages;
      // what goes into the other buffer: this tile's next stage, or stage 0 of the block's next tile (the very last
      // stage of the block re-stages the next-pointer set anyway: no branch in the pipeline)
      const size_t adv = last ? 0 : (size_t)(s + 1) * stepMine;
      f16x8 ah[8], al[8];
#pragma unroll
      for (int ms = 0; ms < 8; ++ms) {
        ah[ms] = *reinterpret_cast<const f16x8*>(cur + aoff[ms]);
        al[ms] = *reinterpret_cast<const f16x8*>(cur + A_PLANE + aoff[ms]);
      }
      f16x8 bh[2], bl[2];   // ping-pong: the next channel fragment is read while the current one is multiplied
      bh[0] = *reinterpret_cast<const f16x8*>(cur + boff);
      bl[0] = *reinterpret_cast<const f16x8*>(cur + boff + B_PLANE);
      __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cs = 0; cs < 8; ++cs) {
        const int c = cs & 1;
        if (cs < 7) {
          bh[c ^ 1] = *reinterpret_cast<const f16x8*>(cur + boff + (cs + 1) * 1024);
          bl[c ^ 1] = *reinterpret_cast<const f16x8*>(cur + boff + B_PLANE + (cs + 1) * 1024);
        }
        if (cs < 2) {   // this wave's 16 staging loads go out under the first two channel fragments
#pragma unroll
          for (int j = 0; j < 8; ++j) dma((last ? srcNext[8 * cs + j] : srcCur[8 * cs + j]) + adv, par ^ 1, 8 * cs + j);
        }
#pragma unroll
        for (int ms = 0; ms < 8; ++ms) {
          acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[c], ah[ms], acc[ms][cs], 0, 0, 0);
          acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[c], al[ms], acc[ms][cs], 0, 0, 0);
          acc[ms][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[c], ah[ms], acc[ms][cs], 0, 0, 0);
        }
        if (cs < 7) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 21, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // epilogue (the next tile's first stage is already in LDS).  With A-operand = B-matrix fragment (16 n x 32 k) and
    // B-operand = A rows (16 m x 32 k): acc[r] = C[m = li][n = 4*lq + r] of the 16 x 16 tile
    const int tm = t / tilesN, tn = t - tm * tilesN;
#pragma unroll
    for (int ms = 0; ms < 8; ++ms)
#pragma unroll
      for (int cs = 0; cs < 8; ++cs) {
        const size_t row = (size_t)tm * BM + wm * 128 + ms * 16 + li;
        const size_t col = (size_t)tn * BN + wn * 128 + cs * 16 + 4 * lq;
        *reinterpret_cast<f32x4*>(a.c + row * a.N + col) = acc[ms][cs];
      }
#pragma unroll
    for (int j = 0; j < 16; ++j) srcCur[j] = srcNext[j];
  }
}

static void split(float v, _Float16& h, _Float16& l) {
  h = (_Float16)v;
  l = (_Float16)(v - (float)h);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  if (M % BM || N % BN || K % BK) {
    printf("sizes must be multiples of %d, %d, %d\n", BM, BN, BK);
    return 1;
  }
  std::vector<float> A((size_t)M * K), B((size_t)K * N);
  srand(3);
  auto rnd = [] {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd() * 0.05f;
  std::vector<_Float16> aH(A.size()), aL(A.size()), bH(B.size()), bL(B.size());
  for (size_t i = 0; i < A.size(); ++i) split(A[i], aH[i], aL[i]);
  // B fragment order: fragment (ks, nf): lane (j = lane & 15, q = lane >> 4) holds B[k = ks*32 + q*8 + e][n = nf*16 + j]
  for (int ks = 0; ks < K / 32; ++ks)
    for (int nf = 0; nf < N / 16; ++nf)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int k = ks * 32 + (lane >> 4) * 8 + e, n = nf * 16 + (lane & 15);
          const size_t o = (((size_t)ks * (N / 16) + nf) * 64 + lane) * 8 + e;
          split(B[(size_t)k * N + n], bH[o], bL[o]);
        }
  Args a;
  _Float16 *daH, *daL, *dbH, *dbL;
  float* dc;
  hipMalloc(&daH, aH.size() * 2);
  hipMalloc(&daL, aL.size() * 2);
  hipMalloc(&dbH, bH.size() * 2);
  hipMalloc(&dbL, bL.size() * 2);
  hipMalloc(&dc, (size_t)M * N * 4);
  hipMemcpy(daH, aH.data(), aH.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(daL, aL.data(), aL.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dbH, bH.data(), bH.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dbL, bL.data(), bL.size() * 2, hipMemcpyHostToDevice);
  a.aHi = daH;
  a.aLo = daL;
  a.bHi = dbH;
  a.bLo = dbL;
  a.c = dc;
  a.M = M;
  a.N = N;
  a.K = K;
  if (hipFuncSetAttribute((const void*)gemm_x3, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
    printf("cannot set %d bytes of dynamic LDS\n", LDS_BYTES);
    return 1;
  }
  const int blocks = std::min((M / BM) * (N / BN), 256) / 8 * 8;   // persistent: at most one block per CU
  hipLaunchKernelGGL(gemm_x3, dim3(blocks), dim3(256), LDS_BYTES, 0, a);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed\n");
    return 1;
  }
  std::vector<float> C(8 * (size_t)N);
  double worst = 0;
  for (int t = 0; t < 8; ++t) {
    const int m = (t * 7919 + 13) % M;
    hipMemcpy(C.data(), dc + (size_t)m * N, N * 4, hipMemcpyDeviceToHost);
    for (int n = 0; n < N; n += 37) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)B[(size_t)k * N + n];
      worst = fmax(worst, fabs(ref - C[n]));
    }
  }
  printf("M %d N %d K %d: max |err| on sampled entries %.3e (values ~%.2f)\n", M, N, K, worst, 0.05 * sqrt((double)K));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {   // second round after ~1 s of load: settled clocks
    hipEventRecord(e0, 0);
    int n = 0;
    do {
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_x3, dim3(blocks), dim3(256), LDS_BYTES, 0, a);
      n += 5;
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    } while (ms < 1000.f);
    const double alg = 2.0 * M * N * (double)K * n / (ms * 1e-3) / 1e12;
    printf("round %d: %.3f ms per GEMM, %.1f TFLOP/s algorithmic, %.1f TFLOP/s executed (3 fp16 MFMAs per product)\n", rep,
           ms / n, alg, 3 * alg);
  }
  return 0;
}
