// HBM-bound kernels of the training step (SURVEY.md section 8 rows a2 train-mode BatchNorm, a4 backward,
// a12 BCE-with-logits, a13 Adam; reference README.md:1453, :1694-1709, :2060-2084).
//
// All reductions over pixels are two-stage and deterministic: a grid of blocks writes fp32 partial sums
// [block][k][C]; a finalize kernel adds them in double in block order.  No float atomics, so gradients
// are bitwise reproducible run to run (and across data-parallel ranks given the same inputs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_x3_ws.h"   // split_pk_f16: the hi / lo operand form of the split-operand kernels

namespace unet {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 f4zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f4 ldf4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void stf4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// Geometry shared by the column reductions: a block is `rows` x `cols` threads (rows a power of two,
// rows*cols <= 256); thread (r, c) owns float4 column c of the current 256-wide column group and walks
// pixels r, r+rows*gridDim.x... of its block's slice.
struct ColGeom {
  int cols, rows;
};
__host__ __device__ inline ColGeom col_geom(int c4) {
  ColGeom g;
  g.cols = c4 < 256 ? c4 : 256;
  int r = 256 / g.cols, p = 1;
  while (p * 2 <= r) p *= 2;
  g.rows = p;
  return g;
}

// Block-level tree reduction of K float4 accumulators over the `rows` dimension, result written by row 0.
template <int K>
__device__ __forceinline__ void block_reduce_store(f4 (&acc)[K], const ColGeom g, int r, int c, bool active,
                                                   float* partial, int C, int colBase) {
  __shared__ f4 red[K][256];
#pragma unroll
  for (int k = 0; k < K; ++k) red[k][threadIdx.x] = active ? acc[k] : f4zero();
  __syncthreads();
  for (int s = g.rows >> 1; s > 0; s >>= 1) {
    if (active && r < s) {
#pragma unroll
      for (int k = 0; k < K; ++k) red[k][r * g.cols + c] += red[k][(r + s) * g.cols + c];
    }
    __syncthreads();
  }
  if (active && r == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) stf4(partial + ((size_t)blockIdx.x * K + k) * C + (colBase + c) * 4, red[k][c]);
  }
  __syncthreads();
}

// Finalize stage of the column reductions: FIN_CH channels per block, FIN_PARTS threads per channel each
// adding a slice of the `nb` partial rows in double (four loads in flight), combined through LDS in a fixed
// order (deterministic).  Returns the K sums of channel `c` to the thread with part == 0.  All 256 threads
// of the block must call it.  (One thread per channel walking all rows took 0.2-0.3 ms per call.)
constexpr int FIN_CH = 16;
constexpr int FIN_PARTS = 256 / FIN_CH;
template <int K>
__device__ __forceinline__ void finalize_sums(const float* __restrict__ partial, int nb, int C, int c, int part,
                                              double (&out)[K]) {
  __shared__ double fin[K][FIN_PARTS][FIN_CH];
  double s[K];
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = 0.0;
  if (c < C) {
    const int per = (nb + FIN_PARTS - 1) / FIN_PARTS;
    const int b0 = part * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    int b = b0;
    for (; b + 3 < b1; b += 4)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float v0 = partial[((size_t)b * K + k) * C + c], v1 = partial[((size_t)(b + 1) * K + k) * C + c],
                    v2 = partial[((size_t)(b + 2) * K + k) * C + c], v3 = partial[((size_t)(b + 3) * K + k) * C + c];
        s[k] += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
      }
    for (; b < b1; ++b)
#pragma unroll
      for (int k = 0; k < K; ++k) s[k] += (double)partial[((size_t)b * K + k) * C + c];
  }
  const int i = threadIdx.x & (FIN_CH - 1);
#pragma unroll
  for (int k = 0; k < K; ++k) fin[k][part][i] = s[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < FIN_PARTS; ++q) t += fin[k][q][i];
    out[k] = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm statistics (training forward): per-channel sum and sum of squares of z (P pixels, C channels)
// partial: [gridDim.x][2][C]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, size_t P, int C,
                                                               float* __restrict__ partial) {
  const int c4 = C >> 2;
  const ColGeom g = col_geom(c4);
  const int r = threadIdx.x / g.cols, c = threadIdx.x - r * g.cols;
  const size_t per = (P + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per;
  const size_t p1 = p0 + per < P ? p0 + per : P;
  for (int colBase = 0; colBase < c4; colBase += 256) {
    const bool active = r < g.rows && colBase + c < c4;
    f4 acc[2] = {f4zero(), f4zero()};
    if (active) {
      // four independent loads in flight per thread (the single-load loop ran at 0.6-2.3 TB/s)
      const float* zp = z + (colBase + c) * 4;
      size_t p = p0 + r;
      const size_t st = g.rows;
      for (; p + 3 * st < p1; p += 4 * st) {
        const f4 v0 = ldf4(zp + p * C), v1 = ldf4(zp + (p + st) * C), v2 = ldf4(zp + (p + 2 * st) * C),
                 v3 = ldf4(zp + (p + 3 * st) * C);
        acc[0] += (v0 + v1) + (v2 + v3);
        acc[1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
      for (; p < p1; p += st) {
        const f4 v = ldf4(zp + p * C);
        acc[0] += v;
        acc[1] += v * v;
      }
    }
    block_reduce_store<2>(acc, g, r, c, active, partial, C, colBase);
  }
}

// mean/var from the partials (double), fused scale/shift for the apply pass, saved mean/invstd for the
// backward pass, running statistics with momentum 0.1 and the unbiased variance (reference README.md:1453
// nn.BatchNorm2d defaults).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nb, int C, double M,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ saveMean,
                                                          float* __restrict__ saveInvstd,
                                                          float* __restrict__ runMean, float* __restrict__ runVar) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), part = threadIdx.x / FIN_CH;
  double sums[2];
  finalize_sums<2>(partial, nb, C, c, part, sums);
  if (c >= C || part != 0) return;
  const double s = sums[0], ss = sums[1];
  const double mean = s / M;
  double var = ss / M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  saveMean[c] = (float)mean;
  saveInvstd[c] = invstd;
  if (runMean) {
    const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
    runMean[c] = (1.f - momentum) * runMean[c] + momentum * (float)mean;
    runVar[c] = (1.f - momentum) * runVar[c] + momentum * (float)unbiased;
  }
}

// a = relu(z*scale + shift), written with pixel stride ldo at channel offset off (concat slice) as fp32 (out, may
// be null) and / or as fp16 hi + lo planes (pHi, may be null; lo plane pLo2 32-bit words behind it; pixel stride ldp
// and channel offset offp in halfs) - the operand form of the split-operand convolutions (conv_x3_ws.h)
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ z,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, size_t P, int C,
                                                            float* __restrict__ out, int ldo, int off,
                                                            uint32_t* __restrict__ pHi, size_t pLo2, int ldp, int offp,
                                                            unsigned* err = nullptr,
                                                            const float* __restrict__ headW = nullptr,
                                                            const float* __restrict__ headB = nullptr,
                                                            float* __restrict__ logits = nullptr) {
  // headW / headB / logits (C == 64 only: the 16 threads of a pixel are the 16 lanes of head1x1_kernel<16>): the 1x1 head
  // on the activation while it is in registers, in that kernel's summation order - the same bits, one pass less
  float amax = 0.f;   // range watch of the planes (conv_x3_ws.h)
  const int c4 = C >> 2;
  const size_t total = P * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const f4 v = ldf4(z + p * C + c), sc = ldf4(scale + c), sh = ldf4(shift + c);
    f4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = v[e] * sc[e] + sh[e];
      y[e] = t > 0.f ? t : 0.f;
    }
    if (headW) {
      const f4 wv = ldf4(headW + c);
      float hs = 0.f;
      hs = fmaf(y[0], wv[0], hs);
      hs = fmaf(y[1], wv[1], hs);
      hs = fmaf(y[2], wv[2], hs);
      hs = fmaf(y[3], wv[3], hs);
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) hs += __shfl_xor(hs, d, 64);
      if (c == 0) logits[p] = hs + *headB;
    }
    if (out) stf4(out + p * (size_t)ldo + off + c, y);
    if (pHi) {
      uint32_t h0, l0, h1, l1;
      split_pk_f16(y[0], y[1], h0, l0, amax);
      split_pk_f16(y[2], y[3], h1, l1, amax);
      const size_t o = (p * (size_t)ldp + offp + c) >> 1;
      *reinterpret_cast<uint2*>(pHi + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(pHi + pLo2 + o) = make_uint2(l0, l1);
    }
  }
  x3_report_range(amax, err);
}

// bn_apply_relu_kernel for a unit whose output is also max-pooled (the encoder's second convolutions): one thread takes
// the 2 x 2 window of a pooled pixel x 4 channels, writes the four activations like bn_apply_relu_kernel does (fp32 with
// ldo / off, planes with ldp / offp) and their maximum as dense planes (qHi; lo plane qLo2 words behind) - the values
// maxpool2x2_to_planes_kernel would read back, so the results are the same bits.  h and w even.
__global__ __launch_bounds__(256) void bn_apply_relu_pool_kernel(const float* __restrict__ z,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int n, int h, int w, int C,
                                                                 float* __restrict__ out, int ldo, int off,
                                                                 uint32_t* __restrict__ pHi, size_t pLo2, int ldp, int offp,
                                                                 uint32_t* __restrict__ qHi, size_t qLo2,
                                                                 unsigned* err = nullptr) {
  float amax = 0.f;
  const int c4 = C >> 2;
  const int oh = h >> 1, ow = w >> 1;
  const size_t total = (size_t)n * oh * ow * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const size_t img = t / oh;
    const size_t pix00 = (img * h + (size_t)oy * 2) * w + (size_t)ox * 2;
    const f4 sc = ldf4(scale + c), sh = ldf4(shift + c);
    f4 m = f4zero();   // activations are >= 0
    f4 zq[4];          // the window's four loads before the first store (else each waits for its own round trip)
#pragma unroll
    for (int q = 0; q < 4; ++q) zq[q] = ldf4(z + (pix00 + (q >> 1) * (size_t)w + (q & 1)) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t p = pix00 + (q >> 1) * (size_t)w + (q & 1);
      const f4 v = zq[q];
      f4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float tt = v[e] * sc[e] + sh[e];
        y[e] = tt > 0.f ? tt : 0.f;
        m[e] = fmaxf(m[e], y[e]);
      }
      if (out) stf4(out + p * (size_t)ldo + off + c, y);
      if (pHi) {
        uint32_t h0, l0, h1, l1;
        split_pk_f16(y[0], y[1], h0, l0, amax);
        split_pk_f16(y[2], y[3], h1, l1, amax);
        const size_t o = (p * (size_t)ldp + offp + c) >> 1;
        *reinterpret_cast<uint2*>(pHi + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(pHi + pLo2 + o) = make_uint2(l0, l1);
      }
    }
    uint32_t h0, l0, h1, l1;
    split_pk_f16(m[0], m[1], h0, l0);
    split_pk_f16(m[2], m[3], h1, l1);
    *reinterpret_cast<uint2*>(qHi + i * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(qHi + qLo2 + i * 2) = make_uint2(l0, l1);
  }
  x3_report_range(amax, err);
}

// fp32 (pixel stride ldi, channel offset offi, C channels) -> hi / lo planes (pixel stride ldp, offset offp halfs)
__global__ __launch_bounds__(256) void split_planes_strided_kernel(const float* __restrict__ x, int ldi, int offi,
                                                                   size_t P, int C, uint32_t* __restrict__ pHi,
                                                                   size_t pLo2, int ldp, int offp) {
  const int c4 = C >> 2;
  const size_t total = P * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const f4 y = ldf4(x + p * (size_t)ldi + offi + c);
    uint32_t h0, l0, h1, l1;
    split_pk_f16(y[0], y[1], h0, l0);
    split_pk_f16(y[2], y[3], h1, l1);
    const size_t o = (p * (size_t)ldp + offp + c) >> 1;
    *reinterpret_cast<uint2*>(pHi + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(pHi + pLo2 + o) = make_uint2(l0, l1);
  }
}

// MaxPool2d(2,2) of an fp32 tensor (pixel stride ldi) written as dense hi / lo planes (n, h/2, w/2, c)
__global__ __launch_bounds__(256) void maxpool2x2_to_planes_kernel(const float* __restrict__ x, int n, int h, int w,
                                                                   int c, int ldi, uint32_t* __restrict__ pHi,
                                                                   size_t pLo2) {
  const int c4 = c >> 2;
  const int oh = h >> 1, ow = w >> 1;
  const size_t total = (size_t)n * oh * ow * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const size_t img = t / oh;
    const size_t pix00 = (img * h + (size_t)oy * 2) * w + (size_t)ox * 2;
    const f4 a = ldf4(x + pix00 * (size_t)ldi + cv), b = ldf4(x + (pix00 + 1) * (size_t)ldi + cv);
    const f4 cc = ldf4(x + (pix00 + w) * (size_t)ldi + cv), d = ldf4(x + (pix00 + w + 1) * (size_t)ldi + cv);
    f4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(cc[e], d[e]));
    uint32_t h0, l0, h1, l1;
    split_pk_f16(m[0], m[1], h0, l0);
    split_pk_f16(m[2], m[3], h1, l1);
    const size_t o = i * 2;
    *reinterpret_cast<uint2*>(pHi + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(pHi + pLo2 + o) = make_uint2(l0, l1);
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward.  dA: gradient w.r.t. the post-ReLU activation (pixel stride ldd, offset offd).
//   dY = dA * [z*scale+shift > 0];  xhat = (z - mean) * invstd
//   pass 1: partial sums of dY (-> dbeta) and dY*xhat (-> dgamma)       partial: [grid][2][C]
//   pass 2: dZ = scale * (dY - dbeta/M - xhat * dgamma/M)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dA, int ldd, int offd,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, size_t P, int C,
                                                             float* __restrict__ partial,
                                                             unsigned* __restrict__ boundKeys = nullptr,
                                                             const float* __restrict__ r1p = nullptr,
                                                             const float* __restrict__ r1c = nullptr) {
  // r1p / r1c (the unit under the 1x1 head): dA[p, c] = r1p[p] * r1c[c] is formed here instead of being read - the same
  // single multiplication head_bwd_kernel would have stored, so the sums are the same bits (dA is then ignored)
  const int c4 = C >> 2;
  const ColGeom g = col_geom(c4);
  const int r = threadIdx.x / g.cols, c = threadIdx.x - r * g.cols;
  const size_t per = (P + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per;
  const size_t p1 = p0 + per < P ? p0 + per : P;
  float mdy = 0.f, mxh = 0.f;   // max |dY|, max |xhat| seen by this thread (boundKeys: see bn_bwd_scale_exponent)
  for (int colBase = 0; colBase < c4; colBase += 256) {
    const bool active = r < g.rows && colBase + c < c4;
    f4 acc[2] = {f4zero(), f4zero()};
    if (active) {
      const int ch = (colBase + c) * 4;
      const f4 sc = ldf4(scale + ch), sh = ldf4(shift + ch), mu = ldf4(mean + ch), is = ldf4(invstd + ch);
      const f4 r1w = r1p ? ldf4(r1c + ch) : f4zero();
      size_t p = p0 + r;
      const size_t st = g.rows;
      for (; p + st < p1; p += 2 * st) {   // two pixels (four loads) in flight per thread
        const f4 z0 = ldf4(z + p * C + ch), z1 = ldf4(z + (p + st) * C + ch);
        const f4 d0 = r1p ? r1w * r1p[p] : ldf4(dA + p * (size_t)ldd + offd + ch);
        const f4 d1 = r1p ? r1w * r1p[p + st] : ldf4(dA + (p + st) * (size_t)ldd + offd + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dy0 = (z0[e] * sc[e] + sh[e] > 0.f) ? d0[e] : 0.f;
          const float dy1 = (z1[e] * sc[e] + sh[e] > 0.f) ? d1[e] : 0.f;
          const float xh0 = (z0[e] - mu[e]) * is[e], xh1 = (z1[e] - mu[e]) * is[e];
          acc[0][e] += dy0 + dy1;
          acc[1][e] += dy0 * xh0 + dy1 * xh1;
          mdy = fmaxf(mdy, fmaxf(fabsf(dy0), fabsf(dy1)));
          mxh = fmaxf(mxh, fmaxf(fabsf(xh0), fabsf(xh1)));
        }
      }
      for (; p < p1; p += st) {
        const f4 zv = ldf4(z + p * C + ch);
        const f4 d = r1p ? r1w * r1p[p] : ldf4(dA + p * (size_t)ldd + offd + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dy = (zv[e] * sc[e] + sh[e] > 0.f) ? d[e] : 0.f;
          const float xh = (zv[e] - mu[e]) * is[e];
          acc[0][e] += dy;
          acc[1][e] += dy * xh;
          mdy = fmaxf(mdy, fabsf(dy));
          mxh = fmaxf(mxh, fabsf(xh));
        }
      }
    }
    block_reduce_store<2>(acc, g, r, c, active, partial, C, colBase);
  }
  if (boundKeys) {   // order keys of non-negative floats; maxima are exact and order independent
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      mdy = fmaxf(mdy, __shfl_xor(mdy, m, 64));
      mxh = fmaxf(mxh, __shfl_xor(mxh, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      if (__float_as_uint(mdy) > *reinterpret_cast<volatile unsigned*>(boundKeys + 0)) atomicMax(boundKeys + 0, __float_as_uint(mdy));
      if (__float_as_uint(mxh) > *reinterpret_cast<volatile unsigned*>(boundKeys + 1)) atomicMax(boundKeys + 1, __float_as_uint(mxh));
    }
  }
}

// Sums the partials in double; writes dbeta, dgamma (the parameter gradients) as floats.
// boundKeys / scale (optional): folds max |scale|, max |out0|, max |out1| over the channels into boundKeys[2..4]
__global__ __launch_bounds__(256) void reduce2_finalize_kernel(const float* __restrict__ partial, int nb, int C,
                                                               float* __restrict__ out0, float* __restrict__ out1,
                                                               unsigned* __restrict__ boundKeys = nullptr,
                                                               const float* __restrict__ scale = nullptr) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), part = threadIdx.x / FIN_CH;
  double sums[2];
  finalize_sums<2>(partial, nb, C, c, part, sums);
  if (c >= C || part != 0) return;
  if (out0) out0[c] = (float)sums[0];
  if (out1) out1[c] = (float)sums[1];
  if (boundKeys) {
    atomicMax(boundKeys + 2, __float_as_uint(fabsf(scale[c])));
    atomicMax(boundKeys + 3, __float_as_uint(fabsf((float)sums[0])));
    atomicMax(boundKeys + 4, __float_as_uint(fabsf((float)sums[1])));
  }
}

// Power-of-two exponent k that brings dZ = scale (dY - dbeta/M - xhat dgamma/M) into the fp16 range without ever
// leaving it: |dZ| <= max|scale| (max|dY| + max|dbeta|/M + max|xhat| max|dgamma|/M) =: B, and 2^k B lies in
// [2^14, 2^15).  The five maxima are collected by bn_bwd_partial_kernel and reduce2_finalize_kernel; every thread
// that evaluates this gets the same k.  (A bound, not the maximum: the planes can be written in the same pass that
// computes dZ.  It overshoots the true maximum by the spread of the per-channel scales, a few bits, which the
// fp16 exponent range absorbs: values 2^-12 of the largest keep all 22 bits.)
__device__ __forceinline__ int bn_bwd_scale_exponent(const unsigned* __restrict__ keys, float invM) {
  const float mdy = __uint_as_float(keys[0]), mxh = __uint_as_float(keys[1]), msc = __uint_as_float(keys[2]);
  const float mdb = __uint_as_float(keys[3]), mdg = __uint_as_float(keys[4]);
  const float bound = msc * (mdy + mdb * invM + mxh * mdg * invM);
  const unsigned bits = __float_as_uint(bound);
  int k = 0;
  if ((bits >> 23) != 0 && (bits >> 23) < 255) k = 14 - ((int)(bits >> 23) - 127);   // normal, finite bound
  return k > 100 ? 100 : (k < -100 ? -100 : k);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dA, int ldd, int offd,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float invM, size_t P,
                                                           int C, float* __restrict__ dZ,
                                                           unsigned* __restrict__ absmaxKey,
                                                           const unsigned* __restrict__ boundKeys = nullptr,
                                                           uint32_t* __restrict__ pHi = nullptr, size_t pLo2 = 0,
                                                           float* __restrict__ invOut = nullptr,
                                                           const float* __restrict__ r1p = nullptr,
                                                           const float* __restrict__ r1c = nullptr) {
  // r1p / r1c: dA[p, c] = r1p[p] * r1c[c] formed here (bn_bwd_partial_kernel)
  // dZ (fp32, may be null) and / or, with boundKeys, dZ * 2^k as dense fp16 hi + lo planes (pHi; lo plane pLo2 32-bit
  // words behind), 2^-k left in *invOut
  const int c4 = C >> 2;
  const size_t total = P * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  float up = 1.f;
  if (pHi) {
    const int k = bn_bwd_scale_exponent(boundKeys, invM);
    up = ldexpf(1.f, k);
    if (blockIdx.x == 0 && threadIdx.x == 0) *invOut = ldexpf(1.f, -k);
  }
  float amax = 0.f;   // max |dZ| of this thread: the fp16 input-gradient convolution scales dZ into range with it
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const f4 zv = ldf4(z + p * C + c);
    const f4 d = r1p ? ldf4(r1c + c) * r1p[p] : ldf4(dA + p * (size_t)ldd + offd + c);
    const f4 sc = ldf4(scale + c), sh = ldf4(shift + c), mu = ldf4(mean + c), is = ldf4(invstd + c);
    const f4 db = ldf4(dbeta + c), dg = ldf4(dgamma + c);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dy = (zv[e] * sc[e] + sh[e] > 0.f) ? d[e] : 0.f;
      const float xh = (zv[e] - mu[e]) * is[e];
      o[e] = sc[e] * (dy - db[e] * invM - xh * dg[e] * invM);
      amax = fmaxf(amax, fabsf(o[e]));
    }
    if (dZ) stf4(dZ + p * C + c, o);
    if (pHi) {
      uint32_t h0, l0, h1, l1;
      split_pk_f16(o[0] * up, o[1] * up, h0, l0);
      split_pk_f16(o[2] * up, o[3] * up, h1, l1);
      *reinterpret_cast<uint2*>(pHi + i * 2) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(pHi + pLo2 + i * 2) = make_uint2(l0, l1);
    }
  }
  if (absmaxKey) {   // non-negative floats order like their bit patterns; max is exact and order independent
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
    // tens of thousands of waves on one address would serialise (~12 ns per atomic): only a wave that beats the value
    // it can see (a plain, possibly stale read of a monotone word) issues one
    if ((threadIdx.x & 63) == 0 && __float_as_uint(amax) > *reinterpret_cast<volatile unsigned*>(absmaxKey))
      atomicMax(absmaxKey, __float_as_uint(amax));
  }
}

// ---------------------------------------------------------------------------------------------------
// MaxPool2d(2,2) backward fused with the skip-connection add:
//   dA[n,y,x,c] = dSkip[n,y,x,c] + (first argmax of the 2x2 window of a ? dPool[n,y/2,x/2,c] : 0)
// a and dSkip are read with pixel strides (they live in concat-shaped buffers); dA is dense.
// PyTorch routes the gradient to the FIRST maximum in row-major window order; ties are kept identical.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_bwd_add_kernel(const float* __restrict__ a, int lda,
                                                              const float* __restrict__ dSkip, int lds, int offs,
                                                              const float* __restrict__ dPool, int n, int h, int w,
                                                              int c, float* __restrict__ dA) {
  const int c4 = c >> 2;
  const int oh = h >> 1, ow = w >> 1;
  const size_t total = (size_t)n * oh * ow * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const size_t img = t / oh;
    const size_t pix00 = (img * h + (size_t)oy * 2) * w + (size_t)ox * 2;
    const size_t pix[4] = {pix00, pix00 + 1, pix00 + w, pix00 + w + 1};
    f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ldf4(a + pix[k] * (size_t)lda + cv);
    const f4 g = ldf4(dPool + i * 4);
    f4 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = dSkip ? ldf4(dSkip + pix[k] * (size_t)lds + offs + cv) : f4zero();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int best = 0;
      float m = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > m) {
          m = v[k][e];
          best = k;
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] += (k == best) ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) stf4(dA + pix[k] * (size_t)c + cv, o[k]);
  }
}

// maxpool_bwd_add_kernel + the first pass of the BatchNorm backward of the unit that produced `a` (bn_bwd_partial_kernel)
// in one pass: while a window's four gradients are in registers they are also masked (a > 0: the ReLU), multiplied with
// xhat = (z - mean) * invstd and summed per channel - partial [grid][2][C] and the two maxima exactly as
// bn_bwd_partial_kernel leaves them - so the separate pass does not read dA back.  A block takes a contiguous slice of
// the pooled pixels; thread (r, c) keeps float4 channel column c (bn_bwd_partial_kernel's geometry).
__global__ __launch_bounds__(256) void maxpool_bwd_add_bnstat_kernel(const float* __restrict__ a, int lda,
                                                                     const float* __restrict__ dSkip, int lds, int offs,
                                                                     const float* __restrict__ dPool, int n, int h, int w,
                                                                     int C, float* __restrict__ dA,
                                                                     const float* __restrict__ z,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     float* __restrict__ partial,
                                                                     unsigned* __restrict__ boundKeys) {
  const int c4 = C >> 2;
  const ColGeom g = col_geom(c4);
  const int r = threadIdx.x / g.cols, c = threadIdx.x - r * g.cols;
  const int oh = h >> 1, ow = w >> 1;
  const size_t Pw = (size_t)n * oh * ow;
  const size_t per = (Pw + gridDim.x - 1) / gridDim.x;
  const size_t w0 = (size_t)blockIdx.x * per;
  const size_t w1 = w0 + per < Pw ? w0 + per : Pw;
  float mdy = 0.f, mxh = 0.f;
  for (int colBase = 0; colBase < c4; colBase += 256) {
    const bool active = r < g.rows && colBase + c < c4;
    f4 acc[2] = {f4zero(), f4zero()};
    if (active) {
      const int cv = (colBase + c) * 4;
      const f4 mu = ldf4(mean + cv), is = ldf4(invstd + cv);
      for (size_t wi = w0 + r; wi < w1; wi += g.rows) {
        const int ox = (int)(wi % ow);
        const size_t t = wi / ow;
        const int oy = (int)(t % oh);
        const size_t img = t / oh;
        const size_t pix00 = (img * h + (size_t)oy * 2) * w + (size_t)ox * 2;
        const size_t pix[4] = {pix00, pix00 + 1, pix00 + w, pix00 + w + 1};
        f4 v[4], zv[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ldf4(a + pix[k] * (size_t)lda + cv);
#pragma unroll
        for (int k = 0; k < 4; ++k) zv[k] = ldf4(z + pix[k] * (size_t)C + cv);
        const f4 gp = ldf4(dPool + wi * (size_t)C + cv);
        if (dSkip) {   // (one uniform branch around four loads: a select per load serialised them)
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = ldf4(dSkip + pix[k] * (size_t)lds + offs + cv);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = f4zero();
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int best = 0;
          float m = v[0][e];
#pragma unroll
          for (int k = 1; k < 4; ++k)
            if (v[k][e] > m) {
              m = v[k][e];
              best = k;
            }
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k][e] += (k == best) ? gp[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          stf4(dA + pix[k] * (size_t)C + cv, o[k]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dy = v[k][e] > 0.f ? o[k][e] : 0.f;
            const float xh = (zv[k][e] - mu[e]) * is[e];
            acc[0][e] += dy;
            acc[1][e] += dy * xh;
            mdy = fmaxf(mdy, fabsf(dy));
            mxh = fmaxf(mxh, fabsf(xh));
          }
        }
      }
    }
    block_reduce_store<2>(acc, g, r, c, active, partial, C, colBase);
  }
  if (boundKeys) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      mdy = fmaxf(mdy, __shfl_xor(mdy, m, 64));
      mxh = fmaxf(mxh, __shfl_xor(mxh, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      if (__float_as_uint(mdy) > *reinterpret_cast<volatile unsigned*>(boundKeys + 0)) atomicMax(boundKeys + 0, __float_as_uint(mdy));
      if (__float_as_uint(mxh) > *reinterpret_cast<volatile unsigned*>(boundKeys + 1)) atomicMax(boundKeys + 1, __float_as_uint(mxh));
    }
  }
}

// Space-to-depth of a strided hi-res gradient slice: S[n,y,x,(a*2+b)*C + co] = dY[n,2y+a,2x+b, off+co]
// (turns the ConvTranspose2d backward into 1x1 GEMMs).  One thread per float4 of S.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ dY, int ldd, int off, int n,
                                                             int h, int w, int c, float* __restrict__ S) {
  const int c4 = c >> 2;
  const size_t total = (size_t)n * h * w * 4 * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ab = (int)(t & 3);
    t >>= 2;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const size_t img = t / h;
    const size_t src = ((img * 2 * h + (size_t)2 * y + (ab >> 1)) * (2 * (size_t)w) + (size_t)2 * x + (ab & 1));
    stf4(S + i * 4, ldf4(dY + src * (size_t)ldd + off + cv));
  }
}

// Per-channel sum over pixels (bias gradients): partial [grid][1][C]
// absKey (optional): max |x| over the slice folded into *absKey as an order key (zero it first)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ldx, int off, size_t P,
                                                             int C, float* __restrict__ partial,
                                                             unsigned* __restrict__ absKey = nullptr) {
  const int c4 = C >> 2;
  const ColGeom g = col_geom(c4);
  const int r = threadIdx.x / g.cols, c = threadIdx.x - r * g.cols;
  const size_t per = (P + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per;
  const size_t p1 = p0 + per < P ? p0 + per : P;
  f4 am = f4zero();
  for (int colBase = 0; colBase < c4; colBase += 256) {
    const bool active = r < g.rows && colBase + c < c4;
    f4 acc[1] = {f4zero()};
    if (active) {
      const float* xp = x + off + (colBase + c) * 4;
      size_t p = p0 + r;
      const size_t st = g.rows;
      for (; p + 3 * st < p1; p += 4 * st) {
        const f4 v0 = ldf4(xp + p * (size_t)ldx), v1 = ldf4(xp + (p + st) * (size_t)ldx);
        const f4 v2 = ldf4(xp + (p + 2 * st) * (size_t)ldx), v3 = ldf4(xp + (p + 3 * st) * (size_t)ldx);
        acc[0] += (v0 + v1) + (v2 + v3);
        if (absKey) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            am[e] = fmaxf(fmaxf(am[e], fmaxf(fabsf(v0[e]), fabsf(v1[e]))), fmaxf(fabsf(v2[e]), fabsf(v3[e])));
        }
      }
      for (; p < p1; p += st) {
        const f4 v = ldf4(xp + p * (size_t)ldx);
        acc[0] += v;
        if (absKey) {
#pragma unroll
          for (int e = 0; e < 4; ++e) am[e] = fmaxf(am[e], fabsf(v[e]));
        }
      }
    }
    block_reduce_store<1>(acc, g, r, c, active, partial, C, colBase);
  }
  if (absKey) {
    float m = fmaxf(fmaxf(am[0], am[1]), fmaxf(am[2], am[3]));
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *reinterpret_cast<volatile unsigned*>(absKey))
      atomicMax(absKey, __float_as_uint(m));
  }
}

// Space-to-depth of a strided hi-res gradient slice straight into operand form: S[n,y,x,(a*2+b)*C + co] * 2^k as
// dense fp16 hi / lo planes, k from the slice's max |.| (absKey, as split_planes_scaled_kernel); 2^-k left in *invOut
__global__ __launch_bounds__(256) void space_to_depth_planes_kernel(const float* __restrict__ dY, int ldd, int off, int n,
                                                                    int h, int w, int c,
                                                                    const unsigned* __restrict__ absKey,
                                                                    uint32_t* __restrict__ pHi, size_t pLo2,
                                                                    float* __restrict__ invOut) {
  const unsigned key = *absKey;
  int k = 0;
  if ((key >> 23) != 0 && (key >> 23) < 255) k = 13 - ((int)(key >> 23) - 127);
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float up = ldexpf(1.f, k);
  if (blockIdx.x == 0 && threadIdx.x == 0) *invOut = ldexpf(1.f, -k);
  const int c4 = c >> 2;
  const size_t total = (size_t)n * h * w * 4 * c4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int cv = (int)(i % c4) * 4;
    size_t t = i / c4;
    const int ab = (int)(t & 3);
    t >>= 2;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const size_t img = t / h;
    const size_t src = ((img * 2 * h + (size_t)2 * y + (ab >> 1)) * (2 * (size_t)w) + (size_t)2 * x + (ab & 1));
    const f4 v = ldf4(dY + src * (size_t)ldd + off + cv);
    uint32_t h0, l0, h1, l1;
    split_pk_f16(v[0] * up, v[1] * up, h0, l0);
    split_pk_f16(v[2] * up, v[3] * up, h1, l1);
    *reinterpret_cast<uint2*>(pHi + i * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(pHi + pLo2 + i * 2) = make_uint2(l0, l1);
  }
}

// out[c % Cfold] += sum_b partial[b][c]: folds the 4 (a,b) groups of the space-to-depth layout into one bias
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nb, int C,
                                                              int Cfold, float* __restrict__ out) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), part = threadIdx.x / FIN_CH;
  double total = 0.0;
  for (int cc = c; cc < C; cc += Cfold) {   // uniform trip count across the block except for c >= Cfold lanes
    double sums[1];
    finalize_sums<1>(partial, nb, C, c < Cfold ? cc : C, part, sums);
    __syncthreads();
    total += sums[0];
  }
  if (c < Cfold && part == 0) out[c] = (float)total;
}

// ---------------------------------------------------------------------------------------------------
// BCE-with-logits, mean reduction (reference README.md:1694-1709): loss_i = max(x,0) - x*t + log1p(exp(-|x|)),
// dlogit_i = (sigmoid(x) - t) / numel.  partial: [grid] floats of loss sums.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_loss_grad_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            size_t n, float invN, float* __restrict__ dx,
                                                            float* __restrict__ partial) {
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float xv = x[i], tv = t[i];
    const float ax = fabsf(xv);
    const float e = expf(-ax);
    s += fmaxf(xv, 0.f) - xv * tv + log1pf(e);
    // sigmoid without overflow: x>=0: 1/(1+e), x<0: e/(1+e)
    const float sg = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dx[i] = (sg - tv) * invN;
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------
// BCE + Dice loss of the reference's training script (README.md:1855-1893, used at :2169-2170 with
// bce_weight = dice_weight = 0.5, pos_weight = 3):
//   L = wb * mean_i( -[pw t log s + (1-t) log(1-s)] ) + wd * (1 - (2 I + eps) / (P + T + eps)),
//   s = sigmoid(x), I = sum s t, P = sum s, T = sum t.
// Pass 1 reduces {BCE sum, I, P, T} (partials [grid][4]); the finalize kernel turns them into the three
// loss values and the two scalars the gradient pass needs; pass 2 writes
//   dL/dx_i = wb/N * (s (1 - t + pw t) - pw t) - wd * s (1 - s) * (2 t (P+T+eps) - (2I+eps)) / (P+T+eps)^2.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_dice_partial_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ t, size_t n, float pw,
                                                               float* __restrict__ partial) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float xv = x[i], tv = t[i];
    const float e = expf(-fabsf(xv));
    const float l1p = log1pf(e);
    // log s = min(x,0) - log1p(e),  log(1-s) = -max(x,0) - log1p(e)
    const float logs = fminf(xv, 0.f) - l1p, log1ms = -fmaxf(xv, 0.f) - l1p;
    const float sg = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    acc[0] += -(pw * tv * logs + (1.f - tv) * log1ms);
    acc[1] += sg * tv;
    acc[2] += sg;
    acc[3] += tv;
  }
  __shared__ float red[4][256];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// out[0] = total, out[1] = bce, out[2] = dice loss; coef[0] = 1/(P+T+eps), coef[1] = (2I+eps)/(P+T+eps)^2
__global__ void bce_dice_finalize_kernel(const float* __restrict__ partial, int nb, double n, float wb, float wd,
                                         float smooth, float* __restrict__ out, float* __restrict__ coef) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double s[4] = {0, 0, 0, 0};
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < 4; ++k) s[k] += (double)partial[(size_t)b * 4 + k];
  const double bce = s[0] / n;
  const double den = s[2] + s[3] + (double)smooth;
  const double dice = (2.0 * s[1] + (double)smooth) / den;
  out[0] = (float)(wb * bce + wd * (1.0 - dice));
  out[1] = (float)bce;
  out[2] = (float)(1.0 - dice);
  coef[0] = (float)(1.0 / den);
  coef[1] = (float)((2.0 * s[1] + (double)smooth) / (den * den));
}

__global__ __launch_bounds__(256) void bce_dice_grad_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            size_t n, float wbOverN, float wd, float pw,
                                                            const float* __restrict__ coef, float* __restrict__ dx) {
  const float invDen = coef[0], numOverDen2 = coef[1];
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float xv = x[i], tv = t[i];
    const float e = expf(-fabsf(xv));
    const float sg = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    const float dbce = sg * (1.f - tv + pw * tv) - pw * tv;
    const float ddice = sg * (1.f - sg) * (2.f * tv * invDen - numOverDen2);   // dD/dx
    dx[i] = wbOverN * dbce - wd * ddice;
  }
}

// ---------------------------------------------------------------------------------------------------
// Dice metric of the reference's validation loop (README.md:2115-2120, called at :2103-2104):
//   pred = sigmoid(logits) > 0.5  (= logit > thr), dice = (2 sum(pred t) + eps) / (sum pred + sum t + eps).
// pred and t are 0/1, so per-thread counts are exact in float up to 2^24 elements per thread; block partials are
// added in double.  partial: [grid][3]; out[0] = dice, out[1..3] = intersection, sum pred, sum t.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_metric_partial_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ t, size_t n, float thr,
                                                                  float* __restrict__ partial) {
  float acc[3] = {0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pv = x[i] > thr ? 1.f : 0.f, tv = t[i];
    acc[0] += pv * tv;
    acc[1] += pv;
    acc[2] += tv;
  }
  __shared__ float red[3][256];
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3) partial[(size_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void dice_metric_finalize_kernel(const float* __restrict__ partial, int nb, float smooth,
                                            float* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double s[3] = {0, 0, 0};
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < 3; ++k) s[k] += (double)partial[(size_t)b * 3 + k];
  out[0] = (float)((2.0 * s[0] + (double)smooth) / (s[1] + s[2] + (double)smooth));
  out[1] = (float)s[0];
  out[2] = (float)s[1];
  out[3] = (float)s[2];
}

__global__ void scalar_sum_finalize_kernel(const float* __restrict__ partial, int nb, double scale,
                                           float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += (double)partial[b];
    *out = (float)(s * scale);
  }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 head backward: dA[p,c] = dl[p]*w[c]; partial[b][0][c] = sum_p dl[p]*a[p,c]; partial[b][1][c] = sum_p dl[p]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ a,
                                                       const float* __restrict__ w, size_t P, int C,
                                                       float* __restrict__ dA, float* __restrict__ partial) {
  const int c4 = C >> 2;
  const ColGeom g = col_geom(c4);
  const int r = threadIdx.x / g.cols, c = threadIdx.x - r * g.cols;
  const size_t per = (P + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per;
  const size_t p1 = p0 + per < P ? p0 + per : P;
  for (int colBase = 0; colBase < c4; colBase += 256) {
    const bool active = r < g.rows && colBase + c < c4;
    f4 acc[2] = {f4zero(), f4zero()};
    if (active) {
      const int ch = (colBase + c) * 4;
      const f4 wv = ldf4(w + ch);
      for (size_t p = p0 + r; p < p1; p += g.rows) {
        const float d = dl[p];
        const f4 av = ldf4(a + p * C + ch);
        acc[0] += av * d;
        acc[1] += (f4){d, d, d, d};
        if (dA) stf4(dA + p * C + ch, wv * d);   // null: the consumer forms dl[p] * w[c] itself (bn_bwd_*_kernel, r1p / r1c)
      }
    }
    block_reduce_store<2>(acc, g, r, c, active, partial, C, colBase);
  }
}

// ---------------------------------------------------------------------------------------------------
// Adam / AdamW over the flat parameter buffer (torch.optim.Adam semantics, reference README.md:2071-2079,
// :2173).  bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed on the host in double.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                   float beta1, float beta2, float eps, float wd, int decoupled,
                                                   float bc1, float bc2sqrt, float gradScale) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pv = p[i];
    float gv = g[i] * gradScale;
    if (wd != 0.f) {
      if (decoupled)
        pv *= (1.f - lr * wd);
      else
        gv += wd * pv;
    }
    const float mv = beta1 * m[i] + (1.f - beta1) * gv;
    const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
    const float denom = sqrtf(vv) / bc2sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
  }
}

// ---------------------------------------------------------------------------------------------------
// Device-side repack of PyTorch-layout weights into MFMA B-fragment order (see igemm_f32.h):
//   out[(((cs*nChunks + kc)*taps + t)*64 + lane)*KPL + e] = W(n = cs*16 + (lane&15), ci = kc*CK + (lane>>4)*KPL + e, t)
// mode 0: conv forward     W(n,ci,t) = w[(n*cinReal + ci)*9 + t]                 (w is (O,I,3,3))
// mode 1: conv dgrad       W(n,ci,t) = w[(ci*nReal + n)*9 + (8-t)]               (n = fwd in-channel, ci = fwd out-channel)
// mode 2: upconv forward   n = ab*coutPad + co:  W = w[(ci*coutReal + co)*4 + ab]   (w is (I,O,2,2), taps = 1)
// mode 4/5: Winograd-transformed forward / dgrad filters (taps = 16 points), see the kernel body
// mode 3: upconv dgrad     GEMM K = (ab, co) over the space-to-depth gradient, N = fwd in-channel:
//                          ci = ab*coutReal + co:  W(n,ci) = w[(n*coutReal + co)*4 + ab]
// ---------------------------------------------------------------------------------------------------
struct PackArgs {
  const float* w;
  float* out;
  int mode, ck, taps, nChunks, nSub;
  int nReal;    // valid columns (mode 2: coutReal per group)
  int kReal;    // valid k (input channels of this GEMM)
  int coutPad;  // mode 2 only
  int aux;      // mode 0: cinReal; mode 1: nReal (= fwd cin); mode 2/3: coutReal
};

// elements of one packed operand
__host__ __device__ inline size_t pack_total(const PackArgs& a) {
  return (size_t)a.nSub * a.nChunks * a.taps * 64 * (a.ck / 4);
}

// block `blk` of `nblk` blocks working on operand a
__device__ __forceinline__ void pack_weights_body(const PackArgs& a, unsigned blk, unsigned nblk) {
  const int kpl = a.ck / 4;
  const size_t total = pack_total(a);
  const size_t stride = (size_t)nblk * 256;
  for (size_t i = (size_t)blk * 256 + threadIdx.x; i < total; i += stride) {
    const int e = (int)(i % kpl);
    size_t t = i / kpl;
    const int lane = (int)(t & 63);
    t >>= 6;
    const int tap = (int)(t % a.taps);
    t /= a.taps;
    const int kc = (int)(t % a.nChunks);
    const int cs = (int)(t / a.nChunks);
    const int n = cs * 16 + (lane & 15);
    const int ci = kc * a.ck + (lane >> 4) * kpl + e;
    float v = 0.f;
    if (a.mode == 0) {
      if (n < a.nReal && ci < a.kReal) v = a.w[((size_t)n * a.kReal + ci) * 9 + tap];
    } else if (a.mode == 1) {
      if (n < a.nReal && ci < a.kReal) v = a.w[((size_t)ci * a.nReal + n) * 9 + (8 - tap)];
    } else if (a.mode == 4 || a.mode == 5) {
      // Winograd F(2x2,3x3) weight transform U = G g G^T at point tap = pa*4+pb (taps == 16);
      // mode 4 reads g as the forward filter, mode 5 as the dgrad filter (transposed, flipped)
      if (n < a.nReal && ci < a.kReal) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const int pa = tap >> 2, pb = tap & 3;
        const float* g = a.mode == 4 ? a.w + ((size_t)n * a.kReal + ci) * 9 : a.w + ((size_t)ci * a.nReal + n) * 9;
        float u = 0.f;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int y = 0; y < 3; ++y) {
            const float gv = a.mode == 4 ? g[x * 3 + y] : g[8 - (x * 3 + y)];
            u += G[pa][x] * gv * G[pb][y];
          }
        v = u;
      }
    } else if (a.mode == 2) {
      const int ab = n / a.coutPad, co = n - ab * a.coutPad;
      if (ab < 4 && co < a.nReal && ci < a.kReal) v = a.w[((size_t)ci * a.nReal + co) * 4 + ab];
    } else {
      const int ab = ci / a.aux, co = ci - ab * a.aux;
      if (n < a.nReal && ab < 4) v = a.w[((size_t)n * a.aux + co) * 4 + ab];
    }
    a.out[i] = v;
  }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) { pack_weights_body(a, blockIdx.x, gridDim.x); }

// Every packed operand of the network in ONE launch (the repack after each optimizer step was ~100 launches of a few
// microseconds of work each): descs[d] owns blocks [blockStart[d], blockStart[d+1]).
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackArgs* __restrict__ descs,
                                                                 const unsigned* __restrict__ blockStart, int nDesc) {
  int lo = 0, hi = nDesc;   // largest d with blockStart[d] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blockStart[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const PackArgs a = descs[lo];
  pack_weights_body(a, blockIdx.x - blockStart[lo], blockStart[lo + 1] - blockStart[lo]);
}

// max |x| as an order key (the bit pattern of a non-negative float), atomically folded into *key (zero it first)
__global__ __launch_bounds__(256) void absmax_key_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ key) {
  float amax = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) amax = fmaxf(amax, fabsf(x[i]));
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(key, __float_as_uint(amax));
}

}  // namespace unet
