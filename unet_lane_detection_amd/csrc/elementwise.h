// HBM-bound helper kernels of the forward path: input packing (SURVEY.md section 8
// rows a8/a9), MaxPool2d(2,2) (a4) and the 1x1 head (a7, a10 threshold).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {

typedef float f32x4e __attribute__((ext_vector_type(4)));

// (N,H,W,3) uint8 RGB -> (N,H,W,4) fp32, (u8 - mean) / std per channel, pad channel = 0.
// Reference: normalisation constants README.md:3110-3111, input layout src/unet.py:39-40.
// One thread per pixel: 3 byte loads (contiguous across lanes: 192 B per wave), one 16-byte store.
__global__ __launch_bounds__(256) void pack_u8_nhwc4_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                            size_t npix, float m0, float m1, float m2, float s0,
                                                            float s1, float s2) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < npix; i += stride) {
    const uint8_t* p = in + i * 3;
    // same operation order as the oracle: (x - mean) / std, division kept exact
    f32x4e v;
    v[0] = ((float)p[0] - m0) / s0;
    v[1] = ((float)p[1] - m1) / s1;
    v[2] = ((float)p[2] - m2) / s2;
    v[3] = 0.f;
    *reinterpret_cast<f32x4e*>(out + i * 4) = v;
  }
}

// (N,3,H,W) fp32 NCHW (already normalised) -> (N,H,W,4) fp32, pad channel = 0.
__global__ __launch_bounds__(256) void pack_nchw_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int n, size_t hw, int cin) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n * hw;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < total; i += stride) {
    const size_t img = i / hw;
    const size_t px = i - img * hw;
    const float* p = in + img * (size_t)cin * hw + px;
    f32x4e v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cin && c < 4; ++c) v[c] = p[(size_t)c * hw];
    *reinterpret_cast<f32x4e*>(out + i * 4) = v;
  }
}

// MaxPool2d(2,2): in (N,H,W,C) with pixel stride ldi (>= C, lets the pool read the skip half of a
// concat buffer) -> out (N,H/2,W/2,C) dense.  One thread per 4 output channels.
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                         int h, int w, int c, int ldi) {
  const int c4 = c >> 2;
  const int oh = h >> 1, ow = w >> 1;
  const size_t total = (size_t)n * oh * ow * c4;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int cv = (int)(i % c4);
    size_t t = i / c4;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const size_t img = t / oh;
    const float* p = in + ((img * h + (size_t)oy * 2) * w + (size_t)ox * 2) * (size_t)ldi + cv * 4;
    const f32x4e a0 = *reinterpret_cast<const f32x4e*>(p);
    const f32x4e a1 = *reinterpret_cast<const f32x4e*>(p + ldi);
    const f32x4e b0 = *reinterpret_cast<const f32x4e*>(p + (size_t)w * ldi);
    const f32x4e b1 = *reinterpret_cast<const f32x4e*>(p + (size_t)w * ldi + ldi);
    f32x4e m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(b0[e], b1[e]));
    *reinterpret_cast<f32x4e*>(out + i * 4) = m;
  }
}

// 1x1 head: logits[p] = dot(x[p, 0:C], w) + bias; optional sigmoid and threshold outputs.
// LPP lanes cooperate on one pixel (each a float4 slice of the channel vector, 16*LPP bytes
// contiguous per pixel), partial sums combined with wavefront shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void head1x1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      float bias, const float* __restrict__ biasPtr, size_t npix,
                                                      int c, float* __restrict__ logits,
                                                      float* __restrict__ probs, uint8_t* __restrict__ mask,
                                                      float thr) {
  if (biasPtr) bias = *biasPtr;  // training keeps the bias in the device-resident parameter buffer
  const int sub = threadIdx.x % LPP;
  const size_t pixPerBlock = 256 / LPP;
  size_t p = (size_t)blockIdx.x * pixPerBlock + threadIdx.x / LPP;
  const size_t stride = (size_t)gridDim.x * pixPerBlock;
  // loop bound is uniform per wave group of LPP lanes; keep all lanes alive for the shuffles
  const size_t pEnd = (npix + pixPerBlock - 1) / pixPerBlock * pixPerBlock;
  for (; p < pEnd; p += stride) {
    float s = 0.f;
    if (p < npix) {
      const float* x = in + p * (size_t)c;
      for (int k = sub * 4; k < c; k += LPP * 4) {
        const f32x4e xv = *reinterpret_cast<const f32x4e*>(x + k);
        const f32x4e wv = *reinterpret_cast<const f32x4e*>(w + k);
        s = fmaf(xv[0], wv[0], s);
        s = fmaf(xv[1], wv[1], s);
        s = fmaf(xv[2], wv[2], s);
        s = fmaf(xv[3], wv[3], s);
      }
    }
#pragma unroll
    for (int d = LPP >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (sub == 0 && p < npix) {
      const float z = s + bias;
      if (logits) logits[p] = z;
      if (probs) probs[p] = 1.f / (1.f + __expf(-z));
      if (mask) mask[p] = z > thr ? 255 : 0;
    }
  }
}

}  // namespace unet
