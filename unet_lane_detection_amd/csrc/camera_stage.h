// Camera stage around the network on the GPU: what the reference's ROS callback does
// with OpenCV on the CPU (src/unet_ros_node.py:296-311, src/unet.py:33, :70) -
//   warpPerspective(M, (1055,685)) -> [INTER_AREA resize at scale 1 = copy] -> BGR2RGB -> resize((224,224))
// - fused into one kernel that computes only the warped pixels the 224x224 bilinear resize reads (4 of every ~14),
// and the mask's resize back.  Integer arithmetic restated from OpenCV 4.x (see oracle/camera_oracle.py for the
// algorithm and for why parity against cv2 itself is unpinned); bit-exact against that restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {

struct CameraArgs {
  const uint8_t* img;   // (height, step) bytes, 3 channels interleaved
  uint8_t* out;         // (out_h, out_w, 3) RGB
  double minv[9];       // inverse of the perspective matrix (destination -> source)
  double scale_x, scale_y;   // 1 / (out / warp)
  int height, width, step, swap_rb;
  int warp_w, warp_h, out_w, out_h;
};

// cv::resize INTER_LINEAR, 8-bit: source index and the two 2^11 fixed-point coefficients of destination index d
__device__ __forceinline__ void resize_coeff(int d, double scale, int srcSize, int& s, int& c0, int& c1) {
#pragma clang fp contract(off)
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f = f - (float)si;
  if (si < 0) {
    si = 0;
    f = 0.f;
  }
  if (si >= srcSize - 1) {
    si = srcSize - 1;
    f = 0.f;
  }
  s = si;
  c1 = (int)rintf(f * 2048.f);
  c0 = (int)rintf((1.f - f) * 2048.f);
}

// one pixel of cv::warpPerspective (INTER_LINEAR, constant border 0): three channels
__device__ __forceinline__ void warp_pixel(const CameraArgs& a, int x, int y, int (&v)[3]) {
#pragma clang fp contract(off)
  const double xs = (double)x, ys = (double)y;
  double w = a.minv[6] * xs + a.minv[7] * ys + a.minv[8];
  w = w != 0.0 ? 32.0 / w : 0.0;
  double fx = (a.minv[0] * xs + a.minv[1] * ys + a.minv[2]) * w;
  double fy = (a.minv[3] * xs + a.minv[4] * ys + a.minv[5]) * w;
  fx = fmin(fmax(fx, -2147483648.0), 2147483647.0);
  fy = fmin(fmax(fy, -2147483648.0), 2147483647.0);
  const long long X = (long long)rint(fx), Y = (long long)rint(fy);
  const long long sx = X >> 5, sy = Y >> 5;
  const int fa = (int)(X & 31), fb = (int)(Y & 31);
  const int wgt[4] = {(32 - fb) * (32 - fa) * 32, (32 - fb) * fa * 32, fb * (32 - fa) * 32, fb * fa * 32};
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long long yy = sy + (t >> 1), xx = sx + (t & 1);
    if (yy >= 0 && yy < a.height && xx >= 0 && xx < a.width) {
      const uint8_t* p = a.img + (size_t)yy * a.step + (size_t)xx * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wgt[t] * (int)p[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = (acc[c] + (1 << 14)) >> 15;
}

__global__ __launch_bounds__(256) void ipm_prestage_kernel(const CameraArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.out_w * a.out_h) return;
  const int oy = i / a.out_w, ox = i - oy * a.out_w;
  int res[3];
  if (a.out_w == a.warp_w && a.out_h == a.warp_h) {
    warp_pixel(a, ox, oy, res);   // cv::resize to the same size is a copy
  } else {
    int sx, a0, a1, sy, b0, b1;
    resize_coeff(ox, a.scale_x, a.warp_w, sx, a0, a1);
    resize_coeff(oy, a.scale_y, a.warp_h, sy, b0, b1);
    const int sx1 = sx + 1 < a.warp_w ? sx + 1 : a.warp_w - 1;
    const int sy1 = sy + 1 < a.warp_h ? sy + 1 : a.warp_h - 1;
    int p00[3], p01[3], p10[3], p11[3];
    warp_pixel(a, sx, sy, p00);
    warp_pixel(a, sx1, sy, p01);
    warp_pixel(a, sx, sy1, p10);
    warp_pixel(a, sx1, sy1, p11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d0 = p00[c] * a0 + p01[c] * a1, d1 = p10[c] * a0 + p11[c] * a1;
      res[c] = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
    }
  }
  uint8_t* o = a.out + (size_t)i * 3;
  o[0] = (uint8_t)(a.swap_rb ? res[2] : res[0]);
  o[1] = (uint8_t)res[1];
  o[2] = (uint8_t)(a.swap_rb ? res[0] : res[2]);
}

// cv::resize INTER_LINEAR of an 8-bit image with `cn` interleaved channels (the mask's way back: src/unet.py:70)
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int height, int width, int cn,
                                                        double scale_x, double scale_y, int out_w, int out_h,
                                                        uint8_t* __restrict__ dst) {
  const size_t total = (size_t)out_w * out_h * cn;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % cn);
    const size_t p = i / cn;
    const int ox = (int)(p % out_w), oy = (int)(p / out_w);
    if (out_w == width && out_h == height) {
      dst[i] = src[i];
      continue;
    }
    int sx, a0, a1, sy, b0, b1;
    resize_coeff(ox, scale_x, width, sx, a0, a1);
    resize_coeff(oy, scale_y, height, sy, b0, b1);
    const int sx1 = sx + 1 < width ? sx + 1 : width - 1;
    const int sy1 = sy + 1 < height ? sy + 1 : height - 1;
    const uint8_t* r0 = src + (size_t)sy * width * cn;
    const uint8_t* r1 = src + (size_t)sy1 * width * cn;
    const int d0 = (int)r0[sx * cn + c] * a0 + (int)r0[sx1 * cn + c] * a1;
    const int d1 = (int)r1[sx * cn + c] * a0 + (int)r1[sx1 * cn + c] * a1;
    dst[i] = (uint8_t)((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2);
  }
}

}  // namespace unet
