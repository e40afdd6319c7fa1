// crop.hip -- roi_align crop of the observation straight into the CNN input tensor + depth normalisation.
//
// Reference: src/megapose/lib3d/cropping.py:113-144 crop_images -> torchvision.ops.roi_align(
//   images, boxes, output_size=(240,320), sampling_ratio=4)  (torchvision 0.12.0, aligned=False,
//   spatial_scale=1; algorithm = torchvision/csrc/ops/cpu/roi_align_kernel.cpp) including the RGBD
//   validity rule (:131-142): depth crop is zeroed where roi_align(depth>0) < 0.99.
// The reference first gathers one full frame per row (inference/pose_estimator.py:389, 3.7 MB/row);
// here every row reads the single observation frame selected by batch_im_id.
// Also: models/pose_rigid.py:466-496 normalize_depth.
// Roofline: HBM/L2-bound gather; algorithmic bytes/row = C*4*out_h*out_w written + covered source window read.
#include "common.h"

namespace mp {

struct Tap {
  int lo, hi;
  float l, h;
  bool valid;
};

// pre_calc_for_bilinear_interpolate, one axis
__device__ __forceinline__ Tap make_tap(float c, int size) {
  Tap t;
  t.valid = !(c < -1.0f || c > (float)size);
  if (c <= 0.f) c = 0.f;
  int lo = (int)c;
  int hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    c = (float)lo;
  } else {
    hi = lo + 1;
  }
  t.lo = lo;
  t.hi = hi;
  t.l = c - (float)lo;
  t.h = 1.0f - t.l;
  return t;
}

template <int C>
__global__ __launch_bounds__(256) void crop_roi_align_kernel(const float* __restrict__ images, int H, int W,
                                                             const int32_t* __restrict__ im_ids,
                                                             const float* __restrict__ boxes, int out_h, int out_w,
                                                             float* __restrict__ out, long long stride_b, long long stride_y,
                                                             long long stride_x, int c0) {
  const int row = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= out_h * out_w) return;
  const int py = pix / out_w, px = pix % out_w;
  const float* bx = boxes + (size_t)row * 4;
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  const float roi_w = fmaxf(x2 - x1, 1.0f), roi_h = fmaxf(y2 - y1, 1.0f);
  const float bin_h = roi_h / (float)out_h, bin_w = roi_w / (float)out_w;
  const float* img = images + (size_t)im_ids[row] * C * H * W;
  const size_t plane = (size_t)H * W;
  float acc[C];
  float acc_valid = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  // The 4x4 sample grid is a tensor product and bilinear weights are separable, so
  //   sum_{iy,ix} bilinear(y_iy, x_ix) = sum_r sum_c wy[r] * wx[c] * img[r0 + r][c0 + c]
  // with per-axis weights accumulated over the 4 samples (invalid samples weigh 0, edge clamping is per axis).
  // For crop scales up to ~2.6 source px per output px the patch is <= 4x4: 16 loads per channel instead of 64.
  Tap ty[4], tx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ty[i] = make_tap(y1 + (float)py * bin_h + ((float)i + 0.5f) * bin_h / 4.0f, H);
    tx[i] = make_tap(x1 + (float)px * bin_w + ((float)i + 0.5f) * bin_w / 4.0f, W);
  }
  const int r0 = ty[0].lo, c0p = tx[0].lo;  // sample coordinates are monotone, so the first tap has the smallest index
  constexpr int P = 4;
  if (ty[3].hi - r0 < P && tx[3].hi - c0p < P) {
    float wy[P], wx[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ty[i].valid) a += (ty[i].lo - r0 == r ? ty[i].h : 0.f) + (ty[i].hi - r0 == r ? ty[i].l : 0.f);
        if (tx[i].valid) b += (tx[i].lo - c0p == r ? tx[i].h : 0.f) + (tx[i].hi - c0p == r ? tx[i].l : 0.f);
      }
      wy[r] = a;
      wx[r] = b;
    }
#pragma unroll
    for (int r = 0; r < P; ++r) {
      if (wy[r] == 0.f) continue;
      const int rr = min(r0 + r, H - 1);
#pragma unroll
      for (int cc = 0; cc < P; ++cc) {
        const float wgt = wy[r] * wx[cc];
        const size_t o = (size_t)rr * W + min(c0p + cc, W - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v = img[c * plane + o];
          acc[c] = fmaf(wgt, v, acc[c]);
          if (C == 4 && c == 3) acc_valid = fmaf(wgt, v > 0.f ? 1.f : 0.f, acc_valid);
        }
      }
    }
  } else {
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        if (!(ty[iy].valid && tx[ix].valid)) continue;  // contributes 0
        const float w1 = ty[iy].h * tx[ix].h, w2 = ty[iy].h * tx[ix].l, w3 = ty[iy].l * tx[ix].h, w4 = ty[iy].l * tx[ix].l;
        const size_t o1 = (size_t)ty[iy].lo * W + tx[ix].lo, o2 = (size_t)ty[iy].lo * W + tx[ix].hi;
        const size_t o3 = (size_t)ty[iy].hi * W + tx[ix].lo, o4 = (size_t)ty[iy].hi * W + tx[ix].hi;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* p = img + c * plane;
          const float v1 = p[o1], v2 = p[o2], v3 = p[o3], v4 = p[o4];
          acc[c] += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
          if (C == 4 && c == 3) {
            const float m1 = v1 > 0.f ? 1.f : 0.f, m2 = v2 > 0.f ? 1.f : 0.f, m3 = v3 > 0.f ? 1.f : 0.f,
                        m4 = v4 > 0.f ? 1.f : 0.f;
            acc_valid += w1 * m1 + w2 * m2 + w3 * m3 + w4 * m4;
          }
        }
      }
    }
  }
  float* o = out + (size_t)row * stride_b + (size_t)py * stride_y + (size_t)px * stride_x + c0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v = acc[c] / 16.0f;
    if (C == 4 && c == 3 && (acc_valid / 16.0f) < 0.99f) v = 0.f;  // cropping.py:140-142
    o[c] = v;
  }
}

__global__ void normalize_depth_kernel(float* __restrict__ x, int h, int w, int border, int C, int ch,
                                       const float* __restrict__ tCR, int mode) {
  const int row = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= h * w) return;
  const int py = pix / w, px = pix % w;
  const int Wp = w + 2 * border, Hp = h + 2 * border;
  float* p = x + (((size_t)row * Hp + py + border) * Wp + px + border) * C + ch;
  const float zr = tCR[3 * row + 2];
  float d = *p;
  if (mode == 1) d = d / zr;
  else if (mode == 2) d = fminf(fmaxf(d / zr, 0.f), 2.f) - 1.f;
  else if (mode == 3) d = fminf(fmaxf(d - zr, -2.f), 2.f);
  *p = d;
}

}  // namespace mp

using namespace mp;

extern "C" int mp_crop_roi_align(const float* d_images, int n_im, int C, int H, int W, const int32_t* d_im_ids,
                                 const float* d_boxes, int b, int out_h, int out_w, float* d_out, int64_t stride_b,
                                 int64_t stride_y, int64_t stride_x, int c0, mp_stream stream) {
  MP_REQUIRE(d_images && d_im_ids && d_boxes && d_out, "mp_crop_roi_align: null pointer");
  MP_REQUIRE(C == 3 || C == 4, "mp_crop_roi_align: C must be 3 or 4 (cropping.py:119)");
  MP_REQUIRE(n_im > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && b >= 0 && b <= 65535, "mp_crop_roi_align: bad size");
  if (b == 0) return MP_OK;
  dim3 grid(ceil_div((long)out_h * out_w, 256), b);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof("crop_roi_align", 0.0, (double)b * C * 4.0 * out_h * out_w * 2.0, s);  // write + (<=) equal-sized source window read
  if (C == 3)
    hipLaunchKernelGGL(crop_roi_align_kernel<3>, grid, dim3(256), 0, s, d_images, H, W, d_im_ids, d_boxes, out_h, out_w, d_out,
                       (long long)stride_b, (long long)stride_y, (long long)stride_x, c0);
  else
    hipLaunchKernelGGL(crop_roi_align_kernel<4>, grid, dim3(256), 0, s, d_images, H, W, d_im_ids, d_boxes, out_h, out_w, d_out,
                       (long long)stride_b, (long long)stride_y, (long long)stride_x, c0);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_normalize_depth(float* d_x, int b, int h, int w, int border, int C, const int32_t* h_channels, int n_ch,
                                  const float* d_tCR, int mode, mp_stream stream) {
  MP_REQUIRE(d_x && d_tCR && (n_ch == 0 || h_channels), "mp_normalize_depth: null pointer");
  MP_REQUIRE(mode >= 0 && mode <= 3, "mp_normalize_depth: unknown mode %d", mode);
  if (mode == 0 || b == 0) return MP_OK;
  dim3 grid(ceil_div((long)h * w, 256), b);
  for (int i = 0; i < n_ch; ++i)
    hipLaunchKernelGGL(normalize_depth_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_x, h, w, border, C, h_channels[i],
                       d_tCR, mode);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
