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
#include "crop_device.h"

namespace mp {

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
  crop_pixel<C>(img, H, W, x1, y1, bin_w, bin_h, px, py,
                out + (size_t)row * stride_b + (size_t)py * stride_y + (size_t)px * stride_x + c0);
}

// [n_im][C][H][W] -> [n_im][H][W][4] (channel 3 = 0 for RGB): the layout the fused crop role of the rasteriser reads
__global__ void pack_nhwc4_kernel(const float* __restrict__ in, int C, int HW, float4* __restrict__ out) {
  const int im = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const float* p = in + (size_t)im * C * HW + i;
  float4 v;
  v.x = p[0]; v.y = p[HW]; v.z = p[2 * (size_t)HW];
  v.w = C == 4 ? p[3 * (size_t)HW] : 0.f;
  out[(size_t)im * HW + i] = v;
}

// T = float, or _Float16 for the half-precision CNN input of the "fp16 renders" mode (value read, normalised in fp32, rounded back)
template <typename T>
__global__ void normalize_depth_kernel(T* __restrict__ x, int h, int w, int border, int C, int ch,
                                       const float* __restrict__ tCR, int mode) {
  const int row = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= h * w) return;
  const int py = pix / w, px = pix % w;
  const int Wp = w + 2 * border, Hp = h + 2 * border;
  T* p = x + (((size_t)row * Hp + py + border) * Wp + px + border) * C + ch;
  const float zr = tCR[3 * row + 2];
  float d = (float)*p;
  if (mode == 1) d = d / zr;
  else if (mode == 2) d = fminf(fmaxf(d / zr, 0.f), 2.f) - 1.f;
  else if (mode == 3) d = fminf(fmaxf(d - zr, -2.f), 2.f);
  *p = (T)d;
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

static int normalize_depth_impl(void* d_x, bool f16, int b, int h, int w, int border, int C, const int32_t* h_channels, int n_ch,
                                const float* d_tCR, int mode, mp_stream stream) {
  MP_REQUIRE(d_x && d_tCR && (n_ch == 0 || h_channels), "mp_normalize_depth: null pointer");
  MP_REQUIRE(mode >= 0 && mode <= 3, "mp_normalize_depth: unknown mode %d", mode);
  if (mode == 0 || b == 0) return MP_OK;
  dim3 grid(ceil_div((long)h * w, 256), b);
  for (int i = 0; i < n_ch; ++i) {
    if (f16)
      hipLaunchKernelGGL(normalize_depth_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (_Float16*)d_x, h, w, border, C,
                         h_channels[i], d_tCR, mode);
    else
      hipLaunchKernelGGL(normalize_depth_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)d_x, h, w, border, C,
                         h_channels[i], d_tCR, mode);
  }
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_normalize_depth(float* d_x, int b, int h, int w, int border, int C, const int32_t* h_channels, int n_ch,
                                  const float* d_tCR, int mode, mp_stream stream) {
  return normalize_depth_impl(d_x, false, b, h, w, border, C, h_channels, n_ch, d_tCR, mode, stream);
}

extern "C" int mp_normalize_depth_f16(void* d_x_half, int b, int h, int w, int border, int C, const int32_t* h_channels, int n_ch,
                                      const float* d_tCR, int mode, mp_stream stream) {
  return normalize_depth_impl(d_x_half, true, b, h, w, border, C, h_channels, n_ch, d_tCR, mode, stream);
}

extern "C" int mp_pack_observation_nhwc4(const float* d_images, int n_im, int C, int H, int W, float* d_out, mp_stream stream) {
  MP_REQUIRE(d_images && d_out && n_im > 0 && (C == 3 || C == 4) && H > 0 && W > 0, "mp_pack_observation_nhwc4: bad arguments");
  ProfScope prof("pack_observation_nhwc4", 0.0, (double)n_im * H * W * (C + 4) * 4.0, (hipStream_t)stream);
  hipLaunchKernelGGL(pack_nhwc4_kernel, dim3(ceil_div((long)H * W, 256L), n_im), dim3(256), 0, (hipStream_t)stream, d_images, C, H * W,
                     (float4*)d_out);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
