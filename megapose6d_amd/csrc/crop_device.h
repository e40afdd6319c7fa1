// crop_device.h -- per-pixel roi_align sample shared by the stand-alone crop kernel (crop.hip) and the fused crop role of the
// rasteriser's band kernel (raster.hip).  Reference: torchvision.ops.roi_align(sampling_ratio=4, aligned=False) as used by
// src/megapose/lib3d/cropping.py:113-144, incl. the RGBD validity rule (:131-142).
#pragma once
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

// source pixel o = y * W + x of an image stored [C][H][W] (NHWC4 = false) or [H][W][4] (true: one 16-byte load per tap instead of C
// 4-byte loads -- the fused crop role is bound by the number of vector-memory instructions)
template <int C, bool NHWC4>
__device__ __forceinline__ void crop_fetch(const float* __restrict__ img, size_t plane, size_t o, float (&v)[C]) {
  if constexpr (NHWC4) {
    const float4 t = reinterpret_cast<const float4*>(img)[o];
    v[0] = t.x; v[1] = t.y; v[2] = t.z;
    if constexpr (C == 4) v[3] = t.w;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = img[c * plane + o];
  }
}

// one output pixel (px, py) of the roi [x1, y1, x1 + out_w * bin_w, y1 + out_h * bin_h] of `img`; writes C floats to o
template <int C, bool NHWC4 = false>
__device__ __forceinline__ void crop_pixel(const float* __restrict__ img, int H, int W, float x1, float y1, float bin_w, float bin_h,
                                           int px, int py, float* __restrict__ o) {
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
        float vv[C];
        crop_fetch<C, NHWC4>(img, plane, o, vv);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v = vv[c];
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
        float q1[C], q2[C], q3[C], q4[C];
        crop_fetch<C, NHWC4>(img, plane, o1, q1);
        crop_fetch<C, NHWC4>(img, plane, o2, q2);
        crop_fetch<C, NHWC4>(img, plane, o3, q3);
        crop_fetch<C, NHWC4>(img, plane, o4, q4);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v1 = q1[c], v2 = q2[c], v3 = q3[c], v4 = q4[c];
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
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v = acc[c] / 16.0f;
    if (C == 4 && c == 3 && (acc_valid / 16.0f) < 0.99f) v = 0.f;  // cropping.py:140-142
    o[c] = v;
  }
}

}  // namespace mp
