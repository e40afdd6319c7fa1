// pool_fc.hip -- 3x3/s2 max pool and the global-average-pool + fc + head tail of the backbones.
//
// Reference: src/megapose/models/torchvision_resnet.py:216 (maxpool), :311-314 (avgpool, flatten, fc);
//            src/megapose/models/wide_resnet.py:70-73,106; src/megapose/models/pose_rigid.py:326-333
//            (mean over H*W for 4-D backbone outputs, then the Linear heads :122-130), sigmoid :627.
// Roofline: HBM-bound streaming (max pool reads C*4*H*W, writes a quarter of it per row).
#include "common.h"

namespace mp {

// one thread per (output pixel, 4 channels); input is post-ReLU (>= 0) so the zero border == -inf padding
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                           int ib, float* __restrict__ y, int ob, int Ho, int Wo,
                                                           float* __restrict__ y_act, const float* __restrict__ sc,
                                                           const float* __restrict__ sh) {
  const int c4n = C / 4;
  const long total = (long)N * Ho * Wo * c4n;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long t = idx / c4n;
  const int wo = (int)(t % Wo);
  t /= Wo;
  const int ho = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const int Hp = H + 2 * ib, Wp = W + 2 * ib;
  // window rows 2*ho-1 .. 2*ho+1 in logical coords -> +ib in the padded buffer (ib >= 1)
  const float* base = x + (((size_t)n * Hp + (2 * ho - 1 + ib)) * Wp + (2 * wo - 1 + ib)) * C + c4 * 4;
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  bool first = true;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = 2 * ho - 1 + dy, xx = 2 * wo - 1 + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // true -inf padding semantics
      const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)dy * Wp + dx) * C);
      if (first) { m = v; first = false; }
      else { m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w); }
    }
  const int Hop = Ho + 2 * ob, Wop = Wo + 2 * ob;
  const size_t o = (((size_t)n * Hop + ho + ob) * Wop + wo + ob) * C + c4 * 4;
  if (y) *reinterpret_cast<float4*>(y + o) = m;
  if (y_act) {
    const float4 s = *reinterpret_cast<const float4*>(sc + c4 * 4);
    const float4 h = *reinterpret_cast<const float4*>(sh + c4 * 4);
    float4 a;
    a.x = fmaxf(fmaf(m.x, s.x, h.x), 0.f);
    a.y = fmaxf(fmaf(m.y, s.y, h.y), 0.f);
    a.z = fmaxf(fmaf(m.z, s.z, h.z), 0.f);
    a.w = fmaxf(fmaf(m.w, s.w, h.w), 0.f);
    *reinterpret_cast<float4*>(y_act + o) = a;
  }
}

// y_act = relu(x * scale[c] + shift[c]) over the interior of a padded NHWC map: one thread per (pixel, 4 channels).  The pre-activation
// WideResNets need relu(bn1(pooled)) next to the pooled stem map; with the max pool fused into the stem's epilogue (windows that straddle
// tiles are completed by atomics, so no workgroup sees the final maximum) this small pass produces it -- the same fmaf / fmaxf as the
// second output of maxpool3x3s2_kernel, bit for bit.
__global__ __launch_bounds__(256) void bn_relu_kernel(const float* __restrict__ x, int N, int H, int W, int C, int b, float* __restrict__ y_act,
                                                      const float* __restrict__ sc, const float* __restrict__ sh) {
  const int c4n = C / 4;
  const long total = (long)N * H * W * c4n;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long t = idx / c4n;
  const int xx = (int)(t % W);
  t /= W;
  const int yy = (int)(t % H);
  const int n = (int)(t / H);
  const size_t o = (((size_t)n * (H + 2 * b) + yy + b) * (W + 2 * b) + xx + b) * C + c4 * 4;
  const float4 m = *reinterpret_cast<const float4*>(x + o);
  const float4 s = *reinterpret_cast<const float4*>(sc + c4 * 4);
  const float4 h = *reinterpret_cast<const float4*>(sh + c4 * 4);
  float4 a;
  a.x = fmaxf(fmaf(m.x, s.x, h.x), 0.f);
  a.y = fmaxf(fmaf(m.y, s.y, h.y), 0.f);
  a.z = fmaxf(fmaf(m.z, s.z, h.z), 0.f);
  a.w = fmaxf(fmaf(m.w, s.w, h.w), 0.f);
  *reinterpret_cast<float4*>(y_act + o) = a;
}

// one workgroup per batch row: mean over H*W (sequential, row-major like torch), optional fc, heads, sigmoid
__global__ __launch_bounds__(256) void pool_fc_heads_kernel(const float* __restrict__ x, int H, int W, int C, int ib,
                                                            const float* __restrict__ fc_w, const float* __restrict__ fc_b,
                                                            int n_feat, const float* __restrict__ head_w,
                                                            const float* __restrict__ head_b, int n_out,
                                                            float* __restrict__ feat_out, float* __restrict__ out,
                                                            float* __restrict__ sig) {
  extern __shared__ float sm[];
  float* pooled = sm;       // [C]
  float* feat = sm + C;     // [n_feat]
  const int n = blockIdx.x;
  const int Hp = H + 2 * ib, Wp = W + 2 * ib;
  const float inv = 1.0f / (float)(H * W);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int yy = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx) s += x[(((size_t)n * Hp + yy + ib) * Wp + xx + ib) * C + c];
    pooled[c] = s * inv;
  }
  __syncthreads();
  if (fc_w) {
    for (int j = threadIdx.x; j < n_feat; j += blockDim.x) {
      const float* wr = fc_w + (size_t)j * C;
      float s = 0.f;
      for (int c = 0; c < C; ++c) s = fmaf(pooled[c], wr[c], s);
      feat[j] = s + fc_b[j];
    }
  } else {
    for (int j = threadIdx.x; j < n_feat; j += blockDim.x) feat[j] = pooled[j];
  }
  __syncthreads();
  if (feat_out)
    for (int j = threadIdx.x; j < n_feat; j += blockDim.x) feat_out[(size_t)n * n_feat + j] = feat[j];
  // heads: one wave per output
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = wave; o < n_out; o += blockDim.x / 64) {
    const float* wr = head_w + (size_t)o * n_feat;
    float s = 0.f;
    for (int j = lane; j < n_feat; j += 64) s = fmaf(feat[j], wr[j], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) {
      const float v = s + head_b[o];
      out[(size_t)n * n_out + o] = v;
      if (sig) sig[(size_t)n * n_out + o] = 1.0f / (1.0f + expf(-v));
    }
  }
}

}  // namespace mp

using namespace mp;

extern "C" int mp_maxpool3x3s2(const float* d_x, int N, int H, int W, int C, int in_border, float* d_y, int out_border,
                               float* d_y_act, const float* d_sc, const float* d_sh, mp_stream stream) {
  MP_REQUIRE(d_x && (d_y || d_y_act), "mp_maxpool3x3s2: null pointer");
  MP_REQUIRE(C % 4 == 0 && in_border >= 1, "mp_maxpool3x3s2: C %% 4 == 0 and in_border >= 1 required");
  MP_REQUIRE(!d_y_act || (d_sc && d_sh), "mp_maxpool3x3s2: y_act needs scale/shift");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  if (total == 0) return MP_OK;
  ProfScope prof("maxpool3x3s2", 0.0, 4.0 * C * ((double)N * H * W + (double)N * Ho * Wo * (d_y && d_y_act ? 2 : 1)), (hipStream_t)stream);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, d_x, N, H, W, C,
                     in_border, d_y, out_border, Ho, Wo, d_y_act, d_sc, d_sh);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_bn_relu_nhwc(const float* d_x, int N, int H, int W, int C, int border, float* d_y_act, const float* d_sc, const float* d_sh,
                              mp_stream stream) {
  MP_REQUIRE(d_x && d_y_act && d_sc && d_sh, "mp_bn_relu_nhwc: null pointer");
  MP_REQUIRE(C % 4 == 0 && border >= 0, "mp_bn_relu_nhwc: C %% 4 == 0 required");
  const long total = (long)N * H * W * (C / 4);
  if (total == 0) return MP_OK;
  ProfScope prof("bn_relu", 0.0, 8.0 * C * (double)N * H * W, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_relu_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, d_x, N, H, W, C, border, d_y_act, d_sc, d_sh);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_pool_fc_heads(const float* d_x, int N, int H, int W, int C, int in_border, const float* d_fc_w,
                                const float* d_fc_b, int n_feat, const float* d_head_w, const float* d_head_b, int n_out,
                                float* d_feat, float* d_out, float* d_sigmoid, mp_stream stream) {
  MP_REQUIRE(d_x && d_head_w && d_head_b && d_out, "mp_pool_fc_heads: null pointer");
  MP_REQUIRE(d_fc_w ? (d_fc_b != nullptr) : (n_feat == C), "mp_pool_fc_heads: fc bias missing or n_feat != C without fc");
  if (N == 0) return MP_OK;
  const size_t lds = (size_t)(C + n_feat) * sizeof(float);
  ProfScope prof("pool_fc_heads", 2.0 * N * ((d_fc_w ? (double)C * n_feat : 0.0) + (double)n_feat * n_out), 4.0 * N * (double)H * W * C, (hipStream_t)stream);
  hipLaunchKernelGGL(pool_fc_heads_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, d_x, H, W, C, in_border, d_fc_w,
                     d_fc_b, n_feat, d_head_w, d_head_b, n_out, d_feat, d_out, d_sigmoid);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
