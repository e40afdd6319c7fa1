// detector.hip -- the 2D detector network (Mask R-CNN, ResNet-50 + FPN) as one native, device-resident inference graph.
//
// Replaces `self.model([image_n for image_n in images])` of the reference's Detector
//   (/root/reference/src/megapose/inference/detector.py:92; model = DetectorMaskRCNN, src/megapose/models/mask_rcnn.py:23-46 =
//    torchvision.models.detection.MaskRCNN(resnet_fpn_backbone("resnet50"), n_classes, AnchorGenerator(anchor_sizes, (0.5, 1, 2)),
//    min_size, max_size), created by training/detector_models_cfg.py:31-38).
// torchvision (0.12.0) is third-party code that is not under /root/reference; the inference algorithm restated here is the
// published one, file by file as listed at the top of oracle/mask_rcnn.py -- the independent CPU restatement this file is tested
// against ("parity unpinned" vs torchvision itself).
//
// Design (MI355X-first, not a port of torchvision's op-by-op graph):
//   * every feature map is padded NHWC fp32 in ONE caller-provided workspace; all 100+ convolutions (ResNet-50 bottlenecks with the
//     FrozenBatchNorm folded in, FPN, RPN head, mask head) and the box head's FC layers (as 1x1 convolutions over [R,1,1,12544]) run
//     on the library's fp32-MFMA implicit-GEMM kernel (conv.hip) with fused bias / residual / ReLU epilogues;
//   * the RPN's two 1x1 heads are one convolution (3 + 12 -> 16 channels), the predictor's two Linear layers one (C + 4C channels),
//     the mask predictor's 2x2 stride-2 transposed convolution is a 1x1 convolution onto 4 x 256 sub-pixel channels;
//   * selection (top-k, sort) is a stable rank sort on the device -- rank(i) = #{j: key_j > key_i or (key_j == key_i and j < i)} --,
//     NMS a sequential-greedy sweep with a parallel inner loop, one workgroup per (image, level | class) segment: deterministic,
//     no host round trip, no library sort;
//   * no synchronisation: detections come back as fixed-size [n, D] arrays + counts.
// None of this is on the pose hot path (one call per frame set); it is bounded by the ResNet-50 convolutions (MFMA).
#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace mp {

constexpr int DET_LEVELS = 5;       // P2..P6
constexpr int DET_A = 3;            // anchors per location
constexpr int DET_FPN_C = 256;
constexpr int DET_MAX_SEG = 1024;   // NMS segment capacity (LDS)
constexpr float DET_XFORM_CLIP = 4.135166556742356f;   // log(1000 / 16)

// ---------------------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------------------
// GeneralizedRCNNTransform: (x - mean) / std, bilinear resize (align_corners = False, scale = in / out), zero pad; NCHW -> padded NHWC4
__global__ __launch_bounds__(256) void det_preprocess_kernel(const float* __restrict__ img, int H, int W, int hr, int wr, float sy, float sx,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out, int Hp, int Wp, int border) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hr * wr) return;
  const int y = i / wr, x = i - y * wr;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  float v[3];
  const float* base = img + (size_t)n * 3 * H * W;
  if (hr == H && wr == W) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (base[((size_t)c * H + y) * W + x] - mean[c]) / sd[c];
  } else {
    float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = base + (size_t)c * H * W;
      const float a = (p[(size_t)y0 * W + x0] - mean[c]) / sd[c], b = (p[(size_t)y0 * W + x1] - mean[c]) / sd[c];
      const float cc = (p[(size_t)y1 * W + x0] - mean[c]) / sd[c], d = (p[(size_t)y1 * W + x1] - mean[c]) / sd[c];
      v[c] = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
    }
  }
  const int Wq = Wp + 2 * border, Hq = Hp + 2 * border;
  *reinterpret_cast<float4*>(out + (((size_t)n * Hq + y + border) * Wq + x + border) * 4) = make_float4(v[0], v[1], v[2], 0.f);
}

// nearest-neighbour resize of a padded-NHWC map (FPN top-down path: F.interpolate(size=..., mode="nearest"))
__global__ __launch_bounds__(256) void det_resize_nearest_kernel(const float* __restrict__ src, int hs, int ws, float* __restrict__ dst, int hd,
                                                                 int wd, int C, int N, int sb, int db, int step_y, int step_x) {
  // step > 0: dst(y, x) = src(y * step, x * step) (LastLevelMaxPool = max_pool2d(kernel 1, stride 2)); step == 0: src index = floor(dst * in / out)
  const int c4n = C / 4;
  const long total = (long)N * hd * wd * c4n;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long t = idx / c4n;
  const int x = (int)(t % wd);
  t /= wd;
  const int y = (int)(t % hd);
  const int n = (int)(t / hd);
  const int ys = step_y > 0 ? y * step_y : min((int)(((long)y * hs) / hd), hs - 1);
  const int xs = step_x > 0 ? x * step_x : min((int)(((long)x * ws) / wd), ws - 1);
  const float4 v = *reinterpret_cast<const float4*>(src + (((size_t)n * (hs + 2 * sb) + ys + sb) * (ws + 2 * sb) + xs + sb) * C + c4 * 4);
  *reinterpret_cast<float4*>(dst + (((size_t)n * (hd + 2 * db) + y + db) * (wd + 2 * db) + x + db) * C + c4 * 4) = v;
}

struct RpnLevels {
  const float* head[DET_LEVELS];   // [n, gh, gw, 16]: 3 objectness logits, 12 deltas (a * 4 + k), 1 pad
  int gh[DET_LEVELS], gw[DET_LEVELS], off[DET_LEVELS + 1];   // off: anchor offset of the level inside one image's key row
  int stride_y[DET_LEVELS], stride_x[DET_LEVELS];
  float base[DET_LEVELS][DET_A][4];   // base anchors (rounded), anchor_utils.py generate_anchors
};

// objectness logits of every anchor, (level, y, x, a) order = torchvision's concat_box_prediction_layers order
__global__ __launch_bounds__(256) void det_rpn_keys_kernel(RpnLevels L, int n_images, float* __restrict__ keys) {
  const int total = L.off[DET_LEVELS];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_images * total) return;
  const int n = (int)(idx / total), i = (int)(idx % total);
  int l = 0;
#pragma unroll
  for (int k = 1; k < DET_LEVELS; ++k) l += i >= L.off[k];
  const int j = i - L.off[l], pix = j / DET_A, a = j - pix * DET_A;
  keys[idx] = L.head[l][((size_t)n * L.gh[l] * L.gw[l] + pix) * 16 + a];
}

// segment tables of the sorts, built on the device (no H2D copy whose source could die before the copy runs)
__global__ void det_rpn_seg_offsets_kernel(RpnLevels L, int n_images, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = L.off[DET_LEVELS];
  if (i < n_images * DET_LEVELS) out[i] = (i / DET_LEVELS) * total + L.off[i % DET_LEVELS];
  if (i == n_images * DET_LEVELS) out[i] = n_images * total;
}
__global__ void det_seg_offsets_kernel(int n_seg, int stride, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_seg) out[i] = i * stride;
}

// Stable rank sort: element i of a segment goes to position #{j: key_j > key_i or (key_j == key_i and j < i)}; only the first K
// positions are stored, -inf keys are "absent".  grid = (ceil(max_len / 256), n_segments).  out_cnt must be zero on entry.
__global__ __launch_bounds__(256) void det_rank_topk_kernel(const float* __restrict__ keys, const int* __restrict__ seg_off, int K,
                                                            int* __restrict__ out_idx, int* __restrict__ out_cnt) {
  __shared__ float tile[1024];
  const int seg = blockIdx.y;
  const int beg = seg_off[seg], len = seg_off[seg + 1] - beg;
  if ((int)(blockIdx.x * 256) >= len) return;   // whole workgroup beyond the segment (uniform)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < len;
  const float mine = live ? keys[beg + i] : 0.f;
  int rank = 0;
  for (int t0 = 0; t0 < len; t0 += 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = t0 + k * 256 + threadIdx.x;
      tile[k * 256 + threadIdx.x] = j < len ? keys[beg + j] : -INFINITY;
    }
    __syncthreads();
    const int m = min(1024, len - t0);
    for (int j = 0; j < m; ++j) {
      const float kj = tile[j];
      rank += (kj > mine || (kj == mine && t0 + j < i)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (live && mine > -INFINITY) {
    atomicAdd(&out_cnt[seg], 1);
    if (rank < K) out_idx[(size_t)seg * K + rank] = i;
  }
}

// BoxCoder.decode_single for one box (models/detection/_utils.py), then clip_boxes_to_image
__device__ __forceinline__ float4 det_decode(float4 box, float dx, float dy, float dw, float dh, float wx, float wy, float ww, float wh,
                                             float clip_h, float clip_w) {
  const float width = box.z - box.x, height = box.w - box.y;
  const float ctr_x = box.x + 0.5f * width, ctr_y = box.y + 0.5f * height;
  dx = dx / wx; dy = dy / wy;
  dw = fminf(dw / ww, DET_XFORM_CLIP); dh = fminf(dh / wh, DET_XFORM_CLIP);
  const float pcx = dx * width + ctr_x, pcy = dy * height + ctr_y;
  const float pw = expf(dw) * width, ph = expf(dh) * height;
  float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
  o.x = fminf(fmaxf(o.x, 0.f), clip_w); o.z = fminf(fmaxf(o.z, 0.f), clip_w);
  o.y = fminf(fmaxf(o.y, 0.f), clip_h); o.w = fminf(fmaxf(o.w, 0.f), clip_h);
  return o;
}

// the pre-NMS top-k anchors of every (image, level): decode, clip, sigmoid, small-box / score filter (rpn.py filter_proposals)
__global__ __launch_bounds__(256) void det_rpn_gather_kernel(RpnLevels L, int n_images, const float* __restrict__ keys, const int* __restrict__ idx,
                                                             const int* __restrict__ cnt, int K, float clip_h, float clip_w, float min_size,
                                                             float score_thresh, float4* __restrict__ cand_box, float* __restrict__ cand_score,
                                                             int* __restrict__ cand_keep) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_images * DET_LEVELS * K) return;
  const int r = (int)(t % K), seg = (int)(t / K), l = seg % DET_LEVELS, n = seg / DET_LEVELS;
  const int len = L.off[l + 1] - L.off[l];
  const int c = min(min(cnt[seg], K), len);
  float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
  float score = -INFINITY;
  int keep = 0;
  if (r < c) {
    const int j = idx[(size_t)seg * K + r];
    const int pix = j / DET_A, a = j - pix * DET_A;
    const int y = pix / L.gw[l], x = pix - y * L.gw[l];
    const float shx = (float)(x * L.stride_x[l]), shy = (float)(y * L.stride_y[l]);
    const float4 anchor = make_float4(shx + L.base[l][a][0], shy + L.base[l][a][1], shx + L.base[l][a][2], shy + L.base[l][a][3]);
    const float* h = L.head[l] + ((size_t)n * L.gh[l] * L.gw[l] + pix) * 16;
    box = det_decode(anchor, h[3 + a * 4], h[4 + a * 4], h[5 + a * 4], h[6 + a * 4], 1.f, 1.f, 1.f, 1.f, clip_h, clip_w);
    const float logit = keys[(size_t)n * L.off[DET_LEVELS] + L.off[l] + j];
    score = 1.f / (1.f + expf(-logit));
    keep = (box.z - box.x >= min_size && box.w - box.y >= min_size && score >= score_thresh) ? 1 : 0;
  }
  cand_box[t] = box;
  cand_score[t] = score;
  cand_keep[t] = keep;
}

// greedy NMS of one score-sorted segment per workgroup (torchvision/csrc/ops/cpu/nms_kernel.cpp semantics: suppress IoU > thr)
__global__ __launch_bounds__(256) void det_nms_kernel(const float4* __restrict__ boxes, int* __restrict__ keep, const int* __restrict__ cnt, int stride,
                                                      float thr) {
  __shared__ float4 sb[DET_MAX_SEG];
  __shared__ int sk[DET_MAX_SEG];
  const int seg = blockIdx.x;
  const int n = min(cnt ? cnt[seg] : stride, stride);
  for (int i = threadIdx.x; i < n; i += 256) {
    sb[i] = boxes[(size_t)seg * stride + i];
    sk[i] = keep[(size_t)seg * stride + i];
  }
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    if (sk[i]) {   // (uniform: every thread reads the same word)
      const float4 bi = sb[i];
      const float ai = (bi.z - bi.x) * (bi.w - bi.y);
      for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
        if (!sk[j]) continue;
        const float4 bj = sb[j];
        const float w = fmaxf(0.f, fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x)), h = fmaxf(0.f, fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y));
        const float inter = w * h;
        const float ovr = inter / (ai + (bj.z - bj.x) * (bj.w - bj.y) - inter);
        if (ovr > thr) sk[j] = 0;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += 256) keep[(size_t)seg * stride + i] = sk[i];
}

// keys[i] = keep[i] ? score[i] : -inf
__global__ __launch_bounds__(256) void det_masked_keys_kernel(const float* __restrict__ score, const int* __restrict__ keep, long n, float* __restrict__ keys) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = keep[i] ? score[i] : -INFINITY;
}

// proposals[n][r] = cand_box[n][idx[n][r]] for r < min(cnt[n], K), zero boxes beyond; counts clamped to K
__global__ __launch_bounds__(256) void det_gather_boxes_kernel(const float4* __restrict__ cand_box, const float* __restrict__ cand_score, int per_image,
                                                               const int* __restrict__ idx, int* __restrict__ cnt, int K, int n_images,
                                                               float4* __restrict__ out_box, float* __restrict__ out_score) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_images * K) return;
  const int n = (int)(t / K), r = (int)(t % K);
  const int c = min(cnt[n], K);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  float s = 0.f;
  if (r < c) {
    const int j = idx[t];
    b = cand_box[(size_t)n * per_image + j];
    s = cand_score[(size_t)n * per_image + j];
  }
  out_box[t] = b;
  if (out_score) out_score[t] = s;
}
__global__ void det_clamp_counts_kernel(int* __restrict__ cnt, int n, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = min(cnt[i], K);
}

struct PyramidRef {
  const float* feat[4];   // P2..P5, padded NHWC (border 1), 256 channels
  int h[4], w[4];
};

// MultiScaleRoIAlign(["0","1","2","3"], P, sampling_ratio = 2) (ops/poolers.py LevelMapper + ops/roi_align, aligned = False); one
// thread per (roi, bin, 4 channels); output [roi][P][P][256] with border `ob` (row stride (P + 2 ob)^2 * 256)
__global__ __launch_bounds__(256) void det_roi_align_kernel(PyramidRef py, const float4* __restrict__ rois, const int* __restrict__ cnt, int R, int n_images,
                                                            int P, int ob, float* __restrict__ out) {
  constexpr int C = DET_FPN_C, C4 = C / 4;
  const long total = (long)n_images * R * P * P * C4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  long t = idx / C4;
  const int pw = (int)(t % P);
  t /= P;
  const int ph = (int)(t % P);
  t /= P;
  const int r = (int)(t % R), n = (int)(t / R);
  const int Pq = P + 2 * ob;
  float* o = out + ((((size_t)n * R + r) * Pq + ph + ob) * Pq + pw + ob) * C + c4 * 4;
  if (r >= min(cnt[n], R)) {
    *reinterpret_cast<float4*>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4 b = rois[(size_t)n * R + r];
  // LevelMapper: k = floor(4 + log2(sqrt(area) / 224) + 1e-6) clamped to [2, 5]
  const float s = sqrtf((b.z - b.x) * (b.w - b.y));
  float lv = floorf(4.f + log2f(s / 224.f) + 1e-6f);
  lv = fminf(fmaxf(lv, 2.f), 5.f);
  const int l = (int)lv - 2;
  const float scale = 1.f / (float)(4 << l);
  const int H = py.h[l], W = py.w[l];
  const float* f = py.feat[l] + (size_t)n * (H + 2) * (W + 2) * C + c4 * 4;
  const float x1 = b.x * scale, y1 = b.y * scale, x2 = b.z * scale, y2 = b.w * scale;
  const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int iy = 0; iy < 2; ++iy) {
    float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / 2.f;
#pragma unroll
    for (int ix = 0; ix < 2; ++ix) {
      float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / 2.f;
      float yy = y;
      if (yy < -1.f || yy > (float)H || x < -1.f || x > (float)W) continue;
      if (yy <= 0.f) yy = 0.f;
      if (x <= 0.f) x = 0.f;
      int yl = (int)yy, xl = (int)x, yh, xh;
      if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
      if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
      const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
      const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
      const float4 v1 = *reinterpret_cast<const float4*>(f + ((size_t)(yl + 1) * (W + 2) + xl + 1) * C);
      const float4 v2 = *reinterpret_cast<const float4*>(f + ((size_t)(yl + 1) * (W + 2) + xh + 1) * C);
      const float4 v3 = *reinterpret_cast<const float4*>(f + ((size_t)(yh + 1) * (W + 2) + xl + 1) * C);
      const float4 v4 = *reinterpret_cast<const float4*>(f + ((size_t)(yh + 1) * (W + 2) + xh + 1) * C);
      acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
      acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
      acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
      acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
    }
  }
  *reinterpret_cast<float4*>(o) = make_float4(acc.x / 4.f, acc.y / 4.f, acc.z / 4.f, acc.w / 4.f);
}

// roi_heads.py postprocess_detections up to the NMS: softmax, per-class decode (weights 10, 10, 5, 5), clip, score / size filter.
// pred row = [C logits | 4C deltas (c * 4 + k)]; candidates are written class-major: seg = n * (C - 1) + (c - 1), element r
__global__ __launch_bounds__(256) void det_box_post_kernel(const float* __restrict__ pred, int row_stride, int C, const float4* __restrict__ props,
                                                           const int* __restrict__ prop_cnt, int R, int n_images, float clip_h, float clip_w,
                                                           float score_thresh, float min_size, float4* __restrict__ cand_box,
                                                           float* __restrict__ cand_key) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_images * R) return;
  const int n = (int)(t / R), r = (int)(t % R);
  const bool live = r < min(prop_cnt[n], R);
  const float* row = pred + (size_t)t * row_stride;
  float m = -INFINITY, sum = 0.f;
  if (live) {
    for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
    for (int c = 0; c < C; ++c) sum += expf(row[c] - m);
  }
  const float4 p = props[t];
  for (int c = 1; c < C; ++c) {
    const size_t o = ((size_t)n * (C - 1) + (c - 1)) * R + r;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float key = -INFINITY;
    if (live) {
      const float sc = expf(row[c] - m) / sum;
      const float* d = row + C + c * 4;
      b = det_decode(p, d[0], d[1], d[2], d[3], 10.f, 10.f, 5.f, 5.f, clip_h, clip_w);
      if (sc > score_thresh && b.z - b.x >= min_size && b.w - b.y >= min_size) key = sc;
    }
    cand_box[o] = b;
    cand_key[o] = key;
  }
}

// sorted copies of a segment's candidates: out[seg][r] = in[seg][idx[seg][r]] for r < min(cnt, K); keep = 1 there, 0 beyond
__global__ __launch_bounds__(256) void det_sorted_gather_kernel(const float4* __restrict__ box, const float* __restrict__ key, const int* __restrict__ idx,
                                                                const int* __restrict__ cnt, int K, long n_total, float4* __restrict__ sbox,
                                                                float* __restrict__ skey, int* __restrict__ keep) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_total) return;
  const int seg = (int)(t / K), r = (int)(t % K);
  const bool live = r < min(cnt[seg], K);
  const int j = live ? idx[t] : 0;
  sbox[t] = live ? box[(size_t)seg * K + j] : make_float4(0.f, 0.f, 0.f, 0.f);
  skey[t] = live ? key[(size_t)seg * K + j] : -INFINITY;
  keep[t] = live ? 1 : 0;
}

// final outputs of one image: the D best surviving candidates over all classes; boxes mapped back to the original frame
__global__ __launch_bounds__(256) void det_final_kernel(const float4* __restrict__ sbox, const float* __restrict__ skey, const int* __restrict__ idx,
                                                        const int* __restrict__ cnt, int D, int per_image, int R, int n_images, float ratio_h,
                                                        float ratio_w, float4* __restrict__ out_box, float* __restrict__ out_score,
                                                        int* __restrict__ out_label, int* __restrict__ out_cnt, float4* __restrict__ resized_box) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_images * D) return;
  const int n = (int)(t / D), d = (int)(t % D);
  const int c = min(cnt[n], D);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f), br = b;
  float s = 0.f;
  int lab = 0;
  if (d < c) {
    const int j = idx[t];
    br = sbox[(size_t)n * per_image + j];
    s = skey[(size_t)n * per_image + j];
    lab = j / R + 1;
    b = make_float4(br.x * ratio_w, br.y * ratio_h, br.z * ratio_w, br.w * ratio_h);   // transform.py resize_boxes
  }
  out_box[t] = b;
  out_score[t] = s;
  out_label[t] = lab;
  resized_box[t] = br;
  if (d == 0) out_cnt[n] = c;
}

// roi_heads.py maskrcnn_inference (sigmoid, class selection) + paste_masks_in_image: one thread per pixel of [n, D, H, W]
__global__ __launch_bounds__(256) void det_mask_paste_kernel(const float* __restrict__ logits /*[n*D*14*14*4][Cs]*/, int Cs, const float4* __restrict__ boxes,
                                                             const int* __restrict__ labels, const int* __restrict__ cnt, int D, int H, int W,
                                                             float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int det = blockIdx.y;   // n * D + d
  if (t >= (long)H * W) return;
  const int n = det / D, d = det - n * D;
  float* o = out + (size_t)det * H * W + t;
  if (d >= min(cnt[n], D)) { *o = 0.f; return; }
  const int Y = (int)(t / W), X = (int)(t % W);
  const float4 b = boxes[det];
  constexpr int M = 28, PAD = 1, MP = M + 2 * PAD;
  const float scale = (float)MP / (float)M;
  const float w_half = (b.z - b.x) * 0.5f * scale, h_half = (b.w - b.y) * 0.5f * scale;
  const float xc = (b.z + b.x) * 0.5f, yc = (b.w + b.y) * 0.5f;
  const int bx0 = (int)(xc - w_half), by0 = (int)(yc - h_half), bx1 = (int)(xc + w_half), by1 = (int)(yc + h_half);   // .to(int64): truncation
  const int w = max(bx1 - bx0 + 1, 1), h = max(by1 - by0 + 1, 1);
  const int x_0 = max(bx0, 0), x_1 = min(bx1 + 1, W), y_0 = max(by0, 0), y_1 = min(by1 + 1, H);
  if (X < x_0 || X >= x_1 || Y < y_0 || Y >= y_1) { *o = 0.f; return; }
  // bilinear resize of the zero-padded 30 x 30 probability map to (h, w), align_corners = False
  const int dx = X - bx0, dy = Y - by0;
  float fy = ((float)MP / (float)h) * ((float)dy + 0.5f) - 0.5f, fx = ((float)MP / (float)w) * ((float)dx + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < MP - 1 ? 1 : 0), x1 = x0 + (x0 < MP - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const int lab = labels[det];
  auto prob = [&](int py, int px) -> float {
    if (py < PAD || py >= M + PAD || px < PAD || px >= M + PAD) return 0.f;
    const int my = py - PAD, mx = px - PAD;   // 28 x 28 mask pixel = (2 y + a, 2 x + b) of the transposed convolution
    const size_t row = (((size_t)det * 14 + (my >> 1)) * 14 + (mx >> 1)) * 4 + ((my & 1) * 2 + (mx & 1));
    return 1.f / (1.f + expf(-logits[row * Cs + lab]));
  };
  *o = hy * (hx * prob(y0, x0) + lx * prob(y0, x1)) + ly * (hx * prob(y1, x0) + lx * prob(y1, x1));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side: weights
// ---------------------------------------------------------------------------------------------------------------------------------
struct DetConv {
  int Cin = 0, Cin_p = 0, Cout = 0, K = 1, stride = 1, pad = 0;   // Cout = channels the kernel writes (padded heads included)
  float* d_w = nullptr;
  float* d_b = nullptr;
};

struct Bottleneck {
  DetConv c1, c2, c3, down;
  bool has_down = false;
};

typedef std::map<std::string, std::pair<const float*, int64_t>> DetState;

struct SpecEntry {
  std::string name;
  int64_t shape[4];
  int n_dims;
};

static std::vector<SpecEntry> det_spec(int C) {
  std::vector<SpecEntry> v;
  auto add = [&](const std::string& n, std::initializer_list<int64_t> s) {
    SpecEntry e;
    e.name = n;
    e.n_dims = (int)s.size();
    int k = 0;
    for (int64_t d : s) e.shape[k++] = d;
    for (; k < 4; ++k) e.shape[k] = 1;
    v.push_back(e);
  };
  auto bn = [&](const std::string& p, int64_t c) {
    for (const char* s : {"weight", "bias", "running_mean", "running_var"}) add(p + "." + s, {c});
  };
  const std::string B = "backbone.body.";
  add(B + "conv1.weight", {64, 3, 7, 7});
  bn(B + "bn1", 64);
  static const int nb[4] = {3, 4, 6, 3}, pl[4] = {64, 128, 256, 512};
  int64_t inplanes = 64;
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < nb[li]; ++bi) {
      const std::string P = B + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
      const int64_t p = pl[li];
      add(P + "conv1.weight", {p, inplanes, 1, 1});
      bn(P + "bn1", p);
      add(P + "conv2.weight", {p, p, 3, 3});
      bn(P + "bn2", p);
      add(P + "conv3.weight", {4 * p, p, 1, 1});
      bn(P + "bn3", 4 * p);
      if (bi == 0) {
        add(P + "downsample.0.weight", {4 * p, inplanes, 1, 1});
        bn(P + "downsample.1", 4 * p);
      }
      inplanes = 4 * p;
    }
  static const int64_t fc[4] = {256, 512, 1024, 2048};
  for (int i = 0; i < 4; ++i) {
    const std::string s = std::to_string(i);
    add("backbone.fpn.inner_blocks." + s + ".weight", {256, fc[i], 1, 1});
    add("backbone.fpn.inner_blocks." + s + ".bias", {256});
    add("backbone.fpn.layer_blocks." + s + ".weight", {256, 256, 3, 3});
    add("backbone.fpn.layer_blocks." + s + ".bias", {256});
  }
  add("rpn.head.conv.weight", {256, 256, 3, 3});
  add("rpn.head.conv.bias", {256});
  add("rpn.head.cls_logits.weight", {DET_A, 256, 1, 1});
  add("rpn.head.cls_logits.bias", {DET_A});
  add("rpn.head.bbox_pred.weight", {4 * DET_A, 256, 1, 1});
  add("rpn.head.bbox_pred.bias", {4 * DET_A});
  add("roi_heads.box_head.fc6.weight", {1024, 256 * 7 * 7});
  add("roi_heads.box_head.fc6.bias", {1024});
  add("roi_heads.box_head.fc7.weight", {1024, 1024});
  add("roi_heads.box_head.fc7.bias", {1024});
  add("roi_heads.box_predictor.cls_score.weight", {C, 1024});
  add("roi_heads.box_predictor.cls_score.bias", {C});
  add("roi_heads.box_predictor.bbox_pred.weight", {4 * (int64_t)C, 1024});
  add("roi_heads.box_predictor.bbox_pred.bias", {4 * (int64_t)C});
  for (int i = 1; i <= 4; ++i) {
    add("roi_heads.mask_head.mask_fcn" + std::to_string(i) + ".weight", {256, 256, 3, 3});
    add("roi_heads.mask_head.mask_fcn" + std::to_string(i) + ".bias", {256});
  }
  add("roi_heads.mask_predictor.conv5_mask.weight", {256, 256, 2, 2});
  add("roi_heads.mask_predictor.conv5_mask.bias", {256});
  add("roi_heads.mask_predictor.mask_fcn_logits.weight", {C, 256, 1, 1});
  add("roi_heads.mask_predictor.mask_fcn_logits.bias", {C});
  return v;
}

}  // namespace mp

using namespace mp;

// workspace plan: every buffer of a forward at (n, H, W), bump-allocated in floats (64-float aligned, + slack the conv's last K chunk
// may read past a tensor's end)
struct DetPlan {
  int n, H, W, hr, wr, Hp, Wp;          // input, resized, padded (multiple of 32)
  float ratio_h, ratio_w, sy, sx;
  int fh[DET_LEVELS], fw[DET_LEVELS];   // P2..P6
  int a_total;                          // anchors per image
  int n_seg_rpn;
  size_t total = 0;
  std::map<std::string, size_t> off;    // float offsets
  size_t take(const std::string& name, size_t floats) {
    const size_t o = total;
    off[name] = o;
    total += (floats + 63) / 64 * 64;
    return o;
  }
};

struct mp_detector {
  mp_detector_config cfg;
  int C, Cpred_s, Cmask_s;   // classes; padded row strides of the predictor / mask-logit outputs
  DetConv stem;
  std::vector<Bottleneck> blocks;
  std::vector<int> stage_of_block;
  DetConv fpn_inner[4], fpn_layer[4], rpn_conv, rpn_head, fc6, fc7, pred, mask_fcn[4], mask_deconv, mask_logits;
  float base_anchors[DET_LEVELS][DET_A][4];
  std::vector<void*> allocs;
  // last forward (debug taps)
  bool ran = false;
  DetPlan last;
  void* last_ws = nullptr;
};

namespace {

constexpr size_t DET_SPLITK_FLOATS = 16u << 20;

inline size_t tensor_floats(int N, int H, int W, int C, int b) { return (size_t)N * (H + 2 * b) * (W + 2 * b) * C + (size_t)(W + 2 * b) * C + 64; }

int det_upload(mp_detector* d, const std::vector<float>& h, float** p) {
  MP_CHECK_HIP(hipMalloc(p, h.size() * sizeof(float)));
  MP_CHECK_HIP(hipMemcpy(*p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  d->allocs.push_back(*p);
  return MP_OK;
}

const float* det_find(const DetState& sm, const std::string& k, int64_t numel) {
  auto it = sm.find(k);
  if (it == sm.end()) {
    set_error("mp_detector_create: missing state_dict key '%s'", k.c_str());
    return nullptr;
  }
  if (it->second.second != numel) {
    set_error("mp_detector_create: key '%s' has %ld elements, expected %ld", k.c_str(), (long)it->second.second, (long)numel);
    return nullptr;
  }
  return it->second.first;
}

// OIHW weights (+ optional FrozenBatchNorm `bn`, + optional bias key) -> packed conv; cout_pad > Cout appends zero output channels
int det_make_conv(mp_detector* d, const DetState& sm, const std::vector<float>& w_oihw, int Cout, int Cin, int K, int stride, int pad,
                  const std::string& bn, const float* bias, int cout_pad, DetConv* L) {
  const int Co = std::max(Cout, cout_pad);
  L->Cin = Cin; L->Cin_p = (Cin + 3) / 4 * 4; L->Cout = Co; L->K = K; L->stride = stride; L->pad = pad;
  std::vector<float> scale, shift(Co, 0.f);
  bool has_shift = false;
  if (!bn.empty()) {
    const float* g = det_find(sm, bn + ".weight", Cout);
    const float* b = det_find(sm, bn + ".bias", Cout);
    const float* m = det_find(sm, bn + ".running_mean", Cout);
    const float* v = det_find(sm, bn + ".running_var", Cout);
    if (!g || !b || !m || !v) return MP_ERR_INVALID;
    scale.assign(Co, 0.f);
    for (int c = 0; c < Cout; ++c) {   // FrozenBatchNorm2d: eps = 1e-5 (ops/misc.py)
      const float s = g[c] / sqrtf(v[c] + 1e-5f);
      scale[c] = s;
      shift[c] = b[c] - m[c] * s;
    }
    has_shift = true;
  }
  if (bias) {
    for (int c = 0; c < Cout; ++c) shift[c] += bias[c];
    has_shift = true;
  }
  std::vector<float> w(w_oihw);
  w.resize((size_t)Co * Cin * K * K, 0.f);
  std::vector<float> packed(mp_conv_packed_floats(L->Cin_p, Co, K, K));
  int rc = mp_conv_pack_weights(w.data(), Co, Cin, K, K, L->Cin_p, scale.empty() ? nullptr : scale.data(), packed.data());
  if (rc) return rc;
  rc = det_upload(d, packed, &L->d_w);
  if (rc) return rc;
  if (has_shift) rc = det_upload(d, shift, &L->d_b);
  return rc;
}

int det_conv_from_key(mp_detector* d, const DetState& sm, const std::string& wkey, int Cout, int Cin, int K, int stride, int pad,
                      const std::string& bn, const std::string& bias_key, DetConv* L) {
  const float* w = det_find(sm, wkey, (int64_t)Cout * Cin * K * K);
  if (!w) return MP_ERR_INVALID;
  const float* b = nullptr;
  if (!bias_key.empty()) {
    b = det_find(sm, bias_key, Cout);
    if (!b) return MP_ERR_INVALID;
  }
  return det_make_conv(d, sm, std::vector<float>(w, w + (size_t)Cout * Cin * K * K), Cout, Cin, K, stride, pad, bn, b, 0, L);
}

int det_run_conv(const DetConv& L, const float* x, int N, int H, int W, int in_border, float* y, int out_border, const float* res, int relu,
                 hipStream_t s, float* sk) {
  mp_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.d_x = x; c.N = N; c.H = H; c.W = W; c.C = L.Cin_p; c.c_real = L.Cin; c.in_border = in_border;
  c.d_w = L.d_w; c.d_bias = L.d_b; c.Cout = L.Cout; c.KH = L.K; c.KW = L.K; c.stride = L.stride; c.pad = L.pad;
  c.d_y = y; c.out_border = out_border; c.d_residual = res; c.relu = relu;
  c.d_splitk_ws = sk;
  c.splitk_ws_floats = sk ? (int64_t)DET_SPLITK_FLOATS : 0;
  return mp_conv2d_nhwc(&c, s);
}

int det_make_plan(const mp_detector* d, int n, int H, int W, DetPlan* p) {
  const mp_detector_config& c = d->cfg;
  p->n = n; p->H = H; p->W = W;
  // transform.py _resize_image_and_masks: scale = min(min_size / min(h, w), max_size / max(h, w)) in float32; out = floor(in * scale)
  const float scale = std::min((float)c.min_size / (float)std::min(H, W), (float)c.max_size / (float)std::max(H, W));
  p->hr = (int)std::floor((double)H * (double)scale);
  p->wr = (int)std::floor((double)W * (double)scale);
  MP_REQUIRE(p->hr >= 32 && p->wr >= 32, "mp_detector: resized image %dx%d too small", p->hr, p->wr);
  p->Hp = (p->hr + 31) / 32 * 32;
  p->Wp = (p->wr + 31) / 32 * 32;
  p->sy = (float)H / (float)p->hr;
  p->sx = (float)W / (float)p->wr;
  p->ratio_h = (float)H / (float)p->hr;
  p->ratio_w = (float)W / (float)p->wr;
  for (int l = 0; l < 4; ++l) { p->fh[l] = p->Hp >> (l + 2); p->fw[l] = p->Wp >> (l + 2); }
  p->fh[4] = (p->fh[3] - 1) / 2 + 1;
  p->fw[4] = (p->fw[3] - 1) / 2 + 1;
  p->a_total = 0;
  for (int l = 0; l < DET_LEVELS; ++l) p->a_total += p->fh[l] * p->fw[l] * DET_A;
  p->n_seg_rpn = n * DET_LEVELS;
  p->total = 0;
  p->off.clear();
  const int C = d->C, R = c.rpn_post_nms_top_n, Kp = c.rpn_pre_nms_top_n, D = c.box_detections_per_img;
  p->take("x0", tensor_floats(n, p->Hp, p->Wp, 4, 3));
  p->take("stem", tensor_floats(n, p->Hp / 2, p->Wp / 2, 64, 1));
  p->take("pool", tensor_floats(n, p->Hp / 4, p->Wp / 4, 64, 1));
  static const int pl[4] = {64, 128, 256, 512};
  for (int s = 0; s < 4; ++s) {
    const int h = p->fh[s], w = p->fw[s];
    const std::string S = "s" + std::to_string(s);
    p->take(S + ".xa", tensor_floats(n, h, w, 4 * pl[s], 1));
    p->take(S + ".xb", tensor_floats(n, h, w, 4 * pl[s], 1));
    p->take(S + ".t1", tensor_floats(n, h, w, pl[s], 1));
    // block 0 runs its first 1x1 at the INPUT resolution of the stage: its own buffer (a buffer must keep ONE geometry, its zero
    // border is only established once per forward)
    p->take(S + ".t1in", tensor_floats(n, s == 0 ? h : 2 * h, s == 0 ? w : 2 * w, pl[s], 1));
    p->take(S + ".t2", tensor_floats(n, h, w, pl[s], 1));
    p->take(S + ".d", tensor_floats(n, h, w, 4 * pl[s], 1));
    p->take("L" + std::to_string(s), tensor_floats(n, h, w, DET_FPN_C, 1));
    p->take("U" + std::to_string(s), tensor_floats(n, h, w, DET_FPN_C, 1));
  }
  for (int l = 0; l < DET_LEVELS; ++l) {
    p->take("P" + std::to_string(l + 2), tensor_floats(n, p->fh[l], p->fw[l], DET_FPN_C, 1));
    p->take("rpn_t" + std::to_string(l), tensor_floats(n, p->fh[l], p->fw[l], DET_FPN_C, 0));
    p->take("rpn_h" + std::to_string(l), tensor_floats(n, p->fh[l], p->fw[l], 16, 0));
  }
  p->take("keys", (size_t)n * p->a_total);
  p->take("seg_off", 64 + (size_t)n * std::max(DET_LEVELS, C) + 2);   // int32 segment tables of the sorts
  p->take("seg_off2", 64 + (size_t)n + 2);
  p->take("seg_off3", 64 + (size_t)n + 2);
  p->take("idx1", (size_t)n * DET_LEVELS * Kp);
  p->take("cnt1", (size_t)n * DET_LEVELS + 64);
  p->take("cand_box", (size_t)n * DET_LEVELS * Kp * 4);
  p->take("cand_score", (size_t)n * DET_LEVELS * Kp);
  p->take("cand_keep", (size_t)n * DET_LEVELS * Kp);
  p->take("keys2", (size_t)n * DET_LEVELS * Kp);
  p->take("idx2", (size_t)n * R);
  p->take("proposals", (size_t)n * R * 4);
  p->take("proposal_scores", (size_t)n * R);
  p->take("proposal_counts", (size_t)n + 64);
  p->take("roi7", tensor_floats(n * R, 1, 1, 7 * 7 * DET_FPN_C, 0));
  p->take("fc6", tensor_floats(n * R, 1, 1, 1024, 0));
  p->take("fc7", tensor_floats(n * R, 1, 1, 1024, 0));
  p->take("class_logits", tensor_floats(n * R, 1, 1, d->Cpred_s, 0));
  const size_t n_c2 = (size_t)n * (C - 1) * R;
  p->take("c2_box", n_c2 * 4);
  p->take("c2_key", n_c2);
  p->take("c2_idx", n_c2);
  p->take("c2_cnt", (size_t)n * (C - 1) + 64);
  p->take("c2_sbox", n_c2 * 4);
  p->take("c2_skey", n_c2);
  p->take("c2_keep", n_c2);
  p->take("c2_fkey", n_c2);
  p->take("f_idx", (size_t)n * D);
  p->take("f_cnt", (size_t)n + 64);
  p->take("det_resized", (size_t)n * D * 4);
  p->take("roi14", tensor_floats(n * D, 14, 14, DET_FPN_C, 1));
  p->take("m_a", tensor_floats(n * D, 14, 14, DET_FPN_C, 1));
  p->take("m_b", tensor_floats(n * D, 14, 14, DET_FPN_C, 1));
  p->take("m_up", tensor_floats(n * D, 14, 14, 4 * DET_FPN_C, 0));
  p->take("mask_logits", tensor_floats(n * D * 14 * 14 * 4, 1, 1, d->Cmask_s, 0));
  p->take("splitk", DET_SPLITK_FLOATS);
  p->take("end", 4096);
  return MP_OK;
}

}  // namespace

extern "C" int mp_detector_default_config(mp_detector_config* cfg, int n_classes, int min_size, int max_size) {
  MP_REQUIRE(cfg && n_classes >= 2 && min_size > 0 && max_size >= min_size, "mp_detector_default_config: bad arguments");
  memset(cfg, 0, sizeof(*cfg));
  cfg->n_classes = n_classes;
  cfg->min_size = min_size;
  cfg->max_size = max_size;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
  const int sizes[5] = {32, 64, 128, 256, 512};
  const float ar[3] = {0.5f, 1.0f, 2.0f};
  memcpy(cfg->image_mean, mean, sizeof(mean));
  memcpy(cfg->image_std, sd, sizeof(sd));
  memcpy(cfg->anchor_sizes, sizes, sizeof(sizes));
  memcpy(cfg->aspect_ratios, ar, sizeof(ar));
  cfg->rpn_pre_nms_top_n = 1000; cfg->rpn_post_nms_top_n = 1000;
  cfg->rpn_nms_thresh = 0.7f; cfg->rpn_score_thresh = 0.0f; cfg->rpn_min_size = 1e-3f;
  cfg->box_score_thresh = 0.05f; cfg->box_nms_thresh = 0.5f; cfg->box_min_size = 1e-2f;
  cfg->box_detections_per_img = 100;
  return MP_OK;
}

extern "C" int mp_detector_state_spec(int n_classes, int idx, char* name, int name_len, int64_t* shape4, int32_t* n_dims) {
  MP_REQUIRE(n_classes >= 2 && idx >= 0 && name && name_len > 0 && shape4 && n_dims, "mp_detector_state_spec: bad arguments");
  const std::vector<SpecEntry> v = det_spec(n_classes);
  if (idx >= (int)v.size()) return 1;
  snprintf(name, name_len, "%s", v[idx].name.c_str());
  for (int k = 0; k < 4; ++k) shape4[k] = v[idx].shape[k];
  *n_dims = v[idx].n_dims;
  return MP_OK;
}

extern "C" int mp_detector_destroy(mp_detector* d) {
  if (!d) return MP_OK;
  for (void* p : d->allocs) (void)hipFree(p);
  delete d;
  return MP_OK;
}

extern "C" int mp_detector_create(const mp_detector_config* cfg, const mp_named_tensor* st, int n_tensors, mp_detector** out) {
  MP_REQUIRE(cfg && st && n_tensors > 0 && out, "mp_detector_create: bad arguments");
  MP_REQUIRE(cfg->n_classes >= 2 && cfg->n_classes <= 1024, "mp_detector_create: n_classes %d", cfg->n_classes);
  MP_REQUIRE(cfg->rpn_pre_nms_top_n >= 1 && cfg->rpn_pre_nms_top_n <= DET_MAX_SEG && cfg->rpn_post_nms_top_n >= 1 &&
                 cfg->rpn_post_nms_top_n <= DET_MAX_SEG && cfg->box_detections_per_img >= 1 && cfg->box_detections_per_img <= DET_MAX_SEG,
             "mp_detector_create: top-n sizes must lie in [1, %d]", DET_MAX_SEG);
  for (int k = 0; k < 3; ++k) MP_REQUIRE(cfg->image_std[k] > 0.f && cfg->aspect_ratios[k] > 0.f, "mp_detector_create: bad std / aspect ratio");
  DetState sm;
  for (int i = 0; i < n_tensors; ++i) sm[st[i].name] = std::make_pair(st[i].h_data, st[i].numel);
  mp_detector* d = new mp_detector();
  d->cfg = *cfg;
  const int C = d->C = cfg->n_classes;
  d->Cpred_s = (5 * C + 3) / 4 * 4;
  d->Cmask_s = (C + 3) / 4 * 4;
  int rc;
#define DET_TRY(e) do { rc = (e); if (rc) { mp_detector_destroy(d); return rc; } } while (0)
  const std::string B = "backbone.body.";
  DET_TRY(det_conv_from_key(d, sm, B + "conv1.weight", 64, 3, 7, 2, 3, B + "bn1", "", &d->stem));
  static const int nb[4] = {3, 4, 6, 3}, pl[4] = {64, 128, 256, 512};
  int inplanes = 64;
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < nb[li]; ++bi) {
      const std::string P = B + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
      const int p = pl[li], stride = (bi == 0 && li > 0) ? 2 : 1;
      Bottleneck blk;
      DET_TRY(det_conv_from_key(d, sm, P + "conv1.weight", p, inplanes, 1, 1, 0, P + "bn1", "", &blk.c1));
      DET_TRY(det_conv_from_key(d, sm, P + "conv2.weight", p, p, 3, stride, 1, P + "bn2", "", &blk.c2));     // (torchvision: stride on the 3x3)
      DET_TRY(det_conv_from_key(d, sm, P + "conv3.weight", 4 * p, p, 1, 1, 0, P + "bn3", "", &blk.c3));
      blk.has_down = bi == 0;
      if (blk.has_down) DET_TRY(det_conv_from_key(d, sm, P + "downsample.0.weight", 4 * p, inplanes, 1, stride, 0, P + "downsample.1", "", &blk.down));
      d->blocks.push_back(blk);
      d->stage_of_block.push_back(li);
      inplanes = 4 * p;
    }
  static const int fc[4] = {256, 512, 1024, 2048};
  for (int i = 0; i < 4; ++i) {
    const std::string s = std::to_string(i);
    DET_TRY(det_conv_from_key(d, sm, "backbone.fpn.inner_blocks." + s + ".weight", 256, fc[i], 1, 1, 0, "", "backbone.fpn.inner_blocks." + s + ".bias", &d->fpn_inner[i]));
    DET_TRY(det_conv_from_key(d, sm, "backbone.fpn.layer_blocks." + s + ".weight", 256, 256, 3, 1, 1, "", "backbone.fpn.layer_blocks." + s + ".bias", &d->fpn_layer[i]));
  }
  DET_TRY(det_conv_from_key(d, sm, "rpn.head.conv.weight", 256, 256, 3, 1, 1, "", "rpn.head.conv.bias", &d->rpn_conv));
  {  // cls_logits (A) + bbox_pred (4A) as ONE 1x1 convolution with 16 output channels
    const float* wc = det_find(sm, "rpn.head.cls_logits.weight", DET_A * 256);
    const float* bc = det_find(sm, "rpn.head.cls_logits.bias", DET_A);
    const float* wb = det_find(sm, "rpn.head.bbox_pred.weight", 4 * DET_A * 256);
    const float* bb = det_find(sm, "rpn.head.bbox_pred.bias", 4 * DET_A);
    if (!wc || !bc || !wb || !bb) { mp_detector_destroy(d); return MP_ERR_INVALID; }
    std::vector<float> w(16 * 256, 0.f), b(16, 0.f);
    memcpy(w.data(), wc, sizeof(float) * DET_A * 256);
    memcpy(w.data() + DET_A * 256, wb, sizeof(float) * 4 * DET_A * 256);
    memcpy(b.data(), bc, sizeof(float) * DET_A);
    memcpy(b.data() + DET_A, bb, sizeof(float) * 4 * DET_A);
    DET_TRY(det_make_conv(d, sm, w, 16, 256, 1, 1, 0, "", b.data(), 16, &d->rpn_head));
  }
  {  // fc6: torchvision flattens [R, 256, 7, 7] channel-major; the RoIAlign output here is [R, 7, 7, 256]
    const float* w6 = det_find(sm, "roi_heads.box_head.fc6.weight", (int64_t)1024 * 12544);
    const float* b6 = det_find(sm, "roi_heads.box_head.fc6.bias", 1024);
    if (!w6 || !b6) { mp_detector_destroy(d); return MP_ERR_INVALID; }
    std::vector<float> w((size_t)1024 * 12544);
    for (int o = 0; o < 1024; ++o)
      for (int c = 0; c < 256; ++c)
        for (int q = 0; q < 49; ++q) w[(size_t)o * 12544 + (size_t)q * 256 + c] = w6[(size_t)o * 12544 + (size_t)c * 49 + q];
    DET_TRY(det_make_conv(d, sm, w, 1024, 12544, 1, 1, 0, "", b6, 0, &d->fc6));
  }
  DET_TRY(det_conv_from_key(d, sm, "roi_heads.box_head.fc7.weight", 1024, 1024, 1, 1, 0, "", "roi_heads.box_head.fc7.bias", &d->fc7));
  {  // cls_score (C) + bbox_pred (4C) as one layer
    const float* wc = det_find(sm, "roi_heads.box_predictor.cls_score.weight", (int64_t)C * 1024);
    const float* bc = det_find(sm, "roi_heads.box_predictor.cls_score.bias", C);
    const float* wb = det_find(sm, "roi_heads.box_predictor.bbox_pred.weight", (int64_t)4 * C * 1024);
    const float* bb = det_find(sm, "roi_heads.box_predictor.bbox_pred.bias", 4 * C);
    if (!wc || !bc || !wb || !bb) { mp_detector_destroy(d); return MP_ERR_INVALID; }
    std::vector<float> w((size_t)d->Cpred_s * 1024, 0.f), b(d->Cpred_s, 0.f);
    memcpy(w.data(), wc, sizeof(float) * C * 1024);
    memcpy(w.data() + (size_t)C * 1024, wb, sizeof(float) * 4 * C * 1024);
    memcpy(b.data(), bc, sizeof(float) * C);
    memcpy(b.data() + C, bb, sizeof(float) * 4 * C);
    DET_TRY(det_make_conv(d, sm, w, d->Cpred_s, 1024, 1, 1, 0, "", b.data(), d->Cpred_s, &d->pred));
  }
  for (int i = 0; i < 4; ++i) {
    const std::string k = "roi_heads.mask_head.mask_fcn" + std::to_string(i + 1);
    DET_TRY(det_conv_from_key(d, sm, k + ".weight", 256, 256, 3, 1, 1, "", k + ".bias", &d->mask_fcn[i]));
  }
  {  // ConvTranspose2d(256, 256, 2, 2): out[o, 2y+a, 2x+b] = sum_i in[i, y, x] W[i, o, a, b] + bias[o]  ->  1x1 conv onto (a, b, o)
    const float* wt = det_find(sm, "roi_heads.mask_predictor.conv5_mask.weight", 256 * 256 * 4);
    const float* bt = det_find(sm, "roi_heads.mask_predictor.conv5_mask.bias", 256);
    if (!wt || !bt) { mp_detector_destroy(d); return MP_ERR_INVALID; }
    std::vector<float> w((size_t)1024 * 256), b(1024);
    for (int ab = 0; ab < 4; ++ab)
      for (int o = 0; o < 256; ++o) {
        b[ab * 256 + o] = bt[o];
        for (int i = 0; i < 256; ++i) w[((size_t)ab * 256 + o) * 256 + i] = wt[((size_t)i * 256 + o) * 4 + ab];
      }
    DET_TRY(det_make_conv(d, sm, w, 1024, 256, 1, 1, 0, "", b.data(), 0, &d->mask_deconv));
  }
  {
    const float* wl = det_find(sm, "roi_heads.mask_predictor.mask_fcn_logits.weight", (int64_t)C * 256);
    const float* bl = det_find(sm, "roi_heads.mask_predictor.mask_fcn_logits.bias", C);
    if (!wl || !bl) { mp_detector_destroy(d); return MP_ERR_INVALID; }
    std::vector<float> b(d->Cmask_s, 0.f);
    memcpy(b.data(), bl, sizeof(float) * C);
    DET_TRY(det_make_conv(d, sm, std::vector<float>(wl, wl + (size_t)C * 256), C, 256, 1, 1, 0, "", b.data(), d->Cmask_s, &d->mask_logits));
  }
#undef DET_TRY
  // anchor_utils.py generate_anchors: h_ratios = sqrt(ar), w_ratios = 1 / h_ratios, base = round([-w, -h, w, h] / 2) (float32, half to even)
  for (int l = 0; l < DET_LEVELS; ++l)
    for (int a = 0; a < DET_A; ++a) {
      const float hr = sqrtf(cfg->aspect_ratios[a]), wr = 1.0f / hr;
      const float ws = wr * (float)cfg->anchor_sizes[l], hs = hr * (float)cfg->anchor_sizes[l];
      d->base_anchors[l][a][0] = nearbyintf(-ws / 2.f);
      d->base_anchors[l][a][1] = nearbyintf(-hs / 2.f);
      d->base_anchors[l][a][2] = nearbyintf(ws / 2.f);
      d->base_anchors[l][a][3] = nearbyintf(hs / 2.f);
    }
  *out = d;
  return MP_OK;
}

extern "C" size_t mp_detector_workspace_bytes(const mp_detector* d, int n_images, int H, int W) {
  if (!d || n_images <= 0 || H <= 0 || W <= 0) return 0;
  DetPlan p;
  if (det_make_plan(d, n_images, H, W, &p)) return 0;
  return p.total * sizeof(float);
}

extern "C" int mp_detector_forward(mp_detector* d, const float* d_images, int n, int H, int W, float* d_boxes, float* d_scores, int32_t* d_labels,
                                   int32_t* d_counts, float* d_masks, void* d_ws, size_t ws_bytes, mp_stream stream) {
  MP_REQUIRE(d && d_images && d_boxes && d_scores && d_labels && d_counts && d_ws, "mp_detector_forward: null pointer");
  MP_REQUIRE(n > 0 && n <= 4096 && H > 0 && W > 0, "mp_detector_forward: bad size");
  DetPlan p;
  int rc = det_make_plan(d, n, H, W, &p);
  if (rc) return rc;
  MP_REQUIRE(ws_bytes >= p.total * sizeof(float), "mp_detector_forward: workspace %zu < %zu bytes", ws_bytes, p.total * sizeof(float));
  MP_REQUIRE((long)n * p.a_total < (1L << 31) && (long)n * (d->C - 1) * d->cfg.rpn_post_nms_top_n < (1L << 31),
             "mp_detector_forward: %d images of %d anchors exceed the 32-bit index range of the selection kernels; split the batch", n, p.a_total);
  // The selection kernels rank by counting (O(len^2) compares per segment: deterministic, no library sort).  That is ~3e9 compares for
  // the 57.6 k P2 anchors of a 480x640 frame (1.5 ms) but grows quadratically: bound the segment sizes instead of degrading silently
  // (torchvision's default 800x1333 transform gives ~200 k anchors on P2 = 4e10 compares per frame -- use a radix select there).
  MP_REQUIRE((long)p.hr * p.wr <= 1024L * 1024L && d->C <= 256,
             "mp_detector_forward: resized frame %dx%d / %d classes exceed what the rank-sort selection is sized for (<= 1024x1024, <= 256 classes)",
             p.hr, p.wr, d->C);
  const mp_detector_config& cfg = d->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)d_ws;
  auto F = [&](const std::string& k) { return ws + p.off.at(k); };
  auto I = [&](const std::string& k) { return reinterpret_cast<int*>(ws + p.off.at(k)); };
  float* SK = F("splitk");
  // zero everything: borders of every padded map, the pad region of the batched input, counters (cheap next to ResNet-50)
  MP_CHECK_HIP(hipMemsetAsync(d_ws, 0, p.total * sizeof(float), s));
  const int C = d->C, R = cfg.rpn_post_nms_top_n, Kp = cfg.rpn_pre_nms_top_n, D = cfg.box_detections_per_img;

  // ---- transform + backbone + FPN --------------------------------------------------------------------------------------------
  {
    ProfScope prof("det_preprocess", 0.0, (double)n * (12.0 * H * W + 16.0 * p.hr * p.wr), s);
    hipLaunchKernelGGL(det_preprocess_kernel, dim3(ceil_div((long)p.hr * p.wr, 256), n), dim3(256), 0, s, d_images, H, W, p.hr, p.wr, p.sy, p.sx,
                       cfg.image_mean[0], cfg.image_mean[1], cfg.image_mean[2], cfg.image_std[0], cfg.image_std[1], cfg.image_std[2], F("x0"),
                       p.Hp, p.Wp, 3);
  }
  rc = det_run_conv(d->stem, F("x0"), n, p.Hp, p.Wp, 3, F("stem"), 1, nullptr, 1, s, SK);
  if (rc) return rc;
  rc = mp_maxpool3x3s2(F("stem"), n, p.Hp / 2, p.Wp / 2, 64, 1, F("pool"), 1, nullptr, nullptr, nullptr, s);
  if (rc) return rc;
  const float* x = F("pool");
  int xh = p.Hp / 4, xw = p.Wp / 4;
  const float* stage_out[4] = {nullptr, nullptr, nullptr, nullptr};
  for (size_t i = 0; i < d->blocks.size(); ++i) {
    const Bottleneck& b = d->blocks[i];
    const int st = d->stage_of_block[i];
    const std::string S = "s" + std::to_string(st);
    const int oh = p.fh[st], ow = p.fw[st];
    float* t1 = b.has_down ? F(S + ".t1in") : F(S + ".t1");
    rc = det_run_conv(b.c1, x, n, xh, xw, 1, t1, 1, nullptr, 1, s, SK);
    if (rc) return rc;
    rc = det_run_conv(b.c2, t1, n, xh, xw, 1, F(S + ".t2"), 1, nullptr, 1, s, SK);
    if (rc) return rc;
    const float* idn = x;
    if (b.has_down) {
      rc = det_run_conv(b.down, x, n, xh, xw, 1, F(S + ".d"), 1, nullptr, 0, s, SK);
      if (rc) return rc;
      idn = F(S + ".d");
    }
    float* y = (x == F(S + ".xa")) ? F(S + ".xb") : F(S + ".xa");
    rc = det_run_conv(b.c3, F(S + ".t2"), n, oh, ow, 1, y, 1, idn, 1, s, SK);
    if (rc) return rc;
    x = y; xh = oh; xw = ow;
    stage_out[st] = y;
  }
  // FPN (ops/feature_pyramid_network.py): last_inner = inner[3](C5); P5 = layer[3](last_inner); going down: lateral + nearest upsample
  rc = det_run_conv(d->fpn_inner[3], stage_out[3], n, p.fh[3], p.fw[3], 1, F("L3"), 1, nullptr, 0, s, SK);
  if (rc) return rc;
  rc = det_run_conv(d->fpn_layer[3], F("L3"), n, p.fh[3], p.fw[3], 1, F("P5"), 1, nullptr, 0, s, SK);
  if (rc) return rc;
  for (int l = 2; l >= 0; --l) {
    const std::string sl = std::to_string(l), su = std::to_string(l + 1);
    {
      const long total = (long)n * p.fh[l] * p.fw[l] * (DET_FPN_C / 4);
      ProfScope prof("det_resize_nearest", 0.0, (double)total * 32.0, s);
      hipLaunchKernelGGL(det_resize_nearest_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, F("L" + su), p.fh[l + 1], p.fw[l + 1], F("U" + sl),
                         p.fh[l], p.fw[l], DET_FPN_C, n, 1, 1, 0, 0);
    }
    rc = det_run_conv(d->fpn_inner[l], stage_out[l], n, p.fh[l], p.fw[l], 1, F("L" + sl), 1, F("U" + sl), 0, s, SK);
    if (rc) return rc;
    rc = det_run_conv(d->fpn_layer[l], F("L" + sl), n, p.fh[l], p.fw[l], 1, F("P" + std::to_string(l + 2)), 1, nullptr, 0, s, SK);
    if (rc) return rc;
  }
  {  // LastLevelMaxPool: F.max_pool2d(P5, 1, 2, 0)
    const long total = (long)n * p.fh[4] * p.fw[4] * (DET_FPN_C / 4);
    hipLaunchKernelGGL(det_resize_nearest_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, F("P5"), p.fh[3], p.fw[3], F("P6"), p.fh[4], p.fw[4],
                       DET_FPN_C, n, 1, 1, 2, 2);
  }

  // ---- RPN (models/detection/rpn.py) --------------------------------------------------------------------------------------------
  RpnLevels L;
  L.off[0] = 0;
  for (int l = 0; l < DET_LEVELS; ++l) {
    const std::string sl = std::to_string(l);
    rc = det_run_conv(d->rpn_conv, F("P" + std::to_string(l + 2)), n, p.fh[l], p.fw[l], 1, F("rpn_t" + sl), 0, nullptr, 1, s, SK);
    if (rc) return rc;
    rc = det_run_conv(d->rpn_head, F("rpn_t" + sl), n, p.fh[l], p.fw[l], 0, F("rpn_h" + sl), 0, nullptr, 0, s, SK);
    if (rc) return rc;
    L.head[l] = F("rpn_h" + sl);
    L.gh[l] = p.fh[l]; L.gw[l] = p.fw[l];
    L.off[l + 1] = L.off[l] + p.fh[l] * p.fw[l] * DET_A;
    L.stride_y[l] = p.Hp / p.fh[l];   // anchor_utils.py: strides = image_size // grid_size (padded batch size)
    L.stride_x[l] = p.Wp / p.fw[l];
    memcpy(L.base[l], d->base_anchors[l], sizeof(L.base[l]));
  }
  {
    const long total = (long)n * p.a_total;
    hipLaunchKernelGGL(det_rpn_keys_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, L, n, F("keys"));
  }
  int max_len = 0;
  for (int l = 0; l < DET_LEVELS; ++l) max_len = std::max(max_len, L.off[l + 1] - L.off[l]);
  hipLaunchKernelGGL(det_rpn_seg_offsets_kernel, dim3(ceil_div((long)n * DET_LEVELS + 1, 256)), dim3(256), 0, s, L, n, I("seg_off"));
  {
    ProfScope prof("det_rank_topk", 0.0, 0.0, s);
    hipLaunchKernelGGL(det_rank_topk_kernel, dim3(ceil_div(max_len, 256), n * DET_LEVELS), dim3(256), 0, s, F("keys"), I("seg_off"), Kp, I("idx1"),
                       I("cnt1"));
  }
  {
    const long total = (long)n * DET_LEVELS * Kp;
    hipLaunchKernelGGL(det_rpn_gather_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, L, n, F("keys"), I("idx1"), I("cnt1"), Kp, (float)p.hr,
                       (float)p.wr, cfg.rpn_min_size, cfg.rpn_score_thresh, reinterpret_cast<float4*>(F("cand_box")), F("cand_score"),
                       I("cand_keep"));
    ProfScope prof("det_nms", 0.0, 0.0, s);
    hipLaunchKernelGGL(det_nms_kernel, dim3(n * DET_LEVELS), dim3(256), 0, s, reinterpret_cast<const float4*>(F("cand_box")), I("cand_keep"),
                       (const int*)nullptr, Kp, cfg.rpn_nms_thresh);
    hipLaunchKernelGGL(det_masked_keys_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, F("cand_score"), I("cand_keep"), total, F("keys2"));
  }
  hipLaunchKernelGGL(det_seg_offsets_kernel, dim3(ceil_div((long)n + 1, 256)), dim3(256), 0, s, n, DET_LEVELS * Kp, I("seg_off2"));
  hipLaunchKernelGGL(det_rank_topk_kernel, dim3(ceil_div(DET_LEVELS * Kp, 256), n), dim3(256), 0, s, F("keys2"), I("seg_off2"), R, I("idx2"),
                     I("proposal_counts"));
  hipLaunchKernelGGL(det_gather_boxes_kernel, dim3(ceil_div((long)n * R, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(F("cand_box")),
                     F("cand_score"), DET_LEVELS * Kp, I("idx2"), I("proposal_counts"), R, n, reinterpret_cast<float4*>(F("proposals")),
                     F("proposal_scores"));
  hipLaunchKernelGGL(det_clamp_counts_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, I("proposal_counts"), n, R);

  // ---- box head (roi_heads.py) ----------------------------------------------------------------------------------------------------
  PyramidRef py;
  for (int l = 0; l < 4; ++l) { py.feat[l] = F("P" + std::to_string(l + 2)); py.h[l] = p.fh[l]; py.w[l] = p.fw[l]; }
  {
    const long total = (long)n * R * 49 * (DET_FPN_C / 4);
    ProfScope prof("det_roi_align", 0.0, (double)total * 16.0 * 5.0, s);
    hipLaunchKernelGGL(det_roi_align_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, py, reinterpret_cast<const float4*>(F("proposals")),
                       I("proposal_counts"), R, n, 7, 0, F("roi7"));
  }
  rc = det_run_conv(d->fc6, F("roi7"), n * R, 1, 1, 0, F("fc6"), 0, nullptr, 1, s, SK);
  if (rc) return rc;
  rc = det_run_conv(d->fc7, F("fc6"), n * R, 1, 1, 0, F("fc7"), 0, nullptr, 1, s, SK);
  if (rc) return rc;
  rc = det_run_conv(d->pred, F("fc7"), n * R, 1, 1, 0, F("class_logits"), 0, nullptr, 0, s, SK);
  if (rc) return rc;
  hipLaunchKernelGGL(det_box_post_kernel, dim3(ceil_div((long)n * R, 256)), dim3(256), 0, s, F("class_logits"), d->Cpred_s, C,
                     reinterpret_cast<const float4*>(F("proposals")), I("proposal_counts"), R, n, (float)p.hr, (float)p.wr, cfg.box_score_thresh,
                     cfg.box_min_size, reinterpret_cast<float4*>(F("c2_box")), F("c2_key"));
  const int n_seg2 = n * (C - 1);
  // (seg_off is free again: the RPN sort that used it was enqueued earlier on the same stream)
  hipLaunchKernelGGL(det_seg_offsets_kernel, dim3(ceil_div((long)n_seg2 + 1, 256)), dim3(256), 0, s, n_seg2, R, I("seg_off"));
  hipLaunchKernelGGL(det_rank_topk_kernel, dim3(ceil_div(R, 256), n_seg2), dim3(256), 0, s, F("c2_key"), I("seg_off"), R, I("c2_idx"), I("c2_cnt"));
  {
    const long total = (long)n_seg2 * R;
    hipLaunchKernelGGL(det_sorted_gather_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(F("c2_box")), F("c2_key"),
                       I("c2_idx"), I("c2_cnt"), R, total, reinterpret_cast<float4*>(F("c2_sbox")), F("c2_skey"), I("c2_keep"));
    hipLaunchKernelGGL(det_nms_kernel, dim3(n_seg2), dim3(256), 0, s, reinterpret_cast<const float4*>(F("c2_sbox")), I("c2_keep"), I("c2_cnt"), R,
                       cfg.box_nms_thresh);
    hipLaunchKernelGGL(det_masked_keys_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, F("c2_skey"), I("c2_keep"), total, F("c2_fkey"));
  }
  hipLaunchKernelGGL(det_seg_offsets_kernel, dim3(ceil_div((long)n + 1, 256)), dim3(256), 0, s, n, (C - 1) * R, I("seg_off3"));
  hipLaunchKernelGGL(det_rank_topk_kernel, dim3(ceil_div((long)(C - 1) * R, 256), n), dim3(256), 0, s, F("c2_fkey"), I("seg_off3"), D, I("f_idx"),
                     I("f_cnt"));
  hipLaunchKernelGGL(det_final_kernel, dim3(ceil_div((long)n * D, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(F("c2_sbox")), F("c2_skey"),
                     I("f_idx"), I("f_cnt"), D, (C - 1) * R, R, n, p.ratio_h, p.ratio_w, reinterpret_cast<float4*>(d_boxes), d_scores, d_labels,
                     d_counts, reinterpret_cast<float4*>(F("det_resized")));

  // ---- mask head --------------------------------------------------------------------------------------------------------------------
  if (d_masks) {
    {
      const long total = (long)n * D * 196 * (DET_FPN_C / 4);
      hipLaunchKernelGGL(det_roi_align_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, py, reinterpret_cast<const float4*>(F("det_resized")),
                         d_counts, D, n, 14, 1, F("roi14"));
    }
    const float* mx = F("roi14");
    for (int i = 0; i < 4; ++i) {
      float* my = (i & 1) ? F("m_b") : F("m_a");
      rc = det_run_conv(d->mask_fcn[i], mx, n * D, 14, 14, 1, my, 1, nullptr, 1, s, SK);
      if (rc) return rc;
      mx = my;
    }
    rc = det_run_conv(d->mask_deconv, mx, n * D, 14, 14, 1, F("m_up"), 0, nullptr, 1, s, SK);
    if (rc) return rc;
    rc = det_run_conv(d->mask_logits, F("m_up"), n * D * 196 * 4, 1, 1, 0, F("mask_logits"), 0, nullptr, 0, s, SK);
    if (rc) return rc;
    ProfScope prof("det_mask_paste", 0.0, (double)n * D * H * W * 4.0, s);
    hipLaunchKernelGGL(det_mask_paste_kernel, dim3(ceil_div((long)H * W, 256), n * D), dim3(256), 0, s, F("mask_logits"), d->Cmask_s,
                       reinterpret_cast<const float4*>(d_boxes), d_labels, d_counts, D, H, W, d_masks);
  }
  MP_CHECK_HIP(hipGetLastError());
  d->ran = true;
  d->last = p;
  d->last_ws = d_ws;
  return MP_OK;
}

extern "C" int mp_detector_debug_tensor(const mp_detector* d, const char* what, const void** d_ptr, int64_t* shape4, int32_t* border,
                                        int64_t* row_stride, int64_t* n_elements) {
  MP_REQUIRE(d && what && d_ptr && shape4 && border && row_stride && n_elements, "mp_detector_debug_tensor: null pointer");
  MP_REQUIRE(d->ran, "mp_detector_debug_tensor: no forward has run yet");
  const DetPlan& p = d->last;
  const std::string w = what;
  auto it = p.off.find(w);
  MP_REQUIRE(it != p.off.end(), "mp_detector_debug_tensor: unknown tensor '%s'", what);
  *d_ptr = (const float*)d->last_ws + it->second;
  *border = 0;
  const int R = d->cfg.rpn_post_nms_top_n, D = d->cfg.box_detections_per_img;
  if (w.size() == 2 && w[0] == 'P' && w[1] >= '2' && w[1] <= '6') {
    const int l = w[1] - '2';
    shape4[0] = p.n; shape4[1] = p.fh[l]; shape4[2] = p.fw[l]; shape4[3] = DET_FPN_C;
    *border = 1;
    *row_stride = DET_FPN_C;
  } else if (w == "proposals") {
    shape4[0] = p.n; shape4[1] = R; shape4[2] = 4; shape4[3] = 1; *row_stride = 4;
  } else if (w == "proposal_scores") {
    shape4[0] = p.n; shape4[1] = R; shape4[2] = 1; shape4[3] = 1; *row_stride = 1;
  } else if (w == "proposal_counts" || w == "f_cnt") {
    shape4[0] = p.n; shape4[1] = 1; shape4[2] = 1; shape4[3] = 1; *row_stride = 1;
  } else if (w == "class_logits") {
    shape4[0] = (int64_t)p.n * R; shape4[1] = 5 * d->C; shape4[2] = 1; shape4[3] = 1; *row_stride = d->Cpred_s;
  } else if (w == "mask_logits") {
    shape4[0] = (int64_t)p.n * D * 196 * 4; shape4[1] = d->C; shape4[2] = 1; shape4[3] = 1; *row_stride = d->Cmask_s;
  } else if (w == "det_resized") {
    shape4[0] = p.n; shape4[1] = D; shape4[2] = 4; shape4[3] = 1; *row_stride = 4;
  } else if (w == "x0") {
    shape4[0] = p.n; shape4[1] = p.Hp; shape4[2] = p.Wp; shape4[3] = 4; *border = 3; *row_stride = 4;
  } else {
    set_error("mp_detector_debug_tensor: '%s' is an internal buffer without a published shape", what);
    return MP_ERR_INVALID;
  }
  if (*border) *n_elements = shape4[0] * (shape4[1] + 2 * *border) * (shape4[2] + 2 * *border) * shape4[3];
  else if (shape4[2] == 4) *n_elements = shape4[0] * shape4[1] * 4;          // box lists
  else if (*row_stride > 1) *n_elements = shape4[0] * *row_stride;            // row matrices with a padded row
  else *n_elements = shape4[0] * shape4[1];                                   // flat per-image lists / counters
  return MP_OK;
}
