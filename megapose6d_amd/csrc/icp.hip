// icp.hip -- on-device depth refiner (point-to-plane ICP) for the "-icp" model variant.
//
// Replaces the per-object CPU loop of /root/reference/src/megapose/inference/icp_refiner.py:195-262 (refine_poses) and
// :128-175 (icp_refinement): full-resolution depth render (our rasteriser), masks (refiner_utils.py:30-56, threshold 0.1 m),
// back-projection (getXYZ :98-125), target normals (get_normal :37-95), centroid pre-alignment (:157-162) and the ICP itself.
// The reference delegates the ICP to OpenCV-contrib `cv2.ppf_match_3d_ICP(100, tolerence=0.05, numLevels=4)` -- third-party code
// that is not under /root/reference ("parity unpinned").  This file implements a GPU-shaped algorithm of the same family with the
// same budget (4 levels x 25 iterations, tolerance 0.05 m, n_min_points = 1000, accept/reject rule on the residual):
//   * source points  = rendered depth back-projected at the mask pixels, target = measured depth (+ normals from central
//     differences of the back-projected measured points),
//   * data association is PROJECTIVE (transform the source point, project it with K, take the target pixel it lands on)
//     instead of OpenCV's kd-tree search -- fully parallel, no host involvement,
//   * each iteration accumulates the 6x6 point-to-plane normal equations per object with block reductions into per-block partial
//     slots that the one-thread Cholesky solve adds up in a fixed order (no float atomics: results are bit-reproducible); levels subsample the mask pixels (stride 8,4,2,1) and tighten the
//     rejection distance (0.20, 0.15, 0.10, 0.05 m).
// The CPU oracle of THIS algorithm is oracle/icp.py.  Roofline: latency-bound (200 tiny launches), negligible next to the CNN.
#include "common.h"

namespace mp {

constexpr int ICP_ACC_BLOCKS = 16, ICP_STAT_BLOCKS = 32;

struct IcpRow {        // per object state, device resident
  float T[12];         // current increment [R|t] (row-major 3x4) applied to camera-frame source points
  float acc_part[ICP_ACC_BLOCKS][29];   // per-block partials of: 21 upper-triangular JtJ, 6 Jtr, sum r^2, inlier count
  float stat_part[ICP_STAT_BLOCKS][8];  // per-block partials of: n_mask, (unused), centroid_tgt xyz, centroid_src xyz
  int status;          // 1 = running/ok, 0 = failed (too few points / singular / diverged)
  float residual;      // RMS point-to-plane residual of the last iteration
};

__device__ __forceinline__ bool mask_at(const float* dm, const float* dr, int idx, float delta_thresh) {
  const float m = dm[idx], r = dr[idx];
  // refiner_utils.py:45-51 (threshold mask; delta_thresh = +inf when the caller supplied its own masks: icp_refiner.py:249-250
  // then uses that mask alone, which the host has already applied to the measured depth) and :142-143 (0.2 < depth < 5)
  return m > 0.f && r > 0.f && fabsf(m - r) <= delta_thresh && m > 0.2f && m < 5.0f;
}

// normals of the measured depth images (one per frame): cross product of central differences of back-projected points
__global__ void icp_target_normals(const float* __restrict__ depth, const float* __restrict__ K, int H, int W, float* __restrict__ normals) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx % W;
  const float* d = depth + (size_t)b * H * W;
  const float* Kb = K + (size_t)b * 9;
  const float ifx = 1.0f / Kb[0], ify = 1.0f / Kb[4], cx = Kb[2], cy = Kb[5];
  float n[3] = {0.f, 0.f, 0.f};
  const int xm = max(x - 2, 0), xp = min(x + 2, W - 1), ym = max(y - 2, 0), yp = min(y + 2, H - 1);
  const float dl = d[y * W + xm], dr = d[y * W + xp], du = d[ym * W + x], dd = d[yp * W + x], dc = d[idx];
  if (dc > 0.f && dl > 0.f && dr > 0.f && du > 0.f && dd > 0.f) {
    const float ax = ((float)xp - cx) * dr * ifx - ((float)xm - cx) * dl * ifx, ay = ((float)y - cy) * (dr - dl) * ify, az = dr - dl;
    const float bx = ((float)x - cx) * (dd - du) * ifx, by = ((float)yp - cy) * dd * ify - ((float)ym - cy) * du * ify, bz = dd - du;
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    if (len > 0.f) {
      nx /= len; ny /= len; nz /= len;
      if (nz > 0.f) { nx = -nx; ny = -ny; nz = -nz; }  // face the camera
      n[0] = nx; n[1] = ny; n[2] = nz;
    }
  }
  float* o = normals + ((size_t)b * H * W + idx) * 3;
  o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// mask size + centroids of target / source points (icp_refiner.py:157-162)
__global__ __launch_bounds__(256) void icp_stats(const float* __restrict__ depth_meas, const int32_t* __restrict__ im_ids,
                                                 const float* __restrict__ depth_rend, const float* __restrict__ K, int H, int W,
                                                 float delta_thresh, IcpRow* __restrict__ rows) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  const float* dm = depth_meas + (size_t)im_ids[n] * H * W;
  const float* dr = depth_rend + (size_t)n * H * W;
  const float* Kn = K + (size_t)n * 9;
  const float ifx = 1.0f / Kn[0], ify = 1.0f / Kn[4], cx = Kn[2], cy = Kn[5];
  float v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * W; idx += gridDim.x * blockDim.x) {
    if (!mask_at(dm, dr, idx, delta_thresh)) continue;
    const float u = (float)(idx % W) - cx, w_ = (float)(idx / W) - cy;
    const float zt = dm[idx], zs = dr[idx];
    v[0] += 1.f;
    v[1] += u * zt * ifx; v[2] += w_ * zt * ify; v[3] += zt;
    v[4] += u * zs * ifx; v[5] += w_ * zs * ify; v[6] += zs;
  }
  for (int k = 0; k < 7; ++k) {
    const float s = block_sum(v[k], red);
    if (threadIdx.x == 0) rows[n].stat_part[blockIdx.x][k == 0 ? 0 : k + 1] = s;
  }
}

__global__ void icp_init(IcpRow* __restrict__ rows, int N, int n_min_points) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  IcpRow& r = rows[n];
  float stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < ICP_STAT_BLOCKS; ++b)   // fixed order
    for (int k = 0; k < 8; ++k) stats[k] += r.stat_part[b][k];
  const float cnt = stats[0];
  for (int k = 0; k < 12; ++k) r.T[k] = (k == 0 || k == 5 || k == 10) ? 1.f : 0.f;
  r.residual = -1.f;
  if (cnt < (float)n_min_points) { r.status = 0; return; }
  r.status = 1;
  r.T[3] = (stats[2] - stats[5]) / cnt;   // centroid_tgt - centroid_src
  r.T[7] = (stats[3] - stats[6]) / cnt;
  r.T[11] = (stats[4] - stats[7]) / cnt;
}

__global__ __launch_bounds__(256) void icp_accumulate(const float* __restrict__ depth_meas, const float* __restrict__ normals,
                                                      const int32_t* __restrict__ im_ids, const float* __restrict__ depth_rend,
                                                      const float* __restrict__ K, int H, int W, int stride, float d_max,
                                                      float delta_thresh, IcpRow* __restrict__ rows) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  if (rows[n].status == 0) return;
  const float* dm = depth_meas + (size_t)im_ids[n] * H * W;
  const float* nm = normals + (size_t)im_ids[n] * H * W * 3;
  const float* dr = depth_rend + (size_t)n * H * W;
  const float* Kn = K + (size_t)n * 9;
  const float fx = Kn[0], fy = Kn[4], cx = Kn[2], cy = Kn[5], ifx = 1.0f / fx, ify = 1.0f / fy;
  float T[12];
  for (int k = 0; k < 12; ++k) T[k] = rows[n].T[k];
  float a[29];
  for (int k = 0; k < 29; ++k) a[k] = 0.f;
  const int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hs * Ws; i += gridDim.x * blockDim.x) {
    const int y = (i / Ws) * stride, x = (i % Ws) * stride;
    const int idx = y * W + x;
    if (!mask_at(dm, dr, idx, delta_thresh)) continue;
    const float zs = dr[idx];
    const float px = ((float)x - cx) * zs * ifx, py = ((float)y - cy) * zs * ify;
    const float sx = T[0] * px + T[1] * py + T[2] * zs + T[3];
    const float sy = T[4] * px + T[5] * py + T[6] * zs + T[7];
    const float sz = T[8] * px + T[9] * py + T[10] * zs + T[11];
    if (!(sz > 0.05f)) continue;
    const int qx = (int)rintf(fx * sx / sz + cx), qy = (int)rintf(fy * sy / sz + cy);
    if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
    const int q = qy * W + qx;
    const float zt = dm[q];
    if (!(zt > 0.2f && zt < 5.0f)) continue;
    const float nx = nm[3 * q], ny = nm[3 * q + 1], nz = nm[3 * q + 2];
    if (nx == 0.f && ny == 0.f && nz == 0.f) continue;
    const float tx = ((float)qx - cx) * zt * ifx, ty = ((float)qy - cy) * zt * ify;
    const float dx = sx - tx, dy = sy - ty, dz = sz - zt;
    if (dx * dx + dy * dy + dz * dz > d_max * d_max) continue;
    const float r = nx * dx + ny * dy + nz * dz;
    const float J[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx, nx, ny, nz};
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int c = p; c < 6; ++c) a[k++] += J[p] * J[c];
#pragma unroll
    for (int p = 0; p < 6; ++p) a[21 + p] += J[p] * r;
    a[27] += r * r;
    a[28] += 1.f;
  }
  for (int k = 0; k < 29; ++k) {
    const float s = block_sum(a[k], red);
    if (threadIdx.x == 0) rows[n].acc_part[blockIdx.x][k] = s;
  }
}

// one thread per object: solve (JtJ) x = -Jtr by Cholesky, compose the increment, reset the accumulators
__global__ void icp_solve(IcpRow* __restrict__ rows, int N, float min_inliers) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  IcpRow& r = rows[n];
  if (r.status == 0) return;
  float acc[29];
  for (int q = 0; q < 29; ++q) acc[q] = 0.f;
  for (int blk = 0; blk < ICP_ACC_BLOCKS; ++blk)   // fixed order
    for (int q = 0; q < 29; ++q) acc[q] += r.acc_part[blk][q];
  double A[6][6], b[6];
  int k = 0;
  for (int p = 0; p < 6; ++p)
    for (int c = p; c < 6; ++c) { A[p][c] = acc[k]; A[c][p] = acc[k]; ++k; }
  for (int p = 0; p < 6; ++p) b[p] = -(double)acc[21 + p];
  const float cnt = acc[28];
  const float res = cnt > 0.f ? sqrtf(acc[27] / cnt) : -1.f;
  if (cnt < min_inliers) { r.status = 0; r.residual = -1.f; return; }
  r.residual = res;
  for (int p = 0; p < 6; ++p) A[p][p] += 1e-9 * (double)cnt;  // Levenberg damping against rank deficiency (planar / symmetric views)
  double L[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
      if (i == j) {
        if (!(s > 0.0)) { r.status = 0; return; }
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double yv[6], x[6];
  for (int i = 0; i < 6; ++i) { double s = b[i]; for (int q = 0; q < i; ++q) s -= L[i][q] * yv[q]; yv[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; --i) { double s = yv[i]; for (int q = i + 1; q < 6; ++q) s -= L[q][i] * x[q]; x[i] = s / L[i][i]; }
  // Rodrigues for omega = x[0:3]
  const double wx = x[0], wy = x[1], wz = x[2];
  const double th = sqrt(wx * wx + wy * wy + wz * wz);
  double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > 1e-12) {
    const double kx = wx / th, ky = wy / th, kz = wz / th, c = cos(th), s = sin(th), v = 1 - c;
    Rm[0] = c + kx * kx * v;      Rm[1] = kx * ky * v - kz * s; Rm[2] = kx * kz * v + ky * s;
    Rm[3] = ky * kx * v + kz * s; Rm[4] = c + ky * ky * v;      Rm[5] = ky * kz * v - kx * s;
    Rm[6] = kz * kx * v - ky * s; Rm[7] = kz * ky * v + kx * s; Rm[8] = c + kz * kz * v;
  }
  double Tn[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) Tn[i * 4 + j] = Rm[i * 3] * r.T[j] + Rm[i * 3 + 1] * r.T[4 + j] + Rm[i * 3 + 2] * r.T[8 + j];
    Tn[i * 4 + 3] += x[3 + i];
  }
  bool finite = true;
  for (int q = 0; q < 12; ++q) finite = finite && isfinite(Tn[q]);
  if (!finite) { r.status = 0; return; }
  for (int q = 0; q < 12; ++q) r.T[q] = (float)Tn[q];
}

// TCO_refined = T_inc @ TCO when the refinement succeeded and residual <= tolerance, else the input pose (icp_refiner.py:172-175, :257-258)
__global__ void icp_finalize(const IcpRow* __restrict__ rows, const float* __restrict__ TCO, int N, float tolerance, float* __restrict__ TCO_out,
                             int32_t* __restrict__ retval, float* __restrict__ residual) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const IcpRow& r = rows[n];
  const float* T = TCO + (size_t)n * 16;
  float* O = TCO_out + (size_t)n * 16;
  const bool ok = r.status == 1 && r.residual >= 0.f && r.residual <= tolerance;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      float v = T[i * 4 + j];
      if (ok) v = r.T[i * 4] * T[j] + r.T[i * 4 + 1] * T[4 + j] + r.T[i * 4 + 2] * T[8 + j] + (j == 3 ? r.T[i * 4 + 3] : 0.f);
      O[i * 4 + j] = v;
    }
  O[12] = T[12]; O[13] = T[13]; O[14] = T[14]; O[15] = T[15];
  if (retval) retval[n] = ok ? 0 : -1;
  if (residual) residual[n] = r.residual;
}

}  // namespace mp

using namespace mp;

extern "C" size_t mp_icp_workspace_bytes(int n_images, int n_rows, int H, int W) {
  return (size_t)n_images * H * W * 3 * sizeof(float) + (size_t)n_rows * sizeof(IcpRow) + 256;
}

extern "C" int mp_icp_refine(const float* d_depth_meas, int n_images, const int32_t* d_im_ids, const float* d_depth_rend,
                             const float* d_K_images, const float* d_K_rows, const float* d_TCO, int n_rows, int H, int W,
                             int n_iterations, int n_levels, float tolerance, int n_min_points, int user_masks, float* d_TCO_out,
                             int32_t* d_retval, float* d_residual, void* d_ws, size_t ws_bytes, mp_stream stream) {
  MP_REQUIRE(d_depth_meas && d_im_ids && d_depth_rend && d_K_images && d_K_rows && d_TCO && d_TCO_out && d_ws, "mp_icp_refine: null pointer");
  MP_REQUIRE(n_images > 0 && n_rows >= 0 && H > 0 && W > 0 && n_iterations > 0 && n_levels >= 1 && n_levels <= 8, "mp_icp_refine: bad size");
  MP_REQUIRE(ws_bytes >= mp_icp_workspace_bytes(n_images, n_rows, H, W), "mp_icp_refine: workspace too small");
  if (n_rows == 0) return MP_OK;
  MP_REQUIRE(n_rows <= 65535 && n_images <= 65535, "mp_icp_refine: at most 65535 rows / images");
  hipStream_t s = (hipStream_t)stream;
  float* normals = (float*)d_ws;
  IcpRow* rows = (IcpRow*)(((uintptr_t)(normals + (size_t)n_images * H * W * 3) + 255) & ~(uintptr_t)255);
  MP_CHECK_HIP(hipMemsetAsync(rows, 0, (size_t)n_rows * sizeof(IcpRow), s));
  ProfScope prof("icp_refine", 0.0, (double)n_rows * H * W * 8.0 * n_iterations, s);
  hipLaunchKernelGGL(icp_target_normals, dim3(ceil_div((long)H * W, 256), n_images), dim3(256), 0, s, d_depth_meas, d_K_images, H, W, normals);
  // user_masks: the caller's masks were already applied to d_depth_meas; the |measured - rendered| <= 0.1 m test is then skipped
  const float delta_thresh = user_masks ? INFINITY : 0.1f;
  hipLaunchKernelGGL(icp_stats, dim3(ICP_STAT_BLOCKS, n_rows), dim3(256), 0, s, d_depth_meas, d_im_ids, d_depth_rend, d_K_rows, H, W, delta_thresh, rows);
  hipLaunchKernelGGL(icp_init, dim3(ceil_div(n_rows, 64)), dim3(64), 0, s, rows, n_rows, n_min_points);
  const int per_level = ceil_div(n_iterations, n_levels);
  for (int l = 0; l < n_levels; ++l) {
    const int stride = 1 << (n_levels - 1 - l);
    const float d_max = tolerance * (float)(n_levels - l);  // 0.20, 0.15, 0.10, 0.05 for the reference's (0.05, 4 levels)
    for (int it = 0; it < per_level; ++it) {
      hipLaunchKernelGGL(icp_accumulate, dim3(ICP_ACC_BLOCKS, n_rows), dim3(256), 0, s, d_depth_meas, normals, d_im_ids, d_depth_rend, d_K_rows, H, W,
                         stride, d_max, delta_thresh, rows);
      hipLaunchKernelGGL(icp_solve, dim3(ceil_div(n_rows, 64)), dim3(64), 0, s, rows, n_rows, 50.0f);
    }
  }
  // (the reported residual is the RMS point-to-plane error measured by the last accumulation, i.e. before the last increment)
  hipLaunchKernelGGL(icp_finalize, dim3(ceil_div(n_rows, 64)), dim3(64), 0, s, rows, d_TCO, n_rows, tolerance, d_TCO_out, d_retval, d_residual);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
