// icp_nn.hip -- the reference's depth refiner, step for step, on the device (nearest-neighbour association).
//
// Replaces /root/reference/src/megapose/inference/icp_refiner.py:37-175: get_normal (:37-95: hole filling, Gaussian sigma 2, gradient
// with spacing 2, cross product of the tangents built from an int16-truncated pixel-offset table), getXYZ (:98-125), the masks and the
// >= 1000-point rule (:140-155), the centroid pre-shift (:157-162) and `cv2.ppf_match_3d_ICP(100, tolerence=0.05, rejectionScale=2.5,
// numLevels=4).registerModelToScene` (:166-169).  OpenCV-contrib is third-party code absent from /root/reference ("parity unpinned");
// the algorithm implemented here is the one oracle/icp_opencv.py restates from its published source (surface_matching/src/icp.cpp):
//   mean / scale normalisation; 4 levels, every 8th / 4th / 2nd / every point, at most 25 / 33 / 50 / 100 iterations, relative-change
//   stop at tolerance * (level + 1)^2; per iteration: exact nearest neighbour of every (moved) model point in the sub-sampled scene,
//   robust rejection at median + 2.5 * 1.48257968 * MAD of the squared distances (LOWER medians), "picky" one-to-one filtering (a scene
//   point keeps its closest model point), linearised point-to-plane least squares for the FULL level transform from the level's start
//   points, residual = Frobenius norm of the matched 6-d rows / number of model points.
// cv2.inpaint is replaced (there as here) by an onion-peel fill: a hole pixel gets the mean of its valid 8-neighbours, ring by ring.
// Only the rings that can reach a used normal (Gaussian radius 8 + gradient 1) are computed: 10 rings give the values of the full fill.
//
// Structure: image-sized kernels for the per-frame / per-object preparation (fill rings, separable Gaussian, normals + back-projection,
// ordered compaction of the mask pixels), then the multi-level ICP as a fixed train of (search, step) kernel pairs -- the worst-case
// iteration count is enqueued up front and every object keeps its own level / iteration / stop state in device memory, so the
// data-dependent loop never touches the host and a finished object just turns its remaining launches into no-ops.  search: exact
// brute-force nearest neighbours in float64 over the whole chip (scene tiled through LDS); step (one workgroup per object): radix-select
// medians, 64-bit atomicMin one-to-one filter, float64 normal equations reduced in a fixed order (deterministic), Cholesky solve.
// Arithmetic follows the restatement (float32 where numpy / OpenCV hold float32, float64 where they compute in double).
// Roofline: latency / VALU-bound tail work of config 5 (once per detection after the CNN stages), negligible next to them.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace mp {

constexpr int NN_THREADS = 1024;
constexpr int FILL_RINGS = 10;
constexpr int IDX_BITS = 18;           // scene index bits of the nearest-neighbour merge key -> at most 2^18 points per object
constexpr int GAUSS_RADIUS = 8;        // scipy.ndimage.gaussian_filter(sigma=2): truncate 4.0 -> radius int(4 * 2 + 0.5) = 8

// ---- preparation ------------------------------------------------------------------------------------------------------------------
// one ring of the hole fill: an invalid pixel with >= 1 valid 8-neighbour becomes their mean (sum in double -> float32, / count in float32)
__global__ void icpnn_fill_ring(const float* __restrict__ d_in, const unsigned char* __restrict__ v_in, float* __restrict__ d_out,
                                unsigned char* __restrict__ v_out, int H, int W) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const size_t o = (size_t)b * H * W;
  const int y = idx / W, x = idx % W;
  float val = d_in[o + idx];
  unsigned char ok = v_in[o + idx];
  if (!ok) {
    double s = 0.0;
    int c = 0;
    // (ndimage.convolve with the all-ones 3x3 kernel, mode='constant': the flipped kernel walks the neighbourhood from the
    //  bottom-right to the top-left; every term is a float32, the running sum a double)
    for (int dy = 1; dy >= -1; --dy)
      for (int dx = 1; dx >= -1; --dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        if (v_in[o + yy * W + xx]) { s += (double)d_in[o + yy * W + xx]; ++c; }
      }
    if (c > 0) { val = (float)s / (float)c; ok = 1; }
  }
  d_out[o + idx] = val;
  v_out[o + idx] = ok;
}

__global__ void icpnn_init_valid(const float* __restrict__ d_in, float* __restrict__ d_out, unsigned char* __restrict__ valid, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = d_in[i];
  if (!(v == v)) v = 0.f;                      // np.nan_to_num
  if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  d_out[i] = v;
  valid[i] = v != 0.f;
}

struct GaussW { double w[GAUSS_RADIUS + 1]; };   // w[k] = weight at distance k (normalised)

__device__ __forceinline__ int reflect_idx(int i, int n) {   // scipy 'reflect': d c b a | a b c d | d c b a
  if (n == 1) return 0;
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

// correlate1d with symmetric weights along one axis (ni_filters.c: tmp = x[l] * w0; for ll = -r .. -1: tmp += (x[l+ll] + x[l-ll]) * w[-ll]),
// double accumulation, float32 output after EACH axis (gaussian_filter filters axis 0 first, then axis 1)
__global__ void icpnn_gauss_axis(const float* __restrict__ in, float* __restrict__ out, int H, int W, int axis, GaussW g) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const size_t o = (size_t)b * H * W;
  const int y = idx / W, x = idx % W;
  const int n = axis == 0 ? H : W, l = axis == 0 ? y : x;
  auto at = [&](int i) -> double {
    const int r = reflect_idx(i, n);
    return (double)(axis == 0 ? in[o + (size_t)r * W + x] : in[o + (size_t)y * W + r]);
  };
  double tmp = at(l) * g.w[0];
  for (int ll = -GAUSS_RADIUS; ll < 0; ++ll) tmp += (at(l + ll) + at(l - ll)) * g.w[-ll];
  out[o + idx] = (float)tmp;
}

// np.gradient(d, 2, edge_order=2) along one axis at position i of a line of n float32 values (float32 arithmetic)
__device__ __forceinline__ float grad2(float fm, float f0, float fp, float f1, float f2, int i, int n) {
  // interior: (f[i+1] - f[i-1]) / 4 ; edges (edge_order 2, dx = 2): a = -1.5/dx, b = 2/dx, c = -0.5/dx
  if (n < 3) return 0.f;
  if (i > 0 && i < n - 1) return (fp - fm) / 4.0f;
  const float a = -0.75f, b = 1.0f, c = -0.25f;
  if (i == 0) return (a * f0 + b * f1) + c * f2;          // f1 = f[1], f2 = f[2]
  return (-c * f2 + (-b) * f1) + (-a) * f0;               // i == n-1: (0.5/dx) f[n-3] - (2/dx) f[n-2] + (1.5/dx) f[n-1]; f1 = f[n-2], f2 = f[n-3]
}

// back-projection (getXYZ, raw depth) + normals (get_normal, filtered depth) -> pts [b][H*W][6] float32
__global__ void icpnn_points(const float* __restrict__ depth_raw, const float* __restrict__ d, const float* __restrict__ K,
                             const int32_t* __restrict__ k_ids, int H, int W, float* __restrict__ pts) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const size_t o = (size_t)b * H * W;
  const int y = idx / W, x = idx % W;
  const float* Kb = K + (size_t)(k_ids ? k_ids[b] : b) * 9;
  const float cxf = Kb[2], cyf = Kb[5];
  // the reference passes numpy float32 SCALARS of K (icp_refiner.py:135-150): 1 / fx is a float32 division, and every product below that
  // involves it is float32 arithmetic (int16 array x float32 scalar -> float32 under numpy 1.x value-based casting and NEP 50 alike)
  const float constant_x = 1.0f / Kb[0], constant_y = 1.0f / Kb[4];
  // uv table: int16(arange - c): float -> int16 truncates toward zero (icp_refiner.py:62-66, :112-116); the subtraction is float64
  // (np.arange is int64, cx a python float)
  const int u1 = (int)(short)(int)((double)x - (double)cxf), u0 = (int)(short)(int)((double)y - (double)cyf);
  const float* dl = d + o;
  const float f0 = dl[idx];
  // gradient along axis 0 (rows) and axis 1 (columns)
  float g0, g1;
  {
    // (e1 / e2 = the two values next to an EDGE pixel, read only there: an interior pixel one step from the border must not reach past it)
    const bool ey = y == 0 || y == H - 1, ex = x == 0 || x == W - 1;
    const int sy = y == 0 ? W : -W, sx = x == 0 ? 1 : -1;
    const float fm = y > 0 ? dl[idx - W] : 0.f, fp = y < H - 1 ? dl[idx + W] : 0.f;
    const float e1 = ey ? dl[idx + sy] : 0.f, e2 = ey ? dl[idx + 2 * sy] : 0.f;
    g0 = grad2(fm, f0, fp, e1, e2, y, H);
    const float gm = x > 0 ? dl[idx - 1] : 0.f, gp = x < W - 1 ? dl[idx + 1] : 0.f;
    const float h1 = ex ? dl[idx + sx] : 0.f, h2 = ex ? dl[idx + 2 * sx] : 0.f;
    g1 = grad2(gm, f0, gp, h1, h2, x, W);
  }
  // v_y = [uv1 * cx * dig0, d * cy + (uv0 * cy) * dig0, dig0], v_x = [d * cx + uv1 * cx * dig1, uv0 * cy * dig1, dig1], evaluated left to
  // right in float32, stored into float64 arrays; the cross product and its normalisation are float64
  const float u1f = (float)u1, u0f = (float)u0;
  const double vy0 = (double)((u1f * constant_x) * g0), vy1 = (double)(f0 * constant_y + (u0f * constant_y) * g0), vy2 = (double)g0;
  const double vx0 = (double)(f0 * constant_x + (u1f * constant_x) * g1), vx1 = (double)((u0f * constant_y) * g1), vx2 = (double)g1;
  double c0 = vx1 * vy2 - vx2 * vy1, c1 = vx2 * vy0 - vx0 * vy2, c2 = vx0 * vy1 - vx1 * vy0;   // np.cross(v_x, v_y)
  double nrm = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
  if (nrm == 0.0) nrm = 1.0;
  c0 /= nrm; c1 /= nrm; c2 /= nrm;
  if (!(c0 == c0)) c0 = 0.0;
  if (!(c1 == c1)) c1 = 0.0;
  if (!(c2 == c2)) c2 = 0.0;
  float* p = pts + (o + idx) * 6;
  // getXYZ: uv * depth * 1 / fx, left to right: ((uv * depth) * 1) / fx -- int16 * float32 -> float32 (numpy promotion), then / float64
  // (and `/ fx` with fx a python float keeps float32: the division is a float32 one)
  const float xd = (float)u1 * depth_raw[o + idx], yd = (float)u0 * depth_raw[o + idx];
  p[0] = xd / Kb[0];
  p[1] = yd / Kb[4];
  p[2] = depth_raw[o + idx];
  p[3] = (float)c0; p[4] = (float)c1; p[5] = (float)c2;
}

// ---- per-object state ----------------------------------------------------------------------------------------------------------------
struct NnRow {
  int n;               // model points (rendered depth at the mask pixels where it is > 0)
  int m;               // scene points (measured depth at the mask pixels)
  int status;          // 1 ok, 0 rejected (too few points / degenerate)
  int iters[8];        // iterations run per level (telemetry)
  // loop state of the multi-level ICP (one (search, step) kernel pair per iteration; a finished object makes its launches no-ops)
  int active;          // 1 while this object still iterates
  int level, it, nl, ml, max_it;
  double tol_p;
  double fval_old, fval_perc, fval_min;
  double pose_x[16];   // the current level's transform (from the level's start points)
  double scale, mean_avg[3];
  double shift[3];     // centroid pre-shift (float32 values held in doubles)
  double pose[16];     // accumulated over the finished levels (normalised frame); after the last level: the result in metres
  double residual;
};

// ordered compaction (row-major, like boolean indexing): scene points = measured depth at the mask pixels inside the depth range
// (icp_refiner.py:142-146), model points = rendered depth at those pixels where it is > 0 (:150).  mask = the threshold mask
// |measured - rendered| <= 0.1 m, both > 0 (refiner_utils.py:45-51) -- or the caller's per-frame mask (:249-250; it only SELECTS points,
// the normals still come from the whole frame).  One workgroup per object.
__global__ __launch_bounds__(NN_THREADS) void icpnn_compact(const float* __restrict__ depth_meas, const int32_t* __restrict__ im_ids,
                                                            const float* __restrict__ depth_rend, const float* __restrict__ pts_meas,
                                                            const float* __restrict__ pts_rend, int H, int W, const unsigned char* __restrict__ masks, int cap,
                                                            int n_min_points, float* __restrict__ src, float* __restrict__ dst,
                                                            NnRow* __restrict__ rows) {
  __shared__ int wave_sum[2][NN_THREADS / 64];
  __shared__ int base_s[2];
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* dm = depth_meas + (size_t)im_ids[n] * H * W;
  const float* dr = depth_rend + (size_t)n * H * W;
  const unsigned char* mk = masks ? masks + (size_t)im_ids[n] * H * W : nullptr;
  const float* pm = pts_meas + (size_t)im_ids[n] * H * W * 6;
  const float* pr = pts_rend + (size_t)n * H * W * 6;
  float* so = src + (size_t)n * cap * 6;
  float* dq = dst + (size_t)n * cap * 6;
  if (tid < 2) base_s[tid] = 0;
  __syncthreads();
  for (int start = 0; start < H * W; start += NN_THREADS) {
    const int idx = start + tid;
    bool in_t = false, in_s = false;
    if (idx < H * W) {
      const float m = dm[idx], r = dr[idx];
      const bool mask = mk ? (mk[idx] != 0) : (m > 0.f && r > 0.f && !(fabsf(m - r) > 0.1f));
      in_t = mask && m > 0.2f && m < 5.0f;
      in_s = in_t && r > 0.f;
    }
    const unsigned long long bt = __ballot(in_t), bs = __ballot(in_s);
    const unsigned long long lower = (1ull << lane) - 1ull;
    if (lane == 0) { wave_sum[0][wave] = __popcll(bt); wave_sum[1][wave] = __popcll(bs); }
    __syncthreads();
    int off_t = base_s[0], off_s = base_s[1];
    for (int w = 0; w < wave; ++w) { off_t += wave_sum[0][w]; off_s += wave_sum[1][w]; }
    const int pos_t = off_t + __popcll(bt & lower), pos_s = off_s + __popcll(bs & lower);
    if (in_t && pos_t < cap)
      for (int k = 0; k < 6; ++k) dq[(size_t)pos_t * 6 + k] = pm[(size_t)idx * 6 + k];
    if (in_s && pos_s < cap)
      for (int k = 0; k < 6; ++k) so[(size_t)pos_s * 6 + k] = pr[(size_t)idx * 6 + k];
    __syncthreads();
    if (tid < 2) {
      int t = 0;
      for (int w = 0; w < NN_THREADS / 64; ++w) t += wave_sum[tid][w];
      base_s[tid] += t;
    }
    __syncthreads();
  }
  if (tid == 0) {
    NnRow& r = rows[n];
    r.m = base_s[0];
    r.n = base_s[1];
    r.status = (r.n >= n_min_points && r.m >= n_min_points && r.n <= cap && r.m <= cap) ? 1 : 0;   // icp_refiner.py:152-155
    r.residual = -1.0;
    r.active = 0;
    for (int k = 0; k < 8; ++k) r.iters[k] = 0;
  }
}

// ---- the ICP ---------------------------------------------------------------------------------------------------------------------------
struct NnScratch {   // per object, global memory (cap points each)
  float* src;        // [cap][6] model points (shifted, then normalised)
  float* dst;        // [cap][6] scene points (normalised)
  float* src_t;      // [cap][6] level start points (sub-sampled)
  float* moved;      // [cap][6]
  float* dst_s;      // [cap][6] sub-sampled scene
  float* d2;         // [cap]
  float* dev;        // [cap]
  int* nn;           // [cap]
  unsigned long long* nnkey;   // [cap] nearest-neighbour merge across scene segments: (distance bits | scene index) per model point
  unsigned long long* key;   // [cap] picky filter: (d2 bits << 32 | model index) per scene point
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {   // deterministic: fixed tree inside the wave, fixed wave order
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < NN_THREADS / 64; ++w) s += red[w];
  __syncthreads();
  return s;
}

// lower median (element (n-1)/2 of the sorted array) of n non-negative float32 values: 4-pass radix select on the bit patterns
__device__ float block_median(const float* __restrict__ a, int n, unsigned* hist, unsigned* sh) {
  unsigned prefix = 0;
  int rank = (n - 1) / 2;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned b = __float_as_uint(a[i]);
      if ((b & mask) == (prefix & mask)) atomicAdd(&hist[(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int r = rank;
      unsigned bin = 0;
      for (; bin < 256; ++bin) {
        if (r < (int)hist[bin]) break;
        r -= (int)hist[bin];
      }
      sh[0] = bin;
      sh[1] = (unsigned)r;
    }
    __syncthreads();
    prefix |= sh[0] << shift;
    rank = (int)sh[1];
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

__device__ __forceinline__ void transform_point(const float* p, const double* T, float* o) {   // transformPCPose (float32 cloud)
  const double x = p[0], y = p[1], z = p[2];
  o[0] = (float)((x * T[0] + y * T[1] + z * T[2]) + T[3]);
  o[1] = (float)((x * T[4] + y * T[5] + z * T[6]) + T[7]);
  o[2] = (float)((x * T[8] + y * T[9] + z * T[10]) + T[11]);
  const double nx = p[3], ny = p[4], nz = p[5];
  double a = nx * T[0] + ny * T[1] + nz * T[2], b = nx * T[4] + ny * T[5] + nz * T[6], c = nx * T[8] + ny * T[9] + nz * T[10];
  const double ln = sqrt(a * a + b * b + c * c);
  if (ln > 1e-12) { a /= ln; b /= ln; c /= ln; }
  o[3] = (float)a; o[4] = (float)b; o[5] = (float)c;
}

struct NnPtrs { float *src, *dst, *src_t, *moved, *dst_s, *d2, *dev; int* nn; unsigned long long *key, *nnkey; };

__device__ __forceinline__ NnPtrs row_ptrs(const NnScratch& sc, int row, int cap) {
  NnPtrs p;
  p.src = sc.src + (size_t)row * cap * 6; p.dst = sc.dst + (size_t)row * cap * 6; p.src_t = sc.src_t + (size_t)row * cap * 6;
  p.moved = sc.moved + (size_t)row * cap * 6; p.dst_s = sc.dst_s + (size_t)row * cap * 6;
  p.d2 = sc.d2 + (size_t)row * cap; p.dev = sc.dev + (size_t)row * cap; p.nn = sc.nn + (size_t)row * cap; p.key = sc.key + (size_t)row * cap; p.nnkey = sc.nnkey + (size_t)row * cap;
  return p;
}

// start of a pyramid level (whole workgroup; `pose` = the transform accumulated so far, in LDS): sub-sample the moved model and the scene
// (samplePCUniform: every step-th point), reset the loop variables.  Thread 0 writes the scalars of the level into the row.
__device__ void level_start(NnRow& R, const NnPtrs& P, const double* pose, int level, int iterations, float tolerance) {
  const int tid = threadIdx.x, n = R.n, m = R.m;
  const double div = (double)(1 << level);
  const int num_samples = (int)rint((double)n / div);               // cvRound
  int step = num_samples > 0 ? (int)rint((double)n / num_samples) : 1;
  if (step < 1) step = 1;
  const int nl = (n + step - 1) / step, ml = (m + step - 1) / step;   // len(x[::step]) of the model / the scene
  for (int i = tid; i < nl; i += NN_THREADS) {
    float o[6];
    transform_point(P.src + (size_t)i * step * 6, pose, o);
    for (int k = 0; k < 6; ++k) { P.src_t[(size_t)i * 6 + k] = o[k]; P.moved[(size_t)i * 6 + k] = o[k]; }
    P.nnkey[i] = ~0ull;
  }
  for (int i = tid; i < ml; i += NN_THREADS)
    for (int k = 0; k < 6; ++k) P.dst_s[(size_t)i * 6 + k] = P.dst[(size_t)i * step * 6 + k];
  if (tid == 0) {
    R.level = level; R.it = 0; R.nl = nl; R.ml = ml;
    R.max_it = (int)rint((double)iterations / (level + 1));
    R.tol_p = (double)tolerance * (level + 1) * (level + 1);
    R.fval_old = 9999999999.0; R.fval_perc = 0.0; R.fval_min = 9999999999.0;
    for (int k = 0; k < 16; ++k) R.pose_x[k] = (k % 5 == 0) ? 1.0 : 0.0;
  }
}

// centroid pre-shift, OpenCV's mean / scale normalisation, start of the coarsest level.  One workgroup per object.
__global__ __launch_bounds__(NN_THREADS) void icpnn_begin(NnScratch sc, int cap, NnRow* __restrict__ rows, int iterations, float tolerance,
                                                          int num_levels) {
  __shared__ double red[NN_THREADS / 64];
  __shared__ double pose_s[16];
  __shared__ float stage[6 * NN_THREADS];
  __shared__ float msum[6];
  const int row = blockIdx.x, tid = threadIdx.x;
  NnRow& R = rows[row];
  if (R.status == 0) return;
  const int n = R.n, m = R.m;
  const NnPtrs P = row_ptrs(sc, row, cap);
  float *src = P.src, *dst = P.dst;
  // ---- centroid pre-shift (icp_refiner.py:157-162): src += mean(dst) - mean(src).  np.mean over axis 0 of a float32 [n, 3] view is a
  // SEQUENTIAL float32 accumulation (numpy sums pairwise only along the fast axis), then a float32 division by the count: six lanes walk
  // the six columns in order, the rows staged through LDS by the whole workgroup -------------------------------------------------------------
  {
    float acc = 0.f;
    const int longest = n > m ? n : m;
    for (int c0 = 0; c0 < longest; c0 += NN_THREADS) {
      const int i = c0 + tid;
      for (int k = 0; k < 3; ++k) {
        stage[k * NN_THREADS + tid] = i < m ? dst[(size_t)i * 6 + k] : 0.f;
        stage[(3 + k) * NN_THREADS + tid] = i < n ? src[(size_t)i * 6 + k] : 0.f;
      }
      __syncthreads();
      if (tid < 6) {
        const int cnt = min(NN_THREADS, (tid < 3 ? m : n) - c0);
        const float* col = stage + tid * NN_THREADS;
        for (int q = 0; q < cnt; ++q) acc += col[q];
      }
      __syncthreads();
    }
    if (tid < 6) msum[tid] = acc;
    __syncthreads();
  }
  float shift[3];
  for (int k = 0; k < 3; ++k) shift[k] = msum[k] / (float)m - msum[3 + k] / (float)n;
  for (int i = tid; i < n; i += NN_THREADS)
    for (int k = 0; k < 3; ++k) src[(size_t)i * 6 + k] += shift[k];
  __syncthreads();
  // ---- normalisation: mean_avg (float64 means of the float32 clouds), scale = n / mean of the summed norms -----------------------------
  double mean_avg[3];
  for (int k = 0; k < 3; ++k) {
    double a = 0.0, b = 0.0;
    for (int i = tid; i < n; i += NN_THREADS) a += (double)src[(size_t)i * 6 + k];
    for (int i = tid; i < m; i += NN_THREADS) b += (double)dst[(size_t)i * 6 + k];
    const double sa = block_sum_d(a, red), sb = block_sum_d(b, red);
    mean_avg[k] = 0.5 * (sa / n + sb / m);
  }
  for (int i = tid; i < n; i += NN_THREADS)
    for (int k = 0; k < 3; ++k) src[(size_t)i * 6 + k] = (float)((double)src[(size_t)i * 6 + k] - mean_avg[k]);
  for (int i = tid; i < m; i += NN_THREADS)
    for (int k = 0; k < 3; ++k) dst[(size_t)i * 6 + k] = (float)((double)dst[(size_t)i * 6 + k] - mean_avg[k]);
  __syncthreads();
  double ds = 0.0, dd = 0.0;
  for (int i = tid; i < n; i += NN_THREADS) {
    const double a0 = src[(size_t)i * 6], a1 = src[(size_t)i * 6 + 1], a2 = src[(size_t)i * 6 + 2];
    ds += sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  }
  for (int i = tid; i < m; i += NN_THREADS) {
    const double b0 = dst[(size_t)i * 6], b1 = dst[(size_t)i * 6 + 1], b2 = dst[(size_t)i * 6 + 2];
    dd += sqrt(b0 * b0 + b1 * b1 + b2 * b2);
  }
  ds = block_sum_d(ds, red);
  dd = block_sum_d(dd, red);
  const double scale = (double)n / ((ds + dd) * 0.5);
  const float scale_f = (float)scale;
  for (int i = tid; i < n; i += NN_THREADS)
    for (int k = 0; k < 3; ++k) src[(size_t)i * 6 + k] *= scale_f;
  for (int i = tid; i < m; i += NN_THREADS)
    for (int k = 0; k < 3; ++k) dst[(size_t)i * 6 + k] *= scale_f;
  if (tid < 16) pose_s[tid] = (tid % 5 == 0) ? 1.0 : 0.0;
  if (tid == 0) {
    for (int k = 0; k < 3; ++k) { R.shift[k] = (double)shift[k]; R.mean_avg[k] = mean_avg[k]; }
    R.scale = scale;
    for (int k = 0; k < 16; ++k) R.pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
    R.residual = 0.0;
    R.active = 1;
  }
  __syncthreads();
  level_start(R, P, pose_s, num_levels - 1, iterations, tolerance);
}

// exact nearest neighbour (float64 distances of the float32 clouds, like a kd-tree's exact answer) of the moved model points in the level's
// scene.  The (model chunk of 512) x (scene segment of 1024) work items of an object are spread over SEARCH_WGS workgroups so that the
// critical path does not grow with the scene; a workgroup keeps its best per model point in registers (exact float64 comparison, ties ->
// the lowest scene index) and merges across segments with one 64-bit atomicMin per point on (top 46 bits of the float64 distance | 18-bit
// scene index).  icpnn_step re-evaluates the exact distance to the winner, so the 34-bit mantissa only decides between candidates of
// DIFFERENT segments that agree to 6e-11 relative.  Objects that are finished exit at once.
constexpr int SEARCH_THREADS = 256, SEARCH_PER = 2, SEARCH_TILE = 1024, SEARCH_SEG = 1024, SEARCH_WGS = 256;
__global__ __launch_bounds__(SEARCH_THREADS) void icpnn_search(NnScratch sc, int cap, const NnRow* __restrict__ rows) {
  __shared__ double tile[SEARCH_TILE * 3];
  const int row = blockIdx.y, tid = threadIdx.x;
  const NnRow& R = rows[row];
  if (R.status == 0 || R.active == 0) return;
  const int nl = R.nl, ml = R.ml;
  const int chunks = (nl + SEARCH_THREADS * SEARCH_PER - 1) / (SEARCH_THREADS * SEARCH_PER), segs = (ml + SEARCH_SEG - 1) / SEARCH_SEG;
  const float* moved = sc.moved + (size_t)row * cap * 6;
  const float* dst_s = sc.dst_s + (size_t)row * cap * 6;
  unsigned long long* nnkey = sc.nnkey + (size_t)row * cap;
  for (int item = blockIdx.x; item < chunks * segs; item += gridDim.x) {
    const int base = (item / segs) * SEARCH_THREADS * SEARCH_PER;
    const int s0 = (item % segs) * SEARCH_SEG, s1 = min(ml, s0 + SEARCH_SEG);
    double px[SEARCH_PER], py[SEARCH_PER], pz[SEARCH_PER], best[SEARCH_PER];
    int bi[SEARCH_PER];
#pragma unroll
    for (int q = 0; q < SEARCH_PER; ++q) {
      const int i = base + q * SEARCH_THREADS + tid;
      const int ii = i < nl ? i : base;
      px[q] = (double)moved[(size_t)ii * 6]; py[q] = (double)moved[(size_t)ii * 6 + 1]; pz[q] = (double)moved[(size_t)ii * 6 + 2];
      best[q] = INFINITY;
      bi[q] = s0;
    }
    for (int t0 = s0; t0 < s1; t0 += SEARCH_TILE) {
      const int tn = min(SEARCH_TILE, s1 - t0);
      __syncthreads();
      for (int j = tid; j < tn; j += SEARCH_THREADS) {
        tile[j * 3] = (double)dst_s[(size_t)(t0 + j) * 6]; tile[j * 3 + 1] = (double)dst_s[(size_t)(t0 + j) * 6 + 1];
        tile[j * 3 + 2] = (double)dst_s[(size_t)(t0 + j) * 6 + 2];
      }
      __syncthreads();
#pragma unroll 4
      for (int j = 0; j < tn; ++j) {   // (unrolled: the LDS reads of four scene points are in flight under the arithmetic of the previous ones)
        const double qx = tile[j * 3], qy = tile[j * 3 + 1], qz = tile[j * 3 + 2];
#pragma unroll
        for (int q = 0; q < SEARCH_PER; ++q) {
          const double dx = px[q] - qx, dy = py[q] - qy, dz = pz[q] - qz;
          const double dsq = (dx * dx + dy * dy) + dz * dz;
          if (dsq < best[q]) { best[q] = dsq; bi[q] = t0 + j; }   // ties: the lowest scene index
        }
      }
    }
#pragma unroll
    for (int q = 0; q < SEARCH_PER; ++q) {
      const int i = base + q * SEARCH_THREADS + tid;
      if (i < nl) atomicMin(&nnkey[i], ((unsigned long long)__double_as_longlong(best[q]) & ~((1ull << IDX_BITS) - 1ull)) | (unsigned long long)bi[q]);
    }
  }
}

// the rest of one ICP iteration + the loop control (one workgroup per object): robust rejection threshold, one-to-one filter, linearised
// point-to-plane solve for the level transform, moved = pose_x(src_t), stop rule; at the end of a level: fold pose_x into the pose and start
// the next level -- or, after the finest one, undo the normalisation and retire the object.
__global__ __launch_bounds__(NN_THREADS) void icpnn_step(NnScratch sc, int cap, NnRow* __restrict__ rows, int iterations, float tolerance,
                                                         float rejection_scale) {
  __shared__ double red[NN_THREADS / 64];
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_u[2];
  __shared__ double sh_d[40];
  __shared__ double pose_s[16], posex_s[16];
  __shared__ int sh_i[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  NnRow& R = rows[row];
  if (R.status == 0 || R.active == 0) return;
  const NnPtrs P = row_ptrs(sc, row, cap);
  const int nl = R.nl, ml = R.ml, level = R.level, max_it = R.max_it;
  if (tid < 16) posex_s[tid] = R.pose_x[tid];
  int it = R.it;
  const double tol_p = R.tol_p;
  double fval_old = R.fval_old, fval_perc = R.fval_perc, fval_min = R.fval_min;
  const float *src_t = P.src_t, *dst_s = P.dst_s;
  float *d2 = P.d2, *dev = P.dev, *moved = P.moved;
  const int* nn = P.nn;
  unsigned long long* key = P.key;
  __syncthreads();   // (every thread has read the row before thread 0 may rewrite it)
  // ---- the search's winners: scene index from the merged key, the distance again in exact float64 -> float32 (FLANN's L2 functor returns
  // the SQUARED distance of the float32 cloud); the keys are reset for the next search -----------------------------------------------------
  for (int i = tid; i < nl; i += NN_THREADS) {
    const int j = (int)(P.nnkey[i] & ((1ull << IDX_BITS) - 1ull));
    P.nnkey[i] = ~0ull;
    const double dx = (double)moved[(size_t)i * 6] - (double)dst_s[(size_t)j * 6], dy = (double)moved[(size_t)i * 6 + 1] - (double)dst_s[(size_t)j * 6 + 1],
                 dz = (double)moved[(size_t)i * 6 + 2] - (double)dst_s[(size_t)j * 6 + 2];
    P.nn[i] = j;
    d2[i] = (float)((dx * dx + dy * dy) + dz * dz);
  }
  __syncthreads();
  // ---- robust rejection threshold: median + scale * 1.48257968 * MAD (lower medians) --------------------------------------------------
  float thr = INFINITY;
  if (rejection_scale > 0.f) {
    const float med = block_median(d2, nl, hist, sh_u);
    for (int i = tid; i < nl; i += NN_THREADS) dev[i] = (float)fabs((double)d2[i] - (double)med);
    __syncthreads();
    const float mad = block_median(dev, nl, hist, sh_u);
    const double s = 1.48257968 * (double)mad;
    thr = (float)((double)rejection_scale * s + (double)med);
  }
  // ---- picky one-to-one filter: a scene point keeps its closest accepted model point (ties: lowest model index) ------------------------
  for (int j = tid; j < ml; j += NN_THREADS) key[j] = ~0ull;
  __syncthreads();
  for (int i = tid; i < nl; i += NN_THREADS)
    if (d2[i] < thr) atomicMin(&key[nn[i]], ((unsigned long long)__float_as_uint(d2[i]) << 32) | (unsigned long long)(unsigned)i);
  __syncthreads();
  // ---- linearised point-to-plane normal equations over the matched pairs (float64), + the residual's sum of squares ---------------------
  double a[28];
  for (int k = 0; k < 28; ++k) a[k] = 0.0;
  int cnt = 0;
  double fsq = 0.0;
  for (int j = tid; j < ml; j += NN_THREADS) {
    const unsigned long long kk = key[j];
    if (kk == ~0ull) continue;
    const int i = (int)(kk & 0xFFFFFFFFull);
    const float* s6 = src_t + (size_t)i * 6;
    const float* d6 = dst_s + (size_t)j * 6;
    const double sx = s6[0], sy = s6[1], sz = s6[2], nx = d6[3], ny = d6[4], nz = d6[5];
    const double J[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx, nx, ny, nz};   // [cross(src, n_dst), n_dst]
    const double bb = ((double)d6[0] - sx) * nx + ((double)d6[1] - sy) * ny + ((double)d6[2] - sz) * nz;
    int k = 0;
    for (int p = 0; p < 6; ++p)
      for (int c = p; c < 6; ++c) a[k++] += J[p] * J[c];
    for (int p = 0; p < 6; ++p) a[21 + p] += J[p] * bb;
    for (int c = 0; c < 6; ++c) { const double e = (double)s6[c] - (double)d6[c]; fsq += e * e; }
    ++cnt;
  }
  for (int k = 0; k < 27; ++k) {
    const double s = block_sum_d(a[k], red);
    if (tid == 0) sh_d[k] = s;
  }
  {
    const double s = block_sum_d(fsq, red);
    const double c = block_sum_d((double)cnt, red);
    if (tid == 0) { sh_d[27] = s; sh_d[28] = c; }
  }
  __syncthreads();
  const int n_match = (int)sh_d[28];
  bool level_over = n_match < 6;            // `break`: the level ends with the transform of the previous iteration
  if (!level_over) {
    // ---- solve (thread 0): Cholesky of A^T A (the restatement uses an SVD least-squares solve of the same system) --------------------
    if (tid == 0) {
      double A[6][6], b[6], L[6][6];
      int k = 0;
      for (int p = 0; p < 6; ++p)
        for (int c = p; c < 6; ++c) { A[p][c] = sh_d[k]; A[c][p] = sh_d[k]; ++k; }
      for (int p = 0; p < 6; ++p) b[p] = sh_d[21 + p];
      bool ok = true;
      for (int i = 0; i < 6 && ok; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = A[i][j];
          for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
          if (i == j) {
            if (!(s > 0.0)) { ok = false; break; }
            L[i][i] = sqrt(s);
          } else {
            L[i][j] = s / L[j][j];
          }
        }
      double yv[6], x[6];
      if (ok) {
        for (int i = 0; i < 6; ++i) { double s = b[i]; for (int q = 0; q < i; ++q) s -= L[i][q] * yv[q]; yv[i] = s / L[i][i]; }
        for (int i = 5; i >= 0; --i) { double s = yv[i]; for (int q = i + 1; q < 6; ++q) s -= L[q][i] * x[q]; x[i] = s / L[i][i]; }
        for (int i = 0; i < 6; ++i) ok = ok && isfinite(x[i]);
      }
      sh_i[0] = ok ? 1 : 0;
      if (ok) {   // eulerToDCM: Rx(e0) (Ry(e1) Rz(e2))
        const double cx = cos(x[0]), sx = sin(x[0]), cy = cos(x[1]), sy = sin(x[1]), cz = cos(x[2]), sz = sin(x[2]);
        const double Ry_Rz[9] = {cy * cz, -cy * sz, sy, sz, cz, 0, -sy * cz, sy * sz, cy};
        double Rm[9];
        for (int c = 0; c < 3; ++c) {
          Rm[c] = Ry_Rz[c];
          Rm[3 + c] = cx * Ry_Rz[3 + c] - sx * Ry_Rz[6 + c];
          Rm[6 + c] = sx * Ry_Rz[3 + c] + cx * Ry_Rz[6 + c];
        }
        const double fval = sqrt(sh_d[27]) / (double)nl;   // ||s_m - d_m||_F / len(moved)
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) sh_d[r * 4 + c] = Rm[r * 3 + c];
          sh_d[r * 4 + 3] = x[3 + r];
        }
        sh_d[12] = 0; sh_d[13] = 0; sh_d[14] = 0; sh_d[15] = 1;
        sh_d[30] = fval;
        for (int k2 = 0; k2 < 16; ++k2) { R.pose_x[k2] = sh_d[k2]; posex_s[k2] = sh_d[k2]; }
      }
    }
    __syncthreads();
    if (!sh_i[0]) {
      level_over = true;                     // `break` on a non-finite solution
    } else {
      const double fval = sh_d[30];
      for (int i = tid; i < nl; i += NN_THREADS) transform_point(src_t + (size_t)i * 6, sh_d, moved + (size_t)i * 6);
      fval_perc = fval / fval_old;
      fval_old = fval;
      fval_min = fval < fval_min ? fval : fval_min;
      ++it;
      level_over = (1.0 - tol_p < fval_perc && fval_perc < 1.0 + tol_p) || it >= max_it;
    }
  }
  __syncthreads();
  if (!level_over) {
    if (tid == 0) { R.it = it; R.fval_old = fval_old; R.fval_perc = fval_perc; R.fval_min = fval_min; }
    return;
  }
  // ---- end of the level: pose = pose_x @ pose, residual = the level's best fval -----------------------------------------------------------
  if (tid < 16) {
    const int r = tid >> 2, c = tid & 3;
    double s = 0.0;
    for (int q = 0; q < 4; ++q) s += posex_s[r * 4 + q] * R.pose[q * 4 + c];
    pose_s[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    R.iters[level] = it;
    R.residual = fval_min;
    for (int k = 0; k < 16; ++k) R.pose[k] = pose_s[k];
  }
  if (level > 0) {
    level_start(R, P, pose_s, level - 1, iterations, tolerance);
    return;
  }
  // ---- undo the normalisation: t = t / scale + mean_avg - R mean_avg ------------------------------------------------------------------------
  if (tid == 0) {
    const double* mean_avg = R.mean_avg;
    for (int r = 0; r < 3; ++r)
      R.pose[r * 4 + 3] = pose_s[r * 4 + 3] / R.scale + mean_avg[r] - (pose_s[r * 4] * mean_avg[0] + pose_s[r * 4 + 1] * mean_avg[1] + pose_s[r * 4 + 2] * mean_avg[2]);
    R.active = 0;
  }
}

// TCO_refined = pose @ (TCO with the centroid shift added to its translation) when residual in [0, tolerance], else the input pose
__global__ void icpnn_finalize(const NnRow* __restrict__ rows, const float* __restrict__ TCO, int N, float tolerance, float* __restrict__ TCO_out,
                               int32_t* __restrict__ retval, float* __restrict__ residual) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const NnRow& r = rows[n];
  const float* T = TCO + (size_t)n * 16;
  float* O = TCO_out + (size_t)n * 16;
  const bool ran = r.status == 1;
  const bool ok = ran && r.residual >= 0.0 && r.residual <= (double)tolerance;
  for (int k = 0; k < 16; ++k) O[k] = T[k];
  if (ran) {   // (the reference returns the refined matrix even when it flags the result rejected; its caller then keeps the input pose)
    double Ts[16];
    for (int k = 0; k < 16; ++k) Ts[k] = (double)T[k];
    for (int k = 0; k < 3; ++k) Ts[k * 4 + 3] = (double)(T[k * 4 + 3] + (float)r.shift[k]);   // (the reference's pose is float32: a float32 +=)
    if (ok)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
          double s = 0.0;
          for (int q = 0; q < 4; ++q) s += r.pose[i * 4 + q] * Ts[q * 4 + j];
          O[i * 4 + j] = (float)s;
        }
  }
  if (retval) retval[n] = ok ? 0 : -1;
  if (residual) residual[n] = ran ? (float)r.residual : -1.0f;
}

}  // namespace mp

using namespace mp;

static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int mp_icp_nn_max_points(void) { return 1 << IDX_BITS; }   // (the search packs a scene index into IDX_BITS bits)

// points per object the buffers of a call are sized for: every pixel of the frame, up to the key's limit
static size_t nn_cap(int H, int W) { return std::min((size_t)H * W, (size_t)1 << IDX_BITS); }

extern "C" size_t mp_icp_nn_workspace_bytes(int n_images, int n_rows, int H, int W) {
  const size_t px = (size_t)H * W, cap = nn_cap(H, W);
  const size_t imgs = (size_t)n_images + n_rows;
  size_t b = 0;
  b += 3 * a256(imgs * px * 4);            // fill ping / pong, gaussian temp
  b += 2 * a256(imgs * px);                // valid ping / pong
  b += a256(imgs * px * 6 * 4);            // points + normals of every image
  b += 5 * a256((size_t)n_rows * cap * 6 * 4) + 2 * a256((size_t)n_rows * cap * 4) + a256((size_t)n_rows * cap * 4) + 2 * a256((size_t)n_rows * cap * 8);
  b += a256((size_t)n_rows * sizeof(NnRow)) + a256(imgs * 4) + 4096;
  return b;
}

extern "C" int mp_icp_refine_nn(const float* d_depth_meas, int n_images, const int32_t* d_im_ids, const float* d_depth_rend,
                                const float* d_K_images, const float* d_K_rows, const float* d_TCO, int n_rows, int H, int W,
                                int n_iterations, int n_levels, float tolerance, int n_min_points, const unsigned char* d_masks /*[n_images][H][W] or NULL*/,
                                float* d_TCO_out,
                                int32_t* d_retval, float* d_residual, int32_t* d_iters /*[n_rows][8] or NULL*/, void* d_ws, size_t ws_bytes,
                                mp_stream stream) {
  MP_REQUIRE(d_depth_meas && d_im_ids && d_depth_rend && d_K_images && d_K_rows && d_TCO && d_TCO_out && d_ws, "mp_icp_refine_nn: null pointer");
  MP_REQUIRE(n_images > 0 && n_rows >= 0 && H > 2 && W > 2 && n_iterations > 0 && n_levels >= 1 && n_levels <= 8, "mp_icp_refine_nn: bad size");
  MP_REQUIRE(ws_bytes >= mp_icp_nn_workspace_bytes(n_images, n_rows, H, W), "mp_icp_refine_nn: workspace too small");
  if (n_rows == 0) return MP_OK;
  MP_REQUIRE(n_rows <= 65535 && n_images <= 65535, "mp_icp_refine_nn: at most 65535 rows / images");
  MP_REQUIRE(n_min_points >= 6, "mp_icp_refine_nn: n_min_points must be >= 6 (an empty / degenerate cloud has no centroid and no 6-dof solve; the reference uses 1000)");
  hipStream_t s = (hipStream_t)stream;
  const size_t px = (size_t)H * W, cap = nn_cap(H, W);
  const int imgs = n_images + n_rows;   // image batch: the measured frames first, then the rendered depth of every object
  unsigned char* w = (unsigned char*)d_ws;
  auto take = [&](size_t bytes) { unsigned char* p = w; w += a256(bytes); return p; };
  float* fa = (float*)take(imgs * px * 4);
  float* fb = (float*)take(imgs * px * 4);
  float* raw = (float*)take(imgs * px * 4);
  unsigned char* va = take(imgs * px);
  unsigned char* vb = take(imgs * px);
  float* pts = (float*)take(imgs * px * 6 * 4);
  NnScratch sc;
  sc.src = (float*)take((size_t)n_rows * cap * 24);
  sc.dst = (float*)take((size_t)n_rows * cap * 24);
  sc.src_t = (float*)take((size_t)n_rows * cap * 24);
  sc.moved = (float*)take((size_t)n_rows * cap * 24);
  sc.dst_s = (float*)take((size_t)n_rows * cap * 24);
  sc.d2 = (float*)take((size_t)n_rows * cap * 4);
  sc.dev = (float*)take((size_t)n_rows * cap * 4);
  sc.nn = (int*)take((size_t)n_rows * cap * 4);
  sc.key = (unsigned long long*)take((size_t)n_rows * cap * 8);
  sc.nnkey = (unsigned long long*)take((size_t)n_rows * cap * 8);
  NnRow* rows = (NnRow*)take((size_t)n_rows * sizeof(NnRow));
  ProfScope prof("icp_refine_nn", 0.0, (double)imgs * px * 40.0, s);
  // batch the images: raw = [measured frames | rendered depths]
  MP_CHECK_HIP(hipMemcpyAsync(raw, d_depth_meas, (size_t)n_images * px * 4, hipMemcpyDeviceToDevice, s));
  MP_CHECK_HIP(hipMemcpyAsync(raw + (size_t)n_images * px, d_depth_rend, (size_t)n_rows * px * 4, hipMemcpyDeviceToDevice, s));
  const long total = (long)imgs * (long)px;
  hipLaunchKernelGGL(icpnn_init_valid, dim3(ceil_div(total, 256L)), dim3(256), 0, s, raw, fa, va, total);
  const dim3 ig(ceil_div((long)px, 256L), imgs);
  float *cur = fa, *nxt = fb;
  unsigned char *vc = va, *vn = vb;
  for (int r = 0; r < FILL_RINGS; ++r) {
    hipLaunchKernelGGL(icpnn_fill_ring, ig, dim3(256), 0, s, cur, vc, nxt, vn, H, W);
    std::swap(cur, nxt);
    std::swap(vc, vn);
  }
  GaussW g;
  {
    double sum = 0.0;
    for (int k = -GAUSS_RADIUS; k <= GAUSS_RADIUS; ++k) sum += exp(-0.5 / (2.0 * 2.0) * (double)(k * k));
    for (int k = 0; k <= GAUSS_RADIUS; ++k) g.w[k] = exp(-0.5 / (2.0 * 2.0) * (double)(k * k)) / sum;
  }
  hipLaunchKernelGGL(icpnn_gauss_axis, ig, dim3(256), 0, s, cur, nxt, H, W, 0, g);
  hipLaunchKernelGGL(icpnn_gauss_axis, ig, dim3(256), 0, s, nxt, cur, H, W, 1, g);
  // back-projection + normals: the frames with their own intrinsics, the rendered depths with their row's
  hipLaunchKernelGGL(icpnn_points, dim3(ceil_div((long)px, 256L), n_images), dim3(256), 0, s, raw, cur, d_K_images, (const int32_t*)nullptr, H, W, pts);
  hipLaunchKernelGGL(icpnn_points, dim3(ceil_div((long)px, 256L), n_rows), dim3(256), 0, s, raw + (size_t)n_images * px, cur + (size_t)n_images * px,
                     d_K_rows, (const int32_t*)nullptr, H, W, pts + (size_t)n_images * px * 6);
  hipLaunchKernelGGL(icpnn_compact, dim3(n_rows), dim3(NN_THREADS), 0, s, d_depth_meas, d_im_ids, d_depth_rend, pts, pts + (size_t)n_images * px * 6, H, W,
                     d_masks, (int)cap, n_min_points, sc.src, sc.dst, rows);
  // the ICP: data-dependent iteration counts without a host round trip -- the worst-case number of (search, step) pairs is enqueued, each
  // object tracks its own level / iteration in its row and turns the launches it no longer needs into no-ops
  hipLaunchKernelGGL(icpnn_begin, dim3(n_rows), dim3(NN_THREADS), 0, s, sc, (int)cap, rows, n_iterations, tolerance, n_levels);
  int total_it = 0;
  for (int level = 0; level < n_levels; ++level) total_it += (int)rint((double)n_iterations / (level + 1));
  const dim3 sg(SEARCH_WGS, n_rows);
  for (int k = 0; k < total_it; ++k) {
    hipLaunchKernelGGL(icpnn_search, sg, dim3(SEARCH_THREADS), 0, s, sc, (int)cap, rows);
    hipLaunchKernelGGL(icpnn_step, dim3(n_rows), dim3(NN_THREADS), 0, s, sc, (int)cap, rows, n_iterations, tolerance, 2.5f);
  }
  hipLaunchKernelGGL(icpnn_finalize, dim3(ceil_div(n_rows, 64)), dim3(64), 0, s, rows, d_TCO, n_rows, tolerance, d_TCO_out, d_retval, d_residual);
  if (d_iters) {
    // telemetry: iterations per level (NnRow::iters sits right after n / status)
    MP_CHECK_HIP(hipMemcpy2DAsync(d_iters, 8 * sizeof(int32_t), (const char*)rows + offsetof(NnRow, iters), sizeof(NnRow), 8 * sizeof(int32_t), n_rows,
                                  hipMemcpyDeviceToDevice, s));
  }
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
