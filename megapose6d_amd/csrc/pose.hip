// pose.hip -- the small fp32 pose/geometry ops of the hot path, fused into a handful of launches.
//
// Reference (paths under /root/reference/src/megapose/):
//   normalize_T                         lib3d/transform_ops.py:106-119, lib3d/rotations.py:25-40
//   TCO_init_from_boxes_autodepth_with_R lib3d/cosypose_ops.py:169-218
//   project_points_robust / boxes_from_uv lib3d/camera_geometry.py:40-64
//   deepim_boxes / deepim_crops_robust   lib3d/cropping.py:30-67, :84-110
//   get_K_crop_resize                    lib3d/camera_geometry.py:67-115  (operation order kept)
//   make_TCO_multiview                   lib3d/multiview.py:165-246, :31-92 (closed form)
//   crop_inputs / compute_crops_multiview models/pose_rigid.py:180-303, :540-552
//   update_pose / pose_update_with_reference_point  models/pose_rigid.py:305-312, lib3d/cosypose_ops.py:33-58
// The reference does these as dozens of tiny torch kernels plus a D2H sync and a per-row Python/Panda3D loop
// for the multiview cameras (multiview.py:186-219); here one workgroup per (row, view) does it all on device.
#include "common.h"

namespace mp {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 normalized(V3 a) {
  const float n = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  return v3(a.x / n, a.y / n, a.z / n);
}

// ortho6d Gram-Schmidt (rotations.py:25-40): columns of R = (x, y, z)
__device__ __forceinline__ void ortho6d(V3 x_raw, V3 y_raw, float* R /*3x3 row-major*/) {
  const V3 x = normalized(x_raw);
  const V3 z = normalized(cross(x, y_raw));
  const V3 y = cross(z, x);
  R[0] = x.x; R[1] = y.x; R[2] = z.x;
  R[3] = x.y; R[4] = y.y; R[5] = z.y;
  R[6] = x.z; R[7] = y.z; R[8] = z.z;
}

__device__ __forceinline__ void normalize_T_dev(const float* T, float* O) {
  float R[9];
  ortho6d(v3(T[0], T[4], T[8]), v3(T[1], T[5], T[9]), R);
  O[0] = R[0]; O[1] = R[1]; O[2] = R[2]; O[3] = T[3];
  O[4] = R[3]; O[5] = R[4]; O[6] = R[5]; O[7] = T[7];
  O[8] = R[6]; O[9] = R[7]; O[10] = R[8]; O[11] = T[11];
  O[12] = 0.f; O[13] = 0.f; O[14] = 0.f; O[15] = 1.f;
}

__global__ void normalize_T_kernel(const float* __restrict__ T, int b, float* __restrict__ O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  float o[16];
  normalize_T_dev(T + (size_t)i * 16, o);
  for (int k = 0; k < 16; ++k) O[(size_t)i * 16 + k] = o[k];
}

// block reduce of (min, max) pairs -------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// extents of R*p over a mesh's (padded) point set: one block per (mesh, rotation)
__global__ __launch_bounds__(256) void init_extents_kernel(const float* __restrict__ pts, int n_pts, const float* __restrict__ R,
                                                           int n_rot, float* __restrict__ ext) {
  __shared__ float red[4][4];
  const int r = blockIdx.x, mesh = blockIdx.y;
  const float* Rm = R + (size_t)r * 9;
  const float* P = pts + (size_t)mesh * n_pts * 3;
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const float px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
    const float x = Rm[0] * px + Rm[1] * py + Rm[2] * pz;
    const float y = Rm[3] * px + Rm[4] * py + Rm[5] * pz;
    xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
    ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = xmin; red[wave][1] = xmax; red[wave][2] = ymin; red[wave][3] = ymax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      xmin = fminf(xmin, red[w][0]); xmax = fmaxf(xmax, red[w][1]);
      ymin = fminf(ymin, red[w][2]); ymax = fmaxf(ymax, red[w][3]);
    }
    ext[((size_t)mesh * n_rot + r) * 2 + 0] = xmax - xmin;
    ext[((size_t)mesh * n_rot + r) * 2 + 1] = ymax - ymin;
  }
}

__global__ void init_poses_kernel(const float* __restrict__ boxes, const float* __restrict__ K, const int32_t* __restrict__ mesh_ids,
                                  const int32_t* __restrict__ rot_ids, const float* __restrict__ R, int n_rot,
                                  const float* __restrict__ ext, int b, float* __restrict__ TCO) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const float* bx = boxes + (size_t)i * 4;
  const float* Ki = K + (size_t)i * 9;
  const int r = rot_ids[i];
  const float* Rm = R + (size_t)r * 9;
  const float fx = Ki[0], fy = Ki[4], cx = Ki[2], cy = Ki[5];
  const float ucx = (bx[0] + bx[2]) / 2.0f, ucy = (bx[1] + bx[3]) / 2.0f;
  const float dx3 = ext[((size_t)mesh_ids[i] * n_rot + r) * 2 + 0];
  const float dy3 = ext[((size_t)mesh_ids[i] * n_rot + r) * 2 + 1];
  const float bdx = (bx[2] - bx[0]) + 1.0f, bdy = (bx[3] - bx[1]) + 1.0f;
  const float z_from_dx = fx * dx3 / bdx;
  const float z_from_dy = fy * dy3 / bdy;
  const float z = (z_from_dy + z_from_dx) / 2.0f;
  float* T = TCO + (size_t)i * 16;
  T[0] = Rm[0]; T[1] = Rm[1]; T[2] = Rm[2]; T[3] = ((ucx - cx) * z) / fx;
  T[4] = Rm[3]; T[5] = Rm[4]; T[6] = Rm[5]; T[7] = ((ucy - cy) * z) / fy;
  T[8] = Rm[6]; T[9] = Rm[7]; T[10] = Rm[8]; T[11] = z;
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// look_at with Panda3D semantics (forward exact): y = normalize(target - pos), x = normalize(y x up), z = x x y
__device__ __forceinline__ void look_at(V3 pos, V3 target, V3 up, V3& x, V3& y, V3& z) {
  y = normalized(v3(target.x - pos.x, target.y - pos.y, target.z - pos.z));
  x = normalized(cross(y, up));
  z = cross(x, y);
}

// One workgroup per (row, view).
__global__ __launch_bounds__(256) void pose_prepare_kernel(
    const float* __restrict__ TCO_in, const float* __restrict__ K, const int32_t* __restrict__ mesh_ids,
    const float* __restrict__ points, int n_pts_stride, int n_pts_main, int n_pts_views, int V, int multiview, int im_h,
    int im_w, int out_h, int out_w, float lamb, float* __restrict__ TCO_n, float* __restrict__ tCR_out,
    float* __restrict__ TCV_O, float* __restrict__ KV_crop, float* __restrict__ boxes_rend, float* __restrict__ boxes_crop,
    float* __restrict__ K_main) {
  __shared__ float Tn[16];
  __shared__ float Tv[16];
  __shared__ float P[12];
  __shared__ float red[4][4];
  const int row = blockIdx.x, view = blockIdx.y;
  const float* Ki = K + (size_t)row * 9;
  // view list of the mode (lib3d/multiview.py:197-246): [TCO unless removed] + the mode's camera offsets; with in-plane rotations
  // every entry is repeated 4x (rotated by 0 / 90 / 180 / 270 degrees about the optical axis).  blockIdx.y == V (only launched when
  // the TCO rendering is removed) is the MAIN unit: crop_inputs of the observation from TCO itself (models/pose_rigid.py:180-247).
  const int mode = multiview & 255;
  const bool remove_tco = (multiview & MP_MV_REMOVE_TCO) != 0, inplane = (multiview & MP_MV_INPLANE) != 0;
  const bool main_only = view == V;
  const int base = inplane ? view >> 2 : view, quarter = inplane ? view & 3 : 0;
  const bool is_tco = main_only || mode == 0 || (!remove_tco && base == 0);
  const bool is_main = remove_tco ? main_only : (view == 0);
  if (threadIdx.x == 0) {
    normalize_T_dev(TCO_in + (size_t)row * 16, Tn);
    if (is_tco) {
      for (int k = 0; k < 16; ++k) Tv[k] = Tn[k];
    } else {
      // TOC = inv(TCO_n)
      const V3 r0 = v3(Tn[0], Tn[1], Tn[2]), r1 = v3(Tn[4], Tn[5], Tn[6]), r2 = v3(Tn[8], Tn[9], Tn[10]);
      const V3 t = v3(Tn[3], Tn[7], Tn[11]);
      // p0 = -R^T t ; columns of R are (r0.x,r1.x,r2.x) ...
      const V3 p0 = v3(-(r0.x * t.x + r1.x * t.y + r2.x * t.z), -(r0.y * t.x + r1.y * t.y + r2.y * t.z),
                       -(r0.z * t.x + r1.z * t.y + r2.z * t.z));
      const V3 up = v3(-r1.x, -r1.y, -r1.z);  // -TOC[:3,1]
      // ref = R^T tCR + p0 with tCR = t
      const V3 ref = v3((r0.x * t.x + r1.x * t.y + r2.x * t.z) + p0.x, (r0.y * t.x + r1.y * t.y + r2.y * t.z) + p0.y,
                        (r0.z * t.x + r1.z * t.y + r2.z * t.z) + p0.z);
      const float radius = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
      V3 lx, ly, lz;
      look_at(p0, ref, up, lx, ly, lz);
      // camera offsets in the look-at frame (x right, y forward, z up), in units of |tCR|:
      //   "TCO+front_3views" (multiview.py:104-112): (0,0,0), (+1,0,0), (-1,0,0);  "TCO+front_1view" (:95-101): (0,0,0);
      //   "sphere_26views" (:150-162): y in [0,1,2] x x in [0,-1,1] x z in [0,1,-1] without (0,1,0)
      const int oi = base - (remove_tco ? 0 : 1);
      float fx_o = 0.f, fy_o = 0.f, fz_o = 0.f;
      if (mode == 1) {
        fx_o = oi == 1 ? 1.0f : (oi == 2 ? -1.0f : 0.0f);
      } else if (mode == 3) {
        const int raw = oi >= 9 ? oi + 1 : oi;   // the skipped combination (x, y, z) = (0, 1, 0) is number 9 of the 27
        const int yi = raw / 9, xi = (raw / 3) % 3, zi = raw % 3;
        fy_o = (float)yi;
        fx_o = xi == 0 ? 0.f : (xi == 1 ? -1.f : 1.f);
        fz_o = zi == 0 ? 0.f : (zi == 1 ? 1.f : -1.f);
      }
      const float ox = fx_o * radius, oy = fy_o * radius, oz = fz_o * radius;
      const V3 pn = v3(p0.x + lx.x * ox + ly.x * oy + lz.x * oz, p0.y + lx.y * ox + ly.y * oy + lz.y * oz,
                       p0.z + lx.z * ox + ly.z * oy + lz.z * oz);
      V3 nx, ny, nz;
      look_at(pn, ref, up, nx, ny, nz);
      // TCV_O = [TCCGL Rn^T | -TCCGL Rn^T pn] : rows (x, -z, y)
      Tv[0] = nx.x; Tv[1] = nx.y; Tv[2] = nx.z; Tv[3] = -dot(nx, pn);
      Tv[4] = -nz.x; Tv[5] = -nz.y; Tv[6] = -nz.z; Tv[7] = dot(nz, pn);
      Tv[8] = ny.x; Tv[9] = ny.y; Tv[10] = ny.z; Tv[11] = -dot(ny, pn);
      Tv[12] = 0.f; Tv[13] = 0.f; Tv[14] = 0.f; Tv[15] = 1.f;
    }
    if (quarter) {   // in-plane copy: R' = Rz(quarter * 90 deg) R, translation unchanged (multiview.py:236-245)
      const float c = quarter == 2 ? -1.f : 0.f, sn = quarter == 1 ? 1.f : (quarter == 3 ? -1.f : 0.f);
      for (int j = 0; j < 3; ++j) {
        const float a0 = Tv[j], a1 = Tv[4 + j];
        Tv[j] = c * a0 - sn * a1;
        Tv[4 + j] = sn * a0 + c * a1;
      }
    }
    // P = K @ Tv[:3]  (3x4)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) P[i * 4 + j] = Ki[i * 3] * Tv[j] + Ki[i * 3 + 1] * Tv[4 + j] + Ki[i * 3 + 2] * Tv[8 + j];
    if (is_main) {
      for (int k = 0; k < 16; ++k) TCO_n[(size_t)row * 16 + k] = Tn[k];
      tCR_out[(size_t)row * 3 + 0] = Tn[3];
      tCR_out[(size_t)row * 3 + 1] = Tn[7];
      tCR_out[(size_t)row * 3 + 2] = Tn[11];
    }
    if (!main_only)
      for (int k = 0; k < 16; ++k) TCV_O[((size_t)row * V + view) * 16 + k] = Tv[k];
  }
  __syncthreads();
  // crop_inputs samples 2000 points, compute_crops_multiview 200 (pose_rigid.py:213, :279); KV_crop[:, 0] = K_crop only when the TCO
  // view is rendered (:551-552)
  const int n_pts = is_main ? n_pts_main : n_pts_views;
  const float* pts = points + (size_t)mesh_ids[row] * n_pts_stride * 3;
  float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const float su = P[0] * px + P[1] * py + P[2] * pz + P[3];
    const float sv = P[4] * px + P[5] * py + P[6] * pz + P[7];
    float sz = P[8] * px + P[9] * py + P[10] * pz + P[11];
    sz = fmaxf(0.1f, sz);
    const float u = su / sz, v = sv / sz;
    umin = fminf(umin, u); umax = fmaxf(umax, u);
    vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
  }
  umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = umin; red[wave][1] = umax; red[wave][2] = vmin; red[wave][3] = vmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      umin = fminf(umin, red[w][0]); umax = fmaxf(umax, red[w][1]);
      vmin = fminf(vmin, red[w][2]); vmax = fmaxf(vmax, red[w][3]);
    }
    // crop centre: projection of the reference point = translation of the view transform (tOR = 0)
    const float cz = fmaxf(0.1f, P[11]);
    const float xc = P[3] / cz, yc = P[7] / cz;
    const float xdist = fmaxf(fabsf(umin - xc), fabsf(umax - xc));
    const float ydist = fmaxf(fabsf(vmin - yc), fabsf(vmax - yc));
    const float w_im = (float)max(im_h, im_w), h_im = (float)min(im_h, im_w);
    const float r = w_im / h_im;
    const float width = fmaxf(xdist, ydist * r) * 2.0f * lamb;
    const float height = fmaxf(xdist / r, ydist) * 2.0f * lamb;
    const float x1 = xc - width / 2.0f, y1 = yc - height / 2.0f, x2 = xc + width / 2.0f, y2 = yc + height / 2.0f;
    if (is_main) {
      float* br = boxes_rend + (size_t)row * 4;
      br[0] = umin; br[1] = vmin; br[2] = umax; br[3] = vmax;
      float* bc = boxes_crop + (size_t)row * 4;
      bc[0] = x1; bc[1] = y1; bc[2] = x2; bc[3] = y2;
    }
    // get_K_crop_resize, reference operation order (camera_geometry.py:87-114)
    const float final_width = (float)max(out_h, out_w), final_height = (float)min(out_h, out_w);
    const float crop_width = x2 - x1, crop_height = y2 - y1;
    const float crop_cj = (x1 + x2) / 2.0f, crop_ci = (y1 + y2) / 2.0f;
    float cx = Ki[2] + (crop_width - 1.0f) / 2.0f - crop_cj;
    float cy = Ki[5] + (crop_height - 1.0f) / 2.0f - crop_ci;
    const float center_x = (crop_width - 1.0f) / 2.0f, center_y = (crop_height - 1.0f) / 2.0f;
    const float orig_cx_diff = cx - center_x, orig_cy_diff = cy - center_y;
    const float scale_x = final_width / crop_width, scale_y = final_height / crop_height;
    const float scaled_center_x = (final_width - 1.0f) / 2.0f, scaled_center_y = (final_height - 1.0f) / 2.0f;
    const float fx = scale_x * Ki[0], fy = scale_y * Ki[4];
    cx = scaled_center_x + scale_x * orig_cx_diff;
    cy = scaled_center_y + scale_y * orig_cy_diff;
    if (!main_only) {
      float* Ko = KV_crop + ((size_t)row * V + view) * 9;
      Ko[0] = fx; Ko[1] = Ki[1]; Ko[2] = cx;
      Ko[3] = Ki[3]; Ko[4] = fy; Ko[5] = cy;
      Ko[6] = Ki[6]; Ko[7] = Ki[7]; Ko[8] = Ki[8];
    }
    if (is_main && K_main) {   // K_crop of crop_inputs: what update_pose consumes (pose_rigid.py:305-312)
      float* Ko = K_main + (size_t)row * 9;
      Ko[0] = fx; Ko[1] = Ki[1]; Ko[2] = cx;
      Ko[3] = Ki[3]; Ko[4] = fy; Ko[5] = cy;
      Ko[6] = Ki[6]; Ko[7] = Ki[7]; Ko[8] = Ki[8];
    }
  }
}

__global__ void pose_update_kernel(const float* __restrict__ TCO, const float* __restrict__ Kc, int k_stride,
                                   const float* __restrict__ out9, const float* __restrict__ tCR, int b,
                                   float* __restrict__ TCO_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const float* T = TCO + (size_t)i * 16;
  const float* K = Kc + (size_t)i * k_stride;
  const float* o = out9 + (size_t)i * 9;
  const float* c = tCR + (size_t)i * 3;
  float dR[9];
  ortho6d(v3(o[0], o[1], o[2]), v3(o[3], o[4], o[5]), dR);
  const float vx = o[6], vy = o[7], vz = o[8];
  const float zsrc = c[2];
  const float ztgt = vz * zsrc;
  const float tx = ((vx / K[0]) + (c[0] / zsrc)) * ztgt;
  const float ty = ((vy / K[4]) + (c[1] / zsrc)) * ztgt;
  const float dx = T[3] - c[0], dy = T[7] - c[1], dz = T[11] - c[2];
  float* O = TCO_out + (size_t)i * 16;
  O[3] = (dR[0] * dx + dR[1] * dy + dR[2] * dz) + tx;
  O[7] = (dR[3] * dx + dR[4] * dy + dR[5] * dz) + ty;
  O[11] = (dR[6] * dx + dR[7] * dy + dR[8] * dz) + ztgt;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) O[r * 4 + cc] = dR[r * 3] * T[cc] + dR[r * 3 + 1] * T[4 + cc] + dR[r * 3 + 2] * T[8 + cc];
  O[12] = T[12]; O[13] = T[13]; O[14] = T[14]; O[15] = T[15];
}

}  // namespace mp

using namespace mp;

extern "C" int mp_normalize_T(const float* d_T, int b, float* d_out, mp_stream stream) {
  MP_REQUIRE(d_T && d_out && b >= 0, "mp_normalize_T: bad arguments");
  if (b == 0) return MP_OK;
  hipLaunchKernelGGL(normalize_T_kernel, dim3(ceil_div(b, 128)), dim3(128), 0, (hipStream_t)stream, d_T, b, d_out);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_init_extents(const float* d_points, int n_mesh, int n_pts, const float* d_R, int n_rot, float* d_ext,
                               mp_stream stream) {
  MP_REQUIRE(d_points && d_R && d_ext && n_mesh > 0 && n_pts > 0 && n_rot > 0 && n_mesh <= 65535, "mp_init_extents: bad arguments");
  hipLaunchKernelGGL(init_extents_kernel, dim3(n_rot, n_mesh), dim3(256), 0, (hipStream_t)stream, d_points, n_pts, d_R, n_rot,
                     d_ext);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_init_poses_from_boxes(const float* d_boxes, const float* d_K, const int32_t* d_mesh_ids,
                                        const int32_t* d_rot_ids, const float* d_R, int n_rot, const float* d_ext, int b,
                                        float* d_TCO, mp_stream stream) {
  MP_REQUIRE(d_boxes && d_K && d_mesh_ids && d_rot_ids && d_R && d_ext && d_TCO && b >= 0, "mp_init_poses_from_boxes: bad arguments");
  if (b == 0) return MP_OK;
  hipLaunchKernelGGL(init_poses_kernel, dim3(ceil_div(b, 128)), dim3(128), 0, (hipStream_t)stream, d_boxes, d_K, d_mesh_ids,
                     d_rot_ids, d_R, n_rot, d_ext, b, d_TCO);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_pose_multiview_n_views(int multiview) {
  const int mode = multiview & 255;
  if (mode < 0 || mode > 3) return -1;
  if (mode == 0) return 1;
  const int n_off = mode == 1 ? 3 : (mode == 2 ? 1 : 26);
  const int n_base = ((multiview & MP_MV_REMOVE_TCO) ? 0 : 1) + n_off;
  return n_base * ((multiview & MP_MV_INPLANE) ? 4 : 1);
}

extern "C" int mp_pose_prepare_ex(const float* d_TCO_in, const float* d_K, const int32_t* d_mesh_ids, const float* d_points,
                                  int n_pts_stride, int n_pts_main, int n_pts_views, int b, int V, int multiview, int im_h,
                                  int im_w, int out_h, int out_w, float lamb, float* d_TCO_n, float* d_tCR, float* d_TCV_O,
                                  float* d_KV_crop, float* d_boxes_rend, float* d_boxes_crop, float* d_K_main, mp_stream stream) {
  MP_REQUIRE(d_TCO_in && d_K && d_mesh_ids && d_points && d_TCO_n && d_tCR && d_TCV_O && d_KV_crop && d_boxes_rend && d_boxes_crop,
             "mp_pose_prepare: null pointer");
  int v_need = mp_pose_multiview_n_views(multiview);
  // one view: make_TCO_multiview returns [TCO] whatever the type (lib3d/multiview.py:186-193); remove_TCO_rendering still decides
  // whether that view's intrinsics are crop_inputs' K_crop or the 200-point multiview crop (models/pose_rigid.py:550-552)
  if (v_need == 1) multiview &= MP_MV_REMOVE_TCO;
  MP_REQUIRE(v_need > 0 && (multiview & ~(255 | MP_MV_REMOVE_TCO | MP_MV_INPLANE)) == 0, "mp_pose_prepare: unknown multiview code 0x%x", multiview);
  MP_REQUIRE(V == v_need, "mp_pose_prepare: multiview 0x%x has %d views, got V=%d", multiview, v_need, V);
  MP_REQUIRE(!(multiview & MP_MV_INPLANE) || (multiview & MP_MV_REMOVE_TCO), "mp_pose_prepare: views_inplane_rotations needs remove_TCO_rendering "
             "(lib3d/multiview.py:237)");
  MP_REQUIRE(n_pts_main <= n_pts_stride && n_pts_views <= n_pts_stride && n_pts_main > 0, "mp_pose_prepare: bad point counts");
  if (b == 0) return MP_OK;
  const bool extra = (multiview & MP_MV_REMOVE_TCO) != 0;   // no view carries crop_inputs' 2000-point crop: one more unit computes it
  ProfScope prof("pose_prepare", 0.0, (double)b * (12.0 * n_pts_main + (V - 1) * 12.0 * n_pts_views), (hipStream_t)stream);
  hipLaunchKernelGGL(pose_prepare_kernel, dim3(b, V + (extra ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, d_TCO_in, d_K, d_mesh_ids, d_points,
                     n_pts_stride, n_pts_main, n_pts_views, V, multiview, im_h, im_w, out_h, out_w, lamb, d_TCO_n, d_tCR, d_TCV_O,
                     d_KV_crop, d_boxes_rend, d_boxes_crop, d_K_main);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_pose_prepare(const float* d_TCO_in, const float* d_K, const int32_t* d_mesh_ids, const float* d_points,
                               int n_pts_stride, int n_pts_main, int n_pts_views, int b, int V, int multiview, int im_h,
                               int im_w, int out_h, int out_w, float lamb, float* d_TCO_n, float* d_tCR, float* d_TCV_O,
                               float* d_KV_crop, float* d_boxes_rend, float* d_boxes_crop, mp_stream stream) {
  return mp_pose_prepare_ex(d_TCO_in, d_K, d_mesh_ids, d_points, n_pts_stride, n_pts_main, n_pts_views, b, V, multiview, im_h, im_w, out_h,
                            out_w, lamb, d_TCO_n, d_tCR, d_TCV_O, d_KV_crop, d_boxes_rend, d_boxes_crop, nullptr, stream);
}

extern "C" int mp_pose_update(const float* d_TCO, const float* d_K_crop, int k_stride_floats, const float* d_out9,
                              const float* d_tCR, int b, float* d_TCO_out, mp_stream stream) {
  MP_REQUIRE(d_TCO && d_K_crop && d_out9 && d_tCR && d_TCO_out && k_stride_floats >= 9, "mp_pose_update: bad arguments");
  if (b == 0) return MP_OK;
  hipLaunchKernelGGL(pose_update_kernel, dim3(ceil_div(b, 128)), dim3(128), 0, (hipStream_t)stream, d_TCO, d_K_crop,
                     k_stride_floats, d_out9, d_tCR, b, d_TCO_out);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
