// raster.hip -- on-device triangle rasteriser for gfx950 (replaces the Panda3D/OpenGL render loop).
//
// Reference contract: Panda3dBatchRenderer.render
//   (/root/reference/src/megapose/panda3d_renderer/panda3d_batch_renderer.py:217-282, worker_loop :89-150,
//    Panda3dSceneRenderer.render_scene panda3d_scene_renderer.py:298-358, camera model types.py:75-101,
//    eye-normal LUT utils.py:58-68 + panda3d_scene_renderer.py:210-216, depth utils.py:44-55,
//    lights panda3d_scene_renderer.py:104-136).
//
// Pixel contract (identical, operation for operation, to oracle/raster.c -- see DESIGN.md "Rasteriser"):
//   * camera-space vertex  Pc = R p + t  as an fmaf chain; screen  sx = fmaf(fx, x/z.., cx)
//   * screen coords snapped to 1/256 px fixed point; coverage by exact integer edge functions (evaluated in fp64) sampled at
//     pixel centres (x+.5, y+.5) with the top-left fill rule; two-sided (no back-face culling)
//   * depth test on wsum = sum b_i/z_i (perspective-correct 1/z), nearest wins, ties -> lowest triangle id;
//     fragments outside [near, far] = [0.1, 10] m are discarded
//   * attributes interpolated perspective-correctly; RGB = albedo * light; normals through the 32^3 LUT
//     (separable, linear filter, repeat wrap); uint8 quantisation then /255.
//   * albedo = vertex colour, modulated for UV-textured meshes by a bilinear, repeat-wrapped sample of a host-built
//     RGBA8 mip chain; the mip level is chosen per triangle from its texel-area / pixel-area ratio (thresholds 2, 8, 32, ...).
//
// Structure: (1) raster_transform: one thread per (view, vertex) -> {X, Y (fixed point), 1/z, valid}
//            (2) raster_bands: one workgroup per (view, band of BAND_H rows): 64-bit {depth,tri} z-buffer in
//                LDS, row-bounds prefilter + LDS compaction, thread-per-triangle LDS atomicMin for tiny triangles and a
//                wave-per-triangle queue for the rest,
//                then a pixel-parallel resolve/shade that writes straight into the CNN input tensor slice.
// Roofline: HBM-bound on the output writes ((3+3[+1])*4*h*w bytes per view).
#include <cmath>
#include <vector>

#include "common.h"
#include "crop_device.h"

namespace mp {

constexpr int SUBPIX = 256;
constexpr float GUARD = 16384.f;  // |screen coord| limit (pixels) for the fixed-point path
constexpr float Z_EPS = 1e-6f;
constexpr float Z_NEAR = 0.1f, Z_FAR = 10.0f;
#ifndef MP_BAND_H
#define MP_BAND_H 16
#endif
constexpr int BAND_H = MP_BAND_H;
constexpr int BAND_THREADS = 512;
constexpr int BIG_TRI_AREA = 128;   // clipped-bbox pixels above which a triangle is rasterised by a whole wave (a lane that
                                    // walks a several-hundred-pixel bbox alone stalls its wave and the block's barrier)
constexpr int BIG_QUEUE = 1024;
constexpr int SCAN_CHUNK = 2048;  // triangles scanned between two looks at the list fill level
constexpr int LIST_CAP = 4096;    // compacted triangle list (drained when another scan chunk might not fit)
constexpr int STAGE_CH = 7;  // rgb(3) + normals(3) + depth(1) staged per pixel for coalesced stores

struct MeshDev {
  const float* verts;
  const float* normals;
  const float* colors;
  const int32_t* faces;
  int n_verts, n_faces;
  float radius;
  float center[3];
  const float* uvs;  // UV texture (optional): per-corner uv [n_faces][3][2], NULL = vertex colours only
};

// texture of mesh i (kept out of MeshDev so that the untextured path does not carry it in registers):
// RGBA8 mip chain, level l at texels + tex_off[l], size max(1, w>>l) x max(1, h>>l)
struct TexDev {
  const uint32_t* texels;
  int tex_w, tex_h, tex_levels;
  int tex_off[MP_TEX_MAX_LEVELS];
};

struct VtxRec {
  int X, Y;     // fixed-point screen coordinates (1/256 px)
  float invz;   // 1 / camera z
  int valid;
};

// optional fused crop role (one extra workgroup per (item, band)): roi_align of the observation into channels c0.. of the same
// pixel lines the views write, so that the XCD's L2 merges all slices of a line (see the work-to-workgroup map in raster_bands)
struct CropArgs {
  const float* images;     // [n_im][C][H][W], or [n_im][H][W][4] when nhwc4; NULL = no crop role
  const int32_t* im_ids;   // [n_items]
  const float* boxes;      // [n_items][4]
  int C, H, W, c0, nhwc4;
};

struct LightsDev {
  float ambient[3];
  int n_point;
  float dir[8][3];
  float color[8][3];
};

__device__ __forceinline__ float dot3p(float a0, float a1, float a2, float x, float y, float z, float t) {
  return fmaf(a2, z, fmaf(a1, y, fmaf(a0, x, t)));
}

__device__ __forceinline__ bool view_finite(const float* T, const float* K) {
  bool ok = true;
  for (int i = 0; i < 16; ++i) ok = ok && isfinite(T[i]);
  for (int i = 0; i < 9; ++i) ok = ok && isfinite(K[i]);
  return ok;
}

__global__ void raster_transform(const MeshDev* __restrict__ meshes, const int32_t* __restrict__ mesh_ids,
                                 const float* __restrict__ TCO, const float* __restrict__ K, int max_verts,
                                 VtxRec* __restrict__ out) {
  const int view = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const MeshDev m = meshes[mesh_ids[view]];
  if (v >= m.n_verts) return;
  const float* T = TCO + (size_t)view * 16;
  const float* Kv = K + (size_t)view * 9;
  VtxRec rec;
  rec.X = 0; rec.Y = 0; rec.invz = 0.f; rec.valid = 0;
  if (view_finite(T, Kv)) {
    const float px = m.verts[3 * v + 0], py = m.verts[3 * v + 1], pz = m.verts[3 * v + 2];
    const float x = dot3p(T[0], T[1], T[2], px, py, pz, T[3]);
    const float y = dot3p(T[4], T[5], T[6], px, py, pz, T[7]);
    const float z = dot3p(T[8], T[9], T[10], px, py, pz, T[11]);
    if (z > Z_EPS) {
      const float iz = 1.0f / z;
      const float sx = fmaf(Kv[0], x * iz, Kv[2]);
      const float sy = fmaf(Kv[4], y * iz, Kv[5]);
      if (fabsf(sx) < GUARD && fabsf(sy) < GUARD) {
        rec.X = (int)rintf(sx * (float)SUBPIX);
        rec.Y = (int)rintf(sy * (float)SUBPIX);
        rec.invz = iz;
        rec.valid = 1;
      }
    }
  }
  out[(size_t)view * max_verts + v] = rec;
}

// Per (view, triangle): pixel-row range [ymin, ymax] the triangle can touch, packed ymin | ymax << 16 (0xFFFF = culled).
// Lets every band skip non-overlapping triangles with one coalesced 4-byte read instead of 3 index + 3 vertex gathers.
__global__ void raster_tri_bounds(const MeshDev* __restrict__ meshes, const int32_t* __restrict__ mesh_ids,
                                  const VtxRec* __restrict__ vtx, int max_verts, int max_faces, int h,
                                  unsigned* __restrict__ bounds) {
  const int view = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const MeshDev m = meshes[mesh_ids[view]];
  if (t >= m.n_faces) return;
  const VtxRec* vv = vtx + (size_t)view * max_verts;
  const VtxRec v0 = vv[m.faces[3 * t]], v1 = vv[m.faces[3 * t + 1]], v2 = vv[m.faces[3 * t + 2]];
  unsigned packed = 0xFFFFu;
  if (v0.valid && v1.valid && v2.valid) {
    const int Ymin = min(v0.Y, min(v1.Y, v2.Y)), Ymax = max(v0.Y, max(v1.Y, v2.Y));
    const int ymin = max(0, (Ymin - 128 + 255) >> 8), ymax = min(h - 1, (Ymax - 128) >> 8);
    if (ymin <= ymax) packed = (unsigned)ymin | ((unsigned)ymax << 16);
  }
  bounds[(size_t)view * max_faces + t] = packed;
}

struct TriSetup {
  // E_i(sx,sy) = A_i*sx + B_i*sy + C_i over fixed-point sample coordinates.  All quantities are integers below 2^47, held
  // in fp64 where they (and every partial sum of the two-FMA evaluation) are exact; fp64 FMA runs at half the fp32 rate on
  // gfx950, whereas 64-bit integer multiplies decompose into quarter-rate 32-bit multiplies.
  double A0, B0, C0, A1, B1, C1, A2, B2, C2;
  double t0, t1, t2;  // fill-rule thresholds: sample is inside iff E_i >= t_i (0 or 1)
  float inv_area;     // 1 / (float)(2 * signed area) -- barycentric b_i = (float)E_i * inv_area
  float iz0, iz1, iz2;
  int xmin, xmax, ymin, ymax;  // pixel bbox (inclusive), clipped to the band
  bool ok;
};

// Edge a->b: E(p) = (bx-ax)*(py-ay) - (by-ay)*(px-ax).  With y pointing down and a positively oriented
// triangle the interior is E >= 0; an edge is "left" if it goes up (dy < 0) and "top" if dy == 0 && dx > 0.
// Top-left edges own their boundary samples (threshold 0); the others exclude E == 0 (threshold 1).
__device__ __forceinline__ void edge_setup(int ax, int ay, int bx, int by, double& A, double& B, double& C, double& thr) {
  const int dx = bx - ax, dy = by - ay;  // |.| < 2^24
  A = -(double)dy;
  B = (double)dx;
  C = (double)dy * (double)ax - (double)dx * (double)ay;  // exact: products < 2^47
  thr = ((dy < 0) || (dy == 0 && dx > 0)) ? 0.0 : 1.0;
}

// v1/v2 (and i1/i2) are swapped in place when the screen-space orientation is negative (two-sided rendering).
__device__ __forceinline__ TriSetup tri_setup(const VtxRec& v0, VtxRec& v1, VtxRec& v2, int& i1, int& i2, int w, int y_lo,
                                              int y_hi) {
  TriSetup s;
  s.ok = false;
  if (!(v0.valid && v1.valid && v2.valid)) return s;
  double area = (double)(v1.X - v0.X) * (double)(v2.Y - v0.Y) - (double)(v1.Y - v0.Y) * (double)(v2.X - v0.X);  // exact
  if (area == 0.0) return s;
  if (area < 0.0) {
    const VtxRec t = v1; v1 = v2; v2 = t;
    const int ti = i1; i1 = i2; i2 = ti;
    area = -area;
  }
  const int Xmin = min(v0.X, min(v1.X, v2.X)), Xmax = max(v0.X, max(v1.X, v2.X));
  const int Ymin = min(v0.Y, min(v1.Y, v2.Y)), Ymax = max(v0.Y, max(v1.Y, v2.Y));
  // samples sit at x*256+128: ceil((Xmin-128)/256) .. floor((Xmax-128)/256)
  s.xmin = max(0, (Xmin - 128 + 255) >> 8);
  s.xmax = min(w - 1, (Xmax - 128) >> 8);
  s.ymin = max(y_lo, (Ymin - 128 + 255) >> 8);
  s.ymax = min(y_hi, (Ymax - 128) >> 8);
  if (s.xmin > s.xmax || s.ymin > s.ymax) return s;
  edge_setup(v1.X, v1.Y, v2.X, v2.Y, s.A0, s.B0, s.C0, s.t0);  // edge opposite vertex 0
  edge_setup(v2.X, v2.Y, v0.X, v0.Y, s.A1, s.B1, s.C1, s.t1);
  edge_setup(v0.X, v0.Y, v1.X, v1.Y, s.A2, s.B2, s.C2, s.t2);
  s.inv_area = 1.0f / (float)area;
  s.iz0 = v0.invz;
  s.iz1 = v1.invz;
  s.iz2 = v2.invz;
  s.ok = true;
  return s;
}

__device__ __forceinline__ bool sample_tri(const TriSetup& s, int px, int py, float& b0, float& b1, float& b2, float& wsum) {
  const double sx = (double)(px * SUBPIX + 128), sy = (double)(py * SUBPIX + 128);
  const double e0 = fma(s.A0, sx, fma(s.B0, sy, s.C0));  // exact integer arithmetic in fp64
  const double e1 = fma(s.A1, sx, fma(s.B1, sy, s.C1));
  const double e2 = fma(s.A2, sx, fma(s.B2, sy, s.C2));
  if (e0 < s.t0 || e1 < s.t1 || e2 < s.t2) return false;
  b0 = (float)e0 * s.inv_area;
  b1 = (float)e1 * s.inv_area;
  b2 = (float)e2 * s.inv_area;
  wsum = fmaf(b2, s.iz2, fmaf(b1, s.iz1, b0 * s.iz0));
  return true;
}

__device__ __forceinline__ void raster_pixel(const TriSetup& s, int px, int py, int tri, int band_y0, int w,
                                             unsigned long long* zbuf) {
  float b0, b1, b2, wsum;
  if (!sample_tri(s, px, py, b0, b1, b2, wsum)) return;
  if (!(wsum >= 1.0f / Z_FAR && wsum <= 1.0f / Z_NEAR)) return;
  const unsigned long long key = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(wsum)) << 32) | (unsigned)tri;
  atomicMin(&zbuf[(py - band_y0) * w + px], key);
}

__device__ __forceinline__ float lut_val(int i) { return (float)((i * 255) >> 5); }  // floor(i*255/32), utils.py:65

// Eye-normal LUT lookup: 32-texel separable ramp, GL_LINEAR filter, repeat wrap; returns the value on the 0..255 scale
__device__ __forceinline__ float normal_lut(float n) {
  const float u = n - floorf(n);
  const float t = fmaf(u, 32.0f, -0.5f);
  const float fl = floorf(t);
  const float f = t - fl;
  const int i0 = ((int)fl + 32) & 31;
  const int i1 = (i0 + 1) & 31;
  const float a = lut_val(i0), b = lut_val(i1);
  return fmaf(b - a, f, a);
}

// Texture sample (contract shared with oracle/raster.c): repeat wrap, texel centres at (i + .5) / size, bilinear, result on the
// 0..255 scale per channel.
__device__ __forceinline__ void tex_sample(const TexDev& m, int level, float u, float v, float& r, float& g, float& b) {
  const int tw = max(1, m.tex_w >> level), th = max(1, m.tex_h >> level);
  const uint32_t* tx = m.texels + m.tex_off[level];
  const float fu = fmaf(u - floorf(u), (float)tw, -0.5f), fv = fmaf(v - floorf(v), (float)th, -0.5f);
  const float flu = floorf(fu), flv = floorf(fv);
  const float au = fu - flu, av = fv - flv;
  int x0 = (int)flu, y0 = (int)flv;
  if (x0 < 0) x0 = tw - 1;
  if (y0 < 0) y0 = th - 1;
  if (x0 >= tw) x0 = tw - 1;   // u - floor(u) can round to 1.0f for tiny negative u
  if (y0 >= th) y0 = th - 1;
  const int x1 = (x0 + 1 == tw) ? 0 : x0 + 1, y1 = (y0 + 1 == th) ? 0 : y0 + 1;
  const uint32_t t00 = tx[y0 * tw + x0], t01 = tx[y0 * tw + x1], t10 = tx[y1 * tw + x0], t11 = tx[y1 * tw + x1];
  float out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a00 = (float)((t00 >> (8 * c)) & 255u), a01 = (float)((t01 >> (8 * c)) & 255u);
    const float a10 = (float)((t10 >> (8 * c)) & 255u), a11 = (float)((t11 >> (8 * c)) & 255u);
    const float top = fmaf(a01 - a00, au, a00), bot = fmaf(a11 - a10, au, a10);
    out[c] = fmaf(bot - top, av, top);
  }
  r = out[0]; g = out[1]; b = out[2];
}

// mip level of a triangle: texels per pixel r = |uv area| * w * h / (screen area); level = #thresholds {2, 8, 32, ...} below r
__device__ __forceinline__ int tex_level(const TexDev& m, const float* uv, float inv_area2) {
  const float du1 = uv[2] - uv[0], dv1 = uv[3] - uv[1], du2 = uv[4] - uv[0], dv2 = uv[5] - uv[1];
  const float at = fabsf(du1 * dv2 - du2 * dv1) * ((float)m.tex_w * (float)m.tex_h);
  const float r = at * (inv_area2 * 65536.0f);   // inv_area2 = 1 / (2 * area in 1/256-px units)
  int level = 0;
  float thr = 2.0f;
  while (level + 1 < m.tex_levels && r > thr) { ++level; thr *= 4.0f; }
  return level;
}

__device__ __forceinline__ float quant8(float v255) {
  const float q = floorf(fminf(fmaxf(v255, 0.f), 255.f) + 0.5f);
  return q / 255.0f;
}

template <bool kUnused = false>
__global__ __launch_bounds__(BAND_THREADS) void raster_bands(
    const MeshDev* __restrict__ meshes, const TexDev* __restrict__ texs, const int32_t* __restrict__ mesh_ids, const float* __restrict__ TCO,
    const VtxRec* __restrict__ vtx, const unsigned* __restrict__ bounds, int max_verts, int max_faces, int h, int w, uint32_t flags,
    LightsDev lights, float* __restrict__ out, long long stride_v, int views_per_item, int n_items, long long stride_view, long long stride_y,
    long long stride_x, int c_rgb, int c_normals, int c_depth, CropArgs crop) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long zbuf[];  // [BAND_H*w] + big-triangle queue
  int* big_queue = (int*)(zbuf + (size_t)BAND_H * w);
  int* list = big_queue + BIG_QUEUE;  // [LIST_CAP]
  float* stage = (float*)big_queue;   // [BAND_THREADS][STAGE_CH]: resolve-phase staging, aliases the (then dead) queue + list
  static_assert((BIG_QUEUE + LIST_CAP) * sizeof(int) >= BAND_THREADS * STAGE_CH * sizeof(float), "stage must fit in queue + list");
  __shared__ int list_n;
  __shared__ int q_head;
  __shared__ int big_count;

  // Work-to-workgroup map: consecutive workgroup ids go round-robin to the 8 XCDs, and the `views_per_item` views of one
  // (item, band) write interleaved channel slices of the SAME pixel lines of the CNN input.  Placing them in consecutive slots
  // of one XCD lets its L2 merge the 24-byte slices into whole lines before they are written back (otherwise every slice is a
  // partial-line write from a different L2).
  const int n_bands = (h + BAND_H - 1) / BAND_H;
  const int roles = views_per_item + (crop.images ? 1 : 0);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = (slot / roles) * 8 + xcd;   // (item, band) index
  if (q >= n_items * n_bands) return;
  const int band = q % n_bands;
  const int y0 = band * BAND_H;
  if (slot % roles == views_per_item) {  // crop role (whole workgroup): the band's rows of the observation crop
    const int item = q / n_bands;
    const int yl = min(h, y0 + BAND_H) - 1;
    const float* bx = crop.boxes + (size_t)item * 4;
    const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    const float roi_w = fmaxf(x2 - x1, 1.0f), roi_h = fmaxf(y2 - y1, 1.0f);
    const float bin_h = roi_h / (float)h, bin_w = roi_w / (float)w;
    const float* img = crop.images + (size_t)crop.im_ids[item] * (crop.nhwc4 ? 4 : crop.C) * crop.H * crop.W;
    float* o_item = out + (size_t)item * stride_v + crop.c0;
    for (int i = threadIdx.x; i < (yl - y0 + 1) * w; i += BAND_THREADS) {
      const int py = y0 + i / w, px = i % w;
      float* o = o_item + (size_t)py * stride_y + (size_t)px * stride_x;
      if (crop.nhwc4) {
        if (crop.C == 4) crop_pixel<4, true>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, o);
        else crop_pixel<3, true>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, o);
      } else {
        if (crop.C == 4) crop_pixel<4, false>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, o);
        else crop_pixel<3, false>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, o);
      }
    }
    return;
  }
  const int view = (q / n_bands) * views_per_item + slot % roles;
  const int y1 = min(h, y0 + BAND_H) - 1;
  const int npix = (y1 - y0 + 1) * w;
  const MeshDev m = meshes[mesh_ids[view]];
  const VtxRec* vv = vtx + (size_t)view * max_verts;
  const unsigned* tb = bounds + (size_t)view * max_faces;

  for (int i = threadIdx.x; i < npix; i += BAND_THREADS) zbuf[i] = ~0ull;
  if (threadIdx.x == 0) { big_count = 0; q_head = 0; }
  __syncthreads();

  // ---- pass 1: triangle-parallel coverage + depth ------------------------------------------------
  // Two steps so that the expensive part runs on dense lanes: (a) scan the packed row
  // bounds and compact the triangles overlapping this band into an LDS list (wave ballot + one LDS atomic per wave),
  // (b) every thread takes list entries.  (A plain "if (!overlap) continue" loop leaves ~1 lane in 9 active.)
  const int lane = threadIdx.x & 63;
  // The list is only drained when the next scan chunk could overflow it (or at the end), so that the drain runs with (nearly)
  // all 512 threads busy instead of once per scan chunk with a fifth of them.
  const int n_faces_eff = m.n_faces;
  if (threadIdx.x == 0) list_n = 0;
  __syncthreads();
  // scan: one 16-byte load = the packed bounds of 4 consecutive triangles per thread and chunk (the bounds array is padded to a
  // multiple of 4 entries per view with "culled"), the next chunk's load is issued before the current one is compacted
  const uint4* tb4 = reinterpret_cast<const uint4*>(tb);
  const uint4 culled4 = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
  uint4 nxt = (4 * (int)threadIdx.x < n_faces_eff) ? tb4[threadIdx.x] : culled4;
  for (int c0 = 0; c0 < n_faces_eff; c0 += SCAN_CHUNK) {
    const int c1 = min(n_faces_eff, c0 + SCAN_CHUNK);
    const uint4 cur = nxt;
    const int tn = c0 + SCAN_CHUNK + 4 * (int)threadIdx.x;
    nxt = (tn < n_faces_eff) ? tb4[tn >> 2] : culled4;
    const int t0 = c0 + 4 * (int)threadIdx.x;
    const unsigned pbs[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned pb = pbs[k];
      const bool hit = (t0 + k < c1) && (int)(pb & 0xFFFFu) <= y1 && (int)(pb >> 16) >= y0;  // culled triangles carry ymin = 0xFFFF
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        int wbase = 0;
        if (lane == 0) wbase = atomicAdd(&list_n, __popcll(mask));
        wbase = __shfl(wbase, 0);
        if (hit) list[wbase + __popcll(mask & ((1ull << lane) - 1ull))] = t0 + k;
      }
    }
    __syncthreads();
    const int n_list = list_n;
    if (n_list + SCAN_CHUNK <= LIST_CAP && c1 < n_faces_eff) continue;   // room for another scan chunk: keep collecting
    for (int e = threadIdx.x; e < n_list; e += BAND_THREADS) {
      const int t = list[e];
      int i0 = m.faces[3 * t], i1 = m.faces[3 * t + 1], i2 = m.faces[3 * t + 2];
      const VtxRec v0 = vv[i0];
      VtxRec v1 = vv[i1], v2 = vv[i2];
      const TriSetup s = tri_setup(v0, v1, v2, i1, i2, w, y0, y1);
      if (!s.ok) continue;
      const int area = (s.xmax - s.xmin + 1) * (s.ymax - s.ymin + 1);
      if (area > BIG_TRI_AREA) {
        const int slot = atomicAdd(&big_count, 1);
        if (slot < BIG_QUEUE) {
          big_queue[slot] = t;
          continue;
        }
      }
      for (int py = s.ymin; py <= s.ymax; ++py)
        for (int px = s.xmin; px <= s.xmax; ++px) raster_pixel(s, px, py, t, y0, w, zbuf);
    }
    __syncthreads();
    if (threadIdx.x == 0) list_n = 0;
    __syncthreads();
  }
  // ---- pass 1b: larger triangles: waves pull them from the queue, 64 lanes share one bbox -----------------
  const int nbig = min(big_count, BIG_QUEUE);
  for (;;) {
    int q = 0;
    if (lane == 0) q = atomicAdd(&q_head, 1);
    q = __shfl(q, 0);
    if (q >= nbig) break;
    const int t = big_queue[q];
    int i0 = m.faces[3 * t], i1 = m.faces[3 * t + 1], i2 = m.faces[3 * t + 2];
    const VtxRec v0 = vv[i0];
    VtxRec v1 = vv[i1], v2 = vv[i2];
    const TriSetup s = tri_setup(v0, v1, v2, i1, i2, w, y0, y1);
    const int bw = s.xmax - s.xmin + 1, bh = s.ymax - s.ymin + 1;
    for (int i = lane; i < bw * bh; i += 64) raster_pixel(s, s.xmin + i % bw, s.ymin + i / bw, t, y0, w, zbuf);
  }
  __syncthreads();

  // ---- pass 2: resolve + shade, pixel-parallel ----------------------------------------------------
  const float* T = TCO + (size_t)view * 16;
  const bool do_norm = (flags & MP_RASTER_NORMALS) && c_normals >= 0;
  const bool do_depth = (flags & MP_RASTER_DEPTH) && c_depth >= 0;
  const bool gl_eye = flags & MP_RASTER_NORMALS_GL;
  const bool no_quant = flags & MP_RASTER_NO_QUANT;
  float* out_v = out + (size_t)(view / views_per_item) * stride_v + (size_t)(view % views_per_item) * stride_view;
  // channel map of the staged values -> output channel (only the enabled groups), so the store loop below can walk
  // (pixel, channel) pairs with consecutive lanes on consecutive addresses (24/28-byte pieces instead of 4-byte scatters)
  int n_ch = 0;
  int ch_src[STAGE_CH], ch_dst[STAGE_CH];
  if (c_rgb >= 0) { for (int k = 0; k < 3; ++k) { ch_src[n_ch] = k; ch_dst[n_ch++] = c_rgb + k; } }
  if (do_norm) { for (int k = 0; k < 3; ++k) { ch_src[n_ch] = 3 + k; ch_dst[n_ch++] = c_normals + k; } }
  if (do_depth) { ch_src[n_ch] = 6; ch_dst[n_ch++] = c_depth; }
  for (int base = 0; base < npix; base += BAND_THREADS) {
    const int i = base + threadIdx.x;
    const bool live = i < npix;
    const int py = y0 + (live ? i : 0) / w, px = (live ? i : 0) % w;
    const unsigned long long key = live ? zbuf[i] : ~0ull;
    float r = 0.f, g = 0.f, b = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, depth = 0.f;
    if (key != ~0ull) {
      const int t = (int)(key & 0xFFFFFFFFu);
      int i0 = m.faces[3 * t], i1 = m.faces[3 * t + 1], i2 = m.faces[3 * t + 2];
      const VtxRec v0 = vv[i0];
      VtxRec v1 = vv[i1], v2 = vv[i2];
      const int i1_in = i1;
      const TriSetup s = tri_setup(v0, v1, v2, i1, i2, w, py, py);
      float b0, b1, b2, wsum;
      (void)sample_tri(s, px, py, b0, b1, b2, wsum);
      const float w0 = b0 * s.iz0, w1 = b1 * s.iz1, w2 = b2 * s.iz2;
      const float z = 1.0f / wsum;
      depth = z;
      const float* c0 = m.colors + 3 * i0; const float* c1 = m.colors + 3 * i1; const float* c2 = m.colors + 3 * i2;
      float ar = fmaf(w2, c2[0], fmaf(w1, c1[0], w0 * c0[0])) * z;
      float ag = fmaf(w2, c2[1], fmaf(w1, c1[1], w0 * c0[1])) * z;
      float ab = fmaf(w2, c2[2], fmaf(w1, c1[2], w0 * c0[2])) * z;
      if (m.uvs) {
        const float* uv = m.uvs + 6 * (size_t)t;
        const int k1 = (i1 == i1_in) ? 1 : 2, k2 = 3 - k1;   // corner slots follow the orientation swap
        const float u = fmaf(w2, uv[2 * k2], fmaf(w1, uv[2 * k1], w0 * uv[0])) * z;
        const float v = fmaf(w2, uv[2 * k2 + 1], fmaf(w1, uv[2 * k1 + 1], w0 * uv[1])) * z;
        float tr, tg, tb2;
        const TexDev& tx = texs[mesh_ids[view]];
        tex_sample(tx, tex_level(tx, uv, s.inv_area), u, v, tr, tg, tb2);
        ar *= tr / 255.0f; ag *= tg / 255.0f; ab *= tb2 / 255.0f;
      }
      const float* n0 = m.normals + 3 * i0; const float* n1 = m.normals + 3 * i1; const float* n2 = m.normals + 3 * i2;
      // interpolated object-frame normal (perspective-correct, NOT renormalised: texcoord semantics)
      const float onx = fmaf(w2, n2[0], fmaf(w1, n1[0], w0 * n0[0])) * z;
      const float ony = fmaf(w2, n2[1], fmaf(w1, n1[1], w0 * n0[1])) * z;
      const float onz = fmaf(w2, n2[2], fmaf(w1, n1[2], w0 * n0[2])) * z;
      float lr = lights.ambient[0], lg = lights.ambient[1], lb = lights.ambient[2];
      if (lights.n_point > 0) {
        const float* p0 = m.verts + 3 * i0; const float* p1 = m.verts + 3 * i1; const float* p2 = m.verts + 3 * i2;
        const float ox = fmaf(w2, p2[0], fmaf(w1, p1[0], w0 * p0[0])) * z;
        const float oy = fmaf(w2, p2[1], fmaf(w1, p1[1], w0 * p0[1])) * z;
        const float oz = fmaf(w2, p2[2], fmaf(w1, p1[2], w0 * p0[2])) * z;
        const float nn = sqrtf(fmaf(onz, onz, fmaf(ony, ony, onx * onx)));
        const float inn = nn > 0.f ? 1.0f / nn : 0.f;
        const float R10 = 10.0f * m.radius;
        for (int l = 0; l < lights.n_point; ++l) {
          const float lx = fmaf(lights.dir[l][0], R10, -ox);
          const float ly = fmaf(lights.dir[l][1], R10, -oy);
          const float lz = fmaf(lights.dir[l][2], R10, -oz);
          const float ln = sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
          const float d = fmaf(lz, onz, fmaf(ly, ony, lx * onx)) * inn / ln;
          const float dd = fmaxf(d, 0.f);
          lr = fmaf(lights.color[l][0], dd, lr);
          lg = fmaf(lights.color[l][1], dd, lg);
          lb = fmaf(lights.color[l][2], dd, lb);
        }
      }
      ar *= lr; ag *= lg; ab *= lb;
      if (no_quant) { r = ar; g = ag; b = ab; }
      else { r = quant8(ar * 255.0f); g = quant8(ag * 255.0f); b = quant8(ab * 255.0f); }
      if (do_norm) {
        // eye-space normal: camera (OpenCV) frame first, then the eye-axis convention
        const float cx = fmaf(T[2], onz, fmaf(T[1], ony, T[0] * onx));
        const float cy = fmaf(T[6], onz, fmaf(T[5], ony, T[4] * onx));
        const float cz = fmaf(T[10], onz, fmaf(T[9], ony, T[8] * onx));
        float ex, ey, ez;
        if (gl_eye) { ex = cx; ey = -cy; ez = -cz; }   // GL eye: x right, y up, z back
        else { ex = cx; ey = cz; ez = -cy; }           // Panda view: x right, y forward, z up (TCCGL, types.py:40)
        if (no_quant) { nx = normal_lut(ex) / 255.0f; ny = normal_lut(ey) / 255.0f; nz = normal_lut(ez) / 255.0f; }
        else { nx = quant8(normal_lut(ex)); ny = quant8(normal_lut(ey)); nz = quant8(normal_lut(ez)); }
      }
    }
    float* st = stage + threadIdx.x * STAGE_CH;
    st[0] = r; st[1] = g; st[2] = b; st[3] = nx; st[4] = ny; st[5] = nz; st[6] = depth;
    __syncthreads();
    const int n_here = min(BAND_THREADS, npix - base);
    // 8 lanes per pixel (channel slot k = lane & 7, active while k < n_ch), 64 pixels per pass: consecutive lanes write
    // consecutive floats of one pixel, no per-element integer division (the chunk's first (row, col) is wave-uniform).
    const int k = threadIdx.x & 7;
    int src = ch_src[0], dst = ch_dst[0];
#pragma unroll
    for (int q = 1; q < STAGE_CH; ++q) if (k == q) { src = ch_src[q]; dst = ch_dst[q]; }
    const int row0 = base / w, col0 = base - row0 * w;
    for (int pl = threadIdx.x >> 3; pl < n_here; pl += BAND_THREADS / 8) {
      int col = col0 + pl, rowp = row0;
      while (col >= w) { col -= w; ++rowp; }
      if (k < n_ch) out_v[(size_t)(y0 + rowp) * stride_y + (size_t)col * stride_x + dst] = stage[pl * STAGE_CH + src];
    }
    __syncthreads();
  }
}

}  // namespace mp

using namespace mp;

struct mp_mesh_db {
  int n;
  int max_verts, max_faces;
  MeshDev* d_meshes;
  TexDev* d_texs;
  std::vector<MeshDev> h_meshes;
  std::vector<TexDev> h_texs;
  std::vector<void*> allocs;
};

extern "C" int mp_mesh_db_create(const mp_mesh_desc* hm, int n, mp_mesh_db** out) {
  MP_REQUIRE(hm && out && n > 0, "mp_mesh_db_create: bad arguments");
  mp_mesh_db* db = new mp_mesh_db();
  db->n = n;
  db->max_verts = db->max_faces = 0;
  db->d_meshes = nullptr;
  db->d_texs = nullptr;
  for (int i = 0; i < n; ++i) {
    const mp_mesh_desc& d = hm[i];
    MP_REQUIRE(d.h_vertices && d.h_normals && d.h_colors && d.h_faces && d.n_vertices > 0 && d.n_faces > 0,
               "mp_mesh_db_create: mesh %d incomplete", i);
    for (int f = 0; f < 3 * d.n_faces; ++f)
      MP_REQUIRE(d.h_faces[f] >= 0 && d.h_faces[f] < d.n_vertices, "mp_mesh_db_create: mesh %d face index out of range", i);
    MeshDev m;
    float *dv, *dn, *dc;
    int32_t* df;
    const size_t vb = (size_t)d.n_vertices * 3 * sizeof(float);
    MP_CHECK_HIP(hipMalloc(&dv, vb));
    MP_CHECK_HIP(hipMalloc(&dn, vb));
    MP_CHECK_HIP(hipMalloc(&dc, vb));
    MP_CHECK_HIP(hipMalloc(&df, (size_t)d.n_faces * 3 * sizeof(int32_t)));
    MP_CHECK_HIP(hipMemcpy(dv, d.h_vertices, vb, hipMemcpyHostToDevice));
    MP_CHECK_HIP(hipMemcpy(dn, d.h_normals, vb, hipMemcpyHostToDevice));
    MP_CHECK_HIP(hipMemcpy(dc, d.h_colors, vb, hipMemcpyHostToDevice));
    MP_CHECK_HIP(hipMemcpy(df, d.h_faces, (size_t)d.n_faces * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
    db->allocs.push_back(dv); db->allocs.push_back(dn); db->allocs.push_back(dc); db->allocs.push_back(df);
    m.verts = dv; m.normals = dn; m.colors = dc; m.faces = df;
    m.n_verts = d.n_vertices; m.n_faces = d.n_faces;
    m.uvs = nullptr;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = 0; v < d.n_vertices; ++v)
      for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(lo[k], d.h_vertices[3 * v + k]);
        hi[k] = fmaxf(hi[k], d.h_vertices[3 * v + k]);
      }
    for (int k = 0; k < 3; ++k) m.center[k] = 0.5f * (lo[k] + hi[k]);
    float r2 = 0.f;
    for (int v = 0; v < d.n_vertices; ++v) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) {
        const float dd = d.h_vertices[3 * v + k] - m.center[k];
        s += dd * dd;
      }
      r2 = fmaxf(r2, s);
    }
    m.radius = sqrtf(r2);
    db->h_meshes.push_back(m);
    db->max_verts = std::max(db->max_verts, d.n_vertices);
    db->max_faces = std::max(db->max_faces, d.n_faces);
  }
  MP_CHECK_HIP(hipMalloc(&db->d_meshes, n * sizeof(MeshDev)));
  MP_CHECK_HIP(hipMemcpy(db->d_meshes, db->h_meshes.data(), n * sizeof(MeshDev), hipMemcpyHostToDevice));
  db->h_texs.assign(n, TexDev{});
  MP_CHECK_HIP(hipMalloc(&db->d_texs, n * sizeof(TexDev)));
  MP_CHECK_HIP(hipMemcpy(db->d_texs, db->h_texs.data(), n * sizeof(TexDev), hipMemcpyHostToDevice));
  *out = db;
  return MP_OK;
}

extern "C" int mp_mesh_db_set_texture(mp_mesh_db* db, int mesh_id, const float* h_uvs, const uint32_t* h_texels, int tex_w, int tex_h,
                                      int n_levels) {
  MP_REQUIRE(db && mesh_id >= 0 && mesh_id < db->n && h_uvs && h_texels, "mp_mesh_db_set_texture: bad arguments");
  MP_REQUIRE(tex_w > 0 && tex_h > 0 && tex_w <= 16384 && tex_h <= 16384 && n_levels >= 1 && n_levels <= MP_TEX_MAX_LEVELS,
             "mp_mesh_db_set_texture: bad texture size %dx%d / %d levels", tex_w, tex_h, n_levels);
  MeshDev& m = db->h_meshes[mesh_id];
  TexDev& tx = db->h_texs[mesh_id];
  size_t total = 0;
  for (int l = 0; l < n_levels; ++l) {
    tx.tex_off[l] = (int)total;
    total += (size_t)std::max(1, tex_w >> l) * std::max(1, tex_h >> l);
  }
  float* duv;
  uint32_t* dtex;
  MP_CHECK_HIP(hipMalloc(&duv, (size_t)m.n_faces * 6 * sizeof(float)));
  MP_CHECK_HIP(hipMalloc(&dtex, total * sizeof(uint32_t)));
  MP_CHECK_HIP(hipMemcpy(duv, h_uvs, (size_t)m.n_faces * 6 * sizeof(float), hipMemcpyHostToDevice));
  MP_CHECK_HIP(hipMemcpy(dtex, h_texels, total * sizeof(uint32_t), hipMemcpyHostToDevice));
  db->allocs.push_back(duv); db->allocs.push_back(dtex);
  m.uvs = duv;
  tx.texels = dtex; tx.tex_w = tex_w; tx.tex_h = tex_h; tx.tex_levels = n_levels;
  MP_CHECK_HIP(hipMemcpy(db->d_texs + mesh_id, &tx, sizeof(TexDev), hipMemcpyHostToDevice));
  MP_CHECK_HIP(hipMemcpy(db->d_meshes + mesh_id, &m, sizeof(MeshDev), hipMemcpyHostToDevice));
  return MP_OK;
}

extern "C" int mp_mesh_db_destroy(mp_mesh_db* db) {
  if (!db) return MP_OK;
  for (void* p : db->allocs) (void)hipFree(p);
  if (db->d_meshes) (void)hipFree(db->d_meshes);
  if (db->d_texs) (void)hipFree(db->d_texs);
  delete db;
  return MP_OK;
}

extern "C" int mp_mesh_db_max_vertices(const mp_mesh_db* db) { return db ? db->max_verts : 0; }
extern "C" float mp_mesh_db_radius(const mp_mesh_db* db, int i) {
  return (db && i >= 0 && i < db->n) ? db->h_meshes[i].radius : 0.f;
}

static inline int faces_stride(const mp_mesh_db* db) { return (db->max_faces + 3) & ~3; }  // 16-byte aligned bounds rows

extern "C" size_t mp_raster_workspace_bytes(const mp_mesh_db* db, int n_views) {
  return db ? (size_t)n_views * ((size_t)db->max_verts * sizeof(VtxRec) + (size_t)faces_stride(db) * sizeof(unsigned)) : 0;
}

static int raster_render_impl(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K,
                              int n_views, int h, int w, uint32_t flags, const mp_lights* lights, float* d_out,
                              int64_t stride_v, int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x,
                              int c_rgb, int c_normals, int c_depth, void* d_ws, size_t ws_bytes, mp_stream stream, const CropArgs& crop) {
  MP_REQUIRE(db && d_mesh_ids && d_TCO && d_K && d_out && lights, "mp_raster_render: null pointer");
  MP_REQUIRE(n_views >= 0 && h > 0 && w > 0 && w <= 1024 && views_per_item >= 1, "mp_raster_render: bad size");
  MP_REQUIRE(lights->n_point >= 0 && lights->n_point <= 8, "mp_raster_render: too many point lights");
  if (n_views == 0) return MP_OK;
  MP_REQUIRE(ws_bytes >= mp_raster_workspace_bytes(db, n_views), "mp_raster_render: workspace too small");
  MP_REQUIRE(n_views <= 65535, "mp_raster_render: at most 65535 views per call");
  hipStream_t s = (hipStream_t)stream;
  LightsDev L;
  memcpy(L.ambient, lights->ambient, sizeof(L.ambient));
  L.n_point = lights->n_point;
  memcpy(L.dir, lights->point_dir, sizeof(L.dir));
  memcpy(L.color, lights->point_color, sizeof(L.color));
  VtxRec* vtx = (VtxRec*)d_ws;
  dim3 g1(ceil_div(db->max_verts, 256), n_views);
  {
  ProfScope prof("raster_transform", 0.0, (double)n_views * db->max_verts * (12.0 + sizeof(VtxRec)), s);
  hipLaunchKernelGGL(raster_transform, g1, dim3(256), 0, s, db->d_meshes, d_mesh_ids, d_TCO, d_K, db->max_verts, vtx);
  }
  unsigned* tri_bounds = (unsigned*)(vtx + (size_t)n_views * db->max_verts);
  {
    ProfScope prof("raster_tri_bounds", 0.0, (double)n_views * db->max_faces * (12.0 + 4.0), s);
    hipLaunchKernelGGL(raster_tri_bounds, dim3(ceil_div(db->max_faces, 256), n_views), dim3(256), 0, s, db->d_meshes, d_mesh_ids, vtx,
                       db->max_verts, faces_stride(db), h, tri_bounds);
  }
  const size_t lds = (size_t)BAND_H * w * sizeof(unsigned long long) + (BIG_QUEUE + LIST_CAP) * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)raster_bands<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    attr_set = true;
  }
  MP_REQUIRE(lds <= 160 * 1024 - 64, "mp_raster_render: image too wide for the LDS z-buffer");
  MP_REQUIRE(n_views % views_per_item == 0, "mp_raster_render: n_views (%d) must be a multiple of views_per_item (%d)", n_views, views_per_item);
  const int n_items = n_views / views_per_item;
  const int roles = views_per_item + (crop.images ? 1 : 0);
  dim3 g2(ceil_div(n_items * ceil_div(h, BAND_H), 8) * 8 * roles);
  const int n_ch = (c_rgb >= 0 ? 3 : 0) + (((flags & MP_RASTER_NORMALS) && c_normals >= 0) ? 3 : 0) + (((flags & MP_RASTER_DEPTH) && c_depth >= 0) ? 1 : 0);
  // algorithmic bytes: output channels written once + the mesh (32 B/vertex, 12 B/triangle) read once per view
  // (+ the fused crop role: C output channels written + at most the same-sized source window read per item)
  const double crop_bytes = crop.images ? (double)n_items * 2.0 * crop.C * 4.0 * h * w : 0.0;
  ProfScope prof("raster_bands", 0.0,
                 (double)n_views * ((double)n_ch * 4.0 * h * w + 32.0 * db->max_verts + 12.0 * db->max_faces) + crop_bytes, s);
  hipLaunchKernelGGL(raster_bands<false>, g2, dim3(BAND_THREADS), lds, s, db->d_meshes, db->d_texs, d_mesh_ids, d_TCO, vtx, tri_bounds,
                     db->max_verts, faces_stride(db), h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items, (long long)stride_view, (long long)stride_y,
                     (long long)stride_x, c_rgb, c_normals, c_depth, crop);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

extern "C" int mp_raster_render(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K,
                                int n_views, int h, int w, uint32_t flags, const mp_lights* lights, float* d_out,
                                int64_t stride_v, int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x,
                                int c_rgb, int c_normals, int c_depth, void* d_ws, size_t ws_bytes, mp_stream stream) {
  CropArgs none;
  memset(&none, 0, sizeof(none));
  return raster_render_impl(db, d_mesh_ids, d_TCO, d_K, n_views, h, w, flags, lights, d_out, stride_v, views_per_item, stride_view, stride_y,
                            stride_x, c_rgb, c_normals, c_depth, d_ws, ws_bytes, stream, none);
}

extern "C" int mp_raster_render_crop(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K,
                                     int n_views, int h, int w, uint32_t flags, const mp_lights* lights, float* d_out,
                                     int64_t stride_v, int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x,
                                     int c_rgb, int c_normals, int c_depth, void* d_ws, size_t ws_bytes,
                                     const float* d_images, int images_nhwc4, int n_im, int C, int H, int W,
                                     const int32_t* d_im_ids, const float* d_boxes, int c0_crop, mp_stream stream) {
  MP_REQUIRE(d_images && d_im_ids && d_boxes && n_im > 0 && (C == 3 || C == 4) && H > 0 && W > 0 && c0_crop >= 0,
             "mp_raster_render_crop: bad crop arguments");
  CropArgs crop;
  crop.images = d_images; crop.im_ids = d_im_ids; crop.boxes = d_boxes;
  crop.C = C; crop.H = H; crop.W = W; crop.c0 = c0_crop; crop.nhwc4 = images_nhwc4 ? 1 : 0;
  return raster_render_impl(db, d_mesh_ids, d_TCO, d_K, n_views, h, w, flags, lights, d_out, stride_v, views_per_item, stride_view, stride_y,
                            stride_x, c_rgb, c_normals, c_depth, d_ws, ws_bytes, stream, crop);
}
