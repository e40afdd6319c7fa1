// raster_core.h -- the arithmetic of the rasteriser's pixel contract (v2), shared by the HIP kernels (raster.hip) and by a
// host emulation of those kernels that the CPU tests run against the independent oracle (oracle/raster.c) without a GPU
// (tests/raster_emul.cpp).  Everything here is plain scalar code: geometry set-up incl. near-plane clipping, exact edge
// functions, the per-sample depth rule, per-(pixel, piece) shading and the multisample resolve.  The contract itself is stated
// at the top of oracle/raster.c; reference lines: panda3d_renderer/panda3d_scene_renderer.py:71-74, 99-136, 210-216,
// panda3d_batch_renderer.py:109-135, 261-274, types.py:63-64, utils.py:44-68.
// Built with -ffp-contract=off on both sides: every fused operation is an explicit fmaf().
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MP_MUL24(a, b) __mul24((a), (b))   // full-rate 24-bit multiply; callers guarantee |a|, |b| < 2^23 and |a * b| < 2^31
#else
#define MP_MUL24(a, b) ((a) * (b))
#endif

#ifndef MP_TEX_MAX_LEVELS
#define MP_TEX_MAX_LEVELS 15
#endif

namespace mp {
namespace rc {

constexpr int SUBPIX = 256;
constexpr float GUARD = 16384.f;  // |screen coord| limit (pixels) of the fixed-point path
constexpr float Z_NEAR = 0.1f, Z_FAR = 10.0f;
constexpr int TILE = 8;           // raster tile = 8 x 8 pixels = one wave

// sample positions inside a pixel, 1/256 px: index 0 = single-sample pattern, 1..4 = the standard 4x pattern
MP_HD int sample_off_x(int ns, int s) { return ns == 1 ? 128 : (s == 0 ? 96 : s == 1 ? 224 : s == 2 ? 32 : 160); }
MP_HD int sample_off_y(int ns, int s) { return ns == 1 ? 128 : (s == 0 ? 32 : s == 1 ? 96 : s == 2 ? 160 : 224); }
MP_HD int sample_off_min(int ns) { return ns == 1 ? 128 : 32; }
MP_HD int sample_off_max(int ns) { return ns == 1 ? 128 : 224; }

struct MeshRef {
  const float* verts;
  const float* normals;
  const float* colors;
  const int32_t* faces;
  int n_verts, n_faces;
  float radius;
  const float* uvs;  // per-corner uv [n_faces][3][2] or NULL
};

struct TexRef {
  const uint32_t* texels;
  int tex_w, tex_h, tex_levels;
  int tex_off[MP_TEX_MAX_LEVELS];
};

struct Lights {
  float ambient[3];
  int n_point;
  float dir[8][3];
  float color[8][3];
  float offset[8][3];
};

// One piece of a triangle as seen by one view: snapped screen coordinates (positively oriented), 1/z per vertex and the
// depth-tie id.  `bary[k][j]` = weight of the ORIGINAL corner j in piece vertex k (a permutation matrix when unclipped).
struct Piece {
  int X[3], Y[3];
  float iz[3];
  int id;     // < 0: no piece
  int tri;
  int flags;  // bit 0: vertices 1 and 2 were exchanged (negative screen orientation); bit 1: the piece was clipped
  float bary[3][3];
};

// bary rows of an UNCLIPPED piece: the permutation its orientation swap applied to the corners
MP_HD void piece_bary_from_flags(Piece& p) {
  const bool swap = (p.flags & 1) != 0;
  p.bary[0][0] = 1.f; p.bary[0][1] = 0.f; p.bary[0][2] = 0.f;
  p.bary[1][0] = 0.f; p.bary[1][1] = swap ? 0.f : 1.f; p.bary[1][2] = swap ? 1.f : 0.f;
  p.bary[2][0] = 0.f; p.bary[2][1] = swap ? 1.f : 0.f; p.bary[2][2] = swap ? 0.f : 1.f;
}

MP_HD int imin(int a, int b) { return a < b ? a : b; }
MP_HD int imax(int a, int b) { return a > b ? a : b; }

MP_HD float dot3p(float a0, float a1, float a2, float x, float y, float z, float t) { return fmaf(a2, z, fmaf(a1, y, fmaf(a0, x, t))); }

struct CVert {
  float x, y, z;
  float bary[3];
};

// intersection of the edge in -> out with the near plane; in.bary / out.bary are the rows of the original corners (unit
// vectors), so fmaf(t, out.bary, (1 - t) * in.bary) is exactly {1 - t at in's corner, t at out's corner, 0 elsewhere}
MP_HD CVert clip_edge(const CVert& in, const CVert& out) {
  CVert r;
  const float t = (Z_NEAR - in.z) / (out.z - in.z);
  r.x = fmaf(t, out.x - in.x, in.x);
  r.y = fmaf(t, out.y - in.y, in.y);
  r.z = Z_NEAR;
  const float s = 1.0f - t;
#pragma unroll
  for (int j = 0; j < 3; ++j) r.bary[j] = fmaf(t, out.bary[j], s * in.bary[j]);
  return r;
}

MP_HD CVert sel(bool c, const CVert& a, const CVert& b) {  // c ? a : b, field by field (no dynamically indexed arrays: they
  CVert r;                                                  // would live in scratch memory on the GPU)
  r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z;
  r.bary[0] = c ? a.bary[0] : b.bary[0]; r.bary[1] = c ? a.bary[1] : b.bary[1]; r.bary[2] = c ? a.bary[2] : b.bary[2];
  return r;
}

// project + snap + orient; id < 0 on rejection
template <bool WITH_BARY>
MP_HD void finish_piece(const CVert& v0, const CVert& v1, const CVert& v2, const float* Kv, int tri, int id, bool clipped, Piece& p) {
  p.id = -1;
  p.tri = tri;
  p.flags = 0;
  const float iz0 = 1.0f / v0.z, iz1 = 1.0f / v1.z, iz2 = 1.0f / v2.z;
  const float sx0 = fmaf(Kv[0], v0.x * iz0, Kv[2]), sy0 = fmaf(Kv[4], v0.y * iz0, Kv[5]);
  const float sx1 = fmaf(Kv[0], v1.x * iz1, Kv[2]), sy1 = fmaf(Kv[4], v1.y * iz1, Kv[5]);
  const float sx2 = fmaf(Kv[0], v2.x * iz2, Kv[2]), sy2 = fmaf(Kv[4], v2.y * iz2, Kv[5]);
  const bool ok = (fabsf(sx0) < GUARD) && (fabsf(sy0) < GUARD) && (fabsf(sx1) < GUARD) && (fabsf(sy1) < GUARD) && (fabsf(sx2) < GUARD) &&
                  (fabsf(sy2) < GUARD);
  if (!ok) return;
  const int X0 = (int)rintf(sx0 * (float)SUBPIX), Y0 = (int)rintf(sy0 * (float)SUBPIX);
  const int X1 = (int)rintf(sx1 * (float)SUBPIX), Y1 = (int)rintf(sy1 * (float)SUBPIX);
  const int X2 = (int)rintf(sx2 * (float)SUBPIX), Y2 = (int)rintf(sy2 * (float)SUBPIX);
  const long long area = (long long)(X1 - X0) * (long long)(Y2 - Y0) - (long long)(Y1 - Y0) * (long long)(X2 - X0);
  if (area == 0) return;
  const bool swap = area < 0;  // two-sided: a negatively oriented piece is drawn with its vertices 1 and 2 exchanged
  p.X[0] = X0; p.Y[0] = Y0; p.iz[0] = iz0;
  p.X[1] = swap ? X2 : X1; p.Y[1] = swap ? Y2 : Y1; p.iz[1] = swap ? iz2 : iz1;
  p.X[2] = swap ? X1 : X2; p.Y[2] = swap ? Y1 : Y2; p.iz[2] = swap ? iz1 : iz2;
  if (WITH_BARY) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      p.bary[0][j] = v0.bary[j];
      p.bary[1][j] = swap ? v2.bary[j] : v1.bary[j];
      p.bary[2][j] = swap ? v1.bary[j] : v2.bary[j];
    }
  }
  p.id = id;
  p.flags = (swap ? 1 : 0) | (clipped ? 2 : 0);
}

// Piece `which` (0 = first, 1 = second) of triangle `tri` under pose T / intrinsics Kv.  Stateless: the binning pass, the
// coverage pass and the shading pass each recompute the piece they need from the mesh (which stays L2-resident) instead of
// round-tripping per-view set-up records through HBM.
// Returns how many pieces the triangle has at most under this view (0, 1 or 2; a piece can still be rejected: p.id < 0).
template <bool WITH_BARY>
MP_HD int make_piece(const MeshRef& m, const float* T, const float* Kv, int tri, int which, Piece& p) {
  p.id = -1;
  p.tri = tri;
  p.flags = 0;
  CVert c0, c1, c2;
  {
    const int v0 = m.faces[3 * tri], v1 = m.faces[3 * tri + 1], v2 = m.faces[3 * tri + 2];
    const float* q0 = m.verts + 3 * (size_t)v0;
    const float* q1 = m.verts + 3 * (size_t)v1;
    const float* q2 = m.verts + 3 * (size_t)v2;
    c0.x = dot3p(T[0], T[1], T[2], q0[0], q0[1], q0[2], T[3]); c0.y = dot3p(T[4], T[5], T[6], q0[0], q0[1], q0[2], T[7]);
    c0.z = dot3p(T[8], T[9], T[10], q0[0], q0[1], q0[2], T[11]);
    c1.x = dot3p(T[0], T[1], T[2], q1[0], q1[1], q1[2], T[3]); c1.y = dot3p(T[4], T[5], T[6], q1[0], q1[1], q1[2], T[7]);
    c1.z = dot3p(T[8], T[9], T[10], q1[0], q1[1], q1[2], T[11]);
    c2.x = dot3p(T[0], T[1], T[2], q2[0], q2[1], q2[2], T[3]); c2.y = dot3p(T[4], T[5], T[6], q2[0], q2[1], q2[2], T[7]);
    c2.z = dot3p(T[8], T[9], T[10], q2[0], q2[1], q2[2], T[11]);
    c0.bary[0] = 1.f; c0.bary[1] = 0.f; c0.bary[2] = 0.f;
    c1.bary[0] = 0.f; c1.bary[1] = 1.f; c1.bary[2] = 0.f;
    c2.bary[0] = 0.f; c2.bary[1] = 0.f; c2.bary[2] = 1.f;
  }
  const bool in0 = c0.z >= Z_NEAR, in1 = c1.z >= Z_NEAR, in2 = c2.z >= Z_NEAR;
  const int n_in = (int)in0 + (int)in1 + (int)in2;
  if (n_in == 0) return 0;
  // the piece's three vertices (a, b, c) and its id; ONE finish_piece call at the end (code size: the kernels inline all of this)
  CVert a = c0, b = c1, c = c2;
  int id = tri, n_pieces = 1;
  bool exists = which == 0;
  if (n_in != 3) {
    // rotate the corners (cyclic order kept) so that r0 is the single inside vertex (n_in == 1) or r2 the single outside one
    const int k = n_in == 1 ? (in0 ? 0 : (in1 ? 1 : 2)) : (!in0 ? 1 : (!in1 ? 2 : 0));
    const CVert r0 = sel(k == 0, c0, sel(k == 1, c1, c2));
    const CVert r1 = sel(k == 0, c1, sel(k == 1, c2, c0));
    const CVert r2 = sel(k == 0, c2, sel(k == 1, c0, c1));
    a = r0;
    if (n_in == 1) {
      b = clip_edge(r0, r1);
      c = clip_edge(r0, r2);
    } else {
      n_pieces = 2;
      exists = true;
      const CVert P = clip_edge(r1, r2);
      if (which == 0) {
        b = r1;
        c = P;
      } else {
        b = P;
        c = clip_edge(r0, r2);
        id = m.n_faces + tri;
      }
    }
  }
  if (exists) finish_piece<WITH_BARY>(a, b, c, Kv, tri, id, n_in != 3, p);
  return n_pieces;
}

// piece index space of a view: [0, F) first pieces, [F, 2F) second pieces
template <bool WITH_BARY>
MP_HD void piece_from_index(const MeshRef& m, const float* T, const float* Kv, int idx, Piece& p) {
  const int tri = idx < m.n_faces ? idx : idx - m.n_faces;
  (void)make_piece<WITH_BARY>(m, T, Kv, tri, idx < m.n_faces ? 0 : 1, p);
}

// pixel bbox (inclusive) of the pixels that own a sample inside the piece's snapped bbox
MP_HD void piece_pixel_bbox(const Piece& p, int ns, int w, int h, int& x0, int& y0, int& x1, int& y1) {
  const int Xmin = imin(p.X[0], imin(p.X[1], p.X[2])), Xmax = imax(p.X[0], imax(p.X[1], p.X[2]));
  const int Ymin = imin(p.Y[0], imin(p.Y[1], p.Y[2])), Ymax = imax(p.Y[0], imax(p.Y[1], p.Y[2]));
  const int omin = sample_off_min(ns), omax = sample_off_max(ns);
  x0 = imax(0, (Xmin - omax + 255) >> 8);   // arithmetic shift = floor division (coordinates may be negative)
  x1 = imin(w - 1, (Xmax - omin) >> 8);
  y0 = imax(0, (Ymin - omax + 255) >> 8);
  y1 = imin(h - 1, (Ymax - omin) >> 8);
}

// Edge a->b: E(p) = (bx-ax)*(py-ay) - (by-ay)*(px-ax) = A*px + B*py + C.  With y pointing down and a positively oriented piece
// the interior is E >= 0; an edge is "left" if it goes up (dy < 0) and "top" if dy == 0 && dx > 0.  Top-left edges own their
// boundary samples (threshold 0); the others exclude E == 0 (threshold 1).
struct Edges {
  long long A[3], B[3], C[3];
  int thr[3];
  float inv_area;
};

MP_HD void edge_setup(int ax, int ay, int bx, int by, long long& A, long long& B, long long& C, int& thr) {
  const long long dx = (long long)bx - ax, dy = (long long)by - ay;
  A = -dy;
  B = dx;
  C = dy * ax - dx * ay;
  thr = ((dy < 0) || (dy == 0 && dx > 0)) ? 0 : 1;
}

MP_HD void piece_edges(const Piece& p, Edges& e) {
  edge_setup(p.X[1], p.Y[1], p.X[2], p.Y[2], e.A[0], e.B[0], e.C[0], e.thr[0]);  // edge opposite vertex 0
  edge_setup(p.X[2], p.Y[2], p.X[0], p.Y[0], e.A[1], e.B[1], e.C[1], e.thr[1]);
  edge_setup(p.X[0], p.Y[0], p.X[1], p.Y[1], e.A[2], e.B[2], e.C[2], e.thr[2]);
  const long long area = (long long)(p.X[1] - p.X[0]) * (long long)(p.Y[2] - p.Y[0]) - (long long)(p.Y[1] - p.Y[0]) * (long long)(p.X[2] - p.X[0]);
  e.inv_area = 1.0f / (float)area;
}

// barycentrics and wsum at fixed-point position (sx, sy); returns coverage
MP_HD bool eval_at(const Piece& p, const Edges& e, long long sx, long long sy, float b[3], float& wsum) {
  bool inside = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const long long v = e.A[i] * sx + e.B[i] * sy + e.C[i];
    inside = inside && (v >= e.thr[i]);
    b[i] = (float)v * e.inv_area;
  }
  wsum = fmaf(b[2], p.iz[2], fmaf(b[1], p.iz[1], b[0] * p.iz[0]));
  return inside;
}

// ---- tile-list records -------------------------------------------------------------------------------------------------------
// The binning pass stores, for every (small piece, touched tile) pair, the piece as the coverage pass needs it: screen coordinates
// RELATIVE to the tile's first sample column / row (they fit 16 bits for a small piece), 1/z per vertex and the depth-tie id --
// 32 bytes, read back with two coalesced 16-byte loads (no index -> faces -> vertices pointer chase, no re-transformation).
struct __attribute__((aligned(16))) TileRec {
  short rx0, ry0, rx1, ry1, rx2, ry2;
  short pad0, pad1;
  float iz0, iz1, iz2;
  int id;
};
static_assert(sizeof(TileRec) == 32, "TileRec is two 16-byte words");

// a piece whose extent is <= SMALL_EXTENT sub-pixel units is "small" (piece_is_small) for EVERY tile its sample bbox touches
constexpr int SMALL_EXTENT = 23170 - 2048 - 256;

MP_HD int piece_extent(const Piece& p) {
  const int Xmin = imin(p.X[0], imin(p.X[1], p.X[2])), Xmax = imax(p.X[0], imax(p.X[1], p.X[2]));
  const int Ymin = imin(p.Y[0], imin(p.Y[1], p.Y[2])), Ymax = imax(p.Y[0], imax(p.Y[1], p.Y[2]));
  return imax(Xmax - Xmin, Ymax - Ymin);
}

MP_HD TileRec pack_tile_rec(const Piece& p, int tile_x0, int tile_y0) {
  TileRec r;
  const int ox = tile_x0 * SUBPIX, oy = tile_y0 * SUBPIX;
  r.rx0 = (short)(p.X[0] - ox); r.ry0 = (short)(p.Y[0] - oy);
  r.rx1 = (short)(p.X[1] - ox); r.ry1 = (short)(p.Y[1] - oy);
  r.rx2 = (short)(p.X[2] - ox); r.ry2 = (short)(p.Y[2] - oy);
  r.pad0 = (short)p.flags;   // orientation swap / clipped: what the shading pass needs to map piece vertices to corners
  r.pad1 = 0;
  r.iz0 = p.iz[0]; r.iz1 = p.iz[1]; r.iz2 = p.iz[2];
  r.id = p.id;
  return r;
}

MP_HD void unpack_tile_rec(const TileRec& r, int tile_x0, int tile_y0, Piece& p) {
  const int ox = tile_x0 * SUBPIX, oy = tile_y0 * SUBPIX;
  p.X[0] = ox + r.rx0; p.Y[0] = oy + r.ry0;
  p.X[1] = ox + r.rx1; p.Y[1] = oy + r.ry1;
  p.X[2] = ox + r.rx2; p.Y[2] = oy + r.ry2;
  p.iz[0] = r.iz0; p.iz[1] = r.iz1; p.iz[2] = r.iz2;
  p.id = r.id;
  p.tri = -1;   // the caller derives it from the id (first piece: id, second: id - n_faces)
  p.flags = r.pad0;
}

struct Sample {
  float wsum;
  int id;  // < 0: empty
};

MP_HD bool depth_in_range(float wsum) { return wsum >= 1.0f / Z_FAR && wsum <= 1.0f / Z_NEAR; }

MP_HD void sample_update(Sample& s, float wsum, int id) {
  if (s.id < 0 || wsum > s.wsum || (wsum == s.wsum && id < s.id)) {
    s.wsum = wsum;
    s.id = id;
  }
}

// ---- 32-bit fast path of the edge functions (the GPU's inner loop) -----------------------------------------------------------
// For a piece whose extent D = max(Xmax - Xmin, Ymax - Ymin) and whose distance R to the farthest sample of the 8x8 tile are both
// <= 23170 sub-pixel units (90 px), every |E| = |dx * ry - dy * rx| <= 2 * 23170^2 < 2^31 and every factor fits 24 bits, so the
// edge functions can be evaluated in int32 relative to the edge's first vertex -- the same integers as the 64-bit form.
struct Edges32 {
  int dx[3], dy[3];   // edge vector b - a
  int ax[3], ay[3];   // edge origin
  int thr[3];
  float inv_area;
};

MP_HD bool piece_is_small(const Piece& p, int tile_x0, int tile_y0) {
  const int Xmin = imin(p.X[0], imin(p.X[1], p.X[2])), Xmax = imax(p.X[0], imax(p.X[1], p.X[2]));
  const int Ymin = imin(p.Y[0], imin(p.Y[1], p.Y[2])), Ymax = imax(p.Y[0], imax(p.Y[1], p.Y[2]));
  const int sx0 = tile_x0 * SUBPIX, sx1 = sx0 + TILE * SUBPIX, sy0 = tile_y0 * SUBPIX, sy1 = sy0 + TILE * SUBPIX;
  const long long D = imax(Xmax - Xmin, Ymax - Ymin);
  const long long R = (long long)imax(imax(sx1 - Xmin, Xmax - sx0), imax(sy1 - Ymin, Ymax - sy0));
  // (differences of ints below 2^23 in magnitude: no overflow)
  return D <= 23170 && R <= 23170 && sx1 - Xmin >= -23170 && Xmax - sx0 >= -23170 && sy1 - Ymin >= -23170 && Ymax - sy0 >= -23170;
}

MP_HD void piece_edges32(const Piece& p, Edges32& e) {
  const int a[3] = {1, 2, 0}, b[3] = {2, 0, 1};  // edge i runs a[i] -> b[i] (edge opposite vertex i)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    e.dx[i] = p.X[b[i]] - p.X[a[i]];
    e.dy[i] = p.Y[b[i]] - p.Y[a[i]];
    e.ax[i] = p.X[a[i]];
    e.ay[i] = p.Y[a[i]];
    e.thr[i] = ((e.dy[i] < 0) || (e.dy[i] == 0 && e.dx[i] > 0)) ? 0 : 1;
  }
  const long long area = (long long)(p.X[1] - p.X[0]) * (long long)(p.Y[2] - p.Y[0]) - (long long)(p.Y[1] - p.Y[0]) * (long long)(p.X[2] - p.X[0]);
  e.inv_area = 1.0f / (float)area;
}

// E_i at sample (sx, sy), int32 (valid under piece_is_small)
MP_HD int edge32(const Edges32& e, int i, int sx, int sy) { return MP_MUL24(e.dx[i], sy - e.ay[i]) - MP_MUL24(e.dy[i], sx - e.ax[i]); }

// conservative per-lane test of the 32-bit path: can ANY sample of pixel (px, py) be inside the (small) piece?
// (no sample is farther than 96/256 px from the pixel centre in x or y)
MP_HD bool maybe_covered32(const Edges32& e, int px, int py) {
  const int cx = px * SUBPIX + 128, cy = py * SUBPIX + 128;
  bool maybe = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) maybe = maybe && (edge32(e, i, cx, cy) + (abs(e.dx[i]) + abs(e.dy[i])) * 96 >= e.thr[i]);
  return maybe;
}

// One SMALL piece (piece_is_small for the pixel's tile) against the NS samples of pixel (px, py): emit(s, wsum) is called for every
// sample that is covered (top-left rule) and inside the depth range.  This is the arithmetic of both coverage forms of the kernel:
// the wave-per-piece sweep (64 lanes = the tile's pixels) and the lane-per-piece scatter (a lane walks its piece's bbox).
template <int NS, class Emit>
MP_HD void cover_pixel32(const Piece& p, const Edges32& e, int px, int py, Emit&& emit) {
  const int cx = px * SUBPIX + 128, cy = py * SUBPIX + 128;
  int ec[3];
  bool maybe = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ec[i] = edge32(e, i, cx, cy);
    if (NS > 1) maybe = maybe && (ec[i] + (abs(e.dx[i]) + abs(e.dy[i])) * 96 >= e.thr[i]);
  }
  if (!maybe) return;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int ox = sample_off_x(NS, s) - 128, oy = sample_off_y(NS, s) - 128;
    int es[3];
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      es[i] = ec[i] + (e.dx[i] * oy - e.dy[i] * ox);
      inside = inside && (es[i] >= e.thr[i]);
    }
    if (inside) {
      const float b0 = (float)es[0] * e.inv_area, b1 = (float)es[1] * e.inv_area, b2 = (float)es[2] * e.inv_area;
      const float wsum = fmaf(b2, p.iz[2], fmaf(b1, p.iz[1], b0 * p.iz[0]));
      if (depth_in_range(wsum)) emit(s, wsum);
    }
  }
}

// the general (64-bit) form for pieces that are not small
template <int NS, class Emit>
MP_HD void cover_pixel64(const Piece& p, const Edges& e, int px, int py, Emit&& emit) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float b[3], wsum;
    const bool inside = eval_at(p, e, (long long)(px * SUBPIX + sample_off_x(NS, s)), (long long)(py * SUBPIX + sample_off_y(NS, s)), b, wsum);
    if (inside && depth_in_range(wsum)) emit(s, wsum);
  }
}

// 64-bit z-buffer key of a covered sample: larger wsum (nearer) wins, equal depth -> lower piece id wins; 0 = empty.
// (wsum > 0 inside the depth range, so its bit pattern orders like the value.)  Layout: [63:32] wsum bits, [31:9] 0x7FFFFF - id
// (ids < 2^23: meshes have < 2^22 faces), [8:0] the piece's position in the tile's record list (SLOT_NONE = not a binned record):
// the shading pass re-reads that record instead of re-deriving the piece from the mesh.  The slot never decides a comparison
// (equal depth and id = same piece = same slot).
constexpr int SLOT_NONE = 511;
MP_HD unsigned long long depth_key(float wsum, int id, int slot) {
  uint32_t wb;
  memcpy(&wb, &wsum, 4);
  return ((unsigned long long)wb << 32) | ((unsigned long long)(0x7FFFFFu - (uint32_t)id) << 9) | (unsigned long long)(uint32_t)slot;
}
MP_HD float key_wsum(unsigned long long key) {
  const uint32_t wb = (uint32_t)(key >> 32);
  float f;
  memcpy(&f, &wb, 4);
  return f;
}
MP_HD int key_id(unsigned long long key) { return key ? (int)(0x7FFFFFu - (uint32_t)((key >> 9) & 0x7FFFFFu)) : -1; }
MP_HD int key_slot(unsigned long long key) { return (int)(key & 511u); }

template <int NS>
MP_HD void cover_lane(const Piece& p, int tile_x0, int tile_y0, int px, int py, Sample (&st)[NS]) {
  if (piece_is_small(p, tile_x0, tile_y0)) {
    Edges32 e;
    piece_edges32(p, e);
    cover_pixel32<NS>(p, e, px, py, [&](int s, float wsum) { sample_update(st[s], wsum, p.id); });
  } else {
    Edges e;
    piece_edges(p, e);
    cover_pixel64<NS>(p, e, px, py, [&](int s, float wsum) { sample_update(st[s], wsum, p.id); });
  }
}

// ---- block-visit coverage form (every binned record) ---------------------------------------------------------------------------
// The wave owns one 8x8 tile; its lanes are the SAMPLES of one block of the tile at a time (NS = 4: four 4x4-pixel blocks, lane =
// pixel * 4 + sample; NS = 1: one 8x8 block, lane = pixel), the depth state of a lane's sample lives in registers, and the pieces
// that can touch a block are visited one after the other (wave-uniform).  A visit evaluates each edge function with ONE
// instruction: E_i = E0_i + dx_i * ry - dy_i * rx, where (rx, ry) is the lane's sample position relative to the tile's first
// sample column / row (a per-lane constant) and E0_i the edge function there -- v_dot2_i32_i16 on (dx | -dy << 16) . (ry | rx << 16).
// The integers are those of edge32() (|dx|, |dy| <= SMALL_EXTENT < 2^15 for a binned piece; sums wrap mod 2^32 and the true
// value fits), so coverage, barycentrics and depth are bit-identical to the other coverage forms.
MP_HD int dot2_i16(uint32_t a, uint32_t b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef short mp_s2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(mp_s2, a), __builtin_bit_cast(mp_s2, b), c, false);
#else
  const uint32_t p0 = (uint32_t)((int)(int16_t)(a & 0xFFFFu) * (int)(int16_t)(b & 0xFFFFu));
  const uint32_t p1 = (uint32_t)((int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16));
  return (int)((uint32_t)c + p0 + p1);
#endif
}

struct __attribute__((aligned(16))) BlkRec {   // a binned piece as the visits need it: three 16-byte words
  uint32_t dxy[3];    // edge i: low half dx_i, high half -dy_i
  uint32_t thr_bits;  // bit i: threshold of edge i (top-left rule)
  int e0[3];          // E_i at the tile's first sample position
  float inv_area;
  float iz[3];
  uint32_t key_lo;    // low word of the depth key: (0x7FFFFF - id) << 9 | record slot
};
static_assert(sizeof(BlkRec) == 48, "BlkRec is three 16-byte words");

MP_HD uint32_t pack_i16x2(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

MP_HD BlkRec make_blk_rec(const Piece& p, const Edges32& e, int tile_x0, int tile_y0, int slot) {
  BlkRec r;
  r.thr_bits = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r.dxy[i] = pack_i16x2(e.dx[i], -e.dy[i]);
    r.e0[i] = edge32(e, i, tile_x0 * SUBPIX, tile_y0 * SUBPIX);
    r.thr_bits |= (uint32_t)e.thr[i] << i;
    r.iz[i] = p.iz[i];
  }
  r.inv_area = e.inv_area;
  r.key_lo = ((0x7FFFFFu - (uint32_t)p.id) << 9) | (uint32_t)slot;
  return r;
}

// block geometry: NS = 4 -> four 4x4-pixel blocks (k & 1 = column, k >> 1 = row), NS = 1 -> the whole tile
MP_HD int blk_count(int ns) { return ns == 1 ? 1 : 4; }
MP_HD int blk_size(int ns) { return ns == 1 ? 8 : 4; }
// pixel (inside the tile, 0..63) and sample of lane `lane` in block k
MP_HD int blk_lane_pixel(int ns, int k, int lane) {
  if (ns == 1) return lane;
  const int p16 = lane >> 2;
  return ((((k >> 1) << 2) + (p16 >> 2)) << 3) | (((k & 1) << 2) + (p16 & 3));
}
MP_HD int blk_lane_sample(int ns, int lane) { return ns == 1 ? 0 : (lane & 3); }
// the lane's sample position relative to the tile's first sample column / row, packed (ry | rx << 16)
MP_HD uint32_t blk_lane_rel(int ns, int k, int lane) {
  const int pix = blk_lane_pixel(ns, k, lane), s = blk_lane_sample(ns, lane);
  return pack_i16x2((pix >> 3) * SUBPIX + sample_off_y(ns, s), (pix & 7) * SUBPIX + sample_off_x(ns, s));
}

// can the piece own a sample of block k?  (conservative: separating-edge test against the block's sample rectangle)
// (rxmin .. rymax: the piece's snapped bounding box relative to the tile's first sample column / row)
MP_HD bool blk_touched(const BlkRec& r, int rxmin, int rxmax, int rymin, int rymax, int ns, int k) {
  if (ns == 1) return true;
  const int bx = (k & 1) * 4, by = (k >> 1) * 4;
  const int x_lo = bx * SUBPIX + sample_off_min(ns), x_hi = (bx + 3) * SUBPIX + sample_off_max(ns);
  const int y_lo = by * SUBPIX + sample_off_min(ns), y_hi = (by + 3) * SUBPIX + sample_off_max(ns);
  bool ok = rxmin <= x_hi && rxmax >= x_lo && rymin <= y_hi && rymax >= y_lo;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int dx = (int)(int16_t)(r.dxy[i] & 0xFFFFu), ndy = (int)(int16_t)(r.dxy[i] >> 16);
    const int e = dot2_i16(r.dxy[i], pack_i16x2(dx >= 0 ? y_hi : y_lo, ndy >= 0 ? x_hi : x_lo), r.e0[i]);
    ok = ok && (e >= 0);
  }
  return ok;
}

// one visit: the lane's sample (position `rel`) against the piece; emit(wsum) if covered (top-left rule) and inside the depth range
template <class Emit>
MP_HD void cover_sample_rel(const BlkRec& r, uint32_t rel, Emit&& emit) {
  const int E0 = dot2_i16(r.dxy[0], rel, r.e0[0]), E1 = dot2_i16(r.dxy[1], rel, r.e0[1]), E2 = dot2_i16(r.dxy[2], rel, r.e0[2]);
  // (bitwise &: all three compares are issued back to back; && would make the GPU branch after every edge)
  const bool inside = (E0 >= (int)(r.thr_bits & 1u)) & (E1 >= (int)((r.thr_bits >> 1) & 1u)) & (E2 >= (int)((r.thr_bits >> 2) & 1u));
  if (inside) {
    const float b0 = (float)E0 * r.inv_area, b1 = (float)E1 * r.inv_area, b2 = (float)E2 * r.inv_area;
    const float wsum = fmaf(b2, r.iz[2], fmaf(b1, r.iz[1], b0 * r.iz[0]));
    if (depth_in_range(wsum)) emit(wsum);
  }
}

MP_HD unsigned long long depth_key_lo(float wsum, uint32_t key_lo) {
  uint32_t wb;
  memcpy(&wb, &wsum, 4);
  return ((unsigned long long)wb << 32) | (unsigned long long)key_lo;
}

// piece index (in the [0, 2F) space) from a depth-tie id: they coincide (first piece: tri, second: F + tri)
MP_HD int index_of_id(int id) { return id; }

MP_HD float lut_val(int i) { return (float)((i * 255) >> 5); }  // floor(i*255/32), utils.py:65

// Eye-normal LUT lookup: 32-texel separable ramp, GL_LINEAR filter, repeat wrap; value on the 0..255 scale
MP_HD float normal_lut(float n) {
  const float u = n - floorf(n);
  const float t = fmaf(u, 32.0f, -0.5f);
  const float fl = floorf(t);
  const float f = t - fl;
  const int i0 = ((int)fl + 32) & 31;
  const int i1 = (i0 + 1) & 31;
  const float a = lut_val(i0), b = lut_val(i1);
  return fmaf(b - a, f, a);
}

// clamp to [0, 255] and round half up (the 8-bit colour buffer); NaN -> 0
MP_HD float q255(float v255) { return floorf(fminf(fmaxf(v255, 0.f), 255.f) + 0.5f); }

MP_HD void tex_sample(const TexRef& m, int level, float u, float v, float out[3]) {
  const int tw = imax(1, m.tex_w >> level), th = imax(1, m.tex_h >> level);
  const uint32_t* tx = m.texels + m.tex_off[level];
  const float fu = fmaf(u - floorf(u), (float)tw, -0.5f), fv = fmaf(v - floorf(v), (float)th, -0.5f);
  const float flu = floorf(fu), flv = floorf(fv);
  const float au = fu - flu, av = fv - flv;
  int x0 = (int)flu, y0 = (int)flv;
  if (x0 < 0) x0 = tw - 1;
  if (y0 < 0) y0 = th - 1;
  if (x0 >= tw) x0 = tw - 1;  // u - floor(u) can round to 1.0f for tiny negative u
  if (y0 >= th) y0 = th - 1;
  const int x1 = (x0 + 1 == tw) ? 0 : x0 + 1, y1 = (y0 + 1 == th) ? 0 : y0 + 1;
  const uint32_t t00 = tx[y0 * tw + x0], t01 = tx[y0 * tw + x1], t10 = tx[y1 * tw + x0], t11 = tx[y1 * tw + x1];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a00 = (float)((t00 >> (8 * c)) & 255u), a01 = (float)((t01 >> (8 * c)) & 255u);
    const float a10 = (float)((t10 >> (8 * c)) & 255u), a11 = (float)((t11 >> (8 * c)) & 255u);
    const float top = fmaf(a01 - a00, au, a00), bot = fmaf(a11 - a10, au, a10);
    out[c] = fmaf(bot - top, av, top);
  }
}

constexpr int MAX_ANISO = 16;   // the reference's `texture-anisotropic-degree` (panda3d_scene_renderer.py:72)

// level of detail from rho^2 (squared texel footprint): lambda = log2(rho) ~ 0.5 * (exponent + mantissa fraction) of rho^2
MP_HD void tex_lod(int levels, float rho2, int& level, float& frac) {
  level = 0;
  frac = 0.f;
  if (!(rho2 > 1.0f)) return;
  if (!(rho2 < 1e30f)) { level = levels - 1; return; }
  uint32_t bits;
  memcpy(&bits, &rho2, 4);
  const int e = (int)(bits >> 23) - 127;
  const float m = (float)(bits & 0x7FFFFFu) * (1.0f / 8388608.0f);
  const float lambda = 0.5f * ((float)e + m);
  const float fl = floorf(lambda);
  int l0 = (int)fl;
  float f = lambda - fl;
  if (l0 >= levels - 1) { l0 = levels - 1; f = 0.f; }
  level = l0;
  frac = f;
}

// attribute `comp` (stride 3) of piece vertex k = bary-weighted combination of the original corners' attributes
MP_HD float pv_attr(const Piece& p, int k, const float* attr, int i0, int i1, int i2, int comp) {
  return fmaf(p.bary[k][2], attr[3 * i2 + comp], fmaf(p.bary[k][1], attr[3 * i1 + comp], p.bary[k][0] * attr[3 * i0 + comp]));
}

// Shade piece p at the centre of pixel (px, py): col255 = RGB on the 0..255 scale before clamping/rounding, nrm255 = eye-normal
// LUT values on the 0..255 scale.  (Barycentrics are extrapolated when the centre lies outside the piece.)
// FULL = false: the instance for vertex-coloured meshes under ambient light only (what the pose networks render): the texture and
// point-light code is not compiled in (registers, code size); the caller guarantees tex == NULL / no uvs and L.n_point == 0.
template <bool FULL = true>
MP_HD void shade(const MeshRef& m, const TexRef* tex, const Lights& L, const float* T, bool gl_eye, bool want_normals, const Piece& p,
                 int px, int py, float col255[3], float nrm255[3]) {
  // barycentrics at the pixel centre and the edge slopes (for the texture derivatives): the 32-bit form for a piece that is small
  // for the pixel's tile (the same integers as the 64-bit form, at a fifth of the instructions)
  float b[3], wsum, inv_area;
  long long eA[3], eB[3];
  if (piece_is_small(p, px & ~(TILE - 1), py & ~(TILE - 1))) {
    Edges32 e;
    piece_edges32(p, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      b[i] = (float)edge32(e, i, px * SUBPIX + 128, py * SUBPIX + 128) * e.inv_area;
      eA[i] = -(long long)e.dy[i];
      eB[i] = (long long)e.dx[i];
    }
    inv_area = e.inv_area;
    wsum = fmaf(b[2], p.iz[2], fmaf(b[1], p.iz[1], b[0] * p.iz[0]));
  } else {
    Edges e;
    piece_edges(p, e);
    (void)eval_at(p, e, (long long)px * SUBPIX + 128, (long long)py * SUBPIX + 128, b, wsum);
#pragma unroll
    for (int i = 0; i < 3; ++i) { eA[i] = e.A[i]; eB[i] = e.B[i]; }
    inv_area = e.inv_area;
  }
  const float w0 = b[0] * p.iz[0], w1 = b[1] * p.iz[1], w2 = b[2] * p.iz[2];
  const float z = 1.0f / wsum;
  const int i0 = m.faces[3 * p.tri], i1 = m.faces[3 * p.tri + 1], i2 = m.faces[3 * p.tri + 2];
  float col[3], on[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    col[k] = fmaf(w2, pv_attr(p, 2, m.colors, i0, i1, i2, k), fmaf(w1, pv_attr(p, 1, m.colors, i0, i1, i2, k), w0 * pv_attr(p, 0, m.colors, i0, i1, i2, k))) * z;
    on[k] = fmaf(w2, pv_attr(p, 2, m.normals, i0, i1, i2, k), fmaf(w1, pv_attr(p, 1, m.normals, i0, i1, i2, k), w0 * pv_attr(p, 0, m.normals, i0, i1, i2, k))) * z;
  }
  if (FULL && m.uvs && tex && tex->texels) {
    const float* uv = m.uvs + 6 * (size_t)p.tri;
    float pu[3], pv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pu[k] = fmaf(p.bary[k][2], uv[4], fmaf(p.bary[k][1], uv[2], p.bary[k][0] * uv[0]));
      pv[k] = fmaf(p.bary[k][2], uv[5], fmaf(p.bary[k][1], uv[3], p.bary[k][0] * uv[1]));
    }
    const float u = fmaf(w2, pu[2], fmaf(w1, pu[1], w0 * pu[0])) * z;
    const float v = fmaf(w2, pv[2], fmaf(w1, pv[1], w0 * pv[0])) * z;
    float dbx[3], dby[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dbx[i] = (float)(eA[i] * 256) * inv_area * p.iz[i];
      dby[i] = (float)(eB[i] * 256) * inv_area * p.iz[i];
    }
    const float dDx = dbx[2] + (dbx[1] + dbx[0]), dDy = dby[2] + (dby[1] + dby[0]);
    const float dNux = fmaf(dbx[2], pu[2], fmaf(dbx[1], pu[1], dbx[0] * pu[0])), dNuy = fmaf(dby[2], pu[2], fmaf(dby[1], pu[1], dby[0] * pu[0]));
    const float dNvx = fmaf(dbx[2], pv[2], fmaf(dbx[1], pv[1], dbx[0] * pv[0])), dNvy = fmaf(dby[2], pv[2], fmaf(dby[1], pv[1], dby[0] * pv[0]));
    const float tw = (float)tex->tex_w, th = (float)tex->tex_h;
    const float dudx = fmaf(-u, dDx, dNux) * z * tw, dvdx = fmaf(-v, dDx, dNvx) * z * th;
    const float dudy = fmaf(-u, dDy, dNuy) * z * tw, dvdy = fmaf(-v, dDy, dNvy) * z * th;
    const float rx2 = fmaf(dvdx, dvdx, dudx * dudx), ry2 = fmaf(dvdy, dvdy, dudy * dudy);
    // anisotropic filtering, degree 16 (panda3d_scene_renderer.py:72 `texture-anisotropic-degree 16`), by the formula of the OpenGL
    // extension's specification (EXT_texture_filter_anisotropic): N = min(ceil(Pmax / Pmin), 16) trilinear probes spread along the major
    // axis of the pixel's footprint at x - 1/2 + i / (N + 1), their level of detail from Pmax / N.  N = 1 is the isotropic sample.
    const bool x_major = rx2 >= ry2;
    const float r2max = x_major ? rx2 : ry2, r2min = x_major ? ry2 : rx2;
    int n_probe = 1;
    while (n_probe < MAX_ANISO && (float)(n_probe * n_probe) * r2min < r2max) ++n_probe;
    int level;
    float frac;
    tex_lod(tex->tex_levels, r2max / (float)(n_probe * n_probe), level, frac);
    const float du = (x_major ? dudx : dudy) / tw, dv = (x_major ? dvdx : dvdy) / th;   // the major axis in (u, v)
    float acc3[3] = {0.f, 0.f, 0.f};
    for (int i = 1; i <= n_probe; ++i) {
      const float t = (float)i / (float)(n_probe + 1) - 0.5f;
      const float ui = fmaf(t, du, u), vi = fmaf(t, dv, v);
      float tc[3];
      tex_sample(*tex, level, ui, vi, tc);
      if (frac > 0.f) {
        float tc1[3];
        tex_sample(*tex, level + 1, ui, vi, tc1);
#pragma unroll
        for (int k = 0; k < 3; ++k) tc[k] = fmaf(tc1[k] - tc[k], frac, tc[k]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) acc3[k] += tc[k];
    }
    const float inv_n = 1.0f / (float)n_probe;
#pragma unroll
    for (int k = 0; k < 3; ++k) col[k] *= (acc3[k] * inv_n) / 255.0f;
  }
  float lr = L.ambient[0], lg = L.ambient[1], lb = L.ambient[2];
  if (FULL && L.n_point > 0) {
    float op[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      op[k] = fmaf(w2, pv_attr(p, 2, m.verts, i0, i1, i2, k), fmaf(w1, pv_attr(p, 1, m.verts, i0, i1, i2, k), w0 * pv_attr(p, 0, m.verts, i0, i1, i2, k))) * z;
    const float nn = sqrtf(fmaf(on[2], on[2], fmaf(on[1], on[1], on[0] * on[0])));
    const float inn = nn > 0.f ? 1.0f / nn : 0.f;
    const float R10 = 10.0f * m.radius;
    for (int l = 0; l < L.n_point; ++l) {
      const float lx = fmaf(L.dir[l][0], R10, L.offset[l][0]) - op[0];
      const float ly = fmaf(L.dir[l][1], R10, L.offset[l][1]) - op[1];
      const float lz = fmaf(L.dir[l][2], R10, L.offset[l][2]) - op[2];
      const float ln = sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
      const float d = fmaf(lz, on[2], fmaf(ly, on[1], lx * on[0])) * inn / ln;
      const float dd = fmaxf(d, 0.f);
      lr = fmaf(L.color[l][0], dd, lr);
      lg = fmaf(L.color[l][1], dd, lg);
      lb = fmaf(L.color[l][2], dd, lb);
    }
  }
  col255[0] = col[0] * lr * 255.0f;
  col255[1] = col[1] * lg * 255.0f;
  col255[2] = col[2] * lb * 255.0f;
  nrm255[0] = nrm255[1] = nrm255[2] = 0.f;
  if (want_normals) {
    // eye-space normal: camera (OpenCV) frame first, then the eye-axis convention
    const float cx = fmaf(T[2], on[2], fmaf(T[1], on[1], T[0] * on[0]));
    const float cy = fmaf(T[6], on[2], fmaf(T[5], on[1], T[4] * on[0]));
    const float cz = fmaf(T[10], on[2], fmaf(T[9], on[1], T[8] * on[0]));
    float ex, ey, ez;
    if (gl_eye) { ex = cx; ey = -cy; ez = -cz; }   // GL eye: x right, y up, z back
    else { ex = cx; ey = cz; ez = -cy; }           // Panda view: x right, y forward, z up (TCCGL, types.py:40)
    nrm255[0] = normal_lut(ex);
    nrm255[1] = normal_lut(ey);
    nrm255[2] = normal_lut(ez);
  }
}

// final value of a channel from the per-sample sum (already q255'ed unless no_quant)
MP_HD float resolve_channel(float acc, int ns, bool no_quant) {
  const float inv_ns = 1.0f / (float)ns;  // 1 or 0.25: exact
  return no_quant ? (acc * inv_ns) / 255.0f : floorf(fmaf(acc, inv_ns, 0.5f)) / 255.0f;
}

// the resolved 8-bit value itself (what resolve_channel divides by 255): the integer the stem-record output stores
MP_HD float resolve_k(float acc, int ns) { return floorf(fmaf(acc, 1.0f / (float)ns, 0.5f)); }

MP_HD bool view_finite(const float* T, const float* K) {
  bool ok = true;
  for (int i = 0; i < 16; ++i) ok = ok && isfinite(T[i]);
  for (int i = 0; i < 9; ++i) ok = ok && isfinite(K[i]);
  return ok;
}

}  // namespace rc
}  // namespace mp
