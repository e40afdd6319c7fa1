/* oracle/raster.c -- CPU software rasteriser.  TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Stands in for Panda3D/OpenGL, which the reference delegates to and which is absent here
 * ("parity unpinned vs Panda3D pixels"); it DEFINES the pixel contract that the HIP rasteriser
 * (megapose6d_amd/csrc/raster.hip) must reproduce bit-for-bit.
 * Reference contract being restated:
 *   /root/reference/src/megapose/panda3d_renderer/panda3d_batch_renderer.py:217-282 (render),
 *   :109-135 (non-finite pose -> zero images), :261-274 (uint8 -> /255, depth float),
 *   types.py:63-64 (near 0.1 / far 10), :75-101 (pinhole K, pixel (i,j) covers [i,i+1)x[j,j+1)),
 *   panda3d_scene_renderer.py:71-74 (texture-minfilter mipmap, framebuffer-multisample 1, multisamples 4),
 *   :99-101 (two-sided), :210-216 + utils.py:58-68 (eye-normal 32^3 LUT),
 *   :104-136 (ambient + point lights placed by a positioning function, 10 x bounding radius for make_scene_lights),
 *   utils.py:44-55 (metric depth, 0 = background).
 *
 * Contract v2 (round 2):
 *   geometry   Pc = R p + t (fmaf chain).  Triangles are CLIPPED against the near plane z = 0.1 m in camera space (a vertex is
 *              inside iff z >= 0.1): 1 inside vertex -> 1 piece, 2 inside -> 2 pieces; an intersection is always computed from the
 *              inside vertex towards the outside one, t = (0.1 - z_in) / (z_out - z_in), P = fmaf(t, P_out - P_in, P_in), z = 0.1
 *              (so the two triangles sharing an edge produce the identical point).  Piece ids for the depth tie rule: first piece
 *              = triangle index, second piece = n_faces + triangle index.  Screen: sx = fmaf(fx, x * (1/z), cx), snapped to 1/256
 *              px; a piece with a vertex beyond |16384| px is dropped.
 *   coverage   exact integer edge functions with the top-left rule, two-sided, evaluated per SAMPLE: 1 sample at the pixel centre
 *              (msaa = 1) or the standard 4-sample pattern (6,2) (14,6) (2,10) (10,14) / 16 px (msaa = 4: D3D / Vulkan "standard
 *              sample locations"; what Panda3D's GL driver uses is unpinned).
 *   depth      per sample: wsum = sum b_i / z_i at the SAMPLE position; kept iff 1/far <= wsum <= 1/near; nearest wins, equal depth
 *              -> lower piece id.
 *   shading    once per (pixel, winning piece) at the PIXEL CENTRE (OpenGL default, non-centroid: barycentrics are extrapolated
 *              when the centre is outside the piece), perspective-correct; per-sample colour / normal-LUT value is clamped and
 *              rounded to uint8 like the 8-bit multisample colour buffer, the pixel is the rounded mean of its samples (background
 *              samples = 0), then / 255.  Depth output = sample 0's own depth (a resolve of a multisample depth buffer picks one
 *              sample; 0 = background).
 *   texture    albedo = vertex colour x texture (modulate); repeat wrap; TRILINEAR probes with a per-pixel level of detail from the
 *              analytic screen-space derivatives of (u, v) at the pixel centre, and ANISOTROPIC filtering of degree 16 (the reference's
 *              `texture-anisotropic-degree 16`) by the formula of the OpenGL extension specification (contract v2.1, round 6):
 *              Px^2 = |d(uv)/dx|^2, Py^2 = |d(uv)/dy|^2 in texels, n = min(ceil(Pmax / Pmin), 16) probes along the major axis at
 *              x - 1/2 + i / (n + 1), i = 1..n, averaged; rho^2 = Pmax^2 / n^2; lambda = log2(rho) approximated piecewise-linearly
 *              from the float's exponent and mantissa (exact integer ops, no libm), level = floor(lambda), blend = lambda - level.
 *              (n = 1 is the isotropic trilinear sample of contract v2.  What Panda3D's GL driver really does with the degree is
 *              implementation-defined and unpinned here, as every pixel of the third-party renderer is.)
 *   lights     RGB = albedo * (ambient + sum_l color_l * max(0, n.l)), light l at dir_l * 10 * radius + offset_l (object frame).
 *
 * Written as a straightforward "for every piece, for every pixel of its bbox, for every sample" loop with plain per-sample
 * z-buffers, i.e. structurally different from the binned, tile/wave-parallel GPU kernel.
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile); fmaf() must be a real fused op.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SUBPIX 256
#define GUARD 16384.0f
#define Z_NEAR 0.1f
#define Z_FAR 10.0f
#define MAX_SAMPLES 4

static const int SAMPLE_OFF_1[1][2] = {{128, 128}};
static const int SAMPLE_OFF_4[4][2] = {{96, 32}, {224, 96}, {32, 160}, {160, 224}};

typedef struct {
  float x, y, z;   /* camera space */
  float bary[3];   /* weights on the ORIGINAL triangle's corners (unit vector for an original vertex) */
} cvert_t;

typedef struct {
  int X[3], Y[3];   /* snapped screen coordinates (1/256 px), positively oriented */
  float iz[3];
  float bary[3][3]; /* per piece vertex: weights on the original corners */
  int64_t A[3], B[3], C[3];
  int thr[3];
  float inv_area;
  int tri;          /* original triangle */
  int id;           /* depth tie id */
  int clipped;
} piece_t;

static float dot3p(float a0, float a1, float a2, float x, float y, float z, float t) {
  return fmaf(a2, z, fmaf(a1, y, fmaf(a0, x, t)));
}

static float lut_val(int i) { return (float)((i * 255) >> 5); }

static float normal_lut(float n) {
  const float u = n - floorf(n);
  const float t = fmaf(u, 32.0f, -0.5f);
  const float fl = floorf(t);
  const float f = t - fl;
  const int i0 = ((int)fl + 32) & 31;
  const int i1 = (i0 + 1) & 31;
  const float a = lut_val(i0), b = lut_val(i1);
  return fmaf(b - a, f, a);
}

/* clamp to [0,255] and round half up; NaN -> 0 (fmaxf returns the non-NaN operand) */
static float q255(float v255) { return floorf(fminf(fmaxf(v255, 0.f), 255.f) + 0.5f); }

typedef struct {
  const float* uvs;        /* [n_faces][3][2] or NULL */
  const uint32_t* texels;
  int w, h, levels;
} tex_t;

static int tex_offset(const tex_t* tx, int level) {
  int off = 0;
  for (int l = 0; l < level; ++l) {
    const int tw = (tx->w >> l) > 1 ? (tx->w >> l) : 1, th = (tx->h >> l) > 1 ? (tx->h >> l) : 1;
    off += tw * th;
  }
  return off;
}

/* degree of the anisotropic filter: 16 = the contract; oracle_set_max_aniso(1) is a TEST HOOK that renders the isotropic trilinear sample
 * of contract v2 instead, so that a test can show what the anisotropic probes change (tests/test_raster_contract_cpu.py) */
static int g_max_aniso = 16;
void oracle_set_max_aniso(int degree) { g_max_aniso = degree < 1 ? 1 : (degree > 16 ? 16 : degree); }

static void tex_sample(const tex_t* tx, int level, float u, float v, float out[3]) {
  const int tw = (tx->w >> level) > 1 ? (tx->w >> level) : 1, th = (tx->h >> level) > 1 ? (tx->h >> level) : 1;
  const uint32_t* t = tx->texels + tex_offset(tx, level);
  const float fu = fmaf(u - floorf(u), (float)tw, -0.5f), fv = fmaf(v - floorf(v), (float)th, -0.5f);
  const float flu = floorf(fu), flv = floorf(fv);
  const float au = fu - flu, av = fv - flv;
  int x0 = (int)flu, y0 = (int)flv;
  if (x0 < 0) x0 = tw - 1;
  if (y0 < 0) y0 = th - 1;
  if (x0 >= tw) x0 = tw - 1;
  if (y0 >= th) y0 = th - 1;
  const int x1 = (x0 + 1 == tw) ? 0 : x0 + 1, y1 = (y0 + 1 == th) ? 0 : y0 + 1;
  const uint32_t t00 = t[y0 * tw + x0], t01 = t[y0 * tw + x1], t10 = t[y1 * tw + x0], t11 = t[y1 * tw + x1];
  for (int c = 0; c < 3; ++c) {
    const float a00 = (float)((t00 >> (8 * c)) & 255u), a01 = (float)((t01 >> (8 * c)) & 255u);
    const float a10 = (float)((t10 >> (8 * c)) & 255u), a11 = (float)((t11 >> (8 * c)) & 255u);
    const float top = fmaf(a01 - a00, au, a00), bot = fmaf(a11 - a10, au, a10);
    out[c] = fmaf(bot - top, av, top);
  }
}

/* level of detail from rho^2 (squared texel footprint): lambda = log2(rho) ~ 0.5 * (exponent + mantissa fraction) of rho^2 */
static void tex_lod(const tex_t* tx, float rho2, int* level, float* frac) {
  *level = 0;
  *frac = 0.f;
  if (!(rho2 > 1.0f)) return;                    /* magnification (or NaN): base level */
  if (!(rho2 < 1e30f)) { *level = tx->levels - 1; return; }
  uint32_t bits;
  memcpy(&bits, &rho2, 4);
  const int e = (int)(bits >> 23) - 127;
  const float m = (float)(bits & 0x7FFFFFu) * (1.0f / 8388608.0f);
  const float lambda = 0.5f * ((float)e + m);
  const float fl = floorf(lambda);
  int l0 = (int)fl;
  float f = lambda - fl;
  if (l0 >= tx->levels - 1) { l0 = tx->levels - 1; f = 0.f; }
  *level = l0;
  *frac = f;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int floordiv256(int a) { return (a >= 0) ? (a / 256) : -((-a + 255) / 256); }

static void edge(int ax, int ay, int bx, int by, int64_t* A, int64_t* B, int64_t* C, int* thr) {
  const int64_t dx = (int64_t)bx - ax, dy = (int64_t)by - ay;
  *A = -dy;
  *B = dx;
  *C = dy * ax - dx * ay;
  *thr = ((dy < 0) || (dy == 0 && dx > 0)) ? 0 : 1; /* top-left edges own their boundary */
}

/* project + snap the three vertices of a piece, orient it; returns 0 if it is dropped */
static int finish_piece(const cvert_t v[3], const float* Kv, int tri, int id, int clipped, piece_t* p) {
  int X[3], Y[3];
  float iz[3];
  for (int k = 0; k < 3; ++k) {
    iz[k] = 1.0f / v[k].z;
    const float sx = fmaf(Kv[0], v[k].x * iz[k], Kv[2]);
    const float sy = fmaf(Kv[4], v[k].y * iz[k], Kv[5]);
    if (!(fabsf(sx) < GUARD && fabsf(sy) < GUARD)) return 0;
    X[k] = (int)rintf(sx * (float)SUBPIX);
    Y[k] = (int)rintf(sy * (float)SUBPIX);
  }
  int64_t area = ((int64_t)X[1] - X[0]) * ((int64_t)Y[2] - Y[0]) - ((int64_t)Y[1] - Y[0]) * ((int64_t)X[2] - X[0]);
  if (area == 0) return 0;
  int o[3] = {0, 1, 2};
  if (area < 0) { o[1] = 2; o[2] = 1; area = -area; }
  for (int k = 0; k < 3; ++k) {
    p->X[k] = X[o[k]]; p->Y[k] = Y[o[k]]; p->iz[k] = iz[o[k]];
    for (int j = 0; j < 3; ++j) p->bary[k][j] = v[o[k]].bary[j];
  }
  edge(p->X[1], p->Y[1], p->X[2], p->Y[2], &p->A[0], &p->B[0], &p->C[0], &p->thr[0]);
  edge(p->X[2], p->Y[2], p->X[0], p->Y[0], &p->A[1], &p->B[1], &p->C[1], &p->thr[1]);
  edge(p->X[0], p->Y[0], p->X[1], p->Y[1], &p->A[2], &p->B[2], &p->C[2], &p->thr[2]);
  p->inv_area = 1.0f / (float)area;
  p->tri = tri;
  p->id = id;
  p->clipped = clipped;
  return 1;
}

static cvert_t clip_edge(const cvert_t* in, const cvert_t* out, int i_in, int i_out) {
  cvert_t r;
  const float t = (Z_NEAR - in->z) / (out->z - in->z);
  r.x = fmaf(t, out->x - in->x, in->x);
  r.y = fmaf(t, out->y - in->y, in->y);
  r.z = Z_NEAR;
  r.bary[0] = r.bary[1] = r.bary[2] = 0.f;
  r.bary[i_in] = 1.0f - t;
  r.bary[i_out] = t;
  return r;
}

/* 0, 1 or 2 pieces of triangle `tri` (camera-space corners c[3]); returns the count */
static int make_pieces(const cvert_t c[3], const float* Kv, int tri, int n_faces, piece_t out[2]) {
  const int in0 = c[0].z >= Z_NEAR, in1 = c[1].z >= Z_NEAR, in2 = c[2].z >= Z_NEAR;
  const int n_in = in0 + in1 + in2;
  if (n_in == 0) return 0;
  if (n_in == 3) return finish_piece(c, Kv, tri, tri, 0, &out[0]);
  int n = 0;
  if (n_in == 1) {
    const int a = in0 ? 0 : (in1 ? 1 : 2), b = (a + 1) % 3, d = (a + 2) % 3; /* cyclic order a, b, d; a inside */
    cvert_t v[3];
    v[0] = c[a];
    v[1] = clip_edge(&c[a], &c[b], a, b);
    v[2] = clip_edge(&c[a], &c[d], a, d);
    n += finish_piece(v, Kv, tri, tri, 1, &out[n]);
  } else {
    const int o = !in0 ? 0 : (!in1 ? 1 : 2), a = (o + 1) % 3, b = (o + 2) % 3; /* cyclic order a, b, o; o outside */
    const cvert_t P = clip_edge(&c[b], &c[o], b, o), Q = clip_edge(&c[a], &c[o], a, o);
    cvert_t v[3];
    v[0] = c[a]; v[1] = c[b]; v[2] = P;
    n += finish_piece(v, Kv, tri, tri, 1, &out[n]);
    v[0] = c[a]; v[1] = P; v[2] = Q;
    n += finish_piece(v, Kv, tri, n_faces + tri, 1, &out[n]);
  }
  return n;
}

/* barycentrics + wsum of piece p at fixed-point position (sx, sy); returns coverage (top-left rule) */
static int eval_at(const piece_t* p, int64_t sx, int64_t sy, float b[3], float* wsum) {
  int inside = 1;
  for (int i = 0; i < 3; ++i) {
    const int64_t e = p->A[i] * sx + p->B[i] * sy + p->C[i];
    if (e < p->thr[i]) inside = 0;
    b[i] = (float)e * p->inv_area;
  }
  *wsum = fmaf(b[2], p->iz[2], fmaf(b[1], p->iz[1], b[0] * p->iz[0]));
  return inside;
}

typedef struct {
  float ambient[3];
  int32_t n_point;
  float dir[8][3];
  float color[8][3];
  float offset[8][3];
} lights_t;

/* attribute of piece vertex k: combination of the original corners' attributes with the piece vertex's weights */
static float pv_attr(const piece_t* p, int k, const float* attr, const int32_t* f3, int comp, int stride) {
  return fmaf(p->bary[k][2], attr[(size_t)stride * f3[2] + comp],
              fmaf(p->bary[k][1], attr[(size_t)stride * f3[1] + comp], p->bary[k][0] * attr[(size_t)stride * f3[0] + comp]));
}

/* shade piece p at the centre of pixel (px, py): col255[3] = RGB on the 0..255 scale BEFORE clamping/rounding,
 * nrm255[3] = eye-normal LUT values on the 0..255 scale */
static void shade(const piece_t* p, int px, int py, const float* verts, const float* normals, const float* colors, const int32_t* faces,
                  const tex_t* tx, const lights_t* L, float radius, const float* T, int gl_eye, float col255[3], float nrm255[3]) {
  float b[3], wsum;
  eval_at(p, (int64_t)px * SUBPIX + 128, (int64_t)py * SUBPIX + 128, b, &wsum);
  const float w0 = b[0] * p->iz[0], w1 = b[1] * p->iz[1], w2 = b[2] * p->iz[2];
  const float z = 1.0f / wsum;
  const int32_t* f3 = faces + 3 * (size_t)p->tri;
  float col[3], on[3];
  for (int k = 0; k < 3; ++k) {
    col[k] = fmaf(w2, pv_attr(p, 2, colors, f3, k, 3), fmaf(w1, pv_attr(p, 1, colors, f3, k, 3), w0 * pv_attr(p, 0, colors, f3, k, 3))) * z;
    on[k] = fmaf(w2, pv_attr(p, 2, normals, f3, k, 3), fmaf(w1, pv_attr(p, 1, normals, f3, k, 3), w0 * pv_attr(p, 0, normals, f3, k, 3))) * z;
  }
  if (tx->uvs && tx->texels) {
    const float* uv = tx->uvs + 6 * (size_t)p->tri; /* per-corner (u,v) of the original triangle, corner order of `faces` */
    float pu[3], pv[3];
    for (int k = 0; k < 3; ++k) {
      pu[k] = fmaf(p->bary[k][2], uv[4], fmaf(p->bary[k][1], uv[2], p->bary[k][0] * uv[0]));
      pv[k] = fmaf(p->bary[k][2], uv[5], fmaf(p->bary[k][1], uv[3], p->bary[k][0] * uv[1]));
    }
    const float u = fmaf(w2, pu[2], fmaf(w1, pu[1], w0 * pu[0])) * z;
    const float v = fmaf(w2, pv[2], fmaf(w1, pv[1], w0 * pv[0])) * z;
    /* analytic derivatives at the centre: b_i is affine in pixel coordinates, d b_i / dx = 256 * A_i * inv_area */
    float dbx[3], dby[3];
    for (int i = 0; i < 3; ++i) {
      dbx[i] = (float)(p->A[i] * 256) * p->inv_area * p->iz[i];
      dby[i] = (float)(p->B[i] * 256) * p->inv_area * p->iz[i];
    }
    const float dDx = dbx[2] + (dbx[1] + dbx[0]), dDy = dby[2] + (dby[1] + dby[0]);
    const float dNux = fmaf(dbx[2], pu[2], fmaf(dbx[1], pu[1], dbx[0] * pu[0])), dNuy = fmaf(dby[2], pu[2], fmaf(dby[1], pu[1], dby[0] * pu[0]));
    const float dNvx = fmaf(dbx[2], pv[2], fmaf(dbx[1], pv[1], dbx[0] * pv[0])), dNvy = fmaf(dby[2], pv[2], fmaf(dby[1], pv[1], dby[0] * pv[0]));
    const float tw = (float)tx->w, th = (float)tx->h;
    const float dudx = fmaf(-u, dDx, dNux) * z * tw, dvdx = fmaf(-v, dDx, dNvx) * z * th;
    const float dudy = fmaf(-u, dDy, dNuy) * z * tw, dvdy = fmaf(-v, dDy, dNvy) * z * th;
    const float rx2 = fmaf(dvdx, dvdx, dudx * dudx), ry2 = fmaf(dvdy, dvdy, dudy * dudy);
    /* anisotropic filtering of degree 16 (the reference's `texture-anisotropic-degree 16`, panda3d_scene_renderer.py:72) as the OpenGL
     * extension specifies it: n = min(ceil(Pmax / Pmin), 16) trilinear probes along the major axis of the footprint, at
     * x - 1/2 + i / (n + 1); level of detail from Pmax / n.  (Compared on squares: n is the first integer with n^2 * Pmin^2 >= Pmax^2.) */
    const int along_x = rx2 >= ry2;
    const float big2 = along_x ? rx2 : ry2, small2 = along_x ? ry2 : rx2;
    int n = 1;
    while (n < g_max_aniso && (float)(n * n) * small2 < big2) n++;
    int level;
    float frac;
    tex_lod(tx, big2 / (float)(n * n), &level, &frac);
    const float axis_u = (along_x ? dudx : dudy) / tw, axis_v = (along_x ? dvdx : dvdy) / th;
    float sum[3] = {0.f, 0.f, 0.f};
    for (int i = 1; i <= n; ++i) {
      const float t = (float)i / (float)(n + 1) - 0.5f;
      const float us = fmaf(t, axis_u, u), vs = fmaf(t, axis_v, v);
      float tc[3], tc1[3];
      tex_sample(tx, level, us, vs, tc);
      if (frac > 0.f) {
        tex_sample(tx, level + 1, us, vs, tc1);
        for (int k = 0; k < 3; ++k) tc[k] = fmaf(tc1[k] - tc[k], frac, tc[k]);
      }
      for (int k = 0; k < 3; ++k) sum[k] += tc[k];
    }
    const float inv_n = 1.0f / (float)n;
    for (int k = 0; k < 3; ++k) col[k] *= (sum[k] * inv_n) / 255.0f;
  }
  float lr = L->ambient[0], lg = L->ambient[1], lb = L->ambient[2];
  if (L->n_point > 0) {
    float op[3];
    for (int k = 0; k < 3; ++k)
      op[k] = fmaf(w2, pv_attr(p, 2, verts, f3, k, 3), fmaf(w1, pv_attr(p, 1, verts, f3, k, 3), w0 * pv_attr(p, 0, verts, f3, k, 3))) * z;
    const float nn = sqrtf(fmaf(on[2], on[2], fmaf(on[1], on[1], on[0] * on[0])));
    const float inn = nn > 0.f ? 1.0f / nn : 0.f;
    const float R10 = 10.0f * radius;
    for (int l = 0; l < L->n_point; ++l) {
      const float lx = fmaf(L->dir[l][0], R10, L->offset[l][0]) - op[0];
      const float ly = fmaf(L->dir[l][1], R10, L->offset[l][1]) - op[1];
      const float lz = fmaf(L->dir[l][2], R10, L->offset[l][2]) - op[2];
      const float ln = sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
      const float d = fmaf(lz, on[2], fmaf(ly, on[1], lx * on[0])) * inn / ln;
      const float dd = fmaxf(d, 0.f);
      lr = fmaf(L->color[l][0], dd, lr);
      lg = fmaf(L->color[l][1], dd, lg);
      lb = fmaf(L->color[l][2], dd, lb);
    }
  }
  col255[0] = col[0] * lr * 255.0f; col255[1] = col[1] * lg * 255.0f; col255[2] = col[2] * lb * 255.0f;
  const float cx = fmaf(T[2], on[2], fmaf(T[1], on[1], T[0] * on[0]));
  const float cy = fmaf(T[6], on[2], fmaf(T[5], on[1], T[4] * on[0]));
  const float cz = fmaf(T[10], on[2], fmaf(T[9], on[1], T[8] * on[0]));
  float e[3];
  if (gl_eye) { e[0] = cx; e[1] = -cy; e[2] = -cz; }
  else { e[0] = cx; e[1] = cz; e[2] = -cy; }
  for (int k = 0; k < 3; ++k) nrm255[k] = normal_lut(e[k]);
}

/* flags: 1 normals, 2 depth, 4 GL eye axes, 8 no quantisation, 16 msaa 4x.
 * Outputs (any may be NULL): rgb [n,h,w,3], normals [n,h,w,3], depth [n,h,w]. */
void oracle_raster_render(const float* verts, const float* normals, const float* colors, const int32_t* faces, int n_verts,
                          int n_faces, float radius, const float* TCO, const float* K, int n_views, int h, int w,
                          uint32_t flags, const lights_t* L, float* out_rgb, float* out_normals, float* out_depth,
                          const float* uvs, const uint32_t* texels, int tex_w, int tex_h, int tex_levels) {
  const tex_t tx = {uvs, texels, tex_w, tex_h, tex_levels};
  const int ns = (flags & 16u) ? 4 : 1;
  const int (*soff)[2] = (flags & 16u) ? SAMPLE_OFF_4 : SAMPLE_OFF_1;
  cvert_t* cv = (cvert_t*)malloc(sizeof(cvert_t) * (size_t)n_verts);
  float* zb = (float*)malloc(sizeof(float) * (size_t)h * w * ns);   /* best wsum per sample */
  int* ib = (int*)malloc(sizeof(int) * (size_t)h * w * ns);         /* winning piece slot per sample (-1 = none) */
  piece_t* pieces = (piece_t*)malloc(sizeof(piece_t) * (size_t)n_faces * 2);
  const int do_norm = (flags & 1u) && out_normals, do_depth = (flags & 2u) && out_depth;
  const int gl_eye = (flags & 4u) != 0, no_quant = (flags & 8u) != 0;
  int offmin = 256, offmax = 0;
  for (int s = 0; s < ns; ++s) { offmin = imin(offmin, imin(soff[s][0], soff[s][1])); offmax = imax(offmax, imax(soff[s][0], soff[s][1])); }
  for (int view = 0; view < n_views; ++view) {
    const float* T = TCO + (size_t)view * 16;
    const float* Kv = K + (size_t)view * 9;
    float* rgb = out_rgb ? out_rgb + (size_t)view * h * w * 3 : NULL;
    float* nrm = do_norm ? out_normals + (size_t)view * h * w * 3 : NULL;
    float* dep = do_depth ? out_depth + (size_t)view * h * w : NULL;
    if (rgb) memset(rgb, 0, sizeof(float) * (size_t)h * w * 3);
    if (nrm) memset(nrm, 0, sizeof(float) * (size_t)h * w * 3);
    if (dep) memset(dep, 0, sizeof(float) * (size_t)h * w);
    int finite = 1;
    for (int i = 0; i < 16; ++i) finite = finite && isfinite(T[i]);
    for (int i = 0; i < 9; ++i) finite = finite && isfinite(Kv[i]);
    if (!finite) continue; /* panda3d_batch_renderer.py:109-135 */
    for (int v = 0; v < n_verts; ++v) {
      const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
      cv[v].x = dot3p(T[0], T[1], T[2], px, py, pz, T[3]);
      cv[v].y = dot3p(T[4], T[5], T[6], px, py, pz, T[7]);
      cv[v].z = dot3p(T[8], T[9], T[10], px, py, pz, T[11]);
    }
    int n_pieces = 0;
    for (int t = 0; t < n_faces; ++t) {
      cvert_t c[3];
      for (int k = 0; k < 3; ++k) {
        c[k] = cv[faces[3 * t + k]];
        c[k].bary[0] = c[k].bary[1] = c[k].bary[2] = 0.f;
        c[k].bary[k] = 1.0f;
      }
      n_pieces += make_pieces(c, Kv, t, n_faces, pieces + n_pieces);
    }
    for (int i = 0; i < h * w * ns; ++i) { zb[i] = 0.f; ib[i] = -1; }
    for (int q = 0; q < n_pieces; ++q) {
      const piece_t* p = &pieces[q];
      const int Xmin = imin(p->X[0], imin(p->X[1], p->X[2])), Xmax = imax(p->X[0], imax(p->X[1], p->X[2]));
      const int Ymin = imin(p->Y[0], imin(p->Y[1], p->Y[2])), Ymax = imax(p->Y[0], imax(p->Y[1], p->Y[2]));
      const int x0 = imax(0, floordiv256(Xmin - offmax + 255)), x1 = imin(w - 1, floordiv256(Xmax - offmin));
      const int y0 = imax(0, floordiv256(Ymin - offmax + 255)), y1 = imin(h - 1, floordiv256(Ymax - offmin));
      for (int py = y0; py <= y1; ++py)
        for (int px = x0; px <= x1; ++px)
          for (int s = 0; s < ns; ++s) {
            float b[3], wsum;
            if (!eval_at(p, (int64_t)px * SUBPIX + soff[s][0], (int64_t)py * SUBPIX + soff[s][1], b, &wsum)) continue;
            if (!(wsum >= 1.0f / Z_FAR && wsum <= 1.0f / Z_NEAR)) continue;
            const size_t k = ((size_t)py * w + px) * ns + s;
            if (ib[k] < 0 || wsum > zb[k] || (wsum == zb[k] && p->id < pieces[ib[k]].id)) {
              zb[k] = wsum;
              ib[k] = q;
            }
          }
    }
    /* resolve */
    for (int py = 0; py < h; ++py)
      for (int px = 0; px < w; ++px) {
        const size_t k0 = ((size_t)py * w + px) * ns;
        float acc_c[3] = {0.f, 0.f, 0.f}, acc_n[3] = {0.f, 0.f, 0.f};
        int any = 0;
        for (int s = 0; s < ns; ++s) {
          const int q = ib[k0 + s];
          if (q < 0) continue;
          any = 1;
          float c255[3], n255[3];
          shade(&pieces[q], px, py, verts, normals, colors, faces, &tx, L, radius, T, gl_eye, c255, n255);
          for (int c = 0; c < 3; ++c) {
            acc_c[c] += no_quant ? c255[c] : q255(c255[c]);
            acc_n[c] += no_quant ? n255[c] : q255(n255[c]);
          }
        }
        if (dep && ib[k0] >= 0) dep[(size_t)py * w + px] = 1.0f / zb[k0];
        if (!any) continue;
        const float inv_ns = 1.0f / (float)ns; /* 1 or 0.25: exact */
        for (int c = 0; c < 3; ++c) {
          if (rgb) rgb[((size_t)py * w + px) * 3 + c] = no_quant ? (acc_c[c] * inv_ns) / 255.0f : floorf(fmaf(acc_c[c], inv_ns, 0.5f)) / 255.0f;
          if (nrm) nrm[((size_t)py * w + px) * 3 + c] = no_quant ? (acc_n[c] * inv_ns) / 255.0f : floorf(fmaf(acc_n[c], inv_ns, 0.5f)) / 255.0f;
        }
      }
  }
  free(cv);
  free(zb);
  free(ib);
  free(pieces);
}
