/* oracle/raster.c -- CPU software rasteriser.  TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Stands in for Panda3D/OpenGL, which the reference delegates to and which is absent here
 * ("parity unpinned vs Panda3D pixels"); it DEFINES the pixel contract that the HIP rasteriser
 * (megapose6d_amd/csrc/raster.hip) must reproduce bit-for-bit.
 * Reference contract being restated:
 *   /root/reference/src/megapose/panda3d_renderer/panda3d_batch_renderer.py:217-282 (render),
 *   :109-135 (non-finite pose -> zero images), :261-274 (uint8 -> /255, depth float),
 *   types.py:63-64 (near 0.1 / far 10), :75-101 (pinhole K, pixel (i,j) covers [i,i+1)x[j,j+1)),
 *   panda3d_scene_renderer.py:99-101 (two-sided), :210-216 + utils.py:58-68 (eye-normal 32^3 LUT),
 *   :104-136 (ambient + 6 point lights at 10 x bounding radius), utils.py:44-55 (metric depth, 0 = background).
 *
 * Written as a straightforward scanline-free "for every triangle, for every pixel of its bbox" loop with a
 * plain z-buffer, i.e. structurally different from the banded/atomic GPU kernel.
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile); fmaf() must be a real fused op.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SUBPIX 256
#define GUARD 16384.0f
#define Z_EPS 1e-6f
#define Z_NEAR 0.1f
#define Z_FAR 10.0f

typedef struct {
  int X, Y;
  float invz;
  int valid;
} vtx_t;

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

static float quant8(float v255) {
  const float q = floorf(fminf(fmaxf(v255, 0.f), 255.f) + 0.5f);
  return q / 255.0f;
}

/* UV texture: RGBA8 mip chain (levels concatenated, level l = max(1,w>>l) x max(1,h>>l)), repeat wrap, bilinear.
 * Replaces the texture stage of Panda3D's auto-shader (panda3d_scene_renderer.py:192-207 loads the textured model);
 * pixel parity with OpenGL's trilinear/anisotropic filtering is unpinned -- this is the engine's contract. */
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

static int tex_level(const tex_t* tx, const float* uv, float inv_area2) {
  const float du1 = uv[2] - uv[0], dv1 = uv[3] - uv[1], du2 = uv[4] - uv[0], dv2 = uv[5] - uv[1];
  const float at = fabsf(du1 * dv2 - du2 * dv1) * ((float)tx->w * (float)tx->h);
  const float r = at * (inv_area2 * 65536.0f);
  int level = 0;
  float thr = 2.0f;
  while (level + 1 < tx->levels && r > thr) { ++level; thr *= 4.0f; }
  return level;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* floor division by 256 for possibly negative ints */
static int floordiv256(int a) { return (a >= 0) ? (a / 256) : -((-a + 255) / 256); }

typedef struct {
  int64_t A[3], B[3], Cc[3];
  int thr[3];
  float inv_area; /* 1 / (float)(2 * area) */
  float iz[3];
  int idx[3];
} tri_t;

static void edge(const vtx_t* a, const vtx_t* b, int64_t* A, int64_t* B, int64_t* C, int* thr) {
  const int64_t dx = (int64_t)b->X - a->X, dy = (int64_t)b->Y - a->Y;
  *A = -dy;
  *B = dx;
  *C = dy * a->X - dx * a->Y;
  *thr = ((dy < 0) || (dy == 0 && dx > 0)) ? 0 : 1; /* top-left edges own their boundary */
}

/* returns 0 if the triangle is culled */
static int setup(const vtx_t* vv, int i0, int i1, int i2, tri_t* t) {
  if (!(vv[i0].valid && vv[i1].valid && vv[i2].valid)) return 0;
  int64_t area = ((int64_t)vv[i1].X - vv[i0].X) * ((int64_t)vv[i2].Y - vv[i0].Y) -
                 ((int64_t)vv[i1].Y - vv[i0].Y) * ((int64_t)vv[i2].X - vv[i0].X);
  if (area == 0) return 0;
  if (area < 0) { int s = i1; i1 = i2; i2 = s; area = -area; }
  t->idx[0] = i0; t->idx[1] = i1; t->idx[2] = i2;
  edge(&vv[i1], &vv[i2], &t->A[0], &t->B[0], &t->Cc[0], &t->thr[0]);
  edge(&vv[i2], &vv[i0], &t->A[1], &t->B[1], &t->Cc[1], &t->thr[1]);
  edge(&vv[i0], &vv[i1], &t->A[2], &t->B[2], &t->Cc[2], &t->thr[2]);
  t->inv_area = 1.0f / (float)area;
  t->iz[0] = vv[i0].invz; t->iz[1] = vv[i1].invz; t->iz[2] = vv[i2].invz;
  return 1;
}

static int sample(const tri_t* t, int px, int py, float b[3], float* wsum) {
  const int64_t sx = (int64_t)px * SUBPIX + 128, sy = (int64_t)py * SUBPIX + 128;
  for (int i = 0; i < 3; ++i) {
    const int64_t e = t->A[i] * sx + t->B[i] * sy + t->Cc[i];
    if (e < t->thr[i]) return 0;
    b[i] = (float)e * t->inv_area; /* int64 -> float is correctly rounded; the GPU converts the same integer from fp64 */
  }
  *wsum = fmaf(b[2], t->iz[2], fmaf(b[1], t->iz[1], b[0] * t->iz[0]));
  return 1;
}

/* flags: 1 normals, 2 depth, 4 GL eye axes, 8 no quantisation.
 * lights: ambient[3], n_point, dir[8][3], color[8][3] packed as floats/ints exactly like mp_lights.
 * Outputs (any may be NULL): rgb [n,h,w,3], normals [n,h,w,3], depth [n,h,w]. */
typedef struct {
  float ambient[3];
  int32_t n_point;
  float dir[8][3];
  float color[8][3];
} lights_t;

void oracle_raster_render(const float* verts, const float* normals, const float* colors, const int32_t* faces, int n_verts,
                          int n_faces, float radius, const float* TCO, const float* K, int n_views, int h, int w,
                          uint32_t flags, const lights_t* L, float* out_rgb, float* out_normals, float* out_depth,
                          const float* uvs, const uint32_t* texels, int tex_w, int tex_h, int tex_levels) {
  const tex_t tx = {uvs, texels, tex_w, tex_h, tex_levels};
  vtx_t* vv = (vtx_t*)malloc(sizeof(vtx_t) * (size_t)n_verts);
  float* zb = (float*)malloc(sizeof(float) * (size_t)h * w);   /* best wsum so far (0 = empty) */
  int* tb = (int*)malloc(sizeof(int) * (size_t)h * w);
  const int do_norm = (flags & 1u) && out_normals, do_depth = (flags & 2u) && out_depth;
  const int gl_eye = flags & 4u, no_quant = flags & 8u;
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
      vtx_t r = {0, 0, 0.f, 0};
      const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
      const float x = dot3p(T[0], T[1], T[2], px, py, pz, T[3]);
      const float y = dot3p(T[4], T[5], T[6], px, py, pz, T[7]);
      const float z = dot3p(T[8], T[9], T[10], px, py, pz, T[11]);
      if (z > Z_EPS) {
        const float iz = 1.0f / z;
        const float sx = fmaf(Kv[0], x * iz, Kv[2]);
        const float sy = fmaf(Kv[4], y * iz, Kv[5]);
        if (fabsf(sx) < GUARD && fabsf(sy) < GUARD) {
          r.X = (int)rintf(sx * (float)SUBPIX);
          r.Y = (int)rintf(sy * (float)SUBPIX);
          r.invz = iz;
          r.valid = 1;
        }
      }
      vv[v] = r;
    }
    for (int i = 0; i < h * w; ++i) { zb[i] = 0.f; tb[i] = -1; }
    /* coverage + depth: triangles in index order, strictly-closer replaces (ties keep the lower index) */
    for (int t = 0; t < n_faces; ++t) {
      tri_t tr;
      if (!setup(vv, faces[3 * t], faces[3 * t + 1], faces[3 * t + 2], &tr)) continue;
      const vtx_t *a = &vv[tr.idx[0]], *b = &vv[tr.idx[1]], *c = &vv[tr.idx[2]];
      const int Xmin = imin(a->X, imin(b->X, c->X)), Xmax = imax(a->X, imax(b->X, c->X));
      const int Ymin = imin(a->Y, imin(b->Y, c->Y)), Ymax = imax(a->Y, imax(b->Y, c->Y));
      const int x0 = imax(0, floordiv256(Xmin - 128 + 255)), x1 = imin(w - 1, floordiv256(Xmax - 128));
      const int y0 = imax(0, floordiv256(Ymin - 128 + 255)), y1 = imin(h - 1, floordiv256(Ymax - 128));
      for (int py = y0; py <= y1; ++py)
        for (int px = x0; px <= x1; ++px) {
          float bb[3], wsum;
          if (!sample(&tr, px, py, bb, &wsum)) continue;
          if (!(wsum >= 1.0f / Z_FAR && wsum <= 1.0f / Z_NEAR)) continue;
          if (tb[py * w + px] < 0 || wsum > zb[py * w + px]) {
            zb[py * w + px] = wsum;
            tb[py * w + px] = t;
          }
        }
    }
    /* shading */
    for (int py = 0; py < h; ++py)
      for (int px = 0; px < w; ++px) {
        const int t = tb[py * w + px];
        if (t < 0) continue;
        tri_t tr;
        setup(vv, faces[3 * t], faces[3 * t + 1], faces[3 * t + 2], &tr);
        float b[3], wsum;
        sample(&tr, px, py, b, &wsum);
        const int i0 = tr.idx[0], i1 = tr.idx[1], i2 = tr.idx[2];
        const float w0 = b[0] * tr.iz[0], w1 = b[1] * tr.iz[1], w2 = b[2] * tr.iz[2];
        const float z = 1.0f / wsum;
        float col[3], on[3];
        for (int k = 0; k < 3; ++k) {
          col[k] = fmaf(w2, colors[3 * i2 + k], fmaf(w1, colors[3 * i1 + k], w0 * colors[3 * i0 + k])) * z;
          on[k] = fmaf(w2, normals[3 * i2 + k], fmaf(w1, normals[3 * i1 + k], w0 * normals[3 * i0 + k])) * z;
        }
        if (tx.uvs && tx.texels) {
          const float* uv = tx.uvs + 6 * (size_t)t;
          const int k1 = (i1 == faces[3 * t + 1]) ? 1 : 2, k2 = 3 - k1; /* corner slots follow the orientation swap */
          const float u = fmaf(w2, uv[2 * k2], fmaf(w1, uv[2 * k1], w0 * uv[0])) * z;
          const float v = fmaf(w2, uv[2 * k2 + 1], fmaf(w1, uv[2 * k1 + 1], w0 * uv[1])) * z;
          float tc[3];
          tex_sample(&tx, tex_level(&tx, uv, tr.inv_area), u, v, tc);
          for (int k = 0; k < 3; ++k) col[k] *= tc[k] / 255.0f;
        }
        float lr = L->ambient[0], lg = L->ambient[1], lb = L->ambient[2];
        if (L->n_point > 0) {
          float op[3];
          for (int k = 0; k < 3; ++k)
            op[k] = fmaf(w2, verts[3 * i2 + k], fmaf(w1, verts[3 * i1 + k], w0 * verts[3 * i0 + k])) * z;
          const float nn = sqrtf(fmaf(on[2], on[2], fmaf(on[1], on[1], on[0] * on[0])));
          const float inn = nn > 0.f ? 1.0f / nn : 0.f;
          const float R10 = 10.0f * radius;
          for (int l = 0; l < L->n_point; ++l) {
            const float lx = fmaf(L->dir[l][0], R10, -op[0]);
            const float ly = fmaf(L->dir[l][1], R10, -op[1]);
            const float lz = fmaf(L->dir[l][2], R10, -op[2]);
            const float ln = sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
            const float d = fmaf(lz, on[2], fmaf(ly, on[1], lx * on[0])) * inn / ln;
            const float dd = fmaxf(d, 0.f);
            lr = fmaf(L->color[l][0], dd, lr);
            lg = fmaf(L->color[l][1], dd, lg);
            lb = fmaf(L->color[l][2], dd, lb);
          }
        }
        col[0] *= lr; col[1] *= lg; col[2] *= lb;
        if (rgb) {
          float* o = rgb + ((size_t)py * w + px) * 3;
          for (int k = 0; k < 3; ++k) o[k] = no_quant ? col[k] : quant8(col[k] * 255.0f);
        }
        if (nrm) {
          const float cx = fmaf(T[2], on[2], fmaf(T[1], on[1], T[0] * on[0]));
          const float cy = fmaf(T[6], on[2], fmaf(T[5], on[1], T[4] * on[0]));
          const float cz = fmaf(T[10], on[2], fmaf(T[9], on[1], T[8] * on[0]));
          float e[3];
          if (gl_eye) { e[0] = cx; e[1] = -cy; e[2] = -cz; }
          else { e[0] = cx; e[1] = cz; e[2] = -cy; }
          float* o = nrm + ((size_t)py * w + px) * 3;
          for (int k = 0; k < 3; ++k) o[k] = no_quant ? normal_lut(e[k]) / 255.0f : quant8(normal_lut(e[k]));
        }
        if (dep) dep[(size_t)py * w + px] = z;
      }
  }
  free(vv);
  free(zb);
  free(tb);
}
