// tests/raster_emul.cpp -- HOST emulation of the two rasteriser kernels (megapose6d_amd/csrc/raster.hip), TEST INFRASTRUCTURE ONLY.
// It executes the same algorithm in the same order -- binning of the pieces into 8x8 tiles (+ the "large" list and the overflow
// fallback), per-tile coverage with the lane arithmetic of raster_core.h (incl. the 32-bit small-piece path), one shading task
// per (pixel, distinct winning piece), the 8-bit multisample resolve and the channel-run staging -- serially on the CPU, so that
// the pixel contract implemented by the DEVICE code's shared core can be compared with the independent oracle (oracle/raster.c)
// in the CPU test tier.  What it cannot cover is the device-only glue (wave broadcasts, LDS hand-offs, atomics): the -m gpu
// tests compare the real kernels with the oracle bit for bit.
// Build: g++ -O2 -std=c++17 -ffp-contract=off -mfma -shared -fPIC -I megapose6d_amd/csrc tests/raster_emul.cpp
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "raster_core.h"

using namespace mp::rc;

namespace {
constexpr int LARGE_TILES = 16;
constexpr int TILE_WAVES = 4;
constexpr int SCATTER_MAX_AREA = 32;

struct Lists {
  std::vector<int> tile_off, large;
  std::vector<TileRec> list;     // per (small piece, touched tile) records, as the binning kernel stores them
  bool overflow = false;
};

void tile_range(const Piece& p, int ns, int w, int h, int& tx0, int& ty0, int& tx1, int& ty1) {
  int x0, y0, x1, y1;
  piece_pixel_bbox(p, ns, w, h, x0, y0, x1, y1);
  tx0 = x0 >> 3; ty0 = y0 >> 3; tx1 = x1 >> 3; ty1 = y1 >> 3;
  if (x0 > x1 || y0 > y1) { tx1 = tx0 - 1; ty1 = ty0 - 1; }
}

// the binning kernel's conservative tile test (exact integers)
struct TileTest {
  long long A[3], B[3], C[3];
  int omin, omax;
};
TileTest tile_test_setup(const Piece& p, int ns) {
  TileTest t;
  const int a[3] = {1, 2, 0}, b[3] = {2, 0, 1};
  for (int i = 0; i < 3; ++i) {
    const long long dx = p.X[b[i]] - p.X[a[i]], dy = p.Y[b[i]] - p.Y[a[i]];
    t.A[i] = -dy; t.B[i] = dx; t.C[i] = dy * p.X[a[i]] - dx * p.Y[a[i]];
  }
  t.omin = sample_off_min(ns); t.omax = sample_off_max(ns);
  return t;
}
bool tile_touched(const TileTest& t, int tx, int ty) {
  const long long x_lo = tx * TILE * SUBPIX + t.omin, x_hi = (tx * TILE + TILE - 1) * SUBPIX + t.omax;
  const long long y_lo = ty * TILE * SUBPIX + t.omin, y_hi = (ty * TILE + TILE - 1) * SUBPIX + t.omax;
  for (int i = 0; i < 3; ++i)
    if (t.A[i] * (t.A[i] >= 0 ? x_hi : x_lo) + t.B[i] * (t.B[i] >= 0 ? y_hi : y_lo) + t.C[i] < 0) return false;
  return true;
}

Lists bin_view(const MeshRef& m, const float* T, const float* Kv, int h, int w, int ns, int cap_list) {
  const int tiles_x = (w + TILE - 1) / TILE, tiles_y = (h + TILE - 1) / TILE, n_tiles = tiles_x * tiles_y;
  Lists L;
  std::vector<int> counts(n_tiles, 0);
  const int F = view_finite(T, Kv) ? m.n_faces : 0;
  for (int t = 0; t < F; ++t) {
    int n_pieces = 1;
    for (int which = 0; which < n_pieces; ++which) {
      Piece p;
      n_pieces = make_piece<false>(m, T, Kv, t, which, p);
      if (p.id < 0) continue;
      int tx0, ty0, tx1, ty1;
      tile_range(p, ns, w, h, tx0, ty0, tx1, ty1);
      if (tx0 > tx1 || ty0 > ty1) continue;
      if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > LARGE_TILES || piece_extent(p) > SMALL_EXTENT) L.large.push_back(p.id);
      else {
        const TileTest tt = tile_test_setup(p, ns);
        for (int ty = ty0; ty <= ty1; ++ty)
          for (int tx = tx0; tx <= tx1; ++tx)
            if (tile_touched(tt, tx, ty)) ++counts[ty * tiles_x + tx];
      }
    }
  }
  L.tile_off.assign(n_tiles + 1, 0);
  for (int i = 0; i < n_tiles; ++i) L.tile_off[i + 1] = L.tile_off[i] + counts[i];
  L.overflow = L.tile_off[n_tiles] > cap_list;
  if (L.overflow) return L;
  L.list.assign(L.tile_off[n_tiles], TileRec{});
  std::vector<int> cursor(L.tile_off.begin(), L.tile_off.end() - 1);
  for (int t = 0; t < F; ++t) {
    int n_pieces = 1;
    for (int which = 0; which < n_pieces; ++which) {
      Piece p;
      n_pieces = make_piece<false>(m, T, Kv, t, which, p);
      if (p.id < 0) continue;
      int tx0, ty0, tx1, ty1;
      tile_range(p, ns, w, h, tx0, ty0, tx1, ty1);
      if (tx0 > tx1 || ty0 > ty1 || (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > LARGE_TILES || piece_extent(p) > SMALL_EXTENT) continue;
      const TileTest tt = tile_test_setup(p, ns);
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          if (tile_touched(tt, tx, ty)) L.list[cursor[ty * tiles_x + tx]++] = pack_tile_rec(p, tx * TILE, ty * TILE);
    }
  }
  return L;
}

template <int NS>
void render_views(const MeshRef* meshes, const TexRef* texs, const int32_t* mesh_ids, const float* TCO, const float* K, int n_views, int h,
                  int w, uint32_t flags, const Lights& lights, float* out, long long stride_v, int views_per_item, long long stride_view,
                  long long stride_y, long long stride_x, int c_rgb, int c_normals, int c_depth, int cap_list, int reverse_lists) {
  const bool do_norm = (flags & 1u) && c_normals >= 0, do_depth = (flags & 2u) && c_depth >= 0, gl_eye = flags & 4u;
  const int tiles_x = (w + TILE - 1) / TILE, tiles_y = (h + TILE - 1) / TILE;
  for (int view = 0; view < n_views; ++view) {
    const int item = view / views_per_item, r = view % views_per_item;
    const MeshRef& m = meshes[mesh_ids[view]];
    const TexRef* tex = m.uvs ? &texs[mesh_ids[view]] : nullptr;
    const float* T = TCO + (size_t)view * 16;
    const float* Kv = K + (size_t)view * 9;
    Lists L = bin_view(m, T, Kv, h, w, NS, cap_list);
    if (reverse_lists) {  // the fill order of the device lists is not deterministic: the result must not depend on it
      std::reverse(L.large.begin(), L.large.end());
      for (int t = 0; t + 1 < (int)L.tile_off.size() && !L.overflow; ++t) std::reverse(L.list.begin() + L.tile_off[t], L.list.begin() + L.tile_off[t + 1]);
    }
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int tx = 0; tx < tiles_x; ++tx) {
        const int tile = ty * tiles_x + tx, tile_x0 = tx * TILE, tile_y0 = ty * TILE;
        // the wave's z-buffer: 64-bit keys, as in the kernel (scatter form for small footprints, sweep form for the rest)
        unsigned long long zb[64 * NS];
        for (int i = 0; i < 64 * NS; ++i) zb[i] = 0ull;
        std::vector<Piece> entries;   // binned records first (unpacked relative to THIS tile), then the recomputed large pieces
        std::vector<int> slots;       // position in the tile's record list (SLOT_NONE for recomputed pieces), as carried by the keys
        if (L.overflow) {
          for (int i = 0; i < 2 * m.n_faces; ++i) { Piece q; piece_from_index<true>(m, T, Kv, i, q); entries.push_back(q); slots.push_back(SLOT_NONE); }
        } else {
          for (int e = L.tile_off[tile]; e < L.tile_off[tile + 1]; ++e) {
            Piece q;
            unpack_tile_rec(L.list[e], tile_x0, tile_y0, q);
            entries.push_back(q);
            slots.push_back(imin(e - L.tile_off[tile], SLOT_NONE));
          }
          for (int idx : L.large) {   // the tile's large list = the large pieces that can own a sample in it (exact tile test, as binned)
            Piece q;
            piece_from_index<true>(m, T, Kv, idx, q);
            if (q.id < 0 || !tile_touched(tile_test_setup(q, NS), tx, ty)) continue;
            entries.push_back(q);
            slots.push_back(SLOT_NONE);
          }
        }
        for (size_t ei = 0; ei < entries.size(); ++ei) {
          const Piece& p = entries[ei];
          const int pslot = slots[ei];
          const bool binned = !L.overflow && (int)ei < L.tile_off[tile + 1] - L.tile_off[tile];
          if (p.id < 0) continue;
          int x0, y0, x1, y1;
          piece_pixel_bbox(p, NS, w, h, x0, y0, x1, y1);
          x0 = imax(x0, tile_x0); y0 = imax(y0, tile_y0); x1 = imin(x1, tile_x0 + TILE - 1); y1 = imin(y1, tile_y0 + TILE - 1);
          if (x0 > x1 || y0 > y1) continue;
          const bool small = piece_is_small(p, tile_x0, tile_y0);
          if (binned) {   // block-visit form (every binned record): the arithmetic and the lane layout of the kernel
            Edges32 e;
            piece_edges32(p, e);
            const BlkRec br = make_blk_rec(p, e, tile_x0, tile_y0, pslot);
            const int ox = tile_x0 * SUBPIX, oy = tile_y0 * SUBPIX;
            const int rxmin = imin(p.X[0], imin(p.X[1], p.X[2])) - ox, rxmax = imax(p.X[0], imax(p.X[1], p.X[2])) - ox;
            const int rymin = imin(p.Y[0], imin(p.Y[1], p.Y[2])) - oy, rymax = imax(p.Y[0], imax(p.Y[1], p.Y[2])) - oy;
            for (int k = 0; k < blk_count(NS); ++k) {
              if (!blk_touched(br, rxmin, rxmax, rymin, rymax, NS, k)) continue;
              for (int l = 0; l < 64; ++l) {
                const int pix = blk_lane_pixel(NS, k, l), s = blk_lane_sample(NS, l);
                if (tile_x0 + (pix & 7) >= w || tile_y0 + (pix >> 3) >= h) continue;
                cover_sample_rel(br, blk_lane_rel(NS, k, l), [&](float wsum) {
                  const unsigned long long key = depth_key_lo(wsum, br.key_lo);
                  if (key > zb[pix * NS + s]) zb[pix * NS + s] = key;
                });
              }
            }
          } else {                                                             // sweep form
            const int Xmin = imin(p.X[0], imin(p.X[1], p.X[2])), Xmax = imax(p.X[0], imax(p.X[1], p.X[2]));
            const int Ymin = imin(p.Y[0], imin(p.Y[1], p.Y[2])), Ymax = imax(p.Y[0], imax(p.Y[1], p.Y[2]));
            const int sx0 = tile_x0 * SUBPIX, sy0 = tile_y0 * SUBPIX;
            if (Xmax < sx0 || Xmin >= sx0 + TILE * SUBPIX || Ymax < sy0 || Ymin >= sy0 + TILE * SUBPIX) continue;
            for (int l = 0; l < 64; ++l) {
              auto emit = [&](int s, float wsum) {
                const unsigned long long key = depth_key(wsum, p.id, SLOT_NONE);
                if (key > zb[l * NS + s]) zb[l * NS + s] = key;
              };
              if (small) {
                Edges32 e;
                piece_edges32(p, e);
                cover_pixel32<NS>(p, e, tile_x0 + (l & 7), tile_y0 + (l >> 3), emit);
              } else {
                Edges e;
                piece_edges(p, e);
                cover_pixel64<NS>(p, e, tile_x0 + (l & 7), tile_y0 + (l >> 3), emit);
              }
            }
          }
        }
        Sample st[64][NS];
        int st_slot[64][NS];
        for (int l = 0; l < 64; ++l)
          for (int s = 0; s < NS; ++s) { st[l][s].wsum = key_wsum(zb[l * NS + s]); st[l][s].id = key_id(zb[l * NS + s]); st_slot[l][s] = key_slot(zb[l * NS + s]); }
        // tasks + shading + resolve
        for (int l = 0; l < 64; ++l) {
          const int px = tile_x0 + (l & 7), py = tile_y0 + (l >> 3);
          if (px >= w || py >= h) continue;
          float q[NS][6];
          float acc[6] = {0, 0, 0, 0, 0, 0};
          for (int s = 0; s < NS; ++s) {
            if (st[l][s].id < 0) continue;
            int src = s;
            for (int k = s - 1; k >= 0; --k)
              if (st[l][k].id == st[l][s].id) src = k;
            if (src == s) {
              Piece pf;
              pf.flags = 2;
              if (st_slot[l][s] != SLOT_NONE) {   // shade from the binned record, as the kernel does
                unpack_tile_rec(L.list[L.tile_off[tile] + st_slot[l][s]], tile_x0, tile_y0, pf);
                pf.tri = pf.id < m.n_faces ? pf.id : pf.id - m.n_faces;
                piece_bary_from_flags(pf);
              }
              if (pf.flags & 2) piece_from_index<true>(m, T, Kv, st[l][s].id, pf);
              float c255[3], n255[3];
              shade(m, tex, lights, T, gl_eye, do_norm, pf, px, py, c255, n255);
              for (int c = 0; c < 3; ++c) { q[s][c] = (float)(unsigned)q255(c255[c]); q[s][3 + c] = (float)(unsigned)q255(n255[c]); }
            }
            for (int c = 0; c < 6; ++c) acc[c] += q[src][c];
          }
          float* o = out + (size_t)item * stride_v + (size_t)py * stride_y + (size_t)px * stride_x + (long long)r * stride_view;
          if (c_rgb >= 0) for (int c = 0; c < 3; ++c) o[c_rgb + c] = resolve_channel(acc[c], NS, false);
          if (do_norm) for (int c = 0; c < 3; ++c) o[c_normals + c] = resolve_channel(acc[3 + c], NS, false);
          if (do_depth) o[c_depth] = st[l][0].id >= 0 ? 1.0f / st[l][0].wsum : 0.f;
        }
      }
  }
}
}  // namespace

extern "C" void raster_emul_render(const float* verts, const float* normals, const float* colors, const int32_t* faces, int n_verts, int n_faces,
                                   float radius, const float* uvs, const uint32_t* texels, int tex_w, int tex_h, int tex_levels,
                                   const float* TCO, const float* K, int n_views, int h, int w, uint32_t flags, const Lights* lights, float* out,
                                   long long stride_v, int views_per_item, long long stride_view, long long stride_y, long long stride_x,
                                   int c_rgb, int c_normals, int c_depth, int cap_list, int reverse_lists) {
  MeshRef m;
  m.verts = verts; m.normals = normals; m.colors = colors; m.faces = faces; m.n_verts = n_verts; m.n_faces = n_faces; m.radius = radius;
  m.uvs = (uvs && texels) ? uvs : nullptr;
  TexRef tx;
  memset(&tx, 0, sizeof(tx));
  tx.texels = texels; tx.tex_w = tex_w; tx.tex_h = tex_h; tx.tex_levels = tex_levels;
  int off = 0;
  for (int l = 0; l < tex_levels && l < MP_TEX_MAX_LEVELS; ++l) {
    tx.tex_off[l] = off;
    off += std::max(1, tex_w >> l) * std::max(1, tex_h >> l);
  }
  std::vector<int32_t> ids(n_views, 0);
  if (cap_list <= 0) cap_list = 4 * n_faces + 2048;
  if (flags & 16u)
    render_views<4>(&m, &tx, ids.data(), TCO, K, n_views, h, w, flags, *lights, out, stride_v, views_per_item, stride_view, stride_y, stride_x,
                    c_rgb, c_normals, c_depth, cap_list, reverse_lists);
  else
    render_views<1>(&m, &tx, ids.data(), TCO, K, n_views, h, w, flags, *lights, out, stride_v, views_per_item, stride_view, stride_y, stride_x,
                    c_rgb, c_normals, c_depth, cap_list, reverse_lists);
}
