// raster.hip -- on-device triangle rasteriser for gfx950 (replaces the Panda3D/OpenGL render loop).
//
// Reference contract: Panda3dBatchRenderer.render
//   (/root/reference/src/megapose/panda3d_renderer/panda3d_batch_renderer.py:217-282, worker_loop :89-150,
//    Panda3dSceneRenderer.render_scene panda3d_scene_renderer.py:298-358, renderer configuration :71-74 (4x multisampling,
//    mip-mapped textures), camera model types.py:75-101, eye-normal LUT utils.py:58-68 + panda3d_scene_renderer.py:210-216,
//    depth utils.py:44-55, lights panda3d_scene_renderer.py:104-136).
//
// Pixel contract v2: stated at the top of oracle/raster.c (the independent CPU restatement this file must match bit for bit);
// its arithmetic lives in raster_core.h, which tests/raster_emul.cpp also compiles for the host so that the contract is checked
// against the oracle without a GPU.
//
// Structure (v4, stateless -- nothing but small index lists goes through HBM between the two kernels):
//   (1) raster_bin    one workgroup per view: every triangle is transformed, clipped against the near plane, projected and its
//                     pieces are binned by their pixel bbox into 8x8-pixel tiles (LDS counters -> prefix scan -> fill).  Pieces
//                     touching more than LARGE_TILES tiles go to a per-view "large" list that every tile walks.
//   (2) raster_tiles  one wave per (item, tile), four tiles per workgroup: for each of the item's views the wave sets up 64 listed
//                     pieces at a time lane-parallel (recomputed from the L2-resident mesh), broadcasts them with v_readlane and
//                     tests the tile's 64 pixels x 1|4 samples against each (exact integer edge functions, 32-bit when the piece
//                     is small); depth state lives in registers -- no LDS z-buffer, no atomics.  Shading runs once per (pixel,
//                     distinct winning piece) from a dense per-wave task list; the observation crop (roi_align) of the item is
//                     another role of the same wave.  All channels of the tile's pixels are staged in LDS and leave as whole,
//                     contiguous pixel records (one launch fills the CNN input with full-line writes).
// Roofline: HBM-bound on the output writes ((3+3[+1])*4*h*w bytes per view + the mesh once per view).
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "crop_device.h"
#include "raster_core.h"

namespace mp {

using rc::Piece;
using rc::Sample;
using rc::SUBPIX;
using rc::TILE;

typedef rc::MeshRef MeshDev;
typedef rc::TexRef TexDev;
typedef rc::Lights LightsDev;

#ifndef MP_RASTER_WAVES
#define MP_RASTER_WAVES 3   // waves per SIMD the tile kernel is compiled for (register budget 512 / MP_RASTER_WAVES).  Measured on one
                            // box, config-2 step (profiles/r03_raster_waves_ab.txt): 4 waves = 128 VGPRs + 284 B/lane of spills (16.7 GB written
                            // per 4.9-GB launch) 60.4 ms; 3 waves = 164 VGPRs, no spills, 55.2 ms
#endif
constexpr int BIN_THREADS = 512;
constexpr int LARGE_TILES = 16;    // a piece whose bbox touches more tiles is not replicated into tile lists
constexpr int TILE_WAVES = 4;      // tiles (waves) per workgroup of raster_tiles: a 32 x 8 pixel strip
constexpr int MAX_RUN = 32;        // channels per pixel one launch may stage
constexpr int HDR_INTS = 4;        // per-view header: n_large, n_entries, overflow, unused

// optional fused crop role: roi_align of the observation into channels c0.. of the same pixels
struct CropArgs {
  const float* images;     // [n_im][C][H][W], or [n_im][H][W][4] when nhwc4; NULL = no crop role
  const int32_t* im_ids;   // [n_items]
  const float* boxes;      // [n_items][4]
  int C, H, W, c0, nhwc4;
};

// per-view workspace, in ints (every section starts 16-byte aligned):
//   [hdr HDR_INTS][tile_off n_tiles + 1][tile_off_l n_tiles + 1][list_l: cap_large piece indices][list: cap_list TileRec records of 8 ints]
// tile_off / list: the binned (small) pieces of every tile as 32-byte records; tile_off_l / list_l: the indices of the LARGE pieces
// (too big for the 32-bit edge functions or touching > LARGE_TILES tiles) per tile they can own a sample in -- recomputed from the mesh
// by the tile kernel, but only by the tiles they touch.
struct BinLayout {
  long long view_ints;
  int n_tiles, tiles_x, tiles_y, cap_list, cap_large, max_faces;
  int off_tl, off_large, off_list;   // int offsets of tile_off_l / list_l / list inside a view's block
};

__device__ __forceinline__ void tile_range(const Piece& p, int ns, int w, int h, int& tx0, int& ty0, int& tx1, int& ty1) {
  int x0, y0, x1, y1;
  rc::piece_pixel_bbox(p, ns, w, h, x0, y0, x1, y1);
  tx0 = x0 >> 3; ty0 = y0 >> 3; tx1 = x1 >> 3; ty1 = y1 >> 3;
  if (x0 > x1 || y0 > y1) { tx1 = tx0 - 1; ty1 = ty0 - 1; }  // empty
}

// Can the piece own a sample inside tile (tx, ty)?  Conservative separating-edge test: for every edge, the sample-rectangle corner
// that maximises the edge function must not be strictly outside.  (Exact integers in fp64: |values| < 2^53.)
struct TileTest {
  double A[3], B[3], C[3];
  int omin, omax;
};
__device__ __forceinline__ TileTest tile_test_setup(const Piece& p, int ns) {
  TileTest t;
  const int a[3] = {1, 2, 0}, b[3] = {2, 0, 1};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double dx = (double)(p.X[b[i]] - p.X[a[i]]), dy = (double)(p.Y[b[i]] - p.Y[a[i]]);
    t.A[i] = -dy;
    t.B[i] = dx;
    t.C[i] = dy * (double)p.X[a[i]] - dx * (double)p.Y[a[i]];
  }
  t.omin = rc::sample_off_min(ns);
  t.omax = rc::sample_off_max(ns);
  return t;
}
__device__ __forceinline__ bool tile_touched(const TileTest& t, int tx, int ty) {
  const double x_lo = (double)(tx * TILE * SUBPIX + t.omin), x_hi = (double)((tx * TILE + TILE - 1) * SUBPIX + t.omax);
  const double y_lo = (double)(ty * TILE * SUBPIX + t.omin), y_hi = (double)((ty * TILE + TILE - 1) * SUBPIX + t.omax);
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double e = fma(t.A[i], t.A[i] >= 0.0 ? x_hi : x_lo, fma(t.B[i], t.B[i] >= 0.0 ? y_hi : y_lo, t.C[i]));
    ok = ok && (e >= 0.0);
  }
  return ok;
}

__global__ __launch_bounds__(BIN_THREADS) void raster_bin(const MeshDev* __restrict__ meshes, const int32_t* __restrict__ mesh_ids,
                                                          const float* __restrict__ TCO, const float* __restrict__ K, int h, int w, int ns,
                                                          int* __restrict__ ws, BinLayout lay, int* __restrict__ counters,
                                                          unsigned char* __restrict__ view_flags) {
  extern __shared__ int counts[];  // [2][n_tiles]: counters of the binned / the large pieces, then (in place) exclusive offsets = fill cursors
  __shared__ int partial[2][BIN_THREADS];
  __shared__ unsigned nearest;     // max over the pieces of (bits of the nearest vertex's 1/z, low bit replaced by the piece's orientation flag)
  const int view = blockIdx.x;
  const int tid = threadIdx.x;
  int* hdr = ws + (size_t)view * lay.view_ints;
  int* tile_off = hdr + HDR_INTS;
  int* tile_off_l = hdr + lay.off_tl;
  int* list_l = hdr + lay.off_large;
  rc::TileRec* list = reinterpret_cast<rc::TileRec*>(hdr + lay.off_list);
  int* counts_l = counts + lay.n_tiles;
  const MeshDev m = meshes[mesh_ids[view]];
  const float* T = TCO + (size_t)view * 16;
  const float* Kv = K + (size_t)view * 9;
  for (int i = tid; i < 2 * lay.n_tiles; i += BIN_THREADS) counts[i] = 0;
  if (tid == 0) nearest = 0u;
  if (view == 0 && tid < 4 && counters) counters[tid] = 0;   // counters of the light-job list (raster_classify runs after this kernel)
  __syncthreads();
  const bool finite = rc::view_finite(T, Kv);  // non-finite pose / intrinsics: empty lists -> zero image (panda3d_batch_renderer.py:109-135)
  const int F = finite ? m.n_faces : 0;
  // ---- phase 1: count ------------------------------------------------------------------------------------------------------
  unsigned my_nearest = 0u;
  for (int t = tid; t < F; t += BIN_THREADS) {
    int n_pieces = 1;
    for (int which = 0; which < n_pieces; ++which) {
      Piece p;
      n_pieces = rc::make_piece<false>(m, T, Kv, t, which, p);
      if (p.id < 0) continue;
      int tx0, ty0, tx1, ty1;
      tile_range(p, ns, w, h, tx0, ty0, tx1, ty1);
      if (tx0 > tx1 || ty0 > ty1) continue;
      my_nearest = max(my_nearest, (__float_as_uint(fmaxf(p.iz[0], fmaxf(p.iz[1], p.iz[2]))) & ~1u) | (unsigned)(p.flags & 1));
      int* cnt = ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > LARGE_TILES || rc::piece_extent(p) > rc::SMALL_EXTENT) ? counts_l : counts;
      const TileTest tt = tile_test_setup(p, ns);
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx)
          if (tile_touched(tt, tx, ty)) atomicAdd(&cnt[ty * lay.tiles_x + tx], 1);
    }
  }
  if (my_nearest) atomicMax(&nearest, my_nearest);
  __syncthreads();
  // ---- phase 2: exclusive scans of the two counter arrays (in place) ---------------------------------------------------------
  const int per = (lay.n_tiles + BIN_THREADS - 1) / BIN_THREADS;
  const int i0 = min(tid * per, lay.n_tiles), i1 = min(i0 + per, lay.n_tiles);
  int sum[2] = {0, 0};
  for (int i = i0; i < i1; ++i) { sum[0] += counts[i]; sum[1] += counts_l[i]; }
  partial[0][tid] = sum[0];
  partial[1][tid] = sum[1];
  __syncthreads();
  for (int d = 1; d < BIN_THREADS; d <<= 1) {  // Hillis-Steele inclusive scan over the 512 partial sums (both arrays at once)
    const int v0 = tid >= d ? partial[0][tid - d] : 0, v1 = tid >= d ? partial[1][tid - d] : 0;
    __syncthreads();
    partial[0][tid] += v0;
    partial[1][tid] += v1;
    __syncthreads();
  }
  const int total = partial[0][BIN_THREADS - 1], total_l = partial[1][BIN_THREADS - 1];
  const bool overflow = total > lay.cap_list || total_l > lay.cap_large;
  int run = partial[0][tid] - sum[0], run_l = partial[1][tid] - sum[1];
  for (int i = i0; i < i1; ++i) {
    const int c = counts[i], cl = counts_l[i];
    counts[i] = run;
    tile_off[i] = run;
    run += c;
    counts_l[i] = run_l;
    tile_off_l[i] = run_l;
    run_l += cl;
    // does this view reach the tile?  (one byte per (view, tile): what raster_classify ORs over an item's views; an overflowed view's tiles
    // all count as reached -- its tiles walk every piece)
    if (view_flags) view_flags[(size_t)view * lay.n_tiles + i] = (c + cl > 0 || overflow) ? 1 : 0;
  }
  if (tid == 0) {
    tile_off[lay.n_tiles] = total;
    tile_off_l[lay.n_tiles] = total_l;
    hdr[0] = total_l;
    hdr[1] = total;
    hdr[2] = overflow ? 1 : 0;  // the lists do not fit: raster_tiles walks ALL piece indices of this view instead (slow, correct)
    hdr[3] = (int)(nearest & 1u);   // orientation flag of the view's nearest piece: which pieces raster_tiles visits first (a hint, see there)
  }
  __syncthreads();
  if (overflow) return;
  // ---- phase 3: fill (pieces recomputed: cheaper than keeping 2F bboxes) -------------------------------------------------------
  for (int t = tid; t < F; t += BIN_THREADS) {
    int n_pieces = 1;
    for (int which = 0; which < n_pieces; ++which) {
      Piece p;
      n_pieces = rc::make_piece<false>(m, T, Kv, t, which, p);
      if (p.id < 0) continue;
      int tx0, ty0, tx1, ty1;
      tile_range(p, ns, w, h, tx0, ty0, tx1, ty1);
      if (tx0 > tx1 || ty0 > ty1) continue;
      const bool is_large = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > LARGE_TILES || rc::piece_extent(p) > rc::SMALL_EXTENT;
      const TileTest tt = tile_test_setup(p, ns);
      for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
          if (!tile_touched(tt, tx, ty)) continue;
          if (is_large) list_l[atomicAdd(&counts_l[ty * lay.tiles_x + tx], 1)] = p.id;   // piece index == depth-tie id
          else list[atomicAdd(&counts[ty * lay.tiles_x + tx], 1)] = rc::pack_tile_rec(p, tx * TILE, ty * TILE);
        }
    }
  }
}

// a TileRec as two 16-byte loads (a struct copy is split into six ushort + four dword loads by the compiler)
__device__ __forceinline__ rc::TileRec load_tile_rec(const rc::TileRec* __restrict__ p) {
  const int4 a = reinterpret_cast<const int4*>(p)[0], b = reinterpret_cast<const int4*>(p)[1];
  rc::TileRec r;
  r.rx0 = (short)(a.x & 0xFFFF); r.ry0 = (short)(a.x >> 16);
  r.rx1 = (short)(a.y & 0xFFFF); r.ry1 = (short)(a.y >> 16);
  r.rx2 = (short)(a.z & 0xFFFF); r.ry2 = (short)(a.z >> 16);
  r.pad0 = (short)(a.w & 0xFFFF);   // flags (orientation swap / clipped)
  r.pad1 = 0;
  r.iz0 = __int_as_float(b.x); r.iz1 = __int_as_float(b.y); r.iz2 = __int_as_float(b.z);
  r.id = b.w;
  return r;
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
// Intra-wave LDS hand-off: the LDS unit executes one wave's DS operations in issue order, so all that is needed for lane A's write
// to be seen by lane B's later read is that the compiler keeps the program order (the asm is a compiler barrier) and that pending DS
// results have landed.  Unlike a workgroup-scope fence this does NOT wait for vmcnt: global loads issued as prefetches for the next
// view stay in flight across it.
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

#ifdef MP_RASTER_PROF   // scripts/microbench build only (never in libmp_engine.so): per-phase shader-cycle totals of raster_tiles,
// accumulated in registers and flushed once per wave by every 32nd workgroup (so the probe does not perturb what it measures)
__device__ unsigned long long g_raster_prof[16];
#define PROF_T0                                              \
  unsigned long long prof_t = __builtin_readcyclecounter(); \
  unsigned long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   /* 9..12 = counters: visits, records, tile-views, batches; 13 / 14 = cycles of a batch's set-up / visits */
#define PROF(slot)                                                  \
  {                                                                 \
    const unsigned long long prof_n = __builtin_readcyclecounter(); \
    prof_acc[slot] += prof_n - prof_t;                              \
    prof_t = prof_n;                                                \
  }
#define PROF_FLUSH                                                                  \
  if (lane == 0 && (blockIdx.x & 31) == 0) {                                        \
    for (int k = 0; k < 16; ++k) atomicAdd(&g_raster_prof[k], prof_acc[k]);         \
  }
#define PROF_COUNT(slot, n) prof_acc[slot] += (unsigned long long)(n);
#else
#define PROF_T0
#define PROF(slot)
#define PROF_FLUSH
#define PROF_COUNT(slot, n)
#endif

// minimum of a 32-bit value over the 64 lanes (wave-uniform result): four DPP steps inside the rows of 16, the rows through SGPRs
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

// ---- coverage form 1 (every binned record): block visits (arithmetic and lane layout: raster_core.h "block-visit coverage form").
// The batch's <= 64 pieces sit one per lane; each lane turns its piece into a 48-byte BlkRec in the wave's LDS slice and tests it
// against the tile's blocks (NS = 4: four 4x4-pixel blocks; NS = 1: the tile).  Then, block by block, the wave walks the set bits of
// the "touches this block" ballot: a visit reads the piece's record with three broadcast ds_read_b128, evaluates the three edge
// functions at the lane's SAMPLE with one v_dot2_i32_i16 each, and a covered sample updates the lane's depth key in registers --
// no LDS z-buffer, no atomics, no cross-lane search; ~20 VALU per visit, ~2 visits per record.
template <int NS>
__device__ __forceinline__ void cover_batch_blocks(const Piece& p, bool active, int tile_x0, int tile_y0, int lane, int slot,
                                                   const uint32_t (&rel)[NS == 1 ? 1 : 4], unsigned ok_mask, uint4* blk,
                                                   unsigned long long (&best)[NS == 1 ? 1 : 4], int front_swap, unsigned long long* n_visits = nullptr) {
  constexpr int NB = NS == 1 ? 1 : 4;
#ifdef MP_RASTER_PROF
  const unsigned long long prof_ta = __builtin_readcyclecounter();
#endif
  rc::BlkRec mine;
  memset(&mine, 0, sizeof(mine));
  int rxmin = 0, rxmax = -1, rymin = 0, rymax = -1;
  if (active) {
    rc::Edges32 e;
    rc::piece_edges32(p, e);
    mine = rc::make_blk_rec(p, e, tile_x0, tile_y0, min(slot, rc::SLOT_NONE));
    const int ox = tile_x0 * SUBPIX, oy = tile_y0 * SUBPIX;
    rxmin = min(p.X[0], min(p.X[1], p.X[2])) - ox; rxmax = max(p.X[0], max(p.X[1], p.X[2])) - ox;
    rymin = min(p.Y[0], min(p.Y[1], p.Y[2])) - oy; rymax = max(p.Y[0], max(p.Y[1], p.Y[2])) - oy;
    uint4* d = blk + lane * 3;
    d[0] = make_uint4(mine.dxy[0], mine.dxy[1], mine.dxy[2], mine.thr_bits);
    d[1] = make_uint4((unsigned)mine.e0[0], (unsigned)mine.e0[1], (unsigned)mine.e0[2], __float_as_uint(mine.inv_area));
    d[2] = make_uint4(__float_as_uint(mine.iz[0]), __float_as_uint(mine.iz[1]), __float_as_uint(mine.iz[2]), mine.key_lo);
  }
  wave_lds_fence();
#ifdef MP_RASTER_PROF   // (probe only: the block tests are hoisted so that set-up and visits can be timed apart)
  unsigned long long touched[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) touched[k] = __ballot(active && rc::blk_touched(mine, rxmin, rxmax, rymin, rymax, NS, k));
  const unsigned long long prof_tb = __builtin_readcyclecounter();
  if (n_visits) n_visits[4] += prof_tb - prof_ta;
#endif
  // Occlusion bound (exact, hierarchical-z style).  Half of a closed mesh's pieces face away from the camera and can never win a sample
  // that a facing piece covers; the two-sided contract still requires them wherever they might.  So per block the batch's pieces are
  // visited in two phases: first the pieces whose screen orientation is the one of the view's NEAREST piece (`front_swap`, a hint from
  // the binning pass: whichever phase order is taken, the result is the same), then the others -- and a piece of the second phase is
  // visited only if it could still win a sample: if every sample of the block already holds a piece, a piece whose nearest vertex lies
  // behind the FARTHEST of those holders loses every depth comparison (its 1/z at any covered sample is a convex combination of its
  // vertices' 1/z; the bound carries a 2^-20 margin for the rounding of that combination, and ties are never culled).
  const bool phase_a = active && (int)(p.flags & 1u) == front_swap;
  const unsigned ub_hi = __float_as_uint(fmaxf(p.iz[0], fmaxf(p.iz[1], p.iz[2])) * (1.0f + 9.5367431640625e-7f));
#pragma unroll
  for (int k = 0; k < NB; ++k) {
#ifdef MP_RASTER_PROF
    const unsigned long long m_all = touched[k];
#else
    const unsigned long long m_all = __ballot(active && rc::blk_touched(mine, rxmin, rxmax, rymin, rymax, NS, k));
#endif
    if (m_all == 0ull) continue;
    const unsigned long long m_a = m_all & __ballot(phase_a);
    const bool ok = (ok_mask >> k) & 1u;
    const uint32_t rel_k = rel[k];
    unsigned long long bk = best[k];
    // visits, software-pipelined: the record of the NEXT visit is requested before the current one is evaluated (the LDS latency
    // of a visit would otherwise be exposed on every trip: the chain read -> 3 dot products -> compare -> depth is serial)
    auto visit_all = [&](unsigned long long m) {
#ifdef MP_RASTER_PROF
      if (n_visits) *n_visits += __popcll(m);
#endif
      if (!m) return;
      int j = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      uint4 a = blk[j * 3], b = blk[j * 3 + 1], c = blk[j * 3 + 2];   // wave-uniform address: broadcast reads
      while (true) {
        const bool more = m != 0ull;
        const int jn = more ? __ffsll((long long)m) - 1 : j;
        m &= m - 1ull;
        const uint4 an = blk[jn * 3], bn = blk[jn * 3 + 1], cn = blk[jn * 3 + 2];
        // rc::cover_sample_rel, branch-free (the same operations in the same order for a covered sample; an uncovered one computes
        // garbage that the final select drops): coverage is the common case of a visit, and the exec-mask branch around the depth
        // arithmetic cost more than the arithmetic
        const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.w);   // scalar: the three thresholds become SALU bit tests
        const int E0 = rc::dot2_i16(a.x, rel_k, (int)b.x), E1 = rc::dot2_i16(a.y, rel_k, (int)b.y), E2 = rc::dot2_i16(a.z, rel_k, (int)b.z);
        bool in = ok & (E0 >= (int)(thr & 1u)) & (E1 >= (int)((thr >> 1) & 1u)) & (E2 >= (int)((thr >> 2) & 1u));
        const float ia = __uint_as_float(b.w);
        const float b0 = (float)E0 * ia, b1 = (float)E1 * ia, b2 = (float)E2 * ia;
        const float wsum = fmaf(b2, __uint_as_float(c.z), fmaf(b1, __uint_as_float(c.y), b0 * __uint_as_float(c.x)));
        in = in & rc::depth_in_range(wsum);
        const unsigned long long key = rc::depth_key_lo(wsum, c.w);
        bk = (in & (key > bk)) ? key : bk;
        if (!more) break;
        a = an; b = bn; c = cn;
      }
    };
    visit_all(m_a);
    unsigned long long m_b = m_all & ~m_a;
    if (m_b != 0ull && __ballot(bk != 0ull || !ok) == ~0ull) {   // every sample of the block (inside the image) holds a piece
      const unsigned far_hi = wave_min_u32(ok ? (unsigned)(bk >> 32) : 0xFFFFFFFFu);   // 1/z bits of the farthest holder
      m_b &= __ballot(ub_hi >= far_hi);
    }
    visit_all(m_b);
    best[k] = bk;
  }
  wave_lds_fence();   // the next batch rewrites the records
#ifdef MP_RASTER_PROF
  if (n_visits) n_visits[5] += __builtin_readcyclecounter() - prof_tb;
#endif
}

// ---- coverage form 2 (the view's "large" list and the overflow fallback only): wave-per-piece sweep with the 64-bit edge functions.
// The piece is wave-uniform, lane l tests ITS pixel and updates its own z-buffer slots (no conflicts). ----------------------------
template <int NS>
__device__ __noinline__ void sweep_piece(int X0, int Y0, int X1, int Y1, int X2, int Y2, float iz0, float iz1, float iz2, int id, int tile_x0,
                                         int tile_y0, int px, int py, int lane, unsigned long long* zb) {
  // (scalar arguments travel in registers; a `const Piece&` would be built in scratch memory before every call)
  Piece p;
  p.X[0] = X0; p.Y[0] = Y0; p.X[1] = X1; p.Y[1] = Y1; p.X[2] = X2; p.Y[2] = Y2;
  p.iz[0] = iz0; p.iz[1] = iz1; p.iz[2] = iz2;
  p.id = id;
  const int Xmin = min(p.X[0], min(p.X[1], p.X[2])), Xmax = max(p.X[0], max(p.X[1], p.X[2]));
  const int Ymin = min(p.Y[0], min(p.Y[1], p.Y[2])), Ymax = max(p.Y[0], max(p.Y[1], p.Y[2]));
  const int sx0 = tile_x0 * SUBPIX, sy0 = tile_y0 * SUBPIX;
  if (Xmax < sx0 || Xmin >= sx0 + TILE * SUBPIX || Ymax < sy0 || Ymin >= sy0 + TILE * SUBPIX) return;
  rc::Edges e;
  rc::piece_edges(p, e);
  rc::cover_pixel64<NS>(p, e, px, py, [&](int s, float wsum) {
    const unsigned long long key = rc::depth_key(wsum, p.id, rc::SLOT_NONE);
    unsigned long long* slot = zb + lane * NS + s;
    if (key > *slot) *slot = key;
  });
}

// roi_align of one output pixel; one instance for the four (C, layout) cases (code size)
template <bool INLINE>
__device__ __forceinline__ float4 crop_lane_body(const CropArgs& crop, int item, int h, int w, int px, int py);
__device__ __noinline__ float4 crop_lane(CropArgs crop, int item, int h, int w, int px, int py) { return crop_lane_body<false>(crop, item, h, w, px, py); }
template <bool INLINE>
__device__ __forceinline__ float4 crop_lane_body(const CropArgs& crop, int item, int h, int w, int px, int py) {
  float cvals[4];   // (arguments and result by value = in registers: references would be built in scratch memory before the call)
  const float* bx = crop.boxes + (size_t)item * 4;
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  const float roi_w = fmaxf(x2 - x1, 1.0f), roi_h = fmaxf(y2 - y1, 1.0f);
  const float bin_h = roi_h / (float)h, bin_w = roi_w / (float)w;
  const float* img = crop.images + (size_t)crop.im_ids[item] * (crop.nhwc4 ? 4 : crop.C) * crop.H * crop.W;
  cvals[3] = 0.f;
  if (crop.nhwc4) {
    if (crop.C == 4) crop_pixel<4, true>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, cvals);
    else crop_pixel<3, true>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, cvals);
  } else {
    if (crop.C == 4) crop_pixel<4, false>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, cvals);
    else crop_pixel<3, false>(img, crop.H, crop.W, x1, y1, bin_w, bin_h, px, py, cvals);
  }
  return make_float4(cvals[0], cvals[1], cvals[2], cvals[3]);
}

// per-wave LDS bytes in front of the channel staging area: z-buffer keys [64 * NS] u64 + shading tasks [64 * NS] u32, and at least the
// 64 x 48-byte piece records of the block-visit coverage form (which alias them)
__host__ __device__ constexpr size_t tiles_zt_bytes(int ns) {
  return (size_t)64 * ns * (sizeof(unsigned long long) + sizeof(unsigned)) > 64 * sizeof(rc::BlkRec)
             ? (size_t)64 * ns * (sizeof(unsigned long long) + sizeof(unsigned)) : 64 * sizeof(rc::BlkRec);
}

constexpr size_t HDR_LDS_BYTES = 128;   // per wave: the list headers of an item's first four views (6 ints each)
struct ViewHdr {   // what a wave needs to know about one view's lists for its tile (wave-uniform)
  int begin, n_list, begin_l, n_large, overflow;
  int front_swap;   // screen orientation (Piece.flags & 1) of the view's nearest piece: a hint for the order of the two visit phases
};

// OUT = OUT_F16 (MP_RASTER_F16, the "fp16 renders" mode of BASELINE.json configs[4]): `out` holds IEEE binary16 elements -- same element
// strides, every written channel (renders and the fused crop) is rounded to nearest-even on its way out; nothing else changes.
// OUT = OUT_XREC (MP_RASTER_XREC): `out` holds the bf16 pixel RECORDS the exact-piece stem convolution consumes (conv_stem.hip):
// [x1,x2,x3 of every crop channel | the 8-bit integer k of every render channel | zero padding], stride_x = record length; the
// record is staged in LDS in that form and leaves as 16-byte chunks.  Channel numbers (c_rgb, c_normals, stride_view, crop.c0) stay
// logical channel numbers; stride_v / stride_y / stride_x count bf16 elements.
constexpr int OUT_F32 = 0, OUT_F16 = 1, OUT_XREC = 2;
// DEPTHREC (OUT_XREC only): the record's fp32-kind channels are a general mask and depth channels are normalised here (RGBD models); false =
// the fp32-kind channels are the crop's leading ones, no depth channel (the RGB models: the round-4 form, free of the extra arithmetic --
// this kernel sits exactly at its register budget)
template <int NS, int OUT = OUT_F32, bool FULL = true, bool DEPTHREC = false>
__global__ __launch_bounds__(64 * TILE_WAVES) __attribute__((amdgpu_waves_per_eu(MP_RASTER_WAVES, MP_RASTER_WAVES))) void raster_tiles(
    const MeshDev* __restrict__ meshes, const TexDev* __restrict__ texs, const int32_t* __restrict__ mesh_ids, const float* __restrict__ TCO,
    const float* __restrict__ K, const int* __restrict__ ws, BinLayout lay, int h, int w, uint32_t flags, LightsDev lights,
    float* __restrict__ out, long long stride_v, int views_per_item, int n_items, long long stride_view, long long stride_y,
    long long stride_x, int c_rgb, int c_normals, int c_depth, int c_lo, int run, uint32_t run_mask, CropArgs crop, uint32_t xrec_mask,
    const float* __restrict__ depth_tcr, int depth_mode, const unsigned char* __restrict__ job_flags) {
  // LDS per wave: zb [64 * NS] u64 (z-buffer, later the shading results) | tasks [64 * NS] u32 | stage [64][run] floats
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // (a readfirstlane'd wave index makes the LDS bases scalar -- and the kernel 4 % slower, measured)
  constexpr int NB = NS == 1 ? 1 : 4;
  constexpr size_t ZT_BYTES = tiles_zt_bytes(NS);
  const size_t per_wave = ZT_BYTES + (((size_t)64 * run * sizeof(float) + 15) & ~(size_t)15) + HDR_LDS_BYTES;
  unsigned char* mine = lds_raw + (size_t)wave * per_wave;
  unsigned long long* zb = (unsigned long long*)mine;
  uint2* res = (uint2*)mine;                                    // aliases zb once the samples are in registers
  uint4* blk = (uint4*)mine;                                    // aliases zb + tasks while the binned records are being covered
  unsigned* tasks = (unsigned*)(mine + (size_t)64 * NS * sizeof(unsigned long long));
  float* stage = (float*)(mine + ZT_BYTES);
  float* my_stage = stage + (size_t)lane * run;
  // OUT_XREC: the staging area holds the pixel records themselves, [64][stride_x] bf16 (never larger than [64][run] floats: the
  // launcher checks); `put` files a channel value into the lane's pixel in whichever form the launch stages
  // (OUT_XREC) logical channel c is fp32-kind (three record slots) iff bit c of xrec_mask is set -- the crop's channels and, for RGBD
  // models, every depth channel --; the fp32-kind channels come first in the record, in channel order, then the integer channels
  const int xrec_nf = DEPTHREC ? __builtin_popcount(xrec_mask) : crop.C;
  unsigned short* my_rec = reinterpret_cast<unsigned short*>(stage) + (size_t)lane * (int)stride_x;
  auto put = [&](int ch, float v) {   // ch = logical channel number
    if constexpr (OUT == OUT_XREC) {
      const int f_below = DEPTHREC ? __builtin_popcount(xrec_mask & ((1u << ch) - 1u)) : min(ch, xrec_nf);   // fp32-kind channels in front of ch (ch < 32)
      if (DEPTHREC ? (bool)((xrec_mask >> ch) & 1u) : ch < xrec_nf) {   // exact truncation split x = x1 + x2 + x3 (three bf16 pieces)
        const unsigned b1 = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(b1);
        const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(b2);
        my_rec[3 * f_below] = (unsigned short)(b1 >> 16); my_rec[3 * f_below + 1] = (unsigned short)(b2 >> 16);
        my_rec[3 * f_below + 2] = (unsigned short)(__float_as_uint(r2) >> 16);
      } else {              // an integer 0..255: one bf16, exactly
        my_rec[3 * xrec_nf + (ch - f_below)] = (unsigned short)(__float_as_uint(v) >> 16);
      }
    } else {
      my_stage[ch - c_lo] = v;
    }
  };
  if constexpr (OUT == OUT_XREC) {   // zero the whole record once (the padding slots are never written again)
    uint4* z = reinterpret_cast<uint4*>(my_rec);
    for (int q = 0; q < (int)stride_x / 8; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int groups_x = (lay.tiles_x + TILE_WAVES - 1) / TILE_WAVES;
  int b = blockIdx.x;
  const int gx = b % groups_x; b /= groups_x;
  const int ty = b % lay.tiles_y;
  const int item = b / lay.tiles_y;
  const int tx = gx * TILE_WAVES + wave;
  if (tx >= lay.tiles_x) return;   // (no workgroup-level barrier below: waves are independent)
  const int tile_x0 = tx * TILE, tile_y0 = ty * TILE;
  const int tile = ty * lay.tiles_x + tx;
  // Compacted launch form (round 5): raster_classify marked the (item, tile) pairs that no view reaches -- 65 % of a pose-pipeline launch --
  // and raster_tiles_light writes those (background + crop).  A wave of such a pair leaves here, after one byte load, instead of waiting
  // for four list headers and running the crop under this kernel's register budget.  Which kernel writes a tile does not enter the
  // arithmetic: the pixels are the same bit for bit (test_raster_compacted_launch_equals_the_direct_form).
  if (job_flags && job_flags[(size_t)item * lay.n_tiles + tile] == 0) return;
  const int px = tile_x0 + (lane & 7), py = tile_y0 + (lane >> 3);
  const bool do_norm = (flags & MP_RASTER_NORMALS) && c_normals >= 0;
  const bool do_depth = (flags & MP_RASTER_DEPTH) && c_depth >= 0;
  const bool gl_eye = flags & MP_RASTER_NORMALS_GL;
  const bool need_shade = c_rgb >= 0 || do_norm;   // a depth-only render (the depth refiner's) has nothing to shade
  // (OUT_XREC) depth channels enter the record NORMALISED, with the operations of normalize_depth_kernel (crop.hip; reference
  // models/pose_rigid.py:466-496) in the same order, so that the three pieces add up to the fp32 tensor path's value bit for bit
  float depth_zr = 1.f;
  if constexpr (OUT == OUT_XREC && DEPTHREC) {
    if (depth_mode != 0 && depth_tcr) depth_zr = depth_tcr[3 * (size_t)item + 2];
  }
  auto nd = [&](float d) {
    if constexpr (OUT == OUT_XREC && DEPTHREC) {
      if (depth_mode == 1) d = d / depth_zr;
      else if (depth_mode == 2) d = fminf(fmaxf(d / depth_zr, 0.f), 2.f) - 1.f;
      else if (depth_mode == 3) d = fminf(fmaxf(d - depth_zr, -2.f), 2.f);
    }
    return d;
  };
  // block-visit coverage: the lane's sample position in each block (relative to the tile) and whether its pixel is inside the image
  uint32_t rel[NB];
  unsigned ok_mask = 0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    rel[k] = rc::blk_lane_rel(NS, k, lane);
    const int pix = rc::blk_lane_pixel(NS, k, lane);
    if (tile_x0 + (pix & 7) < w && tile_y0 + (pix >> 3) < h) ok_mask |= 1u << k;
  }
  PROF_T0

  // The list headers of the item's first FAST_VIEWS views are fetched at once, one view per lane (one memory round trip instead of one
  // per view: most tiles of a crop are empty in every view, and for those the header is all there is to wait for), and parked in a
  // 80-byte slot of the wave's LDS slice; a view reads its five ints back with broadcast loads + readfirstlane.  (They must not live in
  // per-lane registers across the view loop: a value that is read back with v_readlane is invisible to the register allocator's
  // liveness -- a spill under a partial exec mask inside the loop would lose the lanes that are inactive there.  As an array of scalars
  // the compiler put them in SCRATCH: 80 B per lane, 5 KB of dead stores per wave.)  Further views (never in the pose pipeline: 1 or 4 views per item) read their header when
  // they are reached.  The first 64 records of view r + 1 are requested before view r is processed.
  constexpr int FAST_VIEWS = 4;
  const int view0 = item * views_per_item;
  auto fetch_hdr = [&](int view) {   // scalar loads (uniform address)
    const int* hdr = ws + (size_t)view * lay.view_ints;
    ViewHdr v;
    v.overflow = hdr[2];
    v.front_swap = hdr[3];
    v.begin = 0; v.begin_l = 0; v.n_large = 0;
    if (v.overflow) {   // rare: the view's lists did not fit -> its tiles walk all 2F piece indices through the recompute path
      v.n_list = 2 * meshes[mesh_ids[view]].n_faces;
    } else {
      v.begin = hdr[HDR_INTS + tile];
      v.n_list = hdr[HDR_INTS + tile + 1] - v.begin;
      v.begin_l = hdr[lay.off_tl + tile];
      v.n_large = hdr[lay.off_tl + tile + 1] - v.begin_l;
    }
    return v;
  };
  int* hdr_lds = (int*)(mine + per_wave - HDR_LDS_BYTES);   // [FAST_VIEWS][6] ints of this wave (behind its staging area)
  {
    if (lane < min(views_per_item, FAST_VIEWS)) {
      int h_begin = 0, h_nlist = 0, h_begin_l = 0, h_nlarge = 0, h_over = 0;
      const int* hdr = ws + (size_t)(view0 + lane) * lay.view_ints;
      h_over = hdr[2];
      const int h_front = hdr[3];
      if (h_over) {
        h_nlist = 2 * meshes[mesh_ids[view0 + lane]].n_faces;
      } else {
        h_begin = hdr[HDR_INTS + tile];
        h_nlist = hdr[HDR_INTS + tile + 1] - h_begin;
        h_begin_l = hdr[lay.off_tl + tile];
        h_nlarge = hdr[lay.off_tl + tile + 1] - h_begin_l;
      }
      int* d = hdr_lds + lane * 6;
      d[0] = h_begin; d[1] = h_nlist; d[2] = h_begin_l; d[3] = h_nlarge; d[4] = h_over; d[5] = h_front;
    }
    wave_lds_fence();
  }
  auto hdr_of = [&](int r) {   // r is wave-uniform: broadcast LDS reads, straight into scalar registers
    if (r >= FAST_VIEWS) return fetch_hdr(view0 + r);
    const int* d = hdr_lds + r * 6;
    ViewHdr v;
    v.begin = __builtin_amdgcn_readfirstlane(d[0]); v.n_list = __builtin_amdgcn_readfirstlane(d[1]);
    v.begin_l = __builtin_amdgcn_readfirstlane(d[2]); v.n_large = __builtin_amdgcn_readfirstlane(d[3]);
    v.overflow = __builtin_amdgcn_readfirstlane(d[4]); v.front_swap = __builtin_amdgcn_readfirstlane(d[5]);
    return v;
  };
  ViewHdr vh_next = hdr_of(0);
  rc::TileRec rec_nxt;
  rec_nxt.id = -1;
  {
    const rc::TileRec* list0 = reinterpret_cast<const rc::TileRec*>(ws + (size_t)view0 * lay.view_ints + lay.off_list);
    if (!vh_next.overflow && lane < vh_next.n_list) rec_nxt = load_tile_rec(list0 + vh_next.begin + lane);
  }
  for (int r = 0; r < views_per_item; ++r) {
    const int view = view0 + r;
    const float* T = TCO + (size_t)view * 16;
    const float* Kv = K + (size_t)view * 9;
    const int* vhdr = ws + (size_t)view * lay.view_ints;
    const int* large = vhdr + lay.off_large;
    const rc::TileRec* list = reinterpret_cast<const rc::TileRec*>(vhdr + lay.off_list);
    const ViewHdr vh = vh_next;
    const rc::TileRec rec_first = rec_nxt;
    if (r + 1 < views_per_item) {   // prefetch: first records of the next view
      vh_next = hdr_of(r + 1);
      const rc::TileRec* list_n = reinterpret_cast<const rc::TileRec*>(ws + (size_t)(view + 1) * lay.view_ints + lay.off_list);
      rec_nxt.id = -1;
      if (!vh_next.overflow && lane < vh_next.n_list) rec_nxt = load_tile_rec(list_n + vh_next.begin + lane);
    }
    const int n_total = vh.n_list + vh.n_large;
    const long long cv = (long long)r * stride_view;
    if (n_total == 0) {   // nothing of this view reaches the tile (most tiles of a crop): background
      if constexpr (OUT != OUT_XREC) {   // (a stem record was cleared as a whole when the wave started)
        if (c_rgb >= 0) { put(c_rgb + (int)cv, 0.f); put(c_rgb + (int)cv + 1, 0.f); put(c_rgb + (int)cv + 2, 0.f); }
        if (do_norm) { put(c_normals + (int)cv, 0.f); put(c_normals + (int)cv + 1, 0.f); put(c_normals + (int)cv + 2, 0.f); }
        if (do_depth) put(c_depth + (int)cv, 0.f);
      } else {
        if (do_depth) put(c_depth + (int)cv, nd(0.f));   // background depth 0 is normalised like any other value (e.g. to -1)
      }
      PROF(0)
      continue;
    }
    const int mesh_id = mesh_ids[view];
    const MeshDev m = meshes[mesh_id];
    PROF(0)
    PROF_COUNT(11, 1)
    // ---- coverage + depth, binned records: block visits, the depth keys of the lane's samples in registers ---------------------
    unsigned long long best[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) best[k] = 0ull;
    bool any_sweep = vh.overflow != 0;   // (wave-uniform) is there a piece left for the sweep form?
    if (!vh.overflow) {
      // batches of <= 64 pieces, one per lane: first the tile's binned records, then the view's "large" list.  A piece of that list
      // touches too many tiles to be replicated into their lists, but if it still fits the 32-bit edge functions (extent <=
      // SMALL_EXTENT = 81 pixels: the cap fans of a lathe mesh, long slivers) it takes the same block visits -- re-derived from the mesh
      // (no record: slot SLOT_NONE), with arithmetic that is bit for bit the one of the sweep form.
      const int nb_list = (vh.n_list + 63) >> 6, nb_all = nb_list + ((vh.n_large + 63) >> 6);
      for (int b = 0; b < nb_all; ++b) {
        const bool from_list = b < nb_list;
        const int e = ((from_list ? b : b - nb_list) << 6) + lane;
        Piece mine_p;
        mine_p.id = -1;
        if (from_list) {
          if (e < vh.n_list) {     // two coalesced 16-byte loads, nothing to recompute
            const rc::TileRec rec = b == 0 ? rec_first : load_tile_rec(list + vh.begin + e);
            rc::unpack_tile_rec(rec, tile_x0, tile_y0, mine_p);
          }
        } else if (e < vh.n_large) {
          rc::piece_from_index<true>(m, T, Kv, large[vh.begin_l + e], mine_p);
        }
#ifdef MP_RASTER_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (probe only: the wait for the records counts as list fetch, not as coverage)
#endif
        PROF(1)
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
        if (mine_p.id >= 0) {
          rc::piece_pixel_bbox(mine_p, NS, w, h, x0, y0, x1, y1);
          x0 = max(x0, tile_x0); y0 = max(y0, tile_y0); x1 = min(x1, tile_x0 + TILE - 1); y1 = min(y1, tile_y0 + TILE - 1);
        }
        const bool touches = mine_p.id >= 0 && x0 <= x1 && y0 <= y1;
        const bool hit = touches && (from_list || rc::piece_extent(mine_p) <= rc::SMALL_EXTENT);
        any_sweep = any_sweep || __ballot(touches && !hit) != 0ull;
        const int slot = from_list ? e : rc::SLOT_NONE;
#ifdef MP_RASTER_PROF
        if (__ballot(hit) != 0ull) cover_batch_blocks<NS>(mine_p, hit, tile_x0, tile_y0, lane, slot, rel, ok_mask, blk, best, vh.front_swap, &prof_acc[9]);
        PROF_COUNT(10, __popcll(__ballot(hit)))
        PROF_COUNT(12, 1)
#else
        if (__ballot(hit) != 0ull) cover_batch_blocks<NS>(mine_p, hit, tile_x0, tile_y0, lane, slot, rel, ok_mask, blk, best, vh.front_swap);
#endif
        PROF(2)
      }
    }
    // the keys move to the wave's LDS z-buffer in pixel-major order (what the sweep form, the task builder and the resolve index)
#pragma unroll
    for (int k = 0; k < NB; ++k) zb[rc::blk_lane_pixel(NS, k, lane) * NS + rc::blk_lane_sample(NS, lane)] = best[k];
    wave_lds_fence();
    // ---- coverage + depth, what is left of the "large" list (pieces beyond the 32-bit edge functions) / the overflow fallback:
    // recomputed from the mesh, wave-per-piece sweep with the 64-bit edge functions -------------------------------------------------
    const int n_idx = !any_sweep ? 0 : vh.overflow ? vh.n_list : vh.n_large;
    for (int base = 0; base < n_idx; base += 64) {
      const int e = base + lane;
      Piece mine_p;
      mine_p.id = -1;
      if (e < n_idx) rc::piece_from_index<true>(m, T, Kv, vh.overflow ? e : large[vh.begin_l + e], mine_p);
      int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
      if (mine_p.id >= 0) {
        rc::piece_pixel_bbox(mine_p, NS, w, h, x0, y0, x1, y1);
        x0 = max(x0, tile_x0); y0 = max(y0, tile_y0); x1 = min(x1, tile_x0 + TILE - 1); y1 = min(y1, tile_y0 + TILE - 1);
      }
      unsigned long long big = __ballot(mine_p.id >= 0 && x0 <= x1 && y0 <= y1 && (vh.overflow || rc::piece_extent(mine_p) > rc::SMALL_EXTENT));
      while (big) {
        const int j = __ffsll((long long)big) - 1;
        big &= big - 1ull;
        sweep_piece<NS>(rl(mine_p.X[0], j), rl(mine_p.Y[0], j), rl(mine_p.X[1], j), rl(mine_p.Y[1], j), rl(mine_p.X[2], j), rl(mine_p.Y[2], j),
                        rlf(mine_p.iz[0], j), rlf(mine_p.iz[1], j), rlf(mine_p.iz[2], j), rl(mine_p.id, j), tile_x0, tile_y0, px, py, lane, zb);
      }
      wave_lds_fence();
      PROF(3)
    }
    Sample st[NS];
    int st_slot[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const unsigned long long key = zb[lane * NS + s];
      st[s].wsum = rc::key_wsum(key);
      st[s].id = rc::key_id(key);
      st_slot[s] = rc::key_slot(key);
    }
    wave_lds_fence();   // zb is reused for the shading results below
    // ---- shading tasks: one per (pixel, distinct winning piece), ordered by (sample, lane) -------------------------------------
    int n_tasks = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bool nw = need_shade && st[s].id >= 0;
#pragma unroll
      for (int k = 0; k < s; ++k) nw = nw && !(st[k].id == st[s].id);
      const unsigned long long mask = __ballot(nw);
      // task word: lane | sample << 6 | (bit 31 set: record slot << 8) or (bit 31 clear: piece id << 8, for pieces without a record)
      if (nw) tasks[n_tasks + __popcll(mask & ((1ull << lane) - 1ull))] =
          (unsigned)lane | ((unsigned)s << 6) | (st_slot[s] != rc::SLOT_NONE ? (0x80000000u | ((unsigned)st_slot[s] << 8)) : ((unsigned)st[s].id << 8));
      n_tasks += __popcll(mask);
    }
    wave_lds_fence();
    PROF(4)
    const TexDev* tex = (FULL && m.uvs) ? &texs[mesh_id] : nullptr;
    for (int k0 = 0; k0 < n_tasks; k0 += 64) {
      const int k = k0 + lane;
      if (k < n_tasks) {
        const unsigned tk = tasks[k];
        const int tl = tk & 63, ts = (tk >> 6) & 3;
        int id = (int)((tk >> 8) & 0x7FFFFFu);
        Piece pf;
        pf.flags = 2;
        if (tk & 0x80000000u) {   // the winner is a binned record: re-read it (L2-hot) instead of re-deriving the piece
          rc::unpack_tile_rec(load_tile_rec(list + vh.begin + (int)((tk >> 8) & 511u)), tile_x0, tile_y0, pf);
          id = pf.id;
          pf.tri = id < m.n_faces ? id : id - m.n_faces;
          rc::piece_bary_from_flags(pf);
        }
        if (pf.flags & 2) rc::piece_from_index<true>(m, T, Kv, id, pf);   // clipped pieces (their bary rows are not in the record), large-list pieces
        float c255[3], n255[3];
        rc::shade<FULL>(m, tex, lights, T, gl_eye, do_norm, pf, tile_x0 + (tl & 7), tile_y0 + (tl >> 3), c255, n255);
        uint2 q;
        q.x = (unsigned)rc::q255(c255[0]) | ((unsigned)rc::q255(c255[1]) << 8) | ((unsigned)rc::q255(c255[2]) << 16);
        q.y = (unsigned)rc::q255(n255[0]) | ((unsigned)rc::q255(n255[1]) << 8) | ((unsigned)rc::q255(n255[2]) << 16);
        res[tl * NS + ts] = q;
      }
    }
    wave_lds_fence();
    PROF(5)
    // ---- resolve: mean of the samples' 8-bit values, background samples = 0 ----------------------------------------------------
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (st[s].id < 0 || !need_shade) continue;
      int src = s;
#pragma unroll
      for (int k = s - 1; k >= 0; --k)
        if (st[k].id == st[s].id) src = k;   // ends at the first sample holding this piece (the one that was shaded)
      const uint2 q = res[lane * NS + src];
      acc[0] += (float)(q.x & 255u); acc[1] += (float)((q.x >> 8) & 255u); acc[2] += (float)((q.x >> 16) & 255u);
      acc[3] += (float)(q.y & 255u); acc[4] += (float)((q.y >> 8) & 255u); acc[5] += (float)((q.y >> 16) & 255u);
    }
    if (c_rgb >= 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) put(c_rgb + (int)cv + c, OUT == OUT_XREC ? rc::resolve_k(acc[c], NS) : rc::resolve_channel(acc[c], NS, false));
    }
    if (do_norm) {
#pragma unroll
      for (int c = 0; c < 3; ++c) put(c_normals + (int)cv + c, OUT == OUT_XREC ? rc::resolve_k(acc[3 + c], NS) : rc::resolve_channel(acc[3 + c], NS, false));
    }
    if (do_depth) put(c_depth + (int)cv, nd(st[0].id >= 0 ? 1.0f / st[0].wsum : 0.f));
    wave_lds_fence();  // the z-buffer / task arrays are reused by the next view
    PROF(6)
  }
  if (crop.images && px < w && py < h) {  // crop role: roi_align of the item's observation for this lane's pixel
    const float4 cv4 = crop_lane(crop, item, h, w, px, py);
    put(crop.c0, cv4.x); put(crop.c0 + 1, cv4.y); put(crop.c0 + 2, cv4.z);
    if (crop.C == 4) put(crop.c0 + 3, nd(cv4.w));   // (the observation's depth channel: normalised like the rendered ones)
  }
  wave_lds_fence();
  PROF(7)
  // ---- store: each of the tile's 8 rows leaves as one contiguous run of 8 pixels x `run` channels (only written channels) -------
  const int cols = min(TILE, w - tile_x0), rows = min(TILE, h - tile_y0);
  const int per_row = cols * run;   // <= 256 floats
  if constexpr (OUT == OUT_XREC) {
    // the tile's records sit in LDS as [8 rows][8 pixels][stride_x bf16]; a tile row = cols * stride_x / 8 contiguous 16-byte chunks
    const int rowlen = cols * ((int)stride_x / 8);   // <= 40 chunks
    unsigned short* out_item = reinterpret_cast<unsigned short*>(out) + (size_t)item * stride_v;
    const uint4* recs = reinterpret_cast<const uint4*>(stage);
    if (lane < rowlen)
      for (int row = 0; row < rows; ++row)
        *reinterpret_cast<uint4*>(out_item + (size_t)(tile_y0 + row) * stride_y + (size_t)tile_x0 * stride_x + (size_t)lane * 8) =
            recs[row * 8 * ((int)stride_x / 8) + lane];
  } else if constexpr (OUT == OUT_F16) {
    _Float16* out_item = reinterpret_cast<_Float16*>(out) + (size_t)item * stride_v + c_lo;
    for (int i = lane; i < per_row; i += 64) {
      const int x = i / run, c = i - x * run;
      if (!((run_mask >> c) & 1u)) continue;
      _Float16* o = out_item + (size_t)tile_y0 * stride_y + (size_t)(tile_x0 + x) * stride_x + c;
      const float* sp = stage + (size_t)x * run + c;
      for (int row = 0; row < rows; ++row) o[(size_t)row * stride_y] = (_Float16)sp[(size_t)row * 8 * run];   // v_cvt_f16_f32: round to nearest even
    }
  } else {
  float* out_item = out + (size_t)item * stride_v + c_lo;
  for (int i = lane; i < per_row; i += 64) {
    const int x = i / run, c = i - x * run;   // once per lane and 64-float slice, reused for all 8 rows
    if (!((run_mask >> c) & 1u)) continue;
    float* o = out_item + (size_t)tile_y0 * stride_y + (size_t)(tile_x0 + x) * stride_x + c;
    const float* sp = stage + (size_t)x * run + c;
    for (int row = 0; row < rows; ++row) o[(size_t)row * stride_y] = sp[(size_t)row * 8 * run];
  }
  }
  PROF(8)
  PROF_FLUSH
}

// ---- compacted launch form: classification of the (item, tile) pairs and the kernel for the pairs no view reaches --------------------------
// one thread per pair: heavy iff some view of the item has a piece (binned or large) in the tile, or a view's lists overflowed;
// flags[pair] = 1 | 0, the light pairs are appended to a list (order arbitrary: every job is independent)
__global__ __launch_bounds__(256) void raster_classify(const unsigned char* __restrict__ view_flags, BinLayout lay, int views_per_item, int n_items,
                                                       unsigned char* __restrict__ flags, int* __restrict__ counters, int* __restrict__ light_list) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n_items * lay.n_tiles;
  const bool valid = idx < total;
  bool heavy = false;
  if (valid) {
    const int item = (int)(idx / lay.n_tiles), tile = (int)(idx - (long)item * lay.n_tiles);
    for (int r = 0; r < views_per_item; ++r) heavy = heavy || view_flags[(size_t)(item * views_per_item + r) * lay.n_tiles + tile] != 0;
    flags[idx] = heavy ? 1 : 0;
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long ml = __ballot(valid && !heavy);
  int base = 0;
  if (lane == 0 && ml) base = atomicAdd(counters, __popcll(ml));
  base = __builtin_amdgcn_readfirstlane(base);
  if (valid && !heavy) light_list[base + __popcll(ml & ((1ull << lane) - 1ull))] = (int)idx;
}

// A light job: the tile of an item that no view reaches = background in every view channel + the observation crop, stored exactly as
// raster_tiles stores a tile (same staging layout, same `put` forms).  One wave per job, strided over the list; no list headers, no mesh,
// four waves per SIMD (the unrolled roi_align taps want 127 registers; the 80 of six waves spill): one more than raster_tiles, and no list headers to wait for.
template <int OUT, bool DEPTHREC>
__global__ __launch_bounds__(64 * TILE_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void raster_tiles_light(const int* __restrict__ counters, const int* __restrict__ light_list, BinLayout lay,
                                                                     int h, int w, uint32_t flags, float* __restrict__ out, long long stride_v,
                                                                     int views_per_item, long long stride_view, long long stride_y, long long stride_x,
                                                                     int c_rgb, int c_normals, int c_depth, int c_lo, int run, uint32_t run_mask,
                                                                     CropArgs crop, uint32_t xrec_mask, const float* __restrict__ depth_tcr, int depth_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wave = ((size_t)64 * run * sizeof(float) + 15) & ~(size_t)15;
  float* stage = (float*)(lds_raw + (size_t)wave * per_wave);
  float* my_stage = stage + (size_t)lane * run;
  unsigned short* my_rec = reinterpret_cast<unsigned short*>(stage) + (size_t)lane * (int)stride_x;
  const int xrec_nf = DEPTHREC ? __builtin_popcount(xrec_mask) : crop.C;
  const bool do_norm = (flags & MP_RASTER_NORMALS) && c_normals >= 0;
  const bool do_depth = (flags & MP_RASTER_DEPTH) && c_depth >= 0;
  auto put = [&](int ch, float v) {   // (raster_tiles' `put`, verbatim)
    if constexpr (OUT == OUT_XREC) {
      const int f_below = DEPTHREC ? __builtin_popcount(xrec_mask & ((1u << ch) - 1u)) : min(ch, xrec_nf);
      if (DEPTHREC ? (bool)((xrec_mask >> ch) & 1u) : ch < xrec_nf) {
        const unsigned b1 = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(b1);
        const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(b2);
        my_rec[3 * f_below] = (unsigned short)(b1 >> 16); my_rec[3 * f_below + 1] = (unsigned short)(b2 >> 16);
        my_rec[3 * f_below + 2] = (unsigned short)(__float_as_uint(r2) >> 16);
      } else {
        my_rec[3 * xrec_nf + (ch - f_below)] = (unsigned short)(__float_as_uint(v) >> 16);
      }
    } else {
      my_stage[ch - c_lo] = v;
    }
  };
  const int n_jobs = counters[0];
  const int stride = gridDim.x * TILE_WAVES;
  for (int j = blockIdx.x * TILE_WAVES + wave; j < n_jobs; j += stride) {
    const int idx = light_list[j];
    const int item = idx / lay.n_tiles, tile = idx - item * lay.n_tiles;
    const int tx = tile % lay.tiles_x, ty = tile / lay.tiles_x;
    const int tile_x0 = tx * TILE, tile_y0 = ty * TILE;
    const int px = tile_x0 + (lane & 7), py = tile_y0 + (lane >> 3);
    float depth_zr = 1.f;
    if constexpr (OUT == OUT_XREC && DEPTHREC) {
      if (depth_mode != 0 && depth_tcr) depth_zr = depth_tcr[3 * (size_t)item + 2];
    }
    auto nd = [&](float d) {   // (raster_tiles' `nd`, verbatim)
      if constexpr (OUT == OUT_XREC && DEPTHREC) {
        if (depth_mode == 1) d = d / depth_zr;
        else if (depth_mode == 2) d = fminf(fmaxf(d / depth_zr, 0.f), 2.f) - 1.f;
        else if (depth_mode == 3) d = fminf(fmaxf(d - depth_zr, -2.f), 2.f);
      }
      return d;
    };
    if constexpr (OUT == OUT_XREC) {
      uint4* z = reinterpret_cast<uint4*>(my_rec);
      for (int q = 0; q < (int)stride_x / 8; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int r = 0; r < views_per_item; ++r) {
      const int cv = (int)((long long)r * stride_view);
      if constexpr (OUT != OUT_XREC) {
        if (c_rgb >= 0) { put(c_rgb + cv, 0.f); put(c_rgb + cv + 1, 0.f); put(c_rgb + cv + 2, 0.f); }
        if (do_norm) { put(c_normals + cv, 0.f); put(c_normals + cv + 1, 0.f); put(c_normals + cv + 2, 0.f); }
        if (do_depth) put(c_depth + cv, 0.f);
      } else {
        if (do_depth) put(c_depth + cv, nd(0.f));
      }
    }
    if (crop.images && px < w && py < h) {
      const float4 cv4 = crop_lane_body<true>(crop, item, h, w, px, py);   // (inlined: the shared out-of-line copy would impose raster_tiles' register count)
      put(crop.c0, cv4.x); put(crop.c0 + 1, cv4.y); put(crop.c0 + 2, cv4.z);
      if (crop.C == 4) put(crop.c0 + 3, nd(cv4.w));
    }
    wave_lds_fence();
    const int cols = min(TILE, w - tile_x0), rows = min(TILE, h - tile_y0);
    const int per_row = cols * run;
    if constexpr (OUT == OUT_XREC) {
      const int rowlen = cols * ((int)stride_x / 8);
      unsigned short* out_item = reinterpret_cast<unsigned short*>(out) + (size_t)item * stride_v;
      const uint4* recs = reinterpret_cast<const uint4*>(stage);
      if (lane < rowlen)
        for (int row = 0; row < rows; ++row)
          *reinterpret_cast<uint4*>(out_item + (size_t)(tile_y0 + row) * stride_y + (size_t)tile_x0 * stride_x + (size_t)lane * 8) =
              recs[row * 8 * ((int)stride_x / 8) + lane];
    } else if constexpr (OUT == OUT_F16) {
      _Float16* out_item = reinterpret_cast<_Float16*>(out) + (size_t)item * stride_v + c_lo;
      for (int i = lane; i < per_row; i += 64) {
        const int x = i / run, c = i - x * run;
        if (!((run_mask >> c) & 1u)) continue;
        _Float16* o = out_item + (size_t)tile_y0 * stride_y + (size_t)(tile_x0 + x) * stride_x + c;
        const float* sp = stage + (size_t)x * run + c;
        for (int row = 0; row < rows; ++row) o[(size_t)row * stride_y] = (_Float16)sp[(size_t)row * 8 * run];
      }
    } else {
      float* out_item = out + (size_t)item * stride_v + c_lo;
      for (int i = lane; i < per_row; i += 64) {
        const int x = i / run, c = i - x * run;
        if (!((run_mask >> c) & 1u)) continue;
        float* o = out_item + (size_t)tile_y0 * stride_y + (size_t)(tile_x0 + x) * stride_x + c;
        const float* sp = stage + (size_t)x * run + c;
        for (int row = 0; row < rows; ++row) o[(size_t)row * stride_y] = sp[(size_t)row * 8 * run];
      }
    }
    wave_lds_fence();   // the next job restages
  }
}

}  // namespace mp

using namespace mp;

struct mp_mesh_db {
  int n;
  int max_verts, max_faces;
  bool any_texture;   // some mesh has uvs + a texture: raster_tiles needs its FULL instance
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
  db->any_texture = false;
  db->d_meshes = nullptr;
  db->d_texs = nullptr;
  for (int i = 0; i < n; ++i) {
    const mp_mesh_desc& d = hm[i];
    MP_REQUIRE(d.h_vertices && d.h_normals && d.h_colors && d.h_faces && d.n_vertices > 0 && d.n_faces > 0,
               "mp_mesh_db_create: mesh %d incomplete", i);
    MP_REQUIRE(d.n_faces < (1 << 22), "mp_mesh_db_create: mesh %d has %d faces (limit 4 194 303: piece ids are packed in 24 bits)", i, d.n_faces);
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
    float center[3];
    for (int k = 0; k < 3; ++k) center[k] = 0.5f * (lo[k] + hi[k]);
    float r2 = 0.f;
    for (int v = 0; v < d.n_vertices; ++v) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) {
        const float dd = d.h_vertices[3 * v + k] - center[k];
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
  db->any_texture = true;
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

static BinLayout bin_layout(const mp_mesh_db* db, int h, int w);
static size_t job_tail_offset_ints(const BinLayout& lay, int n_views);
// what the last launch on a workspace left behind its view blocks (host-side record): the job flags are only handed out for the
// launch that wrote them -- compacted form, same number of views, same image size
struct LastLaunch { bool compact; int n_views, h, w; };
static std::mutex g_last_mu;
static std::map<const void*, LastLaunch> g_last_launch;
static void note_launch(const void* d_ws, bool compact, int n_views, int h, int w) {
  std::lock_guard<std::mutex> lock(g_last_mu);
  if (g_last_launch.size() > 256) g_last_launch.clear();   // (workspaces come and go with their allocations: the record is a cache)
  g_last_launch[d_ws] = LastLaunch{compact, n_views, h, w};
}

extern "C" const unsigned char* mp_raster_job_flags(const mp_mesh_db* db, const void* d_ws, int n_views, int h, int w) {
  if (!db || !d_ws || n_views <= 0 || h <= 0 || w <= 0) return nullptr;
  {
    std::lock_guard<std::mutex> lock(g_last_mu);
    auto it = g_last_launch.find(d_ws);
    if (it == g_last_launch.end() || !it->second.compact || it->second.n_views != n_views || it->second.h != h || it->second.w != w) return nullptr;
  }
  const BinLayout lay = bin_layout(db, h, w);
  const int* counters = (const int*)d_ws + job_tail_offset_ints(lay, n_views);
  return (const unsigned char*)(counters + 4 + (size_t)n_views * lay.n_tiles);
}

static BinLayout bin_layout(const mp_mesh_db* db, int h, int w) {
  BinLayout lay;
  lay.tiles_x = ceil_div(w, TILE);
  lay.tiles_y = ceil_div(h, TILE);
  lay.n_tiles = lay.tiles_x * lay.tiles_y;
  lay.max_faces = db->max_faces;
  lay.cap_list = 3 * db->max_faces + 2048;
  lay.cap_large = 4 * db->max_faces + 4096;
  lay.off_tl = (HDR_INTS + lay.n_tiles + 1 + 3) & ~3;
  lay.off_large = (lay.off_tl + lay.n_tiles + 1 + 3) & ~3;
  lay.off_list = (lay.off_large + lay.cap_large + 3) & ~3;
  lay.view_ints = (long long)lay.off_list + (long long)lay.cap_list * (long long)(sizeof(rc::TileRec) / sizeof(int));
  return lay;
}

// behind the per-view blocks: [4 counters][light job list: one int per (view, tile)][job flags: one byte per (item, tile), sized for items =
// views][per-view tile flags: one byte per (view, tile)]
static size_t job_tail_offset_ints(const BinLayout& lay, int n_views) { return ((size_t)n_views * (size_t)lay.view_ints + 3) & ~(size_t)3; }

extern "C" size_t mp_raster_workspace_bytes(const mp_mesh_db* db, int n_views, int h, int w) {
  if (!db || n_views <= 0 || h <= 0 || w <= 0) return 0;
  const BinLayout lay = bin_layout(db, h, w);
  const size_t pairs = (size_t)n_views * lay.n_tiles;
  return (job_tail_offset_ints(lay, n_views) + 4 + pairs) * sizeof(int) + 2 * ((pairs + 15) & ~(size_t)15);   // + job flags + per-view tile flags
}

static int raster_render_impl(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K,
                              int n_views, int h, int w, uint32_t flags, const mp_lights* lights, float* d_out,
                              int64_t stride_v, int views_per_item, int64_t stride_view, int64_t stride_y, int64_t stride_x,
                              int c_rgb, int c_normals, int c_depth, void* d_ws, size_t ws_bytes, mp_stream stream, const CropArgs& crop,
                              uint32_t f32_mask = 0u, const float* d_tcr = nullptr, int depth_mode = 0) {
  MP_REQUIRE(db && d_mesh_ids && d_TCO && d_K && d_out && lights, "mp_raster_render: null pointer");
  MP_REQUIRE(n_views >= 0 && h > 0 && w > 0 && w <= 1024 && h <= 1024 && views_per_item >= 1 && views_per_item <= 64,
             "mp_raster_render: bad size (h, w <= 1024; 1 <= views_per_item <= 64)");
  MP_REQUIRE(lights->n_point >= 0 && lights->n_point <= 8, "mp_raster_render: too many point lights");
  MP_REQUIRE((flags & ~(MP_RASTER_NORMALS | MP_RASTER_DEPTH | MP_RASTER_NORMALS_GL | MP_RASTER_MSAA4 | MP_RASTER_F16 | MP_RASTER_XREC)) == 0,
             "mp_raster_render: unknown flag bits 0x%x", flags);
  if (n_views == 0) return MP_OK;
  MP_REQUIRE(ws_bytes >= mp_raster_workspace_bytes(db, n_views, h, w), "mp_raster_render: workspace too small");
  MP_REQUIRE(n_views % views_per_item == 0, "mp_raster_render: n_views (%d) must be a multiple of views_per_item (%d)", n_views, views_per_item);
  hipStream_t s = (hipStream_t)stream;
  LightsDev L;
  memcpy(L.ambient, lights->ambient, sizeof(L.ambient));
  L.n_point = lights->n_point;
  memcpy(L.dir, lights->point_dir, sizeof(L.dir));
  memcpy(L.color, lights->point_color, sizeof(L.color));
  memcpy(L.offset, lights->point_offset, sizeof(L.offset));
  const BinLayout lay = bin_layout(db, h, w);
  const int ns = (flags & MP_RASTER_MSAA4) ? 4 : 1;
  const int n_items = n_views / views_per_item;
  // compacted launch form (light-job list + per-pair flags behind the view blocks) unless MP_RASTER_COMPACT=0 (read per launch: the A/B test flips it)
  int* const counters = (int*)d_ws + job_tail_offset_ints(lay, n_views);
  int* const light_list = counters + 4;
  unsigned char* const job_flags = (unsigned char*)(light_list + (size_t)n_views * lay.n_tiles);
  unsigned char* const view_flags = job_flags + (((size_t)n_views * lay.n_tiles + 15) & ~(size_t)15);
  const char* compact_env = getenv("MP_RASTER_COMPACT");
  const bool compact = !(compact_env && compact_env[0] && atoi(compact_env) == 0);   // (the ONE place the switch is parsed: "0" / "00" = direct form)
  note_launch(d_ws, compact, n_views, h, w);
  const bool do_norm = (flags & MP_RASTER_NORMALS) && c_normals >= 0, do_depth = (flags & MP_RASTER_DEPTH) && c_depth >= 0;
  // channel run [c_lo, c_hi) one pixel record of this launch spans, and which of its channels are written
  int c_lo = 1 << 30, c_hi = -1;
  auto span = [&](int c0, int n) { c_lo = std::min(c_lo, c0); c_hi = std::max(c_hi, c0 + n); };
  for (int r = 0; r < views_per_item; ++r) {
    if (c_rgb >= 0) span(c_rgb + r * (int)stride_view, 3);
    if (do_norm) span(c_normals + r * (int)stride_view, 3);
    if (do_depth) span(c_depth + r * (int)stride_view, 1);
  }
  if (crop.images) span(crop.c0, crop.C);
  MP_REQUIRE(c_hi > c_lo, "mp_raster_render: nothing to write");
  const int run = c_hi - c_lo;
  MP_REQUIRE(run <= MAX_RUN, "mp_raster_render: one launch writes at most %d channels per pixel (asked for %d)", MAX_RUN, run);
  MP_REQUIRE(stride_x >= run, "mp_raster_render: stride_x (%lld) smaller than the channel run (%d)", (long long)stride_x, run);
  uint32_t mask = 0;
  auto mark = [&](int c0, int n) { for (int c = c0; c < c0 + n; ++c) mask |= 1u << (c - c_lo); };
  for (int r = 0; r < views_per_item; ++r) {
    if (c_rgb >= 0) mark(c_rgb + r * (int)stride_view, 3);
    if (do_norm) mark(c_normals + r * (int)stride_view, 3);
    if (do_depth) mark(c_depth + r * (int)stride_view, 1);
  }
  if (crop.images) mark(crop.c0, crop.C);
  {
    const size_t lds = (size_t)2 * lay.n_tiles * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
      MP_CHECK_HIP(hipFuncSetAttribute((const void*)raster_bin, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      attr_set = true;
    }
    ProfScope prof("raster_bin", 0.0, (double)n_views * (12.0 * db->max_faces + 12.0 * db->max_verts + 4.0 * lay.n_tiles), s);
    hipLaunchKernelGGL(raster_bin, dim3(n_views), dim3(BIN_THREADS), lds, s, db->d_meshes, d_mesh_ids, d_TCO, d_K, h, w, ns, (int*)d_ws, lay, counters,
                       compact ? view_flags : (unsigned char*)nullptr);
  }
  const int groups_x = ceil_div(lay.tiles_x, TILE_WAVES);
  const long long n_wg = (long long)n_items * lay.tiles_y * groups_x;
  MP_REQUIRE(n_wg < (1LL << 31), "mp_raster_render: grid too large");
  const int n_ch = (c_rgb >= 0 ? 3 : 0) + (do_norm ? 3 : 0) + (do_depth ? 1 : 0);
  // algorithmic bytes: output channels written once + the mesh (32 B/vertex, 12 B/triangle) read once per view
  // (+ the fused crop role: C output channels written + at most the same-sized source window read per item)
  const bool f16 = (flags & MP_RASTER_F16) != 0, xrec = (flags & MP_RASTER_XREC) != 0;
  if (xrec) {   // stem records: ONE launch writes the whole record of every pixel
    MP_REQUIRE(!f16 && crop.images && crop.c0 == 0 && c_lo == 0 && mask == (run >= 32 ? 0xFFFFFFFFu : (1u << run) - 1u),
               "mp_raster_render: MP_RASTER_XREC needs the fused crop at channel 0 and every channel of the record written by this launch");
    if (f32_mask == 0u) f32_mask = (1u << crop.C) - 1u;   // mp_raster_render_crop: the crop's channels are the fp32-kind ones
    // which channels are fp32-kind is the caller's statement; it must agree with what this launch writes where
    uint32_t want = (1u << crop.C) - 1u;
    for (int r = 0; r < views_per_item; ++r)
      if (do_depth) want |= 1u << (c_depth + r * (int)stride_view);
    MP_REQUIRE(f32_mask == want, "mp_raster_render: MP_RASTER_XREC: fp32-kind channel mask 0x%x, but this launch writes the crop + depth channels 0x%x "
               "(rgb / normal channels are 8-bit integers, crop and depth channels fp32)", f32_mask, want);
    MP_REQUIRE(!do_depth || depth_mode == 0 || d_tcr, "mp_raster_render_xrec: depth normalisation needs d_tCR");
    MP_REQUIRE(depth_mode >= 0 && depth_mode <= 3, "mp_raster_render_xrec: unknown depth mode %d", depth_mode);
    const int n_f32 = __builtin_popcount(f32_mask);
    MP_REQUIRE(stride_x == mp_xrec_elements(n_f32, run - n_f32) && stride_x <= 48,
               "mp_raster_render: MP_RASTER_XREC: stride_x (%lld) must be the record length mp_xrec_elements(%d, %d) <= 48", (long long)stride_x, n_f32,
               run - n_f32);
  }
  // LDS staging per pixel, in floats: the channel run -- or, for stem records, the record itself if that is longer (a 6-channel model, 3 crop
  // + 3 render channels, has a 16-element = 32-byte record but a run of only 6 floats: the record's zero padding is part of what is staged)
  const int run_lds = xrec ? std::max(run, ((int)stride_x + 1) / 2) : run;
  const size_t lds = (size_t)TILE_WAVES * (tiles_zt_bytes(ns) + (((size_t)64 * run_lds * sizeof(float) + 15) & ~(size_t)15) + HDR_LDS_BYTES);
  const double out_es = f16 ? 2.0 : 4.0;   // bytes per output element
  const double alg_bytes = xrec ? (double)n_views * (32.0 * db->max_verts + 12.0 * db->max_faces) + (double)n_items * (2.0 * stride_x + 4.0 * crop.C) * h * w :
                           (double)n_views * ((double)n_ch * out_es * h * w + 32.0 * db->max_verts + 12.0 * db->max_faces) +
                               (crop.images ? (double)n_items * crop.C * (out_es + 4.0) * h * w : 0.0);   // crop: C channels written + <= the same-sized fp32 source window read
  // FULL = texture + point-light code compiled in; the pose networks' renders (vertex colours, ambient light) take the lean instance
  const bool full = db->any_texture || L.n_point > 0;
#define MP_LAUNCH_TILES(NSV, OUTV, FULLV)                                                                                              \
  hipLaunchKernelGGL((raster_tiles<NSV, OUTV, FULLV>), dim3((unsigned)n_wg), dim3(64 * TILE_WAVES), lds, s, db->d_meshes, db->d_texs,    \
                     d_mesh_ids, d_TCO, d_K, (const int*)d_ws, lay, h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items,  \
                     (long long)stride_view, (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop,       \
                     f32_mask, d_tcr, depth_mode, compact ? job_flags : (const unsigned char*)nullptr)
  const bool depthrec = xrec && (do_depth || f32_mask != (1u << crop.C) - 1u);
  if (compact) {
    ProfScope prof_c("raster_classify", 0.0, (double)n_views * lay.n_tiles * 16.0, s);
    const long total = (long)n_items * lay.n_tiles;
    hipLaunchKernelGGL(raster_classify, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const unsigned char*)view_flags, lay, views_per_item,
                       n_items, job_flags, counters, light_list);
  }
  const int sel = depthrec ? (ns == 4 ? 16 : 17) + (full ? 2 : 0) : (ns == 4 ? 8 : 0) | (xrec ? 4 : f16 ? 2 : 0) | (full ? 1 : 0);
  {
  ProfScope prof(f16 ? "raster_tiles/f16" : xrec ? "raster_tiles/xrec" : "raster_tiles", 0.0, alg_bytes, s);
  switch (sel) {
    case 16: hipLaunchKernelGGL((raster_tiles<4, OUT_XREC, false, true>), dim3((unsigned)n_wg), dim3(64 * TILE_WAVES), lds, s, db->d_meshes, db->d_texs,
                     d_mesh_ids, d_TCO, d_K, (const int*)d_ws, lay, h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items,
                     (long long)stride_view, (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop,
                     f32_mask, d_tcr, depth_mode, compact ? job_flags : (const unsigned char*)nullptr); break;
    case 17: hipLaunchKernelGGL((raster_tiles<1, OUT_XREC, false, true>), dim3((unsigned)n_wg), dim3(64 * TILE_WAVES), lds, s, db->d_meshes, db->d_texs,
                     d_mesh_ids, d_TCO, d_K, (const int*)d_ws, lay, h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items,
                     (long long)stride_view, (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop,
                     f32_mask, d_tcr, depth_mode, compact ? job_flags : (const unsigned char*)nullptr); break;
    case 18: hipLaunchKernelGGL((raster_tiles<4, OUT_XREC, true, true>), dim3((unsigned)n_wg), dim3(64 * TILE_WAVES), lds, s, db->d_meshes, db->d_texs,
                     d_mesh_ids, d_TCO, d_K, (const int*)d_ws, lay, h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items,
                     (long long)stride_view, (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop,
                     f32_mask, d_tcr, depth_mode, compact ? job_flags : (const unsigned char*)nullptr); break;
    case 19: hipLaunchKernelGGL((raster_tiles<1, OUT_XREC, true, true>), dim3((unsigned)n_wg), dim3(64 * TILE_WAVES), lds, s, db->d_meshes, db->d_texs,
                     d_mesh_ids, d_TCO, d_K, (const int*)d_ws, lay, h, w, flags, L, d_out, (long long)stride_v, views_per_item, n_items,
                     (long long)stride_view, (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop,
                     f32_mask, d_tcr, depth_mode, compact ? job_flags : (const unsigned char*)nullptr); break;
    case 0: MP_LAUNCH_TILES(1, OUT_F32, false); break;
    case 1: MP_LAUNCH_TILES(1, OUT_F32, true); break;
    case 2: MP_LAUNCH_TILES(1, OUT_F16, false); break;
    case 3: MP_LAUNCH_TILES(1, OUT_F16, true); break;
    case 4: MP_LAUNCH_TILES(1, OUT_XREC, false); break;
    case 5: MP_LAUNCH_TILES(1, OUT_XREC, true); break;
    case 8: MP_LAUNCH_TILES(4, OUT_F32, false); break;
    case 9: MP_LAUNCH_TILES(4, OUT_F32, true); break;
    case 10: MP_LAUNCH_TILES(4, OUT_F16, false); break;
    case 11: MP_LAUNCH_TILES(4, OUT_F16, true); break;
    case 12: MP_LAUNCH_TILES(4, OUT_XREC, false); break;
    default: MP_LAUNCH_TILES(4, OUT_XREC, true); break;
  }
  }
#undef MP_LAUNCH_TILES
  if (compact) {   // the pairs no view reaches: background + crop, strided over the light list, four waves per SIMD
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      MP_CHECK_HIP(hipGetDevice(&dev));
      MP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const size_t lds_l = (size_t)TILE_WAVES * (((size_t)64 * run_lds * sizeof(float) + 15) & ~(size_t)15);
    const long long n_res = std::min<long long>((long long)n_cu * 8, std::max<long long>(1, ((long long)n_items * lay.n_tiles + TILE_WAVES - 1) / TILE_WAVES));
    ProfScope prof_l(f16 ? "raster_tiles_light/f16" : xrec ? "raster_tiles_light/xrec" : "raster_tiles_light", 0.0, 0.0, s);
#define MP_LAUNCH_LIGHT(OUTV, DR)                                                                                                              \
    hipLaunchKernelGGL((raster_tiles_light<OUTV, DR>), dim3((unsigned)n_res), dim3(64 * TILE_WAVES), lds_l, s, (const int*)counters,           \
                       (const int*)light_list, lay, h, w, flags, d_out, (long long)stride_v, views_per_item, (long long)stride_view,           \
                       (long long)stride_y, (long long)stride_x, c_rgb, c_normals, c_depth, c_lo, run_lds, mask, crop, f32_mask, d_tcr, depth_mode)
    if (depthrec) MP_LAUNCH_LIGHT(OUT_XREC, true);
    else if (xrec) MP_LAUNCH_LIGHT(OUT_XREC, false);
    else if (f16) MP_LAUNCH_LIGHT(OUT_F16, false);
    else MP_LAUNCH_LIGHT(OUT_F32, false);
#undef MP_LAUNCH_LIGHT
  }
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

#ifdef MP_RASTER_PROF
extern "C" int mp_raster_prof_read(unsigned long long* out16, int reset) {
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_raster_prof), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {0};
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_raster_prof), z, sizeof(z)));
  }
  return MP_OK;
}
#endif

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

extern "C" int mp_raster_render_xrec(const mp_mesh_db* db, const int32_t* d_mesh_ids, const float* d_TCO, const float* d_K, int n_views, int h,
                                     int w, uint32_t flags, const mp_lights* lights, void* d_out_records, int64_t stride_v, int views_per_item,
                                     int64_t stride_view, int64_t stride_y, int64_t stride_x, int c_rgb, int c_normals, int c_depth, void* d_ws,
                                     size_t ws_bytes, const float* d_images, int images_nhwc4, int n_im, int C, int H, int W,
                                     const int32_t* d_im_ids, const float* d_boxes, uint32_t f32_mask, const float* d_tCR, int depth_mode,
                                     mp_stream stream) {
  MP_REQUIRE(d_images && d_im_ids && d_boxes && n_im > 0 && (C == 3 || C == 4) && H > 0 && W > 0, "mp_raster_render_xrec: bad crop arguments");
  MP_REQUIRE(f32_mask != 0u, "mp_raster_render_xrec: empty fp32-kind channel mask");
  CropArgs crop;
  crop.images = d_images; crop.im_ids = d_im_ids; crop.boxes = d_boxes;
  crop.C = C; crop.H = H; crop.W = W; crop.c0 = 0; crop.nhwc4 = images_nhwc4 ? 1 : 0;
  return raster_render_impl(db, d_mesh_ids, d_TCO, d_K, n_views, h, w, (flags | MP_RASTER_XREC) & ~MP_RASTER_F16, lights, (float*)d_out_records,
                            stride_v, views_per_item, stride_view, stride_y, stride_x, c_rgb, c_normals, c_depth, d_ws, ws_bytes, stream, crop,
                            f32_mask, d_tCR, depth_mode);
}
