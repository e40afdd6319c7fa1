// conv_split.hip -- fp32 convolution computed with bf16 MFMA through exact 3-way operand splitting ("bf16x9" / "bf16x6").
//
// OPTIONAL fast mode of the conv stack in conv.hip (same call sites, same padded-NHWC fp32 tensors in HBM, same epilogue).
// Every fp32 operand x is split by truncation into three bf16 pieces  x = hi + mid + lo  EXACTLY (24 mantissa bits = 3 x 8);
// every bf16 x bf16 partial product is exact in fp32, so
//     a*b = sum over the 9 (piece_a, piece_b) pairs,
// and only the ORDER of the fp32 accumulations differs from the native v_mfma_f32_32x32x2_f32 path -- the result is
// fp32-faithful (same error class as any re-association), not a reduced-precision approximation.  bf16x6 drops the three
// smallest pairs (mid*lo, lo*mid, lo*lo: relative weight <= 2^-24, 2^-24, 2^-32).
// Why: on gfx950 fp32 MFMA peaks at 157 TFLOP/s while v_mfma_f32_32x32x16_bf16 peaks at ~2.5 PFLOP/s, so 9 (6) bf16 MFMAs per
// fp32 MFMA-equivalent is a 1.78x (2.67x) higher ceiling: 278 (417) TFLOP/s of fp32-equivalent work.
//
// Structure: same "row-run implicit GEMM" as conv.hip (tile 128 x {128,64}, 4 waves, BK = 32 floats of K per chunk).  The A
// tile is loaded as fp32, split in registers and written to LDS as three bf16 planes; the weight pieces are pre-split on the
// host (mp_conv_pack_weights_split).  Fragments: lane l supplies row (l & 31), k = 8*(l >> 5) .. +7 of a 32x16 slab as 8 bf16.
#include <cstdlib>

#include "common.h"

namespace mp {
namespace split {

constexpr int BK = 32;         // fp32 K elements per chunk = two k16 MFMA steps
constexpr int LDH = BK + 8;    // LDS row length in bf16 (80 B): conflict-free 16-B fragment reads

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Params {
  const float* __restrict__ x;
  const unsigned short* __restrict__ w;  // [nblk][chunk][3][BN][32] bf16 pieces
  const float* __restrict__ bias;
  const float* __restrict__ residual;
  const float* __restrict__ act_scale;
  const float* __restrict__ act_shift;
  float* __restrict__ y;
  float* __restrict__ y_act;
  int M, Ho, Wo, Hp, Wp, C, in_off, stride, Cout, Hop, Wop, out_border;
  int run, n_chunks, relu, n_mblocks, n_nblocks;
};

template <int TM, int TN, bool RES, bool RELU, bool ACT>
__device__ __forceinline__ void epilogue(const Params& p, const f32x16 (&acc)[TM][TN], const int* row_off, int row0, int n_first) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_first + j * 32;
    if (n >= p.Cout) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
    float sc = 1.f, sh = 0.f;
    if (ACT) { sc = p.act_scale[n]; sh = p.act_shift[n]; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int offs[16];
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) offs[r] = row_off[row0 + i * 32 + (r & 3) + 8 * (r >> 2)];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = offs[r] >= 0 ? p.residual[offs[r] + n] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (offs[r] < 0) continue;
        float v = acc[i][j][r] + bias;
        if (RES) v += res[r];
        if (RELU) v = fmaxf(v, 0.f);
        if (p.y) p.y[offs[r] + n] = v;
        if (ACT) p.y_act[offs[r] + n] = fmaxf(fmaf(v, sc, sh), 0.f);
      }
    }
  }
}

// exact 3-way truncation split of 4 floats -> three packed bf16x4 (8 bytes each)
// v_perm_b32 selector 0x07060302: result = {hi16(second arg) in the low half, hi16(first arg) in the high half}
__device__ __forceinline__ unsigned pack_hi16(unsigned e1, unsigned e0) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = __float_as_uint(f[i]);                                   // only the upper 16 bits are kept by the packing below
    const float r = f[i] - __uint_as_float(h[i] & 0xFFFF0000u);     // exact
    m[i] = __float_as_uint(r);
    const float q = r - __uint_as_float(m[i] & 0xFFFF0000u);        // exact, <= 8 significant bits
    l[i] = __float_as_uint(q);
  }
  hi = make_uint2(pack_hi16(h[1], h[0]), pack_hi16(h[3], h[2]));
  mid = make_uint2(pack_hi16(m[1], m[0]), pack_hi16(m[3], m[2]));
  lo = make_uint2(pack_hi16(l[1], l[0]), pack_hi16(l[3], l[2]));
}

template <int NW, int BM, int BN, int WM, int WN, int NPROD, bool RAGGED, bool DBUF>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv_nhwc_f32_bf16split(Params p) {
  static_assert(BM == 32 * NW && (BN == 128 || BN == 64) && (BM / WM) * (BN / WN) == NW, "tile shape");
  constexpr int NT = NW * 64;      // threads
  constexpr int RSTEP = NT / 8;    // A rows covered by one pass of the threads (8 lanes per 32-float row piece)
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int B_LD = 3 * BN * BK * 2 / 16 / NT;  // 16-byte loads per thread for the weight pieces (6 or 3)

  extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
  constexpr int NBUF = DBUF ? 2 : 1;
  constexpr int A_SZ = 3 * BM * LDH, B_SZ = 3 * BN * LDH;   // bf16 elements per buffer
  unsigned short* As0 = smem_h;                      // [NBUF][3][BM][LDH]
  unsigned short* Bs0 = smem_h + NBUF * A_SZ;        // [NBUF][3][BN][LDH]
  int* row_off = (int*)(Bs0 + NBUF * B_SZ);          // [BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / (BN / WN), wn = wave % (BN / WN);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int nblk = lb % p.n_nblocks, mblk = lb / p.n_nblocks;
  const int m0 = mblk * BM, n0 = nblk * BN;

  const int a_c4 = tid & 7, a_r0 = tid >> 3;
  const float* a_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + a_r0 + RSTEP * i;
    m = m < p.M ? m : p.M - 1;
    const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, n = t / p.Ho;
    const size_t pix = ((size_t)n * p.Hp + (size_t)(ho * p.stride + p.in_off)) * p.Wp + (size_t)(wo * p.stride + p.in_off);
    a_ptr[i] = p.x + pix * p.C;
  }
  for (int r = tid; r < BM; r += NT) {
    const int m = m0 + r;
    int off = -1;
    if (m < p.M) {
      const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, n = t / p.Ho;
      off = (((n * p.Hop) + ho + p.out_border) * p.Wop + wo + p.out_border) * p.Cout;
    }
    row_off[r] = off;
  }
  const float* a_ptr0 = a_ptr[0];
  const float* a_ptr1 = a_ptr[1];
  const float* a_ptr2 = a_ptr[2];
  const float* a_ptr3 = a_ptr[3];
  const uint4* bp = reinterpret_cast<const uint4*>(p.w) + (size_t)nblk * p.n_chunks * (3 * BN * BK * 2 / 16) + tid;
  const int row_stride = p.Wp * p.C, row_wrap = row_stride - p.run;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 a0, a1, a2, a3;
  uint4 b0, b1, b2, b3, b4, b5;  // explicit scalars (arrays + sched_barrier end up in scratch, see conv.hip)
  static_assert(B_LD == 3 || B_LD == 6, "weight staging is written for BN in {64, 128}");
  int j = a_c4 * 4, aoff = a_c4 * 4, ju = 0;
  if constexpr (RAGGED) {
    while (j >= p.run) { j -= p.run; aoff += row_wrap; }
  }
#define MPS_LOAD()                                                   \
  a0 = *reinterpret_cast<const float4*>(a_ptr0 + aoff);              \
  a1 = *reinterpret_cast<const float4*>(a_ptr1 + aoff);              \
  a2 = *reinterpret_cast<const float4*>(a_ptr2 + aoff);              \
  a3 = *reinterpret_cast<const float4*>(a_ptr3 + aoff);              \
  b0 = bp[0]; b1 = bp[NT]; b2 = bp[2 * NT];                          \
  if constexpr (B_LD == 6) { b3 = bp[3 * NT]; b4 = bp[4 * NT]; b5 = bp[5 * NT]; }
#define MPS_BST(Q, V)                                                                                     \
  {                                                                                                       \
    const int idx = tid + NT * (Q); /* 16-byte piece inside [3][BN][4 groups of 8 bf16] */                \
    const int s_ = idx / (BN * 4), rem = idx - s_ * (BN * 4);                                             \
    *reinterpret_cast<uint4*>(Bs_w + (s_ * BN + (rem >> 2)) * LDH + (rem & 3) * 8) = (V);                   \
  }
#define MPS_SPLIT()            \
  split4(a0, h0, m0_, l0);     \
  split4(a1, h1, m1_, l1);     \
  split4(a2, h2, m2_, l2);     \
  split4(a3, h3, m3_, l3);
#define MPS_WRITE(BUF)                                                                                    \
  {                                                                                                       \
    unsigned short* Bs_w = Bs0 + (BUF) * B_SZ;                                                            \
    unsigned short* aw = As0 + (BUF) * A_SZ + a_r0 * LDH + a_c4 * 4;                                      \
    *reinterpret_cast<uint2*>(aw) = h0; *reinterpret_cast<uint2*>(aw + BM * LDH) = m0_; *reinterpret_cast<uint2*>(aw + 2 * BM * LDH) = l0; \
    aw += RSTEP * LDH;                                                                                     \
    *reinterpret_cast<uint2*>(aw) = h1; *reinterpret_cast<uint2*>(aw + BM * LDH) = m1_; *reinterpret_cast<uint2*>(aw + 2 * BM * LDH) = l1; \
    aw += RSTEP * LDH;                                                                                     \
    *reinterpret_cast<uint2*>(aw) = h2; *reinterpret_cast<uint2*>(aw + BM * LDH) = m2_; *reinterpret_cast<uint2*>(aw + 2 * BM * LDH) = l2; \
    aw += RSTEP * LDH;                                                                                     \
    *reinterpret_cast<uint2*>(aw) = h3; *reinterpret_cast<uint2*>(aw + BM * LDH) = m3_; *reinterpret_cast<uint2*>(aw + 2 * BM * LDH) = l3; \
    MPS_BST(0, b0) MPS_BST(1, b1) MPS_BST(2, b2)                                                          \
    if constexpr (B_LD == 6) { MPS_BST(3, b3) MPS_BST(4, b4) MPS_BST(5, b5) }                             \
  }
#define MPS_STORE(BUF) MPS_SPLIT() MPS_WRITE(BUF)
  uint2 h0, m0_, l0, h1, m1_, l1, h2, m2_, l2, h3, m3_, l3;

  MPS_LOAD()
  MPS_STORE(0)
  __syncthreads();

  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
    const int buf = DBUF ? (chunk & 1) : 0;
    if (chunk + 1 < p.n_chunks) {
      bp += 3 * BN * BK * 2 / 16;
      aoff += BK;
      if constexpr (RAGGED) {
        j += BK;
        while (j >= p.run) { j -= p.run; aoff += row_wrap; }
      } else {
        ju += BK;
        if (ju == p.run) { ju = 0; aoff += row_wrap; }
      }
    }
    MPS_LOAD()
    __builtin_amdgcn_sched_barrier(0);
    const unsigned short* as = As0 + buf * A_SZ + (wm * WM + frow) * LDH + fk;
    const unsigned short* bs = Bs0 + buf * B_SZ + (wn * WN + frow) * LDH + fk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          af[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(as + (s * BM + i * 32) * LDH + ks * 16));
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          bf[jn][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bs + (s * BN + jn * 32) * LDH + ks * 16));
      if constexpr (DBUF) {
        if (ks == 1) {  // split + write the prefetched chunk into the other buffer under the last MFMA group
          __builtin_amdgcn_sched_barrier(0);
          MPS_STORE(buf ^ 1)
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if (ks == 1) {  // split the prefetched A rows in registers under the last MFMA group (VALU and MFMA pipes overlap)
          MPS_SPLIT()
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          // smallest terms first
          if constexpr (NPROD == 9) {
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[jn][2], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[jn][2], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[jn][1], acc[i][jn], 0, 0, 0);
          }
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[jn][2], acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[jn][1], acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[jn][0], acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[jn][1], acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[jn][0], acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[jn][0], acc[i][jn], 0, 0, 0);
        }
    }
    if constexpr (DBUF) {
      __syncthreads();
    } else {
      __syncthreads();  // every wave is done reading the tile
      MPS_WRITE(0)
      __syncthreads();
    }
  }
#undef MPS_LOAD
#undef MPS_STORE
#undef MPS_SPLIT
#undef MPS_WRITE
#undef MPS_BST

  const int emode = (p.residual ? 1 : 0) | (p.relu ? 2 : 0) | (p.y_act ? 4 : 0);
  const int erow0 = wm * WM + (lane >> 5) * 4, en0 = n0 + wn * WN + (lane & 31);
  switch (emode) {
    case 0: epilogue<TM, TN, false, false, false>(p, acc, row_off, erow0, en0); break;
    case 1: epilogue<TM, TN, true, false, false>(p, acc, row_off, erow0, en0); break;
    case 2: epilogue<TM, TN, false, true, false>(p, acc, row_off, erow0, en0); break;
    case 3: epilogue<TM, TN, true, true, false>(p, acc, row_off, erow0, en0); break;
    case 4: epilogue<TM, TN, false, false, true>(p, acc, row_off, erow0, en0); break;
    case 5: epilogue<TM, TN, true, false, true>(p, acc, row_off, erow0, en0); break;
    case 6: epilogue<TM, TN, false, true, true>(p, acc, row_off, erow0, en0); break;
    default: epilogue<TM, TN, true, true, true>(p, acc, row_off, erow0, en0); break;
  }
}

static inline int bn_tile(int Cout) { return Cout <= 64 ? 64 : 128; }

template <int NW, int BN, int WM, int WN, int NPROD, bool RAGGED, bool DBUF>
static int launch(Params p, hipStream_t s, double flops, double bytes) {
  constexpr int BM = 32 * NW;
  p.n_mblocks = ceil_div(p.M, BM);
  p.n_nblocks = ceil_div(p.Cout, BN);
  const size_t lds = (size_t)(DBUF ? 2 : 1) * (3 * BM * LDH + 3 * BN * LDH) * sizeof(unsigned short) + BM * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv_nhwc_f32_bf16split<NW, BM, BN, WM, WN, NPROD, RAGGED, DBUF>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  ProfScope prof(BN == 64 ? (NPROD == 9 ? "conv_nhwc_f32_bf16x9<128,64>" : "conv_nhwc_f32_bf16x6<128,64>")
                 : NW == 8 ? (NPROD == 9 ? "conv_nhwc_f32_bf16x9<256,128>" : "conv_nhwc_f32_bf16x6<256,128>")
                           : (NPROD == 9 ? "conv_nhwc_f32_bf16x9<128,128>" : "conv_nhwc_f32_bf16x6<128,128>"), flops, bytes, s);
  hipLaunchKernelGGL((conv_nhwc_f32_bf16split<NW, BM, BN, WM, WN, NPROD, RAGGED, DBUF>), dim3(p.n_mblocks * p.n_nblocks), dim3(NW * 64), lds, s, p);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}

}  // namespace split
}  // namespace mp

using namespace mp;

extern "C" size_t mp_conv_packed_split_bytes(int Cin_p, int Cout, int KH, int KW) {
  const int BN = split::bn_tile(Cout);
  return (size_t)ceil_div(Cout, BN) * ceil_div((long)KH * KW * Cin_p, split::BK) * 3 * BN * split::BK * sizeof(unsigned short);
}

// bf16 pieces (hi, mid, lo by truncation) of w * scale, packed [nb][chunk][piece][n_local][32] over the concatenated K axis
extern "C" int mp_conv_pack_weights_split(const float* w, int Cout, int Cin, int KH, int KW, int Cin_p, const float* scale,
                                          void* packed_bytes) {
  MP_REQUIRE(w && packed_bytes && Cin_p >= Cin && (Cin_p % 4) == 0, "mp_conv_pack_weights_split: bad arguments");
  const int BN = split::bn_tile(Cout), nblk = ceil_div(Cout, BN), run = KW * Cin_p, k_total = KH * run;
  const int n_chunks = ceil_div(k_total, split::BK);
  unsigned short* out = (unsigned short*)packed_bytes;
  memset(out, 0, mp_conv_packed_split_bytes(Cin_p, Cout, KH, KW));
  for (int nb = 0; nb < nblk; ++nb)
    for (int ch = 0; ch < n_chunks; ++ch) {
      unsigned short* tile = out + ((size_t)nb * n_chunks + ch) * 3 * BN * split::BK;
      for (int nl = 0; nl < BN; ++nl) {
        const int n = nb * BN + nl;
        if (n >= Cout) continue;
        const float s = scale ? scale[n] : 1.f;
        for (int k = 0; k < split::BK; ++k) {
          const int kidx = ch * split::BK + k;
          if (kidx >= k_total) continue;
          const int kh = kidx / run, jj = kidx % run, kw = jj / Cin_p, c = jj % Cin_p;
          if (c >= Cin) continue;
          const float v = w[(((size_t)n * Cin + c) * KH + kh) * KW + kw] * s;
          unsigned vb, rb, qb;
          memcpy(&vb, &v, 4);
          const unsigned h = vb & 0xFFFF0000u;
          float hf; memcpy(&hf, &h, 4);
          const float r = v - hf;
          memcpy(&rb, &r, 4);
          const unsigned m = rb & 0xFFFF0000u;
          float mf; memcpy(&mf, &m, 4);
          const float q = r - mf;
          memcpy(&qb, &q, 4);
          tile[(0 * BN + nl) * split::BK + k] = (unsigned short)(h >> 16);
          tile[(1 * BN + nl) * split::BK + k] = (unsigned short)(m >> 16);
          tile[(2 * BN + nl) * split::BK + k] = (unsigned short)(qb >> 16);
        }
      }
    }
  return MP_OK;
}

// same descriptor as mp_conv2d_nhwc, d_w points at the split blob; n_products = 9 or 6
extern "C" int mp_conv2d_nhwc_split(const mp_conv_desc* d, int n_products, mp_stream stream) {
  MP_REQUIRE(d && d->d_x && d->d_w && (d->d_y || d->d_y_act), "mp_conv2d_nhwc_split: null pointer");
  MP_REQUIRE(d->C % 4 == 0 && d->in_border >= d->pad && (n_products == 9 || n_products == 6), "mp_conv2d_nhwc_split: bad arguments");
  MP_REQUIRE(!d->d_y_act || (d->d_act_scale && d->d_act_shift), "mp_conv2d_nhwc_split: y_act needs scale/shift");
  MP_REQUIRE(!d->x_f16, "mp_conv2d_nhwc_split: half-precision inputs (x_f16) are implemented by mp_conv2d_nhwc only");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const long M = (long)d->N * Ho * Wo;
  MP_REQUIRE(M > 0 && M < (1L << 31) && (long)d->N * (Ho + 2 * d->out_border) * (Wo + 2 * d->out_border) * d->Cout < (1L << 31),
             "mp_conv2d_nhwc_split: size out of range");
  split::Params p;
  p.x = d->d_x; p.w = (const unsigned short*)d->d_w; p.bias = d->d_bias; p.residual = d->d_residual;
  p.act_scale = d->d_act_scale; p.act_shift = d->d_act_shift; p.y = d->d_y; p.y_act = d->d_y_act;
  p.M = (int)M; p.Ho = Ho; p.Wo = Wo; p.Hp = d->H + 2 * d->in_border; p.Wp = d->W + 2 * d->in_border; p.C = d->C;
  p.in_off = d->in_border - d->pad; p.stride = d->stride; p.Cout = d->Cout; p.Hop = Ho + 2 * d->out_border;
  p.Wop = Wo + 2 * d->out_border; p.out_border = d->out_border; p.run = d->KW * d->C;
  p.n_chunks = ceil_div((long)d->KH * p.run, split::BK); p.relu = d->relu;
  const double flops = 2.0 * (double)M * d->Cout * d->KH * d->KW * (d->c_real > 0 ? d->c_real : d->C);
  const double bytes = 4.0 * ((double)M * d->stride * d->stride * d->C + (double)M * d->Cout) + 6.0 * p.n_chunks * split::BK * d->Cout;
  hipStream_t s = (hipStream_t)stream;
  static const bool dbuf = getenv("MP_SPLIT_DBUF") ? atoi(getenv("MP_SPLIT_DBUF")) != 0 : false;  // tuning knob: single-buffered LDS keeps 2 workgroups per CU and measured faster
  const bool small = split::bn_tile(d->Cout) == 64, ragged = p.run % split::BK != 0, nine = n_products == 9;
#define MPS_GO(NW, BN, WM, WN)                                                                               \
  if (dbuf)                                                                                                   \
    return nine ? (ragged ? split::launch<NW, BN, WM, WN, 9, true, true>(p, s, flops, bytes) : split::launch<NW, BN, WM, WN, 9, false, true>(p, s, flops, bytes)) \
                : (ragged ? split::launch<NW, BN, WM, WN, 6, true, true>(p, s, flops, bytes) : split::launch<NW, BN, WM, WN, 6, false, true>(p, s, flops, bytes)); \
  return nine ? (ragged ? split::launch<NW, BN, WM, WN, 9, true, false>(p, s, flops, bytes) : split::launch<NW, BN, WM, WN, 9, false, false>(p, s, flops, bytes)) \
              : (ragged ? split::launch<NW, BN, WM, WN, 6, true, false>(p, s, flops, bytes) : split::launch<NW, BN, WM, WN, 6, false, false>(p, s, flops, bytes));
  static const bool big = getenv("MP_SPLIT_BIG") ? atoi(getenv("MP_SPLIT_BIG")) != 0 : false;  // 256x128 tile, 8 waves: measured equal to 128x128 (tuning knob)
  if (small) { MPS_GO(4, 64, 64, 32) }
  if (big && !dbuf) { MPS_GO(8, 128, 64, 64) }
  MPS_GO(4, 128, 64, 64)
#undef MPS_GO
}
