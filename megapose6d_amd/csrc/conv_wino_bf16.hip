// conv_wino_bf16.hip -- the fused Winograd F(2x2, 3x3) convolution with its 16 frequency-plane GEMMs on the bf16 MFMA through EXACT
// operand pieces (v_mfma_f32_32x32x16_bf16), gfx950.
//
// Same call sites and the same structure as conv_wino.hip (reference: src/megapose/models/torchvision_resnet.py:74-120 BasicBlock
// conv1 / conv2, src/megapose/models/wide_resnet.py:29-56 behind src/megapose/models/pose_rigid.py:323): workgroup = 64 tiles x 64
// output channels, wave w owns row w of the 4x4 frequency grid, input transform V = B^T d B computed in fp32 and handed over through
// the double-buffered fp32 LDS tile, output transform + fused epilogue in fp32.  What changes is the multiplication itself:
//   * U = G g G^T (fp32, eval-BN scale folded) is split on the host by truncation into three bf16 pieces U = U1 + U2 + U3 EXACTLY
//     (24 = 3 x 8 mantissa bits) and packed in MFMA fragment order;
//   * every V fragment (8 fp32 channels per lane, straight from LDS) is split the same way in registers, V = V1 + V2 + V3 (4 VALU per
//     element + the packing, issued in the shadow of the MFMAs);
//   * sum_c V U = sum_c sum_{i,j} V_i U_j: ALL nine piece products are evaluated -- each is a bf16 x bf16 product, exact in the
//     MFMA's fp32 accumulator -- so the result differs from the fp32-MFMA kernel only in the ORDER of fp32 additions (it is NOT a
//     reduced-precision mode: no product is rounded, none is dropped).
// Why: v_mfma_f32_32x32x16_bf16 retires 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32 -- nine piece products cost 9/16
// of one fp32 product.  Together with Winograd's 16/36 the matrix time is 0.25 of the direct fp32 convolution's.
// Per 16-channel step and wave: 16 ds_read_b128 (as before), 24 weight-fragment loads of 1 KB (L2 -> registers, one frequency point
// ahead), 4 x 36 MFMAs of 32 cycles.  The schedule is placed by hand (sched_barrier fences, one wave per SIMD issues in order): the other
// work of a wave -- the split of the NEXT point's fragments, the next step's input-transform planes, patch / weight requests, fragment
// reads -- sits in the gaps behind the MFMAs according to slot tables that scripts/gen_wino_schedule.py generates AND checks
// (conv_wino_bf16_sched.h: at most 5 instructions per gap, 7 in a gap with a ds_write -- what the gap hides, measured).
// Structure of a launch (round 6): PERSISTENT -- one workgroup per CU walks the 64-tile x 64-channel units; per unit: prologue (transform of
// step 0, first splits; its requests were issued under the previous unit's epilogue), K loop, a peeled LEAN last step whose free gaps
// carry the residual requests and the next unit's index arithmetic, epilogue (exchange through LDS with the next unit's patch requests
// behind the accumulator blocks, store loop with the next unit's accumulator reset behind its passes).
// Inline asm and hazards: hipcc does not pad hazards whose producer or consumer sits inside an asm statement -- scripts/isa_hazards.py
// lints the compiled ISA (tests/test_wino_isa_hazards_cpu.py, product + deliberately permuted builds); see the accumulator reset below.
// Roofline: bf16 MFMA; algorithmic work = 2 * MACs of the direct convolution; executed = 9 x 16/36 of that in bf16 FLOPs.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "wino_common.h"

namespace mp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (round 6: a packed-fp32 form of the transform / split arithmetic -- v_pk_add_f32, 653 instead of 781 instructions per step -- was built
//  and measured SLOWER, 7.5 k vs 6.4 k cycles per step: one v_pk_add_f32 in an MFMA gap costs 17 cycles, five v_sub_f32 cost 1.5;
//  profiles/r06_wino_kloop_experiments.txt.  v_dot2_f32_bf16 costs the same 17, a ds_write_b128 20, SALU beyond three per gap 8 each.)

// v_perm_b32 selector 0x07060302: {hi16(second arg) in the low half, hi16(first arg) in the high half}
__device__ __forceinline__ unsigned pack_hi16(unsigned e1, unsigned e0) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
// the nine piece pairs (V piece, U piece), small terms first
__device__ constexpr int WB_PA[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, WB_PB[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};

constexpr int UB_F_BYTES = 2 * 3 * 1024;          // one frequency point of one step: [cout block j][piece][lane][16 B]
constexpr int UB_STEP_BYTES = 16 * UB_F_BYTES;    // 96 KB per 16-channel step
// LDS: the two V stages (128 KB) + the tile table during the K loop; the exchange S[4 waves][2][64 tiles][pitch] in the epilogue (sized for a pitch of up to 72 floats: A/B builds)
constexpr size_t WB_LDS_BYTES = (size_t)4 * 2 * WT * 72 * sizeof(float);
static_assert(WB_LDS_BYTES >= WINO_LDS_BYTES && WB_LDS_BYTES <= 160 * 1024, "LDS budget of conv3x3_wino_bf16x9");

// Clock telemetry (as conv.hip's): every 64th workgroup adds the shader cycles (s_memtime), the 100 MHz real-time ticks (s_memrealtime)
// and the number of steps of its K loop: mp_conv_wino_bf16_clock reports the effective shader clock and the cycles per 16-channel step.
__device__ unsigned long long g_wb_clk[6];   // K-loop cycles, K-loop 100 MHz ticks, steps | prologue cycles, epilogue cycles, sampled workgroups
#ifdef MP_WINO_PHASES
// Profiling build only (scripts/microbench/build_wino_variants.sh phases): cycle stamps inside the prologue and the epilogue of every 64th
// workgroup, relative to the workgroup's start / the end of its K loop: [0..4] prologue (requests issued, first patch row transformed,
// V of step 0 written, barrier passed, first fragments split = loop entry), [5..8] epilogue (exchange written, barrier passed, stores
// issued, stores retired), [9] samples.
__device__ unsigned long long g_wb_phase[10];
#define WB_PHASE(K, T0) if (ph_sample) atomicAdd(&g_wb_phase[K], __builtin_readcyclecounter() - (T0));
#else
#define WB_PHASE(K, T0)
#endif

// DIAG (timing experiments only, wrong results; MP_WINO_DIAG): 1 = no split work, 2 = no patch requests / transform, 4 = no weight requests
// RES: the launch has a residual input; its 16 loads per thread ride under the MFMAs of the LAST K step (round 6).
// ACT: the launch writes the second, pre-activated output relu(y * scale + shift) (WideResNet blocks).  Both compile-time, and ReLU is a
// maximum with 0 or -inf: the store loop of the epilogue is straight-line code (it was 48 branches + their mask bookkeeping per workgroup).
// PERSIST (round 6, the default launch form; MP_WINO_PERSIST=0 = one workgroup per unit): one workgroup per CU walks the units blockIdx,
// blockIdx + gridDim, ...; the next unit's indices and tile table are computed before the exchange of the current unit's epilogue, its 16
// patch requests go out four at a time behind the four accumulator blocks of the exchange (the memory pipe is idle there, the patch registers
// are dead) and its accumulator reset four MFMAs at a time behind the passes of the store loop -- the next prologue finds its data arrived
// instead of waiting ~3 k cycles with nothing to overlap (prologue 6.6 k -> 3.2 k, epilogue 6.2 k -> 8.0 k cycles: r6 calls 11 - 14).
template <int DIAG, bool RES, bool ACT, bool PERSIST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wino_bf16x9(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;
  int* tile_tab = (int*)(smem + 4 * 2 * WT * WCOUT);   // [64][2]: output element offset of pixel (2ty, 2tx) (-1: no such tile), validity bits

  unsigned long long clk_start = __builtin_readcyclecounter();
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef MP_WINO_PHASES
  const bool ph_sample = (blockIdx.x & 63) == 0 && threadIdx.x == 0;
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, -1, 0x00020000);
  const int row_bytes = p.Wp * p.C * 4, pix_bytes = p.C * 4;
  const int u_voff = lane * 16;
  const int ptile = tid >> 2, pc4 = tid & 3;
  float4 patch[4][4];
  // the unit (= what a workgroup of the non-persistent launch does): cout block cb (fastest: the units that share an input tile set run
  // together) of tile group tile0 / 64
  int unit = (int)blockIdx.x;
  const int n_units = PERSIST ? p.n_units : (int)gridDim.x;
  int cb, tile0, x_voff;
  __amdgpu_buffer_rsrc_t u_rsrc;   // weights: [cb][step][f][j][piece][lane][8 bf16]; this wave's four frequency points of a step are 24 KB contiguous
#define WB_UNIT_INDICES(U)                                                                               \
  {                                                                                                      \
    const int wg_ = xcd_remap((U), n_units);                                                             \
    const int tg_ = (int)wino_fastdiv((unsigned)wg_, (unsigned)p.n_cblocks, p.mg_cb, p.sh_cb);           \
    cb = wg_ - tg_ * p.n_cblocks;                                                                        \
    tile0 = tg_ * WT;                                                                                    \
    const unsigned char* ub_ = reinterpret_cast<const unsigned char*>(p.u) + (size_t)cb * p.n_steps * UB_STEP_BYTES + (size_t)(4 * wave) * UB_F_BYTES;  \
    u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ub_, 0, -1, 0x00020000);                            \
    int t_ = tile0 + ptile;                                                                              \
    t_ = t_ < p.n_tiles ? t_ : p.n_tiles - 1;                                                            \
    const int r_ = (int)wino_fastdiv((unsigned)t_, (unsigned)p.tiles_x, p.mg_tx, p.sh_tx), tx_ = t_ - r_ * p.tiles_x;   \
    const int n_ = (int)wino_fastdiv((unsigned)r_, (unsigned)p.tiles_y, p.mg_ty, p.sh_ty), ty_ = r_ - n_ * p.tiles_y;   \
    const size_t pix_ = ((size_t)n_ * p.Hp + (size_t)(2 * ty_ + p.in_off)) * p.Wp + (size_t)(2 * tx_ + p.in_off);  \
    x_voff = (int)((pix_ * p.C + pc4 * 4) * sizeof(float));                                               \
  }
  // V[stage][f][tile][16 floats], 16-byte slot s of a row holds channels 4s..4s+3, slots XOR-swizzled by (tile >> 2) & 3 (as conv_wino.hip)
  float* vw = Vs + ptile * WCK + ((pc4 ^ ((ptile >> 2) & 3)) * 4);



  f32x16 acc[4][2][2];
  const int ns = p.n_steps;
  u32x4 Ua[2][3], Ub[2][3];   // weight fragments [cout block][piece] of the current / the next frequency point
  u32x4 AA[2][3], AB[2][3];   // V fragments [tile block][piece], likewise
  float4 raw[2][2];           // fp32 fragment of the point after next: [tile block][channels 0..3 | 4..7 of the lane's K group]
  unsigned sm0, sm1, sm2, sm3;            // the split in flight: masked values,
  float sr0, sr1, sr2, sr3, sq0, sq1, sq2, sq3;   // first and second remainders of the four elements
  float4 trw[4], tplane;      // row combination of the transform row in flight, the plane on its way to LDS
  if (DIAG & (7 | 32 | 64 | 128)) {   // (timing experiments: whatever the skipped work would have produced just has to be defined)
    _Pragma("unroll") for (int i = 0; i < 2; ++i)
      _Pragma("unroll") for (int q = 0; q < 3; ++q) { AA[i][q] = AB[i][q] = Ua[i][q] = Ub[i][q] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; }
    _Pragma("unroll") for (int a = 0; a < 4; ++a) { trw[a] = make_float4(0.f, 0.f, 0.f, 0.f);
      _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) patch[a][bb] = make_float4(1.f, 1.f, 1.f, 1.f); }
    tplane = make_float4(0.f, 0.f, 0.f, 0.f);
    sm0 = sm1 = sm2 = sm3 = 0u; sr0 = sr1 = sr2 = sr3 = sq0 = sq1 = sq2 = sq3 = 0.f;
  }
#define WB_SB __builtin_amdgcn_sched_barrier(0);
#define WB_LOAD_U1(DST, ST, FI, K) \
  if (!(DIAG & 4)) DST[(K) / 3][(K) % 3] = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff + ((K) & 3) * 1024, (ST) * UB_STEP_BYTES + (FI) * UB_F_BYTES + ((K) >> 2) * 4096, 0);   /* (the low part of the fragment offset travels in the instruction: two scalar adds per point instead of six) */
#define WB_LOAD_PATCH1(A, B, CS) if (!(DIAG & (2 | 128))) patch[A][B] = buf4(x_rsrc, x_voff, (CS) + (A) * row_bytes + (B) * pix_bytes);
  // transform row A of the patch in registers: T(A, b) forms the row combination (B^T d)[A][b]; O(A, col) one frequency plane -> LDS
#define WB_F4ASM(OP, D, X, Y)                                                                             \
  asm volatile(OP " %0, %4, %8\n\t" OP " %1, %5, %9\n\t" OP " %2, %6, %10\n\t" OP " %3, %7, %11"           \
               : "=&v"(D.x), "=&v"(D.y), "=&v"(D.z), "=&v"(D.w)                                            \
               : "v"(X.x), "v"(X.y), "v"(X.z), "v"(X.w), "v"(Y.x), "v"(Y.y), "v"(Y.z), "v"(Y.w));
#define WB_OP_SUB "v_sub_f32"
#define WB_OP_ADD "v_add_f32"
#define WB_TR_T(A, B)                                                                                    \
  if (DIAG & (2 | 64)) {} else if ((A) == 0) { WB_F4ASM(WB_OP_SUB, trw[B], patch[0][B], patch[2][B]) }                              \
  else if ((A) == 1) { WB_F4ASM(WB_OP_ADD, trw[B], patch[1][B], patch[2][B]) }                         \
  else if ((A) == 2) { WB_F4ASM(WB_OP_SUB, trw[B], patch[2][B], patch[1][B]) }                         \
  else { WB_F4ASM(WB_OP_SUB, trw[B], patch[1][B], patch[3][B]) }
  // Exact truncation split of the four elements of raw[I][H] (elements 4H .. 4H+3 of fragment I) into the three pieces of AN, spread over
  // four slots so that no instruction of a slot depends on another one of the same slot (one wave per SIMD issues in order: a dependent
  // VALU pair stalls for the pipeline latency, and that stall comes straight out of the MFMA shadow):
  //   S0: m = v & hi16, piece-1 pairs (v_perm of the unmasked v)      S1: r = v - m
  //   S2: n = r & hi16, piece-2 pairs (v_perm of the unmasked r)      S3: q = r - n      (S4, in the next block's S0: piece-3 pairs of q)
  // (inline asm: the compiler's machine-sink pass otherwise moves every `and` next to the `sub` that consumes it -- sched_barrier fences
  //  only the scheduler -- and the dependent pairs are back)
#define WB_SP0(AN, I, H)                                                                                 \
  if (!(DIAG & 1)) {                                                                                                     \
    unsigned p0_, p1_;                                                                                  \
    asm volatile("v_and_b32 %0, 0xffff0000, %6\n\tv_and_b32 %1, 0xffff0000, %7\n\tv_and_b32 %2, 0xffff0000, %8\n\t"                \
                 "v_and_b32 %3, 0xffff0000, %9\n\tv_perm_b32 %4, %7, %6, %10\n\tv_perm_b32 %5, %9, %8, %10"                       \
                 : "=&v"(sm0), "=&v"(sm1), "=&v"(sm2), "=&v"(sm3), "=&v"(p0_), "=&v"(p1_)                                         \
                 : "v"(raw[I][H].x), "v"(raw[I][H].y), "v"(raw[I][H].z), "v"(raw[I][H].w), "s"(0x07060302u));                     \
    AN[I][0][2 * (H)] = p0_; AN[I][0][2 * (H) + 1] = p1_;                                               \
  }
#define WB_SP1(I, H)                                                                                     \
  if (!(DIAG & 1)) asm volatile("v_sub_f32 %0, %4, %8\n\tv_sub_f32 %1, %5, %9\n\tv_sub_f32 %2, %6, %10\n\tv_sub_f32 %3, %7, %11"                    \
               : "=&v"(sr0), "=&v"(sr1), "=&v"(sr2), "=&v"(sr3)                                                                   \
               : "v"(raw[I][H].x), "v"(raw[I][H].y), "v"(raw[I][H].z), "v"(raw[I][H].w), "v"(sm0), "v"(sm1), "v"(sm2), "v"(sm3));
#define WB_SP2(AN, I, H)                                                                                 \
  if (!(DIAG & 1)) {                                                                                                     \
    unsigned p0_, p1_;                                                                                  \
    asm volatile("v_and_b32 %0, 0xffff0000, %6\n\tv_and_b32 %1, 0xffff0000, %7\n\tv_and_b32 %2, 0xffff0000, %8\n\t"                \
                 "v_and_b32 %3, 0xffff0000, %9\n\tv_perm_b32 %4, %7, %6, %10\n\tv_perm_b32 %5, %9, %8, %10"                       \
                 : "=&v"(sm0), "=&v"(sm1), "=&v"(sm2), "=&v"(sm3), "=&v"(p0_), "=&v"(p1_)                                         \
                 : "v"(sr0), "v"(sr1), "v"(sr2), "v"(sr3), "s"(0x07060302u));                                                     \
    AN[I][1][2 * (H)] = p0_; AN[I][1][2 * (H) + 1] = p1_;                                               \
  }
#define WB_SP3()                                                                                         \
  if (!(DIAG & 1)) asm volatile("v_sub_f32 %0, %4, %8\n\tv_sub_f32 %1, %5, %9\n\tv_sub_f32 %2, %6, %10\n\tv_sub_f32 %3, %7, %11"                    \
               : "=&v"(sq0), "=&v"(sq1), "=&v"(sq2), "=&v"(sq3)                                                                   \
               : "v"(sr0), "v"(sr1), "v"(sr2), "v"(sr3), "v"(sm0), "v"(sm1), "v"(sm2), "v"(sm3));
#define WB_SP4(AN, I, H)                                                                                 \
  if (!(DIAG & 1)) {                                                                                                     \
    unsigned p0_, p1_;                                                                                  \
    asm volatile("v_perm_b32 %0, %3, %2, %6\n\tv_perm_b32 %1, %5, %4, %6"                                \
                 : "=&v"(p0_), "=&v"(p1_) : "v"(sq0), "v"(sq1), "v"(sq2), "v"(sq3), "s"(0x07060302u));   \
    AN[I][2][2 * (H)] = p0_; AN[I][2][2 * (H) + 1] = p1_;                                               \
  }
#define WB_READ_RAW1(VB, FI, K) \
  raw[(K) >> 1][(K) & 1] = *reinterpret_cast<const float4*>((VB) + (FI) * (WT * WCK) + ((K) >> 1) * (32 * WCK) + (((K) & 1) ? fo_hi : fo_lo));
  // MFMA of slot S (0..35): piece pair S / 4 (small terms first), accumulator (tile block, cout block) = S % 4
#define WB_M(S, AC, UC)                                                                                  \
  acc[fi][((S) & 3) >> 1][(S) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                            \
      __builtin_bit_cast(bf16x8, AC[((S) & 3) >> 1][WB_PA[(S) >> 2]]), __builtin_bit_cast(bf16x8, UC[(S) & 1][WB_PB[(S) >> 2]]), \
      acc[fi][((S) & 3) >> 1][(S) & 1], 0, 0, 0);
  // one frequency plane of the transform row in flight -> tplane (written to LDS one slot later: the store does not wait for its data)
#define WB_TR_P(COL)                                                                                     \
  if (DIAG & (2 | 64)) {} else if ((COL) == 0) { WB_F4ASM(WB_OP_SUB, tplane, trw[0], trw[2]) }                                      \
  else if ((COL) == 1) { WB_F4ASM(WB_OP_ADD, tplane, trw[1], trw[2]) }                                 \
  else if ((COL) == 2) { WB_F4ASM(WB_OP_SUB, tplane, trw[2], trw[1]) }                                 \
  else { WB_F4ASM(WB_OP_SUB, tplane, trw[1], trw[3]) }
#define WB_TR_W(A, COL, VW) if (!(DIAG & (2 | 32))) *reinterpret_cast<float4*>((VW) + ((A) * 4 + (COL)) * (WT * WCK)) = tplane;
  // ---- the LAST K step (peeled, round 6): no next step to prepare -- no transform, no patch requests, no weights / fragments for a
  //      following point -- so its free slots carry the epilogue's residual loads instead (16 per thread, into the registers of the patch)
#define WB_LOAD_RES1(K) if (RES) res[(K) >> 2][(K) & 3] = __builtin_amdgcn_raw_buffer_load_b128(r_res, voff[(K) >> 2][(K) & 3], 0, 0);
  // The split as single 4-instruction groups and single v_perm instructions (round 6): the slot tables place ONE group per MFMA gap
  // (conv_wino_bf16_sched.h, generated and checked by scripts/gen_wino_schedule.py; the prologue still uses the fused WB_SP0..WB_SP4).
  //   A: m = v & hi16    B (= WB_SP1): r = v - m    C: n = r & hi16    D (= WB_SP3): q = r - n
  //   PRM_A / PRM_B / PRM_C: piece-1 / -2 / -3 pair j of the four elements = v_perm of (v | r | q)[2j + 1], [2j]
#define WB_SPA(I, H)                                                                                     \
  if (!(DIAG & 1)) asm volatile("v_and_b32 %0, 0xffff0000, %4\n\tv_and_b32 %1, 0xffff0000, %5\n\tv_and_b32 %2, 0xffff0000, %6\n\tv_and_b32 %3, 0xffff0000, %7"  \
                                : "=&v"(sm0), "=&v"(sm1), "=&v"(sm2), "=&v"(sm3)                           \
                                : "v"(raw[I][H].x), "v"(raw[I][H].y), "v"(raw[I][H].z), "v"(raw[I][H].w));
#define WB_SPC()                                                                                         \
  if (!(DIAG & 1)) asm volatile("v_and_b32 %0, 0xffff0000, %4\n\tv_and_b32 %1, 0xffff0000, %5\n\tv_and_b32 %2, 0xffff0000, %6\n\tv_and_b32 %3, 0xffff0000, %7"  \
                                : "=&v"(sm0), "=&v"(sm1), "=&v"(sm2), "=&v"(sm3) : "v"(sr0), "v"(sr1), "v"(sr2), "v"(sr3));
#define WB_PRM1(DST, E1, E0)                                                                             \
  if (!(DIAG & 1)) { unsigned p_; asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p_) : "v"(E1), "v"(E0), "s"(0x07060302u)); DST = p_; }
#define WB_PRM_A(AN, I, H, J) WB_PRM1(AN[I][0][2 * (H) + (J)], ((J) ? raw[I][H].w : raw[I][H].y), ((J) ? raw[I][H].z : raw[I][H].x))
#define WB_PRM_B(AN, I, H, J) WB_PRM1(AN[I][1][2 * (H) + (J)], ((J) ? sr3 : sr1), ((J) ? sr2 : sr0))
#define WB_PRM_C(AN, I, H, J) WB_PRM1(AN[I][2][2 * (H) + (J)], ((J) ? sq3 : sq1), ((J) ? sq2 : sq0))
#include "conv_wino_bf16_sched.h"

  // point 3 of the last step: nothing follows it -- 36 MFMAs, residual loads 8..15 in the first gaps
#define WB_POINT_LAST(AC, UC)                                                                            \
  WB_M(0, AC, UC) WB_SB WB_LOAD_RES1(8) WB_SB                                                            \
  WB_M(1, AC, UC) WB_SB WB_LOAD_RES1(9) WB_SB                                                            \
  WB_M(2, AC, UC) WB_SB WB_LOAD_RES1(10) WB_SB                                                           \
  WB_M(3, AC, UC) WB_SB WB_LOAD_RES1(11) WB_SB                                                           \
  WB_M(4, AC, UC) WB_SB WB_LOAD_RES1(12) WB_SB                                                           \
  WB_M(5, AC, UC) WB_SB WB_LOAD_RES1(13) WB_SB                                                           \
  WB_M(6, AC, UC) WB_SB WB_LOAD_RES1(14) WB_SB                                                           \
  WB_M(7, AC, UC) WB_SB WB_LOAD_RES1(15) WB_SB                                                           \
  WB_M(8, AC, UC) WB_M(9, AC, UC) WB_M(10, AC, UC) WB_M(11, AC, UC) WB_NEXT_INDICES() WB_M(12, AC, UC) WB_M(13, AC, UC) WB_M(14, AC, UC)  \
  WB_M(15, AC, UC) WB_M(16, AC, UC) WB_M(17, AC, UC) WB_M(18, AC, UC) WB_M(19, AC, UC) WB_M(20, AC, UC) WB_M(21, AC, UC) \
  WB_M(22, AC, UC) WB_M(23, AC, UC) WB_M(24, AC, UC) WB_M(25, AC, UC) WB_M(26, AC, UC) WB_M(27, AC, UC) WB_M(28, AC, UC) \
  WB_M(29, AC, UC) WB_M(30, AC, UC) WB_M(31, AC, UC) WB_M(32, AC, UC) WB_M(33, AC, UC) WB_M(34, AC, UC) WB_M(35, AC, UC) WB_SB

  // fragment read position: lane (tile row = lane & 31 (+ 32), K group = lane >> 5 = channels 8h .. 8h+7 = 16-byte slots 2h, 2h+1)
  const int fsw = ((lane & 31) >> 2) & 3;
  const float* vr = Vs + ((4 * wave) * WT + (lane & 31)) * WCK;
  const int fo_lo = ((2 * (lane >> 5)) ^ fsw) * 4, fo_hi = ((2 * (lane >> 5) + 1) ^ fsw) * 4;

  // ---- prologue: V of step 0 in stage 0, the patch of step 1 in registers, weights + split fragments of (step 0, point 0), the raw
  //      fragments of point 1 ---------------------------------------------------------------------------------------------------------
  // Request order (round 6; stamps in profiles/r06_wino_kloop_experiments.txt): a wave's vector-memory requests are accepted at the rate
  // they come back (~16 outstanding per wave, ~2 k cycles each from HBM), so the 38 requests of the round-4/5 prologue took 6 k cycles to
  // ISSUE, with everything behind them in program order waiting.  Now: the 16 patch requests of step 0 first, then the accumulator reset
  // and the tile table (matrix pipe / LDS work in the shadow of the queue), the 6 weight requests, and the 16 requests of step 1's patch
  // only after the first transform row -- they fly under the remaining transform rows, the barrier and the first splits.
  float4 pnext[4][4];
  // the first requests of a unit: 16 patch loads of step 0, then (in the shadow of the queue) the tile table of the epilogue and the
  // accumulator reset by the matrix pipe itself (0 x 0 + 0: 16 instructions instead of 256 register writes; asm volatile: as a builtin the 16
  // identical products are merged into one and copied), then the 6 weight loads of (step 0, point 0).
  // `s_nop 1` INSIDE the reset's string: the compiler materialises the zero operand with v_mov right in front of the statement, and a VALU
  // write of a register an MFMA reads as A / B needs two wait states that the hazard recogniser does not insert for a consumer inside inline
  // asm.  Without them the first MFMA multiplies whatever the registers held BEFORE the v_mov -- rounds 4 / 5 shipped that way and were
  // right only because the allocator happened to pick registers that held small integers (bf16 pairs whose products underflow to 0); any
  // edit that moved the operand to never-written registers (stale data of the previous kernel) gave inf / NaN on every shape: DESIGN.md
  // 3.1.1, round 6.
#define WB_UNIT_REQUESTS() \
  _Pragma("unroll") for (int a = 0; a < 4; ++a) \
    _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) { WB_LOAD_PATCH1(a, bb, 0) } \
  WB_SB \
  if (tid < WT) { \
    const int t = tile0 + tid; \
    int off = -1, bits = 0; \
    if (t < p.n_tiles) { \
      const int r = (int)wino_fastdiv((unsigned)t, (unsigned)p.tiles_x, p.mg_tx, p.sh_tx), tx = t - r * p.tiles_x; \
      const int n = (int)wino_fastdiv((unsigned)r, (unsigned)p.tiles_y, p.mg_ty, p.sh_ty), ty = r - n * p.tiles_y; \
      off = ((n * p.Hop + 2 * ty + p.out_border) * p.Wop + 2 * tx + p.out_border) * p.Cout; \
      bits = ((2 * ty + 1 < p.Ho) ? 1 : 0) | ((2 * tx + 1 < p.Wo) ? 2 : 0); \
    } \
    tile_tab[2 * tid] = off; \
    tile_tab[2 * tid + 1] = bits; \
  } \
  { \
    const u32x4 z4 = {0u, 0u, 0u, 0u}; \
    _Pragma("unroll") for (int fi = 0; fi < 4; ++fi) \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) \
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[fi][i][j]) : "v"(z4)); \
  } \
  WB_SB
  // the tile table alone (persistent form: patch loads and accumulator reset are spread over the store loop of the previous unit)
#define WB_UNIT_TABLE() \
  if (tid < WT) { \
    const int t = tile0 + tid; \
    int off = -1, bits = 0; \
    if (t < p.n_tiles) { \
      const int r = (int)wino_fastdiv((unsigned)t, (unsigned)p.tiles_x, p.mg_tx, p.sh_tx), tx = t - r * p.tiles_x; \
      const int n = (int)wino_fastdiv((unsigned)r, (unsigned)p.tiles_y, p.mg_ty, p.sh_ty), ty = r - n * p.tiles_y; \
      off = ((n * p.Hop + 2 * ty + p.out_border) * p.Wop + 2 * tx + p.out_border) * p.Cout; \
      bits = ((2 * ty + 1 < p.Ho) ? 1 : 0) | ((2 * tx + 1 < p.Wo) ? 2 : 0); \
    } \
    tile_tab[2 * tid] = off; \
    tile_tab[2 * tid + 1] = bits; \
  }
  WB_UNIT_INDICES(unit)
  WB_UNIT_REQUESTS()
  for (;;) {   // (one pass unless PERSIST)
  _Pragma("unroll") for (int k = 0; k < 6; ++k) { WB_LOAD_U1(Ua, 0, 0, k) }   // the 6 weight loads of (step 0, point 0)
  WB_SB
  WB_PHASE(0, clk_start)
#define WB_TR_ROW(A, VW) WB_TR_T(A, 0) WB_TR_T(A, 1) WB_TR_T(A, 2) WB_TR_T(A, 3) WB_TR_P(0) WB_TR_W(A, 0, VW) WB_TR_P(1) WB_TR_W(A, 1, VW) WB_TR_P(2) WB_TR_W(A, 2, VW) WB_TR_P(3) WB_TR_W(A, 3, VW)
  WB_TR_ROW(0, vw) WB_PHASE(1, clk_start)
  WB_SB
  // step 1's 16 patch requests, one behind every second statement of transform rows 1..3 (each occupies the address unit for 16 cycles x 4
  // waves: back to back they stalled the wave ~600 cycles)
  const int cs1 = (ns > 1 ? 1 : 0) * (WCK * 4);
#define WB_PN(K) WB_SB pnext[(K) >> 2][(K) & 3] = buf4(x_rsrc, x_voff, cs1 + ((K) >> 2) * row_bytes + ((K) & 3) * pix_bytes); WB_SB
#define WB_TR_ROW_PN(A, VW, K0)                                                                          \
  WB_TR_T(A, 0) WB_TR_T(A, 1) WB_PN((K0) + 0) WB_TR_T(A, 2) WB_TR_T(A, 3) WB_PN((K0) + 1) WB_TR_P(0) WB_TR_W(A, 0, VW) WB_PN((K0) + 2)  \
  WB_TR_P(1) WB_TR_W(A, 1, VW) WB_PN((K0) + 3) WB_TR_P(2) WB_TR_W(A, 2, VW) WB_PN((K0) + 4) WB_TR_P(3) WB_TR_W(A, 3, VW)
  WB_TR_ROW_PN(1, vw, 0) WB_PN(5) WB_TR_ROW_PN(2, vw, 6) WB_TR_ROW_PN(3, vw, 11)
#undef WB_TR_ROW_PN
#undef WB_PN
  WB_PHASE(2, clk_start)
  _Pragma("unroll") for (int a = 0; a < 4; ++a)
    _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) patch[a][bb] = pnext[a][bb];
  __syncthreads();
  WB_PHASE(3, clk_start)
  // the epilogue's store offsets, computed HERE (round 6): the tile table is visible after the barrier, its four reads fly under the
  // splits below, and the offsets are what the last K step needs to request the residual under its MFMAs.  Branch-free: a tile or a
  // pixel that does not exist gets the out-of-range offset (buffer accesses are range-checked: loads return 0, stores are dropped).
  constexpr unsigned WOOB = 0xFFFFFFF0u;
  int2 ttab[4];
  _Pragma("unroll") for (int it = 0; it < 4; ++it) ttab[it] = *reinterpret_cast<const int2*>(tile_tab + 2 * (it * 16 + (tid >> 4)));
  WB_READ_RAW1(vr, 0, 0) WB_READ_RAW1(vr, 0, 1) WB_READ_RAW1(vr, 0, 2) WB_READ_RAW1(vr, 0, 3)
#define WB_SPLIT4(AN, I, H) WB_SP0(AN, I, H) WB_SP1(I, H) WB_SP2(AN, I, H) WB_SP3() WB_SP4(AN, I, H)
  WB_SPLIT4(AA, 0, 0) WB_SPLIT4(AA, 0, 1) WB_SPLIT4(AA, 1, 0) WB_SPLIT4(AA, 1, 1)
  WB_READ_RAW1(vr, 1, 0) WB_READ_RAW1(vr, 1, 1) WB_READ_RAW1(vr, 1, 2) WB_READ_RAW1(vr, 1, 3)
  const int n = cb * WCOUT + (tid & 15) * 4;
  unsigned voff[4][4];
  {
    const int e_row = p.Wop * p.Cout * 4, e_pix = p.Cout * 4;
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {
      const int off = ttab[it].x, bits = ttab[it].y;
      const unsigned base = (unsigned)(off + n) * 4u;
      const bool t_ok = off >= 0;
      voff[it][0] = t_ok ? base : WOOB;
      voff[it][1] = (t_ok && (bits & 2)) ? base + (unsigned)e_pix : WOOB;
      voff[it][2] = (t_ok && (bits & 1)) ? base + (unsigned)e_row : WOOB;
      voff[it][3] = (t_ok && (bits & 3) == 3) ? base + (unsigned)(e_row + e_pix) : WOOB;
    }
  }
  const int out_bytes = p.out_bytes;
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, (RES && p.residual) ? out_bytes : 0, 0x00020000);
  u32x4 res[4][4];

#ifdef MP_WINO_PERMUTE
  // TEST BUILD ONLY (tests/test_gpu_wino_permuted.py): MP_WINO_PERMUTE extra values are kept live across the K loop (read from the bias
  // vector before it, folded into a never-taken store after it), so that the register allocator lays the loop's vector registers out
  // differently from the product build -- same opcodes, another assignment.  The kernel must not care: the parity tests run on this build too.
  u32x4 perm_live[MP_WINO_PERMUTE];
  {
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 4096, 0x00020000);
    _Pragma("unroll") for (int k = 0; k < MP_WINO_PERMUTE; ++k) perm_live[k] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, lane * 16, k * 1024, 0);
  }
#endif
  // ---- K loop: 16 input channels per step, four frequency points per wave and step, 36 MFMAs per point.  Point f multiplies the
  //      fragments split during point f-1 (raw fp32 read during point f-2) with the weights requested during point f-1.  ONE barrier per
  //      step, between points 1 and 2: V of this step is last read in point 1 (for point 3), V of the next step is complete after point 1
  //      (its transform rides in points 0 and 1) and first read in point 2.
  WB_PHASE(4, clk_start)
  const bool clk_sample = p.telemetry != 0 && (unit & 63) == 0 && tid == 0;
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk_sample) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_wb_clk[3], clk_c0 - clk_start); }
  for (int st = 0; st < ns - 1; ++st) {
    const int buf = st & 1;
    const float* vb = vr + buf * WV_STAGE;          // V of this step
    const float* vbn = vr + (buf ^ 1) * WV_STAGE;   // V of the next step
    float* vwn = vw + (buf ^ 1) * WV_STAGE;
    const int cs_patch = (st + 2 < ns ? st + 2 : ns - 1) * (WCK * 4);   // (the step before the last re-requests the last patch: unused)
#ifdef MP_WINO_PERMUTE
    _Pragma("unroll") for (int k = 0; k < MP_WINO_PERMUTE; ++k) asm volatile("" : "+v"(perm_live[k]));   // (in registers, not in scratch)
#endif
    { constexpr int fi = 0; WB_BLK_TR(AA, AB, Ua, Ub, st, 1, 0, 1, vwn, vb, 2) }
    { constexpr int fi = 1; WB_BLK_TR(AB, AA, Ub, Ua, st, 2, 2, 3, vwn, vb, 3) }
    __syncthreads();
#ifdef MP_CONV_EXPERIMENTS
    if (DIAG & 8) { for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(1); }    // skew the four waves by 64 cycles each
    if (DIAG & 16) { for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(2); }   // ... by 128 cycles each
#endif
    { constexpr int fi = 2; WB_BLK_PL(AA, AB, Ua, Ub, st, 3, 0, 1, cs_patch, vbn, 0) }
    { constexpr int fi = 3; WB_BLK_PL(AB, AA, Ub, Ua, st + 1, 0, 2, 3, cs_patch, vbn, 1) }
  }
  const int next_unit = unit + (int)gridDim.x;
  const bool more = PERSIST && next_unit < n_units;
  {   // the last step: nothing to prepare for a following one (see WB_TAIL_RD / WB_TAIL_RES / WB_POINT_LAST)
    const int st = ns - 1;
    const float* vb = vr + (st & 1) * WV_STAGE;
#ifdef MP_WINO_PERMUTE
    _Pragma("unroll") for (int k = 0; k < MP_WINO_PERMUTE; ++k) asm volatile("" : "+v"(perm_live[k]));
#endif
    { constexpr int fi = 0; WB_BLK_RD(AA, AB, Ua, Ub, st, 1, vb, 2) }
    { constexpr int fi = 1; WB_BLK_RD(AB, AA, Ub, Ua, st, 2, vb, 3) }
    { constexpr int fi = 2; WB_BLK_RES(AA, AB, Ua, Ub, st, 3) }
    // persistent form: the next unit's index arithmetic (~60 instructions) inside the unfenced run of MFMAs of the last point, for the
    // scheduler to spread under them (this unit needs cb / tile0 / x_voff / u_rsrc no more: its last weight request is in point 2)
#define WB_NEXT_INDICES() if constexpr (PERSIST) { WB_UNIT_INDICES(next_unit) }   /* (unconditional: a branch would end the scheduling region; unused after the last unit) */
    { constexpr int fi = 3; WB_POINT_LAST(AB, Ub) }
#undef WB_NEXT_INDICES
  }
  if (clk_sample) {
    atomicAdd(&g_wb_clk[0], __builtin_readcyclecounter() - clk_c0);
    atomicAdd(&g_wb_clk[1], __builtin_amdgcn_s_memrealtime() - clk_r0);
    atomicAdd(&g_wb_clk[2], (unsigned long long)ns);
  }
  const unsigned long long clk_epi = __builtin_readcyclecounter();
#ifdef MP_WINO_PERMUTE
  {
    unsigned fold = 0;
    _Pragma("unroll") for (int k = 0; k < MP_WINO_PERMUTE; ++k) fold |= perm_live[k].x ^ perm_live[k].y ^ perm_live[k].z ^ perm_live[k].w;
    if (p.n_steps < 0) tile_tab[tid & 127] = (int)fold;   // never taken (n_steps >= 1): keeps the values alive to here
  }
#endif
  __syncthreads();   // (the epilogue reuses the V stages)

  // ---- epilogue: output transform through LDS, bias + residual + ReLU (+ second activated output).  The store offsets and the residual
  //      are already in registers (computed after the prologue's barrier / requested under the last K step). -------------------------------
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + n);
  if (ACT && p.y_act) {   // (p.y_act is set whenever ACT is, except in the timing-experiment instances)
    sc = *reinterpret_cast<const float4*>(p.act_scale + n);
    sh = *reinterpret_cast<const float4*>(p.act_shift + n);
  }
  // Persistent form: the NEXT unit's indices and tile table now (the table is dead since the prologue), its 16 patch requests four at a time
  // behind each of the four accumulator blocks of the exchange below (rows 0, 2, 1, 3: the order the first transform rows need them; the
  // memory pipe is idle during the exchange) and its accumulator reset four MFMAs at a time behind the passes of the store loop.  All 16
  // requests at once in front of the stores filled the wave's request queue and every store waited an HBM round trip (r6 calls 11 - 13).
  if (more) {   // (indices: computed under the MFMAs of the last K step)
    WB_UNIT_TABLE()
  }
  // S[wave][2][tile][WS]
#ifndef MP_WINO_WS
#define MP_WINO_WS 64
#endif
  constexpr int WS = MP_WINO_WS;   // (a pitch of 72 floats -- lanes 32..63 of a fragment 32 banks away from lanes 0..31 -- measured 130 cycles SLOWER per workgroup: r6 call 10)
  float* S = smem;
  {
    float* sw = S + (size_t)wave * (2 * WT * WS) + ((lane >> 5) * 4) * WS + (lane & 31);
    // one (tile block, cout block) at a time: 64 accumulators leave the matrix registers, become 32 values and go to LDS before the next
    // 64 are touched (as one block the compiler copies all 256 out first and spills)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (the empty statements pin these four accumulators to the matrix registers up to HERE: the allocator cannot split their live
        //  ranges at the loop exit and copy all 256 to vector registers at once)
        asm volatile("" : "+a"(acc[0][i][j]), "+a"(acc[1][i][j]), "+a"(acc[2][i][j]), "+a"(acc[3][i][j]));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][i][j][r], m1 = acc[1][i][j][r], m2 = acc[2][i][j][r], m3 = acc[3][i][j][r];
          const int row = i * 32 + (r & 3) + 8 * (r >> 2);
          sw[row * WS + j * 32] = (m0 + m1) + m2;
          sw[WT * WS + row * WS + j * 32] = (m1 - m2) - m3;
        }
        if (more) {   // the next unit's patch requests, four behind each accumulator block (spreading them further makes the patch registers
          constexpr int prow[4] = {0, 2, 1, 3};   // live early enough for the allocator to spill 13 loop invariants, whose reloads then queue
          __builtin_amdgcn_sched_barrier(0);      // behind these very requests)
          _Pragma("unroll") for (int bb = 0; bb < 4; ++bb) { WB_LOAD_PATCH1(prow[i * 2 + j], bb, 0) }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  WB_PHASE(5, clk_epi)
  __syncthreads();
  WB_PHASE(6, clk_epi)
  // (round 6: an L2 prefetch of the following workgroup's first patch lines from here -- 4 dword loads per thread, results unused -- was
  //  built and measured: that workgroup's prologue gets 1.1 k cycles shorter, this epilogue 1.5 - 2.9 k cycles longer (the scattered requests
  //  queue in front of the stores): a net loss on every layer, removed.  profiles/r06_wino_kloop_experiments.txt)
  const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y ? out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)p.y_act, 0, (ACT && p.y_act) ? out_bytes : 0, 0x00020000);
  const float relu_floor = p.relu != 0 ? 0.f : -__builtin_inff();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int tl = it * 16 + (tid >> 4);
    float4 s4[4][2];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) s4[w][jj] = *reinterpret_cast<const float4*>(S + ((size_t)(w * 2 + jj) * WT + tl) * WS + (tid & 15) * 4);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float4 v;
        if (ii == 0) {
          v.x = (s4[0][jj].x + s4[1][jj].x) + s4[2][jj].x; v.y = (s4[0][jj].y + s4[1][jj].y) + s4[2][jj].y;
          v.z = (s4[0][jj].z + s4[1][jj].z) + s4[2][jj].z; v.w = (s4[0][jj].w + s4[1][jj].w) + s4[2][jj].w;
        } else {
          v.x = (s4[1][jj].x - s4[2][jj].x) - s4[3][jj].x; v.y = (s4[1][jj].y - s4[2][jj].y) - s4[3][jj].y;
          v.z = (s4[1][jj].z - s4[2][jj].z) - s4[3][jj].z; v.w = (s4[1][jj].w - s4[2][jj].w) - s4[3][jj].w;
        }
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (RES) {
          const u32x4 rr = res[it][ii * 2 + jj];
          v.x += __uint_as_float(rr.x); v.y += __uint_as_float(rr.y); v.z += __uint_as_float(rr.z); v.w += __uint_as_float(rr.w);
        }
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
        u32x4 o;
        o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(o, r_y, voff[it][ii * 2 + jj], 0, 0);
        if (ACT) {
          u32x4 a;
          a.x = __float_as_uint(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f)); a.y = __float_as_uint(fmaxf(fmaf(v.y, sc.y, sh.y), 0.f));
          a.z = __float_as_uint(fmaxf(fmaf(v.z, sc.z, sh.z), 0.f)); a.w = __float_as_uint(fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
          __builtin_amdgcn_raw_buffer_store_b128(a, r_act, voff[it][ii * 2 + jj], 0, 0);
        }
      }
    if (more) {
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      _Pragma("unroll") for (int bb = 0; bb < 4; ++bb)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[it][bb >> 1][bb & 1]) : "v"(z4));
    }
  }
#ifdef MP_WINO_PHASES
  WB_PHASE(7, clk_epi)
  if (ph_sample) { __builtin_amdgcn_s_waitcnt(0); WB_PHASE(8, clk_epi) atomicAdd(&g_wb_phase[9], 1ull); }
#endif
  if (clk_sample) {
    if (!more) __builtin_amdgcn_s_waitcnt(0);   // the stores are out (persistent form: the next unit's requests are in flight, do not wait for them)
    atomicAdd(&g_wb_clk[4], __builtin_readcyclecounter() - clk_epi);
    atomicAdd(&g_wb_clk[5], 1ull);
  }
  if (!more) break;
  unit = next_unit;
  clk_start = __builtin_readcyclecounter();
  __syncthreads();   // every wave has read the exchange: the next unit's transform may overwrite it
  }   // units
#undef WB_POINT_LAST
#undef WB_BLK_RES
#undef WB_BLK_RD
#undef WB_BLK_PL
#undef WB_BLK_TR
#undef WB_PRM_C
#undef WB_PRM_B
#undef WB_PRM_A
#undef WB_PRM1
#undef WB_SPC
#undef WB_SPA
#undef WB_LOAD_RES1
#undef WB_M
#undef WB_READ_RAW1
#undef WB_SPLIT4
#undef WB_SP4
#undef WB_SP3
#undef WB_SP2
#undef WB_SP1
#undef WB_SP0
#undef WB_TR_W
#undef WB_TR_P
#undef WB_TR_ROW
#undef WB_TR_T
#undef WB_F4ASM
#undef WB_OP_SUB
#undef WB_OP_ADD
#undef WB_LOAD_PATCH1
#undef WB_LOAD_U1
#undef WB_SB
#undef WB_UNIT_TABLE
#undef WB_UNIT_REQUESTS
#undef WB_UNIT_INDICES
}

}  // namespace mp

using namespace mp;

extern "C" size_t mp_conv_wino_bf16_packed_bytes(int Cin_p, int Cout) { return (size_t)(Cout / WCOUT) * (Cin_p / WCK) * UB_STEP_BYTES; }

// U = G g G^T per (cout, cin) in double, rounded once to fp32, split into three bf16 pieces, in MFMA fragment order:
// packed[cb][step][f][j][piece][lane][e] = piece of U_f[cin = step*16 + (lane >> 5)*8 + e][cout = cb*64 + j*32 + (lane & 31)]
extern "C" int mp_conv_wino_bf16_pack_weights(const float* w, int Cout, int Cin, int Cin_p, const float* scale, void* packed_bytes) {
  MP_REQUIRE(w && packed_bytes && Cin_p >= Cin && Cin_p % WCK == 0 && Cout % WCOUT == 0,
             "mp_conv_wino_bf16_pack_weights: bad arguments (Cin_p %% 16, Cout %% 64)");
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int n_steps = Cin_p / WCK;
  unsigned short* out = (unsigned short*)packed_bytes;
  memset(out, 0, mp_conv_wino_bf16_packed_bytes(Cin_p, Cout));
  for (int n = 0; n < Cout; ++n) {
    const double s = scale ? (double)scale[n] : 1.0;
    const int cb = n / WCOUT, j = (n % WCOUT) / 32, nl = n % 32;
    for (int c = 0; c < Cin; ++c) {
      const float* g = w + ((size_t)n * Cin + c) * 9;
      double t[4][3], U[4][4];
      for (int a = 0; a < 4; ++a)
        for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] * s + G[a][1] * g[1 * 3 + k] * s + G[a][2] * g[2 * 3 + k] * s;
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) U[a][b] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
      const int st = c / WCK, kh = (c % WCK) / 8, e = c % 8;
      const int lane = kh * 32 + nl;
      for (int f = 0; f < 16; ++f) {
        const float v = (float)U[f / 4][f % 4];
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
        const unsigned short pc[3] = {(unsigned short)(h >> 16), (unsigned short)(m >> 16), (unsigned short)(qb >> 16)};
        for (int piece = 0; piece < 3; ++piece)
          out[(((((size_t)(cb * n_steps + st) * 16 + f) * 2 + j) * 3 + piece) * 64 + lane) * 8 + e] = pc[piece];
      }
    }
  }
  return MP_OK;
}

// host-side totals over the launches since the last reset (launches may come from several threads: one lock, taken once per launch)
static std::mutex g_wb_mu;
static double g_wb_direct = 0.0, g_wb_executed = 0.0;
static std::atomic<int> g_wb_telemetry{0};
extern "C" int mp_conv_wino_bf16_stats(double* direct_flops, double* executed_bf16_flops, int reset) {
  std::lock_guard<std::mutex> lock(g_wb_mu);
  if (direct_flops) *direct_flops = g_wb_direct;
  if (executed_bf16_flops) *executed_bf16_flops = g_wb_executed;
  if (reset) g_wb_direct = g_wb_executed = 0.0;
  return MP_OK;
}

// In-kernel clock telemetry (s_memtime / s_memrealtime of every 64th workgroup, six global atomics each): OFF unless switched on here --
// bench.py and the microbenchmarks do; the pose pipeline never pays for it.  Returns the previous setting.
extern "C" int mp_conv_wino_bf16_telemetry(int on) { return g_wb_telemetry.exchange(on ? 1 : 0); }

extern "C" int mp_conv_wino_bf16_phases(double* prologue_cycles, double* epilogue_cycles) {   // per workgroup, since the last clock reset
  unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wb_clk), sizeof(h)));
  if (prologue_cycles) *prologue_cycles = h[5] ? (double)h[3] / (double)h[5] : 0.0;
  if (epilogue_cycles) *epilogue_cycles = h[5] ? (double)h[4] / (double)h[5] : 0.0;
  return MP_OK;
}

#ifdef MP_WINO_PHASES
extern "C" int mp_conv_wino_bf16_phase_probe(double* out9, int reset) {   // average cycle stamps per sampled workgroup (profiling build only)
  unsigned long long h[10];
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wb_phase), sizeof(h)));
  for (int k = 0; k < 9; ++k) out9[k] = h[9] ? (double)h[k] / (double)h[9] : 0.0;
  if (reset) { const unsigned long long z[10] = {0}; MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wb_phase), z, sizeof(z))); }
  return MP_OK;
}
#endif

extern "C" int mp_conv_wino_bf16_clock(double* shader_mhz, double* cycles_per_step, int reset) {
  unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
  MP_CHECK_HIP(hipDeviceSynchronize());
  MP_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wb_clk), sizeof(h)));
  if (shader_mhz) *shader_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
  if (cycles_per_step) *cycles_per_step = h[2] ? (double)h[0] / (double)h[2] : 0.0;
  if (reset) {
    const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
    MP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wb_clk), z, sizeof(z)));
  }
  return MP_OK;
}

extern "C" int mp_conv3x3_wino_bf16_nhwc(const mp_conv_desc* d, const void* d_u_pieces, mp_stream stream) {
  MP_REQUIRE(d && d->d_x && d_u_pieces && (d->d_y || d->d_y_act), "mp_conv3x3_wino_bf16_nhwc: null pointer");
  MP_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "mp_conv3x3_wino_bf16_nhwc: 3x3 / stride 1 / pad 1 only");
  MP_REQUIRE(d->C % WCK == 0 && d->Cout % WCOUT == 0 && d->in_border >= 1, "mp_conv3x3_wino_bf16_nhwc: C %% 16, Cout %% 64, in_border >= 1");
  MP_REQUIRE(!d->d_y_act || (d->d_act_scale && d->d_act_shift), "mp_conv3x3_wino_bf16_nhwc: y_act needs scale/shift");
  WinoParams p;
  p.x = d->d_x; p.u = (const float*)d_u_pieces; p.bias = d->d_bias; p.residual = d->d_residual; p.act_scale = d->d_act_scale; p.act_shift = d->d_act_shift;
  p.y = d->d_y; p.y_act = d->d_y_act;
  p.N = d->N; p.Ho = d->H; p.Wo = d->W;
  p.Hp = d->H + 2 * d->in_border; p.Wp = d->W + 2 * d->in_border; p.C = d->C;
  p.in_off = d->in_border - 1;
  p.Cout = d->Cout;
  p.Hop = d->H + 2 * d->out_border; p.Wop = d->W + 2 * d->out_border; p.out_border = d->out_border;
  p.tiles_x = (d->W + 1) / 2; p.tiles_y = (d->H + 1) / 2;
  const long n_tiles = (long)d->N * p.tiles_x * p.tiles_y;
  const long in_bytes = ((long)d->N * p.Hp + 2) * p.Wp * p.C * 4, out_elems = (long)d->N * p.Hop * p.Wop * d->Cout;
  MP_REQUIRE(n_tiles < (1L << 30) && in_bytes < (1L << 31) && out_elems < (1L << 29), "mp_conv3x3_wino_bf16_nhwc: tensor too large for 32-bit offsets");
  p.out_bytes = (int)(out_elems * 4);
  p.n_tiles = (int)n_tiles;
  p.n_chunks = d->C / 8;
  p.n_steps = d->C / WCK;
  wino_fastdiv_make((unsigned)p.tiles_x, &p.mg_tx, &p.sh_tx);
  wino_fastdiv_make((unsigned)p.tiles_y, &p.mg_ty, &p.sh_ty);
  p.relu = d->relu;
  p.n_cblocks = d->Cout / WCOUT;
  wino_fastdiv_make((unsigned)p.n_cblocks, &p.mg_cb, &p.sh_cb);
  int dev = 0;
  MP_CHECK_HIP(hipGetDevice(&dev));
  static int attr_dev = -1;
  if (attr_dev != dev) {   // (per device: the attribute does not travel with the process)
#define WB_ATTR(...) MP_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_wino_bf16x9<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WB_LDS_BYTES));
    WB_ATTR(0, false, false) WB_ATTR(0, true, false) WB_ATTR(0, false, true) WB_ATTR(0, true, true)
    WB_ATTR(0, false, false, true) WB_ATTR(0, true, false, true) WB_ATTR(0, false, true, true) WB_ATTR(0, true, true, true)
#ifdef MP_CONV_EXPERIMENTS
    WB_ATTR(1, true, true) WB_ATTR(2, true, true) WB_ATTR(4, true, true) WB_ATTR(7, true, true) WB_ATTR(8, true, true) WB_ATTR(16, true, true)
    WB_ATTR(32, true, true) WB_ATTR(64, true, true) WB_ATTR(128, true, true)
#endif
#undef WB_ATTR
    attr_dev = dev;
  }
  const long n_wg = ((n_tiles + WT - 1) / WT) * p.n_cblocks;
  hipStream_t s = (hipStream_t)stream;
  const double c_real = d->c_real > 0 ? d->c_real : d->C;
  const double direct = 2.0 * 9.0 * (double)d->N * d->H * d->W * c_real * d->Cout;
  const double executed = 9.0 * 2.0 * 16.0 * (double)n_tiles * c_real * d->Cout;   // bf16 FLOPs: nine piece products per Winograd multiplication
  {
    std::lock_guard<std::mutex> lock(g_wb_mu);
    g_wb_direct += direct;
    g_wb_executed += executed;
  }
  p.telemetry = g_wb_telemetry.load(std::memory_order_relaxed);
  ProfScope prof("conv3x3_wino_bf16x9<64t,64c>", direct, 4.0 * ((double)d->N * d->H * d->W * (d->C + d->Cout)) + 6.0 * 16.0 * d->C * d->Cout, s, executed, 2500.0);
  const dim3 grid((unsigned)n_wg), block(256);
  // persistent form (default since round 6, MP_WINO_PERSIST=0 = one workgroup per unit): one workgroup per CU walks the units
  static const int persist_env = getenv("MP_WINO_PERSIST") ? atoi(getenv("MP_WINO_PERSIST")) : 1;
  static int n_cu = 0;
  if (!n_cu) { hipDeviceProp_t prop; MP_CHECK_HIP(hipGetDeviceProperties(&prop, dev)); n_cu = prop.multiProcessorCount; }
  p.n_units = (int)n_wg;
  bool launched = false;
#ifdef MP_CONV_EXPERIMENTS
  const int diag = getenv("MP_WINO_DIAG") ? atoi(getenv("MP_WINO_DIAG")) : 0;
#define WB_DIAG_LAUNCH(D) if (diag == D) { hipLaunchKernelGGL((conv3x3_wino_bf16x9<D, true, true>), grid, block, WB_LDS_BYTES, s, p); launched = true; }
  WB_DIAG_LAUNCH(1) WB_DIAG_LAUNCH(2) WB_DIAG_LAUNCH(4) WB_DIAG_LAUNCH(7) WB_DIAG_LAUNCH(8) WB_DIAG_LAUNCH(16) WB_DIAG_LAUNCH(32) WB_DIAG_LAUNCH(64) WB_DIAG_LAUNCH(128)
#undef WB_DIAG_LAUNCH
#endif
  if (launched) {
  } else if (persist_env && n_wg > n_cu) {
    const dim3 pgrid((unsigned)n_cu);
    if (d->d_residual && d->d_y_act) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, true, true, true>), pgrid, block, WB_LDS_BYTES, s, p);
    else if (d->d_residual) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, true, false, true>), pgrid, block, WB_LDS_BYTES, s, p);
    else if (d->d_y_act) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, false, true, true>), pgrid, block, WB_LDS_BYTES, s, p);
    else hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, false, false, true>), pgrid, block, WB_LDS_BYTES, s, p);
  } else if (d->d_residual && d->d_y_act) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, true, true>), grid, block, WB_LDS_BYTES, s, p);
  else if (d->d_residual) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, true, false>), grid, block, WB_LDS_BYTES, s, p);
  else if (d->d_y_act) hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, false, true>), grid, block, WB_LDS_BYTES, s, p);
  else hipLaunchKernelGGL((conv3x3_wino_bf16x9<0, false, false>), grid, block, WB_LDS_BYTES, s, p);
  MP_CHECK_HIP(hipGetLastError());
  return MP_OK;
}
