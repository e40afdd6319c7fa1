// native_wino_check.cpp -- torch-free check + timing of the fused Winograd F(2x2, 3x3) convolution (csrc/conv_wino.hip) against the
// direct fp32-MFMA convolution (mp_conv2d_nhwc, itself checked against torch fp32 by tests/test_gpu_kernels.py) on the same device
// buffers: small shapes (even / odd sizes, partial tile groups, every epilogue mode) by max |difference| relative to the output scale,
// then the config-2 layer shapes (576 rows) timed back to back with the direct kernel.
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/microbench/native_wino_check.cpp \
//              -o scripts/microbench/_build/native_wino_check -Lmegapose6d_amd -lmp_engine -Wl,-rpath,'$ORIGIN/../../../megapose6d_amd'
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "mp_engine.h"
#include "mp_engine_debug.h"

#define HIP_OK(e)                                                                      \
  do {                                                                                 \
    hipError_t err_ = (e);                                                             \
    if (err_ != hipSuccess) {                                                          \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(err_), __FILE__, __LINE__);  \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define MP_OKAY(e)                                                                      \
  do {                                                                                  \
    int rc_ = (e);                                                                      \
    if (rc_ != 0) {                                                                     \
      printf("mp error %d (%s) at %s:%d\n", rc_, mp_last_error(), __FILE__, __LINE__);  \
      return 2;                                                                         \
    }                                                                                   \
  } while (0)

struct Case {
  const char* name;
  int N, C, H, W, Cout, residual, relu, act;
  int timed;
};

static int run_case(const Case& s, int* n_bad) {
  const int Hp = s.H + 2, Wp = s.W + 2;
  const size_t slack = (size_t)(Wp + 1) * s.C + 64;
  const size_t n_in = (size_t)s.N * Hp * Wp * s.C + slack;
  const size_t n_out = (size_t)s.N * Hp * Wp * s.Cout + 64;
  std::mt19937 rng(s.C * 131 + s.H);
  std::normal_distribution<float> G(0.f, 1.f);
  const int n_gen = std::min(s.N, 4);
  const size_t per = (size_t)Hp * Wp * s.C;
  std::vector<float> x((size_t)n_gen * per, 0.f);
  for (int n = 0; n < n_gen; ++n)
    for (int y = 0; y < s.H; ++y)
      for (int xx = 0; xx < s.W; ++xx)
        for (int c = 0; c < s.C; ++c) x[(((size_t)n * Hp + y + 1) * Wp + xx + 1) * s.C + c] = std::fmax(G(rng), 0.f);
  std::vector<float> w((size_t)s.Cout * s.C * 9), bias(s.Cout), scl(s.Cout), sc2(s.Cout), sh2(s.Cout);
  const float a = std::sqrt(2.f / (s.C * 9));
  for (auto& q : w) q = G(rng) * a;
  for (int i = 0; i < s.Cout; ++i) { bias[i] = 0.1f * G(rng); scl[i] = 0.5f + 0.05f * (i % 11); sc2[i] = 0.7f + 0.01f * (i % 13); sh2[i] = 0.05f * G(rng); }
  std::vector<float> packed(mp_conv_packed_floats(s.C, s.Cout, 3, 3)), u(mp_conv_wino_packed_floats(s.C, s.Cout));
  std::vector<unsigned char> ub(mp_conv_wino_bf16_packed_bytes(s.C, s.Cout));
  MP_OKAY(mp_conv_wino_bf16_pack_weights(w.data(), s.Cout, s.C, s.C, scl.data(), ub.data()));
  MP_OKAY(mp_conv_pack_weights(w.data(), s.Cout, s.C, 3, 3, s.C, scl.data(), packed.data()));
  MP_OKAY(mp_conv_wino_pack_weights(w.data(), s.Cout, s.C, s.C, scl.data(), u.data()));
  float *d_x, *d_w, *d_u, *d_b, *d_y0, *d_y1, *d_y2, *d_a0 = nullptr, *d_a1 = nullptr, *d_a2 = nullptr, *d_r = nullptr, *d_s2, *d_h2, *d_sk;
  unsigned char* d_ub;
  HIP_OK(hipMalloc(&d_ub, ub.size()));
  HIP_OK(hipMemcpy(d_ub, ub.data(), ub.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&d_x, n_in * 4));
  HIP_OK(hipMalloc(&d_w, packed.size() * 4));
  HIP_OK(hipMalloc(&d_u, u.size() * 4));
  HIP_OK(hipMalloc(&d_b, s.Cout * 4));
  HIP_OK(hipMalloc(&d_s2, s.Cout * 4));
  HIP_OK(hipMalloc(&d_h2, s.Cout * 4));
  HIP_OK(hipMalloc(&d_y0, n_out * 4));
  HIP_OK(hipMalloc(&d_y1, n_out * 4));
  HIP_OK(hipMalloc(&d_y2, n_out * 4));
  HIP_OK(hipMemset(d_y2, 0, n_out * 4));
  HIP_OK(hipMalloc(&d_sk, (size_t)(12u << 20) * 4));
  HIP_OK(hipMemset(d_x, 0, n_in * 4));
  HIP_OK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  for (int n = n_gen; n < s.N; n += n_gen)
    HIP_OK(hipMemcpy(d_x + (size_t)n * per, d_x, (size_t)std::min(n_gen, s.N - n) * per * 4, hipMemcpyDeviceToDevice));
  // the slack behind the tensor is POISONED with NaN: whatever the kernel reads there must never reach an output
  {
    std::vector<float> nanv(slack, NAN);
    HIP_OK(hipMemcpy(d_x + (size_t)s.N * per, nanv.data(), slack * 4, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_u, u.data(), u.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_b, bias.data(), s.Cout * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_s2, sc2.data(), s.Cout * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_h2, sh2.data(), s.Cout * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(d_y0, 0, n_out * 4));
  HIP_OK(hipMemset(d_y1, 0, n_out * 4));
  if (s.act) {
    HIP_OK(hipMalloc(&d_a0, n_out * 4));
    HIP_OK(hipMalloc(&d_a1, n_out * 4));
    HIP_OK(hipMemset(d_a0, 0, n_out * 4));
    HIP_OK(hipMemset(d_a1, 0, n_out * 4));
    HIP_OK(hipMalloc(&d_a2, n_out * 4));
    HIP_OK(hipMemset(d_a2, 0, n_out * 4));
  }
  if (s.residual) {
    HIP_OK(hipMalloc(&d_r, n_out * 4));
    std::vector<float> r(std::min<size_t>(n_out, (size_t)1 << 22));
    for (auto& q : r) q = 0.5f * G(rng);
    for (size_t o = 0; o < n_out; o += r.size()) HIP_OK(hipMemcpy(d_r + o, r.data(), std::min(r.size(), n_out - o) * 4, hipMemcpyHostToDevice));
  }
  mp_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.d_x = d_x; d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.in_border = 1;
  d.d_w = d_w; d.d_bias = d_b; d.Cout = s.Cout; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1;
  d.d_y = d_y0; d.out_border = 1; d.relu = s.relu; d.d_residual = d_r;
  d.d_y_act = d_a0; d.d_act_scale = s.act ? d_s2 : nullptr; d.d_act_shift = s.act ? d_h2 : nullptr;
  d.d_splitk_ws = d_sk; d.splitk_ws_floats = 12u << 20;
  MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
  mp_conv_desc e = d;
  e.d_y = d_y1; e.d_y_act = d_a1;
  MP_OKAY(mp_conv3x3_wino_nhwc(&e, d_u, nullptr));
  mp_conv_desc g = d;
  g.d_y = d_y2; g.d_y_act = d_a2;
  MP_OKAY(mp_conv3x3_wino_bf16_nhwc(&g, d_ub, nullptr));
  HIP_OK(hipDeviceSynchronize());
  // compare (all of a small case; the first 2 and the last image of a timed one)
  const size_t per_out = (size_t)Hp * Wp * s.Cout;
  std::vector<size_t> imgs;
  for (int n = 0; n < s.N; ++n)
    if (!s.timed || n < 2 || n == s.N - 1) imgs.push_back(n);
  double max_err = 0, max_ref = 0, max_err_a = 0, max_err_b = 0, max_err_ba = 0;
  size_t n_nan = 0;
  std::vector<float> y0(per_out), y1(per_out), y2(per_out);
  for (size_t n : imgs) {
    for (int which = 0; which < (s.act ? 2 : 1); ++which) {
      HIP_OK(hipMemcpy(y0.data(), (which ? d_a0 : d_y0) + n * per_out, per_out * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y1.data(), (which ? d_a1 : d_y1) + n * per_out, per_out * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y2.data(), (which ? d_a2 : d_y2) + n * per_out, per_out * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < per_out; ++i) {
        if (std::isnan(y1[i]) || std::isnan(y2[i])) { ++n_nan; continue; }
        const double eb_ = std::fabs((double)y0[i] - (double)y2[i]);
        if (which) max_err_ba = std::max(max_err_ba, eb_); else max_err_b = std::max(max_err_b, eb_);
        const double e_ = std::fabs((double)y0[i] - (double)y1[i]);
        if (which) max_err_a = std::max(max_err_a, e_); else max_err = std::max(max_err, e_);
        max_ref = std::max(max_ref, (double)std::fabs(y0[i]));
      }
    }
  }
  const bool ok = n_nan == 0 && max_err <= 2e-5 * std::max(1.0, max_ref) && max_err_a <= 2e-5 * std::max(1.0, max_ref) &&
                  max_err_b <= 2e-5 * std::max(1.0, max_ref) && max_err_ba <= 2e-5 * std::max(1.0, max_ref);
  printf("CASE %-34s | max|wino - direct| fp32 %.3e (act %.3e)  bf16x9 %.3e (act %.3e) at output scale %.2f, NaN %zu -> %s\n", s.name, max_err,
         max_err_a, max_err_b, max_err_ba, max_ref, n_nan, ok ? "ok" : "MISMATCH");
  *n_bad += !ok;
  if (s.timed) {
    hipEvent_t e0, e1, e2, e3;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2)); HIP_OK(hipEventCreate(&e3));
    const int reps = 6;
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv3x3_wino_nhwc(&e, d_u, nullptr));
    HIP_OK(hipEventRecord(e2, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv3x3_wino_bf16_nhwc(&g, d_ub, nullptr));
    HIP_OK(hipEventRecord(e3, nullptr));
    if (getenv("MP_WINO_DIAG_SWEEP")) {   // timing experiments of an MP_CONV_EXPERIMENTS build of the library (results of diag != 0 are wrong)
      for (int diag : {0, 1, 2, 4, 7, 32, 64, 128}) {
        char buf[8];
        snprintf(buf, sizeof(buf), "%d", diag);
        setenv("MP_WINO_DIAG", buf, 1);
        hipEvent_t a0, a1;
        HIP_OK(hipEventCreate(&a0)); HIP_OK(hipEventCreate(&a1));
        MP_OKAY(mp_conv3x3_wino_bf16_nhwc(&g, d_ub, nullptr));
        { double t0_, t1_; MP_OKAY(mp_conv_wino_bf16_clock(&t0_, &t1_, 1)); }
        HIP_OK(hipEventRecord(a0, nullptr));
        for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv3x3_wino_bf16_nhwc(&g, d_ub, nullptr));
        HIP_OK(hipEventRecord(a1, nullptr));
        HIP_OK(hipDeviceSynchronize());
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a0, a1));
        double mhz = 0, cps = 0;
        MP_OKAY(mp_conv_wino_bf16_clock(&mhz, &cps, 1));
        printf("DIAG %-34s | diag %d (1 no split, 2 no patch/transform, 4 no weight loads, 32 no V writes, 64 no transform arithmetic, 128 no patch requests): %7.3f ms, %6.0f MHz, %7.0f cycles per step = %5.1f per MFMA\n",
               s.name, diag, ms / reps, mhz, cps, cps / 144.0);
      }
      unsetenv("MP_WINO_DIAG");
    }
    HIP_OK(hipDeviceSynchronize());
    float ms_d = 0, ms_w = 0, ms_b = 0;
    HIP_OK(hipEventElapsedTime(&ms_d, e0, e1));
    HIP_OK(hipEventElapsedTime(&ms_w, e1, e2));
    HIP_OK(hipEventElapsedTime(&ms_b, e2, e3));
    const double flops = 2.0 * s.N * s.H * s.W * (double)s.Cout * 9 * s.C;
    const double tiles = (double)s.N * ((s.H + 1) / 2) * ((s.W + 1) / 2);
    const double exec = 2.0 * 16.0 * tiles * s.C * s.Cout;
    printf("TIME %-34s | direct %7.3f ms %6.1f TFLOP/s | winograd %7.3f ms: executed %6.1f TFLOP/s, direct-equivalent %6.1f TFLOP/s | x%.2f\n", s.name,
           ms_d / reps, flops * reps / (ms_d * 1e-3) / 1e12, ms_w / reps, exec * reps / (ms_w * 1e-3) / 1e12, flops * reps / (ms_w * 1e-3) / 1e12,
           ms_d / ms_w);
    {
      double mhz = 0, cps = 0, pro = 0, epi = 0;
      MP_OKAY(mp_conv_wino_bf16_phases(&pro, &epi));
      {   // profiling builds (-DMP_WINO_PHASES) export finer stamps
        typedef int (*probe_fn)(double*, int);
        probe_fn probe = (probe_fn)dlsym(RTLD_DEFAULT, "mp_conv_wino_bf16_phase_probe");
        double ph[9];
        if (probe && probe(ph, 1) == 0)
          printf("PHASE %-33s | prologue: requests out %5.0f, row 0 transformed %5.0f, V written %5.0f, barrier %5.0f, loop entry %5.0f | epilogue: exchange written %5.0f, barrier %5.0f, stores issued %5.0f, retired %5.0f\n",
                 s.name, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8]);
      }
      MP_OKAY(mp_conv_wino_bf16_clock(&mhz, &cps, 1));
      printf("CLK  %-34s | bf16x9 K loop: %6.0f MHz, %7.0f cycles per 16-channel step = %5.1f per MFMA; per workgroup: prologue %6.0f, K loop %7.0f, epilogue %6.0f cycles\n",
             s.name, mhz, cps, cps / 144.0, pro, cps * (s.C / 16), epi);
    }
    printf("TIME %-34s | bf16x9 winograd %7.3f ms: executed bf16 %7.1f TFLOP/s (%.2f of 2500), direct-equivalent %6.1f TFLOP/s | x%.2f vs direct, x%.2f vs fp32 winograd\n",
           s.name, ms_b / reps, 9.0 * exec * reps / (ms_b * 1e-3) / 1e12, 9.0 * exec * reps / (ms_b * 1e-3) / 1e12 / 2500.0,
           flops * reps / (ms_b * 1e-3) / 1e12, ms_d / ms_b, ms_w / ms_b);
  }
  (void)hipFree(d_y2); (void)hipFree(d_ub);
  if (d_a2) (void)hipFree(d_a2);
  (void)hipFree(d_x); (void)hipFree(d_w); (void)hipFree(d_u); (void)hipFree(d_b); (void)hipFree(d_y0); (void)hipFree(d_y1); (void)hipFree(d_sk);
  (void)hipFree(d_s2); (void)hipFree(d_h2);
  if (d_a0) (void)hipFree(d_a0);
  if (d_a1) (void)hipFree(d_a1);
  if (d_r) (void)hipFree(d_r);
  return 0;
}

int main(int argc, char** argv) {
  mp_conv_wino_bf16_telemetry(1);   // the CLK lines below read the in-kernel clock counters (off by default in the library)
  static const Case CASES[] = {
      {"3x 64->64 @12x16 plain", 3, 64, 12, 16, 64, 0, 0, 0, 0},
      {"5x 64->128 @15x20 res+relu (odd H)", 5, 64, 15, 20, 128, 1, 1, 0, 0},
      {"2x 128->64 @13x11 relu (odd H, W)", 2, 128, 13, 11, 64, 0, 1, 0, 0},
      {"4x 256->256 @8x10 res + act", 4, 256, 8, 10, 256, 1, 0, 1, 0},
      {"7x 16->64 @6x6 bias only", 7, 16, 6, 6, 64, 0, 0, 0, 0},
      {"layer1 64->64 @60x80 x576", 576, 64, 60, 80, 64, 1, 1, 0, 1},
      {"layer2 128->128 @30x40 x576", 576, 128, 30, 40, 128, 1, 1, 0, 1},
      {"layer3 256->256 @15x20 x576", 576, 256, 15, 20, 256, 1, 1, 0, 1},
      {"layer4 512->512 @8x10 x576", 576, 512, 8, 10, 512, 1, 1, 0, 1},
  };
  const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
  int n_bad = 0;
  for (const Case& c : CASES) {
    if (quick && c.timed) continue;
    const int rc = run_case(c, &n_bad);
    if (rc) return rc;
  }
  printf("%s\n", n_bad ? "FAILED" : "ALL OK");
  return n_bad ? 1 : 0;
}
