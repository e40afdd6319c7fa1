// native_stem_check.cpp -- torch-free check + timing of the exact-piece bf16 stem convolution (csrc/conv_stem.hip, mp_conv_stem_xrec)
// against the fp32-MFMA convolution (mp_conv2d_nhwc, itself checked against torch fp32 by tests/test_gpu_kernels.py) on the SAME values:
// the xrec records (three bf16 pieces of every fp32 channel, the integer k of every render channel) are generated on the host, the fp32
// tensor is rebuilt from them (x = x1 + x2 + x3 exactly; k / 255 rounded once, as the reference's uint8 -> float path does).
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/microbench/native_stem_check.cpp \
//              -o scripts/microbench/_build/native_stem_check -Lmegapose6d_amd -lmp_engine -Wl,-rpath,'$ORIGIN/../../../megapose6d_amd'
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
  int N, n_f32, n_u8, H, W, KS, Cout, relu, timed;
};

static void split3(float v, unsigned short out[3]) {
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
  out[0] = (unsigned short)(h >> 16); out[1] = (unsigned short)(m >> 16); out[2] = (unsigned short)(qb >> 16);
}

static int run_case(const Case& s, int* n_bad) {
  const int Cin = s.n_f32 + s.n_u8, Cp = (Cin + 3) / 4 * 4, B = s.KS / 2, pad = s.KS / 2;
  const int R = mp_xrec_elements(s.n_f32, s.n_u8);
  const int Hp = s.H + 2 * B, Wp = s.W + 2 * B;
  const int Ho = (s.H + 2 * pad - s.KS) / 2 + 1, Wo = (s.W + 2 * pad - s.KS) / 2 + 1;
  const size_t per_x = (size_t)Hp * Wp * Cp, per_r = (size_t)Hp * Wp * R, per_out = (size_t)(Ho + 2) * (Wo + 2) * s.Cout;
  std::mt19937 rng(Cin * 131 + s.H + s.KS);
  std::normal_distribution<float> G(0.f, 1.f);
  std::uniform_int_distribution<int> U8(0, 255);
  const int n_gen = std::min(s.N, 3);
  std::vector<float> x((size_t)n_gen * per_x, 0.f);
  std::vector<unsigned short> xr((size_t)n_gen * per_r, 0);
  for (int n = 0; n < n_gen; ++n)
    for (int y = 0; y < s.H; ++y)
      for (int xx = 0; xx < s.W; ++xx) {
        const size_t pix = ((size_t)n * Hp + y + B) * Wp + xx + B;
        const bool bg = ((xx / 16 + y / 16 + n) % 3) == 0;   // a third of the pixels are background (renders 0), like real crops
        for (int c = 0; c < Cin; ++c) {
          if (c < s.n_f32) {
            const float v = 0.5f + 0.25f * G(rng);
            x[pix * Cp + c] = v;
            split3(v, &xr[pix * R + 3 * c]);
          } else {
            const int k = bg ? 0 : U8(rng);
            x[pix * Cp + c] = (float)k / 255.f;
            const float kf = (float)k;
            unsigned kb; memcpy(&kb, &kf, 4);
            xr[pix * R + 3 * s.n_f32 + (c - s.n_f32)] = (unsigned short)(kb >> 16);
          }
        }
      }
  std::vector<float> w((size_t)s.Cout * Cin * s.KS * s.KS), bias(s.Cout), scl(s.Cout);
  const float a = std::sqrt(2.f / (Cin * s.KS * s.KS));
  for (auto& q : w) q = G(rng) * a;
  for (int i = 0; i < s.Cout; ++i) { bias[i] = 0.1f * G(rng); scl[i] = 0.5f + 0.05f * (i % 11); }
  std::vector<float> packed(mp_conv_packed_floats(Cp, s.Cout, s.KS, s.KS));
  std::vector<unsigned char> pk(mp_conv_stem_packed_bytes(s.KS, s.n_f32, s.n_u8, s.Cout));
  MP_OKAY(mp_conv_pack_weights(w.data(), s.Cout, Cin, s.KS, s.KS, Cp, scl.data(), packed.data()));
  MP_OKAY(mp_conv_stem_pack_weights(w.data(), s.Cout, Cin, s.KS, s.n_f32, scl.data(), pk.data()));
  float *d_x, *d_w, *d_b, *d_y0, *d_y1, *d_sk;
  unsigned short* d_xr;
  unsigned char* d_pk;
  const size_t n_out = (size_t)s.N * per_out + 64;
  HIP_OK(hipMalloc(&d_x, (size_t)s.N * per_x * 4 + 4096));
  HIP_OK(hipMalloc(&d_xr, (size_t)s.N * per_r * 2));
  HIP_OK(hipMalloc(&d_w, packed.size() * 4));
  HIP_OK(hipMalloc(&d_pk, pk.size()));
  HIP_OK(hipMalloc(&d_b, s.Cout * 4));
  HIP_OK(hipMalloc(&d_y0, n_out * 4));
  HIP_OK(hipMalloc(&d_y1, n_out * 4));
  HIP_OK(hipMalloc(&d_sk, (size_t)(12u << 20) * 4));
  HIP_OK(hipMemset(d_x, 0, (size_t)s.N * per_x * 4 + 4096));
  for (int n = 0; n < s.N; n += n_gen) {
    const int m = std::min(n_gen, s.N - n);
    HIP_OK(hipMemcpy(d_x + (size_t)n * per_x, x.data(), (size_t)m * per_x * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_xr + (size_t)n * per_r, xr.data(), (size_t)m * per_r * 2, hipMemcpyHostToDevice));
  }
  HIP_OK(hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_pk, pk.data(), pk.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_b, bias.data(), s.Cout * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(d_y0, 0, n_out * 4));
  HIP_OK(hipMemset(d_y1, 0, n_out * 4));
  mp_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.d_x = d_x; d.N = s.N; d.H = s.H; d.W = s.W; d.C = Cp; d.c_real = Cin; d.in_border = B;
  d.d_w = d_w; d.d_bias = d_b; d.Cout = s.Cout; d.KH = s.KS; d.KW = s.KS; d.stride = 2; d.pad = pad;
  d.d_y = d_y0; d.out_border = 1; d.relu = s.relu;
  d.d_splitk_ws = d_sk; d.splitk_ws_floats = 12u << 20;
  MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
  mp_conv_desc e = d;
  e.d_x = (const float*)d_xr; e.d_y = d_y1; e.d_w = nullptr; e.d_splitk_ws = nullptr;
  MP_OKAY(mp_conv_stem_xrec(&e, d_pk, s.n_f32, nullptr));
  HIP_OK(hipDeviceSynchronize());
  std::vector<size_t> imgs;
  for (int n = 0; n < s.N; ++n)
    if (!s.timed || n < 2 || n == s.N - 1) imgs.push_back(n);
  double max_err = 0, max_ref = 0;
  size_t n_nan = 0, n_border = 0;
  std::vector<float> y0(per_out), y1(per_out);
  for (size_t n : imgs) {
    HIP_OK(hipMemcpy(y0.data(), d_y0 + n * per_out, per_out * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(y1.data(), d_y1 + n * per_out, per_out * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < per_out; ++i) {
      if (std::isnan(y1[i])) { ++n_nan; continue; }
      const size_t pixi = i / s.Cout;
      const int yy = (int)(pixi / (Wo + 2)), xx = (int)(pixi % (Wo + 2));
      if ((yy == 0 || yy == Ho + 1 || xx == 0 || xx == Wo + 1) && y1[i] != 0.f) ++n_border;   // borders must stay untouched
      max_err = std::max(max_err, std::fabs((double)y0[i] - (double)y1[i]));
      max_ref = std::max(max_ref, (double)std::fabs(y0[i]));
    }
  }
  const bool ok = n_nan == 0 && n_border == 0 && max_err <= 2e-5 * std::max(1.0, max_ref);
  printf("CASE %-40s | max|stem - fp32 direct| %.3e at output scale %.2f, NaN %zu, border writes %zu -> %s\n", s.name, max_err, max_ref, n_nan,
         n_border, ok ? "ok" : "MISMATCH");
  *n_bad += !ok;
  if (s.timed) {
    hipEvent_t e0, e1, e2;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2));
    const int reps = 6;
    for (int r = 0; r < 2; ++r) MP_OKAY(mp_conv_stem_xrec(&e, d_pk, s.n_f32, nullptr));
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv_stem_xrec(&e, d_pk, s.n_f32, nullptr));
    HIP_OK(hipEventRecord(e2, nullptr));
    HIP_OK(hipDeviceSynchronize());
    float ms_d = 0, ms_w = 0;
    HIP_OK(hipEventElapsedTime(&ms_d, e0, e1));
    HIP_OK(hipEventElapsedTime(&ms_w, e1, e2));
    const double flops = 2.0 * s.N * Ho * Wo * (double)s.Cout * s.KS * s.KS * Cin;
    const int R8 = R / 8, steps = ((s.KS * s.KS * R8 + 3) / 4 + 1) / 2 * 2;
    const double exec = 2.0 * s.N * Ho * Wo * (double)s.Cout * steps * 32 * 3;
    printf("TIME %-40s | fp32 direct %7.3f ms %6.1f TFLOP/s | bf16x3 stem %7.3f ms: algorithmic %6.1f TFLOP/s, executed bf16 %7.1f TFLOP/s | x%.2f\n",
           s.name, ms_d / reps, flops * reps / (ms_d * 1e-3) / 1e12, ms_w / reps, flops * reps / (ms_w * 1e-3) / 1e12,
           exec * reps / (ms_w * 1e-3) / 1e12, ms_d / ms_w);
  }
  (void)hipFree(d_x); (void)hipFree(d_xr); (void)hipFree(d_w); (void)hipFree(d_pk); (void)hipFree(d_b); (void)hipFree(d_y0); (void)hipFree(d_y1);
  (void)hipFree(d_sk);
  return 0;
}

int main(int argc, char** argv) {
  static const Case CASES[] = {
      {"2x 3+6 ch @32x48 7x7 relu", 2, 3, 6, 32, 48, 7, 64, 1, 0},
      {"3x 3+24 ch @30x44 7x7 (ragged tiles)", 3, 3, 24, 30, 44, 7, 64, 0, 0},
      {"2x 3+24 ch @26x38 5x5 relu, Cout 128", 2, 3, 24, 26, 38, 5, 128, 1, 0},
      {"2x 3+12 ch @24x40 7x7 relu (Q = 3)", 2, 3, 12, 24, 40, 7, 64, 1, 0},
      {"2x 3+18 ch @18x34 5x5 (Q = 4)", 2, 3, 18, 18, 34, 5, 64, 0, 0},
      {"1x 3+6 ch @240x320 7x7 relu", 1, 3, 6, 240, 320, 7, 64, 1, 0},
      {"coarse 3+6 ch @240x320 7x7 x576", 576, 3, 6, 240, 320, 7, 64, 1, 1},
      {"refiner 3+24 ch @240x320 7x7 x576", 576, 3, 24, 240, 320, 7, 64, 1, 1},
      {"wide refiner 3+24 ch @240x320 5x5 x576", 576, 3, 24, 240, 320, 5, 64, 1, 1},
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
