// native_f16_check.cpp -- torch-free self-consistency check of the "fp16 renders" mode through the C-ABI (starts in milliseconds,
// so it fits a GPU slot that a Python process would spend importing torch).  Every check has an exact statement:
//   raster:     MP_RASTER_F16 output            == round-to-nearest-even(fp32 output of the same launch)          (bitwise)
//   crop role:  mp_raster_render_crop, F16      == the same                                                       (bitwise)
//   stem conv:  mp_conv2d_nhwc(x_f16 = 1, x)    == mp_conv2d_nhwc(x widened to fp32)                              (bitwise)
//   depth norm: mp_normalize_depth_f16(x)       == round(mp_normalize_depth(float(x)))                            (bitwise)
// Build (container, no GPU needed):
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/microbench/native_f16_check.cpp -o scripts/microbench/_build/native_f16_check \
//         -Lmegapose6d_amd -lmp_engine -Wl,-rpath,'$ORIGIN/../../../megapose6d_amd'
// Run on the GPU box:  scripts/microbench/_build/native_f16_check   (exit code 0 = every check passed)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "mp_engine.h"

#define HIP_OK(e)                                                                          \
  do {                                                                                     \
    hipError_t err_ = (e);                                                                 \
    if (err_ != hipSuccess) {                                                              \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(err_), __FILE__, __LINE__);      \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)
#define MP_OKAY(e)                                                                \
  do {                                                                            \
    int rc_ = (e);                                                                \
    if (rc_ != 0) {                                                               \
      printf("mp error %d (%s) at %s:%d\n", rc_, mp_last_error(), __FILE__, __LINE__); \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

static uint16_t f2h(float f) {  // round to nearest even, IEEE binary16 (host compiler's _Float16 conversion)
  _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static float h2f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}

template <typename T>
static T* dev_upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

static int n_fail = 0;
static void report(const char* what, size_t n, size_t bad, size_t nonzero) {
  printf("%-58s %s  (%zu elements, %zu non-zero, %zu mismatching)\n", what, bad == 0 && nonzero > 0 ? "PASS" : "FAIL", n, nonzero, bad);
  if (bad != 0 || nonzero == 0) ++n_fail;
}

// lathe surface (a bottle-like closed-ish shape), nt x nz quads -> 2 triangles each
static void make_mesh(int nt, int nz, std::vector<float>& v, std::vector<float>& nrm, std::vector<float>& col, std::vector<int32_t>& f) {
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  for (int iz = 0; iz <= nz; ++iz)
    for (int it = 0; it < nt; ++it) {
      const float z = -0.08f + 0.16f * iz / nz, th = 6.2831853f * it / nt;
      const float r = 0.03f + 0.015f * std::sin(9.f * z / 0.16f);
      v.insert(v.end(), {r * std::cos(th), r * std::sin(th), z});
      nrm.insert(nrm.end(), {std::cos(th), std::sin(th), 0.f});
      col.insert(col.end(), {U(rng), U(rng), U(rng)});
    }
  for (int iz = 0; iz < nz; ++iz)
    for (int it = 0; it < nt; ++it) {
      const int a = iz * nt + it, b = iz * nt + (it + 1) % nt, c = a + nt, d = b + nt;
      f.insert(f.end(), {a, b, c});
      f.insert(f.end(), {b, d, c});
    }
}

int main() {
  int n_cu = 0, lds = 0;
  char arch[64];
  MP_OKAY(mp_device_info(&n_cu, &lds, arch, sizeof(arch)));
  printf("device %s, %d CUs\n", arch, n_cu);
  hipStream_t s = nullptr;

  // ---------------------------------------------------------------- rasteriser -----------------------------------------------------
  std::vector<float> mv, mn, mc;
  std::vector<int32_t> mf;
  make_mesh(48, 40, mv, mn, mc, mf);
  mp_mesh_desc md = {mv.data(), mn.data(), mc.data(), mf.data(), (int32_t)(mv.size() / 3), (int32_t)(mf.size() / 3)};
  mp_mesh_db* db = nullptr;
  MP_OKAY(mp_mesh_db_create(&md, 1, &db));
  const int n_items = 3, V = 4, n_views = n_items * V, h = 240, w = 320;
  std::vector<float> TCO(n_views * 16, 0.f), K(n_views * 9, 0.f);
  for (int i = 0; i < n_views; ++i) {
    const float a = 0.3f * i, ca = std::cos(a), sa = std::sin(a);
    const float R[9] = {ca, 0.f, sa, 0.f, 1.f, 0.f, -sa, 0.f, ca};   // rotation about y ...
    const float Rx[9] = {1.f, 0.f, 0.f, 0.f, 0.f, -1.f, 0.f, 1.f, 0.f};  // ... after tipping the lathe axis into the image plane
    float* T = &TCO[i * 16];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int k = 0; k < 3; ++k) acc += R[r * 3 + k] * Rx[k * 3 + c];
        T[r * 4 + c] = acc;
      }
    T[3] = 0.01f * (i % 3 - 1); T[7] = 0.005f * (i % 2); T[11] = 0.35f + 0.02f * (i % 4); T[15] = 1.f;
    float* Kk = &K[i * 9];
    Kk[0] = 520.f; Kk[4] = 520.f; Kk[2] = 160.f; Kk[5] = 120.f; Kk[8] = 1.f;
  }
  std::vector<int32_t> ids(n_views, 0);
  int32_t* d_ids = dev_upload(ids);
  float* d_T = dev_upload(TCO);
  float* d_K = dev_upload(K);
  mp_lights L;
  memset(&L, 0, sizeof(L));
  L.ambient[0] = L.ambient[1] = L.ambient[2] = 0.3f;
  L.n_point = 2;
  L.point_dir[0][0] = 1.f; L.point_dir[1][2] = -1.f;
  for (int c = 0; c < 3; ++c) L.point_color[0][c] = L.point_color[1][c] = 0.4f;
  const size_t ws_bytes = mp_raster_workspace_bytes(db, n_views, h, w);
  void* d_ws = nullptr;
  HIP_OK(hipMalloc(&d_ws, ws_bytes));
  // observation frames for the crop role: [1,3,480,640] in [0,1]
  const int H = 480, W = 640;
  std::vector<float> img((size_t)3 * H * W);
  {
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (auto& x : img) x = U(rng);
  }
  float* d_img = dev_upload(img);
  std::vector<int32_t> im_ids(n_items, 0);
  std::vector<float> boxes = {100.f, 80.f, 420.f, 320.f, 200.f, 100.f, 360.f, 220.f, 50.5f, 40.25f, 610.f, 460.75f};
  int32_t* d_im_ids = dev_upload(im_ids);
  float* d_boxes = dev_upload(boxes);
  if (!d_ids || !d_T || !d_K || !d_img || !d_im_ids || !d_boxes) { printf("upload failed\n"); return 2; }

  for (int pass = 0; pass < 4; ++pass) {
    const bool msaa = pass & 1, fused = pass & 2;
    // CNN-input layout: per item one padded-NHWC row, C = 28 (3 crop + 4 x (3 rgb + 3 normals) = 27, padded), border 3
    const int Cp = fused ? 28 : 8, B = fused ? 3 : 0, vpi = fused ? V : 1;
    const int items = fused ? n_items : n_views;
    const size_t Wp = w + 2 * B, Hp = h + 2 * B;
    const size_t n_el = (size_t)items * Hp * Wp * Cp;
    float* d_o32 = nullptr;
    uint16_t* d_o16 = nullptr;
    HIP_OK(hipMalloc(&d_o32, n_el * 4));
    HIP_OK(hipMalloc(&d_o16, n_el * 2));
    HIP_OK(hipMemset(d_o32, 0, n_el * 4));
    HIP_OK(hipMemset(d_o16, 0, n_el * 2));
    const int64_t stride_v = (int64_t)Hp * Wp * Cp, stride_y = (int64_t)Wp * Cp, stride_x = Cp, off = ((int64_t)B * Wp + B) * Cp;
    const uint32_t flags = MP_RASTER_NORMALS | (fused ? 0u : MP_RASTER_DEPTH) | (msaa ? MP_RASTER_MSAA4 : 0u);
    for (int half = 0; half < 2; ++half) {
      float* out = half ? (float*)(d_o16 + off) : d_o32 + off;
      const uint32_t fl = flags | (half ? MP_RASTER_F16 : 0u);
      if (fused)
        MP_OKAY(mp_raster_render_crop(db, d_ids, d_T, d_K, n_views, h, w, fl, &L, out, stride_v, vpi, 6, stride_y, stride_x, 3, 6, -1, d_ws,
                                      ws_bytes, d_img, 0, 1, 3, H, W, d_im_ids, d_boxes, 0, s));
      else
        MP_OKAY(mp_raster_render(db, d_ids, d_T, d_K, n_views, h, w, fl, &L, out, stride_v, vpi, 0, stride_y, stride_x, 0, 3, 6, d_ws, ws_bytes, s));
    }
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> o32(n_el);
    std::vector<uint16_t> o16(n_el);
    HIP_OK(hipMemcpy(o32.data(), d_o32, n_el * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o16.data(), d_o16, n_el * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, nz = 0;
    for (size_t i = 0; i < n_el; ++i) {
      bad += f2h(o32[i]) != o16[i];
      nz += o32[i] != 0.f;
    }
    char name[96];
    snprintf(name, sizeof(name), "raster %s, %s samples: f16 == rne(f32)", fused ? "+ fused crop (CNN input)" : "rgb+normals+depth", msaa ? "4" : "1");
    report(name, n_el, bad, nz);
    (void)hipFree(d_o32);
    (void)hipFree(d_o16);
  }

  // ---------------------------------------------------------------- stem convolution -----------------------------------------------
  struct Case { int N, Cin, H, W, K, stride, pad, border; };
  const Case cases[] = {{2, 9, 48, 64, 7, 2, 3, 3}, {1, 27, 30, 40, 7, 2, 3, 3}, {3, 32, 30, 40, 5, 2, 2, 2}, {2, 32, 24, 32, 7, 2, 3, 3},
                        {2, 27, 240, 320, 7, 2, 3, 3}};
  for (const Case& c : cases) {
    const int Cp = (c.Cin + 3) / 4 * 4, Cout = 64;
    const int Hp = c.H + 2 * c.border, Wp = c.W + 2 * c.border;
    const int Ho = (c.H + 2 * c.pad - c.K) / c.stride + 1, Wo = (c.W + 2 * c.pad - c.K) / c.stride + 1;
    const size_t n_in = (size_t)c.N * Hp * Wp * Cp + (size_t)Wp * Cp + 64;   // + read slack (engine.padded_nhwc)
    std::vector<float> x32(n_in, 0.f);
    std::vector<uint16_t> x16(n_in, 0);
    std::mt19937 rng(c.Cin * 31 + c.K);
    std::normal_distribution<float> G(0.f, 1.f);
    for (int n = 0; n < c.N; ++n)
      for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x)
          for (int ch = 0; ch < c.Cin; ++ch) {
            const size_t i = (((size_t)n * Hp + y + c.border) * Wp + x + c.border) * Cp + ch;
            x16[i] = f2h(G(rng));
            x32[i] = h2f(x16[i]);
          }
    std::vector<float> wt((size_t)Cout * c.Cin * c.K * c.K), scale(Cout), bias(Cout);
    for (auto& v : wt) v = G(rng) * std::sqrt(2.f / (c.Cin * c.K * c.K));
    for (int i = 0; i < Cout; ++i) { scale[i] = 0.5f + 0.01f * i; bias[i] = 0.1f * G(rng); }
    std::vector<float> packed(mp_conv_packed_floats(Cp, Cout, c.K, c.K));
    MP_OKAY(mp_conv_pack_weights(wt.data(), Cout, c.Cin, c.K, c.K, Cp, scale.data(), packed.data()));
    float* d_w = dev_upload(packed);
    float* d_b = dev_upload(bias);
    float* d_x32 = dev_upload(x32);
    uint16_t* d_x16 = dev_upload(x16);
    const size_t n_out = (size_t)c.N * (Ho + 2) * (Wo + 2) * Cout + 64;
    float *d_y32 = nullptr, *d_y16 = nullptr;
    HIP_OK(hipMalloc(&d_y32, n_out * 4));
    HIP_OK(hipMalloc(&d_y16, n_out * 4));
    HIP_OK(hipMemset(d_y32, 0, n_out * 4));
    HIP_OK(hipMemset(d_y16, 0, n_out * 4));
    for (int half = 0; half < 2; ++half) {
      mp_conv_desc d;
      memset(&d, 0, sizeof(d));
      d.d_x = half ? (const float*)d_x16 : d_x32;
      d.x_f16 = half;
      d.N = c.N; d.H = c.H; d.W = c.W; d.C = Cp; d.c_real = c.Cin; d.in_border = c.border;
      d.d_w = d_w; d.d_bias = d_b; d.Cout = Cout; d.KH = c.K; d.KW = c.K; d.stride = c.stride; d.pad = c.pad;
      d.d_y = half ? d_y16 : d_y32; d.out_border = 1; d.relu = 1;
      MP_OKAY(mp_conv2d_nhwc(&d, s));
    }
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> y32(n_out), y16(n_out);
    HIP_OK(hipMemcpy(y32.data(), d_y32, n_out * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(y16.data(), d_y16, n_out * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, nz = 0;
    for (size_t i = 0; i < n_out; ++i) {
      bad += memcmp(&y32[i], &y16[i], 4) != 0;
      nz += y32[i] != 0.f;
    }
    // and the fp32 result itself against a host reference on a few output pixels (so that "both wrong alike" cannot pass)
    double max_err = 0.0, max_ref = 1.0;
    for (int probe = 0; probe < 64; ++probe) {
      const int n = probe % c.N, ho = (probe * 7) % Ho, wo = (probe * 13) % Wo, co = (probe * 5) % Cout;
      double acc = 0.0;
      for (int kh = 0; kh < c.K; ++kh)
        for (int kw = 0; kw < c.K; ++kw) {
          const int y = ho * c.stride - c.pad + kh, x = wo * c.stride - c.pad + kw;
          if (y < 0 || y >= c.H || x < 0 || x >= c.W) continue;
          for (int ch = 0; ch < c.Cin; ++ch)
            acc += (double)x32[(((size_t)n * Hp + y + c.border) * Wp + x + c.border) * Cp + ch] *
                   (double)(wt[(((size_t)co * c.Cin + ch) * c.K + kh) * c.K + kw] * scale[co]);
        }
      const double ref = std::fmax(acc + bias[co], 0.0);
      const double got = y16[(((size_t)n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * Cout + co];
      max_err = std::fmax(max_err, std::fabs(got - ref));
      max_ref = std::fmax(max_ref, std::fabs(ref));
    }
    char name[96];
    snprintf(name, sizeof(name), "stem conv N=%d C=%d %dx%d k%d: half input == fp32 input", c.N, c.Cin, c.H, c.W, c.K);
    report(name, n_out, bad + (max_err > 2e-4 * max_ref ? 1 : 0), nz);
    printf("    (vs host fp64 on 64 probes: max err %.3g, scale %.3g)\n", max_err, max_ref);
    (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_x32); (void)hipFree(d_x16); (void)hipFree(d_y32); (void)hipFree(d_y16);
  }

  // ---------------------------------------------------------------- depth normalisation --------------------------------------------
  {
    const int b = 2, hh = 12, ww = 20, C = 8, border = 2;
    const size_t n_el = (size_t)b * (hh + 2 * border) * (ww + 2 * border) * C;
    std::vector<float> x32(n_el);
    std::vector<uint16_t> x16(n_el);
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(0.f, 2.f);
    for (size_t i = 0; i < n_el; ++i) { x16[i] = f2h(U(rng)); x32[i] = h2f(x16[i]); }
    std::vector<float> tCR = {0.f, 0.f, 0.7f, 0.1f, 0.f, 1.3f};
    float* d_t = dev_upload(tCR);
    const int32_t ch[2] = {3, 6};
    for (int mode = 1; mode <= 3; ++mode) {
      float* d32 = dev_upload(x32);
      uint16_t* d16 = dev_upload(x16);
      MP_OKAY(mp_normalize_depth(d32, b, hh, ww, border, C, ch, 2, d_t, mode, s));
      MP_OKAY(mp_normalize_depth_f16(d16, b, hh, ww, border, C, ch, 2, d_t, mode, s));
      HIP_OK(hipDeviceSynchronize());
      std::vector<float> o32(n_el);
      std::vector<uint16_t> o16(n_el);
      HIP_OK(hipMemcpy(o32.data(), d32, n_el * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(o16.data(), d16, n_el * 2, hipMemcpyDeviceToHost));
      size_t bad = 0, nz = 0;
      for (size_t i = 0; i < n_el; ++i) { bad += f2h(o32[i]) != o16[i]; nz += o32[i] != 0.f; }
      char name[96];
      snprintf(name, sizeof(name), "normalize_depth_f16 mode %d == rne(normalize_depth)", mode);
      report(name, n_el, bad, nz);
      (void)hipFree(d32); (void)hipFree(d16);
    }
  }
  mp_mesh_db_destroy(db);
  printf(n_fail ? "RESULT: %d check(s) FAILED\n" : "RESULT: all checks passed\n", n_fail);
  return n_fail ? 1 : 0;
}
