// native_detector_check.cpp -- torch-free parity run of the detector network (csrc/detector.hip) against the CPU oracle's fixture
// (oracle/make_detector_fixture.py -> tests/_build/detector_fixture_<case>.bin).  Weights and images are regenerated here from the
// same hash the oracle used (oracle/mask_rcnn.py synthetic_tensor / synthetic_images), so only the oracle's RESULTS travel.
// Build:  hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/microbench/native_detector_check.cpp \
//               -o scripts/microbench/_build/native_detector_check -Lmegapose6d_amd -lmp_engine -Wl,-rpath,'$ORIGIN/../../../megapose6d_amd'
// Run:    scripts/microbench/_build/native_detector_check tests/_build/detector_fixture_native.bin [more fixtures]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mp_engine.h"

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

// ---- the oracle's synthetic data (oracle/mask_rcnn.py) -------------------------------------------------------------------------
static uint64_t fnv1a(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char c : s) h = (h ^ c) * 0x100000001B3ull;
  return h;
}
static inline float hash_unit(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * 1.1920928955078125e-07f - 1.0f;   // u24 * 2^-23 - 1
}
static bool ends_with(const std::string& s, const char* t) {
  const size_t n = strlen(t);
  return s.size() >= n && s.compare(s.size() - n, n, t) == 0;
}
static void synthetic_tensor(const std::string& name, const int64_t* shape, int nd, std::vector<float>& v) {
  size_t n = 1;
  for (int k = 0; k < nd; ++k) n *= (size_t)shape[k];
  v.resize(n);
  const uint64_t seed = fnv1a(name);
  const bool is_bn = name.find(".bn") != std::string::npos || name.find("downsample.1.") != std::string::npos;
  int mode;   // 0 weight, 1 var / bn scale, 2 bn3 scale, 3 mean / bias
  if (ends_with(name, ".running_var")) mode = 1;
  else if (is_bn && ends_with(name, ".weight")) mode = name.find(".bn3.") != std::string::npos ? 2 : 1;
  else if (ends_with(name, ".running_mean") || ends_with(name, ".bias")) mode = 3;
  else mode = 0;
  const float a = (float)std::sqrt(3.0 / (double)(n / (size_t)shape[0]));
  for (size_t i = 0; i < n; ++i) {
    const float u = hash_unit(seed, i);
    float x;
    if (mode == 1) x = u * 0.5f + 1.0f;
    else if (mode == 2) x = (u * 0.5f + 1.0f) * 0.3f;
    else if (mode == 3) x = u * 0.1f;
    else x = u * a;
    v[i] = x;
  }
}
static void synthetic_images(int n, int h, int w, std::vector<float>& img) {
  img.resize((size_t)n * 3 * h * w);
  const uint64_t seed = fnv1a("images/7");
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
          const size_t k = (((size_t)i * 3 + c) * h + y) * w + x;
          const float ys = (float)y / (float)h, xs = (float)x / (float)w;
          const float base = 0.5f + 0.25f * ys - 0.2f * xs;
          float v = base + 0.25f * hash_unit(seed, k);
          img[k] = std::fmin(std::fmax(v, 0.0f), 0.999f);
        }
}

// ---- fixture ----------------------------------------------------------------------------------------------------------------------
struct Rec {
  std::vector<int64_t> dims;
  int dtype = 0;
  std::vector<float> f;
  std::vector<int32_t> i;
  size_t numel() const { size_t n = 1; for (int64_t d : dims) n *= (size_t)d; return n; }
};
static bool read_fixture(const char* path, std::map<std::string, Rec>& out) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return false;
  for (;;) {
    uint32_t nl;
    if (fread(&nl, 4, 1, fp) != 1) break;
    std::string name(nl, 0);
    if (fread(&name[0], 1, nl, fp) != nl) return false;
    uint32_t code, nd;
    if (fread(&code, 4, 1, fp) != 1 || fread(&nd, 4, 1, fp) != 1) return false;
    Rec r;
    r.dtype = (int)code;
    r.dims.resize(nd);
    if (nd && fread(r.dims.data(), 8, nd, fp) != nd) return false;
    const size_t n = r.numel();
    if (code) { r.i.resize(n); if (fread(r.i.data(), 4, n, fp) != n) return false; }
    else { r.f.resize(n); if (fread(r.f.data(), 4, n, fp) != n) return false; }
    out[name] = std::move(r);
  }
  fclose(fp);
  return true;
}

static int n_fail = 0;
static void verdict(const char* what, bool ok, const char* detail) {
  printf("  %-44s %s  %s\n", what, ok ? "PASS" : "FAIL", detail);
  if (!ok) ++n_fail;
}

static int run_case(const char* path) {
  std::map<std::string, Rec> fx;
  if (!read_fixture(path, fx)) { printf("cannot read %s\n", path); return 2; }
  const Rec& cf = fx["config"];
  const int n = cf.i[0], H = cf.i[1], W = cf.i[2], mn = cf.i[3], mx = cf.i[4], C = cf.i[5];
  printf("== %s: %d image(s) %dx%d, min/max size %d/%d, %d classes\n", path, n, H, W, mn, mx, C);
  mp_detector_config cfg;
  MP_OKAY(mp_detector_default_config(&cfg, C, mn, mx));
  // weights
  std::vector<std::vector<float>> store;
  std::vector<std::string> names;
  for (int i = 0;; ++i) {
    char name[160];
    int64_t shp[4];
    int32_t nd;
    const int rc = mp_detector_state_spec(C, i, name, sizeof(name), shp, &nd);
    if (rc == 1) break;
    MP_OKAY(rc);
    names.emplace_back(name);
    store.emplace_back();
    synthetic_tensor(names.back(), shp, nd, store.back());
  }
  std::vector<mp_named_tensor> st(names.size());
  for (size_t i = 0; i < names.size(); ++i) st[i] = {names[i].c_str(), store[i].data(), (int64_t)store[i].size()};
  mp_detector* det = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  MP_OKAY(mp_detector_create(&cfg, st.data(), (int)st.size(), &det));
  printf("  create: %.2f s (%zu tensors)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), st.size());
  store.clear();
  std::vector<float> img;
  synthetic_images(n, H, W, img);
  float* d_img = nullptr;
  HIP_OK(hipMalloc(&d_img, img.size() * 4));
  HIP_OK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  const int D = cfg.box_detections_per_img, R = cfg.rpn_post_nms_top_n;
  const size_t ws_bytes = mp_detector_workspace_bytes(det, n, H, W);
  printf("  workspace %.1f MB\n", ws_bytes / 1e6);
  void* d_ws = nullptr;
  float *d_boxes = nullptr, *d_scores = nullptr, *d_masks = nullptr;
  int32_t *d_labels = nullptr, *d_counts = nullptr;
  HIP_OK(hipMalloc(&d_ws, ws_bytes));
  HIP_OK(hipMalloc(&d_boxes, (size_t)n * D * 16));
  HIP_OK(hipMalloc(&d_scores, (size_t)n * D * 4));
  HIP_OK(hipMalloc(&d_labels, (size_t)n * D * 4));
  HIP_OK(hipMalloc(&d_counts, (size_t)n * 4 + 64));
  HIP_OK(hipMalloc(&d_masks, (size_t)n * D * H * W * 4));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {   // second run: warm, timed, and must reproduce the first bit for bit
    HIP_OK(hipEventRecord(e0, nullptr));
    MP_OKAY(mp_detector_forward(det, d_img, n, H, W, d_boxes, d_scores, d_labels, d_counts, d_masks, d_ws, ws_bytes, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("  forward #%d: %.2f ms\n", rep, ms);
  }
  auto fetch = [&](const char* what, std::vector<float>& host, int64_t shp[4], int32_t& border, int64_t& rs) -> int {
    const void* p = nullptr;
    int64_t n_el = 0;
    MP_OKAY(mp_detector_debug_tensor(det, what, &p, shp, &border, &rs, &n_el));
    host.resize((size_t)n_el);
    HIP_OK(hipMemcpy(host.data(), p, (size_t)n_el * 4, hipMemcpyDeviceToHost));
    return 0;
  };
  char msg[256];
  // ---- preprocessed input and pyramid -------------------------------------------------------------------------------------------
  {
    std::vector<float> h;
    int64_t s4[4]; int32_t b; int64_t rs;
    if (fetch("x0", h, s4, b, rs)) return 2;
    const Rec& ref = fx["batch"];   // [n,3,Hp,Wp]
    const int Hp = (int)ref.dims[2], Wp = (int)ref.dims[3];
    double err = 0;
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c)
        for (int y = 0; y < Hp; ++y)
          for (int x = 0; x < Wp; ++x) {
            const float g = h[(((size_t)i * (Hp + 6) + y + 3) * (Wp + 6) + x + 3) * 4 + c];
            err = std::fmax(err, std::fabs(g - ref.f[(((size_t)i * 3 + c) * Hp + y) * Wp + x]));
          }
    snprintf(msg, sizeof(msg), "max abs err %.3g (padded %dx%d)", err, Hp, Wp);
    verdict("transform (normalise / resize / pad)", (int)s4[1] == Hp && (int)s4[2] == Wp && err < 2e-5, msg);
  }
  for (int l = 2; l <= 6; ++l) {
    std::vector<float> h;
    int64_t s4[4]; int32_t b; int64_t rs;
    const std::string nm = "P" + std::to_string(l);
    if (fetch(nm.c_str(), h, s4, b, rs)) return 2;
    const Rec& ref = fx[nm];   // [n,h,w,256]
    const int fh = (int)ref.dims[1], fw = (int)ref.dims[2];
    double err = 0, scale = 0;
    bool shape_ok = (int)s4[1] == fh && (int)s4[2] == fw;
    if (shape_ok)
      for (int i = 0; i < n; ++i)
        for (int y = 0; y < fh; ++y)
          for (int x = 0; x < fw; ++x)
            for (int c = 0; c < 256; ++c) {
              const float g = h[(((size_t)i * (fh + 2) + y + 1) * (fw + 2) + x + 1) * 256 + c];
              const float r = ref.f[(((size_t)i * fh + y) * fw + x) * 256 + c];
              err = std::fmax(err, std::fabs(g - r));
              scale = std::fmax(scale, std::fabs(r));
            }
    snprintf(msg, sizeof(msg), "%dx%d max abs err %.3g, scale %.3g (rel %.2g)", fh, fw, err, scale, err / std::fmax(scale, 1e-9));
    verdict((nm + " (ResNet-50 + FPN)").c_str(), shape_ok && err < 1e-3 * std::fmax(scale, 1.0), msg);
  }
  // ---- proposals --------------------------------------------------------------------------------------------------------------------
  {
    std::vector<float> pb, ps, pc;
    int64_t s4[4]; int32_t b; int64_t rs;
    if (fetch("proposals", pb, s4, b, rs) || fetch("proposal_scores", ps, s4, b, rs) || fetch("proposal_counts", pc, s4, b, rs)) return 2;
    const Rec &rb = fx["proposals"], &rc = fx["proposal_counts"], &rsc = fx["proposal_scores"];
    for (int i = 0; i < n; ++i) {
      const int got = ((const int32_t*)pc.data())[i], want = rc.i[i];
      int same_pos = 0, matched = 0;
      const int m = std::min(got, want);
      for (int r = 0; r < m; ++r) {
        const float* g = &pb[((size_t)i * R + r) * 4];
        const float* w = &rb.f[((size_t)i * R + r) * 4];
        double e = 0;
        for (int k = 0; k < 4; ++k) e = std::fmax(e, std::fabs(g[k] - w[k]));
        same_pos += (e < 2e-2 && std::fabs(ps[(size_t)i * R + r] - rsc.f[(size_t)i * R + r]) < 1e-4);
      }
      for (int r = 0; r < want; ++r) {   // order-free: every oracle proposal has a twin somewhere in ours
        const float* w = &rb.f[((size_t)i * R + r) * 4];
        for (int q = 0; q < got; ++q) {
          const float* g = &pb[((size_t)i * R + q) * 4];
          if (std::fabs(g[0] - w[0]) < 2e-2 && std::fabs(g[1] - w[1]) < 2e-2 && std::fabs(g[2] - w[2]) < 2e-2 && std::fabs(g[3] - w[3]) < 2e-2) { ++matched; break; }
        }
      }
      snprintf(msg, sizeof(msg), "image %d: %d vs %d proposals, %d identical in place, %d of the oracle's found", i, got, want, same_pos, matched);
      verdict("RPN proposals", std::abs(got - want) <= want / 50 + 2 && matched >= want - want / 50 - 2, msg);
    }
  }
  // ---- class logits / box regression of the rows whose proposals agree in place ------------------------------------------------------
  {
    std::vector<float> cl, pb;
    int64_t s4[4], t4[4]; int32_t b; int64_t rs, rs2;
    if (fetch("class_logits", cl, s4, b, rs) || fetch("proposals", pb, t4, b, rs2)) return 2;
    const Rec &rl = fx["class_logits"], &rr = fx["box_regression"], &rb = fx["proposals"], &rc = fx["proposal_counts"];
    double e_l = 0, e_r = 0, sc_l = 0;
    int rows = 0, off = 0;
    for (int i = 0; i < n; ++i) {
      for (int r = 0; r < rc.i[i]; ++r) {
        const float* g = &pb[((size_t)i * R + r) * 4];
        const float* w = &rb.f[((size_t)i * R + r) * 4];
        if (std::fabs(g[0] - w[0]) > 1e-3 || std::fabs(g[1] - w[1]) > 1e-3 || std::fabs(g[2] - w[2]) > 1e-3 || std::fabs(g[3] - w[3]) > 1e-3) continue;
        ++rows;
        const float* row = &cl[((size_t)i * R + r) * rs];
        for (int c = 0; c < C; ++c) {
          e_l = std::fmax(e_l, std::fabs(row[c] - rl.f[(size_t)(off + r) * C + c]));
          sc_l = std::fmax(sc_l, std::fabs(rl.f[(size_t)(off + r) * C + c]));
        }
        for (int c = 0; c < 4 * C; ++c) e_r = std::fmax(e_r, std::fabs(row[C + c] - rr.f[(size_t)(off + r) * 4 * C + c]));
      }
      off += rc.i[i];
    }
    snprintf(msg, sizeof(msg), "%d rows: logits err %.3g (scale %.3g), deltas err %.3g", rows, e_l, sc_l, e_r);
    verdict("RoIAlign + box head (fc6, fc7, predictor)", rows > 0 && e_l < 1e-3 * std::fmax(sc_l, 1.0) && e_r < 1e-3, msg);
  }
  // ---- detections ---------------------------------------------------------------------------------------------------------------------
  std::vector<float> hb((size_t)n * D * 4), hs((size_t)n * D);
  std::vector<int32_t> hl((size_t)n * D), hc(n);
  HIP_OK(hipMemcpy(hb.data(), d_boxes, hb.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hs.data(), d_scores, hs.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hl.data(), d_labels, hl.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hc.data(), d_counts, hc.size() * 4, hipMemcpyDeviceToHost));
  const Rec &ob = fx["boxes"], &os = fx["scores"], &ol = fx["labels"], &oc = fx["counts"];
  std::vector<int> twin((size_t)n * D, -1);   // oracle detection -> ours
  for (int i = 0; i < n; ++i) {
    int found = 0, in_place = 0;
    for (int r = 0; r < oc.i[i]; ++r) {
      const float* w = &ob.f[((size_t)i * D + r) * 4];
      for (int q = 0; q < hc[i]; ++q) {
        const float* g = &hb[((size_t)i * D + q) * 4];
        if (hl[(size_t)i * D + q] == ol.i[(size_t)i * D + r] && std::fabs(hs[(size_t)i * D + q] - os.f[(size_t)i * D + r]) < 2e-4 &&
            std::fabs(g[0] - w[0]) < 5e-2 && std::fabs(g[1] - w[1]) < 5e-2 && std::fabs(g[2] - w[2]) < 5e-2 && std::fabs(g[3] - w[3]) < 5e-2) {
          twin[(size_t)i * D + r] = q;
          ++found;
          in_place += q == r;
          break;
        }
      }
    }
    snprintf(msg, sizeof(msg), "image %d: %d vs %d detections, %d of the oracle's found (%d in place); first: label %d score %.4f", i, hc[i], oc.i[i],
             found, in_place, hl[(size_t)i * D], hs[(size_t)i * D]);
    verdict("detections (softmax, decode, per-class NMS)", std::abs(hc[i] - oc.i[i]) <= 3 && found >= oc.i[i] - 3 - oc.i[i] / 20, msg);
  }
  // ---- masks ------------------------------------------------------------------------------------------------------------------------
  {
    std::vector<float> ml;
    int64_t s4[4]; int32_t b; int64_t rs;
    if (fetch("mask_logits", ml, s4, b, rs)) return 2;
    const Rec &m28 = fx["masks28"], &mp = fx["masks_pasted_first8"];
    double e28 = 0, ep = 0;
    int n28 = 0, np_ = 0;
    std::vector<float> pasted((size_t)H * W);
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < oc.i[i]; ++r) {
        const int q = twin[(size_t)i * D + r];
        if (q < 0) continue;
        const int det = i * D + q, lab = hl[(size_t)i * D + q];
        ++n28;
        for (int my = 0; my < 28; ++my)
          for (int mx_ = 0; mx_ < 28; ++mx_) {
            const size_t row = (((size_t)det * 14 + (my >> 1)) * 14 + (mx_ >> 1)) * 4 + ((my & 1) * 2 + (mx_ & 1));
            const float prob = 1.f / (1.f + std::exp(-ml[row * rs + lab]));
            e28 = std::fmax(e28, std::fabs(prob - m28.f[(((size_t)i * D + r) * 28 + my) * 28 + mx_]));
          }
        if (r < 8) {
          ++np_;
          HIP_OK(hipMemcpy(pasted.data(), d_masks + (size_t)det * H * W, pasted.size() * 4, hipMemcpyDeviceToHost));
          for (size_t k = 0; k < pasted.size(); ++k) ep = std::fmax(ep, std::fabs(pasted[k] - mp.f[((size_t)i * 8 + r) * H * W + k]));
        }
      }
    snprintf(msg, sizeof(msg), "%d masks: 28x28 prob err %.3g; %d pasted: err %.3g", n28, e28, np_, ep);
    verdict("mask head + paste", n28 > 0 && e28 < 2e-3 && (np_ == 0 || ep < 5e-2), msg);
  }
  // ---- determinism ------------------------------------------------------------------------------------------------------------------
  {
    std::vector<float> hb2(hb.size());
    MP_OKAY(mp_detector_forward(det, d_img, n, H, W, d_boxes, d_scores, d_labels, d_counts, nullptr, d_ws, ws_bytes, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(hb2.data(), d_boxes, hb2.size() * 4, hipMemcpyDeviceToHost));
    verdict("bit-reproducible (and masks optional)", memcmp(hb.data(), hb2.data(), hb.size() * 4) == 0, "");
  }
  mp_detector_destroy(det);
  (void)hipFree(d_img); (void)hipFree(d_ws); (void)hipFree(d_boxes); (void)hipFree(d_scores); (void)hipFree(d_labels); (void)hipFree(d_counts); (void)hipFree(d_masks);
  return 0;
}

// --checksums C: print a checksum of every synthetic tensor and of the synthetic image (host only; compared with the oracle's
// generator by tests/test_detector_cpu.py so that both sides provably run the same network on the same input)
static int dump_checksums(int C) {
  for (int i = 0;; ++i) {
    char name[160];
    int64_t shp[4];
    int32_t nd;
    const int rc = mp_detector_state_spec(C, i, name, sizeof(name), shp, &nd);
    if (rc == 1) break;
    MP_OKAY(rc);
    std::vector<float> v;
    synthetic_tensor(name, shp, nd, v);
    uint64_t acc = 0;
    for (size_t k = 0; k < v.size(); ++k) { uint32_t b; memcpy(&b, &v[k], 4); acc += (uint64_t)b * (k % 251 + 1); }
    printf("%s %zu %llu\n", name, v.size(), (unsigned long long)acc);
  }
  std::vector<float> img;
  synthetic_images(1, 24, 32, img);
  uint64_t acc = 0;
  for (size_t k = 0; k < img.size(); ++k) { uint32_t b; memcpy(&b, &img[k], 4); acc += (uint64_t)b * (k % 251 + 1); }
  printf("images/1x24x32 %zu %llu\n", img.size(), (unsigned long long)acc);
  return 0;
}

// --time n H W C: forward time of the synthetic network at a given size (no oracle involved), with the library's per-kernel profile
static int time_case(int n, int H, int W, int C) {
  mp_detector_config cfg;
  MP_OKAY(mp_detector_default_config(&cfg, C, std::min(H, W), std::max(H, W)));
  std::vector<std::vector<float>> store;
  std::vector<std::string> names;
  for (int i = 0;; ++i) {
    char name[160];
    int64_t shp[4];
    int32_t nd;
    const int rc = mp_detector_state_spec(C, i, name, sizeof(name), shp, &nd);
    if (rc == 1) break;
    MP_OKAY(rc);
    names.emplace_back(name);
    store.emplace_back();
    synthetic_tensor(names.back(), shp, nd, store.back());
  }
  std::vector<mp_named_tensor> st(names.size());
  for (size_t i = 0; i < names.size(); ++i) st[i] = {names[i].c_str(), store[i].data(), (int64_t)store[i].size()};
  mp_detector* det = nullptr;
  MP_OKAY(mp_detector_create(&cfg, st.data(), (int)st.size(), &det));
  std::vector<float> img;
  synthetic_images(n, H, W, img);
  float* d_img = nullptr;
  HIP_OK(hipMalloc(&d_img, img.size() * 4));
  HIP_OK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  const int D = cfg.box_detections_per_img;
  const size_t ws_bytes = mp_detector_workspace_bytes(det, n, H, W);
  void* d_ws = nullptr;
  float *d_boxes = nullptr, *d_scores = nullptr, *d_masks = nullptr;
  int32_t *d_labels = nullptr, *d_counts = nullptr;
  HIP_OK(hipMalloc(&d_ws, ws_bytes));
  HIP_OK(hipMalloc(&d_boxes, (size_t)n * D * 16));
  HIP_OK(hipMalloc(&d_scores, (size_t)n * D * 4));
  HIP_OK(hipMalloc(&d_labels, (size_t)n * D * 4));
  HIP_OK(hipMalloc(&d_counts, (size_t)n * 4 + 64));
  HIP_OK(hipMalloc(&d_masks, (size_t)n * D * H * W * 4));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  printf("== timing: %d image(s) %dx%d, %d classes, workspace %.0f MB\n", n, H, W, C, ws_bytes / 1e6);
  for (int with_masks = 1; with_masks >= 0; --with_masks) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      float ms = 0.f;
      HIP_OK(hipEventRecord(e0, nullptr));
      MP_OKAY(mp_detector_forward(det, d_img, n, H, W, d_boxes, d_scores, d_labels, d_counts, with_masks ? d_masks : nullptr, d_ws, ws_bytes, nullptr));
      HIP_OK(hipEventRecord(e1, nullptr));
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::fmin(best, ms);
    }
    printf("  forward %s: %.2f ms (best of 3 warm runs)\n", with_masks ? "with masks   " : "without masks", best);
  }
  MP_OKAY(mp_profile_begin());
  MP_OKAY(mp_detector_forward(det, d_img, n, H, W, d_boxes, d_scores, d_labels, d_counts, d_masks, d_ws, ws_bytes, nullptr));
  MP_OKAY(mp_profile_end());
  for (int i = 0;; ++i) {
    char name[128];
    int64_t launches;
    double ms, fl, by;
    if (mp_profile_query(i, name, sizeof(name), &launches, &ms, &fl, &by) == 1) break;
    printf("    %-44s %4ld launches %8.3f ms  %7.1f TFLOP/s\n", name, (long)launches, ms, ms > 0 ? fl / ms / 1e9 : 0.0);
  }
  std::vector<int32_t> hc(n);
  HIP_OK(hipMemcpy(hc.data(), d_counts, (size_t)n * 4, hipMemcpyDeviceToHost));
  printf("  detections per image:");
  for (int i = 0; i < n; ++i) printf(" %d", hc[i]);
  printf("\n");
  mp_detector_destroy(det);
  (void)hipFree(d_img); (void)hipFree(d_ws); (void)hipFree(d_boxes); (void)hipFree(d_scores); (void)hipFree(d_labels); (void)hipFree(d_counts); (void)hipFree(d_masks);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--checksums") return dump_checksums(atoi(argv[2]));
  if (argc == 6 && std::string(argv[1]) == "--time") return time_case(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
  int n_cu = 0, lds = 0;
  char arch[64];
  MP_OKAY(mp_device_info(&n_cu, &lds, arch, sizeof(arch)));
  printf("device %s, %d CUs\n", arch, n_cu);
  for (int i = 1; i < argc; ++i) {
    const int rc = run_case(argv[i]);
    if (rc) { printf("case %s aborted (rc %d)\n", argv[i], rc); ++n_fail; }
  }
  printf(n_fail ? "RESULT: %d check(s) FAILED\n" : "RESULT: all checks passed\n", n_fail);
  return n_fail ? 1 : 0;
}
