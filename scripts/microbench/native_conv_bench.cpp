// native_conv_bench.cpp -- torch-free A/B bench of mp_conv2d_nhwc on the layer shapes of the config-2 workload (starts in
// milliseconds: a whole variant sweep fits ~15 s of GPU-box time through gpurun, where a Python process spends that importing).
//   native_conv_bench                      one process per variant in VARIANTS (re-executes itself with MP_CONV_VARIANT set, because the
//                                          library reads the variable once), then a table: TFLOP/s per (shape, variant), the in-kernel
//                                          shader clock, and whether every variant's output is bit-identical to the default's
//   native_conv_bench --one                the current process' variant only (what the children run)
//   native_conv_bench --variants 8449,257  another variant list
// Operands are N(0,1) activations through a ReLU (half zeros) and N(0, 2/K) weights: real data toggles the datapath and costs clock
// (DESIGN.md 3.1); all-zero operands would flatter every variant alike.
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/microbench/native_conv_bench.cpp \
//              -o scripts/microbench/_build/native_conv_bench -Lmegapose6d_amd -lmp_engine -Wl,-rpath,'$ORIGIN/../../../megapose6d_amd'
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
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

struct Shape {
  const char* name;
  int N, C, H, W, Cout, K, stride, pad, in_border, residual;
};
// the convolutions that carry the config-2 step (576 rows): stem of the refiner, layer1..layer4 3x3
static const Shape SHAPES[] = {
    {"stem 7x7 27->64 @240x320", 576, 28, 240, 320, 64, 7, 2, 3, 3, 0},
    {"layer1 3x3 64->64 @60x80", 576, 64, 60, 80, 64, 3, 1, 1, 1, 1},
    {"layer2 3x3 128->128 @30x40", 576, 128, 30, 40, 128, 3, 1, 1, 1, 1},
    {"layer3 3x3 256->256 @15x20", 576, 256, 15, 20, 256, 3, 1, 1, 1, 1},
    {"layer4 3x3 512->512 @8x10", 576, 512, 8, 10, 512, 3, 1, 1, 1, 1},
};

static int run_one() {
  const char* v = getenv("MP_CONV_VARIANT");
  printf("VARIANT %s\n", v ? v : "default");
  for (const Shape& s : SHAPES) {
    const int Hp = s.H + 2 * s.in_border, Wp = s.W + 2 * s.in_border;
    const int Ho = (s.H + 2 * s.pad - s.K) / s.stride + 1, Wo = (s.W + 2 * s.pad - s.K) / s.stride + 1;
    const size_t n_in = (size_t)s.N * Hp * Wp * s.C + (size_t)Wp * s.C + 64;
    const size_t n_out = (size_t)s.N * (Ho + 2) * (Wo + 2) * s.Cout + 64;
    std::mt19937 rng(s.C * 7 + s.K);
    std::normal_distribution<float> G(0.f, 1.f);
    // an 8-image pattern generated on the host and repeated over the batch ON THE DEVICE (the 576-row stem input is 5 GB)
    const int n_gen = std::min(s.N, 8);
    const size_t per = (size_t)Hp * Wp * s.C;
    std::vector<float> x((size_t)n_gen * per, 0.f);
    for (int n = 0; n < n_gen; ++n)
      for (int y = 0; y < s.H; ++y)
        for (int xx = 0; xx < s.W; ++xx)
          for (int c = 0; c < s.C; ++c) x[(((size_t)n * Hp + y + s.in_border) * Wp + xx + s.in_border) * s.C + c] = std::fmax(G(rng), 0.f);
    std::vector<float> w((size_t)s.Cout * s.C * s.K * s.K), bias(s.Cout);
    const float a = std::sqrt(2.f / (s.C * s.K * s.K));
    for (auto& q : w) q = G(rng) * a;
    for (auto& q : bias) q = 0.1f * G(rng);
    std::vector<float> packed(mp_conv_packed_floats(s.C, s.Cout, s.K, s.K));
    MP_OKAY(mp_conv_pack_weights(w.data(), s.Cout, s.C, s.K, s.K, s.C, nullptr, packed.data()));
    float *d_x, *d_w, *d_b, *d_y, *d_r = nullptr, *d_sk;
    HIP_OK(hipMalloc(&d_x, n_in * 4));
    HIP_OK(hipMalloc(&d_w, packed.size() * 4));
    HIP_OK(hipMalloc(&d_b, bias.size() * 4));
    HIP_OK(hipMalloc(&d_y, n_out * 4));
    HIP_OK(hipMalloc(&d_sk, (size_t)(12u << 20) * 4));
    HIP_OK(hipMemset(d_x, 0, n_in * 4));
    HIP_OK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    for (int n = n_gen; n < s.N; n += n_gen)
      HIP_OK(hipMemcpy(d_x + (size_t)n * per, d_x, (size_t)std::min(n_gen, s.N - n) * per * 4, hipMemcpyDeviceToDevice));
    HIP_OK(hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_y, 0, n_out * 4));
    if (s.residual) {
      HIP_OK(hipMalloc(&d_r, n_out * 4));
      HIP_OK(hipMemcpy(d_r, d_x, std::min(n_in, n_out) * 4, hipMemcpyDeviceToDevice));   // same geometry for the 3x3 stride-1 layers
    }
    mp_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.d_x = d_x; d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.in_border = s.in_border;
    d.d_w = d_w; d.d_bias = d_b; d.Cout = s.Cout; d.KH = s.K; d.KW = s.K; d.stride = s.stride; d.pad = s.pad;
    d.d_y = d_y; d.out_border = 1; d.relu = 1; d.d_residual = d_r;
    d.d_splitk_ws = d_sk; d.splitk_ws_floats = 12u << 20;   // the backbone executor always offers the scratch: same launch plan as in the product
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
    HIP_OK(hipDeviceSynchronize());
    double mhz = 0;
    MP_OKAY(mp_conv_clock_read(&mhz, 1));
    const int reps = 6;
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) MP_OKAY(mp_conv2d_nhwc(&d, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipDeviceSynchronize());
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    MP_OKAY(mp_conv_clock_read(&mhz, 1));
    // checksum of the first 16 and the last output image (every tile position modulo the batch pattern, incl. the grid's tail)
    const size_t per_out = (size_t)(Ho + 2) * (Wo + 2) * s.Cout, n_head = std::min<size_t>(16, s.N) * per_out;
    std::vector<float> y(n_head + per_out);
    HIP_OK(hipMemcpy(y.data(), d_y, n_head * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(y.data() + n_head, d_y + (size_t)(s.N - 1) * per_out, per_out * 4, hipMemcpyDeviceToHost));
    uint64_t ck = 0;
    for (size_t i = 0; i < y.size(); ++i) { uint32_t b; memcpy(&b, &y[i], 4); ck += (uint64_t)b * (i % 1021 + 1); }
    const double flops = 2.0 * s.N * Ho * Wo * (double)s.Cout * s.K * s.K * s.C;
    printf("ROW %-30s | %8.3f ms | %7.2f TFLOP/s | %6.0f MHz | ck %016llx\n", s.name, ms / reps, flops * reps / (ms * 1e-3) / 1e12, mhz,
           (unsigned long long)ck);
    (void)hipFree(d_x); (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_y); (void)hipFree(d_sk);
    if (d_r) (void)hipFree(d_r);
  }
  return 0;
}

int main(int argc, char** argv) {
  std::string variants = "8449,257,4097,12289";
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--one")) return run_one();
    if (!strcmp(argv[i], "--variants") && i + 1 < argc) variants = argv[++i];
  }
  // parent: one child per variant (the library latches MP_CONV_VARIANT at its first convolution)
  std::vector<std::string> list;
  for (size_t p = 0; p <= variants.size();) {
    const size_t q = variants.find(',', p);
    list.push_back(variants.substr(p, q == std::string::npos ? std::string::npos : q - p));
    if (q == std::string::npos) break;
    p = q + 1;
  }
  std::vector<std::vector<std::string>> cks(list.size());
  for (size_t v = 0; v < list.size(); ++v) {
    setenv("MP_CONV_VARIANT", list[v].c_str(), 1);
    const std::string cmd = std::string(argv[0]) + " --one";
    FILE* fp = popen(cmd.c_str(), "r");
    if (!fp) { printf("cannot start %s\n", cmd.c_str()); return 2; }
    char line[512];
    while (fgets(line, sizeof(line), fp)) {
      fputs(line, stdout);
      const char* c = strstr(line, "| ck ");
      if (!strncmp(line, "ROW ", 4) && c) cks[v].push_back(std::string(c + 5, 16));
    }
    const int rc = pclose(fp);
    if (rc) printf("variant %s: child exited with %d\n", list[v].c_str(), rc);
  }
  int bad = 0;
  for (size_t v = 1; v < list.size(); ++v) {
    const bool same = cks[v] == cks[0] && !cks[0].empty();
    printf("variant %s vs %s: outputs %s\n", list[v].c_str(), list[0].c_str(), same ? "bit-identical" : "DIFFER");
    bad += !same;
  }
  return bad ? 1 : 0;
}
