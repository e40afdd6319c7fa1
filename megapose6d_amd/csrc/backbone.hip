// backbone.hip -- native executor for the CNN backbones + heads (host-side C++; kernels live in conv.hip / pool_fc.hip).
//
// Reference graph: src/megapose/models/torchvision_resnet.py:181-316 (vanilla_resnet34 = torchvision ResNet-34 with
// n_input_channels and fc 512->512), src/megapose/models/wide_resnet.py:29-126 (pre-activation WideResNet18/34),
// heads src/megapose/models/pose_rigid.py:122-130, selection src/megapose/training/pose_models_cfg.py:90-138.
// Weights arrive as the reference checkpoint's state_dict (SURVEY.md App. F key layout); eval-mode BatchNorm is folded
// into the preceding conv at load time (w' = w * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps)); the WideResNet
// block-input BN (bn1) cannot fold and is emitted by the PRODUCER's epilogue as a second "activated" output.
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace mp {

struct ConvLayer {
  int Cin, Cin_p, Cout, K, stride, pad;
  float* d_w = nullptr;   // packed fp32 weights
  float* d_u = nullptr;   // Winograd-transformed weights of an eligible 3x3 / stride-1 layer: fp32 (conv_wino.hip, MP_CONV_WINO=1) ...
  void* d_ub = nullptr;   // ... or split into three exact bf16 pieces (conv_wino_bf16.hip, the default)
  float* d_b = nullptr;  // folded BN shift (may be null)
};

struct BnAct {  // unfoldable pre-activation BN: relu(x*scale + shift)
  float* d_scale = nullptr;
  float* d_shift = nullptr;
};

struct Block {
  ConvLayer conv1, conv2, down;
  bool has_down = false;
  BnAct pre;   // WideResNet: bn1 of THIS block (applied by the producer of its input)
};

}  // namespace mp

using namespace mp;

struct mp_backbone {
  int kind, c_in, c_in_p, in_border, head_kind, n_out, n_feat;
  int width = 1;      // WideResNet width multiplier (`resnet34_width=N`, training/pose_models_cfg.py:114-116): stage widths 64N .. 512N
  int stageC[4] = {64, 128, 256, 512};
  bool wide;
  ConvLayer stem;
  std::vector<Block> blocks;
  std::vector<int> stage_of_block;  // 0..3
  float *d_fc_w = nullptr, *d_fc_b = nullptr, *d_head_w = nullptr, *d_head_b = nullptr;
  std::vector<void*> allocs;
  // exact-piece bf16 stem (conv_stem.hip): the stem's OIHW weights + folded BN scale stay on the host so that the piece blob can be
  // packed for the record layout (number of fp32-kind channels) the caller's rasteriser launch writes
  std::vector<float> stem_w_host, stem_scale_host;
  std::mutex blob_mu;                     // guards the two maps below: prepare (one thread) vs forwards looking blobs up on other threads / streams
  std::map<uint32_t, void*> stem_blobs;   // f32-kind channel mask -> device blob (mp_backbone_xrec_prepare); never freed before destroy
  std::map<uint32_t, void*> stem_blobs_sparse;   // ... -> blob of the background-tile walk (leading-channel masks with something to skip)
  // workspace bookkeeping: borders are zeroed once per (pointer, batch, h, w); several workspaces may be live at once
  // (one per HIP stream when half-batches are interleaved on two streams)
  struct WsKey { void* ptr; int batch, h, w; };
  std::vector<WsKey> ws_known;
};

namespace {

typedef std::map<std::string, std::pair<const float*, int64_t>> StateMap;

int upload(mp_backbone* bb, const std::vector<float>& h, float** d) {
  MP_CHECK_HIP(hipMalloc(d, h.size() * sizeof(float)));
  MP_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  bb->allocs.push_back(*d);
  return MP_OK;
}

const float* find(const StateMap& sm, const std::string& k, int64_t numel) {
  auto it = sm.find(k);
  if (it == sm.end()) {
    set_error("mp_backbone_create: missing state_dict key '%s'", k.c_str());
    return nullptr;
  }
  if (it->second.second != numel) {
    set_error("mp_backbone_create: key '%s' has %ld elements, expected %ld", k.c_str(), (long)it->second.second, (long)numel);
    return nullptr;
  }
  return it->second.first;
}

// eval BatchNorm -> per-channel (scale, shift)
int bn_affine(const StateMap& sm, const std::string& prefix, int C, std::vector<float>& scale, std::vector<float>& shift) {
  const float* g = find(sm, prefix + ".weight", C);
  const float* b = find(sm, prefix + ".bias", C);
  const float* m = find(sm, prefix + ".running_mean", C);
  const float* v = find(sm, prefix + ".running_var", C);
  if (!g || !b || !m || !v) return MP_ERR_INVALID;
  scale.resize(C);
  shift.resize(C);
  for (int c = 0; c < C; ++c) {
    const float s = g[c] / sqrtf(v[c] + 1e-5f);
    scale[c] = s;
    shift[c] = b[c] - m[c] * s;
  }
  return MP_OK;
}

int make_conv(mp_backbone* bb, const StateMap& sm, const std::string& wkey, const std::string& bnkey /* "" = none */, int Cin,
              int Cin_p, int Cout, int K, int stride, int pad, ConvLayer* L) {
  L->Cin = Cin; L->Cin_p = Cin_p; L->Cout = Cout; L->K = K; L->stride = stride; L->pad = pad;
  const float* w = find(sm, wkey, (int64_t)Cout * Cin * K * K);
  if (!w) return MP_ERR_INVALID;
  std::vector<float> scale, shift;
  if (!bnkey.empty()) {
    int rc = bn_affine(sm, bnkey, Cout, scale, shift);
    if (rc) return rc;
  }
  int rc;
  if (L == &bb->stem) {
    bb->stem_w_host.assign(w, w + (size_t)Cout * Cin * K * K);
    bb->stem_scale_host = scale;
  }
  std::vector<float> packed(mp_conv_packed_floats(Cin_p, Cout, K, K));
  rc = mp_conv_pack_weights(w, Cout, Cin, K, K, Cin_p, bnkey.empty() ? nullptr : scale.data(), packed.data());
  if (rc) return rc;
  rc = upload(bb, packed, &L->d_w);
  // 3x3 / stride-1 layers of the residual stages also get their Winograd F(2x2, 3x3) form.  MP_CONV_WINO: 2 (default) = the bf16x9
  // exact-piece kernel, 1 = the fp32-MFMA kernel, 0 = keep the direct kernel
  static const int wino_mode = getenv("MP_CONV_WINO") ? atoi(getenv("MP_CONV_WINO")) : 2;
  if (!rc && wino_mode != 0 && K == 3 && stride == 1 && pad == 1 && Cin_p % 16 == 0 && Cout % 64 == 0) {
    if (wino_mode == 1) {
      std::vector<float> u(mp_conv_wino_packed_floats(Cin_p, Cout));
      rc = mp_conv_wino_pack_weights(w, Cout, Cin, Cin_p, bnkey.empty() ? nullptr : scale.data(), u.data());
      if (!rc) rc = upload(bb, u, &L->d_u);
    } else {
      std::vector<float> u((mp_conv_wino_bf16_packed_bytes(Cin_p, Cout) + 3) / 4);
      rc = mp_conv_wino_bf16_pack_weights(w, Cout, Cin, Cin_p, bnkey.empty() ? nullptr : scale.data(), u.data());
      float* d = nullptr;
      if (!rc) rc = upload(bb, u, &d);
      L->d_ub = d;
    }
  }
  if (rc) return rc;
  if (!bnkey.empty()) {
    rc = upload(bb, shift, &L->d_b);
    if (rc) return rc;
  }
  return MP_OK;
}

int make_bnact(mp_backbone* bb, const StateMap& sm, const std::string& bnkey, int C, BnAct* a) {
  std::vector<float> scale, shift;
  int rc = bn_affine(sm, bnkey, C, scale, shift);
  if (rc) return rc;
  rc = upload(bb, scale, &a->d_scale);
  if (rc) return rc;
  return upload(bb, shift, &a->d_shift);
}

constexpr size_t SPLITK_WS_FLOATS = 12u << 20;  // 48 MB of split-K scratch at the end of the workspace (>= 512 tiles of 128 x 128)

int run_conv(const mp_backbone* bb, const ConvLayer& L, const float* x, int N, int H, int W, int in_border, float* y, int out_border,
             const float* res, int relu, float* y_act, const BnAct* act, hipStream_t s, float* splitk_ws = nullptr, bool x_f16 = false) {
  mp_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x_f16 = x_f16 ? 1 : 0;
  d.d_x = x; d.N = N; d.H = H; d.W = W; d.C = L.Cin_p; d.c_real = L.Cin; d.in_border = in_border;
  d.d_w = L.d_w; d.d_bias = L.d_b; d.Cout = L.Cout; d.KH = L.K; d.KW = L.K; d.stride = L.stride; d.pad = L.pad;
  d.d_y = y; d.out_border = out_border; d.d_residual = res; d.relu = relu;
  d.d_y_act = y_act;
  if (y_act) { d.d_act_scale = act->d_scale; d.d_act_shift = act->d_shift; }
  d.d_splitk_ws = splitk_ws;
  d.splitk_ws_floats = splitk_ws ? (int64_t)SPLITK_WS_FLOATS : 0;
  if ((L.d_u || L.d_ub) && !x_f16) {
    static int n_cu = 0, n_cu_dev = -1, lds_ok = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (n_cu_dev != dev) {
      if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
      int lds = 0;   // the Winograd kernels take 128.5 KB of LDS per workgroup (gfx950: 160 KB per CU); a part with less keeps the direct kernel
      lds_ok = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && lds >= 132 * 1024;
      n_cu_dev = dev;
    }
    if (lds_ok && mp_conv_wino_eligible(&d, n_cu))   // (the workspace buffers carry the read slack the Winograd kernels need)
      return L.d_ub ? mp_conv3x3_wino_bf16_nhwc(&d, L.d_ub, s) : mp_conv3x3_wino_nhwc(&d, L.d_u, s);
  }
  return mp_conv2d_nhwc(&d, s);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// (+ one padded row + one pixel of slack: the Winograd kernel reads -- and discards -- that much past an odd-sized tensor)
inline size_t buf_floats(int N, int H, int W, int C) { return (size_t)N * (H + 2) * (W + 2) * C + (size_t)(W + 3) * C + 64; }

struct Geometry {
  int h1, w1;       // stem output
  int hs[4], ws[4]; // stage resolutions
};

Geometry geometry(const mp_backbone* bb, int h, int w) {
  Geometry g;
  g.h1 = (h + 2 * bb->stem.pad - bb->stem.K) / 2 + 1;
  g.w1 = (w + 2 * bb->stem.pad - bb->stem.K) / 2 + 1;
  g.hs[0] = (g.h1 + 2 - 3) / 2 + 1;
  g.ws[0] = (g.w1 + 2 - 3) / 2 + 1;
  for (int s = 1; s < 4; ++s) {
    g.hs[s] = (g.hs[s - 1] + 2 - 3) / 2 + 1;
    g.ws[s] = (g.ws[s - 1] + 2 - 3) / 2 + 1;
  }
  return g;
}


}  // namespace

extern "C" int mp_backbone_create(int kind, int c_in, int head_kind, int n_head_out, const mp_named_tensor* st, int n_tensors,
                                  mp_backbone** out) {
  return mp_backbone_create_wide(kind, 1, c_in, head_kind, n_head_out, st, n_tensors, out);
}

extern "C" int mp_backbone_create_wide(int kind, int width, int c_in, int head_kind, int n_head_out, const mp_named_tensor* st, int n_tensors,
                                       mp_backbone** out) {
  MP_REQUIRE(out && st && n_tensors > 0, "mp_backbone_create: bad arguments");
  MP_REQUIRE(kind >= 0 && kind <= 2, "mp_backbone_create: unknown backbone kind %d", kind);
  MP_REQUIRE(width >= 1 && width <= 8 && (width == 1 || kind != MP_BACKBONE_VANILLA_RESNET34),
             "mp_backbone_create: width multiplier %d (1..8, WideResNets only: training/pose_models_cfg.py:114-116)", width);
  MP_REQUIRE(c_in >= 1 && c_in <= 512, "mp_backbone_create: bad c_in %d", c_in);   // (sphere_26views: 3 + 27 * 6 = 165 input channels)
  StateMap sm;
  for (int i = 0; i < n_tensors; ++i) sm[st[i].name] = std::make_pair(st[i].h_data, st[i].numel);
  mp_backbone* bb = new mp_backbone();
  bb->kind = kind;
  bb->wide = kind != MP_BACKBONE_VANILLA_RESNET34;
  bb->c_in = c_in;
  bb->c_in_p = (c_in + 3) / 4 * 4;
  bb->head_kind = head_kind;
  bb->n_out = n_head_out;
  bb->width = width;
  for (int k = 0; k < 4; ++k) bb->stageC[k] = bb->stageC[k] * width;
  const int C0 = bb->stageC[0], NF = bb->stageC[3];
  bb->n_feat = NF;
  int rc;
#define MP_TRY(e) do { rc = (e); if (rc) { mp_backbone_destroy(bb); return rc; } } while (0)
  const std::string B = "backbone.";
  if (!bb->wide) {
    bb->in_border = 3;
    MP_TRY(make_conv(bb, sm, B + "conv1.weight", B + "bn1", c_in, bb->c_in_p, C0, 7, 2, 3, &bb->stem));
  } else {
    bb->in_border = 2;
    MP_TRY(make_conv(bb, sm, B + "conv1.weight", B + "bn1", c_in, bb->c_in_p, C0, 5, 2, 2, &bb->stem));
  }
  static const int n34[4] = {3, 4, 6, 3}, n18[4] = {2, 2, 2, 2};
  const int* nblocks = (kind == MP_BACKBONE_WIDE_RESNET18) ? n18 : n34;
  int inplanes = C0;
  for (int s = 0; s < 4; ++s) {
    const int planes = bb->stageC[s];
    for (int i = 0; i < nblocks[s]; ++i) {
      Block blk;
      const int stride = (i == 0 && s > 0) ? 2 : 1;
      const std::string P = B + "layer" + std::to_string(s + 1) + "." + std::to_string(i) + ".";
      blk.has_down = (i == 0) && (stride != 1 || inplanes != planes);
      if (!bb->wide) {
        MP_TRY(make_conv(bb, sm, P + "conv1.weight", P + "bn1", inplanes, inplanes, planes, 3, stride, 1, &blk.conv1));
        MP_TRY(make_conv(bb, sm, P + "conv2.weight", P + "bn2", planes, planes, planes, 3, 1, 1, &blk.conv2));
        if (blk.has_down)
          MP_TRY(make_conv(bb, sm, P + "downsample.0.weight", P + "downsample.1", inplanes, inplanes, planes, 1, stride, 0, &blk.down));
      } else {
        // out = conv2(relu(bn2(conv1(a)))) + residual,  a = relu(bn1(x))   (wide_resnet.py:50-56)
        MP_TRY(make_bnact(bb, sm, P + "bn1", inplanes, &blk.pre));
        MP_TRY(make_conv(bb, sm, P + "conv1.weight", P + "bn2", inplanes, inplanes, planes, 3, stride, 1, &blk.conv1));
        MP_TRY(make_conv(bb, sm, P + "conv2.weight", "", planes, planes, planes, 3, 1, 1, &blk.conv2));
        if (blk.has_down) MP_TRY(make_conv(bb, sm, P + "downsample.weight", "", inplanes, inplanes, planes, 1, stride, 0, &blk.down));
      }
      bb->blocks.push_back(blk);
      bb->stage_of_block.push_back(s);
      inplanes = planes;
    }
  }
  if (!bb->wide) {
    const float* fw = find(sm, B + "fc.weight", 512 * 512);
    const float* fb = find(sm, B + "fc.bias", 512);
    if (!fw || !fb) { mp_backbone_destroy(bb); return MP_ERR_INVALID; }
    MP_TRY(upload(bb, std::vector<float>(fw, fw + 512 * 512), &bb->d_fc_w));
    MP_TRY(upload(bb, std::vector<float>(fb, fb + 512), &bb->d_fc_b));
  }
  const std::string H = head_kind == 0 ? "pose_fc" : "views_logits_head";
  const float* hw = find(sm, H + ".weight", (int64_t)n_head_out * NF);
  const float* hb = find(sm, H + ".bias", n_head_out);
  if (!hw || !hb) { mp_backbone_destroy(bb); return MP_ERR_INVALID; }
  MP_TRY(upload(bb, std::vector<float>(hw, hw + (size_t)n_head_out * NF), &bb->d_head_w));
  MP_TRY(upload(bb, std::vector<float>(hb, hb + n_head_out), &bb->d_head_b));
#undef MP_TRY
  *out = bb;
  return MP_OK;
}

extern "C" int mp_backbone_destroy(mp_backbone* bb) {
  if (!bb) return MP_OK;
  for (void* p : bb->allocs) (void)hipFree(p);
  delete bb;
  return MP_OK;
}

extern "C" int mp_backbone_input_channels_padded(const mp_backbone* bb) { return bb ? bb->c_in_p : 0; }
extern "C" int mp_backbone_input_border(const mp_backbone* bb) { return bb ? bb->in_border : 0; }

// workspace layout: [stem out][per stage: A, A_act (wide only), B, C]
extern "C" size_t mp_backbone_workspace_bytes(const mp_backbone* bb, int batch, int h, int w) {
  if (!bb || batch <= 0) return 0;
  const Geometry g = geometry(bb, h, w);
  size_t fl = align_up(buf_floats(batch, g.h1, g.w1, bb->stageC[0]), 64);
  const int per_stage = bb->wide ? 4 : 3;
  for (int s = 0; s < 4; ++s) fl += per_stage * align_up(buf_floats(batch, g.hs[s], g.ws[s], bb->stageC[s]), 64);
  fl += SPLITK_WS_FLOATS;
  return fl * sizeof(float);
}

extern "C" int mp_backbone_workspace_reset(mp_backbone* bb, const void* d_ws) {
  MP_REQUIRE(bb, "mp_backbone_workspace_reset: null handle");
  for (size_t i = 0; i < bb->ws_known.size();)
    if (bb->ws_known[i].ptr == d_ws) bb->ws_known.erase(bb->ws_known.begin() + i); else ++i;
  return MP_OK;
}

// The piece blob of the stem for records whose fp32-kind channels are the set bits of `f32_mask` (packed and uploaded on the first call
// for a mask: host work + a synchronous copy, so this is the explicit PREPARE step -- once, outside stream capture, from one thread; the
// forward only looks the blob up); returns the record length in bf16 elements, 0 if this backbone's stem has no exact-piece form.
extern "C" int mp_backbone_xrec_prepare(mp_backbone* bb, uint32_t f32_mask) {
  if (!bb || bb->stem_w_host.empty() || bb->c_in > 32) return 0;
  if (bb->c_in < 32 && (f32_mask >> bb->c_in) != 0u) return 0;
  const int n_f32 = __builtin_popcount(f32_mask), n_u8 = bb->c_in - n_f32;
  if (!mp_conv_stem_supported(bb->stem.K, n_f32, n_u8) || bb->stem.Cout % 64 != 0) return 0;
  std::lock_guard<std::mutex> blob_lock(bb->blob_mu);
  if (!bb->stem_blobs.count(f32_mask)) {
    std::vector<unsigned char> blob(mp_conv_stem_packed_bytes(bb->stem.K, n_f32, n_u8, bb->stem.Cout));
    if (mp_conv_stem_pack_weights_mask(bb->stem_w_host.data(), bb->stem.Cout, bb->c_in, bb->stem.K, f32_mask,
                                       bb->stem_scale_host.empty() ? nullptr : bb->stem_scale_host.data(), blob.data()) != MP_OK)
      return 0;
    void* d = nullptr;
    if (hipMalloc(&d, blob.size()) != hipSuccess) return 0;
    if (hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return 0; }
    bb->allocs.push_back(d);
    bb->stem_blobs[f32_mask] = d;
    // the background-tile walk: only for leading-channel masks (no depth channels) with fewer fp32-kind chunks than the record has
    if (f32_mask == (n_f32 >= 32 ? 0xFFFFFFFFu : (1u << n_f32) - 1u) && mp_conv_stem_sparse_chunks(bb->stem.K, n_f32, n_u8) > 0) {
      std::vector<unsigned char> sb(mp_conv_stem_sparse_packed_bytes(bb->stem.K, n_f32, n_u8, bb->stem.Cout));
      void* ds = nullptr;
      if (mp_conv_stem_pack_weights_sparse(bb->stem_w_host.data(), bb->stem.Cout, bb->c_in, bb->stem.K, n_f32,
                                           bb->stem_scale_host.empty() ? nullptr : bb->stem_scale_host.data(), sb.data()) == MP_OK &&
          hipMalloc(&ds, sb.size()) == hipSuccess) {
        if (hipMemcpy(ds, sb.data(), sb.size(), hipMemcpyHostToDevice) == hipSuccess) {
          bb->allocs.push_back(ds);
          bb->stem_blobs_sparse[f32_mask] = ds;
        } else {
          (void)hipFree(ds);
        }
      }
    }
  }
  return mp_xrec_elements(n_f32, n_u8);
}

static uint32_t leading_mask(int n_f32) { return n_f32 >= 32 ? 0xFFFFFFFFu : n_f32 <= 0 ? 0u : (1u << n_f32) - 1u; }

extern "C" int mp_backbone_xrec_elements(mp_backbone* bb, int n_f32) {
  if (!bb || n_f32 < 0 || n_f32 > bb->c_in) return 0;
  return mp_backbone_xrec_prepare(bb, leading_mask(n_f32));
}

// x_mode: 0 = fp32 padded NHWC, 1 = binary16 elements (MP_RASTER_F16), 2 = bf16 stem records whose fp32-kind channels are f32_mask (MP_RASTER_XREC)
static int backbone_forward_impl(mp_backbone* bb, const float* d_x, int x_mode, uint32_t f32_mask, int batch, int h, int w, float* d_out, float* d_sigmoid,
                                 float* d_feat, void* d_ws, size_t ws_bytes, mp_stream stream, const unsigned char* d_tile_flags = nullptr) {
  const bool x_f16 = x_mode == 1;
  MP_REQUIRE(bb && d_x && d_out && d_ws, "mp_backbone_forward: null pointer");
  if (batch == 0) return MP_OK;
  const size_t need = mp_backbone_workspace_bytes(bb, batch, h, w);
  MP_REQUIRE(ws_bytes >= need, "mp_backbone_forward: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const Geometry g = geometry(bb, h, w);
  bool known = false;
  for (const auto& k : bb->ws_known) known = known || (k.ptr == d_ws && k.batch == batch && k.h == h && k.w == w);
  if (!known) {
    MP_CHECK_HIP(hipMemsetAsync(d_ws, 0, need, s));  // zero borders once; interiors are always overwritten
    for (size_t i = 0; i < bb->ws_known.size();)     // a pointer re-used with another geometry invalidates its old entry
      if (bb->ws_known[i].ptr == d_ws) bb->ws_known.erase(bb->ws_known.begin() + i); else ++i;
    if (bb->ws_known.size() >= 8) bb->ws_known.erase(bb->ws_known.begin());
    bb->ws_known.push_back({d_ws, batch, h, w});
  }
  float* p = (float*)d_ws;
  float* S = p; p += align_up(buf_floats(batch, g.h1, g.w1, bb->stageC[0]), 64);
  float *A[4], *Aact[4], *Bf[4], *Cf[4];
  for (int st = 0; st < 4; ++st) {
    const size_t n = align_up(buf_floats(batch, g.hs[st], g.ws[st], bb->stageC[st]), 64);
    A[st] = p; p += n;
    Aact[st] = nullptr;
    if (bb->wide) { Aact[st] = p; p += n; }
    Bf[st] = p; p += n;
    Cf[st] = p; p += n;
  }
  float* SK = p;  // split-K scratch (SPLITK_WS_FLOATS)
  int rc;
  bool stem_pooled = false;
  // stem: conv + folded bn + relu, then 3x3/s2 max pool (+ first block's pre-activation for the wide nets)
  if (x_mode == 2) {
    const void* d_stem_pieces = nullptr;
    const void* d_sparse_blob = nullptr;
    {
      std::lock_guard<std::mutex> blob_lock(bb->blob_mu);
      const auto blob_it = bb->stem_blobs.find(f32_mask);
      if (blob_it != bb->stem_blobs.end()) d_stem_pieces = blob_it->second;
      const auto sp_it = bb->stem_blobs_sparse.find(f32_mask);
      if (sp_it != bb->stem_blobs_sparse.end()) d_sparse_blob = sp_it->second;
    }
    MP_REQUIRE(d_stem_pieces != nullptr, "mp_backbone_forward_xrec: no piece blob for the fp32-kind channel mask 0x%x of this %d-channel stem: "
               "call mp_backbone_xrec_prepare / mp_backbone_xrec_elements first (it returns 0 if the stem has no exact-piece form)", f32_mask, bb->c_in);
    const int n_f32 = __builtin_popcount(f32_mask);
    mp_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.d_x = d_x; d.N = batch; d.H = h; d.W = w; d.C = bb->stem.Cin_p; d.c_real = bb->c_in; d.in_border = bb->in_border;
    d.d_bias = bb->stem.d_b; d.Cout = bb->stem.Cout; d.KH = bb->stem.K; d.KW = bb->stem.K; d.stride = 2; d.pad = bb->stem.pad;
    d.d_y = S; d.out_border = 1; d.relu = 1;
    // the max pool rides in the stem's epilogue and the stem map is never written (MP_STEM_POOL=0: separate kernels)
    static const bool fuse_pool = !(getenv("MP_STEM_POOL") && atoi(getenv("MP_STEM_POOL")) == 0);
    const void* d_sparse = d_tile_flags ? d_sparse_blob : nullptr;   // background-tile walk where it applies
    if (fuse_pool) {
      d.d_y = nullptr;
      rc = d_sparse ? mp_conv_stem_xrec_sparse(&d, d_stem_pieces, d_sparse, n_f32, d_tile_flags, A[0], 1, s)
                    : mp_conv_stem_xrec_pool(&d, d_stem_pieces, n_f32, A[0], 1, s);
      // pre-activation WideResNets (round 6): relu(bn1(pooled)) by a small elementwise pass over the pooled map (the fused pool completes
      // straddling windows with atomics, so the stem itself cannot emit it) instead of the separate pool kernel's pass over the 4x larger stem map
      if (!rc && bb->wide) rc = mp_bn_relu_nhwc(A[0], batch, g.hs[0], g.ws[0], bb->stageC[0], 1, Aact[0], bb->blocks[0].pre.d_scale, bb->blocks[0].pre.d_shift, s);
      stem_pooled = true;
    } else {
      rc = d_sparse ? mp_conv_stem_xrec_sparse(&d, d_stem_pieces, d_sparse, n_f32, d_tile_flags, nullptr, 0, s)
                    : mp_conv_stem_xrec(&d, d_stem_pieces, n_f32, s);
    }
  } else {
    rc = run_conv(bb, bb->stem, d_x, batch, h, w, bb->in_border, S, 1, nullptr, 1, nullptr, nullptr, s, SK, x_f16);
  }
  if (rc) return rc;
  const Block& b0 = bb->blocks[0];
  if (!stem_pooled) {
    rc = mp_maxpool3x3s2(S, batch, g.h1, g.w1, bb->stageC[0], 1, A[0], 1, bb->wide ? Aact[0] : nullptr, bb->wide ? b0.pre.d_scale : nullptr,
                         bb->wide ? b0.pre.d_shift : nullptr, s);
    if (rc) return rc;
  }
  const int nb = (int)bb->blocks.size();
  for (int i = 0; i < nb; ++i) {
    const Block& blk = bb->blocks[i];
    const int so = bb->stage_of_block[i];                 // output stage
    const int si = (blk.has_down && so > 0) ? so - 1 : so;  // input stage
    const int Hi = g.hs[si], Wi = g.ws[si];
    const bool last = (i + 1 == nb);
    if (!bb->wide) {
      // y1 = relu(bn1(conv1(x))); idn = bn(down(x)) | x; out = relu(bn2(conv2(y1)) + idn)
      rc = run_conv(bb, blk.conv1, A[si], batch, Hi, Wi, 1, Bf[so], 1, nullptr, 1, nullptr, nullptr, s, SK);
      if (rc) return rc;
      const float* idn = A[si];
      if (blk.has_down) {
        rc = run_conv(bb, blk.down, A[si], batch, Hi, Wi, 1, Cf[so], 1, nullptr, 0, nullptr, nullptr, s, SK);
        if (rc) return rc;
        idn = Cf[so];
      }
      rc = run_conv(bb, blk.conv2, Bf[so], batch, g.hs[so], g.ws[so], 1, A[so], 1, idn, 1, nullptr, nullptr, s, SK);
      if (rc) return rc;
    } else {
      // a = relu(bn1(x)) was produced upstream into Aact[si]; residual = down(a) | x
      rc = run_conv(bb, blk.conv1, Aact[si], batch, Hi, Wi, 1, Bf[so], 1, nullptr, 1, nullptr, nullptr, s, SK);
      if (rc) return rc;
      const float* idn = A[si];
      if (blk.has_down) {
        rc = run_conv(bb, blk.down, Aact[si], batch, Hi, Wi, 1, Cf[so], 1, nullptr, 0, nullptr, nullptr, s, SK);
        if (rc) return rc;
        idn = Cf[so];
      }
      const BnAct* next_pre = last ? nullptr : &bb->blocks[i + 1].pre;
      rc = run_conv(bb, blk.conv2, Bf[so], batch, g.hs[so], g.ws[so], 1, A[so], 1, idn, 0, last ? nullptr : Aact[so], next_pre, s, SK);
      if (rc) return rc;
    }
  }
  return mp_pool_fc_heads(A[3], batch, g.hs[3], g.ws[3], bb->stageC[3], 1, bb->d_fc_w, bb->d_fc_b, bb->n_feat, bb->d_head_w, bb->d_head_b, bb->n_out,
                          d_feat, d_out, d_sigmoid, s);
}

extern "C" int mp_backbone_forward(mp_backbone* bb, const float* d_x, int batch, int h, int w, float* d_out, float* d_sigmoid,
                                   float* d_feat, void* d_ws, size_t ws_bytes, mp_stream stream) {
  return backbone_forward_impl(bb, d_x, 0, 0, batch, h, w, d_out, d_sigmoid, d_feat, d_ws, ws_bytes, stream);
}

extern "C" int mp_backbone_forward_xrec(mp_backbone* bb, const void* d_xrec, int n_f32, int batch, int h, int w, float* d_out, float* d_sigmoid,
                                        float* d_feat, void* d_ws, size_t ws_bytes, mp_stream stream) {
  MP_REQUIRE(bb && n_f32 >= 0 && n_f32 <= 32, "mp_backbone_forward_xrec: bad arguments");
  return backbone_forward_impl(bb, (const float*)d_xrec, 2, leading_mask(n_f32), batch, h, w, d_out, d_sigmoid, d_feat, d_ws, ws_bytes, stream);
}

extern "C" int mp_backbone_forward_xrec_mask(mp_backbone* bb, const void* d_xrec, uint32_t f32_mask, int batch, int h, int w, float* d_out,
                                             float* d_sigmoid, float* d_feat, void* d_ws, size_t ws_bytes, mp_stream stream) {
  return backbone_forward_impl(bb, (const float*)d_xrec, 2, f32_mask, batch, h, w, d_out, d_sigmoid, d_feat, d_ws, ws_bytes, stream);
}

extern "C" int mp_backbone_forward_xrec_sparse(mp_backbone* bb, const void* d_xrec, uint32_t f32_mask, const unsigned char* d_tile_flags, int batch,
                                               int h, int w, float* d_out, float* d_sigmoid, float* d_feat, void* d_ws, size_t ws_bytes,
                                               mp_stream stream) {
  return backbone_forward_impl(bb, (const float*)d_xrec, 2, f32_mask, batch, h, w, d_out, d_sigmoid, d_feat, d_ws, ws_bytes, stream, d_tile_flags);
}

extern "C" int mp_backbone_forward_f16(mp_backbone* bb, const void* d_x_half, int batch, int h, int w, float* d_out, float* d_sigmoid,
                                       float* d_feat, void* d_ws, size_t ws_bytes, mp_stream stream) {
  return backbone_forward_impl(bb, (const float*)d_x_half, 1, 0, batch, h, w, d_out, d_sigmoid, d_feat, d_ws, ws_bytes, stream);
}

extern "C" double mp_backbone_flops(const mp_backbone* bb, int batch, int h, int w) {
  if (!bb) return 0.0;
  const Geometry g = geometry(bb, h, w);
  auto conv_flops = [](const ConvLayer& L, int Ho, int Wo) { return 2.0 * L.Cout * L.Cin * L.K * L.K * (double)Ho * Wo; };
  double f = conv_flops(bb->stem, g.h1, g.w1);
  for (size_t i = 0; i < bb->blocks.size(); ++i) {
    const Block& blk = bb->blocks[i];
    const int so = bb->stage_of_block[i];
    f += conv_flops(blk.conv1, g.hs[so], g.ws[so]) + conv_flops(blk.conv2, g.hs[so], g.ws[so]);
    if (blk.has_down) f += conv_flops(blk.down, g.hs[so], g.ws[so]);
  }
  if (!bb->wide) f += 2.0 * 512 * 512;
  f += 2.0 * bb->n_feat * bb->n_out;
  return f * batch;
}
