// libprobunet engine: static plan (tensors in one HBM arena), forward/backward sequencing, C ABI.
// Mirrors, as a static graph, ProbabilisticUNet.forward/elbo (src/prob_unet.py:194-317) and UNet/UNetBlock
// (src/networks.py:134-333) of the reference; see include/probunet.h for the boundary.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <string>
#include <vector>

#include "../../include/probunet.h"
#include "../../include/probunet_testing.h"
#include "pu_kernels.h"

using namespace pu;

#define PU_ABI 2

namespace {

struct ConvL {
  int cin = 0, cout = 0, ks = 3;
  int64_t w_off = -1, b_off = -1;
  long pk_fwd = -1, pk_bwd = -1;       // element offsets into the packed-weight buffer
  int cin_pk = 0, rows_fwd = 0;        // forward pack: [rows_fwd][taps][cin_pk]
  int cout_pk = 0, rows_bwd = 0;       // dgrad pack:   [rows_bwd][taps][cout_pk]
  bool need_dgrad = true;
  bool frag = false;                   // packed weights are fragment-major (conv3 kernel)
  bool m16_fwd = false, m16_bwd = false;   // ... as v_mfma_f32_16x16x32 fragments (forward / data-gradient pack)
};
struct GNL {
  int C = 0, G = 0, nchunk = 1;
  int64_t g_off = -1, b_off = -1, ss_off = -1;   // gamma, beta, affine.bias (scale|shift) offsets
  float* stat = nullptr; float* coef = nullptr;
  uint32_t drop_stream = 0; bool dropout = false; int resample = RS_NONE;
  int drop_site = -1;                  // index into pu_ctx::drop_sites when this GroupNorm is followed by dropout
  bool fuse_bwd = false;               // pass 1 of the backward rides in the epilogue of the data gradient that produces dy (GNBwdFuse)
  uint8_t* keep_bits = nullptr;        // dropout keep decisions saved by the forward apply, read by the backward (16-bit engines)
};
struct Act { TV v; TV g; int flag = -1; };          // value view, gradient view, index of the shared "grad written" flag

enum SkipKind { SK_NONE, SK_IDENTITY, SK_RESAMPLE, SK_CONV };
struct Block {
  std::string name; bool is_block = true;
  int cin = 0, cout = 0; bool up = false, down = false; int skip = SK_NONE;
  Act x, a0, c0, h1, out, skin;
  GNL n0, n1; ConvL conv0, conv1, skipc;
  // GroupNorm statistics fused into the producing convolutions (16-bit engines): rows written by the final writer of `out`
  // (conv1, or conv0 of the stem) and of `c0` (conv0); sl_* = slot count per image reported by the launcher for this forward
  float *st_out = nullptr, *st_c0 = nullptr; int cap_out = 0, cap_c0 = 0, sl_out = 0, sl_c0 = 0;
  int src0 = -1, src1 = -1;              // producers of x: index into enc (>= 0) or dec (1000 + j); src1 = skip half of a concat
  int64_t p_lo = -1;                     // first flat parameter offset of this block
  int bucket_close = -1;                 // gradient bucket whose last writer is this block's backward (data-parallel hand-off), or -1
};
struct GaussNet {
  std::vector<ConvL> convs; std::vector<Act> outs; std::vector<Act> ins;   // ins[i] is the input of conv i
  std::vector<int> pool_before;                                            // 1 if a maxpool precedes conv i
  int64_t wmu = -1, bmu = -1, wls = -1, bls = -1;
  float *hbuf = nullptr, *mu = nullptr, *ls = nullptr, *dmu = nullptr, *dls = nullptr;
  float* enc_inv = nullptr;            // f16: 1 / (device-chosen power-of-two scale of this encoder's backward), see launch_enc_rescale
  int lastB = 0;
};

}  // namespace

struct pu_ctx {
  pu_config cfg; int device = 0; std::string err;
  int dt = 0; size_t esz = 4;
  std::vector<pu_param_desc> table; int64_t nparams = 0;
  float* params = nullptr; float* grads = nullptr; bool packed_valid = false;
  char* arena = nullptr; size_t arena_size = 0, arena_used = 0; bool planning = true;
  void* packed = nullptr; long packed_elems = 0; std::vector<PackDesc> descs; PackDesc* descs_dev = nullptr;
  std::vector<char> flags;
  // network
  std::vector<Block> enc, dec; GNL out_norm; ConvL out_conv; Act out_a, feat;
  Act x_in, xy_in;
  GaussNet prior, post;
  int64_t fc_w0 = -1, fc_b0 = -1, fc_w1 = -1, fc_b1 = -1, fc_w2 = -1, fc_b2 = -1;
  // scratch
  float* wg_slab = nullptr; long wg_slab_floats = 0;
  float* gn_part = nullptr; float* gn_part2 = nullptr; float* gn_coef2 = nullptr; float* bias_part = nullptr; TV dv_scratch;
  float* gn_rows = nullptr; size_t gn_rows_per_img = 0; int gn_rows_cap = 0;   // pass-1 rows written by data-gradient epilogues: [B][slots][C][2]
  long fuse_bwd_minhw = 256L * 256;     // GroupNorm sites with at least this many pixels use the fused backward (PU_GN_FUSE_BWD_MINHW; 0 = off)
  float *z = nullptr, *dz = nullptr, *preds = nullptr, *dpreds = nullptr, *kl = nullptr, *kl2 = nullptr, *scal = nullptr;
  Act fc_feat;                      // standalone fcomb input (converted) + its gradient
  float* fc_z = nullptr; int fc_B = 0; int fc_bcast = 0;
  int unet_B = 0, unet_train = 0; uint64_t unet_seed = 0;
  int max_gn_c = 0;
  float inv_scale = 1.f;            // 1 / (loss scale) applied to every parameter-gradient write of the current backward
  // weight-gradient kernels (MFMA-bound) run on a side stream, overlapping the HBM-bound dgrad -> GroupNorm-backward chain
  bool fused_stats = true;
  float wm_alpha = 0.007f, wm_beta = 0.048f, wm_lam = 0.f, wm_range = -1.f;     // wmse_ms_ssim_loss defaults (prob_unet.py:231-233)
  const float* wm_range_dev = nullptr;                                           // pu_set_recon_range_dev
  float* ms_ws = nullptr; size_t ms_ws_floats = 0;                              // MS-SSIM pyramid workspace, allocated on first use
  static constexpr int NSLAB = 1;
  hipStream_t side = nullptr, side2 = nullptr; std::vector<hipEvent_t> evs; size_t ev_next = 0; bool side_dirty = false; bool use_side = true;
  // injected dropout masks (parity tests): one site per UNetBlock in execution order
  struct DropSite { std::string name; int C, H, W; size_t off; };
  std::vector<DropSite> drop_sites; uint8_t* drop_masks = nullptr; size_t drop_masks_cap = 0; int drop_masks_B = 0;
  // data-parallel hand-off: flat gradient ranges in completion order + the events that mark them complete
  struct Bucket { int64_t lo = 0, hi = 0; hipEvent_t ev[3] = {nullptr, nullptr, nullptr}; bool rec[3] = {false, false, false}; };
  int want_buckets = 0; std::vector<Bucket> buckets; std::vector<int64_t> cuts;      // cuts: descending flat offsets closing the U-Net buckets
  // cfg5: captured launch sequences of pu_sample / pu_sample_hr
  struct SampleGraph { std::vector<const void*> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  bool sample_graph_on = false; std::vector<SampleGraph> sample_graphs; std::vector<std::vector<const void*>> sample_seen;
  long graph_captures = 0, graph_replays = 0, graph_eager = 0;      // pu_sample_graph_stats
  // fused ELBO backward, 16-bit engines: convolution weights / biases are WRITTEN by their (single) weight-gradient reduction, so only the
  // accumulated parameters (GroupNorm, Fcomb, heads, dead parameters) are zeroed beforehand: `zero_ranges` = complement of the conv ranges
  std::vector<std::pair<int64_t, int64_t>> conv_ranges; long* zero_ranges_dev = nullptr; int n_zero_ranges = 0; bool grad_overwrite = false;
  size_t max_tensor_elems = 0;                                      // largest B*H*W*ld of the plan (kernels index pixels with 32-bit element offsets)
};

static std::string g_create_err;

// ------------------------------------------------------------------ helpers
#define CKH(expr)                                                                                       \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) {                                                                             \
      char _b[512]; snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      c->err = _b; return PU_ERR_HIP;                                                                   \
    }                                                                                                   \
  } while (0)
#define FAIL(code, ...)                                            \
  do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); c->err = _b; return code; } while (0)

template <typename F> static int dispatch(pu_ctx* c, F&& f) {
  switch (c->dt) {
    case PU_F32: return f(float{});
    case PU_F16: return f(f16{});
    case PU_BF16: return f(bf16{});
  }
  return PU_ERR_INVALID;
}

static void* arena_alloc(pu_ctx* c, size_t bytes) {
  size_t off = (c->arena_used + 255) & ~(size_t)255;
  c->arena_used = off + bytes;
  if (c->planning) return nullptr;
  return c->arena + off;
}
static TV alloc_tv(pu_ctx* c, int B, int H, int W, int C) {
  TV t; t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
  if ((size_t)B * H * W * C > c->max_tensor_elems) c->max_tensor_elems = (size_t)B * H * W * C;
  t.p = arena_alloc(c, (size_t)B * H * W * C * c->esz);
  return t;
}
static float* alloc_f32(pu_ctx* c, size_t n) { return (float*)arena_alloc(c, n * sizeof(float)); }
static TV view_c(const TV& t, int coff, int C, size_t esz) {
  TV v = t; v.C = C; v.p = t.p ? (char*)t.p + (size_t)coff * esz : nullptr; return v;
}
static TV with_b(TV t, int B) { t.B = B; return t; }
static int new_flag(pu_ctx* c) { c->flags.push_back(0); return (int)c->flags.size() - 1; }
static Act alloc_act(pu_ctx* c, int B, int H, int W, int C, bool need_grad = true) {
  Act a; a.v = alloc_tv(c, B, H, W, C);
  if (need_grad) { a.g = alloc_tv(c, B, H, W, C); a.flag = new_flag(c); }
  return a;
}
static Act view_act(const Act& a, int coff, int C, size_t esz) {
  Act v; v.v = view_c(a.v, coff, C, esz); v.g = view_c(a.g, coff, C, esz); v.flag = a.flag; return v;
}
// returns whether the gradient already holds a contribution (=> accumulate) and marks it written
static int take_acc(pu_ctx* c, const Act& a) { int r = c->flags[a.flag]; c->flags[a.flag] = 1; return r; }

static int64_t add_param(pu_ctx* c, const std::string& name, std::initializer_list<int> shape, bool is_buffer = false) {
  pu_param_desc d; memset(&d, 0, sizeof d);
  snprintf(d.name, sizeof d.name, "%s", name.c_str());
  d.ndim = (int)shape.size(); int i = 0; int64_t n = 1;
  for (int s : shape) { d.shape[i++] = s; n *= s; }
  d.is_buffer = is_buffer ? 1 : 0;
  if (is_buffer) d.offset = -1;
  else { d.offset = c->nparams; c->nparams += n; }
  c->table.push_back(d);
  return d.offset;
}

static void setup_conv(pu_ctx* c, ConvL& L, int cin, int cout, int ks, int64_t w_off, int64_t b_off, bool need_dgrad, int rh, int rw) {
  L.cin = cin; L.cout = cout; L.ks = ks; L.w_off = w_off; L.b_off = b_off; L.need_dgrad = need_dgrad;
  c->conv_ranges.push_back({w_off, w_off + (int64_t)cout * cin * ks * ks});
  if (b_off >= 0) c->conv_ranges.push_back({b_off, b_off + cout});
  L.frag = conv_uses_frag_layout((int)c->esz, rh, rw);
  const int taps = ks * ks;
  L.cin_pk = rup(cin, 32); L.rows_fwd = rup(cout, 32);
  L.pk_fwd = c->packed_elems; c->packed_elems += (long)L.rows_fwd * taps * L.cin_pk;
  L.m16_fwd = L.frag && conv_uses_mfma16((int)c->esz, taps, L.cin_pk, cout, rh, rw);
  PackDesc d; d.src_off = w_off; d.dst_off = L.pk_fwd; d.Cout = cout; d.Cin = cin; d.taps = taps; d.rows_pk = L.rows_fwd; d.k_pk = L.cin_pk;
  d.mode = L.frag ? (L.m16_fwd ? 6 : 2) : 0;
  c->descs.push_back(d);
  if (need_dgrad) {
    L.cout_pk = rup(cout, 32); L.rows_bwd = rup(cin, 32);
    L.pk_bwd = c->packed_elems; c->packed_elems += (long)L.rows_bwd * taps * L.cout_pk;
    L.m16_bwd = L.frag && conv_uses_mfma16((int)c->esz, taps, L.cout_pk, rup(cin, 8), rh, rw);      // data gradient: K = cout, N = the stored cin planes
    d.dst_off = L.pk_bwd; d.rows_pk = L.rows_bwd; d.k_pk = L.cout_pk; d.mode = L.frag ? (L.m16_bwd ? 7 : 3) : 1;
    c->descs.push_back(d);
  }
}
static int gn_groups(int C) { int g = C / 4; return g < 32 ? g : 32; }
static int gn_chunks(long HW) { long n = HW / 256; if (n < 1) n = 1; if (n > 64) n = 64; return (int)n; }
static void setup_gn(pu_ctx* c, GNL& n, int C, long HW, int64_t g_off, int64_t b_off, int64_t ss_off, int resample, bool dropout, uint32_t stream) {
  n.C = C; n.G = gn_groups(C); n.nchunk = gn_chunks(HW); n.g_off = g_off; n.b_off = b_off; n.ss_off = ss_off; n.resample = resample;
  n.dropout = dropout; n.drop_stream = stream;
  const int mb = c->cfg.max_batch;
  n.stat = alloc_f32(c, (size_t)mb * n.G * 2);
  n.coef = alloc_f32(c, (size_t)mb * C * 4);
  if (C > c->max_gn_c) c->max_gn_c = C;
  if (dropout && c->esz == 2 && resample == RS_NONE && !getenv("PU_NO_KEEP_BITS")) n.keep_bits = (uint8_t*)arena_alloc(c, (size_t)mb * HW * C / 8);
  n.fuse_bwd = c->esz == 2 && resample == RS_NONE && c->fuse_bwd_minhw > 0 && HW >= c->fuse_bwd_minhw && HW > 1024;
  if (n.fuse_bwd) {
    const size_t per = (size_t)(HW / 64) * C * 2;          // at most one row per 64 pixels (a wave's share of a tile) and channel
    if (per > c->gn_rows_per_img) c->gn_rows_per_img = per;
  }
}

// ------------------------------------------------------------------ plan construction
struct Spec { std::string name; bool is_block; int cin, cout; bool up, down, concat; int level; };

static int build_plan(pu_ctx* c) {
  const pu_config& cf = c->cfg;
  const int mb = cf.max_batch, D = cf.depth, mc = cf.model_channels;
  const size_t esz = c->esz;
  c->table.clear(); c->nparams = 0; c->packed_elems = 0; c->descs.clear(); c->flags.clear();
  c->enc.clear(); c->dec.clear(); c->arena_used = 0; c->max_gn_c = 0; c->drop_sites.clear(); c->max_tensor_elems = 0; c->gn_rows_per_img = 0; c->conv_ranges.clear();
  c->prior = GaussNet(); c->post = GaussNet();
  const int emb = mc * 4;
  add_param(c, "unet.map_label.weight", {emb, 1});

  // ---- specs (networks.py:259-295)
  std::vector<Spec> es, ds;
  int cout = cf.input_channels;
  for (int lv = 0; lv < D; ++lv) {
    const int r = 128 >> lv;
    char p[64]; snprintf(p, sizeof p, "unet.enc.%dx%d", r, r);
    if (lv == 0) { int cin = cout; cout = mc * cf.channel_mult[0]; es.push_back({std::string(p) + "_conv", false, cin, cout, false, false, false, lv}); }
    else es.push_back({std::string(p) + "_down", true, cout, cout, false, true, false, lv});
    for (int i = 0; i < 2; ++i) { int cin = cout; cout = mc * cf.channel_mult[lv]; es.push_back({std::string(p) + "_block" + std::to_string(i), true, cin, cout, false, false, false, lv}); }
  }
  std::vector<int> skips; for (auto& s : es) skips.push_back(s.cout);
  std::vector<int> skip_of_dec;
  {
    std::vector<int> stack; for (int i = 0; i < (int)es.size(); ++i) stack.push_back(i);
    for (int lv = D - 1; lv >= 0; --lv) {
      const int r = 128 >> lv;
      char p[64]; snprintf(p, sizeof p, "unet.dec.%dx%d", r, r);
      if (lv == D - 1) {
        ds.push_back({std::string(p) + "_in0", true, cout, cout, false, false, false, lv}); skip_of_dec.push_back(-1);
        ds.push_back({std::string(p) + "_in1", true, cout, cout, false, false, false, lv}); skip_of_dec.push_back(-1);
      } else { ds.push_back({std::string(p) + "_up", true, cout, cout, true, false, false, lv}); skip_of_dec.push_back(-1); }
      for (int i = 0; i < 3; ++i) {
        const int si = stack.back(); stack.pop_back();
        const int cin = cout + es[si].cout; cout = mc * cf.channel_mult[lv];
        ds.push_back({std::string(p) + "_block" + std::to_string(i), true, cin, cout, false, false, true, lv}); skip_of_dec.push_back(si);
      }
    }
  }
  const int c_last = cout;
  auto chk8 = [&](int ch) { return ch % 8 == 0; };
  for (auto& s : es) if (!chk8(s.cout)) FAIL(PU_ERR_INVALID, "channel count %d of %s is not a multiple of 8", s.cout, s.name.c_str());
  for (int i = 0; i < D; ++i) if (!chk8(cf.num_filters[i])) FAIL(PU_ERR_INVALID, "num_filters[%d]=%d is not a multiple of 8", i, cf.num_filters[i]);

  // ---- tensors: inputs
  const int H = cf.H, W = cf.W;
  const int cin_pad = rup(cf.input_channels, 8), cxy_pad = rup(cf.input_channels + cf.num_classes, 8);
  c->x_in = alloc_act(c, mb, H, W, cin_pad, false);
  c->xy_in = alloc_act(c, mb, H, W, cxy_pad, false);

  // ---- concat buffers for decoder blocks; encoder outputs live inside them (zero-copy torch.cat, networks.py:328-329)
  std::vector<Act> cat(ds.size());
  for (size_t j = 0; j < ds.size(); ++j)
    if (ds[j].concat) { const int r_h = H >> ds[j].level, r_w = W >> ds[j].level; cat[j] = alloc_act(c, mb, r_h, r_w, ds[j].cin); }
  std::vector<Act> enc_out(es.size());
  for (size_t j = 0; j < ds.size(); ++j)
    if (ds[j].concat) { const int si = skip_of_dec[j]; const int cx = ds[j].cin - es[si].cout; enc_out[si] = view_act(cat[j], cx, es[si].cout, esz); }

  uint32_t drop_stream = 1;
  auto build_block = [&](const Spec& s, Block& b, const Act& xin, const Act& outa, int inH, int inW) {
    b.name = s.name; b.is_block = s.is_block; b.cin = s.cin; b.cout = s.cout; b.up = s.up; b.down = s.down;
    b.x = xin; b.out = outa;
    const int oH = s.down ? inH / 2 : (s.up ? inH * 2 : inH), oW = s.down ? inW / 2 : (s.up ? inW * 2 : inW);
    const std::string& p = s.name;
    if (conv_uses_frag_layout((int)esz, oH, oW)) {         // conv3-class producers: at least 64 pixels per wave
      b.cap_out = oH * oW / 64; b.st_out = alloc_f32(c, (size_t)mb * b.cap_out * s.cout * 2);
      if (s.is_block) { b.cap_c0 = b.cap_out; b.st_c0 = alloc_f32(c, (size_t)mb * b.cap_c0 * s.cout * 2); }
    }
    b.p_lo = c->nparams;
    if (!s.is_block) {
      const int64_t w = add_param(c, p + ".weight", {s.cout, s.cin, 3, 3});
      const int64_t bb = add_param(c, p + ".bias", {s.cout});
      setup_conv(c, b.conv0, s.cin, s.cout, 3, w, bb, false, oH, oW);
      return;
    }
    const int64_t g0 = add_param(c, p + ".norm0.weight", {s.cin});
    const int64_t b0 = add_param(c, p + ".norm0.bias", {s.cin});
    const int64_t w0 = add_param(c, p + ".conv0.weight", {s.cout, s.cin, 3, 3});
    const int64_t bb0 = add_param(c, p + ".conv0.bias", {s.cout});
    if (s.up || s.down) add_param(c, p + ".conv0.resample_filter", {1, 1, 2, 2}, true);
    add_param(c, p + ".affine.weight", {2 * s.cout, emb});
    const int64_t ab = add_param(c, p + ".affine.bias", {2 * s.cout});
    const int64_t g1 = add_param(c, p + ".norm1.weight", {s.cout});
    const int64_t b1 = add_param(c, p + ".norm1.bias", {s.cout});
    const int64_t w1 = add_param(c, p + ".conv1.weight", {s.cout, s.cout, 3, 3});
    const int64_t bb1 = add_param(c, p + ".conv1.bias", {s.cout});
    b.skip = s.cin != s.cout ? SK_CONV : ((s.up || s.down) ? SK_RESAMPLE : SK_IDENTITY);
    if (b.skip == SK_CONV) {
      const int64_t ws = add_param(c, p + ".skip.weight", {s.cout, s.cin, 1, 1});
      const int64_t bs = add_param(c, p + ".skip.bias", {s.cout});
      if (s.up || s.down) add_param(c, p + ".skip.resample_filter", {1, 1, 2, 2}, true);
      setup_conv(c, b.skipc, s.cin, s.cout, 1, ws, bs, true, oH, oW);
    } else if (b.skip == SK_RESAMPLE) add_param(c, p + ".skip.resample_filter", {1, 1, 2, 2}, true);
    const int rs = s.down ? RS_DOWN : (s.up ? RS_UP : RS_NONE);
    setup_gn(c, b.n0, s.cin, (long)inH * inW, g0, b0, -1, rs, false, 0);
    setup_conv(c, b.conv0, s.cin, s.cout, 3, w0, bb0, true, oH, oW);
    setup_gn(c, b.n1, s.cout, (long)oH * oW, g1, b1, ab, RS_NONE, true, drop_stream++);
    {
      size_t off = 0;
      if (!c->drop_sites.empty()) { const auto& q = c->drop_sites.back(); off = q.off + (size_t)q.C * q.H * q.W; }   // per-sample offsets
      b.n1.drop_site = (int)c->drop_sites.size();
      c->drop_sites.push_back({s.name, s.cout, oH, oW, off});
    }
    setup_conv(c, b.conv1, s.cout, s.cout, 3, w1, bb1, true, oH, oW);
    b.a0 = alloc_act(c, mb, oH, oW, s.cin);
    b.c0 = alloc_act(c, mb, oH, oW, s.cout);
    b.h1 = alloc_act(c, mb, oH, oW, s.cout);
    if ((s.up || s.down) && b.skip != SK_IDENTITY) b.skin = alloc_act(c, mb, oH, oW, s.cin);
  };

  // ---- encoder
  c->enc.resize(es.size());
  {
    Act cur = c->x_in; int curH = H, curW = W;
    for (size_t i = 0; i < es.size(); ++i) {
      build_block(es[i], c->enc[i], cur, enc_out[i], curH, curW);
      c->enc[i].src0 = (int)i - 1;
      if (es[i].down) { curH /= 2; curW /= 2; }
      cur = enc_out[i];
    }
    // ---- decoder
    c->dec.resize(ds.size());
    for (size_t j = 0; j < ds.size(); ++j) {
      Act xin = ds[j].concat ? cat[j] : cur;
      const int oH = ds[j].up ? curH * 2 : curH, oW = ds[j].up ? curW * 2 : curW;
      Act outa;
      if (j + 1 < ds.size() && ds[j + 1].concat) outa = view_act(cat[j + 1], 0, ds[j].cout, esz);
      else outa = alloc_act(c, mb, oH, oW, ds[j].cout);
      build_block(ds[j], c->dec[j], xin, outa, curH, curW);
      c->dec[j].src0 = j == 0 ? (int)es.size() - 1 : 1000 + (int)j - 1;
      if (ds[j].concat) c->dec[j].src1 = skip_of_dec[j];
      curH = oH; curW = oW; cur = outa;
    }
    // ---- output head (networks.py:296-297,331)
    const int64_t og = add_param(c, "unet.out_norm.weight", {c_last});
    const int64_t ob = add_param(c, "unet.out_norm.bias", {c_last});
    const int64_t ow = add_param(c, "unet.out_conv.weight", {cf.num_filters[0], c_last, 3, 3});
    const int64_t obb = add_param(c, "unet.out_conv.bias", {cf.num_filters[0]});
    setup_gn(c, c->out_norm, c_last, (long)H * W, og, ob, -1, RS_NONE, false, 0);
    setup_conv(c, c->out_conv, c_last, cf.num_filters[0], 3, ow, obb, true, H, W);
    c->out_a = alloc_act(c, mb, H, W, c_last);
    c->feat = alloc_act(c, mb, H, W, cf.num_filters[0]);
  }

  // ---- Gaussian encoders (prob_unet.py:19-54)
  auto build_gauss = [&](GaussNet& g, const char* net, const Act& in0, int cin0_logical, int cin0_alloc) {
    Act cur = in0; int curH = H, curW = W; int cin = cin0_logical; int idx = 0; (void)cin0_alloc;
    for (int lv = 0; lv < D; ++lv) {
      bool pooled = false;
      if (lv != 0) {
        idx += 1;
        Act pl = alloc_act(c, mb, curH / 2, curW / 2, cin);
        g.ins.push_back(pl); pooled = true; curH /= 2; curW /= 2; cur = pl;
      }
      for (int k = 0; k < 3; ++k) {
        const int co = cf.num_filters[lv];
        const int64_t w = add_param(c, std::string(net) + ".encoder." + std::to_string(idx) + ".weight", {co, cin, 3, 3});
        const int64_t b = add_param(c, std::string(net) + ".encoder." + std::to_string(idx) + ".bias", {co});
        ConvL L; setup_conv(c, L, cin, co, 3, w, b, !(lv == 0 && k == 0), curH, curW);
        g.convs.push_back(L);
        if (!(pooled && k == 0)) g.ins.push_back(cur);
        g.pool_before.push_back(pooled && k == 0 ? 1 : 0);
        Act o = alloc_act(c, mb, curH, curW, co);
        g.outs.push_back(o); cur = o; cin = co; idx += 2;
      }
    }
    const int cf_last = cf.num_filters[D - 1], L = cf.latent_dim;
    g.wmu = add_param(c, std::string(net) + ".conv_mu.weight", {L, cf_last, 1, 1});
    g.bmu = add_param(c, std::string(net) + ".conv_mu.bias", {L});
    g.wls = add_param(c, std::string(net) + ".conv_log_sigma.weight", {L, cf_last, 1, 1});
    g.bls = add_param(c, std::string(net) + ".conv_log_sigma.bias", {L});
    g.hbuf = alloc_f32(c, (size_t)mb * cf_last);
    g.mu = alloc_f32(c, (size_t)mb * L); g.ls = alloc_f32(c, (size_t)mb * L);
    g.dmu = alloc_f32(c, (size_t)mb * L); g.dls = alloc_f32(c, (size_t)mb * L);
    g.enc_inv = alloc_f32(c, 16);
  };
  build_gauss(c->prior, "prior", c->x_in, cf.input_channels, cin_pad);
  build_gauss(c->post, "posterior", c->xy_in, cf.input_channels + cf.num_classes, cxy_pad);

  // ---- Fcomb (prob_unet.py:99-105)
  const int F = cf.num_filters[0], L = cf.latent_dim, Co = cf.num_classes, MM_ = cf.max_members;
  c->fc_w0 = add_param(c, "fcomb.layers.0.weight", {F, F + L, 1, 1});
  c->fc_b0 = add_param(c, "fcomb.layers.0.bias", {F});
  c->fc_w1 = add_param(c, "fcomb.layers.2.weight", {F, F, 1, 1});
  c->fc_b1 = add_param(c, "fcomb.layers.2.bias", {F});
  c->fc_w2 = add_param(c, "fcomb.layers.4.weight", {Co, F, 1, 1});
  c->fc_b2 = add_param(c, "fcomb.layers.4.bias", {Co});
  if (F != 8 && F != 16 && F != 32) FAIL(PU_ERR_INVALID, "num_filters[0]=%d unsupported by the fused Fcomb kernel (8, 16 or 32)", F);
  if (Co > F) FAIL(PU_ERR_INVALID, "num_classes %d > num_filters[0] %d unsupported", Co, F);
  if (MM_ > 256) FAIL(PU_ERR_INVALID, "max_members %d > 256 unsupported", MM_);

  // ---- scratch
  const long HW = (long)H * W;
  c->gn_part = alloc_f32(c, (size_t)mb * 64 * c->max_gn_c * 2);
  c->gn_part2 = alloc_f32(c, (size_t)mb * 64 * c->max_gn_c * 2);
  c->gn_coef2 = alloc_f32(c, (size_t)mb * c->max_gn_c * 3);
  c->gn_rows = c->gn_rows_per_img ? alloc_f32(c, (size_t)mb * c->gn_rows_per_img) : nullptr;
  c->bias_part = alloc_f32(c, (size_t)256 * 1024);
  c->wg_slab_floats = c->dt == PU_F32 ? 0 : 32L * 1024 * 1024;
  c->wg_slab = c->wg_slab_floats ? alloc_f32(c, (size_t)c->wg_slab_floats * pu_ctx::NSLAB) : nullptr;
  {  // dv scratch: the largest GroupNorm input
    size_t mx = 0;
    auto upd = [&](const Block& b) { if (b.is_block) { size_t a = (size_t)b.x.v.H * b.x.v.W * b.x.v.C, q = (size_t)b.c0.v.H * b.c0.v.W * b.c0.v.C; if (a > mx) mx = a; if (q > mx) mx = q; } };
    for (auto& b : c->enc) upd(b);
    for (auto& b : c->dec) upd(b);
    size_t o = (size_t)H * W * c_last; if (o > mx) mx = o;
    c->dv_scratch = TV(); c->dv_scratch.p = arena_alloc(c, mx * mb * esz);
  }
  c->z = alloc_f32(c, (size_t)MM_ * mb * L); c->dz = alloc_f32(c, (size_t)MM_ * mb * L);
  c->preds = alloc_f32(c, (size_t)mb * MM_ * Co * HW); c->dpreds = alloc_f32(c, (size_t)mb * MM_ * Co * HW);
  c->kl = alloc_f32(c, mb); c->kl2 = alloc_f32(c, mb); c->scal = alloc_f32(c, PU_NUM_SCALARS);
  c->fc_feat = alloc_act(c, mb, H, W, F);
  c->fc_z = alloc_f32(c, (size_t)mb * L);
  return PU_OK;
}

// ------------------------------------------------------------------ op wrappers
static const float* P(pu_ctx* c, int64_t off) { return off < 0 ? nullptr : c->params + off; }
static float* G(pu_ctx* c, int64_t off) { return off < 0 ? nullptr : c->grads + off; }

static int ensure_packed(pu_ctx* c, hipStream_t s) {
  if (!c->params) FAIL(PU_ERR_STATE, "pu_bind_params has not been called");
  if (c->packed_valid) return PU_OK;
  int r = dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    CKH(launch_pack<T>(c->params, c->packed, c->descs_dev, (int)c->descs.size(), s));
    return PU_OK;
  });
  if (r == PU_OK) c->packed_valid = true;
  return r;
}

template <typename T>
static int conv_fwd(pu_ctx* c, const ConvL& L, TV in, TV out, int B, bool relu, const TV* res, int accumulate, hipStream_t s,
                    float* stat = nullptr, int stat_cap = 0, int* stat_slots = nullptr) {
  ConvArgs a; memset(&a, 0, sizeof a);
  if (stat_slots) *stat_slots = 0;
  if (stat && c->fused_stats) { a.stat_out = stat; a.stat_cap = stat_cap; a.stat_slots = stat_slots; }
  a.in = in.p; a.in_ld = in.ld;
  a.Cin = in.C;                       // channels present in the tensor (>= L.cin; extra planes meet zero-padded weights)
  a.wpk = (char*)c->packed + (size_t)L.pk_fwd * c->esz; a.cin_pk = L.cin_pk; a.cout_pk = L.rows_fwd; a.taps = L.ks * L.ks;
  a.bias = P(c, L.b_off);
  a.res = res ? res->p : nullptr; a.res_ld = res ? res->ld : 0;
  a.out = out.p; a.out_ld = out.ld; a.Cout = L.cout;
  a.B = B; a.H = out.H; a.W = out.W; a.relu = relu ? 1 : 0; a.accumulate = accumulate; a.frag_layout = L.frag ? 1 : 0;
  a.mfma16 = L.m16_fwd ? 1 : 0;
  if (a.Cin > L.cin_pk) a.Cin = L.cin_pk;
  CKH(launch_conv<T>(a, s));
  return PU_OK;
}
// dx (+)= dgrad(dy)
struct GnbReq { const GNL* n = nullptr; TV x; int train = 0; uint64_t seed = 0; int slots = 0; };   // in: the GroupNorm whose dy this data gradient is; out: slots
template <typename T>
static int conv_dgrad(pu_ctx* c, const ConvL& L, TV dy, TV dx, int B, int accumulate, hipStream_t s, GnbReq* gq = nullptr,
                      const TV* relu_mask = nullptr) {
  ConvArgs a; memset(&a, 0, sizeof a);
  if (gq) gq->slots = 0;
  if (gq && gq->n && gq->n->fuse_bwd && c->gn_rows && !accumulate && dx.H > 0) {
    const GNL& n = *gq->n;
    a.gnb.x = gq->x.p; a.gnb.x_ld = gq->x.ld; a.gnb.coef = n.coef; a.gnb.C = n.C;
    a.gnb.drop.drop_p = (n.dropout && gq->train) ? c->cfg.dropout_p : 0.f; a.gnb.drop.seed = gq->seed; a.gnb.drop.stream = n.drop_stream; a.gnb.drop.b0 = 0;
    if (a.gnb.drop.drop_p > 0.f && c->drop_masks && n.drop_site >= 0 && B == c->drop_masks_B)
      a.gnb.drop.mask = c->drop_masks + c->drop_sites[n.drop_site].off * (size_t)c->drop_masks_B;
    if (a.gnb.drop.drop_p > 0.f) a.gnb.drop.bits = n.keep_bits;
    a.gnb.part = c->gn_rows; a.gnb.cap = (int)(c->gn_rows_per_img / ((size_t)n.C * 2)); a.gnb.slots = &gq->slots;
  }
  a.in = dy.p; a.in_ld = dy.ld; a.Cin = L.cout;
  a.wpk = (char*)c->packed + (size_t)L.pk_bwd * c->esz; a.cin_pk = L.cout_pk; a.cout_pk = L.rows_bwd; a.taps = L.ks * L.ks;
  a.bias = nullptr; a.res = nullptr;
  if (relu_mask) { a.relu_mask = relu_mask->p; a.relu_mask_ld = relu_mask->ld; }
  a.out = dx.p; a.out_ld = dx.ld;
  a.Cout = dx.C;                      // write every allocated plane (planes >= L.cin receive zeros from zero-padded weights)
  if (a.Cout > L.rows_bwd) a.Cout = L.rows_bwd;
  a.B = B; a.H = dx.H; a.W = dx.W; a.relu = 0; a.accumulate = accumulate; a.frag_layout = L.frag ? 1 : 0;
  a.mfma16 = L.m16_bwd ? 1 : 0;
  CKH(launch_conv<T>(a, s));
  return PU_OK;
}
// fork: the side stream waits for everything enqueued on `s` so far; returns the stream the weight gradient runs on
static int fork_side(pu_ctx* c, hipStream_t s, hipStream_t* out) {
  *out = s;
  if (!c->use_side) return PU_OK;
  hipEvent_t e = c->evs[c->ev_next++ % c->evs.size()];
  CKH(hipEventRecord(e, s));
  CKH(hipStreamWaitEvent(c->side, e, 0));
  c->side_dirty = true; *out = c->side;
  return PU_OK;
}
// join: `s` waits for the side stream (called once at the end of every backward)
static int join_side(pu_ctx* c, hipStream_t s) {
  if (!c->use_side || !c->side_dirty) return PU_OK;
  hipEvent_t e = c->evs[c->ev_next++ % c->evs.size()];
  CKH(hipEventRecord(e, c->side));
  CKH(hipStreamWaitEvent(s, e, 0));
  c->side_dirty = false;
  return PU_OK;
}
// second side stream: the two latent encoders run beside the U-Net (forward and backward)
static int fork2(pu_ctx* c, hipStream_t s, hipStream_t* out) {
  *out = s;
  if (!c->use_side || !c->side2) return PU_OK;
  hipEvent_t e = c->evs[c->ev_next++ % c->evs.size()];
  CKH(hipEventRecord(e, s));
  CKH(hipStreamWaitEvent(c->side2, e, 0));
  *out = c->side2;
  return PU_OK;
}
static int join2(pu_ctx* c, hipStream_t s, hipStream_t s2) {
  if (s2 == s) return PU_OK;
  hipEvent_t e = c->evs[c->ev_next++ % c->evs.size()];
  CKH(hipEventRecord(e, s2));
  CKH(hipStreamWaitEvent(s, e, 0));
  return PU_OK;
}
template <typename T> static int conv_bgrad(pu_ctx* c, TV dy, int B, float* d0, float* d1, hipStream_t s);
// weight gradient (+ bias gradient d0/d1 = column sums of dy when d0 != null)
template <typename T>
static int conv_wgrad(pu_ctx* c, const ConvL& L, TV dy, TV in, int B, hipStream_t s, float* d0 = nullptr, float* d1 = nullptr,
                      const float* inv_dev = nullptr) {
  WgradArgs a; memset(&a, 0, sizeof a);
  a.inv_scale_dev = inv_dev;
  if (sizeof(T) == 2) { a.dbias0 = d0; a.dbias1 = d1; }
  else if (d0) { int r = conv_bgrad<T>(c, dy, B, d0, d1, s); if (r) return r; }
  a.dy = dy.p; a.dy_ld = dy.ld; a.Cout = L.cout; a.in = in.p; a.in_ld = in.ld; a.Cin = L.cin;
  a.dw = G(c, L.w_off); a.B = B; a.H = dy.H; a.W = dy.W; a.taps = L.ks * L.ks;
  a.slab = c->wg_slab; a.slab_floats = c->wg_slab_floats; a.inv_scale = c->inv_scale;
  a.overwrite = (c->grad_overwrite && sizeof(T) == 2) ? 1 : 0;
  hipStream_t ws = s;
  if constexpr (sizeof(T) == 2) {
    int r = fork_side(c, s, &ws); if (r) return r;
    // (chaining the slab reduction of weight gradient k into the prologue of weight gradient k + 1 - no stand-alone reduce kernel -
    //  was measured and rejected: the HBM-latency-bound prologue delays every main kernel, 40.7 vs 37.4 ms per step)
  }
  CKH(launch_wgrad<T>(a, ws));
  return PU_OK;
}
static int bias_chunks(long npix) { long n = npix / 256; if (n < 1) n = 1; if (n > 256) n = 256; return (int)n; }
template <typename T>
static int conv_bgrad(pu_ctx* c, TV dy, int B, float* d0, float* d1, hipStream_t s) {
  TV t = with_b(dy, B);
  CKH(launch_bias_grad<T>(t, d0, d1, c->bias_part, bias_chunks((long)B * dy.H * dy.W), c->inv_scale, s));
  return PU_OK;
}

static GNArgs gn_args(pu_ctx* c, const GNL& n, TV x, TV y, int B, int train, uint64_t seed) {
  GNArgs a; memset(&a, 0, sizeof a);
  a.x = with_b(x, B); a.y = with_b(y, B); a.G = n.G; a.eps = 1e-5f;
  a.gamma = P(c, n.g_off); a.beta = P(c, n.b_off);
  a.scale = n.ss_off >= 0 ? P(c, n.ss_off) : nullptr; a.shift = n.ss_off >= 0 ? P(c, n.ss_off + n.C) : nullptr;
  a.resample = n.resample;
  a.drop_p = (n.dropout && train) ? c->cfg.dropout_p : 0.f; a.drop_seed = seed; a.drop_stream = n.drop_stream;
  if (a.drop_p > 0.f && c->drop_masks && n.drop_site >= 0 && B == c->drop_masks_B)      // injected masks: [site][B][H][W][C] uint8
    a.drop_mask = c->drop_masks + c->drop_sites[n.drop_site].off * (size_t)c->drop_masks_B;
  if (a.drop_p > 0.f) a.keep_bits = n.keep_bits;
  a.part = c->gn_part; a.stat = n.stat; a.coef = n.coef; a.nchunk = n.nchunk;
  return a;
}
struct StatSrc { const float* p0 = nullptr; int n0 = 0, c0 = 0; const float* p1 = nullptr; int n1 = 0; };
static Block& blk(pu_ctx* c, int id) { return id >= 1000 ? c->dec[id - 1000] : c->enc[id]; }
// statistics of a block's input, when every producer of x delivered them in this forward
static StatSrc src_of_x(pu_ctx* c, const Block& b) {
  StatSrc q;
  if (b.src0 < 0) return q;
  const Block& p0 = blk(c, b.src0);
  if (!p0.st_out || p0.sl_out <= 0) return q;
  if (b.src1 >= 0) {
    const Block& p1 = blk(c, b.src1);
    if (!p1.st_out || p1.sl_out <= 0) return q;
    q.p1 = p1.st_out; q.n1 = p1.sl_out;
  }
  q.p0 = p0.st_out; q.n0 = p0.sl_out; q.c0 = p0.cout;
  return q;
}
template <typename T>
static int gn_fwd(pu_ctx* c, const GNL& n, TV x, TV y, int B, int train, uint64_t seed, hipStream_t s, const StatSrc* st = nullptr) {
  GNArgs a = gn_args(c, n, x, y, B, train, seed);
  if (st && st->p0) { a.ps0 = st->p0; a.ns0 = st->n0; a.pc0 = st->c0; a.ps1 = st->p1; a.ns1 = st->n1; }
  CKH(launch_gn_fwd<T>(a, s));
  return PU_OK;
}
template <typename T>
static int gn_bwd(pu_ctx* c, const GNL& n, TV x, TV y, TV dy, TV dx, int accumulate, int B, int train, uint64_t seed, hipStream_t s,
                  const TV* add = nullptr, int dv_slots = 0) {
  GNBwdArgs a; memset(&a, 0, sizeof a);
  if (dv_slots > 0) { a.rows = c->gn_rows; a.nrows = dv_slots; }      // dy holds dv, pass 1 was done by the data-gradient epilogue
  if (add && n.resample == RS_NONE) a.add = with_b(*add, B);
  a.f = gn_args(c, n, x, y, B, train, seed);
  a.dy = with_b(dy, B);
  a.dv = with_b(x, B); a.dv.p = c->dv_scratch.p; a.dv.ld = x.C;
  a.dx = with_b(dx, B); a.accumulate = accumulate;
  a.dgamma = G(c, n.g_off); a.dbeta = G(c, n.b_off);
  a.dscale = n.ss_off >= 0 ? G(c, n.ss_off) : nullptr; a.dshift = n.ss_off >= 0 ? G(c, n.ss_off + n.C) : nullptr;
  a.part2 = c->gn_part2; a.coef2 = c->gn_coef2; a.inv_scale = c->inv_scale;
  CKH(launch_gn_bwd<T>(a, s));
  return PU_OK;
}

// ------------------------------------------------------------------ data-parallel gradient buckets
// The backward completes the flat gradient buffer from its top (out_conv) downwards: out, decoder blocks reversed, encoder blocks
// reversed is exactly descending flat order, so the U-Net buckets are contiguous ranges cut at block boundaries; the two latent
// encoders + Fcomb (the tail of the buffer) form one more bucket, finished on the second side stream.
static void plan_buckets(pu_ctx* c) {
  for (auto& b : c->enc) b.bucket_close = -1;
  for (auto& b : c->dec) b.bucket_close = -1;
  for (auto& q : c->buckets) for (auto& e : q.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  c->buckets.clear();
  const int n = c->want_buckets;
  if (n <= 0) return;
  const int64_t unet_hi = c->prior.convs[0].w_off;          // first parameter after the U-Net
  std::vector<Block*> order;
  for (int j = (int)c->dec.size() - 1; j >= 0; --j) order.push_back(&c->dec[j]);
  for (int i = (int)c->enc.size() - 1; i >= 1; --i) order.push_back(&c->enc[i]);     // enc[0] (stem) always closes the last bucket
  int64_t hi = unet_hi; int k = 0;
  for (Block* b : order) {
    if (k >= n - 1) break;
    const int64_t target = unet_hi - (int64_t)((double)unet_hi * (k + 1) / n);
    if (b->p_lo <= target) {
      pu_ctx::Bucket q; q.lo = b->p_lo; q.hi = hi; c->buckets.push_back(q);
      b->bucket_close = (int)c->buckets.size() - 1; hi = b->p_lo; ++k;
    }
  }
  { pu_ctx::Bucket q; q.lo = unet_hi; q.hi = c->nparams; c->buckets.push_back(q); }     // latent encoders + Fcomb
  { pu_ctx::Bucket q; q.lo = 0; q.hi = hi; c->buckets.push_back(q); }                   // shallow encoder levels: closes with the backward
  for (auto& q : c->buckets) for (auto& e : q.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
}
// mark bucket k complete at this point of the enqueue order: everything already enqueued on `s` (and on the weight-gradient
// side stream, and on `s2` when given) belongs to it or to an earlier bucket
static int record_bucket(pu_ctx* c, int k, hipStream_t s, hipStream_t s2 = nullptr) {
  if (k < 0 || k >= (int)c->buckets.size()) return PU_OK;
  pu_ctx::Bucket& q = c->buckets[k];
  q.rec[0] = q.rec[1] = q.rec[2] = false;
  if (q.ev[0]) { CKH(hipEventRecord(q.ev[0], s)); q.rec[0] = true; }
  if (c->use_side && c->side_dirty && q.ev[1]) { CKH(hipEventRecord(q.ev[1], c->side)); q.rec[1] = true; }
  if (s2 && s2 != s && q.ev[2]) { CKH(hipEventRecord(q.ev[2], s2)); q.rec[2] = true; }
  return PU_OK;
}

// ------------------------------------------------------------------ U-Net forward / backward
template <typename T>
static int block_fwd(pu_ctx* c, Block& b, int B, int train, uint64_t seed, hipStream_t s) {
  int r;
  if (!b.is_block) return conv_fwd<T>(c, b.conv0, b.x.v, b.out.v, B, false, nullptr, 0, s, b.st_out, b.cap_out, &b.sl_out);
  // x = conv0(resample(silu(norm0(x))))            networks.py:168
  const StatSrc sx = src_of_x(c, b);
  if ((r = gn_fwd<T>(c, b.n0, b.x.v, b.a0.v, B, train, seed, s, &sx))) return r;
  if ((r = conv_fwd<T>(c, b.conv0, b.a0.v, b.c0.v, B, false, nullptr, 0, s, b.st_c0, b.cap_c0, &b.sl_c0))) return r;
  // x = silu(shift + norm1(x) * (scale + 1)); dropout   networks.py:170-177
  StatSrc sc; if (b.sl_c0 > 0) { sc.p0 = b.st_c0; sc.n0 = b.sl_c0; sc.c0 = b.cout; }
  if ((r = gn_fwd<T>(c, b.n1, b.c0.v, b.h1.v, B, train, seed, s, &sc))) return r;
  // x = conv1(x) + skip(orig)                        networks.py:177-179   (conv1 is always the final writer of `out`)
  if (b.skip == SK_CONV) {
    TV sin = b.x.v;
    if (b.up || b.down) { CKH(launch_resample<T>(with_b(b.x.v, B), with_b(b.skin.v, B), b.down ? RS_DOWN : RS_UP, s)); sin = b.skin.v; }
    if ((r = conv_fwd<T>(c, b.skipc, sin, b.out.v, B, false, nullptr, 0, s))) return r;
    return conv_fwd<T>(c, b.conv1, b.h1.v, b.out.v, B, false, nullptr, 1, s, b.st_out, b.cap_out, &b.sl_out);
  }
  if (b.skip == SK_RESAMPLE) {
    CKH(launch_resample<T>(with_b(b.x.v, B), with_b(b.skin.v, B), b.down ? RS_DOWN : RS_UP, s));
    return conv_fwd<T>(c, b.conv1, b.h1.v, b.out.v, B, false, &b.skin.v, 0, s, b.st_out, b.cap_out, &b.sl_out);
  }
  return conv_fwd<T>(c, b.conv1, b.h1.v, b.out.v, B, false, &b.x.v, 0, s, b.st_out, b.cap_out, &b.sl_out);
}

template <typename T>
static int block_bwd(pu_ctx* c, Block& b, int B, int train, uint64_t seed, bool need_dx, hipStream_t s) {
  int r;
  TV dout = b.out.g;
  if (!b.is_block) {
    return conv_wgrad<T>(c, b.conv0, dout, b.x.v, B, s, G(c, b.conv0.b_off), nullptr);
  }
  // conv1
  if ((r = conv_wgrad<T>(c, b.conv1, dout, b.h1.v, B, s, G(c, b.conv1.b_off), b.skip == SK_CONV ? G(c, b.skipc.b_off) : nullptr))) return r;
  GnbReq q1; q1.n = &b.n1; q1.x = b.c0.v; q1.train = train; q1.seed = seed;
  if ((r = conv_dgrad<T>(c, b.conv1, dout, b.h1.g, B, 0, s, &q1))) return r;
  // skip path
  if (b.skip == SK_CONV) {
    TV sin = (b.up || b.down) ? b.skin.v : b.x.v;
    if ((r = conv_wgrad<T>(c, b.skipc, dout, sin, B, s))) return r;
    if (need_dx) {
      if (b.up || b.down) {
        if ((r = conv_dgrad<T>(c, b.skipc, dout, b.skin.g, B, 0, s))) return r;
        CKH(launch_resample_bwd<T>(with_b(b.skin.g, B), with_b(b.x.g, B), b.down ? RS_DOWN : RS_UP, take_acc(c, b.x), s));
      } else if ((r = conv_dgrad<T>(c, b.skipc, dout, b.x.g, B, take_acc(c, b.x), s))) return r;
    }
  } else if (need_dx) {
    if (b.skip == SK_RESAMPLE) CKH(launch_resample_bwd<T>(with_b(dout, B), with_b(b.x.g, B), b.down ? RS_DOWN : RS_UP, take_acc(c, b.x), s));
    else if (b.n0.resample != RS_NONE || sizeof(T) == 4) CKH(launch_add<T>(with_b(dout, B), with_b(b.x.g, B), take_acc(c, b.x), s));
    // else (identity skip, 16-bit engines): dout is added inside the pass 2 of norm0's backward below
  }
  // norm1 (+scale/shift, dropout) -> c0.g
  if ((r = gn_bwd<T>(c, b.n1, b.c0.v, b.h1.v, b.h1.g, b.c0.g, 0, B, train, seed, s, nullptr, q1.slots))) return r;
  // conv0
  if ((r = conv_wgrad<T>(c, b.conv0, b.c0.g, b.a0.v, B, s, G(c, b.conv0.b_off), nullptr))) return r;
  GnbReq q0; q0.n = &b.n0; q0.x = b.x.v; q0.train = train; q0.seed = seed;
  if ((r = conv_dgrad<T>(c, b.conv0, b.c0.g, b.a0.g, B, 0, s, &q0))) return r;
  // norm0 (+resample) -> x.g   (parameter gradients are needed even when dx is not)
  TV dx = b.x.g; int acc = 0;
  if (need_dx) acc = take_acc(c, b.x);
  else { dx = with_b(b.x.v, B); dx.p = c->dv_scratch.p; dx.ld = b.x.v.C; }      // never happens for blocks (x always needs grad)
  const bool fold_skip = need_dx && b.skip == SK_IDENTITY && b.n0.resample == RS_NONE && sizeof(T) != 4;
  return gn_bwd<T>(c, b.n0, b.x.v, b.a0.v, b.a0.g, dx, acc, B, train, seed, s, fold_skip ? &dout : nullptr, q0.slots);
}

template <typename T>
static int unet_forward(pu_ctx* c, int B, int train, uint64_t seed, hipStream_t s) {
  int r;
  for (auto& b : c->enc) if ((r = block_fwd<T>(c, b, B, train, seed, s))) return r;
  for (auto& b : c->dec) if ((r = block_fwd<T>(c, b, B, train, seed, s))) return r;
  Act& last = c->dec.back().out;
  StatSrc so; { const Block& lb = c->dec.back(); if (lb.st_out && lb.sl_out > 0) { so.p0 = lb.st_out; so.n0 = lb.sl_out; so.c0 = lb.cout; } }
  if ((r = gn_fwd<T>(c, c->out_norm, last.v, c->out_a.v, B, train, seed, s, &so))) return r;
  if ((r = conv_fwd<T>(c, c->out_conv, c->out_a.v, c->feat.v, B, false, nullptr, 0, s))) return r;
  c->unet_B = B; c->unet_train = train; c->unet_seed = seed;
  return PU_OK;
}
// expects feat.g filled; clears/uses the gradient flags of all U-Net tensors
template <typename T, typename Mid>
static int unet_backward(pu_ctx* c, hipStream_t s, Mid&& mid) {
  int r; const int B = c->unet_B, train = c->unet_train; const uint64_t seed = c->unet_seed;
  if (B <= 0) FAIL(PU_ERR_STATE, "U-Net backward without a forward");
  std::fill(c->flags.begin(), c->flags.end(), 0);
  Act& last = c->dec.back().out;
  if ((r = conv_wgrad<T>(c, c->out_conv, c->feat.g, c->out_a.v, B, s, G(c, c->out_conv.b_off), nullptr))) return r;
  GnbReq qo; qo.n = &c->out_norm; qo.x = last.v; qo.train = train; qo.seed = seed;
  if ((r = conv_dgrad<T>(c, c->out_conv, c->feat.g, c->out_a.g, B, 0, s, &qo))) return r;
  if ((r = gn_bwd<T>(c, c->out_norm, last.v, c->out_a.v, c->out_a.g, last.g, take_acc(c, last), B, train, seed, s, nullptr, qo.slots))) return r;
  const int jmid = (int)c->dec.size() * 2 / 3;           // `mid` is enqueued after the first third of the decoder blocks
  for (int j = (int)c->dec.size() - 1; j >= 0; --j) {
    if ((r = block_bwd<T>(c, c->dec[j], B, train, seed, true, s))) return r;
    if (c->dec[j].bucket_close >= 0 && (r = record_bucket(c, c->dec[j].bucket_close, s))) return r;
    if (j == jmid && (r = mid())) return r;
  }
  for (int i = (int)c->enc.size() - 1; i >= 0; --i) {
    if ((r = block_bwd<T>(c, c->enc[i], B, train, seed, i > 0, s))) return r;
    if (c->enc[i].bucket_close >= 0 && (r = record_bucket(c, c->enc[i].bucket_close, s))) return r;
  }
  return PU_OK;
}

// ------------------------------------------------------------------ Gaussian encoders
template <typename T>
static int gauss_forward(pu_ctx* c, GaussNet& g, int B, hipStream_t s) {
  int r;
  for (size_t i = 0; i < g.convs.size(); ++i) {
    if (g.pool_before[i]) CKH(launch_maxpool<T>(with_b(g.outs[i - 1].v, B), with_b(g.ins[i].v, B), s));
    if ((r = conv_fwd<T>(c, g.convs[i], g.ins[i].v, g.outs[i].v, B, true, nullptr, 0, s))) return r;
  }
  const int L = c->cfg.latent_dim;
  CKH(launch_heads_fwd<T>(with_b(g.outs.back().v, B), P(c, g.wmu), P(c, g.bmu), P(c, g.wls), P(c, g.bls), L, g.hbuf, g.mu, g.ls, s));
  g.lastB = B;
  return PU_OK;
}
// expects g.dmu / g.dls filled
template <typename T>
static int gauss_backward(pu_ctx* c, GaussNet& g, hipStream_t s) {
  int r; const int B = g.lastB, L = c->cfg.latent_dim;
  if (B <= 0) FAIL(PU_ERR_STATE, "Gaussian-encoder backward without a forward");
  // f16: this sub-graph gets its own power-of-two scale, chosen ON THE DEVICE from the max-abs of its entry gradients.  The
  // static loss scale is sized for the reconstruction gradient; the KL gradient entering here can be 10^6 times larger at
  // initialisation (overflow) or, during the beta_1 = 0 warm-up, only the tiny reconstruction part remains (underflow).
  const float* inv_dev = nullptr;
  if (c->dt == PU_F16) { CKH(launch_enc_rescale(g.dmu, g.dls, B * L, 4096.f, g.enc_inv, s)); inv_dev = g.enc_inv; }
  TV lastg = with_b(g.outs.back().g, B);
  CKH(launch_heads_bwd<T>(with_b(g.outs.back().v, B), lastg, g.hbuf, P(c, g.wmu), P(c, g.wls), g.dmu, g.dls, L,
                          G(c, g.wmu), G(c, g.bmu), G(c, g.wls), G(c, g.bls), c->inv_scale, s, inv_dev));
  static const bool no_mask_fuse = getenv("PU_NO_RELU_MASK_FUSE") != nullptr;      // diagnostic: every ReLU backward as its own pass
  bool masked_by_dgrad = false;
  for (int i = (int)g.convs.size() - 1; i >= 0; --i) {
    TV dy = with_b(g.outs[i].g, B);
    // ReLU backward of this layer's output: folded into the max-pool backward that produced dy when a pool follows the layer
    // (networks: conv -> ReLU -> pool; the routed gradient is dropped where the window maximum is not positive), else its own pass
    // ... or into the data gradient of the layer above, which wrote dy (no pool in between: ins[i + 1] aliases outs[i])
    const bool masked_by_pool = i + 1 < (int)g.convs.size() && g.pool_before[i + 1];
    if (!masked_by_pool && !masked_by_dgrad) CKH(launch_relu_bwd<T>(with_b(g.outs[i].v, B), dy, s));
    masked_by_dgrad = false;
    if ((r = conv_wgrad<T>(c, g.convs[i], g.outs[i].g, g.ins[i].v, B, s, G(c, g.convs[i].b_off), nullptr, inv_dev))) return r;
    if (i == 0) break;
    const bool fuse_mask = !g.pool_before[i] && g.convs[i].frag && !no_mask_fuse;
    const TV mask = with_b(g.outs[i - 1].v, B);
    if ((r = conv_dgrad<T>(c, g.convs[i], g.outs[i].g, g.ins[i].g, B, 0, s, nullptr, fuse_mask ? &mask : nullptr))) return r;
    masked_by_dgrad = fuse_mask;
    if (g.pool_before[i]) CKH(launch_maxpool_bwd<T>(with_b(g.outs[i - 1].v, B), with_b(g.ins[i].g, B), with_b(g.outs[i - 1].g, B), s, true));
    // else ins[i] aliases outs[i-1] (same Act): gradient already in place
  }
  return PU_OK;
}

static FcombArgs fcomb_args(pu_ctx* c, TV feat, int bcast, const float* z, int B, int M, float* out) {
  FcombArgs a; memset(&a, 0, sizeof a);
  a.feat = feat; a.bcast = bcast; a.z = z;
  a.w0 = P(c, c->fc_w0); a.b0 = P(c, c->fc_b0); a.w1 = P(c, c->fc_w1); a.b1 = P(c, c->fc_b1); a.w2 = P(c, c->fc_w2); a.b2 = P(c, c->fc_b2);
  a.F = c->cfg.num_filters[0]; a.L = c->cfg.latent_dim; a.Cout = c->cfg.num_classes; a.B = B; a.M = M; a.out = out;
  return a;
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int pu_abi_version(void) { return PU_ABI; }
const char* pu_last_error(pu_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int pu_create(const pu_config* cfg, int device, pu_ctx** out) {
  if (!cfg || !out) { g_create_err = "null argument"; return PU_ERR_INVALID; }
  pu_ctx* c = new pu_ctx();
  c->cfg = *cfg; c->device = device; c->dt = cfg->dtype;
  auto bail = [&](int code) { g_create_err = c->err; delete c; return code; };
  if (cfg->dtype < 0 || cfg->dtype > 2) { c->err = "bad dtype"; return bail(PU_ERR_INVALID); }
  c->esz = cfg->dtype == PU_F32 ? 4 : 2;
  if (cfg->depth < 1 || cfg->depth > PU_MAX_LEVELS || cfg->max_batch < 1 || cfg->max_members < 1 || cfg->latent_dim < 1 || cfg->latent_dim > 64) {
    c->err = "bad depth/max_batch/max_members/latent_dim"; return bail(PU_ERR_INVALID);
  }
  const int div = 1 << (cfg->depth - 1);
  if (cfg->H % div || cfg->W % div || (cfg->H / div) % 8 || (cfg->W / div) % 8) {
    c->err = "H, W must be divisible by 2^(depth-1) with the deepest level a multiple of 8 (networks.py concat constraint + 8x8 MFMA pixel tile)";
    return bail(PU_ERR_INVALID);
  }
  if (cfg->dropout_p < 0.f || cfg->dropout_p >= 1.f) { c->err = "bad dropout_p"; return bail(PU_ERR_INVALID); }
  if (const char* e = getenv("PU_GN_FUSE_BWD_MINHW")) c->fuse_bwd_minhw = atol(e);
  c->planning = true;
  int r = build_plan(c);
  if (r != PU_OK) return bail(r);
  if (c->max_tensor_elems >= ((size_t)1 << 32)) {
    // the convolution kernels keep per-pixel ELEMENT offsets in 32 bits (kernels_conv.hip staging plan): a tensor of 2^32 elements or
    // more would wrap silently
    char b_[256]; snprintf(b_, sizeof b_, "max_batch %d x %d x %d: the largest activation has %zu elements (>= 2^32, the engine's 32-bit pixel "
                           "offset range); lower max_batch", cfg->max_batch, cfg->H, cfg->W, c->max_tensor_elems);
    c->err = b_; return bail(PU_ERR_INVALID);
  }
  DeviceGuard dg(device);                                  // the caller's current device is restored on every return path
  int cur_dev = -1;
  hipError_t e = hipGetDevice(&cur_dev);
  if (e != hipSuccess || cur_dev != device) { c->err = std::string("hipSetDevice(") + std::to_string(device) + ") failed"; return bail(PU_ERR_HIP); }
  c->arena_size = c->arena_used + 4096;
  if ((e = hipMalloc(&c->arena, c->arena_size)) != hipSuccess) { c->err = std::string("hipMalloc(arena): ") + hipGetErrorString(e); return bail(PU_ERR_NOMEM); }
  if ((e = hipMemset(c->arena, 0, c->arena_size)) != hipSuccess) { c->err = "hipMemset(arena)"; (void)hipFree(c->arena); return bail(PU_ERR_HIP); }
  c->planning = false;
  r = build_plan(c);                                      // second pass: identical walk, real pointers
  if (r != PU_OK) { (void)hipFree(c->arena); return bail(r); }
  if ((e = hipMalloc(&c->packed, (size_t)c->packed_elems * c->esz + 256)) != hipSuccess) { c->err = "hipMalloc(packed)"; (void)hipFree(c->arena); return bail(PU_ERR_NOMEM); }
  if ((e = hipMalloc(&c->descs_dev, c->descs.size() * sizeof(PackDesc))) != hipSuccess) { c->err = "hipMalloc(descs)"; (void)hipFree(c->arena); (void)hipFree(c->packed); return bail(PU_ERR_NOMEM); }
  if ((e = hipMemcpy(c->descs_dev, c->descs.data(), c->descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice)) != hipSuccess) {
    c->err = "hipMemcpy(descs)"; (void)hipFree(c->arena); (void)hipFree(c->packed); (void)hipFree(c->descs_dev); return bail(PU_ERR_HIP);
  }
  {
    // complement of the convolution parameter ranges inside [0, nparams)
    std::vector<std::pair<int64_t, int64_t>> cr = c->conv_ranges;
    std::sort(cr.begin(), cr.end());
    std::vector<long> zr; int64_t pos = 0;
    for (auto& q : cr) { if (q.first > pos) { zr.push_back((long)pos); zr.push_back((long)q.first); } if (q.second > pos) pos = q.second; }
    if (pos < c->nparams) { zr.push_back((long)pos); zr.push_back((long)c->nparams); }
    c->n_zero_ranges = (int)zr.size() / 2;
    if (c->n_zero_ranges > 0) {
      if ((e = hipMalloc(&c->zero_ranges_dev, zr.size() * sizeof(long))) != hipSuccess ||
          (e = hipMemcpy(c->zero_ranges_dev, zr.data(), zr.size() * sizeof(long), hipMemcpyHostToDevice)) != hipSuccess) {
        c->err = "zero-range table"; (void)hipFree(c->arena); (void)hipFree(c->packed); (void)hipFree(c->descs_dev); return bail(PU_ERR_HIP);
      }
    }
  }
  {
    // the engine's side streams (weight gradients, latent encoders) get the HIGHEST stream priority: whatever shares the GPU with them at
    // default priority - in particular RCCL's reduction kernels of the bucketed gradient all-reduce, which run on torch's
    // communication stream while the rest of the backward is still being computed - yields to the compute kernels at dispatch
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { prio_least = prio_greatest = 0; (void)hipGetLastError(); }
    int prio = prio_greatest;
    if (const char* sp = getenv("PU_SIDE_PRIO")) prio = sp[0] == 'l' ? prio_least : sp[0] == 'n' ? 0 : prio_greatest;   // diagnostic: low / normal / high
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio) != hipSuccess) { c->side = nullptr; (void)hipGetLastError(); }
    if (hipStreamCreateWithPriority(&c->side2, hipStreamNonBlocking, prio) != hipSuccess) { c->side2 = nullptr; (void)hipGetLastError(); }
  }

  if (c->side) {
    c->evs.resize(256);
    for (auto& e : c->evs) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { c->use_side = false; e = nullptr; }
  } else c->use_side = false;
  if (getenv("PU_NO_FUSED_STATS") || c->dt == PU_F32) c->fused_stats = false;
  if (getenv("PU_NO_SIDE_STREAM") || c->dt == PU_F32) c->use_side = false;   // the fp32 parity path shares scratch (bias_part) and stays serial
  *out = c;
  return PU_OK;
}

int pu_destroy(pu_ctx* c) {
  if (!c) return PU_OK;
  DeviceGuard dg(c->device);
  for (auto& g : c->sample_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
  for (auto& q : c->buckets) for (auto& e : q.ev) if (e) (void)hipEventDestroy(e);
  if (c->drop_masks) (void)hipFree(c->drop_masks);
  for (auto e : c->evs) if (e) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->side2) (void)hipStreamDestroy(c->side2);

  if (c->ms_ws) (void)hipFree(c->ms_ws);
  if (c->arena) (void)(void)hipFree(c->arena);
  if (c->packed) (void)(void)hipFree(c->packed);
  if (c->descs_dev) (void)(void)hipFree(c->descs_dev);
  if (c->zero_ranges_dev) (void)hipFree(c->zero_ranges_dev);
  delete c;
  return PU_OK;
}

int pu_param_table(pu_ctx* c, const pu_param_desc** out, int* n) {
  if (!c || !out || !n) return PU_ERR_INVALID;
  *out = c->table.data(); *n = (int)c->table.size();
  return PU_OK;
}
int64_t pu_param_count(pu_ctx* c) { return c ? c->nparams : -1; }
int pu_profile_enable(int on) { prof_enable(on != 0); return PU_OK; }
int pu_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int64_t step, const float* skip_flag, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return PU_ERR_INVALID;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const hipError_t e = launch_adamw_flat(params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay,
                                         (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), (hipStream_t)stream, skip_flag);
  return e == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, void* stream) {
  return pu_adamw_step_guarded(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, stream);
}
int pu_set_overlap(pu_ctx* c, int on) {
  if (!c) return PU_ERR_INVALID;
  c->use_side = on != 0 && c->side && c->side2 && c->dt != PU_F32 && !getenv("PU_NO_SIDE_STREAM");      // (not on a hot path)
  return PU_OK;
}
int pu_profile_collect(pu_prof_entry* out, int max_entries) {
  static_assert(sizeof(pu_prof_entry) == sizeof(ProfEntry), "pu_prof_entry layout");
  return out ? prof_collect(reinterpret_cast<ProfEntry*>(out), max_entries) : 0;
}
int64_t pu_workspace_bytes(pu_ctx* c) { return c ? (int64_t)(c->arena_size + (size_t)c->packed_elems * c->esz) : -1; }

int pu_bind_params(pu_ctx* c, float* p, float* g) {
  if (!c || !p) return PU_ERR_INVALID;
  c->params = p; c->grads = g; c->packed_valid = false;
  return PU_OK;
}
int pu_bind_grads(pu_ctx* c, float* g) {
  if (!c || !g) return PU_ERR_INVALID;
  c->grads = g;                          // the packed compute-dtype weights stay valid: only the gradient destination moves
  return PU_OK;
}
int pu_params_changed(pu_ctx* c) { if (!c) return PU_ERR_INVALID; c->packed_valid = false; return PU_OK; }

double pu_elbo_fwd_flops(pu_ctx* c, int B, int M) {
  if (!c) return 0;
  const pu_config& cf = c->cfg;
  double f = 0;
  auto conv = [&](const ConvL& L, const TV& out) { f += 2.0 * out.H * out.W * (double)L.cout * L.cin * L.ks * L.ks; };
  auto blk = [&](const Block& b) { conv(b.conv0, b.is_block ? b.c0.v : b.out.v); if (b.is_block) { conv(b.conv1, b.out.v); if (b.skip == SK_CONV) conv(b.skipc, b.out.v); } };
  for (auto& b : c->enc) blk(b);
  for (auto& b : c->dec) blk(b);
  conv(c->out_conv, c->feat.v);
  for (GaussNet* g : {&c->prior, &c->post}) {
    for (size_t i = 0; i < g->convs.size(); ++i) conv(g->convs[i], g->outs[i].v);
    f += 2.0 * 2 * cf.latent_dim * cf.num_filters[cf.depth - 1];
  }
  const double F = cf.num_filters[0], HW = (double)cf.H * cf.W;
  f += (double)M * 2.0 * HW * (F * (F + cf.latent_dim) + F * F + F * cf.num_classes);
  return f * B;
}

static int check_B(pu_ctx* c, int B) {
  if (B < 1 || B > c->cfg.max_batch) FAIL(PU_ERR_INVALID, "batch %d outside [1, max_batch=%d]", B, c->cfg.max_batch);
  return PU_OK;
}

int pu_unet_fwd(pu_ctx* c, const float* x, float* feat, int B, int train, uint64_t seed, void* stream) {
  if (!c || !x) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  int r; if ((r = check_B(c, B))) return r;
  hipStream_t s = (hipStream_t)stream;
  if ((r = ensure_packed(c, s))) return r;
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    const long HW = (long)c->cfg.H * c->cfg.W;
    CKH(launch_nchw_to_nhwc<T>(x, (long)c->cfg.input_channels * HW, c->cfg.input_channels, nullptr, 0, with_b(c->x_in.v, B), s));
    int q = unet_forward<T>(c, B, train, seed, s); if (q) return q;
    if (feat) CKH(launch_nhwc_to_nchw<T>(with_b(c->feat.v, B), c->cfg.num_filters[0], feat, 0, s));
    return PU_OK;
  });
}
int pu_unet_bwd(pu_ctx* c, const float* dfeat, void* stream) {
  if (!c || !dfeat) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  if (!c->grads) FAIL(PU_ERR_STATE, "no gradient buffer bound");
  hipStream_t s = (hipStream_t)stream;
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    const int B = c->unet_B; if (B <= 0) FAIL(PU_ERR_STATE, "pu_unet_bwd without pu_unet_fwd");
    c->inv_scale = 1.f;
    const int F = c->cfg.num_filters[0]; const long HW = (long)c->cfg.H * c->cfg.W;
    CKH(launch_nchw_to_nhwc<T>(dfeat, (long)F * HW, F, nullptr, 0, with_b(c->feat.g, B), s));
    int q = unet_backward<T>(c, s, []() -> int { return PU_OK; }); if (q) return q;
    return join_side(c, s);
  });
}

static int gauss_input(pu_ctx* c, int which, const float* x, const float* target, int B, hipStream_t s) {
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    const long HW = (long)c->cfg.H * c->cfg.W; const int ci = c->cfg.input_channels;
    if (which == PU_PRIOR) CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, nullptr, 0, with_b(c->x_in.v, B), s));
    else CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, target, c->cfg.num_classes, with_b(c->xy_in.v, B), s));
    return PU_OK;
  });
}

int pu_gauss_fwd(pu_ctx* c, int which, const float* x, const float* target, float* mu, float* ls, int B, void* stream) {
  if (!c || !x || (which == PU_POSTERIOR && !target)) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  int r; if ((r = check_B(c, B))) return r;
  hipStream_t s = (hipStream_t)stream;
  if ((r = ensure_packed(c, s))) return r;
  if ((r = gauss_input(c, which, x, target, B, s))) return r;
  GaussNet& g = which == PU_PRIOR ? c->prior : c->post;
  r = dispatch(c, [&](auto t) -> int { typedef decltype(t) T; return gauss_forward<T>(c, g, B, s); });
  if (r) return r;
  const size_t n = (size_t)B * c->cfg.latent_dim * sizeof(float);
  if (mu) CKH(hipMemcpyAsync(mu, g.mu, n, hipMemcpyDeviceToDevice, s));
  if (ls) CKH(hipMemcpyAsync(ls, g.ls, n, hipMemcpyDeviceToDevice, s));
  return PU_OK;
}
int pu_gauss_bwd(pu_ctx* c, int which, const float* dmu, const float* dls, void* stream) {
  if (!c || !dmu || !dls) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  if (!c->grads) FAIL(PU_ERR_STATE, "no gradient buffer bound");
  hipStream_t s = (hipStream_t)stream;
  GaussNet& g = which == PU_PRIOR ? c->prior : c->post;
  if (g.lastB <= 0) FAIL(PU_ERR_STATE, "pu_gauss_bwd without pu_gauss_fwd");
  c->inv_scale = 1.f;
  const size_t n = (size_t)g.lastB * c->cfg.latent_dim * sizeof(float);
  CKH(hipMemcpyAsync(g.dmu, dmu, n, hipMemcpyDeviceToDevice, s));
  CKH(hipMemcpyAsync(g.dls, dls, n, hipMemcpyDeviceToDevice, s));
  return dispatch(c, [&](auto t) -> int { typedef decltype(t) T; int q = gauss_backward<T>(c, g, s); if (q) return q; return join_side(c, s); });
}

int pu_fcomb_fwd(pu_ctx* c, const float* feat, int64_t bstride, const float* z, float* out, int B, void* stream) {
  if (!c || !feat || !z || !out) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  int r; if ((r = check_B(c, B))) return r;
  if (!c->params) FAIL(PU_ERR_STATE, "pu_bind_params has not been called");
  hipStream_t s = (hipStream_t)stream;
  const int F = c->cfg.num_filters[0], L = c->cfg.latent_dim; const long HW = (long)c->cfg.H * c->cfg.W;
  const int bcast = bstride == 0 ? 1 : 0;
  if (!bcast && bstride != (int64_t)F * HW) FAIL(PU_ERR_INVALID, "feature batch stride must be 0 or F0*H*W");
  CKH(hipMemcpyAsync(c->fc_z, z, (size_t)B * L * sizeof(float), hipMemcpyDeviceToDevice, s));
  c->fc_B = B; c->fc_bcast = bcast;
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    TV fv = with_b(c->fc_feat.v, bcast ? 1 : B);
    CKH(launch_nchw_to_nhwc<T>(feat, (long)F * HW, F, nullptr, 0, fv, s));
    FcombArgs a = fcomb_args(c, fv, bcast, c->fc_z, B, 1, out);
    CKH(launch_fcomb_fwd<T>(a, s));
    return PU_OK;
  });
}
int pu_fcomb_bwd(pu_ctx* c, const float* dout, float* dfeat, float* dz, void* stream) {
  if (!c || !dout) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  if (!c->grads) FAIL(PU_ERR_STATE, "no gradient buffer bound");
  if (c->fc_B <= 0) FAIL(PU_ERR_STATE, "pu_fcomb_bwd without pu_fcomb_fwd");
  hipStream_t s = (hipStream_t)stream;
  const int B = c->fc_B, F = c->cfg.num_filters[0];
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    TV fv = with_b(c->fc_feat.v, c->fc_bcast ? 1 : B);
    FcombBwdArgs a; memset(&a, 0, sizeof a);
    a.f = fcomb_args(c, fv, c->fc_bcast, c->fc_z, B, 1, nullptr);
    a.dout = dout; a.dfeat = with_b(c->fc_feat.g, B); a.dfeat_accumulate = 0; if (!dfeat) a.dfeat.p = nullptr;
    a.dz = dz; a.inv_scale = 1.f;
    a.dw0 = G(c, c->fc_w0); a.db0 = G(c, c->fc_b0); a.dw1 = G(c, c->fc_w1); a.db1 = G(c, c->fc_b1); a.dw2 = G(c, c->fc_w2); a.db2 = G(c, c->fc_b2);
    CKH(launch_fcomb_bwd<T>(a, s));
    if (dfeat) CKH(launch_nhwc_to_nchw<T>(with_b(c->fc_feat.g, B), F, dfeat, 0, s));
    return PU_OK;
  });
}

int pu_elbo_fwd_bwd(pu_ctx* c, const float* x, const float* target, const float* eps, int B, int M, int recon_kind,
                    float beta0, float beta1, float beta2, float alpha, int train, uint64_t seed, int with_backward,
                    float* out_scalars, float* out_kl, float* out_kl2, void* stream) {
  if (!c || !x || !target || !eps) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  int r; if ((r = check_B(c, B))) return r;
  if (M < 1 || M > c->cfg.max_members) FAIL(PU_ERR_INVALID, "M=%d outside [1, max_members=%d]", M, c->cfg.max_members);
  if (recon_kind == PU_RECON_AFCRPS && M < 2) FAIL(PU_ERR_INVALID, "M must be at least 2 to compute afCRPS but got M=%d", M);
  if (recon_kind == PU_RECON_AFCRPS && M > 32) FAIL(PU_ERR_INVALID, "M=%d > 32 unsupported by the fused afCRPS kernel", M);
  if (recon_kind != PU_RECON_AFCRPS && recon_kind != PU_RECON_L1 && recon_kind != PU_RECON_WMSE_MSSSIM) FAIL(PU_ERR_INVALID, "unknown recon kind %d", recon_kind);
  const bool msssim = recon_kind == PU_RECON_WMSE_MSSSIM;
  if (msssim) {
    if (c->cfg.H <= 96 || c->cfg.W <= 96) FAIL(PU_ERR_INVALID, "Image size should be larger than 96 due to the 4 downsamplings in ms-ssim (got %dx%d)", c->cfg.H, c->cfg.W);
    if (!c->ms_ws) {
      c->ms_ws_floats = msssim_ws_floats(c->cfg.max_batch, c->cfg.max_members, c->cfg.num_classes, c->cfg.H, c->cfg.W);
      if (hipMalloc(&c->ms_ws, c->ms_ws_floats * sizeof(float)) != hipSuccess) { c->ms_ws = nullptr; FAIL(PU_ERR_NOMEM, "hipMalloc(MS-SSIM workspace, %zu bytes)", c->ms_ws_floats * sizeof(float)); }
    }
  }
  if (with_backward && !c->grads) FAIL(PU_ERR_STATE, "no gradient buffer bound");
  hipStream_t s = (hipStream_t)stream;
  if ((r = ensure_packed(c, s))) return r;
  const pu_config& cf = c->cfg;
  const long HW = (long)cf.H * cf.W; const int L = cf.latent_dim, Co = cf.num_classes, ci = cf.input_channels;
  const bool l1 = recon_kind == PU_RECON_L1;
  const int Mf = l1 ? 1 : M;
  // static loss scale (see pu_config.grad_scale)
  float S = cf.grad_scale > 0.f ? cf.grad_scale : 1.f;
  if (cf.grad_scale <= 0.f && c->dt == PU_F16) {
    const double norm = l1 ? 1.0 / ((double)B * Co * HW) : 1.0 / ((double)B * M * (M - 1) * Co * HW);
    double want = 1.0 / (2.0 * fmax(fabs((double)beta0), 1e-6) * norm * (l1 ? 1.0 : (double)(M - 1)));
    if (msssim) want = 1.0 / (32.0 * fmax(fabs((double)beta0), 1e-6)) * (double)B * M * Co * HW;   // SSIM gradients: ~16x the L1 headroom
    int e2 = (int)floor(log2(fmax(want, 1.0))); if (e2 > 24) e2 = 24;
    S = (float)ldexp(1.0, e2);
  }
  c->inv_scale = 1.f / S;
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    int q;
    CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, nullptr, 0, with_b(c->x_in.v, B), s));
    CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, target, Co, with_b(c->xy_in.v, B), s));
    // host enqueue order matters (~5 us per launch): the critical-path U-Net goes first, the latent encoders are enqueued on
    // the second side stream afterwards and still finish long before the U-Net does
    hipStream_t s2;
    if ((q = fork2(c, s, &s2))) return q;
    if ((q = unet_forward<T>(c, B, train, seed, s))) return q;
    if ((q = gauss_forward<T>(c, c->prior, B, s2))) return q;
    if ((q = gauss_forward<T>(c, c->post, B, s2))) return q;
    if ((q = join2(c, s, s2))) return q;
    CKH(hipMemsetAsync(c->scal, 0, PU_NUM_SCALARS * sizeof(float), s));
    LatentArgs la; memset(&la, 0, sizeof la);
    la.mu_q = c->post.mu; la.ls_q = c->post.ls; la.mu_p = c->prior.mu; la.ls_p = c->prior.ls; la.eps = eps; la.z = c->z;
    la.kl = c->kl; la.kl2 = c->kl2; la.scalars = c->scal; la.B = B; la.L = L; la.M = Mf;
    CKH(launch_latent_fwd(la, s));
    FcombArgs fa = fcomb_args(c, with_b(c->feat.v, B), 0, c->z, B, Mf, c->preds);
    CKH(launch_fcomb_fwd<T>(fa, s));
    if (msssim) {
      MsssimArgs ma; memset(&ma, 0, sizeof ma);
      ma.pred = c->preds; ma.target = target; ma.B = B; ma.M = Mf; ma.C = Co; ma.H = cf.H; ma.W = cf.W;
      ma.alpha_w = c->wm_alpha; ma.beta_w = c->wm_beta; ma.lam_w = c->wm_lam; ma.data_range = c->wm_range; ma.data_range_dev = c->wm_range_dev; ma.gscale = beta0 * S;
      ma.ws = c->ms_ws; ma.ws_floats = c->ms_ws_floats; ma.scalars = c->scal; ma.dpred = with_backward ? c->dpreds : nullptr;
      CKH(launch_wmse_msssim(ma, s));
    } else
    CKH(launch_recon(recon_kind, c->preds, target, with_backward ? c->dpreds : nullptr, c->scal, B, Mf, Co, HW, alpha, beta0 * S, s));
    CKH(launch_finish_scalars(c->scal, beta0, beta1, beta2, l1 ? 1 : 0, s));
    if (out_scalars) CKH(hipMemcpyAsync(out_scalars, c->scal, PU_NUM_SCALARS * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_kl) CKH(hipMemcpyAsync(out_kl, c->kl, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_kl2) CKH(hipMemcpyAsync(out_kl2, c->kl2, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!with_backward) return PU_OK;
    // ---------------- backward
    static const bool full_memset = getenv("PU_GRAD_MEMSET") != nullptr;      // diagnostic: the round-2 behaviour (zero everything, accumulate)
    struct OverwriteScope { pu_ctx* c; ~OverwriteScope() { c->grad_overwrite = false; } } ow_scope{c};
    if (sizeof(T) == 2 && c->zero_ranges_dev && !full_memset) {
      CKH(launch_zero_ranges(c->grads, c->zero_ranges_dev, c->n_zero_ranges, s));
      c->grad_overwrite = true;                    // every convolution weight / bias is written exactly once below
    } else CKH(hipMemsetAsync(c->grads, 0, (size_t)c->nparams * sizeof(float), s));
    FcombBwdArgs fb; memset(&fb, 0, sizeof fb);
    fb.f = fa; fb.dout = c->dpreds; fb.dfeat = with_b(c->feat.g, B); fb.dfeat_accumulate = 0; fb.dz = c->dz;
    fb.dw0 = G(c, c->fc_w0); fb.db0 = G(c, c->fc_b0); fb.dw1 = G(c, c->fc_w1); fb.db1 = G(c, c->fc_b1); fb.dw2 = G(c, c->fc_w2); fb.db2 = G(c, c->fc_b2);
    fb.inv_scale = c->inv_scale;
    CKH(launch_fcomb_bwd<T>(fb, s));
    LatentBwdArgs lb; memset(&lb, 0, sizeof lb);
    lb.f = la; lb.dz = c->dz; lb.beta1 = beta1 * S; lb.beta2 = l1 ? beta2 * S : 0.f;
    lb.dmu_q = c->post.dmu; lb.dls_q = c->post.dls; lb.dmu_p = c->prior.dmu; lb.dls_p = c->prior.dls;
    CKH(launch_latent_bwd(lb, s));
    // host enqueue order: the latent encoders' backward goes to the second side stream once the first U-Net decoder blocks
    // are queued (the host runs only ~1 ms ahead of the GPU here: enqueued last, they would start when the U-Net chain ends
    // and leave a ~4 ms tail; enqueued first, their ~100 launches would delay the critical chain)
    if ((q = fork2(c, s, &s2))) return q;
    if ((q = unet_backward<T>(c, s, [&]() -> int {
          int e;
          if ((e = gauss_backward<T>(c, c->post, s2))) return e;
          if ((e = gauss_backward<T>(c, c->prior, s2))) return e;
          // latent encoders + Fcomb: complete once s2 and the weight gradients enqueued so far have run
          return c->buckets.empty() ? PU_OK : record_bucket(c, (int)c->buckets.size() - 2, s, s2);
        }))) return q;
    if ((q = join2(c, s, s2))) return q;
    if ((q = join_side(c, s))) return q;
    if (!c->buckets.empty() && (q = record_bucket(c, (int)c->buckets.size() - 1, s))) return q;
    if (c->dt == PU_F16 && c->buckets.empty()) {
      // fp16 activations can overflow (diverging training, a loss scale set too high): flag non-finite parameter gradients so
      // that the optimizer step can be skipped on the device (pu_adamw_step_guarded); one read of the flat buffer, ~60 us
      CKH(launch_nonfinite_flag(c->grads, (long)c->nparams, c->scal + PU_S_NONFINITE, s));
      if (out_scalars) CKH(hipMemcpyAsync(out_scalars + PU_S_NONFINITE, c->scal + PU_S_NONFINITE, sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return PU_OK;
  });
}

__global__ void export_mu_sigma_kernel(const float* mu, const float* ls, float* omu, float* osig, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { if (omu) omu[i] = mu[i]; if (osig) osig[i] = expf(ls[i]) + 1e-7f; }
}

int pu_last_latent(pu_ctx* c, int which, float* mu, float* sigma, int B, void* stream) {
  if (!c || (!mu && !sigma)) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  GaussNet& g = which == PU_POSTERIOR ? c->post : c->prior;
  if (g.lastB <= 0) FAIL(PU_ERR_STATE, "no forward of that latent encoder yet");
  if (B != g.lastB) FAIL(PU_ERR_INVALID, "B=%d but the last forward of that encoder had %d samples", B, g.lastB);
  const int n = B * c->cfg.latent_dim;
  hipLaunchKernelGGL(export_mu_sigma_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, g.mu, g.ls, mu, sigma, n);
  return hipGetLastError() == hipSuccess ? PU_OK : PU_ERR_HIP;
}

static int sample_body(pu_ctx* c, const float* x, const float* target, const float* eps, int B, int n, float* out, float* mu, float* sigma,
                       const float* lrinterp, const float* resid_std, float epsilon, int softplus, float softplus_c, hipStream_t s) {
  const pu_config& cf = c->cfg;
  const long HW = (long)cf.H * cf.W; const int L = cf.latent_dim, ci = cf.input_channels;
  return dispatch(c, [&](auto t) -> int {
    typedef decltype(t) T;
    int q;
    CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, nullptr, 0, with_b(c->x_in.v, B), s));
    if (target) CKH(launch_nchw_to_nhwc<T>(x, (long)ci * HW, ci, target, cf.num_classes, with_b(c->xy_in.v, B), s));
    if ((q = unet_forward<T>(c, B, 0, 0, s))) return q;
    GaussNet& g = target ? c->post : c->prior;
    if ((q = gauss_forward<T>(c, g, B, s))) return q;
    LatentArgs la; memset(&la, 0, sizeof la);
    la.mu_q = g.mu; la.ls_q = g.ls; la.eps = eps; la.z = c->z; la.B = B; la.L = L; la.M = n;
    CKH(launch_latent_fwd(la, s));
    FcombArgs fa = fcomb_args(c, with_b(c->feat.v, B), 0, c->z, B, n, out);
    fa.hr_base = lrinterp; fa.hr_std = resid_std; fa.hr_eps = epsilon; fa.hr_softplus = softplus; fa.hr_softplus_c = softplus_c;
    CKH(launch_fcomb_fwd<T>(fa, s));
    if (mu || sigma) hipLaunchKernelGGL(export_mu_sigma_kernel, dim3(cdiv((long)B * L, 256)), dim3(256), 0, s, g.mu, g.ls, mu, sigma, B * L);
    return PU_OK;
  });
}

static int sample_impl(pu_ctx* c, const float* x, const float* target, const float* eps, int B, int n, float* out, float* mu, float* sigma,
                       const float* lrinterp, const float* resid_std, float epsilon, int softplus, float softplus_c, void* stream) {
  if (!c || !x || !eps || !out) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  int r; if ((r = check_B(c, B))) return r;
  if (n < 1 || n > c->cfg.max_members) FAIL(PU_ERR_INVALID, "n=%d outside [1, max_members=%d]", n, c->cfg.max_members);
  hipStream_t s = (hipStream_t)stream;
  if ((r = ensure_packed(c, s))) return r;                 // eager, never inside a capture (the weights may have changed)
  auto eager = [&]() { if (c->sample_graph_on) ++c->graph_eager; return sample_body(c, x, target, eps, B, n, out, mu, sigma, lrinterp, resid_std, epsilon, softplus, softplus_c, s); };
  if (!c->sample_graph_on || !c->side2 || prof_enabled()) return eager();
  // ---- hipGraph path: the first call with a given argument tuple runs eagerly (it also performs the one-time
  // hipFuncSetAttribute calls, which are illegal inside a capture), the second captures, later ones replay
  union { float f; uintptr_t u; } e0, e1; e0.u = 0; e1.u = 0; e0.f = epsilon; e1.f = softplus_c;
  std::vector<const void*> key = {x, target, eps, out, mu, sigma, lrinterp, resid_std, (const void*)(uintptr_t)B, (const void*)(uintptr_t)n,
                                  (const void*)(uintptr_t)softplus, (const void*)e0.u, (const void*)e1.u};
  for (auto& g : c->sample_graphs) if (g.key == key) { CKH(hipGraphLaunch(g.exec, s)); ++c->graph_replays; c->unet_B = B; return PU_OK; }
  bool seen = false;
  for (auto& k : c->sample_seen) if (k == key) { seen = true; break; }
  if (!seen) { if (c->sample_seen.size() >= 64) c->sample_seen.clear(); c->sample_seen.push_back(key); return eager(); }
  // capture on the (otherwise idle, non-blocking) second side stream: torch's current stream is often the legacy default
  // stream, which cannot be captured; the instantiated graph is launched on the caller's stream
  hipStream_t cs = c->side2;
  if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return eager(); }
  const int rb = sample_body(c, x, target, eps, B, n, out, mu, sigma, lrinterp, resid_std, epsilon, softplus, softplus_c, cs);
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(cs, &graph);
  if (rb != PU_OK || ee != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return rb != PU_OK ? rb : eager(); }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) { (void)hipGraphDestroy(graph); (void)hipGetLastError(); return eager(); }
  if (c->sample_graphs.size() >= 8) {                       // bounded cache: drop the oldest
    auto& o = c->sample_graphs.front();
    (void)hipGraphExecDestroy(o.exec); (void)hipGraphDestroy(o.graph);
    c->sample_graphs.erase(c->sample_graphs.begin());
  }
  pu_ctx::SampleGraph sg; sg.key = key; sg.graph = graph; sg.exec = exec;
  c->sample_graphs.push_back(sg);
  ++c->graph_captures;
  CKH(hipGraphLaunch(exec, s));
  ++c->graph_replays;
  return PU_OK;
}
int pu_set_sample_graph(pu_ctx* c, int on) {
  if (!c) return PU_ERR_INVALID;
  c->sample_graph_on = on != 0;
  return PU_OK;
}
int pu_sample_graph_stats(pu_ctx* c, int64_t* captures, int64_t* replays, int64_t* eager_fallbacks) {
  if (!c) return PU_ERR_INVALID;
  if (captures) *captures = c->graph_captures;
  if (replays) *replays = c->graph_replays;
  if (eager_fallbacks) *eager_fallbacks = c->graph_eager;
  return PU_OK;
}
int pu_drop_site_count(pu_ctx* c) { return c ? (int)c->drop_sites.size() : -1; }
int pu_drop_site(pu_ctx* c, int i, char name[96], int* C, int* H, int* W) {
  if (!c || i < 0 || i >= (int)c->drop_sites.size()) return PU_ERR_INVALID;
  const auto& q = c->drop_sites[i];
  if (name) snprintf(name, 96, "%s", q.name.c_str());
  if (C) *C = q.C; if (H) *H = q.H; if (W) *W = q.W;
  return PU_OK;
}
int pu_set_drop_masks(pu_ctx* c, const float* masks, int B, void* stream) {
  if (!c) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  if (!masks) { c->drop_masks_B = 0; return PU_OK; }        // back to the counter-hash stream (the buffer is kept for reuse)
  int r; if ((r = check_B(c, B))) return r;
  if (c->drop_sites.empty()) return PU_OK;
  const auto& last = c->drop_sites.back();
  const size_t per = last.off + (size_t)last.C * last.H * last.W, need = per * (size_t)B;
  if (need > c->drop_masks_cap) {
    if (c->drop_masks) { CKH(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(c->drop_masks); c->drop_masks = nullptr; c->drop_masks_cap = 0; }
    if (hipMalloc(&c->drop_masks, need) != hipSuccess) { c->drop_masks = nullptr; FAIL(PU_ERR_NOMEM, "hipMalloc(dropout masks, %zu bytes)", need); }
    c->drop_masks_cap = need;
  }
  for (const auto& q : c->drop_sites) {
    const size_t site = (size_t)q.C * q.H * q.W;
    CKH(launch_mask_to_nhwc_u8(masks + q.off * (size_t)B, c->drop_masks + q.off * (size_t)B, B, q.C, (long)q.H * q.W, (hipStream_t)stream));
    (void)site;
  }
  c->drop_masks_B = B;
  return PU_OK;
}
int pu_set_grad_buckets(pu_ctx* c, int n) {
  if (!c || n < 0 || n > 16) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  if (n != c->want_buckets) { c->want_buckets = n; plan_buckets(c); }
  return PU_OK;
}
int pu_grad_buckets(pu_ctx* c, int64_t* lo, int64_t* hi, int max, int* n) {
  if (!c || !n) return PU_ERR_INVALID;
  *n = (int)c->buckets.size();
  for (int k = 0; k < *n && k < max; ++k) { if (lo) lo[k] = c->buckets[k].lo; if (hi) hi[k] = c->buckets[k].hi; }
  return PU_OK;
}
int pu_grad_bucket_wait(pu_ctx* c, int k, void* stream) {
  if (!c || k < 0 || k >= (int)c->buckets.size()) return PU_ERR_INVALID;
  DeviceGuard dg(c->device);
  const auto& q = c->buckets[k];
  if (!q.rec[0]) FAIL(PU_ERR_STATE, "bucket %d has not been recorded: call pu_elbo_fwd_bwd(with_backward) first", k);
  for (int i = 0; i < 3; ++i) if (q.rec[i]) CKH(hipStreamWaitEvent((hipStream_t)stream, q.ev[i], 0));
  return PU_OK;
}
int pu_scale_grads(float* g, int64_t n, const float* scale_dev, float host_factor, void* stream) {
  if (!g || n < 0) return PU_ERR_INVALID;
  return launch_scale_grads(g, (long)n, scale_dev, host_factor, (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_nonfinite_flag(const float* g, int64_t n, float* flag, void* stream) {
  if (!g || !flag || n < 0) return PU_ERR_INVALID;
  return launch_nonfinite_flag(g, (long)n, flag, (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_adamw_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float* state, const float* skip_flag, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state || n < 0) return PU_ERR_INVALID;
  return launch_adamw_flat_dev(params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, state, skip_flag,
                               (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_adamw_prepare(float* state, const float* skip_flag, float lr, float beta1, float beta2, void* stream) {
  if (!state) return PU_ERR_INVALID;
  return launch_adamw_flat_dev(nullptr, nullptr, nullptr, nullptr, 0, lr, beta1, beta2, 0.f, 0.f, state, skip_flag, (hipStream_t)stream, 1) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_adamw_apply(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const float* state, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state || n < 0) return PU_ERR_INVALID;
  return launch_adamw_flat_dev(params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, const_cast<float*>(state), nullptr,
                               (hipStream_t)stream, 2) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_softplus_transform(float* data, int64_t n, int inverse, float threshold, float cc, void* stream) {
  if (!data || n < 0) return PU_ERR_INVALID;
  return launch_softplus_transform(data, (long)n, inverse, threshold, cc, (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_destandardize(const float* x, const float* base, const float* std_hr, const float* mean_hr, float epsilon, int B, int n, int C, int H, int W,
                     float* out, void* stream) {
  if (!x || !std_hr || !out || B < 1 || n < 1 || C < 1 || H < 1 || W < 1) return PU_ERR_INVALID;
  return launch_destandardize(x, base, std_hr, mean_hr, epsilon, B, n, (long)C * H * W, out, (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_set_recon_wmse_msssim(pu_ctx* c, float alpha_w, float beta_w, float lam_w, float data_range) {
  if (!c) return PU_ERR_INVALID;
  c->wm_alpha = alpha_w; c->wm_beta = beta_w; c->wm_lam = lam_w; c->wm_range = data_range;
  return PU_OK;
}
int pu_set_recon_range_dev(pu_ctx* c, const float* range_dev) {
  if (!c) return PU_ERR_INVALID;
  c->wm_range_dev = range_dev;
  return PU_OK;
}
int pu_lr_stats(const float* hr, int N, int C, int H, int W, int k, float* mean_lr, float* std_lr, float* mean_hr, float* std_hr, void* stream) {
  if (!hr || N < 1 || C < 1 || k < 1 || H % k || W % k) return PU_ERR_INVALID;
  return launch_lr_stats(hr, N, C, H, W, k, mean_lr, std_lr, mean_hr, std_hr, (hipStream_t)stream) == hipSuccess ? PU_OK : PU_ERR_HIP;
}
int pu_lrinterp_to_residuals(const float* hr, int B, int C, int H, int W, int k, const float* mean_hr, const float* std_hr, float epsilon,
                             float* inputs, float* targets, float* lrinterp, float* lr, void* stream) {
  if (!hr || !mean_hr || !std_hr || !inputs || !targets || B < 1 || C < 1 || k < 1 || H % k || W % k) return PU_ERR_INVALID;
  return launch_lrinterp_residuals(hr, B, C, H, W, k, mean_hr, std_hr, epsilon, inputs, targets, lrinterp, lr, (hipStream_t)stream) == hipSuccess
             ? PU_OK : PU_ERR_HIP;
}
int pu_op_wmse_msssim(const float* pred, const float* target, int B, int M, int C, int H, int W, float alpha_w, float beta_w, float lam_w,
                      float data_range, float gscale, float* out_scalars, float* dpred, void* stream) {
  if (!pred || !target || !out_scalars || B < 1 || M < 1 || C < 1 || H <= 96 || W <= 96) return PU_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  MsssimArgs ma; memset(&ma, 0, sizeof ma);
  ma.pred = pred; ma.target = target; ma.B = B; ma.M = M; ma.C = C; ma.H = H; ma.W = W;
  ma.alpha_w = alpha_w; ma.beta_w = beta_w; ma.lam_w = lam_w; ma.data_range = data_range; ma.gscale = gscale;
  ma.ws_floats = msssim_ws_floats(B, M, C, H, W); ma.scalars = out_scalars; ma.dpred = dpred;
  if (hipMalloc(&ma.ws, ma.ws_floats * sizeof(float)) != hipSuccess) return PU_ERR_NOMEM;
  int rc = PU_OK;
  if (hipMemsetAsync(out_scalars, 0, PU_NUM_SCALARS * sizeof(float), s) != hipSuccess) rc = PU_ERR_HIP;
  if (rc == PU_OK && launch_wmse_msssim(ma, s) != hipSuccess) rc = PU_ERR_HIP;
  if (hipStreamSynchronize(s) != hipSuccess) rc = PU_ERR_HIP;
  (void)hipFree(ma.ws);
  return rc;
}
int pu_sample(pu_ctx* c, const float* x, const float* target, const float* eps, int B, int n, float* out, float* mu, float* sigma, void* stream) {
  return sample_impl(c, x, target, eps, B, n, out, mu, sigma, nullptr, nullptr, 0.f, 0, 0.f, stream);
}
int pu_sample_hr(pu_ctx* c, const float* x, const float* target, const float* eps, int B, int n, const float* lrinterp, const float* resid_std,
                 float epsilon, int softplus, float softplus_c, float* out, float* mu, float* sigma, void* stream) {
  if (!lrinterp || !resid_std) return PU_ERR_INVALID;
  return sample_impl(c, x, target, eps, B, n, out, mu, sigma, lrinterp, resid_std, epsilon, softplus, softplus_c, stream);
}

}  // extern "C"

// ------------------------------------------------------------------ single-op test entry points
#define CK0(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { g_create_err = std::string(#expr) + ": " + hipGetErrorString(_e); rc = PU_ERR_HIP; goto done; } } while (0)

template <typename T>
static int op_conv_t(int mode, int ks, int relu, int B, int Cin, int Cout, int H, int W, const float* x, const float* w, const float* bias,
                     const float* dy, float* out, hipStream_t s) {
  int rc = PU_OK;
  const size_t esz = sizeof(T); const int taps = ks * ks;
  const int cin_a = rup(Cin, 8), cout_a = rup(Cout, 8);
  T *xin = nullptr, *yb = nullptr, *wp = nullptr; PackDesc* dd = nullptr; float* dwtmp = nullptr;
  const long npix = (long)B * H * W;
  PackDesc d; d.src_off = 0; d.dst_off = 0; d.Cout = Cout; d.Cin = Cin; d.taps = taps;
  TV tx, ty;
  CK0(hipMalloc(&xin, npix * cin_a * esz)); CK0(hipMalloc(&yb, npix * cout_a * esz));
  CK0(hipMalloc(&dd, sizeof(PackDesc)));
  tx.p = xin; tx.B = B; tx.H = H; tx.W = W; tx.C = cin_a; tx.ld = cin_a;
  ty.p = yb; ty.B = B; ty.H = H; ty.W = W; ty.C = cout_a; ty.ld = cout_a;
  if (mode == 0) {
    d.rows_pk = rup(Cout, 32); d.k_pk = rup(Cin, 32); d.mode = conv_uses_frag_layout((int)esz, H, W) ? 2 : 0;
    const bool m16 = d.mode == 2 && conv_uses_mfma16((int)esz, taps, d.k_pk, Cout, H, W);
    if (m16) d.mode = 6;
    CK0(hipMalloc(&wp, (size_t)d.rows_pk * taps * d.k_pk * esz));
    CK0(hipMemcpyAsync(dd, &d, sizeof d, hipMemcpyHostToDevice, s));
    CK0(launch_pack<T>(w, wp, dd, 1, s));
    CK0(launch_nchw_to_nhwc<T>(x, (long)Cin * H * W, Cin, nullptr, 0, tx, s));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = xin; a.in_ld = cin_a; a.Cin = cin_a; a.wpk = wp; a.cin_pk = d.k_pk; a.cout_pk = d.rows_pk; a.taps = taps; a.bias = bias;
    a.out = yb; a.out_ld = cout_a; a.Cout = Cout; a.B = B; a.H = H; a.W = W; a.relu = relu; a.frag_layout = d.mode >= 2; a.mfma16 = m16;
    CK0(hipMemsetAsync(yb, 0, npix * cout_a * esz, s));
    CK0(launch_conv<T>(a, s));
    CK0(launch_nhwc_to_nchw<T>(ty, Cout, out, 0, s));
  } else if (mode == 1) {
    d.rows_pk = rup(Cin, 32); d.k_pk = rup(Cout, 32); d.mode = conv_uses_frag_layout((int)esz, H, W) ? 3 : 1;
    const bool m16 = d.mode == 3 && conv_uses_mfma16((int)esz, taps, d.k_pk, cin_a, H, W);
    if (m16) d.mode = 7;
    CK0(hipMalloc(&wp, (size_t)d.rows_pk * taps * d.k_pk * esz));
    CK0(hipMemcpyAsync(dd, &d, sizeof d, hipMemcpyHostToDevice, s));
    CK0(launch_pack<T>(w, wp, dd, 1, s));
    CK0(launch_nchw_to_nhwc<T>(dy, (long)Cout * H * W, Cout, nullptr, 0, ty, s));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = yb; a.in_ld = cout_a; a.Cin = cout_a; a.wpk = wp; a.cin_pk = d.k_pk; a.cout_pk = d.rows_pk; a.taps = taps;
    a.out = xin; a.out_ld = cin_a; a.Cout = cin_a; a.B = B; a.H = H; a.W = W; a.frag_layout = d.mode >= 2; a.mfma16 = m16;
    CK0(launch_conv<T>(a, s));
    CK0(launch_nhwc_to_nchw<T>(tx, Cin, out, 0, s));
  } else {
    CK0(launch_nchw_to_nhwc<T>(x, (long)Cin * H * W, Cin, nullptr, 0, tx, s));
    CK0(launch_nchw_to_nhwc<T>(dy, (long)Cout * H * W, Cout, nullptr, 0, ty, s));
    CK0(hipMemsetAsync(out, 0, (size_t)Cout * Cin * taps * sizeof(float), s));
    WgradArgs a; memset(&a, 0, sizeof a);
    a.dy = yb; a.dy_ld = cout_a; a.Cout = Cout; a.in = xin; a.in_ld = cin_a; a.Cin = Cin; a.dw = out; a.B = B; a.H = H; a.W = W; a.taps = taps;
    a.inv_scale = 1.f;
    if (sizeof(T) == 2) { a.slab_floats = 16L * 1024 * 1024; CK0(hipMalloc(&dwtmp, a.slab_floats * sizeof(float))); a.slab = dwtmp; }
    CK0(launch_wgrad<T>(a, s));
  }
  CK0(hipStreamSynchronize(s));
done:
  (void)hipFree(xin); (void)hipFree(yb); (void)hipFree(wp); (void)hipFree(dd); (void)hipFree(dwtmp);
  return rc;
}

extern "C" int pu_op_conv(int dtype, int mode, int ks, int relu, int B, int Cin, int Cout, int H, int W, const float* x, const float* w,
               const float* bias, const float* dy, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((ks != 1 && ks != 3) || mode < 0 || mode > 2) return PU_ERR_INVALID;
  if (dtype == PU_F32) return op_conv_t<float>(mode, ks, relu, B, Cin, Cout, H, W, x, w, bias, dy, out, s);
  if (dtype == PU_F16) return op_conv_t<f16>(mode, ks, relu, B, Cin, Cout, H, W, x, w, bias, dy, out, s);
  if (dtype == PU_BF16) return op_conv_t<bf16>(mode, ks, relu, B, Cin, Cout, H, W, x, w, bias, dy, out, s);
  return PU_ERR_INVALID;
}

template <typename T>
static int op_gn_t(int resample, int B, int C, int H, int W, const float* x, const float* gamma, const float* beta, const float* ss, float* y,
                   const float* dy, float* dx, float* dgamma, float* dbeta, float* dss, float drop_p, uint64_t drop_seed, hipStream_t s) {
  int rc = PU_OK;
  const size_t esz = sizeof(T);
  const int OH = resample == RS_DOWN ? H / 2 : (resample == RS_UP ? H * 2 : H), OW = resample == RS_DOWN ? W / 2 : (resample == RS_UP ? W * 2 : W);
  const long nin = (long)B * H * W * C, nout = (long)B * OH * OW * C;
  T *xb = nullptr, *yb = nullptr, *dyb = nullptr, *dxb = nullptr, *dvb = nullptr; float* ws = nullptr;
  const int G = gn_groups(C), nchunk = gn_chunks((long)H * W);
  const size_t nws = (size_t)B * nchunk * C * 2 * 2 + (size_t)B * G * 2 + 4 + (size_t)B * C * 4 + (size_t)B * C * 3;
  GNArgs a; GNBwdArgs bw; TV tx, ty;
  CK0(hipMalloc(&xb, nin * esz)); CK0(hipMalloc(&yb, nout * esz)); CK0(hipMalloc(&dyb, nout * esz)); CK0(hipMalloc(&dxb, nin * esz));
  CK0(hipMalloc(&dvb, nin * esz)); CK0(hipMalloc(&ws, nws * sizeof(float)));
  tx.p = xb; tx.B = B; tx.H = H; tx.W = W; tx.C = C; tx.ld = C;
  ty.p = yb; ty.B = B; ty.H = OH; ty.W = OW; ty.C = C; ty.ld = C;
  memset(&a, 0, sizeof a);
  a.x = tx; a.y = ty; a.G = G; a.eps = 1e-5f; a.gamma = gamma; a.beta = beta; a.scale = ss; a.shift = ss ? ss + C : nullptr; a.resample = resample;
  a.drop_p = resample == RS_NONE ? drop_p : 0.f; a.drop_seed = drop_seed; a.drop_stream = 7;
  a.part = ws; a.nchunk = nchunk; a.stat = ws + (size_t)B * nchunk * C * 2; a.coef = a.stat + (((size_t)B * G * 2 + 3) & ~(size_t)3);
  CK0(launch_nchw_to_nhwc<T>(x, (long)C * H * W, C, nullptr, 0, tx, s));
  CK0(launch_gn_fwd<T>(a, s));
  CK0(launch_nhwc_to_nchw<T>(ty, C, y, 0, s));
  if (dy) {
    TV tdy = ty; tdy.p = dyb; TV tdx = tx; tdx.p = dxb; TV tdv = tx; tdv.p = dvb;
    CK0(launch_nchw_to_nhwc<T>(dy, (long)C * OH * OW, C, nullptr, 0, tdy, s));
    memset(&bw, 0, sizeof bw);
    bw.f = a; bw.dy = tdy; bw.dv = tdv; bw.dx = tdx; bw.accumulate = 0;
    bw.dgamma = dgamma; bw.dbeta = dbeta; bw.dscale = ss ? dss : nullptr; bw.dshift = ss ? dss + C : nullptr;
    bw.part2 = a.coef + (size_t)B * C * 4; bw.coef2 = bw.part2 + (size_t)B * nchunk * C * 2; bw.inv_scale = 1.f;
    CK0(hipMemsetAsync(dgamma, 0, C * sizeof(float), s)); CK0(hipMemsetAsync(dbeta, 0, C * sizeof(float), s));
    if (ss) CK0(hipMemsetAsync(dss, 0, 2 * C * sizeof(float), s));
    CK0(launch_gn_bwd<T>(bw, s));
    CK0(launch_nhwc_to_nchw<T>(tdx, C, dx, 0, s));
  }
  CK0(hipStreamSynchronize(s));
done:
  (void)hipFree(xb); (void)hipFree(yb); (void)hipFree(dyb); (void)hipFree(dxb); (void)hipFree(dvb); (void)hipFree(ws);
  return rc;
}

// uniform 16-bit random values in [-1, 1) (bench on random data: zero / constant operands run at a higher clock, MI355X_MICROARCH.md DVFS)
template <typename T>
__global__ void rand_fill_kernel(T* p, long n, uint32_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    ET<T>::st(p + i, (float)(int)(hash32((uint32_t)i * 2654435761u + seed) >> 8) * (1.0f / 8388608.0f) - 1.0f);
}
// micro-benchmark of one convolution launch on random NHWC data (no layout conversion in the timed region)
template <typename T>
static int bench_conv_t(int mode, int ks, int B, int Cin, int Cout, int H, int W, int iters, float* out_us, hipStream_t s) {
  int rc = PU_OK;
  const size_t esz = sizeof(T); const int taps = ks * ks;
  const long npix = (long)B * H * W;
  T *xin = nullptr, *yb = nullptr, *wp = nullptr; float *wf = nullptr, *slab = nullptr, *dw = nullptr; PackDesc* dd = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr; float ms = 0.f;
  PackDesc d; d.src_off = 0; d.dst_off = 0; d.Cout = Cout; d.Cin = Cin; d.taps = taps;
  const long nw = (long)Cout * Cin * taps;
  CK0(hipMalloc(&xin, npix * Cin * esz)); CK0(hipMalloc(&yb, npix * Cout * esz)); CK0(hipMalloc(&wf, nw * 4)); CK0(hipMalloc(&dw, nw * 4));
  CK0(hipMalloc(&dd, sizeof(PackDesc))); CK0(hipMalloc(&slab, 32L * 1024 * 1024 * 4));
  hipLaunchKernelGGL(rand_fill_kernel<float>, dim3(1024), dim3(256), 0, s, wf, nw, 11u);
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, xin, npix * Cin, 22u);
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, yb, npix * Cout, 33u);
  CK0(hipGetLastError());
  CK0(hipEventCreate(&e0)); CK0(hipEventCreate(&e1));
  if (mode == 0 || mode == 1) {
    const int ci = mode == 0 ? Cin : Cout, co = mode == 0 ? Cout : Cin;
    d.Cout = co; d.Cin = ci; d.rows_pk = rup(co, 32); d.k_pk = rup(ci, 32); d.mode = conv_uses_frag_layout((int)esz, H, W) ? 2 : 0;
    const bool m16 = d.mode == 2 && conv_uses_mfma16((int)esz, taps, d.k_pk, co, H, W);
    if (m16) d.mode = 6;
    CK0(hipMalloc(&wp, (size_t)d.rows_pk * taps * d.k_pk * esz));
    CK0(hipMemcpyAsync(dd, &d, sizeof d, hipMemcpyHostToDevice, s));
    CK0(launch_pack<T>(wf, wp, dd, 1, s));
    ConvArgs a; memset(&a, 0, sizeof a);
    a.in = mode == 0 ? (void*)xin : (void*)yb; a.in_ld = ci; a.Cin = ci; a.wpk = wp; a.cin_pk = d.k_pk; a.cout_pk = d.rows_pk; a.taps = taps;
    a.out = mode == 0 ? (void*)yb : (void*)xin; a.out_ld = co; a.Cout = co; a.B = B; a.H = H; a.W = W; a.frag_layout = d.mode >= 2; a.mfma16 = m16;
    for (int i = 0; i < 3; ++i) CK0(launch_conv<T>(a, s));
    CK0(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK0(launch_conv<T>(a, s));
    CK0(hipEventRecord(e1, s));
  } else {
    WgradArgs a; memset(&a, 0, sizeof a);
    a.dy = yb; a.dy_ld = Cout; a.Cout = Cout; a.in = xin; a.in_ld = Cin; a.Cin = Cin; a.dw = dw; a.B = B; a.H = H; a.W = W; a.taps = taps;
    a.slab = slab; a.slab_floats = 32L * 1024 * 1024; a.inv_scale = 1.f;
    for (int i = 0; i < 3; ++i) CK0(launch_wgrad<T>(a, s));
    CK0(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK0(launch_wgrad<T>(a, s));
    CK0(hipEventRecord(e1, s));
  }
  CK0(hipEventSynchronize(e1)); CK0(hipEventElapsedTime(&ms, e0, e1));
  *out_us = 1e3f * ms / iters;
done:
  (void)hipFree(xin); (void)hipFree(yb); (void)hipFree(wp); (void)hipFree(wf); (void)hipFree(dd); (void)hipFree(slab); (void)hipFree(dw);
  if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
  return rc;
}
extern "C" int pu_bench_conv(int dtype, int mode, int ks, int B, int Cin, int Cout, int H, int W, int iters, float* out_us, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PU_F16) return bench_conv_t<f16>(mode, ks, B, Cin, Cout, H, W, iters, out_us, s);
  if (dtype == PU_BF16) return bench_conv_t<bf16>(mode, ks, B, Cin, Cout, H, W, iters, out_us, s);
  if (dtype == PU_F32) return bench_conv_t<float>(mode, ks, B, Cin, Cout, H, W, iters, out_us, s);
  return PU_ERR_INVALID;
}

// micro-benchmark of the GroupNorm kernels on random NHWC data: out_us[0] apply, [1] backward pass 1, [2] backward pass 2,
// [3] single-kernel small-tensor backward (0 where it does not apply); flags: 1 = accumulate into dx, 2 = extra addend
template <typename T>
static int bench_gn_t(int resample, int B, int C, int H, int W, float drop_p, int flags, int iters, float* out_us, hipStream_t s) {
  int rc = PU_OK;
  const size_t esz = sizeof(T);
  const int OH = resample == RS_DOWN ? H / 2 : (resample == RS_UP ? H * 2 : H), OW = resample == RS_DOWN ? W / 2 : (resample == RS_UP ? W * 2 : W);
  const long nin = (long)B * H * W * C, nout = (long)B * OH * OW * C;
  T *xb = nullptr, *yb = nullptr, *dyb = nullptr, *dxb = nullptr, *adb = nullptr; float *ws = nullptr, *par = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr; float ms = 0.f;
  const int G = gn_groups(C), nchunk = gn_chunks((long)H * W);
  const size_t nws = (size_t)B * nchunk * C * 2 * 2 + (size_t)B * G * 2 + 4 + (size_t)B * C * 4 + (size_t)B * C * 3;
  GNArgs a; GNBwdArgs bw; TV tx, ty;
  CK0(hipMalloc(&xb, nin * esz)); CK0(hipMalloc(&yb, nout * esz)); CK0(hipMalloc(&dyb, nout * esz)); CK0(hipMalloc(&dxb, nin * esz));
  CK0(hipMalloc(&adb, nin * esz)); CK0(hipMalloc(&ws, nws * sizeof(float))); CK0(hipMalloc(&par, (size_t)C * 8 * sizeof(float)));
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, xb, nin, 5u);
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, dyb, nout, 6u);
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, adb, nin, 7u);
  hipLaunchKernelGGL(rand_fill_kernel<T>, dim3(4096), dim3(256), 0, s, dxb, nin, 8u);
  hipLaunchKernelGGL(rand_fill_kernel<float>, dim3(8), dim3(256), 0, s, par, (long)C * 8, 9u);
  CK0(hipGetLastError());
  CK0(hipEventCreate(&e0)); CK0(hipEventCreate(&e1));
  tx.p = xb; tx.B = B; tx.H = H; tx.W = W; tx.C = C; tx.ld = C;
  ty.p = yb; ty.B = B; ty.H = OH; ty.W = OW; ty.C = C; ty.ld = C;
  memset(&a, 0, sizeof a);
  a.x = tx; a.y = ty; a.G = G; a.eps = 1e-5f; a.gamma = par; a.beta = par + C; a.resample = resample;
  a.drop_p = resample == RS_NONE ? drop_p : 0.f; a.drop_seed = 1234; a.drop_stream = 7;
  a.part = ws; a.nchunk = nchunk; a.stat = ws + (size_t)B * nchunk * C * 2; a.coef = a.stat + (((size_t)B * G * 2 + 3) & ~(size_t)3);
  CK0(launch_gn_fwd<T>(a, s));
  {
    TV tdy = ty; tdy.p = dyb; TV tdx = tx; tdx.p = dxb; TV tad = tx; tad.p = adb;
    memset(&bw, 0, sizeof bw);
    bw.f = a; bw.dy = tdy; bw.dx = tdx; bw.accumulate = flags & 1; if (flags & 2) bw.add = tad;
    bw.dgamma = par + 2 * C; bw.dbeta = par + 3 * C;
    bw.part2 = a.coef + (size_t)B * C * 4; bw.coef2 = bw.part2 + (size_t)B * nchunk * C * 2; bw.inv_scale = 1.f;
  }
  CK0(launch_gn_bwd_parts<T>(bw, 7, s));
  for (int k = 0; k < 4; ++k) {
    out_us[k] = 0.f;
    int cb = 0;
    if (k == 3 && !(sizeof(T) == 2 && resample == RS_NONE && (long)H * W <= 1024 && C % 8 == 0)) continue;
    auto run = [&]() -> hipError_t {
      if (k == 0) return launch_gn_apply<T>(a, s);
      return launch_gn_bwd_parts<T>(bw, k == 1 ? 1 : (k == 2 ? 2 : 8), s);
    };
    (void)cb;
    if (run() != hipSuccess) { (void)hipGetLastError(); continue; }
    CK0(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK0(run());
    CK0(hipEventRecord(e1, s));
    CK0(hipEventSynchronize(e1)); CK0(hipEventElapsedTime(&ms, e0, e1));
    out_us[k] = 1e3f * ms / iters;
  }
done:
  (void)hipFree(xb); (void)hipFree(yb); (void)hipFree(dyb); (void)hipFree(dxb); (void)hipFree(adb); (void)hipFree(ws); (void)hipFree(par);
  if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
  return rc;
}
extern "C" int pu_bench_gn(int dtype, int resample, int B, int C, int H, int W, float drop_p, int flags, int iters, float* out_us, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C % 8 || iters < 1) return PU_ERR_INVALID;
  if (dtype == PU_F16) return bench_gn_t<f16>(resample, B, C, H, W, drop_p, flags, iters, out_us, s);
  if (dtype == PU_BF16) return bench_gn_t<bf16>(resample, B, C, H, W, drop_p, flags, iters, out_us, s);
  return PU_ERR_INVALID;
}

extern "C" int pu_op_gnsilu(int dtype, int resample, int B, int C, int H, int W, const float* x, const float* gamma, const float* beta,
                 const float* ss, float* y, const float* dy, float* dx, float* dgamma, float* dbeta, float* dss,
                 float drop_p, uint64_t drop_seed, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C % 8) return PU_ERR_INVALID;
  if (dtype == PU_F32) return op_gn_t<float>(resample, B, C, H, W, x, gamma, beta, ss, y, dy, dx, dgamma, dbeta, dss, drop_p, drop_seed, s);
  if (dtype == PU_F16) return op_gn_t<f16>(resample, B, C, H, W, x, gamma, beta, ss, y, dy, dx, dgamma, dbeta, dss, drop_p, drop_seed, s);
  if (dtype == PU_BF16) return op_gn_t<bf16>(resample, B, C, H, W, x, gamma, beta, ss, y, dy, dx, dgamma, dbeta, dss, drop_p, drop_seed, s);
  return PU_ERR_INVALID;
}
