// Host-side driver of the CRNN-OCR hot path: parameter layout (Keras weight order), workspace plan,
// and the forward / backward kernel chains on one HIP stream.  Mirrors the graph of utils.py:58-96
// (CRNN.get_model) + utils.py:247-258 (STN) + utils.py:98-103 (CTC Lambda); no allocation, no sync.
#include "common.h"
#include "crnn_mi355x.h"
#include <string.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace {

struct BlockSpec { int cout, ph, pw; };
const BlockSpec kBlocks[7] = {{64, 1, 1}, {128, 1, 1}, {256, 2, 2}, {256, 1, 1}, {512, 1, 2}, {512, 1, 1}, {512, 1, 1}};
// Pooled blocks (round 6): the training forward's BatchNorm-2 + ReLU6 + MaxPool pass also keeps q at each window's first maximum ("qm<i>"), and the statistics
// pass of that BatchNorm's backward reads it instead of the four (two) values of the window: the arg-max carries the window's whole gradient, so the sums are
// the same, bit for bit (crnn_bn_bwd_qmax_ex).  CRNN_FLAG_NO_POOL_ARGMAX_Q: the window scan of rounds 1-5.
bool pool_argmax_q(const crnn_config* cfg, int block) {
  const int ph = kBlocks[block - 1].ph, pw = kBlocks[block - 1].pw;
  return !(cfg->flags & CRNN_FLAG_NO_POOL_ARGMAX_Q) && ((ph == 2 && pw == 2) || (ph == 1 && pw == 2));
}
const float kDropBlock = 0.1f, kDropDense1 = 0.4f, kDropRnn = 0.2f;  // utils.py:56,75,83
const uint32_t kLayerDense1 = 8, kLayerRnn = 9;

struct Tensor { std::string name; long off; long size; int ndim; int dims[4]; int dtype = CRNN_F32; };

struct Dims {
  int B, H0, W0, Hp, Wp, T, feat, C, L, tds, u, G, stn_flat;
  int Hs1, Ws1, Ho1, Wo1, Hs2, Ws2, Ho2, Wo2;  // STN locnet maps
  int bh[8], bw[8], bc[8];                      // input map of block i (1-based), bc[0] = 1
};

Dims make_dims(const crnn_config* c) {
  Dims d;
  d.B = c->batch; d.H0 = c->imgh; d.W0 = c->imgw; d.Hp = d.H0 + 4; d.Wp = d.W0 + 4;
  d.C = c->num_classes; d.L = c->max_len; d.tds = c->tds; d.u = c->units; d.G = (c->gru ? 3 : 4) * c->units;
  d.Hs1 = d.H0 / 2; d.Ws1 = d.W0 / 2; d.Ho1 = d.Hs1 - 4; d.Wo1 = d.Ws1 - 4;
  d.Hs2 = d.Ho1 / 2; d.Ws2 = d.Wo1 / 2; d.Ho2 = d.Hs2 - 4; d.Wo2 = d.Ws2 - 4;
  d.stn_flat = d.Ho2 * d.Wo2 * 20;
  int h = d.Hp, w = d.Wp, ch = 1;
  for (int i = 1; i <= 7; ++i) {
    d.bh[i] = h; d.bw[i] = w; d.bc[i - 1] = ch;
    h /= kBlocks[i - 1].ph; w /= kBlocks[i - 1].pw; ch = kBlocks[i - 1].cout;
  }
  d.bc[7] = ch;
  d.T = h; d.feat = w * ch;
  return d;
}

long pad4(long n) { return (n + 3) & ~3L; }

struct Layout {
  std::vector<Tensor> params;
  long total = 0;
  void add(const std::string& n, std::initializer_list<int> dims) {
    Tensor t; t.name = n; t.off = total; t.ndim = (int)dims.size(); t.size = 1;
    int i = 0; for (int v : dims) { t.dims[i++] = v; t.size *= v; }
    for (; i < 4; ++i) t.dims[i] = 1;
    params.push_back(t); total += pad4(t.size);
  }
  long off(const std::string& n) const { for (auto& t : params) if (t.name == n) return t.off; return -1; }
};

Layout make_layout(const crnn_config* c) {
  Dims d = make_dims(c);
  Layout L;
  L.add("stn_c1_k", {5, 5, 1, 20}); L.add("stn_c1_b", {20});
  L.add("stn_c2_k", {5, 5, 20, 20}); L.add("stn_c2_b", {20});
  L.add("stn_d1_w", {d.stn_flat, 50}); L.add("stn_d1_b", {50});
  L.add("stn_d2_w", {50, 6}); L.add("stn_d2_b", {6});
  for (int i = 1; i <= 7; ++i) {
    std::string p = "b" + std::to_string(i);
    int ci = d.bc[i - 1], co = d.bc[i];
    L.add(p + "_dw", {3, 3, ci}); L.add(p + "_bn1_g", {ci}); L.add(p + "_bn1_b", {ci});
    L.add(p + "_pw", {ci, co}); L.add(p + "_bn2_g", {co}); L.add(p + "_bn2_b", {co});
  }
  L.add("dense1_w", {d.feat, d.tds}); L.add("dense1_b", {d.tds});
  for (int l = 1; l <= 2; ++l)
    for (const char* dir : {"f", "b"}) {
      std::string p = "rnn" + std::to_string(l) + dir;
      L.add(p + "_w", {l == 1 ? d.tds : d.u, d.G}); L.add(p + "_u", {d.u, d.G}); L.add(p + "_b", {d.G});
    }
  L.add("dense2_w", {2 * d.u, d.C}); L.add("dense2_b", {d.C});
  return L;
}

// ---------------------------------------------------------------------------------------------------
struct Plan {
  std::vector<Tensor> t;
  long total = 0;
  // `count` elements of `dtype`; offsets/total are in floats (a bf16 tensor occupies count/2 floats)
  long add(const std::string& n, long count, int dtype = CRNN_F32) {
    Tensor x; x.name = n; x.off = total; x.size = count; x.ndim = 1; x.dims[0] = x.dims[1] = x.dims[2] = x.dims[3] = 1; x.dtype = dtype;
    long floats = (dtype == CRNN_BF16) ? (count + 1) / 2 : count;
    t.push_back(x); total += (floats + 63) & ~63L;  // 256-byte aligned
    return x.off;
  }
  int dt(const std::string& n) const { for (auto& x : t) if (x.name == n) return x.dtype; return CRNN_F32; }
  long off(const std::string& n) const { for (auto& x : t) if (x.name == n) return x.off; return -1; }
  bool has(const std::string& n) const { return off(n) >= 0; }
  long cnt(const std::string& n) const { for (auto& x : t) if (x.name == n) return x.size; return -1; }
};

long lmax(long a, long b) { return a > b ? a : b; }

// dense2 (+ softmax) on the streaming GEMM over a 128-row padded bf16 W^T (forward, round 5): bf16 modes, whole 64-row stripes, 2u a multiple of 64, at most
// 64 classes (the softmax kernels' limit); CRNN_FLAG_GEMM_TILE_KERNELS keeps the tile GEMM
static bool dense2_stream(const crnn_config* cfg, const Dims& d) {
  return cfg->mfma_bf16 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && ((long)d.T * d.B) % 64 == 0 && (2 * d.u) % 64 == 0 && d.C <= 64;
}
// dense1's forward on the 64-row stripe stream (gemm_wgrad.hip, crnn_dense_fwd_stream: x7 bf16 against a bf16 W1^T, bias + ReLU + the rows to time-major +
// Dropout(.4) in the epilogue): bf16 storage mode, whole stripes; CRNN_FLAG_GEMM_TILE_KERNELS keeps the tile GEMM + dropout pass
static bool dense1_stream(const crnn_config* cfg, const Dims& d) {
  return cfg->mfma_bf16 == 2 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && d.feat % 8 == 0 &&
         crnn_dense_fwd_stream_supported((long)d.T * d.B, d.tds, d.feat) == CRNN_OK;
}
// dense1's data gradient gA [T*B][feat] = gbm [T*B][tds] . W1^T on the weights-resident GEMM (gemm_wres.hip: 36 slices of 128 features keep their
// 128 x 128 weights in registers, the 3.4 MB operand streams; bf16 storage mode: both operands and the result are bf16 there anyway -- the same
// products as the tile GEMM); CRNN_FLAG_GEMM_TILE_KERNELS off
static bool dense1_dgrad_wres(const crnn_config* cfg, const Dims& d) {
  return cfg->mfma_bf16 == 2 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && crnn_gemm_wres_supported(d.feat, d.tds) == CRNN_OK &&
         (long)d.T * d.B * d.feat < (1L << 31);
}
// dense2's backward (both gradients, the bias gradient and the dropout multiplier of its input) as one fp32 kernel with the weights in registers
// (dense.hip, round 5) in every precision mode; CRNN_FLAG_GEMM_TILE_KERNELS keeps the tile GEMMs + column reduce + dropout pass
static bool dense2_bwd_fused(const crnn_config* cfg, const Dims& d) {
  return !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && crnn_dense_bwd_small_supported((long)d.T * d.B, 2 * d.u, d.C) == CRNN_OK;
}
// The BatchNorm-2 fusions into the depthwise row-stream kernels (fuse_bn2_dw below): opt-in flags for bf16 tensors (the re-forming is VALU work the bf16
// kernels have no issue slots for: measured neutral), the default schedule for fp32 tensors since round 4 (half the elements per byte: -0.6 ms of 14.8 per
// step), CRNN_FLAG_NO_BN2_DW_FUSION switches them off.
// spatial transformer: the localisation net as one workgroup per sample (crnn_loc_net_fwd / crnn_loc_net_bwd, stn.hip) where a sample's maps fit
bool loc_net_fused(const crnn_config* c, const Dims& d) {
  return !(c->flags & CRNN_FLAG_LOC_NET_KERNELS) && crnn_loc_net_fused_supported(d.H0, d.W0) == CRNN_OK;
}
// block 1's single-channel stage with BatchNorm-1 folded into the neighbouring kernels (conv.hip: crnn_dwconv3x3_c1_fwd, crnn_pw1_bn_fwd, crnn_pw1_bn_bwd)
bool block1_fused(const crnn_config* c, int ci, int dtd) { return ci == 1 && dtd == CRNN_F32 && !(c->flags & CRNN_FLAG_BLOCK1_KERNELS) && c->imgw + 4 <= 255; }
bool bn2_dw_fusion_on(const crnn_config* c) {
  if (c->flags & CRNN_FLAG_NO_BN2_DW_FUSION) return false;
  return (c->flags & CRNN_FLAG_BN2_DW_FUSION) || c->mfma_bf16 != 2;
}
bool bn2_stats_fusion_on(const crnn_config* c) { return bn2_dw_fusion_on(c) && ((c->flags & CRNN_FLAG_BN2_STATS_FUSION) || c->mfma_bf16 != 2); }
#ifndef CRNN_BF16_POOL_FUSION
#define CRNN_BF16_POOL_FUSION 0   // experiment (scripts/gpu_ab_libs.sh): bf16 tensors, the POOLED blocks' outputs through the prologue kernels (q at the windows' arg-max), statistics fused
#endif
// ... per block: bf16 tensors keep the fusion opt-in for the un-pooled blocks (measured neutral to slower); block = the block whose output it is
bool bn2_dw_fusion_block(const crnn_config* c, int block) {
  if (bn2_dw_fusion_on(c)) return true;
  return CRNN_BF16_POOL_FUSION && !(c->flags & CRNN_FLAG_NO_BN2_DW_FUSION) && c->mfma_bf16 == 2 && kBlocks[block - 1].ph * kBlocks[block - 1].pw != 1;
}
bool bn2_stats_fusion_block(const crnn_config* c, int block) {
  return bn2_stats_fusion_on(c) || (CRNN_BF16_POOL_FUSION && bn2_dw_fusion_block(c, block) && kBlocks[block - 1].ph * kBlocks[block - 1].pw != 1);
}

// storage of the recurrent weights the recurrences multiply with: bf16 copies in the bf16 modes (u % 128 == 0), else fp32
int rnn_dtu(const crnn_config* c) { return (c->mfma_bf16 && c->units % 128 == 0) ? CRNN_BF16 : CRNN_F32; }
// recurrences as persistent one-launch-per-layer kernels (rnn_persist.hip, gru_persist.hip) unless switched off or unsupported
bool rnn_persist(const crnn_config* c) {
  if (c->flags & CRNN_FLAG_RNN_STEP_KERNELS) return false;
  return (c->gru ? crnn_gru_persist_supported(c->units, rnn_dtu(c)) : crnn_lstm_persist_supported(c->units, rnn_dtu(c))) == 0;
}

size_t deferred_scratch_bytes(const crnn_config* cfg, const Dims& d);   // (defined with the deferred second stages below)

// uw argument of the persistent recurrences: automatic workgroup size, XCD-local clusters unless the linear map is asked for.
// Measured at B = 256, u = 256 (profiles/r03_lstm_cache_policy.txt): with the XCD-local map and, once a cluster has verified that its
// members share an XCD, plain exchange stores the LSTM forward takes 98 us per layer (linear map + write-through stores: 131) and the
// BPTT 142 us (198).
int rnn_uw(const crnn_config* c) { return (c->flags & CRNN_FLAG_RNN_LINEAR_CLUSTERS) ? 0 : CRNN_RNN_XCD_LOCAL; }

Plan make_plan(const crnn_config* c) {
  Dims d = make_dims(c);
  Plan P;
  const long B = d.B;
  // BatchNorm state blocks [mean|var|scale|shift] first: their offsets must stay small (int) for crnn_bn_update
  for (int i = 1; i <= 7; ++i) {
    P.add("bn1s" + std::to_string(i), 4L * d.bc[i - 1]);
    P.add("bn2s" + std::to_string(i), 4L * d.bc[i]);
  }
  // STN locnet
  P.add("pool1", B * d.Hs1 * d.Ws1);
  P.add("c1", B * d.Ho1 * d.Wo1 * 20);
  P.add("pool2", B * d.Hs2 * d.Ws2 * 20);
  P.add("flat", B * d.stn_flat);
  P.add("fc1", B * 50);
  P.add("theta", B * 6);
  P.add("x0", B * d.Hp * d.Wp);
  long maxact = 0, maxparts = 0;
  for (int i = 1; i <= 7; ++i) {
    std::string p = std::to_string(i);
    long M = B * d.bh[i] * d.bw[i];
    int ci = d.bc[i - 1], co = d.bc[i];
    long Mo = B * (d.bh[i] / kBlocks[i - 1].ph) * (d.bw[i] / kBlocks[i - 1].pw);
    // storage mode 2 (bf16 tensors): every conv-stack tensor with >= 4 channels is kept as bf16
    const int sdt_in = (c->mfma_bf16 == 2 && ci % 4 == 0) ? CRNN_BF16 : CRNN_F32;
    const int sdt_out = (c->mfma_bf16 == 2) ? CRNN_BF16 : CRNN_F32;
    P.add("d" + p, M * ci, sdt_in); P.add("a" + p, M * ci, sdt_in); P.add("q" + p, M * co, sdt_out); P.add("x" + p, Mo * co, sdt_out);
    if (pool_argmax_q(c, i)) P.add("qm" + p, Mo * co, sdt_out);   // q at each pool window's arg-max (training forward -> the backward's statistics pass)
    // dropout keep bytes of the block output (one per 8 elements) for the prologue depthwise kernels of block i+1 (fuse_bn2_dw)
    if (i < 7 && (kBlocks[i - 1].ph * kBlocks[i - 1].pw == 1 || pool_argmax_q(c, i)) && co % 8 == 0 && (c->mfma_bf16 == 2 || bn2_dw_fusion_on(c)))
      P.add("dm" + p, (Mo * co / 8 + 3) / 4);
    maxact = lmax(maxact, M * co);
    long tiles = crnn_dwconv_num_tiles(d.B, d.bh[i], d.bw[i]);
    maxparts = lmax(maxparts, tiles * 9L * ci);
    maxparts = lmax(maxparts, (long)crnn_dwconv_bwd_fused_rows(d.B, d.bh[i], d.bw[i], ci) * 9L * ci);
    maxparts = lmax(maxparts, (long)crnn_dwconv_fwd_stream_rows(d.B, d.bh[i], d.bw[i], ci) * 2L * ci);
    maxparts = lmax(maxparts, (long)crnn_dwconv_fwd_stream_rows_ex(d.B, d.bh[i], d.bw[i], ci, CRNN_F32) * 2L * ci);
    maxparts = lmax(maxparts, (long)crnn_dwconv_bwd_stream_rows(d.B, d.bh[i], d.bw[i], ci) * 9L * ci);
    maxparts = lmax(maxparts, (long)crnn_dwconv_bwd_stream_rows_ex(d.B, d.bh[i], d.bw[i], ci, CRNN_F32) * 9L * ci);
    maxparts = lmax(maxparts, (long)crnn_colreduce_chunks(M) * 2L * lmax(ci, co));
    if (ci == 1) maxparts = lmax(maxparts, lmax((long)crnn_dwconv_c1_stat_rows(d.B, d.bh[i], d.bw[i]) * 2L, (long)crnn_pw1_bn_bwd_rows(M) * (co + 2L) + 64));
    maxparts = lmax(maxparts, (long)crnn_pwconv_stat_rows(M) * 2L * co);
    maxparts = lmax(maxparts, (long)crnn_pwconv_fwd_wres_rows(M, co, ci) * 2L * co);   // one row per IO wave and stripe lane: more rows than tiles at small batches
    maxparts = lmax(maxparts, (long)crnn_bn_bwd_chunks(M) * 2L * lmax(ci, co));
    maxparts = lmax(maxparts, (long)crnn_gemm_wres_bnstats_rows(M, ci, co) * 2L * ci);
    maxparts = lmax(maxparts, (long)crnn_gemm_wres3_stat_rows(M, co, ci) * 2L * co);   // parity mode, weights-resident plane kernels (gemm_wres3.hip): forward ...
    maxparts = lmax(maxparts, (long)crnn_gemm_wres3_stat_rows(M, ci, co) * 2L * ci);   // ... and data gradient
  }
  const long TB = (long)d.T * B;
  P.add("dn1", TB * d.tds);
  for (int l = 1; l <= 2; ++l) {
    std::string p = std::to_string(l);
    for (const char* dir : {"f", "b"}) {
      P.add("xw" + p + dir, TB * d.G); P.add("cs" + p + dir, TB * d.u); P.add("gt" + p + dir, TB * d.G);
      P.add("ut" + p + dir, (long)d.G * d.u); P.add("dz" + p + dir, TB * d.G);
      P.add("dbp" + p + dir, (long)crnn_rnn_db_rows((int)B) * d.G);   // bias-gradient partials of the persistent LSTM backward (one row per 16-row batch tile)
      P.add("wt" + p + dir, (long)d.G * (l == 1 ? d.tds : d.u));   // bf16 W^T of the input projection (streaming xw kernel), oversized by 2
    }
  }
  P.add("h1f", TB * d.u); P.add("h1b", TB * d.u); P.add("r1", TB * d.u);
  P.add("h2", TB * 2 * d.u); P.add("r2d", TB * 2 * d.u);
  P.add("logits", TB * d.C); P.add("ypred", TB * d.C);
  // backward
  P.add("dlogits", TB * d.C); P.add("dr2", TB * 2 * d.u); P.add("dr1", TB * d.u);
  P.add("dcf", B * d.u); P.add("dcb", B * d.u); P.add("dhpf", B * d.u); P.add("dhpb", B * d.u);
  P.add("ddn1", TB * d.tds); P.add("gbm", TB * d.tds);
  if (dense1_dgrad_wres(c, d)) P.add("gbm16", TB * d.tds, CRNN_BF16);   // bf16 copy of dense1's output gradient (operand of the weights-resident GEMM)
  maxact = lmax(maxact, TB * d.feat);
  // gradient ping-pong buffers: sized for fp32, hold bf16 tensors in storage mode 2
  P.add("gA", maxact); P.add("gB", maxact); P.add("gC", maxact);   // conv-stack gradient buffers (three: a weight-gradient GEMM on the side stream may still read one)
  P.add("dtheta", B * 6); P.add("dfc1", B * 50); P.add("dflat", B * d.stn_flat);
  if (c->stn && loc_net_fused(c, d)) P.add("locterms", crnn_loc_net_bwd_scratch((int)B));   // the samples' convolution weight-gradient terms (crnn_loc_net_bwd)
  P.add("dpool2", B * d.Hs2 * d.Ws2 * 20); P.add("dc1", B * d.Ho1 * d.Wo1 * 20);
  maxparts = lmax(maxparts, (long)crnn_colreduce_chunks(TB) * lmax(d.G, lmax(d.tds, d.C)));
  maxparts = lmax(maxparts, (long)crnn_colreduce_chunks(B * d.Ho1 * d.Wo1) * 64);
  if (c->stn) {
    maxparts = lmax(maxparts, ((long)crnn_loc_conv_wgrad_chunks(d.B, d.Hs1, d.Ws1) + 1) * (25L * 20 + 20));
    maxparts = lmax(maxparts, ((long)crnn_loc_conv_wgrad_chunks(d.B, d.Hs2, d.Ws2) + 1) * (25L * 20 * 20 + 20));
  }
  P.add("partials", maxparts);
  { long n = 0;   // BatchNorm-2 backward statistics taken by the next block's depthwise-stage backward (fuse_bn2_dw): they outlive that block's other partials
    for (int i = 1; i <= 6; ++i)
      n = lmax(n, (long)crnn_dwconv_bwd_stream_rows_ex(d.B, d.bh[i + 1], d.bw[i + 1], d.bc[i], c->mfma_bf16 == 2 ? CRNN_BF16 : CRNN_F32) * 2L * d.bc[i]);
    if ((c->mfma_bf16 == 2 || bn2_stats_fusion_on(c)) && n) P.add("bn2parts", n); }
  { long pw = 0; for (int i = 2; i <= 7; ++i) pw += (long)d.bc[i - 1] * d.bc[i];
    // bf16 W^T copies of the pointwise-conv weights (bf16 modes) + dense2's W^T as 128 rows of 2u (rows >= num_classes are never written: the streaming
    // GEMM's columns for them are never read -- dense2_stream below)
    P.add("pwT", pw + 128L * 2 * d.u + (dense1_stream(c, d) ? (long)d.feat * d.tds : 0), CRNN_BF16); }   // (+ dense1's W^T [tds][feat], dense1_stream)
  if (c->mfma_bf16) P.add("lg128", TB * 128);   // dense2's raw products over the padded weight matrix
  if (dense2_bwd_fused(c, d)) {
    P.add("d2part", (long)(crnn_dense_bwd_small_scratch_bytes(TB, 2 * d.u, d.C) / sizeof(float)));   // per-workgroup partial gradients of dense2
    P.add("keep9", (TB * 2 * d.u / 8 + 3) / 4);   // keep bytes of the Dropout(.2) under dense2 (written by the forward's dropout pass, read by the one-pass backward)
  }
  P.add("pbf", make_layout(c).total, CRNN_BF16);   // bf16 shadow of the parameter buffer (GEMM B operands in the bf16 modes)
  if (!c->mfma_bf16) {   // parity mode: bf16 planes of the pointwise-conv weights (CRNN_FLAG_WEIGHT_PLANES, weight_planes below); 3 planes x the b2_pw .. b7_pw span
    const Layout L = make_layout(c);
    P.add("p3", 3 * (L.off("b7_pw") + pad4((long)d.bc[6] * d.bc[7]) - L.off("b2_pw")), CRNN_BF16);
  }
  P.add("coef", 2 * 1024);
  P.add("fold", 32 * 2 * 1024);   // chunk sums of long BatchNorm partial lists (crnn_bn_finalize_folded)
  P.add("gemm_scratch", 16L * 1024 * 1024);   // 64 MiB of split-reduction partials (main stream)
  P.add("gemm_scratch2", 16L * 1024 * 1024);  // the same for the side stream of the backward
  P.add("partials2", lmax((long)crnn_colreduce_chunks(TB) * lmax(d.G, lmax(d.tds, d.C)), 1024));
  if (c->mfma_bf16 && (c->flags & CRNN_FLAG_DEFERRED_SUMS) && !(c->flags & CRNN_FLAG_GEMM_TILE_KERNELS))
    P.add("wgrad_scratch", (long)(deferred_scratch_bytes(c, d) / sizeof(float)) + 64);   // partial tiles of the deferred weight-gradient second stages
  if (rnn_persist(c)) P.add("rnnx", (long)((crnn_lstm_persist_xbuf_bytes(d.T, d.B, d.u, rnn_dtu(c)) + 3) / 4));   // h_t / dz_t exchange tiles
  return P;
}

const size_t kGemmScratchBytes = 64UL * 1024 * 1024;

// Second stages of the streaming weight gradients, collected over a backward stage and run as one launch (crnn_wgrad_sum_batch): each
// deferred first stage keeps its partial tiles in its own piece of the "wgrad_scratch" workspace tensor until the flush.
struct Deferred { std::vector<crnn_sum_job> jobs; size_t used = 0, cap = 0; float* base = nullptr; };

struct Ctx {
  const crnn_config* cfg; Dims d; Layout L; Plan P;
  const float* params; float* grads; float* ws; hipStream_t s;
  Deferred* def = nullptr;   // null: every second stage right after its first stage
  const float* p(const std::string& n) const { return params + L.off(n); }
  float* g(const std::string& n) const { return grads + L.off(n); }
  float* w(const std::string& n) const { return ws + P.off(n); }
  int dt(const std::string& n) const { return P.dt(n); }
  int gdt() const { return cfg->mfma_bf16 == 2 ? CRNN_BF16 : CRNN_F32; }   // storage of the conv-stack gradients
  bool side = false;   // side-stream context: its own split-reduction scratch and reduction partials
  float* scratch() const { return ws + P.off(side ? "gemm_scratch2" : "gemm_scratch"); }
  float* partials() const { return ws + P.off(side ? "partials2" : "partials"); }
};

// ... and the shapes crnn_bn_act_pool_drop_qmax_ex / crnn_bn_bwd_qmax_ex take (whole windows, 32-bit element counter, 16-byte aligned tensors): the training
// forward and the backward both decide with this
bool use_qmax(const Ctx& c, int i) {
  if (!pool_argmax_q(c.cfg, i) || !c.P.has("qm" + std::to_string(i))) return false;
  const int ph = kBlocks[i - 1].ph, pw = kBlocks[i - 1].pw, H = c.d.bh[i], W = c.d.bw[i], co = c.d.bc[i];
  const long Mo = (long)c.d.B * (H / ph) * (W / pw);
  const std::string p = std::to_string(i);
  return H % ph == 0 && W % pw == 0 && co % 4 == 0 && Mo * co + 8192L * 256 * 8 < (1L << 31) &&
         ((((uintptr_t)c.w("q" + p) | (uintptr_t)c.w("x" + p) | (uintptr_t)c.w("qm" + p) | (uintptr_t)c.w("bn2s" + p)) & 15) == 0);
}
// In the bf16 modes a weight operand (B of the NN / NT GEMMs, i.e. a pointer into the parameter buffer) is read from
// the bf16 shadow copy refreshed at the start of every forward: half the L2->LDS bytes, identical rounding (RNE).
const float* weight_operand(const Ctx& c, int mode, const float* B, int* dtB) {
  if (c.cfg->mfma_bf16 && mode != 2 && B >= c.params && B < c.params + c.L.total) {
    *dtB = CRNN_BF16;
    return reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(c.ws + c.P.off("pbf")) + (B - c.params));
  }
  return B;
}
// GEMMs of the conv stack / dense layers / RNN input projections: bf16 products when cfg->mfma_bf16
// planes: bf16 planes per operand of the parity mode's products (3 = fp32-accurate; 2 = 16 significant bits per factor, conv_planes below)
int gemm(const Ctx& c, int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
         const float* bias = nullptr, int act = 0, int acc = 0, int perm = 0, int planes = 3) {
  if (c.cfg->mfma_bf16) {
    int dtB = CRNN_F32;
    B = weight_operand(c, mode, B, &dtB);
    return crnn_gemm_bf16_ex(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, CRNN_F32, dtB, CRNN_F32, c.s);
  }
  // parity mode: fp32 tensors; the products from three bf16 planes per operand (fp32-level accuracy on the 16x faster bf16 matrix path,
  // crnn_gemm_f32x3).  CRNN_FLAG_F32_MFMA_GEMMS: fp32 MFMA (an fmaf chain bit for bit)
  if (c.cfg->flags & CRNN_FLAG_F32_MFMA_GEMMS)
    return crnn_gemm_f32(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, c.s);
  if (planes == 2) return crnn_gemm_f32x2(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, c.s);
  return crnn_gemm_f32x3(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, c.s);
}
// product selector of crnn_pwconv_fwd: 1 = bf16 products (bf16 modes), 2 = three-plane fp32-accurate products (parity mode), 0 = fp32 MFMA
int pw_products(const crnn_config* cfg) { return cfg->mfma_bf16 ? 1 : ((cfg->flags & CRNN_FLAG_F32_MFMA_GEMMS) ? 0 : 2); }
// Parity mode: bf16 planes per operand of the step's GEMMs.  Forward: three (every kept partial product exact: fp32-level accuracy, the 1e-3 logit /
// CTC / bit-exact arg-max parity is asserted on this path); CRNN_FLAG_TWO_PLANE_FORWARD (opt-in): two in the conv stack's pointwise convolutions.
// Backward (every weight- and data-gradient GEMM: conv stack, dense layers, RNN projections; the recurrences themselves stay on the fp32 MFMA):
// two -- hi*hi + hi*mid + mid*hi, 16 significant bits per factor, relative error of a product <= 3 * 2^-18 (between TF32's 2^-11 and
// fp32's 2^-24), half the MFMA work: gradients within 1e-5 of the three-plane ones; CRNN_FLAG_THREE_PLANE_BACKWARD: three there too.
int conv_planes(const crnn_config* cfg, bool backward) {
  return backward ? ((cfg->flags & CRNN_FLAG_THREE_PLANE_BACKWARD) ? 3 : 2) : ((cfg->flags & CRNN_FLAG_TWO_PLANE_FORWARD) ? 2 : 3);
}
// Parity mode, round 6: the pointwise convolutions' forward product and data gradient on the weights-resident plane kernels (gemm_wres3.hip) for reductions
// of at most 256 channels; CRNN_FLAG_GEMM_TILE_KERNELS keeps the tile kernel (same planes and products, another accumulation order).  At a reduction of 512
// the resident kernel is slower than the tile kernel (its 64-channel slices ingest and split every pixel row eight times: profiles/r06_wres3_bench.txt).
bool wres3_on(const crnn_config* cfg, int reduction) {
  return !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && reduction <= crnn_knob("CRNN_W3_MAXK", 256);
}
// ... as crnn_pwconv_fwd's product selector for a forward conv of the stack (3 = two planes)
int pw_products_fwd(const crnn_config* cfg) { return (pw_products(cfg) == 2 && conv_planes(cfg, false) == 2) ? 3 : pw_products(cfg); }
// Parity mode with three-plane GEMMs, CRNN_FLAG_WEIGHT_PLANES (opt-in): the pointwise weights of blocks 2..7 are split into their bf16 planes ONCE, at
// the start of the forward pass (crnn_split3_planes over the b2_pw .. b7_pw span of the parameter buffer -> workspace tensor "p3"), instead of by
// every tile of the forward and data-gradient GEMMs that stages them (a tile of 128 rows re-splits the whole weight matrix).  Same words in LDS:
// bit-identical.  Not the default: the staging waves pay more for three 8-byte loads per item than for the split arithmetic (include/crnn_mi355x.h).
// `stride` = elements between planes; null: no planes for this weight.
struct WeightPlanes { long lo = 0, n = 0; };
WeightPlanes weight_planes_span(const Ctx& c) {
  WeightPlanes s;
  if (pw_products(c.cfg) != 2 || !(c.cfg->flags & CRNN_FLAG_WEIGHT_PLANES) || c.P.off("p3") < 0) return s;
  s.lo = c.L.off("b2_pw"); s.n = c.L.off("b7_pw") + pad4((long)c.d.bc[6] * c.d.bc[7]) - s.lo;
  if (((uintptr_t)(c.params + s.lo) & 15) || (s.n & 3)) s.n = 0;
  return s;
}
const void* weight_planes(const Ctx& c, const float* w, long* stride, bool backward) {
  *stride = 0;
  if (conv_planes(c.cfg, backward) != 3) return nullptr;        // (the planes tensor holds three planes)
  const WeightPlanes s = weight_planes_span(c);
  *stride = s.n;
  if (!s.n || w < c.params + s.lo || w >= c.params + s.lo + s.n || ((w - c.params - s.lo) & 3)) return nullptr;
  return reinterpret_cast<const bf16_t*>(c.ws + c.P.off("p3")) + (w - c.params - s.lo);
}
// GEMM with explicit operand / result storage types (storage mode 2); falls back to the plain entry points otherwise
int gemm_t(const Ctx& c, int mode, const float* A, int dtA, const float* B, int dtB, float* C, int dtC, int M, int N, int K, int lda,
           int ldb, int ldc, const float* bias = nullptr, int act = 0, int acc = 0, int perm = 0, int planes = 3) {
  if (c.cfg->mfma_bf16 == 2) {
    B = weight_operand(c, mode, B, &dtB);
    return crnn_gemm_bf16_ex(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, dtA, dtB, dtC, c.s);
  }
  if (dtA != CRNN_F32 || dtB != CRNN_F32 || dtC != CRNN_F32) return CRNN_ERR_ARG;
  return gemm(c, mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, planes);
}
// Training with bf16 conv-stack tensors: the BatchNorm + ReLU6 between a block's depthwise and pointwise convolutions is
// applied by the pointwise GEMMs themselves (forward and weight gradient) while they stage the operand; the activated
// tensor `a` is not written (one read + one write pass per block less).  CRNN_FLAG_NO_DW_BN_FUSION keeps the two-pass path.
bool fuse_dw_bn(const crnn_config* cfg, int dtd, int dtq, int ci) {
  return !(cfg->flags & CRNN_FLAG_NO_DW_BN_FUSION) && cfg->mfma_bf16 == 2 && dtd == CRNN_BF16 && dtq == CRNN_BF16 && ci % 8 == 0 && ci <= 512;
}
// parity mode with three-plane GEMMs: the same fusion in the staging waves of crnn_gemm_f32x3's kernel (forward product and weight gradient)
bool fuse_dw_bn_x3(const crnn_config* cfg, int dtd, int dtq, int ci) {
  return !(cfg->flags & CRNN_FLAG_NO_DW_BN_FUSION) && pw_products(cfg) == 2 && dtd == CRNN_F32 && dtq == CRNN_F32 && ci % 4 == 0 && ci >= 16 && ci <= 512;
}
// The fused entry points taken on that decision have no fallback once the activated tensor was skipped: every pointer they are handed must
// satisfy their 16-byte rule up front (workspace tensors are 256-byte aligned by make_plan; the caller's parameter / gradient / workspace
// base pointers are what can break it)
bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}
// Training with bf16 conv-stack tensors: the output x_i = Dropout(ReLU6(BatchNorm-2(q_i))) of an un-pooled block i (1, 2, 4, 6; utils.py:48-56) is
// not materialised -- block i+1's depthwise row-stream kernels apply it to q_i after the rows have landed in LDS (forward:
// crnn_dwconv3x3_fwd_stream_pro) and re-form it the same way for the depthwise weight gradient (backward: crnn_dwconv3x3_bwd_stream_pro): one write
// and two read passes of x_i less per step.  ONE decision for both passes (the backward has no x_i to fall back to).
// bf16 tensors: opt-in (CRNN_FLAG_BN2_DW_FUSION; bit-identical to crnn_bn_act_pool_drop_ex + the plain kernels): the re-forming is VALU work on a few
// waves of kernels that are otherwise bandwidth-bound, and what the step gains in bytes it loses in issue slots -- measured -1.5 % ... +2.5 % step
// time depending on the box.  fp32 tensors (the parity mode): the default -- half the elements per byte, the same kernels stay
// bandwidth-bound (bn2_dw_fusion_on above).
bool fuse_bn2_dw_shape(const crnn_config* cfg, const Dims& d, const Plan& P, int i) {
  if (i < 1 || i > 6) return false;
  if (!bn2_dw_fusion_block(cfg, i) || (cfg->flags & (CRNN_FLAG_DW_TILE_KERNEL | CRNN_FLAG_NO_DW_BWD_FUSION))) return false;
  const std::string p = std::to_string(i), n = std::to_string(i + 1);
  // a pooled block (round 6): its output is Dropout(ReLU6(BatchNorm-2(.))) of q at each window's arg-max, element by element -- the tensor "qm<i>" the training
  // forward keeps (pool_argmax_q) stands in for q_i in both prologue kernels, which then also take the statistics pass of that BatchNorm's backward
  const bool pooled = kBlocks[i - 1].ph * kBlocks[i - 1].pw != 1;
  if (pooled && (!pool_argmax_q(cfg, i) || !P.has("qm" + p) || d.bh[i] % kBlocks[i - 1].ph || d.bw[i] % kBlocks[i - 1].pw || d.bc[i] % 4 ||
                 (long)d.B * d.bh[i + 1] * d.bw[i + 1] * d.bc[i] + 8192L * 256 * 8 >= (1L << 31))) return false;
  const int dt = P.dt("q" + p);                       // bf16 tensors (throughput mode) or fp32 tensors (round 4: the parity mode's forms of the same kernels)
  if (P.dt("x" + p) != dt || P.dt("d" + n) != dt || P.off("dm" + p) < 0) return false;
  if (bn2_stats_fusion_block(cfg, i) && P.off("bn2parts") < 0) return false;
  const int H = d.bh[i + 1], W = d.bw[i + 1], C = d.bc[i];
  return crnn_dwconv_fwd_stream_pro_supported_ex(d.B, H, W, C, dt) == CRNN_OK && crnn_dwconv_bwd_stream_pro_supported_ex(d.B, H, W, C, dt) == CRNN_OK &&
         (dt == CRNN_F32 || crnn_dwconv_bwd_fused_supported(H, W, C) == CRNN_OK);
}
// the tensor the prologue kernels of block i + 1 form x_i from: q_i, or q_i at the pool windows' arg-max
float* pro_src(const Ctx& c, int i) { return c.w((kBlocks[i - 1].ph * kBlocks[i - 1].pw != 1 ? "qm" : "q") + std::to_string(i)); }
bool fuse_bn2_dw(const Ctx& c, int i) {
  if (!fuse_bn2_dw_shape(c.cfg, c.d, c.P, i)) return false;
  const std::string p = std::to_string(i), n = std::to_string(i + 1);
  return aligned16(c.w("q" + p), c.w("bn2s" + p), c.w("d" + n), c.p("b" + n + "_dw")) && aligned16(pro_src(c, i));
}
const float* keep_bytes(const Ctx& c, int i) { return c.cfg->dropout ? c.w("dm" + std::to_string(i)) : nullptr; }
// always-fp32 GEMM (spatial-transformer localisation net: tiny, and theta is precision-sensitive)
int gemm32(const Ctx& c, int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
           const float* bias = nullptr, int act = 0, int acc = 0, int perm = 0) {
  return crnn_gemm_f32(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, acc, perm, c.scratch(), kGemmScratchBytes, c.s);
}

int flush_deferred(const Ctx& c) {
  if (!c.def || c.def->jobs.empty()) return CRNN_OK;
  const int rc = crnn_wgrad_sum_batch(c.def->jobs.data(), (int)c.def->jobs.size(), c.s);
  c.def->jobs.clear(); c.def->used = 0;
  return rc;
}
// a piece of the deferred scratch for `bytes` of partial tiles (null: not deferring, or it can never fit); flushes first when full
float* deferred_scratch(const Ctx& c, size_t bytes, int* rc) {
  *rc = CRNN_OK;
  if (!c.def || bytes == 0 || bytes > c.def->cap) return nullptr;
  if (c.def->used + bytes > c.def->cap || (int)c.def->jobs.size() >= CRNN_SUM_BATCH_MAX) { *rc = flush_deferred(c); if (*rc) return nullptr; }
  float* ptr = c.def->base + c.def->used / sizeof(float);
  c.def->used += (bytes + 255) & ~(size_t)255;
  return ptr;
}
// bytes of deferred scratch one backward stage needs at most (make_plan)
size_t deferred_scratch_bytes(const crnn_config* cfg, const Dims& d) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const long TB = (long)d.T * d.B, K1 = (long)(d.T - 1) * d.B;
  const int Nh = cfg->gru ? 2 * d.u : d.G;
  size_t top = 0, bottom = 0;
  for (int l = 1; l <= 2; ++l) {
    const int din = l == 1 ? d.tds : d.u;
    top += 2 * al(crnn_pwconv_wgrad_stream_scratch_bytes(TB, d.G, din)) + 2 * al(crnn_pwconv_wgrad_stream_scratch_bytes(K1, Nh, d.u));
  }
  for (int i = 2; i <= 7; ++i) bottom += al(crnn_pwconv_wgrad_stream_scratch_bytes((long)d.B * d.bh[i] * d.bw[i], d.bc[i], d.bc[i - 1]));
  return top > bottom ? top : bottom;
}

int colsum(const Ctx& c, const float* x, long M, int C, int ld, float* out) {
  CRNN_TRY(crnn_colreduce(x, c.partials(), M, C, ld, 1, c.s));
  return crnn_partials_sum(c.partials(), crnn_colreduce_chunks(M), C, out, 1.f, c.s);
}

int check_cfg(const crnn_config* c) {
  if (!c || c->batch <= 0) return CRNN_ERR_ARG;
  if (c->units < 64 || c->units % 64) return CRNN_ERR_UNSUPPORTED;
  if (c->mfma_bf16 < 0 || c->mfma_bf16 > 2) return CRNN_ERR_ARG;
  if (c->num_classes > 64 || c->num_classes < 2) return CRNN_ERR_UNSUPPORTED;
  if (2 * c->max_len + 1 > 64) return CRNN_ERR_UNSUPPORTED;
  if (c->tds % 4) return CRNN_ERR_UNSUPPORTED;
  Dims d = make_dims(c);
  if (c->stn && (d.Ho2 < 1 || d.Wo2 < 1)) return CRNN_ERR_UNSUPPORTED;
  if (d.T < 3 || d.feat < 1) return CRNN_ERR_UNSUPPORTED;
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" int crnn_num_params(const crnn_config* cfg) { return (int)make_layout(cfg).params.size(); }
extern "C" long crnn_params_total(const crnn_config* cfg) { return make_layout(cfg).total; }
extern "C" int crnn_param_info(const crnn_config* cfg, int idx, char* name, int name_cap, long* offset, long* size,
                               int* ndim, int* dims) {
  Layout L = make_layout(cfg);
  if (idx < 0 || idx >= (int)L.params.size()) return CRNN_ERR_ARG;
  const Tensor& t = L.params[idx];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = t.off;
  if (size) *size = t.size;
  if (ndim) *ndim = t.ndim;
  if (dims) for (int i = 0; i < 4; ++i) dims[i] = t.dims[i];
  return 0;
}
extern "C" int crnn_time_steps(const crnn_config* cfg) { return make_dims(cfg).T; }
extern "C" int crnn_bn_total(const crnn_config* cfg) {
  Dims d = make_dims(cfg); int n = 0;
  for (int i = 1; i <= 7; ++i) n += d.bc[i - 1] + d.bc[i];
  return n;
}
// idx 0..13: b1_bn1, b1_bn2, b2_bn1, ...
extern "C" int crnn_bn_info(const crnn_config* cfg, int idx, char* name, int name_cap, int* offset, int* channels, long* count) {
  if (idx < 0 || idx >= 14) return CRNN_ERR_ARG;
  Dims d = make_dims(cfg);
  int off = 0;
  for (int k = 0; k < idx; ++k) { int i = k / 2 + 1; off += (k % 2 == 0) ? d.bc[i - 1] : d.bc[i]; }
  int i = idx / 2 + 1;
  if (name && name_cap > 0) snprintf(name, name_cap, "b%d_bn%d", i, idx % 2 + 1);
  if (offset) *offset = off;
  if (channels) *channels = (idx % 2 == 0) ? d.bc[i - 1] : d.bc[i];
  if (count) *count = (long)d.B * d.bh[i] * d.bw[i];
  return 0;
}
extern "C" size_t crnn_workspace_bytes(const crnn_config* cfg) {
  if (check_cfg(cfg)) return 0;
  return (size_t)make_plan(cfg).total * sizeof(float);
}
extern "C" int crnn_ws_tensor_info(const crnn_config* cfg, const char* name, long* offset, long* count, int* dtype) {
  Plan P = make_plan(cfg);
  long o = P.off(name);
  if (o < 0) return CRNN_ERR_ARG;
  if (offset) *offset = o;
  if (count) *count = P.cnt(name);
  if (dtype) *dtype = P.dt(name);
  return 0;
}
// 1 when training does not materialise the output x_block of conv block `block` (1..7): the next block's depthwise row-stream kernels form it
// from q_block in LDS (fp32 tensors: the default; bf16 tensors: CRNN_FLAG_BN2_DW_FUSION; un-pooled blocks, image width 32) -- the workspace tensor "x<block>" is
// then never written by a training forward.  (16-byte aligned parameter / workspace base pointers assumed, as torch allocations are.)
extern "C" int crnn_block_output_fused(const crnn_config* cfg, int block) {
  if (check_cfg(cfg)) return 0;
  return fuse_bn2_dw_shape(cfg, make_dims(cfg), make_plan(cfg), block) ? 1 : 0;
}
extern "C" int crnn_ws_tensor(const crnn_config* cfg, const char* name, long* offset, long* count) {
  Plan P = make_plan(cfg);
  long o = P.off(name);
  if (o < 0) return CRNN_ERR_ARG;
  if (offset) *offset = o;
  if (count) *count = P.cnt(name);
  return 0;
}

namespace {
constexpr int kForkMaxDevices = 64;
// The side stream an entry point may fork to: NULL (run everything on `stream`) when the caller passed none, the same stream, or a stream on a device ordinal
// beyond the fork/join event table (round 6, ADVICE: such a box used to fail the whole step with -3)
hipStream_t side_stream(hipStream_t stream, hipStream_t aux) {
  if (!aux || aux == stream) return nullptr;
  hipDevice_t d = 0;
  if (hipStreamGetDevice(aux, &d) != hipSuccess || d < 0 || d >= kForkMaxDevices) return nullptr;
  return aux;
}
// Two streams with event links (fork: the side stream continues after everything enqueued on the main stream so far; join: the reverse)
struct ForkJoin {
  // The events are created once per (host thread, device) and reused by every call (recording an event again while an earlier
  // wait on it is still queued is well defined: the wait captured the earlier record); nothing is created or destroyed
  // on the step's path.  Keyed by the streams' device: an event belongs to the device it was created on, and a second engine on
  // another GPU driven from the same host thread must not record the first one's events on its streams (invalid resource handle).
  // The events live as long as the thread (a fixed pool of 24 per device, like the runtime's own per-device pools); they are never
  // destroyed because a wait on one of them may still be queued when the thread exits.
  // A device ordinal beyond the table (more than 64 GPUs / partitions visible to one process) does not fail the step: the fork is simply not taken
  // and the side work runs on the main stream (same results; round 6, ADVICE).
  hipStream_t main, aux; int n = 0; bool on; int dev = -1;
  static constexpr int kMaxDevices = kForkMaxDevices, kEvents = 24;
  ForkJoin(hipStream_t m, hipStream_t a) : main(m), aux(a), on(a != nullptr) {
    if (on) {   // the device the side stream lives on
      hipDevice_t d = 0;
      if (hipStreamGetDevice(aux, &d) != hipSuccess || d < 0 || d >= kMaxDevices) on = false; else dev = d;
    }
  }
  int event(int i, hipEvent_t* out) {
    static thread_local hipEvent_t ev[kMaxDevices][kEvents] = {};
    if (i >= kEvents || dev < 0) return CRNN_ERR_ARG;
    hipEvent_t& e = ev[dev][i];
    if (!e) { hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (r != hipSuccess) return (int)r; }
    *out = e;
    return CRNN_OK;
  }
  int link(hipStream_t from, hipStream_t to) {   // `to` continues after everything enqueued on `from` so far
    if (!on) return CRNN_OK;
    hipEvent_t e; CRNN_TRY(event(n++, &e));
    hipError_t r = hipEventRecord(e, from); if (r != hipSuccess) return (int)r;
    r = hipStreamWaitEvent(to, e, 0); return r == hipSuccess ? CRNN_OK : (int)r;
  }
  int fork() { return link(main, aux); }
  int join() { return link(aux, main); }
  // split form of a join: mark the side stream's progress now, make the main stream wait for that point later
  int mark(hipEvent_t* e) {
    if (!on) return CRNN_OK;
    CRNN_TRY(event(n++, e));
    hipError_t r = hipEventRecord(*e, aux); return r == hipSuccess ? CRNN_OK : (int)r;
  }
  int wait(hipEvent_t e) {
    if (!on || !e) return CRNN_OK;
    hipError_t r = hipStreamWaitEvent(main, e, 0); return r == hipSuccess ? CRNN_OK : (int)r;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" int crnn_forward(const crnn_config* cfg, const float* params, const float* bn_mean, const float* bn_var,
                            const float* x, float* ws, size_t ws_bytes, float* y_pred, int train, uint64_t seed,
                            hipStream_t stream) {
  return crnn_forward_ex(cfg, params, bn_mean, bn_var, x, ws, ws_bytes, y_pred, train, seed, stream, nullptr);
}
// aux_stream != NULL (and != stream): work that nothing at the head of the forward waits for runs there next to the spatial transformer's
// small kernels -- today the dropout keep bytes of the block outputs that only exist inside the next depthwise kernels (a function of
// (seed, site, index), 0.1 ms of VALU work at batch 256), joined before block 2's depthwise kernel.  Same results.
extern "C" int crnn_forward_ex(const crnn_config* cfg, const float* params, const float* bn_mean, const float* bn_var,
                               const float* x, float* ws, size_t ws_bytes, float* y_pred, int train, uint64_t seed,
                               hipStream_t stream, hipStream_t aux_stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, nullptr, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  const Dims& d = c.d;
  const int B = d.B;
  aux_stream = side_stream(stream, aux_stream);
  ForkJoin fj(stream, aux_stream);
  bool keep_pending = false;
  if (train && cfg->dropout) {   // the dropout decisions of the block outputs that only exist inside the next depthwise kernels (fuse_bn2_dw)
    void* outs[CRNN_KEEP_BATCH_MAX]; long ng[CRNN_KEEP_BATCH_MAX]; uint32_t lay[CRNN_KEEP_BATCH_MAX]; int n = 0;
    for (int i = 1; i <= 6; ++i)
      if (fuse_bn2_dw(c, i)) { outs[n] = c.w("dm" + std::to_string(i)); ng[n] = (long)B * d.bh[i + 1] * d.bw[i + 1] * d.bc[i] / 8; lay[n] = (uint32_t)i; ++n; }   // (block i + 1's map = block i's output, pooled or not)
    if (n) {
      CRNN_TRY(fj.fork());
      CRNN_TRY(crnn_dropout_keep_bytes_batch(n, outs, ng, lay, kDropBlock, seed, fj.on ? aux_stream : stream));
      keep_pending = fj.on;
    }
  }
  if (cfg->mfma_bf16) CRNN_TRY(crnn_convert_f32_to_bf16(params, c.ws + c.P.off("pbf"), c.L.total, stream));
  { const WeightPlanes wp = weight_planes_span(c);   // parity mode, opt-in: the pointwise weights' bf16 planes, once per step
    if (wp.n) CRNN_TRY(crnn_split3_planes(params + wp.lo, c.ws + c.P.off("p3"), wp.n, wp.n, stream)); }
  // ---- spatial transformer (utils.py:247-258) + ZeroPadding2D (utils.py:63)
  if (cfg->stn && loc_net_fused(cfg, d) && aligned16(c.p("stn_c1_k"), c.p("stn_c2_k"), c.p("stn_c1_b"), c.p("stn_c2_b")) &&
      aligned16(c.w("c1"), c.w("pool2"), c.w("flat"))) {
    CRNN_TRY(crnn_loc_net_fwd(x, c.p("stn_c1_k"), c.p("stn_c1_b"), c.p("stn_c2_k"), c.p("stn_c2_b"), c.p("stn_d1_w"), c.p("stn_d1_b"), c.p("stn_d2_w"),
                              c.p("stn_d2_b"), c.w("pool1"), c.w("c1"), c.w("pool2"), c.w("flat"), c.w("fc1"), c.w("theta"), B, d.H0, d.W0, stream));
    CRNN_TRY(crnn_sampler_fwd(x, c.w("theta"), c.w("x0"), B, d.H0, d.W0, 2, stream));
  } else if (cfg->stn) {
    CRNN_TRY(crnn_maxpool_fwd(x, c.w("pool1"), B, d.H0, d.W0, 1, 2, 2, stream));
    CRNN_TRY(crnn_loc_conv_fwd(c.w("pool1"), c.p("stn_c1_k"), c.p("stn_c1_b"), c.w("c1"), B, d.Hs1, d.Ws1, 1, stream));
    CRNN_TRY(crnn_maxpool_fwd(c.w("c1"), c.w("pool2"), B, d.Ho1, d.Wo1, 20, 2, 2, stream));
    CRNN_TRY(crnn_loc_conv_fwd(c.w("pool2"), c.p("stn_c2_k"), c.p("stn_c2_b"), c.w("flat"), B, d.Hs2, d.Ws2, 20, stream));
    CRNN_TRY(crnn_loc_fc_fwd(c.w("flat"), c.p("stn_d1_w"), c.p("stn_d1_b"), c.p("stn_d2_w"), c.p("stn_d2_b"), c.w("fc1"), c.w("theta"), B,
                             d.stn_flat, stream));
    CRNN_TRY(crnn_sampler_fwd(x, c.w("theta"), c.w("x0"), B, d.H0, d.W0, 2, stream));
  } else {
    CRNN_TRY(crnn_pad_copy(x, c.w("x0"), B, d.H0, d.W0, 2, stream));
  }
  // bf16 modes: W^T (bf16) copies of the pointwise weights of blocks 2..7, one launch
  long pwT_off[8]; for (int i = 0; i < 8; ++i) pwT_off[i] = -1;
  long d2T_off = -1;                       // element offset of dense2's padded W^T inside "pwT" (dense2_stream)
  long d1T_off = -1;                       // ... of dense1's W^T (dense1_stream)
  if (cfg->mfma_bf16) {
    long in_off[8], out_off[8]; int R[8], Cc[8]; int n = 0; long acc = 0;
    for (int i = 2; i <= 7; ++i) {
      const int ci = d.bc[i - 1], co = d.bc[i];
      if (ci % 8 || co % 8) continue;
      in_off[n] = c.L.off("b" + std::to_string(i) + "_pw"); out_off[n] = acc; R[n] = ci; Cc[n] = co;
      pwT_off[i] = acc; acc += (long)ci * co; ++n;
    }
    if (dense2_stream(cfg, d) && n < 8) {   // dense2's W [2u][C] -> W^T rows 0..C-1 of a 128-row matrix behind the pointwise copies
      in_off[n] = c.L.off("dense2_w"); out_off[n] = acc; R[n] = 2 * d.u; Cc[n] = d.C; d2T_off = acc; ++n;
    }
    acc += 128L * 2 * d.u;
    if (dense1_stream(cfg, d) && n < 8) {   // dense1's W [feat][tds] -> W^T [tds][feat] behind it
      in_off[n] = c.L.off("dense1_w"); out_off[n] = acc; R[n] = d.feat; Cc[n] = d.tds; d1T_off = acc; ++n;
    }
    if (n) CRNN_TRY(crnn_transpose_batch(params, c.w("pwT"), n, in_off, out_off, R, Cc, CRNN_BF16, stream));
  }
  // ---- 7 depthwise-separable blocks (utils.py:43-56, 64-70)
  const float* in = c.w("x0");
  const float* pro_q = nullptr; const float* pro_s2 = nullptr;   // pending BatchNorm-2 prologue of the next depthwise kernel (training, fuse_bn2_dw)
  int bn_off = 0;
  if (!train) {   // inference: every BatchNorm's [mean|var|scale|shift] is known before the first conv -- one launch for all 14
    const float *mm[14], *mv[14], *gg[14], *bb[14]; float* st[14]; int cc[14]; int n = 0, off = 0;
    for (int i = 1; i <= 7; ++i) {
      const std::string bp = "b" + std::to_string(i);
      for (int h = 0; h < 2; ++h) {
        const int ch = h ? d.bc[i] : d.bc[i - 1];
        mm[n] = bn_mean + off; mv[n] = bn_var + off; gg[n] = c.p(bp + (h ? "_bn2_g" : "_bn1_g")); bb[n] = c.p(bp + (h ? "_bn2_b" : "_bn1_b"));
        st[n] = c.w((h ? "bn2s" : "bn1s") + std::to_string(i)); cc[n] = ch; off += ch; ++n;
      }
    }
    CRNN_TRY(crnn_bn_infer_state_batch(n, mm, mv, gg, bb, cc, st, stream));
  }
  for (int i = 1; i <= 7; ++i) {
    std::string p = std::to_string(i), bp = "b" + p;
    const int H = d.bh[i], W = d.bw[i], ci = d.bc[i - 1], co = d.bc[i];
    const long M = (long)B * H * W;
    float* dd = c.w("d" + p); float* aa = c.w("a" + p); float* qq = c.w("q" + p); float* xo = c.w("x" + p);
    float* s1 = c.w("bn1s" + p); float* s2 = c.w("bn2s" + p);
    float* parts = c.w("partials");
    const int dtd = c.dt("d" + p), dtq = c.dt("q" + p);
    const int ph = kBlocks[i - 1].ph, pw = kBlocks[i - 1].pw;
    const int slab = (dtd == CRNN_BF16) ? 64 : 32;
    // bf16 maps whose rows fill the 9 KiB step row: the row-stream kernel (dwconv_stream.hip), bit-identical to the halo-tile kernel
    const bool dws = dtd == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_DW_TILE_KERNEL) && crnn_dwconv_fwd_stream_supported(B, H, W, ci) == CRNN_OK;
    if (!train) {
      // inference (learning_phase 0): the BatchNorm scale/shift are known up front, so BN + ReLU6 fold into the
      // epilogue of the conv that feeds them -- the depthwise kernel writes `a` directly and, when the block has no
      // pooling, the pointwise GEMM writes the block output directly: two passes per block instead of four
      bn_off += ci;
      // a pooled block on bf16 maps: BN + ReLU6 + MaxPooling2D all in the pointwise GEMM's epilogue (groups of 2 | 4 consecutive rows); for the
      // (2,2) window the depthwise kernel writes its rows in window-major order -- the un-pooled map q never exists
      int pool_rows = 1;
      if (ph * pw > 1 && cfg->mfma_bf16 && pwT_off[i] >= 0 && dtd == CRNN_BF16 && dtq == CRNN_BF16 && c.dt("x" + p) == CRNN_BF16 &&
          aligned16(aa, xo, s2, c.w("pwT")) &&
          !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && crnn_pwconv_fwd_wres_folded_pool_supported(M, co, ci, ph * pw) == CRNN_OK &&
          ((ph == 1 && pw == 2 && W % 2 == 0) || (ph == 2 && pw == 2 && dws && H % 2 == 0 && W % 2 == 0)))
        pool_rows = ph * pw;
      if (dws) {
        CRNN_TRY(crnn_dwconv3x3_fwd_stream_ex(in, c.p(bp + "_dw"), aa, nullptr, s1, B, H, W, ci, 0, pool_rows == 4 ? 1 : 0, stream));
      } else if (ci % slab == 0) {
        CRNN_TRY(crnn_dwconv3x3_bn_relu6_fwd(in, c.p(bp + "_dw"), s1, aa, B, H, W, ci, dtd, stream));
      } else {
        CRNN_TRY(crnn_dwconv3x3_fwd_ex(in, c.p(bp + "_dw"), dd, nullptr, B, H, W, ci, 0, dtd, stream));
        CRNN_TRY(crnn_bn_act_pool_drop_ex(dd, s1, aa, 1, 1, (int)M, ci, 1, 1, 0.f, 0, 0, dtd, dtd, stream));
      }
      bn_off += co;
      int dtw = CRNN_F32, wt = 0;
      const float* wq = weight_operand(c, 0, c.p(bp + "_pw"), &dtw);
      if (cfg->mfma_bf16 && pwT_off[i] >= 0) {
        wq = reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(c.w("pwT")) + pwT_off[i]);
        dtw = CRNN_BF16; wt = 1;
      }
      const bool one = (ci == 1 && dtd == CRNN_F32);                      // block 1: an outer product, not a GEMM
      const bool fold = (ph * pw == 1 || pool_rows > 1) && (c.dt("x" + p) == dtq);
      if (one) CRNN_TRY(fold ? crnn_pw1_fwd_folded(aa, c.p(bp + "_pw"), xo, M, co, s2, dtq, stream) : crnn_pw1_fwd(aa, c.p(bp + "_pw"), qq, M, co, nullptr, dtq, stream));
      else {
        // bf16 tensors + W^T: the weights-resident kernel (folded BatchNorm in its MFMA waves' epilogue, or the plain product)
        int rc = CRNN_ERR_UNSUPPORTED;
        if (wt && dtd == CRNN_BF16 && dtq == CRNN_BF16 && dtw == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && M <= 0x7fffffffL)
          rc = pool_rows > 1 ? crnn_pwconv_fwd_wres_folded_pool(aa, wq, xo, M, co, ci, s2, pool_rows, stream)
               : fold ? crnn_pwconv_fwd_wres_folded(aa, wq, xo, M, co, ci, s2, stream) : crnn_gemm_wres_bf16(aa, wq, qq, (int)M, co, ci, stream);
        if (pool_rows > 1) { CRNN_TRY(rc); in = xo; continue; }        // (its shape rules were checked above: no fallback that could read window-major rows)
        if (rc == CRNN_ERR_UNSUPPORTED)
          rc = crnn_pwconv_fwd(aa, wq, fold ? xo : qq, M, co, ci, nullptr, fold ? s2 : nullptr, pw_products_fwd(cfg), dtd, dtw, dtq, wt, stream);
        CRNN_TRY(rc);
      }
      if (!fold) CRNN_TRY(crnn_bn_act_pool_drop_ex(qq, s2, xo, B, H, W, co, ph, pw, 0.f, seed, (uint32_t)i, dtq, c.dt("x" + p), stream));
      in = xo;
      continue;
    }
    if (pro_q) {   // the previous block's output was not materialised: its BatchNorm-2 + ReLU6 + dropout run inside this depthwise kernel (fuse_bn2_dw)
      if (keep_pending) { CRNN_TRY(fj.join()); keep_pending = false; }   // the keep bytes are complete
      CRNN_TRY(crnn_dwconv3x3_fwd_stream_pro_ex(pro_q, pro_s2, cfg->dropout ? kDropBlock : 0.f, keep_bytes(c, i - 1), c.p(bp + "_dw"), dd, parts, B, H, W, ci, dtd, stream));
      CRNN_TRY(crnn_bn_finalize_folded(parts, crnn_dwconv_fwd_stream_rows_ex(B, H, W, ci, dtd), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, c.w("fold"), stream));
      pro_q = nullptr; pro_s2 = nullptr;
    } else if (dws) {
      CRNN_TRY(crnn_dwconv3x3_fwd_stream(in, c.p(bp + "_dw"), dd, parts, nullptr, B, H, W, ci, 0, stream));
      CRNN_TRY(crnn_bn_finalize_folded(parts, crnn_dwconv_fwd_stream_rows(B, H, W, ci), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, c.w("fold"), stream));
    } else if (dtd == CRNN_F32 && i > 1 && !(cfg->flags & CRNN_FLAG_DW_TILE_KERNEL) && crnn_dwconv_fwd_stream_supported_ex(B, H, W, ci, CRNN_F32) == CRNN_OK &&
               aligned16(in, dd, c.p(bp + "_dw"))) {
      // fp32 maps (parity mode, round 4): the row-stream kernel's fp32 form -- same outputs as the halo-tile kernel, statistics per workgroup band
      CRNN_TRY(crnn_dwconv3x3_fwd_stream_dt(in, c.p(bp + "_dw"), dd, parts, B, H, W, ci, 0, CRNN_F32, stream));
      CRNN_TRY(crnn_bn_finalize_folded(parts, crnn_dwconv_fwd_stream_rows_ex(B, H, W, ci, CRNN_F32), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, c.w("fold"), stream));
    } else if (ci % 32 == 0 && ci % slab == 0) {
      CRNN_TRY(crnn_dwconv3x3_fwd_ex(in, c.p(bp + "_dw"), dd, parts, B, H, W, ci, 0, dtd, stream));
      CRNN_TRY(crnn_bn_finalize_folded(parts, crnn_dwconv_num_tiles(B, H, W), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, c.w("fold"), stream));
    } else if (block1_fused(cfg, ci, dtd)) {                           // block 1: the one-channel depthwise kernel takes its own statistics
      CRNN_TRY(crnn_dwconv3x3_c1_fwd(in, c.p(bp + "_dw"), dd, parts, B, H, W, stream));
      CRNN_TRY(crnn_bn_finalize(parts, crnn_dwconv_c1_stat_rows(B, H, W), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, stream));
    } else {
      CRNN_TRY(crnn_dwconv3x3_fwd_ex(in, c.p(bp + "_dw"), dd, nullptr, B, H, W, ci, 0, dtd, stream));
      CRNN_TRY(crnn_colreduce_ex(dd, parts, M, ci, ci, 2, dtd, stream));
      CRNN_TRY(crnn_bn_finalize(parts, crnn_colreduce_chunks(M), ci, M, c.p(bp + "_bn1_g"), c.p(bp + "_bn1_b"), s1, stream));
    }
    bn_off += ci;
    const bool fuse_a = fuse_dw_bn(cfg, dtd, dtq, ci);                  // BN + ReLU6 applied while the GEMM stages its operand
    const bool fuse_x3 = fuse_dw_bn_x3(cfg, dtd, dtq, ci) && aligned16(dd, qq, s1, c.p(bp + "_pw"));   // ... by the staging waves of the parity mode's three-plane kernel
    const bool blk1 = block1_fused(cfg, ci, dtd);                       // ... by the outer product of block 1 (crnn_pw1_bn_fwd): no activated tensor either
    if (!fuse_a && !fuse_x3 && !blk1) CRNN_TRY(crnn_bn_act_pool_drop_ex(dd, s1, aa, 1, 1, (int)M, ci, 1, 1, 0.f, 0, 0, dtd, dtd, stream));
    int stat_rows = crnn_pwconv_stat_rows(M);
    {  // pointwise conv; its epilogue also produces the batch statistics of the BatchNorm that follows
      int dtw = CRNN_F32, wt = 0;
      const float* wq = weight_operand(c, 0, c.p(bp + "_pw"), &dtw);
      if (cfg->mfma_bf16 && pwT_off[i] >= 0) {   // bf16 W^T copy made above: both operands contiguous along the reduction
        wq = reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(c.w("pwT")) + pwT_off[i]);
        dtw = CRNN_BF16; wt = 1;
      }
      if (blk1) CRNN_TRY(crnn_pw1_bn_fwd(dd, s1, c.p(bp + "_pw"), qq, M, co, parts, dtq, stream));                      // block 1: outer product of relu6(BN(d))
      else if (ci == 1 && dtd == CRNN_F32) CRNN_TRY(crnn_pw1_fwd(aa, c.p(bp + "_pw"), qq, M, co, parts, dtq, stream));
      else if (fuse_x3) {
        long wps = 0; const void* wpl = weight_planes(c, c.p(bp + "_pw"), &wps, false);
        int rc = wpl ? crnn_pwconv_bnrelu6_fwd_f32x3_pl(dd, s1, c.p(bp + "_pw"), wpl, wps, qq, M, co, ci, parts, stream) : CRNN_ERR_UNSUPPORTED;
        // round 6: the weights' planes resident in registers, the pixel rows streamed once (gemm_wres3.hip) where that kernel wins: K <= 256
        // (at K = 512 the planes of a 128-channel slice do not fit a CU's registers next to the accumulators: the tile kernel stays)
        if (rc == CRNN_ERR_UNSUPPORTED && wres3_on(cfg, ci) && crnn_gemm_wres3_supported(M, co, ci) == CRNN_OK) {
          rc = crnn_pwconv_bnrelu6_fwd_wres3(dd, s1, c.p(bp + "_pw"), qq, M, co, ci, conv_planes(cfg, false), parts, stream);
          if (rc == CRNN_OK) stat_rows = crnn_gemm_wres3_stat_rows(M, co, ci);
        }
        if (rc == CRNN_ERR_UNSUPPORTED)      // (no planes, or ragged tiles: split while staging)
          rc = conv_planes(cfg, false) == 2 ? crnn_pwconv_bnrelu6_fwd_f32x2(dd, s1, c.p(bp + "_pw"), qq, M, co, ci, parts, stream)
                                            : crnn_pwconv_bnrelu6_fwd_f32x3(dd, s1, c.p(bp + "_pw"), qq, M, co, ci, parts, stream);
        CRNN_TRY(rc);
      }
      else if (fuse_a) {
        // weights resident in registers, IO waves transform / drain / take the statistics (gemm_wres.hip) where its shape rules hold
        int rc = CRNN_ERR_UNSUPPORTED;
        if (wt && dtq == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && crnn_pwconv_fwd_wres_supported(M, co, ci) == CRNN_OK) {
          rc = crnn_pwconv_bnrelu6_fwd_wres(dd, s1, wq, qq, M, co, ci, parts, stream);
          if (rc == CRNN_OK) stat_rows = crnn_pwconv_fwd_wres_rows(M, co, ci);
        }
        if (rc == CRNN_ERR_UNSUPPORTED) rc = crnn_pwconv_bnrelu6_fwd(dd, s1, wq, qq, M, co, ci, parts, dtq, wt, stream);
        CRNN_TRY(rc);
      } else CRNN_TRY(crnn_pwconv_fwd(aa, wq, qq, M, co, ci, parts, nullptr, pw_products_fwd(cfg),
                                      dtd, dtw, dtq, wt, stream));
    }
    CRNN_TRY(crnn_bn_finalize_folded(parts, stat_rows, co, M, c.p(bp + "_bn2_g"), c.p(bp + "_bn2_b"), s2, c.w("fold"), stream));
    bn_off += co;
    if (fuse_bn2_dw(c, i)) {   // x_i is formed by block i+1's depthwise kernel from q_i (and again by its backward): not written
      if (ph * pw != 1)          // pooled: from q at each window's arg-max, which one pass over q selects (no BatchNorm output is written)
        CRNN_TRY(crnn_bn_act_pool_drop_qmax_ex(qq, s2, nullptr, pro_src(c, i), B, H, W, co, ph, pw, 0.f, seed, (uint32_t)i, dtq, dtq, stream));
      pro_q = pro_src(c, i); pro_s2 = s2; in = nullptr;
      continue;
    }
    if (use_qmax(c, i))   // (no fallback: the backward takes the same decision)
      CRNN_TRY(crnn_bn_act_pool_drop_qmax_ex(qq, s2, xo, c.w("qm" + p), B, H, W, co, ph, pw, cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, dtq, c.dt("x" + p), stream));
    else
    CRNN_TRY(crnn_bn_act_pool_drop_ex(qq, s2, xo, B, H, W, co, ph, pw, cfg->dropout ? kDropBlock : 0.f, seed,
                                      (uint32_t)i, dtq, c.dt("x" + p), stream));
    in = xo;
  }
  if (keep_pending) { CRNN_TRY(fj.join()); keep_pending = false; }
  // ---- Reshape + dense1 (relu) + Dropout(.4) (utils.py:72-75); output time-major [T][B][tds]
  const int T = d.T, TB = T * B, u = d.u, G = d.G;
  int rc1 = CRNN_ERR_UNSUPPORTED;
  if (d1T_off >= 0 && c.dt("x7") == CRNN_BF16)
    rc1 = crnn_dense_fwd_stream(in, reinterpret_cast<const bf16_t*>(c.w("pwT")) + d1T_off, c.p("dense1_b"), c.w("dn1"), TB, d.tds, d.feat, d.feat, d.feat, 1, T,
                                (train && cfg->dropout) ? kDropDense1 : 0.f, seed, kLayerDense1, stream);
  if (rc1 != CRNN_OK && rc1 != CRNN_ERR_UNSUPPORTED) return rc1;
  if (rc1 != CRNN_OK) {
    CRNN_TRY(gemm_t(c, 0, in, c.dt("x7"), c.p("dense1_w"), CRNN_F32, c.w("dn1"), CRNN_F32, TB, d.tds, d.feat, d.feat, d.tds, d.tds, c.p("dense1_b"), 1, 0, T));
    if (train && cfg->dropout) CRNN_TRY(crnn_dropout(c.w("dn1"), c.w("dn1"), TB, d.tds, d.tds, d.tds, kDropDense1, seed, kLayerDense1, stream));
  }
  // ---- 2 x Bidirectional(LSTM) (utils.py:78-79)
  // bf16 modes: the recurrent products run on the bf16 MFMA from a bf16 U^T (u % 128 == 0); parity mode: fp32
  const int dtu = rnn_dtu(cfg);
  const bool persist = rnn_persist(cfg);
  const size_t xbytes = persist ? crnn_lstm_persist_xbuf_bytes(T, B, u, dtu) : 0;
  // bf16 U^T: the input projections stream too (64-row stripes of X against a bf16 W^T through LDS, gemm_wgrad.hip) unless the tile schedule is asked for
  const bool xw_stream = dtu == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && TB % 64 == 0 && G % 128 == 0 && d.tds % 64 == 0 && u % 64 == 0;
  {  // U -> U^T for the four recurrences (and W -> W^T for the streamed input projections), one launch
    long in_off[8], out_off[8]; int R[8], Cc[8]; int n = 0;
    const char* names[4] = {"1f", "1b", "2f", "2b"};
    const long esz = (dtu == CRNN_BF16) ? 2 : 4;
    for (const char* nm : names) {
      in_off[n] = c.L.off(std::string("rnn") + nm + "_u");
      // element offset of ut<nm> from ut1f in the destination type (the workspace offsets are in floats)
      out_off[n] = (c.P.off(std::string("ut") + nm) - c.P.off("ut1f")) * 4 / esz;
      R[n] = u; Cc[n] = G; ++n;
    }
    if (xw_stream)
      for (const char* nm : names) {
        in_off[n] = c.L.off(std::string("rnn") + nm + "_w");
        out_off[n] = (c.P.off(std::string("wt") + nm) - c.P.off("ut1f")) * 4 / esz;
        R[n] = nm[0] == '1' ? d.tds : u; Cc[n] = G; ++n;
      }
    CRNN_TRY(crnn_transpose_batch(params, c.w("ut1f"), n, in_off, out_off, R, Cc, dtu, stream));
  }
  auto xw = [&](const float* xin, int din, const char* nm) -> int {
    const std::string s(nm);
    if (xw_stream)
      return crnn_gemm_nt_f32_stream_bias(xin, c.w("wt" + s), nullptr, nullptr, c.w("xw" + s), c.p("rnn" + s + "_b"), TB, G, din, din, din, G, stream);
    return gemm(c, 0, xin, c.p("rnn" + s + "_w"), c.w("xw" + s), TB, G, din, din, G, G, c.p("rnn" + s + "_b"));
  };
  // both directions of a layer in one launch of persistent workgroups (round 5: crnn_rnn_input_proj; bit-identical to the two stripe launches)
  auto xw2 = [&](const float* xin, int din, const char* l) -> int {
    const std::string f = std::string(l) + "f", b = std::string(l) + "b";
    if (xw_stream && crnn_rnn_input_proj_supported(TB, G, din) == CRNN_OK) {
      const int rc = crnn_rnn_input_proj(xin, c.w("wt" + f), c.w("wt" + b), c.p("rnn" + f + "_b"), c.p("rnn" + b + "_b"), c.w("xw" + f), c.w("xw" + b), TB, G, din,
                                         din, din, G, stream);
      if (rc != CRNN_ERR_UNSUPPORTED) return rc;
    }
    CRNN_TRY(xw(xin, din, f.c_str()));
    return xw(xin, din, b.c_str());
  };
  CRNN_TRY(xw2(c.w("dn1"), d.tds, "1"));
  if (cfg->gru && persist)   // "cs" holds r*h_prev for the GRU (cell state for the LSTM)
    CRNN_TRY(crnn_gru_fwd_persist(c.w("xw1f"), c.w("xw1b"), c.w("ut1f"), c.w("ut1b"), c.w("h1f"), c.w("h1b"), u, c.w("gt1f"), c.w("gt1b"),
                                  c.w("cs1f"), c.w("cs1b"), T, B, u, dtu, c.w("rnnx"), xbytes, rnn_uw(cfg), stream));
  else if (cfg->gru)
    CRNN_TRY(crnn_gru_fwd_ex(c.w("xw1f"), c.w("xw1b"), c.w("ut1f"), c.w("ut1b"), c.w("h1f"), c.w("h1b"), u, c.w("gt1f"), c.w("gt1b"),
                             c.w("cs1f"), c.w("cs1b"), T, B, u, dtu, stream));
  else if (persist)
    CRNN_TRY(crnn_lstm_fwd_persist(c.w("xw1f"), c.w("xw1b"), c.w("ut1f"), c.w("ut1b"), c.w("h1f"), c.w("h1b"), u, c.w("cs1f"), c.w("cs1b"),
                                   c.w("gt1f"), c.w("gt1b"), T, B, u, dtu, c.w("rnnx"), xbytes, 0, rnn_uw(cfg), stream));
  else
    CRNN_TRY(crnn_lstm_fwd_ex(c.w("xw1f"), c.w("xw1b"), c.w("ut1f"), c.w("ut1b"), c.w("h1f"), c.w("h1b"), u, c.w("cs1f"), c.w("cs1b"),
                              c.w("gt1f"), c.w("gt1b"), T, B, u, dtu, stream));
  CRNN_TRY(crnn_add(c.w("h1f"), c.w("h1b"), c.w("r1"), (long)TB * u, stream));  // merge_mode='sum'
  CRNN_TRY(xw2(c.w("r1"), u, "2"));
  if (cfg->gru && persist)
    CRNN_TRY(crnn_gru_fwd_persist(c.w("xw2f"), c.w("xw2b"), c.w("ut2f"), c.w("ut2b"), c.w("h2"), c.w("h2") + u, 2 * u, c.w("gt2f"), c.w("gt2b"),
                                  c.w("cs2f"), c.w("cs2b"), T, B, u, dtu, c.w("rnnx"), xbytes, rnn_uw(cfg), stream));
  else if (cfg->gru)
    CRNN_TRY(crnn_gru_fwd_ex(c.w("xw2f"), c.w("xw2b"), c.w("ut2f"), c.w("ut2b"), c.w("h2"), c.w("h2") + u, 2 * u, c.w("gt2f"), c.w("gt2b"),
                             c.w("cs2f"), c.w("cs2b"), T, B, u, dtu, stream));
  else if (persist)
    CRNN_TRY(crnn_lstm_fwd_persist(c.w("xw2f"), c.w("xw2b"), c.w("ut2f"), c.w("ut2b"), c.w("h2"), c.w("h2") + u, 2 * u, c.w("cs2f"), c.w("cs2b"),
                                   c.w("gt2f"), c.w("gt2b"), T, B, u, dtu, c.w("rnnx"), xbytes, 0, rnn_uw(cfg), stream));    // merge_mode='concat'
  else
    CRNN_TRY(crnn_lstm_fwd_ex(c.w("xw2f"), c.w("xw2b"), c.w("ut2f"), c.w("ut2b"), c.w("h2"), c.w("h2") + u, 2 * u, c.w("cs2f"), c.w("cs2b"),
                              c.w("gt2f"), c.w("gt2b"), T, B, u, dtu, stream));    // merge_mode='concat'
  const float* r2 = c.w("h2");
  if (train) {  // Dropout(.2) (utils.py:83)
    int rcd = CRNN_ERR_UNSUPPORTED;
    if (cfg->dropout && dense2_bwd_fused(cfg, d))   // the dropped activations and the decisions as keep bytes in one pass (dense2's backward reads them)
      rcd = crnn_dropout_keep(c.w("h2"), c.w("r2d"), c.w("keep9"), TB, 2 * u, 2 * u, 2 * u, kDropRnn, seed, kLayerRnn, stream);
    if (rcd != CRNN_OK && rcd != CRNN_ERR_UNSUPPORTED) return rcd;
    if (rcd != CRNN_OK) {
      CRNN_TRY(crnn_dropout(c.w("h2"), c.w("r2d"), TB, 2 * u, 2 * u, 2 * u, cfg->dropout ? kDropRnn : 0.f, seed, kLayerRnn, stream));
      if (cfg->dropout && dense2_bwd_fused(cfg, d)) CRNN_TRY(crnn_dropout_keep_bytes(c.w("keep9"), (long)TB * 2 * u / 8, kDropRnn, seed, kLayerRnn, stream));
    }
    r2 = c.w("r2d");
  }
  // ---- dense2 + softmax (utils.py:85-86); back to batch-major [B][T][C]
  bool ypred_out = false;                  // y_pred already written by the fused epilogue
  if (d2T_off >= 0) {
    // round 5: the product on the streaming kernel (64-row stripes against the bf16 W^T through LDS; the tile GEMM ran 104 workgroups for 35 us), bias + row
    // permutation + softmax + the y_pred copy in one pass over its output: dropout aside, four launches (56 us) -> two (18 us)
    const bf16_t* wT = reinterpret_cast<const bf16_t*>(c.w("pwT")) + d2T_off;
    int rc = crnn_gemm_nt_f32_stream_bias(r2, wT, nullptr, nullptr, c.w("lg128"), nullptr, TB, 128, 2 * u, 2 * u, 2 * u, 128, stream);
    if (rc == CRNN_OK) {
      CRNN_TRY(crnn_softmax_rows_perm(c.w("lg128"), 128, c.p("dense2_b"), c.w("logits"), c.w("ypred"), y_pred, TB, d.C, B, stream));
      ypred_out = y_pred != nullptr;
    } else if (rc != CRNN_ERR_UNSUPPORTED) return rc;
    else d2T_off = -1;
  }
  if (d2T_off < 0) {
    CRNN_TRY(gemm(c, 0, r2, c.p("dense2_w"), c.w("logits"), TB, d.C, 2 * u, 2 * u, d.C, d.C, c.p("dense2_b"), 0, 0, B));
    CRNN_TRY(crnn_softmax_rows(c.w("logits"), c.w("ypred"), TB, d.C, stream));
  }
  if (y_pred && y_pred != c.w("ypred") && !ypred_out) {
    hipError_t e = hipMemcpyAsync(y_pred, c.w("ypred"), (size_t)TB * d.C * sizeof(float), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return (int)e;
  }
  return CRNN_OK;
}

// ---------------------------------------------------------------------------------------------------
// One Bidirectional layer's backward in three pieces so that the caller can put the weight-gradient GEMMs on a second
// stream: (1) the BPTT chain (T dependent step launches, latency-bound), (2) dW/dU/db from the finished dz (throughput
// work nobody downstream of the chain waits for), (3) dX = dZ W^T (what the layer below needs).
// the persistent LSTM backward of layer `layer` runs (and leaves the bias-gradient partials "dbp<layer>f/b"): one decision for rnn_bwd_chain and rnn_bwd_wgrads
static bool lstm_bwd_persistent(const Ctx& c, int layer) {     // (LSTM or GRU: whichever cell the configuration has)
  if (!rnn_persist(c.cfg)) return false;
  std::string l = std::to_string(layer);
  int dtu = CRNN_F32;
  const float* uf = c.p("rnn" + l + "f_u"); const float* ub = c.p("rnn" + l + "b_u");
  if (c.cfg->mfma_bf16 && c.d.u % 128 == 0) { uf = weight_operand(c, 0, uf, &dtu); ub = weight_operand(c, 0, ub, &dtu); }
  return !(((uintptr_t)uf | (uintptr_t)ub) & 15) && c.P.off("dbp" + l + "f") >= 0;
}
static int rnn_bwd_chain(const Ctx& c, int layer, const float* hf, const float* hb, int ldh, const float* doutf, const float* doutb, int ldo) {
  const Dims& d = c.d;
  const int T = d.T, B = d.B, u = d.u;
  std::string l = std::to_string(layer);
  float* dzf = c.w("dz" + l + "f"); float* dzb = c.w("dz" + l + "b");
  // bf16 modes: U is read from the bf16 shadow of the parameter buffer (refreshed by the forward)
  int dtu = CRNN_F32;
  const float* uf = c.p("rnn" + l + "f_u"); const float* ub = c.p("rnn" + l + "b_u");
  if (c.cfg->mfma_bf16 && u % 128 == 0) { uf = weight_operand(c, 0, uf, &dtu); ub = weight_operand(c, 0, ub, &dtu); }
  if (c.cfg->gru && lstm_bwd_persistent(c, layer))
    return crnn_gru_bwd_persist_db(uf, ub, hf, hb, ldh, c.w("gt" + l + "f"), c.w("gt" + l + "b"), doutf, doutb, ldo, dzf, dzb, c.w("dbp" + l + "f"),
                                   c.w("dbp" + l + "b"), T, B, u, dtu, c.w("rnnx"), crnn_lstm_persist_xbuf_bytes(T, B, u, dtu), rnn_uw(c.cfg), c.s);
  if (c.cfg->gru && rnn_persist(c.cfg) && !(((uintptr_t)uf | (uintptr_t)ub) & 15))
    return crnn_gru_bwd_persist(uf, ub, hf, hb, ldh, c.w("gt" + l + "f"), c.w("gt" + l + "b"), doutf, doutb, ldo, dzf, dzb, T, B, u, dtu,
                                c.w("rnnx"), crnn_lstm_persist_xbuf_bytes(T, B, u, dtu), rnn_uw(c.cfg), c.s);
  if (c.cfg->gru)
    return crnn_gru_bwd_ex(uf, ub, hf, hb, ldh, c.w("gt" + l + "f"), c.w("gt" + l + "b"), doutf, doutb,
                           ldo, dzf, dzb, c.w("dcf"), c.w("dcb"), c.w("dhpf"), c.w("dhpb"), T, B, u, dtu, c.s);
  if (!c.cfg->gru && lstm_bwd_persistent(c, layer))      // (also leaves the bias-gradient partials: rnn_bwd_wgrads sums them instead of reading dz once more)
    return crnn_lstm_bwd_persist_db(uf, ub, c.w("cs" + l + "f"), c.w("cs" + l + "b"), c.w("gt" + l + "f"), c.w("gt" + l + "b"), doutf, doutb, ldo,
                                    dzf, dzb, c.w("dbp" + l + "f"), c.w("dbp" + l + "b"), T, B, u, dtu, c.w("rnnx"),
                                    crnn_lstm_persist_xbuf_bytes(T, B, u, dtu), 0, rnn_uw(c.cfg), c.s);
  if (rnn_persist(c.cfg) && !(((uintptr_t)uf | (uintptr_t)ub) & 15))
    return crnn_lstm_bwd_persist(uf, ub, c.w("cs" + l + "f"), c.w("cs" + l + "b"), c.w("gt" + l + "f"), c.w("gt" + l + "b"), doutf, doutb, ldo,
                                 dzf, dzb, T, B, u, dtu, c.w("rnnx"), crnn_lstm_persist_xbuf_bytes(T, B, u, dtu), 0, rnn_uw(c.cfg), c.s);
  return crnn_lstm_bwd_ex(uf, ub, c.w("cs" + l + "f"), c.w("cs" + l + "b"), c.w("gt" + l + "f"),
                          c.w("gt" + l + "b"), doutf, doutb, ldo, dzf, dzb, c.w("dcf"), c.w("dcb"), T, B, u, dtu, c.s);
}
// C[M][N] = A^T B over K rows (weight gradients of the recurrent layers): the streaming kernel in the bf16 modes where its shape
// rules hold (gemm_wgrad.hip, fp32 operands rounded to bf16 on the way in like the tile GEMM does), else the tile GEMM
static int gemm_tn(const Ctx& c, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
  if (c.cfg->mfma_bf16 && !(c.cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS)) {
    const size_t need = c.def ? crnn_pwconv_wgrad_stream_scratch_bytes(K, N, M) : 0;
    int rc0 = CRNN_OK;
    if (float* sc = deferred_scratch(c, need, &rc0)) {      // first stage now, the fixed-order sum with the stage's other second stages
      crnn_sum_job job;
      const int rc = crnn_gemm_tn_stream_defer(A, lda, B, ldb, C, ldc, M, N, K, sc, need, &job, c.s);
      if (rc == CRNN_OK) { c.def->jobs.push_back(job); return CRNN_OK; }
      if (rc != CRNN_ERR_UNSUPPORTED) return rc;
    }
    CRNN_TRY(rc0);
    const int rc = crnn_gemm_tn_stream(A, lda, B, ldb, C, ldc, M, N, K, c.scratch(), kGemmScratchBytes, c.s);
    if (rc != CRNN_ERR_UNSUPPORTED) return rc;
  }
  // parity mode, two-plane backward: the pixel-stream form of the two-plane weight gradient (gemm_wgrad3.hip) where its shape rules hold
  if (!c.cfg->mfma_bf16 && conv_planes(c.cfg, true) == 2 && !(c.cfg->flags & (CRNN_FLAG_GEMM_TILE_KERNELS | CRNN_FLAG_F32_MFMA_GEMMS))) {
    const int rc = crnn_gemm_tn_planes_stream(A, lda, B, ldb, C, ldc, M, N, K, c.scratch(), kGemmScratchBytes, c.s);
    if (rc != CRNN_ERR_UNSUPPORTED) return rc;
  }
  return gemm(c, 2, A, B, C, M, N, K, lda, ldb, ldc, nullptr, 0, 0, 0, conv_planes(c.cfg, true));
}
static int rnn_bwd_wgrads(const Ctx& c, int layer, const float* xin, int ldx, int din, const float* hf, const float* hb, int ldh) {
  const Dims& d = c.d;
  const int T = d.T, B = d.B, TB = T * B, u = d.u, G = d.G;
  std::string l = std::to_string(layer);
  float* dzf = c.w("dz" + l + "f"); float* dzb = c.w("dz" + l + "b");
  // dW = X^T dZ ; dU = Hprev^T dZ ; db = colsum(dZ)
  CRNN_TRY(gemm_tn(c, xin, dzf, c.g("rnn" + l + "f_w"), din, G, TB, ldx, G, G));
  CRNN_TRY(gemm_tn(c, xin, dzb, c.g("rnn" + l + "b_w"), din, G, TB, ldx, G, G));
  const int K1 = (T - 1) * B;
  // forward direction: h_{t-1} pairs with dz_t ; backward direction: h_{t+1} pairs with dz_t
  const int Nh = c.cfg->gru ? 2 * u : G;   // GRU: only the z,r columns see h_prev; the candidate sees r*h_prev
  CRNN_TRY(gemm_tn(c, hf, dzf + (long)B * G, c.g("rnn" + l + "f_u"), u, Nh, K1, ldh, G, G));
  CRNN_TRY(gemm_tn(c, hb + (long)B * ldh, dzb, c.g("rnn" + l + "b_u"), u, Nh, K1, ldh, G, G));
  if (c.cfg->gru) {
    CRNN_TRY(gemm(c, 2, c.w("cs" + l + "f"), dzf + 2 * u, c.g("rnn" + l + "f_u") + 2 * u, u, u, TB, u, G, G, nullptr, 0, 0, 0, conv_planes(c.cfg, true)));
    CRNN_TRY(gemm(c, 2, c.w("cs" + l + "b"), dzb + 2 * u, c.g("rnn" + l + "b_u") + 2 * u, u, u, TB, u, G, G, nullptr, 0, 0, 0, conv_planes(c.cfg, true)));
  }
  if (lstm_bwd_persistent(c, layer)) {    // the persistent backward summed dz over time per 16-row batch tile: the tiles in a fixed order
    CRNN_TRY(crnn_partials_sum(c.w("dbp" + l + "f"), crnn_rnn_db_rows(B), G, c.g("rnn" + l + "f_b"), 1.f, c.s));
    return crnn_partials_sum(c.w("dbp" + l + "b"), crnn_rnn_db_rows(B), G, c.g("rnn" + l + "b_b"), 1.f, c.s);
  }
  CRNN_TRY(colsum(c, dzf, TB, G, G, c.g("rnn" + l + "f_b")));
  return colsum(c, dzb, TB, G, G, c.g("rnn" + l + "b_b"));
}
static int rnn_bwd_dx(const Ctx& c, int layer, int din, float* dxin) {
  const Dims& d = c.d;
  const int TB = d.T * d.B, G = d.G;
  std::string l = std::to_string(layer);
  if (c.cfg->mfma_bf16 && !(c.cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS)) {   // both directions in one streaming launch (gemm_wgrad.hip)
    int dtf = CRNN_F32, dtb = CRNN_F32;
    const float* wf = weight_operand(c, 1, c.p("rnn" + l + "f_w"), &dtf);
    const float* wb = weight_operand(c, 1, c.p("rnn" + l + "b_w"), &dtb);
    if (dtf == CRNN_BF16 && dtb == CRNN_BF16) {
      const int rc = crnn_gemm_nt_f32_stream(c.w("dz" + l + "f"), wf, c.w("dz" + l + "b"), wb, dxin, TB, din, G, G, G, din, c.s);
      if (rc != CRNN_ERR_UNSUPPORTED) return rc;
    }
  }
  // parity mode, two-plane backward: both directions in one stripe-stream launch (gemm_wgrad3.hip)
  if (!c.cfg->mfma_bf16 && conv_planes(c.cfg, true) == 2 && !(c.cfg->flags & (CRNN_FLAG_GEMM_TILE_KERNELS | CRNN_FLAG_F32_MFMA_GEMMS))) {
    const int rc = crnn_gemm_nt_f32x2_stream(c.w("dz" + l + "f"), c.p("rnn" + l + "f_w"), c.w("dz" + l + "b"), c.p("rnn" + l + "b_w"), dxin, TB, din, G, G, G, din, c.s);
    if (rc != CRNN_ERR_UNSUPPORTED) return rc;
  }
  CRNN_TRY(gemm(c, 1, c.w("dz" + l + "f"), c.p("rnn" + l + "f_w"), dxin, TB, din, G, G, G, din, nullptr, 0, 0, 0, conv_planes(c.cfg, true)));
  return gemm(c, 1, c.w("dz" + l + "b"), c.p("rnn" + l + "b_w"), dxin, TB, din, G, G, G, din, nullptr, 0, 1, 0, conv_planes(c.cfg, true));
}

// The backward runs in two stages so that a data-parallel host can start the gradient all-reduce of the upper
// layers (dense1, the recurrent layers, dense2: the tail of the flat buffer, 75 % of its bytes) while the
// conv-stack / STN stage is still running.
namespace {
int backward_top(const Ctx& c, const int* labels, const int* input_length, const int* label_length, float* loss, uint64_t seed, hipStream_t aux);
int backward_bottom(const Ctx& c, const float* x, uint64_t seed, hipStream_t aux);
}
extern "C" long crnn_grad_split_offset(const crnn_config* cfg) { return make_layout(cfg).off("dense1_w"); }
extern "C" int crnn_backward_top_ex(const crnn_config* cfg, const float* params, float* grads, const int* labels,
                                    const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss,
                                    uint64_t seed, hipStream_t stream, hipStream_t aux_stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, grads, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  return backward_top(c, labels, input_length, label_length, loss, seed, side_stream(stream, aux_stream));
}
extern "C" int crnn_backward_top(const crnn_config* cfg, const float* params, float* grads, const int* labels,
                                 const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss,
                                 uint64_t seed, hipStream_t stream) {
  return crnn_backward_top_ex(cfg, params, grads, labels, input_length, label_length, ws, ws_bytes, loss, seed, stream, nullptr);
}
extern "C" int crnn_backward_bottom(const crnn_config* cfg, const float* params, float* grads, const float* x, float* ws,
                                    size_t ws_bytes, uint64_t seed, hipStream_t stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, grads, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  return backward_bottom(c, x, seed, nullptr);
}
extern "C" int crnn_backward_bottom_ex(const crnn_config* cfg, const float* params, float* grads, const float* x, float* ws,
                                       size_t ws_bytes, uint64_t seed, hipStream_t stream, hipStream_t aux_stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, grads, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  return backward_bottom(c, x, seed, side_stream(stream, aux_stream));
}
extern "C" int crnn_backward(const crnn_config* cfg, const float* params, float* grads, const float* x, const int* labels,
                             const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss,
                             uint64_t seed, hipStream_t stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, grads, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  CRNN_TRY(backward_top(c, labels, input_length, label_length, loss, seed, nullptr));
  return backward_bottom(c, x, seed, nullptr);
}
extern "C" int crnn_backward_ex(const crnn_config* cfg, const float* params, float* grads, const float* x, const int* labels,
                                const int* input_length, const int* label_length, float* ws, size_t ws_bytes, float* loss,
                                uint64_t seed, hipStream_t stream, hipStream_t aux_stream) {
  CRNN_TRY(check_cfg(cfg));
  Ctx c{cfg, make_dims(cfg), make_layout(cfg), make_plan(cfg), params, grads, ws, stream};
  if (ws_bytes < (size_t)c.P.total * sizeof(float)) return CRNN_ERR_ARG;
  CRNN_TRY(backward_top(c, labels, input_length, label_length, loss, seed, side_stream(stream, aux_stream)));
  return backward_bottom(c, x, seed, side_stream(stream, aux_stream));
}

namespace {
// aux != nullptr: a second stream takes the weight-gradient GEMMs that nothing on the critical path waits for (dense2's
// during the layer-2 BPTT chain, layer 2's during the layer-1 chain).  The chains are T dependent launches of a few
// microseconds each and leave the GPU almost idle; the GEMMs fill it.  Only the aux stream touches the split-reduction
// scratch and the reduction partials between the fork and the join, and every gradient tensor still has a single
// writer in a fixed order, so the result is bit-identical to the serial schedule.
// deferred second stages: serial schedule only (a side stream's first stages could not share one flush), bf16 modes with the streaming kernels
void deferred_setup(const Ctx& c0, Ctx& c, Deferred& def, hipStream_t aux) {
  c = c0;
  const long off = c0.P.off("wgrad_scratch");
  if (aux || off < 0) return;
  def.base = c0.ws + off; def.cap = (size_t)(c0.P.cnt("wgrad_scratch") - 64) * sizeof(float);
  c.def = &def;
}

int backward_top(const Ctx& c0, const int* labels, const int* input_length, const int* label_length, float* loss, uint64_t seed, hipStream_t aux) {
  Deferred def; Ctx c; deferred_setup(c0, c, def, aux);
  const crnn_config* cfg = c.cfg; float* grads = c.grads; hipStream_t stream = c.s;
  const Dims& d = c.d;
  const int B = d.B, T = d.T, TB = T * B, u = d.u;
  ForkJoin fj(stream, aux);
  Ctx ca = c; if (aux) { ca.s = aux; ca.side = true; }   // same tensors, side stream, its own reduction scratch
  hipError_t e = hipMemsetAsync(grads, 0, (size_t)c.L.total * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  // ---- CTC (utils.py:98-103): loss per sample + d mean(loss)/d logits (time-major)
  CRNN_TRY(crnn_ctc_loss_grad(c.w("ypred"), labels, input_length, label_length, loss, c.w("dlogits"), B, T, d.C, d.L, 2, 1.0f / (float)B, stream));
  // ---- dense2: weight / bias gradients on the side stream, the data gradient feeds the recurrent layers
  const float* r2 = c.w("r2d");
  int rc2 = CRNN_ERR_UNSUPPORTED;
  if (dense2_bwd_fused(cfg, d))   // one pass over r2 / dlogits: dW, db, and the data gradient with the forward's dropout multiplier (dense.hip)
    rc2 = crnn_dense_bwd_small(r2, c.w("dlogits"), c.p("dense2_w"), c.w("dr2"), c.g("dense2_w"), c.g("dense2_b"), c.w("d2part"),
                               (size_t)c.P.cnt("d2part") * sizeof(float), TB, 2 * u, d.C, 2 * u, 2 * u, cfg->dropout ? c.w("keep9") : nullptr,
                               cfg->dropout ? kDropRnn : 0.f, seed, kLayerRnn, stream);
  if (rc2 != CRNN_OK && rc2 != CRNN_ERR_UNSUPPORTED) return rc2;
  CRNN_TRY(fj.fork());
  if (rc2 != CRNN_OK) {
    CRNN_TRY(gemm(ca, 2, r2, c.w("dlogits"), c.g("dense2_w"), 2 * u, d.C, TB, 2 * u, d.C, d.C, nullptr, 0, 0, 0, conv_planes(cfg, true)));
    CRNN_TRY(colsum(ca, c.w("dlogits"), TB, d.C, d.C, c.g("dense2_b")));
    CRNN_TRY(gemm(c, 1, c.w("dlogits"), c.p("dense2_w"), c.w("dr2"), TB, 2 * u, d.C, d.C, d.C, 2 * u, nullptr, 0, 0, 0, conv_planes(cfg, true)));
    if (cfg->dropout) CRNN_TRY(crnn_dropout(c.w("dr2"), c.w("dr2"), TB, 2 * u, 2 * u, 2 * u, kDropRnn, seed, kLayerRnn, stream));
  }
  // ---- Bidirectional LSTM / GRU x2
  CRNN_TRY(rnn_bwd_chain(c, 2, c.w("h2"), c.w("h2") + u, 2 * u, c.w("dr2"), c.w("dr2") + u, 2 * u));
  CRNN_TRY(fj.fork());                               // dz of layer 2 is complete: its weight gradients go to the side stream
  CRNN_TRY(rnn_bwd_wgrads(ca, 2, c.w("r1"), u, u, c.w("h2"), c.w("h2") + u, 2 * u));
  CRNN_TRY(rnn_bwd_dx(c, 2, u, c.w("dr1")));
  CRNN_TRY(rnn_bwd_chain(c, 1, c.w("h1f"), c.w("h1b"), u, c.w("dr1"), c.w("dr1"), u));
  CRNN_TRY(fj.join());                               // the scratch buffers are the main stream's again
  CRNN_TRY(rnn_bwd_wgrads(c, 1, c.w("dn1"), d.tds, d.tds, c.w("h1f"), c.w("h1b"), u));
  CRNN_TRY(rnn_bwd_dx(c, 1, d.tds, c.w("ddn1")));
  // ---- Dropout(.4) + relu of dense1, rows back to batch-major
  const bool wres1 = dense1_dgrad_wres(cfg, d);
  CRNN_TRY(crnn_relu_bwd_ex(c.w("dn1"), c.w("ddn1"), c.w("gbm"), wres1 ? c.w("gbm16") : nullptr, TB, d.tds, cfg->dropout ? 1.0f / (1.0f - kDropDense1) : 1.0f, B, stream));
  const float* feat = c.w("x7");
  int rcw = CRNN_ERR_UNSUPPORTED;
  if (wres1 && c.dt("x7") == CRNN_BF16)   // both operands bf16: the pixel-streaming weight-gradient kernel, 36 feature tiles x 7 row ranges (gemm_wgrad.hip)
    rcw = crnn_gemm_tn_bf16_stream(feat, d.feat, c.w("gbm16"), d.tds, c.g("dense1_w"), d.tds, d.feat, d.tds, TB, c.scratch(), kGemmScratchBytes, stream);
  if (rcw != CRNN_OK && rcw != CRNN_ERR_UNSUPPORTED) return rcw;
  // parity mode, two-plane backward: the pixel-stream form (gemm_wgrad3.hip; 36 feature tiles x 7 row ranges)
  if (rcw != CRNN_OK && !cfg->mfma_bf16 && c.dt("x7") == CRNN_F32 && conv_planes(cfg, true) == 2 && !(cfg->flags & (CRNN_FLAG_GEMM_TILE_KERNELS | CRNN_FLAG_F32_MFMA_GEMMS))) {
    rcw = crnn_gemm_tn_planes_stream(feat, d.feat, c.w("gbm"), d.tds, c.g("dense1_w"), d.tds, d.feat, d.tds, TB, c.scratch(), kGemmScratchBytes, stream);
    if (rcw != CRNN_OK && rcw != CRNN_ERR_UNSUPPORTED) return rcw;
  }
  if (rcw != CRNN_OK)
    CRNN_TRY(gemm_t(c, 2, feat, c.dt("x7"), c.w("gbm"), CRNN_F32, c.g("dense1_w"), CRNN_F32, d.feat, d.tds, TB, d.feat, d.tds, d.tds, nullptr, 0, 0, 0, conv_planes(cfg, true)));
  CRNN_TRY(colsum(c, c.w("gbm"), TB, d.tds, d.tds, c.g("dense1_b")));
  float* gA = c.w("gA"); float* gB = c.w("gB");
  int rc1 = CRNN_ERR_UNSUPPORTED;
  if (wres1) rc1 = crnn_gemm_wres_bf16(c.w("gbm16"), reinterpret_cast<const bf16_t*>(c.w("pbf")) + c.L.off("dense1_w"), gA, TB, d.feat, d.tds, stream);
  if (rc1 != CRNN_OK && rc1 != CRNN_ERR_UNSUPPORTED) return rc1;
  if (rc1 != CRNN_OK)
    CRNN_TRY(gemm_t(c, 1, c.w("gbm"), CRNN_F32, c.p("dense1_w"), CRNN_F32, gA, c.gdt(), TB, d.feat, d.tds, d.tds, d.tds, d.feat, nullptr, 0, 0, 0, conv_planes(cfg, true)));
  return flush_deferred(c);                             // every gradient of this stage is final (a data-parallel host exchanges them now)
}

// aux != nullptr: the pointwise weight-gradient GEMM of every block runs on the side stream next to the rest of the block's
// backward (data-gradient GEMM, BatchNorm statistics pass, fused depthwise stage -- HBM- and VALU-bound kernels that leave the
// matrix cores and most of the vector-memory path idle).  The GEMM reads the BatchNorm-2 input gradient of its block, so the
// gradient buffers rotate over three allocations and the main stream waits for GEMM i before the buffer it reads is written
// again (by the depthwise stage of block i-1).  One writer per gradient tensor, fixed order: bit-identical to the serial schedule.
int backward_bottom(const Ctx& c0, const float* x, uint64_t seed, hipStream_t aux) {
  Deferred def; Ctx c; deferred_setup(c0, c, def, aux);
  const crnn_config* cfg = c.cfg; hipStream_t stream = c.s;
  const Dims& d = c.d;
  const int B = d.B;
  ForkJoin fj(stream, aux);
  Ctx ca = c; if (aux) { ca.s = aux; ca.side = true; }
  float* gA = c.w("gA"); float* gB = c.w("gB"); float* gC = c.w("gC");   // gA holds d loss / d x7 (written by backward_top)
  hipEvent_t gB_free = nullptr, gC_free = nullptr;     // side-stream GEMMs still reading gB / gC (null: none)
  int bn2_stats_rows = 0;                              // > 0: block i+1's depthwise-stage backward left the statistics of block i's BatchNorm-2 backward in "bn2parts"
  // ---- conv stack
  for (int i = 7; i >= 1; --i) {
    std::string p = std::to_string(i), bp = "b" + p;
    const int H = d.bh[i], W = d.bw[i], ci = d.bc[i - 1], co = d.bc[i];
    const long M = (long)B * H * W;
    const int dtd = c.dt("d" + p), dtq = c.dt("q" + p);   // dtq == storage of the incoming gradient (gdt)
    const bool fused_bf = i > 1 && dtd == CRNN_BF16 && c.dt("x" + std::to_string(i - 1)) == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_NO_DW_BWD_FUSION) &&
                          crnn_dwconv_bwd_fused_supported(H, W, ci) == CRNN_OK;
    // fp32 tensors (parity mode, round 4): the row-stream kernel's fp32 form where its shape rule holds (no halo-tile form: else the three-kernel sequence)
    const bool fused_f32 = i > 1 && dtd == CRNN_F32 && dtq == CRNN_F32 && c.dt("x" + std::to_string(i - 1)) == CRNN_F32 &&
                           !(cfg->flags & (CRNN_FLAG_NO_DW_BWD_FUSION | CRNN_FLAG_DW_TILE_KERNEL)) &&
                           crnn_dwconv_bwd_stream_supported_ex(B, H, W, ci, CRNN_F32) == CRNN_OK &&
                           aligned16(c.w("d" + p), c.w("x" + std::to_string(i - 1)), c.p(bp + "_dw"), c.w("coef")) && aligned16(gA, gB, gC, c.w("bn1s" + p));
    const bool fused_dw = fused_bf || fused_f32;
    int bn1_stats_rows = 0;                               // > 0: the data-gradient GEMM left the BatchNorm-1 backward statistics in `partials`
    // Parity mode, two-plane backward (round 6): dq -- BatchNorm-2's input gradient, read only by this block's two pointwise GEMMs -- is WRITTEN as its two bf16
    // planes (the bytes of the fp32 tensor) by the BatchNorm backward, so that neither GEMM splits it: the data gradient runs from the planes by LDS-DMA
    // with the weight planes resident (gemm_pres.hip), the weight gradient's IO waves copy them (gemm_wgrad3.hip).  Same words, same products.
    const bool dq_planes = !fused_bf && pw_products(cfg) == 2 && conv_planes(cfg, true) == 2 && dtq == CRNN_F32 && dtd == CRNN_F32 && fuse_dw_bn_x3(cfg, dtd, dtq, ci) &&
                           !(cfg->flags & (CRNN_FLAG_GEMM_TILE_KERNELS | CRNN_FLAG_NO_BN_STATS_FUSION | CRNN_FLAG_NO_GRADIENT_PLANES | CRNN_FLAG_WEIGHT_PLANES)) &&
                           (kBlocks[i - 1].ph * kBlocks[i - 1].pw == 1 || crnn_knob("CRNN_DQPL_POOL", 1)) &&
                           crnn_gemm_pres_supported(M, ci, co, 2) == CRNN_OK && crnn_pwconv_wgrad_planes_stream_supported(M, co, ci) == CRNN_OK &&
                           crnn_pwconv_wgrad_planes_stream_scratch_bytes(M, co, ci) <= kGemmScratchBytes &&
                           aligned16(c.w("d" + p), c.w("q" + p), c.w("bn1s" + p), c.p(bp + "_pw")) && aligned16(gA, gB, gC, c.w("partials"));
    CRNN_TRY(fj.wait(gB_free)); gB_free = nullptr;        // gB is written next
    if (bn2_stats_rows > 0) {   // the depthwise-stage backward of block i+1 took this BatchNorm's statistics pass: finalize, then pass 2 alone
      CRNN_TRY(crnn_bn_bwd_finalize(c.w("bn2parts"), bn2_stats_rows, co, M, c.g(bp + "_bn2_g"), c.g(bp + "_bn2_b"), c.w("coef"), stream));
      if (dq_planes)
        CRNN_TRY(crnn_bn_bwd_apply_planes_ex(c.w("q" + p), gA, c.w("bn2s" + p), c.w("coef"), gB, M * co, 2, B, H, W, co, kBlocks[i - 1].ph, kBlocks[i - 1].pw,
                                             cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, stream));
      else
      CRNN_TRY(crnn_bn_bwd_apply_ex(c.w("q" + p), gA, c.w("bn2s" + p), c.w("coef"), gB, B, H, W, co, kBlocks[i - 1].ph, kBlocks[i - 1].pw,
                                    cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, dtq, stream));
      bn2_stats_rows = 0;
    } else if (use_qmax(c, i))   // pooled block: the statistics pass reads the forward's arg-max values
      CRNN_TRY(crnn_bn_bwd_qmax_ex(c.w("q" + p), c.w("qm" + p), gA, c.w("bn2s" + p), c.p(bp + "_bn2_g"), dq_planes ? nullptr : gB, dq_planes ? gB : nullptr, M * co,
                                   dq_planes ? 2 : 0, c.g(bp + "_bn2_g"), c.g(bp + "_bn2_b"), c.w("partials"), c.w("coef"), B, H, W, co, kBlocks[i - 1].ph,
                                   kBlocks[i - 1].pw, cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, dtq, stream));
    else if (dq_planes)
      CRNN_TRY(crnn_bn_bwd_planes_ex(c.w("q" + p), gA, c.w("bn2s" + p), c.p(bp + "_bn2_g"), gB, M * co, 2, c.g(bp + "_bn2_g"), c.g(bp + "_bn2_b"), c.w("partials"),
                                     c.w("coef"), B, H, W, co, kBlocks[i - 1].ph, kBlocks[i - 1].pw, cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, stream));
    else
    CRNN_TRY(crnn_bn_bwd_ex(c.w("q" + p), gA, c.w("bn2s" + p), c.p(bp + "_bn2_g"), gB, c.g(bp + "_bn2_g"), c.g(bp + "_bn2_b"), c.w("partials"),
                            c.w("coef"), B, H, W, co, kBlocks[i - 1].ph, kBlocks[i - 1].pw, cfg->dropout ? kDropBlock : 0.f, seed, (uint32_t)i, dtq, stream));
    if (block1_fused(cfg, ci, dtd)) {   // block 1: weight gradient, data gradient and BatchNorm-1's backward statistics from one pass over dq
      const int rows = crnn_pw1_bn_bwd_rows(M);
      float* bnp = c.w("partials");       // [rows][2] for the finalize below; the weight-gradient partials behind them
      CRNN_TRY(crnn_pw1_bn_bwd(c.w("d" + p), c.w("bn1s" + p), c.p(bp + "_pw"), gB, gA, c.g(bp + "_pw"), bnp + ((2L * rows + 63) & ~63L), bnp, M, co, dtq, stream));
      bn1_stats_rows = rows;
    } else if (ci == 1 && dtd == CRNN_F32) {   // block 1: outer-product weight / data gradients
      CRNN_TRY(crnn_pw1_bwd(c.w("a" + p), c.p(bp + "_pw"), gB, gA, c.g(bp + "_pw"), c.w("partials"), M, co, dtq, stream));
    } else {
      const bool side = fj.on && fused_dw;
      const Ctx& cw = side ? ca : c;
      if (side) CRNN_TRY(fj.fork());
      if (fuse_dw_bn(cfg, dtd, dtq, ci)) {  // the activated tensor was never written: re-form it from d while staging (as the forward did)
        int rc = CRNN_ERR_UNSUPPORTED;        // pixel-streaming kernel (gemm_wgrad.hip) where its shape rules hold, else the tile GEMM
        if (!(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS) && crnn_pwconv_wgrad_stream_supported(M, co, ci) == CRNN_OK) {
          const size_t need = cw.def ? crnn_pwconv_wgrad_stream_scratch_bytes(M, co, ci) : 0;
          int rc0 = CRNN_OK;
          if (float* sc = deferred_scratch(cw, need, &rc0)) {
            crnn_sum_job job;
            rc = crnn_pwconv_bnrelu6_wgrad_stream_defer(c.w("d" + p), c.w("bn1s" + p), gB, c.g(bp + "_pw"), M, co, ci, sc, need, &job, cw.s);
            if (rc == CRNN_OK) cw.def->jobs.push_back(job);
          }
          CRNN_TRY(rc0);
          if (rc == CRNN_ERR_UNSUPPORTED)
            rc = crnn_pwconv_bnrelu6_wgrad_stream(c.w("d" + p), c.w("bn1s" + p), gB, c.g(bp + "_pw"), M, co, ci, cw.scratch(), kGemmScratchBytes, cw.s);
        }
        if (rc == CRNN_ERR_UNSUPPORTED)
          rc = crnn_pwconv_bnrelu6_wgrad(c.w("d" + p), c.w("bn1s" + p), gB, c.g(bp + "_pw"), M, co, ci, cw.scratch(), kGemmScratchBytes, cw.s);
        CRNN_TRY(rc);
      }
      else if (fuse_dw_bn_x3(cfg, dtd, dtq, ci) && aligned16(c.w("d" + p), c.w("q" + p), c.w("bn1s" + p), c.p(bp + "_pw"))) {   // parity mode: likewise (the forward's own predicate)
        int rc = CRNN_ERR_UNSUPPORTED;
        // round 6: two-plane operands on the pixel stream (gemm_wgrad3.hip): one workgroup per 128 x 128 tile and pixel range, the tile in registers; dq from
        // its planes (no fallback: the planes are what gB holds -- the predicate above is the kernel's own rule)
        if (dq_planes) { CRNN_TRY(crnn_pwconv_bnrelu6_wgrad_planes_stream_gp(c.w("d" + p), c.w("bn1s" + p), gB, M * co, c.g(bp + "_pw"), M, co, ci, cw.scratch(), kGemmScratchBytes, cw.s)); rc = CRNN_OK; }
        if (rc == CRNN_ERR_UNSUPPORTED)
          rc = (conv_planes(cfg, true) == 2 ? crnn_pwconv_bnrelu6_wgrad_f32x2 : crnn_pwconv_bnrelu6_wgrad_f32x3)(
              c.w("d" + p), c.w("bn1s" + p), gB, c.g(bp + "_pw"), M, co, ci, cw.scratch(), kGemmScratchBytes, cw.s);
        CRNN_TRY(rc);
      }
      else CRNN_TRY(gemm_t(cw, 2, c.w("a" + p), dtd, gB, dtq, c.g(bp + "_pw"), CRNN_F32, ci, co, (int)M, ci, co, co, nullptr, 0, 0, 0, conv_planes(cfg, true)));
      if (side) CRNN_TRY(fj.mark(&gB_free));
      // data gradient da[M][ci] = dq[M][co] . W[ci][co]^T: the persistent LDS-DMA kernels where their shape rules hold
      int rc = CRNN_ERR_UNSUPPORTED;
      if (dq_planes) {   // (no fallback either)
        CRNN_TRY(crnn_gemm_pres_bnstats(gB, M * co, c.p(bp + "_pw"), gA, M, ci, co, 2, c.w("d" + p), c.w("bn1s" + p), c.w("partials"), nullptr, 0, 0, stream));
        bn1_stats_rows = crnn_gemm_pres_stat_rows(M, ci, co, 2);
        rc = CRNN_OK;
      }
      if (cfg->mfma_bf16 == 2 && dtq == CRNN_BF16 && dtd == CRNN_BF16 && !(cfg->flags & CRNN_FLAG_GEMM_TILE_KERNELS)) {
        int dtw = CRNN_F32;
        const float* wsh = weight_operand(c, 1, c.p(bp + "_pw"), &dtw);
        if (dtw == CRNN_BF16) {
          // weights resident in registers; where the fused depthwise stage follows, the storer waves also take the statistics pass of
          // the depthwise BatchNorm's backward (they hold the finished da stripe: the stand-alone pass would read da and d again)
          if (fused_dw && !(cfg->flags & CRNN_FLAG_NO_BN_STATS_FUSION) && crnn_gemm_wres_bnstats_supported(M, ci, co) == CRNN_OK) {
            rc = crnn_gemm_wres_bf16_bnstats(gB, wsh, gA, M, ci, co, c.w("d" + p), c.w("bn1s" + p), c.w("partials"), stream);
            bn1_stats_rows = (rc == CRNN_OK) ? crnn_gemm_wres_bnstats_rows(M, ci, co) : 0;
          }
          if (rc == CRNN_ERR_UNSUPPORTED) rc = crnn_gemm_wres_bf16(gB, wsh, gA, (int)M, ci, co, stream);
          if (rc == CRNN_ERR_UNSUPPORTED) rc = crnn_gemm_nt_bf16(gB, wsh, gA, (int)M, ci, co, stream);
        }
      }
      // parity mode: the three-plane GEMM's epilogue takes the statistics pass of the depthwise BatchNorm's backward (it holds the finished da tile)
      if (rc == CRNN_ERR_UNSUPPORTED && !fused_bf && pw_products(cfg) == 2 && dtq == CRNN_F32 && dtd == CRNN_F32 &&
          !(cfg->flags & CRNN_FLAG_NO_BN_STATS_FUSION) && crnn_gemm_f32x3_bnstats_supported(M, ci, co) == CRNN_OK) {
        long wps = 0; const void* wpl = weight_planes(c, c.p(bp + "_pw"), &wps, true);
        int w3_rows = 0;
        if (wres3_on(cfg, co) && !wpl && crnn_gemm_wres3_supported(M, ci, co) == CRNN_OK) {   // round 6: W^T planes resident, dq streamed once (gemm_wres3.hip)
          rc = crnn_gemm_wres3_bnstats(gB, c.p(bp + "_pw"), gA, M, ci, co, conv_planes(cfg, true), c.w("d" + p), c.w("bn1s" + p), c.w("partials"), stream);
          if (rc == CRNN_OK) { w3_rows = crnn_gemm_wres3_stat_rows(M, ci, co); }
        }
        if (rc != CRNN_ERR_UNSUPPORTED) {}
        else if (conv_planes(cfg, true) == 2) rc = crnn_gemm_f32x2_bnstats(gB, c.p(bp + "_pw"), gA, M, ci, co, c.w("d" + p), c.w("bn1s" + p), c.w("partials"), stream);
        else rc = crnn_gemm_f32x3_bnstats_pl(gB, nullptr, 0, c.p(bp + "_pw"), wpl, wps, gA, M, ci, co, c.w("d" + p), c.w("bn1s" + p), c.w("partials"), stream);
        if (rc == CRNN_ERR_UNSUPPORTED && wpl) rc = crnn_gemm_f32x3_bnstats(gB, c.p(bp + "_pw"), gA, M, ci, co, c.w("d" + p), c.w("bn1s" + p), c.w("partials"), stream);
        bn1_stats_rows = (rc == CRNN_OK) ? (w3_rows ? w3_rows : crnn_gemm_f32x3_bnstats_rows(M)) : 0;
      }
      if (rc == CRNN_ERR_UNSUPPORTED) rc = gemm_t(c, 1, gB, dtq, c.p(bp + "_pw"), CRNN_F32, gA, dtd, (int)M, ci, co, co, co, ci, nullptr, 0, 0, 0, conv_planes(cfg, true));
      CRNN_TRY(rc);
    }
    const float* xin = (i == 1) ? c.w("x0") : c.w("x" + std::to_string(i - 1));
    if (fused_dw) {
      // depthwise stage in one kernel: BatchNorm statistics pass, then BN-backward pass 2 + depthwise weight and data gradients together
      if (bn1_stats_rows > 0 && fused_f32)     // (one row per GEMM tile row: folded first)
        CRNN_TRY(crnn_bn_bwd_finalize_folded(c.w("partials"), bn1_stats_rows, ci, M, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("coef"), c.w("fold"), stream));
      else if (bn1_stats_rows > 0)
        CRNN_TRY(crnn_bn_bwd_finalize(c.w("partials"), bn1_stats_rows, ci, M, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("coef"), stream));
      else
        CRNN_TRY(crnn_bn_bwd_ex(c.w("d" + p), gA, c.w("bn1s" + p), c.p(bp + "_bn1_g"), nullptr, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("partials"),
                                c.w("coef"), B, H, W, ci, 1, 1, 0.f, 0, 0, dtd, stream));
      CRNN_TRY(fj.wait(gC_free)); gC_free = nullptr;      // gC is written next
      int rc = CRNN_ERR_UNSUPPORTED;
      if (fuse_bn2_dw(c, i - 1)) {                           // the forward did not keep x_{i-1}: re-formed from q_{i-1} in LDS (no fallback: same decision)
        const std::string pp = std::to_string(i - 1);
        // (its dropout decisions: the keep bytes the forward of this step left in the workspace -- same seed)
        float* st2 = bn2_stats_fusion_block(cfg, i - 1) ? c.w("bn2parts") : nullptr;   // (bf16 tensors: opt-in, measured neutral -- include/crnn_mi355x.h)
        CRNN_TRY(crnn_dwconv3x3_bwd_stream_pro_ex(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), pro_src(c, i - 1), c.w("bn2s" + pp), cfg->dropout ? kDropBlock : 0.f,
                                                  keep_bytes(c, i - 1), c.p(bp + "_dw"), gC, c.g(bp + "_dw"), c.w("partials"), st2, B, H, W, ci, dtd, stream));
        if (st2) bn2_stats_rows = crnn_dwconv_bwd_stream_rows_ex(B, H, W, ci, dtd);
        rc = CRNN_OK;
      } else if (fused_f32) {                                   // (no fallback: the predicate above is the kernel's own rule)
        CRNN_TRY(crnn_dwconv3x3_bwd_stream_ex(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), xin, c.p(bp + "_dw"), gC, c.g(bp + "_dw"), c.w("partials"),
                                              B, H, W, ci, CRNN_F32, stream));
        rc = CRNN_OK;
      } else if (!(cfg->flags & CRNN_FLAG_DW_TILE_KERNEL))         // rows streamed through LDS where the shape rule holds (dwconv_bwd_stream.hip)
        rc = crnn_dwconv3x3_bwd_stream(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), xin, c.p(bp + "_dw"), gC, c.g(bp + "_dw"), c.w("partials"),
                                       B, H, W, ci, stream);
      if (rc == CRNN_ERR_UNSUPPORTED)
        rc = crnn_dwconv3x3_bwd_fused(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), xin, c.p(bp + "_dw"), gC, c.g(bp + "_dw"), c.w("partials"),
                                      B, H, W, ci, stream);
      CRNN_TRY(rc);
      // the block below finds its incoming gradient in gA, writes gB; this block's side-stream GEMM may still read the old gB
      float* t = gA; gA = gC; gC = gB; gB = t;
      gC_free = gB_free; gB_free = nullptr;
      continue;
    }
    if (bn1_stats_rows > 0 && block1_fused(cfg, ci, dtd)) {   // block 1: finalize, then pass 2 + both depthwise gradients in one kernel
      CRNN_TRY(crnn_bn_bwd_finalize_folded(c.w("partials"), bn1_stats_rows, ci, M, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("coef"), c.w("fold"), stream));
      CRNN_TRY(crnn_dwconv3x3_c1_bwd(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), xin, c.p(bp + "_dw"), (i > 1 || cfg->stn) ? gB : nullptr, c.g(bp + "_dw"),
                                     c.w("partials"), B, H, W, stream));
      if (i > 1 || cfg->stn) { float* t = gA; gA = gB; gB = t; }   // (the spatial transformer's backward finds d loss / d x0 in gA)
      continue;
    }
    if (bn1_stats_rows > 0) {   // statistics from the data-gradient GEMM: finalize, then pass 2 alone
      CRNN_TRY(crnn_bn_bwd_finalize_folded(c.w("partials"), bn1_stats_rows, ci, M, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("coef"), c.w("fold"), stream));
      CRNN_TRY(crnn_bn_bwd_apply_ex(c.w("d" + p), gA, c.w("bn1s" + p), c.w("coef"), gB, B, H, W, ci, 1, 1, 0.f, 0, 0, dtd, stream));
    } else
    CRNN_TRY(crnn_bn_bwd_ex(c.w("d" + p), gA, c.w("bn1s" + p), c.p(bp + "_bn1_g"), gB, c.g(bp + "_bn1_g"), c.g(bp + "_bn1_b"), c.w("partials"),
                            c.w("coef"), B, H, W, ci, 1, 1, 0.f, 0, 0, dtd, stream));
    CRNN_TRY(crnn_dwconv3x3_wgrad_ex(xin, gB, c.g(bp + "_dw"), c.w("partials"), B, H, W, ci, dtd, stream));
    if (i > 1 || cfg->stn) CRNN_TRY(crnn_dwconv3x3_fwd_ex(gB, c.p(bp + "_dw"), gA, nullptr, B, H, W, ci, 1, dtd, stream));
  }
  CRNN_TRY(fj.join());                                   // every weight gradient is complete in the main stream's order
  CRNN_TRY(flush_deferred(c));
  // ---- spatial transformer
  if (cfg->stn) {
    CRNN_TRY(crnn_sampler_bwd(x, c.w("theta"), gA, c.w("dtheta"), B, d.H0, d.W0, 2, stream));
    if (loc_net_fused(cfg, d) && c.P.off("locterms") >= 0 && aligned16(c.p("stn_c2_k"), c.w("pool2"), c.w("locterms"), c.p("stn_d1_w")) && aligned16(c.w("dfc1"), c.w("fc1"), c.w("dtheta")))
      return crnn_loc_net_bwd(c.w("dtheta"), c.w("flat"), c.w("fc1"), c.w("pool1"), c.w("c1"), c.w("pool2"), c.p("stn_d1_w"), c.p("stn_d2_w"), c.p("stn_c2_k"),
                              c.w("dfc1"), c.w("locterms"), c.g("stn_c1_k"), c.g("stn_c1_b"), c.g("stn_c2_k"), c.g("stn_c2_b"), c.g("stn_d1_w"), c.g("stn_d1_b"),
                              c.g("stn_d2_w"), c.g("stn_d2_b"), B, d.H0, d.W0, stream);
    CRNN_TRY(crnn_loc_fc_bwd(c.w("flat"), c.w("fc1"), c.w("dtheta"), c.p("stn_d1_w"), c.p("stn_d2_w"), c.w("dfc1"), c.w("dflat"),
                             c.g("stn_d1_w"), c.g("stn_d1_b"), c.g("stn_d2_w"), c.g("stn_d2_b"), B, d.stn_flat, stream));
    CRNN_TRY(crnn_loc_conv_wgrad(c.w("pool2"), c.w("dflat"), c.g("stn_c2_k"), c.g("stn_c2_b"), c.w("partials"), B, d.Hs2, d.Ws2, 20, stream));
    CRNN_TRY(crnn_loc_conv_dgrad(c.w("dflat"), c.p("stn_c2_k"), c.w("dpool2"), B, d.Hs2, d.Ws2, stream));
    CRNN_TRY(crnn_maxpool_bwd(c.w("c1"), c.w("dpool2"), c.w("dc1"), B, d.Ho1, d.Wo1, 20, 2, 2, stream));
    CRNN_TRY(crnn_loc_conv_wgrad(c.w("pool1"), c.w("dc1"), c.g("stn_c1_k"), c.g("stn_c1_b"), c.w("partials"), B, d.Hs1, d.Ws1, 1, stream));
  }
  return CRNN_OK;
}
}  // namespace

extern "C" int crnn_train_step_adam(const crnn_config* cfg, float* params, float* grads, float* m, float* v, float* bn_mean, float* bn_var,
                                    const float* x, const int* labels, const int* input_length, const int* label_length, float* ws,
                                    size_t ws_bytes, float* y_pred, float* loss, void* norm_scratch, float* norm_out, float lr_t,
                                    float beta1, float beta2, float eps, float clipnorm, uint64_t seed, hipStream_t stream) {
  CRNN_TRY(crnn_forward_ex(cfg, params, bn_mean, bn_var, x, ws, ws_bytes, y_pred, 1, seed, stream, nullptr));
  CRNN_TRY(crnn_backward(cfg, params, grads, x, labels, input_length, label_length, ws, ws_bytes, loss, seed, stream));
  const long n = make_layout(cfg).total;
  CRNN_TRY(crnn_global_norm(grads, n, clipnorm, norm_scratch, norm_out, stream));
  CRNN_TRY(crnn_adam_step(params, grads, m, v, n, lr_t, beta1, beta2, eps, norm_out, stream));
  return crnn_bn_update(cfg, bn_mean, bn_var, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm moving statistics (momentum .99): m <- .99 m + .01 mean ;
// v <- .99 v + .01 var * n/(n-1) * n/(n-(1+eps))   (TF fused-BN Bessel correction x Keras 2.2.2 factor)
struct BnTable { int state_off[14]; int C[14]; int cum[15]; float count[14]; };
__global__ void bn_moving_kernel(BnTable tab, float* __restrict__ mmean, float* __restrict__ mvar, const float* __restrict__ ws,
                                 int total, float momentum, float eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int k = 0;
  while (k < 13 && i >= tab.cum[k + 1]) ++k;
  int ch = i - tab.cum[k];
  float n = tab.count[k];
  float mean = ws[tab.state_off[k] + ch], var = ws[tab.state_off[k] + tab.C[k] + ch];
  float vhat = var * (n / (n - 1.f)) * (n / (n - (1.f + eps)));
  mmean[i] = momentum * mmean[i] + (1.f - momentum) * mean;
  mvar[i] = momentum * mvar[i] + (1.f - momentum) * vhat;
}

extern "C" int crnn_bn_update(const crnn_config* cfg, float* bn_mean, float* bn_var, float* ws, size_t ws_bytes, hipStream_t stream) {
  CRNN_TRY(check_cfg(cfg));
  Dims d = make_dims(cfg);
  Plan P = make_plan(cfg);
  if (ws_bytes < (size_t)P.total * sizeof(float)) return CRNN_ERR_ARG;
  BnTable tab;
  int k = 0, cum = 0;
  for (int i = 1; i <= 7; ++i)
    for (int j = 1; j <= 2; ++j, ++k) {
      int C = (j == 1) ? d.bc[i - 1] : d.bc[i];
      tab.state_off[k] = (int)P.off(std::string("bn") + (j == 1 ? "1s" : "2s") + std::to_string(i));
      tab.C[k] = C; tab.cum[k] = cum; tab.count[k] = (float)((long)d.B * d.bh[i] * d.bw[i]);
      cum += C;
    }
  tab.cum[14] = cum;
  hipLaunchKernelGGL(bn_moving_kernel, dim3(cdiv(cum, 256)), dim3(256), 0, stream, tab, bn_mean, bn_var, ws, cum, 0.99f, 1e-3f);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
