// CTC beam-search decoding, one 64-lane wavefront per sample; beam state lives in registers (lane i = beam
// entry i = TopN slot i), classes map to lanes for the expansion, the prefix-node table sits in LDS.
//
// Restates tf.nn.ctc_beam_search_decoder (TF r1.8 ctc_beam_search.h; reached by the reference through
// K.ctc_decode(greedy=False, beam_width, top_paths=1), utils.py:353) EXACTLY, including the order-dependent side
// effects of its sequential "grow new leaves" loop:
//   * entries are visited in descending previous-score order, their children in label order;
//   * a child is offered only if it is not currently in the beam; an accepted child evicts the current bottom;
//   * an entry that has been evicted earlier in the same step and is then re-offered by its parent fails the
//     candidate test and gets its old probabilities reset -- so it no longer expands when its own turn comes.
// A prefix trie is replaced by node identity = (parent node, label) in an LDS table, so a prefix that drops out
// and re-enters later is the same node again (its descendants regain their parent term).  The per-entry loop over
// the 37 children only iterates over "events" (labels that can enter the beam, or that name an existing entry),
// found with one wave ballot; every decision inside it is wave-uniform.
#include "common.h"

#define BEAM_MAX 64          // one beam entry per lane
#define BEAM_EPS 1e-7f
#define BNEG (-INFINITY)

__device__ __forceinline__ float blse(float a, float b) {
  if (a == BNEG) return b;
  if (b == BNEG) return a;
  float m = fmaxf(a, b), n = fminf(a, b);
  return m + log1pf(expf(n - m));
}

// Value of lane l (wave-uniform l) as a scalar: v_readlane_b32 instead of the LDS round trip of ds_bpermute_b32 -- the decoder is one wave per
// sample, a chain of several hundred dependent cross-lane reads per time step, so their latency IS its run time.
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// (min value, its lane) over lanes < cnt; ties -> lowest lane.  Result is wave-uniform.  The first four butterfly steps stay inside a row of
// 16 lanes (DPP: quad permutes, half-row mirror, row mirror -- any pairing works for an idempotent reduction); beams of at most 16 entries
// (the reference decodes with 5 or 10) never leave the row.
template <int CTRL>
__device__ __forceinline__ float max_dpp(float v) {
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
// maximum over the 64 lanes, wave-uniform: four in-row DPP steps, then the four row maxima as scalars
__device__ __forceinline__ float wave_max64(float v) {
  v = max_dpp<0xB1>(v); v = max_dpp<0x4E>(v); v = max_dpp<0x141>(v); v = max_dpp<0x140>(v);
  return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}
template <int CTRL>
__device__ __forceinline__ void argmin_dpp(float& mv, int& ml) {
  const float ov = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(mv), __float_as_int(mv), CTRL, 0xf, 0xf, false));
  const int ol = __builtin_amdgcn_update_dpp(ml, ml, CTRL, 0xf, 0xf, false);
  if (ov < mv || (ov == mv && ol < ml)) { mv = ov; ml = ol; }
}
__device__ __forceinline__ void wave_argmin(float v, int lane, int cnt, float& mv, int& ml) {
  mv = (lane < cnt) ? v : INFINITY; ml = lane;
  argmin_dpp<0xB1>(mv, ml);      // quad_perm [1,0,3,2]
  argmin_dpp<0x4E>(mv, ml);      // quad_perm [2,3,0,1]
  argmin_dpp<0x141>(mv, ml);     // row_half_mirror
  argmin_dpp<0x140>(mv, ml);     // row_mirror
  if (cnt > 16) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      float ov = __shfl_xor(mv, o, 64); int ol = __shfl_xor(ml, o, 64);
      if (ov < mv || (ov == mv && ol < ml)) { mv = ov; ml = ol; }
    }
  }
  mv = rl(mv, 0); ml = rl(ml, 0);
}

__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ y, const int* __restrict__ input_len,
                                                      int* __restrict__ out, int* __restrict__ out_len,
                                                      float* __restrict__ scores, int T, int C, int bw, int merge_repeated,
                                                      int nmax) {
  extern __shared__ int smem_i[];
  int* nodes = smem_i;                                   // [nmax] ((parent+1)<<8)|(label+1); node 0 = root
  int* s_ref = nodes + nmax;                             // [BEAM_MAX] sorted leaves: branch index or -1
  int* s_par = s_ref + BEAM_MAX;                         // parent branch index of a new child
  int* s_lab = s_par + BEAM_MAX;
  float* s_val = reinterpret_cast<float*>(s_lab + BEAM_MAX);
  const int b = blockIdx.x, lane = threadIdx.x, blank = C - 1;
  int Tb = input_len ? input_len[b] : T; if (Tb > T) Tb = T; if (Tb < 0) Tb = 0;

  // beam entry `lane` (valid for lane < n)
  int b_node = 0, b_par = -1, b_lab = -1, b_act = 0;
  float b_ob = BNEG, b_ol = BNEG, b_ot = BNEG, b_nb = 0.f, b_nl = BNEG, b_nt = 0.f;
  // TopN slot `lane` (valid for lane < nle)
  float l_v = BNEG; int l_ref = -1, l_par = -1, l_lab = -1;
  int n = 1, nnodes = 1;
  if (lane == 0) nodes[0] = 0;
  __syncthreads();
  // register copy of the node table, 64 nodes per register (node 64 k + lane in nd[k]): the search for a re-entering prefix compares against
  // registers instead of walking LDS -- up to ten searches per time step, each up to nnodes / 64 dependent LDS round trips before.  Tables
  // beyond kNodeRegs * 64 nodes (T * beam_width > 1023) keep the LDS walk.
  constexpr int kNodeRegs = 16;
  const bool regtab = nmax <= kNodeRegs * 64;
  int nd[kNodeRegs];
#pragma unroll
  for (int k = 0; k < kNodeRegs; ++k) nd[k] = -1;

  float ynext = (Tb > 0 && lane < C) ? y[(long)b * T * C + lane] : 0.f;        // the next step's posteriors are requested a step ahead
  for (int t = 0; t < Tb; ++t) {
    const float ycur = ynext;
    if (t + 1 < Tb && lane < C) ynext = y[((long)b * T + t + 1) * C + lane];
    float lg = (lane < C) ? logf(ycur + BEAM_EPS) : BNEG;
    const float inp = lg - wave_max64(lg);               // lane = class
    const float inp_blank = rl(inp, blank);
    // ---- oldp <- newp; re-score the entries (parent term only while the parent is in the beam)
    b_ob = b_nb; b_ol = b_nl; b_ot = b_nt;
    {
      float prev = BNEG; bool found = false;
      for (int j = 0; j < n; ++j) {
        int nj = rl(b_node, j), lj = rl(b_lab, j);
        float obj = rl(b_ob, j), otj = rl(b_ot, j);
        if (lane < n && b_node != 0 && nj == b_par) { found = true; prev = (b_lab == lj) ? obj : otj; }
      }
      float in_lab = __shfl(inp, b_lab & 63, 64);
      if (lane < n) {
        float nl = BNEG;
        if (b_node != 0) {
          nl = found ? blse(b_ol, prev) : b_ol;
          nl = (nl == BNEG) ? BNEG : nl + in_lab;
        }
        b_nb = b_ot + inp_blank; b_nl = nl; b_nt = blse(b_nb, nl);
      }
    }
    // ---- TopN <- all entries
    int nle = n;
    l_v = b_nt; l_ref = lane; l_par = -1; l_lab = -1; b_act = (lane < n) ? 1 : 0;
    float bval; int bslot;
    wave_argmin(l_v, lane, nle, bval, bslot);
    // ---- grow new leaves, entry by entry
    for (int bi = 0; bi < n; ++bi) {
      const float bot = rl(b_ot, bi), bob = rl(b_ob, bi);
      const int blab = rl(b_lab, bi), bnode = rl(b_node, bi);
      if (!(bot > BNEG && (nle < bw || bot > bval))) continue;
      const float prev = (lane == blab) ? bob : bot;
      const float v = (lane < blank && prev > BNEG) ? inp + prev : BNEG;     // lane = child label
      int cb = -1;
      for (int j = 0; j < n; ++j) {
        int pj = rl(b_par, j), lj = rl(b_lab, j);
        if (pj == bnode && lj == lane) cb = j;                                  // this child is beam entry j
      }
      const bool ev = (lane < blank) && (cb >= 0 || (v > BNEG && (nle < bw || v > bval)));
      unsigned long long mask = __ballot(ev);
      while (mask) {
        const int c = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float vc = rl(v, c);
        const int ccb = rl(cb, c);
        if (ccb >= 0 && rl(b_act, ccb)) continue;                       // child already in the beam
        if (vc > BNEG && (nle < bw || vc > bval)) {
          int slot;
          if (nle == bw) {                                                      // evict the bottom
            slot = bslot;
            int k = rl(l_ref, bslot);
            if (k >= 0 && lane == k) b_act = 0;
          } else {
            slot = nle++;
          }
          if (lane == slot) { l_v = vc; l_ref = -1; l_par = bi; l_lab = c; }
          wave_argmin(l_v, lane, nle, bval, bslot);
        } else if (ccb >= 0 && lane == ccb) {                                   // re-offered, rejected: reset oldp
          b_ob = b_ol = b_ot = BNEG;
        }
      }
    }
    // ---- new beam = TopN sorted by descending score (ties: lower slot first)
    int rank = 0;
    for (int j = 0; j < nle; ++j) {
      float vj = rl(l_v, j);
      if (vj > l_v || (vj == l_v && j < lane)) ++rank;
    }
    __syncthreads();
    if (lane < nle) { s_ref[rank] = l_ref; s_par[rank] = l_par; s_lab[rank] = l_lab; s_val[rank] = l_v; }
    __syncthreads();
    int r_ref = -1, r_par = 0, r_lab = 0; float r_val = BNEG;
    if (lane < nle) { r_ref = s_ref[lane]; r_par = s_par[lane]; r_lab = s_lab[lane]; r_val = s_val[lane]; }
    // surviving entries: copy from their old lane; new children: parent node from the parent's lane
    const int src = (r_ref >= 0) ? r_ref : (r_par & 63);
    int g_node = __shfl(b_node, src, 64), g_par = __shfl(b_par, src, 64), g_lab = __shfl(b_lab, src, 64);
    float g_nb = __shfl(b_nb, src, 64), g_nl = __shfl(b_nl, src, 64), g_nt = __shfl(b_nt, src, 64);
    int new_node = g_node, new_par = g_par, new_lab = g_lab;
    float new_nb = g_nb, new_nl = g_nl, new_nt = g_nt;
    const bool is_new = (lane < nle) && (r_ref < 0);
    if (is_new) { new_par = g_node; new_lab = r_lab; new_nb = BNEG; new_nl = r_val; new_nt = r_val; new_node = -1; }
    // resolve node ids of the new children one at a time (re-entering prefix -> reuse its node)
    unsigned long long nm = __ballot(is_new);
    while (nm) {
      const int r = __ffsll((long long)nm) - 1;
      nm &= nm - 1;
      const int packed = ((rl(new_par, r) + 1) << 8) | (rl(new_lab, r) + 1);
      int found = -1;
      if (regtab) {
#pragma unroll
        for (int k = 0; k < kNodeRegs; ++k) {
          if (found < 0 && k * 64 < nnodes) {                                   // (packed > 0 = nodes[0], and unused slots hold -1: no index test needed)
            const unsigned long long m = __ballot(nd[k] == packed);
            if (m) found = k * 64 + __ffsll((long long)m) - 1;
          }
        }
      } else {
        for (int base = 1; base < nnodes; base += 64) {
          int idx = base + lane;
          bool hit = (idx < nnodes) && (nodes[idx] == packed);
          unsigned long long m = __ballot(hit);
          if (m) { found = base + __ffsll((long long)m) - 1; break; }
        }
      }
      if (found < 0) {
        found = nnodes;
        if (nnodes < nmax) {
          if (lane == 0) nodes[nnodes] = packed;
          if (regtab) {
#pragma unroll
            for (int k = 0; k < kNodeRegs; ++k) if (k == (nnodes >> 6) && lane == (nnodes & 63)) nd[k] = packed;
          }
          ++nnodes;
        }
        __syncthreads();
      }
      if (lane == r) new_node = found;
    }
    b_node = new_node; b_par = new_par; b_lab = new_lab; b_nb = new_nb; b_nl = new_nl; b_nt = new_nt;
    n = nle;
  }
  // ---- best path = entry 0; walk to the root, merge_repeated on the collapsed sequence, reverse
  for (int i = lane; i < T; i += 64) out[(long)b * T + i] = -1;
  const int best_node = rl(b_node, 0);
  const float best_score = rl(b_nt, 0);
  __syncthreads();
  if (lane == 0) {
    int len = 0, nd = best_node, prev = -1;
    while (nd != 0) {                       // labels are emitted leaf->root straight into `out`, then flipped
      int pk = nodes[nd];
      int lab = (pk & 255) - 1;
      if (!merge_repeated || lab != prev) out[(long)b * T + len++] = lab;
      prev = lab;
      nd = (pk >> 8) - 1;
    }
    for (int i = 0; i < len / 2; ++i) {
      int a = out[(long)b * T + i]; out[(long)b * T + i] = out[(long)b * T + len - 1 - i]; out[(long)b * T + len - 1 - i] = a;
    }
    scores[b] = best_score;
    out_len[b] = len;
  }
}

extern "C" int crnn_ctc_beam_decode(const float* y, const int* input_len, int* out, int* out_len, float* scores, int B, int T,
                                    int C, int beam_width, int merge_repeated, hipStream_t stream) {
  if (C > 64 || C < 2 || beam_width < 1 || beam_width > BEAM_MAX) return CRNN_ERR_UNSUPPORTED;
  int nmax = 1 + T * beam_width;
  size_t lds = (size_t)nmax * 4 + BEAM_MAX * 16 + 64;
  if (lds > 64 * 1024) return CRNN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(64), lds, stream, y, input_len, out, out_len, scores, T, C, beam_width, merge_repeated, nmax);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
