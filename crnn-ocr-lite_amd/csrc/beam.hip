// CTC beam-search decoding, one 64-lane wavefront per sample, everything resident in LDS.
// Restates tf.nn.ctc_beam_search_decoder (TF r1.8 ctc_beam_search.h; reached by the reference through
// K.ctc_decode(greedy=False, beam_width, top_paths=1), utils.py:353): prefix trie with (blank,label,total)
// log-probabilities per node, parent contribution only while the parent is still in the beam, children of a
// beam entry offered unless already in the beam, top-`beam_width` kept, merge_repeated applied to the final
// label sequence.  Flat-beam formulation that is exactly equivalent to the trie walk (ties aside):
//   * node identity = (parent node, label), kept in an LDS node table so a prefix that drops out of the beam
//     and re-enters later reuses its node (its descendants see it as their parent again);
//   * the sequential "push, evict the bottom" of the TopN container == top-N selection over
//     {re-scored beam entries} U {offered children}, earlier insertion winning ties.
// Lanes = classes for the expansion (C <= 64), lanes = beam slots for the per-entry update.
#include "common.h"

#define BEAM_MAX 16
#define BEAM_EPS 1e-7f
#define BNEG (-INFINITY)

__device__ __forceinline__ float blse(float a, float b) {
  if (a == BNEG) return b;
  if (b == BNEG) return a;
  float m = fmaxf(a, b), n = fminf(a, b);
  return m + log1pf(expf(n - m));
}

struct BeamSet {  // one beam (<= BEAM_MAX entries) in LDS
  int node[BEAM_MAX], par[BEAM_MAX], lab[BEAM_MAX];
  float ob[BEAM_MAX], ol[BEAM_MAX], ot[BEAM_MAX], nb[BEAM_MAX], nl[BEAM_MAX], nt[BEAM_MAX];
};

__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ y, const int* __restrict__ input_len,
                                                      int* __restrict__ out, int* __restrict__ out_len,
                                                      float* __restrict__ scores, int T, int C, int bw, int merge_repeated,
                                                      int nmax) {
  extern __shared__ int smem_i[];
  int* nodes = smem_i;                                        // [nmax] ((parent+1)<<8)|(label+1)
  float* cand = reinterpret_cast<float*>(nodes + nmax);       // [(BEAM_MAX+1)][64]
  float* inp = cand + (BEAM_MAX + 1) * 64;                    // [64]
  int* pick_row = reinterpret_cast<int*>(inp + 64);           // [BEAM_MAX]
  int* pick_lane = pick_row + BEAM_MAX;
  float* pick_val = reinterpret_cast<float*>(pick_lane + BEAM_MAX);
  BeamSet* sets = reinterpret_cast<BeamSet*>(pick_val + BEAM_MAX);  // [2]
  const int b = blockIdx.x, lane = threadIdx.x, blank = C - 1;
  int Tb = input_len ? input_len[b] : T; if (Tb > T) Tb = T; if (Tb < 0) Tb = 0;

  BeamSet* cur = &sets[0]; BeamSet* nxt = &sets[1];
  int n = 1, nnodes = 1;
  if (lane == 0) {
    nodes[0] = 0;  // root: parent -1, label -1
    cur->node[0] = 0; cur->par[0] = -1; cur->lab[0] = -1;
    cur->ob[0] = cur->ol[0] = cur->ot[0] = BNEG;
    cur->nb[0] = 0.f; cur->nl[0] = BNEG; cur->nt[0] = 0.f;
  }
  __syncthreads();

  for (int t = 0; t < Tb; ++t) {
    float lg = (lane < C) ? logf(y[((long)b * T + t) * C + lane] + BEAM_EPS) : BNEG;
    float mx = wave_max(lg);
    inp[lane] = lg - mx;
    // ---- oldp <- newp for every beam entry
    if (lane < n) { cur->ob[lane] = cur->nb[lane]; cur->ol[lane] = cur->nl[lane]; cur->ot[lane] = cur->nt[lane]; }
    __syncthreads();
    // ---- re-score the entries that stay (TF: second loop of Step)
    if (lane < n) {
      int nd = cur->node[lane], par = cur->par[lane], lab = cur->lab[lane];
      float nlv = BNEG;
      if (nd != 0) {
        nlv = cur->ol[lane];
        for (int j = 0; j < n; ++j)
          if (cur->node[j] == par) {  // parent still in the beam (Active)
            int plab = (nodes[par] & 255) - 1;
            nlv = blse(nlv, (lab == plab) ? cur->ob[j] : cur->ot[j]);
          }
        nlv = (nlv == BNEG) ? BNEG : nlv + inp[lab];
      }
      float nbv = cur->ot[lane] + inp[blank];
      cur->nb[lane] = nbv; cur->nl[lane] = nlv; cur->nt[lane] = blse(nbv, nlv);
    }
    __syncthreads();
    // ---- offer children: row i+1 = children of entry i, column = label; row 0 = the entries themselves
    cand[lane] = (lane < n) ? cur->nt[lane] : BNEG;
    for (int i = 0; i < n; ++i) {
      float v = BNEG;
      if (lane < blank) {
        bool active = false;
        int nd = cur->node[i];
        for (int j = 0; j < n; ++j) active |= (cur->par[j] == nd && cur->lab[j] == lane);
        if (!active) {
          float prev = (lane == cur->lab[i]) ? cur->ob[i] : cur->ot[i];
          v = (prev == BNEG) ? BNEG : inp[lane] + prev;
        }
      }
      cand[(i + 1) * 64 + lane] = v;
    }
    __syncthreads();
    // ---- keep the top bw (value desc; ties: smaller row, then smaller lane = earlier insertion)
    int newn = 0;
    for (int r = 0; r < bw; ++r) {
      float best = BNEG; int brow = 0;
      for (int row = 0; row <= n; ++row) { float v = cand[row * 64 + lane]; if (v > best) { best = v; brow = row; } }
      int key = brow * 64 + lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int ok = __shfl_xor(key, o, 64);
        if (ov > best || (ov == best && ok < key)) { best = ov; key = ok; }
      }
      if (best == BNEG) break;
      if (lane == (key & 63)) cand[key] = BNEG;
      if (lane == 0) { pick_row[r] = key >> 6; pick_lane[r] = key & 63; pick_val[r] = best; }
      ++newn;
      __syncthreads();
    }
    __syncthreads();
    // ---- build the new beam in pick order (already descending)
    for (int r = 0; r < newn; ++r) {
      int row = pick_row[r], pl = pick_lane[r];
      if (row == 0) {
        if (lane == 0) {
          nxt->node[r] = cur->node[pl]; nxt->par[r] = cur->par[pl]; nxt->lab[r] = cur->lab[pl];
          nxt->ob[r] = cur->ob[pl]; nxt->ol[r] = cur->ol[pl]; nxt->ot[r] = cur->ot[pl];
          nxt->nb[r] = cur->nb[pl]; nxt->nl[r] = cur->nl[pl]; nxt->nt[r] = cur->nt[pl];
        }
      } else {
        int parent = cur->node[row - 1];
        int packed = ((parent + 1) << 8) | (pl + 1);
        int found = -1;
        for (int base = 1; base < nnodes; base += 64) {  // re-entering prefix? reuse its node
          int idx = base + lane;
          bool hit = (idx < nnodes) && (nodes[idx] == packed);
          unsigned long long m = __ballot(hit);
          if (m) { found = base + __ffsll((long long)m) - 1; break; }
        }
        if (found < 0) {
          found = nnodes;
          if (nnodes < nmax) { if (lane == 0) nodes[nnodes] = packed; ++nnodes; }
        }
        if (lane == 0) {
          float v = pick_val[r];
          nxt->node[r] = found; nxt->par[r] = parent; nxt->lab[r] = pl;
          nxt->ob[r] = nxt->ol[r] = nxt->ot[r] = BNEG;
          nxt->nb[r] = BNEG; nxt->nl[r] = v; nxt->nt[r] = v;
        }
        __syncthreads();
      }
    }
    __syncthreads();
    BeamSet* tmp = cur; cur = nxt; nxt = tmp;
    n = newn;
  }
  // ---- best path = slot 0; walk to the root, merge_repeated on the collapsed sequence, reverse
  for (int i = lane; i < T; i += 64) out[(long)b * T + i] = -1;
  __syncthreads();
  if (lane == 0) {
    int len = 0;
    if (n > 0) {
      int nd = cur->node[0], prev = -1;
      // labels are emitted leaf->root straight into `out`, then flipped in place
      while (nd != 0) {
        int pk = nodes[nd];
        int lab = (pk & 255) - 1;
        if (!merge_repeated || lab != prev) out[(long)b * T + len++] = lab;
        prev = lab;
        nd = (pk >> 8) - 1;
      }
      for (int i = 0; i < len / 2; ++i) {
        int a = out[(long)b * T + i]; out[(long)b * T + i] = out[(long)b * T + len - 1 - i]; out[(long)b * T + len - 1 - i] = a;
      }
      scores[b] = cur->nt[0];
    } else {
      scores[b] = 0.f;
    }
    out_len[b] = len;
  }
}

extern "C" int crnn_ctc_beam_decode(const float* y, const int* input_len, int* out, int* out_len, float* scores, int B, int T,
                                    int C, int beam_width, int merge_repeated, hipStream_t stream) {
  if (C > 64 || C < 2 || beam_width < 1 || beam_width > BEAM_MAX) return CRNN_ERR_UNSUPPORTED;
  int nmax = 1 + T * beam_width;
  size_t lds = (size_t)nmax * 4 + (size_t)(BEAM_MAX + 1) * 64 * 4 + 64 * 4 + BEAM_MAX * 12 + 2 * sizeof(BeamSet) + 64;
  if (lds > 64 * 1024) return CRNN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(64), lds, stream, y, input_len, out, out_len, scores, T, C, beam_width, merge_repeated, nmax);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
