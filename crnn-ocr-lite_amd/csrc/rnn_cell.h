// LSTM / GRU cell arithmetic shared by the per-step (rnn.hip) and persistent (rnn_persist.hip) recurrences
// (Keras 2.2.2 LSTMCell: hard_sigmoid gates, tanh, gate order i,f,c,o; utils.py:77-79).  Floating-point contraction
// is switched off inside these functions: every product and sum is rounded exactly as written, so two kernels that
// call them produce bit-identical results regardless of how the compiler schedules the surrounding code.
#pragma once
#include "common.h"

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 + v_rcp_f32 (about 1 ulp each): absolute error < 3e-7 everywhere, exact
// limits +-1 (exp -> inf / 0), ~10 instructions instead of libm's ~45 -- the recurrences are latency-bound chains of
// T dependent steps and every step ends in two of these.
__device__ __forceinline__ float lstm_tanh(float x) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x) = 2^(2x log2 e)
  const float r = __builtin_amdgcn_rcpf(e + 1.f);
  return 1.f - 2.f * r;
}

struct LstmFwdOut { float ig, fg, gg, og, cn, hn; };
__device__ __forceinline__ LstmFwdOut lstm_cell_fwd(const float (&z)[4], float cprev) {
#pragma clang fp contract(off)
  LstmFwdOut o;
  o.ig = hard_sigmoid(z[0]); o.fg = hard_sigmoid(z[1]); o.gg = lstm_tanh(z[2]); o.og = hard_sigmoid(z[3]);
  const float a = o.fg * cprev, b = o.ig * o.gg;
  o.cn = a + b;
  o.hn = o.og * lstm_tanh(o.cn);
  return o;
}

struct LstmBwdOut { float dz[4]; float dc; };
// dh = gradient w.r.t. h_t (recurrent part + upstream), dcin = cell-gradient carry from the step processed before
__device__ __forceinline__ LstmBwdOut lstm_cell_bwd(float dh, float ig, float fg, float gg, float og, float ct, float cprev, float dcin) {
#pragma clang fp contract(off)
  LstmBwdOut o;
  const float tc = lstm_tanh(ct);
  const float dog = dh * tc;
  const float t1 = tc * tc;
  const float t2 = 1.f - t1;
  const float t3 = dh * og;
  const float t4 = t3 * t2;
  const float dct = t4 + dcin;
  o.dz[0] = (dct * gg) * hs_grad_from_out(ig);
  o.dz[1] = (dct * cprev) * hs_grad_from_out(fg);
  const float g2 = gg * gg;
  o.dz[2] = (dct * ig) * (1.f - g2);
  o.dz[3] = dog * hs_grad_from_out(og);
  o.dc = dct * fg;
  return o;
}

// ---- GRU (Keras 2.2.2 GRUCell, reset_after=False, gate order z,r,h; utils.py:80-82), shared by the per-step kernels of rnn.hip and
// the persistent ones of gru_persist.hip: same operations in the same order, contraction off => bit-identical results.
struct GruZR { float zg, rg, rh; };
// zz, rr: pre-activations (recurrent product + x W + b); rh = r * h_prev feeds the candidate's recurrent product
__device__ __forceinline__ GruZR gru_cell_zr(float zz, float rr, float hprev) {
#pragma clang fp contract(off)
  GruZR o;
  o.zg = hard_sigmoid(zz); o.rg = hard_sigmoid(rr);
  o.rh = o.rg * hprev;
  return o;
}
struct GruH { float hh, hn; };
__device__ __forceinline__ GruH gru_cell_h(float pre, float zg, float hprev) {
#pragma clang fp contract(off)
  GruH o;
  o.hh = lstm_tanh(pre);
  const float a = zg * hprev, b = (1.f - zg) * o.hh;
  o.hn = a + b;
  return o;
}
// backward, first half: dh = gradient w.r.t. h_t -> gradients of the z and candidate pre-activations
struct GruBwdB { float dzz, dhh; };
__device__ __forceinline__ GruBwdB gru_cell_bwd_b(float dh, float zg, float hh, float hprev) {
#pragma clang fp contract(off)
  GruBwdB o;
  const float t0 = hprev - hh;
  o.dzz = (dh * t0) * hs_grad_from_out(zg);
  const float t1 = 1.f - zg, t2 = hh * hh, t3 = 1.f - t2;
  o.dhh = (dh * t1) * t3;
  return o;
}
// second half: drh = gradient w.r.t. r * h_prev -> gradient of the r pre-activation and the carry to h_prev
struct GruBwdA { float dzr, dhp; };
__device__ __forceinline__ GruBwdA gru_cell_bwd_a(float drh, float dh, float zg, float rg, float hprev) {
#pragma clang fp contract(off)
  GruBwdA o;
  o.dzr = (drh * hprev) * hs_grad_from_out(rg);
  const float a = dh * zg, b = drh * rg;
  o.dhp = a + b;
  return o;
}
