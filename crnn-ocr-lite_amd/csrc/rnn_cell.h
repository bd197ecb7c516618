// LSTM cell arithmetic shared by the per-step (rnn.hip) and persistent (rnn_persist.hip) recurrences
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
