// disc_bn_step.h — the ORDER of the phases of disc_bn.h: one discriminator step with BatchNorm blocks (train mode) and the eval-mode forward.
// Written once against a launcher `LN` so that the device build (ilsx_disc.hip: every phase one kernel launch on the ctx stream) and the
// host emulation (tests/harness/disc_bn_host.cpp: every phase a serial loop) run the SAME sequence:
//   LN::par(count, f)   f(idx) for idx in [0, count)                      (rows / elements)
//   LN::col(H, f)       f(j, lane) for every feature column j — or row r  (one wavefront per column / row on the device)
//   LN::gemm(g)         C (+)= op(A) op(B) (+ bias), DbnGemm of disc_bn.h    (LDS-tiled on the device; up to two independent products per launch)
// Reference: adv_irl.py:133-216 (_do_reward_training), :268-274 (eval-mode logits), simple_disc_models.py:8-48.
#pragma once
#include "disc_bn.h"

#ifdef DBN_HOST_EMU
#define DBN_LAMBDA [=]
#else
#define DBN_LAMBDA [=] __device__
#endif

#define DBN_MAX_BLK 3

// parameters in the flat order of torch's parameters(): per block W [H][in] | b [H] | gamma [H] | beta [H]; then w [H] | c
struct DbnNet {
  int D, H, nblk, act;
  float clampv;
  float *P, *G, *M, *V;        // n_params floats each
  float *rmean, *rvar;         // running statistics [nblk][H]
  int in_of(int l) const { return l == 0 ? D : H; }
  int off_W(int l) const { int o = 0; for (int i = 0; i < l; ++i) o += H * in_of(i) + 3 * H; return o; }
  int off_b(int l) const { return off_W(l) + H * in_of(l); }
  int off_g(int l) const { return off_b(l) + H; }
  int off_be(int l) const { return off_b(l) + 2 * H; }
  int off_w() const { return off_W(nblk); }
  int off_c() const { return off_w() + H; }
  int n_params() const { return off_c() + 1; }
};
// workspace for up to `rows` = 3 * max_batch rows (2B stacked expert | policy rows of the CE pass + B interpolates of the gradient penalty)
struct DbnWs {
  float *X, *XH;                                        // [2B][D] stacked CE input, [B][D] interpolates
  float *ch[DBN_MAX_BLK], *ah[DBN_MAX_BLK], *h[DBN_MAX_BLK], *p[DBN_MAX_BLK], *s[DBN_MAX_BLK];          // forward tape of the pass in flight
  float *uh[DBN_MAX_BLK], *uy[DBN_MAX_BLK], *uah[DBN_MAX_BLK], *tt[DBN_MAX_BLK], *ua[DBN_MAX_BLK], *m2[DBN_MAX_BLK];   // the penalty's first backward
  float *ybar[DBN_MAX_BLK], *ahbar[DBN_MAX_BLK], *sbar[DBN_MAX_BLK];
  float *t0, *t1;                                       // [2B][max(H, D)] scratch
  float *logit, *dlogit, *gate, *ce_row, *correct, *gp_row;   // [2B] / [B]
};

// forward of `n` rows from x (row stride D) through all blocks; train: batch statistics (+ running update), eval: running statistics
template <class LN>
void dbn_forward(LN& L, const DbnNet& N, const DbnWs& W, const float* x, int n, int train, int update_running, bool tape) {
  const int H = N.H, act = N.act;
  const float* in = x;
  int K = N.D;
  for (int l = 0; l < N.nblk; ++l) {
    const float *Wl = N.P + N.off_W(l), *bl = N.P + N.off_b(l), *gl = N.P + N.off_g(l), *bel = N.P + N.off_be(l);
    float *ch = W.ch[l], *ah = tape ? W.ah[l] : nullptr, *h = W.h[l], *p = tape ? W.p[l] : nullptr, *s = tape ? W.s[l] : nullptr;
    float *rm = N.rmean + (size_t)l * H, *rv = N.rvar + (size_t)l * H;
    const int Kl = K;
    L.gemm(dbn_g_dense(in, Kl, Wl, bl, ch, n, H, Kl));
    L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_fwd(j, lane, ch, ah, h, p, s, gl, bel, rm, rv, n, H, act, train, update_running); });
    in = h; K = H;
  }
}

// eval-mode clamped logits of n rows of cat(obs, second)  (adv_irl.py:268-274)
template <class LN>
void dbn_logits_eval(LN& L, const DbnNet& N, const DbnWs& W, const float* x, int n, float* logits) {
  dbn_forward(L, N, W, x, n, /*train=*/0, 0, false);
  const int H = N.H;
  const float *hL = W.h[N.nblk - 1], *w = N.P + N.off_w(), *cp = N.P + N.off_c();
  const float clampv = N.clampv;
  L.col(n, DBN_LAMBDA(int r, int lane) { dbn_head(r, lane, hL, w, cp[0], clampv, H, logits, nullptr); });
}

// AdvIRL._do_reward_training: gradients of BCE(2B rows) + gp_w * penalty(B interpolates) into N.G (no optimiser step).
// X [2B][D] and (use_gp) XH [B][D] are filled by the caller (dbn_prep).
template <class LN>
void dbn_backward(LN& L, const DbnNet& N, const DbnWs& W, int B, int use_gp, float gp_w) {
  const int H = N.H, D = N.D, nb = N.nblk, act = N.act, n1 = 2 * B, np = N.n_params();
  const float clampv = N.clampv;
  float* G = N.G;
  (void)np;   // no zeroing launch: the cross-entropy pass ASSIGNS every gradient word (W, b, gamma, beta of every block, w, c), the penalty adds
  // ---- cross-entropy pass: train-mode forward of the 2B rows (running statistics move), backward
  dbn_forward(L, N, W, W.X, n1, 1, 1, true);
  {
    const float *hL = W.h[nb - 1], *w = N.P + N.off_w(), *cp = N.P + N.off_c();
    float *lg = W.logit, *dl = W.dlogit, *ce = W.ce_row, *co = W.correct;
    L.col(n1, DBN_LAMBDA(int r, int lane) { dbn_head_ce(r, lane, hL, w, cp[0], clampv, B, H, lg, dl, ce, co); });
    float* gw = G + N.off_w();
    float* gc = G + N.off_c();
    L.col(H + 1, DBN_LAMBDA(int j, int lane) {   // column H: the output bias, dc = sum_r dlogit
      if (j < H) dbn_col_dot(j, lane, dl, hL, gw, n1, H, 0);
      else dbn_vec_sum(lane, dl, n1, gc);
    });
    const float* uh = nullptr;
    for (int l = nb - 1; l >= 0; --l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l);
      const float *p = W.p[l], *ah = W.ah[l], *s = W.s[l], *xin = l > 0 ? W.h[l - 1] : W.X;
      float *ua = W.t0, *dg = G + N.off_g(l), *dbe = G + N.off_be(l), *db = G + N.off_b(l), *dW = G + N.off_W(l);
      const int K = N.in_of(l);
      const float* uhl = uh;
      L.col(H, DBN_LAMBDA(int j, int lane) {
        dbn_col_bwd(j, lane, uhl, dl, w, p, ah, s, gl, ua, nullptr, nullptr, nullptr, nullptr, nullptr, dg, dbe, db, n1, H, 0);
      });
      if (l > 0) {   // the weight gradient and the cotangent for the block below read the same ua: one launch
        float* nxt = W.t1;
        L.gemm(dbn_g_outer(ua, xin, K, dW, n1, H, K, 0), dbn_g_dense_t(ua, Wl, nxt, n1, H, K));   // K == H here
        uh = nxt;   // the block below reads it in its column phase (writing ua into t0) before its own dx lands in t1 again
      } else {
        L.gemm(dbn_g_outer(ua, xin, K, dW, n1, H, K, 0));
      }
    }
  }
  if (!use_gp) return;
  // ---- penalty pass: train-mode forward of the B interpolates (their OWN batch statistics; the running statistics move again)
  dbn_forward(L, N, W, W.XH, B, 1, 1, true);
  const float *w = N.P + N.off_w(), *cp = N.P + N.off_c();
  {
    const float* hL = W.h[nb - 1];
    float* gt = W.gate;
    L.col(B, DBN_LAMBDA(int r, int lane) { dbn_head(r, lane, hL, w, cp[0], clampv, H, nullptr, gt); });
  }
  // first backward: g = d(sum_r D(xh_r)) / d xh through the batch statistics, tape kept
  {
    const float* uh = nullptr;
    for (int l = nb - 1; l >= 0; --l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l), *p = W.p[l], *ah = W.ah[l], *s = W.s[l], *gt = W.gate;
      float *ua = W.ua[l], *uho = W.uh[l], *uy = W.uy[l], *uah = W.uah[l], *tt = W.tt[l], *m2 = W.m2[l];
      const int K = N.in_of(l);
      const float* uhl = uh;
      L.col(H, DBN_LAMBDA(int j, int lane) {
        dbn_col_bwd(j, lane, uhl, gt, w, p, ah, s, gl, ua, uho, uy, uah, tt, m2, nullptr, nullptr, nullptr, B, H, 1);
      });
      float* ux = l > 0 ? W.t1 : W.t0;   // l == 0: dD/dx [B][D] in t0
      L.gemm(dbn_g_dense_t(ua, Wl, ux, B, H, K));
      uh = ux;                            // read by the block below's column phase (which stores its own copy in W.uh[l-1]) before t1 is rewritten
    }
  }
  float* xbar = W.t1;   // [B][D]
  {
    const float* g = W.t0;
    float* gpr = W.gp_row;
    L.col(B, DBN_LAMBDA(int r, int lane) { dbn_gp_row(r, lane, g, xbar, gpr, B, D, gp_w); });
  }
  // reverse of the first backward, bottom block first
  {
    const float* xb = xbar;   // adjoint of the block's ux, [B][K]
    for (int l = 0; l < nb; ++l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l);
      const int K = N.in_of(l);
      float *dW = G + N.off_W(l), *dg = G + N.off_g(l);
      const float *ua = W.ua[l], *xbl = xb;
      float* uabar = W.t0;
      L.gemm(dbn_g_outer(ua, xbl, K, dW, B, H, K, 1),               // ux = ua W: its weight gradient
             dbn_g_dense(xbl, K, Wl, nullptr, uabar, B, H, K));      // uabar = xbar W^T   (both read xbar only: one launch)
      const float *tt = W.tt[l], *s = W.s[l], *ah = W.ah[l], *uah = W.uah[l], *uy = W.uy[l], *uhl = W.uh[l], *p = W.p[l], *h = W.h[l], *m2 = W.m2[l];
      float *yb = W.ybar[l], *ahb = W.ahbar[l], *sb = W.sbar[l], *up = W.t1;   // xbar (t1) was consumed by the two phases above
      L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_rev(j, lane, uabar, tt, s, ah, uah, uy, uhl, p, h, m2, gl, yb, ahb, sb, up, dg, B, H, act); });
      xb = up;
    }
    const float* gt = W.gate;
    float* gw = G + N.off_w();
    const float* top = xb;
    L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_dot(j, lane, gt, top, gw, B, H, 1); });   // uh_L = gate w
  }
  // and down the forward graph
  {
    const float* hbar = nullptr;
    for (int l = nb - 1; l >= 0; --l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l), *p = W.p[l], *ah = W.ah[l], *ch = W.ch[l], *s = W.s[l];
      const float *yb = W.ybar[l], *ahb = W.ahbar[l], *sb = W.sbar[l], *xin = l > 0 ? W.h[l - 1] : W.XH;
      float *ab = W.t0, *dg = G + N.off_g(l), *dbe = G + N.off_be(l), *db = G + N.off_b(l), *dW = G + N.off_W(l);
      const int K = N.in_of(l);
      const float* hb = hbar;
      L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_down(j, lane, yb, hb, p, ah, ahb, ch, s, sb, gl, ab, dg, dbe, db, B, H); });
      if (l > 0) {
        float* nxt = W.t1;
        L.gemm(dbn_g_outer(ab, xin, K, dW, B, H, K, 1), dbn_g_dense_t(ab, Wl, nxt, B, H, K));
        hbar = nxt;
      } else {
        L.gemm(dbn_g_outer(ab, xin, K, dW, B, H, K, 1));
      }
    }
  }
}

// statistics of the step (adv_irl.py:205-216): stats[0] = mean BCE, [1] = accuracy, [2] = mean (|g| - 1)^2
template <class LN>
void dbn_stats(LN& L, const DbnWs& W, int B, int use_gp, float* stats) {
  const float *ce = W.ce_row, *co = W.correct, *gp = W.gp_row;
  L.col(1, DBN_LAMBDA(int, int lane) {   // one wavefront: lanes split the rows
    float a = 0.0f, b = 0.0f, c = 0.0f;
    for (int r = lane; r < 2 * B; r += DBN_LANES) { a += ce[r]; b += co[r]; }
    if (use_gp) for (int r = lane; r < B; r += DBN_LANES) c += gp[r];
    a = dbn_wsum(a); b = dbn_wsum(b); c = dbn_wsum(c);
    if (lane == 0) { stats[0] = a / (float)(2 * B); stats[1] = b / (float)(2 * B); stats[2] = use_gp ? c / (float)B : 0.0f; }
  });
}
// Adam(lr, betas = (b1, 0.999)) over every parameter, step count t (1-based) — adv_irl.py:75-77
template <class LN>
void dbn_adam_step(LN& L, const DbnNet& N, float lr, float b1, int t) {
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
  const float step = (float)((double)lr / bc1), bc2s = (float)sqrt(bc2);
  float *P = N.P, *G = N.G, *M = N.M, *V = N.V;
  L.par(N.n_params(), DBN_LAMBDA(int i) { dbn_adam(i, P, G, M, V, step, bc2s, b1, 0.999f, 1e-8f); });
}
