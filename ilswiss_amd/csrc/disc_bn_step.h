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
// workspace for up to `rows` = 3 * max_batch rows (2B stacked expert | policy rows of the CE pass + B interpolates of the gradient penalty).
// The two passes of a training step run SIDE BY SIDE (paired launches), so each has a forward tape and scratch of its own.
struct DbnWs {
  float *X, *XH;                                        // [2B][D] stacked CE input, [B][D] interpolates
  float *ch[DBN_MAX_BLK], *ah[DBN_MAX_BLK], *h[DBN_MAX_BLK], *p[DBN_MAX_BLK], *s[DBN_MAX_BLK];          // forward tape of the cross-entropy pass (and of an eval forward)
  float *gch[DBN_MAX_BLK], *gah[DBN_MAX_BLK], *gh[DBN_MAX_BLK], *gp[DBN_MAX_BLK], *gs[DBN_MAX_BLK];     // forward tape of the penalty pass
  float *uh[DBN_MAX_BLK], *uy[DBN_MAX_BLK], *uah[DBN_MAX_BLK], *tt[DBN_MAX_BLK], *ua[DBN_MAX_BLK], *m2[DBN_MAX_BLK];   // the penalty's first backward
  float *ybar[DBN_MAX_BLK], *ahbar[DBN_MAX_BLK], *sbar[DBN_MAX_BLK];
  float *t0, *t1;                                       // [2B][max(H, D)] scratch of the cross-entropy pass
  float *gt0, *gt1;                                     // [B][max(H, D)] scratch of the penalty pass
  float *bstat;                                         // [2 passes][nblk][mean | var][H] batch statistics for the deferred running update
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
    L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_fwd(j, lane, ch, ah, h, p, s, gl, bel, rm, rv, n, H, act, train, update_running, nullptr); });
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
// The cross-entropy pass (forward, head, backward) and the penalty pass's forward + FIRST backward do not depend on each other: they run as
// PAIRED launches — one launch holds the same phase of both (two matrix products; the columns of both passes) — on tapes of their own, and
// the running statistics, which only the eval forward reads, get both updates afterwards in the reference's order (dbn_running_update in
// dbn_finish).  A dependent launch of this size costs ~5.8 us whatever it does: the step's length is its launch count (round 5: 33 -> 20).
template <class LN>
void dbn_backward(LN& L, const DbnNet& N, const DbnWs& W, int B, int use_gp, float gp_w) {
  const int H = N.H, D = N.D, nb = N.nblk, act = N.act, n1 = 2 * B;
  const float clampv = N.clampv;
  float* G = N.G;   // no zeroing launch: the cross-entropy pass ASSIGNS every gradient word (W, b, gamma, beta of every block, w, c), the penalty adds
  const float *w = N.P + N.off_w(), *cp = N.P + N.off_c();
  // ---- train-mode forward of the 2B rows and of the B interpolates (each with its OWN batch statistics), block by block
  {
    const float *inC = W.X, *inG = W.XH;
    int K = D;
    for (int l = 0; l < nb; ++l) {
      const float *Wl = N.P + N.off_W(l), *bl = N.P + N.off_b(l), *gl = N.P + N.off_g(l), *bel = N.P + N.off_be(l);
      float *chC = W.ch[l], *ahC = W.ah[l], *hC = W.h[l], *pC = W.p[l], *sC = W.s[l];
      float *chG = W.gch[l], *ahG = W.gah[l], *hG = W.gh[l], *pG = W.gp[l], *sG = W.gs[l];
      float *bsC = W.bstat + (size_t)((0 * nb + l) * 2) * H, *bsG = W.bstat + (size_t)((1 * nb + l) * 2) * H;
      const int Kl = K;
      if (use_gp) L.gemm(dbn_g_dense(inC, Kl, Wl, bl, chC, n1, H, Kl), dbn_g_dense(inG, Kl, Wl, bl, chG, B, H, Kl));
      else L.gemm(dbn_g_dense(inC, Kl, Wl, bl, chC, n1, H, Kl));
      L.col(use_gp ? 2 * H : H, DBN_LAMBDA(int j, int lane) {
        if (j < H) dbn_col_fwd(j, lane, chC, ahC, hC, pC, sC, gl, bel, nullptr, nullptr, n1, H, act, 1, 0, bsC);
        else dbn_col_fwd(j - H, lane, chG, ahG, hG, pG, sG, gl, bel, nullptr, nullptr, B, H, act, 1, 0, bsG);
      });
      inC = hC; inG = hG; K = H;
    }
  }
  // ---- heads: CE terms and dlogit of the 2B rows ; the clamp's gate of the interpolates
  const float *hLC = W.h[nb - 1], *hLG = W.gh[nb - 1];
  float *dl = W.dlogit, *gt = W.gate;
  {
    float *lg = W.logit, *ce = W.ce_row, *co = W.correct;
    L.col(n1 + (use_gp ? B : 0), DBN_LAMBDA(int r, int lane) {
      if (r < n1) dbn_head_ce(r, lane, hLC, w, cp[0], clampv, B, H, lg, dl, ce, co);
      else dbn_head(r - n1, lane, hLG, w, cp[0], clampv, H, nullptr, gt);
    });
  }
  // ---- the cross-entropy backward, and the penalty's first backward g = d(sum_r D(xh_r)) / d xh through the batch statistics (tape kept),
  //      block by block from the top; the head's column sums ride in the top block's column launch
  {
    const float *uhC = nullptr, *uhG = nullptr;
    float *gw = G + N.off_w(), *gc = G + N.off_c();
    for (int l = nb - 1; l >= 0; --l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l);
      const float *pC = W.p[l], *ahC = W.ah[l], *sC = W.s[l], *xinC = l > 0 ? W.h[l - 1] : W.X;
      const float *pG = W.gp[l], *ahG = W.gah[l], *sG = W.gs[l];
      float *uaC = W.t0, *dg = G + N.off_g(l), *dbe = G + N.off_be(l), *db = G + N.off_b(l), *dW = G + N.off_W(l);
      float *uaG = W.ua[l], *uho = W.uh[l], *uy = W.uy[l], *uah = W.uah[l], *tt = W.tt[l], *m2 = W.m2[l];
      const int K = N.in_of(l), top = l == nb - 1 ? 1 : 0, c1 = use_gp ? 2 * H : H;
      const float *uhCl = uhC, *uhGl = uhG;
      L.col(c1 + (top ? H + 1 : 0), DBN_LAMBDA(int j, int lane) {
        if (j < H) dbn_col_bwd(j, lane, uhCl, dl, w, pC, ahC, sC, gl, uaC, nullptr, nullptr, nullptr, nullptr, nullptr, dg, dbe, db, n1, H, 0);
        else if (j < c1) dbn_col_bwd(j - H, lane, uhGl, gt, w, pG, ahG, sG, gl, uaG, uho, uy, uah, tt, m2, nullptr, nullptr, nullptr, B, H, 1);
        else if (j < c1 + H) dbn_col_dot(j - c1, lane, dl, hLC, gw, n1, H, 0);
        else dbn_vec_sum(lane, dl, n1, gc);   // the output bias: dc = sum_r dlogit
      });
      // weight gradient of the CE pass, its cotangent for the block below, the penalty pass's cotangent (l == 0: dD/dx [B][D] in gt0)
      float *nxtC = W.t1, *uxG = l > 0 ? W.gt1 : W.gt0;
      const DbnGemm gW = dbn_g_outer(uaC, xinC, K, dW, n1, H, K, 0), gC = dbn_g_dense_t(uaC, Wl, nxtC, n1, H, K), gG = dbn_g_dense_t(uaG, Wl, uxG, B, H, K);
      if (l > 0 && use_gp) L.gemm(gW, gC, gG);
      else if (l > 0) L.gemm(gW, gC);
      else if (use_gp) L.gemm(gW, gG);
      else L.gemm(gW);
      uhC = nxtC;   // the block below reads it in its column phase (writing ua into t0) before its own dx lands in t1 again
      uhG = uxG;    // likewise (its own copy goes to W.uh[l - 1])
    }
  }
  if (!use_gp) return;
  float* xbar = W.gt1;   // [B][D]
  {
    const float* g = W.gt0;
    float* gpr = W.gp_row;
    L.col(B, DBN_LAMBDA(int r, int lane) { dbn_gp_row(r, lane, g, xbar, gpr, B, D, gp_w); });
  }
  // reverse of the first backward, bottom block first
  const float* xb = xbar;   // adjoint of the block's ux, [B][K]
  for (int l = 0; l < nb; ++l) {
    const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l);
    const int K = N.in_of(l);
    float *dW = G + N.off_W(l), *dg = G + N.off_g(l);
    const float *ua = W.ua[l], *xbl = xb;
    float* uabar = W.gt0;
    L.gemm(dbn_g_outer(ua, xbl, K, dW, B, H, K, 1),               // ux = ua W: its weight gradient
           dbn_g_dense(xbl, K, Wl, nullptr, uabar, B, H, K));      // uabar = xbar W^T   (both read xbar only: one launch)
    const float *tt = W.tt[l], *s = W.gs[l], *ah = W.gah[l], *uah = W.uah[l], *uy = W.uy[l], *uhl = W.uh[l], *p = W.gp[l], *h = W.gh[l], *m2 = W.m2[l];
    float *yb = W.ybar[l], *ahb = W.ahbar[l], *sb = W.sbar[l], *up = W.gt1;   // xbar (gt1) was consumed by the launch above
    L.col(H, DBN_LAMBDA(int j, int lane) { dbn_col_rev(j, lane, uabar, tt, s, ah, uah, uy, uhl, p, h, m2, gl, yb, ahb, sb, up, dg, B, H, act); });
    xb = up;
  }
  // and down the forward graph; uh_L = gate w: the head weights' share rides in the top block's column launch
  {
    float* gw = G + N.off_w();
    const float* topx = xb;
    const float* hbar = nullptr;
    for (int l = nb - 1; l >= 0; --l) {
      const float *gl = N.P + N.off_g(l), *Wl = N.P + N.off_W(l), *p = W.gp[l], *ah = W.gah[l], *ch = W.gch[l], *s = W.gs[l];
      const float *yb = W.ybar[l], *ahb = W.ahbar[l], *sb = W.sbar[l], *xin = l > 0 ? W.gh[l - 1] : W.XH;
      float *ab = W.gt0, *dg = G + N.off_g(l), *dbe = G + N.off_be(l), *db = G + N.off_b(l), *dW = G + N.off_W(l);
      const int K = N.in_of(l), top = l == nb - 1 ? 1 : 0;
      const float* hb = hbar;
      L.col(top ? 2 * H : H, DBN_LAMBDA(int j, int lane) {
        if (j < H) dbn_col_down(j, lane, yb, hb, p, ah, ahb, ch, s, sb, gl, ab, dg, dbe, db, B, H);
        else dbn_col_dot(j - H, lane, gt, topx, gw, B, H, 1);   // reads gt1 before the launch below rewrites it
      });
      if (l > 0) {
        float* nxt = W.gt1;
        L.gemm(dbn_g_outer(ab, xin, K, dW, B, H, K, 1), dbn_g_dense_t(ab, Wl, nxt, B, H, K));
        hbar = nxt;
      } else {
        L.gemm(dbn_g_outer(ab, xin, K, dW, B, H, K, 1));
      }
    }
  }
}

// The step's last launch: statistics (adv_irl.py:205-216: stats[0] = mean BCE, [1] = accuracy, [2] = mean (|g| - 1)^2), the two deferred
// running-statistics updates, and Adam(lr, betas = (b1, 0.999)) over every parameter, step count t (1-based) — adv_irl.py:75-77.  Three
// ranges of "columns": one wavefront for the statistics, then lanes over (block, column), then lanes over the parameters.
template <class LN>
void dbn_finish(LN& L, const DbnNet& N, const DbnWs& W, int B, int use_gp, float* stats, float lr, float b1, int t) {
  const float *ce = W.ce_row, *co = W.correct, *gp = W.gp_row, *bstat = W.bstat;
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
  const float step = (float)((double)lr / bc1), bc2s = (float)sqrt(bc2);
  float *P = N.P, *G = N.G, *M = N.M, *V = N.V, *rmean = N.rmean, *rvar = N.rvar;
  const int nbh = N.nblk * N.H, np = N.n_params(), nblk = N.nblk, H = N.H;
  const int cr = (nbh + DBN_LANES - 1) / DBN_LANES, ca = (np + DBN_LANES - 1) / DBN_LANES;
  L.col(1 + cr + ca, DBN_LAMBDA(int j, int lane) {
    if (j == 0) {   // one wavefront: lanes split the rows
      float a = 0.0f, b = 0.0f, c = 0.0f;
      for (int r = lane; r < 2 * B; r += DBN_LANES) { a += ce[r]; b += co[r]; }
      if (use_gp) for (int r = lane; r < B; r += DBN_LANES) c += gp[r];
      a = dbn_wsum(a); b = dbn_wsum(b); c = dbn_wsum(c);
      if (lane == 0) { stats[0] = a / (float)(2 * B); stats[1] = b / (float)(2 * B); stats[2] = use_gp ? c / (float)B : 0.0f; }
    } else if (j <= cr) {
      const int idx = (j - 1) * DBN_LANES + lane;
      if (idx < nbh) dbn_running_update(idx, rmean, rvar, bstat, nblk, H, 2 * B, use_gp ? B : 0);
    } else {
      const int i = (j - 1 - cr) * DBN_LANES + lane;
      if (i < np) dbn_adam(i, P, G, M, V, step, bc2s, b1, 0.999f, 1e-8f);
    }
  });
}
