// disc_bn.h — the discriminator step with BatchNorm1d blocks: MLPDisc(use_bn=True), the reference constructor's DEFAULT
// (rlkit/torch/algorithms/adv_irl/disc_models/simple_disc_models.py:15,30-31,36-37), trained by AdvIRL._do_reward_training
// (adv_irl.py:133-216) with the module in train mode and evaluated in eval mode by _do_policy_training (adv_irl.py:268-274).
//
// Every block is Linear -> BatchNorm1d -> act.  Batch statistics couple the rows of a batch, so the step is a sequence of PHASES that are
// each embarrassingly parallel over rows x features, over features (one wavefront per feature COLUMN: lanes split the rows, the column's
// means are wave reductions) or over a weight matrix; a phase is one launch.  The gradient penalty's gradient is a double backward
// THROUGH the batch statistics, derived by hand (reverse over reverse; oracle/disc.py:DiscBNOracle states it in numpy and is pinned to
// the reference's autograd by tests/golden/g26_disc_bn.npz):
//
//   forward (n rows)        a = x W^T + b ; mu, var = column mean / biased variance ; s = (var + eps)^-1/2 ; ch = a - mu ; ah = ch s ;
//                           y = gamma ah + beta ; h = phi(y) ; p = phi'(y)
//   backward (CE and the first backward of the penalty, cotangent uh of h)
//                           uy = uh p ; uah = uy gamma ; m1 = mean(uah) ; m2 = mean(uah ah) ; tt = uah - m1 - ah m2 ; ua = s tt ; ux = ua W
//   reverse of that (xbar = adjoint of ux, bottom block first)
//                           W += ua^T xbar ; uabar = xbar W^T ; sbar = sum(uabar tt) ; ttbar = uabar s ; m1bar = -sum ttbar ;
//                           m2bar = -sum(ttbar ah) ; uahbar = ttbar + m1bar / n + ah m2bar / n ; ahbar = -ttbar m2 + uah m2bar / n ;
//                           gamma += sum(uahbar uy) ; uybar = uahbar gamma ; ybar = uybar uh phi''(y) ; xbar' = uybar p  (next block up)
//   and down the forward graph (hbar = adjoint of h from the block above)
//                           yb = ybar + hbar p ; gamma += sum(yb ah) ; beta += sum yb ; ahb = ahbar + yb gamma ; sb = sbar + sum(ahb ch) ;
//                           chb = ahb s - sb s^3 ch / n ; ab = chb - mean(chb) ; W += ab^T x ; b += sum ab ; hbar' = ab W
//
// This path is simple fp32 FMA code, not MFMA tiles: it exists so that the constructor default WORKS (no exp_spec of the reference turns
// batch norm on); the use_bn=False path keeps the fused kernels.  Natural (row-major) parameter layout, any hid_dim.
// The same text compiles for the host (tests/harness/disc_bn_host.cpp, DBN_HOST_EMU: every phase as a serial loop) so that the CPU
// suite checks the phases against the oracle and the reference's vectors without a GPU.
#pragma once
#include <math.h>

#ifdef DBN_HOST_EMU
#define DBN_HD inline
#define DBN_HDH inline
#define DBN_HOST inline
#define DBN_LANES 1
static inline float dbn_wsum(float v) { return v; }
#else
#define DBN_HD __device__ __forceinline__
#define DBN_HDH __host__ __device__ __forceinline__
#define DBN_HOST inline   // descriptors are built by the host-side step sequence (disc_bn_step.h)
#define DBN_LANES 64
__device__ __forceinline__ float dbn_wsum(float v) {   // all 64 lanes active
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif

#define DBN_EPS 1e-5f
#define DBN_MOM 0.1f
enum { DBN_RELU = 0, DBN_TANH = 1 };

DBN_HD float dbn_act(float z, int act) { return act == DBN_RELU ? fmaxf(z, 0.0f) : tanhf(z); }
DBN_HD float dbn_dact(float h, int act) { return act == DBN_RELU ? (h > 0.0f ? 1.0f : 0.0f) : 1.0f - h * h; }
DBN_HD float dbn_d2act(float h, int act) { return act == DBN_RELU ? 0.0f : -2.0f * h * (1.0f - h * h); }   // phi''(y) written with h

// ---- dense phases: every matrix product of the step is ONE shape, C (+)= op(A) op(B) (+ bias), stated with element strides
//        C[i][j] = sum_k A[i sai + k sak] * B[k sbk + j sbj]        i < M, j < N, k < Kd ascending, one fmaf chain from 0.0f
//   forward / uabar = xbar W^T   out[r][j] = sum_k x[r][k] W[j][k] + b[j]     A = x (sai ldx, sak 1)   B = W (sbk 1, sbj K)
//   dx                           out[r][k] = sum_j d[r][j] W[j][k]            A = d (sai H, sak 1)     B = W (sbk K, sbj 1)
//   weight gradient              G[j][k] (+)= sum_r a[r][j] x[r][k]           A = a (sai 1, sak H)     B = x (sbk ldx, sbj 1)
// The launcher's `gemm` phase runs it: 16 x 16 tiles on the exact-fp32 matrix pipe on the device (k_dbn_gemm / dbn_gemm_tile, ilsx_disc.hip), a serial
// loop of dbn_gemm_elem on the host; the chain of one element is the same in both (four ranges, k ascending), so the two give the same bits.
struct DbnGemm {
  const float* A; int sai, sak;
  const float* B; int sbk, sbj;
  float* C; int ldc;
  const float* bias;   // nullable, per column j
  int M, N, Kd, acc;   // acc: C += (the penalty's weight gradients land on the cross-entropy pass's)
};
DBN_HOST DbnGemm dbn_g_dense(const float* x, int ldx, const float* W, const float* b, float* out, int n, int H, int K) {
  return DbnGemm{x, ldx, 1, W, 1, K, out, H, b, n, H, K, 0};
}
DBN_HOST DbnGemm dbn_g_dense_t(const float* d, const float* W, float* out, int n, int H, int K) {
  return DbnGemm{d, H, 1, W, K, 1, out, K, nullptr, n, K, H, 0};
}
DBN_HOST DbnGemm dbn_g_outer(const float* a, const float* x, int ldx, float* G, int n, int H, int K, int acc) {
  return DbnGemm{a, 1, H, x, ldx, 1, G, K, nullptr, H, K, n, acc};
}
// the contraction runs in FOUR consecutive ranges of dbn_kq(Kd) terms (one per wave of the device tile), each an fmaf chain from 0.0f in
// ascending k; the four partial sums are added in range order: c = ((p0 + p1) + p2) + p3 (+ bias).  dbn_gemm_elem states that chain for the
// host emulation, k_dbn_gemm (ilsx_disc.hip) runs it on a 16 x 16 tile — one v_mfma_f32_16x16x4_f32 per four terms, itself an fmaf chain over its
// k slots in ascending order: the two give the same bits.
DBN_HDH int dbn_kq(int Kd) { return (((Kd + 3) / 4) + 31) & ~31; }
DBN_HD void dbn_gemm_elem(const DbnGemm& g, int i, int j) {
  const int kq = dbn_kq(g.Kd);
  float s = 0.0f;
  for (int q = 0; q < 4; ++q) {
    float p = 0.0f;
    const int k1 = (q + 1) * kq < g.Kd ? (q + 1) * kq : g.Kd;
    for (int k = q * kq; k < k1; ++k) p = fmaf(g.A[(size_t)i * g.sai + (size_t)k * g.sak], g.B[(size_t)k * g.sbk + (size_t)j * g.sbj], p);
    s = q == 0 ? p : s + p;
  }
  float* c = g.C + (size_t)i * g.ldc + j;
  if (g.bias) s = s + g.bias[j];
  *c = g.acc ? *c + s : s;
}

// ---- column phases: one wavefront per feature column j; lanes split the rows (host emulation: one lane)
// A lane's rows are walked in BATCHES of DBN_U: every load of a batch is issued before the first value is used (a plain loop around
// load + use waited for memory on every trip: 8 serial L2 round trips per pass and array at 512 rows — the column launches ran 9 us each).
// When the column is ONE batch (n <= DBN_LANES * DBN_U: 512 rows on the device) the values stay in registers across the phase's passes — one
// memory round trip per launch instead of one per pass.  The order in which a lane adds its rows is unchanged (ascending), so are the results.
#ifdef DBN_HOST_EMU
#define DBN_U 64   // the host's single lane holds 64 rows per batch: the reference vectors (<= 40 rows) take the one-batch path, the oracle sizes the general one
#else
#define DBN_U 8
#endif
#define DBN_BATCH(rb) for (int rb = lane; rb < n; rb += DBN_LANES * DBN_U)
#define DBN_EACH(u, r, rb) _Pragma("unroll") for (int u = 0, r = rb; u < DBN_U; ++u, r += DBN_LANES)
// v[u] = x[r][j] for the batch's rows (0 past the end)
#define DBN_LD(v, x, rb) DBN_EACH(u_, r_, rb) v[u_] = r_ < n ? (x)[(size_t)r_ * H + j] : 0.0f
// forward of a block after the dense phase: a (in buf `ch`, overwritten by a - mu) -> s[j], ah, h, p; train: batch statistics (+ running
// update), eval: running statistics
// bstat (nullable, train): the column's batch mean and BIASED variance are parked at bstat[j], bstat[H + j] for dbn_running_update — the
// training step runs its two forward passes side by side and applies their running-statistics updates afterwards, in the reference's order
DBN_HD void dbn_col_fwd(int j, int lane, float* ch, float* ah, float* h, float* p, float* s_out, const float* gamma, const float* beta,
                        float* rmean, float* rvar, int n, int H, int act, int train, int update_running, float* bstat) {
  const bool one = n <= DBN_LANES * DBN_U;
  float x[DBN_U];
  float mu, var;
  if (train) {
    float sm = 0.0f;
    DBN_BATCH(rb) { DBN_LD(x, ch, rb); DBN_EACH(u, r, rb) if (r < n) sm += x[u]; }
    mu = dbn_wsum(sm) / (float)n;
    float sv = 0.0f;
    DBN_BATCH(rb) { if (!one) DBN_LD(x, ch, rb); DBN_EACH(u, r, rb) if (r < n) { const float c = x[u] - mu; sv += c * c; } }
    var = dbn_wsum(sv) / (float)n;
    if (update_running && lane == 0) {   // torch: running = (1 - momentum) running + momentum batch, the variance UNBIASED
      rmean[j] = (1.0f - DBN_MOM) * rmean[j] + DBN_MOM * mu;
      rvar[j] = (1.0f - DBN_MOM) * rvar[j] + DBN_MOM * var * ((float)n / (float)(n - 1));
    }
    if (bstat && lane == 0) { bstat[j] = mu; bstat[H + j] = var; }
  } else {
    mu = rmean[j]; var = rvar[j];
  }
  const float s = 1.0f / sqrtf(var + DBN_EPS), g = gamma[j], be = beta[j];
  if (lane == 0 && s_out) s_out[j] = s;
  DBN_BATCH(rb) {
    if (!one || !train) DBN_LD(x, ch, rb);
    DBN_EACH(u, r, rb) if (r < n) {
      const size_t at = (size_t)r * H + j;
      const float c = x[u] - mu, a_ = c * s, hh = dbn_act(g * a_ + be, act);
      ch[at] = c;
      if (ah) ah[at] = a_;
      h[at] = hh;
      if (p) p[at] = dbn_dact(hh, act);
    }
  }
}
// backward through act -> BN (the CE backward, and the FIRST backward of the penalty): cotangent uh of h -> ua of the dense output.
// uh == null: the top block, uh[r][j] = top[r] * w[j].  Optional outputs uy / uah / tt (the penalty's tape), m2 ; optional gradient
// accumulations dgamma (+)= sum(uy ah), dbeta (+)= sum(uy), db (+)= sum(ua)  (CE backward only; `acc` = add to what is there).
DBN_HD void dbn_col_bwd(int j, int lane, const float* uh, const float* top, const float* w, const float* p, const float* ah, const float* s,
                        const float* gamma, float* ua, float* uh_out, float* uy, float* uah, float* tt, float* m2_out, float* dgamma, float* dbeta,
                        float* db, int n, int H, int acc) {
  const bool one = n <= DBN_LANES * DBN_U;
  const float g = gamma[j], sj = s[j], wj = w ? w[j] : 0.0f;
  float u_[DBN_U], pp[DBN_U], aa[DBN_U];
  float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f;
  DBN_BATCH(rb) {
    DBN_EACH(u, r, rb) u_[u] = r < n ? (uh ? uh[(size_t)r * H + j] : top[r] * wj) : 0.0f;
    DBN_LD(pp, p, rb); DBN_LD(aa, ah, rb);
    DBN_EACH(u, r, rb) if (r < n) {
      const float y_ = u_[u] * pp[u], q = y_ * g;
      a1 += q; a2 += q * aa[u]; a3 += y_ * aa[u]; a4 += y_;
      if (uh_out) uh_out[(size_t)r * H + j] = u_[u];
    }
  }
  const float m1 = dbn_wsum(a1) / (float)n, m2 = dbn_wsum(a2) / (float)n;
  a3 = dbn_wsum(a3); a4 = dbn_wsum(a4);
  float sa = 0.0f;
  DBN_BATCH(rb) {
    if (!one) {
      DBN_EACH(u, r, rb) u_[u] = r < n ? (uh ? uh[(size_t)r * H + j] : top[r] * wj) : 0.0f;
      DBN_LD(pp, p, rb); DBN_LD(aa, ah, rb);
    }
    DBN_EACH(u, r, rb) if (r < n) {
      const size_t at = (size_t)r * H + j;
      const float y_ = u_[u] * pp[u], q = y_ * g, t = q - m1 - aa[u] * m2, a_ = sj * t;
      ua[at] = a_;
      sa += a_;
      if (uy) { uy[at] = y_; uah[at] = q; tt[at] = t; }
    }
  }
  sa = dbn_wsum(sa);
  if (lane == 0) {
    if (m2_out) m2_out[j] = m2;
    if (dgamma) {   // acc == 0: the cross-entropy pass comes first and ASSIGNS (no zeroing launch in front of the step)
      dgamma[j] = acc ? dgamma[j] + a3 : a3; dbeta[j] = acc ? dbeta[j] + a4 : a4; db[j] = acc ? db[j] + sa : sa;
    }
  }
}
// reverse of the first backward for one block: uabar (adjoint of ua) -> ybar, ahbar, sbar[j], xbar_up (adjoint of this block's uh) ; dgamma
DBN_HD void dbn_col_rev(int j, int lane, const float* uabar, const float* tt, const float* s, const float* ah, const float* uah, const float* uy,
                        const float* uh, const float* p, const float* h, const float* m2, const float* gamma, float* ybar, float* ahbar,
                        float* sbar, float* xbar_up, float* dgamma, int n, int H, int act) {
  const bool one = n <= DBN_LANES * DBN_U;
  const float sj = s[j], g = gamma[j], m2j = m2[j];
  float ub[DBN_U], t_[DBN_U], aa[DBN_U], ua_[DBN_U], uy_[DBN_U], uh_[DBN_U], hh[DBN_U], pp[DBN_U];
  float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  DBN_BATCH(rb) {
    DBN_LD(ub, uabar, rb); DBN_LD(t_, tt, rb); DBN_LD(aa, ah, rb);
    if (one) { DBN_LD(ua_, uah, rb); DBN_LD(uy_, uy, rb); DBN_LD(uh_, uh, rb); DBN_LD(hh, h, rb); DBN_LD(pp, p, rb); }   // the second pass's operands ride along
    DBN_EACH(u, r, rb) if (r < n) {
      const float tb = ub[u] * sj;
      a1 += ub[u] * t_[u]; a2 += tb; a3 += tb * aa[u];
    }
  }
  const float sb = dbn_wsum(a1), m1bar = -dbn_wsum(a2), m2bar = -dbn_wsum(a3);
  float dg = 0.0f;
  DBN_BATCH(rb) {
    if (!one) {
      DBN_LD(ub, uabar, rb); DBN_LD(aa, ah, rb); DBN_LD(ua_, uah, rb); DBN_LD(uy_, uy, rb);
      DBN_LD(uh_, uh, rb); DBN_LD(hh, h, rb); DBN_LD(pp, p, rb);
    }
    DBN_EACH(u, r, rb) if (r < n) {
      const size_t at = (size_t)r * H + j;
      const float tb = ub[u] * sj;
      const float uahb = tb + m1bar / (float)n + (m2bar / (float)n) * aa[u];
      ahbar[at] = -tb * m2j + (m2bar / (float)n) * ua_[u];
      dg += uahb * uy_[u];
      const float uyb = uahb * g;
      ybar[at] = uyb * uh_[u] * dbn_d2act(hh[u], act);
      xbar_up[at] = uyb * pp[u];
    }
  }
  dg = dbn_wsum(dg);
  if (lane == 0) { sbar[j] = sb; dgamma[j] += dg; }
}
// the adjoints of the forward quantities go down the forward graph: (ybar, ahbar, sbar, hbar from above [null at the top]) -> ab ; dgamma, dbeta, db
DBN_HD void dbn_col_down(int j, int lane, const float* ybar, const float* hbar, const float* p, const float* ah, const float* ahbar,
                         const float* ch, const float* s, const float* sbar, const float* gamma, float* ab, float* dgamma, float* dbeta,
                         float* db, int n, int H) {
  const bool one = n <= DBN_LANES * DBN_U;
  const float sj = s[j], g = gamma[j];
  float yb_[DBN_U], aa[DBN_U], ab_[DBN_U], cc[DBN_U], hp[DBN_U], cb[DBN_U];
  float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  DBN_BATCH(rb) {
    DBN_LD(yb_, ybar, rb); DBN_LD(aa, ah, rb); DBN_LD(ab_, ahbar, rb); DBN_LD(cc, ch, rb);
    DBN_EACH(u, r, rb) hp[u] = (hbar && r < n) ? hbar[(size_t)r * H + j] * p[(size_t)r * H + j] : 0.0f;
    DBN_EACH(u, r, rb) if (r < n) {
      const float yb = yb_[u] + (hbar ? hp[u] : 0.0f);
      const float ahb = ab_[u] + yb * g;
      a1 += yb * aa[u]; a2 += yb; a3 += ahb * cc[u];
    }
  }
  a1 = dbn_wsum(a1); a2 = dbn_wsum(a2);
  const float sb = sbar[j] + dbn_wsum(a3), vb = -0.5f * sb * sj * sj * sj;
  float cm = 0.0f;
  DBN_BATCH(rb) {
    if (!one) {
      DBN_LD(yb_, ybar, rb); DBN_LD(ab_, ahbar, rb); DBN_LD(cc, ch, rb);
      DBN_EACH(u, r, rb) hp[u] = (hbar && r < n) ? hbar[(size_t)r * H + j] * p[(size_t)r * H + j] : 0.0f;
    }
    DBN_EACH(u, r, rb) if (r < n) {
      const float yb = yb_[u] + (hbar ? hp[u] : 0.0f);
      const float ahb = ab_[u] + yb * g;
      const float chb = ahb * sj + vb * 2.0f * cc[u] / (float)n;
      if (!one) ab[(size_t)r * H + j] = chb;
      cb[u] = chb;
      cm += chb;
    }
  }
  cm = dbn_wsum(cm) / (float)n;
  float sa = 0.0f;
  DBN_BATCH(rb) {
    if (!one) DBN_LD(cb, ab, rb);
    DBN_EACH(u, r, rb) if (r < n) {
      const float v = cb[u] - cm;
      ab[(size_t)r * H + j] = v;
      sa += v;
    }
  }
  sa = dbn_wsum(sa);
  if (lane == 0) { dgamma[j] += a1; dbeta[j] += a2; db[j] += sa; }
}
// column sums for the head: gw[j] (+)= sum_r v[r] * m[r][j]       (v = dlogit with m = h_L ; v = gate with m = xbar_up)
DBN_HD void dbn_col_dot(int j, int lane, const float* v, const float* m, float* out, int n, int H, int acc) {
  float x[DBN_U];
  float a = 0.0f;
  DBN_BATCH(rb) {
    DBN_LD(x, m, rb);
    DBN_EACH(u, r, rb) if (r < n) a += v[r] * x[u];
  }
  a = dbn_wsum(a);
  if (lane == 0) out[j] = acc ? out[j] + a : a;
}
// sum of a vector as a column phase of ONE column (lanes split the rows): out[0] = sum_r v[r]
DBN_HD void dbn_vec_sum(int lane, const float* v, int n, float* out) {
  float a = 0.0f;
  for (int r = lane; r < n; r += DBN_LANES) a += v[r];
  a = dbn_wsum(a);
  if (lane == 0) out[0] = a;
}

// the running-statistics updates of a training step, after the fact: per (block, column) the cross-entropy pass's (n1 rows), then the
// penalty pass's (n2 rows; n2 = 0: no penalty pass) — the same two expressions, in the same order, as dbn_col_fwd(update_running = 1) applies
// them.  bstat = [pass][block][mean | var][H].        idx in [0, nblk * H)
DBN_HD void dbn_running_update(int idx, float* rmean, float* rvar, const float* bstat, int nblk, int H, int n1, int n2) {
  const int l = idx / H, j = idx - l * H;
  float rm = rmean[idx], rv = rvar[idx];
  for (int ps = 0; ps < (n2 > 0 ? 2 : 1); ++ps) {
    const float* bs = bstat + (size_t)((ps * nblk + l) * 2) * H;
    const int n = ps == 0 ? n1 : n2;
    rm = (1.0f - DBN_MOM) * rm + DBN_MOM * bs[j];
    rv = (1.0f - DBN_MOM) * rv + DBN_MOM * bs[H + j] * ((float)n / (float)(n - 1));
  }
  rmean[idx] = rm; rvar[idx] = rv;
}

// ---- row phases: one wavefront per ROW r, lanes split the features (coalesced; host emulation: one lane); dot = sum_j h[r][j] w[j]
DBN_HD float dbn_row_dot(int lane, const float* hr, const float* w, int H) {
  float a = 0.0f;
  for (int j = lane; j < H; j += DBN_LANES) a = fmaf(hr[j], w[j], a);
  return dbn_wsum(a);
}
// head of the cross-entropy pass (2B rows: the first B are expert rows, target 1): raw = h w + c ; clamp ; BCE-with-logits row term, accuracy,
// dlogit = (sigmoid(logit) - t) / (2B) * gate          r in [0, 2B)
DBN_HD void dbn_head_ce(int r, int lane, const float* h, const float* w, float c, float clampv, int B, int H, float* logit_out, float* dlogit,
                        float* ce_row, float* correct) {
  const float raw = dbn_row_dot(lane, h + (size_t)r * H, w, H) + c;
  if (lane != 0) return;
  const float x = fminf(fmaxf(raw, -clampv), clampv), t = r < B ? 1.0f : 0.0f;
  const float gate = (raw >= -clampv && raw <= clampv) ? 1.0f : 0.0f;
  ce_row[r] = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
  correct[r] = ((x > 0.0f ? 1.0f : 0.0f) == t) ? 1.0f : 0.0f;
  dlogit[r] = (1.0f / (1.0f + expf(-x)) - t) / (float)(2 * B) * gate;
  if (logit_out) logit_out[r] = x;
}
// head of a plain forward: clamped logit (+ the clamp's gate)
DBN_HD void dbn_head(int r, int lane, const float* h, const float* w, float c, float clampv, int H, float* logit, float* gate) {
  const float raw = dbn_row_dot(lane, h + (size_t)r * H, w, H) + c;
  if (lane != 0) return;
  if (logit) logit[r] = fminf(fmaxf(raw, -clampv), clampv);
  if (gate) gate[r] = (raw >= -clampv && raw <= clampv) ? 1.0f : 0.0f;
}
// interpolates xh = eps x_exp + (1 - eps) x_pol (adv_irl.py:187-189) and the stacked CE input [x_exp ; x_pol]        idx in [0, B * D)
DBN_HD void dbn_prep(int idx, const float* eo, const float* ea, const float* po, const float* pa, const float* eps, float* X, float* XH, int B, int o,
                     int a) {
  const int D = o + a, r = idx / D, k = idx - r * D;
  const float xe = k < o ? eo[(size_t)r * o + k] : ea[(size_t)r * a + (k - o)], xp = k < o ? po[(size_t)r * o + k] : pa[(size_t)r * a + (k - o)];
  X[(size_t)r * D + k] = xe;
  X[(size_t)(B + r) * D + k] = xp;
  if (XH) { const float e = eps[r]; XH[(size_t)r * D + k] = e * xe + (1.0f - e) * xp; }
}
// penalty rows: n = |g_r| ; row term (n - 1)^2 ; xbar = w_gp / B * 2 (n - 1) / n * g  (0 where n == 0)        one wavefront per row r in [0, B)
DBN_HD void dbn_gp_row(int r, int lane, const float* g, float* xbar, float* gp_row, int B, int D, float gp_w) {
  const float* gr = g + (size_t)r * D;
  const float ss = dbn_row_dot(lane, gr, gr, D);
  const float nn = sqrtf(ss);
  if (lane == 0) gp_row[r] = (nn - 1.0f) * (nn - 1.0f);
  const float coef = nn > 0.0f ? gp_w / (float)B * 2.0f * (nn - 1.0f) / nn : 0.0f;
  for (int k = lane; k < D; k += DBN_LANES) xbar[(size_t)r * D + k] = coef * gr[k];
}
// torch 1.9 Adam (no weight decay): elementwise                  idx in [0, n_params)
DBN_HD void dbn_adam(int i, float* P, const float* G, float* M, float* V, float lr_over_bc1, float bc2_sqrt, float b1, float b2, float eps) {
  const float g = G[i];
  const float m = M[i] * b1 + (1.0f - b1) * g;
  const float v = V[i] * b2 + (1.0f - b2) * g * g;
  M[i] = m; V[i] = v;
  P[i] = P[i] - lr_over_bc1 * (m / (sqrtf(v) / bc2_sqrt + eps));
}
