// TEST HARNESS, not product: compiles ilswiss_amd/csrc/disc_bn.h + disc_bn_step.h (the phases of the BatchNorm discriminator step and their
// order) for the HOST, every phase as a serial loop, so that the CPU suite can check them against oracle/disc.py:DiscBNOracle and the
// reference's vectors (tests/golden/g26_disc_bn.npz) without a GPU.  Built by tests/test_disc_bn_host.py into a temp dir; nothing in
// ilswiss_amd/ loads it.
#define DBN_HOST_EMU 1
#include <cstring>
#include <vector>
#include "../../ilswiss_amd/csrc/disc_bn_step.h"

struct HostLaunch {
  template <class F> void par(int n, F f) { for (int i = 0; i < n; ++i) f(i); }
  template <class F> void col(int H, F f) { for (int j = 0; j < H; ++j) f(j, 0); }
  void gemm(const DbnGemm& g) { for (int i = 0; i < g.M; ++i) for (int j = 0; j < g.N; ++j) dbn_gemm_elem(g, i, j); }
  void gemm(const DbnGemm& g1, const DbnGemm& g2) { gemm(g1); gemm(g2); }
  void gemm(const DbnGemm& g1, const DbnGemm& g2, const DbnGemm& g3) { gemm(g1); gemm(g2); gemm(g3); }
};

struct HostDisc {
  DbnNet N;
  DbnWs W;
  std::vector<std::vector<float>> store;
  int maxB, t = 0;
  float* buf(size_t n) { store.emplace_back(n, 0.0f); return store.back().data(); }
};

extern "C" void* dbnh_create(int D, int H, int nblk, int act, float clampv, int max_batch) {
  HostDisc* d = new HostDisc();
  DbnNet& N = d->N;
  N.D = D; N.H = H; N.nblk = nblk; N.act = act; N.clampv = clampv;
  const int np = N.n_params();
  N.P = d->buf(np); N.G = d->buf(np); N.M = d->buf(np); N.V = d->buf(np);
  N.rmean = d->buf((size_t)nblk * H); N.rvar = d->buf((size_t)nblk * H);
  for (int i = 0; i < nblk * H; ++i) N.rvar[i] = 1.0f;
  d->maxB = max_batch;
  const size_t n2 = 2 * (size_t)max_batch, w = (size_t)(H > D ? H : D);
  DbnWs& W = d->W;
  W.X = d->buf(n2 * D); W.XH = d->buf(n2 * D);
  for (int l = 0; l < nblk; ++l) {
    W.ch[l] = d->buf(n2 * H); W.ah[l] = d->buf(n2 * H); W.h[l] = d->buf(n2 * H); W.p[l] = d->buf(n2 * H); W.s[l] = d->buf(H);
    W.gch[l] = d->buf(n2 * H); W.gah[l] = d->buf(n2 * H); W.gh[l] = d->buf(n2 * H); W.gp[l] = d->buf(n2 * H); W.gs[l] = d->buf(H);
    W.uh[l] = d->buf(n2 * H); W.uy[l] = d->buf(n2 * H); W.uah[l] = d->buf(n2 * H); W.tt[l] = d->buf(n2 * H); W.ua[l] = d->buf(n2 * H); W.m2[l] = d->buf(H);
    W.ybar[l] = d->buf(n2 * H); W.ahbar[l] = d->buf(n2 * H); W.sbar[l] = d->buf(H);
  }
  W.t0 = d->buf(n2 * w); W.t1 = d->buf(n2 * w); W.gt0 = d->buf(n2 * w); W.gt1 = d->buf(n2 * w);
  W.bstat = d->buf((size_t)2 * nblk * 2 * H);
  W.logit = d->buf(n2); W.dlogit = d->buf(n2); W.gate = d->buf(n2); W.ce_row = d->buf(n2); W.correct = d->buf(n2); W.gp_row = d->buf(n2);
  return d;
}
extern "C" void dbnh_destroy(void* h) { delete (HostDisc*)h; }
extern "C" int dbnh_num_params(void* h) { return ((HostDisc*)h)->N.n_params(); }
extern "C" void dbnh_set_params(void* h, const float* flat) { HostDisc* d = (HostDisc*)h; memcpy(d->N.P, flat, sizeof(float) * d->N.n_params()); }
extern "C" void dbnh_get(void* h, float* params, float* grads, float* rmean, float* rvar) {
  HostDisc* d = (HostDisc*)h;
  const int np = d->N.n_params(), nb = d->N.nblk * d->N.H;
  if (params) memcpy(params, d->N.P, sizeof(float) * np);
  if (grads) memcpy(grads, d->N.G, sizeof(float) * np);
  if (rmean) memcpy(rmean, d->N.rmean, sizeof(float) * nb);
  if (rvar) memcpy(rvar, d->N.rvar, sizeof(float) * nb);
}
// one _do_reward_training step on explicit batches; stats3 = {ce, accuracy, mean (|g| - 1)^2}
extern "C" int dbnh_train_step(void* h, const float* eo, const float* ea, const float* po, const float* pa, const float* eps, int B, int o, int a,
                               int use_gp, float gp_w, float lr, float b1, float* stats3) {
  HostDisc* d = (HostDisc*)h;
  if (B > d->maxB || o + a != d->N.D) return -1;
  HostLaunch L;
  float *X = d->W.X, *XH = use_gp ? d->W.XH : nullptr;
  L.par(B * d->N.D, [=](int idx) { dbn_prep(idx, eo, ea, po, pa, eps, X, XH, B, o, a); });
  dbn_backward(L, d->N, d->W, B, use_gp, gp_w);
  dbn_finish(L, d->N, d->W, B, use_gp, stats3, lr, b1, ++d->t);
  return 0;
}
extern "C" int dbnh_logits_eval(void* h, const float* x, int n, float* logits) {
  HostDisc* d = (HostDisc*)h;
  if (n > 2 * d->maxB) return -1;
  HostLaunch L;
  dbn_logits_eval(L, d->N, d->W, x, n, logits);
  return 0;
}
