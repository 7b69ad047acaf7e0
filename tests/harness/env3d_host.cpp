// TEST HARNESS, not product: compiles ilswiss_amd/csrc/env3d.h (the device code of the 3-D stepper) for the HOST so the CPU suite
// can check its recursions (CRBA / RNE / Cholesky / PGS / RK4 / observation) against oracle/spatial_env.py without a GPU.  Built
// by tests/test_env3d_host.py into a temp dir; nothing in ilswiss_amd/ loads it.  One "lane" at a time: n_env = 1, env = 0.
#define __device__
#define __forceinline__ inline
#define E3W_HOST_EMU 1
#include <vector>
#include "../../ilswiss_amd/csrc/env3d.h"
#include "../../ilswiss_amd/csrc/env3d_wave.h"

int e3w_reverse = 0;   // order in which the emulated "parallel" loops of env3d_wave.h run: 0 ascending, 1 descending

extern "C" int e3h_scratch_doubles() { return E3Off::TOTAL; }
extern "C" int e3h_obs_dim(const ilsx_spatial_model* sm) {
  Spatial3Dev m;
  return e3_build_model(sm, m) ? -1 : m.obs_dim;
}
// q [nq], v [nv] in/out; act [n_act] float; obs [obs_dim] out; returns 0, or -1 when the model is refused
extern "C" int e3h_step(const ilsx_spatial_model* sm, double* q, double* v, const float* act, double* obs, double* reward, int* done) {
  static Spatial3Dev m;
  if (e3_build_model(sm, m)) return -1;
  std::vector<double> buf(E3Off::TOTAL, 0.0);
  double* scr = buf.data();
  const int n_env = 1, env = 0;
  const E3Ctx C{scr, n_env, env, &m};
  for (int i = 0; i < m.nq; ++i) E3S(E3St::Q0 + i) = q[i];
  for (int i = 0; i < m.nv; ++i) E3S(E3St::V0 + i) = v[i];
  bool d; double r;
  e3_task_step(C, act, r, d);
  e3_observe(C, [&](int i, double val) { obs[i] = val; });
  *reward = r; *done = d ? 1 : 0;
  for (int i = 0; i < m.nq; ++i) q[i] = E3S(E3St::Q0 + i);
  for (int i = 0; i < m.nv; ++i) v[i] = E3S(E3St::V0 + i);
  return 0;
}
// acceleration of (q, v, ctrl) — the quantity every stage of the integrator evaluates
extern "C" int e3h_qacc(const ilsx_spatial_model* sm, const double* q, const double* v, const double* ctrl, double* qacc) {
  static Spatial3Dev m;
  if (e3_build_model(sm, m)) return -1;
  std::vector<double> buf(E3Off::TOTAL, 0.0);
  double* scr = buf.data();
  const int n_env = 1, env = 0;
  const E3Ctx C{scr, n_env, env, &m};
  for (int i = 0; i < m.nq; ++i) E3S(E3St::Q0 + i) = q[i];
  for (int i = 0; i < m.nv; ++i) E3S(E3St::V0 + i) = v[i];
  for (int k = 0; k < m.n_act; ++k) E3S(E3St::CTRL + k) = ctrl[k];
  e3_dynamics(C, E3St::Q0, E3St::V0, E3St::CTRL, E3St::ACC);
  for (int i = 0; i < m.nv; ++i) qacc[i] = E3S(E3St::ACC + i);
  return 0;
}

// ---- the wave-per-env form (env3d_wave.h) under host emulation: parallel loops run serially, ascending or descending
extern "C" int e3hw_scratch_doubles() { return E3WOff::TOTAL; }
extern "C" int e3hw_step(const ilsx_spatial_model* sm, int reverse, double* q, double* v, const float* act, double* obs, double* reward, int* done) {
  static Spatial3Dev m;
  if (e3_build_model(sm, m)) return -1;
  e3w_reverse = reverse;
  std::vector<double> S(E3WOff::TOTAL, 0.0);
  const int lane = 0;
  for (int i = 0; i < m.nq; ++i) S[E3WOff::Q0 + i] = q[i];
  for (int i = 0; i < m.nv; ++i) S[E3WOff::V0 + i] = v[i];
  bool d; double r;
  E3WRegs regs[64];
  for (int ln = 0; ln < 64; ++ln) e3w_regs_init(regs[ln], m, ln);
  const int snv = reverse >> 1;   // bit 1: take the dof count at compile time (the device's Humanoid / Ant instantiations)
  e3w_reverse = reverse & 1;
  if (snv && m.nv == 23) e3w_task_step<23>(S.data(), m, lane, regs, act, r, d);
  else if (snv && m.nv == 14) e3w_task_step<14>(S.data(), m, lane, regs, act, r, d);
  else e3w_task_step<0>(S.data(), m, lane, regs, act, r, d);
  for (int i = 0; i < m.obs_dim; ++i) obs[i] = -12345.0;
  e3w_observe(S.data(), m, lane, [&](int i, double val) { obs[i] = val; });
  *reward = r; *done = d ? 1 : 0;
  for (int i = 0; i < m.nq; ++i) q[i] = S[E3WOff::Q0 + i];
  for (int i = 0; i < m.nv; ++i) v[i] = S[E3WOff::V0 + i];
  return 0;
}
extern "C" int e3hw_qacc(const ilsx_spatial_model* sm, int reverse, const double* q, const double* v, const double* ctrl, double* qacc) {
  static Spatial3Dev m;
  if (e3_build_model(sm, m)) return -1;
  e3w_reverse = reverse;
  std::vector<double> S(E3WOff::TOTAL, 0.0);
  const int lane = 0;
  for (int i = 0; i < m.nq; ++i) S[E3WOff::Q0 + i] = q[i];
  for (int i = 0; i < m.nv; ++i) S[E3WOff::V0 + i] = v[i];
  for (int k = 0; k < m.n_act; ++k) S[E3WOff::CTRL + k] = ctrl[k];
  E3WRegs regs[64];
  for (int ln = 0; ln < 64; ++ln) e3w_regs_init(regs[ln], m, ln);
  const int snv = reverse >> 1;
  e3w_reverse = reverse & 1;
  if (snv && m.nv == 23) e3w_dynamics<23>(S.data(), m, lane, regs, E3WOff::Q0, E3WOff::V0, E3WOff::CTRL, E3WOff::ACC);
  else if (snv && m.nv == 14) e3w_dynamics<14>(S.data(), m, lane, regs, E3WOff::Q0, E3WOff::V0, E3WOff::CTRL, E3WOff::ACC);
  else e3w_dynamics<0>(S.data(), m, lane, regs, E3WOff::Q0, E3WOff::V0, E3WOff::CTRL, E3WOff::ACC);
  for (int i = 0; i < m.nv; ++i) qacc[i] = S[E3WOff::ACC + i];
  return 0;
}
