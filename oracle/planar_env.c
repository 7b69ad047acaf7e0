/* TEST INFRASTRUCTURE, not product: a scalar C restatement of oracle/planar_env.py (the planar articulated-body stepper the HIP
 * kernel k_env_step must reproduce), for ONE purpose — the CPU leg of bench.py's `cpu_baseline` for env-steps/s (SURVEY section 8d asks
 * for a scalar compiled env stepper on the host's cores; the numpy statement is ~1000x slower than compiled code and would say nothing
 * about a CPU).  Same dense formulation as the numpy file, line for line: dense Jacobians, M = sum m Jc^T Jc + I Jphi^T Jphi, Gaussian
 * elimination, constraint rows (capsule ends within the margin: normal + tangent; violated joint limits), projected Gauss-Seidel, RK4
 * with the constraint solve in every stage; reward / termination / observation rules of rlkit/envs/mujoco/hopper.py:11-40,
 * walker2d.py:11-36 and gym's HalfCheetah-v2.  Pinned against oracle/planar_env.py by tests/test_env_oracle.py (1e-9 over chained
 * steps).  Only tests/ and bench.py's cpu_baseline load it; nothing under ilswiss_amd/ does.  Build: make -C oracle.
 * The model arrives as the C-ABI struct of include/ilsx.h (the boundary's own type), filled by the test / bench. */
#include <math.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/ilsx.h"

#define NMAX (ILSX_ENV_MAX_BODY + 2)
#define RMAX 16

static double impedance(double r_abs, const double* solimp) {
  const double d0 = solimp[0], dmax = solimp[1], width = solimp[2];
  const double x = width > 0 ? fmin(r_abs / width, 1.0) : 1.0;
  const double y = x < 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return d0 + y * (dmax - d0);
}

/* solve M X = B for nrhs right-hand sides (Gaussian elimination with partial pivoting, like numpy.linalg.solve) */
static void solve(int n, const double M[NMAX][NMAX], int nrhs, double B[][NMAX], double X[][NMAX]) {
  double A[NMAX][NMAX];
  int piv[NMAX];
  memcpy(A, M, sizeof A);
  for (int i = 0; i < n; ++i) piv[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i) if (fabs(A[i][k]) > fabs(A[p][k])) p = i;
    if (p != k) { for (int j = 0; j < n; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; } int t = piv[k]; piv[k] = piv[p]; piv[p] = t; }
    for (int i = k + 1; i < n; ++i) {
      A[i][k] /= A[k][k];
      for (int j = k + 1; j < n; ++j) A[i][j] -= A[i][k] * A[k][j];
    }
  }
  for (int r = 0; r < nrhs; ++r) {
    double y[NMAX];
    for (int i = 0; i < n; ++i) { double s = B[r][piv[i]]; for (int j = 0; j < i; ++j) s -= A[i][j] * y[j]; y[i] = s; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= A[i][j] * X[r][j]; X[r][i] = s / A[i][i]; }
  }
}

typedef struct { double J[NMAX], r, rdist, mu; int kind; const double *solref, *solimp; } Row;   /* kind 0 normal, 1 tangent, 2 limit */

/* oracle/planar_env.py: kin() + dynamics() */
static void dynamics(const ilsx_planar_model* m, const double* q, const double* v, const double* ctrl, double* qacc) {
  const int nb = m->n_body, n = nb + 2;
  double phi[ILSX_ENV_MAX_BODY], phid[ILSX_ENV_MAX_BODY], Jphi[ILSX_ENV_MAX_BODY][NMAX], Jo[ILSX_ENV_MAX_BODY][2][NMAX], o[ILSX_ENV_MAX_BODY][2],
      ao[ILSX_ENV_MAX_BODY][2];
  memset(Jphi, 0, sizeof Jphi); memset(Jo, 0, sizeof Jo); memset(ao, 0, sizeof ao);
  for (int b = 0; b < nb; ++b) {
    const int p = m->parent[b];
    if (p < 0) {
      phi[b] = m->jsign[b] * q[2];
      Jphi[b][2] = m->jsign[b];
      o[b][0] = q[0]; o[b][1] = q[1];
      Jo[b][0][0] = 1.0; Jo[b][1][1] = 1.0;
    } else {
      phi[b] = phi[p] + m->jsign[b] * q[2 + b];
      memcpy(Jphi[b], Jphi[p], sizeof Jphi[b]); Jphi[b][2 + b] += m->jsign[b];
      const double c = cos(phi[p]), s = sin(phi[p]), ax = m->anchor[b][0], az = m->anchor[b][1];
      const double rx = c * ax - s * az, rz = s * ax + c * az;          /* rot(phi_p) a */
      const double dx = -s * ax - c * az, dz = c * ax - s * az;         /* drot(phi_p) a */
      o[b][0] = o[p][0] + rx; o[b][1] = o[p][1] + rz;
      double pd = 0.0;
      for (int i = 0; i < n; ++i) { Jo[b][0][i] = Jo[p][0][i] + dx * Jphi[p][i]; Jo[b][1][i] = Jo[p][1][i] + dz * Jphi[p][i]; pd += Jphi[p][i] * v[i]; }
      ao[b][0] = ao[p][0] - pd * pd * rx; ao[b][1] = ao[p][1] - pd * pd * rz;
    }
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += Jphi[b][i] * v[i];
    phid[b] = s;
  }
  double M[NMAX][NMAX], rhs[1][NMAX], q0[1][NMAX];
  memset(M, 0, sizeof M); memset(rhs, 0, sizeof rhs);
  for (int b = 0; b < nb; ++b) {
    const double c = cos(phi[b]), s = sin(phi[b]), rx0 = m->com[b][0], rz0 = m->com[b][1];
    const double wx = c * rx0 - s * rz0, wz = s * rx0 + c * rz0, dx = -s * rx0 - c * rz0, dz = c * rx0 - s * rz0;
    double Jc[2][NMAX];
    for (int i = 0; i < n; ++i) { Jc[0][i] = Jo[b][0][i] + dx * Jphi[b][i]; Jc[1][i] = Jo[b][1][i] + dz * Jphi[b][i]; }
    const double acx = ao[b][0] - phid[b] * phid[b] * wx, acz = ao[b][1] - phid[b] * phid[b] * wz;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) M[i][j] += m->mass[b] * (Jc[0][i] * Jc[0][j] + Jc[1][i] * Jc[1][j]) + m->inertia[b] * Jphi[b][i] * Jphi[b][j];
      rhs[0][i] += m->mass[b] * (Jc[0][i] * (0.0 - acx) + Jc[1][i] * (-m->gravity - acz));
    }
  }
  int k = 0;
  for (int b = 0; b < nb; ++b) {
    M[2 + b][2 + b] += m->armature[b];
    rhs[0][2 + b] -= m->damping[b] * v[2 + b] + m->stiffness[b] * q[2 + b];
    if (m->gear[b] != 0.0) rhs[0][2 + b] += m->gear[b] * ctrl[k++];     /* actuator order = body order */
  }
  solve(n, M, 1, rhs, q0);
  /* ---- constraint rows: distal geoms first (p1, p2), then limits */
  Row rows[RMAX];
  int nr = 0;
  const int max_rows = m->max_rows ? m->max_rows : (nb == 4 ? 8 : 12);
  for (int gi = m->n_geom - 1; gi >= 0; --gi) {
    const int b = m->geom_body[gi];
    const double c = cos(phi[b]), s = sin(phi[b]);
    for (int e = 0; e < 2; ++e) {
      const double* pe = e == 0 ? m->geom_p1[gi] : m->geom_p2[gi];
      const double wx = c * pe[0] - s * pe[1], wz = s * pe[0] + c * pe[1], rad = m->geom_radius[gi];
      const double dist = o[b][1] + wz - rad;
      if (dist < m->contact_margin && nr + 2 <= max_rows) {
        const double cx = wx, cz = wz - (rad + 0.5 * dist);
        const double mu = fmax(m->geom_friction[gi], 0.0);
        Row *rn = &rows[nr], *rt = &rows[nr + 1];
        for (int i = 0; i < n; ++i) { rn->J[i] = Jo[b][1][i] + cx * Jphi[b][i]; rt->J[i] = Jo[b][0][i] - cz * Jphi[b][i]; }
        rn->r = dist; rn->rdist = dist; rn->kind = 0; rn->mu = mu; rn->solref = m->contact_solref; rn->solimp = m->contact_solimp;
        rt->r = 0.0; rt->rdist = dist; rt->kind = 1; rt->mu = mu; rt->solref = m->contact_solref; rt->solimp = m->contact_solimp;
        nr += 2;
      }
    }
  }
  for (int b = m->limited[0] == 0 ? 1 : 0; b < nb; ++b) {
    if (!m->limited[b] || nr + 1 > max_rows) continue;
    const double lo = m->range[b][0], hi = m->range[b][1];
    double sgn = 0.0, r = 0.0;
    if (q[2 + b] - lo < 0.0) { sgn = 1.0; r = q[2 + b] - lo; }
    else if (hi - q[2 + b] < 0.0) { sgn = -1.0; r = hi - q[2 + b]; }
    if (sgn == 0.0) continue;
    Row* rl = &rows[nr++];
    memset(rl->J, 0, sizeof rl->J);
    rl->J[2 + b] = sgn; rl->r = r; rl->rdist = r; rl->kind = 2; rl->mu = 0.0; rl->solref = m->limit_solref; rl->solimp = m->limit_solimp;
  }
  if (nr == 0) { memcpy(qacc, q0[0], n * sizeof(double)); return; }
  double JT[RMAX][NMAX], MiJ[RMAX][NMAX], A[RMAX][RMAX], R[RMAX], rc[RMAX], f[RMAX];
  for (int r = 0; r < nr; ++r) memcpy(JT[r], rows[r].J, sizeof JT[r]);
  solve(n, M, nr, JT, MiJ);
  for (int r = 0; r < nr; ++r)
    for (int c2 = 0; c2 < nr; ++c2) { double s = 0.0; for (int i = 0; i < n; ++i) s += rows[r].J[i] * MiJ[c2][i]; A[r][c2] = s; }
  for (int r = 0; r < nr; ++r) {
    const double tcs = rows[r].solref[0], drs = rows[r].solref[1], dmax = rows[r].solimp[1];
    const double d = impedance(fabs(rows[r].rdist), rows[r].solimp);
    const double bdamp = 2.0 / (dmax * tcs), kst = 1.0 / (dmax * dmax * tcs * tcs * drs * drs);
    double jv = 0.0, jq = 0.0;
    for (int i = 0; i < n; ++i) { jv += rows[r].J[i] * v[i]; jq += rows[r].J[i] * q0[0][i]; }
    R[r] = (1.0 - d) / d * A[r][r];
    rc[r] = (-bdamp * jv - kst * d * rows[r].r) - jq;
    f[r] = 0.0;
  }
  for (int it = 0; it < m->pgs_iters; ++it)
    for (int r = 0; r < nr; ++r) {
      double res = rc[r];
      for (int c2 = 0; c2 < nr; ++c2) res -= A[r][c2] * f[c2];
      res += A[r][r] * f[r];
      double fi = res / (A[r][r] + R[r]);
      if (rows[r].kind == 1) { const double lim = rows[r].mu * f[r - 1]; fi = fmin(fmax(fi, -lim), lim); }
      else fi = fmax(fi, 0.0);
      f[r] = fi;
    }
  for (int i = 0; i < n; ++i) { double s = q0[0][i]; for (int r = 0; r < nr; ++r) s += MiJ[r][i] * f[r]; qacc[i] = s; }
}

static void substep(const ilsx_planar_model* m, double* q, double* v, const double* ctrl) {
  const int n = m->n_body + 2;
  const double h = m->timestep;
  double a1[NMAX], a2[NMAX], a3[NMAX], a4[NMAX], q2[NMAX], v2[NMAX], q3[NMAX], v3[NMAX], q4[NMAX], v4[NMAX];
  dynamics(m, q, v, ctrl, a1);
  for (int i = 0; i < n; ++i) { q2[i] = q[i] + 0.5 * h * v[i]; v2[i] = v[i] + 0.5 * h * a1[i]; }
  dynamics(m, q2, v2, ctrl, a2);
  for (int i = 0; i < n; ++i) { q3[i] = q[i] + 0.5 * h * v2[i]; v3[i] = v[i] + 0.5 * h * a2[i]; }
  dynamics(m, q3, v3, ctrl, a3);
  for (int i = 0; i < n; ++i) { q4[i] = q[i] + h * v3[i]; v4[i] = v[i] + h * a3[i]; }
  dynamics(m, q4, v4, ctrl, a4);
  for (int i = 0; i < n; ++i) {
    q[i] += h / 6.0 * (v[i] + 2 * v2[i] + 2 * v3[i] + v4[i]);
    v[i] += h / 6.0 * (a1[i] + 2 * a2[i] + 2 * a3[i] + a4[i]);
  }
}

/* oracle/planar_env.py step(): q, v [n] in / out, action [n_act], obs [2n - 1] */
int orc_planar_step(const ilsx_planar_model* m, double* q, double* v, const double* action, double* obs, double* reward, int* done) {
  const int nb = m->n_body, n = nb + 2;
  double a[ILSX_ENV_MAX_BODY], asq = 0.0;
  int na = 0;
  for (int b = 0; b < nb; ++b) if (m->gear[b] != 0.0) ++na;
  for (int k = 0; k < na; ++k) { a[k] = fmin(fmax(action[k], -1.0), 1.0); asq += a[k] * a[k]; }
  const double x0 = q[0];
  for (int s = 0; s < m->frame_skip; ++s) substep(m, q, v, a);
  const double dt = m->timestep * m->frame_skip;
  *reward = (q[0] - x0) / dt + m->alive_bonus - m->ctrl_cost * asq;
  int ok = 1;
  if (m->task == 0) {
    for (int i = 0; i < n; ++i) ok = ok && isfinite(q[i]) && isfinite(v[i]) && fabs(v[i]) < m->state_max && (i < 2 || fabs(q[i]) < m->state_max);
    ok = ok && q[1] > m->z_min && fabs(q[2]) < m->ang_max;
  } else if (m->task == 1) {
    ok = q[1] > m->z_min && q[1] < m->z_max && q[2] > -m->ang_max && q[2] < m->ang_max;
  }
  *done = !ok;
  for (int i = 1; i < n; ++i) obs[i - 1] = q[i];
  for (int i = 0; i < n; ++i) obs[n - 1 + i] = m->qvel_clip > 0 ? fmin(fmax(v[i], -m->qvel_clip), m->qvel_clip) : v[i];
  return 0;
}

/* CPU baseline: n_env envs stepped n_steps times with uniform[-1,1] actions from a per-env LCG, auto-reset on termination or after
 * max_path_length steps, `threads` OpenMP threads over envs.  Returns seconds; *checksum keeps the work alive. */
double orc_planar_bench(const ilsx_planar_model* m, int n_env, int n_steps, int max_path_length, int threads, double* checksum) {
  const int nb = m->n_body, n = nb + 2;
  double total = 0.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) reduction(+ : total) schedule(static)
#endif
  for (int e = 0; e < n_env; ++e) {
    unsigned long long rng = 0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1);
    double q[NMAX], v[NMAX], obs[2 * NMAX], act[ILSX_ENV_MAX_BODY], r;
    int done, len = 0;
#define ORC_U01() (rng = rng * 6364136223846793005ull + 1442695040888963407ull, (double)(rng >> 11) * (1.0 / 9007199254740992.0))
    for (int i = 0; i < n; ++i) { q[i] = m->init_qpos[i] + m->reset_noise * (2.0 * ORC_U01() - 1.0); v[i] = m->reset_noise * (2.0 * ORC_U01() - 1.0); }
    for (int s = 0; s < n_steps; ++s) {
      for (int k = 0; k < nb; ++k) act[k] = 2.0 * ORC_U01() - 1.0;
      orc_planar_step(m, q, v, act, obs, &r, &done);
      total += r;
      if (done || ++len >= max_path_length) {
        len = 0;
        for (int i = 0; i < n; ++i) { q[i] = m->init_qpos[i] + m->reset_noise * (2.0 * ORC_U01() - 1.0); v[i] = m->reset_noise * (2.0 * ORC_U01() - 1.0); }
      }
    }
#undef ORC_U01
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  (void)threads;
  if (checksum) *checksum = total;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
