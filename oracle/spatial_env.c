/* TEST INFRASTRUCTURE, not product: a scalar C restatement of oracle/spatial_env.py (the 3-D articulated-body stepper the HIP kernels
 * k_env3dw_step / k_env3d_step must reproduce: Ant-v2, Humanoid-v2), for ONE purpose — the CPU leg of bench.py's `cpu_baseline` for the
 * Humanoid env-steps/s of BASELINE config 5 ("env-step CPU baseline: the build's scalar C++ stepper, 1 thread and all
 * cores"; the numpy statement is ~1000x slower than compiled code and says nothing about a CPU).  Same DENSE formulation as the numpy
 * file, function for function — per-link 3 x nv Jacobians, M = sum m Jc^T Jc + Jw^T Iw Jw, bias from the velocity-product accelerations,
 * LU solves, contact spheres (normal + two pyramidal tangent rows) and violated joint limits as soft constraint rows, projected
 * Gauss-Seidel, RK4 with the constraint solve in every stage and positions advanced on the quaternion manifold — and therefore NOT the
 * formulation of the device code (composite-rigid-body / Newton-Euler recursions over the link tree, Cholesky, A = Z Z^T).  Reward /
 * termination / observation rules: rlkit/envs/mujoco/humanoid.py:24-73, ant.py:11-43; action map wrappers.py:342-346.
 * Pinned against oracle/spatial_env.py by tests/test_env3d_oracle.py (1e-9 over chained steps with contacts, limits, terminations).
 * Only tests/ and bench.py's cpu_baseline load it; nothing under ilswiss_amd/ does.  Build: make -C oracle.
 * The model arrives as the C-ABI struct of include/ilsx.h (the boundary's own type), filled by the test / bench. */
#include <math.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/ilsx.h"

#define NL ILSX_ENV3_MAX_LINK
#define NV (ILSX_ENV3_MAX_LINK + 6)   /* nv = 6 + (n_link - 1) < NV */
#define NQ (ILSX_ENV3_MAX_LINK + 7)
#define RM 64                        /* constraint rows kept (model->max_rows <= RM is checked) */

typedef struct { double R[NL][3][3], o[NL][3], w[NL][3], Jw[NL][3][NV], Jo[NL][3][NV], al[NL][3], ao[NL][3], vo[NL][3]; } Kin;

static void quat_to_R(const double* q, double R[3][3]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z); R[0][2] = 2 * (x * z + w * y);
  R[1][0] = 2 * (x * y + w * z); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
  R[2][0] = 2 * (x * z - w * y); R[2][1] = 2 * (y * z + w * x); R[2][2] = 1 - 2 * (x * x + y * y);
}
static void quat_mul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
static void mat3_vec(const double A[3][3], const double* x, double* y) {
  for (int i = 0; i < 3; ++i) y[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2];
}
static void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
/* Rodrigues: I + sin(ang) K + (1 - cos(ang)) K K */
static void axis_angle_R(const double* ax, double ang, double R[3][3]) {
  const double K[3][3] = {{0, -ax[2], ax[1]}, {ax[2], 0, -ax[0]}, {-ax[1], ax[0], 0}};
  double KK[3][3];
  mat3_mul(K, K, KK);
  const double s = sin(ang), c1 = 1.0 - cos(ang);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = (i == j ? 1.0 : 0.0) + s * K[i][j] + c1 * KK[i][j];
}
/* J_out = J_a - skew(r) J_b  (3 x nv) */
static void jac_shift(int nv, const double Ja[3][NV], const double* r, const double Jb[3][NV], double Jout[3][NV]) {
  for (int i = 0; i < nv; ++i) {
    const double b0 = Jb[0][i], b1 = Jb[1][i], b2 = Jb[2][i];
    Jout[0][i] = Ja[0][i] - (-r[2] * b1 + r[1] * b2);
    Jout[1][i] = Ja[1][i] - (r[2] * b0 - r[0] * b2);
    Jout[2][i] = Ja[2][i] - (-r[1] * b0 + r[0] * b1);
  }
}
static double impedance(double r_abs, const double* solimp) {
  const double d0 = solimp[0], dmax = solimp[1], width = solimp[2];
  const double x = width > 0 ? fmin(r_abs / width, 1.0) : 1.0;
  const double y = x < 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return d0 + y * (dmax - d0);
}
/* LU with partial pivoting of M (n x n), then nrhs solves (numpy.linalg.solve's algorithm) */
static void lu_solve(int n, const double M[NV][NV], int nrhs, double B[][NV], double X[][NV]) {
  double A[NV][NV];
  int piv[NV];
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
    double y[NV];
    for (int i = 0; i < n; ++i) { double s = B[r][piv[i]]; for (int j = 0; j < i; ++j) s -= A[i][j] * y[j]; y[i] = s; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= A[i][j] * X[r][j]; X[r][i] = s / A[i][i]; }
  }
}

/* SpatialOracle.kin */
static void kin(const ilsx_spatial_model* m, const double* q, const double* v, Kin* K) {
  const int nl = m->n_link, nv = 6 + nl - 1;
  memset(K->Jw, 0, sizeof K->Jw); memset(K->Jo, 0, sizeof K->Jo);
  double qn[4], q0[4];
  const double nrm = sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[6] * q[6]);
  for (int i = 0; i < 4; ++i) qn[i] = q[3 + i] / nrm;
  quat_mul(qn, m->quat0[0], q0);
  quat_to_R(q0, K->R[0]);
  for (int i = 0; i < 3; ++i) {
    K->o[0][i] = q[i]; K->vo[0][i] = v[i]; K->al[0][i] = 0.0; K->ao[0][i] = 0.0;
    K->Jo[0][i][i] = 1.0;
    for (int j = 0; j < 3; ++j) K->Jw[0][i][3 + j] = K->R[0][i][j];      /* omega_world = R omega_body */
  }
  mat3_vec(K->R[0], v + 3, K->w[0]);
  for (int l = 1; l < nl; ++l) {
    const int p = m->parent[l];
    double Rq[3][3], Ra[3][3], Rrel[3][3], rp[3], ax_l[3], aw[3], t[3], t2[3];
    quat_to_R(m->quat0[l], Rq);
    axis_angle_R(m->axis[l], q[7 + l - 1], Ra);
    mat3_mul(Rq, Ra, Rrel);
    mat3_mul(K->R[p], Rrel, K->R[l]);
    mat3_vec(K->R[p], m->anchor[l], rp);
    mat3_vec(Rq, m->axis[l], ax_l);
    mat3_vec(K->R[p], ax_l, aw);                                        /* hinge axis in the world (fixed in the parent link) */
    const double qd = v[6 + l - 1];
    for (int i = 0; i < 3; ++i) K->o[l][i] = K->o[p][i] + rp[i];
    memcpy(K->Jw[l], K->Jw[p], sizeof K->Jw[l]);
    for (int i = 0; i < 3; ++i) { K->Jw[l][i][6 + l - 1] += aw[i]; K->w[l][i] = K->w[p][i] + aw[i] * qd; t[i] = aw[i] * qd; }
    cross(K->w[p], t, t2);
    for (int i = 0; i < 3; ++i) K->al[l][i] = K->al[p][i] + t2[i];
    jac_shift(nv, K->Jo[p], rp, K->Jw[p], K->Jo[l]);
    cross(K->w[p], rp, t);
    for (int i = 0; i < 3; ++i) K->vo[l][i] = K->vo[p][i] + t[i];
    cross(K->w[p], t, t2);                                               /* w x (w x rp) */
    cross(K->al[p], rp, t);
    for (int i = 0; i < 3; ++i) K->ao[l][i] = K->ao[p][i] + t[i] + t2[i];
  }
}

static void inertia_world(const ilsx_spatial_model* m, int l, const double R[3][3], double Iw[3][3]) {
  const double* c = m->inertia[l];   /* xx yy zz xy xz yz */
  const double I[3][3] = {{c[0], c[3], c[4]}, {c[3], c[1], c[5]}, {c[4], c[5], c[2]}};
  double RI[3][3], Rt[3][3];
  mat3_mul(R, I, RI);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i][j] = R[j][i];
  mat3_mul(RI, Rt, Iw);
}

/* SpatialOracle.mass_bias */
static void mass_bias(const ilsx_spatial_model* m, const double* q, const double* v, Kin* K, double M[NV][NV], double* c) {
  const int nl = m->n_link, nv = 6 + nl - 1;
  kin(m, q, v, K);
  memset(M, 0, sizeof(double) * NV * NV);
  for (int i = 0; i < nv; ++i) c[i] = 0.0;
  const double g[3] = {0.0, 0.0, m->gravity};
  for (int l = 0; l < nl; ++l) {
    const double ml = m->mass[l];
    if (ml == 0.0) continue;
    double rc[3], Jc[3][NV], ac[3], t[3], t2[3], Iw[3][3], Ial[3], Iwv[3], wIw[3], fa[3], fw[3];
    mat3_vec(K->R[l], m->com[l], rc);
    jac_shift(nv, K->Jo[l], rc, K->Jw[l], Jc);
    cross(K->w[l], rc, t); cross(K->w[l], t, t2); cross(K->al[l], rc, t);
    for (int i = 0; i < 3; ++i) ac[i] = K->ao[l][i] + t[i] + t2[i];
    inertia_world(m, l, K->R[l], Iw);
    mat3_vec(Iw, K->al[l], Ial); mat3_vec(Iw, K->w[l], Iwv); cross(K->w[l], Iwv, wIw);
    for (int i = 0; i < 3; ++i) { fa[i] = ac[i] + g[i]; fw[i] = Ial[i] + wIw[i]; }
    for (int i = 0; i < nv; ++i) {
      double IJ[3];
      for (int a = 0; a < 3; ++a) IJ[a] = Iw[a][0] * K->Jw[l][0][i] + Iw[a][1] * K->Jw[l][1][i] + Iw[a][2] * K->Jw[l][2][i];
      for (int j = 0; j < nv; ++j)
        M[j][i] += ml * (Jc[0][j] * Jc[0][i] + Jc[1][j] * Jc[1][i] + Jc[2][j] * Jc[2][i]) +
                   (K->Jw[l][0][j] * IJ[0] + K->Jw[l][1][j] * IJ[1] + K->Jw[l][2][j] * IJ[2]);
      c[i] += ml * (Jc[0][i] * fa[0] + Jc[1][i] * fa[1] + Jc[2][i] * fa[2]) +
              (K->Jw[l][0][i] * fw[0] + K->Jw[l][1][i] * fw[1] + K->Jw[l][2][i] * fw[2]);
    }
  }
  for (int l = 1; l < nl; ++l) M[6 + l - 1][6 + l - 1] += m->armature[l];
}

typedef struct { double J[NV], r, rdist, mu; int kind; const double *solref, *solimp; } Row3;   /* kind 0 normal, 1 / 2 tangent, 3 limit */

/* SpatialOracle.dynamics */
static void dynamics(const ilsx_spatial_model* m, const double* q, const double* v, const double* ctrl, double* qacc) {
  const int nl = m->n_link, nv = 6 + nl - 1;
  static _Thread_local Kin K;
  double M[NV][NV], c[NV], rhs1[1][NV], q0[1][NV];
  mass_bias(m, q, v, &K, M, c);
  for (int i = 0; i < nv; ++i) rhs1[0][i] = 0.0;
  for (int l = 1; l < nl; ++l) rhs1[0][6 + l - 1] = -m->damping[l] * v[6 + l - 1] - m->stiffness[l] * q[7 + l - 1];
  for (int k = 0; k < m->n_act; ++k) { const int l = m->act_link[k]; rhs1[0][6 + l - 1] += m->gear[l] * ctrl[k]; }
  for (int i = 0; i < nv; ++i) rhs1[0][i] -= c[i];
  lu_solve(nv, M, 1, rhs1, q0);
  static _Thread_local Row3 rows[RM];
  int nr = 0;
  const int max_rows = m->max_rows < RM ? m->max_rows : RM;
  for (int ci = 0; ci < m->n_contact; ++ci) {
    const int l = m->contact_link[ci];
    double rp[3];
    mat3_vec(K.R[l], m->contact_pos[ci], rp);
    const double rad = m->contact_radius[ci], dist = K.o[l][2] + rp[2] - rad;
    if (dist < m->contact_margin && nr + 3 <= max_rows) {
      const double rc[3] = {rp[0], rp[1], rp[2] - (rad + 0.5 * dist)};   /* contact point relative to the link origin */
      double Jp[3][NV];
      jac_shift(nv, K.Jo[l], rc, K.Jw[l], Jp);
      const int ax[3] = {2, 0, 1};
      for (int t = 0; t < 3; ++t) {
        Row3* r = &rows[nr + t];
        memcpy(r->J, Jp[ax[t]], sizeof(double) * NV);
        r->r = t == 0 ? dist : 0.0; r->rdist = dist; r->kind = t; r->mu = m->contact_friction[ci];
        r->solref = m->contact_solref; r->solimp = m->contact_solimp;
      }
      nr += 3;
    }
  }
  for (int l = 1; l < nl; ++l) {
    if (!m->limited[l] || nr + 1 > max_rows) continue;
    const double lo = m->range[l][0], hi = m->range[l][1], ql = q[7 + l - 1];
    double sgn = 0.0, r = 0.0;
    if (ql - lo < 0.0) { sgn = 1.0; r = ql - lo; }
    else if (hi - ql < 0.0) { sgn = -1.0; r = hi - ql; }
    if (sgn == 0.0) continue;
    Row3* rl = &rows[nr++];
    memset(rl->J, 0, sizeof rl->J);
    rl->J[6 + l - 1] = sgn; rl->r = r; rl->rdist = r; rl->kind = 3; rl->mu = 0.0; rl->solref = m->limit_solref; rl->solimp = m->limit_solimp;
  }
  if (nr == 0) { memcpy(qacc, q0[0], nv * sizeof(double)); return; }
  static _Thread_local double JT[RM][NV], MiJ[RM][NV], A[RM][RM];
  double Rg[RM], rc2[RM], f[RM];
  for (int r = 0; r < nr; ++r) memcpy(JT[r], rows[r].J, sizeof JT[r]);
  lu_solve(nv, M, nr, JT, MiJ);
  for (int r = 0; r < nr; ++r)
    for (int c2 = 0; c2 < nr; ++c2) { double s = 0.0; for (int i = 0; i < nv; ++i) s += rows[r].J[i] * MiJ[c2][i]; A[r][c2] = s; }
  for (int r = 0; r < nr; ++r) {
    const double tcs = rows[r].solref[0], drs = rows[r].solref[1], dmax = rows[r].solimp[1];
    const double rr = (rows[r].kind == 1 || rows[r].kind == 2) ? rows[r].rdist : rows[r].r;
    const double d = impedance(fabs(rr), rows[r].solimp);
    const double bdamp = 2.0 / (dmax * tcs), kst = 1.0 / (dmax * dmax * tcs * tcs * drs * drs);
    double jv = 0.0, jq = 0.0;
    for (int i = 0; i < nv; ++i) { jv += rows[r].J[i] * v[i]; jq += rows[r].J[i] * q0[0][i]; }
    Rg[r] = (1.0 - d) / d * A[r][r];
    rc2[r] = (-bdamp * jv - kst * d * rows[r].r) - jq;
    f[r] = 0.0;
  }
  for (int it = 0; it < m->pgs_iters; ++it)
    for (int r = 0; r < nr; ++r) {
      double res = rc2[r];
      for (int c2 = 0; c2 < nr; ++c2) res -= A[r][c2] * f[c2];
      res += A[r][r] * f[r];
      double fi = res / (A[r][r] + Rg[r]);
      if (rows[r].kind == 1) { const double lim = rows[r].mu * f[r - 1]; fi = fmin(fmax(fi, -lim), lim); }
      else if (rows[r].kind == 2) { const double lim = rows[r].mu * f[r - 2]; fi = fmin(fmax(fi, -lim), lim); }
      else fi = fmax(fi, 0.0);
      f[r] = fi;
    }
  for (int i = 0; i < nv; ++i) { double s = q0[0][i]; for (int r = 0; r < nr; ++r) s += MiJ[r][i] * f[r]; qacc[i] = s; }
}

/* mj_integratePos */
static void integrate_pos(int nq, int nv, const double* q, const double* v, double h, double* out) {
  memcpy(out, q, nq * sizeof(double));
  for (int i = 0; i < 3; ++i) out[i] += h * v[i];
  const double wn = sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5]), ang = wn * h;
  if (ang > 0) {
    const double s = sin(0.5 * ang), dq[4] = {cos(0.5 * ang), s * v[3] / wn, s * v[4] / wn, s * v[5] / wn};
    quat_mul(q + 3, dq, out + 3);
  }
  const double n = sqrt(out[3] * out[3] + out[4] * out[4] + out[5] * out[5] + out[6] * out[6]);
  for (int i = 3; i < 7; ++i) out[i] /= n;
  for (int i = 6; i < nv; ++i) out[7 + i - 6] += h * v[i];
}

/* SpatialOracle.substep: RK4, positions on the manifold */
static void substep(const ilsx_spatial_model* m, double* q, double* v, const double* ctrl) {
  const int nl = m->n_link, nv = 6 + nl - 1, nq = nv + 1;
  const double h = m->timestep;
  double a1[NV], a2[NV], a3[NV], a4[NV], q2[NQ], v2[NV], q3[NQ], v3[NV], q4[NQ], v4[NV], vbar[NV], qn[NQ];
  dynamics(m, q, v, ctrl, a1);
  integrate_pos(nq, nv, q, v, 0.5 * h, q2);
  for (int i = 0; i < nv; ++i) v2[i] = v[i] + 0.5 * h * a1[i];
  dynamics(m, q2, v2, ctrl, a2);
  integrate_pos(nq, nv, q, v2, 0.5 * h, q3);
  for (int i = 0; i < nv; ++i) v3[i] = v[i] + 0.5 * h * a2[i];
  dynamics(m, q3, v3, ctrl, a3);
  integrate_pos(nq, nv, q, v3, h, q4);
  for (int i = 0; i < nv; ++i) v4[i] = v[i] + h * a3[i];
  dynamics(m, q4, v4, ctrl, a4);
  for (int i = 0; i < nv; ++i) vbar[i] = (v[i] + 2 * v2[i] + 2 * v3[i] + v4[i]) / 6.0;
  integrate_pos(nq, nv, q, vbar, h, qn);
  memcpy(q, qn, nq * sizeof(double));
  for (int i = 0; i < nv; ++i) v[i] += h / 6.0 * (a1[i] + 2 * a2[i] + 2 * a3[i] + a4[i]);
}

/* mass_center (humanoid.py:6-9): x of sum(m xipos) / sum(m); com3 (nullable) receives the whole vector */
static double com_x(const ilsx_spatial_model* m, const double* q, Kin* K, const double* v_for_kin, double* com3) {
  const int nl = m->n_link;
  double zero[NV] = {0};
  kin(m, q, v_for_kin ? v_for_kin : zero, K);
  double num[3] = {0, 0, 0}, tot = 0.0;
  for (int l = 0; l < nl; ++l) {
    double rc[3];
    mat3_vec(K->R[l], m->com[l], rc);
    for (int i = 0; i < 3; ++i) num[i] += m->mass[l] * (K->o[l][i] + rc[i]);
    tot += m->mass[l];
  }
  if (com3) for (int i = 0; i < 3; ++i) com3[i] = num[i] / tot;
  return num[0] / tot;
}

int orc_spatial_obs_dim(const ilsx_spatial_model* m) {
  const int nv = 6 + m->n_link - 1, nbody = m->n_body + 1;
  return m->task == ILSX_TASK_ANT ? (nv + 1 - 2) + nv + nbody * 6 : (nv + 1 - 2) + nv + nbody * 10 + nbody * 6 + nv + nbody * 6;
}

/* SpatialOracle.obs (+ obs_extras for Humanoid) */
static void observe(const ilsx_spatial_model* m, const double* q, const double* v, const double* ctrl, double* obs) {
  const int nl = m->n_link, nv = 6 + nl - 1, nq = nv + 1, nbody = m->n_body + 1;
  int k = 0;
  for (int i = 2; i < nq; ++i) obs[k++] = q[i];
  for (int i = 0; i < nv; ++i) obs[k++] = v[i];
  if (m->task == ILSX_TASK_ANT) { for (int i = 0; i < nbody * 6; ++i) obs[k++] = 0.0; return; }   /* cfrc_ext: zeros under MuJoCo >= 2.0 */
  static _Thread_local Kin K;
  double com[3];
  com_x(m, q, &K, v, com);
  double* cin = obs + k; k += nbody * 10;
  double* cv = obs + k; k += nbody * 6;
  memset(cin, 0, sizeof(double) * nbody * 10); memset(cv, 0, sizeof(double) * nbody * 6);
  int seen[NL] = {0};
  for (int b = 0; b < m->n_body; ++b) {
    const int l = m->body_link[b];
    double d[3], t[3], rc[3];
    for (int i = 0; i < 3; ++i) d[i] = com[i] - K.o[l][i];
    cross(K.w[l], d, t);
    for (int i = 0; i < 3; ++i) { cv[(b + 1) * 6 + i] = K.w[l][i]; cv[(b + 1) * 6 + 3 + i] = K.vo[l][i] + t[i]; }
    if (seen[l]) continue;   /* a welded body (Humanoid's feet): MuJoCo lists it separately; its mass sits in the parent here */
    seen[l] = 1;
    const double ml = m->mass[l];
    mat3_vec(K.R[l], m->com[l], rc);
    for (int i = 0; i < 3; ++i) d[i] = K.o[l][i] + rc[i] - com[i];
    double Iw[3][3];
    inertia_world(m, l, K.R[l], Iw);
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double I[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I[i][j] = Iw[i][j] + ml * ((i == j ? dd : 0.0) - d[i] * d[j]);
    double* c = cin + (b + 1) * 10;
    c[0] = I[0][0]; c[1] = I[1][1]; c[2] = I[2][2]; c[3] = I[0][1]; c[4] = I[0][2]; c[5] = I[1][2];
    c[6] = ml * d[0]; c[7] = ml * d[1]; c[8] = ml * d[2]; c[9] = ml;
  }
  for (int i = 0; i < nv; ++i) obs[k + i] = 0.0;
  for (int a = 0; a < m->n_act; ++a) { const int l = m->act_link[a]; obs[k + 6 + l - 1] = m->gear[l] * ctrl[a]; }
  k += nv;
  for (int i = 0; i < nbody * 6; ++i) obs[k++] = 0.0;
}

/* SpatialOracle.step: q [nq], v [nv] in / out, action [n_act], obs [orc_spatial_obs_dim] */
int orc_spatial_step(const ilsx_spatial_model* m, double* q, double* v, const double* action, double* obs, double* reward, int* done) {
  const int nl = m->n_link, nv = 6 + nl - 1, nq = nv + 1;
  if (nl < 1 || nl > NL || m->n_act > NL || m->max_rows > RM) return -1;
  static _Thread_local Kin K;
  double ctrl[NL], asq = 0.0, csq = 0.0;
  for (int k = 0; k < m->n_act; ++k) {
    ctrl[k] = fmin(fmax(action[k] * m->ctrl_range, -m->ctrl_range), m->ctrl_range);   /* NormalizedBoxEnv, wrappers.py:342-346 */
    const double a = fmin(fmax(action[k], -1.0), 1.0);
    asq += a * a; csq += ctrl[k] * ctrl[k];
  }
  const double x0 = m->task == ILSX_TASK_HUMANOID ? com_x(m, q, &K, 0, 0) : q[0];
  for (int s = 0; s < m->frame_skip; ++s) substep(m, q, v, ctrl);
  const double x1 = m->task == ILSX_TASK_HUMANOID ? com_x(m, q, &K, 0, 0) : q[0];
  if (m->task == ILSX_TASK_HUMANOID) {   /* humanoid.py:37-49 */
    *reward = m->vel_weight * (x1 - x0) / m->timestep - m->ctrl_cost * csq + m->alive_bonus;
    *done = (q[2] < m->z_min || q[2] > m->z_max) ? 1 : 0;
  } else {                               /* ant.py:11-24 */
    const double dt = m->timestep * m->frame_skip;
    *reward = (x1 - x0) / dt - m->ctrl_cost * asq + m->alive_bonus;
    int ok = 1;
    for (int i = 0; i < nq; ++i) ok = ok && isfinite(q[i]);
    for (int i = 0; i < nv; ++i) ok = ok && isfinite(v[i]);
    ok = ok && q[2] >= m->z_min && q[2] <= m->z_max;
    *done = !ok;
  }
  observe(m, q, v, ctrl, obs);
  return 0;
}

/* CPU baseline: n_env envs stepped n_steps times with uniform[-1,1] actions from a per-env LCG, auto-reset on termination or after
 * max_path_length steps, `threads` OpenMP threads over envs.  Returns seconds; *checksum keeps the work alive. */
double orc_spatial_bench(const ilsx_spatial_model* m, int n_env, int n_steps, int max_path_length, int threads, double* checksum) {
  const int nl = m->n_link, nv = 6 + nl - 1, nq = nv + 1;
  double total = 0.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) reduction(+ : total) schedule(static)
#endif
  for (int e = 0; e < n_env; ++e) {
    unsigned long long rng = 0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1);
    double q[NQ], v[NV], obs[512], act[NL], r;
    int done, len = 0;
#define ORC_U01() (rng = rng * 6364136223846793005ull + 1442695040888963407ull, (double)(rng >> 11) * (1.0 / 9007199254740992.0))
#define ORC_RESET() do { for (int i = 0; i < nq; ++i) q[i] = m->init_qpos[i] + m->reset_noise * (2.0 * ORC_U01() - 1.0); \
      { const double n_ = sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[6] * q[6]); for (int i = 3; i < 7; ++i) q[i] /= n_; } \
      for (int i = 0; i < nv; ++i) v[i] = m->reset_noise * (2.0 * ORC_U01() - 1.0); } while (0)
    ORC_RESET();
    for (int s = 0; s < n_steps; ++s) {
      for (int k = 0; k < m->n_act; ++k) act[k] = 2.0 * ORC_U01() - 1.0;
      orc_spatial_step(m, q, v, act, obs, &r, &done);
      total += r;
      if (done || ++len >= max_path_length) { len = 0; ORC_RESET(); }
    }
#undef ORC_RESET
#undef ORC_U01
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  (void)threads;
  if (checksum) *checksum = total;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
