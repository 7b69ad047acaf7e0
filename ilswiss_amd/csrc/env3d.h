// env3d.h — device code of the 3-D articulated-body stepper (Ant-v2 / Humanoid-v2), included by ilsx_env.hip.
//
// Replaces the MuJoCo call under gym's AntEnv / HumanoidEnv.step (rlkit/envs/envs_dict.py:6,9); reward / termination /
// observation / reset rules as restated in rlkit/envs/mujoco/humanoid.py:24-73 and ant.py:11-43.  The model and its solver are
// stated in oracle/spatial_env.py (dense Jacobians + numpy solves); this file computes the same quantities with recursions over
// the link tree — composite-rigid-body mass matrix, recursive Newton-Euler bias, in-place Cholesky, z_r = L^-1 j_r rows and
// A = Z Z^T for the projected Gauss-Seidel sweep — so that agreement with the oracle (1e-8 over chained steps) means something.
// Physics parity with MuJoCo is UNPINNED (no MuJoCo here), see DESIGN.md.
//
// One lane per env, fp64.  A 23-DoF model needs ~2.5k doubles of working set per env (link frames, composite inertias, the mass
// matrix / Cholesky factor, up to 30 constraint rows, the constraint-space matrix): far beyond registers and, at 64 envs per
// wave, beyond LDS, so it lives in an explicit global scratch laid out [slot][env] — every access of a wave is one coalesced
// 512-byte run — instead of compiler-managed private memory (whose size the runtime multiplies by every wave slot of the chip).
// Not a roofline kernel: a long serial fp64 chain per env; the figure to report is env-steps/s.
#pragma once
#include <math.h>
#include <string.h>
#include "../../include/ilsx.h"

#define E3_MAXL 20     // links (root + hinges)
#define E3_MAXC 32     // contact spheres
#define E3_MAXB 16     // MuJoCo bodies reported in the observation (world excluded)
#define E3_MAXR 30     // constraint rows kept per env
#define E3_MAXOBS 384

struct Spatial3Dev {
  int nl, nv, nq, task, frame_skip, pgs_iters, max_rows, n_act, obs_dim, n_contact, n_body, pad0;
  int parent[E3_MAXL], limited[E3_MAXL], act_link[E3_MAXL], contact_link[E3_MAXC], body_link[E3_MAXB], body_first[E3_MAXB];
  double anchor[E3_MAXL][3], axis[E3_MAXL][3], axis_p[E3_MAXL][3], Rq0[E3_MAXL][9], com[E3_MAXL][3], mass[E3_MAXL], inertia[E3_MAXL][6];
  double armature[E3_MAXL], damping[E3_MAXL], stiffness[E3_MAXL], range[E3_MAXL][2], gear[E3_MAXL];
  double cpos[E3_MAXC][3], crad[E3_MAXC], cfric[E3_MAXC];
  double timestep, gravity, reset_noise, margin, reset_noise_vel_std, ctrl_range, total_mass;
  double c_solref[2], c_solimp[3], l_solref[2], l_solimp[3];
  double ctrl_cost, alive, vel_weight, z_min, z_max;
  double init_qpos[E3_MAXL + 6];
  double obs_shift[E3_MAXOBS], obs_inv_scale[E3_MAXOBS];
  // derived tables of the wave-per-env kernels (env3d_wave.h): links ordered by depth in the tree, ancestor-or-self bit masks,
  // and the actuator on each hinge (-1: none)
  int n_level, lvl_off[E3_MAXL + 1], lvl_link[E3_MAXL], link_act[E3_MAXL], depth[E3_MAXL];
  unsigned anc_mask[E3_MAXL];
};

// ---- scratch map (doubles per env); S(i) = element i of this env
struct E3Off {
  static constexpr int KIN = 0;                                  // per link 27: R[9] o[3] w[3] vo[3] aw[3] al[3] ao[3]
  static constexpr int CRB = KIN + 27 * E3_MAXL;                 // per link 10: m, h[3], Io[6] (xx yy zz xy xz yz) about the world origin
  static constexpr int WR = CRB + 10 * E3_MAXL;                  // per link 6: subtree force f[3], moment about the world origin n[3]
  static constexpr int M = WR + 6 * E3_MAXL;                     // lower triangle, nv(nv+1)/2  (becomes L)
  static constexpr int NVMAX = E3_MAXL + 5;
  static constexpr int Z = M + NVMAX * (NVMAX + 1) / 2;          // rows [r][nv] (become z_r = L^-1 j_r)
  static constexpr int A = Z + E3_MAXR * NVMAX;                  // lower triangle of Z Z^T
  static constexpr int RM = A + E3_MAXR * (E3_MAXR + 1) / 2;     // per row 5: rhs, Rg, f, kind, mu
  static constexpr int VEC = RM + 5 * E3_MAXR;                   // qacc0[nv], c[nv], tmp[nv]
  static constexpr int ST = VEC + 3 * NVMAX;                     // RK4: q0[nq] v0[nv] qs[nq] vs[nv] vsum[nv] asum[nv] acc[nv] ctrl[MAXL]
  static constexpr int TOTAL = ST + 2 * (NVMAX + 1) + 5 * NVMAX + E3_MAXL;
};
#define E3S(i) scr[(size_t)(i) * n_env + env]

struct E3Ctx { double* scr; int n_env, env; const Spatial3Dev* m; };

// multiply-adds may fuse in the fp64 helpers on the device (results are compared with the oracle at 1e-8, not bit for bit)
#ifdef __HIP_DEVICE_COMPILE__
#define E3_FMA _Pragma("clang fp contract(fast)")
#else
#define E3_FMA
#endif
__device__ __forceinline__ void e3_quat_to_R(double w, double x, double y, double z, double* R) {
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void e3_mat3mul(const double* A, const double* B, double* C) {
  E3_FMA
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void e3_matvec(const double* A, const double* x, double* y) {
  E3_FMA
  y[0] = A[0] * x[0] + A[1] * x[1] + A[2] * x[2]; y[1] = A[3] * x[0] + A[4] * x[1] + A[5] * x[2]; y[2] = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
}
__device__ __forceinline__ void e3_cross(const double* a, const double* b, double* c) {
  E3_FMA
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double e3_dot(const double* a, const double* b) {
  E3_FMA
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
__device__ __forceinline__ double e3_impedance(double r_abs, const double* solimp) {
  const double d0 = solimp[0], dmax = solimp[1], width = solimp[2];
  const double x = width > 0.0 ? fmin(r_abs / width, 1.0) : 1.0;
  const double y = x < 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return d0 + y * (dmax - d0);
}
__device__ __forceinline__ void e3_ld3(const E3Ctx& C, int at, double* v) {
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  v[0] = E3S(at); v[1] = E3S(at + 1); v[2] = E3S(at + 2);
}
__device__ __forceinline__ void e3_st3(const E3Ctx& C, int at, const double* v) {
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  E3S(at) = v[0]; E3S(at + 1) = v[1]; E3S(at + 2) = v[2];
}

// Link frames and velocity-product accelerations (oracle kin()); q at ST+qoff (nq), v at ST+voff (nv) of the scratch.
__device__ void e3_kinematics(const E3Ctx& C, int qoff, int voff) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  {
    double qw = E3S(qoff + 3), qx = E3S(qoff + 4), qy = E3S(qoff + 5), qz = E3S(qoff + 6);
    const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw *= nrm; qx *= nrm; qy *= nrm; qz *= nrm;
    double R[9], o[3] = {E3S(qoff), E3S(qoff + 1), E3S(qoff + 2)}, wb[3] = {E3S(voff + 3), E3S(voff + 4), E3S(voff + 5)}, w[3];
    e3_quat_to_R(qw, qx, qy, qz, R);
    e3_matvec(R, wb, w);
    const int k = E3Off::KIN;
#pragma unroll
    for (int i = 0; i < 9; ++i) E3S(k + i) = R[i];
    e3_st3(C, k + 9, o); e3_st3(C, k + 12, w);
    double vo[3] = {E3S(voff), E3S(voff + 1), E3S(voff + 2)}, z3[3] = {0.0, 0.0, 0.0};
    e3_st3(C, k + 15, vo); e3_st3(C, k + 18, z3); e3_st3(C, k + 21, z3); e3_st3(C, k + 24, z3);
  }
  for (int l = 1; l < m.nl; ++l) {
    const int p = m.parent[l], kp = E3Off::KIN + 27 * p, k = E3Off::KIN + 27 * l;
    double Rp[9], op[3], wp[3], vop[3], alp[3], aop[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rp[i] = E3S(kp + i);
    e3_ld3(C, kp + 9, op); e3_ld3(C, kp + 12, wp); e3_ld3(C, kp + 15, vop); e3_ld3(C, kp + 21, alp); e3_ld3(C, kp + 24, aop);
    // Rodrigues about the hinge axis (link frame), then the fixed rotation quat0
    const double ang = E3S(qoff + 7 + l - 1), qd = E3S(voff + 6 + l - 1);
    const double s = sin(ang), c1 = 1.0 - cos(ang);
    const double ax = m.axis[l][0], ay = m.axis[l][1], az = m.axis[l][2];
    double Rh[9] = {1.0 - c1 * (ay * ay + az * az), -s * az + c1 * ax * ay, s * ay + c1 * ax * az,
                    s * az + c1 * ax * ay, 1.0 - c1 * (ax * ax + az * az), -s * ax + c1 * ay * az,
                    -s * ay + c1 * ax * az, s * ax + c1 * ay * az, 1.0 - c1 * (ax * ax + ay * ay)};
    double Rrel[9], R[9];
    e3_mat3mul(m.Rq0[l], Rh, Rrel);
    e3_mat3mul(Rp, Rrel, R);
    double rp[3], aw[3], o[3], w[3], al[3], vo[3], ao[3], t1[3], t2[3];
    e3_matvec(Rp, m.anchor[l], rp);
    e3_matvec(Rp, m.axis_p[l], aw);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[i] = op[i] + rp[i]; w[i] = wp[i] + aw[i] * qd; t1[i] = aw[i] * qd; }
    e3_cross(wp, t1, t2);
#pragma unroll
    for (int i = 0; i < 3; ++i) al[i] = alp[i] + t2[i];
    e3_cross(wp, rp, t1);
#pragma unroll
    for (int i = 0; i < 3; ++i) vo[i] = vop[i] + t1[i];
    e3_cross(wp, t1, t2);         // w x (w x rp)
    e3_cross(alp, rp, t1);
#pragma unroll
    for (int i = 0; i < 3; ++i) ao[i] = aop[i] + t1[i] + t2[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) E3S(k + i) = R[i];
    e3_st3(C, k + 9, o); e3_st3(C, k + 12, w); e3_st3(C, k + 15, vo); e3_st3(C, k + 18, aw); e3_st3(C, k + 21, al); e3_st3(C, k + 24, ao);
  }
}

__device__ __forceinline__ int e3_tri(int r, int c) { return r * (r + 1) / 2 + c; }   // r >= c

// qacc = f(q, v, ctrl) with soft constraints (oracle dynamics()); result at VEC + 2*NVMAX.. (tmp), copied to `out_off` in ST.
__device__ void e3_dynamics(const E3Ctx& C, int qoff, int voff, int ctrl_off, int out_off) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  const int nl = m.nl, nv = m.nv;
  constexpr int NVM = E3Off::NVMAX;
  e3_kinematics(C, qoff, voff);
  // ---- per-link rigid-body quantities: composite inertia seeds (about the world origin) and the RNE wrench
  for (int l = 0; l < nl; ++l) {
    const int k = E3Off::KIN + 27 * l, cb = E3Off::CRB + 10 * l, wr = E3Off::WR + 6 * l;
    const double ml = m.mass[l];
    double R[9], o[3], w[3], al[3], ao[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = E3S(k + i);
    e3_ld3(C, k + 9, o); e3_ld3(C, k + 12, w); e3_ld3(C, k + 21, al); e3_ld3(C, k + 24, ao);
    double rc[3], cw[3];
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) cw[i] = o[i] + rc[i];
    // Iw = R I R^T (I symmetric: xx yy zz xy xz yz)
    const double* I6 = m.inertia[l];
    const double Im[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    double T[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]}, Iw[9];
    e3_mat3mul(R, Im, T);
    e3_mat3mul(T, Rt, Iw);
    const double c2 = e3_dot(cw, cw);
    E3S(cb) = ml;
    E3S(cb + 1) = ml * cw[0]; E3S(cb + 2) = ml * cw[1]; E3S(cb + 3) = ml * cw[2];
    E3S(cb + 4) = Iw[0] + ml * (c2 - cw[0] * cw[0]); E3S(cb + 5) = Iw[4] + ml * (c2 - cw[1] * cw[1]); E3S(cb + 6) = Iw[8] + ml * (c2 - cw[2] * cw[2]);
    E3S(cb + 7) = Iw[1] - ml * cw[0] * cw[1]; E3S(cb + 8) = Iw[2] - ml * cw[0] * cw[2]; E3S(cb + 9) = Iw[5] - ml * cw[1] * cw[2];
    // RNE: force m (a_c + g z), moment I alpha + w x I w about the COM, moved to the world origin
    double t1[3], t2[3], ac[3], F[3], N[3], Iwv[3], n0[3];
    e3_cross(w, rc, t1); e3_cross(w, t1, t2); e3_cross(al, rc, t1);
#pragma unroll
    for (int i = 0; i < 3; ++i) ac[i] = ao[i] + t1[i] + t2[i];
    ac[2] += m.gravity;
#pragma unroll
    for (int i = 0; i < 3; ++i) F[i] = ml * ac[i];
    e3_matvec(Iw, w, Iwv); e3_cross(w, Iwv, t1); e3_matvec(Iw, al, t2); e3_cross(cw, F, n0);
#pragma unroll
    for (int i = 0; i < 3; ++i) N[i] = t2[i] + t1[i] + n0[i];
    e3_st3(C, wr, F); e3_st3(C, wr + 3, N);
  }
  // ---- backward pass: subtree sums (composite inertias, wrenches) and the bias vector c
  const int cv = E3Off::VEC + NVM;   // c
  for (int l = nl - 1; l >= 1; --l) {
    const int p = m.parent[l], k = E3Off::KIN + 27 * l, wr = E3Off::WR + 6 * l, wp = E3Off::WR + 6 * p, cb = E3Off::CRB + 10 * l, cp = E3Off::CRB + 10 * p;
    double f[3], n[3], o[3], aw[3], t[3];
    e3_ld3(C, wr, f); e3_ld3(C, wr + 3, n); e3_ld3(C, k + 9, o); e3_ld3(C, k + 18, aw);
    e3_cross(o, f, t);
    E3S(cv + 6 + l - 1) = aw[0] * (n[0] - t[0]) + aw[1] * (n[1] - t[1]) + aw[2] * (n[2] - t[2]);
#pragma unroll
    for (int i = 0; i < 6; ++i) E3S(wp + i) += E3S(wr + i);
#pragma unroll
    for (int i = 0; i < 10; ++i) E3S(cp + i) += E3S(cb + i);
  }
  {
    double f[3], n[3], o[3], t[3], R[9];
    e3_ld3(C, E3Off::WR, f); e3_ld3(C, E3Off::WR + 3, n); e3_ld3(C, E3Off::KIN + 9, o);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = E3S(E3Off::KIN + i);
    e3_cross(o, f, t);
#pragma unroll
    for (int i = 0; i < 3; ++i) { E3S(cv + i) = f[i]; n[i] -= t[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) E3S(cv + 3 + i) = R[i] * n[0] + R[3 + i] * n[1] + R[6 + i] * n[2];   // R^T n
  }
  // ---- mass matrix (CRBA): M[i][j] = S_j . (Ic_i S_i) for j on the path from i to the root
  for (int i = 0; i < nv * (nv + 1) / 2; ++i) E3S(E3Off::M + i) = 0.0;
  double R0[9], o0[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R0[i] = E3S(E3Off::KIN + i);
  e3_ld3(C, E3Off::KIN + 9, o0);
  // spatial momentum (L about the world origin, p) of a composite moving with (omega = a, v0)
  auto momentum = [&](int cb, const double* a, const double* v0, double* L, double* p) {
    const double mc = E3S(cb), h[3] = {E3S(cb + 1), E3S(cb + 2), E3S(cb + 3)};
    const double Ixx = E3S(cb + 4), Iyy = E3S(cb + 5), Izz = E3S(cb + 6), Ixy = E3S(cb + 7), Ixz = E3S(cb + 8), Iyz = E3S(cb + 9);
    double t[3];
    e3_cross(a, h, t);
    p[0] = mc * v0[0] + t[0]; p[1] = mc * v0[1] + t[1]; p[2] = mc * v0[2] + t[2];
    e3_cross(h, v0, t);
    L[0] = Ixx * a[0] + Ixy * a[1] + Ixz * a[2] + t[0];
    L[1] = Ixy * a[0] + Iyy * a[1] + Iyz * a[2] + t[1];
    L[2] = Ixz * a[0] + Iyz * a[1] + Izz * a[2] + t[2];
  };
  auto root_cols = [&](int row, const double* L, const double* p) {   // the six root columns of row `row`
    for (int kx = 0; kx < 3; ++kx) {
      if (kx <= row) E3S(E3Off::M + e3_tri(row, kx)) = p[kx];
      const double a[3] = {R0[kx], R0[3 + kx], R0[6 + kx]};
      double v0[3];
      e3_cross(o0, a, v0);
      if (3 + kx <= row) E3S(E3Off::M + e3_tri(row, 3 + kx)) = e3_dot(a, L) + e3_dot(v0, p);
    }
  };
  for (int l = nl - 1; l >= 1; --l) {
    const int row = 6 + l - 1, k = E3Off::KIN + 27 * l;
    double a[3], o[3], v0[3], L[3], p[3];
    e3_ld3(C, k + 18, a); e3_ld3(C, k + 9, o);
    e3_cross(o, a, v0);
    momentum(E3Off::CRB + 10 * l, a, v0, L, p);
    E3S(E3Off::M + e3_tri(row, row)) = e3_dot(a, L) + e3_dot(v0, p) + m.armature[l];
    for (int j = m.parent[l]; j >= 1; j = m.parent[j]) {
      const int kj = E3Off::KIN + 27 * j;
      double aj[3], oj[3], vj[3];
      e3_ld3(C, kj + 18, aj); e3_ld3(C, kj + 9, oj);
      e3_cross(oj, aj, vj);
      E3S(E3Off::M + e3_tri(row, 6 + j - 1)) = e3_dot(aj, L) + e3_dot(vj, p);
    }
    root_cols(row, L, p);
  }
  for (int kx = 0; kx < 6; ++kx) {   // root rows
    double a[3] = {0.0, 0.0, 0.0}, v0[3] = {0.0, 0.0, 0.0}, L[3], p[3];
    if (kx < 3) v0[kx] = 1.0;
    else { a[0] = R0[kx - 3]; a[1] = R0[3 + kx - 3]; a[2] = R0[6 + kx - 3]; e3_cross(o0, a, v0); }
    momentum(E3Off::CRB, a, v0, L, p);
    root_cols(kx, L, p);
  }
  // ---- right-hand side tau - c, Cholesky, qacc0
  const int q0v = E3Off::VEC;   // qacc0
  for (int i = 0; i < 6; ++i) E3S(q0v + i) = -E3S(cv + i);
  for (int l = 1; l < nl; ++l)
    E3S(q0v + 6 + l - 1) = -m.damping[l] * E3S(voff + 6 + l - 1) - m.stiffness[l] * E3S(qoff + 7 + l - 1) - E3S(cv + 6 + l - 1);
  for (int k = 0; k < m.n_act; ++k) { const int l = m.act_link[k]; E3S(q0v + 6 + l - 1) += m.gear[l] * E3S(ctrl_off + k); }
  for (int i = 0; i < nv; ++i)
    for (int k = 0; k <= i; ++k) {
      double sum = E3S(E3Off::M + e3_tri(i, k));
      for (int t = 0; t < k; ++t) sum -= E3S(E3Off::M + e3_tri(i, t)) * E3S(E3Off::M + e3_tri(k, t));
      E3S(E3Off::M + e3_tri(i, k)) = (i == k) ? sqrt(sum) : sum / E3S(E3Off::M + e3_tri(k, k));
    }
  auto fwd_sub = [&](int x) {    // x <- L^-1 x
    for (int i = 0; i < nv; ++i) {
      double sum = E3S(x + i);
      for (int t = 0; t < i; ++t) sum -= E3S(E3Off::M + e3_tri(i, t)) * E3S(x + t);
      E3S(x + i) = sum / E3S(E3Off::M + e3_tri(i, i));
    }
  };
  auto bwd_sub = [&](int x) {    // x <- L^-T x
    for (int i = nv - 1; i >= 0; --i) {
      double sum = E3S(x + i);
      for (int t = i + 1; t < nv; ++t) sum -= E3S(E3Off::M + e3_tri(t, i)) * E3S(x + t);
      E3S(x + i) = sum / E3S(E3Off::M + e3_tri(i, i));
    }
  };
  fwd_sub(q0v); bwd_sub(q0v);
  // ---- constraint rows: contact spheres in model order (normal, tangent x, tangent y), then joint limits
  int nr = 0;
  auto finish_row = [&](int r, double rr, double rdist, int kind, double mu, const double* solref, const double* solimp) {
    // needs J.v and J.qacc0 of the raw row (still in Z), then z = L^-1 j
    const int zr = E3Off::Z + r * NVM;
    double jv = 0.0, jq = 0.0;
    for (int i = 0; i < nv; ++i) { const double j = E3S(zr + i); jv += j * E3S(voff + i); jq += j * E3S(q0v + i); }
    fwd_sub(zr);
    double aii = 0.0;
    for (int i = 0; i < nv; ++i) aii += E3S(zr + i) * E3S(zr + i);
    const double d = e3_impedance(fabs(rdist), solimp);
    const double dmax = solimp[1], tc = solref[0], dr = solref[1];
    const double b = 2.0 / (dmax * tc), ks = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    const double aref = -b * jv - ks * d * rr;
    const int rm = E3Off::RM + 5 * r;
    E3S(rm) = aref - jq; E3S(rm + 1) = (1.0 - d) / d * aii; E3S(rm + 2) = 0.0; E3S(rm + 3) = (double)kind; E3S(rm + 4) = mu;
  };
  for (int ci = 0; ci < m.n_contact; ++ci) {
    const int l = m.contact_link[ci], k = E3Off::KIN + 27 * l;
    double R[9], o[3], rp[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = E3S(k + i);
    e3_ld3(C, k + 9, o);
    e3_matvec(R, m.cpos[ci], rp);
    const double rad = m.crad[ci], dist = o[2] + rp[2] - rad;
    if (dist < m.margin && nr + 3 <= m.max_rows) {
      const double pw[3] = {o[0] + rp[0], o[1] + rp[1], o[2] + rp[2] - (rad + 0.5 * dist)};   // contact point (world)
      for (int d3 = 0; d3 < 3; ++d3) {          // rows: normal z, then tangents x, y
        const int dir = d3 == 0 ? 2 : d3 - 1, zr = E3Off::Z + (nr + d3) * NVM;
        for (int i = 0; i < nv; ++i) E3S(zr + i) = 0.0;
        double dvec[3] = {0.0, 0.0, 0.0};
        dvec[dir] = 1.0;
        for (int j = l; j >= 1; j = m.parent[j]) {
          const int kj = E3Off::KIN + 27 * j;
          double aj[3], oj[3], rel[3], t[3];
          e3_ld3(C, kj + 18, aj); e3_ld3(C, kj + 9, oj);
#pragma unroll
          for (int i = 0; i < 3; ++i) rel[i] = pw[i] - oj[i];
          e3_cross(aj, rel, t);
          E3S(zr + 6 + j - 1) = t[dir];
        }
        E3S(zr + dir) = 1.0;
        for (int kx = 0; kx < 3; ++kx) {
          const double a[3] = {R0[kx], R0[3 + kx], R0[6 + kx]}, rel[3] = {pw[0] - o0[0], pw[1] - o0[1], pw[2] - o0[2]};
          double t[3];
          e3_cross(a, rel, t);
          E3S(zr + 3 + kx) = t[dir];
        }
        (void)dvec;
      }
      finish_row(nr, dist, dist, 0, m.cfric[ci], m.c_solref, m.c_solimp);
      finish_row(nr + 1, 0.0, dist, 1, m.cfric[ci], m.c_solref, m.c_solimp);
      finish_row(nr + 2, 0.0, dist, 2, m.cfric[ci], m.c_solref, m.c_solimp);
      nr += 3;
    }
  }
  for (int l = 1; l < nl; ++l) {
    if (!m.limited[l] || nr + 1 > m.max_rows) continue;
    const double ql = E3S(qoff + 7 + l - 1), lo = m.range[l][0], hi = m.range[l][1];
    double sgn = 0.0, rr = 0.0;
    if (ql - lo < 0.0) { sgn = 1.0; rr = ql - lo; }
    else if (hi - ql < 0.0) { sgn = -1.0; rr = hi - ql; }
    if (sgn == 0.0) continue;
    const int zr = E3Off::Z + nr * NVM;
    for (int i = 0; i < nv; ++i) E3S(zr + i) = 0.0;
    E3S(zr + 6 + l - 1) = sgn;
    finish_row(nr, rr, rr, 3, 0.0, m.l_solref, m.l_solimp);
    nr += 1;
  }
  const int outv = out_off;
  if (nr == 0) {
    for (int i = 0; i < nv; ++i) E3S(outv + i) = E3S(q0v + i);
    return;
  }
  // ---- A = Z Z^T, projected Gauss-Seidel
  for (int r = 0; r < nr; ++r)
    for (int c = 0; c <= r; ++c) {
      double s = 0.0;
      for (int i = 0; i < nv; ++i) s += E3S(E3Off::Z + r * NVM + i) * E3S(E3Off::Z + c * NVM + i);
      E3S(E3Off::A + e3_tri(r, c)) = s;
    }
  for (int it = 0; it < m.pgs_iters; ++it)
    for (int r = 0; r < nr; ++r) {
      const int rm = E3Off::RM + 5 * r;
      double res = E3S(rm);
      for (int c = 0; c < nr; ++c) {
        const double a = r >= c ? E3S(E3Off::A + e3_tri(r, c)) : E3S(E3Off::A + e3_tri(c, r));
        res -= a * E3S(E3Off::RM + 5 * c + 2);
      }
      const double arr = E3S(E3Off::A + e3_tri(r, r)), fold = E3S(rm + 2);
      res += arr * fold;
      double fi = res / (arr + E3S(rm + 1));
      const int kind = (int)E3S(rm + 3);
      if (kind == 1 || kind == 2) {
        const double lim = E3S(rm + 4) * E3S(E3Off::RM + 5 * (r - kind) + 2);
        fi = fmin(fmax(fi, -lim), lim);
      } else {
        fi = fmax(fi, 0.0);
      }
      E3S(rm + 2) = fi;
    }
  // qacc = qacc0 + L^-T (sum_r z_r f_r)
  const int tv = E3Off::VEC + 2 * NVM;
  for (int i = 0; i < nv; ++i) {
    double s = 0.0;
    for (int r = 0; r < nr; ++r) s += E3S(E3Off::Z + r * NVM + i) * E3S(E3Off::RM + 5 * r + 2);
    E3S(tv + i) = s;
  }
  bwd_sub(tv);
  for (int i = 0; i < nv; ++i) E3S(outv + i) = E3S(q0v + i) + E3S(tv + i);
}

// mj_integratePos: dst_q = src_q (+) h * vel   (vel = nv doubles at voff)
__device__ void e3_integrate_pos(const E3Ctx& C, int src_q, int voff, double h, int dst_q) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  for (int i = 0; i < 3; ++i) E3S(dst_q + i) = E3S(src_q + i) + h * E3S(voff + i);
  const double wx = E3S(voff + 3), wy = E3S(voff + 4), wz = E3S(voff + 5);
  double qw = E3S(src_q + 3), qx = E3S(src_q + 4), qy = E3S(src_q + 5), qz = E3S(src_q + 6);
  const double wn = sqrt(wx * wx + wy * wy + wz * wz), ang = wn * h;
  if (ang > 0.0) {
    const double sh = sin(0.5 * ang) / wn, ch = cos(0.5 * ang);
    const double bx = sh * wx, by = sh * wy, bz = sh * wz;
    const double nw = qw * ch - qx * bx - qy * by - qz * bz, nx = qw * bx + qx * ch + qy * bz - qz * by;
    const double ny = qw * by - qx * bz + qy * ch + qz * bx, nz = qw * bz + qx * by - qy * bx + qz * ch;
    qw = nw; qx = nx; qy = ny; qz = nz;
  }
  const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  E3S(dst_q + 3) = qw * nrm; E3S(dst_q + 4) = qx * nrm; E3S(dst_q + 5) = qy * nrm; E3S(dst_q + 6) = qz * nrm;
  for (int l = 1; l < m.nl; ++l) E3S(dst_q + 7 + l - 1) = E3S(src_q + 7 + l - 1) + h * E3S(voff + 6 + l - 1);
}

// scratch state slots (offsets from E3Off::ST)
struct E3St {
  static constexpr int NVM = E3Off::NVMAX, NQM = E3Off::NVMAX + 1;
  static constexpr int Q0 = E3Off::ST, V0 = Q0 + NQM, QS = V0 + NVM, VS = QS + NQM, VSUM = VS + NVM, ASUM = VSUM + NVM, ACC = ASUM + NVM, CTRL = ACC + NVM;
};

// one RK4 substep on the state at (Q0, V0), positions on the manifold (oracle substep())
__device__ void e3_substep(const E3Ctx& C) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  const int nv = m.nv;
  const double h = m.timestep;
  // stage 1
  e3_dynamics(C, E3St::Q0, E3St::V0, E3St::CTRL, E3St::ACC);
  for (int i = 0; i < nv; ++i) { E3S(E3St::VSUM + i) = E3S(E3St::V0 + i); E3S(E3St::ASUM + i) = E3S(E3St::ACC + i); }
  const double hs[3] = {0.5 * h, 0.5 * h, h}, wt[3] = {2.0, 2.0, 1.0};
  for (int st = 0; st < 3; ++st) {
    // stage state: q_s = q0 (+) hs * v_prev_stage, v_s = v0 + hs * a_prev_stage
    e3_integrate_pos(C, E3St::Q0, st == 0 ? E3St::V0 : E3St::VS, hs[st], E3St::QS);
    for (int i = 0; i < nv; ++i) E3S(E3St::VS + i) = E3S(E3St::V0 + i) + hs[st] * E3S(E3St::ACC + i);
    e3_dynamics(C, E3St::QS, E3St::VS, E3St::CTRL, E3St::ACC);
    for (int i = 0; i < nv; ++i) { E3S(E3St::VSUM + i) += wt[st] * E3S(E3St::VS + i); E3S(E3St::ASUM + i) += wt[st] * E3S(E3St::ACC + i); }
  }
  for (int i = 0; i < nv; ++i) E3S(E3St::VSUM + i) *= (1.0 / 6.0);
  e3_integrate_pos(C, E3St::Q0, E3St::VSUM, h, E3St::QS);
  for (int i = 0; i < m.nq; ++i) E3S(E3St::Q0 + i) = E3S(E3St::QS + i);
  for (int i = 0; i < nv; ++i) E3S(E3St::V0 + i) += h / 6.0 * E3S(E3St::ASUM + i);
}

// x of the whole model's centre of mass (humanoid.py:6-9) from the link frames in KIN
__device__ double e3_com(const E3Ctx& C, double* com3) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  double s[3] = {0.0, 0.0, 0.0};
  for (int l = 0; l < m.nl; ++l) {
    if (m.mass[l] == 0.0) continue;
    const int k = E3Off::KIN + 27 * l;
    double R[9], o[3], rc[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = E3S(k + i);
    e3_ld3(C, k + 9, o);
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] += m.mass[l] * (o[i] + rc[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) com3[i] = s[i] / m.total_mass;
  return com3[0];
}

// observation of the state at (Q0, V0) (oracle obs()): KIN must hold the kinematics of that state.  `put(i, value)` receives the
// raw components; the caller applies the ScaledEnv map and the float conversion.
template <class Put>
__device__ void e3_observe(const E3Ctx& C, Put put) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  int at = 0;
  for (int i = 2; i < m.nq; ++i) put(at++, E3S(E3St::Q0 + i));
  for (int i = 0; i < m.nv; ++i) put(at++, E3S(E3St::V0 + i));
  const int nb = m.n_body;
  if (m.task == 3) {   // Ant-v2: clip(cfrc_ext, -1, 1) of 14 bodies — zeros (see oracle/spatial_env.py::obs_extras)
    for (int i = 0; i < (nb + 1) * 6; ++i) put(at++, 0.0);
    return;
  }
  double com[3];
  e3_com(C, com);
  for (int i = 0; i < 10; ++i) put(at++, 0.0);             // cinert of the world body
  for (int b = 0; b < nb; ++b) {
    const int l = m.body_link[b];
    if (!m.body_first[b]) { for (int i = 0; i < 10; ++i) put(at++, 0.0); continue; }   // welded body: its mass sits in the link above
    const int k = E3Off::KIN + 27 * l;
    double R[9], o[3], rc[3], d[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = E3S(k + i);
    e3_ld3(C, k + 9, o);
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = o[i] + rc[i] - com[i];
    const double* I6 = m.inertia[l];
    const double Im[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    double T[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]}, Iw[9];
    e3_mat3mul(R, Im, T); e3_mat3mul(T, Rt, Iw);
    const double ml = m.mass[l], d2 = e3_dot(d, d);
    put(at++, Iw[0] + ml * (d2 - d[0] * d[0])); put(at++, Iw[4] + ml * (d2 - d[1] * d[1])); put(at++, Iw[8] + ml * (d2 - d[2] * d[2]));
    put(at++, Iw[1] - ml * d[0] * d[1]); put(at++, Iw[2] - ml * d[0] * d[2]); put(at++, Iw[5] - ml * d[1] * d[2]);
    put(at++, ml * d[0]); put(at++, ml * d[1]); put(at++, ml * d[2]); put(at++, ml);
  }
  for (int i = 0; i < 6; ++i) put(at++, 0.0);              // cvel of the world body
  for (int b = 0; b < nb; ++b) {
    const int k = E3Off::KIN + 27 * m.body_link[b];
    double o[3], w[3], vo[3], rel[3], t[3];
    e3_ld3(C, k + 9, o); e3_ld3(C, k + 12, w); e3_ld3(C, k + 15, vo);
#pragma unroll
    for (int i = 0; i < 3; ++i) rel[i] = com[i] - o[i];
    e3_cross(w, rel, t);
    put(at++, w[0]); put(at++, w[1]); put(at++, w[2]);
    put(at++, vo[0] + t[0]); put(at++, vo[1] + t[1]); put(at++, vo[2] + t[2]);
  }
  for (int i = 0; i < m.nv; ++i) {                         // qfrc_actuator
    double f = 0.0;
    for (int kk = 0; kk < m.n_act; ++kk)
      if (6 + m.act_link[kk] - 1 == i) f = m.gear[m.act_link[kk]] * E3S(E3St::CTRL + kk);
    put(at++, f);
  }
  for (int i = 0; i < (nb + 1) * 6; ++i) put(at++, 0.0);   // cfrc_ext
}

// ---- task layer: one env.step() on the state at (Q0, V0) — action -> ctrl, frame_skip substeps, reward and termination
// (oracle step()).  Leaves KIN holding the kinematics of the new state, which e3_observe needs.
__device__ void e3_task_step(const E3Ctx& C, const float* act, double& reward, bool& done) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  double ctrl_sq = 0.0;
  for (int k = 0; k < m.n_act; ++k) {   // NormalizedBoxEnv: [-1, 1] -> ctrlrange, clip (wrappers.py:343-346)
    const double a = (double)act[k];
    const double u = fmin(fmax(a * m.ctrl_range, -m.ctrl_range), m.ctrl_range);
    E3S(E3St::CTRL + k) = u;
    const double ac = fmin(fmax(a, -1.0), 1.0);
    ctrl_sq += m.task == 4 ? u * u : ac * ac;   // humanoid.py:45 squares data.ctrl, ant.py:16 the action it was given
  }
  double x0 = E3S(E3St::Q0), com[3];
  if (m.task == 4) { e3_kinematics(C, E3St::Q0, E3St::V0); x0 = e3_com(C, com); }
  for (int s = 0; s < m.frame_skip; ++s) e3_substep(C);
  e3_kinematics(C, E3St::Q0, E3St::V0);
  const double z = E3S(E3St::Q0 + 2);
  if (m.task == 4) {   // humanoid.py:37-49
    const double x1 = e3_com(C, com);
    reward = m.vel_weight * (x1 - x0) / m.timestep - m.ctrl_cost * ctrl_sq + m.alive;
    done = z < m.z_min || z > m.z_max;
  } else {             // ant.py:11-24
    reward = (E3S(E3St::Q0) - x0) / (m.timestep * m.frame_skip) - m.ctrl_cost * ctrl_sq + m.alive;
    bool fin = true;
    for (int i = 0; i < m.nq; ++i) fin = fin && isfinite(E3S(E3St::Q0 + i));
    for (int i = 0; i < m.nv; ++i) fin = fin && isfinite(E3S(E3St::V0 + i));
    done = !(fin && z >= m.z_min && z <= m.z_max);
  }
}

// ---- host: ilsx_spatial_model (include/ilsx.h) -> the device-side constant block.  Returns NULL, or why the model is refused.
static inline const char* e3_build_model(const ilsx_spatial_model* sm, Spatial3Dev& m) {
  if (sm->n_link < 2 || sm->n_link > E3_MAXL) return "n_link out of range";
  if (sm->n_contact < 0 || sm->n_contact > E3_MAXC || sm->n_body < 1 || sm->n_body > E3_MAXB || sm->n_act < 1 || sm->n_act >= E3_MAXL)
    return "model sizes out of range";
  if (sm->task != ILSX_TASK_ANT && sm->task != ILSX_TASK_HUMANOID) return "task is not a 3-D task";
  if (sm->max_rows < 3 || sm->max_rows > E3_MAXR) return "max_rows out of range";
  memset(&m, 0, sizeof m);
  m.nl = sm->n_link; m.nv = 6 + m.nl - 1; m.nq = 7 + m.nl - 1; m.task = sm->task; m.frame_skip = sm->frame_skip; m.pgs_iters = sm->pgs_iters;
  m.max_rows = sm->max_rows; m.n_act = sm->n_act; m.n_contact = sm->n_contact; m.n_body = sm->n_body;
  double tot = 0.0;
  for (int l = 0; l < m.nl; ++l) {
    m.parent[l] = sm->parent[l]; m.limited[l] = sm->limited[l];
    if (l > 0 && (sm->parent[l] < 0 || sm->parent[l] >= l)) return "a link's parent must precede it";
    double R[9];
    const double* q = sm->quat0[l];
    R[0] = 1 - 2 * (q[2] * q[2] + q[3] * q[3]); R[1] = 2 * (q[1] * q[2] - q[0] * q[3]); R[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    R[3] = 2 * (q[1] * q[2] + q[0] * q[3]); R[4] = 1 - 2 * (q[1] * q[1] + q[3] * q[3]); R[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
    R[6] = 2 * (q[1] * q[3] - q[0] * q[2]); R[7] = 2 * (q[2] * q[3] + q[0] * q[1]); R[8] = 1 - 2 * (q[1] * q[1] + q[2] * q[2]);
    for (int i = 0; i < 9; ++i) m.Rq0[l][i] = R[i];
    for (int i = 0; i < 3; ++i) {
      m.anchor[l][i] = sm->anchor[l][i]; m.axis[l][i] = sm->axis[l][i]; m.com[l][i] = sm->com[l][i];
      m.axis_p[l][i] = R[3 * i] * sm->axis[l][0] + R[3 * i + 1] * sm->axis[l][1] + R[3 * i + 2] * sm->axis[l][2];
    }
    for (int i = 0; i < 6; ++i) m.inertia[l][i] = sm->inertia[l][i];
    m.mass[l] = sm->mass[l]; tot += sm->mass[l];
    m.armature[l] = sm->armature[l]; m.damping[l] = sm->damping[l]; m.stiffness[l] = sm->stiffness[l];
    m.range[l][0] = sm->range[l][0]; m.range[l][1] = sm->range[l][1]; m.gear[l] = sm->gear[l];
  }
  m.total_mass = tot;
  for (int k = 0; k < m.n_act; ++k) {
    m.act_link[k] = sm->act_link[k];
    if (sm->act_link[k] < 1 || sm->act_link[k] >= m.nl) return "act_link out of range";
  }
  for (int c = 0; c < m.n_contact; ++c) {
    m.contact_link[c] = sm->contact_link[c]; m.crad[c] = sm->contact_radius[c]; m.cfric[c] = sm->contact_friction[c];
    for (int i = 0; i < 3; ++i) m.cpos[c][i] = sm->contact_pos[c][i];
  }
  for (int b = 0; b < m.n_body; ++b) {
    m.body_link[b] = sm->body_link[b];
    m.body_first[b] = 1;
    for (int b2 = 0; b2 < b; ++b2) if (sm->body_link[b2] == sm->body_link[b]) m.body_first[b] = 0;
  }
  m.timestep = sm->timestep; m.gravity = sm->gravity; m.reset_noise = sm->reset_noise; m.margin = sm->contact_margin;
  m.reset_noise_vel_std = sm->reset_noise_vel_std; m.ctrl_range = sm->ctrl_range;
  for (int i = 0; i < 2; ++i) { m.c_solref[i] = sm->contact_solref[i]; m.l_solref[i] = sm->limit_solref[i]; }
  for (int i = 0; i < 3; ++i) { m.c_solimp[i] = sm->contact_solimp[i]; m.l_solimp[i] = sm->limit_solimp[i]; }
  m.ctrl_cost = sm->ctrl_cost; m.alive = sm->alive_bonus; m.vel_weight = sm->vel_weight; m.z_min = sm->z_min; m.z_max = sm->z_max;
  for (int i = 0; i < m.nq; ++i) m.init_qpos[i] = sm->init_qpos[i];
  const int base = (m.nq - 2) + m.nv, nb1 = m.n_body + 1;
  m.obs_dim = m.task == ILSX_TASK_HUMANOID ? base + nb1 * 10 + nb1 * 6 + m.nv + nb1 * 6 : base + nb1 * 6;
  if (m.obs_dim > E3_MAXOBS) return "observation wider than E3_MAXOBS";
  for (int i = 0; i < E3_MAXOBS; ++i) { m.obs_shift[i] = 0.0; m.obs_inv_scale[i] = 1.0; }
  {
    int depth[E3_MAXL], maxd = 0;
    depth[0] = 0; m.anc_mask[0] = 1u;
    for (int l = 1; l < m.nl; ++l) {
      depth[l] = depth[m.parent[l]] + 1;
      m.anc_mask[l] = m.anc_mask[m.parent[l]] | (1u << l);
      if (depth[l] > maxd) maxd = depth[l];
    }
    m.n_level = maxd + 1;
    for (int l = 0; l < E3_MAXL; ++l) m.depth[l] = l < m.nl ? depth[l] : -1;
    int at = 0;
    for (int d = 0; d <= maxd; ++d) {
      m.lvl_off[d] = at;
      for (int l = 0; l < m.nl; ++l) if (depth[l] == d) m.lvl_link[at++] = l;
    }
    m.lvl_off[maxd + 1] = at;
    for (int l = 0; l < E3_MAXL; ++l) m.link_act[l] = -1;
    for (int k = 0; k < m.n_act; ++k) {
      if (m.link_act[m.act_link[k]] >= 0) return "two actuators on one hinge";
      m.link_act[m.act_link[k]] = k;
    }
  }
  return nullptr;
}
