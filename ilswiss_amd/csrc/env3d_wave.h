// env3d_wave.h — the 3-D articulated-body stepper (Ant-v2 / Humanoid-v2), ONE WAVEFRONT PER ENV, working set in LDS.
//
// Same model, same equations and the same oracle as env3d.h (oracle/spatial_env.py; reference rules rlkit/envs/mujoco/humanoid.py:24-73,
// ant.py:11-43).  env3d.h runs one env per lane with its ~2.8k-double working set in global scratch: every access is a memory round
// trip and a 1024-env launch is 16 waves on a 1024-SIMD chip (measured 36.7 ms per vec-env step, profiles/r02).  Here the 64 lanes of a
// wave share one env: link frames, composite inertias, the mass matrix / Cholesky factor, the constraint rows and the constraint-space
// matrix live in 23 KB of LDS owned by that wave (6 envs resident per CU), and every O(n^2) / O(n^3) stage is spread over the lanes:
//   kinematics            parallel over the links of one tree level (levels in sequence), joint rotations of all links first
//   inertias / wrenches   parallel over links; subtree sums parallel over the 16 components, serial up the tree
//   mass matrix (CRBA)    parallel over rows
//   Cholesky, solves      right-looking / column-oriented: one uniform step per pivot, the update spread over the lanes
//   constraint rows       parallel over contacts, then over (row, dof) Jacobian entries, then one row per lane for z_r = L^-1 j_r
//   A = Z Z^T             parallel over entries
//   Gauss-Seidel          rows in sequence (inherent), residuals kept incrementally: one multiply-add per lane per row update
// Control flow is wave-uniform (one env per wave): the number of active rows, early exits and resets cost what THIS env needs.
// Cross-lane hand-offs go through LDS inside one wave, which executes its LDS operations in order, so a hand-off needs only a
// compiler fence (E3W_SYNC), never a workgroup barrier.
//
// The file also compiles for the host (tests/harness/env3d_host.cpp, E3W_HOST_EMU): a parallel loop becomes a serial loop run
// forwards or backwards; any dependence between iterations of one parallel loop shows up as a mismatch between the two orders.
#pragma once
#include "env3d.h"

#ifdef E3W_HOST_EMU
typedef double e3w_lds;
extern int e3w_reverse;
#define E3W_FOR(i, n) for (int i##_c = 0, i##_n = (n), i = e3w_reverse ? i##_n - 1 : 0; i##_c < i##_n; ++i##_c, i += e3w_reverse ? -1 : 1)
#define E3W_SYNC() ((void)0)
#define E3W_ONE if (true)
#else
typedef __attribute__((address_space(3))) double e3w_lds;
#define E3W_FOR(i, n) for (int i = lane; i < (n); i += 64)
#define E3W_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define E3W_ONE if (lane == 0)
#endif
// keep a group of independent LDS loads ahead of the arithmetic that consumes them (the scheduler otherwise pairs each load with
// its use and every pair pays the full LDS latency)
#ifdef E3W_HOST_EMU
#define E3W_LOADS_FIRST() ((void)0)
#else
#define E3W_LOADS_FIRST() __builtin_amdgcn_sched_barrier(0)
#endif

// phase timing for tools/ubench/env3d_phases.hip: cycles per stage of e3w_dynamics accumulated in LDS behind the working set
#ifdef E3W_PROFILE
#define E3W_T(k) do { const unsigned long long e3w_now = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) { S[E3WOff::TOTAL + (k)] += (double)(e3w_now - (unsigned long long)S[E3WOff::TOTAL + 15]); S[E3WOff::TOTAL + 15] = (double)__builtin_amdgcn_s_memtime(); } \
    E3W_SYNC(); } while (0)
#else
#define E3W_T(k) ((void)0)
#endif
#if defined(E3W_ASM_MARKS) && !defined(E3W_HOST_EMU)
#define E3W_MARK(name) asm volatile("; E3W_MARK " name)
#else
#define E3W_MARK(name) ((void)0)
#endif

// Per-lane values that live across phases (registers on the device: one instance per lane; an array of 64 under host emulation)
struct E3WRegs {
  int ri[6], tj[6];   // the lower-triangle entries t = lane + 64 s this lane owns in the factorisation: row base i (i + 1) / 2 and column j (-1: none)
  int tjt[6];         // tj (tj + 1) / 2: the row base of ROW tj (where the factorisation step k reads M[tj][k])
  // constants of link l == lane (the tree recursions map link l to lane l), so that no phase waits on the model in global memory
  int parent, depth, act, limited;
  unsigned anc;
  double anchor[3], axis_p[3], damping, stiffness, armature, gear, lo, hi;
  unsigned long long par_lo, par_hi;   // every link's parent, 5 bits each (wave-uniform): links 0..11, 12..23
  // model scalars (wave-uniform).  A wavefront fence makes the compiler re-read anything it loaded from memory, the model included;
  // these copies are values, not loads
  int nl, n_level, nc, max_rows, pgs_iters;
  double margin, timestep;
};
#ifdef E3W_HOST_EMU
#define E3W_REGS(ln) regs[ln]
#else
#define E3W_REGS(ln) regs[0]
#endif
#ifdef __HIP_DEVICE_COMPILE__
#define E3W_FMA _Pragma("clang fp contract(fast)")
#else
#define E3W_FMA
#endif

struct E3WOff {   // doubles
  static constexpr int NVM = E3_MAXL + 5, NQM = NVM + 1;
  static constexpr int KIN = 0;                                  // per link 27: R[9] o[3] w[3] vo[3] aw[3] al[3] ao[3]
  static constexpr int CRB = KIN + 27 * E3_MAXL;                 // per link 10: m, h[3], Io[6] about the world origin
  static constexpr int WR = CRB + 10 * E3_MAXL;                  // per link 6: subtree force, moment about the world origin
  static constexpr int M = WR + 6 * E3_MAXL;                     // lower triangle (becomes L)
  static constexpr int Z = M + NVM * (NVM + 1) / 2;              // rows [r][NVM]: j_r, then z_r = L^-1 j_r ; row MAXR: tau - c, then y
  static constexpr int A = Z + (E3_MAXR + 1) * NVM;                   // lower triangle of Z Z^T; before that: contact distances and points
  static constexpr int RM = A + E3_MAXR * (E3_MAXR + 1) / 2;     // per row 8: res, Rg, f, kind, mu, 1/(A_rr+Rg), source, r
  static constexpr int VEC = RM + 8 * E3_MAXR;                   // (free)[NVM], 1/L_kk[NVM], (free)[NVM]
  static constexpr int Q0 = VEC + 3 * NVM, V0 = Q0 + NQM, QS = V0 + NVM, VS = QS + NQM, VSUM = VS + NVM, ASUM = VSUM + NVM,
                       ACC = ASUM + NVM, CTRL = ACC + NVM, TOTAL = CTRL + E3_MAXL;
};
static_assert(E3_MAXC * 4 <= E3_MAXR * (E3_MAXR + 1) / 2, "contact scratch is overlaid on A");

__device__ __forceinline__ void e3w_ld3(const e3w_lds* S, int at, double* v) { v[0] = S[at]; v[1] = S[at + 1]; v[2] = S[at + 2]; }
__device__ __forceinline__ void e3w_st3(e3w_lds* S, int at, const double* v) { S[at] = v[0]; S[at + 1] = v[1]; S[at + 2] = v[2]; }
__device__ __forceinline__ int e3w_tri_row(int t) {   // row of lower-triangle index t
  int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (r * (r + 1) / 2 > t) --r;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  return r;
}

// the last ROW of the triangle that slot s (entries 64 s .. 64 s + 63, row-major) holds, for an nv x nv matrix: step k of the factorisation
// touches entries of rows > k only, so a slot whose last row is <= k has nothing left to do — for every lane, known at compile time
__device__ constexpr int e3w_slot_last_row(int s, int nv) {
  int t = 64 * s + 63;
  const int nt = nv * (nv + 1) / 2;
  if (t > nt - 1) t = nt - 1;
  int r = 0;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  return r;
}

__device__ __forceinline__ void e3w_regs_init(E3WRegs& R, const Spatial3Dev& m, int ln) {
  const int nv = m.nv;
  for (int s = 0; s < 6; ++s) {
    const int t = ln + 64 * s;
    R.ri[s] = 0; R.tj[s] = -1; R.tjt[s] = 0;
    if (t < nv * (nv + 1) / 2) { const int i = e3w_tri_row(t); R.ri[s] = i * (i + 1) / 2; R.tj[s] = t - R.ri[s]; R.tjt[s] = R.tj[s] * (R.tj[s] + 1) / 2; }
  }
  const int l = ln < m.nl ? ln : 0;
  R.parent = m.parent[l]; R.depth = ln < m.nl ? m.depth[l] : -1; R.act = m.link_act[l]; R.limited = ln >= 1 && ln < m.nl && m.limited[l];
  R.anc = m.anc_mask[l];
  for (int i = 0; i < 3; ++i) { R.anchor[i] = m.anchor[l][i]; R.axis_p[i] = m.axis_p[l][i]; }
  R.damping = m.damping[l]; R.stiffness = m.stiffness[l]; R.armature = m.armature[l]; R.gear = m.gear[l];
  R.lo = m.range[l][0]; R.hi = m.range[l][1];
  R.nl = m.nl; R.n_level = m.n_level; R.nc = m.n_contact; R.max_rows = m.max_rows; R.pgs_iters = m.pgs_iters;
  R.margin = m.margin; R.timestep = m.timestep;
  R.par_lo = 0; R.par_hi = 0;
  for (int j = 1; j < m.nl; ++j) {
    if (j < 12) R.par_lo |= (unsigned long long)m.parent[j] << (5 * j);
    else R.par_hi |= (unsigned long long)m.parent[j] << (5 * (j - 12));
  }
}
// The values above are loads from read-only memory, which the compiler is free to repeat instead of keeping (it did: three global
// loads per tree level).  Passing them through an empty asm makes them computed values that must stay in registers.
__device__ __forceinline__ void e3w_regs_pin(E3WRegs& R) {
#ifndef E3W_HOST_EMU
  asm volatile("" : "+v"(R.parent), "+v"(R.depth), "+v"(R.act), "+v"(R.anc));
  for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(R.anchor[i]), "+v"(R.axis_p[i]));
  asm volatile("" : "+v"(R.damping), "+v"(R.stiffness), "+v"(R.armature), "+v"(R.gear));
  for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(R.ri[s]), "+v"(R.tj[s]), "+v"(R.tjt[s]));
  // wave-uniform ones: through readfirstlane, so that they are scalar values
  R.nl = __builtin_amdgcn_readfirstlane(R.nl); R.n_level = __builtin_amdgcn_readfirstlane(R.n_level);
  R.nc = __builtin_amdgcn_readfirstlane(R.nc); R.max_rows = __builtin_amdgcn_readfirstlane(R.max_rows);
  R.pgs_iters = __builtin_amdgcn_readfirstlane(R.pgs_iters);
  R.margin = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(R.margin)), __builtin_amdgcn_readfirstlane(__double2loint(R.margin)));
  R.timestep = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(R.timestep)), __builtin_amdgcn_readfirstlane(__double2loint(R.timestep)));
  R.par_lo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(R.par_lo >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)R.par_lo);
  R.par_hi = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(R.par_hi >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)R.par_hi);
#endif
}
__device__ __forceinline__ int e3w_parent(const E3WRegs& R, int l) {
  return (int)((l < 12 ? R.par_lo >> (5 * l) : R.par_hi >> (5 * (l - 12))) & 31ull);
}
// sin and cos of a joint angle: Cody-Waite reduction by pi/2 (two constants: exact for the few quadrants a joint angle spans) and
// the fdlibm kernel polynomials on [-pi/4, pi/4]; ~1e-16 absolute, a fifth of the instructions of the general-range library pair
__device__ __forceinline__ void e3w_sincos(double x, double& sn, double& cs) {
  E3W_FMA
  const double kf = rint(x * 0.63661977236758134308);
  double r = x - kf * 1.57079632673412561417e+00;
  r = r - kf * 6.07710050650619224932e-11;
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s0 = r + r * z * ps, c0 = 1.0 - 0.5 * z + z * z * pc;
  const int q = (int)kf & 3;
  sn = (q & 1) ? c0 : s0; cs = (q & 1) ? s0 : c0;
  if (q == 1 || q == 2) cs = -cs;
  if (q >= 2) sn = -sn;
}

// sum_t S[a + sa t] S[b + sb t], t < n: eight products per trip so that the 16 loads of a trip are in flight together (a loop with a
// run-time trip count otherwise waits for every load before issuing the next)
__device__ __forceinline__ double e3w_dot(const e3w_lds* S, int a, int sa, int b, int sb, int n) {
  double s = 0.0;
  for (int t = 0; t < n; t += 8) {
    double x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int tc = t + u < n ? t + u : 0;
      x[u] = S[a + sa * tc]; y[u] = S[b + sb * tc];
    }
    E3W_LOADS_FIRST();
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (t + u < n ? x[u] : 0.0) * y[u];
  }
  return s;
}
// 1 / a for a pivot: hardware estimate + two Newton steps on the device (a few ulp; the oracle tolerance is 1e-8)
__device__ __forceinline__ double e3w_rcp(double a) {
#ifdef E3W_HOST_EMU
  return 1.0 / a;
#else
  double x = __builtin_amdgcn_rcp(a);
  x = x * (2.0 - a * x);
  return x * (2.0 - a * x);
#endif
}
#ifndef E3W_HOST_EMU
__device__ __forceinline__ double e3w_readlane(double v, int src) {   // src is wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
#ifndef E3W_HOST_EMU
// One Gauss-Seidel sweep over the rows 0 .. nr - 1 (nr wave-uniform), the rows as TEMPLATE recursion.  Written as `#pragma unroll` over
// r < E3_MAXR with a `break` the loop stayed rolled (the compiler's "loop not unrolled" warning on k_env3dw_step): column c of A — 30
// doubles a lane — then lived in a 256-byte private segment and every row step of the chain began with a scratch load of Acol[r].
// Same operations in the same order.
template <int R>
__device__ __forceinline__ void e3w_pgs_sweep(const double (&Acol)[E3_MAXR], double& res, double& f, double& fn, double arr, double inv,
                                              double mu_eff, double open_hi, int lane, int nidx, int nr) {
  if constexpr (R < E3_MAXR) {
    if (R >= nr) return;
    const double lim = mu_eff * fn;
    const double nw = fmin(fmax((res + arr * f) * inv, -lim), lim + open_hi);
    const double sd = e3w_readlane(nw - f, R), sn = e3w_readlane(nw, R);
    res -= Acol[R] * sd;
    f = lane == R ? sn : f;
    fn = nidx == R ? sn : fn;
    e3w_pgs_sweep<R + 1>(Acol, res, f, fn, arr, inv, mu_eff, open_hi, lane, nidx, nr);
  }
}
#endif
#endif

// Link frames and velocity-product accelerations of the state at (qoff, voff) (oracle kin())
__device__ __forceinline__ void e3w_kinematics(e3w_lds* S, const Spatial3Dev& m, int lane, const E3WRegs* regs, int qoff, int voff) {
  E3W_FMA
  const int nl = E3W_REGS(0).nl, n_level = E3W_REGS(0).n_level;
  E3W_MARK("kin0 begin");
  E3W_FOR(l, nl) {
    const int k = E3WOff::KIN + 27 * l;
    if (l == 0) {
      double qw = S[qoff + 3], qx = S[qoff + 4], qy = S[qoff + 5], qz = S[qoff + 6];
      const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
      qw *= nrm; qx *= nrm; qy *= nrm; qz *= nrm;
      double R[9], o[3] = {S[qoff], S[qoff + 1], S[qoff + 2]}, wb[3] = {S[voff + 3], S[voff + 4], S[voff + 5]}, w[3];
      e3_quat_to_R(qw, qx, qy, qz, R);
      e3_matvec(R, wb, w);
#pragma unroll
      for (int i = 0; i < 9; ++i) S[k + i] = R[i];
      e3w_st3(S, k + 9, o); e3w_st3(S, k + 12, w);
      const double vo[3] = {S[voff], S[voff + 1], S[voff + 2]}, z3[3] = {0.0, 0.0, 0.0};
      e3w_st3(S, k + 15, vo); e3w_st3(S, k + 18, z3); e3w_st3(S, k + 21, z3); e3w_st3(S, k + 24, z3);
    } else {   // rotation relative to the parent: fixed quat0, then Rodrigues about the hinge axis (parked in the link's R slot)
      double s, cang;
      e3w_sincos(S[qoff + 7 + l - 1], s, cang);
      const double c1 = 1.0 - cang;
      const double ax = m.axis[l][0], ay = m.axis[l][1], az = m.axis[l][2];
      const double Rh[9] = {1.0 - c1 * (ay * ay + az * az), -s * az + c1 * ax * ay, s * ay + c1 * ax * az,
                            s * az + c1 * ax * ay, 1.0 - c1 * (ax * ax + az * az), -s * ax + c1 * ay * az,
                            -s * ay + c1 * ax * az, s * ax + c1 * ay * az, 1.0 - c1 * (ax * ax + ay * ay)};
      double Rrel[9];
      e3_mat3mul(m.Rq0[l], Rh, Rrel);
#pragma unroll
      for (int i = 0; i < 9; ++i) S[k + i] = Rrel[i];
    }
  }
  E3W_SYNC();
  E3W_MARK("kin levels begin");
  for (int d = 1; d < n_level; ++d) {
    E3W_FOR(l, nl) {
      const E3WRegs& R_ = E3W_REGS(l);
      if (R_.depth != d) continue;
      const int p = R_.parent, kp = E3WOff::KIN + 27 * p, k = E3WOff::KIN + 27 * l;
      double Rp[9], Rrel[9], R[9], op[3], wp[3], vop[3], alp[3], aop[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) { Rp[i] = S[kp + i]; Rrel[i] = S[k + i]; }
      e3w_ld3(S, kp + 9, op); e3w_ld3(S, kp + 12, wp); e3w_ld3(S, kp + 15, vop); e3w_ld3(S, kp + 21, alp); e3w_ld3(S, kp + 24, aop);
      const double qd = S[voff + 6 + l - 1];
      e3_mat3mul(Rp, Rrel, R);
      double rp[3], aw[3], o[3], w[3], al[3], vo[3], ao[3], t1[3], t2[3];
      e3_matvec(Rp, R_.anchor, rp);
      e3_matvec(Rp, R_.axis_p, aw);
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
      for (int i = 0; i < 9; ++i) S[k + i] = R[i];
      e3w_st3(S, k + 9, o); e3w_st3(S, k + 12, w); e3w_st3(S, k + 15, vo); e3w_st3(S, k + 18, aw); e3w_st3(S, k + 21, al); e3w_st3(S, k + 24, ao);
    }
    E3W_SYNC();
  }
  E3W_MARK("kin end");
}

// x <- L^-T x, column by column: one uniform step per pivot, the update of the remaining entries spread over the lanes (host form;
// the device keeps x in registers, see the end of e3w_dynamics)
__device__ __forceinline__ void e3w_bwd_sub(e3w_lds* S, int lane, int nv, int x) {
  for (int k = nv - 1; k >= 0; --k) {
    const double xk = S[x + k] * S[E3WOff::VEC + E3WOff::NVM + k];
    E3W_FOR(i, k) S[x + i] -= S[E3WOff::M + e3_tri(k, i)] * xk;
    E3W_ONE S[x + k] = xk;
    E3W_SYNC();
  }
}

// qacc = f(q, v, ctrl) with soft constraints (oracle dynamics()) -> S[out_off .. out_off + nv)
template <int NV>
__device__ __forceinline__ void e3w_dynamics(e3w_lds* S, const Spatial3Dev& m, int lane, const E3WRegs* regs, int qoff, int voff, int ctrl_off, int out_off) {
  E3W_FMA
  const int nl = E3W_REGS(0).nl, nv = NV > 0 ? NV : m.nv;   // NV > 0: the model's dof count at compile time (row solves and solves fully unrolled)
  constexpr int NVM = E3WOff::NVM;
  E3W_T(12);
  e3w_kinematics(S, m, lane, regs, qoff, voff);
  E3W_T(0);
  // ---- per link: composite-inertia seed about the world origin, Newton-Euler wrench about the world origin
  E3W_FOR(l, nl) {
    const int k = E3WOff::KIN + 27 * l, cb = E3WOff::CRB + 10 * l, wr = E3WOff::WR + 6 * l;
    const double ml = m.mass[l];
    double R[9], o[3], w[3], al[3], ao[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = S[k + i];
    e3w_ld3(S, k + 9, o); e3w_ld3(S, k + 12, w); e3w_ld3(S, k + 21, al); e3w_ld3(S, k + 24, ao);
    double rc[3], cw[3];
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) cw[i] = o[i] + rc[i];
    const double* I6 = m.inertia[l];
    const double Im[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    double T[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]}, Iw[9];
    e3_mat3mul(R, Im, T);
    e3_mat3mul(T, Rt, Iw);
    const double c2 = e3_dot(cw, cw);
    S[cb] = ml;
    S[cb + 1] = ml * cw[0]; S[cb + 2] = ml * cw[1]; S[cb + 3] = ml * cw[2];
    S[cb + 4] = Iw[0] + ml * (c2 - cw[0] * cw[0]); S[cb + 5] = Iw[4] + ml * (c2 - cw[1] * cw[1]); S[cb + 6] = Iw[8] + ml * (c2 - cw[2] * cw[2]);
    S[cb + 7] = Iw[1] - ml * cw[0] * cw[1]; S[cb + 8] = Iw[2] - ml * cw[0] * cw[2]; S[cb + 9] = Iw[5] - ml * cw[1] * cw[2];
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
    e3w_st3(S, wr, F); e3w_st3(S, wr + 3, N);
  }
  E3W_FOR(t, nv * (nv + 1) / 2) S[E3WOff::M + t] = 0.0;
  E3W_SYNC();
  E3W_T(1);
  // ---- subtree sums, one component per lane, children before parents (a link's index exceeds its parent's)
  E3W_FOR(c, 16) {
    const int base = c < 10 ? E3WOff::CRB + c : E3WOff::WR + (c - 10), stride = c < 10 ? 10 : 6;
    for (int l = nl - 1; l >= 1; --l) S[base + stride * e3w_parent(E3W_REGS(c), l)] += S[base + stride * l];
  }
  E3W_SYNC();
  E3W_T(2);
  // ---- right-hand side tau - c (c: the subtree wrench projected on each dof) ; mass-matrix rows (CRBA)
  const int yrow = E3WOff::Z + E3_MAXR * NVM;   // the right-hand side rides through L^-1 as one more row
  E3W_FOR(ln, 64) {
    const E3WRegs& R_ = E3W_REGS(ln);
    const int ix = ln >= 32 ? ln - 32 : (ln >= 1 && ln < nl ? ln + 5 : -1);   // 0..5: root dofs (lanes 32..37) ; 6 + l - 1: link l on lane l
    if (ix < 0 || (ln >= 32 && ix >= 6)) continue;
    double R0[9], o0[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = S[E3WOff::KIN + i];
    e3w_ld3(S, E3WOff::KIN + 9, o0);
    double a[3] = {0.0, 0.0, 0.0}, v0[3] = {0.0, 0.0, 0.0};
    int row, cb, l = 0;
    if (ix < 6) {          // root dof ix: translation along world x/y/z, rotation about the body axes
      row = ix; cb = E3WOff::CRB;
      if (ix < 3) v0[ix] = 1.0;
      else { a[0] = R0[ix - 3]; a[1] = R0[3 + ix - 3]; a[2] = R0[6 + ix - 3]; e3_cross(o0, a, v0); }
    } else {
      l = ix - 6 + 1;
      row = 6 + l - 1; cb = E3WOff::CRB + 10 * l;
      double o[3];
      e3w_ld3(S, E3WOff::KIN + 27 * l + 18, a); e3w_ld3(S, E3WOff::KIN + 27 * l + 9, o);
      e3_cross(o, a, v0);
    }
    // bias: S_row . (subtree wrench about the world origin)
    {
      const int wr = E3WOff::WR + 6 * l;
      double f[3], n[3];
      e3w_ld3(S, wr, f); e3w_ld3(S, wr + 3, n);
      const double cvv = e3_dot(a, n) + e3_dot(v0, f);
      double tau = 0.0;
      if (l >= 1) {
        tau = -R_.damping * S[voff + 6 + l - 1] - R_.stiffness * S[qoff + 7 + l - 1];
        if (R_.act >= 0) tau += R_.gear * S[ctrl_off + R_.act];
      }
      S[yrow + row] = tau - cvv;
    }
    // momentum of the composite below this dof moving with (omega = a, v0): L about the world origin, p
    double L[3], p[3];
    {
      const double mc = S[cb], h[3] = {S[cb + 1], S[cb + 2], S[cb + 3]};
      const double Ixx = S[cb + 4], Iyy = S[cb + 5], Izz = S[cb + 6], Ixy = S[cb + 7], Ixz = S[cb + 8], Iyz = S[cb + 9];
      double t[3];
      e3_cross(a, h, t);
      p[0] = mc * v0[0] + t[0]; p[1] = mc * v0[1] + t[1]; p[2] = mc * v0[2] + t[2];
      e3_cross(h, v0, t);
      L[0] = Ixx * a[0] + Ixy * a[1] + Ixz * a[2] + t[0];
      L[1] = Ixy * a[0] + Iyy * a[1] + Iyz * a[2] + t[1];
      L[2] = Ixz * a[0] + Iyz * a[1] + Izz * a[2] + t[2];
    }
    if (l >= 1) {
      S[E3WOff::M + e3_tri(row, row)] = e3_dot(a, L) + e3_dot(v0, p) + R_.armature;
      for (unsigned left = R_.anc & ~(1u << l) & ~1u; left; left &= left - 1) {   // the hinges above this one
        const int j = __builtin_ctz(left);
        const int kj = E3WOff::KIN + 27 * j;
        double aj[3], oj[3], vj[3];
        e3w_ld3(S, kj + 18, aj); e3w_ld3(S, kj + 9, oj);
        e3_cross(oj, aj, vj);
        S[E3WOff::M + e3_tri(row, 6 + j - 1)] = e3_dot(aj, L) + e3_dot(vj, p);
      }
    }
    for (int kx = 0; kx < 3; ++kx) {   // the six root columns of this row
      if (kx <= row) S[E3WOff::M + e3_tri(row, kx)] = p[kx];
      const double ar[3] = {R0[kx], R0[3 + kx], R0[6 + kx]};
      double vr[3];
      e3_cross(o0, ar, vr);
      if (3 + kx <= row) S[E3WOff::M + e3_tri(row, 3 + kx)] = e3_dot(ar, L) + e3_dot(vr, p);
    }
  }
  E3W_SYNC();
  E3W_T(3);
  // ---- Cholesky in place.  Step k subtracts column k's outer product from the trailing block using the UNSCALED column
  // (M_ik M_jk / M_kk), so a step is one fence-to-fence phase; the columns are scaled by 1 / sqrt(pivot) in one pass at the end.
  E3W_MARK("chol begin");
  constexpr int NSLOT = NV > 0 ? (NV * (NV + 1) / 2 + 63) / 64 : 6;
  if constexpr (NV > 0) {
    // the pivot as a compile-time value: slots that hold no row beyond k are skipped for the whole wavefront (Humanoid: 88 slot steps instead
    // of 115), the test against k is a compare with an immediate and row tj's base comes from a register (E3WRegs::tjt) — the step is a
    // count of issued instructions (§3b): ~110 per pivot before.  Same operations on the entries that are touched.
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double pinv = e3w_rcp(S[E3WOff::M + e3_tri(k, k)]);
      E3W_FOR(ln, 64) {
        const E3WRegs& R = E3W_REGS(ln);
        double ci[NSLOT], cj[NSLOT], mt[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {   // loads are unconditional (every address is inside the triangle), the store is not
          if (e3w_slot_last_row(sl, NV) <= k) continue;
          ci[sl] = S[E3WOff::M + R.ri[sl] + k];
          cj[sl] = S[E3WOff::M + (R.tj[sl] > k ? R.tjt[sl] + k : 0)];
          mt[sl] = S[E3WOff::M + ln + 64 * sl];
        }
        E3W_LOADS_FIRST();
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
          if (e3w_slot_last_row(sl, NV) <= k) continue;
          if (R.tj[sl] > k) S[E3WOff::M + ln + 64 * sl] = mt[sl] - ci[sl] * cj[sl] * pinv;
        }
      }
      E3W_SYNC();
    }
  } else
  for (int k = 0; k < nv; ++k) {
    const double pinv = e3w_rcp(S[E3WOff::M + e3_tri(k, k)]);
    E3W_FOR(ln, 64) {
      const E3WRegs& R = E3W_REGS(ln);
      double ci[NSLOT], cj[NSLOT], mt[NSLOT];
#pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl) {   // loads are unconditional (every address is inside the triangle), the store is not
        ci[sl] = S[E3WOff::M + R.ri[sl] + k];
        cj[sl] = S[E3WOff::M + (R.tj[sl] > k ? R.tj[sl] * (R.tj[sl] + 1) / 2 + k : 0)];
        mt[sl] = S[E3WOff::M + ln + 64 * sl];
      }
      E3W_LOADS_FIRST();
#pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl)
        if (R.tj[sl] > k) S[E3WOff::M + ln + 64 * sl] = mt[sl] - ci[sl] * cj[sl] * pinv;
    }
    E3W_SYNC();
  }
  E3W_MARK("chol end");
  E3W_FOR(k, nv) S[E3WOff::VEC + NVM + k] = 1.0 / sqrt(S[E3WOff::M + e3_tri(k, k)]);
  E3W_SYNC();
  E3W_FOR(ln, 64) {
    const E3WRegs& R = E3W_REGS(ln);
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl)
      if (R.tj[sl] >= 0) S[E3WOff::M + ln + 64 * sl] *= S[E3WOff::VEC + NVM + R.tj[sl]];
  }
  E3W_T(4);
  // ---- contacts (lanes 32 .. 32 + n_contact): distance and contact point of every sphere
  const int nc = E3W_REGS(0).nc, max_rows = E3W_REGS(0).max_rows;
  const double margin = E3W_REGS(0).margin;
  E3W_FOR(ln, 64) {
    const int ci = ln - 32;
    if (ci < 0 || ci >= nc) continue;
    const int l = m.contact_link[ci], k = E3WOff::KIN + 27 * l;
    double R[9], o[3], rp[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = S[k + i];
    e3w_ld3(S, k + 9, o);
    e3_matvec(R, m.cpos[ci], rp);
    const double rad = m.crad[ci], dist = o[2] + rp[2] - rad;
    S[E3WOff::A + ci] = dist;
    S[E3WOff::A + E3_MAXC + 3 * ci] = o[0] + rp[0]; S[E3WOff::A + E3_MAXC + 3 * ci + 1] = o[1] + rp[1];
    S[E3WOff::A + E3_MAXC + 3 * ci + 2] = o[2] + rp[2] - (rad + 0.5 * dist);
  }
  E3W_SYNC();
  // ---- row table, in model order: contacts (normal, tangent x, tangent y) while three rows fit, then violated joint limits while
  // one fits.  Row record: [2] f, [3] kind, [4] friction or limit sign, [6] contact / link index, [7] r.
  int nr = 0;
#ifdef E3W_HOST_EMU
  for (int ci = 0; ci < nc; ++ci) {
    const double dist = S[E3WOff::A + ci];
    if (dist < margin && nr + 3 <= max_rows) {
      for (int d3 = 0; d3 < 3; ++d3) {
        const int rm = E3WOff::RM + 8 * (nr + d3);
        S[rm + 2] = 0.0; S[rm + 3] = (double)d3; S[rm + 4] = m.cfric[ci]; S[rm + 6] = (double)ci; S[rm + 7] = d3 == 0 ? dist : 0.0;
      }
      nr += 3;
    }
  }
  for (int l = 1; l < nl; ++l) {
    if (!m.limited[l] || nr + 1 > max_rows) continue;
    const double ql = S[qoff + 7 + l - 1], lo = m.range[l][0], hi = m.range[l][1];
    double sgn = 0.0, rr = 0.0;
    if (ql - lo < 0.0) { sgn = 1.0; rr = ql - lo; }
    else if (hi - ql < 0.0) { sgn = -1.0; rr = hi - ql; }
    if (sgn == 0.0) continue;
    const int rm = E3WOff::RM + 8 * nr;
    S[rm + 2] = 0.0; S[rm + 3] = 3.0; S[rm + 4] = sgn; S[rm + 6] = (double)l; S[rm + 7] = rr;
    nr += 1;
  }
#else
  {   // contact ci is lane 32 + ci, link l is lane l: a row's position is a population count over the lanes below it
    const int ci = lane - 32;
    const bool isc = ci >= 0 && ci < nc;
    const double dist = S[E3WOff::A + (isc ? ci : 0)];
    const bool cflag = isc && dist < margin;
    double sgn = 0.0, rr = 0.0;
    if (regs[0].limited) {
      const double ql = S[qoff + 7 + lane - 1], lo = regs[0].lo, hi = regs[0].hi;
      if (ql - lo < 0.0) { sgn = 1.0; rr = ql - lo; }
      else if (hi - ql < 0.0) { sgn = -1.0; rr = hi - ql; }
    }
    const bool lflag = sgn != 0.0;
    const unsigned long long cmask = __ballot(cflag), lmask = __ballot(lflag), below = (1ull << lane) - 1ull;
    const int tot_c = __popcll(cmask), acc_c = tot_c < max_rows / 3 ? tot_c : max_rows / 3, nr_c = 3 * acc_c;
    const int tot_l = __popcll(lmask), room = max_rows - nr_c, acc_l = tot_l < room ? tot_l : room;
    nr = nr_c + acc_l;
    if (cflag) {
      const int rank = __popcll(cmask & below);
      if (rank < acc_c) {
        const double mu = m.cfric[ci];
        for (int d3 = 0; d3 < 3; ++d3) {
          const int rm = E3WOff::RM + 8 * (3 * rank + d3);
          S[rm + 2] = 0.0; S[rm + 3] = (double)d3; S[rm + 4] = mu; S[rm + 6] = (double)ci; S[rm + 7] = d3 == 0 ? dist : 0.0;
        }
      }
    }
    if (lflag) {
      const int rank = __popcll(lmask & below);
      if (rank < acc_l) {
        const int rm = E3WOff::RM + 8 * (nr_c + rank);
        S[rm + 2] = 0.0; S[rm + 3] = 3.0; S[rm + 4] = sgn; S[rm + 6] = (double)lane; S[rm + 7] = rr;
      }
    }
  }
#endif
  E3W_SYNC();
  E3W_T(6);
  // ---- Jacobian rows j_r, one (row, dof) entry per lane-iteration
  E3W_FOR(e, nr * nv) {
    const int r = e / nv, i = e - r * nv, rm = E3WOff::RM + 8 * r;
    const int kind = (int)S[rm + 3], src = (int)S[rm + 6];
    double val = 0.0;
    if (kind == 3) {
      val = (i == 6 + src - 1) ? S[rm + 4] : 0.0;
    } else {
      const int dir = kind == 0 ? 2 : kind - 1, l = m.contact_link[src];
      double pw[3];
      e3w_ld3(S, E3WOff::A + E3_MAXC + 3 * src, pw);
      if (i < 3) val = (i == dir) ? 1.0 : 0.0;
      else {
        double aj[3], oj[3], rel[3], t[3];
        bool on = true;
        if (i < 6) {
          aj[0] = S[E3WOff::KIN + i - 3]; aj[1] = S[E3WOff::KIN + 3 + i - 3]; aj[2] = S[E3WOff::KIN + 6 + i - 3];
          e3w_ld3(S, E3WOff::KIN + 9, oj);
        } else {
          const int j = i - 6 + 1;
          on = (m.anc_mask[l] >> j) & 1u;
          e3w_ld3(S, E3WOff::KIN + 27 * j + 18, aj); e3w_ld3(S, E3WOff::KIN + 27 * j + 9, oj);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rel[c] = pw[c] - oj[c];
        e3_cross(aj, rel, t);
        val = on ? t[dir] : 0.0;
      }
    }
    S[E3WOff::Z + r * NVM + i] = val;
  }
  E3W_SYNC();
  E3W_T(7);
  // ---- one vector per lane through L^-1: the nr rows (z_r = L^-1 j_r) and, as lane nr, the right-hand side (y = L^-1 (tau - c)).
  // Row-oriented and private to the lane; the loads of a row of L are broadcasts.
  E3W_MARK("rowsolve begin");
  E3W_FOR(r, nr + 1) {
    const int zr = r == nr ? yrow : E3WOff::Z + r * NVM, rm = E3WOff::RM + 8 * r;
    double jv = 0.0, aii = 0.0;
    if constexpr (NV > 0) {   // the vector stays in registers; rows of L arrive as broadcast loads at compile-time offsets
      double z[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) z[i] = S[zr + i];
#pragma unroll
      for (int i = 0; i < NV; ++i) jv += z[i] * S[voff + i];
      // column by column: once z_t is final every later row takes its term L_it z_t — NV - t - 1 independent loads and multiply-adds, so the
      // dependent chain is two operations per pivot (276 in the row-by-row form for Humanoid, where every term waited for its own broadcast
      // load); a row still receives its terms in ascending t: the same sums
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        z[t] = z[t] * S[E3WOff::VEC + NVM + t];
        aii += z[t] * z[t];
#pragma unroll
        for (int i = t + 1; i < NV; ++i) z[i] -= S[E3WOff::M + i * (i + 1) / 2 + t] * z[t];
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) S[zr + i] = z[i];
    } else {
      jv = r == nr ? 0.0 : e3w_dot(S, zr, 1, voff, 1, nv);
      for (int i = 0; i < nv; ++i) {
        const double sum = (S[zr + i] - e3w_dot(S, E3WOff::M + i * (i + 1) / 2, 1, zr, 1, i)) * S[E3WOff::VEC + NVM + i];
        S[zr + i] = sum;
        aii += sum * sum;
      }
    }
    if (r < nr) { S[rm] = jv; S[rm + 1] = aii; }
  }
  E3W_MARK("rowsolve end");
  E3W_SYNC();
  if (nr > 0) {
    // ---- per row: J.qacc0 = z_r . y, the regulariser and the reference acceleration
    E3W_FOR(r, nr) {
      const int zr = E3WOff::Z + r * NVM, rm = E3WOff::RM + 8 * r;
      const double jv = S[rm], aii = S[rm + 1], jq = e3w_dot(S, zr, 1, yrow, 1, nv);
      const int kind = (int)S[rm + 3];
      const double rr = S[rm + 7];
      const double rdist = kind == 3 ? rr : S[E3WOff::A + (int)S[rm + 6]];
      const double* solref = kind == 3 ? m.l_solref : m.c_solref;
      const double* solimp = kind == 3 ? m.l_solimp : m.c_solimp;
      const double d = e3_impedance(fabs(rdist), solimp);
      const double dmax = solimp[1], tc = solref[0], dr = solref[1];
      const double b = 2.0 / (dmax * tc), ks = 1.0 / (dmax * dmax * tc * tc * dr * dr);
      const double aref = -b * jv - ks * d * rr;
      const double Rg = (1.0 - d) / d * aii;
      S[rm] = aref - jq;          // residual rhs - A f at f = 0
      S[rm + 1] = Rg;
      S[rm + 5] = 1.0 / (aii + Rg);
    }
    E3W_SYNC();
    E3W_T(8);
    // ---- A = Z Z^T (overwrites the contact scratch: every reader of it is behind the fence above)
    E3W_FOR(ln, 64) {
      const int li = ln >> 3, lj = ln & 7;
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
          const int r = li + 8 * a, c = lj + 8 * b;
          if (r < nr && c <= r) S[E3WOff::A + e3_tri(r, c)] = e3w_dot(S, E3WOff::Z + r * NVM, 1, E3WOff::Z + c * NVM, 1, nv);
        }
    }
    E3W_SYNC();
    E3W_T(9);
    // ---- projected Gauss-Seidel: rows in sequence (inherent); each update changes one force and every row's residual
#ifdef E3W_HOST_EMU
    for (int it = 0, n_it = E3W_REGS(0).pgs_iters; it < n_it; ++it)
      for (int r = 0; r < nr; ++r) {
        const int rm = E3WOff::RM + 8 * r;
        const double arr = S[E3WOff::A + e3_tri(r, r)], fold = S[rm + 2];
        double fi = (S[rm] + arr * fold) * S[rm + 5];
        const int kind = (int)S[rm + 3];
        if (kind == 1 || kind == 2) {
          const double lim = S[rm + 4] * S[E3WOff::RM + 8 * (r - kind) + 2];
          fi = fmin(fmax(fi, -lim), lim);
        } else {
          fi = fmax(fi, 0.0);
        }
        const double delta = fi - fold;
        E3W_FOR(c, nr) S[E3WOff::RM + 8 * c] -= (r >= c ? S[E3WOff::A + e3_tri(r, c)] : S[E3WOff::A + e3_tri(c, r)]) * delta;
        E3W_ONE S[rm + 2] = fi;
        E3W_SYNC();
      }
#else
    {   // lane c is row c: residual, force, 1 / (A_cc + R_c), friction bound and column c of A stay in registers.  Every lane works out
        // its own row's candidate each step (same instruction count as one lane doing it); the row whose turn it is publishes its
        // change through readlane and every lane folds it into its residual with one multiply-add.
      const bool on = lane < nr;
      const int rmc = E3WOff::RM + 8 * (on ? lane : 0);
      const int kind = (int)S[rmc + 3];
      const bool fric = kind == 1 || kind == 2;
      double res = S[rmc], f = 0.0, fn = 0.0;
      const double inv = S[rmc + 5], arr = S[E3WOff::A + (on ? e3_tri(lane, lane) : 0)];
      const double mu_eff = fric ? S[rmc + 4] : 0.0, open_hi = fric ? 0.0 : __builtin_inf();   // bounds: [-mu fn, mu fn] or [0, inf)
      const int nidx = lane - kind;          // the row holding this row's normal force (friction rows)
      double Acol[E3_MAXR];
#pragma unroll
      for (int r = 0; r < E3_MAXR; ++r) {
        const int hi = r > lane ? r : lane, lo = r > lane ? lane : r;
        Acol[r] = S[E3WOff::A + ((on && r < nr) ? e3_tri(hi, lo) : 0)];
      }
      for (int it = 0, n_it = E3W_REGS(0).pgs_iters; it < n_it; ++it)
        e3w_pgs_sweep<0>(Acol, res, f, fn, arr, inv, mu_eff, open_hi, lane, nidx, nr);
      if (on) S[rmc + 2] = f;
    }
    E3W_SYNC();
#endif
    E3W_T(10);
    // ---- y + sum_r z_r f_r
    E3W_FOR(i, nv) S[yrow + i] += e3w_dot(S, E3WOff::Z + i, NVM, E3WOff::RM + 2, 8, nr);
    E3W_SYNC();
  }
  // ---- qacc = L^-T (y + sum_r z_r f_r)
#ifdef E3W_HOST_EMU
  e3w_bwd_sub(S, lane, nv, yrow);
  E3W_FOR(i, nv) S[out_off + i] = S[yrow + i];
#else
  {   // lane i keeps x_i; pivot k's value and 1 / L_kk reach every lane through readlane, row k of L is fetched one step ahead
    double x = S[yrow + (lane < nv ? lane : 0)];
    const double invd = S[E3WOff::VEC + NVM + (lane < nv ? lane : 0)];
    double Lnext = S[E3WOff::M + e3_tri(nv - 1, lane < nv - 1 ? lane : 0)];
#pragma unroll
    for (int k = nv - 1; k >= 0; --k) {
      const double Lk = lane < k ? Lnext : 0.0;
      if (k > 0) Lnext = S[E3WOff::M + e3_tri(k - 1, lane < k - 1 ? lane : 0)];
      const double xk = e3w_readlane(x, k) * e3w_readlane(invd, k);
      x = lane == k ? xk : x - Lk * xk;
    }
    if (lane < nv) S[out_off + lane] = x;
  }
#endif
  E3W_SYNC();
  E3W_T(11);
}

// mj_integratePos: dst_q = src_q (+) h * vel.  The quaternion update is computed by every lane (uniform) and written by one.
__device__ __forceinline__ void e3w_integrate_pos(e3w_lds* S, const Spatial3Dev& m, int lane, int src_q, int voff, double h, int dst_q) {
  const double wx = S[voff + 3], wy = S[voff + 4], wz = S[voff + 5];
  double qw = S[src_q + 3], qx = S[src_q + 4], qy = S[src_q + 5], qz = S[src_q + 6];
  const double wn = sqrt(wx * wx + wy * wy + wz * wz), ang = wn * h;
  if (ang > 0.0) {
    double sh, ch;
    e3w_sincos(0.5 * ang, sh, ch);
    sh /= wn;
    const double bx = sh * wx, by = sh * wy, bz = sh * wz;
    const double nw = qw * ch - qx * bx - qy * by - qz * bz, nx = qw * bx + qx * ch + qy * bz - qz * by;
    const double ny = qw * by - qx * bz + qy * ch + qz * bx, nz = qw * bz + qx * by - qy * bx + qz * ch;
    qw = nw; qx = nx; qy = ny; qz = nz;
  }
  const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  const double p0 = S[src_q] + h * S[voff], p1 = S[src_q + 1] + h * S[voff + 1], p2 = S[src_q + 2] + h * S[voff + 2];
  E3W_SYNC();   // src_q may be dst_q's neighbour in a later phase; keep reads ahead of the writes below
  E3W_ONE {
    S[dst_q] = p0; S[dst_q + 1] = p1; S[dst_q + 2] = p2;
    S[dst_q + 3] = qw * nrm; S[dst_q + 4] = qx * nrm; S[dst_q + 5] = qy * nrm; S[dst_q + 6] = qz * nrm;
  }
  E3W_FOR(j, m.nl - 1) S[dst_q + 7 + j] = S[src_q + 7 + j] + h * S[voff + 6 + j];
  E3W_SYNC();
}

// one RK4 substep on the state at (Q0, V0), positions on the manifold (oracle substep()).  Four stages through ONE dynamics call
// site so that the largest piece of code exists once.
template <int NV>
__device__ __forceinline__ void e3w_substep(e3w_lds* S, const Spatial3Dev& m, int lane, const E3WRegs* regs) {
  E3W_FMA
  const int nv = NV > 0 ? NV : m.nv;
  const double h = E3W_REGS(0).timestep;
#pragma unroll 1
  for (int st = 0; st < 4; ++st) {
    const double hs = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h), wt = (st == 1 || st == 2) ? 2.0 : 1.0;
    if (st == 0) {
      E3W_FOR(i, m.nq) S[E3WOff::QS + i] = S[E3WOff::Q0 + i];
      E3W_FOR(i, nv) { S[E3WOff::VS + i] = S[E3WOff::V0 + i]; S[E3WOff::VSUM + i] = 0.0; S[E3WOff::ASUM + i] = 0.0; }
      E3W_SYNC();
    } else {   // stage state: q_s = q0 (+) hs * v_(previous stage), v_s = v0 + hs * a_(previous stage)
      e3w_integrate_pos(S, m, lane, E3WOff::Q0, E3WOff::VS, hs, E3WOff::QS);
      E3W_FOR(i, nv) S[E3WOff::VS + i] = S[E3WOff::V0 + i] + hs * S[E3WOff::ACC + i];
      E3W_SYNC();
    }
    e3w_dynamics<NV>(S, m, lane, regs, E3WOff::QS, E3WOff::VS, E3WOff::CTRL, E3WOff::ACC);
    E3W_FOR(i, nv) { S[E3WOff::VSUM + i] += wt * S[E3WOff::VS + i]; S[E3WOff::ASUM + i] += wt * S[E3WOff::ACC + i]; }
    E3W_SYNC();
  }
  E3W_FOR(i, nv) S[E3WOff::VSUM + i] *= (1.0 / 6.0);
  E3W_SYNC();
  e3w_integrate_pos(S, m, lane, E3WOff::Q0, E3WOff::VSUM, h, E3WOff::QS);
  E3W_FOR(i, m.nq) S[E3WOff::Q0 + i] = S[E3WOff::QS + i];
  E3W_FOR(i, nv) S[E3WOff::V0 + i] += h / 6.0 * S[E3WOff::ASUM + i];
  E3W_SYNC();
}

// whole-model centre of mass from the link frames in KIN (every lane computes it: uniform reads, no hand-off)
__device__ __forceinline__ void e3w_com(const e3w_lds* S, const Spatial3Dev& m, double* com3) {
  double s[3] = {0.0, 0.0, 0.0};
  for (int l = 0; l < m.nl; ++l) {
    if (m.mass[l] == 0.0) continue;
    const int k = E3WOff::KIN + 27 * l;
    double R[9], o[3], rc[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = S[k + i];
    e3w_ld3(S, k + 9, o);
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] += m.mass[l] * (o[i] + rc[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) com3[i] = s[i] / m.total_mass;
}

// env.step() on the state at (Q0, V0) (oracle step()); leaves KIN = kinematics of the new state.  reward / done are wave-uniform.
template <int NV>
__device__ __forceinline__ void e3w_task_step(e3w_lds* S, const Spatial3Dev& m, int lane, const E3WRegs* regs, const float* act, double& reward, bool& done) {
  E3W_FMA
  double ctrl_sq = 0.0;
  for (int k = 0; k < m.n_act; ++k) {   // NormalizedBoxEnv: [-1, 1] -> ctrlrange, clip (wrappers.py:343-346)
    const double a = (double)act[k];
    const double u = fmin(fmax(a * m.ctrl_range, -m.ctrl_range), m.ctrl_range);
    const double ac = fmin(fmax(a, -1.0), 1.0);
    ctrl_sq += m.task == 4 ? u * u : ac * ac;   // humanoid.py:45 squares data.ctrl, ant.py:16 the action it was given
  }
  E3W_FOR(k, m.n_act) S[E3WOff::CTRL + k] = fmin(fmax((double)act[k] * m.ctrl_range, -m.ctrl_range), m.ctrl_range);
  E3W_SYNC();
  double x0 = S[E3WOff::Q0], com[3];
  if (m.task == 4) { e3w_kinematics(S, m, lane, regs, E3WOff::Q0, E3WOff::V0); e3w_com(S, m, com); x0 = com[0]; }
#pragma unroll 1
  for (int s = 0; s < m.frame_skip; ++s) e3w_substep<NV>(S, m, lane, regs);
  e3w_kinematics(S, m, lane, regs, E3WOff::Q0, E3WOff::V0);
  const double z = S[E3WOff::Q0 + 2];
  if (m.task == 4) {   // humanoid.py:37-49
    e3w_com(S, m, com);
    reward = m.vel_weight * (com[0] - x0) / m.timestep - m.ctrl_cost * ctrl_sq + m.alive;
    done = z < m.z_min || z > m.z_max;
  } else {             // ant.py:11-24
    reward = (S[E3WOff::Q0] - x0) / (m.timestep * m.frame_skip) - m.ctrl_cost * ctrl_sq + m.alive;
    bool fin = true;
    for (int i = 0; i < m.nq; ++i) fin = fin && isfinite(S[E3WOff::Q0 + i]);
    for (int i = 0; i < m.nv; ++i) fin = fin && isfinite(S[E3WOff::V0 + i]);
    done = !(fin && z >= m.z_min && z <= m.z_max);
  }
}

// observation of the state at (Q0, V0) (oracle obs()); KIN must hold its kinematics.  put(i, value) is called once per component,
// components spread over the lanes.
template <class Put>
__device__ __forceinline__ void e3w_observe(const e3w_lds* S, const Spatial3Dev& m, int lane, Put put) {
  const int nq2 = m.nq - 2, nv = m.nv, nb = m.n_body;
  E3W_FOR(i, nq2) put(i, S[E3WOff::Q0 + 2 + i]);
  E3W_FOR(i, nv) put(nq2 + i, S[E3WOff::V0 + i]);
  int at = nq2 + nv;
  if (m.task == 3) {   // Ant-v2: clip(cfrc_ext, -1, 1) of 14 bodies — zeros (oracle/spatial_env.py::obs_extras)
    E3W_FOR(i, (nb + 1) * 6) put(at + i, 0.0);
    return;
  }
  double com[3];
  e3w_com(S, m, com);
  E3W_FOR(b1, nb + 1) {            // cinert: row 0 is the world body
    const int base = at + 10 * b1, b = b1 - 1;
    if (b1 == 0 || !m.body_first[b]) { for (int i = 0; i < 10; ++i) put(base + i, 0.0); continue; }   // welded body: mass sits in the link above
    const int l = m.body_link[b], k = E3WOff::KIN + 27 * l;
    double R[9], o[3], rc[3], d[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = S[k + i];
    e3w_ld3(S, k + 9, o);
    e3_matvec(R, m.com[l], rc);
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = o[i] + rc[i] - com[i];
    const double* I6 = m.inertia[l];
    const double Im[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    double T[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]}, Iw[9];
    e3_mat3mul(R, Im, T); e3_mat3mul(T, Rt, Iw);
    const double ml = m.mass[l], d2 = e3_dot(d, d);
    put(base, Iw[0] + ml * (d2 - d[0] * d[0])); put(base + 1, Iw[4] + ml * (d2 - d[1] * d[1])); put(base + 2, Iw[8] + ml * (d2 - d[2] * d[2]));
    put(base + 3, Iw[1] - ml * d[0] * d[1]); put(base + 4, Iw[2] - ml * d[0] * d[2]); put(base + 5, Iw[5] - ml * d[1] * d[2]);
    put(base + 6, ml * d[0]); put(base + 7, ml * d[1]); put(base + 8, ml * d[2]); put(base + 9, ml);
  }
  at += 10 * (nb + 1);
  E3W_FOR(b1, nb + 1) {            // cvel
    const int base = at + 6 * b1;
    if (b1 == 0) { for (int i = 0; i < 6; ++i) put(base + i, 0.0); continue; }
    const int k = E3WOff::KIN + 27 * m.body_link[b1 - 1];
    double o[3], w[3], vo[3], rel[3], t[3];
    e3w_ld3(S, k + 9, o); e3w_ld3(S, k + 12, w); e3w_ld3(S, k + 15, vo);
#pragma unroll
    for (int i = 0; i < 3; ++i) rel[i] = com[i] - o[i];
    e3_cross(w, rel, t);
    put(base, w[0]); put(base + 1, w[1]); put(base + 2, w[2]);
    put(base + 3, vo[0] + t[0]); put(base + 4, vo[1] + t[1]); put(base + 5, vo[2] + t[2]);
  }
  at += 6 * (nb + 1);
  E3W_FOR(i, nv) {                 // qfrc_actuator
    const int l = i - 6 + 1, ka = (i >= 6) ? m.link_act[l] : -1;
    put(at + i, ka >= 0 ? m.gear[l] * S[E3WOff::CTRL + ka] : 0.0);
  }
  at += nv;
  E3W_FOR(i, (nb + 1) * 6) put(at + i, 0.0);   // cfrc_ext
}
