// env2d_group.h — the planar articulated-body stepper with SIXTEEN LANES PER ENV (Hopper, Walker2d, HalfCheetah).
//
// The first form (k_env_step in ilsx_env.hip, kept for A/B runs: ILSX_ENV2D_LANE=1) gives every env ONE lane: 4096 envs are 64
// wavefronts on a chip with 1024 SIMDs, and a vec step lasts as long as one lane needs for 16 dynamics evaluations of dependent fp64
// arithmetic (0.9 ms: rocprofv3 — 165k instructions per wave per env step, half of the wave's cycles spent waiting on the previous
// instruction's result), whatever the number of envs.  Here a DPP row of 16 lanes shares one env (4 envs per wavefront, one wavefront
// per workgroup): lane l is degree of freedom l (0 = x, 1 = z, 2 + b = the hinge of body b), body l - 2, constraint row l and
// contact candidate l at once, so every phase of oracle/planar_env.py::dynamics runs over its natural index in parallel:
//
//   kinematics, COM, inertial forces        one body per lane; ancestors' terms summed root-to-leaf under the ancestor bit mask
//   mass matrix + right-hand side           one ROW per lane (bodies in ascending order: the oracle's summation order)
//   Cholesky M = L L^T                      lane i owns row i; column step k: pivot from lane k, rank-1 update of the trailing rows
//   L y = rhs, L^T x = y                    lane i keeps x_i; the value finished at step k travels by ds_swizzle (a row broadcast)
//   contact / limit rows                    one candidate per lane, row numbers by ballot + population count in the oracle's order
//   z_r = L^-1 j_r, right-hand sides        one row per lane (L read back from the env's LDS blackboard)
//   A = Z Z^T                               one row of A per lane
//   projected Gauss-Seidel                  lane t = row t with its residual, bounds and row t of A in registers; the row whose turn it
//                                           is publishes its new force with one row broadcast, every lane folds the change into its
//                                           residual with one multiply-add (the oracle recomputes the residual from scratch: same
//                                           fixed point, rounding-level differences, 1e-9 against the oracle over chained steps)
//   q.. = qacc0 + L^-T (Z^T f)              one degree of freedom per lane
//
// Hand-offs between lanes go through the env's LDS blackboard (one wavefront: LDS operations execute in order, a hand-off is a
// compiler fence) or, on the latency-critical chains (substitutions, Gauss-Seidel), through ds_swizzle row broadcasts.  Control flow
// is wave-uniform; an env without active rows in a wavefront that has some computes zero forces.
// Included by ilsx_env.hip after PlanarModelDev / EnvStepArgs / impedance_d / env_uniform.
#pragma once

#define EG_LANES 16
#define EG_ENVS 4    // per wavefront == per workgroup
#define EG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// ILSX_EG_PROFILE (measurement build: make VAR=egprof VARFLAGS=-DILSX_EG_PROFILE, tools/env2d_phases.py): clock ticks per stage of
// eg_dynamics, summed over the evaluations of workgroup 0's wavefront into g_eg_prof (read back with ilsx_debug_eg_prof)
#ifdef ILSX_EG_PROFILE
__device__ unsigned long long g_eg_prof[16];
// the stage sums stay in registers (eg_acc, owned by the step function) and reach memory once, at the end of the step
#define EG_PROF_PARAM , unsigned long long (&eg_acc)[16]
#define EG_PROF_ARG , eg_acc
#define EG_PROF_BEGIN() unsigned long long eg_pl = __builtin_amdgcn_s_memtime()
#define EG_PROF(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    eg_acc[i] += t_ - eg_pl; eg_pl = t_; asm volatile("" ::: "memory"); } while (0)
#else
#define EG_PROF_PARAM
#define EG_PROF_ARG
#define EG_PROF_BEGIN() ((void)0)
#define EG_PROF(i) ((void)0)
#endif

// value of lane K of this lane's 16-lane row: DPP row_newbcast (gfx90a+: the one DPP control 64-bit operations accept) — a VALU move, or
// folded into the consuming instruction; the first version went through the LDS crossbar (ds_swizzle bit mode, lane' = (lane & 0x10) | K:
// an LDS round trip on every step of the substitution and Gauss-Seidel chains).  ILSX_EG_SWIZZLE selects that form (measurement).
template <int K> __device__ __forceinline__ double eg_bcast(double x) {
#ifdef ILSX_EG_SWIZZLE
  constexpr int pat = 0x10 | (K << 5);
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), pat);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), pat);
#else
  constexpr int ctl = 0x150 + K;   // DPP_ROW_NEWBCAST_FIRST + K
  // `old` (what a lane keeps when its source lane is disabled) is the value itself: every lane is active at every call site, and a
  // constant there costs two moves per broadcast to materialise
  const int xl = __double2loint(x), xh = __double2hiint(x);
  const int lo = __builtin_amdgcn_update_dpp(xl, xl, ctl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(xh, xh, ctl, 0xf, 0xf, false);
#endif
  return __hiloint2double(hi, lo);
}

// the same with the source lane as a (compile-time after unrolling) value: the swizzle pattern is an instruction immediate
__device__ __forceinline__ double eg_bcast_k(double x, int k) {
  switch (k) {
    case 0: return eg_bcast<0>(x);   case 1: return eg_bcast<1>(x);   case 2: return eg_bcast<2>(x);   case 3: return eg_bcast<3>(x);
    case 4: return eg_bcast<4>(x);   case 5: return eg_bcast<5>(x);   case 6: return eg_bcast<6>(x);   case 7: return eg_bcast<7>(x);
    case 8: return eg_bcast<8>(x);   case 9: return eg_bcast<9>(x);   case 10: return eg_bcast<10>(x); case 11: return eg_bcast<11>(x);
    case 12: return eg_bcast<12>(x); case 13: return eg_bcast<13>(x); case 14: return eg_bcast<14>(x); default: return eg_bcast<15>(x);
  }
}

// One Gauss-Seidel sweep over the rows 0 .. nrmax - 1 (nrmax wave-uniform), rows as TEMPLATE recursion: as `#pragma unroll` over t with a
// `break` the loop stayed rolled for MR = 12 / 16 (the compiler's "loop not unrolled" warning on Walker2d / HalfCheetah: every row step then
// indexed fcopy / Arow through s_set_gpr_idx and picked its row broadcast in a 16-way switch — 220 clock ticks per row step against 40-75
// for Hopper's MR = 8, where it did unroll; tools/env2d_phases.py).  Same operations in the same order.
template <int T, int MR>
__device__ __forceinline__ void eg_pgs_sweep(double (&fcopy)[MR], const double (&Arow)[MR], double& res, double att, double invden, double lo_a,
                                             double hi_a, double hi_b, int nrmax) {
  if constexpr (T < MR) {
    if (T >= nrmax) return;   // wave-uniform
    double fi = (res + att * fcopy[T]) * invden;
    const double fprev = fcopy[T > 0 ? T - 1 : 0];
    fi = fmin(fmax(fi, lo_a * fprev), hi_a * fprev + hi_b);
    const double fb = eg_bcast<T>(fi);
    const double dl = fb - fcopy[T];
    fcopy[T] = fb;
    res -= Arow[T] * dl;
    eg_pgs_sweep<T + 1, MR>(fcopy, Arow, res, att, invden, lo_a, hi_a, hi_b, nrmax);
  }
}
// w = sum_r z_r[l] f_r over the active rows (the same early exit, the same way)
template <int R, int MR, int N>
__device__ __forceinline__ void eg_ztf(const double* Zl, const double (&fcopy)[MR], bool dof, int nr, int nrmax, double& w) {
  if constexpr (R < MR) {
    if (R >= nrmax) return;
    const double z = (dof && R < nr) ? Zl[R * N] : 0.0;
    w += z * fcopy[R];
    eg_ztf<R + 1, MR, N>(Zl, fcopy, dof, nr, nrmax, w);
  }
}

template <int NB, int MR>
struct EgOff {   // the env's LDS blackboard, in doubles
  static constexpr int N = NB + 2;
  static constexpr int BODY_F = 12;                       // c s phid wx wz tax taz ox oz cx cz (+ 1 spare) ; fx fz live in FRC
  static constexpr int QV = 0;                            // q[N] v[N]
  static constexpr int BODY = QV + 2 * N;                 // [NB][BODY_F]
  static constexpr int FRC = BODY + NB * BODY_F;          // [NB][2]
  static constexpr int LMAT = FRC + 2 * NB;               // [N][N] lower triangle of L, row-major
  static constexpr int INVD = LMAT + N * N;               // [N]
  static constexpr int QACC0 = INVD + N;                  // [N]
  static constexpr int ROWJ = QACC0 + N;                  // [MR][N]   J rows, overwritten by z_r
  static constexpr int ROWS = ROWJ + MR * N;              // [MR][4]   residual r, kind, friction, impedance d
  static constexpr int TOTAL = ROWS + MR * 4;
};

// What lane l reads from the model in every evaluation and what depends on nothing but l: read ONCE per step.  A wavefront fence makes the
// compiler re-read whatever came from memory, the model's LDS copy included, so inside eg_dynamics these were LDS round trips at the head
// of dependent chains (ancestor mask -> sums; parent -> the parent's frame; geom -> its body -> the body's frame) and, for the constraint
// rows' damping / stiffness, two fp64 divisions of model constants per evaluation.  Values, not loads; the arithmetic is unchanged.
struct EgLaneK {
  unsigned am, amg;
  int parent, gb, max_rows, pgs_iters;
  bool is_cand, limited;
  double ax, az, cmx, cmz, mb, grav, jsl, arm, damp, stiff;
  double rad, ex, ez, gfric, margin, lo, hi;
  double bd_c, ks_c, bd_l, ks_l;
};
__device__ __forceinline__ void eg_lane_consts(const PlanarModelDev& m, int l, int N, EgLaneK& K) {
  const int jb = l >= 2 && l < N ? l - 2 : 0;
  K.am = (unsigned)m.ancmask[jb]; K.parent = m.parent[jb];
  K.ax = m.anchor[jb][0]; K.az = m.anchor[jb][1]; K.cmx = m.com[jb][0]; K.cmz = m.com[jb][1];
  K.mb = m.mass[jb]; K.grav = m.gravity; K.jsl = m.jsign[jb];
  K.arm = m.armature[jb]; K.damp = m.damping[jb]; K.stiff = m.stiffness[jb];
  K.is_cand = l < 2 * m.ng;
  const int gi = K.is_cand ? m.ng - 1 - (l >> 1) : 0;
  K.gb = m.geom_body[gi]; K.amg = (unsigned)m.ancmask[K.gb];
  K.rad = m.grad[gi];
  K.ex = (l & 1) == 0 ? m.gp1[gi][0] : m.gp2[gi][0]; K.ez = (l & 1) == 0 ? m.gp1[gi][1] : m.gp2[gi][1];
  K.gfric = m.gfric[gi]; K.margin = m.margin; K.max_rows = m.max_rows; K.pgs_iters = m.pgs_iters;
  K.limited = m.limited[jb]; K.lo = m.range[jb][0]; K.hi = m.range[jb][1];
  {
    const double tc = m.c_solref[0], dr = m.c_solref[1], dmax = m.c_solimp[1];
    K.bd_c = 2.0 / (dmax * tc); K.ks_c = 1.0 / (dmax * dmax * tc * tc * dr * dr);
  }
  {
    const double tc = m.l_solref[0], dr = m.l_solref[1], dmax = m.l_solimp[1];
    K.bd_l = 2.0 / (dmax * tc); K.ks_l = 1.0 / (dmax * dmax * tc * tc * dr * dr);
  }
}

// Forward dynamics for the env this 16-lane row serves.  In: this lane's q_l, v_l (0 beyond N), `torque` = gear * ctrl of the actuator
// on this lane's hinge (0 if none).  Out: this lane's acceleration.  E = the env's blackboard.
template <int NB, int MR>
__device__ __forceinline__ double eg_dynamics(const PlanarModelDev& m, double q, double v, double torque, double* E, int l, int grp, const EgLaneK& K EG_PROF_PARAM) {
  constexpr int N = NB + 2;
  using O = EgOff<NB, MR>;
  const bool dof = l < N;
  const int jb = l >= 2 && l < N ? l - 2 : 0;        // the body on this lane (lanes 0, 1 and >= N shadow body 0; they publish nothing)
  const bool body = l >= 2 && l < N;
  EG_PROF_BEGIN();
  // ---- q, v on the blackboard
  if (dof) { E[O::QV + l] = q; E[O::QV + N + l] = v; }
  EG_SYNC();
  // ---- kinematics of body jb: angle and rate = the ancestors' hinge terms, root to leaf
  constexpr bool UK = NB <= 4;   // the lane constants in registers (Hopper); the 7-body models keep reading the model: their step must fit 256 registers
  const unsigned am = UK ? K.am : (unsigned)m.ancmask[jb];
  double phi = 0.0, phid = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double jsj = m.jsign[j];   // read whatever the ancestor bit says: no branch around the load
    const double sg = ((am >> j) & 1u) ? jsj : 0.0;
    phi += sg * E[O::QV + 2 + j];
    phid += sg * E[O::QV + N + 2 + j];
  }
  double sn, cs;
  sincos(phi, &sn, &cs);
  double* Bm = E + O::BODY + jb * O::BODY_F;
  if (body) { Bm[0] = cs; Bm[1] = sn; Bm[2] = phid; }
  EG_SYNC();
  {   // hinge offset from the parent's origin (world frame) and what the origin's acceleration loses there; zero for the root
    const int p = UK ? K.parent : m.parent[jb];
    const double* Pm = E + O::BODY + (p < 0 ? 0 : p) * O::BODY_F;
    const double pc = Pm[0], ps = Pm[1], pphid = Pm[2];
    const double anx = UK ? K.ax : m.anchor[jb][0], anz = UK ? K.az : m.anchor[jb][1];
    double wx = pc * anx - ps * anz, wz = ps * anx + pc * anz;
    double tax = pphid * pphid * wx, taz = pphid * pphid * wz;
    if (p < 0) { wx = 0.0; wz = 0.0; tax = 0.0; taz = 0.0; }
    if (body) { Bm[3] = wx; Bm[4] = wz; Bm[5] = tax; Bm[6] = taz; }
  }
  EG_SYNC();
  double ox = E[O::QV + 0], oz = E[O::QV + 1], aox = 0.0, aoz = 0.0;
  {   // every body's terms are read (no lane-dependent branch around the LDS loads: one round trip for all of them), the ancestors' are kept
    double t3[NB], t4[NB], t5[NB], t6[NB];
#pragma unroll
    for (int j = 1; j < NB; ++j) {
      const double* Jm = E + O::BODY + j * O::BODY_F;
      t3[j] = Jm[3]; t4[j] = Jm[4]; t5[j] = Jm[5]; t6[j] = Jm[6];
    }
#pragma unroll
    for (int j = 1; j < NB; ++j) {
      const bool anc = (am >> j) & 1u;
      ox = anc ? ox + t3[j] : ox; oz = anc ? oz + t4[j] : oz; aox = anc ? aox - t5[j] : aox; aoz = anc ? aoz - t6[j] : aoz;
    }
  }
  {
    const double cmx = UK ? K.cmx : m.com[jb][0], cmz = UK ? K.cmz : m.com[jb][1];
    const double cwx = cs * cmx - sn * cmz, cwz = sn * cmx + cs * cmz;
    const double cx = ox + cwx, cz = oz + cwz;
    const double acx = aox - phid * phid * cwx, acz = aoz - phid * phid * cwz;
    const double mb = UK ? K.mb : m.mass[jb];
    if (body) {
      Bm[7] = ox; Bm[8] = oz; Bm[9] = cx; Bm[10] = cz;
      E[O::FRC + 2 * jb] = mb * (0.0 - acx); E[O::FRC + 2 * jb + 1] = mb * (-(UK ? K.grav : m.gravity) - acz);
    }
  }
  EG_SYNC();
  EG_PROF(0);
  // ---- mass matrix row l and right-hand side l: bodies in ascending order.  Branch-free: as nested ifs (this lane's dof kind, k <= l, the
  // ancestor bits) the block compiled to ~120 EXEC / scalar branches for Walker2d, most of them around ONE LDS load each — 119 serialised LDS
  // round trips, 11.5k clock ticks of a 46k-tick evaluation (tools/env2d_phases.py).  Here every operand is requested up front and lanes / dofs
  // that do not take part are masked by selects; the expressions and their order are the old ones (same bits).
  double Mrow[N];
#pragma unroll
  for (int k = 0; k < N; ++k) Mrow[k] = 0.0;
  double rhs = 0.0;
  {
    double hox[N], hoz[N], hsg[N];   // dof k >= 2: its hinge's origin and sign
#pragma unroll
    for (int k = 2; k < N; ++k) {
      const double* Km = E + O::BODY + (k - 2) * O::BODY_F;
      hox[k] = Km[7]; hoz[k] = Km[8]; hsg[k] = m.jsign[k - 2];
    }
    const double jsl = UK ? K.jsl : m.jsign[jb];
    double bcx[NB], bcz[NB], bf0[NB], bf1[NB], bm[NB], bi[NB];
    unsigned bam[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double* Cm = E + O::BODY + b * O::BODY_F;
      bcx[b] = Cm[9]; bcz[b] = Cm[10]; bf0[b] = E[O::FRC + 2 * b]; bf1[b] = E[O::FRC + 2 * b + 1];
      bm[b] = m.mass[b]; bi[b] = m.inertia[b]; bam[b] = (unsigned)m.ancmask[b];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double cx = bcx[b], cz = bcz[b];
      const unsigned amb = bam[b];
      // this lane's Jacobian entries of body b
      const double sgl = ((amb >> jb) & 1u) ? jsl : 0.0;
      double jxi = -sgl * (cz - oz), jzi = sgl * (cx - ox), jpi = sgl;
      jxi = l == 0 ? 1.0 : (l == 1 ? 0.0 : jxi);
      jzi = l == 0 ? 0.0 : (l == 1 ? 1.0 : jzi);
      jpi = l < 2 ? 0.0 : jpi;
      const double mb = bm[b], ib = bi[b];
      rhs += jxi * bf0[b] + jzi * bf1[b];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double jxk, jzk, jpk;
        if (k == 0) { jxk = 1.0; jzk = 0.0; jpk = 0.0; }
        else if (k == 1) { jxk = 0.0; jzk = 1.0; jpk = 0.0; }
        else {
          const double sg = ((amb >> (k - 2)) & 1u) ? hsg[k] : 0.0;
          jxk = -sg * (cz - hoz[k]); jzk = sg * (cx - hox[k]); jpk = sg;
        }
        const double add = mb * (jxi * jxk + jzi * jzk) + ib * jpi * jpk;
        Mrow[k] = (k <= l) ? Mrow[k] + add : Mrow[k];
      }
    }
  }
  if (body) {
#pragma unroll
    for (int k = 2; k < N; ++k)
      if (k == l) Mrow[k] += UK ? K.arm : m.armature[jb];
    rhs -= (UK ? K.damp : m.damping[jb]) * v + (UK ? K.stiff : m.stiffness[jb]) * q;
    rhs += torque;
  }
  EG_PROF(1);
  // ---- Cholesky, column steps.  Every lane takes the reciprocal of the broadcast pivot itself (same bits everywhere).
  double invd[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double dk = eg_bcast_k(sqrt(Mrow[k]), k);
    const double ik = 1.0 / dk;
    invd[k] = ik;
    if (l == k) Mrow[k] = dk;
    else if (l > k) Mrow[k] = Mrow[k] * ik;          // L[l][k]
#pragma unroll
    for (int j = k + 1; j < N; ++j) {
      const double ljk = eg_bcast_k(Mrow[k], j);        // L[j][k]
      if (l >= j) Mrow[j] -= Mrow[k] * ljk;
    }
  }
  EG_PROF(2);
  // L and the reciprocal pivots on the blackboard (the constraint rows read them back); L^T entries this lane needs for the back-substitutions
  if (dof) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (k <= l) E[O::LMAT + l * N + k] = Mrow[k];
  }
  EG_SYNC();
  double LT[N];   // LT[k] = L[k][l] for k > l
#pragma unroll
  for (int k = 0; k < N; ++k) LT[k] = (k > l && dof) ? E[O::LMAT + k * N + l] : 0.0;
  EG_PROF(3);
  // ---- qacc0 = M^-1 rhs
  double y = dof ? rhs : 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double t = y * invd[k];
    const double yk = eg_bcast_k(t, k);
    if (l == k) y = t;
    else if (l > k) y -= Mrow[k] * yk;
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const double t = y * invd[k];
    const double xk = eg_bcast_k(t, k);
    if (l == k) y = t;
    else if (l < k) y -= LT[k] * xk;
  }
  const double qacc0 = dof ? y : 0.0;
  if (dof) E[O::QACC0 + l] = qacc0;
  EG_PROF(4);
  // ---- constraint rows.  Contacts: candidate c = the capsule end points in the oracle's order (distal geoms first, p1 then p2).
  int ncon = 0, nr = 0;
  {
    const bool is_cand = UK ? K.is_cand : l < 2 * m.ng;
    const int gi = is_cand ? m.ng - 1 - (l >> 1) : 0;   // (unused when the lane constants are)
    const int gb = UK ? K.gb : m.geom_body[gi];
    const int max_rows = UK ? K.max_rows : m.max_rows;
    const double* Gm = E + O::BODY + gb * O::BODY_F;
    const double bc = Gm[0], bs = Gm[1], box = Gm[7], boz = Gm[8], rad = UK ? K.rad : m.grad[gi];
    const double ex = UK ? K.ex : ((l & 1) == 0 ? m.gp1[gi][0] : m.gp2[gi][0]), ez = UK ? K.ez : ((l & 1) == 0 ? m.gp1[gi][1] : m.gp2[gi][1]);
    const double wx = bc * ex - bs * ez, wz = bs * ex + bc * ez;
    const double dist = boz + wz - rad;
    const bool active = is_cand && dist < (UK ? K.margin : m.margin);
    const unsigned bal = (unsigned)((__ballot(active) >> (16 * grp)) & 0xFFFFull);
    const int rank = __popc(bal & ((1u << l) - 1u));
    const bool accept = active && 2 * rank + 2 <= max_rows;
    ncon = min(__popc(bal), max_rows / 2);
    if (accept) {
      const double px = box + wx, pz = boz + wz - (rad + 0.5 * dist);
      double* Jn = E + O::ROWJ + (2 * rank) * N;
      double* Jt = Jn + N;
      Jn[0] = 0.0; Jn[1] = 1.0; Jt[0] = 1.0; Jt[1] = 0.0;
      const unsigned amg = UK ? K.amg : (unsigned)m.ancmask[gb];
      {   // every body's origin and hinge sign requested first (no branch on the ancestor bit around a load: one round trip, not NB)
        double jo7[NB], jo8[NB], jsg[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const double* Jm = E + O::BODY + j * O::BODY_F;
          jo7[j] = Jm[7]; jo8[j] = Jm[8]; jsg[j] = m.jsign[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const double sg = ((amg >> j) & 1u) ? jsg[j] : 0.0;
          Jn[2 + j] = sg * (px - jo7[j]); Jt[2 + j] = -sg * (pz - jo8[j]);
        }
      }
      const double d = impedance_d(fabs(dist), m.c_solimp);
      double* Sn = E + O::ROWS + (2 * rank) * 4;
      const double gfr = UK ? K.gfric : m.gfric[gi];
      Sn[0] = dist; Sn[1] = 0.0; Sn[2] = gfr; Sn[3] = d;
      Sn[4] = 0.0; Sn[5] = 1.0; Sn[6] = gfr; Sn[7] = d;
    }
    // joint limits, bodies in ascending order
    double r = 0.0, sgn = 0.0;
    if (body && (UK ? K.limited : (bool)m.limited[jb])) {
      const double lo = UK ? K.lo : m.range[jb][0], hi = UK ? K.hi : m.range[jb][1];
      if (q - lo < 0.0) { r = q - lo; sgn = 1.0; }
      else if (hi - q < 0.0) { r = hi - q; sgn = -1.0; }
    }
    const bool lact = sgn != 0.0;
    const unsigned lbal = (unsigned)((__ballot(lact) >> (16 * grp)) & 0xFFFFull);
    const int lrank = __popc(lbal & ((1u << l) - 1u));
    const int row = 2 * ncon + lrank;
    if (lact && row + 1 <= max_rows) {
      double* Jl = E + O::ROWJ + row * N;
#pragma unroll
      for (int i = 0; i < N; ++i) Jl[i] = (i == l) ? sgn : 0.0;
      double* Sl = E + O::ROWS + row * 4;
      Sl[0] = r; Sl[1] = 2.0; Sl[2] = 0.0; Sl[3] = impedance_d(fabs(r), m.l_solimp);
    }
    nr = 2 * ncon + min(__popc(lbal), max_rows - 2 * ncon);
  }
  // rows of the busiest env of this wavefront (wave-uniform trip counts below)
  int nrmax = max(max(__builtin_amdgcn_readlane(nr, 0), __builtin_amdgcn_readlane(nr, 16)),
                  max(__builtin_amdgcn_readlane(nr, 32), __builtin_amdgcn_readlane(nr, 48)));
  EG_PROF(5);
  if (nrmax == 0) return qacc0;
  EG_SYNC();
  const bool rowl = l < nr && l < MR;
  double x[N], rres = 0.0, rmu = 0.0, rd = 1.0;
  int rkind = 0;
  {
    const double* Jr = E + O::ROWJ + (l < MR ? l : 0) * N;
    const double* Sr = E + O::ROWS + (l < MR ? l : 0) * 4;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = rowl ? Jr[i] : 0.0;
    if (rowl) { rres = Sr[0]; rkind = (int)Sr[1]; rmu = Sr[2]; rd = Sr[3]; }
  }
  double rhs_c = 0.0;
  {
    double jv = 0.0, ja = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { jv += x[i] * E[O::QV + N + i]; ja += x[i] * E[O::QACC0 + i]; }
    // z_r = L^-1 j_r (forward substitution only)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double sum = x[i];
#pragma unroll
      for (int t = 0; t < i; ++t) sum -= E[O::LMAT + i * N + t] * x[t];
      x[i] = sum * invd[i];
    }
    double bdamp, kstiff;   // 2 / (dmax tc), 1 / (dmax^2 tc^2 dr^2) of the row kind's solref / solimp
    if (UK) { bdamp = rkind == 2 ? K.bd_l : K.bd_c; kstiff = rkind == 2 ? K.ks_l : K.ks_c; }
    else {
      const double* sref = rkind == 2 ? m.l_solref : m.c_solref;
      const double dmax = rkind == 2 ? m.l_solimp[1] : m.c_solimp[1];
      const double tc = sref[0], dr = sref[1];
      bdamp = 2.0 / (dmax * tc); kstiff = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    }
    if (rowl) rhs_c = (-bdamp * jv - kstiff * rd * rres) - ja;
  }
  EG_PROF(6);
  EG_SYNC();   // every lane has read its J row and L: the rows may be overwritten
  if (rowl) {
    double* Zr = E + O::ROWJ + l * N;
#pragma unroll
    for (int i = 0; i < N; ++i) Zr[i] = x[i];
  }
  EG_SYNC();
  // ---- row l of A = Z Z^T
  // CB rows of Z per LDS round trip (row by row, each behind its own wave-uniform test, the block was MR serialised round trips); rows of a
  // batch beyond nrmax read stale blackboard words and are masked like every row beyond this env's nr.  CB = 4 for Hopper, 2 for the 7-body
  // models: their step must stay within 256 registers (two wavefronts per SIMD from 8192 envs on)
  constexpr int CB = N > 6 ? 2 : 4;
  double Arow[MR];
#pragma unroll
  for (int c0 = 0; c0 < MR; c0 += CB) {
    double sum[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) sum[u] = 0.0;
    if (c0 < nrmax) {   // wave-uniform
      double zc[CB][N];
#pragma unroll
      for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int i = 0; i < N; ++i) zc[u][i] = (c0 + u < MR) ? E[O::ROWJ + (c0 + u) * N + i] : 0.0;
#pragma unroll
      for (int u = 0; u < CB; ++u)
#pragma unroll
        for (int i = 0; i < N; ++i) sum[u] += x[i] * zc[u][i];
    }
#pragma unroll
    for (int u = 0; u < CB; ++u)
      if (c0 + u < MR) Arow[c0 + u] = (rowl && c0 + u < nr) ? sum[u] : 0.0;
  }
  double att = 0.0;
#pragma unroll
  for (int c = 0; c < MR; ++c)
    if (c == l) att = Arow[c];
  const double invden = rowl ? 1.0 / (att + (1.0 - rd) / rd * att) : 0.0;
  EG_PROF(7);
  // ---- projected Gauss-Seidel: lane t is row t
  double fcopy[MR];
#pragma unroll
  for (int c = 0; c < MR; ++c) fcopy[c] = 0.0;
  double res = rhs_c;
  // the projection of a row as ONE clamp with per-lane coefficients, f <- min(max(f, lo_a * f_prev), hi_a * f_prev + hi_b): friction rows
  // (lo_a, hi_a, hi_b) = (-mu, mu, 0), normal / limit rows (0, 0, +inf) — the same values as the two-branch form (a select per bound and
  // per lane), five instructions shorter per row and sweep on the chain that is most of this kernel
  const double lo_a = rkind == 1 ? -rmu : 0.0, hi_a = rkind == 1 ? rmu : 0.0, hi_b = rkind == 1 ? 0.0 : __builtin_huge_val();
  for (int it = 0, n_it = UK ? K.pgs_iters : m.pgs_iters; it < n_it; ++it) eg_pgs_sweep<0, MR>(fcopy, Arow, res, att, invden, lo_a, hi_a, hi_b, nrmax);
  EG_PROF(8);
  // ---- q.. = qacc0 + L^-T (Z^T f)
  double w = 0.0;
  eg_ztf<0, MR, N>(E + O::ROWJ + l, fcopy, dof, nr, nrmax, w);
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const double t = w * invd[k];
    const double xk = eg_bcast_k(t, k);
    if (l == k) w = t;
    else if (l < k) w -= LT[k] * xk;
  }
  EG_SYNC();   // the rows are rewritten by the next evaluation
  EG_PROF(9);
  return dof ? qacc0 + w : 0.0;
}

// One vec-env step; the arguments, the reward / termination / record / auto-reset rules are those of k_env_step.
template <int NB, int MR>
__device__ __forceinline__ void envg_step_dev(const EnvStepArgs& A) {
  constexpr int N = NB + 2;
  using O = EgOff<NB, MR>;
  extern __shared__ __attribute__((aligned(16))) double smd_all[];
  constexpr int MODEL_DOUBLES = (sizeof(PlanarModelDev) + 7) / 8;
  {
    const double* src = reinterpret_cast<const double*>(A.m);
    for (int i = threadIdx.x; i < MODEL_DOUBLES; i += 64) smd_all[i] = src[i];
    EG_SYNC();
  }
  const PlanarModelDev& m = *reinterpret_cast<const PlanarModelDev*>(smd_all);
#ifdef ILSX_EG_PROFILE
  const unsigned long long eg_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long eg_acc[16];
  for (int i = 0; i < 16; ++i) eg_acc[i] = 0;
#endif
  const int lane = threadIdx.x, l = lane & 15, grp = lane >> 4;
  double* E = smd_all + MODEL_DOUBLES + grp * O::TOTAL;
  const int t_raw = blockIdx.x * EG_ENVS + grp;
  const int t = t_raw < A.n_ids ? t_raw : A.n_ids - 1;     // surplus rows of the last wavefront shadow the last env and store nothing
  const int env = A.ids ? A.ids[t] : t;
  const bool live = t_raw < A.n_ids && !(A.frozen && A.frozen[env]);
  const bool dof = l < N;
  const int o = m.obs_dim, na = m.n_act;
  double q = dof ? A.qpos[(size_t)l * A.n_env + env] : 0.0, v = dof ? A.qvel[(size_t)l * A.n_env + env] : 0.0;
  // this lane's actuator torque (gear * clipped action) and the env's control cost
  double torque = 0.0, ctrl_sq = 0.0;
  for (int k = 0; k < na; ++k) {   // NormalizedBoxEnv: ctrlrange [-1, 1], clip (wrappers.py:343-346)
    const double a = fmin(fmax((double)A.act[(size_t)t * na + k], -1.0), 1.0);
    ctrl_sq += a * a;
    if (m.act_body[k] + 2 == l) torque += m.gear[l - 2] * a;
  }
  // observation pieces of this lane: element l - 1 (qpos[1:]) and N - 1 + l (qvel)
  auto obs_q = [&](double qq) { return (float)((qq - m.obs_shift[l - 1]) * m.obs_inv_scale[l - 1]); };
  auto obs_v = [&](double vv) {
    return (float)(((m.qvel_clip > 0.0 ? fmin(fmax(vv, -m.qvel_clip), m.qvel_clip) : vv) - m.obs_shift[N - 1 + l]) * m.obs_inv_scale[N - 1 + l]);
  };
  const float obq0 = (l >= 1 && dof) ? obs_q(q) : 0.0f, obv0 = dof ? obs_v(v) : 0.0f;
  const double x0 = eg_bcast<0>(q);
  EgLaneK K;
  if (NB <= 4) eg_lane_consts(m, l, N, K);
#ifdef ILSX_EG_PROFILE
  { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); eg_acc[10] += __builtin_amdgcn_s_memtime() - eg_t0; }
  const unsigned long long eg_t1 = __builtin_amdgcn_s_memtime();
#endif
  const double h = m.timestep;
  for (int s = 0; s < m.frame_skip; ++s) {   // RK4, the oracle's accumulation order (env_substep)
    double qs = q, vs = v, qsum = 0.0, vsum = 0.0;
#pragma unroll 1
    for (int stage = 0; stage < 4; ++stage) {
      const double a = eg_dynamics<NB, MR>(m, qs, vs, torque, E, l, grp, K EG_PROF_ARG);
      const double w = (stage == 1 || stage == 2) ? 2.0 : 1.0;
      const double ch = (stage == 2) ? h : 0.5 * h;
      qsum = stage == 0 ? vs : qsum + w * vs;
      vsum = stage == 0 ? a : vsum + w * a;
      const double vn = v + ch * a;
      qs = q + ch * vs;
      vs = vn;
    }
    q = q + h / 6.0 * qsum;
    v = v + h / 6.0 * vsum;
  }
#ifdef ILSX_EG_PROFILE
  const unsigned long long eg_t2 = __builtin_amdgcn_s_memtime();
  eg_acc[11] += eg_t2 - eg_t1;   // the 16 evaluations and the integrator around them
#endif
  const double dt = m.timestep * m.frame_skip;
  const double xq = eg_bcast<0>(q), zq = eg_bcast<1>(q), aq = eg_bcast<2>(q);
  const double reward = (xq - x0) / dt + m.alive - m.ctrl_cost * ctrl_sq;
  // termination (hopper.py:19-25 / walker2d.py:17-20 / never)
  bool okl = true;
  if (m.task == 0) {
    if (dof) {
      okl = isfinite(q) && isfinite(v) && fabs(v) < m.state_max;
      if (l >= 2) okl = okl && fabs(q) < m.state_max;
    }
  }
  const unsigned okbal = (unsigned)((__ballot(okl) >> (16 * grp)) & 0xFFFFull);
  bool ok;
  if (m.task == 0) ok = okbal == 0xFFFFu && zq > m.z_min && fabs(aq) < m.ang_max;
  else if (m.task == 1) ok = zq > m.z_min && zq < m.z_max && aq > -m.ang_max && aq < m.ang_max;
  else ok = true;
  const bool done = !ok;
  const bool finl = !dof || (isfinite(q) && isfinite(v));
  const bool finite = (unsigned)((__ballot(finl) >> (16 * grp)) & 0xFFFFull) == 0xFFFFu;
  float obq1 = (l >= 1 && dof) ? obs_q(q) : 0.0f, obv1 = dof ? obs_v(v) : 0.0f;
  if (live) {
    if (A.obs) {
      if (l >= 1 && dof) A.obs[(size_t)t * o + l - 1] = obq1;
      if (dof) A.obs[(size_t)t * o + N - 1 + l] = obv1;
    }
    if (l == 0) {
      if (A.rew) A.rew[t] = (float)reward;
      if (A.done) A.done[t] = done ? 1 : 0;
    }
    if (A.replay) {   // fused replay insert: one 128-byte-aligned record per transition
      long long slot = A.top + env;
      if (slot >= A.cap) slot -= A.cap;
      float* rec = A.stage ? A.stage + ((size_t)env * A.stage_len + A.ep_len[env]) * A.rec : A.replay + (size_t)slot * A.rec;
      if (l >= 1 && dof) { rec[l - 1] = obq0; rec[o + na + 2 + l - 1] = obq1; }
      if (dof) { rec[N - 1 + l] = obv0; rec[o + na + 2 + N - 1 + l] = obv1; }
      const float* ra = A.rec_act ? A.rec_act : A.act;   // DAgger stores the expert's label, not the executed action
      if (l < na) rec[o + l] = ra[(size_t)t * na + l];
      if (l == 0) {
        rec[o + na] = (float)reward;
        rec[o + na + 1] = (done && !A.no_terminal) ? 1.0f : 0.0f;   // base_algorithm.py:195-196,208-210
        rec[2 * o + na + 2] = 0.0f; rec[2 * o + na + 3] = 0.0f;      // absorbing = [0, 0] (base_algorithm.py:211-213)
      }
    }
  }
  if (A.auto_reset) {
    const int len = A.ep_len[env] + 1;
    const double ret = A.ep_ret[env] + reward;
    const bool end = (done && !A.no_terminal) || len >= A.max_path_length || !finite;
    EG_SYNC();   // every lane has read ep_len / ep_ret
    if (end) {
      if (dof) {   // env_reset_state, one degree of freedom per lane: the same Philox draws
        q = m.init_qpos[l] + m.reset_noise * (2.0 * env_uniform(A.seed, A.stream, A.step, (uint32_t)env, l) - 1.0);
        if (m.reset_noise_vel_std > 0.0) {
          const double u1 = env_uniform(A.seed, A.stream, A.step, (uint32_t)env, N + 2 * l), u2 = env_uniform(A.seed, A.stream, A.step, (uint32_t)env, N + 2 * l + 1);
          v = m.reset_noise_vel_std * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        } else {
          v = m.reset_noise * (2.0 * env_uniform(A.seed, A.stream, A.step, (uint32_t)env, N + l) - 1.0);
        }
        obq1 = l >= 1 ? obs_q(q) : 0.0f; obv1 = obs_v(v);
      }
      if (live && l == 0) { atomicAdd(&A.stats[0], 1.0); atomicAdd(&A.stats[1], ret); }
    }
    if (live && l == 0) {
      A.ep_len[env] = end ? 0 : len;
      A.ep_ret[env] = end ? 0.0 : ret;
      if (A.flush_len) A.flush_len[env] = end ? (len | ((done && !A.no_terminal) ? (1 << 30) : 0)) : 0;
    }
  }
  if (live) {
    if (A.obs_cur) {
      if (l >= 1 && dof) A.obs_cur[(size_t)env * o + l - 1] = obq1;
      if (dof) A.obs_cur[(size_t)env * o + N - 1 + l] = obv1;
    }
    if (dof) { A.qpos[(size_t)l * A.n_env + env] = q; A.qvel[(size_t)l * A.n_env + env] = v; }
  }
#ifdef ILSX_EG_PROFILE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    eg_acc[12] = t3 - eg_t2; eg_acc[15] = t3 - eg_t0; eg_acc[14] = 1;
    for (int i = 0; i < 16; ++i) g_eg_prof[i] += eg_acc[i];
  }
#endif
}

template <int NB, int MR>
// two wavefronts per SIMD must fit (8192 envs and more: 256 registers each, accumulation registers included) — with launch bounds alone the
// compiler budgets a lone wavefront's 512 and the 7-body instances took 284: 591 against 432 us per 8192-env Walker2d step
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_envg_step(const EnvStepArgs A) { envg_step_dev<NB, MR>(A); }

// The same step for several RUNS in one launch (the lock-step rollout of co-resident seeds, ilsx_rollout_steps_lockstep): blockIdx.y = run,
// every run with its own state, model record, actions, ring / staging area, Philox key and step counter — a row of the grid is exactly the
// launch that run would make alone (same arithmetic, same draws).  Ten 4-env runs are ten single-wavefront launches otherwise.
#define ENVG_MAX_RUNS 16
struct EnvStepGroupArgs { EnvStepArgs a[ENVG_MAX_RUNS]; };
static_assert(sizeof(EnvStepGroupArgs) <= 4096, "the runs' records travel in the kernel-argument segment (4 KB)");
template <int NB, int MR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_envg_step_runs(const EnvStepGroupArgs G) { envg_step_dev<NB, MR>(G.a[blockIdx.y]); }
