// ilsx_env.hip — HIP-resident batched planar articulated-body stepper + fused rollout step.
// Replaces the vec-env path of the reference: rlkit/envs/vecenvs.py:158-257 (reset/step protocol),
// rlkit/envs/worker/subproc.py:59-113 (one OS process + pipe per env), rlkit/envs/wrappers.py:342-352
// (NormalizedBoxEnv action map) and the MuJoCo call under gym's HopperEnv/Walker2dEnv.step
// (reward / termination formulas as restated in rlkit/envs/mujoco/hopper.py:11-40, walker2d.py:11-36).
//
// One lane per env.  State is SoA float64 in HBM (qpos[d][env], qvel[d][env]) so a wave's loads are
// coalesced; the per-env matrices (mass matrix / Cholesky factor, constraint Jacobian, M^-1 J^T, the
// constraint-space matrix A) live in LDS as [element][lane] slices, everything else in registers.
// The model and its solver are stated in oracle/planar_env.py (the CPU oracle); physics parity with
// MuJoCo is UNPINNED (no MuJoCo here) — see DESIGN.md.
// Not a roofline kernel: ~160 B of HBM traffic and O(1e4) fp64 FLOP per env-step on a serial chain.
#include <cmath>

#include "host_common.h"
#include "env3d.h"
#include "env3d_wave.h"

#define ENV_MAXB 8
#define ENV_MAXG 8

struct PlanarModelDev {
  int nb, ng, task, frame_skip, pgs_iters, max_rows, n_act, obs_dim;
  int parent[ENV_MAXB], limited[ENV_MAXB], act_body[ENV_MAXB], geom_body[ENV_MAXG], ancmask[ENV_MAXB];
  double anchor[ENV_MAXB][2], com[ENV_MAXB][2], mass[ENV_MAXB], inertia[ENV_MAXB], jsign[ENV_MAXB];
  double armature[ENV_MAXB], damping[ENV_MAXB], range[ENV_MAXB][2], gear[ENV_MAXB], stiffness[ENV_MAXB];
  double gp1[ENV_MAXG][2], gp2[ENV_MAXG][2], grad[ENV_MAXG], gfric[ENV_MAXG];
  double timestep, gravity, reset_noise, margin, reset_noise_vel_std, qvel_clip;
  double c_solref[2], c_solimp[3], l_solref[2], l_solimp[3];
  double ctrl_cost, alive, z_min, z_max, ang_max, state_max;
  double init_qpos[ENV_MAXB + 2];
  // ScaledEnv / MinmaxEnv (wrappers.py:53-203): every observation the env produces is (raw - shift) * inv_scale
  double obs_shift[2 * (ENV_MAXB + 2)], obs_inv_scale[2 * (ENV_MAXB + 2)];
};

struct ilsx_vecenv {
  ilsx_ctx* ctx = nullptr;
  PlanarModelDev hm;
  PlanarModelDev* dm = nullptr;
  int n_env = 0, n = 0, o = 0, a = 0;
  double *qpos = nullptr, *qvel = nullptr;  // [n][n_env]
  float *obs_cur = nullptr, *act = nullptr, *nobs = nullptr, *rew = nullptr;
  uint8_t* done = nullptr;
  int* ep_len = nullptr;
  double* ep_ret = nullptr;
  double* stats = nullptr;  // [0] finished episodes, [1] sum of their returns, [2] env steps
  int* ids = nullptr;       // device scratch for id lists
  uint64_t seed = 0;
  uint32_t rng_stream = 0;
  unsigned long long step_ctr = 0;
  // running observation statistics (vecenvs.py:299-327, normalizer.py:128-152)
  struct ObsRms* rms = nullptr;
  float* obs_n = nullptr;   // [n_env][o] normalised policy input (norm_obs)
  // evaluation rollouts (ilsx_eval_rollout)
  float* act_label = nullptr;   // [n_env][a] expert labels (ilsx_rollout_step_relabel)
  unsigned char* ev_frozen = nullptr; double* ev_ret = nullptr; int* ev_len = nullptr; double* ev_stats = nullptr; int* ev_alive = nullptr;
  bool norm_obs = false, update_rms = false;
  // path mode (ilsx_vecenv_set_path_mode): whole episodes are staged and enter the ring when they end (base_algorithm.py:509-519)
  bool path_mode = false; float* stage = nullptr; int stage_len = 0, stage_rec = 0; int* flush_len = nullptr;
  int* flush_host = nullptr;            // pinned: the per-step read-back of the episode-end flags is a true asynchronous copy
  ilsx_replay* paths_pending = nullptr; // ilsx_rollout_step_begin enqueued a step whose finished episodes are not in the ring yet (ilsx_rollout_step_end)
  // run 0 of a lock-step group (ilsx_rollout_steps_lockstep): ONE array of episode-end flags for all K runs, so that a lock-step reads them back
  // with one copy instead of K (each a serialised 8 us on the stream)
  int* grp_flush_dev = nullptr; int* grp_flush_host = nullptr; int grp_flush_n = 0;
  // 3-D engine (Ant / Humanoid, env3d.h): model, its device copy, and the per-env working set [E3Off::TOTAL][n_env]
  int engine = 0, nq = 0, nv = 0;
  bool wave3 = true;   // wave-per-env kernels (env3d_wave.h); ILSX_ENV3D_LANE=1 selects the lane-per-env form (env3d.h) for A/B runs
  Spatial3Dev* hm3 = nullptr; Spatial3Dev* dm3 = nullptr; double* scr3 = nullptr;
  const float* policy_obs() const { return norm_obs ? obs_n : obs_cur; }
};

// RunningMeanStd (normalizer.py:128-152): float64, one owner workgroup per observation dimension (count is kept per
// dimension so that no workgroup depends on another's update).
#define ENV_MAX_OBS E3_MAXOBS
struct ObsRms { double mean[ENV_MAX_OBS], var[ENV_MAX_OBS], count[ENV_MAX_OBS]; };

// update(x) with x = the selected rows of src[n_rows][o]: batch mean / population variance in two passes (np.mean,
// np.var), then the parallel-variance merge.  flag != null: only rows whose flag is 0 (episodes that just ended).
__global__ __launch_bounds__(256) void k_obs_rms_update(const float* __restrict__ src, int n_rows, int o, const int* __restrict__ flag,
                                                        ObsRms* rms) {
  __shared__ double sh[4];
  __shared__ double bc;
  const int i = blockIdx.x, tid = threadIdx.x;
  auto bsum = [&](double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
  };
  double s = 0.0, n = 0.0;
  for (int r = tid; r < n_rows; r += 256)
    if (!flag || flag[r] == 0) { s += (double)src[(size_t)r * o + i]; n += 1.0; }
  s = bsum(s); n = bsum(n);
  if (n == 0.0) return;   // block-uniform
  const double bm = s / n;
  double ss = 0.0;
  for (int r = tid; r < n_rows; r += 256)
    if (!flag || flag[r] == 0) { const double d = (double)src[(size_t)r * o + i] - bm; ss += d * d; }
  ss = bsum(ss);
  if (tid == 0) {
    const double bv = ss / n, c = rms->count[i], mean = rms->mean[i], var = rms->var[i];
    const double delta = bm - mean, tot = c + n;
    const double m2 = var * c + bv * n + delta * delta * c * n / tot;
    rms->mean[i] = mean + delta * n / tot;
    rms->var[i] = m2 / tot;
    rms->count[i] = tot;
  }
  (void)bc;
}
// normalize_obs (vecenvs.py:299-327): clip((x - mean) / sqrt(var + eps), +-10), eps = np.finfo(float32).eps
__global__ __launch_bounds__(256) void k_obs_normalize(const float* src, float* dst, int n, int o, const ObsRms* __restrict__ rms) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int i = t % o;
  const double z = ((double)src[t] - rms->mean[i]) / sqrt(rms->var[i] + 1.1920928955078125e-07);
  dst[t] = (float)fmin(fmax(z, -10.0), 10.0);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double impedance_d(double r_abs, const double* solimp) {
  const double d0 = solimp[0], dmax = solimp[1], width = solimp[2];
  double x = width > 0.0 ? fmin(r_abs / width, 1.0) : 1.0;
  const double y = x < 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return d0 + y * (dmax - d0);
}

// LDS slice accessor: element i of this lane's scratch array starting at `base`
#define SL(base, i) sm[((base) + (i)) * BLOCK + lane_in_block]

template <int NB, int MR, int BLOCK>
struct EnvScratch {
  static constexpr int N = NB + 2;
  static constexpr int OFF_M = 0;               // N*N   (becomes the Cholesky factor L)
  static constexpr int OFF_J = OFF_M + N * N;   // MR*N  (row r becomes z_r = L^-1 j_r in place)
  static constexpr int OFF_A = OFF_J + MR * N;  // MR*(MR+1)/2: A = J M^-1 J^T = Z Z^T, lower triangle packed
  static constexpr int TOTAL = OFF_A + MR * (MR + 1) / 2;
  static constexpr int a_idx(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
};

// Forward dynamics with soft constraints: qacc = f(q, v, ctrl).  See oracle/planar_env.py::dynamics.
template <int NB, int MR, int BLOCK>
__device__ void env_dynamics(const PlanarModelDev& m, const double (&q)[NB + 2], const double (&v)[NB + 2],
                             const double* ctrl, double (&qacc)[NB + 2], double* sm, int lane_in_block) {
  constexpr int N = NB + 2;
  using S = EnvScratch<NB, MR, BLOCK>;
  double phi[NB], phid[NB], ox[NB], oz[NB], aox[NB], aoz[NB];
  // cos / sin of every body angle, ONCE per evaluation (one sincos each: a shared range reduction): the kinematics of a child, the
  // mass-matrix loop and the contact geometry all use the same values (the first form called cos and sin separately at each of the
  // three places: ~11 pairs per evaluation of the 4-body model, a sixth of its instructions)
  double cb[NB], sb[NB];
  // ---- kinematics
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int p = m.parent[b];
    if (p < 0) {
      phi[b] = m.jsign[b] * q[2]; phid[b] = m.jsign[b] * v[2];
      ox[b] = q[0]; oz[b] = q[1]; aox[b] = 0.0; aoz[b] = 0.0;
    } else {
      // parent index is data: select with a small unrolled scan (keeps the arrays in registers)
      double pphi = 0, pphid = 0, pox = 0, poz = 0, paox = 0, paoz = 0, c = 1.0, s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j == p) { pphi = phi[j]; pphid = phid[j]; pox = ox[j]; poz = oz[j]; paox = aox[j]; paoz = aoz[j]; c = cb[j]; s = sb[j]; }
      phi[b] = pphi + m.jsign[b] * q[2 + b]; phid[b] = pphid + m.jsign[b] * v[2 + b];
      const double wx = c * m.anchor[b][0] - s * m.anchor[b][1], wz = s * m.anchor[b][0] + c * m.anchor[b][1];
      ox[b] = pox + wx; oz[b] = poz + wz;
      aox[b] = paox - pphid * pphid * wx; aoz[b] = paoz - pphid * pphid * wz;
    }
    sincos(phi[b], &sb[b], &cb[b]);
  }
  // ---- mass matrix + right-hand side
#pragma unroll
  for (int i = 0; i < N * N; ++i) SL(S::OFF_M, i) = 0.0;
  double rhs[N];
#pragma unroll
  for (int i = 0; i < N; ++i) rhs[i] = 0.0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const double c = cb[b], s = sb[b];
    const double cwx = c * m.com[b][0] - s * m.com[b][1], cwz = s * m.com[b][0] + c * m.com[b][1];
    const double cx = ox[b] + cwx, cz = oz[b] + cwz;
    const double acx = aox[b] - phid[b] * phid[b] * cwx, acz = aoz[b] - phid[b] * phid[b] * cwz;
    double jx[N], jz[N], jp[N];  // COM Jacobian rows and angular Jacobian
    jx[0] = 1.0; jz[0] = 0.0; jp[0] = 0.0; jx[1] = 0.0; jz[1] = 1.0; jp[1] = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool anc = (m.ancmask[b] >> j) & 1;
      const double sg = anc ? m.jsign[j] : 0.0;
      jx[2 + j] = -sg * (cz - oz[j]); jz[2 + j] = sg * (cx - ox[j]); jp[2 + j] = sg;
    }
    const double mb = m.mass[b], ib = m.inertia[b];
    const double fx = mb * (0.0 - acx), fz = mb * (-m.gravity - acz);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      rhs[i] += jx[i] * fx + jz[i] * fz;
#pragma unroll
      for (int k = 0; k <= i; ++k) SL(S::OFF_M, i * N + k) += mb * (jx[i] * jx[k] + jz[i] * jz[k]) + ib * jp[i] * jp[k];
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    SL(S::OFF_M, (2 + b) * N + 2 + b) += m.armature[b];
    rhs[2 + b] -= m.damping[b] * v[2 + b] + m.stiffness[b] * q[2 + b];   // joint damper + spring towards 0
  }
  for (int k = 0; k < m.n_act; ++k) {
    const int b = m.act_body[k];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j == b) rhs[2 + j] += m.gear[j] * ctrl[k];
  }
  // ---- Cholesky M = L L^T (lower triangle in place).  Every division by a pivot is a multiplication by its reciprocal, taken once
  // (an fp64 division is ~40 instructions on this pipe; the factorisation, the two solves, one forward substitution per constraint
  // row and the final back-substitution held 30 + 6 per row of them per evaluation).  1 ulp away from the oracle's quotients.
  double invd[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k <= i; ++k) {
      double sum = SL(S::OFF_M, i * N + k);
#pragma unroll
      for (int t = 0; t < k; ++t) sum -= SL(S::OFF_M, i * N + t) * SL(S::OFF_M, k * N + t);
      if (i == k) { const double d = sqrt(sum); SL(S::OFF_M, i * N + k) = d; invd[i] = 1.0 / d; }
      else SL(S::OFF_M, i * N + k) = sum * invd[k];
    }
  }
  auto chol_solve = [&](double (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double sum = x[i];
#pragma unroll
      for (int t = 0; t < i; ++t) sum -= SL(S::OFF_M, i * N + t) * x[t];
      x[i] = sum * invd[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double sum = x[i];
#pragma unroll
      for (int t = i + 1; t < N; ++t) sum -= SL(S::OFF_M, t * N + i) * x[t];
      x[i] = sum * invd[i];
    }
  };
  double qacc0[N];
#pragma unroll
  for (int i = 0; i < N; ++i) qacc0[i] = rhs[i];
  chol_solve(qacc0);

  // ---- constraint rows: contacts (distal geoms first, p1 then p2), then joint limits
  int nr = 0;
  double rr[MR], rmu[MR], rd[MR], rb_[MR], rk[MR];
  int rkind[MR];  // 0 normal, 1 tangent, 2 limit
#pragma unroll
  for (int t = 0; t < MR; ++t) { rr[t] = 0.0; rmu[t] = 0.0; rd[t] = 1.0; rb_[t] = 0.0; rk[t] = 0.0; rkind[t] = 0; }
  auto add_row = [&](const double (&jrow)[N], double r, int kind, double mu, double d, double bdamp, double kstiff) {
#pragma unroll
    for (int i = 0; i < N; ++i) SL(S::OFF_J, nr * N + i) = jrow[i];
#pragma unroll
    for (int t = 0; t < MR; ++t)
      if (t == nr) { rr[t] = r; rkind[t] = kind; rmu[t] = mu; rd[t] = d; rb_[t] = bdamp; rk[t] = kstiff; }
    ++nr;
  };
  {
    const double tc = m.c_solref[0], dr = m.c_solref[1], dmax = m.c_solimp[1];
    const double bdamp = 2.0 / (dmax * tc), kstiff = 1.0 / (dmax * dmax * tc * tc * dr * dr);
    for (int gi = m.ng - 1; gi >= 0; --gi) {
      const int b = m.geom_body[gi];
      double box = 0, boz = 0, c = 1.0, s = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j == b) { box = ox[j]; boz = oz[j]; c = cb[j]; s = sb[j]; }
      const double rad = m.grad[gi];
      for (int e = 0; e < 2; ++e) {
        const double ex = e == 0 ? m.gp1[gi][0] : m.gp2[gi][0], ez = e == 0 ? m.gp1[gi][1] : m.gp2[gi][1];
        const double wx = c * ex - s * ez, wz = s * ex + c * ez;
        const double dist = boz + wz - rad;
        if (dist < m.margin && nr + 2 <= m.max_rows) {
          const double px = box + wx, pz = boz + wz - (rad + 0.5 * dist);  // contact point
          double jn[N], jt[N];
          jn[0] = 0.0; jn[1] = 1.0; jt[0] = 1.0; jt[1] = 0.0;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const bool anc = (m.ancmask[b] >> j) & 1;
            const double sg = anc ? m.jsign[j] : 0.0;
            jn[2 + j] = sg * (px - ox[j]); jt[2 + j] = -sg * (pz - oz[j]);
          }
          const double d = impedance_d(fabs(dist), m.c_solimp);
          add_row(jn, dist, 0, m.gfric[gi], d, bdamp, kstiff);
          add_row(jt, 0.0, 1, m.gfric[gi], d, bdamp, kstiff);
        }
      }
    }
  }
  {
    const double tc = m.l_solref[0], dr = m.l_solref[1], dmax = m.l_solimp[1];
    const double bdamp = 2.0 / (dmax * tc), kstiff = 1.0 / (dmax * dmax * tc * tc * dr * dr);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!m.limited[b] || nr + 1 > m.max_rows) continue;
      const double lo = m.range[b][0], hi = m.range[b][1], qq = q[2 + b];
      double r = 0.0, sgn = 0.0;
      if (qq - lo < 0.0) { r = qq - lo; sgn = 1.0; }
      else if (hi - qq < 0.0) { r = hi - qq; sgn = -1.0; }
      if (sgn != 0.0) {
        double je[N];
#pragma unroll
        for (int i = 0; i < N; ++i) je[i] = (i == 2 + b) ? sgn : 0.0;
        add_row(je, r, 2, 0.0, impedance_d(fabs(r), m.l_solimp), bdamp, kstiff);
      }
    }
  }
  if (nr == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) qacc[i] = qacc0[i];
    return;
  }
  // ---- z_r = L^-1 j_r (forward substitution only, in place), A = Z Z^T (symmetric, packed), PGS on (A + R) f = aref - J qacc0.
  // M^-1 J^T itself is never formed: the constraint acceleration is L^-T (Z^T f), one back-substitution at the end.
  double rhs_c[MR], Rd[MR], f[MR];
#pragma unroll
  for (int t = 0; t < MR; ++t) { rhs_c[t] = 0.0; Rd[t] = 0.0; f[t] = 0.0; }
  for (int r = 0; r < nr; ++r) {
    double x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = SL(S::OFF_J, r * N + i);
    double jv = 0.0, ja = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { jv += x[i] * v[i]; ja += x[i] * qacc0[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double sum = x[i];
#pragma unroll
      for (int t = 0; t < i; ++t) sum -= SL(S::OFF_M, i * N + t) * x[t];
      x[i] = sum * invd[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) SL(S::OFF_J, r * N + i) = x[i];
#pragma unroll
    for (int t = 0; t < MR; ++t)
      if (t == r) rhs_c[t] = (-rb_[t] * jv - rk[t] * rd[t] * rr[t]) - ja;
    for (int c2 = 0; c2 <= r; ++c2) {
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) sum += x[i] * SL(S::OFF_J, c2 * N + i);
      SL(S::OFF_A, r * (r + 1) / 2 + c2) = sum;
    }
  }
  // Rd is reused for 1 / (a_tt + R_t): the 30 sweeps divide by it once per row instead of once per row per sweep
#pragma unroll
  for (int t = 0; t < MR; ++t)
    if (t < nr) { const double att = SL(S::OFF_A, S::a_idx(t, t)); Rd[t] = 1.0 / (att + (1.0 - rd[t]) / rd[t] * att); }
  for (int it = 0; it < m.pgs_iters; ++it) {
#pragma unroll
    for (int t = 0; t < MR; ++t) {
      if (t >= nr) continue;
      const double aii = SL(S::OFF_A, S::a_idx(t, t));
      double res = rhs_c[t] + aii * f[t];
#pragma unroll
      for (int u = 0; u < MR; ++u)
        if (u < nr) res -= SL(S::OFF_A, S::a_idx(t, u)) * f[u];
      double fi = res * Rd[t];
      if (rkind[t] == 1) {
        const double lim = rmu[t] * f[t > 0 ? t - 1 : 0];
        fi = fmin(fmax(fi, -lim), lim);
      } else {
        fi = fmax(fi, 0.0);
      }
      f[t] = fi;
    }
  }
  double w[N];
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = 0.0;
  for (int r = 0; r < nr; ++r) {
    double fr = 0.0;
#pragma unroll
    for (int t = 0; t < MR; ++t)
      if (t == r) fr = f[t];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] += SL(S::OFF_J, r * N + i) * fr;
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double sum = w[i];
#pragma unroll
    for (int t = i + 1; t < N; ++t) sum -= SL(S::OFF_M, t * N + i) * w[t];
    w[i] = sum * invd[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) qacc[i] = qacc0[i] + w[i];
}

template <int NB, int MR, int BLOCK>
__device__ void env_substep(const PlanarModelDev& m, double (&q)[NB + 2], double (&v)[NB + 2], const double* ctrl,
                            double* sm, int lane_in_block) {
  constexpr int N = NB + 2;
  const double h = m.timestep;
  // classic RK4 as ONE dynamics call site in a 4-trip loop (stage weights 1,2,2,1; stage offsets h/2, h/2, h): the sums
  // accumulate in the order the oracle writes them, so the result is bit-identical to the unrolled form at a quarter of
  // the code size and without the ten stage arrays
  double qs[N], vs[N], qsum[N], vsum[N], a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { qs[i] = q[i]; vs[i] = v[i]; qsum[i] = 0.0; vsum[i] = 0.0; }
#pragma unroll 1
  for (int stage = 0; stage < 4; ++stage) {
    env_dynamics<NB, MR, BLOCK>(m, qs, vs, ctrl, a, sm, lane_in_block);
    const double w = (stage == 1 || stage == 2) ? 2.0 : 1.0;
    const double ch = (stage == 2) ? h : 0.5 * h;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      qsum[i] = stage == 0 ? vs[i] : qsum[i] + w * vs[i];
      vsum[i] = stage == 0 ? a[i] : vsum[i] + w * a[i];
      const double vn = v[i] + ch * a[i];
      qs[i] = q[i] + ch * vs[i];
      vs[i] = vn;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double qn = q[i] + h / 6.0 * qsum[i];
    const double vn = v[i] + h / 6.0 * vsum[i];
    q[i] = qn; v[i] = vn;
  }
}

struct EnvStepArgs {
  const PlanarModelDev* m;
  double *qpos, *qvel;
  int n_env;
  const int* ids;         // nullable: step only these envs (compact i/o indexing)
  int n_ids;
  const float* act;       // [n_ids][a]
  float *obs, *rew;       // [n_ids][o], [n_ids]   (obs = observation AFTER the step, pre-reset)
  unsigned char* done;    // [n_ids]
  // fused-rollout extras (all nullable / 0)
  float* obs_cur;         // [n_env][o]: the policy's next input (post auto-reset)
  int auto_reset, max_path_length, no_terminal;
  float* stage; int stage_len; int* flush_len;   // path mode: records go to stage[env][step of the episode], flush_len[env] = length | terminal << 30 when the episode ends
  const float* rec_act;          // nullable: [n_ids][a] action written into the replay record instead of `act`
  const unsigned char* frozen;   // nullable: envs whose flag is set do not step (evaluation: one episode per env)
  int* ep_len; double* ep_ret; double* stats;
  float* replay; int rec; long long cap, top;   // transition record written at slot (top + env) % cap
  uint64_t seed; uint32_t stream; unsigned long long step;
};

template <int NB>
__device__ __forceinline__ void env_write_obs(const PlanarModelDev& m, const double (&q)[NB + 2], const double (&v)[NB + 2],
                                              float* dst) {
  constexpr int N = NB + 2;
#pragma unroll
  for (int i = 1; i < N; ++i) dst[i - 1] = (float)((q[i] - m.obs_shift[i - 1]) * m.obs_inv_scale[i - 1]);   // qpos[1:] (hopper.py:29-30)
#pragma unroll
  for (int i = 0; i < N; ++i)                                                                                // clip(qvel, +-10)
    dst[N - 1 + i] = (float)(((m.qvel_clip > 0.0 ? fmin(fmax(v[i], -m.qvel_clip), m.qvel_clip) : v[i]) - m.obs_shift[N - 1 + i]) *
                             m.obs_inv_scale[N - 1 + i]);
}

__device__ __forceinline__ double env_uniform(uint64_t seed, uint32_t stream, unsigned long long step, uint32_t env,
                                              uint32_t k) {
  uint32_t c[4] = {env, k >> 2, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
  return ((double)c[k & 3] + 0.5) * (1.0 / 4294967296.0);
}

template <int NB>
__device__ __forceinline__ void env_reset_state(const PlanarModelDev& m, uint64_t seed, uint32_t stream,
                                                unsigned long long step, uint32_t env, double (&q)[NB + 2],
                                                double (&v)[NB + 2]) {
  constexpr int N = NB + 2;  // hopper.py:32-40: init + U(+-reset_noise) on qpos and qvel
#pragma unroll
  for (int i = 0; i < N; ++i) {
    q[i] = m.init_qpos[i] + m.reset_noise * (2.0 * env_uniform(seed, stream, step, env, i) - 1.0);
    if (m.reset_noise_vel_std > 0.0) {   // HalfCheetahEnv.reset_model: qvel = init + 0.1 * randn (Box-Muller on two draws)
      const double u1 = env_uniform(seed, stream, step, env, N + 2 * i), u2 = env_uniform(seed, stream, step, env, N + 2 * i + 1);
      v[i] = m.reset_noise_vel_std * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    } else {
      v[i] = m.reset_noise * (2.0 * env_uniform(seed, stream, step, env, N + i) - 1.0);
    }
  }
}

#include "env2d_group.h"
#ifdef ILSX_EG_PROFILE
extern "C" int ilsx_debug_eg_prof(unsigned long long* out16, int reset) {   // measurement build only
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_eg_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_eg_prof), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif

template <int NB, int MR, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_env_step(const EnvStepArgs A) {
  constexpr int N = NB + 2;
  extern __shared__ __attribute__((aligned(16))) double smd_all[];
  // The model record (2 KB of constants every phase of the dynamics reads, many of them under run-time indices: geom -> body, actuator
  // -> body) is copied into LDS once per workgroup.  Read in place from global memory it cost ~100 vector loads per dynamics evaluation,
  // each waited for on the spot by the one wave of its SIMD (rocprofv3: 1546 VMEM reads per wave per env step, SQ_WAIT_ANY 54 % of
  // the wave's cycles): a broadcast LDS read is a tenth of that latency.
  constexpr int MODEL_DOUBLES = (sizeof(PlanarModelDev) + 7) / 8;
  {
    const double* src = reinterpret_cast<const double*>(A.m);
    for (int i = threadIdx.x; i < MODEL_DOUBLES; i += BLOCK) smd_all[i] = src[i];
    __syncthreads();
  }
  const PlanarModelDev& m = *reinterpret_cast<const PlanarModelDev*>(smd_all);
  double* smd = smd_all + MODEL_DOUBLES;
  const int lane_in_block = threadIdx.x;
  const int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t >= A.n_ids) return;
  const int env = A.ids ? A.ids[t] : t;
  if (A.frozen && A.frozen[env]) return;
  const int o = m.obs_dim, na = m.n_act;
  double q[N], v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { q[i] = A.qpos[(size_t)i * A.n_env + env]; v[i] = A.qvel[(size_t)i * A.n_env + env]; }
  double ctrl[ENV_MAXB];
  double ctrl_sq = 0.0;
  for (int k = 0; k < na; ++k) {  // NormalizedBoxEnv: lb + (a+1)/2*(ub-lb), clip; ctrlrange is [-1,1] (wrappers.py:343-346)
    const double a = fmin(fmax((double)A.act[(size_t)t * na + k], -1.0), 1.0);
    ctrl[k] = a; ctrl_sq += a * a;
  }
  float obs_before[2 * N - 1];
  if (A.replay) env_write_obs<NB>(m, q, v, obs_before);
  const double x0 = q[0];
  for (int s = 0; s < m.frame_skip; ++s) env_substep<NB, MR, BLOCK>(m, q, v, ctrl, smd, lane_in_block);
  const double dt = m.timestep * m.frame_skip;
  const double reward = (q[0] - x0) / dt + m.alive - m.ctrl_cost * ctrl_sq;  // hopper.py:16-18
  bool ok;
  if (m.task == 0) {  // hopper.py:19-25
    ok = isfinite(q[0]) && isfinite(q[1]) && q[1] > m.z_min && fabs(q[2]) < m.ang_max;
#pragma unroll
    for (int i = 2; i < N; ++i) ok = ok && isfinite(q[i]) && fabs(q[i]) < m.state_max;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && isfinite(v[i]) && fabs(v[i]) < m.state_max;
  } else if (m.task == 1) {   // walker2d.py:17-20
    ok = q[1] > m.z_min && q[1] < m.z_max && q[2] > -m.ang_max && q[2] < m.ang_max;
  } else {                    // HalfCheetah: done = False always
    ok = true;
  }
  const bool done = !ok;
  float ob[2 * N - 1];
  env_write_obs<NB>(m, q, v, ob);
  if (A.obs) for (int i = 0; i < o; ++i) A.obs[(size_t)t * o + i] = ob[i];
  if (A.rew) A.rew[t] = (float)reward;
  if (A.done) A.done[t] = done ? 1 : 0;
  if (A.replay) {  // fused replay insert: one 128-byte-aligned record per transition
    long long slot = A.top + env;
    if (slot >= A.cap) slot -= A.cap;
    float* rec = A.stage ? A.stage + ((size_t)env * A.stage_len + A.ep_len[env]) * A.rec : A.replay + (size_t)slot * A.rec;
    for (int i = 0; i < o; ++i) rec[i] = obs_before[i];
    const float* ra = A.rec_act ? A.rec_act : A.act;   // DAgger stores the expert's label, not the executed action
    for (int k = 0; k < na; ++k) rec[o + k] = ra[(size_t)t * na + k];
    rec[o + na] = (float)reward;
    rec[o + na + 1] = (done && !A.no_terminal) ? 1.0f : 0.0f;   // base_algorithm.py:195-196,208-210
    for (int i = 0; i < o; ++i) rec[o + na + 2 + i] = ob[i];
    rec[2 * o + na + 2] = 0.0f; rec[2 * o + na + 3] = 0.0f;   // absorbing = [0, 0] (base_algorithm.py:211-213)
  }
  if (A.auto_reset) {
    const int len = A.ep_len[env] + 1;
    const double ret = A.ep_ret[env] + reward;
    // time-limit ends are NOT terminal (base_algorithm.py:264-277).  With no_terminal the reference overwrites `terminals` BEFORE it
    // decides to reset (:195-196 ahead of :215), so an unhealthy env is stepped on until the time limit: the fallen states are visited
    // and rewarded, which is what grounds their values in adversarial IRL.  A state that is no longer finite always resets.
    bool finite = true;
#pragma unroll
    for (int i = 0; i < N; ++i) finite = finite && isfinite(q[i]) && isfinite(v[i]);
    const bool end = (done && !A.no_terminal) || len >= A.max_path_length || !finite;
    if (end) {
      atomicAdd(&A.stats[0], 1.0);
      atomicAdd(&A.stats[1], ret);
      env_reset_state<NB>(m, A.seed, A.stream, A.step, (uint32_t)env, q, v);
      env_write_obs<NB>(m, q, v, ob);
    }
    A.ep_len[env] = end ? 0 : len;
    A.ep_ret[env] = end ? 0.0 : ret;
    if (A.flush_len) A.flush_len[env] = end ? (len | ((done && !A.no_terminal) ? (1 << 30) : 0)) : 0;
  }
  if (A.obs_cur) for (int i = 0; i < o; ++i) A.obs_cur[(size_t)env * o + i] = ob[i];
#pragma unroll
  for (int i = 0; i < N; ++i) { A.qpos[(size_t)i * A.n_env + env] = q[i]; A.qvel[(size_t)i * A.n_env + env] = v[i]; }
}

template <int NB>
__global__ void k_env_reset(const PlanarModelDev* mp, double* qpos, double* qvel, int n_env, const int* ids, int n_ids,
                            float* obs, float* obs_cur, int* ep_len, double* ep_ret, uint64_t seed, uint32_t stream,
                            unsigned long long step) {
  constexpr int N = NB + 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_ids) return;
  const int env = ids ? ids[t] : t;
  const PlanarModelDev& m = *mp;
  double q[N], v[N];
  env_reset_state<NB>(m, seed, stream, step, (uint32_t)env, q, v);
  float ob[2 * N - 1];
  env_write_obs<NB>(m, q, v, ob);
  for (int i = 0; i < m.obs_dim; ++i) {
    if (obs) obs[(size_t)t * m.obs_dim + i] = ob[i];
    if (obs_cur) obs_cur[(size_t)env * m.obs_dim + i] = ob[i];
  }
  ep_len[env] = 0; ep_ret[env] = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) { qpos[(size_t)i * n_env + env] = q[i]; qvel[(size_t)i * n_env + env] = v[i]; }
}

// uniform(-1,1) actions: env.action_space.sample() per env while the replay is short (base_algorithm.py:369-380)
__global__ void k_random_actions(float* act, int n, uint64_t seed, uint32_t stream, unsigned long long step, int a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * a) return;
  const int env = t / a, k = t - env * a;
  act[t] = (float)(2.0 * env_uniform(seed, stream, step, (uint32_t)env, (uint32_t)k) - 1.0);
}

// ================================================================================================ 3-D engine kernels
__device__ void e3_reset_state(const E3Ctx& C, uint64_t seed, uint32_t stream, unsigned long long step, uint32_t envu) {
  const Spatial3Dev& m = *C.m;
  double* scr = C.scr; const int n_env = C.n_env, env = C.env;
  // humanoid.py:62-73 / ant.py:36-43: init + U(+-c) on every qpos component (the quaternion is re-normalised, as MuJoCo does
  // before it uses it); qvel = U(+-c) (Humanoid) or 0.1 * randn (Ant)
  for (int i = 0; i < m.nq; ++i) E3S(E3St::Q0 + i) = m.init_qpos[i] + m.reset_noise * (2.0 * env_uniform(seed, stream, step, envu, i) - 1.0);
  {
    double qw = E3S(E3St::Q0 + 3), qx = E3S(E3St::Q0 + 4), qy = E3S(E3St::Q0 + 5), qz = E3S(E3St::Q0 + 6);
    const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    E3S(E3St::Q0 + 3) = qw * nrm; E3S(E3St::Q0 + 4) = qx * nrm; E3S(E3St::Q0 + 5) = qy * nrm; E3S(E3St::Q0 + 6) = qz * nrm;
  }
  for (int i = 0; i < m.nv; ++i) {
    if (m.reset_noise_vel_std > 0.0) {
      const double u1 = env_uniform(seed, stream, step, envu, m.nq + 2 * i), u2 = env_uniform(seed, stream, step, envu, m.nq + 2 * i + 1);
      E3S(E3St::V0 + i) = m.reset_noise_vel_std * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    } else {
      E3S(E3St::V0 + i) = m.reset_noise * (2.0 * env_uniform(seed, stream, step, envu, m.nq + i) - 1.0);
    }
  }
  for (int k = 0; k < m.n_act; ++k) E3S(E3St::CTRL + k) = 0.0;   // qfrc_actuator of a freshly reset model is zero
}

__global__ __launch_bounds__(64) void k_env3d_step(const EnvStepArgs A, const Spatial3Dev* mp, double* scr) {
  const Spatial3Dev& m = *mp;
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= A.n_ids) return;
  const int env = A.ids ? A.ids[t] : t, n_env = A.n_env;
  if (A.frozen && A.frozen[env]) return;
  const E3Ctx C{scr, n_env, env, mp};
  const int o = m.obs_dim, na = m.n_act;
  for (int i = 0; i < m.nq; ++i) E3S(E3St::Q0 + i) = A.qpos[(size_t)i * n_env + env];
  for (int i = 0; i < m.nv; ++i) E3S(E3St::V0 + i) = A.qvel[(size_t)i * n_env + env];
  float* rec = nullptr;
  if (A.replay) {   // fused replay insert: the observation the policy acted on is the stored current observation
    long long slot = A.top + env;
    if (slot >= A.cap) slot -= A.cap;
    rec = A.stage ? A.stage + ((size_t)env * A.stage_len + A.ep_len[env]) * A.rec : A.replay + (size_t)slot * A.rec;
    for (int i = 0; i < o; ++i) rec[i] = A.obs_cur[(size_t)env * o + i];
  }
  double reward; bool done;
  e3_task_step(C, A.act + (size_t)t * na, reward, done);
  bool end = false; int len = 0; double ret = 0.0;
  if (A.auto_reset) {
    len = A.ep_len[env] + 1; ret = A.ep_ret[env] + reward;
    bool finite = isfinite(reward);
    end = (done && !A.no_terminal) || len >= A.max_path_length || !finite;   // see k_env_step: no_terminal keeps stepping an unhealthy env
  }
  float* obs_out = A.obs ? A.obs + (size_t)t * o : nullptr;
  float* cur = (A.obs_cur && !end) ? A.obs_cur + (size_t)env * o : nullptr;
  float* rnext = rec ? rec + o + na + 2 : nullptr;
  e3_observe(C, [&](int i, double val) {
    const float f = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]);
    if (obs_out) obs_out[i] = f;
    if (cur) cur[i] = f;
    if (rnext) rnext[i] = f;
  });
  if (A.rew) A.rew[t] = (float)reward;
  if (A.done) A.done[t] = done ? 1 : 0;
  if (rec) {
    const float* ra = A.rec_act ? A.rec_act : A.act;
    for (int k = 0; k < na; ++k) rec[o + k] = ra[(size_t)t * na + k];
    rec[o + na] = (float)reward;
    rec[o + na + 1] = (done && !A.no_terminal) ? 1.0f : 0.0f;
    rec[2 * o + na + 2] = 0.0f; rec[2 * o + na + 3] = 0.0f;
  }
  if (A.auto_reset) {
    if (end) {
      atomicAdd(&A.stats[0], 1.0);
      atomicAdd(&A.stats[1], ret);
      e3_reset_state(C, A.seed, A.stream, A.step, (uint32_t)env);
      e3_kinematics(C, E3St::Q0, E3St::V0);
      float* c2 = A.obs_cur + (size_t)env * o;
      e3_observe(C, [&](int i, double val) { c2[i] = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]); });
    }
    A.ep_len[env] = end ? 0 : len;
    A.ep_ret[env] = end ? 0.0 : ret;
    if (A.flush_len) A.flush_len[env] = end ? (len | ((done && !A.no_terminal) ? (1 << 30) : 0)) : 0;
  }
  for (int i = 0; i < m.nq; ++i) A.qpos[(size_t)i * n_env + env] = E3S(E3St::Q0 + i);
  for (int i = 0; i < m.nv; ++i) A.qvel[(size_t)i * n_env + env] = E3S(E3St::V0 + i);
}

__global__ __launch_bounds__(64) void k_env3d_reset(const Spatial3Dev* mp, double* scr, double* qpos, double* qvel, int n_env, const int* ids,
                                                    int n_ids, float* obs, float* obs_cur, int* ep_len, double* ep_ret, uint64_t seed,
                                                    uint32_t stream, unsigned long long step) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n_ids) return;
  const int env = ids ? ids[t] : t;
  const Spatial3Dev& m = *mp;
  const E3Ctx C{scr, n_env, env, mp};
  e3_reset_state(C, seed, stream, step, (uint32_t)env);
  e3_kinematics(C, E3St::Q0, E3St::V0);
  const int o = m.obs_dim;
  e3_observe(C, [&](int i, double val) {
    const float f = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]);
    if (obs) obs[(size_t)t * o + i] = f;
    if (obs_cur) obs_cur[(size_t)env * o + i] = f;
  });
  ep_len[env] = 0; ep_ret[env] = 0.0;
  for (int i = 0; i < m.nq; ++i) qpos[(size_t)i * n_env + env] = E3S(E3St::Q0 + i);
  for (int i = 0; i < m.nv; ++i) qvel[(size_t)i * n_env + env] = E3S(E3St::V0 + i);
}

// ---- wave-per-env form (env3d_wave.h): one 64-lane workgroup per env, working set in that wave's LDS
__device__ __forceinline__ void e3w_reset_state(e3w_lds* S, const Spatial3Dev& m, int lane, uint64_t seed, uint32_t stream, unsigned long long step,
                                                uint32_t envu) {
  // same draws as e3_reset_state: component i of qpos uses counter i, qvel nq + i (uniform) or nq + 2i, nq + 2i + 1 (Box-Muller)
  E3W_FOR(i, m.nq) S[E3WOff::Q0 + i] = m.init_qpos[i] + m.reset_noise * (2.0 * env_uniform(seed, stream, step, envu, i) - 1.0);
  E3W_FOR(i, m.nv) {
    if (m.reset_noise_vel_std > 0.0) {
      const double u1 = env_uniform(seed, stream, step, envu, m.nq + 2 * i), u2 = env_uniform(seed, stream, step, envu, m.nq + 2 * i + 1);
      S[E3WOff::V0 + i] = m.reset_noise_vel_std * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    } else {
      S[E3WOff::V0 + i] = m.reset_noise * (2.0 * env_uniform(seed, stream, step, envu, m.nq + i) - 1.0);
    }
  }
  E3W_FOR(k, m.n_act) S[E3WOff::CTRL + k] = 0.0;   // qfrc_actuator of a freshly reset model is zero
  E3W_SYNC();
  const double qw = S[E3WOff::Q0 + 3], qx = S[E3WOff::Q0 + 4], qy = S[E3WOff::Q0 + 5], qz = S[E3WOff::Q0 + 6];
  const double nrm = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  E3W_SYNC();
  E3W_ONE { S[E3WOff::Q0 + 3] = qw * nrm; S[E3WOff::Q0 + 4] = qx * nrm; S[E3WOff::Q0 + 5] = qy * nrm; S[E3WOff::Q0 + 6] = qz * nrm; }
  E3W_SYNC();
}

template <int NV>
__global__ __launch_bounds__(64) void k_env3dw_step(const EnvStepArgs A, const Spatial3Dev* mp) {
  extern __shared__ __attribute__((aligned(16))) double e3w_smem[];
  e3w_lds* S = (e3w_lds*)e3w_smem;
  const Spatial3Dev& m = *mp;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int env = A.ids ? A.ids[t] : t, n_env = A.n_env;
  if (A.frozen && A.frozen[env]) return;
  const int o = m.obs_dim, na = m.n_act;
  E3W_FOR(i, m.nq) S[E3WOff::Q0 + i] = A.qpos[(size_t)i * n_env + env];
  E3W_FOR(i, m.nv) S[E3WOff::V0 + i] = A.qvel[(size_t)i * n_env + env];
  float* rec = nullptr;
  if (A.replay) {   // fused replay insert: the observation the policy acted on is the stored current observation
    long long slot = A.top + env;
    if (slot >= A.cap) slot -= A.cap;
    rec = A.stage ? A.stage + ((size_t)env * A.stage_len + A.ep_len[env]) * A.rec : A.replay + (size_t)slot * A.rec;
    E3W_FOR(i, o) rec[i] = A.obs_cur[(size_t)env * o + i];
  }
  E3W_SYNC();
  E3WRegs regs[1];
  e3w_regs_init(regs[0], m, lane);
  e3w_regs_pin(regs[0]);
  double reward; bool done;
  e3w_task_step<NV>(S, m, lane, regs, A.act + (size_t)t * na, reward, done);
  bool end = false; int len = 0; double ret = 0.0;
  if (A.auto_reset) {
    len = A.ep_len[env] + 1; ret = A.ep_ret[env] + reward;
    bool finite = isfinite(reward);
    end = (done && !A.no_terminal) || len >= A.max_path_length || !finite;   // see k_env_step: no_terminal keeps stepping an unhealthy env
  }
  float* obs_out = A.obs ? A.obs + (size_t)t * o : nullptr;
  float* cur = (A.obs_cur && !end) ? A.obs_cur + (size_t)env * o : nullptr;
  float* rnext = rec ? rec + o + na + 2 : nullptr;
  e3w_observe(S, m, lane, [&](int i, double val) {
    const float f = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]);
    if (obs_out) obs_out[i] = f;
    if (cur) cur[i] = f;
    if (rnext) rnext[i] = f;
  });
  if (lane == 0) {
    if (A.rew) A.rew[t] = (float)reward;
    if (A.done) A.done[t] = done ? 1 : 0;
  }
  if (rec) {
    const float* ra = A.rec_act ? A.rec_act : A.act;
    E3W_FOR(k, na) rec[o + k] = ra[(size_t)t * na + k];
    if (lane == 0) {
      rec[o + na] = (float)reward;
      rec[o + na + 1] = (done && !A.no_terminal) ? 1.0f : 0.0f;
      rec[2 * o + na + 2] = 0.0f; rec[2 * o + na + 3] = 0.0f;
    }
  }
  if (A.auto_reset) {
    if (end) {
      if (lane == 0) { atomicAdd(&A.stats[0], 1.0); atomicAdd(&A.stats[1], ret); }
      E3W_SYNC();
      e3w_reset_state(S, m, lane, A.seed, A.stream, A.step, (uint32_t)env);
      e3w_kinematics(S, m, lane, regs, E3WOff::Q0, E3WOff::V0);
      float* c2 = A.obs_cur + (size_t)env * o;
      e3w_observe(S, m, lane, [&](int i, double val) { c2[i] = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]); });
    }
    if (lane == 0) {
      A.ep_len[env] = end ? 0 : len; A.ep_ret[env] = end ? 0.0 : ret;
      if (A.flush_len) A.flush_len[env] = end ? (len | ((done && !A.no_terminal) ? (1 << 30) : 0)) : 0;
    }
  }
  E3W_FOR(i, m.nq) A.qpos[(size_t)i * n_env + env] = S[E3WOff::Q0 + i];
  E3W_FOR(i, m.nv) A.qvel[(size_t)i * n_env + env] = S[E3WOff::V0 + i];
}

__global__ __launch_bounds__(64) void k_env3dw_reset(const Spatial3Dev* mp, double* qpos, double* qvel, int n_env, const int* ids, float* obs,
                                                     float* obs_cur, int* ep_len, double* ep_ret, uint64_t seed, uint32_t stream,
                                                     unsigned long long step) {
  extern __shared__ __attribute__((aligned(16))) double e3w_smem[];
  e3w_lds* S = (e3w_lds*)e3w_smem;
  const Spatial3Dev& m = *mp;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int env = ids ? ids[t] : t;
  e3w_reset_state(S, m, lane, seed, stream, step, (uint32_t)env);
  E3WRegs regs[1];
  e3w_regs_init(regs[0], m, lane);
  e3w_regs_pin(regs[0]);
  e3w_kinematics(S, m, lane, regs, E3WOff::Q0, E3WOff::V0);
  const int o = m.obs_dim;
  e3w_observe(S, m, lane, [&](int i, double val) {
    const float f = (float)((val - m.obs_shift[i]) * m.obs_inv_scale[i]);
    if (obs) obs[(size_t)t * o + i] = f;
    if (obs_cur) obs_cur[(size_t)env * o + i] = f;
  });
  if (lane == 0) { ep_len[env] = 0; ep_ret[env] = 0.0; }
  E3W_FOR(i, m.nq) qpos[(size_t)i * n_env + env] = S[E3WOff::Q0 + i];
  E3W_FOR(i, m.nv) qvel[(size_t)i * n_env + env] = S[E3WOff::V0 + i];
}

// ------------------------------------------------------------------------------------------------ host
template <int NB, int MR, int BLOCK>
static int launch_env_step_t(ilsx_vecenv* e, const EnvStepArgs& A) {
  using S = EnvScratch<NB, MR, BLOCK>;
  const size_t lds = (size_t)S::TOTAL * BLOCK * sizeof(double) + ((sizeof(PlanarModelDev) + 7) / 8) * sizeof(double);   // per-env scratch + the model record
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute((const void*)k_env_step<NB, MR, BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  ProfScope ps(e->ctx, ILSX_K_ENV_STEP);
  ILSX_LAUNCH(ps, (k_env_step<NB, MR, BLOCK>), dim3((A.n_ids + BLOCK - 1) / BLOCK), dim3(BLOCK), lds, e->ctx->stream, A);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
// 16 lanes per env (env2d_group.h): one wavefront = one workgroup = 4 envs
template <int NB, int MR>
static int launch_envg_step_t(ilsx_vecenv* e, const EnvStepArgs& A) {
  const size_t lds = (((sizeof(PlanarModelDev) + 7) / 8) + (size_t)EG_ENVS * EgOff<NB, MR>::TOTAL) * sizeof(double);
  ProfScope ps(e->ctx, ILSX_K_ENV_STEP);
  ILSX_LAUNCH(ps, (k_envg_step<NB, MR>), dim3((A.n_ids + EG_ENVS - 1) / EG_ENVS), dim3(64), lds, e->ctx->stream, A);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
static int launch_env_step(ilsx_vecenv* e, const EnvStepArgs& A) {
  if (A.n_ids <= 0) return ILSX_OK;
  if (e->engine == 1) {
    ProfScope ps(e->ctx, ILSX_K_ENV_STEP);
    if (e->wave3 && e->nv == 23)        // Humanoid
      ILSX_LAUNCH(ps, k_env3dw_step<23>, dim3(A.n_ids), dim3(64), (size_t)E3WOff::TOTAL * 8, e->ctx->stream, A, (const Spatial3Dev*)e->dm3);
    else if (e->wave3 && e->nv == 14)   // Ant
      ILSX_LAUNCH(ps, k_env3dw_step<14>, dim3(A.n_ids), dim3(64), (size_t)E3WOff::TOTAL * 8, e->ctx->stream, A, (const Spatial3Dev*)e->dm3);
    else if (e->wave3)                  // any other tree: run-time dof count
      ILSX_LAUNCH(ps, k_env3dw_step<0>, dim3(A.n_ids), dim3(64), (size_t)E3WOff::TOTAL * 8, e->ctx->stream, A, (const Spatial3Dev*)e->dm3);
    else
      ILSX_LAUNCH(ps, k_env3d_step, dim3((A.n_ids + 63) / 64), dim3(64), 0, e->ctx->stream, A, (const Spatial3Dev*)e->dm3, e->scr3);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  static const bool lane_form = getenv("ILSX_ENV2D_LANE") != nullptr;   // A/B: the first form, one lane per env
  if (!lane_form) {
    if (e->hm.nb == 4) return launch_envg_step_t<4, 8>(e, A);
    if (e->hm.nb == 7) return e->hm.max_rows > 12 ? launch_envg_step_t<7, 16>(e, A) : launch_envg_step_t<7, 12>(e, A);
  }
  if (e->hm.nb == 4) return launch_env_step_t<4, 8, 64>(e, A);
  if (e->hm.nb == 7) return e->hm.max_rows > 12 ? launch_env_step_t<7, 16, 32>(e, A) : launch_env_step_t<7, 12, 64>(e, A);
  ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "vec-env kernels are instantiated for 4 (Hopper) and 7 (Walker2d) bodies, got %d", e->hm.nb);
}
static int launch_env_reset(ilsx_vecenv* e, const int* ids_dev, int n_ids, float* obs) {
  if (n_ids <= 0) return ILSX_OK;
  const unsigned long long step = ++e->step_ctr;
  if (e->engine == 1 && e->wave3) {
    hipLaunchKernelGGL(k_env3dw_reset, dim3(n_ids), dim3(64), (size_t)E3WOff::TOTAL * 8, e->ctx->stream, (const Spatial3Dev*)e->dm3, e->qpos, e->qvel,
                       e->n_env, ids_dev, obs, e->obs_cur, e->ep_len, e->ep_ret, e->seed, e->rng_stream, step);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  if (e->engine == 1) {
    hipLaunchKernelGGL(k_env3d_reset, dim3((n_ids + 63) / 64), dim3(64), 0, e->ctx->stream, (const Spatial3Dev*)e->dm3, e->scr3, e->qpos, e->qvel,
                       e->n_env, ids_dev, n_ids, obs, e->obs_cur, e->ep_len, e->ep_ret, e->seed, e->rng_stream, step);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  const dim3 grid((n_ids + 255) / 256), block(256);
  if (e->hm.nb == 4)
    hipLaunchKernelGGL(k_env_reset<4>, grid, block, 0, e->ctx->stream, e->dm, e->qpos, e->qvel, e->n_env, ids_dev, n_ids, obs,
                       e->obs_cur, e->ep_len, e->ep_ret, e->seed, e->rng_stream, step);
  else
    hipLaunchKernelGGL(k_env_reset<7>, grid, block, 0, e->ctx->stream, e->dm, e->qpos, e->qvel, e->n_env, ids_dev, n_ids, obs,
                       e->obs_cur, e->ep_len, e->ep_ret, e->seed, e->rng_stream, step);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// RunningMeanStd update with rows of `src` (all, or those whose flag is 0) then normalise src -> dst
static int env_obs_norm(ilsx_vecenv* e, const float* upd_src, int upd_rows, const int* flag, const float* src, float* dst, int rows) {
  if (!e->norm_obs) return ILSX_OK;
  hipStream_t st = e->ctx->stream;
  if (e->update_rms && upd_src && upd_rows > 0)
    hipLaunchKernelGGL(k_obs_rms_update, dim3(e->o), dim3(256), 0, st, upd_src, upd_rows, e->o, flag, e->rms);
  if (src && rows > 0)
    hipLaunchKernelGGL(k_obs_normalize, dim3((rows * e->o + 255) / 256), dim3(256), 0, st, src, dst, rows * e->o, e->o,
                       (const ObsRms*)e->rms);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_create(ilsx_ctx* ctx, const ilsx_planar_model* pm, int n_env, uint64_t seed, ilsx_vecenv** out) {
  if (!ctx || !pm || !out || n_env < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_vecenv_create: bad argument");
  if (pm->n_body != 4 && pm->n_body != 7)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "n_body=%d: kernels exist for 4 (Hopper) and 7 (Walker2d) bodies", pm->n_body);
  if (pm->n_geom < 1 || pm->n_geom > ENV_MAXG) ILSX_FAIL(ILSX_ERR_ARG, "n_geom=%d out of range", pm->n_geom);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_vecenv* e = new ilsx_vecenv();
  e->ctx = ctx; e->n_env = n_env; e->seed = seed; e->rng_stream = ctx->next_rng_stream++;
  PlanarModelDev& m = e->hm;
  memset(&m, 0, sizeof m);
  m.nb = pm->n_body; m.ng = pm->n_geom; m.task = pm->task; m.frame_skip = pm->frame_skip; m.pgs_iters = pm->pgs_iters;
  m.max_rows = pm->max_rows > 0 ? pm->max_rows : (pm->n_body == 4 ? 8 : 12);
  if (m.max_rows > 16) { delete e; ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "max_rows=%d: the kernels keep at most 16 constraint rows per env", m.max_rows); }
  int na = 0;
  for (int b = 0; b < m.nb; ++b) {
    m.parent[b] = pm->parent[b]; m.limited[b] = pm->limited[b];
    m.anchor[b][0] = pm->anchor[b][0]; m.anchor[b][1] = pm->anchor[b][1];
    m.com[b][0] = pm->com[b][0]; m.com[b][1] = pm->com[b][1];
    m.mass[b] = pm->mass[b]; m.inertia[b] = pm->inertia[b]; m.jsign[b] = pm->jsign[b];
    m.armature[b] = pm->armature[b]; m.damping[b] = pm->damping[b]; m.stiffness[b] = pm->stiffness[b];
    m.range[b][0] = pm->range[b][0]; m.range[b][1] = pm->range[b][1]; m.gear[b] = pm->gear[b];
    if (pm->gear[b] != 0.0) m.act_body[na++] = b;
    int mask = 0;
    for (int j = b; j >= 0; j = pm->parent[j]) mask |= 1 << j;
    m.ancmask[b] = mask;
    if (b > 0 && (pm->parent[b] < 0 || pm->parent[b] >= b)) { delete e; ILSX_FAIL(ILSX_ERR_ARG, "parent[%d] must precede the body", b); }
  }
  m.n_act = na;
  for (int g = 0; g < m.ng; ++g) {
    m.geom_body[g] = pm->geom_body[g];
    m.gp1[g][0] = pm->geom_p1[g][0]; m.gp1[g][1] = pm->geom_p1[g][1];
    m.gp2[g][0] = pm->geom_p2[g][0]; m.gp2[g][1] = pm->geom_p2[g][1];
    m.grad[g] = pm->geom_radius[g]; m.gfric[g] = pm->geom_friction[g];
  }
  m.timestep = pm->timestep; m.gravity = pm->gravity; m.reset_noise = pm->reset_noise; m.margin = pm->contact_margin;
  m.reset_noise_vel_std = pm->reset_noise_vel_std; m.qvel_clip = pm->qvel_clip;
  for (int i = 0; i < 2; ++i) { m.c_solref[i] = pm->contact_solref[i]; m.l_solref[i] = pm->limit_solref[i]; }
  for (int i = 0; i < 3; ++i) { m.c_solimp[i] = pm->contact_solimp[i]; m.l_solimp[i] = pm->limit_solimp[i]; }
  m.ctrl_cost = pm->ctrl_cost; m.alive = pm->alive_bonus; m.z_min = pm->z_min; m.z_max = pm->z_max;
  m.ang_max = pm->ang_max; m.state_max = pm->state_max;
  e->n = m.nb + 2; e->nq = e->nv = e->n; e->o = 2 * e->n - 1; e->a = na;
  m.obs_dim = e->o;
  for (int i = 0; i < e->n; ++i) m.init_qpos[i] = pm->init_qpos[i];
  for (int i = 0; i < 2 * (ENV_MAXB + 2); ++i) { m.obs_shift[i] = 0.0; m.obs_inv_scale[i] = 1.0; }
  const size_t N = (size_t)n_env;
  int rc = ctx_alloc(ctx, sizeof m, (void**)&e->dm);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->n * 8, (void**)&e->qpos);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->n * 8, (void**)&e->qvel);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->obs_cur);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->a * 4, (void**)&e->act);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->nobs);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->rew);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N, (void**)&e->done);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->ep_len);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->obs_n);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(ObsRms), (void**)&e->rms);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 8, (void**)&e->ep_ret);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, 4 * 8, (void**)&e->stats);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->ids);
  if (rc != ILSX_OK) { delete e; return rc; }
  HIPCHK(hipMemcpyAsync(e->dm, &m, sizeof m, hipMemcpyHostToDevice, ctx->stream));
  {
    ObsRms h;   // RunningMeanStd(): mean 0, var 1, count 0 (normalizer.py:133-136)
    for (int i = 0; i < ENV_MAX_OBS; ++i) { h.mean[i] = 0.0; h.var[i] = 1.0; h.count[i] = 0.0; }
    HIPCHK(hipMemcpyAsync(e->rms, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ILSX_TRY(launch_env_reset(e, nullptr, n_env, nullptr));
  *out = e;
  return ILSX_OK;
}

// 3-D models (Ant / Humanoid): same handle type, same protocol entry points; the engine behind step / reset is env3d.h
extern "C" int ilsx_vecenv_create_spatial(ilsx_ctx* ctx, const ilsx_spatial_model* sm, int n_env, uint64_t seed, ilsx_vecenv** out) {
  if (!ctx || !sm || !out || n_env < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_vecenv_create_spatial: bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_vecenv* e = new ilsx_vecenv();
  e->ctx = ctx; e->n_env = n_env; e->seed = seed; e->rng_stream = ctx->next_rng_stream++; e->engine = 1;
  e->wave3 = !(getenv("ILSX_ENV3D_LANE") && atoi(getenv("ILSX_ENV3D_LANE")) != 0);
  if (e->wave3) {
    HIPCHK(hipFuncSetAttribute((const void*)k_env3dw_step<23>, hipFuncAttributeMaxDynamicSharedMemorySize, E3WOff::TOTAL * 8));
    HIPCHK(hipFuncSetAttribute((const void*)k_env3dw_step<14>, hipFuncAttributeMaxDynamicSharedMemorySize, E3WOff::TOTAL * 8));
    HIPCHK(hipFuncSetAttribute((const void*)k_env3dw_step<0>, hipFuncAttributeMaxDynamicSharedMemorySize, E3WOff::TOTAL * 8));
    HIPCHK(hipFuncSetAttribute((const void*)k_env3dw_reset, hipFuncAttributeMaxDynamicSharedMemorySize, E3WOff::TOTAL * 8));
  }
  e->hm3 = new Spatial3Dev();
  Spatial3Dev& m = *e->hm3;
  if (const char* why = e3_build_model(sm, m)) { delete e->hm3; delete e; ILSX_FAIL(ILSX_ERR_ARG, "ilsx_vecenv_create_spatial: %s", why); }
  e->n = m.nq; e->nq = m.nq; e->nv = m.nv; e->o = m.obs_dim; e->a = m.n_act;
  const size_t N = (size_t)n_env;
  int rc = ctx_alloc(ctx, sizeof m, (void**)&e->dm3);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, (size_t)E3Off::TOTAL * N * 8, (void**)&e->scr3);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->nq * 8, (void**)&e->qpos);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->nv * 8, (void**)&e->qvel);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->obs_cur);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->a * 4, (void**)&e->act);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->nobs);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->rew);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N, (void**)&e->done);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->ep_len);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * e->o * 4, (void**)&e->obs_n);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(ObsRms), (void**)&e->rms);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 8, (void**)&e->ep_ret);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, 4 * 8, (void**)&e->stats);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * 4, (void**)&e->ids);
  if (rc != ILSX_OK) { delete e->hm3; delete e; return rc; }
  HIPCHK(hipMemcpyAsync(e->dm3, &m, sizeof m, hipMemcpyHostToDevice, ctx->stream));
  {
    std::vector<ObsRms> h(1);
    for (int i = 0; i < ENV_MAX_OBS; ++i) { h[0].mean[i] = 0.0; h[0].var[i] = 1.0; h[0].count[i] = 0.0; }
    HIPCHK(hipMemcpyAsync(e->rms, h.data(), sizeof(ObsRms), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  ILSX_TRY(launch_env_reset(e, nullptr, n_env, nullptr));
  *out = e;
  return ILSX_OK;
}
extern "C" int ilsx_vecenv_set_path_mode(ilsx_vecenv* e, int on) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  e->path_mode = on != 0;
  return ILSX_OK;
}
extern "C" int ilsx_vecenv_state_dims(const ilsx_vecenv* e, int* nq, int* nv) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  if (nq) *nq = e->nq;
  if (nv) *nv = e->nv;
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_destroy(ilsx_vecenv* e) {
  if (!e) return ILSX_OK;
  void* ps[] = {e->dm, e->qpos, e->qvel, e->obs_cur, e->act, e->nobs, e->rew, e->done, e->ep_len, e->ep_ret, e->stats, e->ids, e->dm3, e->scr3, e->stage, e->flush_len};
  for (void* p : ps) if (p) ctx_free(e->ctx, p);
  if (e->flush_host) hipHostFree(e->flush_host);
  if (e->grp_flush_host) hipHostFree(e->grp_flush_host);
  if (e->grp_flush_dev) ctx_free(e->ctx, e->grp_flush_dev);
  delete e->hm3;
  delete e;
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_dims(const ilsx_vecenv* e, int* obs_dim, int* act_dim, int* n_dof, int* n_env) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  if (obs_dim) *obs_dim = e->o;
  if (act_dim) *act_dim = e->a;
  if (n_dof) *n_dof = e->n;
  if (n_env) *n_env = e->n_env;
  return ILSX_OK;
}

static int env_upload_ids(ilsx_vecenv* e, const int32_t* ids_host, int n_ids, const int** dev) {
  *dev = nullptr;
  if (!ids_host) return ILSX_OK;
  for (int i = 0; i < n_ids; ++i)
    if (ids_host[i] < 0 || ids_host[i] >= e->n_env) ILSX_FAIL(ILSX_ERR_ARG, "env id %d out of range", ids_host[i]);
  HIPCHK(hipMemcpyAsync(e->ids, ids_host, (size_t)n_ids * 4, hipMemcpyHostToDevice, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  *dev = e->ids;
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_reset(ilsx_vecenv* e, const int32_t* ids_host, int n_ids, float* obs) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  HIPCHK(hipSetDevice(e->ctx->device));
  if (!ids_host) n_ids = e->n_env;
  if (n_ids < 0 || n_ids > e->n_env) ILSX_FAIL(ILSX_ERR_ARG, "n_ids=%d out of range", n_ids);
  const int* dev = nullptr;
  ILSX_TRY(env_upload_ids(e, ids_host, n_ids, &dev));
  if (!e->norm_obs) return launch_env_reset(e, dev, n_ids, obs);
  // reset(id) returns normalised observations and feeds the running statistics (vecenvs.py:173-181)
  float* raw = obs ? obs : e->nobs;   // compact [n_ids][o]
  ILSX_TRY(launch_env_reset(e, dev, n_ids, raw));
  ILSX_TRY(env_obs_norm(e, raw, n_ids, nullptr, obs, obs, obs ? n_ids : 0));
  return env_obs_norm(e, nullptr, 0, nullptr, e->obs_cur, e->obs_n, e->n_env);
}

extern "C" int ilsx_vecenv_step(ilsx_vecenv* e, const float* act, const int32_t* ids_host, int n_ids, float* obs,
                                float* rew, uint8_t* done) {
  if (!e || !act) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_vecenv_step: NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  if (!ids_host) n_ids = e->n_env;
  if (n_ids < 0 || n_ids > e->n_env) ILSX_FAIL(ILSX_ERR_ARG, "n_ids=%d out of range", n_ids);
  const int* dev = nullptr;
  ILSX_TRY(env_upload_ids(e, ids_host, n_ids, &dev));
  EnvStepArgs A;
  memset(&A, 0, sizeof A);
  A.m = e->dm; A.qpos = e->qpos; A.qvel = e->qvel; A.n_env = e->n_env;
  A.ids = dev; A.n_ids = n_ids; A.act = act; A.obs = obs; A.rew = rew; A.done = done;
  A.obs_cur = e->obs_cur;
  if (!e->norm_obs) return launch_env_step(e, A);
  if (!obs) A.obs = e->nobs;
  ILSX_TRY(launch_env_step(e, A));   // step() returns normalised observations (vecenvs.py:251-257)
  ILSX_TRY(env_obs_norm(e, A.obs, n_ids, nullptr, obs, obs, obs ? n_ids : 0));
  return env_obs_norm(e, nullptr, 0, nullptr, e->obs_cur, e->obs_n, e->n_env);
}

extern "C" int ilsx_vecenv_get_state(ilsx_vecenv* e, double* qpos_host, double* qvel_host) {
  if (!e || !qpos_host || !qvel_host) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  const size_t N = e->n_env, nq = e->nq, nv = e->nv;
  std::vector<double> tq(N * nq), tv(N * nv);
  HIPCHK(hipMemcpyAsync(tq.data(), e->qpos, N * nq * 8, hipMemcpyDeviceToHost, e->ctx->stream));
  HIPCHK(hipMemcpyAsync(tv.data(), e->qvel, N * nv * 8, hipMemcpyDeviceToHost, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  for (size_t j = 0; j < N; ++j) {
    for (size_t i = 0; i < nq; ++i) qpos_host[j * nq + i] = tq[i * N + j];
    for (size_t i = 0; i < nv; ++i) qvel_host[j * nv + i] = tv[i * N + j];
  }
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_set_state(ilsx_vecenv* e, const double* qpos_host, const double* qvel_host) {
  if (!e || !qpos_host || !qvel_host) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  const size_t N = e->n_env, nq = e->nq, nv = e->nv;
  std::vector<double> tq(N * nq), tv(N * nv);
  for (size_t j = 0; j < N; ++j) {
    for (size_t i = 0; i < nq; ++i) tq[i * N + j] = qpos_host[j * nq + i];
    for (size_t i = 0; i < nv; ++i) tv[i * N + j] = qvel_host[j * nv + i];
  }
  HIPCHK(hipMemcpyAsync(e->qpos, tq.data(), N * nq * 8, hipMemcpyHostToDevice, e->ctx->stream));
  HIPCHK(hipMemcpyAsync(e->qvel, tv.data(), N * nv * 8, hipMemcpyHostToDevice, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  return ILSX_OK;
}

extern "C" int ilsx_vecenv_cur_obs(ilsx_vecenv* e, float** dev_ptr) {
  if (!e || !dev_ptr) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  *dev_ptr = const_cast<float*>(e->policy_obs());
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------------ terminals.py
// One wave per row: lanes stride the row for the all-finite / all-below-100 scans, lane 0 applies the task's thresholds.
__global__ __launch_bounds__(256) void k_is_terminal(int kind, const float* __restrict__ x, int n, int o, unsigned char* done) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* r = x + (size_t)row * o;
  bool fin = true, below = true;
  if (kind == ILSX_TERM_INVERTED_PENDULUM || kind == ILSX_TERM_HOPPER || kind == ILSX_TERM_ANT)
    for (int i = lane; i < o; i += 64) {
      const float v = r[i];
      fin = fin && isfinite(v);
      if (i >= 1) below = below && (v < 100.0f);   // terminals.py:61: abs() wraps the comparison, so only an upper bound
    }
  fin = __all(fin); below = __all(below);
  if (lane != 0) return;
  bool d = false;
  if (kind == ILSX_TERM_INVERTED_PENDULUM) {
    d = !(fin && fabsf(r[1]) <= 0.2f);
  } else if (kind == ILSX_TERM_INVERTED_DOUBLE_PENDULUM) {
    const float th1 = atan2f(r[1], r[3]), th2 = atan2f(r[2], r[4]);
    d = 0.6f * (r[3] + cosf(th1 + th2)) <= 1.0f;
  } else if (kind == ILSX_TERM_HOPPER) {
    d = !(fin && below && r[0] > 0.7f && fabsf(r[1]) < 0.2f);
  } else if (kind == ILSX_TERM_WALKER2D) {
    d = !(r[0] > 0.8f && r[0] < 2.0f && r[1] > -1.0f && r[1] < 1.0f);
  } else if (kind == ILSX_TERM_HUMANOID) {
    d = r[0] < 1.0f || r[0] > 2.0f;
  } else if (kind == ILSX_TERM_ANT) {
    d = !(fin && r[0] >= 0.2f && r[0] <= 1.0f);
  }
  done[row] = d ? 1 : 0;
}

extern "C" int ilsx_is_terminal(ilsx_ctx* ctx, int kind, const float* next_obs, int n, int obs_dim, uint8_t* done) {
  if (!ctx || !next_obs || !done || n < 0) ILSX_FAIL(ILSX_ERR_ARG, "is_terminal: null argument or n < 0");
  static const int min_dim[7] = {2, 5, 2, 2, 1, 1, 1};
  if (kind < 0 || kind > ILSX_TERM_ANT) ILSX_FAIL(ILSX_ERR_ARG, "is_terminal: unknown kind %d", kind);
  if (obs_dim < min_dim[kind]) ILSX_FAIL(ILSX_ERR_ARG, "is_terminal: kind %d reads %d observation columns, got %d", kind, min_dim[kind], obs_dim);
  if (n == 0) return ILSX_OK;
  hipLaunchKernelGGL(k_is_terminal, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, kind, next_obs, n, obs_dim, done);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// ScaledEnv (obs - mean)/(std + EPS) and MinmaxEnv (obs - min)/(max - min + EPS) (wrappers.py:53-203): a fixed affine map of
// every observation the env hands out or records; shift / scale are HOST float64 [obs_dim] (scale already includes EPS).
// The envs are reset so that the observations they currently show follow the new map.
extern "C" int ilsx_vecenv_set_obs_affine(ilsx_vecenv* e, const double* shift_host, const double* scale_host) {
  if (!e || !shift_host || !scale_host) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_vecenv_set_obs_affine: NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  for (int i = 0; i < e->o; ++i) {
    if (!(scale_host[i] != 0.0)) ILSX_FAIL(ILSX_ERR_ARG, "observation scale %d is zero", i);
    if (e->engine == 1) { e->hm3->obs_shift[i] = shift_host[i]; e->hm3->obs_inv_scale[i] = 1.0 / scale_host[i]; }
    else { e->hm.obs_shift[i] = shift_host[i]; e->hm.obs_inv_scale[i] = 1.0 / scale_host[i]; }
  }
  if (e->engine == 1) HIPCHK(hipMemcpyAsync(e->dm3, e->hm3, sizeof(Spatial3Dev), hipMemcpyHostToDevice, e->ctx->stream));
  else HIPCHK(hipMemcpyAsync(e->dm, &e->hm, sizeof e->hm, hipMemcpyHostToDevice, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  ILSX_TRY(launch_env_reset(e, nullptr, e->n_env, nullptr));
  return env_obs_norm(e, nullptr, 0, nullptr, e->obs_cur, e->obs_n, e->n_env);
}

// vecenvs.py:104-113 (obs_rms / norm_obs / update_obs_rms attributes of BaseVectorEnv)
extern "C" int ilsx_vecenv_obs_norm(ilsx_vecenv* e, int norm_obs, int update_obs_rms) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  if (e->o > ENV_MAX_OBS) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "obs_dim %d > %d", e->o, ENV_MAX_OBS);
  HIPCHK(hipSetDevice(e->ctx->device));
  e->norm_obs = norm_obs != 0; e->update_rms = update_obs_rms != 0;
  // the observations the envs currently show become the first batch, like the reset() that follows construction
  return env_obs_norm(e, e->obs_cur, e->n_env, nullptr, e->obs_cur, e->obs_n, e->n_env);
}
extern "C" int ilsx_vecenv_get_obs_rms(ilsx_vecenv* e, double* mean, double* var, double* count) {
  if (!e || !mean || !var || !count) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  ObsRms h;
  HIPCHK(hipMemcpyAsync(&h, e->rms, sizeof h, hipMemcpyDeviceToHost, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  for (int i = 0; i < e->o; ++i) { mean[i] = h.mean[i]; var[i] = h.var[i]; }
  *count = h.count[0];
  return ILSX_OK;
}
extern "C" int ilsx_vecenv_set_obs_rms(ilsx_vecenv* e, const double* mean, const double* var, double count) {
  if (!e || !mean || !var) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(e->ctx->device));
  ObsRms h;
  for (int i = 0; i < ENV_MAX_OBS; ++i) { h.mean[i] = i < e->o ? mean[i] : 0.0; h.var[i] = i < e->o ? var[i] : 1.0; h.count[i] = count; }
  HIPCHK(hipMemcpyAsync(e->rms, &h, sizeof h, hipMemcpyHostToDevice, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  return env_obs_norm(e, nullptr, 0, nullptr, e->obs_cur, e->obs_n, e->n_env);
}

// after an auto-resetting step: statistics see the step's observations, then the reset observations of the envs whose
// episode just ended (the reference's step() followed by reset(ids)), and the policy input is re-normalised
static int env_after_step_norm(ilsx_vecenv* e) {
  if (!e->norm_obs) return ILSX_OK;
  ILSX_TRY(env_obs_norm(e, e->nobs, e->n_env, nullptr, nullptr, nullptr, 0));
  return env_obs_norm(e, e->obs_cur, e->n_env, e->ep_len, e->obs_cur, e->obs_n, e->n_env);
}

// One iteration of BaseAlgorithm's sampling loop (base_algorithm.py:183-277) for ALL envs, on the device:
// actions (policy or uniform random) -> physics -> transition record into the replay ring -> auto-reset.
// The stepper's arguments for one fused rollout step of `e` (auto-reset, record into `rb`'s ring — or, with path mode on, into the per-env
// episode staging area, allocated on first use).
static int rollout_env_args(ilsx_vecenv* e, ilsx_replay* rb, int max_path_length, int no_terminal, unsigned long long step, EnvStepArgs* out) {
  ilsx_ctx* ctx = e->ctx;
  EnvStepArgs& A = *out;
  memset(&A, 0, sizeof A);
  A.m = e->dm; A.qpos = e->qpos; A.qvel = e->qvel; A.n_env = e->n_env;
  A.ids = nullptr; A.n_ids = e->n_env; A.act = e->act;
  A.obs = e->nobs; A.rew = e->rew; A.done = e->done;
  A.obs_cur = e->obs_cur; A.auto_reset = 1; A.max_path_length = max_path_length;
  A.ep_len = e->ep_len; A.ep_ret = e->ep_ret; A.stats = e->stats;
  if (rb) { A.replay = rb->data; A.rec = rb->rec; A.cap = rb->cap; A.top = rb->top; }
  A.seed = e->seed; A.stream = e->rng_stream; A.step = step; A.no_terminal = no_terminal;
  if (rb && e->path_mode) {
    if (e->stage && (e->stage_rec != rb->rec || e->stage_len < max_path_length))
      ILSX_FAIL(ILSX_ERR_ARG, "path mode: staging holds %d-step episodes of %d-float records, asked for %d / %d", e->stage_len, e->stage_rec,
                max_path_length, rb->rec);
    if (!e->stage) {
      if (max_path_length < 1 || max_path_length >= rb->cap) ILSX_FAIL(ILSX_ERR_ARG, "path mode needs 1 <= max_path_length < replay capacity");
      ILSX_TRY(ctx_alloc(ctx, (size_t)e->n_env * max_path_length * rb->rec * 4, (void**)&e->stage));
      ILSX_TRY(ctx_alloc(ctx, (size_t)e->n_env * 4, (void**)&e->flush_len));
      e->stage_len = max_path_length; e->stage_rec = rb->rec;
      HIPCHK(hipHostMalloc((void**)&e->flush_host, (size_t)e->n_env * sizeof(int), hipHostMallocDefault));
    }
    A.stage = e->stage; A.stage_len = e->stage_len; A.flush_len = e->flush_len;
  }
  return ILSX_OK;
}
// path mode, second half of a step: wait for the step, move the episodes that ended in it from the staging area into the ring
// synced: the caller has already waited for the stream the step ran on; flags: where the step's episode-end flags were read back to (default: the env's own buffer)
static int rollout_paths_finish(ilsx_vecenv* e, bool synced = false, const int* flags = nullptr) {
  ilsx_replay* rb = e->paths_pending;
  if (!rb) return ILSX_OK;
  e->paths_pending = nullptr;
  if (!synced) HIPCHK(hipStreamSynchronize(e->ctx->stream));
  if (!flags) flags = e->flush_host;
  std::vector<int> envs, lens; std::vector<uint8_t> term;
  for (int i = 0; i < e->n_env; ++i)
    if (flags[i]) { envs.push_back(i); lens.push_back(flags[i] & ((1 << 30) - 1)); term.push_back((flags[i] >> 30) & 1); }
  return replay_insert_paths(rb, e->stage, e->stage_len, envs.data(), lens.data(), term.data(), (int)envs.size());
}
static int rollout_step_impl(ilsx_vecenv* e, ilsx_net* pi, ilsx_net* label_pi, int label_deterministic, ilsx_replay* rb,
                             int max_path_length, int random_actions, int deterministic, int no_terminal, bool defer_paths = false) {
  if (!e || (!pi && !random_actions)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_step: need a policy or random_actions");
  if (e->paths_pending) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_rollout_step: the previous step was begun with ilsx_rollout_step_begin and never ended (ilsx_rollout_step_end)");
  ilsx_ctx* ctx = e->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  if (rb && (rb->o != e->o || rb->a != e->a)) ILSX_FAIL(ILSX_ERR_ARG, "replay dims (%d,%d) != env dims (%d,%d)", rb->o, rb->a, e->o, e->a);
  if (rb && e->n_env > rb->cap) ILSX_FAIL(ILSX_ERR_ARG, "n_env exceeds the replay capacity");
  // the fused record holds the env's RAW observations while the policy acts on the normalised ones (policy_obs()): an
  // off-policy learner fed from this ring would train on a different observation scale than it acts on.  The reference
  // stores what step() returned, i.e. normalised observations (vecenvs.py:251-257); no shipped spec combines the two.
  if (rb && e->norm_obs)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "ilsx_rollout_step: norm_obs=1 with a replay ring is not supported (the fused insert records raw "
              "observations); use the on-policy rollout (ilsx_ppo_rollout) or turn norm_obs off");
  const unsigned long long step = ++e->step_ctr;
  if (random_actions) {
    const int tot = e->n_env * e->a;
    hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, e->act, e->n_env, e->seed,
                       e->rng_stream ^ 0x5A5A5A5Au, step, e->a);
    HIPCHK(hipGetLastError());
  } else {
    ILSX_TRY(ilsx_policy_act(pi, e->policy_obs(), e->n_env, deterministic, nullptr, e->act, nullptr));
  }
  EnvStepArgs A;
  ILSX_TRY(rollout_env_args(e, rb, max_path_length, no_terminal, step, &A));
  if (label_pi) {   // DAgger._handle_step (dagger.py:45-71): the stored action is the expert's for the observation acted on
    if (!e->act_label) ILSX_TRY(ctx_alloc(ctx, (size_t)e->n_env * e->a * 4, (void**)&e->act_label));
    ILSX_TRY(ilsx_policy_act(label_pi, e->policy_obs(), e->n_env, label_deterministic, nullptr, e->act_label, nullptr));
    A.rec_act = e->act_label;
  }
  const bool paths = rb && e->path_mode;
  ILSX_TRY(launch_env_step(e, A));
  if (paths) {
    HIPCHK(hipMemcpyAsync(e->flush_host, e->flush_len, (size_t)e->n_env * 4, hipMemcpyDeviceToHost, ctx->stream));
    e->paths_pending = rb;
    if (!defer_paths) ILSX_TRY(rollout_paths_finish(e));
  } else if (rb) {
    ILSX_TRY(replay_advance_device_rows(rb, e->n_env));
  }
  return env_after_step_norm(e);
}

extern "C" int ilsx_rollout_step(ilsx_vecenv* e, ilsx_net* pi, ilsx_replay* rb, int max_path_length, int random_actions,
                                 int deterministic, int no_terminal) {
  return rollout_step_impl(e, pi, nullptr, 0, rb, max_path_length, random_actions, deterministic, no_terminal);
}
// The same step in two halves, for a host that advances several runs side by side (each on its own ctx / stream): _begin only ENQUEUES
// (actions, physics, record, and in path mode the read-back of the episode-end flags), _end does what needs the host — in path mode it waits
// for the step and inserts the finished episodes (base_algorithm.py:509-519); otherwise nothing.  begin(all runs) ; end(all runs) lets the
// runs' launches overlap on the GPU instead of serialising on one host wait per run.
extern "C" int ilsx_rollout_step_begin(ilsx_vecenv* e, ilsx_net* pi, ilsx_replay* rb, int max_path_length, int random_actions,
                                       int deterministic, int no_terminal) {
  return rollout_step_impl(e, pi, nullptr, 0, rb, max_path_length, random_actions, deterministic, no_terminal, /*defer_paths=*/true);
}
extern "C" int ilsx_rollout_step_end(ilsx_vecenv* e) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_step_end: NULL env");
  HIPCHK(hipSetDevice(e->ctx->device));
  return rollout_paths_finish(e);
}
// n_steps lock-step sampling iterations of K runs (the inner loop of DeviceRLAlgorithmGroup between two train triggers): per iteration every
// run's step is enqueued on its own stream, then every run's host part is done — K x n_steps x 2 calls through the binding become one.
// Run k acts at random while its ring holds fewer than min_steps_before_training[k] samples (base_algorithm.py:186-188), exactly as the
// per-step form decides it.
// Can the K runs' steps go out as ONE launch per stage?  Planar steppers of one kernel instantiation with the same env count, plain
// observations, policies of one shape (the generic forward takes up to four tasks per launch), everything on one device.
static bool rollout_runs_groupable(ilsx_vecenv* const* envs, ilsx_net* const* pis, int n_runs) {
  static const bool off = getenv("ILSX_ROLLOUT_NO_GROUP") != nullptr || getenv("ILSX_ENV2D_LANE") != nullptr;
  if (off || n_runs < 2) return false;
  const ilsx_vecenv* e0 = envs[0];
  const ilsx_net* p0 = pis[0];
  if (e0->engine != 0 || e0->ctx->prof_on) return false;
  for (int k = 0; k < n_runs; ++k) {
    const ilsx_vecenv* e = envs[k];
    const ilsx_net* p = pis[k];
    if (e->engine != 0 || e->norm_obs || e->n_env != e0->n_env || e->hm.nb != e0->hm.nb || (e->hm.max_rows > 12) != (e0->hm.max_rows > 12) ||
        e->o != e0->o || e->a != e0->a || e->ctx->device != e0->ctx->device)
      return false;
    if (memcmp(&p->lay.cfg, &p0->lay.cfg, sizeof p->lay.cfg) || p->noise_policy != p0->noise_policy || p->out_linear != p0->out_linear ||
        (p->lay.cfg.n_heads != 2 && !p->noise_policy))
      return false;
  }
  return true;
}
template <int NB, int MR>
static int launch_envg_step_runs_t(ilsx_ctx* ctx, const EnvStepGroupArgs& G, int n, int n_env) {
  const size_t lds = (((sizeof(PlanarModelDev) + 7) / 8) + (size_t)EG_ENVS * EgOff<NB, MR>::TOTAL) * sizeof(double);
  hipLaunchKernelGGL((k_envg_step_runs<NB, MR>), dim3((n_env + EG_ENVS - 1) / EG_ENVS, n), dim3(64), lds, ctx->stream, G);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
// One lock-step iteration of K groupable runs on ONE stream (run 0's): the runs' actions — policy inference as launches of up to four tasks,
// each task on its run's Philox key and call counter, or uniform draws while a run is still filling its ring —, ONE stepper launch whose grid
// rows are the runs (k_envg_step_runs), the read-back of the episode-end flags with path mode on.  Per run the launches do exactly what its own
// ilsx_rollout_step would do: same arithmetic, same draws.
static int rollout_lockstep_grouped(ilsx_vecenv* const* envs, ilsx_net* const* pis, ilsx_replay* const* rbs, int n_runs, int max_path_length,
                                    const int64_t* min_steps, int deterministic, int no_terminal) {
  ilsx_ctx* gctx = envs[0]->ctx;
  hipStream_t gs = gctx->stream;
  std::vector<int> pol;   // runs that act through their policy this step
  for (int k = 0; k < n_runs; ++k) {
    ilsx_vecenv* e = envs[k];
    if (e->paths_pending) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_rollout_steps_lockstep: run %d has an unfinished step (ilsx_rollout_step_end)", k);
    if (rbs[k]->o != e->o || rbs[k]->a != e->a || e->n_env > rbs[k]->cap) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_steps_lockstep: replay %d does not match its env", k);
    if (rbs[k]->size < min_steps[k]) {
      const int tot = e->n_env * e->a;
      hipLaunchKernelGGL(k_random_actions, dim3((tot + 255) / 256), dim3(256), 0, gs, e->act, e->n_env, e->seed, e->rng_stream ^ 0x5A5A5A5Au,
                         e->step_ctr + 1, e->a);
    } else pol.push_back(k);
  }
  for (size_t i = 0; i < pol.size(); i += 4) {
    FwdArgs A;
    memset(&A, 0, sizeof A);
    const int nt = (int)std::min<size_t>(4, pol.size() - i);
    for (int j = 0; j < nt; ++j) {
      ilsx_vecenv* e = envs[pol[i + j]];
      ilsx_net* pi = pis[pol[i + j]];
      FwdTask& t = A.t[j];
      t.net = net_view(pi->lay, pi->base);
      t.x0 = e->policy_obs(); t.d0 = pi->lay.cfg.in_dim; t.s0 = pi->lay.cfg.in_dim;
      t.head = deterministic ? HEAD_TANH_DET : HEAD_TANH_SAMPLE;
      if (pi->noise_policy) {
        t.head = pi->out_linear ? HEAD_DET_LIN_NOISE : HEAD_DET_TANH_NOISE;
        t.noise = deterministic ? 0.0f : pi->noise; t.noise_clip = pi->noise_clip; t.max_act = pi->max_act;
      }
      t.action = e->act;
      t.rng_stream = 0x41435400u;  // 'ACT' (ilsx_policy_act)
      t.seed_t = pi->ctx->seed; t.step_t = ++pi->ctx->act_calls;   // (>= 1: the task's own key)
    }
    A.rows = envs[0]->n_env; A.ntasks = nt; A.seed = gctx->seed;
    ILSX_TRY(launch_fwd(gctx, A, pis[0]->lay.cfg.hidden, pis[0]->lay.cfg.act, pis[0]->lay.KP));
  }
  ilsx_vecenv* e0m = envs[0];
  bool all_paths = true;
  for (int k = 0; k < n_runs; ++k) all_paths = all_paths && envs[k]->path_mode;
  const int ne = e0m->n_env;
  if (all_paths && e0m->grp_flush_n < n_runs * ne) {   // the group's flag array (owned by run 0's env, freed with it)
    if (e0m->grp_flush_dev) ILSX_TRY(ctx_free(gctx, e0m->grp_flush_dev));
    if (e0m->grp_flush_host) HIPCHK(hipHostFree(e0m->grp_flush_host));
    e0m->grp_flush_dev = nullptr; e0m->grp_flush_host = nullptr; e0m->grp_flush_n = 0;
    ILSX_TRY(ctx_alloc(gctx, (size_t)n_runs * ne * sizeof(int), (void**)&e0m->grp_flush_dev));
    HIPCHK(hipHostMalloc((void**)&e0m->grp_flush_host, (size_t)n_runs * ne * sizeof(int), hipHostMallocDefault));
    e0m->grp_flush_n = n_runs * ne;
  }
  for (int k0 = 0; k0 < n_runs; k0 += ENVG_MAX_RUNS) {
    EnvStepGroupArgs G;
    const int n = std::min(ENVG_MAX_RUNS, n_runs - k0);
    for (int j = 0; j < n; ++j) {
      ilsx_vecenv* e = envs[k0 + j];
      ILSX_TRY(rollout_env_args(e, rbs[k0 + j], max_path_length, no_terminal, ++e->step_ctr, &G.a[j]));
      if (all_paths) G.a[j].flush_len = e0m->grp_flush_dev + (size_t)(k0 + j) * ne;
    }
    const ilsx_vecenv* e0 = envs[0];
    if (e0->hm.nb == 4) ILSX_TRY((launch_envg_step_runs_t<4, 8>(gctx, G, n, e0->n_env)));
    else if (e0->hm.max_rows > 12) ILSX_TRY((launch_envg_step_runs_t<7, 16>(gctx, G, n, e0->n_env)));
    else ILSX_TRY((launch_envg_step_runs_t<7, 12>(gctx, G, n, e0->n_env)));
  }
  bool any_paths = false;
  if (all_paths) {
    HIPCHK(hipMemcpyAsync(e0m->grp_flush_host, e0m->grp_flush_dev, (size_t)n_runs * ne * sizeof(int), hipMemcpyDeviceToHost, gs));
    for (int k = 0; k < n_runs; ++k) envs[k]->paths_pending = rbs[k];
    any_paths = true;
  } else {
    for (int k = 0; k < n_runs; ++k) {
      ilsx_vecenv* e = envs[k];
      if (e->path_mode) {
        HIPCHK(hipMemcpyAsync(e->flush_host, e->flush_len, (size_t)e->n_env * 4, hipMemcpyDeviceToHost, gs));
        e->paths_pending = rbs[k];
        any_paths = true;
      } else ILSX_TRY(replay_advance_device_rows(rbs[k], e->n_env));
    }
  }
  if (any_paths) {
    HIPCHK(hipStreamSynchronize(gs));
    for (int k = 0; k < n_runs; ++k)
      ILSX_TRY(rollout_paths_finish(envs[k], /*synced=*/true, all_paths ? e0m->grp_flush_host + (size_t)k * ne : nullptr));
  }
  return ILSX_OK;
}
// n_steps lock-step sampling iterations of K runs (the inner loop of DeviceRLAlgorithmGroup between two train triggers): K x n_steps x 2 calls
// through the binding become one.  Run k acts at random while its ring holds fewer than min_steps_before_training[k] samples
// (base_algorithm.py:186-188), exactly as the per-step form decides it.  Runs of one shape (rollout_runs_groupable: the reference's regime of a
// few envs per run and many seeds) go out as one launch per stage on run 0's stream; otherwise every run's step is enqueued on its own stream
// and then every run's host part is done.
extern "C" int ilsx_rollout_steps_lockstep(ilsx_vecenv* const* envs, ilsx_net* const* pis, ilsx_replay* const* rbs, int n_runs, int n_steps,
                                           int max_path_length, const int64_t* min_steps_before_training, int deterministic, int no_terminal) {
  if (!envs || !pis || !rbs || !min_steps_before_training || n_runs < 1 || n_steps < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_steps_lockstep: bad argument");
  for (int k = 0; k < n_runs; ++k)
    if (!envs[k] || !pis[k] || !rbs[k]) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_steps_lockstep: run %d has a NULL env / policy / replay", k);
  if (rollout_runs_groupable(envs, pis, n_runs)) {
    HIPCHK(hipSetDevice(envs[0]->ctx->device));
    // everything the runs' own streams still hold (an evaluation, a train call, a ring insert) happens before the grouped launches, and they
    // are complete when this call returns: host waits (an event wait would slow the stream's later graph launches, profiles/r06_grp_streams.txt)
    for (int k = 1; k < n_runs; ++k)
      if (envs[k]->ctx->stream != envs[0]->ctx->stream) HIPCHK(hipStreamSynchronize(envs[k]->ctx->stream));
    for (int t = 0; t < n_steps; ++t)
      ILSX_TRY(rollout_lockstep_grouped(envs, pis, rbs, n_runs, max_path_length, min_steps_before_training, deterministic, no_terminal));
    HIPCHK(hipStreamSynchronize(envs[0]->ctx->stream));
    return ILSX_OK;
  }
  for (int t = 0; t < n_steps; ++t) {
    for (int k = 0; k < n_runs; ++k)
      ILSX_TRY(rollout_step_impl(envs[k], pis[k], nullptr, 0, rbs[k], max_path_length, rbs[k]->size < min_steps_before_training[k] ? 1 : 0,
                                 deterministic, no_terminal, /*defer_paths=*/true));
    for (int k = 0; k < n_runs; ++k) {
      HIPCHK(hipSetDevice(envs[k]->ctx->device));
      ILSX_TRY(rollout_paths_finish(envs[k]));
    }
  }
  return ILSX_OK;
}
extern "C" int ilsx_rollout_step_relabel(ilsx_vecenv* e, ilsx_net* pi, ilsx_net* expert, int expert_deterministic, ilsx_replay* rb,
                                         int max_path_length, int no_terminal) {
  if (!expert) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_rollout_step_relabel: NULL expert policy");
  return rollout_step_impl(e, pi, expert, expert_deterministic, rb, max_path_length, 0, 0, no_terminal);
}

// ---- on-policy rollout into env-major buffers (sample (env, t) lives at row env*T + t)
__global__ void k_onpolicy_record_pre(const float* __restrict__ obs, const float* __restrict__ act, int n_env, int o, int a, int t,
                                      int T, float* __restrict__ obs_buf, float* __restrict__ act_buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, w = o + a;
  if (i >= n_env * w) return;
  const int env = i / w, j = i - env * w;
  const size_t row = (size_t)env * T + t;
  if (j < o) obs_buf[row * o + j] = obs[(size_t)env * o + j];
  else act_buf[row * a + (j - o)] = act[(size_t)env * a + (j - o)];
}
__global__ void k_onpolicy_record_post(const float* __restrict__ rew, const int* __restrict__ ep_len, int n_env, int t, int T,
                                       float* __restrict__ rew_buf, uint8_t* __restrict__ ends_buf) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n_env) return;
  const size_t row = (size_t)env * T + t;
  rew_buf[row] = rew[env];
  ends_buf[row] = ep_len[env] == 0 ? 1 : 0;   // the env was just reset: terminal or time limit
}

// PPO's sampling phase (torch_rl_algorithm.py:30-32 over base_algorithm.py:183-277) for ALL envs on the device: T vec steps
// with the Gaussian policy of `ppo`; env-major buffers obs[n_env*T,o] (what the policy saw: normalised when norm_obs),
// act[n_env*T,a] (un-clipped policy output; the env clips), rew[n_env*T], ends[n_env*T] (1 = episode ended after this
// sample).  last_values[n_env] (nullable) = V(observation after the last step) for bootstrapping unfinished segments.
extern "C" int ilsx_ppo_rollout(ilsx_ppo* ppo, ilsx_vecenv* e, int T, int max_path_length, float* obs, float* act, float* rew,
                                uint8_t* ends, float* last_values) {
  if (!ppo || !e || !obs || !act || !rew || !ends || T < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_rollout: bad argument");
  ilsx_ctx* ctx = e->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const int n = e->n_env, w = e->o + e->a;
  for (int t = 0; t < T; ++t) {
    const unsigned long long step = ++e->step_ctr;
    ILSX_TRY(ilsx_ppo_policy_act(ppo, e->policy_obs(), n, 0, nullptr, e->act, nullptr));
    hipLaunchKernelGGL(k_onpolicy_record_pre, dim3((n * w + 255) / 256), dim3(256), 0, ctx->stream, e->policy_obs(), (const float*)e->act,
                       n, e->o, e->a, t, T, obs, act);
    EnvStepArgs A;
    memset(&A, 0, sizeof A);
    A.m = e->dm; A.qpos = e->qpos; A.qvel = e->qvel; A.n_env = n;
    A.ids = nullptr; A.n_ids = n; A.act = e->act;
    A.obs = e->nobs; A.rew = e->rew; A.done = e->done;
    A.obs_cur = e->obs_cur; A.auto_reset = 1; A.max_path_length = max_path_length;
    A.ep_len = e->ep_len; A.ep_ret = e->ep_ret; A.stats = e->stats;
    A.seed = e->seed; A.stream = e->rng_stream; A.step = step;
    ILSX_TRY(launch_env_step(e, A));
    hipLaunchKernelGGL(k_onpolicy_record_post, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float*)e->rew,
                       (const int*)e->ep_len, n, t, T, rew, ends);
    HIPCHK(hipGetLastError());
    ILSX_TRY(env_after_step_norm(e));
  }
  if (last_values) ILSX_TRY(ilsx_ppo_values(ppo, e->policy_obs(), n, last_values));
  return ILSX_OK;
}

// ---- evaluation: VecPathSampler.obtain_samples / rollout (samplers/vec_sampler.py:5-93,124-142) on the device.
// Every env plays ONE episode (envs that end are frozen, not reset); the statistics of
// eval_util.get_generic_path_information (:15-80) are accumulated as sums / extrema.
enum { EV_PATHS = 0, EV_STEPS, EV_RET_S, EV_RET_SS, EV_RET_MAX, EV_RET_MIN, EV_LEN_S, EV_LEN_SS, EV_LEN_MAX, EV_LEN_MIN, EV_REW_S,
       EV_REW_SS, EV_REW_MAX, EV_REW_MIN, EV_ACT_S, EV_ACT_SS, EV_ACT_MAX, EV_ACT_MIN, EV_N };
__device__ __forceinline__ void atomic_max_d(double* p, double v) {
  unsigned long long old = __double_as_longlong(*p);
  while (v > __longlong_as_double(old)) {
    const unsigned long long prev = atomicCAS((unsigned long long*)p, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__device__ __forceinline__ void atomic_min_d(double* p, double v) {
  unsigned long long old = __double_as_longlong(*p);
  while (v < __longlong_as_double(old)) {
    const unsigned long long prev = atomicCAS((unsigned long long*)p, old, (unsigned long long)__double_as_longlong(v));
    if (prev == old) break;
    old = prev;
  }
}
__global__ void k_eval_begin(unsigned char* frozen, double* ret, int* len, int n, int* alive) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) { frozen[e] = 0; ret[e] = 0.0; len[e] = 0; }
  if (e == 0) *alive = n;
}
__global__ void k_eval_accum(const float* __restrict__ rew, const unsigned char* __restrict__ done, const float* __restrict__ act,
                             int n, int a, int max_path_length, unsigned char* frozen, double* ret, int* len, double* st, int* alive) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || frozen[e]) return;
  const double r = (double)rew[e];
  const double R = ret[e] + r;
  const int L = len[e] + 1;
  ret[e] = R; len[e] = L;
  atomicAdd(&st[EV_STEPS], 1.0);
  atomicAdd(&st[EV_REW_S], r); atomicAdd(&st[EV_REW_SS], r * r); atomic_max_d(&st[EV_REW_MAX], r); atomic_min_d(&st[EV_REW_MIN], r);
  for (int k = 0; k < a; ++k) {
    const double x = (double)act[(size_t)e * a + k];
    atomicAdd(&st[EV_ACT_S], x); atomicAdd(&st[EV_ACT_SS], x * x); atomic_max_d(&st[EV_ACT_MAX], x); atomic_min_d(&st[EV_ACT_MIN], x);
  }
  if (done[e] || L >= max_path_length) {   // the path ends at its first terminal or at the horizon (vec_sampler.py:61-77)
    frozen[e] = 1;
    atomicAdd(&st[EV_PATHS], 1.0);
    atomicAdd(&st[EV_RET_S], R); atomicAdd(&st[EV_RET_SS], R * R); atomic_max_d(&st[EV_RET_MAX], R); atomic_min_d(&st[EV_RET_MIN], R);
    const double l = (double)L;
    atomicAdd(&st[EV_LEN_S], l); atomicAdd(&st[EV_LEN_SS], l * l); atomic_max_d(&st[EV_LEN_MAX], l); atomic_min_d(&st[EV_LEN_MIN], l);
    atomicSub(alive, 1);
  }
}

extern "C" int ilsx_eval_rollout(ilsx_vecenv* e, ilsx_net* pi, ilsx_ppo* ppo, int max_path_length, int deterministic, int reset_stats,
                                 double* stats_host) {
  if (!e || (!pi && !ppo) || max_path_length < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_eval_rollout: bad argument");
  ilsx_ctx* ctx = e->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const int n = e->n_env;
  if (!e->ev_frozen) {
    ILSX_TRY(ctx_alloc(ctx, n, (void**)&e->ev_frozen)); ILSX_TRY(ctx_alloc(ctx, (size_t)n * 8, (void**)&e->ev_ret));
    ILSX_TRY(ctx_alloc(ctx, (size_t)n * 4, (void**)&e->ev_len)); ILSX_TRY(ctx_alloc(ctx, EV_N * 8, (void**)&e->ev_stats));
    ILSX_TRY(ctx_alloc(ctx, 4, (void**)&e->ev_alive));
    reset_stats = 1;
  }
  hipStream_t st = ctx->stream;
  if (reset_stats) {
    double h[EV_N];
    for (int i = 0; i < EV_N; ++i) h[i] = 0.0;
    h[EV_RET_MAX] = h[EV_LEN_MAX] = h[EV_REW_MAX] = h[EV_ACT_MAX] = -INFINITY;
    h[EV_RET_MIN] = h[EV_LEN_MIN] = h[EV_REW_MIN] = h[EV_ACT_MIN] = INFINITY;
    HIPCHK(hipMemcpyAsync(e->ev_stats, h, sizeof h, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  ILSX_TRY(ilsx_vecenv_reset(e, nullptr, n, nullptr));   // rollout() starts with env.reset(ready_env_ids) (vec_sampler.py:33)
  hipLaunchKernelGGL(k_eval_begin, dim3((n + 255) / 256), dim3(256), 0, st, e->ev_frozen, e->ev_ret, e->ev_len, n, e->ev_alive);
  for (int t = 0; t < max_path_length; ++t) {
    const unsigned long long step = ++e->step_ctr;
    if (pi) ILSX_TRY(ilsx_policy_act(pi, e->policy_obs(), n, deterministic, nullptr, e->act, nullptr));
    else ILSX_TRY(ilsx_ppo_policy_act(ppo, e->policy_obs(), n, deterministic, nullptr, e->act, nullptr));
    EnvStepArgs A;
    memset(&A, 0, sizeof A);
    A.m = e->dm; A.qpos = e->qpos; A.qvel = e->qvel; A.n_env = n;
    A.ids = nullptr; A.n_ids = n; A.act = e->act;
    A.obs = e->nobs; A.rew = e->rew; A.done = e->done; A.obs_cur = e->obs_cur; A.frozen = e->ev_frozen;
    A.seed = e->seed; A.stream = e->rng_stream; A.step = step;
    ILSX_TRY(launch_env_step(e, A));
    hipLaunchKernelGGL(k_eval_accum, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)e->rew, (const unsigned char*)e->done,
                       (const float*)e->act, n, e->a, max_path_length, e->ev_frozen, e->ev_ret, e->ev_len, e->ev_stats, e->ev_alive);
    HIPCHK(hipGetLastError());
    if (e->norm_obs) {   // step() hands out normalised observations and (training envs only) feeds the statistics
      ILSX_TRY(env_obs_norm(e, e->nobs, n, nullptr, nullptr, nullptr, 0));
      ILSX_TRY(env_obs_norm(e, nullptr, 0, nullptr, e->obs_cur, e->obs_n, n));
    }
    if ((t & 31) == 31) {   // every env done? (one 4-byte read-back per 32 vec steps)
      int alive = 0;
      HIPCHK(hipMemcpyAsync(&alive, e->ev_alive, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (alive <= 0) break;
    }
  }
  if (stats_host) {
    HIPCHK(hipMemcpyAsync(stats_host, e->ev_stats, EV_N * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  return ILSX_OK;
}

// ---- the evaluation rollouts of K runs in lock-step (DeviceRLAlgorithmGroup: K small eval envs, one per run)
struct EvalAccumRun { const float* rew; const unsigned char* done; const float* act; unsigned char* frozen; double* ret; int* len; double* st; int* alive; };
struct EvalAccumRuns { EvalAccumRun r[ENVG_MAX_RUNS]; int n, a, max_path_length; };
__global__ void k_eval_accum_runs(const EvalAccumRuns G) {   // blockIdx.y = run; a row of the grid is that run's k_eval_accum launch
  const EvalAccumRun& R = G.r[blockIdx.y];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G.n || R.frozen[e]) return;
  const double r = (double)R.rew[e];
  const double Rt = R.ret[e] + r;
  const int L = R.len[e] + 1;
  R.ret[e] = Rt; R.len[e] = L;
  double* st = R.st;
  atomicAdd(&st[EV_STEPS], 1.0);
  atomicAdd(&st[EV_REW_S], r); atomicAdd(&st[EV_REW_SS], r * r); atomic_max_d(&st[EV_REW_MAX], r); atomic_min_d(&st[EV_REW_MIN], r);
  for (int k = 0; k < G.a; ++k) {
    const double x = (double)R.act[(size_t)e * G.a + k];
    atomicAdd(&st[EV_ACT_S], x); atomicAdd(&st[EV_ACT_SS], x * x); atomic_max_d(&st[EV_ACT_MAX], x); atomic_min_d(&st[EV_ACT_MIN], x);
  }
  if (R.done[e] || L >= G.max_path_length) {
    R.frozen[e] = 1;
    atomicAdd(&st[EV_PATHS], 1.0);
    atomicAdd(&st[EV_RET_S], Rt); atomicAdd(&st[EV_RET_SS], Rt * Rt); atomic_max_d(&st[EV_RET_MAX], Rt); atomic_min_d(&st[EV_RET_MIN], Rt);
    const double l = (double)L;
    atomicAdd(&st[EV_LEN_S], l); atomicAdd(&st[EV_LEN_SS], l * l); atomic_max_d(&st[EV_LEN_MAX], l); atomic_min_d(&st[EV_LEN_MIN], l);
    atomicSub(R.alive, 1);
  }
}
static int eval_buffers(ilsx_vecenv* e) {
  if (e->ev_frozen) return ILSX_OK;
  ilsx_ctx* ctx = e->ctx;
  const int n = e->n_env;
  ILSX_TRY(ctx_alloc(ctx, n, (void**)&e->ev_frozen)); ILSX_TRY(ctx_alloc(ctx, (size_t)n * 8, (void**)&e->ev_ret));
  ILSX_TRY(ctx_alloc(ctx, (size_t)n * 4, (void**)&e->ev_len)); ILSX_TRY(ctx_alloc(ctx, EV_N * 8, (void**)&e->ev_stats));
  ILSX_TRY(ctx_alloc(ctx, 4, (void**)&e->ev_alive));
  return ILSX_OK;
}
// VecPathSampler.obtain_samples (vec_sampler.py:126-146) of K runs at once: every run rolls whole rounds (one episode per env, frozen at its
// end) until ITS statistics hold >= num_steps steps; the runs of one shape step in lock-step as one launch per stage on run 0's stream — a run
// leaves a round at the very check (every 32 vec steps) at which its own ilsx_eval_rollout would, so its policy-call counter, env step counter
// and statistics are those of evaluating it alone.  stats_host: [K][18] doubles (the layout ilsx_eval_rollout fills).  ILSX_ERR_UNSUPPORTED when
// the runs cannot share launches (the caller evaluates them one by one).
extern "C" int ilsx_eval_rollouts_lockstep(ilsx_vecenv* const* envs, ilsx_net* const* pis, int n_runs, int max_path_length, int deterministic,
                                           int64_t num_steps, double* stats_host) {
  if (!envs || !pis || !stats_host || n_runs < 1 || max_path_length < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_eval_rollouts_lockstep: bad argument");
  for (int k = 0; k < n_runs; ++k)
    if (!envs[k] || !pis[k]) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_eval_rollouts_lockstep: run %d has a NULL env / policy", k);
  if (n_runs > ENVG_MAX_RUNS || !rollout_runs_groupable(envs, pis, n_runs)) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "ilsx_eval_rollouts_lockstep: the runs cannot share launches");
  ilsx_vecenv* e0 = envs[0];
  ilsx_ctx* gctx = e0->ctx;
  hipStream_t gs = gctx->stream;
  HIPCHK(hipSetDevice(gctx->device));
  const int ne = e0->n_env;
  for (int k = 0; k < n_runs; ++k) {
    ILSX_TRY(eval_buffers(envs[k]));
    if (envs[k]->ctx->stream != gs) HIPCHK(hipStreamSynchronize(envs[k]->ctx->stream));
  }
  if (e0->grp_flush_n < n_runs) {   // the group's small int array (alive counters here, episode-end flags in the training rollouts)
    if (e0->grp_flush_dev) ILSX_TRY(ctx_free(gctx, e0->grp_flush_dev));
    if (e0->grp_flush_host) HIPCHK(hipHostFree(e0->grp_flush_host));
    e0->grp_flush_dev = nullptr; e0->grp_flush_host = nullptr; e0->grp_flush_n = 0;
    const int cnt = std::max(n_runs, n_runs * ne);
    ILSX_TRY(ctx_alloc(gctx, (size_t)cnt * sizeof(int), (void**)&e0->grp_flush_dev));
    HIPCHK(hipHostMalloc((void**)&e0->grp_flush_host, (size_t)cnt * sizeof(int), hipHostMallocDefault));
    e0->grp_flush_n = cnt;
  }
  int* alive_dev = e0->grp_flush_dev;
  int* alive_host = e0->grp_flush_host;
  std::vector<int> active(n_runs);
  for (int k = 0; k < n_runs; ++k) active[k] = k;
  {   // fresh statistics for every run
    double h[EV_N];
    for (int i = 0; i < EV_N; ++i) h[i] = 0.0;
    h[EV_RET_MAX] = h[EV_LEN_MAX] = h[EV_REW_MAX] = h[EV_ACT_MAX] = -INFINITY;
    h[EV_RET_MIN] = h[EV_LEN_MIN] = h[EV_REW_MIN] = h[EV_ACT_MIN] = INFINITY;
    for (int k = 0; k < n_runs; ++k) HIPCHK(hipMemcpyAsync(envs[k]->ev_stats, h, sizeof h, hipMemcpyHostToDevice, gs));
    HIPCHK(hipStreamSynchronize(gs));
  }
  while (!active.empty()) {
    for (int k : active) {   // rollout() starts with env.reset(ready_env_ids) (vec_sampler.py:33): on the run's own stream, waited for below
      ilsx_vecenv* e = envs[k];
      ILSX_TRY(ilsx_vecenv_reset(e, nullptr, e->n_env, nullptr));
      if (e->ctx->stream != gs) HIPCHK(hipStreamSynchronize(e->ctx->stream));
      hipLaunchKernelGGL(k_eval_begin, dim3((ne + 255) / 256), dim3(256), 0, gs, e->ev_frozen, e->ev_ret, e->ev_len, ne, alive_dev + k);
    }
    std::vector<int> live = active;
    for (int t = 0; t < max_path_length && !live.empty(); ++t) {
      for (size_t i = 0; i < live.size(); i += 4) {
        FwdArgs A;
        memset(&A, 0, sizeof A);
        const int nt = (int)std::min<size_t>(4, live.size() - i);
        for (int j = 0; j < nt; ++j) {
          ilsx_vecenv* e = envs[live[i + j]];
          ilsx_net* pi = pis[live[i + j]];
          FwdTask& ft = A.t[j];
          ft.net = net_view(pi->lay, pi->base);
          ft.x0 = e->policy_obs(); ft.d0 = pi->lay.cfg.in_dim; ft.s0 = pi->lay.cfg.in_dim;
          ft.head = deterministic ? HEAD_TANH_DET : HEAD_TANH_SAMPLE;
          if (pi->noise_policy) {
            ft.head = pi->out_linear ? HEAD_DET_LIN_NOISE : HEAD_DET_TANH_NOISE;
            ft.noise = deterministic ? 0.0f : pi->noise; ft.noise_clip = pi->noise_clip; ft.max_act = pi->max_act;
          }
          ft.action = e->act;
          ft.rng_stream = 0x41435400u;
          ft.seed_t = pi->ctx->seed; ft.step_t = ++pi->ctx->act_calls;
        }
        A.rows = ne; A.ntasks = nt; A.seed = gctx->seed;
        ILSX_TRY(launch_fwd(gctx, A, pis[0]->lay.cfg.hidden, pis[0]->lay.cfg.act, pis[0]->lay.KP));
      }
      EnvStepGroupArgs G;
      EvalAccumRuns Q;
      memset(&Q, 0, sizeof Q);
      Q.n = ne; Q.a = e0->a; Q.max_path_length = max_path_length;
      const int n = (int)live.size();
      for (int j = 0; j < n; ++j) {
        ilsx_vecenv* e = envs[live[j]];
        EnvStepArgs& A = G.a[j];
        memset(&A, 0, sizeof A);
        A.m = e->dm; A.qpos = e->qpos; A.qvel = e->qvel; A.n_env = ne;
        A.ids = nullptr; A.n_ids = ne; A.act = e->act;
        A.obs = e->nobs; A.rew = e->rew; A.done = e->done; A.obs_cur = e->obs_cur; A.frozen = e->ev_frozen;
        A.seed = e->seed; A.stream = e->rng_stream; A.step = ++e->step_ctr;
        Q.r[j] = EvalAccumRun{e->rew, e->done, e->act, e->ev_frozen, e->ev_ret, e->ev_len, e->ev_stats, alive_dev + live[j]};
      }
      if (e0->hm.nb == 4) ILSX_TRY((launch_envg_step_runs_t<4, 8>(gctx, G, n, ne)));
      else if (e0->hm.max_rows > 12) ILSX_TRY((launch_envg_step_runs_t<7, 16>(gctx, G, n, ne)));
      else ILSX_TRY((launch_envg_step_runs_t<7, 12>(gctx, G, n, ne)));
      hipLaunchKernelGGL(k_eval_accum_runs, dim3((ne + 255) / 256, n), dim3(256), 0, gs, Q);
      HIPCHK(hipGetLastError());
      if ((t & 31) == 31) {   // every env of a run done? (ilsx_eval_rollout's check, for all runs in one read-back)
        HIPCHK(hipMemcpyAsync(alive_host, alive_dev, (size_t)n_runs * sizeof(int), hipMemcpyDeviceToHost, gs));
        HIPCHK(hipStreamSynchronize(gs));
        std::vector<int> still;
        for (int k : live) if (alive_host[k] > 0) still.push_back(k);
        live.swap(still);
      }
    }
    for (int k : active) HIPCHK(hipMemcpyAsync(stats_host + (size_t)k * EV_N, envs[k]->ev_stats, EV_N * 8, hipMemcpyDeviceToHost, gs));
    HIPCHK(hipStreamSynchronize(gs));
    std::vector<int> more;
    for (int k : active) if (stats_host[(size_t)k * EV_N + EV_STEPS] < (double)num_steps) more.push_back(k);
    active.swap(more);
  }
  return ILSX_OK;
}

extern "C" int ilsx_rollout_stats(ilsx_vecenv* e, double* episodes, double* return_sum, int reset) {
  if (!e) ILSX_FAIL(ILSX_ERR_ARG, "env is NULL");
  HIPCHK(hipSetDevice(e->ctx->device));
  double h[4];
  HIPCHK(hipMemcpyAsync(h, e->stats, sizeof h, hipMemcpyDeviceToHost, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  if (episodes) *episodes = h[0];
  if (return_sum) *return_sum = h[1];
  if (reset) HIPCHK(hipMemsetAsync(e->stats, 0, sizeof h, e->ctx->stream));
  return ILSX_OK;
}
