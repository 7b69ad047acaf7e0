"""CPU statement of the planar articulated-body stepper (the thing the HIP kernel k_env_step must
reproduce).  numpy float64, one env at a time, generic linear algebra — written independently of the
HIP code (which uses explicit recursions and a hand-rolled Cholesky) so that agreement means something.
Test infrastructure.

**Physics parity with the reference is UNPINNED**: the reference steps MuJoCo 2.1 through mujoco-py
(requirements.txt:12) on gym 0.22's XML models; none of that exists here.  What IS pinned from the
tree: reward / termination / observation / reset-noise formulas (rlkit/envs/mujoco/hopper.py:11-40,
walker2d.py:11-36), frame_skip (hopper.py:8), the NormalizedBoxEnv action map (wrappers.py:342-346), and
the README random-policy known answers (README.md:158-169) as a coarse behavioural check.

Model: body 0 has 3 DoF (x, z, pitch), each further body one hinge.  phi_b = absolute angle (CCW in x-z).
    M(q) qdd + c(q,qd) = tau + J^T f
    M = sum_b m_b Jc_b^T Jc_b + I_b Jphi_b^T Jphi_b + diag(armature)
    c = sum_b m_b Jc_b^T (acc_b(qdd=0) - g)
    tau = gear*ctrl - damping*qd - stiffness*q          (joint springs towards 0: half_cheetah.xml)
Constraints (MuJoCo-style soft constraints, solved by projected Gauss-Seidel in constraint space):
    rows: per capsule end within `contact_margin` of the floor: normal (f >= 0) + tangent (|f| <= mu f_n);
          per violated joint limit: one unilateral row.
    (A + R) f = aref - J qacc0,  A = J M^-1 J^T,  R_i = (1-d_i)/d_i A_ii,
    aref_i = -b v_i - k d_i r_i,  b = 2/(dmax tc),  k = 1/(dmax^2 tc^2 dr^2)      (solref = (tc, dr), solimp = (d0, dmax, width))
Integrator: classic RK4 on (q, qd) with the constraint solve inside every stage, `frame_skip` substeps.
"""
import numpy as np

TASK_HOPPER, TASK_WALKER2D, TASK_HALFCHEETAH = 0, 1, 2


def rot(phi):
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[c, -s], [s, c]])


def drot(phi):  # d rot / d phi
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[-s, -c], [c, -s]])


def impedance(r_abs, solimp):
    """MuJoCo default impedance curve (power 2, midpoint 0.5): d0 at r=0 rising to dmax at |r| >= width."""
    d0, dmax, width = solimp
    x = min(r_abs / width, 1.0) if width > 0 else 1.0
    y = 2.0 * x * x if x < 0.5 else 1.0 - 2.0 * (1.0 - x) ** 2
    return d0 + y * (dmax - d0)


class PlanarOracle:
    def __init__(self, model):
        self.m = model
        self.nb = model["n_body"]
        self.n = self.nb + 2

    # ---------------------------------------------------------------- kinematics
    def kin(self, q, v):
        m, nb, n = self.m, self.nb, self.n
        phi = np.zeros(nb); phid = np.zeros(nb)
        Jphi = np.zeros((nb, n)); Jo = np.zeros((nb, 2, n))
        o = np.zeros((nb, 2)); ao = np.zeros((nb, 2))
        for b in range(nb):
            p = m["parent"][b]
            if p < 0:
                phi[b] = m["jsign"][b] * q[2]
                Jphi[b, 2] = m["jsign"][b]
                o[b] = q[:2]
                Jo[b, 0, 0] = 1.0; Jo[b, 1, 1] = 1.0
            else:
                phi[b] = phi[p] + m["jsign"][b] * q[2 + b]
                Jphi[b] = Jphi[p]; Jphi[b, 2 + b] += m["jsign"][b]
                a = np.asarray(m["anchor"][b])
                o[b] = o[p] + rot(phi[p]) @ a
                Jo[b] = Jo[p] + np.outer(drot(phi[p]) @ a, Jphi[p])
                ao[b] = ao[p] - (Jphi[p] @ v) ** 2 * (rot(phi[p]) @ a)
            phid[b] = Jphi[b] @ v
        return phi, phid, Jphi, o, Jo, ao

    def dynamics(self, q, v, ctrl):
        """Returns qacc (constrained)."""
        m, nb, n = self.m, self.nb, self.n
        phi, phid, Jphi, o, Jo, ao = self.kin(q, v)
        M = np.zeros((n, n)); rhs = np.zeros(n)
        g = np.array([0.0, -m["gravity"]])
        for b in range(nb):
            r = np.asarray(m["com"][b])
            Jc = Jo[b] + np.outer(drot(phi[b]) @ r, Jphi[b])
            ac = ao[b] - phid[b] ** 2 * (rot(phi[b]) @ r)
            M += m["mass"][b] * Jc.T @ Jc + m["inertia"][b] * np.outer(Jphi[b], Jphi[b])
            rhs += m["mass"][b] * Jc.T @ (g - ac)
        for b in range(nb):
            M[2 + b, 2 + b] += m["armature"][b]
            rhs[2 + b] -= m["damping"][b] * v[2 + b] + m.get("stiffness", [0.0] * nb)[b] * q[2 + b]
        for k, b in enumerate(m["act_bodies"]):
            rhs[2 + b] += m["gear"][b] * ctrl[k]
        qacc0 = np.linalg.solve(M, rhs)

        # ---- constraint rows
        rows = []  # (J, r, kind, mu, partner)
        tc, dr = m["contact_solref"]
        max_rows = m.get("max_rows", 0) or (8 if nb == 4 else 12)  # rows: distal geoms first (p1, p2), then limits
        for gi in range(m["n_geom"] - 1, -1, -1):
            b = m["geom_body"][gi]
            for e in (m["geom_p1"][gi], m["geom_p2"][gi]):
                w = rot(phi[b]) @ np.asarray(e)
                rad = m["geom_radius"][gi]
                dist = o[b][1] + w[1] - rad
                if dist < m["contact_margin"] and len(rows) + 2 <= max_rows:
                    wc = w + np.array([0.0, -(rad + 0.5 * dist)])  # contact point, relative to the body origin
                    Jp = Jo[b] + np.outer(np.array([-wc[1], wc[0]]), Jphi[b])
                    mu = max(m["geom_friction"][gi], 0.0)
                    rows.append(dict(J=Jp[1], r=dist, kind="n", mu=mu, solref=(tc, dr), solimp=m["contact_solimp"]))
                    rows.append(dict(J=Jp[0], r=0.0, kind="t", mu=mu, solref=(tc, dr), solimp=m["contact_solimp"], rdist=dist))
        for b in range(1 if m["limited"][0] == 0 else 0, nb):
            if not m["limited"][b] or len(rows) + 1 > max_rows:
                continue
            lo, hi = m["range"][b]
            e = np.zeros(n); e[2 + b] = 1.0
            if q[2 + b] - lo < 0.0:
                rows.append(dict(J=e, r=q[2 + b] - lo, kind="l", mu=0.0, solref=m["limit_solref"], solimp=m["limit_solimp"]))
            elif hi - q[2 + b] < 0.0:
                rows.append(dict(J=-e, r=hi - q[2 + b], kind="l", mu=0.0, solref=m["limit_solref"], solimp=m["limit_solimp"]))
        if not rows:
            return qacc0
        J = np.array([r_["J"] for r_ in rows])
        MinvJT = np.linalg.solve(M, J.T)
        A = J @ MinvJT
        nr = len(rows)
        R = np.zeros(nr); rhs_c = np.zeros(nr)
        for i, r_ in enumerate(rows):
            tcs, drs = r_["solref"]
            d0, dmax, width = r_["solimp"]
            rr = r_["rdist"] if r_["kind"] == "t" else r_["r"]
            d = impedance(abs(rr), r_["solimp"])
            bdamp = 2.0 / (dmax * tcs)
            kstiff = 1.0 / (dmax * dmax * tcs * tcs * drs * drs)
            aref = -bdamp * (r_["J"] @ v) - kstiff * d * r_["r"]
            R[i] = (1.0 - d) / d * A[i, i]
            rhs_c[i] = aref - r_["J"] @ qacc0
        f = np.zeros(nr)
        for _ in range(m["pgs_iters"]):
            for i, r_ in enumerate(rows):
                res = rhs_c[i] - A[i] @ f + A[i, i] * f[i]
                fi = res / (A[i, i] + R[i])
                if r_["kind"] == "t":
                    lim = r_["mu"] * f[i - 1]
                    fi = min(max(fi, -lim), lim)
                else:
                    fi = max(fi, 0.0)
                f[i] = fi
        return qacc0 + MinvJT @ f

    # ---------------------------------------------------------------- integrator
    def substep(self, q, v, ctrl):
        h = self.m["timestep"]
        a1 = self.dynamics(q, v, ctrl)
        q2, v2 = q + 0.5 * h * v, v + 0.5 * h * a1
        a2 = self.dynamics(q2, v2, ctrl)
        q3, v3 = q + 0.5 * h * v2, v + 0.5 * h * a2
        a3 = self.dynamics(q3, v3, ctrl)
        q4, v4 = q + h * v3, v + h * a3
        a4 = self.dynamics(q4, v4, ctrl)
        return q + h / 6.0 * (v + 2 * v2 + 2 * v3 + v4), v + h / 6.0 * (a1 + 2 * a2 + 2 * a3 + a4)

    def obs(self, q, v):  # hopper.py:29-30 / gym v2: qpos[1:], clip(qvel, +-10); HalfCheetah-v2: no clip
        c = self.m.get("qvel_clip", 10.0)
        return np.concatenate([q[1:], np.clip(v, -c, c) if c > 0 else v])

    def step(self, q, v, action):
        """NormalizedBoxEnv action map (identity + clip for ctrlrange [-1,1], wrappers.py:342-346) ->
        frame_skip RK4 substeps -> reward, done (hopper.py:11-27 / walker2d.py:11-22)."""
        m = self.m
        a = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
        x0 = q[0]
        for _ in range(m["frame_skip"]):
            q, v = self.substep(q, v, a)
        dt = m["timestep"] * m["frame_skip"]
        reward = (q[0] - x0) / dt + m["alive_bonus"] - m["ctrl_cost"] * float(np.sum(a * a))
        hl = m["healthy"]
        s = np.concatenate([q, v])
        if m["task"] == TASK_HOPPER:
            ok = np.all(np.isfinite(s)) and np.all(np.abs(s[2:]) < hl["state"]) and q[1] > hl["z_min"] and abs(q[2]) < hl["ang"]
        elif m["task"] == TASK_WALKER2D:
            ok = hl["z_min"] < q[1] < hl["z_max"] and -hl["ang"] < q[2] < hl["ang"]
        else:   # HalfCheetahEnv.step: done = False
            ok = True
        return q, v, self.obs(q, v), reward, (not ok)

    def reset(self, rng):  # hopper.py:32-40: init + U(+-0.005) on qpos and qvel
        m, n = self.m, self.n
        nz = m["reset_noise"]
        q = np.asarray(m["init_qpos"], dtype=np.float64) + rng.uniform(-nz, nz, n)
        sd = m.get("reset_noise_vel_std", 0.0)   # HalfCheetahEnv.reset_model: qvel = 0.1 * randn
        v = sd * rng.standard_normal(n) if sd > 0 else rng.uniform(-nz, nz, n)
        return q, v

    def energy(self, q, v):
        """Kinetic + potential energy (for the conservation test; armature counts as rotor inertia)."""
        m, nb = self.m, self.nb
        phi, phid, Jphi, o, Jo, _ = self.kin(q, v)
        E = 0.0
        for b in range(nb):
            r = np.asarray(m["com"][b])
            Jc = Jo[b] + np.outer(drot(phi[b]) @ r, Jphi[b])
            vc = Jc @ v
            c = o[b] + rot(phi[b]) @ r
            E += 0.5 * m["mass"][b] * vc @ vc + 0.5 * m["inertia"][b] * phid[b] ** 2 + m["mass"][b] * m["gravity"] * c[1]
            E += 0.5 * m["armature"][b] * v[2 + b] ** 2 + 0.5 * m.get("stiffness", [0.0] * nb)[b] * q[2 + b] ** 2
        return E
