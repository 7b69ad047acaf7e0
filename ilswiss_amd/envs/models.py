"""Planar articulated-body models for the HIP vec-env stepper (plain data, no arithmetic).

The reference builds its envs from gym 0.22's MuJoCo XML models (rlkit/envs/envs_dict.py:5-12 ->
gym.envs.mujoco.hopper:HopperEnv etc.); neither gym nor MuJoCo nor the XML files exist in this
environment, so the constants below are authored from public knowledge of `hopper.xml` /
`walker2d.xml` and are UNVERIFIED against MuJoCo.  Reward / termination /
reset-noise formulas are the in-tree ones (rlkit/envs/mujoco/hopper.py:11-40, walker2d.py:11-36).

Conventions: planar (x forward, z up); body 0 is the root with 3 DoF (slide x, slide z, hinge);
every other body hangs off its parent by one hinge.  Angles are counter-clockwise in the (x, z)
plane, which is MuJoCo's positive rotation about the axis (0,-1,0) all these joints use.
Capsule mass / inertia follow MuJoCo's rule (uniform density 1000 kg/m^3, inertia from the geom).
"""
import math

TASK_HOPPER, TASK_WALKER2D, TASK_HALFCHEETAH = 0, 1, 2
DENSITY = 1000.0


def capsule_mass_inertia(p1, p2, r, density=DENSITY):
    """Solid capsule: cylinder + two hemispheres; inertia about the COM, axis normal to the plane."""
    L = math.hypot(p2[0] - p1[0], p2[1] - p1[1])
    m_c = density * math.pi * r * r * L
    # end caps counted as pi*r^3 (not 4/3*pi*r^3): reproduces the body masses MuJoCo 2.1 reports for
    # Hopper-v2 (3.5343, 3.9270, 2.7143, 5.0894 kg; total 15.265), i.e. m = rho*pi*r^2*(L + r)
    m_s = density * math.pi * r ** 3
    i_c = m_c * (L * L / 12.0 + r * r / 4.0)
    i_s = m_s * (2.0 * r * r / 5.0 + L * L / 4.0 + 3.0 * L * r / 8.0)
    return m_c + m_s, i_c + i_s


def _body(name, parent, anchor_world, p1_world, p2_world, radius, friction, extra_geoms=(), **joint):
    """extra_geoms: further capsules (p1_world, p2_world, radius, friction) rigidly attached to the same body."""
    return dict(name=name, parent=parent, anchor_world=anchor_world, p1_world=p1_world, p2_world=p2_world,
                radius=radius, friction=friction, extra_geoms=list(extra_geoms), joint=joint)


def _finish(task, bodies, timestep, frame_skip, reset_noise, healthy, init_z, contact_solimp=(0.8, 0.8, 0.01), jsign=1.0,
            ctrl_cost=1e-3, alive_bonus=1.0, reset_noise_vel_std=0.0, qvel_clip=10.0, max_rows=0, total_mass=None):
    """World-frame description at the zero pose -> parent-relative description the engine uses.  A body's mass, centre of
    mass and inertia are those of the union of its capsules (MuJoCo's rule for bodies without an explicit inertial)."""
    nb = len(bodies)
    m = dict(task=task, n_body=nb, parent=[], anchor=[], com=[], mass=[], inertia=[], jsign=[], armature=[],
             damping=[], stiffness=[], limited=[], range=[], gear=[], geom_body=[], geom_p1=[], geom_p2=[], geom_radius=[],
             geom_friction=[], names=[b["name"] for b in bodies])
    for bi, b in enumerate(bodies):
        ax, az = b["anchor_world"]
        par = b["parent"]
        pax, paz = bodies[par]["anchor_world"] if par >= 0 else (0.0, 0.0)
        geoms = [(b["p1_world"], b["p2_world"], b["radius"], b["friction"])] + list(b["extra_geoms"])
        parts = []
        for p1, p2, rad, _fr in geoms:
            gm, gi = capsule_mass_inertia(p1, p2, rad)
            parts.append((gm, gi, 0.5 * (p1[0] + p2[0]), 0.5 * (p1[1] + p2[1])))
        mass = sum(pt[0] for pt in parts)
        cwx = sum(pt[0] * pt[2] for pt in parts) / mass
        cwz = sum(pt[0] * pt[3] for pt in parts) / mass
        inertia = sum(pt[1] + pt[0] * ((pt[2] - cwx) ** 2 + (pt[3] - cwz) ** 2) for pt in parts)
        j = b["joint"]
        m["parent"].append(par)
        m["anchor"].append((ax - pax, az - paz) if par >= 0 else (0.0, 0.0))
        m["com"].append((cwx - ax, cwz - az))
        m["mass"].append(mass)
        m["inertia"].append(inertia)
        m["jsign"].append(jsign)
        m["armature"].append(j.get("armature", 0.0))
        m["damping"].append(j.get("damping", 0.0))
        m["stiffness"].append(j.get("stiffness", 0.0))
        m["limited"].append(1 if ("range" in j or "range_rad" in j) else 0)
        m["range"].append(tuple(j["range_rad"]) if "range_rad" in j else tuple(math.radians(v) for v in j.get("range", (0.0, 0.0))))
        m["gear"].append(j.get("gear", 0.0))
        for p1, p2, rad, fr in geoms:
            m["geom_body"].append(bi)
            m["geom_p1"].append((p1[0] - ax, p1[1] - az))
            m["geom_p2"].append((p2[0] - ax, p2[1] - az))
            m["geom_radius"].append(rad)
            m["geom_friction"].append(fr)
    if total_mass is not None:   # <compiler settotalmass=...>: masses and inertias scaled uniformly
        k = total_mass / sum(m["mass"])
        m["mass"], m["inertia"] = [v * k for v in m["mass"]], [v * k for v in m["inertia"]]
    m["n_geom"] = len(m["geom_body"])
    m["act_bodies"] = [i for i in range(nb) if m["gear"][i] != 0.0]   # actuator order = body order
    m["init_qpos"] = [0.0, init_z, 0.0] + [0.0] * (nb - 1)
    m.update(timestep=timestep, frame_skip=frame_skip, gravity=9.81, reset_noise=reset_noise,
             # contact: geom margin 0.001 (+ floor 0.001), solref (0.02, 1), solimp per model
             contact_margin=0.002, contact_solref=(0.02, 1.0), contact_solimp=contact_solimp,
             # joint limits: MuJoCo defaults solreflimit (0.02, 1), solimplimit (0.9, 0.95, 0.001)
             limit_solref=(0.02, 1.0), limit_solimp=(0.9, 0.95, 0.001), pgs_iters=30,
             ctrl_cost=ctrl_cost, alive_bonus=alive_bonus, healthy=healthy, reset_noise_vel_std=reset_noise_vel_std,
             qvel_clip=qvel_clip, max_rows=max_rows)
    m["obs_dim"] = 2 * (nb + 2) - 1
    m["act_dim"] = len(m["act_bodies"])
    return m


def hopper():
    """gym hopper.xml (Hopper-v2): torso / thigh / leg / foot, gear 200, dt = 4 x 0.002 (RK4)."""
    hj = dict(armature=1.0, damping=1.0)
    bodies = [
        _body("torso", -1, (0.0, 1.25), (0.0, 1.45), (0.0, 1.05), 0.05, 0.9),
        _body("thigh", 0, (0.0, 1.05), (0.0, 1.05), (0.0, 0.6), 0.05, 0.9, range=(-150.0, 0.0), gear=200.0, **hj),
        _body("leg", 1, (0.0, 0.6), (0.0, 0.6), (0.0, 0.1), 0.04, 0.9, range=(-150.0, 0.0), gear=200.0, **hj),
        _body("foot", 2, (0.0, 0.1), (-0.13, 0.1), (0.26, 0.1), 0.06, 2.0, range=(-45.0, 45.0), gear=200.0, **hj),
    ]
    # healthy (rlkit/envs/mujoco/hopper.py:19-25): |state[2:]| < 100, z > 0.7, |angle| < 0.2
    return _finish(TASK_HOPPER, bodies, 0.002, 4, 0.005, dict(z_min=0.7, z_max=1e30, ang=0.2, state=100.0), 1.25)


def walker2d():
    """gym walker2d.xml (Walker2d-v2): torso + two (thigh, leg, foot) legs, gear 100."""
    hj = dict(armature=0.01, damping=0.1)
    leg = lambda side, fr: [  # noqa: E731
        _body(f"thigh{side}", 0, (0.0, 1.05), (0.0, 1.05), (0.0, 0.6), 0.05, 0.7, range=(-150.0, 0.0), gear=100.0, **hj),
        _body(f"leg{side}", None, (0.0, 0.6), (0.0, 0.6), (0.0, 0.1), 0.04, 0.7, range=(-150.0, 0.0), gear=100.0, **hj),
        _body(f"foot{side}", None, (0.0, 0.1), (0.0, 0.1), (0.2, 0.1), 0.06, fr, range=(-45.0, 45.0), gear=100.0, **hj)]
    bodies = [_body("torso", -1, (0.0, 1.25), (0.0, 1.45), (0.0, 1.05), 0.05, 0.7)]
    for side, fr in (("", 0.9), ("_left", 1.9)):
        base = len(bodies)
        lg = leg(side, fr)
        lg[1]["parent"], lg[2]["parent"] = base, base + 1
        bodies += lg
    # walker2d.py:17-20: done = not (0.8 < z < 2.0 and -1 < angle < 1)
    # walker2d.xml overrides neither solref nor solimp: MuJoCo defaults (0.02, 1), (0.9, 0.95, 0.001)
    return _finish(TASK_WALKER2D, bodies, 0.002, 4, 0.005, dict(z_min=0.8, z_max=2.0, ang=1.0, state=1e30), 1.25,
                   contact_solimp=(0.9, 0.95, 0.001))


def _capsule(center, angle, half_len):
    """MuJoCo capsule given by its centre, an axisangle about +y and a half length: the long axis is z rotated by `angle`."""
    dx, dz = math.sin(angle) * half_len, math.cos(angle) * half_len
    return (center[0] - dx, center[1] - dz), (center[0] + dx, center[1] + dz)


def halfcheetah():
    """gym half_cheetah.xml (HalfCheetah-v2): torso (+ head) and two 3-segment legs; joints about +y (jsign -1 in this file's
    CCW convention) with springs (stiffness) and dampers, armature 0.1, gears 120/90/60/120/60/30, friction 0.4, contact
    solimp (0, 0.8, 0.01), settotalmass 14 (body masses 6.36 / 1.54 / 1.58 / 1.07 / 1.43 / 1.18 / 0.85 kg as MuJoCo reports);
    dt = 5 x 0.01 (RK4).  HalfCheetahEnv: reward = forward velocity - 0.1*|a|^2, never done,
    reset qpos + U(+-0.1), qvel = 0.1*randn, observation qpos[1:] | qvel without clipping."""
    z0, r, fr = 0.7, 0.046, 0.4
    J = lambda lo, hi, damping, stiffness, gear: dict(range_rad=(lo, hi), damping=damping, stiffness=stiffness, gear=gear, armature=0.1)  # noqa: E731
    head = _capsule((0.6, z0 + 0.1), 0.87, 0.15)
    A = {}
    A["bthigh"] = (-0.5, z0)
    A["bshin"] = (A["bthigh"][0] + 0.16, A["bthigh"][1] - 0.25)
    A["bfoot"] = (A["bshin"][0] - 0.28, A["bshin"][1] - 0.14)
    A["fthigh"] = (0.5, z0)
    A["fshin"] = (A["fthigh"][0] - 0.14, A["fthigh"][1] - 0.24)
    A["ffoot"] = (A["fshin"][0] + 0.13, A["fshin"][1] - 0.18)
    seg = lambda name, off, ang, hl: _capsule((A[name][0] + off[0], A[name][1] + off[1]), ang, hl)  # noqa: E731
    g = dict(bthigh=seg("bthigh", (0.1, -0.13), -3.8, 0.145), bshin=seg("bshin", (-0.14, -0.07), -2.03, 0.15),
             bfoot=seg("bfoot", (0.03, -0.097), -0.27, 0.094), fthigh=seg("fthigh", (-0.07, -0.12), 0.52, 0.133),
             fshin=seg("fshin", (0.065, -0.09), -0.6, 0.106), ffoot=seg("ffoot", (0.045, -0.07), -0.6, 0.07))
    bodies = [
        _body("torso", -1, (0.0, z0), (-0.5, z0), (0.5, z0), r, fr, extra_geoms=[(head[0], head[1], r, fr)]),
        _body("bthigh", 0, A["bthigh"], *g["bthigh"], r, fr, **J(-0.52, 1.05, 6.0, 240.0, 120.0)),
        _body("bshin", 1, A["bshin"], *g["bshin"], r, fr, **J(-0.785, 0.785, 4.5, 180.0, 90.0)),
        _body("bfoot", 2, A["bfoot"], *g["bfoot"], r, fr, **J(-0.4, 0.785, 3.0, 120.0, 60.0)),
        _body("fthigh", 0, A["fthigh"], *g["fthigh"], r, fr, **J(-1.0, 0.7, 4.5, 180.0, 120.0)),
        _body("fshin", 4, A["fshin"], *g["fshin"], r, fr, **J(-1.2, 0.87, 3.0, 120.0, 60.0)),
        _body("ffoot", 5, A["ffoot"], *g["ffoot"], r, fr, **J(-0.5, 0.5, 1.5, 60.0, 30.0)),
    ]
    return _finish(TASK_HALFCHEETAH, bodies, 0.01, 5, 0.1, dict(z_min=-1e30, z_max=1e30, ang=1e30, state=1e30), z0,
                   contact_solimp=(0.0, 0.8, 0.01), jsign=-1.0, ctrl_cost=0.1, alive_bonus=0.0, reset_noise_vel_std=0.1,
                   qvel_clip=0.0, max_rows=16, total_mass=14.0)


MODELS = {"hopper": hopper, "walker": walker2d, "walker2d": walker2d, "halfcheetah": halfcheetah, "half_cheetah": halfcheetah}
