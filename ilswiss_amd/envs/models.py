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

TASK_HOPPER, TASK_WALKER2D = 0, 1
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


def _body(name, parent, anchor_world, p1_world, p2_world, radius, friction, **joint):
    return dict(name=name, parent=parent, anchor_world=anchor_world, p1_world=p1_world, p2_world=p2_world,
                radius=radius, friction=friction, joint=joint)


def _finish(task, bodies, timestep, frame_skip, reset_noise, healthy, init_z, contact_solimp=(0.8, 0.8, 0.01)):
    """World-frame description at the zero pose -> parent-relative description the engine uses."""
    nb = len(bodies)
    m = dict(task=task, n_body=nb, parent=[], anchor=[], com=[], mass=[], inertia=[], jsign=[], armature=[],
             damping=[], limited=[], range=[], gear=[], geom_body=[], geom_p1=[], geom_p2=[], geom_radius=[],
             geom_friction=[], names=[b["name"] for b in bodies])
    for b in bodies:
        ax, az = b["anchor_world"]
        par = b["parent"]
        pax, paz = bodies[par]["anchor_world"] if par >= 0 else (0.0, 0.0)
        mass, inertia = capsule_mass_inertia(b["p1_world"], b["p2_world"], b["radius"])
        cx = 0.5 * (b["p1_world"][0] + b["p2_world"][0]) - ax
        cz = 0.5 * (b["p1_world"][1] + b["p2_world"][1]) - az
        j = b["joint"]
        m["parent"].append(par)
        m["anchor"].append((ax - pax, az - paz) if par >= 0 else (0.0, 0.0))
        m["com"].append((cx, cz))
        m["mass"].append(mass)
        m["inertia"].append(inertia)
        m["jsign"].append(1.0)
        m["armature"].append(j.get("armature", 0.0))
        m["damping"].append(j.get("damping", 0.0))
        m["limited"].append(1 if "range" in j else 0)
        m["range"].append(tuple(math.radians(v) for v in j.get("range", (0.0, 0.0))))
        m["gear"].append(j.get("gear", 0.0))
        m["geom_body"].append(len(m["parent"]) - 1)
        m["geom_p1"].append((b["p1_world"][0] - ax, b["p1_world"][1] - az))
        m["geom_p2"].append((b["p2_world"][0] - ax, b["p2_world"][1] - az))
        m["geom_radius"].append(b["radius"])
        m["geom_friction"].append(b["friction"])
    m["n_geom"] = nb
    m["act_bodies"] = [i for i in range(nb) if m["gear"][i] != 0.0]   # actuator order = body order
    m["init_qpos"] = [0.0, init_z, 0.0] + [0.0] * (nb - 1)
    m.update(timestep=timestep, frame_skip=frame_skip, gravity=9.81, reset_noise=reset_noise,
             # contact: geom margin 0.001 (+ floor 0.001), solref (0.02, 1), solimp (0.8, 0.8, 0.01)
             contact_margin=0.002, contact_solref=(0.02, 1.0), contact_solimp=contact_solimp,
             # joint limits: MuJoCo defaults solreflimit (0.02, 1), solimplimit (0.9, 0.95, 0.001)
             limit_solref=(0.02, 1.0), limit_solimp=(0.9, 0.95, 0.001), pgs_iters=30,
             ctrl_cost=1e-3, alive_bonus=1.0, healthy=healthy)
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


MODELS = {"hopper": hopper, "walker": walker2d, "walker2d": walker2d}
