"""3-D articulated-body models for the HIP vec-env stepper: Ant-v2 and Humanoid-v2 (plain data, no arithmetic beyond
mass properties).

The reference builds these envs from gym 0.22's MuJoCo XML models (rlkit/envs/envs_dict.py:6,9 ->
gym.envs.mujoco.ant:AntEnv, gym.envs.mujoco.humanoid:HumanoidEnv); neither gym nor MuJoCo nor the XML files exist in this
environment, so the constants below are authored from public knowledge of `ant.xml` / `humanoid.xml` and are UNVERIFIED
against MuJoCo (same status as envs/models.py).  Reward / termination / observation / reset rules are the in-tree ones:
rlkit/envs/mujoco/humanoid.py:24-73 (with the four observation blocks the local copy comments out restored, as gym's
Humanoid-v2 has them: 376 dims), rlkit/envs/mujoco/ant.py:11-43 (+ the 84 clipped contact-force dims of
Ant-v2: 111), rlkit/envs/terminals.py:94-117.

Engine conventions (oracle/spatial_env.py, csrc/ilsx_env3d.hip): link 0 is the root body on a free joint (qpos = position,
unit quaternion w x y z; qvel = linear velocity in the world frame, angular velocity in the BODY frame — MuJoCo's free-joint
convention); every other link hangs off its parent by ONE hinge.  A MuJoCo body with k hinges becomes a chain of k links, the
first k-1 massless, sharing the body's frame at the zero pose — MuJoCo applies a body's joints one after the other in the
frame the previous ones produced, which is exactly that chain.  A link's frame has its origin at its hinge anchor; `anchor`
is that point in the parent link's frame, `axis` the unit hinge axis (same components in parent and link frame at q = 0),
`quat0` the fixed rotation parent -> link at q = 0.
"""
import math

import numpy as np

TASK_ANT, TASK_HUMANOID = 3, 4


def _q_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def sphere_mi(c, r, density):
    m = density * 4.0 / 3.0 * math.pi * r ** 3
    return m, np.asarray(c, float), np.eye(3) * (0.4 * m * r * r)


def capsule_mi(p1, p2, r, density):
    """Solid capsule (cylinder + two hemispheres), exact mass / inertia about its centre, in the frame p1, p2 are given in."""
    p1, p2 = np.asarray(p1, float), np.asarray(p2, float)
    L = float(np.linalg.norm(p2 - p1))
    mc = density * math.pi * r * r * L
    # both end caps together counted as pi*r^3 (not 4/3*pi*r^3): MuJoCo 2.1's capsule mass is rho*pi*r^2*(L + r) — it reproduces
    # the body masses it reports for Humanoid-v2 (thigh 4.52556, shin 2.63249, torso 8.32208 kg) and Hopper-v2 (envs/models.py)
    ms = density * math.pi * r ** 3
    # axial / transverse moments (axis = capsule axis)
    i_ax = mc * r * r / 2.0 + ms * 2.0 * r * r / 5.0
    h = 3.0 * r / 8.0                                    # hemisphere centroid offset from its flat face
    i_tr_c = mc * (L * L / 12.0 + r * r / 4.0)
    # each hemisphere: about its own centroid (83/320 m r^2), shifted to the capsule centre (L/2 + h)
    i_tr_s = ms * (83.0 / 320.0 * r * r + (L / 2.0 + h) ** 2)
    i_tr = i_tr_c + i_tr_s
    if L < 1e-12:
        return mc + ms, 0.5 * (p1 + p2), np.eye(3) * (0.4 * ms * r * r)
    u = (p2 - p1) / L
    I = i_tr * (np.eye(3) - np.outer(u, u)) + i_ax * np.outer(u, u)
    return mc + ms, 0.5 * (p1 + p2), I


def _combine(parts):
    """Union of rigid parts (m, c, I about c) -> (m, c, I about the common COM)."""
    m = sum(p[0] for p in parts)
    c = sum(p[0] * p[1] for p in parts) / m
    I = np.zeros((3, 3))
    for pm, pc, pI in parts:
        d = pc - c
        I += pI + pm * (d @ d * np.eye(3) - np.outer(d, d))
    return m, c, I


def _unit(a):
    a = np.asarray(a, float)
    return a / np.linalg.norm(a)


class _Builder:
    """Bodies are described MuJoCo-style (position / joints / geoms in the body's own frame); emits the link list."""

    def __init__(self, density):
        self.density = density
        self.links = []          # dicts
        self.body_last_link = []  # MuJoCo body index (1-based, 0 = world) -> its last link
        self.geoms = []          # (link, p1 or centre, p2 or None, radius, friction)

    def body(self, name, parent_body, pos, joints, geoms, quat=(1.0, 0.0, 0.0, 0.0)):
        """parent_body: index returned by a previous body() call, or None for the root.  joints: list of
        dict(name, pos, axis, range (deg), armature, damping, stiffness, gear).  geoms: list of ("sphere", c, r) /
        ("capsule", p1, p2, r), each optionally followed by a friction value.  Returns this body's index."""
        pos = np.asarray(pos, float)
        parts = []
        for g in geoms:
            parts.append(sphere_mi(g[1], g[2], self.density) if g[0] == "sphere" else capsule_mi(g[1], g[2], g[3], self.density))
        mass, com, I = _combine(parts) if parts else (0.0, np.zeros(3), np.zeros((3, 3)))
        if parent_body is None:
            assert not self.links and not joints
            self.links.append(dict(name=name, parent=-1, anchor=np.zeros(3), axis=np.array([0.0, 0.0, 1.0]), quat0=np.array(quat, float),
                                   mass=mass, com=com, inertia=I, armature=0.0, damping=0.0, stiffness=0.0, limited=0, range=(0.0, 0.0),
                                   gear=0.0, joint=name + "_root", body_origin=np.zeros(3)))
            last = 0
            origin_in_link = np.zeros(3)
        else:
            pl = self.body_last_link[parent_body]
            p_origin = self.links[pl]["body_origin"]      # parent BODY origin expressed in the parent's last link frame
            assert joints, "a jointless child body is welded: merge its geoms into the parent"
            last = None
            for k, j in enumerate(joints):
                jp = np.asarray(j.get("pos", (0.0, 0.0, 0.0)), float)
                final = k == len(joints) - 1
                if k == 0:
                    par, anchor, q0 = pl, p_origin + pos + _q_to_R(quat) @ jp, np.array(quat, float)
                else:
                    prev_jp = np.asarray(joints[k - 1].get("pos", (0.0, 0.0, 0.0)), float)
                    par, anchor, q0 = last, jp - prev_jp, np.array([1.0, 0.0, 0.0, 0.0])
                # the body's own origin, seen from this link's frame (origin = this joint's anchor): -jp
                self.links.append(dict(name=name if final else f"{name}~{j['name']}", parent=par, anchor=anchor, axis=_unit(j["axis"]), quat0=q0,
                                       mass=mass if final else 0.0, com=(com - jp) if final else np.zeros(3),
                                       inertia=I if final else np.zeros((3, 3)), armature=j.get("armature", 0.0), damping=j.get("damping", 0.0),
                                       stiffness=j.get("stiffness", 0.0), limited=1 if "range" in j else 0,
                                       range=tuple(math.radians(v) for v in j.get("range", (0.0, 0.0))), gear=j.get("gear", 0.0),
                                       joint=j["name"], body_origin=-jp))
                last = len(self.links) - 1
            origin_in_link = self.links[last]["body_origin"]
        for g in geoms:
            fr = g[-1] if isinstance(g[-1], float) and ((g[0] == "sphere" and len(g) == 4) or (g[0] == "capsule" and len(g) == 5)) else None
            if g[0] == "sphere":
                self.geoms.append((last, origin_in_link + np.asarray(g[1], float), None, g[2], fr))
            else:
                self.geoms.append((last, origin_in_link + np.asarray(g[1], float), origin_in_link + np.asarray(g[2], float), g[3], fr))
        self.body_last_link.append(last)
        return len(self.body_last_link) - 1


def _finish(task, B, act_order, **kw):
    links = B.links
    nl = len(links)
    m = dict(task=task, n_link=nl, nv=6 + nl - 1, nq=7 + nl - 1, names=[l["name"] for l in links], joints=[l["joint"] for l in links],
             parent=[l["parent"] for l in links], anchor=[l["anchor"].tolist() for l in links], axis=[l["axis"].tolist() for l in links],
             quat0=[l["quat0"].tolist() for l in links], mass=[l["mass"] for l in links], com=[l["com"].tolist() for l in links],
             inertia=[l["inertia"].tolist() for l in links], armature=[l["armature"] for l in links], damping=[l["damping"] for l in links],
             stiffness=[l["stiffness"] for l in links], limited=[l["limited"] for l in links], range=[l["range"] for l in links],
             gear=[l["gear"] for l in links], body_link=[0] + [0] * 0)
    m["body_link"] = list(B.body_last_link)     # MuJoCo body b (0-based here, world excluded) -> link carrying its mass
    # contact spheres: (link, centre in link frame, radius, friction); capsules contribute their two end spheres
    cs = []
    for link, p1, p2, r, fr in B.geoms:
        f = kw["friction"] if fr is None else fr
        cs.append((link, p1.tolist(), r, f))
        if p2 is not None:
            cs.append((link, p2.tolist(), r, f))
    m["contact_link"] = [c[0] for c in cs]
    m["contact_pos"] = [c[1] for c in cs]
    m["contact_radius"] = [c[2] for c in cs]
    m["contact_friction"] = [c[3] for c in cs]
    m["n_contact"] = len(cs)
    jn = m["joints"]
    m["act_links"] = [jn.index(a) for a in act_order]      # actuator k drives this link's hinge
    m["act_dim"] = len(act_order)
    m.update(kw)
    m["init_qpos"] = [0.0, 0.0, kw["init_z"], 1.0, 0.0, 0.0, 0.0] + [0.0] * (nl - 1)
    nb = len(m["body_link"])
    m["obs_dim"] = (m["nq"] - 2) + m["nv"] + (((nb + 1) * 10 + (nb + 1) * 6 + m["nv"] + (nb + 1) * 6) if task == TASK_HUMANOID else (nb + 1) * 6)
    return m


def ant():
    """gym ant.xml (Ant-v2): sphere torso, four 2-segment legs; density 5, armature 1, damping 1, gear 150, friction 1, margin 0.01;
    RK4 at 0.01 s, frame_skip 5.  Body order (MuJoCo): torso, front_left_leg, aux_1, (ankle), front_right_leg, aux_2, ..., i.e.
    13 bodies + world = 14 (the cfrc_ext block of Ant-v2's observation is 14 x 6).  The jointless `*_leg` bodies are welded to the
    torso here (their capsules join the torso's mass); `body_link` still lists 13 bodies so that the observation keeps its shape."""
    B = _Builder(5.0)
    legs = [("front_left_leg", (0.2, 0.2, 0), "hip_1", "ankle_1", (-1, 1, 0), (30, 70), 1, 1),
            ("front_right_leg", (-0.2, 0.2, 0), "hip_2", "ankle_2", (1, 1, 0), (-70, -30), -1, 1),
            ("back_leg", (-0.2, -0.2, 0), "hip_3", "ankle_3", (-1, 1, 0), (-70, -30), -1, -1),
            ("right_back_leg", (0.2, -0.2, 0), "hip_4", "ankle_4", (1, 1, 0), (30, 70), 1, -1)]
    torso_geoms = [("sphere", (0, 0, 0), 0.25)] + [("capsule", (0, 0, 0), l[1], 0.08) for l in legs]
    t = B.body("torso", None, (0, 0, 0.75), [], torso_geoms)
    hj = dict(armature=1.0, damping=1.0, gear=150.0)
    body_map = [0]                       # MuJoCo body -> link, in MuJoCo's depth-first order
    for name, off, hip, ank, ank_axis, ank_rng, sx, sy in legs:
        body_map.append(0)               # the welded `*_leg` body moves with the torso
        aux = B.body("aux_" + hip[-1], t, off, [dict(name=hip, pos=(0, 0, 0), axis=(0, 0, 1), range=(-30, 30), **hj)],
                     [("capsule", (0, 0, 0), (0.2 * sx, 0.2 * sy, 0), 0.08)])
        body_map.append(B.body_last_link[aux])
        low = B.body("lower_" + hip[-1], aux, (0.2 * sx, 0.2 * sy, 0), [dict(name=ank, pos=(0, 0, 0), axis=ank_axis, range=ank_rng, **hj)],
                     [("capsule", (0, 0, 0), (0.4 * sx, 0.4 * sy, 0), 0.08)])
        body_map.append(B.body_last_link[low])
    m = _finish(TASK_ANT, B, ["hip_4", "ankle_4", "hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3"],
                timestep=0.01, frame_skip=5, gravity=9.81, friction=1.0, contact_margin=0.01, init_z=0.75, ctrl_range=1.0,
                reset_noise=0.1, reset_noise_vel_std=0.1, contact_solref=(0.02, 1.0), contact_solimp=(0.9, 0.95, 0.001),
                limit_solref=(0.02, 1.0), limit_solimp=(0.9, 0.95, 0.001), pgs_iters=30, max_rows=30,
                ctrl_cost=0.5, alive_bonus=1.0, vel_weight=1.0, z_min=0.2, z_max=1.0)
    m["body_link"] = body_map
    m["obs_dim"] = (m["nq"] - 2) + m["nv"] + 14 * 6
    return m


def humanoid():
    """gym humanoid.xml (Humanoid-v2): 13 bodies, 17 hinges, density 1000; RK4 at 0.003 s, frame_skip 5; motors ctrlrange +-0.4
    (NormalizedBoxEnv maps [-1, 1] onto it, wrappers.py:342-346) with gears 100 / 300 / 200 / 25; geoms are frictionless
    (condim 1) against a condim-3 floor of friction 1 -> contacts have friction 1."""
    B = _Builder(1000.0)
    J = lambda name, axis, rng, pos=(0, 0, 0), armature=1.0, damping=1.0, stiffness=0.0, gear=0.0: dict(  # noqa: E731
        name=name, axis=axis, range=rng, pos=pos, armature=armature, damping=damping, stiffness=stiffness, gear=gear)
    torso = B.body("torso", None, (0, 0, 1.4), [], [("capsule", (0, -.07, 0), (0, .07, 0), 0.07), ("sphere", (0, 0, .19), 0.09),
                                                     ("capsule", (-.01, -.06, -.12), (-.01, .06, -.12), 0.06)])
    tilt = (1.0, 0.0, -0.002, 0.0)
    lwaist = B.body("lwaist", torso, (-.01, 0, -0.260), [J("abdomen_z", (0, 0, 1), (-45, 45), (0, 0, 0.065), 0.02, 5, 20, 100),
                                                          J("abdomen_y", (0, 1, 0), (-75, 30), (0, 0, 0.065), 0.02, 5, 10, 100)],
                    [("capsule", (0, -.06, 0), (0, .06, 0), 0.06)], quat=tilt)
    pelvis = B.body("pelvis", lwaist, (0, 0, -0.165), [J("abdomen_x", (1, 0, 0), (-35, 35), (0, 0, 0.1), 0.02, 5, 10, 100)],
                    [("capsule", (-.02, -.07, 0), (-.02, .07, 0), 0.09)], quat=tilt)
    for side, sy in (("right", -1.0), ("left", 1.0)):
        sx = 1.0 if side == "right" else -1.0      # left hip x / z axes are mirrored (axis = -1 0 0 / 0 0 -1)
        thigh = B.body(f"{side}_thigh", pelvis, (0, 0.1 * sy, -0.04),
                       [J(f"{side}_hip_x", (sx, 0, 0), (-25, 5), (0, 0, 0), 0.01, 5, 10, 100),
                        J(f"{side}_hip_z", (0, 0, sx), (-60, 35), (0, 0, 0), 0.01, 5, 10, 100),
                        J(f"{side}_hip_y", (0, 1, 0), (-110, 20), (0, 0, 0), 0.008 if side == "right" else 0.01, 5, 20, 300)],
                       [("capsule", (0, 0, 0), (0, -0.01 * sy, -.34), 0.06)])
        shin = B.body(f"{side}_shin", thigh, (0, -0.01 * sy, -0.403),
                      [J(f"{side}_knee", (0, -1, 0), (-160, -2), (0, 0, .02), 0.006, 1.0, 0.0 if side == "right" else 1.0, 200)],
                      [("capsule", (0, 0, 0), (0, 0, -.3), 0.049), ("sphere", (0, 0, -0.35), 0.075)])   # the foot body is welded to the shin
    for side, sy in (("right", -1.0), ("left", 1.0)):
        s1 = (2, 1, 1) if side == "right" else (2, -1, 1)
        s2 = (0, -1, 1) if side == "right" else (0, 1, 1)
        rng = (-85, 60) if side == "right" else (-60, 85)
        ua = B.body(f"{side}_upper_arm", torso, (0, 0.17 * sy, 0.06),
                    [J(f"{side}_shoulder1", s1, rng, (0, 0, 0), 0.0068, 1.0, 1.0, 25), J(f"{side}_shoulder2", s2, rng, (0, 0, 0), 0.0051, 1.0, 1.0, 25)],
                    [("capsule", (0, 0, 0), (.16, .16 * sy, -.16), 0.04)])
        B.body(f"{side}_lower_arm", ua, (.18, .18 * sy, -.18),
               [J(f"{side}_elbow", (0, -1, 1) if side == "right" else (0, -1, -1), (-90, 50), (0, 0, 0), 0.0028, 1.0, 0.0, 25)],
               [("capsule", (.01, -.01 * sy, .01), (.17, -.17 * sy, .17), 0.031), ("sphere", (.18, -.18 * sy, .18), 0.04)])
    acts = ["abdomen_y", "abdomen_z", "abdomen_x", "right_hip_x", "right_hip_z", "right_hip_y", "right_knee", "left_hip_x", "left_hip_z",
            "left_hip_y", "left_knee", "right_shoulder1", "right_shoulder2", "right_elbow", "left_shoulder1", "left_shoulder2", "left_elbow"]
    m = _finish(TASK_HUMANOID, B, acts, timestep=0.003, frame_skip=5, gravity=9.81, friction=1.0, contact_margin=0.001, init_z=1.4,
                ctrl_range=0.4, reset_noise=0.01, reset_noise_vel_std=0.0, contact_solref=(0.02, 1.0), contact_solimp=(0.9, 0.95, 0.001),
                limit_solref=(0.02, 1.0), limit_solimp=(0.9, 0.95, 0.001), pgs_iters=30, max_rows=30,
                ctrl_cost=0.1, alive_bonus=5.0, vel_weight=0.25, z_min=1.0, z_max=2.0)
    # MuJoCo's 13 bodies in depth-first order; the welded feet report their shin's link
    last = {n: i for i, n in enumerate(m["names"])}
    order = ["torso", "lwaist", "pelvis", "right_thigh", "right_shin", "right_shin", "left_thigh", "left_shin", "left_shin",
             "right_upper_arm", "right_lower_arm", "left_upper_arm", "left_lower_arm"]
    m["body_link"] = [last[n] for n in order]
    m["obs_dim"] = (m["nq"] - 2) + m["nv"] + 14 * 10 + 14 * 6 + m["nv"] + 14 * 6
    return m


MODELS3D = {"ant": ant, "humanoid": humanoid}
