"""Cheetah skeleton definition: the host-side mirror of cheetah.py:109-200 (`model`), :19-106
(`define_leg`), :203-356 (`add_pyomo_constraints`) and acinoset_misc.py:1581-1659 (`get_pose_state`),
expressed as plain tables for the C ABI (include/cpe.h `cpe_skeleton`).

Geometry conventions are the ones of SURVEY.md Appendix A.2/A.3/A.6: every link carries absolute ZYX
Euler angles; "-x" links (base, bodyF, neck) extend along -x, "+x" links (tails) along +x, "-z" links
(legs) along -z of their own frame.
"""
import json
import math
import os

import numpy as np

from . import abi

LINKS = ["base", "bodyF", "neck", "tail0", "tail1", "UFL", "LFL", "HFL", "UFR", "LFR", "HFR",
         "UBL", "LBL", "UBR", "LBR", "HBL", "HBR"]            # cheetah.py:197-198
PARENT = {"base": None, "bodyF": "base", "neck": "bodyF", "tail0": "base", "tail1": "tail0",
          "UFL": "bodyF", "LFL": "UFL", "HFL": "LFL", "UFR": "bodyF", "LFR": "UFR", "HFR": "LFR",
          "UBL": "base", "LBL": "UBL", "HBL": "LBL", "UBR": "base", "LBR": "UBR", "HBR": "LBR"}
MARKERS = ["nose", "r_eye", "l_eye", "neck_base", "spine", "tail_base", "tail1", "tail2",
           "r_shoulder", "r_front_knee", "r_front_ankle", "r_front_paw",
           "l_shoulder", "l_front_knee", "l_front_ankle", "l_front_paw",
           "r_hip", "r_back_knee", "r_back_ankle", "r_back_paw",
           "l_hip", "l_back_knee", "l_back_ankle", "l_back_paw"]  # acinoset_misc.py:1914-1940
# DLC column of each marker (acinoset_misc.py:1943-1969)
DLC_INDEX = {"nose": 23, "r_eye": 0, "l_eye": 1, "neck_base": 24, "spine": 6, "tail_base": 22,
             "tail1": 11, "tail2": 12, "l_shoulder": 13, "l_front_knee": 14, "l_front_ankle": 15,
             "l_front_paw": 16, "r_shoulder": 2, "r_front_knee": 3, "r_front_ankle": 4,
             "r_front_paw": 5, "l_hip": 17, "l_back_knee": 18, "l_back_ankle": 19, "l_back_paw": 20,
             "r_hip": 7, "r_back_knee": 8, "r_back_ankle": 9, "r_back_paw": 10}

# measurement standard deviations R (acinoset_misc.py:1762-1790); weight = 1/(2 R) (:1850, :231)
R_MEAS = np.array([1.2, 1.24, 1.18, 2.08, 2.04, 2.52, 2.73, 1.83, 3.47, 2.75, 2.69, 2.24, 3.4, 2.91,
                   2.85, 2.27, 3.26, 2.76, 2.33, 2.4, 3.53, 2.69, 2.49, 2.34])
# constant-acceleration model standard deviations Q (acinoset_misc.py:1852-1907)
Q_MODEL = np.array([4, 7, 5, 13, 9, 26, 10, 53, 34, 32, 18, 12, 0, 90, 43, 0, 118, 51, 0, 247, 0, 0,
                    186, 0, 0, 91, 0, 0, 194, 0, 0, 164, 0, 0, 91, 0, 0, 295, 0, 0, 243, 0, 0, 334, 0,
                    0, 149, 0, 0, 132, 0, 0, 132, 0], dtype=float)

_PARAMS_JSON = os.path.join(os.path.dirname(__file__), "data", "skeleton_params.json")
PHI, THETA, PSI = 0, 1, 2


def dof(link: str, angle: int) -> int:
    return 3 + 3 * LINKS.index(link) + angle


def load_params(animal: str) -> dict:
    """Link mass/length/radius exported from cheetah_params.py (tools/export_skeleton_params.py).
    Animal naming follows acinoset_opt.py:455-457."""
    with open(_PARAMS_JSON) as f:
        allp = json.load(f)
    if animal.endswith("-02"):
        animal = animal[:-3]
    if animal not in allp:
        animal = "acinoset"
    return allp[animal]


def _link_param(p: dict, name: str) -> dict:
    if name == "base":
        return p["body_B"]
    if name == "bodyF":
        return p["body_F"]
    if name in ("neck", "tail0", "tail1"):
        return p[name]
    seg = {"U": "thigh", "L": "calf", "H": "hock"}[name[0]]
    return p["front" if name[1] == "F" else "back"][seg]


def measurement_sigma(n_markers: int = 24, kinetic_dataset: bool = False) -> np.ndarray:
    """R_pw[0] of get_uncertainty_models(): 2*R (acinoset_misc.py:1850); 7 for the kinetic dataset
    (:187-188).  An extra 25th marker (benchmark only) reuses the `spine` value."""
    R = 2.0 * R_MEAS
    if kinetic_dataset:
        R = np.full(24, 7.0)
    if n_markers > 24:
        R = np.concatenate([R, np.full(n_markers - 24, R[4])])
    return R[:n_markers]


def build_skeleton(animal: str = "phantom", n_markers: int = 24, kinetic_dataset: bool = False) -> abi.Skeleton:
    p = load_params(animal)
    L = {n: _link_param(p, n)["length"] for n in LINKS}
    r = {n: _link_param(p, n)["radius"] for n in LINKS}
    m = {n: _link_param(p, n)["mass"] for n in LINKS}
    sk = abi.Skeleton()
    sk.n_links = len(LINKS)
    # link-local end points
    bottom, top, com = {}, {}, {}
    for n in LINKS:
        if n == "base":
            top[n], bottom[n], com[n] = (L[n] / 2, 0, 0), (-L[n] / 2, 0, 0), (0, 0, 0)
        elif n in ("bodyF", "neck"):
            top[n], bottom[n], com[n] = (0, 0, 0), (-L[n], 0, 0), (-L[n] / 2, 0, 0)
        elif n in ("tail0", "tail1"):
            top[n], bottom[n], com[n] = (0, 0, 0), (L[n], 0, 0), (L[n] / 2, 0, 0)
        else:
            top[n], bottom[n], com[n] = (0, 0, 0), (0, 0, -L[n]), (0, 0, -L[n] / 2)
    attach = {"base": (0, 0, 0), "bodyF": bottom["base"], "neck": bottom["bodyF"], "tail0": top["base"],
              "tail1": bottom["tail0"]}
    for side, sy in (("L", -1.0), ("R", 1.0)):
        # cheetah.py:32-38: start = body.Pb_I + Rb_I [ -+L/2, +-r, 0 ]; Pb_I of bodyF is its own com point
        attach["UF" + side] = (com["bodyF"][0] - L["bodyF"] / 2, sy * r["bodyF"], 0)
        attach["UB" + side] = (L["base"] / 2, sy * r["base"], 0)
        for fb in "FB":
            attach["L" + fb + side] = bottom["U" + fb + side]
            attach["H" + fb + side] = bottom["L" + fb + side]
    for i, n in enumerate(LINKS):
        sk.parent[i] = -1 if PARENT[n] is None else LINKS.index(PARENT[n])
        for d in range(3):
            sk.attach[i][d] = attach[n][d]
            sk.com[i][d] = com[n][d]
        sk.mass[i] = m[n]

    def add(a, b):
        return tuple(x + y for x, y in zip(a, b))

    # marker = (link, body-frame offset)   acinoset_misc.py:1586-1659
    mk = {
        "nose": ("neck", add(bottom["neck"], (-0.055, 0, -0.055))),
        "r_eye": ("neck", add(bottom["neck"], (0, 0.045, 0))),
        "l_eye": ("neck", add(bottom["neck"], (0, -0.045, 0))),
        "neck_base": ("bodyF", bottom["bodyF"]),                 # neck.top
        "spine": ("base", bottom["base"]),
        "tail_base": ("base", top["base"]),
        "tail1": ("tail0", bottom["tail0"]),                     # tail1.top
        "tail2": ("tail1", bottom["tail1"]),
        "r_shoulder": ("bodyF", add(bottom["bodyF"], (0.06, 0.075, -0.15))),
        "l_shoulder": ("bodyF", add(bottom["bodyF"], (0.06, -0.075, -0.15))),
        "r_hip": ("base", add(top["base"], (-0.06, 0.06, -0.1))),
        "l_hip": ("base", add(top["base"], (-0.06, -0.06, -0.1))),
    }
    for side, s in (("r", "R"), ("l", "L")):
        for fb, FB in (("front", "F"), ("back", "B")):
            mk[f"{side}_{fb}_knee"] = ("U" + FB + s, bottom["U" + FB + s])
            mk[f"{side}_{fb}_ankle"] = ("L" + FB + s, bottom["L" + FB + s])   # hock.top
            mk[f"{side}_{fb}_paw"] = ("H" + FB + s, bottom["H" + FB + s])
    names = list(MARKERS)
    if n_markers > 24:
        for e in range(n_markers - 24):
            names.append(f"extra{e}")
            mk[f"extra{e}"] = ("base", (0, 0, 0))   # benchmark-only 25th marker at the base COM
    sk.n_markers = n_markers
    for l, n in enumerate(names[:n_markers]):
        sk.marker_link[l] = LINKS.index(mk[n][0])
        for d in range(3):
            sk.marker_off[l][d] = mk[n][1][d]

    # joints (cheetah.py:71-72,101,160-161), parents before children
    joints = [("base", "UBL", 0), ("base", "UBR", 0), ("bodyF", "UFL", 0), ("bodyF", "UFR", 0),
              ("UFL", "LFL", 0), ("LFL", "HFL", 0), ("UFR", "LFR", 0), ("LFR", "HFR", 0),
              ("UBL", "LBL", 0), ("LBL", "HBL", 0), ("UBR", "LBR", 0), ("LBR", "HBR", 0),
              ("base", "tail0", 1), ("tail0", "tail1", 1)]
    sk.n_joints = len(joints)
    for j, (a, b, k) in enumerate(joints):
        sk.joint_parent[j], sk.joint_child[j], sk.joint_kind[j] = LINKS.index(a), LINKS.index(b), k

    # bounds lo <= q_a - q_b <= up (cheetah.py:306-352)
    pi = math.pi
    B = []
    if kinetic_dataset:
        B += [("neck", PSI, "bodyF", -0.05, 0.05), ("neck", PHI, "bodyF", -0.05, 0.05),
              ("base", PHI, None, -0.05, 0.05), ("bodyF", PSI, "base", -0.1, 0.1),
              ("bodyF", PHI, "base", -0.1, 0.1), ("base", PSI, "tail0", -0.1, 0.1)]
    else:
        B += [("neck", PSI, "bodyF", -pi / 6, pi / 6), ("neck", PHI, "bodyF", -pi / 6, pi / 6),
              ("base", PHI, None, -pi / 6, pi / 6), ("bodyF", PSI, "base", -pi / 6, pi / 6),
              ("bodyF", PHI, "base", -pi / 6, pi / 6), ("base", PSI, "tail0", -pi / 1.5, pi / 1.5)]
    B += [("neck", THETA, "bodyF", -pi / 6, pi / 6), ("bodyF", THETA, "base", -pi / 6, pi / 6),
          ("base", THETA, "tail0", -pi / 1.5, pi / 1.5), ("tail0", THETA, "tail1", -pi / 1.5, pi / 1.5),
          ("tail0", PSI, "tail1", -pi / 1.5, pi / 1.5)]
    for body, thigh, calf, hock, back in (("bodyF", "UFL", "LFL", "HFL", False), ("bodyF", "UFR", "LFR", "HFR", False),
                                          ("base", "UBL", "LBL", "HBL", True), ("base", "UBR", "LBR", "HBR", True)):
        B.append((body, THETA, thigh, -0.75 * pi, 0.75 * pi))
        B.append((thigh, THETA, calf, *((0.0, pi) if back else (-pi, 0.0))))
        B.append((calf, THETA, hock, *((-0.75 * pi, 0.0) if back else (-pi / 4, 0.75 * pi))))
    sk.n_bounds = len(B)
    for i, (a, ang, b, lo, up) in enumerate(B):
        sk.bound_a[i] = dof(a, ang)
        sk.bound_b[i] = -1 if b is None else dof(b, ang)
        sk.bound_lo[i], sk.bound_up[i] = lo, up

    for pidx in range(sk.nq):
        sk.motion_w[pidx] = 0.0 if Q_MODEL[pidx] == 0 else 1.0 / Q_MODEL[pidx] ** 2
        sk.rel_ref[pidx] = -1
        sk.rel_sign[pidx] = 1.0
    # relative angles (acinoset_misc.py:510-526)
    for n in LINKS[1:]:
        par = PARENT[n]
        for a in range(3):
            sk.rel_ref[dof(n, a)] = dof(par, a)
            sk.rel_sign[dof(n, a)] = 1.0 if n in ("bodyF", "neck") else -1.0
    return sk


def independent_dofs(sk: abi.Skeleton) -> np.ndarray:
    dep = set()
    for j in range(sk.n_joints):
        c = sk.joint_child[j]
        dep.add(3 + 3 * c)
        if sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y:
            dep.add(3 + 3 * c + 2)
    return np.array([p for p in range(sk.nq) if p not in dep], dtype=np.int32)


def jacobian_layout(sk: abi.Skeleton):
    """(slot_marker, slot_dof): the structurally non-zero (marker, dof) pairs, marker-major; within a
    marker: x, y, z, then the 3 angles of each link on the path root -> marker link; then 0-3 zero alignment slots."""
    sm, sd = [], []
    for l in range(sk.n_markers):
        chain = []
        k = sk.marker_link[l]
        while k >= 0:
            chain.append(k)
            k = sk.parent[k]
        chain.reverse()
        dofs = [0, 1, 2] + [3 + 3 * k + a for k in chain for a in range(3)]
        sm += [l] * len(dofs)
        sd += dofs
        if l == 0:
            chain0 = list(chain)
    # alignment slots (cpe_api.hip build_model): the count is padded to a multiple of 4 with structurally zero pairs of
    # marker 0 -- dofs outside its chain, highest first -- so that a camera row of J is a whole number of 64-byte lines
    d = sk.n_links * 3 + 2
    while len(sm) % 4 and d >= 3:
        if (d - 3) // 3 not in chain0:
            sm.append(0)
            sd.append(d)
        d -= 1
    return np.array(sm, dtype=np.int32), np.array(sd, dtype=np.int32)


FEET = ("HFL", "HFR", "HBL", "HBR")                   # pe.foot.feet(robot): add_foot(hock, at="bottom") in link order (cheetah.py:104)
FOOT_MARKERS = ("l_front_paw", "r_front_paw", "l_back_paw", "r_back_paw")


def grf_options(animal: str = "phantom", iterations: int = 2000) -> abi.GrfOptions:
    """Options of the per-frame GRF fit (acinoset_opt.py:176-270): feet = hock bottoms = the paw markers, root inertia =
    solid cylinder along the body axis x (I_axis = m r^2 / 2, I_perp = m L^2 / 12 + m r^2 / 4)."""
    p = load_params(animal)["body_B"]
    m, L, r = p["mass"], p["length"], p["radius"]
    o = abi.GrfOptions()
    o.root_inertia[0] = 0.5 * m * r * r
    o.root_inertia[1] = o.root_inertia[2] = m * L * L / 12.0 + 0.25 * m * r * r
    o.friction_ratio, o.force_max, o.regularisation, o.gravity = 1.3, 5.0, 1e-6, 9.81
    o.n_feet = 4
    for i, name in enumerate(FOOT_MARKERS):
        o.foot_marker[i] = MARKERS.index(name)
    o.iterations = iterations
    return o


def eom_options(animal: str = "phantom") -> abi.EomOptions:
    """gravity and the principal moments of every link: solid cylinders (I_axis = m r^2 / 2,
    I_perp = m L^2 / 12 + m r^2 / 4) along the link's aligned axis -- x for the trunk and tail links, z for the leg links
    (cheetah.py:19-200)."""
    p = load_params(animal)
    o = abi.EomOptions()
    o.gravity = 9.81
    for i, name in enumerate(LINKS):
        lp = _link_param(p, name)
        m, L, r = lp["mass"], lp["length"], lp["radius"]
        ax, perp = 0.5 * m * r * r, m * L * L / 12.0 + 0.25 * m * r * r
        along_x = name in ("base", "bodyF", "neck", "tail0", "tail1")
        o.link_inertia[i][0] = ax if along_x else perp
        o.link_inertia[i][1] = perp
        o.link_inertia[i][2] = perp if along_x else ax
    return o


# motors of the physics-based model: add_torque(first, second, about=...) in cheetah.py:70-165 -- 3 + 3 + 2 + 2 + 4 x 3 = 22 torques
MOTORS = ([("bodyF", "base", a) for a in "xyz"] + [("neck", "bodyF", a) for a in "xyz"] +
          [("base", "tail0", a) for a in "yz"] + [("tail0", "tail1", a) for a in "yz"] +
          [m for leg, body in (("FL", "bodyF"), ("FR", "bodyF"), ("BL", "base"), ("BR", "base"))
           for m in ((body, "U" + leg, "y"), ("U" + leg, "L" + leg, "y"), ("L" + leg, "H" + leg, "y"))])


def dyn_options(animal: str = "phantom") -> abi.DynOptions:
    o = abi.DynOptions()
    o.eom = eom_options(animal)
    o.n_feet = 4
    for i, name in enumerate(FOOT_MARKERS):
        o.foot_marker[i] = MARKERS.index(name)
    o.n_motors = len(MOTORS)
    for i, (a, b, ax) in enumerate(MOTORS):
        o.motor_first[i], o.motor_second[i], o.motor_axis[i] = LINKS.index(a), LINKS.index(b), "xyz".index(ax)
    return o


def motor_groups():
    """[(motor name, [columns of the torque vector])]: one entry per add_torque(first, second, about=...) of cheetah.py:70-165, its components
    in axis order.  Names follow `<first>_<second>_torque` (the naming of the absent physical_education.motor: unpinned)."""
    groups, order = {}, []
    for i, (a, b, ax) in enumerate(MOTORS):
        key = f"{a}_{b}_torque"
        if key not in groups:
            groups[key] = []; order.append(key)
        groups[key].append(i)
    return [(k, groups[k]) for k in order]


def constraint_rows(sk: abi.Skeleton):
    """(parent link, child link) of every joint-equality row, in the order of the constraint forces lambda: two rows per revolute joint
    (parent.y . child.x, parent.y . child.z), one per hooke joint (cheetah.py:71-72,101,160-161)"""
    rows = []
    for j in range(sk.n_joints):
        for _ in range(2 if sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y else 1):
            rows.append((sk.joint_parent[j], sk.joint_child[j]))
    return rows


def without_motion_model(sk: abi.Skeleton) -> abi.Skeleton:
    """copy of the skeleton with the constant-acceleration weights zeroed: the physics-based cost (acinoset_opt.py:905-921) has no such term"""
    import ctypes as C
    out = abi.Skeleton()
    C.memmove(C.byref(out), C.byref(sk), C.sizeof(abi.Skeleton))
    for p in range(len(out.motion_w)):
        out.motion_w[p] = 0.0
    return out


# ---- pairwise pseudo-measurements (PPM, acinoset_misc.py:179, :211-256; enabled for the "flick" clips, run_dataset.py:1323) -----------
# R_pw rows 1 and 2 of get_uncertainty_models (acinoset_misc.py:1791-1847; doubled at :1850 like row 0) and get_pairwise_graph (:1972-1998):
# the two DLC body parts each marker is also predicted FROM (DLC part index, not marker index).  tests/test_oracle_golden.py compares these
# tables with the values the reference's own functions return (tests/golden/misc_golden.npz, misc_names.json).
R_PW1 = np.array([2.71, 3.06, 2.99, 4.07, 5.53, 4.67, 6.05, 5.6, 5.01, 5.11, 5.24, 4.85, 5.18, 5.28, 5.5, 4.9, 4.7, 4.7, 5.21, 5.11, 5.1, 5.27, 5.75, 5.44])
R_PW2 = np.array([2.8, 3.24, 3.42, 3.8, 4.4, 5.43, 5.22, 7.29, 8.19, 6.5, 5.9, 6.18, 8.83, 6.52, 6.22, 6.34, 6.8, 6.12, 5.37, 5.98, 7.83, 6.44, 6.1, 6.38])
PAIRWISE = {"r_eye": (23, 1), "l_eye": (23, 0), "nose": (0, 1), "neck_base": (6, 23), "spine": (22, 24), "tail_base": (6, 11), "tail1": (6, 22),
            "tail2": (11, 22), "l_shoulder": (14, 24), "l_front_knee": (13, 15), "l_front_ankle": (13, 14), "l_front_paw": (14, 15),
            "r_shoulder": (3, 24), "r_front_knee": (2, 4), "r_front_ankle": (2, 3), "r_front_paw": (3, 4), "l_hip": (18, 22), "l_back_knee": (17, 19),
            "l_back_ankle": (17, 18), "l_back_paw": (18, 19), "r_hip": (8, 22), "r_back_knee": (7, 9), "r_back_ankle": (7, 8), "r_back_paw": (8, 9)}


def pairwise_sigma(w: int, kinetic_dataset: bool = False) -> np.ndarray:
    """R_pw[w] (w = 0: the detection itself, 1, 2: the two pairwise predictions), doubled (acinoset_misc.py:1850); 7 for the kinetic dataset (:187-188)"""
    if kinetic_dataset:
        return np.full(24, 7.0)
    return 2.0 * (R_MEAS, R_PW1, R_PW2)[w]
