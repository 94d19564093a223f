"""Synthetic benchmark sequences: a galloping-cheetah-like trajectory of the
17-link skeleton seen by C fisheye cameras modelled on the AcinoSet GoPro calibration, with DLC-like
noise, outliers and likelihood drop-outs.  Data generation only (numpy, host): nothing here is on the
solve path, and nothing is read from the reference at run time.
"""
import math

import numpy as np

from . import abi
from .skeleton import LINKS, dof, PHI, THETA, PSI, measurement_sigma, independent_dofs

IMG_W, IMG_H = 2704, 1520


# ---------------------------------------------------------------------------------------------------
def rot_zyx(ang: np.ndarray) -> np.ndarray:
    """ang[..., 3] = (phi, theta, psi) -> R[..., 3, 3] = Rz(psi) Ry(theta) Rx(phi)."""
    sf, cf = np.sin(ang[..., 0]), np.cos(ang[..., 0])
    st, ct = np.sin(ang[..., 1]), np.cos(ang[..., 1])
    sp, cp = np.sin(ang[..., 2]), np.cos(ang[..., 2])
    R = np.empty(ang.shape[:-1] + (3, 3))
    R[..., 0, 0] = cp * ct; R[..., 0, 1] = sf * st * cp - sp * cf; R[..., 0, 2] = sf * sp + st * cf * cp
    R[..., 1, 0] = sp * ct; R[..., 1, 1] = sf * sp * st + cf * cp; R[..., 1, 2] = -sf * cp + sp * st * cf
    R[..., 2, 0] = -st;     R[..., 2, 1] = sf * ct;                R[..., 2, 2] = cf * ct
    return R


def fk_numpy(sk: abi.Skeleton, q: np.ndarray):
    """q[..., nq] -> (positions[..., L, 3], com[..., 3])."""
    nl = sk.n_links
    R = rot_zyx(q[..., 3:].reshape(q.shape[:-1] + (nl, 3)))
    origin = [None] * nl
    for i in range(nl):
        if sk.parent[i] < 0:
            origin[i] = q[..., 0:3]
        else:
            p = sk.parent[i]
            origin[i] = origin[p] + R[..., p, :, :] @ np.array(sk.attach[i][:])
    pos = np.stack([origin[sk.marker_link[l]] + R[..., sk.marker_link[l], :, :] @ np.array(sk.marker_off[l][:])
                    for l in range(sk.n_markers)], axis=-2)
    M = sum(sk.mass[i] for i in range(nl))
    com = sum(sk.mass[i] * (origin[i] + R[..., i, :, :] @ np.array(sk.com[i][:])) for i in range(nl)) / M
    return pos, com


def project_numpy(cam: abi.Camera, p: np.ndarray) -> np.ndarray:
    """p[..., 3] -> uv[..., 2]; fisheye / pinhole of acinoset_misc.py:1663-1696."""
    Rc = np.array(cam.R[:]).reshape(3, 3)
    X = p @ Rc.T + np.array(cam.t[:])
    a, b = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
    r = np.sqrt(a * a + b * b)
    D = cam.D
    if cam.model == abi.CAM_FISHEYE:
        th = np.arctan(r)
        thd = th * (1 + D[0] * th**2 + D[1] * th**4 + D[2] * th**6 + D[3] * th**8)
        g = thd / (r + 1e-12)
    else:
        g = 1 + D[0] * r**2 + D[1] * r**4 + D[2] * r**6
    return np.stack([cam.fx * a * g + cam.cx, cam.fy * b * g + cam.cy], axis=-1), X[..., 2]


def project_dependents_numpy(sk: abi.Skeleton, q: np.ndarray, branch=None) -> np.ndarray:
    """Closed-form solve of the joint equalities for the dependent angles; q[..., nq].
    `branch[..., n_joints]` (+1 / -1, default +1) selects the sign of cos(phi) of each revolute child: both
    signs satisfy the equalities; +1 is the branch reached from the reference's initial guess."""
    q = q.copy()
    for j in range(sk.n_joints):
        p, c = sk.joint_parent[j], sk.joint_child[j]
        Rp = rot_zyx(q[..., 3 + 3 * p:6 + 3 * p])
        ax, ay, az = Rp[..., 0, 1], Rp[..., 1, 1], Rp[..., 2, 1]
        th = q[..., 3 + 3 * c + 1]
        st, ct = np.sin(th), np.cos(th)
        if sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y:
            sphi = np.clip(az / ct, -1, 1)
            phi = np.arcsin(sphi)
            if branch is not None:
                phi = np.where(np.asarray(branch)[..., j] < 0, np.pi - phi, phi)
            psi = np.arctan2(ay, ax) - np.arctan2(np.cos(phi), sphi * st)
            ref = q[..., 3 + 3 * p + 2]
            psi = psi + 2 * np.pi * np.round((ref - psi) / (2 * np.pi))
            q[..., 3 + 3 * c], q[..., 3 + 3 * c + 2] = phi, psi
        else:
            psi = q[..., 3 + 3 * c + 2]
            sp, cp = np.sin(psi), np.cos(psi)
            q[..., 3 + 3 * c] = np.arctan2(ax * st * cp + ay * st * sp + az * ct, ay * cp - ax * sp)
    return q


def leg_layout(sk: abi.Skeleton):
    """[(child link, body link)] of every revolute (leg) joint, in joint order; the body is the link whose y axis the
    whole leg shares (cheetah.py:71-72,101)."""
    body_of, out = {}, []
    for j in range(sk.n_joints):
        if sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y:
            p, c = sk.joint_parent[j], sk.joint_child[j]
            body_of[c] = body_of.get(p, p)
            out.append((c, body_of[c]))
    return out


def legs_from_alpha(sk: abi.Skeleton, q: np.ndarray, alpha: np.ndarray) -> np.ndarray:
    """Euler angles of the leg links from their rotation alpha[..., nrev] about the body's y axis:
    R_c = R_B Ry(alpha) (principal pitch |theta_c| <= 90 deg, phi_c in (-180, 180] deg)."""
    q = q.copy()
    for r, (c, B) in enumerate(leg_layout(sk)):
        RB = rot_zyx(q[..., 3 + 3 * B:6 + 3 * B])
        ca, sa = np.cos(alpha[..., r]), np.sin(alpha[..., r])
        Ry = np.zeros(alpha.shape[:-1] + (3, 3))
        Ry[..., 0, 0] = ca; Ry[..., 0, 2] = sa; Ry[..., 1, 1] = 1.0; Ry[..., 2, 0] = -sa; Ry[..., 2, 2] = ca
        Rc = RB @ Ry
        q[..., 3 + 3 * c + 1] = np.arcsin(np.clip(-Rc[..., 2, 0], -1, 1))
        q[..., 3 + 3 * c] = np.arctan2(Rc[..., 2, 1], Rc[..., 2, 2])
        psi = np.arctan2(Rc[..., 1, 0], Rc[..., 0, 0])
        ref = q[..., 3 + 3 * B + 2]
        q[..., 3 + 3 * c + 2] = psi + 2 * np.pi * np.round((ref - psi) / (2 * np.pi))
    return q


def cost_view_numpy(sk: abi.Skeleton, q: np.ndarray) -> np.ndarray:
    """What the solver's cost terms see of an Euler trajectory q[..., nq] ('cost pitch'): the pitch slot of every leg link holds
    theta_B + alpha_c -- pitch of the body the leg hangs from + the leg's angle about the body's y axis (alpha_c from R_B^T R_c) -- instead of the
    Euler pitch of the link's own (principal) triple.  Equal to the reference's variable for an unrolled trunk, smooth through +-90 degrees."""
    qc = q.copy()
    for c, B in leg_layout(sk):
        M = np.einsum("...ji,...jk->...ik", rot_zyx(q[..., 3 + 3 * B:6 + 3 * B]), rot_zyx(q[..., 3 + 3 * c:6 + 3 * c]))
        qc[..., 3 + 3 * c + 1] = q[..., 3 + 3 * B + 1] + np.arctan2(M[..., 0, 2], M[..., 0, 0])
    return qc


# ---------------------------------------------------------------------------------------------------
def look_at_camera(pos, target, fx, fy, cx, cy, D, model=abi.CAM_FISHEYE, mult=1.0) -> abi.Camera:
    pos, target = np.asarray(pos, float), np.asarray(target, float)
    fwd = target - pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0])
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rc = np.stack([right, down, fwd])          # world -> camera
    cam = abi.Camera()
    cam.model = model
    cam.fx, cam.fy, cam.cx, cam.cy = fx, fy, cx, cy
    for i in range(4):
        cam.D[i] = D[i]
    for i in range(9):
        cam.R[i] = Rc.reshape(-1)[i]
    t = -Rc @ pos
    for i in range(3):
        cam.t[i] = t[i]
    cam.mult = mult
    return cam


def make_cameras(n_cams: int = 6, seed: int = 1234, track: float = 20.0):
    """3 cameras per side of a `track`-m straight, 4 m lateral, 0.5-1.0 m high, looking at the track
    centre; intrinsics = the AcinoSet GoPro calibration recovered in SURVEY 8c-5, each +-1 %."""
    rng = np.random.default_rng(seed)
    cams = (abi.Camera * n_cams)()
    D0 = np.array([0.0366, 0.0480, -0.0347, 0.0074])
    for c in range(n_cams):
        side = 1.0 if c % 2 == 0 else -1.0
        x = track * (0.15 + 0.35 * (c // 2)) if n_cams > 1 else track / 2
        pos = [x, side * 4.0, rng.uniform(0.5, 1.0)]
        s = rng.uniform(0.99, 1.01, 4)
        cams[c] = look_at_camera(pos, [x + rng.uniform(-1, 1), 0.0, 0.4], 1241.84 * s[0], 1239.92 * s[1],
                                 1346.96 * s[2], 773.02 * s[3], D0 * rng.uniform(0.8, 1.2, 4))
    return cams


def truth_trajectory(sk: abi.Skeleton, N: int, fps: float, rng: np.random.Generator, speed: float = 12.0, wide_limbs: bool = False):
    """Ground-truth q[N, nq]: base x = speed*t, z = 0.55 + 0.03 sin(2 pi 3 t), limb pitch sinusoids at
    3 Hz inside the reference's joint ranges (cheetah.py:333-352); limb roll/yaw from the joint
    equalities."""
    t = np.arange(N) / fps
    w = 2 * math.pi * 3.0
    q = np.zeros((N, sk.nq))
    ph = rng.uniform(0, 2 * math.pi, 16)
    amp = rng.uniform(0.8, 1.2, 16)
    x0 = rng.uniform(0.0, 1.0)
    q[:, 0] = x0 + speed * t
    q[:, 1] = rng.uniform(-0.3, 0.3) + 0.2 * t
    q[:, 2] = 0.55 + 0.03 * np.sin(w * t + ph[0])
    # '-x' aligned body: the head points along -x of the base frame, so running towards +x means
    # psi = pi + heading (acinoset_misc.py:454)
    heading = math.pi + math.atan2(0.2, speed)
    base_pitch = 0.08 * amp[1] * np.sin(w * t + ph[1])
    q[:, dof("base", PHI)] = 0.03 * np.sin(w * t + ph[2])
    q[:, dof("base", THETA)] = base_pitch
    q[:, dof("base", PSI)] = heading + 0.03 * np.sin(0.5 * w * t + ph[3])
    q[:, dof("bodyF", PHI)] = q[:, dof("base", PHI)] + 0.03 * np.sin(w * t + ph[4])
    q[:, dof("bodyF", THETA)] = base_pitch - 0.15 * amp[2] * np.sin(w * t + ph[1])
    q[:, dof("bodyF", PSI)] = q[:, dof("base", PSI)] + 0.04 * np.sin(0.5 * w * t + ph[5])
    q[:, dof("neck", PHI)] = q[:, dof("bodyF", PHI)]
    q[:, dof("neck", THETA)] = q[:, dof("bodyF", THETA)] + 0.2 + 0.05 * np.sin(w * t + ph[6])
    q[:, dof("neck", PSI)] = q[:, dof("bodyF", PSI)]
    q[:, dof("tail0", THETA)] = base_pitch + 0.3 + 0.2 * amp[3] * np.sin(w * t + ph[7])
    q[:, dof("tail0", PSI)] = q[:, dof("base", PSI)] + 0.2 * np.sin(0.5 * w * t + ph[8])
    q[:, dof("tail1", THETA)] = q[:, dof("tail0", THETA)] + 0.2 * np.sin(w * t + ph[9])
    q[:, dof("tail1", PSI)] = q[:, dof("tail0", PSI)] + 0.2 * np.sin(0.5 * w * t + ph[10])
    for i, (U, Lk, H, back, body) in enumerate((("UFL", "LFL", "HFL", False, "bodyF"), ("UFR", "LFR", "HFR", False, "bodyF"),
                                                ("UBL", "LBL", "HBL", True, "base"), ("UBR", "LBR", "HBR", True, "base"))):
        p0 = ph[11 + i]
        thU = 0.45 * amp[11 + i] * np.sin(w * t + p0)
        if back:
            thL = thU - 0.45 - 0.3 * np.sin(w * t + p0 + 0.8)         # thigh - calf in [0.15, 0.75]
            thH = thL + 0.5 + 0.3 * np.sin(w * t + p0 + 1.6)          # calf - hock in [-0.8, -0.2]
        else:
            thL = thU + 0.45 + 0.3 * np.sin(w * t + p0 + 0.8)         # thigh - calf in [-0.75, -0.15]
            thH = thL - 0.3 - 0.3 * np.sin(w * t + p0 + 1.6)          # calf - hock in [0, 0.6]
        for n, th in ((U, thU), (Lk, thL), (H, thH)):
            q[:, dof(n, THETA)] = th
            q[:, dof(n, PSI)] = q[:, dof(body, PSI)]
    q = project_dependents_numpy(sk, q)
    if wide_limbs:
        # limbs swinging through and beyond the horizontal under a rolled trunk (as in the stored AcinoSet runs,
        # tests/golden/fk_csv_pin.npz): generated in the leg-angle coordinates alpha, where this is smooth
        q[:, dof("base", PHI)] += 0.2
        q[:, dof("bodyF", PHI)] += 0.15
        lay = leg_layout(sk)
        alpha = np.zeros((N, len(lay)))
        for r, (c, B) in enumerate(lay):
            alpha[:, r] = q[:, 3 + 3 * c + 1] - q[:, 3 + 3 * B + 1]
            if LINKS[c][0] in "LH":
                alpha[:, r] *= 2.0                                        # calves and hocks: up to ~2.2 rad
        q = legs_from_alpha(sk, q, alpha)
        qh = project_dependents_numpy(sk, q)                              # tails: hooke phi from the (rolled) parent
        for j in range(sk.n_joints):
            if sk.joint_kind[j] == abi.JOINT_HOOKE_YZ:
                c = sk.joint_child[j]
                q[:, 3 + 3 * c] = qh[:, 3 + 3 * c]
    return q


def make_batch(sk: abi.Skeleton, cams, B: int, N: int = 200, fps: float = 120.0, seed: int = 1234,
               noise_px: float = 2.0, outlier_frac: float = 0.10, init_noise: float = 0.05,
               dlc_thresh: float = 0.5, kinetic_dataset: bool = False, wide_limbs: bool = False, shutter_delay=None):
    """B independent sequences (sequence b uses seed + b).  Returns dict of C-contiguous fp64 arrays:
    q_true, q_init [B,N,nq]; meas [B,N,C,L,2]; weight [B,N,C,L].
    shutter_delay [C] (seconds): camera c sees the markers displaced by x' tau_c + x'' tau_c^2 of the base position
    (acinoset_misc.py:283-285), backward differences, from node 2 on as cpe_solve_shutter models it."""
    C, L, nq = len(cams), sk.n_markers, sk.nq
    q_true = np.empty((B, N, nq)); q_init = np.zeros((B, N, nq))
    meas = np.empty((B, N, C, L, 2)); weight = np.empty((B, N, C, L))
    sigma = measurement_sigma(L, kinetic_dataset)
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        qt = truth_trajectory(sk, N, fps, rng, wide_limbs=wide_limbs)
        q_true[b] = qt
        pos, _ = fk_numpy(sk, qt)
        for c in range(C):
            pc = pos
            if shutter_delay is not None and N > 2:
                ta, x = float(shutter_delay[c]), qt[:, 0:3]
                d = np.zeros((N, 3))
                d[2:] = (x[2:] - x[1:-1]) * fps * ta + (x[2:] - 2 * x[1:-1] + x[:-2]) * fps * fps * ta * ta
                pc = pos + d[:, None, :]
            uv, z = project_numpy(cams[c], pc)
            uv = uv + rng.normal(0, noise_px, uv.shape)
            out = rng.random((N, L)) < outlier_frac
            uv[out] = np.stack([rng.uniform(0, IMG_W, out.sum()), rng.uniform(0, IMG_H, out.sum())], axis=-1)
            lik = rng.random((N, L))
            vis = (z > 0.1) & (uv[..., 0] >= 0) & (uv[..., 0] < IMG_W) & (uv[..., 1] >= 0) & (uv[..., 1] < IMG_H)
            wgt = np.where((lik > dlc_thresh) & vis, 1.0 / sigma[None, :], 0.0)    # acinoset_misc.py:231
            uv[~vis] = 0.0
            meas[b, :, c] = uv
            weight[b, :, c] = wgt
        # reference initial guess (acinoset_opt.py:574-583): all angles 0, psi = heading for every link
        dx = np.diff(qt[:, 0]); dy = np.diff(qt[:, 1])
        psi = np.arctan2(dy, dx) if N > 1 else np.zeros(1)
        psi = np.pi + (np.append(psi, psi[-1]) if N > 1 else psi)          # acinoset_misc.py:450-454
        q_init[b, :, 0:3] = qt[:, 0:3] + rng.normal(0, init_noise, (N, 3))
        for i in range(sk.n_links):
            q_init[b, :, 3 + 3 * i + 2] = psi
    return dict(q_true=q_true, q_init=q_init, meas=np.ascontiguousarray(meas), weight=np.ascontiguousarray(weight))


# ---------------------------------------------------------------------------------------------------
# config 4 ("contact schedule = 4 feet rotary gallop 3 Hz, stance 12 frames"): a gallop whose paws are PLANTED during
# their stance windows, so that the physics-based model (no-slip, foot height, ground-reaction forces) has a consistent problem.
def _smoother(s):
    s = np.clip(s, 0.0, 1.0)
    return s * s * s * (10.0 - 15.0 * s + 6.0 * s * s)          # C2 step


def gallop_trajectory(sk: abi.Skeleton, N: int, fps: float, rng: np.random.Generator, speed: float = 7.0, stride_hz: float = 3.0,
                      stance_frames: int = 12, clearance: float = 0.05):
    """q[N, nq] and stance[N, 4] (feet in skeleton.FEET order: HFL, HFR, HBL, HBR) of a rotary gallop along +x: the trunk moves on
    smooth sinusoids, every paw stands still on the ground (z = 0) for `stance_frames` frames per stride -- touch-downs in the order
    back-right, back-left, front-right, front-left, a quarter stride apart -- and swings forward on a C2 curve in between; the leg
    angles follow by planar inverse kinematics (leg links are rotations of their body about its y axis, cheetah.py:71-72,101)."""
    from .skeleton import FEET, FOOT_MARKERS, MARKERS
    T = int(round(fps / stride_hz))                               # frames per stride
    t = np.arange(N) / fps
    w = 2 * math.pi * stride_hz
    q = np.zeros((N, sk.nq))
    ph = rng.uniform(0, 2 * math.pi, 8)
    q[:, 0] = rng.uniform(0.0, 0.5) + speed * t
    q[:, 1] = rng.uniform(-0.2, 0.2)
    q[:, 2] = 0.60 + 0.012 * np.sin(w * t + ph[0])
    for i in range(sk.n_links):
        q[:, 3 + 3 * i + 2] = math.pi                             # '-x' aligned body running towards +x (acinoset_misc.py:454)
    pitch = -0.10 + 0.04 * np.sin(w * t + ph[1])                 # nose down: the fore legs are shorter than the hind legs
    q[:, dof("base", THETA)] = pitch
    q[:, dof("bodyF", THETA)] = pitch - 0.06 - 0.05 * np.sin(w * t + ph[1] + 0.5)
    q[:, dof("neck", THETA)] = q[:, dof("bodyF", THETA)] + 0.2 + 0.04 * np.sin(w * t + ph[2])
    q[:, dof("tail0", THETA)] = pitch + 0.3 + 0.15 * np.sin(w * t + ph[3])
    q[:, dof("tail1", THETA)] = q[:, dof("tail0", THETA)] + 0.15 * np.sin(w * t + ph[4])
    lay = leg_layout(sk)
    alpha = np.zeros((N, len(lay)))
    stance = np.zeros((N, 4), np.int32)
    off0 = int(rng.integers(0, T))
    order = {"HBR": 0, "HBL": 1, "HFR": 2, "HFL": 3}               # touch-down order of the rotary gallop, a quarter stride apart
    nl = sk.n_links
    R = rot_zyx(q[..., 3:].reshape(N, nl, 3))
    org = [None] * nl
    for i in range(nl):
        org[i] = q[:, 0:3] if sk.parent[i] < 0 else org[sk.parent[i]] + R[:, sk.parent[i]] @ np.array(sk.attach[i][:])
    for k, foot in enumerate(FEET):
        leg = foot[1:]                                            # FL, FR, BL, BR
        iU, iL, iH = LINKS.index("U" + leg), LINKS.index("L" + leg), LINKS.index("H" + leg)
        B = sk.parent[iU]
        L1, L2 = abs(sk.attach[iL][2]), abs(sk.attach[iH][2])
        L3 = abs(sk.marker_off[MARKERS.index(FOOT_MARKERS[k])][2])
        back = leg[0] == "B"
        hip = org[iU]                                             # [N, 3] world
        td0 = off0 + order[foot] * (T // 4) - 2 * T                # first touch-down (before the clip starts)
        # paw target in the world, frame by frame
        tgt = np.zeros((N, 3))
        hx = lambda fr: hip[0, 0] + speed * (fr - 0.0) / fps + (hip[:, 0] - (hip[0, 0] + speed * t)).mean()   # hip x on the mean line
        for n in range(N):
            j = (n - td0) // T                                    # stride index
            td = td0 + j * T
            xc = hx(td + 0.5 * (stance_frames - 1))                # planted under the hip at mid-stance
            if n - td < stance_frames:
                tgt[n] = (xc, hip[n, 1], 0.0); stance[n, k] = 1
            else:
                s = (n - (td + stance_frames - 1)) / float(T - stance_frames + 1)
                xn = hx(td + T + 0.5 * (stance_frames - 1))
                tgt[n] = (xc + (xn - xc) * _smoother(s), hip[n, 1], clearance * 64.0 * s**3 * (1 - s)**3)
        d = np.einsum("nji,nj->ni", R[:, B], tgt - hip)             # body frame
        dx, dz = d[:, 0], d[:, 2]
        # three links, two equations: the thigh keeps a fixed angle A* to the hip -> paw line and the hock folds against the calf
        # (delta = hock - calf) as the leg shortens, inside the joint ranges of cheetah.py:345-352
        Astar = 0.7 if back else 0.9
        D = np.minimum(np.hypot(dx, dz), 0.98 * (L1 + L2 + L3))
        Lr = np.sqrt(L1 * L1 + D * D - 2 * L1 * D * math.cos(Astar))
        Lr = np.clip(Lr, abs(L2 - L3) + 0.01, L2 + L3 - 0.005)
        delta = np.arccos(np.clip((Lr * Lr - L2 * L2 - L3 * L3) / (2 * L2 * L3), -1, 1)) * (1.0 if back else -1.0)
        beta = np.arctan2(L3 * np.sin(delta), L2 + L3 * np.cos(delta))
        gam = np.arctan2(-dx, -dz)
        A = np.arccos(np.clip((L1 * L1 + D * D - Lr * Lr) / (2 * L1 * D), -1, 1))
        a1 = gam + A if back else gam - A                          # knee forward for the hind legs, backward for the fore legs
        rx = -D * np.sin(gam) + L1 * np.sin(a1); rz = -D * np.cos(gam) + L1 * np.cos(a1)
        a2 = np.arctan2(-rx, -rz)
        for r, (c, Bk) in enumerate(lay):
            if c == iU:
                alpha[:, r] = a1
            elif c == iL:
                alpha[:, r] = a2 - beta
            elif c == iH:
                alpha[:, r] = a2 - beta + delta
    q = legs_from_alpha(sk, q, alpha)
    q = project_dependents_numpy_hooke(sk, q)
    return q, stance


def project_dependents_numpy_hooke(sk: abi.Skeleton, q: np.ndarray) -> np.ndarray:
    """phi of the hooke children (tails) from their joint equality, everything else untouched"""
    qh = project_dependents_numpy(sk, q)
    q = q.copy()
    for j in range(sk.n_joints):
        if sk.joint_kind[j] == abi.JOINT_HOOKE_YZ:
            c = sk.joint_child[j]
            q[..., 3 + 3 * c] = qh[..., 3 + 3 * c]
    return q


def make_gallop_batch(sk: abi.Skeleton, cams, B: int, N: int = 200, fps: float = 120.0, seed: int = 4321, noise_px: float = 2.0,
                      outlier_frac: float = 0.10, init_noise: float = 0.02, dlc_thresh: float = 0.5, speed: float = 7.0,
                      kinetic_dataset: bool = False, clearance: float = 0.05, stance_frames: int = 12, x0: float = 0.0):
    """config-4 sequences: q_true, q_init (truth + noise on the independent coordinates: the physics-based solve is warm-started
    from a kinematic solution, acinoset_opt.py:739-777), meas, weight as make_batch, and stance [B, N, 4].  kinetic_dataset: the 7 px
    sigma of that data set (acinoset_misc.py:187-188); clearance: height of the swinging paws (the contact heuristic of
    acinoset_misc.py:745-856 needs more than 5 cm to see a flight phase); x0: shift of the run along the track."""
    C, L, nq = len(cams), sk.n_markers, sk.nq
    out = dict(q_true=np.empty((B, N, nq)), q_init=np.empty((B, N, nq)), meas=np.empty((B, N, C, L, 2)), weight=np.empty((B, N, C, L)),
               stance=np.zeros((B, N, 4), np.int32))
    sigma = measurement_sigma(L, kinetic_dataset)
    ind = independent_dofs(sk)
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        qt, st = gallop_trajectory(sk, N, fps, rng, speed=speed, stance_frames=stance_frames, clearance=clearance)
        qt[:, 0] += x0
        out["q_true"][b] = qt; out["stance"][b] = st
        pos, _ = fk_numpy(sk, qt)
        for c in range(C):
            uv, z = project_numpy(cams[c], pos)
            uv = uv + rng.normal(0, noise_px, uv.shape)
            o = rng.random((N, L)) < outlier_frac
            uv[o] = np.stack([rng.uniform(0, IMG_W, o.sum()), rng.uniform(0, IMG_H, o.sum())], axis=-1)
            lik = rng.random((N, L))
            vis = (z > 0.1) & (uv[..., 0] >= 0) & (uv[..., 0] < IMG_W) & (uv[..., 1] >= 0) & (uv[..., 1] < IMG_H)
            uv[~vis] = 0.0
            out["meas"][b, :, c] = uv
            out["weight"][b, :, c] = np.where((lik > dlc_thresh) & vis, 1.0 / sigma[None, :], 0.0)
        qi = qt.copy()
        qi[:, ind] += rng.normal(0, init_noise, (N, len(ind))) * np.where(np.arange(len(ind)) < 3, 0.5, 1.0)
        out["q_init"][b] = qi
    out["meas"] = np.ascontiguousarray(out["meas"]); out["weight"] = np.ascontiguousarray(out["weight"])
    return out
