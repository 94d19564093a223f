#!/usr/bin/env python3
"""Pin the FK / marker / joint model against the reference's STORED 2D outputs (build container only).

`data/test_set/<seq>/fte_kinematic/cam{1..6}_fte.csv` are text files written by the reference
(acinoset_misc.py:1346-1407): the fisheye projection (cv.fisheye.projectPoints) of the 24 marker positions its
own FK produced at its solution, for every camera and frame.  The camera calibration and the solution q are NOT
available (calibration files absent; fte.pickle is a pickle the permitted loaders refuse).  This script shows
that there EXIST camera parameters and joint angles for which THIS repository's FK + marker model + joint
equalities + fisheye projection reproduce all 57 x 6 x 24 x 2 stored numbers, by recovering them:

  1. undistort with the intrinsics recovered in SURVEY 8c-5, two-view geometry (8-point, cameras 1-2),
     triangulation, metric scale from the base-link length, linear PnP for the other cameras;
  2. point-based bundle adjustment (scipy, sparse finite differences);
  3. z-up world frame from the animal itself, per-frame skeleton fit, then a joint least-squares refinement of
     all reduced coordinates u (57 x 28, dependent angles from the joint equalities) and all 6 x 14 camera
     parameters against the stored 2D values.

Output: tests/golden/fk_csv_pin.npz = {uv (the stored numbers), q, cams, residual statistics}.  A wrong link
length, marker offset, chain attachment or joint axis leaves a residual of pixels; the recovered fit is at the
1e-6 px level or below (see DESIGN.md for the achieved figure).
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from cheetah_pose_estimation_amd import abi, skeleton, synth  # noqa: E402
from oracle.initial_guess import undistort_fisheye as _undistort_fisheye, triangulate as _triangulate  # noqa: E402  (tools/ is build-container tooling, not product)

# usage: pin_fk_from_csv.py [sequence [result directory [output file]]]
SEQ = sys.argv[1] if len(sys.argv) > 1 else "2019_03_07/phantom/run"
SUB = sys.argv[2] if len(sys.argv) > 2 else "fte_kinematic"
OUT = sys.argv[3] if len(sys.argv) > 3 else "fk_csv_pin.npz"
ANIMAL = SEQ.split("/")[-2] if "kinetic" not in SEQ else SEQ.split("/")[-2]
SRC = f"/root/reference/data/test_set/{SEQ}/{SUB}"
K0 = np.array([1241.84, 1239.92, 1346.96, 773.02])
D0 = np.array([0.0366, 0.0480, -0.0347, 0.0074])
if SEQ.startswith("2017"):
    K0 = K0 * (1920.0 / 2704.0)        # the 2017 recordings are 1920 x 1080 (largest stored pixel 1917): same lens, scaled sensor read-out; refined below


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def inv_rodrigues(R):
    th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    if th < 1e-12:
        return np.zeros(3)
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def project(cp, X):
    """cp = [fx, fy, cx, cy, D0..3, rvec(3), t(3)]; X[..., 3] -> uv[..., 2] (acinoset_misc.py:1663-1679)"""
    R = rodrigues(cp[8:11])
    Xc = X @ R.T + cp[11:14]
    a, b = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
    r = np.sqrt(a * a + b * b)
    th = np.arctan(r)
    thd = th * (1 + cp[4] * th**2 + cp[5] * th**4 + cp[6] * th**6 + cp[7] * th**8)
    g = thd / (r + 1e-12)
    return np.stack([cp[0] * a * g + cp[2], cp[1] * b * g + cp[3]], axis=-1)


def load_uv():
    arrs = []
    for c in range(1, 7):
        f = os.path.join(SRC, f"cam{c}_fte.csv")
        if not os.path.exists(f):      # a camera the stored result does not cover (4- and 5-camera scenes): all its pixels count as empty
            arrs.append(None); continue
        rows = np.genfromtxt(f, delimiter=",", skip_header=2)
        arrs.append(rows[:, 1:].reshape(len(rows), 24, 3)[:, :, :2])
    shape = next(a for a in arrs if a is not None).shape
    return np.stack([a if a is not None else np.full(shape, np.nan) for a in arrs], 1)          # [N, C, L, 2]


def essential_8pt(n1, n2):
    A = np.stack([n2[:, 0] * n1[:, 0], n2[:, 0] * n1[:, 1], n2[:, 0], n2[:, 1] * n1[:, 0], n2[:, 1] * n1[:, 1], n2[:, 1],
                  n1[:, 0], n1[:, 1], np.ones(len(n1))], axis=1)
    E = np.linalg.svd(A)[2][-1].reshape(3, 3)
    U, S, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    cands = [(U @ W @ Vt, U[:, 2]), (U @ W @ Vt, -U[:, 2]), (U @ W.T @ Vt, U[:, 2]), (U @ W.T @ Vt, -U[:, 2])]
    best, bestn = None, -1
    for R, t in cands:
        X = _triangulate(n1[::7], n2[::7], np.eye(3), np.zeros(3), R, t)
        z1 = X[:, 2]; z2 = (X @ R.T + t)[:, 2]
        n = int(((z1 > 0) & (z2 > 0)).sum())
        if n > bestn:
            best, bestn = (R, t), n
    return best


def pnp_dlt(X, n):
    A = []
    for Xi, ni in zip(X, n):
        Xh = np.append(Xi, 1.0)
        A.append(np.concatenate([Xh, np.zeros(4), -ni[0] * Xh]))
        A.append(np.concatenate([np.zeros(4), Xh, -ni[1] * Xh]))
    P = np.linalg.svd(np.array(A))[2][-1].reshape(3, 4)
    U, S, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt
    sc = S.mean()
    if np.linalg.det(R) < 0:
        R, sc = -R, -sc
    t = P[:, 3] / sc
    if ((X @ R.T + t)[:, 2] > 0).mean() < 0.5:
        R, t = -R, -t            # should not happen after the det fix; kept for safety
    return R, t


def main():
    uv = load_uv()
    N, C, L, _ = uv.shape
    sk = skeleton.build_skeleton(ANIMAL, 24)
    Kmat = np.array([[K0[0], 0, K0[2]], [0, K0[1], K0[3]], [0, 0, 1.0]])
    nrm = [_undistort_fisheye(uv[:, c].reshape(-1, 2), Kmat, D0) for c in range(C)]
    # ---- 1. two-view geometry + scale + PnP
    assert not np.isnan(nrm[0]).any(), "camera 1 is the gauge of the bundle adjustment: it must see every stored point"
    j2 = next(c for c in range(1, C) if not np.isnan(nrm[c]).any())          # second view of the two-view start: the first other camera without empty pixels
    R01, t01 = essential_8pt(nrm[0], nrm[j2])
    X = _triangulate(nrm[0], nrm[j2], np.eye(3), np.zeros(3), R01, t01)
    Xf = X.reshape(N, L, 3)
    L_base = 2.0 * abs(sk.marker_off[5][0])
    s = L_base / np.median(np.linalg.norm(Xf[:, 5] - Xf[:, 4], axis=1))
    X *= s; t01 = t01 * s
    Rs, ts = [np.eye(3)], [np.zeros(3)]
    for c in range(1, C):
        if c == j2:
            Rs.append(R01); ts.append(t01); continue
        ok = ~np.isnan(nrm[c]).any(1)
        if ok.sum() < 12:              # camera without stored pixels: a placeholder that no residual touches
            Rs.append(np.eye(3)); ts.append(np.array([0.0, 0.0, 5.0])); continue
        R, t = pnp_dlt(X[ok], nrm[c][ok]); Rs.append(R); ts.append(t)
    cams = np.array([np.concatenate([K0, D0, inv_rodrigues(Rs[c]), ts[c]]) for c in range(C)])
    res0 = np.concatenate([np.nan_to_num(project(cams[c], X) - uv[:, c].reshape(-1, 2)).ravel() for c in range(C)])     # (NaN = a stored pixel outside the image: no row)
    print(f"after SfM init: rms reprojection {np.sqrt(np.mean(res0**2)):.3f} px")

    # ---- 2. point-based bundle adjustment (camera 0 pose fixed = gauge; scale drifts freely, fixed afterwards)
    npts = X.shape[0]

    def unpack(p):
        cp = p[:C * 14].reshape(C, 14).copy()
        cp[0, 8:14] = 0.0
        return cp, p[C * 14:].reshape(npts, 3)

    def fun_ba(p):
        cp, Xp = unpack(p)
        return np.concatenate([np.nan_to_num(project(cp[c], Xp) - uv[:, c].reshape(-1, 2)).ravel() for c in range(C)])

    spars = lil_matrix((C * npts * 2, C * 14 + npts * 3), dtype=int)
    for c in range(C):
        r0 = c * npts * 2
        spars[r0:r0 + npts * 2, c * 14:(c + 1) * 14] = 1
        for i in range(npts):
            spars[r0 + 2 * i:r0 + 2 * i + 2, C * 14 + 3 * i:C * 14 + 3 * i + 3] = 1
    p0 = np.concatenate([cams.ravel(), X.ravel()])
    sol = least_squares(fun_ba, p0, jac_sparsity=spars, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200, verbose=0)
    cams, X = unpack(sol.x)
    print(f"after point BA: rms {np.sqrt(np.mean(sol.fun**2)):.3e} px, nfev {sol.nfev}")
    Xf = X.reshape(N, L, 3)
    s = L_base / np.median(np.linalg.norm(Xf[:, 5] - Xf[:, 4], axis=1))
    X = X * s; cams[:, 11:14] *= s; Xf = X.reshape(N, L, 3)

    # ---- 3. z-up world frame taken from the animal: up = paws -> spine, x = running direction
    paws = Xf[:, [11, 15, 19, 23]].mean(1)
    up = (Xf[:, 4] - paws).mean(0); up /= np.linalg.norm(up)
    run = Xf[-1, 4] - Xf[0, 4]; run -= up * (run @ up); run /= np.linalg.norm(run)
    Rw = np.stack([run, np.cross(up, run), up])           # world <- old:  Xw = Rw (X - o)
    o = paws.mean(0) - 0.0 * up
    Xw = (X - o) @ Rw.T
    for c in range(C):
        R = rodrigues(cams[c, 8:11]); t = cams[c, 11:14]
        Rn = R @ Rw.T; tn = t + R @ o
        cams[c, 8:11] = inv_rodrigues(Rn); cams[c, 11:14] = tn
    chk = np.concatenate([np.nan_to_num(project(cams[c], Xw) - uv[:, c].reshape(-1, 2)).ravel() for c in range(C)])
    print(f"after re-framing: rms {np.sqrt(np.mean(chk**2)):.3e} px")
    Xw = Xw.reshape(N, L, 3)

    # ---- 4. per-frame skeleton fit to the reconstructed 3D markers (gives u), then joint refinement on the 2D data
    ind = skeleton.independent_dofs(sk)
    M = {m: i for i, m in enumerate(skeleton.MARKERS)}
    D = skeleton.dof
    branch = np.ones((N, sk.n_joints))

    def q_from_u(u):
        q = np.zeros((N, sk.nq)); q[:, ind] = u.reshape(N, len(ind))
        for i in range(1, sk.n_links):
            if (3 + 3 * i + 2) not in ind:
                q[:, 3 + 3 * i + 2] = q[:, 5]                       # psi seed (only used to unwrap the dependents)
        return synth.project_dependents_numpy(sk, q, branch=branch)

    # 4a. trunk first (base, bodyF, neck, tails) on the trunk markers only
    d = Xw[:, 5] - Xw[:, 4]; d /= np.linalg.norm(d, axis=1, keepdims=True)       # base x axis: spine -> tail_base
    q0 = np.zeros((N, sk.nq))
    q0[:, 0:3] = 0.5 * (Xw[:, 5] + Xw[:, 4])
    psi = np.unwrap(np.arctan2(d[:, 1], d[:, 0])); th = -np.arcsin(d[:, 2])
    for i in range(sk.n_links):
        q0[:, 3 + 3 * i + 2] = psi
    q0[:, 4] = th
    cps, sps = np.cos(psi), np.sin(psi)

    def pitch_x(top, bot, sign):     # "+x"/"-x" link: bottom - top = sign * R e_x L
        v = sign * (Xw[:, M[bot]] - Xw[:, M[top]])
        return np.arctan2(-v[:, 2], v[:, 0] * cps + v[:, 1] * sps)

    q0[:, D("bodyF", 1)] = pitch_x("spine", "neck_base", -1.0)
    q0[:, D("tail0", 1)] = pitch_x("tail_base", "tail1", 1.0)
    q0[:, D("tail1", 1)] = pitch_x("tail1", "tail2", 1.0)
    trunk = [M[m] for m in ("nose", "r_eye", "l_eye", "neck_base", "spine", "tail_base", "tail1", "tail2", "r_shoulder", "l_shoulder", "r_hip", "l_hip")]
    sp3 = lil_matrix((N * len(trunk) * 3, N * len(ind)), dtype=int)
    for n in range(N):
        sp3[n * len(trunk) * 3:(n + 1) * len(trunk) * 3, n * len(ind):(n + 1) * len(ind)] = 1
    sA = least_squares(lambda u: (synth.fk_numpy(sk, q_from_u(u))[0][:, trunk] - Xw[:, trunk]).ravel(), q0[:, ind].ravel(),
                       jac_sparsity=sp3, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200)
    print(f"trunk fit: rms {np.sqrt(np.mean(sA.fun**2)) * 1e3:.5f} mm")
    qA = q_from_u(sA.x)
    # 4b. every leg link is the body rotated about the body's y axis: R_c = R_B Ry(alpha).  alpha follows from the
    # observed link direction; the link's Euler pitch and the SIGN of cos(phi) (the branch of the joint
    # equalities) follow from R_c.  Limb pitch passes +-90 degrees in this run, where the pitch alone is ambiguous.
    RB = {b: synth.rot_zyx(qA[:, 3 + 3 * skeleton.LINKS.index(b):6 + 3 * skeleton.LINKS.index(b)]) for b in ("base", "bodyF")}
    posA = None
    q1 = qA.copy()
    for fb, body in (("F", "bodyF"), ("B", "base")):
        for side, S_ in (("r", "R"), ("l", "L")):
            name = "front" if fb == "F" else "back"
            U, Lk, H = "U" + fb + S_, "L" + fb + S_, "H" + fb + S_
            iU = skeleton.LINKS.index(U)
            # thigh start = origin_body + R_body attach  (cheetah.py:32-38)
            ib = skeleton.LINKS.index(body)
            org_b = qA[:, 0:3] if body == "base" else qA[:, 0:3] + np.einsum("nij,j->ni", RB["base"], np.array(sk.attach[1][:]))
            start = org_b + np.einsum("nij,j->ni", RB[body], np.array(sk.attach[iU][:]))
            segs = ((U, start, Xw[:, M[f"{side}_{name}_knee"]]), (Lk, Xw[:, M[f"{side}_{name}_knee"]], Xw[:, M[f"{side}_{name}_ankle"]]),
                    (H, Xw[:, M[f"{side}_{name}_ankle"]], Xw[:, M[f"{side}_{name}_paw"]]))
            for lk, top, bot in segs:
                w = np.einsum("nji,nj->ni", RB[body], -(bot - top))            # R_B^T (R_c e_z L) = L [sin a, 0, cos a]
                al = np.arctan2(w[:, 0], w[:, 2])
                ca, sa = np.cos(al), np.sin(al)
                Ry = np.zeros((N, 3, 3)); Ry[:, 0, 0] = ca; Ry[:, 0, 2] = sa; Ry[:, 1, 1] = 1; Ry[:, 2, 0] = -sa; Ry[:, 2, 2] = ca
                Rc = RB[body] @ Ry
                thc = -np.arcsin(np.clip(Rc[:, 2, 0], -1, 1))
                cphi = Rc[:, 2, 2] / np.cos(thc)
                il = skeleton.LINKS.index(lk)
                q1[:, 3 + 3 * il + 1] = thc
                j = [jj for jj in range(sk.n_joints) if sk.joint_child[jj] == il][0]
                branch[:, j] = np.where(cphi < 0, -1.0, 1.0)
    print("frames x links on the cos(phi) < 0 branch:", int((branch < 0).sum()), "of", branch.size)
    u0 = q1[:, ind].ravel()

    def fun_3d(u):
        return (synth.fk_numpy(sk, q_from_u(u))[0] - Xw).ravel()

    sp3 = lil_matrix((N * L * 3, N * len(ind)), dtype=int)
    for n in range(N):
        sp3[n * L * 3:(n + 1) * L * 3, n * len(ind):(n + 1) * len(ind)] = 1
    print(f"before refinement: rms {np.sqrt(np.mean(fun_3d(u0)**2)) * 1e3:.5f} mm")
    s3 = least_squares(fun_3d, u0, jac_sparsity=sp3, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=300)
    r3 = np.sqrt((s3.fun.reshape(N, -1) ** 2).mean(1)) * 1e3
    print(f"skeleton fit to the 3D points: rms {np.sqrt(np.mean(s3.fun**2)) * 1e3:.6f} mm, worst frame {r3.max():.6f} mm, nfev {s3.nfev}")

    nU = N * len(ind)

    def fun_joint(p):
        pos = synth.fk_numpy(sk, q_from_u(p[:nU]))[0]
        cp = p[nU:].reshape(C, 14)
        return np.concatenate([np.nan_to_num(project(cp[c], pos) - uv[:, c]).ravel() for c in range(C)])

    spj = lil_matrix((C * N * L * 2, nU + C * 14), dtype=int)
    for c in range(C):
        for n in range(N):
            r0 = (c * N + n) * L * 2
            spj[r0:r0 + L * 2, n * len(ind):(n + 1) * len(ind)] = 1
            spj[r0:r0 + L * 2, nU + c * 14:nU + (c + 1) * 14] = 1
    pj = np.concatenate([s3.x, cams.ravel()])
    for rnd in range(3):
        sj = least_squares(fun_joint, pj, jac_sparsity=spj, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
        pj = sj.x
        print(f"joint refinement round {rnd}: rms {np.sqrt(np.mean(sj.fun**2)):.3e} px, max {np.abs(sj.fun).max():.3e} px, nfev {sj.nfev}")
        pf = np.sqrt((sj.fun.reshape(C, N, -1) ** 2).mean(axis=(0, 2)))
        print("   frames with rms > 1e-6 px:", [(int(n), float(np.round(pf[n], 4))) for n in np.nonzero(pf > 1e-6)[0]])
    q = q_from_u(pj[:nU]); cams = pj[nU:].reshape(C, 14)
    out = os.path.join(ROOT, "tests", "golden", OUT)
    np.savez_compressed(out, uv=uv, q=q, cams=cams, branch=branch, seq=SEQ, animal=ANIMAL, rms_px=np.sqrt(np.mean(sj.fun**2)), max_px=np.abs(sj.fun).max())
    print("wrote", out)
    print("intrinsics per camera:\n", np.round(cams[:, :8], 4))


if __name__ == "__main__":
    main()
