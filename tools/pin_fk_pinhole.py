#!/usr/bin/env python3
"""Pin the FK / marker / joint model and the PINHOLE camera model against the reference's stored 2D outputs of the kinetic dataset (build container only).

`data/test_set/kinetic_dataset/<day>/<animal>/<trial>/fte_kinematic/cam{1..4}_fte.csv` are text files the reference wrote with
cv.projectPoints (acinoset_misc.py:1332-1336, :1373-1399) from its own FK at its own solution, for the `-02` skeletons and the four-camera
pinhole rig of that data set; points outside the 1280 x 720 image are empty (NaN).  Neither the calibration nor the solution is available.  As
tools/pin_fk_from_csv.py does for the fisheye rigs, this script shows that camera parameters and joint angles EXIST for which this repository's
FK + marker model + joint equalities + pinhole projection reproduce every stored number, by recovering them from the numbers alone:
  1. focal length by a scan (principal point at the image centre): the value at which the essential matrix of the best-covered camera pair
     fits best; two-view geometry, metric scale from the base link, linear PnP of the other cameras, multi-view triangulation of every point
     seen twice;
  2. bundle adjustment of 4 x 15 camera parameters (fx, fy, cx, cy, k1, k2, p1, p2, k3 of OpenCV's model, pose) and the points;
  3. z-up frame from the animal, per-frame skeleton fit (trunk, then leg links as rotations of their body), joint refinement of all reduced
     coordinates and cameras on the stored pixels.
Output: tests/golden/<out>.npz = {uv, q, cams, rms / max pixel error}.

usage: pin_fk_pinhole.py <sequence under data/test_set> <animal> <output file>"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cheetah_pose_estimation_amd import skeleton, synth  # noqa: E402

SEQ, ANIMAL, OUT = sys.argv[1], sys.argv[2], sys.argv[3]
SRC = f"/root/reference/data/test_set/{SEQ}/fte_kinematic"
W, H = 1280.0, 720.0


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
    """cp = [fx, fy, cx, cy, k1, k2, p1, p2, k3, rvec(3), t(3)]: cv.projectPoints"""
    R = rodrigues(cp[9:12])
    Xc = X @ R.T + cp[12:15]
    a, b = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
    r2 = a * a + b * b
    rad = 1 + cp[4] * r2 + cp[5] * r2**2 + cp[8] * r2**3
    xa = a * rad + 2 * cp[6] * a * b + cp[7] * (r2 + 2 * a * a)
    ya = b * rad + cp[6] * (r2 + 2 * b * b) + 2 * cp[7] * a * b
    return np.stack([cp[0] * xa + cp[2], cp[1] * ya + cp[3]], axis=-1)


def load_uv():
    arrs = []
    for c in range(1, 5):
        rows = np.genfromtxt(os.path.join(SRC, f"cam{c}_fte.csv"), delimiter=",", skip_header=2)
        arrs.append(rows[:, 1:].reshape(len(rows), 24, 3)[:, :, :2])
    return np.stack(arrs, 1)


def tri_multi(Ps, ns):
    """DLT triangulation of one point from normalised image points ns[i] (2,) and 3 x 4 matrices Ps[i]"""
    A = []
    for P, n in zip(Ps, ns):
        A.append(n[0] * P[2] - P[0]); A.append(n[1] * P[2] - P[1])
    Xh = np.linalg.svd(np.array(A))[2][-1]
    return Xh[:3] / Xh[3]


def essential(n1, n2):
    A = np.stack([n2[:, 0] * n1[:, 0], n2[:, 0] * n1[:, 1], n2[:, 0], n2[:, 1] * n1[:, 0], n2[:, 1] * n1[:, 1], n2[:, 1], n1[:, 0], n1[:, 1], np.ones(len(n1))], axis=1)
    E = np.linalg.svd(A)[2][-1].reshape(3, 3)
    U, S, Vt = np.linalg.svd(E)
    err = abs(S[0] - S[1]) / S[0] + S[2] / S[0]                 # an essential matrix has singular values (s, s, 0)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    best, bestn = None, -1
    for R, t in ((U @ Wm @ Vt, U[:, 2]), (U @ Wm @ Vt, -U[:, 2]), (U @ Wm.T @ Vt, U[:, 2]), (U @ Wm.T @ Vt, -U[:, 2])):
        P1 = np.hstack([np.eye(3), np.zeros((3, 1))]); P2 = np.hstack([R, t[:, None]])
        X = np.array([tri_multi([P1, P2], [a, b]) for a, b in zip(n1[::5], n2[::5])])
        n = int(((X[:, 2] > 0) & ((X @ R.T + t)[:, 2] > 0)).sum())
        if n > bestn:
            best, bestn = (R, t), n
    return best, err


def pnp_dlt(X, n):
    A = []
    for Xi, ni in zip(X, n):
        Xh = np.append(Xi, 1.0)
        A.append(np.concatenate([Xh, np.zeros(4), -ni[0] * Xh])); A.append(np.concatenate([np.zeros(4), Xh, -ni[1] * Xh]))
    P = np.linalg.svd(np.array(A))[2][-1].reshape(3, 4)
    U, S, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt; sc = S.mean()
    if np.linalg.det(R) < 0:
        R, sc = -R, -sc
    return R, P[:, 3] / sc


def main():
    uv = load_uv()
    N, C, L, _ = uv.shape
    ok = ~np.isnan(uv).any(-1)                                   # [N, C, L]
    sk = skeleton.build_skeleton(f"{ANIMAL}-02", 24, kinetic_dataset=True)
    flat = lambda c: uv[:, c].reshape(-1, 2)
    okf = lambda c: ok[:, c].reshape(-1)
    pairs = sorted(((int((okf(a) & okf(b)).sum()), a, b) for a in range(C) for b in range(a + 1, C)), reverse=True)
    _, ca, cb = pairs[0]
    if os.environ.get("PIN_PAIR"):                               # another starting pair (the one with most common points can be close to degenerate)
        ca, cb = (int(v) - 1 for v in os.environ["PIN_PAIR"].split(","))
    print("pairs by common points:", [(a + 1, b + 1, n) for n, a, b in pairs])
    both = okf(ca) & okf(cb)
    best = None
    for f in np.geomspace(500, 6000, 60):
        n1 = (flat(ca)[both] - [W / 2, H / 2]) / f; n2 = (flat(cb)[both] - [W / 2, H / 2]) / f
        (R, t), err = essential(n1, n2)
        if best is None or err < best[0]:
            best = (err, f, R, t)
    err, f0, R01, t01 = best
    print(f"camera pair ({ca + 1}, {cb + 1}) with {int(both.sum())} common points; focal scan: f = {f0:.0f} px (essential-matrix defect {err:.2e})")
    nrm = [(flat(c) - [W / 2, H / 2]) / f0 for c in range(C)]
    P = {ca: np.hstack([np.eye(3), np.zeros((3, 1))]), cb: np.hstack([R01, t01[:, None]])}
    npts = N * L
    X = np.full((npts, 3), np.nan)
    for i in np.nonzero(both)[0]:
        X[i] = tri_multi([P[ca], P[cb]], [nrm[ca][i], nrm[cb][i]])
    Xf = X.reshape(N, L, 3)
    L_base = 2.0 * abs(sk.marker_off[5][0])
    s = L_base / np.nanmedian(np.linalg.norm(Xf[:, 5] - Xf[:, 4], axis=1))
    X *= s; P[cb][:, 3] *= s
    for c in range(C):
        if c in P:
            continue
        m = okf(c) & ~np.isnan(X[:, 0])
        R, t = pnp_dlt(X[m], nrm[c][m]); P[c] = np.hstack([R, t[:, None]])
    for i in range(npts):                                       # every point seen at least twice, from all its views
        views = [c for c in range(C) if okf(c)[i]]
        X[i] = tri_multi([P[c] for c in views], [nrm[c][i] for c in views]) if len(views) >= 2 else np.nan
    have = ~np.isnan(X[:, 0])
    cams = np.array([np.concatenate([[f0, f0, W / 2, H / 2, 0, 0, 0, 0, 0], inv_rodrigues(P[c][:, :3]), P[c][:, 3]]) for c in range(C)])
    idx = np.nonzero(have)[0]

    def res_ba(cp, Xp):
        out = []
        for c in range(C):
            m = okf(c)[idx]
            out.append((project(cp[c], Xp[m]) - flat(c)[idx][m]).ravel())
        return np.concatenate(out)
    print(f"after SfM init: rms {np.sqrt(np.mean(res_ba(cams, X[idx])**2)):.3f} px over {int(have.sum())} of {npts} points")

    def unpack(p):
        cp = p[:C * 15].reshape(C, 15).copy()
        cp[ca, 9:15] = 0.0
        return cp, p[C * 15:].reshape(-1, 3)
    rows = []
    for c in range(C):
        for j in np.nonzero(okf(c)[idx])[0]:
            rows.append((c, j))
    spars = lil_matrix((2 * len(rows), C * 15 + 3 * len(idx)), dtype=int)
    for r, (c, j) in enumerate(rows):
        spars[2 * r:2 * r + 2, c * 15:(c + 1) * 15] = 1; spars[2 * r:2 * r + 2, C * 15 + 3 * j:C * 15 + 3 * j + 3] = 1
    sol = least_squares(lambda p: res_ba(*unpack(p)), np.concatenate([cams.ravel(), X[idx].ravel()]), jac_sparsity=spars, method="trf", x_scale="jac",
                        ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=150)
    cams, Xi = unpack(sol.x)
    print(f"after point BA: rms {np.sqrt(np.mean(sol.fun**2)):.3e} px, nfev {sol.nfev}")
    X[idx] = Xi
    Xf = X.reshape(N, L, 3)
    s = L_base / np.nanmedian(np.linalg.norm(Xf[:, 5] - Xf[:, 4], axis=1))
    X = X * s; cams[:, 12:15] *= s; Xf = X.reshape(N, L, 3)
    # z-up world frame from the animal
    paws = np.nanmean(Xf[:, [11, 15, 19, 23]], axis=1)
    up = np.nanmean(Xf[:, 4] - paws, axis=0); up /= np.linalg.norm(up)
    sp = Xf[:, 4]; g = np.nonzero(~np.isnan(sp[:, 0]))[0]
    run = sp[g[-1]] - sp[g[0]]; run -= up * (run @ up); run /= np.linalg.norm(run)
    Rw = np.stack([run, np.cross(up, run), up]); o = np.nanmean(paws, axis=0)
    Xw = ((X - o) @ Rw.T).reshape(N, L, 3)
    for c in range(C):
        R = rodrigues(cams[c, 9:12]); t = cams[c, 12:15]
        cams[c, 9:12] = inv_rodrigues(R @ Rw.T); cams[c, 12:15] = t + R @ o
    # skeleton: reduced coordinates u per frame; dependent angles from the joint equalities
    ind = skeleton.independent_dofs(sk)
    M = {m: i for i, m in enumerate(skeleton.MARKERS)}
    branch = np.ones((N, sk.n_joints))

    def q_from_u(u):
        q = np.zeros((N, sk.nq)); q[:, ind] = u.reshape(N, len(ind))
        for i in range(1, sk.n_links):
            if (3 + 3 * i + 2) not in ind:
                q[:, 3 + 3 * i + 2] = q[:, 5]
        return synth.project_dependents_numpy(sk, q, branch=branch)
    fill = lambda a: np.array([np.interp(np.arange(N), np.nonzero(~np.isnan(a[:, d]))[0], a[~np.isnan(a[:, d]), d]) for d in range(a.shape[1])]).T
    spine, tailb = fill(Xw[:, 4]), fill(Xw[:, 5])
    d = tailb - spine; d /= np.linalg.norm(d, axis=1, keepdims=True)
    q0 = np.zeros((N, sk.nq)); q0[:, 0:3] = 0.5 * (spine + tailb)
    psi = np.unwrap(np.arctan2(d[:, 1], d[:, 0]))
    for i in range(sk.n_links):
        q0[:, 3 + 3 * i + 2] = psi
    q0[:, 4] = -np.arcsin(np.clip(d[:, 2], -1, 1))
    # leg pitches: hanging straight down to start with (theta = pitch of the body), refined below on the 3D points
    wgt = (~np.isnan(Xw[..., 0])).astype(float)
    Xz = np.nan_to_num(Xw)

    def fun_3d(u):
        return ((synth.fk_numpy(sk, q_from_u(u))[0] - Xz) * wgt[..., None]).ravel()
    sp3 = lil_matrix((N * L * 3, N * len(ind)), dtype=int)
    for n in range(N):
        sp3[n * L * 3:(n + 1) * L * 3, n * len(ind):(n + 1) * len(ind)] = 1
    u0 = q0[:, ind].ravel()
    s3 = least_squares(fun_3d, u0, jac_sparsity=sp3, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
    print(f"skeleton fit to the 3D points: rms {np.sqrt(np.mean(s3.fun**2)) * 1e3:.4f} mm, nfev {s3.nfev}")
    nU = N * len(ind)
    okc = ok.transpose(1, 0, 2)                                  # [C, N, L]
    # From here on every leg link is its body rotated about the body's y axis by alpha (the solver's own coordinates): both solution
    # branches of the joint equalities are covered smoothly -- limbs of these gallops swing beyond the horizontal
    lay = synth.leg_layout(sk)
    leg_pos = [list(ind).index(3 + 3 * c + 1) for c, _ in lay]

    def q_from_ua(u):
        U = u.reshape(N, len(ind))
        q = np.zeros((N, sk.nq)); q[:, ind] = U
        for c, _ in lay:
            q[:, 3 + 3 * c + 1] = 0.0
        for i in range(1, sk.n_links):
            if (3 + 3 * i + 2) not in ind:
                q[:, 3 + 3 * i + 2] = q[:, 5]
        q = synth.legs_from_alpha(sk, q, U[:, leg_pos])
        return synth.project_dependents_numpy_hooke(sk, q)
    qs = q_from_u(s3.x)
    ua = qs[:, ind].copy()
    for r, (c, B) in enumerate(lay):
        RB = synth.rot_zyx(qs[:, 3 + 3 * B:6 + 3 * B]); Rc = synth.rot_zyx(qs[:, 3 + 3 * c:6 + 3 * c])
        Mx = np.einsum("nji,njk->nik", RB, Rc)
        ua[:, leg_pos[r]] = np.arctan2(Mx[:, 0, 2], Mx[:, 0, 0])
    s3 = type("S", (), {"x": ua.ravel()})
    q_from_u = q_from_ua

    def fun_joint(p):
        pos = synth.fk_numpy(sk, q_from_u(p[:nU]))[0]
        cp = p[nU:].reshape(C, 15)
        return np.concatenate([np.where(okc[c][..., None], project(cp[c], pos) - np.nan_to_num(uv[:, c]), 0.0).ravel() for c in range(C)])
    spj = lil_matrix((C * N * L * 2, nU + C * 15), dtype=int)
    for c in range(C):
        for n in range(N):
            r0 = (c * N + n) * L * 2
            spj[r0:r0 + L * 2, n * len(ind):(n + 1) * len(ind)] = 1; spj[r0:r0 + L * 2, nU + c * 15:nU + (c + 1) * 15] = 1
    pj = np.concatenate([s3.x, cams.ravel()])
    for rnd in range(6):
        sj = least_squares(fun_joint, pj, jac_sparsity=spj, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
        pj = sj.x
        print(f"joint refinement round {rnd}: rms {np.sqrt((sj.fun**2).sum() / (2 * ok.sum())):.3e} px, max {np.abs(sj.fun).max():.3e} px, nfev {sj.nfev}")
        if np.abs(sj.fun).max() < 1e-6:
            break
    q = q_from_u(pj[:nU]); cams = pj[nU:].reshape(C, 15)
    out = os.path.join(ROOT, "tests", "golden", OUT)
    np.savez_compressed(out, uv=uv, q=q, cams=cams, seq=SEQ, animal=ANIMAL, rms_px=np.sqrt((sj.fun**2).sum() / (2 * ok.sum())), max_px=np.abs(sj.fun).max())
    print("wrote", out); print("intrinsics per camera:\n", np.round(cams[:, :9], 4))


if __name__ == "__main__":
    main()
