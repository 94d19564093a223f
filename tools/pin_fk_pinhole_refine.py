#!/usr/bin/env python3
"""Developer tool: continue the joint refinement of tools/pin_fk_pinhole.py from a stored result (same arguments + the .npz to start from).  The world
frame is a gauge freedom of that fit (cameras and animal can move together); here the first camera's pose is held at its stored value, which removes the
six flat directions that slowed the shiraz fit down.
usage: python tools/pin_fk_pinhole_refine.py <sequence> <animal> <out.npz> <start.npz> [rounds]"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
START = sys.argv[4]
ROUNDS = int(sys.argv[5]) if len(sys.argv) > 5 else 12
import pin_fk_pinhole as P  # noqa: E402  (reads sys.argv[1:4])
from cheetah_pose_estimation_amd import skeleton, synth  # noqa: E402


def main():
    Z = np.load(START)
    uv = P.load_uv()
    N, C, L, _ = uv.shape
    ok = ~np.isnan(uv).any(-1)
    okc = ok.transpose(1, 0, 2)
    sk = skeleton.build_skeleton(f"{P.ANIMAL}-02", 24, kinetic_dataset=True)
    ind = skeleton.independent_dofs(sk)
    lay = synth.leg_layout(sk)
    leg_pos = [list(ind).index(3 + 3 * c + 1) for c, _ in lay]
    qs, cams0 = Z["q"], Z["cams"].copy()
    ua = qs[:, ind].copy()
    for r, (c, B) in enumerate(lay):
        RB = synth.rot_zyx(qs[:, 3 + 3 * B:6 + 3 * B]); Rc = synth.rot_zyx(qs[:, 3 + 3 * c:6 + 3 * c])
        Mx = np.einsum("nji,njk->nik", RB, Rc)
        ua[:, leg_pos[r]] = np.arctan2(Mx[:, 0, 2], Mx[:, 0, 0])

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
    nU = N * len(ind)
    free = np.ones(C * 15, bool); free[9:15] = False                 # camera 1's pose is the gauge
    if os.environ.get("PIN_RADIAL_ONLY"):                            # the reference's pinhole model has no tangential terms (acinoset_misc.py:1682-1696)
        for c in range(C):
            free[c * 15 + 6] = free[c * 15 + 7] = False
            cams0[c, 6] = cams0[c, 7] = 0.0

    def cams_of(pc):
        cp = cams0.ravel().copy(); cp[free] = pc
        return cp.reshape(C, 15)

    def fun(p):
        pos = synth.fk_numpy(sk, q_from_ua(p[:nU]))[0]
        cp = cams_of(p[nU:])
        return np.concatenate([np.where(okc[c][..., None], P.project(cp[c], pos) - np.nan_to_num(uv[:, c]), 0.0).ravel() for c in range(C)])
    col_of = np.cumsum(free) - 1
    spj = lil_matrix((C * N * L * 2, nU + int(free.sum())), dtype=int)
    for c in range(C):
        cols = [nU + col_of[c * 15 + k] for k in range(15) if free[c * 15 + k]]
        for n in range(N):
            r0 = (c * N + n) * L * 2
            spj[r0:r0 + L * 2, n * len(ind):(n + 1) * len(ind)] = 1
            for cc in cols:
                spj[r0:r0 + L * 2, cc] = 1
    pj = np.concatenate([ua.ravel(), cams0.ravel()[free]])
    r0_ = fun(pj)
    print(f"start: rms {np.sqrt((r0_**2).sum() / (2 * ok.sum())):.3e} px, max {np.abs(r0_).max():.3e} px", flush=True)
    for rnd in range(ROUNDS):
        sj = least_squares(fun, pj, jac_sparsity=spj, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
        pj = sj.x
        print(f"round {rnd}: rms {np.sqrt((sj.fun**2).sum() / (2 * ok.sum())):.3e} px, max {np.abs(sj.fun).max():.3e} px, nfev {sj.nfev}", flush=True)
        q = q_from_ua(pj[:nU]); cams = cams_of(pj[nU:])
        out = P.OUT if os.path.isabs(P.OUT) else os.path.join(P.ROOT, "tests", "golden", P.OUT)
        np.savez_compressed(out, uv=uv, q=q, cams=cams, seq=P.SEQ, animal=P.ANIMAL, rms_px=np.sqrt((sj.fun**2).sum() / (2 * ok.sum())), max_px=np.abs(sj.fun).max())
        if np.abs(sj.fun).max() < 1e-5:
            break
    print("wrote", out); print(np.round(cams[:, :9], 6))


if __name__ == "__main__":
    main()
