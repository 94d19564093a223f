"""Learned priors of the monocular "data-driven" model (config 3): the Gaussian-mixture pose prior
(acinoset_misc.py:680-714, acinoset_models.py:277-300) and the linear autoregressive motion prior
(acinoset_misc.py:291-336, acinoset_models.py:173-274).  The fitted numbers are package data
(data/priors_full_pose.npz, produced by tools/fit_priors.py with the reference's own recipe); this module only
packs them into the C-ABI struct `cpe_priors`."""
import os

import numpy as np

from . import abi

_NPZ = os.path.join(os.path.dirname(__file__), "data", "priors_full_pose.npz")


def load_priors(pose: bool = True, motion: bool = True, path: str = _NPZ) -> abi.Priors:
    z = np.load(path)
    pr = abi.Priors()
    if pose:
        w, mu, cov = z["gmm_weights"], z["gmm_means"], z["gmm_covariances"]
        K, D = mu.shape
        assert K <= abi.MAX_GMM and D <= abi.NX
        pr.gmm_k, pr.gmm_dim = K, D
        for k in range(K):
            sign, logdet = np.linalg.slogdet(cov[k])
            assert sign > 0
            pr.gmm_logw[k] = np.log(w[k]) - 0.5 * (D * np.log(2 * np.pi) + logdet)     # log of w_k / sqrt(det(2 pi Sigma_k))
            P = np.linalg.inv(cov[k])
            P = 0.5 * (P + P.T)
            for i in range(D):
                pr.gmm_mu[k][i] = mu[k, i]
                for j in range(D):
                    pr.gmm_P[k][i][j] = P[i, j]
    if motion:
        coef, b, var = z["lr_coef"], z["lr_intercept"], z["lr_error_variance"]
        W = int(z["lr_window"])
        assert W <= abi.MAX_WINDOW and coef.shape == (abi.NX, W * abi.NX)
        pr.lr_window = W
        for p in range(abi.NX):
            pr.lr_b[p] = b[p]
            pr.lr_w[p] = 0.0 if var[p] == 0 else 1.0 / var[p]                           # acinoset_misc.py:307
            for j in range(W * abi.NX):
                pr.lr_coef[p][j] = coef[p, j]
    return pr
