#!/usr/bin/env python3
"""Fit the learned priors of the monocular "data-driven" model (build container only).

The reference fits both models at run time with scikit-learn on its shipped pose tables
(acinoset_models.py:173-274 MotionModel, :277-300 PoseModelGMM; datasets models/data-driven/dataset_full_pose.h5,
whose .csv twin has the same content).  This script repeats exactly those fits --
    MultiTaskLasso(alpha=1e-2, random_state=42, max_iter=20000) on the window-4 supervised framing of x (28),
    error_variance = var(y - y_pred) over the training rows                       (acinoset_models.py:208, :219-220)
    GaussianMixture(n_components=5, random_state=42, max_iter=20000) on x[6:28]   (acinoset_models.py:294)
-- and stores the NUMBERS (coef_, intercept_, error_variance, weights_, means_, covariances_) in
cheetah_pose_estimation_amd/data/priors_full_pose.npz, plus golden evaluations of sklearn's own predict / score on
a few dataset rows in tests/golden/priors_golden.npz.  scikit-learn version drift vs. the authors' environment makes
parity with THEIR fitted numbers unpinned; the fits are deterministic here.
"""
import os
import sys

import numpy as np
import pandas as pd
from sklearn.linear_model import MultiTaskLasso
from sklearn.mixture import GaussianMixture

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = "/root/reference/models/data-driven"
NUM_VARS, EXT_DIM, WINDOW, NCOMP = 28, 6, 4, 5


def series_to_supervised(data: np.ndarray, n_in: int) -> np.ndarray:
    """standard shift framing (common/py_utils data_ops.series_to_supervised is absent):
    columns t-n_in ... t-1 then t, rows with a full history only"""
    n = data.shape[0]
    if n <= n_in:
        return np.zeros((0, data.shape[1] * (n_in + 1)))
    return np.concatenate([data[i:n - n_in + i] for i in range(n_in + 1)], axis=1)


def supervised_xy(df: pd.DataFrame, window: int):
    idx = np.where(df.index.values == 0)[0]
    data = df.iloc[:, :NUM_VARS].to_numpy()
    parts, end = [], 0
    for b, e in zip(idx, idx[1:]):
        parts.append(series_to_supervised(data[b:e], window)); end = e
    parts.append(series_to_supervised(data[end:], window))
    xy = np.concatenate(parts)
    return xy[:, :NUM_VARS * window], xy[:, NUM_VARS * window:]


def main():
    df = pd.read_csv(os.path.join(REF, "dataset_full_pose.csv"), index_col=0)
    X, y = supervised_xy(df, WINDOW)
    lr = MultiTaskLasso(alpha=1e-2, random_state=42, max_iter=20000).fit(X, y)
    err_var = np.var(y - lr.predict(X), axis=0)
    Xg = df.iloc[:, EXT_DIM:NUM_VARS].to_numpy()
    gmm = GaussianMixture(n_components=NCOMP, random_state=42, max_iter=20000).fit(Xg)
    out = os.path.join(ROOT, "cheetah_pose_estimation_amd", "data", "priors_full_pose.npz")
    np.savez_compressed(out, lr_coef=lr.coef_, lr_intercept=lr.intercept_, lr_error_variance=err_var, lr_window=WINDOW,
                        gmm_weights=gmm.weights_, gmm_means=gmm.means_, gmm_covariances=gmm.covariances_)
    print("wrote", out, "LR non-zeros", np.count_nonzero(lr.coef_), "of", lr.coef_.size, "| GMM converged", gmm.converged_,
          "train log-lik", gmm.score(Xg))
    # golden evaluations by sklearn itself
    rows = np.arange(0, X.shape[0], 97)[:12]
    gold = os.path.join(ROOT, "tests", "golden", "priors_golden.npz")
    np.savez_compressed(gold, lr_X=X[rows], lr_y=y[rows], lr_pred=lr.predict(X[rows]), gmm_x=Xg[rows],
                        gmm_logpdf=gmm.score_samples(Xg[rows]))
    print("wrote", gold)
    # a second size (the reference's grid search varies both, run_dataset.py:814-915): 3 components, window 2, plain least squares -- fitted by the
    # PACKAGE's own priors.fit_priors, with scikit-learn's evaluations of independently fitted models beside the numbers
    sys.path.insert(0, ROOT)
    from cheetah_pose_estimation_amd import priors as P
    import tempfile
    from sklearn.linear_model import LinearRegression
    path = P.fit_priors(3, 2, False, dataset=os.path.join(REF, "dataset_full_pose.csv"), cache_dir=tempfile.mkdtemp())
    z = dict(np.load(path))
    X2, y2 = supervised_xy(df, 2)
    lr2 = LinearRegression().fit(X2, y2)
    gmm2 = GaussianMixture(n_components=3, random_state=42, max_iter=20000).fit(Xg)
    rows2 = np.arange(0, X2.shape[0], 97)[:12]
    gold2 = os.path.join(ROOT, "tests", "golden", "priors_k3_w2_dense.npz")
    np.savez_compressed(gold2, lr_X=X2[rows2], lr_pred=lr2.predict(X2[rows2]), gmm_x=Xg[rows2], gmm_logpdf=gmm2.score_samples(Xg[rows2]), **z)
    print("wrote", gold2)


if __name__ == "__main__":
    main()
