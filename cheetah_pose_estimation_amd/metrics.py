"""Trajectory comparison metrics of the post-processing step: host-side reductions over [N, L, 3] arrays,
as `dataset_post_process` (run_dataset.py:365-632) uses them.  Counterparts: acinoset_misc.py `rmse` :93-98,
`traj_smoothness` :1170-1176, `traj_error` :1179-1199.  Pinned by tests/golden/contacts_metrics_golden.npz, which holds the
outputs of the reference's own functions on seeded inputs (tools/gen_golden.py)."""
from typing import Dict, Tuple

import numpy as np

from . import skeleton


def rmse(predictions, targets) -> float:
    """root of the NaN-ignoring mean of the squared differences"""
    d = np.asarray(predictions, dtype=float) - np.asarray(targets, dtype=float)
    return float(np.sqrt(np.nanmean(d * d)))


def traj_smoothness(X, Y) -> float:
    """mean absolute difference of the frame-to-frame marker displacements of two trajectories [N, L, 3] (metres)"""
    step = lambda A: np.linalg.norm(np.diff(np.asarray(A, dtype=float), axis=0), axis=2)
    return float(np.mean(np.abs(step(X) - step(Y))))


def traj_error(X, Y, centered: bool = False) -> Tuple[Dict[str, float], np.ndarray, float]:
    """({marker: mean position error in mm}, per-frame mean error [N] in mm, smoothness error in mm).  `centered` removes the
    per-frame centroid of each trajectory first (MPJPE); otherwise the errors are absolute (MPE).  Unlike the reference the
    inputs are left untouched (it subtracts the centroids in place)."""
    X = np.array(X, dtype=float)
    Y = np.array(Y, dtype=float)
    smooth_mm = 1000.0 * traj_smoothness(X, Y)              # on the uncentred trajectories, as the reference computes it first
    if centered:
        X -= X.mean(axis=1, keepdims=True)
        Y -= Y.mean(axis=1, keepdims=True)
    dist = np.linalg.norm(X - Y, axis=2)
    per_marker = 1000.0 * dist.mean(axis=0)
    names = list(skeleton.MARKERS) + [f"extra{i}" for i in range(max(0, dist.shape[1] - len(skeleton.MARKERS)))]
    return {names[l]: float(per_marker[l]) for l in range(dist.shape[1])}, 1000.0 * dist.mean(axis=1), smooth_mm
