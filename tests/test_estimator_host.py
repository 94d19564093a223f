"""Host logic of the estimator mirror (no GPU): file formats in, measurement tensorisation
(acinoset_misc.py:211-256), initial trajectory estimate (acinoset_misc.py:381-456)."""
import numpy as np
import pytest

from cheetah_pose_estimation_amd import estimator as E, skeleton, synth
from dataset_util import write_dataset


def test_scene_dlc_and_measurement_tensors(tmp_path):
    info = write_dataset(str(tmp_path), N=20)
    import os
    ddir = os.path.join(str(tmp_path), info["data_path"])
    k, d, r, t, res, n_cams, fpath = E.find_scene_file(ddir)
    assert n_cams == 6 and res == (synth.IMG_W, synth.IMG_H) and k.shape == (6, 3, 3) and d.reshape(6, -1).shape == (6, 4)
    tables = [E.load_dlc_table(p) for p in E.dlc_paths(os.path.join(ddir, "dlc"))]
    assert len(tables) == 6 and tables[0][1].shape == (28, 75)
    meas, weight = E.build_measurements(tables, 4, 24, [{"cam": 1, "frame": 2}], 6, 0.5, False)
    assert meas.shape == (20, 6, 24, 2) and weight.shape == (20, 6, 24)
    sig = skeleton.measurement_sigma(24)
    for l, m in enumerate(skeleton.MARKERS):
        j = skeleton.DLC_INDEX[m]
        assert np.array_equal(meas[:, 0, l, 0], tables[0][1][4:24, 3 * j])                 # camera 0: no offset
        assert np.array_equal(meas[:, 1, l, 1], tables[1][1][2:22, 3 * j + 1])             # camera 1: rows shifted by its sync offset
        lik = tables[0][1][4:24, 3 * j + 2]
        assert np.array_equal(weight[:, 0, l], np.where(lik > 0.5, 1.0 / sig[l], 0.0))
    # monocular selection keeps one camera
    m1, w1 = E.build_measurements(tables, 4, 24, None, 6, 0.5, False, cam_idx=2)
    assert m1.shape == (20, 1, 24, 2) and np.array_equal(m1[:, 0], E.build_measurements(tables, 4, 24, None, 6, 0.5, False)[0][:, 2])


def test_initial_trajectory_estimate(tmp_path):
    info = write_dataset(str(tmp_path), N=30, noise_px=0.5)
    import torch
    if torch.cuda.is_available():
        pytest.skip("host-only test")
    import os
    ddir = os.path.join(str(tmp_path), info["data_path"])
    k, d, r, t, res, n_cams, fpath = E.find_scene_file(ddir)
    tables = [E.load_dlc_table(p) for p in E.dlc_paths(os.path.join(ddir, "dlc"))]
    params = E.TrajectoryParams(ddir, 4, 34, 30, 0.5, None, False, False, False, False)
    scene = E.Scene(fpath, k, d.reshape(6, -1), r, t, res, 120.0, 6, None)
    sk = info["sk"]
    x, y, z, psi = E.create_trajectory_estimate(tables, params, scene, 2 * abs(sk.marker_off[5][0]))
    qt = info["q_true"]
    # the reference's rule puts the base at spine + L/2 along x; truth base is within a few cm of that
    assert np.abs(y[4:34] - qt[4:34, 1]).max() < 0.05 and np.abs(z[4:34] - qt[4:34, 2]).max() < 0.08
    assert np.abs(x[4:34] - qt[4:34, 0]).max() < 0.45
    assert np.abs(np.unwrap(psi[4:34]) - np.pi).max() < 0.2
    # undistortion inverts the projection model
    cam = info["cams"][0]
    K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1.0]])
    P = info["pos_true"][5]
    uv, _ = synth.project_numpy(cam, P)
    Xc = P @ np.array(cam.R[:]).reshape(3, 3).T + np.array(cam.t[:])
    assert np.abs(E._undistort_fisheye(uv, K, np.array(cam.D[:])) - Xc[:, :2] / Xc[:, 2:]).max() < 1e-9
