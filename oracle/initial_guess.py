"""CPU checker for the initial-guess ingestion kernels (cpe_triangulate).  TEST INFRASTRUCTURE ONLY, like the rest of
oracle/: numpy restatements of what the reference obtains from OpenCV in triangulate_points[_fisheye]
(acinoset_misc.py:1432-1453) -- cv.fisheye.undistortPoints / cv.undistortPoints and cv.triangulatePoints.  OpenCV is not
installed here, so the pins are the model identities themselves (tests/test_estimator_host.py): undistortion inverts the
projection model of acinoset_misc.py:1663-1696, and triangulating two exact projections returns the 3D point."""
import numpy as np


def undistort_fisheye(uv, K, D):
    x = (uv[:, 0] - K[0, 2]) / K[0, 0]; y = (uv[:, 1] - K[1, 2]) / K[1, 1]
    rd = np.sqrt(x * x + y * y)
    th = rd.copy()
    for _ in range(20):
        t2 = th * th
        f = th * (1 + D[0] * t2 + D[1] * t2**2 + D[2] * t2**3 + D[3] * t2**4) - rd
        df = 1 + 3 * D[0] * t2 + 5 * D[1] * t2**2 + 7 * D[2] * t2**3 + 9 * D[3] * t2**4
        th = th - f / df
    s = np.where(rd > 1e-12, np.tan(th) / np.maximum(rd, 1e-12), 1.0)
    return np.stack([x * s, y * s], axis=1)


def undistort_pinhole(uv, K, D):
    x0 = (uv[:, 0] - K[0, 2]) / K[0, 0]; y0 = (uv[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(20):
        r2 = x * x + y * y
        g = 1 + D[0] * r2 + D[1] * r2**2 + D[2] * r2**3
        x, y = x0 / g, y0 / g
    return np.stack([x, y], axis=1)


def triangulate(n1, n2, R1, t1, R2, t2):
    """linear (DLT) two-view triangulation of normalised image points: smallest right singular vector, dehomogenised"""
    P1 = np.hstack([R1, t1.reshape(3, 1)]); P2 = np.hstack([R2, t2.reshape(3, 1)])
    out = np.empty((len(n1), 3))
    for i in range(len(n1)):
        A = np.stack([n1[i, 0] * P1[2] - P1[0], n1[i, 1] * P1[2] - P1[1], n2[i, 0] * P2[2] - P2[0], n2[i, 1] * P2[2] - P2[1]])
        X = np.linalg.svd(A)[2][-1]
        out[i] = X[:3] / X[3]
    return out


def backproject(n1, R, t, depth):
    Xc = depth * np.c_[n1, np.ones(len(n1))]
    return (Xc - t.reshape(1, 3)) @ R


def camera_arrays(cam):
    """(K, D, R, t, fisheye?) of an abi.Camera"""
    K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1.0]])
    return K, np.array(cam.D[:]), np.array(cam.R[:]).reshape(3, 3), np.array(cam.t[:]), cam.model == 0


def triangulate_pixels(cams, cam_a, cam_b, uv_a, uv_b, depth=3.0):
    """the checker of cpe_triangulate: same arguments, numpy arrays"""
    out = np.empty((len(cam_a), 3))
    for i in range(len(cam_a)):
        Ka, Da, Ra, ta, fa = camera_arrays(cams[int(cam_a[i])])
        na = (undistort_fisheye if fa else undistort_pinhole)(np.asarray(uv_a[i], dtype=float).reshape(1, 2), Ka, Da)
        if cam_b[i] < 0:
            out[i] = backproject(na, Ra, ta, depth)[0]
        else:
            Kb, Db, Rb, tb, fb = camera_arrays(cams[int(cam_b[i])])
            nb = (undistort_fisheye if fb else undistort_pinhole)(np.asarray(uv_b[i], dtype=float).reshape(1, 2), Kb, Db)
            out[i] = triangulate(na, nb, Ra, ta, Rb, tb)[0]
    return out
