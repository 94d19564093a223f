"""Host-side mirror of the reference's estimator API (layer L3, acinoset_opt.py) on top of the HIP back end.

Same entry points, argument meaning and return convention as the reference so that its drivers
(`run_dataset.py:1143-1231` etc., `tests.ipynb`) call in unchanged:

    est = init_trajectory(root_dir, data_path, cheetah_name, kinetic_dataset, solver_path, ...)   # acinoset_opt.py:413-430
    ok  = estimate_kinematics(est, ...)                                                            # acinoset_opt.py:539-547

and the on-disk contract is kept: `fte.pickle` (keys of acinoset_opt.py:330-361) and `cam{i}_fte.csv` in
DeepLabCut layout (acinoset_misc.py:1346-1407).  Where the reference builds a Pyomo model and spawns IPOPT
(acinoset_opt.py:508-525, :611-617) this module fills dense tensors and calls `cpe_solve` through the C ABI.
Everything numerical on the solve path runs on the GPU; numpy here only moves data in and out of files.
"""
import json
import os
import pickle
from dataclasses import dataclass, field
from glob import glob
from time import time
from typing import Sequence, Dict, List, Optional, Tuple

import numpy as np

from . import _lib, abi, skeleton


class _ArrayOnlyUnpickler(pickle.Unpickler):
    """`fte.pickle` reader that executes nothing from the file: only the globals a dict of numpy arrays / floats needs are
    resolvable; anything else (a `fte.pickle` may come from an untrusted tree laid out like the reference's data) raises."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"fte.pickle: global {module}.{name} is not allowed (arrays, numbers, str, dict, list only)")


def load_result_pickle(path: str) -> dict:
    """read back a result file written by CheetahEstimator.save (plain dict of arrays and numbers)"""
    with open(path, "rb") as fh:
        return _ArrayOnlyUnpickler(fh).load()


# ---- data classes mirroring acinoset_misc.py:40-73 ---------------------------------------------------
@dataclass
class TrajectoryParams:
    data_dir: str
    start_frame: int
    end_frame: int
    total_length: int
    dlc_thresh: float
    sync_offset: Optional[List[Dict]]
    hand_labeled_data: bool
    kinetic_dataset: bool
    enable_shutter_delay_estimation: bool
    enable_ppms: bool


@dataclass
class Scene:
    scene_fpath: str
    k_arr: np.ndarray
    d_arr: np.ndarray
    r_arr: np.ndarray
    t_arr: np.ndarray
    cam_res: Tuple[int, int]
    fps: float
    n_cams: int
    cam_idx: Optional[int]


# ---- file formats ---------------------------------------------------------------------------------------
def load_scene(fpath: str):
    """`*_cam_scene_sba.json` (acinoset_misc.py:1496-1516)."""
    with open(fpath, "r", encoding="utf-8") as f:
        data = json.load(f)
    cams = data["cameras"]
    arr = lambda k: np.array([c[k] for c in cams], dtype=np.float64)
    return arr("k"), arr("d"), arr("r"), arr("t"), tuple(data["camera_resolution"])


def find_scene_file(dir_path: str, scene_fname: Optional[str] = None):
    """walk up from the sequence directory to `extrinsic_calib/` (acinoset_misc.py:1519-1544)."""
    if scene_fname is None:
        n_cams = len(glob(os.path.join(dir_path, "cam[1-9].mp4")))
        scene_fname = f"{n_cams}_cam_scene_sba.json" if n_cams else "[1-9]_cam_scene*.json"
    d = dir_path
    while d and d != os.path.sep:
        files = sorted(f for f in glob(os.path.join(d, "extrinsic_calib", scene_fname)) if "before_corrections" not in f)
        if files:
            k, dd, r, t, res = load_scene(files[-1])
            return k, dd, r, t, res, int(os.path.basename(files[-1])[0]), files[-1]
        nd = os.path.dirname(d)
        if nd == d:
            break
        d = nd
    raise FileNotFoundError(os.path.join("extrinsic_calib", scene_fname))


def load_dlc_table(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """One DeepLabCut file -> (frame index [F], values [F, 75]) with columns (x, y, likelihood) x 25 body parts.
    `.h5` needs PyTables (as in the reference, acinoset_misc.py:206); the `.csv` twin DeepLabCut writes beside
    it (3 header rows: scorer / bodyparts / coords) is read without it."""
    if path.endswith(".h5"):
        import pandas as pd
        df = pd.read_hdf(path)
        return np.asarray(df.index), df.to_numpy(dtype=np.float64)
    rows = np.genfromtxt(path, delimiter=",", skip_header=3)
    return rows[:, 0].astype(np.int64), rows[:, 1:].astype(np.float64)


def load_hand_labeled_table(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """One hand-labelled file (`dlc_hand_labeled/cam?.h5`, acinoset_misc.py:1545-1552) in the layout of `load_dlc_table`: (frame index [F], values
    [F, 75]).  The file holds (x, y) per body part and is indexed by image names (`.../img012.png`): the reference numbers its rows from the
    digits [3:6] of the first name to those of the last.  Likelihood 1 for a labelled point; a point without a label (NaN) gets likelihood 0 --
    DEVIATION: the reference compares with `== np.nan` (acinoset_misc.py:221, :247), which is never true, so its NLP would be handed the NaN.
    `.h5` needs PyTables; the `.csv` twin DeepLabCut writes beside it (`CollectedData_*.csv`: header rows scorer / bodyparts / coords, one or
    three leading name columns) is read without it."""
    if path.endswith(".h5"):
        import pandas as pd
        df = pd.read_hdf(path)
        names = [ix[-1] if isinstance(ix, tuple) else str(ix) for ix in df.index]
        xy = df.to_numpy(dtype=np.float64)
    else:
        import csv
        with open(path, "r", encoding="utf-8", newline="") as f:
            rows = list(csv.reader(f))
        coords = rows[2]
        first = next(i for i, c in enumerate(coords) if c.strip() in ("x", "y"))
        body = [r for r in rows[3:] if r]
        names = [r[first - 1] for r in body]
        xy = np.array([[float(v) if v.strip() not in ("", "nan", "NaN") else np.nan for v in r[first:]] for r in body], dtype=np.float64)
    base = lambda nm: os.path.basename(str(nm).replace("\\", "/"))
    start, end = int(base(names[0])[3:6]), int(base(names[-1])[3:6])
    assert end - start + 1 == xy.shape[0], f"{path}: {xy.shape[0]} rows for frames {start}..{end}"
    n_parts = xy.shape[1] // 2
    vals = np.zeros((xy.shape[0], 3 * n_parts))
    vals[:, 0::3], vals[:, 1::3] = xy[:, 0::2], xy[:, 1::2]
    ok = np.isfinite(xy[:, 0::2]) & np.isfinite(xy[:, 1::2])
    vals[:, 2::3] = ok
    vals[:, 0::3][~ok] = 0.0; vals[:, 1::3][~ok] = 0.0
    return np.arange(start, end + 1, dtype=np.int64), vals


def dlc_paths(dlc_dir: str) -> List[str]:
    h5 = sorted(glob(os.path.join(dlc_dir, "*.h5")))
    try:
        import tables  # noqa: F401
        if h5:
            return h5
    except Exception:
        pass
    csv = sorted(glob(os.path.join(dlc_dir, "*.csv")))
    return csv if csv else h5


def build_measurements(tables, start_frame: int, end_frame: int, sync_offset, n_cams: int, dlc_thresh: float,
                       kinetic_dataset: bool, cam_idx: Optional[int] = None, device: int = 0):
    """meas[N,C,24,2] and meas_err_weight[N,C,24] exactly as `init_measurements` / `init_meas_weights` fill the
    Pyomo params (acinoset_misc.py:211-256): row (n + start_frame - sync_offset[c]) of camera c, DLC column of
    each marker (`get_dlc_marker_indices`), weight 1/R_pw[0][l] if likelihood > dlc_thresh else 0.  The gather runs on
    the GPU (cpe_tensorise_dlc), one launch per camera table."""
    off = [0] * n_cams
    if sync_offset is not None:
        for o in sync_offset:
            off[o["cam"]] = o["frame"]
    N = end_frame - start_frame
    cams = list(range(n_cams)) if cam_idx is None else [cam_idx]
    sigma = skeleton.measurement_sigma(24, kinetic_dataset)
    col = [skeleton.DLC_INDEX[m] for m in skeleton.MARKERS]
    from .synth import make_cameras
    h = _lib.Handle(skeleton.build_skeleton("phantom", 24), make_cameras(1), device=device)        # only the marker count is used
    try:
        return h.tensorise_dlc_host([tables[c][1] for c in cams], [start_frame - off[c] for c in cams], col, 1.0 / sigma, dlc_thresh, N)
    finally:
        h.close()


def load_pairwise_table(path: str):
    """pairwise predictions of ONE camera (`dlc_pw/cam?DLC_....pickle`, acinoset_misc.py:202-205): per table row the DLC pose row [75] and the
    pairwise offsets `pws` [1, 25, 25, 2] (offset from body part a to body part b).  Accepted here: an `.npz` with arrays `pose [rows, 75]`,
    `pws [rows, 25, 25, 2]`, or the reference's pickle read with the arrays-only unpickler (a list of {"pose", "pws"} records)."""
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return np.asarray(z["pose"], dtype=np.float64), np.asarray(z["pws"], dtype=np.float64)
    recs = load_result_pickle(path)
    pose = np.array([np.asarray(r["pose"], dtype=np.float64).ravel() for r in recs])
    pws = np.array([np.asarray(r["pws"], dtype=np.float64).reshape(25, 25, 2) for r in recs])
    return pose, pws


def build_pairwise_measurements(pw_tables, start_frame: int, end_frame: int, sync_offset, n_cams: int, dlc_thresh: float, kinetic_dataset: bool,
                                cam_idx: Optional[int] = None):
    """meas[N, 2, C, 24, 2] and weight[N, 2, C, 24] of the two pairwise pseudo-measurements w = 2, 3 as `init_measurements` / `init_meas_weights`
    fill them (acinoset_misc.py:211-256): marker l is predicted from body part b = pair_dict[marker][w - 2] as pose[b] + pws[b, dlc index of the
    marker]; weight 1 / R_pw[w - 1][l] if the likelihood of b exceeds the threshold.  Host gather over a few thousand numbers."""
    off = [0] * n_cams
    if sync_offset is not None:
        for o in sync_offset:
            off[o["cam"]] = o["frame"]
    N = end_frame - start_frame
    cams = list(range(n_cams)) if cam_idx is None else [cam_idx]
    meas = np.zeros((N, 2, len(cams), 24, 2)); weight = np.zeros((N, 2, len(cams), 24))
    own = np.array([skeleton.DLC_INDEX[m] for m in skeleton.MARKERS])
    for w in (0, 1):
        sig = skeleton.pairwise_sigma(w + 1, kinetic_dataset)
        src = np.array([skeleton.PAIRWISE[m][w] for m in skeleton.MARKERS])
        for ci, c in enumerate(cams):
            pose, pws = pw_tables[c]
            rows = np.arange(N) + start_frame - off[c]
            ok = (rows >= 0) & (rows < pose.shape[0])
            r = np.clip(rows, 0, pose.shape[0] - 1)
            xy = np.stack([pose[r][:, 0::3][:, src] + pws[r][:, src, own, 0], pose[r][:, 1::3][:, src] + pws[r][:, src, own, 1]], axis=-1)
            lik = pose[r][:, 2::3][:, src]
            good = ok[:, None] & np.isfinite(xy).all(-1)
            meas[:, w, ci] = np.where(good[..., None], xy, 0.0)
            weight[:, w, ci] = np.where(good & (lik > dlc_thresh), 1.0 / sig[None, :], 0.0)
    return meas, weight


def scene_cameras(scene: Scene, kinetic_dataset: bool, hand_labeled: bool = False):
    """abi.Camera array from the scene; multipliers [1,1,.6,.6] for the kinetic dataset (acinoset_misc.py:462-464) -- not with hand-labelled
    points, whose squared loss has no multiplier (:471-474)."""
    idx = list(range(scene.n_cams)) if scene.cam_idx is None else [scene.cam_idx]
    cams = (abi.Camera * len(idx))()
    mult = [1.0, 1.0, 0.6, 0.6] if kinetic_dataset and not hand_labeled else [1.0] * 6
    for j, c in enumerate(idx):
        cam = cams[j]
        cam.model = abi.CAM_PINHOLE if kinetic_dataset else abi.CAM_FISHEYE
        K = scene.k_arr[c]
        cam.fx, cam.fy, cam.cx, cam.cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        D = np.asarray(scene.d_arr[c]).ravel()
        for i in range(min(4, len(D))):
            cam.D[i] = D[i]
        for i in range(9):
            cam.R[i] = np.asarray(scene.r_arr[c]).reshape(-1)[i]
        for i in range(3):
            cam.t[i] = np.asarray(scene.t_arr[c]).reshape(-1)[i]
        cam.mult = mult[c] if c < len(mult) else 1.0
    return cams


# ---- initial guess (acinoset_misc.py:381-456) ------------------------------------------------------------------------
def create_trajectory_estimate(tables, params: TrajectoryParams, scene: Scene, base_length: float, device: int = 0):
    """x, y, z, psi initial estimate from the `spine` marker (acinoset_misc.py:381-456): pairwise triangulation
    over the camera ring (multi-view) or back-projection to 3 m depth (monocular) -- both on the GPU, cpe_triangulate --
    then the per-frame mean over the pairs, a cubic / linear smoothing spline, heading from finite differences (+pi: the
    skeleton's head points along -x)."""
    from scipy.interpolate import UnivariateSpline
    kin = params.kinetic_dataset
    col = skeleton.DLC_INDEX["spine"]
    off = [0] * scene.n_cams
    if params.sync_offset is not None:
        for o in params.sync_offset:
            off[o["cam"]] = o["frame"]
    obs = {}
    for c in range(scene.n_cams):
        idx, vals = tables[c]
        ok = vals[:, 3 * col + 2] > params.dlc_thresh
        obs[c] = {int(f) + off[c]: vals[i, 3 * col:3 * col + 2] for i, f in enumerate(idx) if ok[i]}
    # one flat list of (frame, camera a, camera b, pixel a, pixel b) records for a single launch
    rec_f, rec_a, rec_b, rec_ua, rec_ub = [], [], [], [], []
    if scene.cam_idx is None:
        ncam = 2 if kin else scene.n_cams                       # kinetic dataset: near-side cameras only (:399-401)
        for a, b in [(i % ncam, (i + 1) % ncam) for i in range(ncam)]:
            for f in sorted(set(obs[a]) & set(obs[b])):
                rec_f.append(f); rec_a.append(a); rec_b.append(b); rec_ua.append(obs[a][f]); rec_ub.append(obs[b][f])
    else:
        c = scene.cam_idx
        for f in sorted(obs[c]):
            rec_f.append(f); rec_a.append(c); rec_b.append(-1); rec_ua.append(obs[c][f]); rec_ub.append(obs[c][f])
    if not rec_f:
        raise ValueError("no usable spine detections for the initial trajectory estimate")
    f32 = lambda a: np.array(a, dtype=np.float32).astype(np.float64)          # the reference hands float32 pixels to OpenCV (:1467-1468)
    all_cams = scene_cameras(Scene(scene.scene_fpath, scene.k_arr, scene.d_arr, scene.r_arr, scene.t_arr, scene.cam_res,
                                   scene.fps, scene.n_cams, None), kin)
    h = _lib.Handle(skeleton.build_skeleton("phantom", 24), all_cams, device=device)      # only the camera table is used
    try:
        X = h.triangulate_host(rec_a, rec_b, f32(rec_ua), f32(rec_ub), depth=3.0)
    finally:
        h.close()
    rec_f = np.array(rec_f)
    frames = np.unique(rec_f)
    xyz = np.array([X[rec_f == f].mean(axis=0) for f in frames])             # groupby(frame).mean() (:1491)
    xyz[:, 0] += base_length / 2.0                                   # :424
    k = 1 if kin else 3
    fr = np.arange(params.end_frame)
    est = [np.asarray(UnivariateSpline(frames, xyz[:, d], k=k)(fr)) for d in range(3)]
    psi = np.arctan2(np.diff(est[1]) * scene.fps, np.diff(est[0]) * scene.fps)
    psi = np.pi + np.append(psi, psi[-1])
    return est[0], est[1], est[2], psi


# ---- estimator object ---------------------------------------------------------------------------------------
@dataclass
class CheetahEstimator:
    """Counterpart of acinoset_opt.CheetahEstimator (acinoset_opt.py:21-70): holds the problem tensors instead
    of a Pyomo model."""
    name: str
    data_path: str
    params: TrajectoryParams
    scene: Scene
    skeleton: abi.Skeleton
    cams: object
    meas: np.ndarray                 # [N, C, 24, 2]
    weight: np.ndarray               # [N, C, 24]
    scale_forces_by: float
    kinematic_model: bool
    tables: object = None
    device: int = 0
    opt_time_s: float = 0.0
    shutter_delay: Optional[np.ndarray] = None   # [C] seconds, set by estimate_kinematics when enable_shutter_delay_estimation
    synthesised_grf: Optional[dict] = None       # {foot: GRFz per frame}: the profile estimate_kinetics(joint_estimation=False) prescribed (acinoset_opt.py:823)
    costs: Dict[str, float] = field(default_factory=dict)
    result: Optional[dict] = None
    com_pos: Optional[np.ndarray] = None
    com_vel: Optional[np.ndarray] = None
    enable_eom_slack: bool = True
    bound_eom_error: Optional[Tuple[float, float]] = None
    kinetic: Optional[dict] = None          # node forces of the last estimate_kinetics (tau, lam, grf, slack, stance)

    def get_objective_cost(self) -> float:
        return float(self.result["stats"][0].cost) if self.result else float("nan")

    def relative_angles(self, q: np.ndarray) -> np.ndarray:
        """x (28) of acinoset_misc.py:508-528 + mask :1699-1757, linear in q"""
        sk = self.skeleton
        ind = skeleton.independent_dofs(sk)
        ref = np.array(sk.rel_ref[:sk.nq]); sgn = np.array(sk.rel_sign[:sk.nq])
        rel = np.where(ref < 0, q, sgn * (q - q[..., np.maximum(ref, 0)]))
        return rel[..., ind]

    def estimate_grf(self, plot: bool = False, monocular: bool = False, out_dir_prefix: Optional[str] = None,
                     contacts: Optional[dict] = None):
        """Per-frame ground-reaction-force fit, same signature and return value as CheetahEstimator.estimate_grf
        (acinoset_opt.py:176-270): reads the kinematic result `fte_kinematic[_<cam>]/fte.pickle` and the contact windows of
        `grf/autogen-contact.json` (written by the reference's determine_contacts; may be passed as `contacts` instead),
        fits the forces of the feet in contact on the GPU (cpe_grf_fit) and returns ({foot: [GRFz per frame]},
        {foot: [[4 friction components] per frame]}) in body weights."""
        params, scene, sk = self.params, self.scene, self.skeleton
        data_dir = params.data_dir if out_dir_prefix is None else os.path.join(out_dir_prefix, self.data_path)
        sub = "fte_kinematic" if not monocular else f"fte_kinematic_{scene.cam_idx}"
        fte = load_result_pickle(os.path.join(data_dir, sub, "fte.pickle"))       # restricted unpickler: arrays and numbers only
        if contacts is None:
            with open(os.path.join(data_dir, "grf", "autogen-contact.json"), "r", encoding="utf-8") as fh:
                contacts = json.load(fh)
        start_frame, end_frame = contacts["start_frame"], contacts["end_frame"]
        N = end_frame - start_frame
        feet = [f"{name}_foot" for name in skeleton.FEET]
        flags = np.zeros((N, len(feet)), np.int32)
        for i, f in enumerate(feet):
            for seq in (contacts["contacts"].get(f) or []):           # [first, last, foot index, "leading"|"trailing"] (acinoset_misc.py:812)
                a, b = int(seq[0]), int(seq[1])
                lo, hi = max(a - start_frame, 0), min(b - start_frame, N)
                if hi > lo:
                    flags[lo:hi, i] = 1
        gopt = skeleton.grf_options(self.name)                                # load_params falls back to the generic animal
        h = _lib.Handle(sk, self.cams, device=self.device)
        try:
            gz, gxy, res = h.grf_fit_host(gopt, fte["q"][None, :N], fte["dq"][None, :N], fte["ddq"][None, :N], flags[None])
        finally:
            h.close()
        self.grf_residual = res[0]
        grfz_est = {f: [float(v) for v in gz[0, :, i]] for i, f in enumerate(feet)}
        grfxy_est = {f: [[float(v) for v in row] for row in gxy[0, :, i]] for i, f in enumerate(feet)}
        return grfz_est, grfxy_est

    def folded_meas_err(self) -> np.ndarray:
        """slack_meas as the reference stores it, [N, C, 24, 2, W] (acinoset_opt.py:325): the solver's camera axis is [w * C + c] with PPM"""
        me = self.result["meas_err"][0]
        W = 3 if self.params.enable_ppms else 1
        C1 = me.shape[1] // W
        return np.ascontiguousarray(np.stack([me[:, w * C1:(w + 1) * C1] for w in range(W)], axis=-1))

    def robot_data(self) -> dict:
        """`cheetah.pickle` of a physics-based run: the layout `System3D.save_data_to_file` gives the reference's files (SURVEY 8b:
        name, description, nfe, ncp, hm, hm0, links: [{name, is_base, mass, length, radius, q, dq, ddq, Fr, nodes}]): per link the
        collocation variables, the joint constraint forces the link takes part in, and its nodes -- a motor (`Tc`) or a foot (`GRFz`,
        `GRFxy`, `foot_height`).  The library that defines the exact field shapes is absent (physical_education): unpinned."""
        sk, res, kin = self.skeleton, self.result, self.kinetic
        q, dq, ddq = res["q"][0], res["dq"][0], res["ddq"][0]
        N = q.shape[0]
        p = skeleton.load_params(self.name)
        groups = dict(skeleton.motor_groups())
        links = []
        for i, name in enumerate(skeleton.LINKS):
            lp = skeleton._link_param(p, name)
            sl = slice(0, 6) if i == 0 else slice(3 + 3 * i, 6 + 3 * i)
            rows = [r for r, (par, ch) in enumerate(skeleton.constraint_rows(sk)) if i in (par, ch)]
            nodes = []
            for mname, cols in groups.items():
                if mname.startswith(name + "_"):
                    nodes.append(dict(name=mname, Tc=kin["tau"][:, cols]))
            if name in skeleton.FEET:
                k = skeleton.FEET.index(name)
                mk = skeleton.MARKERS.index(skeleton.FOOT_MARKERS[k])
                nodes.append(dict(name=f"{name}_foot", GRFz=kin["grf"][:, k, 0], GRFxy=kin["grf"][:, k, 1:5],
                                  foot_height=res["positions"][0][:, mk, 2] - kin["ground_height"], stance=kin["stance"][:, k]))
            links.append(dict(name=name, is_base=i == 0, meta=[], mass=lp["mass"], length=lp["length"], radius=lp["radius"],
                              q=q[:, sl], dq=dq[:, sl], ddq=ddq[:, sl], Fr=kin["lam"][:, rows], nodes=nodes))
        h = 1.0 / self.scene.fps
        return dict(name=f"cheetah-{self.name}", description="Auto Save", repr="cheetah_pose_estimation_amd physics-based estimate",
                    nfe=N, ncp=1, hm=np.full(N, h), hm0=h, slack_eom=kin["slack"], links=links)

    def save(self, out_dir: str, fname: str = "fte", out_dir_prefix: Optional[str] = None):
        """fte.pickle + cam*_fte.csv, acinoset_opt.py:278-373."""
        res, params, scene = self.result, self.params, self.scene
        if out_dir_prefix:
            out_dir = os.path.join(out_dir_prefix, self.data_path, out_dir or "fte")
        else:
            out_dir = os.path.join(params.data_dir, out_dir or "fte")
        os.makedirs(out_dir, exist_ok=True)
        q, dq, ddq = res["q"][0], res["dq"][0], res["ddq"][0]
        tau = {}
        if self.kinetic is not None:                                # {motor name: [N, components]} (acinoset_opt.py:317-324)
            for name, cols in skeleton.motor_groups():
                tau[name] = np.ascontiguousarray(self.kinetic["tau"][:, cols])
        output = dict(positions=res["positions"][0], x=self.relative_angles(q), dx=self.relative_angles(dq),
                      ddx=self.relative_angles(ddq), q=q, dq=dq, ddq=ddq, com_pos=self.com_pos, com_vel=self.com_vel,
                      tau=tau, meas_err=self.folded_meas_err(), obj_cost=self.get_objective_cost(),
                      processing_time_s=self.opt_time_s, start_frame=params.start_frame)
        with open(os.path.join(out_dir, f"{fname}.pickle"), "wb") as f:
            pickle.dump(output, f)
        if not self.kinematic_model and self.kinetic is not None:
            with open(os.path.join(out_dir, "cheetah.pickle"), "wb") as f:      # robot.save_data_to_file (acinoset_opt.py:372-373)
                pickle.dump(self.robot_data(), f)
        off = [0] * scene.n_cams
        if params.sync_offset is not None:
            for o in params.sync_offset:
                off[o["cam"]] = o["frame"]
        all_cams = scene_cameras(Scene(scene.scene_fpath, scene.k_arr, scene.d_arr, scene.r_arr, scene.t_arr, scene.cam_res,
                                       scene.fps, scene.n_cams, None), params.kinetic_dataset)
        pos = res["positions"][0]
        hw = _lib.Handle(self.skeleton, all_cams, device=self.device)          # every camera of the scene, also for monocular runs
        try:
            delays = self.shutter_delay
            if delays is None:
                uv_all = hw.reproject_host(pos[None])[0]                       # [N, C, 24, 2] on the GPU (cpe_reproject)
            else:
                # positions_arr[c] = markers + q'_base tau_c + q''_base tau_c^2 (acinoset_opt.py:342-352), from node 2 on as the solve models it
                shift = np.zeros((scene.n_cams,) + pos.shape)
                for c in range(scene.n_cams):
                    d3 = dq[:, 0:3] * delays[c] + ddq[:, 0:3] * delays[c] ** 2
                    d3[:2] = 0.0
                    shift[c] = pos + d3[:, None, :]
                uv_c = hw.reproject_host(shift)                                # [C, N, C, 24, 2]: camera c reads its own displaced copy
                uv_all = np.stack([uv_c[c, :, c] for c in range(scene.n_cams)], axis=1)
        finally:
            hw.close()
        for i in range(scene.n_cams):
            uv = uv_all[:, i].copy()
            bad = (uv > np.array(scene.cam_res)) | (uv < 0)
            uv[bad.any(-1)] = np.nan                                # acinoset_misc.py:1385-1386
            n_frames = pos.shape[0]
            path = os.path.join(out_dir, f"cam{i + 1}_{fname}.csv")
            with open(path, "w") as f:                              # DLC MultiIndex layout (:1373-1399)
                f.write("bodyparts," + ",".join(f"{m},{m},{m}" for m in skeleton.MARKERS) + "\n")
                f.write("coords," + ",".join("x,y,likelihood" for _ in skeleton.MARKERS) + "\n")
                for n in range(n_frames):
                    row = ",".join((("" if np.isnan(uv[n, l, 0]) else repr(float(uv[n, l, 0]))) + "," +
                                    ("" if np.isnan(uv[n, l, 1]) else repr(float(uv[n, l, 1]))) + ",") for l in range(24))
                    f.write(f"{params.start_frame - off[i] + n},{row}\n")
        return out_dir


def init_trajectory(root_dir: str, data_path: str, cheetah_name: str, kinetic_dataset: bool, solver_path: Optional[str] = None,
                    start_frame: int = -1, end_frame: int = -1, dlc_thresh: float = 0.5,
                    include_camera_constraints: bool = True, enable_eom_slack: bool = True, kinematic_model: bool = False,
                    monocular_enable: bool = False, override_monocular_cam: Optional[int] = None,
                    disable_contact_lcp: bool = True, bound_eom_error: Optional[Tuple[float, float]] = None,
                    shutter_delay_estimation: bool = False, enable_ppm: bool = False, hand_labeled_data: bool = False,
                    device: int = 0) -> CheetahEstimator:
    """Same signature and meaning as acinoset_opt.init_trajectory (acinoset_opt.py:413-536).  `solver_path`
    (the IPOPT binary of the reference) is accepted and ignored."""
    if hand_labeled_data and enable_ppm:
        raise NotImplementedError("hand-labelled points together with pairwise predictions (there are none for hand labels)")
    if shutter_delay_estimation and enable_ppm and not monocular_enable:
        raise NotImplementedError("shutter delays together with pairwise predictions: the three measurement slices of a camera would share one delay")
    if cheetah_name not in ("jules", "phantom", "shiraz", "arabia"):
        cheetah_name = "acinoset"                                   # acinoset_opt.py:455-456
    model_name = f"{cheetah_name}-02" if kinetic_dataset else cheetah_name
    data_dir = os.path.join(root_dir, data_path)
    cam_idx, sync_offset = None, None
    if start_frame < 0 or end_frame < 0:
        with open(os.path.join(data_dir, "metadata.json"), "r", encoding="utf-8") as f:
            md = json.load(f)
        start_frame, end_frame, sync_offset = md["start_frame"], md["end_frame"], md["cam_sync"]
        cam_idx = md["monocular_cam"] if monocular_enable else None
        cam_idx = cam_idx if override_monocular_cam is None else override_monocular_cam
    total_length = end_frame - start_frame
    dlc_dir = os.path.join(data_dir, "dlc_hand_labeled" if hand_labeled_data else "dlc")          # acinoset_opt.py:479
    assert os.path.exists(dlc_dir), dlc_dir
    k_arr, d_arr, r_arr, t_arr, cam_res, n_cams, scene_fpath = find_scene_file(data_dir)
    fps = 200.0
    if not kinetic_dataset and "2019" in data_path:
        fps = 120.0
    elif not kinetic_dataset and "2017" in data_path:
        fps = 90.0                                                   # acinoset_opt.py:483-487
    d_arr = d_arr.reshape((n_cams, -1))
    params = TrajectoryParams(data_dir, start_frame, end_frame, total_length, dlc_thresh, sync_offset, hand_labeled_data,
                              kinetic_dataset, shutter_delay_estimation, enable_ppm)
    scene = Scene(scene_fpath, k_arr, d_arr, r_arr, t_arr, cam_res, fps, n_cams, cam_idx)
    paths = dlc_paths(dlc_dir)
    assert n_cams == len(paths), f"# of dlc files != # of cams in {scene_fpath}"
    sk = skeleton.build_skeleton(model_name, 24, kinetic_dataset)
    if hand_labeled_data:
        # acinoset_misc.py:217-222, :243-246, :471-474: rows by POSITION n + start_frame of the labelled table, no camera offsets, every labelled point at
        # weight 1 / R, squared loss (w r)^2.  The kernels evaluate rho(w r) with rho(e) = e^2 / 2 below the first knot: the knots go out of reach
        # (_kin_prepare) and the weights carry a factor sqrt(2).
        tables = [load_hand_labeled_table(p) for p in paths]
        pos = [(np.arange(len(t[0]), dtype=np.int64), t[1]) for t in tables]
        meas, weight = build_measurements(pos, start_frame, end_frame, None, n_cams, 0.5, kinetic_dataset, cam_idx, device=device)
        weight = weight * np.sqrt(2.0)
    else:
        tables = [load_dlc_table(p) for p in paths]
        meas, weight = build_measurements(tables, start_frame, end_frame, sync_offset, n_cams, dlc_thresh, kinetic_dataset, cam_idx, device=device)
    cams = scene_cameras(scene, kinetic_dataset, hand_labeled_data)
    if enable_ppm:
        # m.W = RangeSet(3) (acinoset_misc.py:179): the SAME projected marker is compared with three detections -- its own and two pairwise
        # predictions --, each with its own weight.  For the kernels that is every camera three times with identical parameters: cameras
        # [w * C + c], measurement slices [N, w * C + c, 24, ...]; CheetahEstimator.save folds the residuals back to [N, C, 24, 2, 3].
        pw_paths = sorted(glob(os.path.join(dlc_dir + "_pw", "*.npz"))) or sorted(glob(os.path.join(dlc_dir + "_pw", "*.pickle")))
        assert n_cams == len(pw_paths), f"# of pairwise files != # of cams in {scene_fpath}"
        pm, pw = build_pairwise_measurements([load_pairwise_table(p) for p in pw_paths], start_frame, end_frame, sync_offset, n_cams, dlc_thresh,
                                             kinetic_dataset, cam_idx)
        C1 = len(cams)
        meas = np.ascontiguousarray(np.concatenate([meas] + [pm[:, w] for w in range(2)], axis=1))
        weight = np.ascontiguousarray(np.concatenate([weight] + [pw[:, w] for w in range(2)], axis=1))
        cams3 = (abi.Camera * (3 * C1))()
        for w in range(3):
            for c in range(C1):
                cams3[w * C1 + c] = cams[c]
        cams = cams3
    total_mass = sum(sk.mass[i] for i in range(sk.n_links))
    return CheetahEstimator(cheetah_name, data_path, params, scene, sk, cams, meas, weight,
                            total_mass * 9.81, kinematic_model, tables, device, enable_eom_slack=enable_eom_slack, bound_eom_error=bound_eom_error)


def _kin_prepare(est: CheetahEstimator, monocular_constraints: bool, disable_pose_prior: bool, disable_motion_prior: bool,
                 pose_model_num_components: int, motion_model_window_size: int, motion_model_sparse_solution: bool,
                 q_init: Optional[np.ndarray], options: Optional[abi.Options]):
    """what estimate_kinematics sets up before the solver call (acinoset_opt.py:565-608): priors, initial guess, options"""
    params, scene, sk = est.params, est.scene, est.skeleton
    pri = None
    if monocular_constraints and scene.cam_idx is not None and not (disable_pose_prior and disable_motion_prior):
        # acinoset_opt.py:593-600.  The fitted numbers of the defaults ship as package data (tools/fit_priors.py re-runs the reference's recipe:
        # 5-component GMM, window-4 multi-task lasso); another size -- the grid search of run_dataset.py:814-915 -- is fitted here as the reference
        # does at run time (priors.fit_priors: up to 8 components, windows up to 4 frames) and cached.
        from . import priors as _priors
        default = (disable_pose_prior or pose_model_num_components == 5) and \
            (disable_motion_prior or (motion_model_window_size == 4 and motion_model_sparse_solution))
        if default:
            pri = _priors.load_priors(pose=not disable_pose_prior, motion=not disable_motion_prior)
        else:
            path = _priors.fit_priors(5 if disable_pose_prior else pose_model_num_components, 4 if disable_motion_prior else motion_model_window_size,
                                      True if disable_motion_prior else motion_model_sparse_solution)
            pri = _priors.load_priors(pose=not disable_pose_prior, motion=not disable_motion_prior, path=path)
    N = params.end_frame - params.start_frame
    if q_init is None:
        base_len = 2.0 * abs(sk.marker_off[5][0])
        x, y, z, psi = create_trajectory_estimate(est.tables, params, scene, base_len, device=est.device)
        q_init = np.zeros((N, sk.nq))                               # acinoset_opt.py:574-583
        sl = slice(params.start_frame, params.start_frame + N)
        q_init[:, 0], q_init[:, 1], q_init[:, 2] = x[sl], y[sl], z[sl]
        for i in range(sk.n_links):
            q_init[:, 3 + 3 * i + 2] = psi[sl]
    opts = options if options is not None else abi.default_options(scene.fps)
    opts.h = 1.0 / scene.fps
    if params.hand_labeled_data:
        opts.loss_a, opts.loss_b, opts.loss_c = 1e6, 2e6, 3e6          # squared loss: no residual reaches the first knot (init_trajectory)
    return q_init, opts, pri


def _kin_finish(est: CheetahEstimator, h, res: dict, seconds: float, solver_output: bool, monocular_constraints: bool,
                out_dir_prefix: Optional[str]) -> bool:
    """what estimate_kinematics does after the solver call (acinoset_opt.py:619-634): centre of mass, costs, files.  `res` holds ONE sequence."""
    params, scene = est.params, est.scene
    N = params.end_frame - params.start_frame
    est.opt_time_s = seconds
    import torch
    dev = torch.device("cuda", est.device)
    qd = torch.tensor(res["q"], device=dev)
    pos = torch.empty((1, N, 24, 3), dtype=torch.float64, device=dev); com = torch.empty((1, N, 3), dtype=torch.float64, device=dev)
    h.forward_kinematics(qd, pos, com); h.synchronize()
    est.com_pos = com[0].cpu().numpy()
    est.com_vel = (est.com_pos[1:] - est.com_pos[:-1]) * scene.fps     # acinoset_misc.py:742
    st = res["stats"][0]
    est.result = res
    est.costs = {"measurement": st.cost_meas, "model": st.cost_model, "pose": st.cost_pose, "motion": st.cost_motion}
    if solver_output:
        print(f"Total cost: {st.cost}\n-- measurement: {st.cost_meas}\n-- model: {st.cost_model}\n-- pose: {st.cost_pose}\n-- motion: {st.cost_motion}\n"
              f"status {st.status}, {st.iterations} LM iterations, {est.opt_time_s:.3f} s")
        if est.shutter_delay is not None:
            print("Shutter delay estimation:", [float(v) for v in est.shutter_delay])           # acinoset_opt.py:397-398
    ok = st.status == abi.OK
    if ok:
        fname = f"fte_kinematic{'_gt' if params.hand_labeled_data else ''}"
        fname = fname if scene.cam_idx is None or monocular_constraints else "fte_kinematic_orig"
        fname = fname if scene.cam_idx is None else f"{fname}_{scene.cam_idx}"    # acinoset_opt.py:626-628
        est.save(fname, out_dir_prefix=out_dir_prefix)
    return ok


def estimate_kinematics(estimator: CheetahEstimator, solver_output: bool = True, monocular_constraints: bool = False,
                        disable_pose_prior: bool = False, disable_motion_prior: bool = False,
                        pose_model_num_components: int = 5, motion_model_window_size: int = 4,
                        motion_model_sparse_solution: bool = True, out_dir_prefix: Optional[str] = None,
                        q_init: Optional[np.ndarray] = None, options: Optional[abi.Options] = None) -> bool:
    """Same signature as acinoset_opt.estimate_kinematics (acinoset_opt.py:539-547) plus two optional
    keyword arguments.  Returns True when the solve converged (IPOPT `ok`+`optimal` in the reference)."""
    est, params, scene, sk = estimator, estimator.params, estimator.scene, estimator.skeleton
    q_init, opts, pri = _kin_prepare(est, monocular_constraints, disable_pose_prior, disable_motion_prior, pose_model_num_components,
                                     motion_model_window_size, motion_model_sparse_solution, q_init, options)
    h = _lib.Handle(sk, est.cams, opts, pri, device=est.device)
    try:
        t0 = time()
        if params.enable_shutter_delay_estimation and scene.cam_idx is None:
            # m.shutter_delay, bounds +-hm0, camera 1 fixed (acinoset_misc.py:182-183, 273-275)
            res = h.solve_shutter_host(q_init[None], est.meas[None], est.weight[None], opts.h, max_rounds=20, tol_tau=1e-5)
            est.shutter_delay = res["tau"][0]
        else:
            res = h.solve_host(q_init[None], est.meas[None], est.weight[None])
            est.shutter_delay = None
        return _kin_finish(est, h, res, time() - t0, solver_output, monocular_constraints, out_dir_prefix)
    finally:
        h.close()


def _struct_bytes(x) -> bytes:
    import ctypes as _C
    return bytes(_C.string_at(_C.addressof(x), _C.sizeof(x)))


def estimate_kinematics_batch(estimators: Sequence[CheetahEstimator], solver_output: bool = False, monocular_constraints: bool = False,
                              disable_pose_prior: bool = False, disable_motion_prior: bool = False, out_dir_prefix: Optional[str] = None,
                              options: Optional[abi.Options] = None) -> List[bool]:
    """estimate_kinematics for MANY sequences at once: the loop of run_dataset.py:1145-1196 (`for seq in dataset: init_trajectory; estimate_kinematics`)
    as batched launches.  Sequences that share a skeleton, a camera rig, a length and a frame rate go through ONE solver handle and ONE cpe_solve
    call (B = the group's size: the GPU solves thousands of sequences per second batched, one at a time it is latency-bound at 7 - 30 ms each);
    every sequence then gets its own centre of mass, costs and files exactly as estimate_kinematics writes them.  All estimators must live on the
    same device; shutter-delay estimation is per sequence (cpe_solve_shutter) and goes through estimate_kinematics.  Returns one bool per
    estimator, in order."""
    ests = list(estimators)
    out: List[Optional[bool]] = [None] * len(ests)
    groups: Dict[tuple, List[int]] = {}
    prepared = {}
    for i, est in enumerate(ests):
        if est.params.enable_shutter_delay_estimation and est.scene.cam_idx is None:
            out[i] = estimate_kinematics(est, solver_output, monocular_constraints, disable_pose_prior, disable_motion_prior, out_dir_prefix=out_dir_prefix, options=options)
            continue
        q_init, opts, pri = _kin_prepare(est, monocular_constraints, disable_pose_prior, disable_motion_prior, 5, 4, True, None,
                                         None if options is None else _copy_options(options))
        prepared[i] = (q_init, opts, pri)
        key = (_struct_bytes(est.skeleton), b"".join(_struct_bytes(est.cams[c]) for c in range(len(est.cams))), q_init.shape[0], _struct_bytes(opts),
               None if pri is None else _struct_bytes(pri), est.device)
        groups.setdefault(key, []).append(i)
    for idx in groups.values():
        e0 = ests[idx[0]]
        _, opts, pri = prepared[idx[0]]
        h = _lib.Handle(e0.skeleton, e0.cams, opts, pri, device=e0.device)
        try:
            t0 = time()
            res = h.solve_host(np.stack([prepared[i][0] for i in idx]), np.stack([ests[i].meas for i in idx]), np.stack([ests[i].weight for i in idx]))
            dt = (time() - t0) / len(idx)                                            # processing_time_s of a sequence: its share of the batched solve
            for b, i in enumerate(idx):
                one = {k: (v[b:b + 1] if isinstance(v, np.ndarray) else v) for k, v in res.items() if k not in ("stats", "status")}
                one["stats"] = [res["stats"][b]]; one["status"] = res["stats"][b].status
                ests[i].shutter_delay = None
                out[i] = _kin_finish(ests[i], h, one, dt, solver_output, monocular_constraints, out_dir_prefix)
        finally:
            h.close()
    return [bool(v) for v in out]


def _copy_options(o: abi.Options) -> abi.Options:
    import ctypes as _C
    c = abi.Options()
    _C.memmove(_C.byref(c), _C.byref(o), _C.sizeof(abi.Options))
    return c


def _dataset_worker(rank: int, n_ranks: int, device: int, jobs: List[dict], estimate_kwargs: dict, queue) -> None:
    """one process of run_kinematics_dataset: builds the estimators of its shard on its GPU and solves them batched"""
    from . import sharding
    try:
        mine = [int(i) for i in sharding.shard_indices(len(jobs), rank, n_ranks)]
        ests = [init_trajectory(device=device, **jobs[i]) for i in mine]
        oks = estimate_kinematics_batch(ests, **estimate_kwargs)
        queue.put((rank, mine, oks, None))
    except Exception as exc:                                                          # the parent re-raises
        import traceback
        queue.put((rank, [], [], f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def run_kinematics_dataset(jobs: Sequence[dict], devices: Sequence[int] = (0,), **estimate_kwargs) -> List[bool]:
    """A whole data set of sequences across the GPUs of one node (BASELINE config 5; the reference loops over them one by one on the CPU,
    run_dataset.py:1143-1231).  jobs: one dictionary of init_trajectory keyword arguments per sequence (root_dir, data_path, cheetah_name,
    kinetic_dataset, kinematic_model=True, ...).  Sequence i goes to rank i mod G (sharding.shard_indices: independent sequences, no collective on
    the solve path); every rank is a FRESH process started before it touches its GPU (spawn, not fork), builds its estimators with init_trajectory
    (DLC tables -> measurement tensors on its GPU), solves them with estimate_kinematics_batch and writes each sequence's files.  Returns one bool
    per job, in order.  devices may name one GPU several times (rehearsal of the multi-rank path on a one-GPU box)."""
    jobs = [dict(j) for j in jobs]
    G = len(devices)
    if G == 1:
        ests = [init_trajectory(device=devices[0], **j) for j in jobs]
        return estimate_kinematics_batch(ests, **estimate_kwargs)
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_dataset_worker, args=(r, G, int(devices[r]), jobs, estimate_kwargs, queue)) for r in range(G)]
    for p_ in procs:
        p_.start()
    results, errors = [None] * len(jobs), []
    for _ in range(G):
        rank, mine, oks, err = queue.get()
        if err:
            errors.append(f"rank {rank}: {err}")
        for i, ok in zip(mine, oks):
            results[i] = bool(ok)
    for p_ in procs:
        p_.join()
    if errors:
        raise RuntimeError("run_kinematics_dataset: " + " | ".join(errors))
    return results


def determine_contacts(estimator: CheetahEstimator, monocular: bool = False, verbose: bool = True,
                       out_dir_prefix: Optional[str] = None):
    """Contact windows from the kinematic reconstruction + template forces, acinoset_opt.py:636-692: reads
    `fte_kinematic[_<cam>]/fte.pickle` (written by `CheetahEstimator.save`), evaluates foot heights and analytic foot
    velocities on the GPU (cpe_forward_kinematics, cpe_marker_velocities), applies the height / zero-velocity / stance-time
    heuristic (contacts.contact_detection) and writes `grf/autogen-contact.json`, `grf/autogen-contact-02.json` and the two
    synthetic force tables.  Returns (contacts, contacts_height_only) -- the reference prints them and returns None."""
    from . import contacts as ct
    params, scene, sk = estimator.params, estimator.scene, estimator.skeleton
    data_dir = params.data_dir if out_dir_prefix is None else os.path.join(out_dir_prefix, estimator.data_path)
    sub = "fte_kinematic" if not monocular else f"fte_kinematic_{scene.cam_idx}"
    fte = load_result_pickle(os.path.join(data_dir, sub, "fte.pickle"))           # restricted unpickler: arrays and numbers only
    estimator.com_vel, estimator.com_pos = fte["com_vel"], fte["com_pos"]
    h = _lib.Handle(sk, estimator.cams, device=estimator.device)
    try:
        pos, vel = h.kinematics_host(fte["q"][None], fte["dq"][None])
    finally:
        h.close()
    feet_idx = [skeleton.MARKERS.index(m) for m in skeleton.FOOT_MARKERS]
    names = [f"{f}_foot" for f in skeleton.FEET]
    speed = float(np.mean(np.linalg.norm(estimator.com_vel, axis=1)))
    contacts, by_height = ct.contact_detection(pos[0][:, feet_idx, 2], vel[0][:, feet_idx, 2], names, params.start_frame, speed, scene.fps)
    if verbose:
        print("Height, velocity, stance time heuristic:")
        print(contacts)
        print("Height:")
        print(by_height)
    grf_dir = os.path.join(data_dir, "grf")
    n_frames = pos.shape[1]
    ct.write_contacts(grf_dir, params.start_frame, n_frames, contacts, by_height)
    direction = 1.0 if float(np.mean(estimator.com_vel, axis=0)[0]) < 0 else -1.0
    for cname, oname in (("autogen-contact.json", "data_synth"), ("autogen-contact-02.json", "data_synth_02")):
        with open(os.path.join(grf_dir, cname), "r", encoding="utf-8") as fh:
            cj = json.load(fh)
        ct.write_synth_grf(os.path.join(grf_dir, f"{oname}.csv"), ct.synth_grf(cj, names, speed, direction))
    return contacts, by_height


def stance_from_contacts(contact_json: dict, n_frames: int) -> np.ndarray:
    """contact windows [first, last, foot, label] of `autogen-contact.json` / metadata.json -> stance[N, 4] in skeleton.FEET order: node
    first - start_frame ... last - start_frame INCLUSIVE with start_frame = the CONTACT FILE's own (`contact_times` of acinoset_opt.py:787-798:
    `(f_contact[0] - start_frame) + 1` in Pyomo's one-based finite elements)"""
    start = contact_json["start_frame"]
    st = np.zeros((n_frames, len(skeleton.FEET)), np.int32)
    for k, foot in enumerate(skeleton.FEET):
        for win in (contact_json["contacts"].get(f"{foot}_foot") or []):
            a, b = int(win[0]) - start, int(win[1]) - start
            st[max(a, 0):max(min(b + 1, n_frames), 0), k] = 1
    return st


def load_force_table(path_csv: str) -> Dict[int, np.ndarray]:
    """`grf/data_synth.csv` (or a `grf/data.csv` twin of the measured force plates): {force plate: array [rows, 3] = (Fx, Fy, Fz)} in file order --
    the table the reference reads with `pd.read_hdf(...).query("force_plate == k")` (acinoset_misc.py:960, :981)"""
    rows = np.genfromtxt(path_csv, delimiter=",", skip_header=1)
    rows = rows.reshape(-1, 5)
    return {int(k): rows[rows[:, 0] == k][:, 2:5] for k in np.unique(rows[:, 0])}


def resample_force_plate(x: np.ndarray, up: int = 2, down: int = 35, dc_samples: int = 500) -> np.ndarray:
    """One channel of the measured force plates (`grf/data.h5`, 3.5 kHz) on the 200 Hz frame grid, as `get_grf_profile(synthetic_data=False)` prepares
    it for the kinetic dataset (acinoset_misc.py:985-1000): the mean of the first `dc_samples` samples is removed (`remove_dc_offset(x, 500)`, :717-719),
    then `scipy.signal.resample_poly(x, up=2, down=35)`: zero-stuffing by 2, the default Kaiser-windowed (beta 5) low-pass FIR of half-length
    10 max(up, down) with cut-off 1 / max(up, down) and gain `up`, decimation by 35, output length ceil(2 len / 35)."""
    from scipy import signal
    x = np.asarray(x, dtype=np.float64)
    return signal.resample_poly(x - np.mean(x[:dc_samples], axis=0), up=up, down=down, axis=0)


def grf_profile(plates: Dict[int, np.ndarray], contact_json: dict, n_frames: int, absolute_rows: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """`misc.get_grf_profile` on per-frame tables (acinoset_misc.py:946-1026): grfz [N, 4] and grfxy [N, 4, 4] (friction-polygon sides +x, +y, -x,
    -y) in skeleton.FEET order.  As there: only the FIRST contact of a foot (it "assumes a single stride"), frames 0 .. N-2
    (`for fe in range(1, nfe)`), and of the horizontal force only the LARGEST positive polygon component is kept.  Row convention: a synthetic table
    (`data_synth`) starts at the contact file's start frame -- frame n reads row n (`Fz[fe - 1]`, :1003-1006); the resampled MEASURED plates of the
    kinetic dataset start with the recording -- frame n reads row start_frame + n (`Fz[start_frame + fe - 1]`, :1007-1010): `absolute_rows=True`."""
    start = contact_json["start_frame"]
    gz = np.zeros((n_frames, len(skeleton.FEET))); gxy = np.zeros((n_frames, len(skeleton.FEET), 4))
    for k, foot in enumerate(skeleton.FEET):
        rec = contact_json["contacts"].get(f"{foot}_foot")
        if not rec:
            continue
        first, last, plate = int(rec[0][0]), int(rec[0][1]), int(rec[0][2]) - 1
        F = plates.get(plate)
        if F is None:
            continue
        for n in range(n_frames - 1):
            row = start + n if absolute_rows else n
            if first <= start + n <= last and 0 <= row < len(F):
                fx, fy, fz = F[row]
                gz[n, k] = fz
                comps = np.array([fx, fy, -fx, -fy])
                i = int(np.argmax(comps))
                if comps[i] > 0:
                    gxy[n, k, i] = comps[i]
    return gz, gxy


def load_measured_plates(grf_dir: str, direction: float, scale_forces_by: float) -> Optional[Dict[int, np.ndarray]]:
    """CSV twins of the reference's `grf/data.h5` (PyTables is absent; columns force_plate, sample, Fx, Fy, Fz as `load_force_table` reads them):
    `data.csv` = one row per video frame from the start of the recording, already resampled and in body weights; `data_3500hz.csv` = the raw 3.5 kHz
    samples in newtons, resampled here exactly as the reference does (`measured_force_plates`).  None if neither exists."""
    per_frame = os.path.join(grf_dir, "data.csv")
    if os.path.exists(per_frame):
        return load_force_table(per_frame)
    raw = os.path.join(grf_dir, "data_3500hz.csv")
    if os.path.exists(raw):
        return measured_force_plates(load_force_table(raw), direction, scale_forces_by)
    return None


def measured_force_plates(raw: Dict[int, np.ndarray], direction: float, scale_forces_by: float) -> Dict[int, np.ndarray]:
    """raw {plate: [samples, 3] = (Fx, Fy, Fz) at 3.5 kHz, newtons} -> the per-frame table `grf_profile(..., absolute_rows=True)` reads: every channel
    resampled by 2 / 35 after its DC offset is removed, Fx and Fy times the running direction, all times `scale_forces_by` = 1 / (M g)
    (acinoset_misc.py:985-1000)"""
    out = {}
    for k, F in raw.items():
        F = np.asarray(F, dtype=np.float64)
        out[k] = np.stack([direction * resample_force_plate(F[:, 0]) * scale_forces_by, direction * resample_force_plate(F[:, 1]) * scale_forces_by,
                           resample_force_plate(F[:, 2]) * scale_forces_by], axis=1)
    return out


def estimate_kinetics(estimator: CheetahEstimator, init_torques: bool = True, auto: bool = True, use_2d_reprojections: bool = True,
                      solver_output: bool = True, init_prev_kinematic_solution: bool = True, synthesised_grf: bool = False,
                      no_slip: bool = True, joint_estimation: bool = False, fix_grf: bool = True, ground_constraint: bool = False,
                      disable_pose_prior: bool = False, disable_motion_prior: bool = False, plot: bool = False, out_fname: str = "fte",
                      out_dir_prefix: Optional[str] = None, options: Optional[abi.Options] = None,
                      kinetic_options: Optional[abi.KineticOptions] = None) -> bool:
    """Same signature and meaning as acinoset_opt.estimate_kinetics (acinoset_opt.py:693-708) for the branch its drivers run for the
    physics-based reconstruction (`joint_estimation=True`, run_dataset.py:1198-1229): torques, joint constraint forces, ground-reaction
    forces and the trajectory are estimated together, warm-started from the kinematic solution on disk, with the contact windows of
    `grf/autogen-contact.json` (auto) or `metadata.json`; and for the branch of the kinetic-dataset driver (`joint_estimation=False`,
    `fix_grf=True`, run_dataset.py:1092-1140; acinoset_opt.py:813-866): the ground-reaction forces are PRESCRIBED -- the synthesised profile of
    `grf/data_synth.csv` (`synthesised_grf=True`) or the per-frame fit of `CheetahEstimator.estimate_grf` -- the feet are held near the ground
    (`ground_constraint`) and still (`no_slip`) while their force is positive, and torques, constraint forces and the trajectory are estimated
    (cpe_solve_kinetic_fixed).  With `fix_grf=False` the same profile only BOXES the forces (acinoset_opt.py:838-850: `GRFz` and every `GRFxy` side
    within `bound_value(profile, 0.2)`; cpe_solve_kinetic_force_box): net z in [0.8, 1.2] x profile, net x / y from the boxes of the two opposite
    polygon sides (a side is non-negative), friction polyhedron kept; feet outside the profile's contact carry no force (the reference would let
    them take up to 0.2 body weights: its box around zero).  The whole NLP runs on the GPU.  `init_torques` has no effect here: the torques are
    minimised out exactly at every evaluation, so they need no starting value."""
    est, params, scene, sk = estimator, estimator.params, estimator.scene, estimator.skeleton
    if est.kinematic_model:
        raise AssertionError("Dynamic model of the cheetah is required.")          # the reference asserts hasattr(model, 'eom_f')
    if not use_2d_reprojections:
        raise NotImplementedError("the 3D kinematic cost (use_2d_reprojections=False) is not built")
    if not est.enable_eom_slack:
        raise NotImplementedError("enable_eom_slack=False (hard equations of motion, no slack cost, acinoset_opt.py:914): the slack IS the residual of this "
                                  "solver's least-squares model; no driver of the reference switches it off")
    if params.enable_shutter_delay_estimation and scene.cam_idx is None:
        raise NotImplementedError("shutter delays are estimated by estimate_kinematics only (cpe_solve_shutter); not inside the physics-based model")
    data_dir = params.data_dir if out_dir_prefix is None else os.path.join(out_dir_prefix, est.data_path)
    mono = scene.cam_idx is not None and init_prev_kinematic_solution
    fte = load_result_pickle(os.path.join(data_dir, f"fte_kinematic_{scene.cam_idx}" if mono else "fte_kinematic", "fte.pickle"))
    est.com_vel, est.com_pos = fte["com_vel"], fte["com_pos"]
    N = params.end_frame - params.start_frame
    if init_prev_kinematic_solution:
        q_init = np.ascontiguousarray(fte["q"][:N], dtype=np.float64)                # acinoset_opt.py:768-777
    else:
        base_len = 2.0 * abs(sk.marker_off[5][0])
        x, y, z, psi = create_trajectory_estimate(est.tables, params, scene, base_len, device=est.device)
        q_init = np.zeros((N, sk.nq)); sl = slice(params.start_frame, params.start_frame + N)
        q_init[:, 0], q_init[:, 1], q_init[:, 2] = x[sl], y[sl], z[sl]
        for i in range(sk.n_links):
            q_init[:, 3 + 3 * i + 2] = psi[sl]
    with open(os.path.join(data_dir, "grf", "autogen-contact.json") if auto else os.path.join(params.data_dir, "metadata.json"), "r", encoding="utf-8") as fh:
        contact_json = json.load(fh)
    grf_fixed = None
    grf_box = None
    if joint_estimation:
        stance = stance_from_contacts(contact_json, N)
    else:
        if synthesised_grf:                                                          # acinoset_opt.py:814-820
            measured = (not auto) and params.kinetic_dataset                         # the resampled plates are indexed by absolute frame (acinoset_misc.py:1007-1010)
            if auto:
                table = os.path.join(data_dir, "grf", "data_synth.csv")
                plates = load_force_table(table) if os.path.exists(table) else None
            else:
                table = os.path.join(params.data_dir, "grf", "data.csv")
                direction = 1.0 if float(np.mean(est.com_vel, axis=0)[0]) < 0 else -1.0      # acinoset_opt.py:817
                plates = load_measured_plates(os.path.join(params.data_dir, "grf"), direction, 1.0 / est.scale_forces_by)
            if plates is None:
                raise FileNotFoundError(f"{table}: the force table of the prescribed-force branch (determine_contacts writes grf/data_synth.csv; the measured "
                                        "force plates ship as grf/data.h5 in the reference, which needs PyTables -- provide a CSV twin, data.csv or data_3500hz.csv)")
            gz, gxy = grf_profile(plates, contact_json, N, absolute_rows=measured)
        else:                                                                        # :821-822: the per-frame fit
            gzd, gxyd = est.estimate_grf(monocular=True, plot=False, out_dir_prefix=out_dir_prefix)
            gz = np.zeros((N, 4)); gxy = np.zeros((N, 4, 4))
            for k, foot in enumerate(skeleton.FEET):
                v = np.asarray(gzd[f"{foot}_foot"], dtype=np.float64); w = np.asarray(gxyd[f"{foot}_foot"], dtype=np.float64).reshape(-1, 4)
                gz[:min(N, len(v)), k] = v[:N]; gxy[:min(N, len(w)), k] = w[:N]
        est.synthesised_grf = {f"{foot}_foot": [float(v) for v in gz[:, k]] for k, foot in enumerate(skeleton.FEET)}
        stance = (gz > 0).astype(np.int32)                                           # height / no-slip rules where a force acts (:832-835, :853-864)
        grf_fixed = np.ascontiguousarray(np.stack([gz, gxy[..., 0] - gxy[..., 2], gxy[..., 1] - gxy[..., 3]], axis=-1))
        if not fix_grf:
            bz = bound_value(gz, 0.2)                                                  # [N, 4, 2]
            bs = np.maximum(bound_value(gxy, 0.2), 0.0)                                # [N, 4, 4 sides, 2]; a polygon side is >= 0
            grf_box = np.zeros((N, 4, 3, 2))
            grf_box[:, :, 0] = bz
            grf_box[:, :, 1, 0] = bs[:, :, 0, 0] - bs[:, :, 2, 1]; grf_box[:, :, 1, 1] = bs[:, :, 0, 1] - bs[:, :, 2, 0]      # x = (+x side) - (-x side)
            grf_box[:, :, 2, 0] = bs[:, :, 1, 0] - bs[:, :, 3, 1]; grf_box[:, :, 2, 1] = bs[:, :, 1, 1] - bs[:, :, 3, 0]
            grf_fixed = None
    pri = None
    if not disable_pose_prior and scene.cam_idx is not None:                         # acinoset_opt.py:916-917
        from . import priors as _priors
        pri = _priors.load_priors(pose=True, motion=False)
    opts = options if options is not None else abi.default_options(scene.fps)
    if params.hand_labeled_data:
        opts.loss_a, opts.loss_b, opts.loss_c = 1e6, 2e6, 3e6          # squared loss of hand-labelled points (init_trajectory)
    opts.h = 1.0 / scene.fps
    if options is None:
        opts.tol_cost, opts.max_iter = 1e-6, 1500          # the physics term is stiff: see DESIGN.md 2b (the reference stops IPOPT at Tol = 1e-3; a monocular start with a missed contact window has taken 1 200 iterations)
    ko = kinetic_options if kinetic_options is not None else abi.default_kinetic_options(skeleton.dyn_options(est.name), scene.fps, params.kinetic_dataset)
    if not no_slip and not joint_estimation:
        ko.slip_max = 0.0                                                            # `no_slip` guards the rules of the prescribed-force branch only (acinoset_opt.py:855-866);
        ko.zvel_max = 0.0                                                            # the joint-estimation branch always has them (:803-810)
    if not joint_estimation and not ground_constraint:
        ko.foot_height_tol = 1e9                                                     # the feet are not tied to the ground (:832)
    if disable_motion_prior:
        ko.w_torque, ko.w_smooth = 0.0, 0.0                                          # acinoset_opt.py:918-920
    if est.bound_eom_error is not None:
        ko.slack_lo, ko.slack_hi = float(est.bound_eom_error[0]), float(est.bound_eom_error[1])      # make_pyomo_model(bound_eom_error=...), acinoset_opt.py:510-514
    skk = skeleton.without_motion_model(sk)              # the physics-based cost has no constant-acceleration term (acinoset_opt.py:905-921)
    h = _lib.Handle(skk, est.cams, opts, pri, device=est.device)
    try:
        t0 = time()
        res = h.solve_kinetic_host(ko, q_init[None], est.meas[None], est.weight[None], stance[None],
                                   grf_fixed=None if grf_fixed is None else grf_fixed[None], grf_box=None if grf_box is None else grf_box[None])
        est.opt_time_s = time() - t0
        import torch
        dev = torch.device("cuda", est.device)
        qd = torch.tensor(res["q"], device=dev)
        pos = torch.empty((1, N, 24, 3), dtype=torch.float64, device=dev); com = torch.empty((1, N, 3), dtype=torch.float64, device=dev)
        h.forward_kinematics(qd, pos, com); h.synchronize()
        est.com_pos = com[0].cpu().numpy()
        est.com_vel = (est.com_pos[1:] - est.com_pos[:-1]) * scene.fps
    finally:
        h.close()
    st, ks = res["stats"][0], res["kstats"][0]
    est.result = res
    est.kinetic = dict(tau=res["tau"][0], lam=res["lam"][0], grf=res["grf"][0], slack=res["slack"][0], stance=stance, ground_height=ko.ground_height)
    est.costs = {"measurement": st.cost_meas, "pose": st.cost_pose, "energy": ks.cost_energy, "eom_error": ks.cost_eom, "torque": ks.cost_torque}
    base_err = float(np.sqrt(np.mean((q_init[:, :6] - res["q"][0][:, :6]) ** 2)))
    rel_err = float(np.sqrt(np.mean((q_init[:, 6:] - res["q"][0][:, 6:]) ** 2)))
    if solver_output:
        print(f"Total cost: {st.cost}\n-- measurement: {st.cost_meas}\n-- pose: {st.cost_pose}\n-- energy: {ks.cost_energy}\n-- eom_error: {ks.cost_eom}\n"
              f"-- torque: {ks.cost_torque}\nstatus {st.status}, {st.iterations} LM iterations, {st.outer} multiplier updates, {est.opt_time_s:.3f} s\n"
              f"max |slack_eom| {ks.max_slack:.3e} (box [{ko.slack_lo}, {ko.slack_hi}]), max |rows 0-2| / Mg {ks.max_base_rows:.3e}, max violated inequality {ks.max_violation:.3e}\n"
              f"RMSE base: {base_err:.4f}\nRMSE links: {rel_err:.4f}")
    ok = st.status == abi.OK and bool((res["slack"][0] >= ko.slack_lo - 1e-4).all() and (res["slack"][0] <= ko.slack_hi + 1e-4).all())
    if scene.cam_idx is not None or ok:                                              # acinoset_opt.py:948-954
        dname = f"fte_kinetic{'_gt' if params.hand_labeled_data else ''}"
        dname = dname if scene.cam_idx is None else f"{dname}_{scene.cam_idx}"
        est.save(dname, fname=out_fname, out_dir_prefix=out_dir_prefix)
    return ok


def bound_value(val, slack_percentage: float) -> np.ndarray:
    """`misc.bound_value` (acinoset_misc.py:84-90), vectorised: [..., 2] = (lower, upper) -- within `slack_percentage` of the value on its own
    side of zero, (-slack, +slack) for a value of exactly zero"""
    v = np.asarray(val, dtype=np.float64)
    lo = np.where(v > 0, (1 - slack_percentage) * v, np.where(v < 0, (1 + slack_percentage) * v, -slack_percentage))
    hi = np.where(v > 0, (1 + slack_percentage) * v, np.where(v < 0, (1 - slack_percentage) * v, slack_percentage))
    return np.stack([lo, hi], axis=-1)


def estimate_grf(estimator: CheetahEstimator, solver_output: bool = True, out_dir_prefix: Optional[str] = None,
                 options: Optional[abi.Options] = None, kinetic_options: Optional[abi.KineticOptions] = None) -> bool:
    """Same signature and meaning as the module-level acinoset_opt.estimate_grf (acinoset_opt.py:966-1048), the last stage of the kinetic-dataset
    pipeline (run_dataset.py:1125-1138): the physics-based model is solved again from the stored `fte_kinetic/fte.pickle` -- trajectory as the
    starting point, every torque within 10 % of its stored value (`Tc.bounds = bound_value(init_tau, 0.1)`, :995-1003) -- with the ground-reaction
    forces FREE where the measured force plates saw the foot on the ground and zero elsewhere (:1005-1017), the feet within 3 cm of the ground
    during those contacts, no pose prior, cost = measurement + torque + 0.1 fps^-2 smoothing + 10e3 slack (:1019-1026).  Runs on the GPU
    (cpe_solve_kinetic_bounded); writes `fte_grf/`.
    Contact pattern: the reference takes the non-zero entries of `get_grf_profile(..., synthetic_data=False)`, i.e. the frames of each foot's FIRST
    window in `metadata.json` (frames 0 .. N-2) at which the resampled plate signal is non-zero; with a per-frame CSV twin `grf/data.csv` of that
    table (the reference ships `grf/data.h5` at 3.5 kHz, which needs PyTables and a resampling step that are not reproduced here) the same rule is
    applied to it, otherwise the window alone decides (a loaded plate never reads exactly zero)."""
    est, params, scene, sk = estimator, estimator.params, estimator.scene, estimator.skeleton
    assert params.kinetic_dataset, "Cannot determine GRF on a dataset other than the kinetic dataset from Penny Hudson and Co."
    if est.kinematic_model:
        raise AssertionError("Dynamic model of the cheetah is required.")
    data_dir = params.data_dir if out_dir_prefix is None else os.path.join(out_dir_prefix, est.data_path)
    fte = load_result_pickle(os.path.join(data_dir, "fte_kinetic", "fte.pickle"))
    N = params.end_frame - params.start_frame
    q_init = np.ascontiguousarray(fte["q"][:N], dtype=np.float64)
    groups = skeleton.motor_groups()
    nm = sum(len(cols) for _, cols in groups)
    init_tau = np.zeros((N, nm))
    for name, cols in groups:                                                       # {motor name: [N, components]} as save() and the reference write it
        init_tau[:, cols] = np.asarray(fte["tau"][name], dtype=np.float64)[:N]
    tau_box = bound_value(init_tau, 0.1)
    with open(os.path.join(params.data_dir, "metadata.json"), "r", encoding="utf-8") as fh:
        contact_json = json.load(fh)
    direction = 1.0 if float(np.mean(np.asarray(fte["com_vel"]), axis=0)[0]) < 0 else -1.0
    plates = load_measured_plates(os.path.join(params.data_dir, "grf"), direction, 1.0 / est.scale_forces_by)
    if plates is not None:
        gz, _ = grf_profile(plates, contact_json, N, absolute_rows=True)                # measured plates: row = absolute frame
        stance = (gz != 0).astype(np.int32)
    else:
        stance = np.zeros((N, len(skeleton.FEET)), np.int32)
        for k, foot in enumerate(skeleton.FEET):
            rec = contact_json["contacts"].get(f"{foot}_foot")
            if rec:
                a, b = int(rec[0][0]) - contact_json["start_frame"], int(rec[0][1]) - contact_json["start_frame"]
                stance[max(a, 0):max(min(b + 1, N - 1), 0), k] = 1                   # first window only, frames 0 .. N-2 (acinoset_misc.py:954, :1003)
    opts = options if options is not None else abi.default_options(scene.fps)
    if params.hand_labeled_data:
        opts.loss_a, opts.loss_b, opts.loss_c = 1e6, 2e6, 3e6          # squared loss of hand-labelled points (init_trajectory)
    opts.h = 1.0 / scene.fps
    if options is None:
        opts.tol_cost, opts.max_iter = 1e-6, 1500
    ko = kinetic_options if kinetic_options is not None else abi.default_kinetic_options(skeleton.dyn_options(est.name), scene.fps, True)
    if kinetic_options is None:
        ko.foot_height_tol = 0.03                                                    # foot_height in [-0.03, 0.03] during a contact (:1010-1012)
        ko.slip_max = 0.0                                                            # this NLP has no no-slip rule (estimate_kinetics adds it to ITS model)
        ko.zvel_max = 0.0                                                            # ... and no `foot_z_vel <= 1` rule either (acinoset_opt.py:1004-1017)
    if est.bound_eom_error is not None:
        ko.slack_lo, ko.slack_hi = float(est.bound_eom_error[0]), float(est.bound_eom_error[1])      # make_pyomo_model(bound_eom_error=...), acinoset_opt.py:510-514
    skk = skeleton.without_motion_model(sk)
    h = _lib.Handle(skk, est.cams, opts, None, device=est.device)
    try:
        t0 = time()
        res = h.solve_kinetic_host(ko, q_init[None], est.meas[None], est.weight[None], stance[None], tau_box=tau_box[None])
        est.opt_time_s = time() - t0
        import torch
        dev = torch.device("cuda", est.device)
        qd = torch.tensor(res["q"], device=dev)
        pos = torch.empty((1, N, 24, 3), dtype=torch.float64, device=dev); com = torch.empty((1, N, 3), dtype=torch.float64, device=dev)
        h.forward_kinematics(qd, pos, com); h.synchronize()
        est.com_pos = com[0].cpu().numpy()
        est.com_vel = (est.com_pos[1:] - est.com_pos[:-1]) * scene.fps
    finally:
        h.close()
    st, ks = res["stats"][0], res["kstats"][0]
    est.result = res
    est.kinetic = dict(tau=res["tau"][0], lam=res["lam"][0], grf=res["grf"][0], slack=res["slack"][0], stance=stance, ground_height=ko.ground_height,
                       tau_box=tau_box)
    est.costs = {"measurement": st.cost_meas, "energy": ks.cost_energy, "eom_error": ks.cost_eom, "torque": ks.cost_torque}
    base_err = float(np.sqrt(np.mean((q_init[:, :6] - res["q"][0][:, :6]) ** 2)))
    rel_err = float(np.sqrt(np.mean((q_init[:, 6:] - res["q"][0][:, 6:]) ** 2)))
    if solver_output:
        print(f"Total cost: {st.cost}\n-- measurement: {st.cost_meas}\n-- energy: {ks.cost_energy}\n-- eom_error: {ks.cost_eom}\n-- torque: {ks.cost_torque}\n"
              f"status {st.status}, {st.iterations} LM iterations, {st.outer} multiplier updates, {est.opt_time_s:.3f} s\n"
              f"max |slack_eom| {ks.max_slack:.3e} (box [{ko.slack_lo}, {ko.slack_hi}]), max violated inequality {ks.max_violation:.3e}\n"
              f"RMSE base: {base_err:.4f}\nRMSE links: {rel_err:.4f}")
    ok = st.status == abi.OK and bool((res["slack"][0] >= ko.slack_lo - 1e-4).all() and (res["slack"][0] <= ko.slack_hi + 1e-4).all())
    if ok:
        est.save("fte_grf", fname="fte", out_dir_prefix=out_dir_prefix)              # acinoset_opt.py:1045-1046
    return ok
