"""Loader + thin ctypes binding of libcpe.so (the HIP / C-ABI product library, include/cpe.h).

There is NO CPU fallback: if the library is missing, or no MI355X is visible, every call fails loudly.
Torch is used only as plumbing for device memory (tensor.data_ptr()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcpe.so")
_CSRC = os.path.join(_HERE, "csrc")
_LIB = None
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


class CpeError(RuntimeError):
    pass


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libcpe.so (in-tree).  hipcc cross-compiles without a GPU."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)] + [os.path.join(_HERE, "..", "include", "cpe.h")]
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH,
           os.path.join(_CSRC, "cpe_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise CpeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the solve path)")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; it must be resident before
    # libcpe.so resolves the same soname, otherwise two runtimes fight over the device.
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    lib.cpe_last_error.restype = C.c_char_p
    lib.cpe_stream.restype = C.c_void_p
    lib.cpe_stream.argtypes = [C.c_void_p]
    lib.cpe_create.argtypes = [C.POINTER(abi.Skeleton), C.POINTER(abi.Camera), C.c_int32, C.POINTER(abi.Options),
                               C.POINTER(abi.Priors), C.c_int32, C.POINTER(C.c_void_p)]
    lib.cpe_destroy.argtypes = [C.c_void_p]
    lib.cpe_synchronize.argtypes = [C.c_void_p]
    lib.cpe_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpe_stream_signal.argtypes = [C.c_void_p, C.c_void_p]
    lib.cpe_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.cpe_profile_get.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
    lib.cpe_jacobian_slots.argtypes = [C.c_void_p]
    lib.cpe_jacobian_layout.argtypes = [C.c_void_p, ip, ip]
    lib.cpe_num_independent.argtypes = [C.c_void_p]
    lib.cpe_independent_dofs.argtypes = [C.c_void_p, ip]
    vp = C.c_void_p
    lib.cpe_eval_resjac.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.cpe_eval_resjac_host.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.cpe_project_joints.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.cpe_forward_kinematics.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    lib.cpe_marker_velocities.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    lib.cpe_reproject.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    lib.cpe_triangulate.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.c_double, vp]
    lib.cpe_tensorise_dlc.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_double, vp, vp]
    lib.cpe_eval_normal.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cpe_solve.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(abi.Stats)]
    lib.cpe_solve_shutter.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_double, C.c_int32, C.c_double, vp, vp, vp, vp, vp, vp,
                                      C.POINTER(abi.Stats), C.POINTER(C.c_int32)]
    lib.cpe_solve_host.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(abi.Stats)]
    lib.cpe_eom_rows.argtypes = [vp, C.POINTER(abi.EomOptions), C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.cpe_eom_residual.argtypes = [vp, C.POINTER(abi.DynOptions), C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.cpe_grf_fit.argtypes = [vp, C.POINTER(abi.GrfOptions), C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.cpe_default_kinetic_options.argtypes = [C.POINTER(abi.KineticOptions), C.c_double, C.c_int32]
    lib.cpe_default_kinetic_options.restype = None
    lib.cpe_solve_kinetic.argtypes = [vp, C.POINTER(abi.KineticOptions), C.c_int32, C.c_int32] + [vp] * 13 + [C.POINTER(abi.Stats), C.POINTER(abi.KineticStats)]
    lib.cpe_solve_kinetic_fixed.argtypes = [vp, C.POINTER(abi.KineticOptions), C.c_int32, C.c_int32] + [vp] * 14 + [C.POINTER(abi.Stats), C.POINTER(abi.KineticStats)]
    lib.cpe_solve_kinetic_force_box.argtypes = [vp, C.POINTER(abi.KineticOptions), C.c_int32, C.c_int32] + [vp] * 14 + [C.POINTER(abi.Stats), C.POINTER(abi.KineticStats)]
    lib.cpe_solve_kinetic_bounded.argtypes = [vp, C.POINTER(abi.KineticOptions), C.c_int32, C.c_int32] + [vp] * 14 + [C.POINTER(abi.Stats), C.POINTER(abi.KineticStats)]
    lib.cpe_eval_kinetic_nodes.argtypes = [vp, C.POINTER(abi.KineticOptions), C.c_int32, C.c_int32] + [vp] * 11
    _LIB = lib
    return lib


def _check(status: int, what: str, allow=(abi.OK,)):
    if status not in allow:
        raise CpeError(f"{what} failed with status {status}: {load().cpe_last_error().decode()}")
    return status


def _ptr(t):
    """device pointer of a torch tensor / host pointer of a numpy array / None"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.dtype == np.float64 and t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    assert t.is_contiguous() and ((t.dtype.is_floating_point and t.element_size() == 8) or str(t.dtype) == "torch.int32")
    return t.data_ptr()


class Handle:
    """One solver instance = one skeleton + one camera rig + options, bound to one GPU and one HIP stream.
    Stands where the reference builds its Pyomo model (acinoset_opt.py:459-525)."""

    def __init__(self, sk: abi.Skeleton, cams, opts: abi.Options = None, priors: abi.Priors = None, device: int = 0):
        self.lib = load()
        self.sk, self.cams, self.n_cams = sk, cams, len(cams)
        self.opts = opts if opts is not None else abi.default_options()
        self._h = C.c_void_p()
        st = self.lib.cpe_create(C.byref(sk), cams, self.n_cams, C.byref(self.opts),
                                 C.byref(priors) if priors is not None else None, device, C.byref(self._h))
        if st == abi.NO_DEVICE:
            raise CpeError("no HIP device visible: the solve path has no CPU fallback (" + self.lib.cpe_last_error().decode() + ")")
        _check(st, "cpe_create")
        self.device = device
        self.S = self.lib.cpe_jacobian_slots(self._h)
        self.nu = self.lib.cpe_num_independent(self._h)
        self.nq, self.L = sk.nq, sk.n_markers

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.cpe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return self.lib.cpe_stream(self._h)

    def synchronize(self):
        _check(self.lib.cpe_synchronize(self._h), "cpe_synchronize")

    # The device-pointer entry points enqueue on the handle's own stream (include/cpe.h, "Synchronisation contract").  Torch
    # tensors are produced and consumed on torch's current stream, so every device-pointer method below brackets its call:
    # the handle first waits for what torch has queued, and torch's stream afterwards waits for what the handle has queued.
    def _torch_stream(self):
        import torch
        return torch.cuda.current_stream(self.device).cuda_stream

    def _enter(self):
        _check(self.lib.cpe_stream_wait(self._h, self._torch_stream()), "cpe_stream_wait")

    def _leave(self):
        _check(self.lib.cpe_stream_signal(self._h, self._torch_stream()), "cpe_stream_signal")

    PROFILE_SLOTS = ("k_frame_normal", "k_lr_band", "k_lm_step", "k_build_act", "k_finalize", "k_dyn_eval", "k_dyn_gather", "k_lm_back",
                     "k_dyn_assemble", "k_dyn_schur", "k_dyn_jac", "_free11")

    def profile(self, on: bool):
        _check(self.lib.cpe_profile_enable(self._h, 1 if on else 0), "cpe_profile_enable")

    def profile_totals(self):
        """{kernel: (milliseconds, launches)} accumulated by the solves since profile(True)"""
        ms = (C.c_double * len(self.PROFILE_SLOTS))(); n = (C.c_int64 * len(self.PROFILE_SLOTS))()
        _check(self.lib.cpe_profile_get(self._h, ms, n), "cpe_profile_get")
        return {k: (ms[i], int(n[i])) for i, k in enumerate(self.PROFILE_SLOTS) if n[i]}

    def jacobian_layout(self):
        sm = np.empty(self.S, dtype=np.int32); sd = np.empty(self.S, dtype=np.int32)
        _check(self.lib.cpe_jacobian_layout(self._h, sm.ctypes.data_as(ip), sd.ctypes.data_as(ip)), "cpe_jacobian_layout")
        return sm, sd

    def independent_dofs(self):
        d = np.empty(self.nu, dtype=np.int32)
        _check(self.lib.cpe_independent_dofs(self._h, d.ctypes.data_as(ip)), "cpe_independent_dofs")
        return d

    # ---- device-pointer entry points (torch-ROCm tensors on self.device) -------------------------------
    def eval_resjac(self, q, meas, weight, r, J, eps, cost=None):
        B, N = q.shape[0], q.shape[1]
        self._enter()
        _check(self.lib.cpe_eval_resjac(self._h, B, N, _ptr(q), _ptr(meas), _ptr(weight), _ptr(r), _ptr(J), _ptr(eps), _ptr(cost)),
               "cpe_eval_resjac")
        self._leave()

    def project_joints(self, q):
        self._enter()
        st = _check(self.lib.cpe_project_joints(self._h, q.shape[0], q.shape[1], _ptr(q)), "cpe_project_joints",
                    allow=(abi.OK, abi.NUMERICAL))
        self._leave()
        return st

    def forward_kinematics(self, q, positions, com=None):
        self._enter()
        _check(self.lib.cpe_forward_kinematics(self._h, q.shape[0], q.shape[1], _ptr(q), _ptr(positions), _ptr(com)),
               "cpe_forward_kinematics")
        self._leave()

    def marker_velocities(self, q, dq, velocities):
        """v_l = (d p_l / d q) dq for every marker (device tensors; velocities [B, N, L, 3])"""
        self._enter()
        _check(self.lib.cpe_marker_velocities(self._h, q.shape[0], q.shape[1], _ptr(q), _ptr(dq), _ptr(velocities)),
               "cpe_marker_velocities")
        self._leave()

    def reproject(self, positions, uv):
        """stored 3D markers -> pixels in every camera (device tensors; uv [B, N, C, L, 2])"""
        self._enter()
        _check(self.lib.cpe_reproject(self._h, positions.shape[0], positions.shape[1], _ptr(positions), _ptr(uv)), "cpe_reproject")
        self._leave()

    def reproject_host(self, positions):
        """numpy [B, N, L, 3] -> numpy [B, N, C, L, 2] (staged through HBM with torch)"""
        import torch
        dev = torch.device("cuda", self.device)
        pd = torch.tensor(np.ascontiguousarray(positions, dtype=np.float64), device=dev)
        uv = torch.empty((pd.shape[0], pd.shape[1], self.n_cams, self.L, 2), dtype=torch.float64, device=dev)
        self.reproject(pd, uv)
        self.synchronize()
        return uv.cpu().numpy()

    def triangulate_host(self, cam_a, cam_b, uv_a, uv_b, depth=3.0):
        """n detection pairs -> xyz [n, 3] (cpe_triangulate; cam_b < 0: back-projection of uv_a to `depth`); numpy in / out"""
        import torch
        dev = torch.device("cuda", self.device)
        ca = torch.tensor(np.ascontiguousarray(cam_a, dtype=np.int32), device=dev)
        cb = torch.tensor(np.ascontiguousarray(cam_b, dtype=np.int32), device=dev)
        ua = torch.tensor(np.ascontiguousarray(uv_a, dtype=np.float64).reshape(-1, 2), device=dev)
        ub = torch.tensor(np.ascontiguousarray(uv_b, dtype=np.float64).reshape(-1, 2), device=dev)
        n = int(ca.shape[0])
        if not (cb.shape[0] == n and ua.shape[0] == n and ub.shape[0] == n):
            raise CpeError("triangulate_host: array lengths differ")
        xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
        self._enter()
        _check(self.lib.cpe_triangulate(self._h, n, _ptr(ca), _ptr(cb), _ptr(ua), _ptr(ub), float(depth), _ptr(xyz)), "cpe_triangulate")
        self.synchronize()
        return xyz.cpu().numpy()

    def tensorise_dlc_host(self, tables, first_rows, part_of_marker, inv_sigma, thresh, N):
        """per-camera DLC tables (numpy [rows, 3*parts]) -> meas [N, C, L, 2], weight [N, C, L] (numpy) through cpe_tensorise_dlc"""
        import torch
        dev = torch.device("cuda", self.device)
        Cn = len(tables)
        meas = torch.empty((N, Cn, self.L, 2), dtype=torch.float64, device=dev)
        weight = torch.empty((N, Cn, self.L), dtype=torch.float64, device=dev)
        pm = torch.tensor(np.ascontiguousarray(part_of_marker, dtype=np.int32), device=dev)
        isg = torch.tensor(np.ascontiguousarray(inv_sigma, dtype=np.float64), device=dev)
        if pm.shape[0] != self.L or isg.shape[0] != self.L:
            raise CpeError("tensorise_dlc_host: one body part and one sigma per marker of the handle")
        for c, tab in enumerate(tables):
            t = torch.tensor(np.ascontiguousarray(tab, dtype=np.float64), device=dev)
            if t.dim() != 2 or t.shape[1] % 3:
                raise CpeError("tensorise_dlc_host: a DLC table has 3 columns per body part")
            self._enter()
            _check(self.lib.cpe_tensorise_dlc(self._h, N, Cn, c, _ptr(t), int(t.shape[0]), int(t.shape[1] // 3), int(first_rows[c]), _ptr(pm), _ptr(isg),
                                              float(thresh), _ptr(meas), _ptr(weight)), "cpe_tensorise_dlc")
            self.synchronize()                      # `t` must outlive the launch
        return meas.cpu().numpy(), weight.cpu().numpy()

    def kinematics_host(self, q, dq):
        """numpy in, numpy out (staged through HBM with torch): positions [B, N, L, 3], marker velocities [B, N, L, 3]"""
        import torch
        dev = torch.device("cuda", self.device)
        qd = torch.tensor(np.ascontiguousarray(q, dtype=np.float64), device=dev)
        dqd = torch.tensor(np.ascontiguousarray(dq, dtype=np.float64), device=dev)
        B, N = qd.shape[0], qd.shape[1]
        pos = torch.empty((B, N, self.L, 3), dtype=torch.float64, device=dev); vel = torch.empty_like(pos)
        self.forward_kinematics(qd, pos, None)
        self.marker_velocities(qd, dqd, vel)
        self.synchronize()
        return pos.cpu().numpy(), vel.cpu().numpy()

    def eval_normal(self, q, meas, weight, g, Bm, cost, gam=None, q_out=None):
        self._enter()
        _check(self.lib.cpe_eval_normal(self._h, q.shape[0], q.shape[1], _ptr(q), _ptr(meas), _ptr(weight), _ptr(g), _ptr(Bm), _ptr(cost),
                                        _ptr(gam), _ptr(q_out)), "cpe_eval_normal")
        self._leave()

    def solve(self, q_init, meas, weight, q, dq, ddq, positions, meas_err):
        B, N = q_init.shape[0], q_init.shape[1]
        stats = (abi.Stats * max(B, 1))()
        self._enter()
        st = self.lib.cpe_solve(self._h, B, N, _ptr(q_init), _ptr(meas), _ptr(weight), _ptr(q), _ptr(dq), _ptr(ddq),
                                _ptr(positions), _ptr(meas_err), stats)
        self._leave()
        _check(st, "cpe_solve", allow=(abi.OK, abi.MAX_ITER, abi.NUMERICAL))
        return st, list(stats)[:B]

    def solve_shutter(self, q_init, meas, weight, tau_bound, q, dq, ddq, positions, meas_err, tau, max_rounds=8, tol_tau=1e-6):
        """trajectory + per-camera shutter delays (acinoset_misc.py:283-285); device tensors, tau [B, C]"""
        B, N = q_init.shape[0], q_init.shape[1]
        stats = (abi.Stats * max(B, 1))()
        rounds = C.c_int32(0)
        self._enter()
        st = self.lib.cpe_solve_shutter(self._h, B, N, _ptr(q_init), _ptr(meas), _ptr(weight), float(tau_bound), int(max_rounds), float(tol_tau),
                                        _ptr(q), _ptr(dq), _ptr(ddq), _ptr(positions), _ptr(meas_err), _ptr(tau), stats, C.byref(rounds))
        self._leave()
        _check(st, "cpe_solve_shutter", allow=(abi.OK, abi.MAX_ITER, abi.NUMERICAL))
        return st, list(stats)[:B], rounds.value

    def solve_shutter_host(self, q_init, meas, weight, tau_bound, max_rounds=8, tol_tau=1e-6):
        """numpy in, numpy out (staged through HBM with torch)"""
        import torch
        dev = torch.device("cuda", self.device)
        T = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
        E = lambda *sh: torch.empty(sh, dtype=torch.float64, device=dev)
        qi, me, we = T(q_init), T(meas), T(weight)
        B, N, nq = qi.shape
        q, dq, ddq = E(B, N, nq), E(B, N, nq), E(B, N, nq)
        pos, err, tau = E(B, N, self.L, 3), E(B, N, self.n_cams, self.L, 2), E(B, self.n_cams)
        st, stats, rounds = self.solve_shutter(qi, me, we, tau_bound, q, dq, ddq, pos, err, tau, max_rounds, tol_tau)
        self.synchronize()
        return dict(status=st, q=q.cpu().numpy(), dq=dq.cpu().numpy(), ddq=ddq.cpu().numpy(), positions=pos.cpu().numpy(),
                    meas_err=err.cpu().numpy(), tau=tau.cpu().numpy(), stats=stats, rounds=rounds)

    def eom_rows(self, eopt, q, dq, ddq, rows):
        """all rows of d/dt dL/dq' - dL/dq (device tensors [B, N, nq])"""
        self._enter()
        _check(self.lib.cpe_eom_rows(self._h, C.byref(eopt), q.shape[0], q.shape[1], _ptr(q), _ptr(dq), _ptr(ddq), _ptr(rows)), "cpe_eom_rows")
        self._leave()

    def eom_residual(self, dopt, q, dq, ddq, tau, lam, grf, residual):
        """rows of the equations of motion minus the generalised forces (device tensors; tau / lam / grf may be None)"""
        self._enter()
        _check(self.lib.cpe_eom_residual(self._h, C.byref(dopt), q.shape[0], q.shape[1], _ptr(q), _ptr(dq), _ptr(ddq), _ptr(tau), _ptr(lam),
                                         _ptr(grf), _ptr(residual)), "cpe_eom_residual")
        self._leave()

    def grf_fit(self, gopt, q, dq, ddq, contact, grfz, grfxy, residual=None):
        """per-frame ground-reaction-force fit (device tensors); contact int32 [B, N, n_feet]"""
        self._enter()
        _check(self.lib.cpe_grf_fit(self._h, C.byref(gopt), q.shape[0], q.shape[1], _ptr(q), _ptr(dq), _ptr(ddq), _ptr(contact),
                                    _ptr(grfz), _ptr(grfxy), _ptr(residual)), "cpe_grf_fit")
        self._leave()

    def grf_fit_host(self, gopt, q, dq, ddq, contact):
        """numpy in, numpy out (staged through HBM with torch): grfz [B, N, nf], grfxy [B, N, nf, 4], residual [B, N, 6]"""
        import torch
        dev = torch.device("cuda", self.device)
        T = lambda a, dt: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
        qd, dqd, ddqd, cd = T(q, np.float64), T(dq, np.float64), T(ddq, np.float64), T(contact, np.int32)
        B, N, nf = qd.shape[0], qd.shape[1], gopt.n_feet
        gz = torch.empty((B, N, nf), dtype=torch.float64, device=dev); gxy = torch.empty((B, N, nf, 4), dtype=torch.float64, device=dev)
        res = torch.empty((B, N, 6), dtype=torch.float64, device=dev)
        self.grf_fit(gopt, qd, dqd, ddqd, cd, gz, gxy, res)
        self.synchronize()
        return gz.cpu().numpy(), gxy.cpu().numpy(), res.cpu().numpy()

    # ---- physics-based trajectory model (config 4) -------------------------------------------------------
    def solve_kinetic(self, kopts, q_init, meas, weight, stance, q, dq, ddq, positions, meas_err, tau=None, lam=None, grf=None, slack=None, grf_fixed=None, tau_box=None, grf_box=None):
        """device tensors (stance int32 [B, N, n_feet]; at most one of: grf_fixed [B, N, n_feet, 3] = prescribed net foot forces, tau_box [B, N, n_motors, 2] =
        (lower, upper) bound of every torque, grf_box [B, N, n_feet, 3, 2] = (lower, upper) of the net (z, x, y) foot forces); returns (status, [Stats], [KineticStats])"""
        B, N = q_init.shape[0], q_init.shape[1]
        stats = (abi.Stats * max(B, 1))(); ks = (abi.KineticStats * max(B, 1))()
        if sum(a is not None for a in (grf_fixed, tau_box, grf_box)) > 1:
            raise ValueError("prescribed foot forces, torque boxes and force boxes are separate entry points")
        fn = self.lib.cpe_solve_kinetic_bounded if tau_box is not None else (self.lib.cpe_solve_kinetic_force_box if grf_box is not None else self.lib.cpe_solve_kinetic_fixed)
        self._enter()
        st = fn(self._h, C.byref(kopts), B, N, _ptr(q_init), _ptr(meas), _ptr(weight), _ptr(stance), _ptr(tau_box if tau_box is not None else (grf_box if grf_box is not None else grf_fixed)), _ptr(q), _ptr(dq), _ptr(ddq),
                _ptr(positions), _ptr(meas_err), _ptr(tau), _ptr(lam), _ptr(grf), _ptr(slack), stats, ks)
        self._leave()
        _check(st, "cpe_solve_kinetic", allow=(abi.OK, abi.MAX_ITER, abi.NUMERICAL))
        return st, list(stats)[:B], list(ks)[:B]

    def n_constraint_rows(self):
        return sum(2 if self.sk.joint_kind[j] == abi.JOINT_REVOLUTE_Y else 1 for j in range(self.sk.n_joints))

    def solve_kinetic_host(self, kopts, q_init, meas, weight, stance, grf_fixed=None, tau_box=None, grf_box=None):
        """numpy in, numpy out (staged through HBM with torch)"""
        import torch
        dev = torch.device("cuda", self.device)
        T = lambda a, dt=np.float64: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
        qi, me, we, stn = T(q_init), T(meas), T(weight), T(stance, np.int32)
        gfx = None if grf_fixed is None else T(grf_fixed)
        tbx = None if tau_box is None else T(tau_box)
        gbx = None if grf_box is None else T(grf_box)
        B, N = qi.shape[0], qi.shape[1]
        E = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        nm, nf, nc = kopts.dyn.n_motors, kopts.dyn.n_feet, self.n_constraint_rows()
        q, dq, ddq = E(B, N, self.nq), E(B, N, self.nq), E(B, N, self.nq)
        pos, err = E(B, N, self.L, 3), E(B, N, self.n_cams, self.L, 2)
        tau, lam, grf, slack = E(B, N, nm), E(B, N, nc), E(B, N, nf, 5), E(B, N, self.nq)
        st, stats, ks = self.solve_kinetic(kopts, qi, me, we, stn, q, dq, ddq, pos, err, tau, lam, grf, slack, grf_fixed=gfx, tau_box=tbx, grf_box=gbx)
        self.synchronize()
        c = lambda t: t.cpu().numpy()
        return dict(status=st, q=c(q), dq=c(dq), ddq=c(ddq), positions=c(pos), meas_err=c(err), tau=c(tau), lam=c(lam), grf=c(grf), slack=c(slack),
                    stats=stats, kstats=ks)

    def eval_kinetic_nodes_host(self, kopts, q, meas, weight, stance):
        """one evaluation of the physics terms per node (cpe_eval_kinetic_nodes); numpy in, dict of numpy arrays out"""
        import torch
        dev = torch.device("cuda", self.device)
        T = lambda a, dt=np.float64: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
        qd, me, we, stn = T(q), T(meas), T(weight), T(stance, np.int32)
        B, N = qd.shape[0], qd.shape[1]
        E = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
        out = dict(f=E(B, N, 64), stat=E(B, N, 8), g=E(B, N, 84), Huu=E(B, N, 84, 84), Hfu=E(B, N, 64, 84), Hff=E(B, N, 64, 64))
        meta = torch.zeros((B, N, 65), dtype=torch.int32, device=dev)
        self._enter()
        _check(self.lib.cpe_eval_kinetic_nodes(self._h, C.byref(kopts), B, N, _ptr(qd), _ptr(me), _ptr(we), _ptr(stn), _ptr(out["f"]), _ptr(out["stat"]),
                                               _ptr(out["g"]), _ptr(out["Huu"]), _ptr(out["Hfu"]), _ptr(out["Hff"]), _ptr(meta)), "cpe_eval_kinetic_nodes")
        self._leave()
        self.synchronize()
        res = {k: v.cpu().numpy() for k, v in out.items()}
        res["meta"] = meta.cpu().numpy()
        return res

    # ---- host-pointer conveniences (numpy in, numpy out; PCIe-inclusive) -------------------------------
    def eval_resjac_host(self, q, meas, weight, want_cost=True):
        q = np.ascontiguousarray(q, dtype=np.float64); meas = np.ascontiguousarray(meas, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        B, N = q.shape[:2]
        r = np.empty((B, N, self.n_cams, self.L, 2)); J = np.empty((B, N, self.n_cams, self.S, 2))
        eps = np.empty((B, N, self.nq)); cost = np.empty((B, N)) if want_cost else None
        _check(self.lib.cpe_eval_resjac_host(self._h, B, N, _ptr(q), _ptr(meas), _ptr(weight), _ptr(r), _ptr(J), _ptr(eps), _ptr(cost)),
               "cpe_eval_resjac_host")
        return r, J, eps, cost

    def solve_host(self, q_init, meas, weight):
        q_init = np.ascontiguousarray(q_init, dtype=np.float64); meas = np.ascontiguousarray(meas, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        B, N = q_init.shape[:2]
        q = np.empty_like(q_init); dq = np.empty_like(q_init); ddq = np.empty_like(q_init)
        pos = np.empty((B, N, self.L, 3)); me = np.empty((B, N, self.n_cams, self.L, 2))
        stats = (abi.Stats * max(B, 1))()
        st = self.lib.cpe_solve_host(self._h, B, N, _ptr(q_init), _ptr(meas), _ptr(weight), _ptr(q), _ptr(dq), _ptr(ddq),
                                     _ptr(pos), _ptr(me), stats)
        _check(st, "cpe_solve_host", allow=(abi.OK, abi.MAX_ITER, abi.NUMERICAL))
        return dict(status=st, q=q, dq=dq, ddq=ddq, positions=pos, meas_err=me, stats=list(stats)[:B])
