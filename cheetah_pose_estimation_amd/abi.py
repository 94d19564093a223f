"""ctypes mirror of include/cpe.h (the C ABI).  Pure data definitions: no compute lives here."""
import ctypes as C

MAX_LINKS = 20
MAX_MARKERS = 32
MAX_CAMS = 18
MAX_JOINTS = 16
MAX_BOUNDS = 32
MAX_NQ = 3 + 3 * MAX_LINKS
MAX_GMM = 8
NX = 28
MAX_WINDOW = 4

OK, MAX_ITER, NUMERICAL, BAD_ARG, NO_DEVICE, HIP_ERROR = 0, 1, 2, -1, -2, -3
JOINT_REVOLUTE_Y, JOINT_HOOKE_YZ = 0, 1
CAM_FISHEYE, CAM_PINHOLE = 0, 1

d3 = C.c_double * 3


class Skeleton(C.Structure):
    _fields_ = [
        ("n_links", C.c_int32), ("n_markers", C.c_int32), ("n_joints", C.c_int32), ("n_bounds", C.c_int32),
        ("parent", C.c_int32 * MAX_LINKS),
        ("attach", d3 * MAX_LINKS),
        ("com", d3 * MAX_LINKS),
        ("mass", C.c_double * MAX_LINKS),
        ("marker_link", C.c_int32 * MAX_MARKERS),
        ("marker_off", d3 * MAX_MARKERS),
        ("joint_parent", C.c_int32 * MAX_JOINTS),
        ("joint_child", C.c_int32 * MAX_JOINTS),
        ("joint_kind", C.c_int32 * MAX_JOINTS),
        ("bound_a", C.c_int32 * MAX_BOUNDS),
        ("bound_b", C.c_int32 * MAX_BOUNDS),
        ("bound_lo", C.c_double * MAX_BOUNDS),
        ("bound_up", C.c_double * MAX_BOUNDS),
        ("motion_w", C.c_double * MAX_NQ),
        ("rel_ref", C.c_int32 * MAX_NQ),
        ("rel_sign", C.c_double * MAX_NQ),
    ]

    @property
    def nq(self):
        return 3 + 3 * self.n_links


class Camera(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("_pad", C.c_int32),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("D", C.c_double * 4), ("R", C.c_double * 9), ("t", C.c_double * 3), ("mult", C.c_double),
    ]


class Priors(C.Structure):
    _fields_ = [
        ("gmm_k", C.c_int32), ("gmm_dim", C.c_int32),
        ("gmm_logw", C.c_double * MAX_GMM),
        ("gmm_mu", (C.c_double * NX) * MAX_GMM),
        ("gmm_P", ((C.c_double * NX) * NX) * MAX_GMM),
        ("lr_window", C.c_int32), ("_pad", C.c_int32),
        ("lr_coef", (C.c_double * (MAX_WINDOW * NX)) * NX),
        ("lr_b", C.c_double * NX),
        ("lr_w", C.c_double * NX),
    ]


class Options(C.Structure):
    _fields_ = [
        ("h", C.c_double), ("loss_a", C.c_double), ("loss_b", C.c_double), ("loss_c", C.c_double),
        ("cost_scale", C.c_double), ("bound_penalty", C.c_double), ("bound_tol", C.c_double), ("lambda0", C.c_double),
        ("tol_step", C.c_double), ("tol_cost", C.c_double),
        ("max_iter", C.c_int32), ("curvature", C.c_int32), ("max_outer", C.c_int32), ("_pad", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iterations", C.c_int32),
        ("cost", C.c_double), ("cost_meas", C.c_double), ("cost_model", C.c_double),
        ("cost_pose", C.c_double), ("cost_motion", C.c_double),
        ("lam", C.c_double), ("max_constraint", C.c_double), ("max_bound_violation", C.c_double),
        ("outer", C.c_int32), ("_pad", C.c_int32),
    ]


class GrfOptions(C.Structure):
    """mirror of cpe_grf_options (include/cpe.h)"""
    _fields_ = [
        ("root_inertia", C.c_double * 3), ("friction_ratio", C.c_double), ("force_max", C.c_double),
        ("regularisation", C.c_double), ("gravity", C.c_double),
        ("n_feet", C.c_int32), ("foot_marker", C.c_int32 * 4), ("iterations", C.c_int32),
    ]


class EomOptions(C.Structure):
    """mirror of cpe_eom_options (include/cpe.h)"""
    _fields_ = [("gravity", C.c_double), ("link_inertia", d3 * MAX_LINKS)]


class DynOptions(C.Structure):
    """mirror of cpe_dyn_options (include/cpe.h)"""
    _fields_ = [("eom", EomOptions), ("n_feet", C.c_int32), ("n_motors", C.c_int32), ("foot_marker", C.c_int32 * 4),
                ("motor_first", C.c_int32 * 32), ("motor_second", C.c_int32 * 32), ("motor_axis", C.c_int32 * 32)]


class KineticOptions(C.Structure):
    """mirror of cpe_kinetic_options (include/cpe.h): the physics-based trajectory model of estimate_kinetics"""
    _fields_ = [("dyn", DynOptions), ("w_slack", C.c_double), ("w_torque", C.c_double), ("w_smooth", C.c_double), ("friction", C.c_double),
                ("force_max", C.c_double), ("grfz_min", C.c_double), ("foot_height_tol", C.c_double), ("foot_height_min", C.c_double),
                ("ground_height", C.c_double), ("slip_max", C.c_double), ("zvel_max", C.c_double), ("slack_lo", C.c_double), ("slack_hi", C.c_double),
                ("reg_force", C.c_double), ("kappa_force", C.c_double), ("kappa_height", C.c_double), ("kappa_slip", C.c_double), ("kappa_slack", C.c_double),

                ("lm_force_damping", C.c_double), ("lm_wall_damping", C.c_double), ("inner_iterations", C.c_int32), ("_pad", C.c_int32)]


class KineticStats(C.Structure):
    """mirror of cpe_kinetic_stats (include/cpe.h)"""
    _fields_ = [("cost_torque", C.c_double), ("cost_energy", C.c_double), ("cost_eom", C.c_double), ("max_slack", C.c_double),
                ("max_base_rows", C.c_double), ("max_violation", C.c_double), ("inner_max", C.c_int32), ("_pad", C.c_int32)]


def default_kinetic_options(dyn: DynOptions, fps: float = 120.0, kinetic_dataset: bool = False) -> KineticOptions:
    """the reference's values (acinoset_opt.py:494-506, :780, :905-921; acinoset_misc.py:1140-1167; run_dataset.py:984); same numbers as
    cpe_default_kinetic_options() in csrc/cpe_api.hip"""
    o = KineticOptions()
    C.memmove(C.byref(o.dyn), C.byref(dyn), C.sizeof(DynOptions))
    o.w_slack, o.w_torque, o.w_smooth = 10e3, 1.0, 0.1 / (fps * fps)
    o.friction, o.force_max, o.grfz_min = 0.8, 5.0, 0.01
    o.foot_height_tol = 0.03 if kinetic_dataset else 0.1
    o.foot_height_min, o.ground_height, o.slip_max = 0.0, 0.0, 1.0
    o.zvel_max = 1.0 if kinetic_dataset else 0.0       # `foot_z_vel <= 1` is a rule of the kinetic dataset only (acinoset_opt.py:807-810)
    o.slack_lo, o.slack_hi = -2.0, 2.0                 # bound_eom_error of run_dataset.py:984
    o.reg_force, o.kappa_force, o.kappa_height, o.kappa_slip, o.kappa_slack = 1e-4, 1e5, 1e6, 1e2, 1e6
    o.lm_force_damping, o.lm_wall_damping = 10.0, 10.0
    o.inner_iterations = 30
    return o


def default_options(fps: float = 120.0) -> Options:
    """Same defaults as cpe_default_options() in csrc/cpe_api.cpp."""
    o = Options()
    o.h = 1.0 / fps
    o.loss_a, o.loss_b, o.loss_c = 3.0, 10.0, 20.0   # acinoset_misc.py:479-481
    o.cost_scale = 1e-3                              # acinoset_opt.py:602
    o.bound_penalty = 1e4
    o.bound_tol = 1e-6
    o.max_outer = 8
    o.lambda0 = 1e-4
    o.tol_step = 1e-8
    o.tol_cost = 1e-9
    o.max_iter = 200
    o.curvature = 0
    return o
