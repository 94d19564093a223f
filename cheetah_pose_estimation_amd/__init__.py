"""MI355X-native trajectory-optimisation back end for cheetah 3D pose (hot path of
zicodasilva/cheetah_pose_estimation): HIP kernels + C ABI (csrc/, include/cpe.h) and the host-side
mirror of the reference's estimator API."""
from . import abi, skeleton, synth  # noqa: F401

__all__ = ["abi", "skeleton", "synth"]
