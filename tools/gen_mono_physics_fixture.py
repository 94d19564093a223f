#!/usr/bin/env python3
"""Writes tests/golden/mono_physics_warm_start.npz: the monocular kinematic estimate (one camera, pose + motion priors) that
tests/test_gpu_kinetic.py::test_monocular_physics_with_pose_prior_and_detected_contacts warm-starts its physics stage from, and the stance table
determine_contacts' rule finds on it.  Runs the test's own kinematic stage on the GPU (python tools/gen_mono_physics_fixture.py [out.npz])."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cheetah_pose_estimation_amd import _lib, abi
import test_gpu_kinetic as T

handles = []
def factory(sk, cams, opts=None, priors=None):
    h = _lib.Handle(sk, cams, opts if opts is not None else abi.default_options(), priors); handles.append(h); return h
sk, cam1, d, hk, kin = T._monocular_stage(factory, 200)
stance, contacts = T._detected_stance(hk, kin["q"][0], kin["dq"][0], 0, 120.0, 200)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "mono_physics_warm_start.npz")
np.savez_compressed(out, q=kin["q"][0], dq=kin["dq"][0], stance=stance.astype(np.int32), iterations=kin["stats"][0].iterations, status=kin["stats"][0].status)
print("wrote", out, "kinematic stage:", kin["stats"][0].iterations, "iterations, status", kin["stats"][0].status, "stance frames", int(stance.sum()))
for h in handles: h.close()
