"""developer helper: run bench.py's resjac leg against an experimental library build"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cheetah_pose_estimation_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu", "--no-solve"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
