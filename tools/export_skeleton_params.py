#!/usr/bin/env python3
"""Export the numeric link parameters of the reference's cheetah skeletons to JSON.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's
pure-data module `cheetah_params.py` (a dict literal, cheetah_params.py:3-566) and writes
numbers -- mass / length / radius per link for every animal -- to
cheetah_pose_estimation_amd/data/skeleton_params.json.  No reference source text is copied.
"""
import importlib.util
import json
import os
import sys

REF = "/root/reference/cheetah_params.py"
OUT = os.path.join(os.path.dirname(__file__), "..", "cheetah_pose_estimation_amd", "data", "skeleton_params.json")


def main():
    spec = importlib.util.spec_from_file_location("_ref_cheetah_params", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    out = {}
    for animal, p in mod.parameters.items():
        rec = {}
        for key in ("neck", "body_F", "body_B", "tail0", "tail1"):
            rec[key] = {k: float(p[key][k]) for k in ("mass", "radius", "length")}
        for side in ("front", "back"):
            rec[side] = {seg: {k: float(p[side][seg][k]) for k in ("mass", "radius", "length")}
                         for seg in ("thigh", "calf", "hock")}
        rec["friction_coeff"] = float(p["friction_coeff"])
        out[animal] = rec
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.abspath(OUT), list(out))


if __name__ == "__main__":
    main()
