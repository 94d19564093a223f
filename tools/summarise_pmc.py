"""Summarise rocprofv3 --pmc passes into profiles/rNN_pmc_<name>.json: HBM bytes per launch of one or more kernels.

Counter split that works on this pool (gfx950, ROCm 7.2; MI355X_MICROARCH.md "rocprofv3 PMC slots": TCC has 4 slots, FETCH_SIZE costs 3,
WRITE_SIZE costs 2, so the two never share a pass; SQ has 8 slots):
    pass 1   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py ...
    pass 2   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py ...
    pass 3   rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ...
ONE derived TCC counter per pass: round 1's `error code 38: Request exceeds the capabilities of the hardware to collect` (gpurun_out/pmc_m2.log)
came from asking for TA_* and several TCC_EA0_* counters together with a derived one -- the profile configuration is rejected before any kernel
runs.  The round-1 pass that stayed silent for 7 minutes had the full-size workload (13.7 GB per launch, every dispatch replayed per counter
group): PMC passes here use `--steps 1 --warmup 0 --batch 256 --solve-batch 256 --no-cpu --no-l24`.  Put `python3 bench.py ...` directly after
`--` (no env / bash -c wrappers: the profiler's preloaded library has initialised the GPU before the program starts).

usage: python tools/summarise_pmc.py <fetch_dir> <write_dir> <out.json> <frames per launch> <kernel-substring>=<algorithmic bytes per frame> [...]
"""
import csv
import glob
import json
import os
import sys


def collect_db(d, counter, kernel):
    """rocprofv3's default output on ROCm 7.2 is a rocpd SQLite database: view `counters_collection` has one row per dispatch, counter (and
    dimension instance)"""
    import sqlite3
    per, meta = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(f)
        for did, val, grid, vgpr, lds, wg in db.execute(
                "select dispatch_id, value, grid_size, vgpr_count, lds_block_size, workgroup_size from counters_collection where counter_name = ? and kernel_name like ?",
                (counter, "%" + kernel + "%")):
            per[did] = per.get(did, 0.0) + float(val)
            meta = dict(grid=grid, vgpr=vgpr, lds=lds, wg=wg)
    return [per[k] for k in sorted(per)], meta


def collect(d, counter, kernel):
    v, m = collect_db(d, counter, kernel)
    if v:
        return v, m
    per, meta = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])     # one row per XCD / dimension: sum per dispatch
                    meta = dict(grid=r.get("Grid_Size"), vgpr=r.get("VGPR_Count"), lds=r.get("LDS_Block_Size"), wg=r.get("Workgroup_Size"))
    return [per[k] for k in sorted(per)], meta


def main():
    fd, wd, out, frames = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    res = {"frames_per_launch": frames, "config": {"C": int(os.environ.get("PMC_C", "6")), "L": int(os.environ.get("PMC_L", "25"))}, "kernels": {},
           "note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM) -> read side doubled; "
                   "WRITE_SIZE exact for 16-B/lane stores; counters in KiB; separate --pmc passes; the largest-grid launches of each kernel are used"}
    for spec in sys.argv[5:]:
        kernel, bpf = spec.split("=")
        fv, fm = collect(fd, "FETCH_SIZE", kernel)
        wv, wm = collect(wd, "WRITE_SIZE", kernel)
        if not fv or not wv:
            res["kernels"][kernel] = None
            continue
        top = lambda v: [x for x in v if x >= 0.5 * max(v)] or v            # full-size launches only (the last windows of a solve are partial)
        mean = lambda v: sum(v) / len(v)
        rd, wr = mean(top(fv)) * 1024.0, mean(top(wv)) * 1024.0
        alg = float(bpf) * frames
        res["kernels"][kernel] = dict(launches=len(fv), read_raw=rd, read_corrected_x2=2 * rd, write=wr, total_corrected=2 * rd + wr, algorithmic=alg,
                                      traffic_over_algorithmic=(2 * rd + wr) / alg, **{"fetch_" + k: v for k, v in fm.items()})
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: (v and round(v["traffic_over_algorithmic"], 3)) for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()
