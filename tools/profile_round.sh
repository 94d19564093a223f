#!/bin/bash
# rocprofv3 passes of one round on the GPU box (run through gpurun from the repo root): kernel-trace statistics of the bench command, then the
# PMC passes -- FETCH_SIZE and WRITE_SIZE in SEPARATE runs, one derived TCC counter per pass, SQ counters in a third -- on a bounded workload
# (512 sequences: one full launch window).  The program follows `--` directly (python3, no wrapper).  Summaries -> profiles/ by the caller.
set -o pipefail
R=${1:-r03}
OUT=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${R}_stats -o ${R} -- python3 /root/repo/bench.py --steps 10 --warmup 10 --solve-batch 2048 --cfg4-batch 16 --no-cpu > $OUT/${R}_bench_rocprof.log 2>&1 && echo "stats pass done" &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${R}_pmc_fetch -o ${R} -- python3 /root/repo/bench.py --steps 1 --warmup 0 --batch 512 --solve-batch 512 --cfg3-batch 256 --cfg4-batch 16 --no-cpu --no-l24 > $OUT/${R}_pmc_fetch.log 2>&1 && echo "fetch pass done" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${R}_pmc_write -o ${R} -- python3 /root/repo/bench.py --steps 1 --warmup 0 --batch 512 --solve-batch 512 --cfg3-batch 256 --cfg4-batch 16 --no-cpu --no-l24 > $OUT/${R}_pmc_write.log 2>&1 && echo "write pass done" &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/${R}_pmc_sq -o ${R} -- python3 /root/repo/bench.py --steps 1 --warmup 0 --batch 512 --solve-batch 512 --cfg3-batch 256 --cfg4-batch 16 --no-cpu --no-l24 > $OUT/${R}_pmc_sq.log 2>&1 && echo "sq pass done"
ls $OUT/${R}_stats $OUT/${R}_pmc_fetch | head
