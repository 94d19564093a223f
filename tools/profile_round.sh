#!/bin/bash
# rocprofv3 passes of one round on the GPU box (run through gpurun from the repo root): kernel-trace statistics of the bench command, then the
# PMC passes -- FETCH_SIZE and WRITE_SIZE in SEPARATE runs, one derived TCC counter per pass, SQ counters in a third -- on a bounded workload
# (512 sequences: one full launch window).  The program follows `--` directly (python3, no wrapper).  The rocpd databases are summarised ON THE
# BOX (tools/export_rocprof_stats.py, tools/summarise_pmc.py) and deleted: only the summaries travel back (gpurun_out/ is capped at 64 MiB).
set -o pipefail
R=${1:-r03}
OUT=/root/repo/gpurun_out
T=/root/repo/tools
PMC_ARGS="--steps 1 --warmup 0 --batch 512 --solve-batch 512 --cfg3-batch 256 --cfg4-batch 16 --no-cpu --no-l24 --gen-workers 1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/${R}_stats -o ${R} -- python3 /root/repo/bench.py --steps 10 --warmup 10 --solve-batch 2048 --cfg4-batch 16 --no-cpu --gen-workers 1 > $OUT/${R}_bench_rocprof.log 2>&1 &&
python3 $T/export_rocprof_stats.py /tmp/${R}_stats/${R}_results.db $OUT/${R}_kernel_stats.csv && echo "stats pass done" &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/${R}_pmc_fetch -o ${R} -- python3 /root/repo/bench.py $PMC_ARGS > $OUT/${R}_pmc_fetch.log 2>&1 && echo "fetch pass done" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/${R}_pmc_write -o ${R} -- python3 /root/repo/bench.py $PMC_ARGS > $OUT/${R}_pmc_write.log 2>&1 && echo "write pass done" &&
python3 $T/summarise_pmc.py /tmp/${R}_pmc_fetch /tmp/${R}_pmc_write $OUT/${R}_pmc.json 102400 "k_resjac<false=32160" "k_lm_step<3, 0>=33776" "k_lm_back<3>=26496" "k_frame_normal<true>=11504" &&
PMC_C=1 python3 $T/summarise_pmc.py /tmp/${R}_pmc_fetch /tmp/${R}_pmc_write $OUT/${R}_pmc_cfg4.json 3168 "k_dyn_eval=93184" "k_dyn_jac=97520" "k_dyn_assemble=225472" "k_dyn_schur=170912" "k_dyn_gather=70112" &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d /tmp/${R}_pmc_sq -o ${R} -- python3 /root/repo/bench.py $PMC_ARGS > $OUT/${R}_pmc_sq.log 2>&1 &&
python3 $T/export_rocprof_stats.py /tmp/${R}_pmc_sq/${R}_results.db $OUT/${R}_pmc_sq.csv counters && echo "sq pass done"
rc=$?
ls -la /tmp/${R}_* | head -20
exit $rc
