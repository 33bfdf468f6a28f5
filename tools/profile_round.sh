# usage (GPU box): bash tools/profile_round.sh [tag] [workload]   -- writes gpurun_out/<tag>/...: the bench line of the
# workload, kernel trace + per-launch table, two timelines, FETCH/WRITE/SQ counter passes, and (base only) the kernel
# trace of the 800x800 max_steps-4096 render
TAG=${1:-r04a}
WL=${2:-base}
set -x
cd /root/repo
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
if [ "$WL" = "base" ]; then
TNL_BENCH_DETAIL=$OUT/bench_detail.json python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
else
TNL_BENCH_DETAIL=$OUT/bench_${WL}_detail.json python bench.py --workload $WL --no-extras --no-cpu-baseline > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 /root/repo/bench.py --workload $WL --steps 32 --warmup 16 --no-cpu-baseline --no-extras > $OUT/bench_traced.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 /root/repo/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 /root/repo/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -o s -- python3 /root/repo/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
if [ "$WL" = "base" ]; then
PYTHONPATH=/root/repo rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_infer -o i -- python3 /root/repo/tools/bench_infer.py > $OUT/infer.txt 2>/dev/null
cp $(find $OUT/trace_infer -name "*kernel_stats.csv" | head -1) $OUT/infer_kernel_stats.csv
fi
cd /root/repo
python tools/per_launch.py $OUT/trace $OUT/per_launch.csv > /dev/null
python tools/step_timeline.py $OUT/trace k_mse_loss 30 > $OUT/step_timeline.txt
python tools/step_timeline.py $OUT/trace k_mse_loss k_packbits > $OUT/refresh_step_timeline.txt
python tools/kernel_per_step.py $OUT/trace "k_field_bwd" $OUT/field_bwd_per_step.csv > /dev/null
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/$TAG 2
python tools/pmc_kernels.py $OUT/pmc_sq1 > $OUT/pmc_sq1.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_${WL}_kernel_stats.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/trace $OUT/trace_infer
ls -la $OUT
