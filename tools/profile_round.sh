# usage (GPU box): bash tools/profile_round.sh   -- writes gpurun_out/<round>/...: default bench line, kernel trace + per-launch table, two timelines, FETCH/WRITE/SQ counter passes
set -x
cd /root/repo
mkdir -p gpurun_out/r02f
python bench.py > gpurun_out/r02f/bench_default.json 2> gpurun_out/r02f/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02f/trace -o t -- python3 /root/repo/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-extras > /root/repo/gpurun_out/r02f/bench_traced.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r02f/pmc_fetch -o f -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/r02f/pmc_write -o w -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /root/repo/gpurun_out/r02f/pmc_sq1 -o s -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd /root/repo
python tools/per_launch.py gpurun_out/r02f/trace gpurun_out/r02f/per_launch.csv > /dev/null
python tools/step_timeline.py gpurun_out/r02f/trace k_mse_loss 30 > gpurun_out/r02f/step_timeline.txt
python tools/step_timeline.py gpurun_out/r02f/trace k_mse_loss k_packbits > gpurun_out/r02f/refresh_step_timeline.txt
python tools/kernel_per_step.py gpurun_out/r02f/trace "k_adam_l1_live" gpurun_out/r02f/adam_per_step.csv > /dev/null
python tools/pmc_summary.py gpurun_out/r02f/pmc_fetch gpurun_out/r02f/pmc_write gpurun_out/r02f/r02f 2
python tools/pmc_kernels.py gpurun_out/r02f/pmc_sq1 > gpurun_out/r02f/pmc_sq1.txt
rm -rf gpurun_out/r02f/pmc_fetch gpurun_out/r02f/pmc_write gpurun_out/r02f/pmc_sq1
rm -f gpurun_out/r02f/trace/t_kernel_trace.csv
ls -la gpurun_out/r02f gpurun_out/r02f/trace
