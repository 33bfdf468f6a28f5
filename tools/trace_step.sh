# usage (GPU box): bash tools/trace_step.sh [workload] [tag] -- kernel trace of a short bench run -> gpurun_out/<tag>_step_timeline.txt,
# <tag>_per_launch.csv (tools/step_timeline.py, tools/per_launch.py)
WL=${1:-base}; TAG=${2:-trace}
OUT=/root/repo/gpurun_out/$TAG.d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 /root/repo/bench.py --workload $WL --steps 32 --warmup 16 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd /root/repo
python tools/per_launch.py $OUT gpurun_out/${TAG}_per_launch.csv > /dev/null
python tools/step_timeline.py $OUT k_mse_loss 30 > gpurun_out/${TAG}_step_timeline.txt
rm -rf $OUT
