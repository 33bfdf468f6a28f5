# kernel-alone A/B of hipcc flag sets for the field backward: bash tools/ab_bwd_kernel.sh [workload] "<flags A>" "<flags B>" ...
cd /root/repo
WL=$1; shift
for flags in "$@"; do
  TNL_HIPCC_FLAGS="$flags" python -m trinerflet_amd.build --force > /dev/null || { echo "build failed: $flags"; continue; }
  echo "[$flags] $(python tools/bench_field_bwd.py $WL)"
done
python -m trinerflet_amd.build --force > /dev/null
