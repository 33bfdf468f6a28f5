cd /root/repo
for flags in "$@"; do
TNL_HIPCC_FLAGS="-DTNL_ROWS_STAMP=1 $flags" python -m trinerflet_amd.build --force > /dev/null || echo build failed
echo "== $flags"
python tools/rows_stamps.py base 2>&1 | grep -v amdgpu.ids
done
python -m trinerflet_amd.build --force > /dev/null
