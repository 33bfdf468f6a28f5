# usage (GPU box): bash tools/ab_old_new.sh -- HEAD against the round's starting tree (_old_r06a = `git archive <commit> | tar -x -C _old_r06a` inside
# the repo directory so that it travels to the box, built there; not tracked, delete it afterwards), with and without the layout pass's 4 rows per
# workgroup; base and small alternating on one box.  Section times differ by where the side chain lands: compare the steps.
cd _old_r06a && python -m trinerflet_amd.build > /dev/null 2>&1; cd ..
line() { (cd $1 && python bench.py --workload $2 --no-cpu-baseline --no-extras --steps 64 --warmup 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],4), round(c["ms_per_step_over_whole_periods"],4), {k[3:]:round(c[k],3) for k in c if k in ("ms_idwt_fwd","ms_march","ms_field_fwd","ms_idwt_adjoint","ms_adam_coef","ms_plane_grad_binned")})'); }
for rep in 1 2 3; do
  for flags in "" "-DTNL_LAYOUT_ROWS=1"; do
    touch trinerflet_amd/csrc/wavelet.hip; TNL_HIPCC_FLAGS="$flags" python -m trinerflet_amd.build > /dev/null 2>&1
    for wl in base small; do echo "$wl new[$flags] rep=$rep $(line . $wl)"; done
  done
  for wl in base small; do echo "$wl old rep=$rep $(line _old_r06a $wl)"; done
done | tee gpurun_out/r06_ab_old_new4.txt
touch trinerflet_amd/csrc/wavelet.hip; python -m trinerflet_amd.build > /dev/null 2>&1
