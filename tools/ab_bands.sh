#!/bin/bash
# A/B in one session: live rectangles cut down to band pieces (default) vs whole rectangles.
mkdir -p gpurun_out
for rep in 1 2; do
  for b in 1 0; do
    TNL_LIVE_BANDS=$b python bench.py --steps 64 --warmup 32 > gpurun_out/ab_bands_${b}_${rep}.json 2> gpurun_out/ab_bands_${b}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_bands_${b}_${rep}.json").read().strip().splitlines()[-1])
k = {x["name"]: x for x in d.get("kernels", [])} if isinstance(d.get("kernels"), list) else d.get("kernels", {})
print("bands=${b} rep=${rep}", d["ms_per_step"], d["value"], d["roofline"].get("kernel"), d["roofline"].get("achieved"), d["roofline"].get("frac"))
ad = d["config"].get("adam_deferred") or d.get("adam_deferred")
print("   share", None if ad is None else ad.get("band_pieces_share_of_rectangle"))
PY
  done
done
