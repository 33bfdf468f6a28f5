#!/bin/bash
# A/B of the walk-kernel forms and builds (GPU box):  bash tools/ab_idwt.sh  -> gpurun_out/ab_idwt.txt
# libs: the in-tree build and trinerflet_amd/_variants/lib_*.so (tools/build_variant.py)
export PYTHONPATH=.
out=gpurun_out/ab_idwt.txt; mkdir -p gpurun_out; : > $out
for lib in "" $(ls trinerflet_amd/_variants/lib_*.so 2>/dev/null); do
  for pair in 0 2 4; do
    for rep in 1 2; do
      echo "== lib=${lib:-in-tree} pair=$pair rep=$rep" >> $out
      TNL_LIB_PATH=$lib python tools/bench_idwt.py 32 1024 512 4=$pair 2>&1 | grep -v "^torch" >> $out
    done
  done
done
cat $out
