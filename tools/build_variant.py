"""Experiment builds of ONE translation unit: python tools/build_variant.py <file.hip> <tag> [-DFLAG=..] ...
-> trinerflet_amd/_variants/lib_<tag>.so (the other objects are the in-tree build's).  Run a tool against it with
TNL_LIB_PATH=trinerflet_amd/_variants/lib_<tag>.so.  Timing experiments only; nothing in the product loads these."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trinerflet_amd import build as B   # noqa: E402

src, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
per_file = [] if "--no-per-file-flags" in flags else None      # drop build.py's PER_FILE flags of this file
flags = [f for f in flags if f != "--no-per-file-flags"]
B.build()
out_dir = os.path.join(ROOT, "trinerflet_amd", "_variants")
os.makedirs(out_dir, exist_ok=True)
base = os.path.basename(src)
obj = os.path.join(out_dir, base[:-4] + "_" + tag + ".o")
subprocess.check_call([B.HIPCC, *B.COMMON, *(B.PER_FILE.get(base, []) if per_file is None else per_file), *flags, "-c", os.path.join(B.CSRC, base), "-o", obj])
objs = [obj if os.path.basename(o) == base[:-4] + ".o" else o
        for o in (os.path.join(B.OBJ, os.path.basename(s)[:-4] + ".o") for s in sorted(os.listdir(B.CSRC)) if s.endswith(".hip"))]
lib = os.path.join(out_dir, f"lib_{tag}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib, *objs])
print(lib)
