"""Builds libtrinerflet_hip.so (gfx950) in-tree with hipcc.  `python -m trinerflet_amd.build`.

One object per csrc/*.hip (so an edit recompiles one file), linked into
trinerflet_amd/libtrinerflet_hip.so, which is the C-ABI library declared in include/trinerflet_hip.h.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libtrinerflet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]
# extra flags for A/B experiments on the GPU box (e.g. TNL_HIPCC_FLAGS="-DTNL_FWD_NT=0"), empty in normal builds
COMMON += os.environ.get("TNL_HIPCC_FLAGS", "").split()
# the marching kernels must not contract a*b+c on their own: bit-exact sample counts (see raymarch.hip)
# the MFMA kernels run one wave per SIMD and post-process every accumulator tile on the VALU: keep MFMA results in
# VGPRs instead of AGPRs (saves the v_accvgpr_read per element; field backward 1.06 -> 1.01 ms at base)
MFMA_VGPR = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# the SLP vectoriser packs adjacent fp32 multiply-adds into v_pk_*_f32: a gain for some kernels (the hidden-64 field forward,
# the replay of deferred optimiser steps), a loss for others -- switched off per object where an A/B on one box said so
# (profiles/r06o_ab_no_slp.txt: hidden-128 field forward -10 %, one-kernel render of a trained field -7 %)
NO_SLP = ["-fno-slp-vectorize"]
PER_FILE = {"raymarch.hip": ["-ffp-contract=off"], "rays.hip": ["-ffp-contract=off"],
            "field_bwd.hip": MFMA_VGPR,
            # the rows backward: the scheduler told to favour ILP (0.583-0.597 -> 0.570-0.574 ms at base in four alternating rounds,
            # small equal: profiles/r06o_ab_no_slp.txt section 6; the same switch costs field.hip and wavelet.hip 0.05-0.08 ms)
            "field_bwd_rows.hip": MFMA_VGPR + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "field.hip": MFMA_VGPR, "scatter.hip": MFMA_VGPR,
            "render.hip": MFMA_VGPR + NO_SLP, "field_h128.hip": MFMA_VGPR + NO_SLP + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}   # (+ max-ilp: 1.26 -> 1.20 ms)
# A/B builds on the GPU box: TNL_HIPCC_FILE_FLAGS="scatter.hip:-fno-slp-vectorize;wavelet.hip:-DX=1 -DY=2" adds flags per object
for _spec in filter(None, os.environ.get("TNL_HIPCC_FILE_FLAGS", "").split(";")):
    _name, _flags = _spec.split(":", 1)
    PER_FILE[_name.strip()] = PER_FILE.get(_name.strip(), []) + _flags.split()
# objects that are another source compiled under a macro
INCLUDES = {"field_h128.hip": ["field.hip"]}


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o, headers + [os.path.join(CSRC, d) for d in INCLUDES.get(os.path.basename(s), [])]):
            jobs.append([HIPCC, *COMMON, *PER_FILE.get(os.path.basename(s), []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
