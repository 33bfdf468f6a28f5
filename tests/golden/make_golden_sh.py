"""Generates tests/golden/sh_reference.npz by EVALUATING THE REFERENCE's own formulas (build container only):

    python tests/golden/make_golden_sh.py

The reference's spherical-harmonics kernel (aux_libs/shencoder/src/shencoder.cu:28-355) is CUDA-only, but its body is
a list of assignments `outputs[k] = <polynomial in x, y, z>;` / `dx[k] = ...; dy[k] = ...; dz[k] = ...;` over the locals
declared at :44-46.  This script reads that file where it lies under /root/reference, takes those statements as they
stand (a C float literal `1.5f` becomes numpy float32 1.5) and evaluates them with numpy in float32 on 96 directions:
the values are the reference kernel's arithmetic, executed here, for every degree 1..8 and for the optional
derivatives.  Nothing of the source text is stored -- only inputs and outputs:

  dirs [96,3]   unit directions (+ the axes, + a few non-unit vectors: the polynomials are evaluated as written)
  out  [96,64]  outputs[0..63]            dx, dy, dz [96,64]  the derivative tables
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/aux_libs/shencoder/src/shencoder.cu"


def main():
    text = open(SRC).read()
    body = text[text.index("__global__ void kernel_sh("):text.index("__global__ void kernel_sh_backward(")]
    rng = np.random.default_rng(8)
    d = rng.standard_normal((96, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:3] = np.eye(3, dtype=np.float32)
    d[3:6] = -np.eye(3, dtype=np.float32)
    d[6:10] *= np.array([[0.5], [1.7], [0.01], [3.0]], np.float32)
    env = {"x": d[:, 0].copy(), "y": d[:, 1].copy(), "z": d[:, 2].copy()}
    lit = lambda s: re.sub(r"(?<![\w.])(\d+\.\d+(?:e[-+]?\d+)?|\d+\.|\.\d+)f\b", r"np.float32(\1)", s)
    # the locals of shencoder.cu:44-46 (`scalar_t xy=x*y, xz=x*z, ...;`)
    for m in re.finditer(r"scalar_t\s+((?:\w+\s*=\s*[\w*]+\s*,\s*)*\w+\s*=\s*[\w*]+)\s*;", body):
        for part in m.group(1).split(","):
            name, expr = (t.strip() for t in part.split("="))
            if name in ("x", "y", "z"):
                continue
            env[name] = eval(expr, {}, env).astype(np.float32)
    out = {k: np.zeros((96, 64), np.float32) for k in ("outputs", "dx", "dy", "dz")}
    n = 0
    for m in re.finditer(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*([^;]+);", body, re.M):
        arr, k, expr = m.group(1), int(m.group(2)), m.group(3)
        val = eval(lit(expr), {"np": np}, env)
        out[arr][:, k] = np.broadcast_to(np.asarray(val, np.float32), (96,))
        n += 1
    assert n == 4 * 64, n
    path = os.path.join(HERE, "sh_reference.npz")
    np.savez_compressed(path, dirs=d, out=out["outputs"], dx=out["dx"], dy=out["dy"], dz=out["dz"])
    print("wrote", path, os.path.getsize(path), "bytes;", n, "statements of", SRC, "evaluated")


if __name__ == "__main__":
    main()
