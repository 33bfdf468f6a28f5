"""Cycle stamps of the split hidden-128 backward (library built with -DTNL_BWD_STAMP=1 or 2 for PART 1 / 2: tools/build_variant.py):
wave 0 of workgroup 0, its first 64 super-tiles; prints the median cycles between consecutive stamps over super-tiles 8..62.
TNL_LIB_PATH=trinerflet_amd/_variants/lib_<tag>.so python tools/bwd_stamps.py large"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tools.bench_field_bwd as B   # noqa: E402
from trinerflet_amd import _lib as L   # noqa: E402

B.main()
buf = np.zeros(64 * 32, dtype=np.uint64)
assert L.lib().tnl_debug_bwd_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf.reshape(64, 32).astype(np.int64)
tiles = st[8:62]
n = int((tiles[0] > 0).sum())
d = np.diff(tiles[:, :n], axis=1)
print("stamps", n, "median cycles between stamps:", [int(x) for x in np.median(d, axis=0)])
tot = tiles[1:, 0] - tiles[:-1, 0]
print("super-tile period median", int(np.median(tot)), "sum of medians", int(np.median(d, axis=0).sum()))
