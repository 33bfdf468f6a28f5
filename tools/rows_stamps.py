"""Per-slot cycle stamps of k_field_bwd_rows (library built with -DTNL_ROWS_STAMP=1): wave 0 of workgroup 0, its first 63
tiles; prints the median cycles per slot over tiles 8..62, the tile total, and the shader clock (s_memtime ticks per
100-MHz s_memrealtime tick).  python tools/rows_stamps.py [workload]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.argv = [sys.argv[0]] + sys.argv[1:]
import tools.bench_field_bwd as B   # noqa: E402
from trinerflet_amd import _lib as L   # noqa: E402

B.main()
buf = np.zeros(64 * 64, dtype=np.uint64)
rc = L.lib().tnl_debug_rows_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
st = buf.reshape(64, 64)
real = st[63, :63].astype(np.int64)
tiles = st[8:62].astype(np.int64)
nslot = int((tiles[0] > 0).sum())
d = np.diff(tiles[:, :nslot], axis=1)
med = np.median(d, axis=0)
print("slots", nslot - 1, "median cycles per slot:", [int(x) for x in med])
tot = tiles[1:, 0] - tiles[:-1, 0]
print("tile period (cycles) median", int(np.median(tot)), "min", int(tot.min()), "max", int(tot.max()), " sum of slot medians", int(med.sum()))
dr = np.diff(real[8:62])
print("clock: %.3f GHz (memtime ticks per memrealtime 10 ns tick = %.2f)" % (np.median(tot / dr) * 0.1, np.median(tot / dr)))
