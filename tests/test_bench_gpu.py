"""bench.py's output contract (one JSON line with the driver's keys + roofline + cpu_baseline) on the tiny workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line(cuda):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "rays/s" and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert (rf["bound"], rf["unit"], rf["peak"]) in (("hbm", "GB/s", 8000.0), ("mfma", "TFLOP/s", 2500.0))
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    # the line names the section with the largest per-step time and carries the three longest with their roofs
    assert len(rf["top"]) == 3 and rf["top"][0]["ms_per_step"] >= rf["top"][1]["ms_per_step"] >= rf["top"][2]["ms_per_step"]
    for e in rf["top"]:
        assert e["bound"] in ("hbm", "mfma") and e["algorithmic_bytes"] > 0 and 0 < e["frac"] < 1
    # (chosen by an instrumented pass before the timed steps; at this tiny size the order of near-equal sections may
    # differ between that pass and the final one, so: one of the three)
    assert any(e["section"] in rf["kernel"] for e in rf["top"])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_device(cuda, scaling):
    """The N > 1 code path of bench.py itself (ray shards, slice-sharded dense work, collectives, max-over-ranks
    timing, rank-0 line), launched exactly as the driver does but with both ranks on cuda:0 over gloo -- the only way
    to execute it on a one-GPU box.  RCCL replaces gloo on a real node; the call sequence is the same."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--backend", "gloo", "--same-device", "--scaling", scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert "dp2" in d["config"]["parallelism"]
    cfg = d["config"]
    assert cfg["rays_per_step_global"] == (4096 if scaling == "strong" else 8192)
    assert cfg["rays_per_step_per_gpu"] * 2 == cfg["rays_per_step_global"]
    assert cfg["collectives"]["world_size"] == 2 and cfg["collectives"]["backend"] == "gloo"
    assert cfg["collectives"]["bytes_on_the_wire"]["reduce_scatter_plane_grad_bytes"] > 0
