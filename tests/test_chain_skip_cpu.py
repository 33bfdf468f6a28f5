"""csrc/chain_skip.h -- the empty-cell skip of the march (`do { t += dt; } while (t < tt);`, raymarching.cu:393-398) computed
without walking its chain of dependent adds -- is plain C for host and device: compiled here with gcc (no FMA contraction,
as the kernels) and held against the literal loop, bit for bit, on millions of random (t, dt, tt)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chain_skip_equals_the_literal_loop(tmp_path):
    exe = str(tmp_path / "chain_skip_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "trinerflet_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "chain_skip_check.c"), "-lm"], check=True)
    r = subprocess.run([exe, "6000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "6000000 cases, 0 mismatches" in r.stdout, r.stdout[-2000:]


def test_the_kernels_use_it_behind_a_knob():
    src = open(os.path.join(ROOT, "trinerflet_amd", "csrc", "march_device.h")).read()
    assert "chain_skip_or_walk(t, m.dt0, tt)" in src and "#if TNL_CHAIN_JUMP" in src and "do { t += m.dt0; } while (t < tt);" in src
