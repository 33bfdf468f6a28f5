"""The shell tools DESIGN.md / README.md cite must at least parse and do what their header says (a bad patch
left tools/knob_ci.sh recursing without ever building)."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shell_tools_parse():
    for f in glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_knob_ci_lists_every_knob_with_existing_tests_and_builds_with_the_flags():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "knob_ci.sh")], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, KNOB_CI_DRY="1"))
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if " | " in l]
    assert len(lines) >= 20, r.stdout
    for l in lines:
        flags, tests = l.split(" | ")
        assert flags.startswith("-DTNL_"), l
        files = [t.split("::")[0] for t in tests.split()]
        assert files and all(os.path.exists(os.path.join(ROOT, t)) for t in files), l
    # every knob named there exists in the sources (an #ifndef default in csrc/)
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "trinerflet_amd", "csrc", "*.h*")))
    for knob in set(re.findall(r"-D(TNL_[A-Z0-9_]+)", r.stdout)):
        assert re.search(r"#\s*if(ndef|def)?\b[^\n]*\b" + knob + r"\b", src), knob
    # run() hands its flags to the build (the line the bad patch lost)
    body = open(os.path.join(ROOT, "tools", "knob_ci.sh")).read()
    assert re.search(r'TNL_HIPCC_FLAGS="\$1" python -m trinerflet_amd\.build --force', body)


def test_python_tools_compile():
    """Every tools/*.py at least byte-compiles (their GPU runs are recorded under profiles/)."""
    import py_compile
    for f in glob.glob(os.path.join(ROOT, "tools", "*.py")):
        py_compile.compile(f, doraise=True)
