"""bench.py's bookkeeping that needs no GPU: which kernels a section's counter traffic is summed over."""
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name


def test_every_profiled_kernel_of_a_section_is_owned_by_it():
    """Round 5's line carried top[2].traffic_over_algorithmic = 0.014: the field backward's traffic was summed over
    k_slab_reduce alone because "k_field_bwd_rows" was matched by equality against "k_field_bwd".  The kernel names of the
    committed rocprof statistics (a real base run) are the fixture: every one of them lands in the section it belongs to,
    and in no other."""
    import bench
    names = set()
    for f in glob.glob(os.path.join(ROOT, "profiles", "r0[56]*_bench_base_kernel_stats.csv")):
        with open(f) as fh:
            names |= {_short(r["Name"]) for r in csv.DictReader(fh) if "k_" in r["Name"]}
    assert {"k_field_bwd_rows", "k_idwt_bwd_walk", "k_to_texel_major_h", "k_adam_l1_live"} <= names
    want = {"k_field_bwd_rows": "field_bwd", "k_slab_reduce": "field_bwd", "k_field_fwd": "field_fwd",
            "k_idwt_bwd_walk": "idwt_adjoint", "k_idwt_bwd_pipe": "idwt_adjoint", "k_idwt_fwd_walk": "idwt_fwd",
            "k_idwt_fwd_pipe": "idwt_fwd", "k_to_texel_major_h": "idwt_fwd", "k_tile_accumulate": "plane_grad_binned",
            "k_adam_l1_live": "adam_coef", "k_adam_l1": "adam_coef", "k_adam_record": "adam_coef"}
    for kn in names:
        owners = [s for s in bench.SECTION_KERNELS if bench.section_owns(s, kn)]
        assert owners == ([want[kn]] if kn in want else []), (kn, owners)
    # the replay of the deferred pass is not part of the per-step coefficient pass
    assert not bench.section_owns("adam_coef", "k_adam_l1_catchup")
    # hidden 128's two-launch backward (templated name, csrc/field_bwd.hip) belongs to the same section
    assert bench.section_owns("field_bwd", _short("void (anonymous namespace)::k_field_bwd<48, 128, 1>(Args)"))
