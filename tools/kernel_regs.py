"""Register / LDS / scratch use of every kernel in an AMDGPU assembly file (hipcc -save-temps): parses the
amdhsa.kernels metadata.  python tools/kernel_regs.py file.s [substring]"""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = get("name")
    if flt in name:
        m = re.search(r"(k_[a-z0-9_]+)(I[^E]*E)?", name)
        print(f"{name[:90]:90s} agpr {blk.split()[0]:>4s} vgpr {get('vgpr_count'):>4s} sgpr {get('sgpr_count'):>4s} "
              f"spill {get('vgpr_spill_count'):>3s} scratch {get('private_segment_fixed_size'):>5s} lds {get('group_segment_fixed_size')}")
