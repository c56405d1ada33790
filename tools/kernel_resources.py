#!/usr/bin/env python3
"""Register / scratch / LDS footprint of every kernel of the library, from the code object metadata of a device-only compile:
    python tools/kernel_resources.py [extra hipcc flags...]
(waves per SIMD on gfx950: floor(512 / (vgpr + agpr)) capped at 8)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__ as g
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "api.s")
        subprocess.run([g.HIPCC, *flags, "--cuda-device-only", "-S", "-o", out, os.path.join(g.CSRC, "api.hip")], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    rows = []
    for block in meta.split("\n  - .agpr_count:")[1:]:
        block = ".agpr_count:" + block
        get = lambda key, block=block: re.search(r"\.%s:\s+(\S+)" % key, block)
        name = get("name").group(1)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows.append((demangled, int(get("vgpr_count").group(1)), int(get("agpr_count").group(1)), int(get("sgpr_count").group(1)),
                     int(get("vgpr_spill_count").group(1)), int(get("sgpr_spill_count").group(1)), int(get("private_segment_fixed_size").group(1))))
    print(f"{'kernel':58s} {'vgpr+agpr':>9s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'waves/SIMD':>10s}")
    for name, vgpr, agpr, sgpr, vs, ss, scratch in sorted(rows):
        print(f"{name[:58]:58s} {vgpr:9d} {agpr:5d} {sgpr:5d} {vs:6d} {ss:6d} {scratch:7d} {min(8, 512 // max(vgpr, 1)):10d}")


if __name__ == "__main__":
    main()
