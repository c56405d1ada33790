#!/usr/bin/env python3
"""Register / scratch footprint of every kernel of the library AS SHIPPED: the code-object metadata inside vsrd_amd/lib/libvsrd_hip.so
(what the loader sees -- a callee's registers and the AGPRs it parks callee-saved ones in count towards its callers here, which a
one-kernel compile does not show: residual_step_pair_kernel<4> is 251 registers compiled alone and 256-260 in the library).
    python tools/kernel_resources.py [path/to/lib.so]          -> table on stdout (profiles/<round>/kernel_resources.txt)
(waves per SIMD on gfx950: floor(512 / unified registers), at most 8; kernels with an amdgpu_waves_per_eu attribute are listed as built)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_object_notes(library):
    """The notes of EVERY gfx950 code object in the library's fat binary (round 5: two translation units, api.hip and split_front.hip,
    one clang offload bundle each)."""
    with tempfile.TemporaryDirectory() as tmp:
        fatbin, device = os.path.join(tmp, "fatbin.bin"), os.path.join(tmp, "device.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", library, fatbin], check=True)
        data = open(fatbin, "rb").read()
        starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
        if not starts:
            raise SystemExit(f"{library}: no device code object found")
        notes = ""
        for begin, end in zip(starts, starts[1:] + [len(data)]):
            open(device, "wb").write(data[begin:end])
            out = subprocess.run([READELF, "--notes", device], capture_output=True, text=True)
            if out.returncode == 0:
                notes += out.stdout
        return notes


def main():
    library = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "vsrd_amd", "lib", "libvsrd_hip.so")
    notes = code_object_notes(library)
    rows = []
    for block in notes.split("  - .agpr_count:")[1:]:
        agpr = int(block.split()[0])
        get = lambda key, block=block: re.search(r"\.%s:\s+(\S+)" % key, block)
        name = get("name").group(1)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows.append((demangled, int(get("vgpr_count").group(1)), agpr, int(get("sgpr_count").group(1)), int(get("vgpr_spill_count").group(1)),
                     int(get("sgpr_spill_count").group(1)), int(get("private_segment_fixed_size").group(1))))
    print(f"# {os.path.relpath(library, ROOT)}: code-object metadata (llvm-readelf --notes)")
    print(f"{'kernel':62s} {'vgpr+agpr':>9s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'waves/SIMD':>10s}")
    for name, vgpr, agpr, sgpr, vs, ss, scratch in sorted(rows):
        print(f"{name[:62]:62s} {vgpr:9d} {agpr:5d} {sgpr:5d} {vs:6d} {ss:6d} {scratch:7d} {min(8, 512 // max(vgpr, 1)):10d}")


if __name__ == "__main__":
    main()
