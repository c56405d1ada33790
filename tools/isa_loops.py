#!/usr/bin/env python3
"""Static view of one kernel's ISA: basic blocks, backward branches (loops) and the instruction mix of every loop body.
    python tools/isa_loops.py <file.s> <substring of the mangled kernel name> [--dump LABEL]
(the .s comes from hipcc ... --cuda-device-only -S; see tools/kernel_resources.py)"""
import re
import sys
from collections import Counter


def kernel_lines(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section") or lines[i].strip().startswith(".amdhsa_kernel"))
    return lines[start + 1:end]


def classify(op):
    if op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub", "v_mac", "v_pk_")):
        return "fp32"
    if op.startswith(("v_rcp", "v_sqrt", "v_exp", "v_log", "v_rsq", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "cmp/sel"
    if op.startswith("v_"):
        return "valu-other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem:" + op.split("_")[0]
    return "?"


def main():
    path, key = sys.argv[1], sys.argv[2]
    body = kernel_lines(path, key)
    insts = []          # (index, label-or-None, op, text)
    labels = {}
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        insts.append((op, s.split(";")[0].strip()))
    if "--dump" in sys.argv:
        a, b = sys.argv[sys.argv.index("--dump") + 1].split(":")
        for i in range(int(a), int(b)):
            print(i, insts[i][1])
        return
    print(f"{len(insts)} instructions")
    total = Counter(classify(op) for op, _ in insts)
    print("whole kernel:", dict(total))
    loops = []
    for i, (op, text) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = text.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    loops.sort()
    for a, b, tgt in loops:
        mix = Counter(classify(op) for op, _ in insts[a:b + 1])
        valu = sum(v for k, v in mix.items() if k in ("fp32", "trans", "mov", "lane", "cmp/sel", "valu-other"))
        print(f"loop {tgt:12s} [{a:5d},{b:5d}] len={b - a + 1:4d} valu={valu:4d}  " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())))


if __name__ == "__main__":
    main()
