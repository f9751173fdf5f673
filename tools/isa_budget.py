#!/usr/bin/env python3
"""Instruction budget of a kernel from hipcc's assembly listing: counts VALU / SALU / LDS / VMEM / branch instructions in
line ranges of the listing (the phases are delimited by hand from landmarks: barriers, loop headers, the cell loads).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -x hip kcfx_kernels.hip -o k.s
    tools/isa_budget.py k.s screen_k31c_kernel [name:first:last ...]

Without ranges it prints the landmarks (loop headers, barriers, global loads / stores) to pick them from."""
import re
import sys


def klass(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm")):
        return "wait/nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return None


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines, on = [], False
    for ln in open(path):
        if re.match(rf"^_ZN\w*{kernel}\w*:", ln):
            on = True
        if on:
            lines.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    if len(sys.argv) == 3:
        for i, ln in enumerate(lines, 1):
            if re.search(r"Loop Header|s_barrier|global_load|global_store|s_endpgm", ln):
                print(i, ln.strip())
        return
    print("| phase | listing lines | VALU | SALU | LDS | VMEM | branch | wait/nop |")
    print("|---|---|---|---|---|---|---|---|")
    for spec in sys.argv[3:]:
        name, a, b = spec.rsplit(":", 2)
        cnt = {k: 0 for k in ("VALU", "SALU", "LDS", "VMEM", "branch", "wait/nop")}
        for ln in lines[int(a) - 1:int(b)]:
            m = re.match(r"^\s+([a-z_0-9]+)\b", ln)
            if m and klass(m.group(1)):
                cnt[klass(m.group(1))] += 1
        print(f"| {name} | {a}-{b} | " + " | ".join(str(cnt[k]) for k in ("VALU", "SALU", "LDS", "VMEM", "branch", "wait/nop")) + " |")


if __name__ == "__main__":
    main()
