#!/usr/bin/env python3
"""Table of hipcc's -Rpass-analysis=kernel-resource-usage remarks: kernel, VGPRs, SGPRs, scratch, spills, occupancy.
usage: rsrc_table.py file.rsrc [...]   (the Makefile writes one .rsrc per object)  --fail: exit 1 if any kernel has scratch"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = int(v)
    return rows


def diagnostics(path):
    """everything in a report that is NOT a resource-usage remark block: genuine -Wall diagnostics with their source-context and
    caret lines intact.  A remark block = the `remark: ... [-Rpass-analysis=kernel-resource-usage]` line plus the indented
    source-context lines (`  NN | ...`, `     | ^`) clang prints under it."""
    in_remark = False
    held = []                      # "In file included from ..." lines: they belong to whatever diagnostic follows them
    for line in open(path, errors="replace"):
        if "-Rpass-analysis=kernel-resource-usage" in line:
            in_remark = True
            held = []
            continue
        if in_remark and re.match(r"^\s*(\d+\s*)?\|", line):
            continue
        in_remark = False
        if line.startswith("In file included from"):
            held.append(line)
            continue
        sys.stdout.write("".join(held) + line)
        held = []


def main():
    if "--diagnostics" in sys.argv:
        for f in [a for a in sys.argv[1:] if not a.startswith("--")]:
            diagnostics(f)
        return
    fail = "--fail" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    bad = 0
    for f in files:
        rows = parse(f)
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*", "", n).replace("void ", "")
            spill = r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0)
            mark = " <-- SCRATCH" if spill else ""
            bad += 1 if spill else 0
            print(f"{n:60s} vgpr {r.get('VGPRs', 0):4d} agpr {r.get('AGPRs', 0):3d} sgpr {r.get('TotalSGPRs', 0):4d} scratch {r.get('ScratchSize', 0):4d} "
                  f"vspill {r.get('VGPRs Spill', 0):3d} sspill {r.get('SGPRs Spill', 0):3d} occ {r.get('Occupancy', 0)}{mark}")
    if fail and bad:
        print(f"{bad} kernel instance(s) with scratch", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
