#!/usr/bin/env python3
"""SGPR-spill traffic of kernel instances, by WHERE it executes: hipcc's resource report counts spilled SGPRs per kernel, not how
often their v_readlane / v_writelane run.  This compiles one .hip file of the product library to assembly (same flags as the
Makefile) and lists, per kernel instance whose mangled name matches a regex, the spill instructions inside loop blocks together with
the branch that guards their block -- a v_readlane in a once-per-column branch is not vector issue in the item loop.

    python tools/loop_spills.py conv3d_xp8w.hip 'ILi2ELi1ELi1ELi2ELi8|ILi1ELi2ELi0ELi0ELi4ELi0|ILi1ELi4ELi2ELi0ELi4ELi0'"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
csrc = os.path.join(ROOT, "values_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", out],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
name, body = None, []
kernels = {}
for l in lines:
    m = re.match(r"^(_Z\w+):\s", l)
    if m:
        name, body = m.group(1), []
        kernels[name] = body
    elif name:
        body.append(l)
        if l.startswith(".Lfunc_end"):
            name = None
for k, body in kernels.items():
    if not pat.search(k):
        continue
    in_loop, guard, rows, total = False, "", [], 0
    for i, l in enumerate(body):
        if re.match(r"^\.LBB\d+_\d+:", l):
            in_loop = "Loop" in l
        if "s_cbranch" in l:
            guard = l.strip()
        if "v_readlane" in l or "v_writelane" in l:
            total += 1
            # a lane select in an SGPR is the statistics' cross-lane read, not a spill slot
            if in_loop and not re.search(r",\s*s\d+\s*$", l.strip()):
                rows.append((i, l.strip(), guard))
    print(f"{k}: {total} v_readlane / v_writelane in the kernel, {len(rows)} inside loop blocks")
    for i, l, gd in rows:
        print(f"    line {i:5d}  {l:44s} behind  {gd}")
