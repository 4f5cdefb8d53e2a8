#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py: HBM-side bytes per launch
of every conv kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the counters are in KB; FETCH_SIZE reports half the bytes of
16-byte-per-lane reads on gfx950, /opt/skills/guides/MI355X_MICROARCH.md, HBM section), stamped with the hash of the
kernel sources so that bench.py only quotes it for the build it was measured on.
    python tools/pmc_traffic.py <FETCH csv glob> <WRITE csv glob> <out.json>"""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha  # noqa: E402


def norm(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name


# every kernel family of the library that moves tensors on the timed paths (round 6: pool_finish_z_kernel was missing)
KERNELS = ("conv", "norm", "unc_reduce", "pool_finish", "affine_gather", "fuse_sum", "bilinear", "bn_finalize", "tta_views")


def mean_per_kernel(pat, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
                acc[norm(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_comment": "HBM-side bytes per launch at the bench workload: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes, "
                   "mean over the launches of each kernel instance; valid for the build whose kernel sources hash to src_sha",
       "src_sha": source_sha(), "kernels": {k: int((2 * fetch[k] + write.get(k, 0.0)) * 1024) for k in sorted(fetch)}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
