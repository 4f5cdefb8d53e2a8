#!/usr/bin/env python3
"""Randomised shapes through the streaming kernels (transposed conv, InstanceNorm apply / pool / concat / fan-out) by
calling the parity tests' bodies with random parameters.      python tools/fuzz_misc.py [cases] [seed]"""
import os, random, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_num_threads(16)
from tests import test_gpu_kernels as T
from values_amd._lib import VxError

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = unsupported = 0
for case in range(cases):
    kind = rng.choice(["convT", "norm", "norm", "c1"])
    try:
        if kind == "convT":
            cin = rng.choice([16, 32, 64, 128, 8, 24, 48])
            cout = rng.choice([8, 16, 32, 64, 24, 40])   # multiples of 8 (others: packed_floats() = -1)
            shape = (rng.randint(1, 3), rng.randint(1, 5), rng.randint(1, 9), rng.randint(1, 17))
            tag = f"convT {cin}->{cout} {shape}"
            T.test_convT_matches_oracle(cin, cout, shape)
        elif kind == "norm":
            c = rng.choice([4, 8, 16, 32, 64, 128, 256, 12])
            pool = rng.random() < 0.5
            if pool:
                shape = (rng.randint(1, 3), 2 * rng.randint(1, 4), 2 * rng.randint(1, 6), 2 * rng.randint(1, 10))
            else:
                shape = (rng.randint(1, 3), rng.randint(1, 7), rng.randint(1, 11), rng.randint(1, 21))
                if shape[1] * shape[2] * shape[3] == 1:      # one element per (sample, channel): torch's instance_norm (the oracle) refuses it,
                    shape = shape[:3] + (2,)                 # as the reference's InstanceNorm3d does at 1^3; seed 23, case 142
            tag = f"norm C={c} {shape} pool={pool}"
            T.test_instnorm_lrelu_drop_pool_matches_oracle(c, shape, pool)
        else:
            cout = rng.choice([8, 16, 32])
            shape = (rng.randint(1, 3), rng.randint(1, 9), rng.randint(1, 20), rng.randint(1, 40))
            flip = rng.randint(0, 7)
            tag = f"c1 ->{cout} {shape} flip={flip}"
            T.test_conv3d_c1_matches_oracle(cout, shape, flip)
    except VxError as e:
        unsupported += 1
        print(f"rejected ({tag}): {e}")
    except Exception:
        bad += 1
        print(f"FAIL case {case}: {tag}")
        traceback.print_exc(limit=2)
print(f"{cases} cases, {bad} failures, {unsupported} rejected loudly")
sys.exit(1 if bad else 0)
