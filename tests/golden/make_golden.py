#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/rt_oracle.c).

The reference ships no golden data for this path and cannot be run here, so these fixtures are
outputs of the oracle, frozen so that (a) later edits to the oracle are caught and (b) GPU tests can
check the kernels against data that does not depend on the oracle library being rebuilt.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    O.build()
    # one set per arithmetic profile: NAME.npz (strict) and NAME_fma.npz (fused multiply-adds at fixed sites)
    for profile in ("strict", "fma"):
        for name, case in cases.CASES.items():
            r = cases.run_oracle(case, profile=profile)
            path = cases.golden_path(name, profile)
            np.savez_compressed(
                path,
                accum=r["accum"], rgba8=r["rgba8"], prim_id=r["prim_id"], chain_hash=r["chain_hash"],
                rng_state=r["rng_state"],
                counter_values=np.array([r["counters"][k] for k in O.COUNTER_NAMES], dtype=np.uint64))
            print(profile, name, os.path.getsize(path), "bytes", r["counters"])


if __name__ == "__main__":
    main()
