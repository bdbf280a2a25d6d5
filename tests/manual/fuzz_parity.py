#!/usr/bin/env python3
"""Longer fuzz run than the test suite's: random triangle soups (tests/test_gpu_parity._random_scene)
of random sizes through every raster mode against the oracle.  python tests/manual/fuzz_parity.py [n=100] [seed0=1000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T

    bad = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        n_tris = int(rng.choice([1, 3, 17, 60, 250, 900, 3000]))
        W = int(rng.integers(1, 700))
        H = int(rng.integers(1, 500))
        spread = float(rng.choice([0.6, 1.0, 1.3, 2.5]))
        draws = T._random_scene(rng, n_tris, W, H, spread=spread)
        if rng.random() < 0.3:  # a second draw over the first
            draws = draws + T._random_scene(rng, max(1, n_tris // 3), W, H, spread=spread)
        o = SU.oracle_render(oracle_ffi, W, H, draws)
        for mode in (0, 1, 2):
            g = SU.gpu_render(W, H, draws, mode=mode)
            try:
                info = SU.assert_parity(o, g)
                assert info["f64_colour_mismatches"] == 0
                assert g["counters"]["fragments"] == o["counters"]["fragments"]
            except AssertionError as e:
                bad += 1
                print(f"FAIL seed {seed0 + k} mode {mode} {W}x{H} tris {n_tris}: {str(e)[:200]}", flush=True)
        if k % 20 == 19:
            print(f"{k + 1} scenes done, {bad} failures", flush=True)
    print(f"fuzz: {n} scenes x 3 modes, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
