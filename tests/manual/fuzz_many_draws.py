#!/usr/bin/env python3
"""Frames of MANY draws (the draw table, the per-draw constants of the shade pass, tri_seq across hundreds of draws):
50-600 draws of small random meshes at random places in one frame, all planes against the oracle, in every raster mode.
    python tests/manual/fuzz_many_draws.py [n=10] [seed0=16000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T
    from tinyrenderer_amd.model_renderer import BuiltinShader

    bad = skipped = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        W = int(rng.integers(64, 500))
        H = int(rng.integers(64, 400))
        pool = [T._random_scene(rng, int(rng.choice([1, 2, 5, 20])), W, H, spread=0.9)[0][0] for _ in range(6)]
        n_draws = int(rng.choice([50, 130, 300, 600]))
        draws = []
        for _ in range(n_draws):
            s = float(rng.uniform(0.05, 0.5))
            mv = np.array([[s, 0, 0, rng.uniform(-0.9, 0.9)], [0, s, 0, rng.uniform(-0.9, 0.9)], [0, 0, 1, rng.uniform(-0.3, 0.3)], [0, 0, 0, 1.0]])
            sh = BuiltinShader(np.eye(4), mv, (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 2.0),
                               use_normal_map=bool(rng.random() < 0.7), light_intensity=float(rng.uniform(0.2, 1.5)), ambient_light=float(rng.uniform(0, 30)))
            draws.append((pool[int(rng.integers(0, 6))], sh))
        try:
            o = SU.oracle_render(oracle_ffi, W, H, draws)
        except AssertionError:
            skipped += 1
            continue
        for mode in (0, 1, 2):
            try:
                g = SU.gpu_render(W, H, draws, mode=mode)
                info = SU.assert_parity(o, g)
                assert info["f64_colour_mismatches"] == 0
                assert g["counters"]["fragments"] == o["counters"]["fragments"]
            except AssertionError as e:
                bad += 1
                print(f"FAIL seed {seed0 + k} mode {mode} {W}x{H} draws {n_draws}: {str(e)[:300]}", flush=True)
        print(f"scene {k}: {n_draws} draws, {W}x{H} done, {bad} failures", flush=True)
    print(f"fuzz_many_draws: {n} scenes x 3 modes ({skipped} skipped: the reference would have panicked), {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
