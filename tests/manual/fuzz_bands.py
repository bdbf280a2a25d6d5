#!/usr/bin/env python3
"""Fuzz of the band path (tr_set_band, multi-GPU sharding of one frame): random triangle soups
(tests/test_gpu_parity._random_scene) of random sizes, cut into 2-5 random bands of whole 16-pixel tile rows, every band
rendered by itself in every raster mode; the stitched depth / winner keys / presented image must equal the ORACLE's whole
frame, the fragments must add up, and nothing may land outside a band.  python tests/manual/fuzz_bands.py [n=100] [seed0=5000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T

    bad = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        n_tris = int(rng.choice([3, 17, 60, 250, 900]))
        W = int(rng.integers(8, 500))
        H = int(rng.integers(33, 400))
        draws = T._random_scene(rng, n_tris, W, H, spread=float(rng.choice([0.6, 1.0, 1.3, 2.5])))
        if rng.random() < 0.3:
            draws = draws + T._random_scene(rng, max(1, n_tris // 3), W, H)
        o = SU.oracle_render(oracle_ffi, W, H, draws)
        rows = (H + 15) // 16
        n_bands = int(rng.integers(2, min(5, rows) + 1))
        cuts = sorted(rng.choice(np.arange(1, rows), size=n_bands - 1, replace=False).tolist()) if rows > 1 else []
        edges = [0] + [min(c * 16, H) for c in cuts] + [H]
        for mode in (0, 1, 2):
            z = np.full_like(o["z"], -np.inf)
            key = np.full_like(o["key"], np.uint64(0xFFFFFFFFFFFFFFFF))
            present = np.zeros_like(o["present"])
            fused = np.zeros_like(o["present"])
            frags = 0
            try:
                for y_lo, y_hi in zip(edges[:-1], edges[1:]):
                    if y_hi <= y_lo:
                        continue
                    g = SU.gpu_render(W, H, draws, mode=mode, band=(y_lo, y_hi))
                    assert g["flags"] == 0, hex(g["flags"])
                    z[:, y_lo:y_hi] = g["z"][:, y_lo:y_hi]
                    key[:, y_lo:y_hi] = g["key"][:, y_lo:y_hi]
                    present[H - y_hi: H - y_lo] = g["present"][H - y_hi: H - y_lo]
                    fused[H - y_hi: H - y_lo] = g["present_fused"][H - y_hi: H - y_lo]
                    frags += g["counters"]["fragments"]
                    outside = np.ones(H, dtype=bool)
                    outside[y_lo:y_hi] = False
                    assert np.isneginf(g["z"][:, outside]).all(), "a sample landed outside its band"
                same = (z.view(np.uint64) == o["z"].view(np.uint64)) | (np.isnan(z) & np.isnan(o["z"]))
                assert same.all(), f"{(~same).sum()} depth mismatches"
                assert np.array_equal(key, o["key"]), "winner keys"
                assert np.array_equal(present, o["present"]), "presented image (exact colour)"
                assert np.abs(fused.astype(np.int16) - o["present"].astype(np.int16)).max() <= 1, "fused colour"
                assert frags == o["counters"]["fragments"], (frags, o["counters"]["fragments"])
            except AssertionError as e:
                bad += 1
                print(f"FAIL seed {seed0 + k} mode {mode} {W}x{H} tris {n_tris} bands {edges}: {str(e)[:200]}", flush=True)
        if k % 20 == 19:
            print(f"{k + 1} scenes done, {bad} failures", flush=True)
    print(f"fuzz_bands: {n} scenes x 3 modes, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
