#!/usr/bin/env python3
"""Fuzz of the band path for LARGE draws (>= 65 536 triangles: cluster boxes, k_cull_clusters, deferred shade records,
k_small / k_mid): african_head subdivided 3x (159 488 triangles) under random rotations, positions, scales and camera
distances, the frame cut into 2-6 random bands; the stitched winner keys, depth and presented image must equal the whole
frame rendered by one context (itself checked against the oracle on every 4th scene).
    python tests/manual/fuzz_bands_big.py [n=20] [seed0=14000] [levels=3]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 14000
    levels = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import oracle_ffi
    import scene_util as SU
    from tinyrenderer_amd import scenes
    from tinyrenderer_amd.model_renderer import MeshComponent

    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, levels), head.texture, head.normal_map)
    bad = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        W = int(rng.integers(200, 1400)) // 4 * 4
        H = int(rng.integers(200, 1000)) // 2 * 2
        fine.transform.set_rotation((float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.6, 0.6))))
        fine.transform.set_position((float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.5, 0.5))))
        near = os.environ.get("FUZZ_NEAR", "0") != "0"  # (the camera inside or just in front of the mesh: corners behind the camera plane, w <= 0)
        eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.6, 1.3) if near else rng.uniform(1.6, 3.0)))
        draws = [(fine, scenes.shader_for(fine, W, H, eye=eye))]
        try:
            full = SU.gpu_render(W, H, draws)
            assert full["flags"] == 0, hex(full["flags"])
            if k % 4 == 0:
                SU.assert_parity(SU.oracle_render(oracle_ffi, W, H, draws), full)
            rows = (H + 15) // 16
            n_bands = int(rng.integers(2, 7))
            cuts = sorted(rng.choice(np.arange(1, rows), size=n_bands - 1, replace=False).tolist())
            edges = [0] + [min(c * 16, H) for c in cuts] + [H]
            z = np.full_like(full["z"], -np.inf)
            key = np.full_like(full["key"], np.uint64(0xFFFFFFFFFFFFFFFF))
            present = np.zeros_like(full["present"])
            frags = 0
            for y_lo, y_hi in zip(edges[:-1], edges[1:]):
                g = SU.gpu_render(W, H, draws, band=(y_lo, y_hi))
                assert g["flags"] == 0, hex(g["flags"])
                z[:, y_lo:y_hi] = g["z"][:, y_lo:y_hi]
                key[:, y_lo:y_hi] = g["key"][:, y_lo:y_hi]
                present[H - y_hi: H - y_lo] = g["present_fused"][H - y_hi: H - y_lo]
                frags += g["counters"]["fragments"]
            assert np.array_equal(z.view(np.uint64), full["z"].view(np.uint64)), "depth"
            assert np.array_equal(key, full["key"]), f"{(key != full['key']).sum()} winner keys differ"
            assert np.array_equal(present, full["present_fused"]), "presented image"
            assert frags == full["counters"]["fragments"], (frags, full["counters"]["fragments"])
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed {seed0 + k} {W}x{H}: {str(e)[:300]}", flush=True)
        if k % 5 == 4:
            print(f"{k + 1} scenes done, {bad} failures", flush=True)
    print(f"fuzz_bands_big: {n} scenes ({fine.mesh.triangles.shape[0]} triangles), {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
