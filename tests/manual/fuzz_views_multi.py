#!/usr/bin/env python3
"""Fuzz of batches of views over SEVERAL meshes (tr_render_views_multi: the reference's loop draws two meshes per
iteration, src/main.rs:272-273): 2-3 random meshes under 2-5 random affine views, every slice against the oracle's frame of
the same draws in the same order -- depth, winner keys (tri_seq continues across the meshes and over the batch), presented
image, counters; then RGBA8-only batches over the same slices.  python tests/manual/fuzz_views_multi.py [n=60] [seed0=13000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 13000
    import big_cases as B
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T
    import test_views as V
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState, render_views_multi

    bad = skipped = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        W = int(rng.integers(8, 260))
        H = int(rng.integers(8, 200))
        n_meshes = int(rng.integers(2, 4))
        meshes = [T._random_scene(rng, int(rng.choice([1, 7, 40, 200, 600])), W, H, spread=float(rng.choice([0.6, 1.0, 1.3])))[0][0] for _ in range(n_meshes)]
        n_tris = [m.mesh.triangles.shape[0] for m in meshes]
        n_views = int(rng.integers(2, 6))
        shaders = []
        for _ in range(n_views):
            per_view = []
            for _m in range(n_meshes):
                ang = float(rng.uniform(-0.4, 0.4)) if rng.random() < 0.5 else 0.0
                c, s = np.cos(ang), np.sin(ang)
                mv = np.array([[c, -s, 0, rng.uniform(-0.3, 0.3)], [s, c, 0, rng.uniform(-0.3, 0.3)], [0, 0, 1, rng.uniform(-0.05, 0.05)], [0, 0, 0, 1.0]])
                per_view.append(BuiltinShader(np.eye(4), mv, (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 2.0)))
            shaders.append(per_view)
        try:
            oracles = [SU.oracle_render(oracle_ffi, W, H, list(zip(meshes, pv))) for pv in shaders]
        except AssertionError:
            skipped += 1  # (the reference would have panicked on a texture index)
            continue
        st = ModelRendererState(W, H, max_views=n_views)
        try:
            Tt = sum(n_tris)
            render_views_multi(st, meshes, shaders, V.MASK)
            assert st.status_flags() == 0, hex(st.status_flags())
            tot = [0, 0, 0, 0]
            for v, o in enumerate(oracles):
                st.select_view(v)
                z = st.z_buffer[..., 0]
                same = (o["z"].view(np.uint64) == z.view(np.uint64)) | (np.isnan(o["z"]) & np.isnan(z))
                assert same.all(), f"view {v}: {(~same).sum()} depth mismatches"
                assert np.array_equal(V._local_key(st.winner_keys(), v, Tt), o["key"]), f"view {v}: winner keys"
                assert np.array_equal(st.present(), o["present"]), f"view {v}: presented image (exact colour)"
                tot[0] += o["counters"]["samples"]; tot[1] += o["counters"]["fragments"]; tot[2] += o["counters"]["scanlines"]
                tot[3] += int((o["key"] != B.NONE_KEY).sum())
            c = st.counters()
            assert [c["samples"], c["fragments"], c["scanlines"], c["covered_pixels"]] == tot, (c, tot)
            for order in (list(range(n_views)), list(range(n_views))[::-1], [0]):
                render_views_multi(st, meshes, [shaders[i] for i in order], _ffi.TR_OUT_RGBA8)
                for v, i in enumerate(order):
                    st.select_view(v)
                    d = np.abs(st.present().astype(np.int16) - oracles[i]["present"].astype(np.int16))
                    assert d.max() <= 1, f"RGBA8-only batch, view {v}: presented colour off by {d.max()} LSB on {(d.max(axis=-1) > 1).sum()} pixels"
                assert st.status_flags() == 0
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed {seed0 + k} {W}x{H} meshes {n_tris} views {n_views}: {str(e)[:300]}", flush=True)
        finally:
            st.close()
        if k % 20 == 19:
            print(f"{k + 1} scenes done, {bad} failures", flush=True)
    print(f"fuzz_views_multi: {n} scenes ({skipped} skipped: the reference would have panicked), {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
