#!/usr/bin/env python3
"""Fuzz of the batched-view path (tr_create_views / tr_render_views): random triangle soups
(tests/test_gpu_parity._random_scene, one mesh) under 2-6 random affine views per batch, every slice against the oracle's
frame of that view -- depth, winner keys, triangle ids, f64 colour, presented image, counters (tests/test_views._check_views).
    python tests/manual/fuzz_views.py [n=100] [seed0=7000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    import oracle_ffi
    import scene_util as SU
    import test_gpu_parity as T
    import test_views as V
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import BuiltinShader, ModelRendererState

    bad = 0
    skipped = 0
    flagged = 0
    for k in range(n):
        rng = np.random.default_rng(seed0 + k)
        n_tris = int(rng.choice([1, 3, 17, 60, 250, 900]))
        W = int(rng.integers(4, 300))
        H = int(rng.integers(4, 220))
        mc = T._random_scene(rng, n_tris, W, H, spread=float(rng.choice([0.6, 1.0, 1.3])))[0][0]
        n_views = int(rng.integers(2, 7))
        shaders = []
        for _ in range(n_views):
            # (half of the views unrotated: the scenes' exactly horizontal degenerate triangles then stay so, and their NaN
            # samples land on pixel (0, 0) of the view's slice)
            ang = float(rng.uniform(-0.4, 0.4)) if rng.random() < 0.5 else 0.0
            c, s = np.cos(ang), np.sin(ang)
            mv = np.array([[c, -s, 0, rng.uniform(-0.2, 0.2)], [s, c, 0, rng.uniform(-0.2, 0.2)], [0, 0, 1, rng.uniform(-0.05, 0.05)], [0, 0, 0, 1.0]])
            shaders.append(BuiltinShader(np.eye(4), mv, (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 2.0)))
        # (a sliver's barycentrics can carry u, v past the texture's edge: the reference panics there, helpers/mod.rs:255-270 --
        # not a scene to compare)
        panics = False
        for sh in shaders:
            try:
                SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])
            except AssertionError:
                panics = True
        st = ModelRendererState(W, H, max_views=n_views)
        try:
            if panics:
                skipped += 1
                V.render_views(st, mc, shaders, V.MASK)
                # (deferred shading: the library shades -- and so flags -- a pixel's FINAL winner only, the reference every sample
                # that passes its z test at the time; a panic on a sample that is overwritten later goes unflagged)
                flagged += 1 if st.status_flags() & 4 else 0
            else:
                V._check_views(oracle_ffi, st, mc, shaders, W, H)
                # the frame loop's way: RGBA8 only (boxed shade launch, self-cleaning planes), the batch and then its views in
                # reverse order over the same slices, so that every slice has to repaint what the other batch left
                for order in (shaders, shaders[::-1], shaders[: max(1, n_views // 2)]):
                    V.render_views(st, mc, order, _ffi.TR_OUT_RGBA8)
                    for v, sh in enumerate(order):
                        o = SU.oracle_render(oracle_ffi, W, H, [(mc, sh)])
                        st.select_view(v)
                        d = np.abs(st.present().astype(np.int16) - o["present"].astype(np.int16))
                        assert d.max() <= 1, f"RGBA8-only batch of {len(order)}, view {v}: presented colour off by {d.max()} LSB on {(d.max(axis=-1) > 1).sum()} pixels"
                    assert st.status_flags() == 0
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed {seed0 + k} {W}x{H} tris {n_tris} views {n_views}: {str(e)[:200]}", flush=True)
        finally:
            st.close()
        if k % 20 == 19:
            print(f"{k + 1} scenes done, {bad} failures", flush=True)
    print(f"fuzz_views: {n} scenes ({skipped} on which the reference would have panicked on a texture index and nothing is compared; TR_FLAG_TEX_OOB raised on {flagged} of them), {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
