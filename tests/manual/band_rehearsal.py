#!/usr/bin/env python3
"""BASELINE.json configs[3] rehearsed on ONE GPU: african_head subdivided `levels` times at WxH, the
frame cut into the tile-row bands that 1/2/4/8 GPUs would own (sharding.band_rows, tr_set_band).
Every band is rendered and timed by itself on this GPU; the slowest band of a cut is what a node of
that many GPUs would take per frame before the RGBA8 gather.  The bands are stitched and compared
with the whole frame (winner keys), and the whole frame with the oracle if asked.
    python tests/manual/band_rehearsal.py [levels=6] [W=7680] [H=4320] [oracle=0] [rounds=8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 7680
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 4320
    with_oracle = (int(sys.argv[4]) if len(sys.argv) > 4 else 0) != 0
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes, sharding
    from tinyrenderer_amd.model_renderer import MeshComponent, ModelRendererState, clear_screen, render_mesh_component

    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, levels), head.texture, head.normal_map)
    fine.transform.set_rotation((0.025, 0.0, 0.0))
    sh = scenes.shader_for(fine, W, H)
    print(f"{fine.mesh.triangles.shape[0]} triangles at {W}x{H}", flush=True)
    st = ModelRendererState(W, H, raster_mode=0)

    def run(band, reps=5):
        ts = []
        for _ in range(reps):
            st.synchronize()
            t0 = time.perf_counter()
            clear_screen(st)
            if band is not None:
                st.set_band(*band)
            render_mesh_component(st, fine, sh)
            st.resolve(_ffi.TR_OUT_RGBA8)
            st.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[1:])) * 1e3

    t_full = run(None)
    key_full = st.winner_keys()
    print(f"whole frame: {t_full:.2f} ms", flush=True)
    # Rebalancing as a node would do it: every rank knows how long its band took last frame; the cost
    # per tile row is taken constant inside a band, and the next frame's cuts equalise it
    # (sharding.balanced_bands), damped; `rounds` frames starting from equal rows.
    n_rows = (H + sharding.TILE - 1) // sharding.TILE
    for world in (2, 4, 8):
        bands = [sharding.band_rows(H, r, world) for r in range(world)]
        for it in range(rounds):
            times, key = [], np.full_like(key_full, np.uint64(0xFFFFFFFFFFFFFFFF))
            for y_lo, y_hi in bands:
                times.append(run((y_lo, y_hi)))
                key[:, y_lo:y_hi] = st.winner_keys()[:, y_lo:y_hi]
            same = bool((key == key_full).all())
            print(f"{world} bands, round {it}: rows {[hi - lo for lo, hi in bands]} ms {[round(t, 2) for t in times]}; slowest "
                  f"{max(times):.2f} ms => {t_full / max(times):.2f}x the one-GPU frame rate; stitched keys equal: {same}", flush=True)
            assert same
            fixed = 0.5 * min(times)  # rough share of a band's time that does not depend on its rows
            row_w = [0.0] * n_rows
            for (y_lo, y_hi), t in zip(bands, times):
                k0, k1 = y_lo // sharding.TILE, (y_hi + sharding.TILE - 1) // sharding.TILE
                for k in range(k0, k1):
                    row_w[k] = max(t - fixed, 1e-3) / (k1 - k0)
            proposed = sharding.balanced_bands(row_w, world, height=H)
            # damped: move every cut half of the way (in whole tile rows)
            cuts = [0] + [((a[1] + b[1]) // 2) // sharding.TILE * sharding.TILE for a, b in zip(bands[:-1], proposed[:-1])] + [H]
            bands = list(zip(cuts[:-1], cuts[1:]))
    if with_oracle:
        import oracle_ffi

        clear_screen(st)
        st.set_band(0, H)  # (a band stays set until it is changed)
        render_mesh_component(st, fine, sh)
        o = SU.oracle_render(oracle_ffi, W, H, [(fine, sh)])
        print("whole frame vs oracle: key mismatches", int((o["key"] != st.winner_keys()).sum()))
    st.close()


if __name__ == "__main__":
    main()
