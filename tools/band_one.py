#!/usr/bin/env python3
"""One tile-row band of BASELINE.json configs[3] (african_head subdivided 6x at 7680x4320), rendered a few times:
for `rocprofv3 --kernel-trace --stats` of what a rank of the band-sharded job runs per frame.
    python tools/band_one.py [band=3] [n_bands=8] [levels=6] [frames=6] [y_lo y_hi]     (y_lo y_hi: these rows instead)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    a = [int(x) for x in sys.argv[1:]]
    band, n_bands, levels, frames = (a + [3, 8, 6, 6][len(a):])[:4]
    import big_cases as B
    from tinyrenderer_amd import _ffi, sharding
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    W, H, draws = B.head_subdivided(levels, 7680, 4320)
    st = ModelRendererState(W, H)
    y_lo, y_hi = sharding.band_rows(H, band, n_bands)
    if len(a) >= 6:
        y_lo, y_hi = a[4], a[5]
    times = []
    for _ in range(frames):
        st.synchronize()
        t0 = time.perf_counter()
        clear_screen(st)
        st.set_band(y_lo, y_hi)
        render_mesh_component(st, *draws[0])
        st.resolve(_ffi.TR_OUT_RGBA8)
        st.synchronize()
        times.append(time.perf_counter() - t0)
    print(f"band {band} of {n_bands} rows [{y_lo}, {y_hi}): ms", [round(t * 1e3, 2) for t in times], st.counters())
    st.close()


if __name__ == "__main__":
    main()
