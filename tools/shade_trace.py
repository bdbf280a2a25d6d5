#!/usr/bin/env python3
"""Per-wave timeline of k_shade (debugging aid).  Run on the GPU box:
    TINYRENDERER_HIP_DIAG=1 TR_SHADE_TRACE=gpurun_out/shade.trace python tools/shade_trace.py [W H]
renders a few frames of the bench scene and summarises the last shade launch: when waves with
covered pixels start and end, how many are in flight per SIMD, how the launch trails off."""
import os

os.environ.setdefault("TINYRENDERER_HIP_DIAG", "1")  # the trace switches exist in the diagnostic build only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    path = os.environ.get("TR_SHADE_TRACE")
    assert path, "set TR_SHADE_TRACE"
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    import bench
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    draws = bench.build_scene(W, H)
    st = ModelRendererState(W, H)
    for _ in range(4):
        clear_screen(st)
        for mc, sh in draws:
            render_mesh_component(st, mc, sh)
        st.resolve()
    st.synchronize()
    st.close()
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t = t[t[:, 1] != 0]
    t0 = t[:, 0].min()
    b, e = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01  # us (100 MHz counter)
    cov = t[:, 2].astype(np.int64)
    hw = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (t[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    skipped = cov >= 0xFFFFFFFE
    work = (~skipped) & (cov > 0)
    print(f"{len(t)} waves traced; skipped-block waves {skipped.sum()}, waves without a covered pixel {((~skipped) & (cov == 0)).sum()}, "
          f"waves with covered pixels {work.sum()} (mean {cov[work].mean():.1f} px)")
    print(f"launch span {e.max():.1f} us" + (f"; skipped waves: last begins at {b[skipped].max():.1f} us" if skipped.any() else ""))
    d = e[work] - b[work]
    print(f"covered waves: life us mean {d.mean():.2f} p10 {np.percentile(d, 10):.2f} p50 {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}")
    print(f"   begin p1 {np.percentile(b[work], 1):.1f} p50 {np.median(b[work]):.1f} p99 {np.percentile(b[work], 99):.1f}; "
          f"end p50 {np.median(e[work]):.1f} p90 {np.percentile(e[work], 90):.1f} p99 {np.percentile(e[work], 99):.1f} max {e[work].max():.1f}")
    # covered waves in flight over time
    grid = np.arange(0.0, e.max() + 1.0, 2.0)
    infl = [(int(((b[work] <= g) & (e[work] > g)).sum())) for g in grid]
    print("   covered waves in flight every 2 us:", infl)
    # per-SIMD load: hw id bits (gfx9): simd [5:4], cu [11:8], sh [12], se [15:13]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh_ = (hw >> 12) & 1
    se = (hw >> 13) & 7
    unit = ((xcc * 8 + se) * 2 + sh_) * 16 + cu
    busy = {}
    for u, s_, dd in zip(unit[work], simd[work], d):
        busy[(u, s_)] = busy.get((u, s_), 0.0) + dd
    v = np.array(list(busy.values()))
    cnt = {}
    for u, s_ in zip(unit[work], simd[work]):
        cnt[(u, s_)] = cnt.get((u, s_), 0) + 1
    c = np.array(list(cnt.values()))
    print(f"   SIMDs that ran covered waves: {len(v)}; covered waves per SIMD: mean {c.mean():.1f} min {c.min()} p50 {np.median(c):.0f} max {c.max()}")
    print(f"   distinct CUs seen: {len(set(unit.tolist()))}")


if __name__ == "__main__":
    main()
