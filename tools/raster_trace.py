#!/usr/bin/env python3
"""Per-chunk timeline of k_raster (debugging aid).  Run on the GPU box:
    TINYRENDERER_HIP_DIAG=1 TR_RASTER_TRACE=gpurun_out/trace python tools/raster_trace.py [W H [head|diablo]]
renders a few frames of the bench scene (or of diablo3_pose), then summarises the last depth / winner launch."""
import os

os.environ.setdefault("TINYRENDERER_HIP_DIAG", "1")  # the trace switches exist in the diagnostic build only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def summarise(path):
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t = t[t[:, 2] != 0]
    if not len(t):
        print(path, "empty")
        return
    t0 = t[:, 0].min()
    b, s, e = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01  # us (100 MHz)
    n = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    blk = (t[:, 3] >> np.uint64(32)).astype(np.int64)
    print(f"== {path}: {len(t)} chunks, {n.sum()} samples, span {e.max():.1f} us")
    print(f"   record phase   us: mean {np.mean(s - b):.2f} p50 {np.median(s - b):.2f} p99 {np.percentile(s - b, 99):.2f} max {np.max(s - b):.2f}")
    print(f"   sample phase  us: mean {np.mean(e - s):.2f} p50 {np.median(e - s):.2f} p99 {np.percentile(e - s, 99):.2f} max {np.max(e - s):.2f}")
    recs = blk  # (row raster: records of the bin in the high word)
    it = np.maximum(1, (n + 255) // 256)
    print(f"   us per 256-sample iteration: mean {np.mean((e - s) / it):.3f} p50 {np.median((e - s) / it):.3f}")
    print(f"   chunk begin us: p50 {np.median(b):.1f} p90 {np.percentile(b, 90):.1f} max {b.max():.1f}; end: p50 {np.median(e):.1f} p90 {np.percentile(e, 90):.1f} p99 {np.percentile(e, 99):.1f}")
    late = np.argsort(e)[-8:]
    for k in late:
        print(f"   late chunk {k}: block {blk[k]} begin {b[k]:.1f} scan {s[k] - b[k]:.2f} samples {n[k]} sample-phase {e[k] - s[k]:.2f} end {e[k]:.1f}")
    hist, edges = np.histogram(e, bins=10)
    print("   end-time histogram:", [f"{edges[i]:.0f}-{edges[i+1]:.0f}:{hist[i]}" for i in range(10)])


def main():
    path = os.environ.get("TR_RASTER_TRACE")
    assert path, "set TR_RASTER_TRACE"
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    import bench
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    draws = bench.build_scene(W, H)
    if len(sys.argv) > 3 and sys.argv[3] == "diablo":
        import scene_util as SU
        from tinyrenderer_amd import scenes

        mc = SU.component("diablo")
        draws = [(mc, scenes.shader_for(mc, W, H))]
    for suffix in (".depth", ".winner", ".rows"):
        if os.path.exists(path + suffix):
            os.remove(path + suffix)
    st = ModelRendererState(W, H)
    for _ in range(4):
        clear_screen(st)
        for mc, sh in draws:
            render_mesh_component(st, mc, sh)
        st.resolve()
    st.synchronize()
    st.close()
    for suffix in (".depth", ".winner", ".rows"):
        if os.path.exists(path + suffix):
            summarise(path + suffix)


if __name__ == "__main__":
    main()
