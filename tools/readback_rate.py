#!/usr/bin/env python3
"""PCIe-inclusive frame rate of the bench workload: every frame is also read back to the host
(tr_read_rgba8: device-to-host copy of W*H*4 bytes into pageable memory + the vertical flip), which
bench.py's `value` deliberately excludes (frames stay in HBM there).   python tools/readback_rate.py [W H]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    import bench
    from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component

    draws = bench.build_scene(W, H)
    st = ModelRendererState(W, H)
    n = 60
    for pinned in (False, True):
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(n):
                clear_screen(st)
                for mc, sh in draws:
                    render_mesh_component(st, mc, sh)
                st.resolve()
                img = st.present(pinned=pinned)
            dt = time.perf_counter() - t0
        print(f"{W}x{H}, {'page-locked' if pinned else 'pageable'} host buffer: {n / dt:.0f} frames/s with every frame read back "
              f"({dt / n * 1e3:.2f} ms per frame, {W * H * 4 / (dt / n) / 1e9:.1f} GB/s of RGBA8 over the link), "
              f"image checksum {int(img.sum(dtype='int64'))}")
    st.close()


if __name__ == "__main__":
    main()
