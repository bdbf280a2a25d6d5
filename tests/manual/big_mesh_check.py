#!/usr/bin/env python3
"""BASELINE.json configs[3] rehearsal on one GPU: african_head subdivided `levels` times
(2492 * 4**levels triangles), rendered whole at WxH, timed, and compared with the oracle.
    python tests/manual/big_mesh_check.py [levels=5] [W=1920] [H=1080] [mode=0] [oracle=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    with_oracle = (int(sys.argv[5]) if len(sys.argv) > 5 else 1) != 0
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes
    from tinyrenderer_amd.model_renderer import MeshComponent, ModelRendererState, clear_screen, render_mesh_component

    head = SU.component("head")
    t0 = time.time()
    fine = MeshComponent(scenes.subdivide(head.mesh, levels), head.texture, head.normal_map)
    fine.transform.set_rotation((0.025, 0.0, 0.0))
    print(f"{fine.mesh.triangles.shape[0]} triangles (subdivide {time.time() - t0:.1f} s)", flush=True)
    sh = scenes.shader_for(fine, W, H)
    st = ModelRendererState(W, H, raster_mode=mode)
    times = []
    for it in range(6):
        st.synchronize()
        t0 = time.perf_counter()
        clear_screen(st)
        render_mesh_component(st, fine, sh)
        st.resolve(_ffi.TR_OUT_RGBA8)
        st.synchronize()
        times.append(time.perf_counter() - t0)
        # (reading the flags settles buffer growth: in row mode an overflowing frame is otherwise
        # redone only at the next read-back, and the loop would time frames that drew nothing)
        flags = st.status_flags()
    print("frame ms:", [round(t * 1e3, 2) for t in times], "flags", hex(st.status_flags()), st.counters(), flush=True)
    key = st.winner_keys()
    present = st.present()
    st.close()
    if with_oracle:
        import oracle_ffi

        t0 = time.time()
        o = SU.oracle_render(oracle_ffi, W, H, [(fine, sh)])
        print(f"oracle {time.time() - t0:.1f} s; key mismatches {(o['key'] != key).sum()}, "
              f"presented pixels differing {(o['present'] != present).any(axis=2).sum()}")


if __name__ == "__main__":
    main()
