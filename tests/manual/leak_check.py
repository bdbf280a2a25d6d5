#!/usr/bin/env python3
"""Device-memory leak check: contexts (plain and batched-view ones) created, used for a frame with every plane requested and
every growth path taken (a large mesh, many draws, a late plane request, a band, a raster mode switch), and destroyed, many
times over; meshes and textures created and destroyed; free device memory (hipMemGetInfo) before and after.
    python tests/manual/leak_check.py [cycles=60]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_bytes(hip):
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes
    from tinyrenderer_amd.model_renderer import MeshComponent, ModelRendererState, clear_screen, render_mesh_component, render_views

    hip = C.CDLL("libamdhip64.so.7")
    head = SU.component("head")
    fine = MeshComponent(scenes.subdivide(head.mesh, 3), head.texture, head.normal_map)  # 159 488 triangles: the large-draw buffers
    MASK = _ffi.TR_OUT_RGBA8 | _ffi.TR_OUT_Z | _ffi.TR_OUT_KEY | _ffi.TR_OUT_RGBA_F64 | _ffi.TR_OUT_TRI_ID

    def cycle(k):
        W, H = 320 + 16 * (k % 7), 200 + 16 * (k % 5)
        st = ModelRendererState(W, H, raster_mode=k % 3)
        clear_screen(st)
        for _ in range(1 + k % 4):
            render_mesh_component(st, head, scenes.shader_for(head, W, H))
        st.resolve(_ffi.TR_OUT_RGBA8)
        st.present()
        st.z_buffer  # (a late plane request)
        clear_screen(st)
        st.set_band(0, 64)
        render_mesh_component(st, fine, scenes.shader_for(fine, W, H))
        st.resolve(MASK)
        st.winner_keys(); st.triangle_ids(); st.frame_buffer
        st.close()  # (meshes and textures the state made go with it)
        sv = ModelRendererState(W, H, max_views=4)
        render_views(sv, head, [scenes.shader_for(head, W, H, eye=scenes.orbit_eye(i, 16)) for i in range(3)], _ffi.TR_OUT_RGBA8)
        sv.select_view(1)
        sv.present()
        sv.close()

    for k in range(3):  # (the runtime's own pools settle)
        cycle(k)
    before = free_bytes(hip)
    for k in range(cycles):
        cycle(k)
    after = free_bytes(hip)
    print(f"free device memory before {before} after {after}: {before - after:+d} bytes over {cycles} cycles")
    return 0 if before - after < (8 << 20) else 1


if __name__ == "__main__":
    sys.exit(main())
