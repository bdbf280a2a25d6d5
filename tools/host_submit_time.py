#!/usr/bin/env python3
"""How long the HOST takes to submit the frames bench.py times, beside how long the GPU takes to render them:
    python tools/host_submit_time.py [contexts=4] [frames=400] [W=3840] [H=2160]
prints the submit time per frame (the loop without the final wait) and the whole time per frame."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    a = [int(x) for x in sys.argv[1:]]
    n_ctx, frames, W, H = (a + [4, 400, 3840, 2160][len(a):])[:4]
    import bench
    from tinyrenderer_amd import _ffi
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    lib = _ffi.load()
    draws = bench.build_scene(W, H)
    states = [ModelRendererState(W, H) for _ in range(n_ctx)]
    prepared = [[(s._mesh_handle(mc.mesh), make_draw_params(s, sh, mc.texture, mc.normal_map)) for mc, sh in draws] for s in states]

    def frame(k):
        cx = states[k]._ctx
        lib.tr_clear(cx)
        for mesh_h, params in prepared[k]:
            lib.tr_draw(cx, mesh_h, C.byref(params))
        lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8)

    def sync():
        for s in states:
            lib.tr_synchronize(s._ctx)

    for i in range(40):
        frame(i % n_ctx)
    sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(frames):
            frame(i % n_ctx)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        print(f"contexts {n_ctx}: submit {1e6 * (t1 - t0) / frames:.1f} us/frame, whole {1e6 * (t2 - t0) / frames:.1f} us/frame "
              f"({frames / (t2 - t0):.0f} frames/s)")


if __name__ == "__main__":
    main()
