#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU through tr_render_views: camera orbit of diablo3_pose at 1920x1080,
`n` views in batches of `batch` views per call on `contexts` contexts in turn; views per second and the
per-stage times of a batch (hipEvents).
    python tools/orbit_batch_bench.py [n=512] [batch=32] [contexts=1] [W=1920] [H=1080]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    a = [int(x) for x in sys.argv[1:]]
    n, batch, n_ctx, W, H = (a + [512, 32, 1, 1920, 1080][len(a):])[:5]
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    lib = _ffi.load()
    mc = SU.component(os.environ.get("ORBIT_MESH", "diablo"))
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    states = [ModelRendererState(W, H, max_views=batch) for _ in range(n_ctx)]
    shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n)) for k in range(n)]
    params = []
    for s in states:
        arr = (_ffi.DrawParams * n)()
        for k, sh in enumerate(shaders):
            arr[k] = make_draw_params(s, sh, mc.texture, mc.normal_map)
        params.append(arr)
    meshes = [s._mesh_handle(mc.mesh) for s in states]
    stride = C.sizeof(_ffi.DrawParams)

    def run(count):
        done, i = 0, 0
        while done < count:
            v = min(batch, count - done)
            s = states[i % n_ctx]
            p = C.cast(C.byref(params[i % n_ctx], done * stride), C.POINTER(_ffi.DrawParams))
            rc = lib.tr_render_views(s._ctx, meshes[i % n_ctx], p, v, _ffi.TR_OUT_RGBA8)
            assert rc == 0, lib.tr_last_error(s._ctx)
            done += v
            i += 1
        for s in states:
            s.synchronize()

    run(min(n, 4 * batch * n_ctx))
    t0 = time.perf_counter()
    run(n)
    dt = time.perf_counter() - t0
    c = states[0].counters()
    print(f"{n} views of diablo3_pose at {W}x{H}, batches of {batch} on {n_ctx} context(s): {n / dt:.0f} views/s "
          f"({dt / n * 1e6:.1f} us per view); last batch on context 0: {c['fragments']} fragments, "
          f"{c['covered_pixels']} covered pixels, flags {states[0].status_flags():#x}")
    st = states[0]
    lib.tr_profile_enable(st._ctx, 1)
    ms, cnt = (C.c_double * 6)(), (C.c_uint64 * 6)()
    lib.tr_profile_read(st._ctx, ms, cnt, 1)
    for _ in range(5):
        lib.tr_render_views(st._ctx, meshes[0], params[0], batch, _ffi.TR_OUT_RGBA8)
    st.synchronize()
    lib.tr_profile_read(st._ctx, ms, cnt, 1)
    names = ["clear", "transform", "setup_scan", "depth_pass", "winner_pass", "shade"]
    print("per batch of", batch, "views (us):", {nm: round(ms[i] / max(1, cnt[i]) * 1e3, 1) for i, nm in enumerate(names)},
          "=> per view", round(sum(ms[i] / max(1, cnt[i]) for i in range(6)) * 1e3 / batch, 1), "us")
    for s in states:
        s.close()


if __name__ == "__main__":
    main()
