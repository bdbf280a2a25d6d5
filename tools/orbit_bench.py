#!/usr/bin/env python3
"""BASELINE.json configs[4] rehearsal on one GPU: camera orbit of diablo3_pose at 1920x1080,
`n` views (eye_k = 2 (sin t, 0, cos t), t = 2 pi k / n), two contexts in flight; views per second.
    python tools/orbit_bench.py [n=512] [W=1920] [H=1080] [contexts=2]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    n_ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    import scene_util as SU
    from tinyrenderer_amd import _ffi, scenes
    from tinyrenderer_amd.model_renderer import ModelRendererState, make_draw_params

    lib = _ffi.load()
    mc = SU.component("diablo")
    states = [ModelRendererState(W, H) for _ in range(n_ctx)]
    # matrices are computed on the host once and passed as inputs
    shaders = [scenes.shader_for(mc, W, H, eye=scenes.orbit_eye(k, n)) for k in range(n)]
    params = [[make_draw_params(s, sh, mc.texture, mc.normal_map) for sh in shaders] for s in states]
    meshes = [s._mesh_handle(mc.mesh) for s in states]

    def run(count):
        for k in range(count):
            i = k % n_ctx
            cx = states[i]._ctx
            lib.tr_clear(cx)
            rc = lib.tr_draw(cx, meshes[i], C.byref(params[i][k % n]))
            assert rc == 0
            rc = lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8)
            assert rc == 0
        for s in states:
            s.synchronize()

    def issue(count):
        for k in range(count):
            i = k % n_ctx
            cx = states[i]._ctx
            lib.tr_clear(cx)
            lib.tr_draw(cx, meshes[i], C.byref(params[i][k % n]))
            lib.tr_resolve(cx, _ffi.TR_OUT_RGBA8)

    run(16)
    t0 = time.perf_counter()
    issue(n)
    t_issue = time.perf_counter() - t0
    for s in states:
        s.synchronize()
    print(f"host side: {t_issue / n * 1e6:.1f} us per view to issue clear + draw + resolve ({n / t_issue:.0f} views/s if the GPU were free)")
    t0 = time.perf_counter()
    run(n)
    dt = time.perf_counter() - t0
    c = states[0].counters()
    print(f"{n} views of diablo3_pose at {W}x{H}, {n_ctx} contexts: {n / dt:.0f} views/s ({dt / n * 1e6:.1f} us per view); "
          f"last view on context 0: {c['fragments']} fragments, {c['covered_pixels']} covered pixels, flags {states[0].status_flags():#x}")
    for s in states:
        s.close()


if __name__ == "__main__":
    main()
