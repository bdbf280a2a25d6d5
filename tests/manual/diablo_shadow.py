#!/usr/bin/env python3
"""BASELINE.json configs[1]: diablo3_pose with normal + specular maps and a shadow-map pass at
1024x1024 on one GPU.  The Phong / specular / shadow shaders are EXTENSIONS (the reference has none):
the frame is compared with the build's own CPU statement of them (oracle/), visibility bit for bit,
colour within 1 LSB.      python tests/manual/diablo_shadow.py [W=1024] [H=1024] [map=1024] [out.png]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    out_png = sys.argv[4] if len(sys.argv) > 4 else None
    import scene_util as SU
    import oracle_ffi
    from tinyrenderer_amd import _ffi, helpers, scenes
    from tinyrenderer_amd.model_renderer import (ModelRendererState, ShadeExt, ShadowMap, clear_screen,
                                                 render_mesh_component, shadow_from_view)

    mc = SU.component("diablo")
    spec = scenes.load_specular("diablo")
    light_eye = (1.2, 0.9, 1.6)
    sh = scenes.shader_for(mc, W, H)
    sh.camera_pos = light_eye  # the shader's light direction (main.rs:249-251 uses the camera's position)
    light_sh = scenes.shader_for(mc, M, M, eye=light_eye)
    m = shadow_from_view(sh, W, H, light_sh, M, M)
    light, cam = ModelRendererState(M, M), ModelRendererState(W, H)

    state = {"sm": None}

    def frame(keep=False, reuse=True):
        clear_screen(light)
        render_mesh_component(light, mc, light_sh)  # (not resolved here: the snapshot resolves for depth only)
        old = None
        if state["sm"] is not None and reuse and not keep:
            state["sm"].update(light)  # no allocation, no host wait
        else:
            old = state["sm"]
            state["sm"] = ShadowMap(light, keep_host_copy=keep)
        sm = state["sm"]
        ext = ShadeExt(model=_ffi.TR_SHADE_PHONG, specular_map=spec, specular_strength=0.6, shadow=sm,
                       shadow_from_view=m, shadow_bias=0.01, shadow_dim=0.3)
        clear_screen(cam)
        if old is not None:
            old.close()  # (only now: the camera context's last frame referred to it until the clear)
        render_mesh_component(cam, mc, sh, ext)
        cam.resolve(_ffi.TR_OUT_RGBA8)
        return sm, ext

    for label, reuse, sync_each in (("a new map per frame, host waits for every frame", False, True),
                                    ("one map updated in place, host waits for every frame", True, True),
                                    ("one map updated in place, 200 frames queued", True, False)):
        n = 12 if sync_each else 200
        for it in range(3):
            frame(reuse=reuse)
        cam.synchronize()
        t0 = time.perf_counter()
        for it in range(n):
            frame(reuse=reuse)
            if sync_each:
                cam.synchronize()
        cam.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{W}x{H}, shadow map {M}x{M}, light pass + camera pass, {label}: {dt * 1e3:.3f} ms per frame "
              f"({1.0 / dt:.0f} frames/s, one context each)", flush=True)
    sm, ext = frame(keep=True)
    key, present = cam.winner_keys(), cam.present()
    print("flags", hex(cam.status_flags()), cam.counters(), flush=True)
    t0 = time.time()
    o = SU.oracle_render(oracle_ffi, W, H, [(mc, sh, ext)])
    d = np.abs(o["present"].astype(int) - present.astype(int))
    print(f"CPU statement {time.time() - t0:.2f} s (camera pass only); winner-key mismatches {(o['key'] != key).sum()}, "
          f"colour max diff {d.max()} LSB, pixels differing {(d.max(axis=2) > 0).sum()}", flush=True)
    if out_png:
        helpers.write_rgba_img_to_file(out_png, present)
    light.close()
    cam.close()
    state["sm"].close()  # (after the contexts that drew with it)


if __name__ == "__main__":
    main()
