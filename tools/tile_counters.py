#!/usr/bin/env python3
"""Diagnostic: work counters of the tile raster for one frame of a scene (GPU needed)."""
import os

os.environ.setdefault("TINYRENDERER_HIP_DIAG", "1")  # TR_TILE_DEBUG exists in the diagnostic build only
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scene_util as SU  # noqa: E402
from tinyrenderer_amd import scenes  # noqa: E402
from tinyrenderer_amd.model_renderer import ModelRendererState, clear_screen, render_mesh_component  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
name = sys.argv[3] if len(sys.argv) > 3 else "head"
mc = SU.component(name)
mc.transform.set_rotation((0.025, 0, 0))
st = ModelRendererState(W, H)
clear_screen(st)
render_mesh_component(st, mc, scenes.shader_for(mc, W, H))
st.resolve()
c = st.counters()
print(c)
if os.environ.get("TR_TILE_DEBUG") == "8":
    p = c["scanlines"]
    print("slowest tile: %.1f us, n_tri %d, tile %d (tx %d, ty %d); sum of tile times %.1f us over tiles" % (
        (p >> 40) / 100.0, (p >> 24) & 0xFFFF, p & 0xFFFFFF, (p & 0xFFFFFF) % ((W + 15) // 16), (p & 0xFFFFFF) // ((W + 15) // 16),
        c["triangles"] / 100.0))
