"""The oracle against the one output artefact the reference holds: sample.jpg (README.md:5), a
2160x3840 screenshot of its window showing african_head + inner eyes, frontal.  It is lossy and
its scene parameters are not recorded, so it cannot pin bits; it does pin everything gross --
projection, winding / z order, texture orientation, tangent-space normal mapping, lighting: the
oracle's frame at the screenshot's aspect (camera and perspective of src/main.rs, rotation 0)
correlates with it at 0.99+, and visibly wrong variants do not."""
import os

import numpy as np
import pytest

import scene_util as SU
from tinyrenderer_amd import scenes

PIL = pytest.importorskip("PIL.Image")
HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 270, 480


def _ncc(a, b):
    a = a - a.mean()
    b = b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def _gray(rgb):
    return np.asarray(PIL.fromarray(np.ascontiguousarray(rgb[..., :3])).convert("L"), dtype=np.float64)


def test_oracle_frame_matches_reference_screenshot(oracle):
    ref_rgb = np.asarray(PIL.open(os.path.join(HERE, "golden", "reference_sample_270x480.png")).convert("RGB"))
    ref = _gray(ref_rgb)
    head, eye = SU.component("head"), SU.component("eye_inner")
    eye.parent = head
    head.transform.set_rotation((0.0, 0.0, 0.0))
    draws = [(head, scenes.shader_for(head, W, H)), (eye, scenes.shader_for(eye, W, H))]
    frame = SU.oracle_render(oracle, W, H, draws)["present"]
    ncc = _ncc(_gray(frame), ref)
    assert ncc > 0.99, ncc
    # colour too, channel by channel, and the silhouette
    for c in range(3):
        assert _ncc(frame[..., c].astype(np.float64), ref_rgb[..., c].astype(np.float64)) > 0.985
    cover_ref = ref > 12
    cover = frame[..., :3].max(axis=2) > 12
    assert (cover_ref ^ cover).mean() < 0.01
    # sensitivity: wrong variants of the same scene fall well below
    assert _ncc(_gray(frame[::-1]), ref) < 0.8                      # upside down
    no_nm = [(mc, scenes.shader_for(mc, W, H, use_normal_map=False)) for mc, _ in draws]
    assert _ncc(_gray(SU.oracle_render(oracle, W, H, no_nm)["present"]), ref) < ncc  # no normal map
    head.transform.set_rotation((0.25, 0.0, 0.0))
    turned = [(head, scenes.shader_for(head, W, H)), (eye, scenes.shader_for(eye, W, H))]
    assert _ncc(_gray(SU.oracle_render(oracle, W, H, turned)["present"]), ref) < 0.9
    head.transform.set_rotation((0.025, 0.0, 0.0))
    eye.parent = None
