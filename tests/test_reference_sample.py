"""The oracle (and the GPU path) against the one output artefact the reference holds: sample.jpg (README.md:5,
"Result"), a 2160x3840 JPEG of its window showing african_head + inner eyes, frontal.  It is lossy, so it cannot
pin bits.  What it does pin:

* coarse (270x480, normalised correlation): projection, winding / z order, texture orientation, tangent-space
  normal mapping, lighting -- the oracle's frame correlates with it at 0.99+, visibly wrong variants do not;
* at FULL resolution (2160x3840, the literals of src/main.rs:62-70,110-142 with the head's rotation at 0, i.e. the
  scene at rest -- the angle was found by scanning it; the loop itself draws its first frame at 0.025 rad): over the 4.8 M covered pixels the oracle's presented image differs from the JPEG by 1.24 LSB
  on average (median 1, 99 % within 4 LSB, 99.7 % within 8 LSB on every channel -- the JPEG's own quantisation noise,
  with its chroma bias of +1.2 in R and B), the silhouettes disagree on 0.13 % of the frame, and every one of the
  following makes it measurably worse: the frame shifted by ONE pixel in any direction (1.43-1.46), the head turned by
  0.001 rad (1.40) or 0.002 rad (1.67), no normal map (4.2).  So the vertex stage, viewport transform, raster coverage,
  z order, the parent fold of the eye mesh, texture addressing and the lighting formula are pinned to a pixel and to
  JPEG noise by the reference's own output; what stays unpinned is the last bits (rounding / association)."""
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


# ---- full resolution -------------------------------------------------------------------------------------------
FW, FH = 2160, 3840


def _full_ref():
    return np.asarray(PIL.open(os.path.join(HERE, "golden", "reference_sample_2160x3840.jpg")).convert("RGB")).astype(np.int16)


def _scene(rot=0.0, use_normal_map=True):
    head, eye = SU.component("head"), SU.component("eye_inner")
    eye.parent = head
    head.transform.set_rotation((rot, 0.0, 0.0))
    draws = [(head, scenes.shader_for(head, FW, FH, use_normal_map=use_normal_map)),
             (eye, scenes.shader_for(eye, FW, FH, use_normal_map=use_normal_map))]
    return head, eye, draws


def _restore(head, eye):
    head.transform.set_rotation((0.025, 0.0, 0.0))
    eye.parent = None


def _check_against_screenshot(frame, ref, what):
    """`frame`: presented RGB (H, W, 3) int16.  Returns (mean |d| over the covered pixels, covered mask)."""
    cov_r, cov_f = ref.max(axis=2) > 12, frame.max(axis=2) > 12
    cov = cov_r | cov_f
    assert (cov_r ^ cov_f).mean() < 0.002, f"{what}: silhouettes disagree on {(cov_r ^ cov_f).mean():.4%} of the frame"
    d = np.abs(frame - ref)[cov]
    assert 4.5e6 < cov.sum() < 5.0e6
    mean = float(d.mean())
    assert mean < 1.5, f"{what}: mean |d| = {mean:.3f} LSB over the covered pixels"
    assert np.median(d) <= 1 and np.percentile(d, 99) <= 6, (what, np.median(d), np.percentile(d, 99))
    assert (d.max(axis=1) > 8).mean() < 0.005, f"{what}: {(d.max(axis=1) > 8).mean():.4%} of the covered pixels are off by more than 8 LSB"
    assert (d.max(axis=1) > 16).mean() < 0.001
    return mean, cov


def _err(frame, ref, cov):
    return float(np.abs(frame - ref)[cov].mean())


def test_oracle_frame_matches_reference_screenshot_at_full_resolution(oracle):
    ref = _full_ref()
    assert ref.shape == (FH, FW, 3)
    head, eye, draws = _scene(0.0)
    try:
        frame = SU.oracle_render(oracle, FW, FH, draws)["present"][..., :3].astype(np.int16)
        mean, cov = _check_against_screenshot(frame, ref, "oracle")
        # no bias, and none locally: the signed error of the green channel (the one a JPEG keeps best; R and B carry the
        # encoder's +1.2 / +1.3, what flooring Y, Cb and Cr by half a step each gives: -0.5 - 1.402 / 2 and -0.5 - 1.772 / 2
        # against +0.03 for G) averages 0.01 over the frame and stays within +-1 over every fully covered 48x48 block --
        # `as u8` truncates (helpers/mod.rs:209; rounding would show as +0.5 here), and no region of the head is shaded differently
        both = (ref.max(axis=2) > 12) & (frame.max(axis=2) > 12)
        dg = (frame[..., 1] - ref[..., 1]).astype(np.float64)
        assert abs(dg[both].mean()) < 0.15, dg[both].mean()
        B = 48
        hb, wb = FH // B, FW // B
        full = both[:hb * B, :wb * B].reshape(hb, B, wb, B).all(axis=(1, 3))
        bm = (dg * both)[:hb * B, :wb * B].reshape(hb, B, wb, B).mean(axis=(1, 3))[full]
        assert full.sum() > 1500 and np.abs(bm).max() < 1.0 and bm.std() < 0.35, (full.sum(), np.abs(bm).max(), bm.std())
        # sensitivity: one pixel in any direction, a thousandth of a radian, the normal map -- all measurably worse
        for axis, step in ((1, 1), (1, -1), (0, 1), (0, -1)):
            assert _err(np.roll(frame, step, axis=axis), ref, cov) > mean + 0.12, (axis, step)
        for rot, margin in ((0.001, 0.08), (-0.002, 0.25), (0.002, 0.25)):
            _, _, turned = _scene(rot)
            assert _err(SU.oracle_render(oracle, FW, FH, turned)["present"][..., :3].astype(np.int16), ref, cov) > mean + margin, rot
        _, _, flat = _scene(0.0, use_normal_map=False)
        assert _err(SU.oracle_render(oracle, FW, FH, flat)["present"][..., :3].astype(np.int16), ref, cov) > 3.0
    finally:
        _restore(head, eye)


@pytest.mark.gpu
def test_gpu_frame_matches_reference_screenshot_at_full_resolution():
    """The HIP path against the reference's own output directly (no oracle in between): both colour arithmetics."""
    ref = _full_ref()
    head, eye, draws = _scene(0.0)
    try:
        g = SU.gpu_render(FW, FH, draws)
        assert g["flags"] == 0
        m_exact, _ = _check_against_screenshot(g["present"][..., :3].astype(np.int16), ref, "GPU, exact colour")
        m_fused, _ = _check_against_screenshot(g["present_fused"][..., :3].astype(np.int16), ref, "GPU, fused colour")
        assert abs(m_exact - m_fused) < 0.01
    finally:
        _restore(head, eye)
