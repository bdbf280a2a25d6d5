#!/usr/bin/env python3
"""The oracle's frame against the reference's own output, sample.jpg (tests/golden/reference_sample_2160x3840.jpg), at full
resolution: scans the head's rotation (the one parameter the screenshot does not record) and prints, per angle, the mean
absolute difference over the covered pixels, its percentiles and the share of the frame on which the silhouettes disagree;
then one-pixel shifts, the normal map left out and the vertical flip at the best angle.  CPU only (oracle/, test
infrastructure).  The numbers quoted in DESIGN.md 5 and asserted by tests/test_reference_sample.py come from here.
    python tests/manual/sample_scan.py [angle ...]        (default: -0.005 -0.002 -0.001 0 0.001 0.002 0.005 0.0125 0.025)"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi  # noqa: E402
import scene_util as SU  # noqa: E402
from tinyrenderer_amd import scenes  # noqa: E402

ref = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "reference_sample_2160x3840.jpg")).convert("RGB")).astype(np.int16)
H_, W_ = ref.shape[:2]


def render(rot, use_nm=True):
    head, eye = SU.component("head"), SU.component("eye_inner")
    eye.parent = head
    head.transform.set_rotation((rot, 0.0, 0.0))
    draws = [(head, scenes.shader_for(head, W_, H_, use_normal_map=use_nm)), (eye, scenes.shader_for(eye, W_, H_, use_normal_map=use_nm))]
    f = SU.oracle_render(oracle_ffi, W_, H_, draws)["present"][..., :3].astype(np.int16)
    eye.parent = None
    return f


def stats(f):
    cov_r, cov_f = ref.max(axis=2) > 12, f.max(axis=2) > 12
    cov = cov_r | cov_f
    d = np.abs(f - ref)[cov]
    return cov, float(d.mean()), float(np.median(d)), float(np.percentile(d, 99)), float((d.max(axis=1) > 8).mean()), float((cov_r ^ cov_f).mean())


angles = [float(x) for x in sys.argv[1:]] or [-0.005, -0.002, -0.001, 0.0, 0.001, 0.002, 0.005, 0.0125, 0.025]
best = None
for rot in angles:
    f = render(rot)
    cov, mean, med, p99, far, xor = stats(f)
    print(f"rotation {rot:+.4f} rad: mean |d| {mean:.3f} LSB over {int(cov.sum())} covered pixels, median {med:.0f}, p99 {p99:.0f}, "
          f"> 8 LSB on {far:.3%}, silhouettes differ on {xor:.3%} of the frame", flush=True)
    if best is None or mean < best[1]:
        best = (rot, mean, f, cov)
rot, mean, f, cov = best
err = lambda g: float(np.abs(g - ref)[cov].mean())  # noqa: E731
print(f"at {rot:+.4f} rad: signed mean per channel {np.round((f - ref)[cov].mean(axis=0), 3).tolist()}")
print("   shifted by one pixel: x+1 %.3f, x-1 %.3f, y+1 %.3f, y-1 %.3f" % (err(np.roll(f, 1, 1)), err(np.roll(f, -1, 1)), err(np.roll(f, 1, 0)), err(np.roll(f, -1, 0))))
print("   no normal map %.3f, upside down %.3f" % (err(render(rot, use_nm=False)), err(f[::-1])))
