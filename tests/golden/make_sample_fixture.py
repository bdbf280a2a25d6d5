#!/usr/bin/env python3
"""Makes tests/golden/reference_sample_270x480.png from the reference's only output artefact,
/root/reference/sample.jpg (2160x3840 JPEG screenshot of its window, README.md:5): a data file the
reference holds, downscaled 8x with PIL (LANCZOS).  Run in the build container only."""
import os

from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
im = Image.open("/root/reference/sample.jpg").convert("RGB")
assert im.size == (2160, 3840)
im.resize((270, 480), Image.LANCZOS).save(os.path.join(HERE, "reference_sample_270x480.png"), optimize=True)
