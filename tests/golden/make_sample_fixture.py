#!/usr/bin/env python3
"""Makes the two fixtures that come from the reference's only output artefact, /root/reference/sample.jpg
(README.md:5, "Result": a 2160x3840 JPEG of its window -- african_head + inner eyes, frontal): a data file the
reference holds, not source.

  reference_sample_2160x3840.jpg   the file itself, byte for byte (the full-resolution comparison of
                                   tests/test_reference_sample.py: geometry to a pixel, colour to JPEG noise)
  reference_sample_270x480.png     the same downscaled 8x with PIL (LANCZOS), for the coarse correlation test

Run in the build container only (the reference does not travel to the GPU box)."""
import os
import shutil

from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/sample.jpg"
im = Image.open(SRC).convert("RGB")
assert im.size == (2160, 3840)
im.resize((270, 480), Image.LANCZOS).save(os.path.join(HERE, "reference_sample_270x480.png"), optimize=True)
shutil.copyfile(SRC, os.path.join(HERE, "reference_sample_2160x3840.jpg"))
