#!/usr/bin/env python3
"""Generates tests/golden/*.npz and hashes.json with the CPU oracle (oracle/tr_oracle.cpp).

The reference ships no golden vectors and cannot be built here (no Rust toolchain), so these
fixtures pin the ORACLE's output on the reference's own scenes; they make regressions in the
oracle or the GPU path visible and give anyone with a Rust toolchain concrete buffers to
check the real reference against.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_ffi  # noqa: E402
import scene_util as SU  # noqa: E402
from tinyrenderer_amd import scenes  # noqa: E402

CASES = {
    # name: (component, W, H, full-buffer fixture?)
    "c1_head_128": ("head", 128, 128, True),
    "c2_diablo_200": ("diablo", 200, 200, True),
    "c1_head_800": ("head", 800, 800, False),
    "c2_diablo_1024": ("diablo", 1024, 1024, False),
}


def render(name):
    comp, W, H, _ = CASES[name]
    mc = SU.component(comp)
    mc.transform.set_rotation((0.025, 0.0, 0.0))
    return SU.oracle_render(oracle_ffi, W, H, [(mc, scenes.shader_for(mc, W, H))])


def digest(o):
    return {"z": hashlib.sha256(np.ascontiguousarray(o["z"]).tobytes()).hexdigest(),
            "tri": hashlib.sha256(np.ascontiguousarray(o["seq"].astype(np.int32)).tobytes()).hexdigest(),
            "key": hashlib.sha256(np.ascontiguousarray(o["key"]).tobytes()).hexdigest(),
            "rgba8": hashlib.sha256(np.ascontiguousarray(o["present"]).tobytes()).hexdigest(),
            "counters": {k: o["counters"][k] for k in ("samples", "fragments", "zpass", "scanlines")}}


def main():
    hashes = {}
    for name, (_, W, H, full) in CASES.items():
        o = render(name)
        hashes[name] = digest(o)
        if full:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), z=o["z"], tri=o["seq"].astype(np.int32),
                                rgba8=o["present"])
        else:  # 1/64 subsample of the big cases (every 8th pixel in x and y)
            np.savez_compressed(os.path.join(HERE, name + "_sub8.npz"), z=o["z"][::8, ::8],
                                tri=o["seq"].astype(np.int32)[::8, ::8], rgba8=o["present"][::8, ::8])
    with open(os.path.join(HERE, "hashes.json"), "w") as fh:
        json.dump(hashes, fh, indent=1, sort_keys=True)
    print(json.dumps(hashes, indent=1))


if __name__ == "__main__":
    main()
