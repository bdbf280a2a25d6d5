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


def big_cases():
    """Full-size configs[2], [3], [4] (tests/big_cases.py): hashes only.  `python tests/golden/make_golden.py --big`
    (the 8K frame of 10.2 M triangles takes the oracle about 8 s and ~6 GB)."""
    import big_cases as B

    out = {}
    W, H, draws = B.head_4k()
    out["c3_head_4k"] = B.oracle_digest(oracle_ffi, W, H, draws)
    for k in B.ORBIT_VIEWS:
        W, H, draws = B.orbit_view(k)
        out[f"c5_orbit_1080p_view{k}"] = B.oracle_digest(oracle_ffi, W, H, draws)
    W, H, draws = B.diablo_1080p_frame1()
    out["c5_diablo_1080p_frame1"] = B.oracle_digest(oracle_ffi, W, H, draws)
    assert out["c5_diablo_1080p_frame1"]["counters"]["samples"] == 1412616
    assert out["c5_diablo_1080p_frame1"]["covered"] == 192579
    W, H, draws = B.head_subdivided(6, 7680, 4320)
    out["c4_head_sub6_8k"] = B.oracle_digest(oracle_ffi, W, H, draws)
    return out


def ext_case():
    """configs[1] at full size with the EXTENSION shaders (Phong + specular map + shadow map): digests of the build's own
    CPU statement (oracle/), not of anything the reference computes.  `python tests/golden/make_golden.py --ext`."""
    import big_cases as B

    o, zl = B.config1_ext_oracle(oracle_ffi)
    d = B.digest_row_major(o["z"].T, o["key"].T, o["present"], o["counters"])
    d["light_z"] = B.sha(np.ascontiguousarray(zl.T).view(np.uint64))
    d["note"] = "extension shaders: the build's own CPU statement, parity with the reference unpinned by construction"
    return {"c2x_diablo_1024_phong_spec_shadow": d}


def main():
    path = os.path.join(HERE, "hashes.json")
    if "--ext" in sys.argv:
        hashes = json.load(open(path))
        hashes.update(ext_case())
        with open(path, "w") as fh:
            json.dump(hashes, fh, indent=1, sort_keys=True)
        print(json.dumps({k: v for k, v in hashes.items() if k.startswith("c2x_")}, indent=1))
        return
    if "--big" in sys.argv:  # (kept apart: the small cases take seconds, these a minute)
        hashes = json.load(open(path))
        hashes.update(big_cases())
        with open(path, "w") as fh:
            json.dump(hashes, fh, indent=1, sort_keys=True)
        print(json.dumps({k: v for k, v in hashes.items() if k.startswith(("c3_", "c4_", "c5_"))}, indent=1))
        return
    hashes = {k: v for k, v in json.load(open(path)).items() if k.startswith(("c2x_", "c3_", "c4_", "c5_"))} if os.path.exists(path) else {}
    for name, (_, W, H, full) in CASES.items():
        o = render(name)
        hashes[name] = digest(o)
        if full:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), z=o["z"], tri=o["seq"].astype(np.int32),
                                rgba8=o["present"])
        else:  # 1/64 subsample of the big cases (every 8th pixel in x and y)
            np.savez_compressed(os.path.join(HERE, name + "_sub8.npz"), z=o["z"][::8, ::8],
                                tri=o["seq"].astype(np.int32)[::8, ::8], rgba8=o["present"][::8, ::8])
    with open(path, "w") as fh:
        json.dump(hashes, fh, indent=1, sort_keys=True)
    print(json.dumps(hashes, indent=1))


if __name__ == "__main__":
    main()
