"""The kernels evaluate Rust's f64::round (half away from zero) as trunc(v + copysign(0.5 - 2^-54, v))
(tr_device.h pixel_coord).  tests/ctools/round_trick.c compares that with C's round() on every
half-integer / integer neighbourhood (+-4 ulps) up to 2^22, around every power of two, and on
4.5e8 random values; IEEE-754 binary64 arithmetic is the same on the host and on gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_round_trick_matches_round(tmp_path):
    exe = str(tmp_path / "round_trick")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "ctools", "round_trick.c"), "-lm"],
                   check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    assert " 0 mismatches" in out, out


def test_div255_matches_division(tmp_path):
    """tr_device.h div255 (x / 255.0 as a multiply and two fmas) against the division on 3e8 values."""
    exe = str(tmp_path / "div255")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "ctools", "div255.c"), "-lm"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    assert " 0 mismatches" in out, out
