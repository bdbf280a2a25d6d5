"""A few seeds of the fuzzers of tests/manual/ in the automated suite (the long runs are made by hand): the
frame state machine with moving bands and a peer context, bands of random scenes, batches of random views -- each against the
oracle.  The seeds include the ones on which the fuzzers found the three pixel-(0, 0) defects of round 3."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, argv, monkeypatch, env=None):
    spec = importlib.util.spec_from_file_location("fuzz_" + script, os.path.join(HERE, "manual", script + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(sys, "argv", [script] + [str(a) for a in argv])
    return mod.main()


@pytest.mark.parametrize("seed", [8201, 9738, 10241])
def test_frame_sequences_that_found_defects(oracle, monkeypatch, seed):
    for mode in (0, 2):
        assert _run("fuzz_frames", [1, seed, mode], monkeypatch, {"FUZZ_BANDS": "1"}) == 0


def test_frame_sequences(oracle, monkeypatch):
    assert _run("fuzz_frames", [10, 20000, 0], monkeypatch, {"FUZZ_BANDS": "1", "FUZZ_PEER": "1"}) == 0
    assert _run("fuzz_frames", [4, 20100, 1], monkeypatch, {"FUZZ_BANDS": "1"}) == 0


def test_bands_of_random_scenes(oracle, monkeypatch):
    assert _run("fuzz_bands", [8, 5000], monkeypatch) == 0


def test_batches_of_random_views(oracle, monkeypatch):
    assert _run("fuzz_views", [6, 7000], monkeypatch) == 0
    assert _run("fuzz_views_multi", [4, 13000], monkeypatch) == 0
