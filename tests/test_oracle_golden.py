"""Pins the oracle (and the scene loader that feeds it) against the only reference-produced outputs that exist for this
path: the author's published 8-bit renders (reference src/results/*.png, stored decoded under tests/golden/).  They come
from an older revision of the reference's sources, so the bound is loose per scene but tight in aggregate; for
cornellbox_recursive the survey recorded what a fresh build of the CURRENT reference sources gives against the same PNG —
27 232 differing pixels, max 220 — and the oracle reproduces exactly that count.

tests/golden/oracle_frames.json (tools/make_golden.py) additionally guards the oracle's own frames against regressions."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FRAMES = json.load(open(os.path.join(GOLD, "oracle_frames.json")))
PNG_SCENES = sorted(n for n, e in FRAMES.items() if "png_within_1" in e)


@pytest.fixture(scope="module")
def frames(rt, orc):
    cache = {}

    def get(name):
        if name not in cache:
            soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))
            hdr, pid, pt, st = orc.OracleScene(soup).render(0, nthreads=8, dynamic=True)
            cache[name] = (hdr, pid, st)
        return cache[name]
    return get


@pytest.mark.parametrize("name", PNG_SCENES)
def test_oracle_vs_published_png(rt, frames, name):
    hdr, pid, st = frames(name)
    img8 = rt.renderer.resolve_8bit(hdr)
    ref = np.load(os.path.join(GOLD, "ref_png_%s.npz" % name))["rgb"]
    assert ref.shape == img8.shape
    d = np.abs(img8.astype(int) - ref.astype(int)).max(axis=2)
    within1 = float((d <= 1).mean())
    assert within1 >= 0.95, (name, within1)                        # SURVEY §8c bound
    assert within1 >= FRAMES[name]["png_within_1"] - 1e-4          # and no worse than when the fixture was made
    if name == "cornellbox_recursive":
        assert int((d > 0).sum()) == 27232 and int(d.max()) == 220  # the figures the survey measured for the real reference


def test_aggregate_png_agreement(rt, frames):
    tot = bad = 0
    for name in PNG_SCENES:
        hdr, _, _ = frames(name)
        ref = np.load(os.path.join(GOLD, "ref_png_%s.npz" % name))["rgb"]
        d = np.abs(rt.renderer.resolve_8bit(hdr).astype(int) - ref.astype(int)).max(axis=2)
        tot += d.size
        bad += int((d > 1).sum())
    assert bad / tot < 0.006, bad / tot


@pytest.mark.parametrize("name", PNG_SCENES)
def test_oracle_frame_digests(frames, name):
    hdr, pid, st = frames(name)
    e = FRAMES[name]
    assert hashlib.sha256(pid.tobytes()).hexdigest() == e["sha256_prim_id"]      # hit ids do not depend on libm
    assert [int(x) for x in st.rays] == e["rays"]
    assert abs(float(hdr.astype(np.float64).sum()) - e["hdr_sum"]) <= 1e-6 * abs(e["hdr_sum"])
