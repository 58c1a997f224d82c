"""Pins the oracle (and the scene loader that feeds it) against the only reference-produced outputs that exist for this
path: the author's published 8-bit renders (reference src/results/*.png, stored decoded under tests/golden/).  They come
from an older revision of the reference's sources, so the bound is loose per scene but tight in aggregate; for
cornellbox_recursive the survey recorded what a fresh build of the CURRENT reference sources gives against the same PNG —
27 232 differing pixels, max 220 (recorded when the reference could still be run) — and the oracle reproduces exactly that count.

tests/golden/oracle_frames.json (tools/make_golden.py) additionally guards the oracle's own frames against regressions."""
import hashlib
import importlib
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
    assert within1 >= 0.95, (name, within1)                        # the author's PNGs come from an older revision: loose bound
    assert within1 >= FRAMES[name]["png_within_1"] - 1e-4          # and no worse than when the fixture was made
    if name == "cornellbox_recursive":
        assert int((d > 0).sum()) == 27232 and int(d.max()) == 220  # the figures measured for the real reference's current sources
        # WHERE the published PNG (older revision) and the current algorithm differ, by the object the primary ray hits: 25 238
        # of the 27 232 pixels lie on the dielectric sphere (99.8 % of its pixels: the older revision shaded glass differently —
        # RecursiveComputation.cpp:124-179 is the code path), 1 317 on the conductor sphere (9.5 % of its pixels), 677 on
        # the five walls (0.11 % of theirs: silhouette and shadow-edge pixels)
        soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))
        sph = (pid >= 0) & ((pid & rt.abi.RT_PRIM_SPHERE) != 0)
        obj = np.full(pid.shape, -1)
        obj[sph] = np.asarray(soup.sph_object)[pid[sph] & rt.abi.RT_PRIM_INDEX_MASK]
        tri = (pid >= 0) & ~sph
        obj[tri] = np.asarray(soup.tri_object)[pid[tri]]
        mtype = np.asarray(soup.materials["type"])[np.asarray(soup.objects["material"])]
        glass, metal = obj == int(np.nonzero(mtype == rt.abi.RT_MAT_DIELECTRIC)[0][0]), obj == int(np.nonzero(mtype == rt.abi.RT_MAT_CONDUCTOR)[0][0])
        assert int((d[glass] > 0).sum()) == 25238 and int((d[metal] > 0).sum()) == 1317
        walls = ~glass & ~metal
        assert int((d[walls] > 0).sum()) == 677 and (d[walls] > 0).mean() < 0.0012


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


# ---- the reference's image-textured scenes against its published renders -------------------------------------------------
TEXTURED = json.load(open(os.path.join(GOLD, "oracle_textured_frames.json")))


@pytest.mark.parametrize("name", sorted(TEXTURED))
def test_oracle_vs_published_textured_png(rt, frames, name):
    """Image textures (replace_kd, blend_kd, bump and normal maps, nearest / bilinear look-up, textures/ of the reference) through
    loader + oracle against the author's render of the same scene, by the means of 16x16-pixel blocks: per pixel the published
    files differ from the current sources by sub-texel amounts (older revision), and the two Perlin ellipsoids of ellipsoids_texture
    by the reference's start-up shuffle, so the bound is on orientation / scale / decal mode / shading, not on the last texel."""
    e = TEXTURED[name]
    if e.get("num_samples"):
        # a sampled scene (depth of field, area light, motion blur, glossy mirrors ...): the reference's generators are time-seeded,
        # so its published frame and the oracle's (counter-based RNG, 16 samples per pixel) can only agree in the block means
        soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))
        orc_mod = importlib.import_module("oracle.orc")
        hdr, pid, pt, st = orc_mod.OracleScene(soup).render(0, nthreads=8, dynamic=True, num_samples=e["num_samples"])
    else:
        hdr, pid, st = frames(name)
    assert hashlib.sha256(pid.tobytes()).hexdigest() == e["sha256_prim_id"] and [int(x) for x in st.rays] == e["rays"]
    img8 = rt.renderer.resolve_8bit(hdr)
    if e.get("tonemapped"):
        # the one scene with <Tonemap>: the author published an 8-bit conversion of the EXR (converter unknown).  The oracle's tone
        # mapper on the float radiance + the scene's display gamma lands within 1.5 levels per block; on the 8-bit image (what the
        # reference's current sources feed it, Q20) 5.8 — a plausibility bound for the operator, not a pin.
        soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))
        key, burn, sat, gamma = soup.tonemap
        mapped, _ = importlib.import_module("oracle.orc").tonemap(hdr, key, burn, sat)
        img8 = np.clip(np.power(np.clip(mapped, 0, None), 1.0 / gamma), 0, 1) * 255.0
    ref_blocks = np.load(os.path.join(GOLD, "ref_png_blocks_%s.npz" % name))["blocks"].astype(np.float64)
    b = 16
    h, w = img8.shape[0] // b * b, img8.shape[1] // b * b
    mine = img8[:h, :w].astype(np.float64).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))
    d = np.abs(mine - ref_blocks).max(axis=2)
    if e.get("tonemapped"):
        assert d.mean() <= 2.0 and (d > 8).mean() <= 0.01 and abs(float(mine.mean()) - float(ref_blocks.mean())) <= 2.5
        return
    assert d.mean() <= max(1.0, 1.02 * e["block_mean_diff"] + 0.01), (name, d.mean())        # 8-bit levels; <= 1 for all but the Perlin scene
    assert (d > 8).mean() <= e["blocks_off_by_8"] + 0.002, (name, (d > 8).mean())
    assert (d > 8).mean() <= (0.06 if name == "ellipsoids_texture" else 0.01)
    assert abs(float(mine.mean()) - float(ref_blocks.mean())) <= 0.6           # whole (block-covered) frame, all channels
