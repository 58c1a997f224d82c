"""One-off wide fuzz: many seeds of tests/test_gpu_fuzz.py's random scenes (hits bit-exact, frames in tolerance, equal ray
counts).  usage: python3 tools/fuzz_sweep.py <first_seed> <n_seeds>"""
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
orc = importlib.import_module("oracle.orc")
fz = importlib.import_module("test_gpu_fuzz")
from parity_util import compare_frames  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = rt.Context(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    n_tris = int(rng.choice([300, 900, 2500, 5000, 8000]))
    depth = int(rng.randint(0, 5))
    fancy = bool(rng.randint(2))
    soup = fz.random_scene(seed, n_tris, depth, fancy)
    ctx.upload(soup)
    osc = orc.OracleScene(soup)
    rays = fz.rays_for(soup, 8000, seed)
    g, c = ctx.intersect(rays), osc.intersect(rays)
    ok = np.array_equal(g["shape"], c["shape"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    try:
        spp = int(rng.choice([1, 4]))
        gh, gi, gt, gs = ctx.render(0, num_samples=spp, seed=seed)
        ch, ci, ct, cs = osc.render(0, nthreads=16, num_samples=spp, seed=seed)
        m = compare_frames(gh, ch, gi, ci, gt, ct)
        ok = ok and list(gs.rays) == list(cs.rays)
    except AssertionError as e:
        ok = False
        m = str(e)[:200]
    osc.close()
    bad += 0 if ok else 1
    print("seed %d tris %d depth %d fancy %d spp %d: %s %s" % (seed, n_tris, depth, fancy, spp, "ok" if ok else "MISMATCH", m if not ok else ""), flush=True)
print("fuzz sweep: %d of %d seeds mismatched" % (bad, count))
sys.exit(1 if bad else 0)
