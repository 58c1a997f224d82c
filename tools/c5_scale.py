"""BASELINE config 5 geometry at (a fraction of) full scale on one GPU: instances of the 1M-triangle mesh, GPU BVH build,
size-independent checks (determinism, tile union == full frame).  usage: c5_scale.py <instances> <nx> <ny> <spp>"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
synth = importlib.import_module("ray-tracer_amd.synth")
ni, nx, ny, spp = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 10), (2, 1920), (3, 1080), (4, 1)))
t0 = time.time()
compact = bool(os.environ.get("C5_COMPACT"))        # instances and primitive runs instead of written-out clones: the library expands them in HBM
soup = synth.instanced_scene(ni, 1_000_000, nx, ny, spp, perlin_table=rt.renderer.perlin_tables(1)[0], compact=compact)
print("scene: %d explicit triangles, %d instances, %d primitive references / runs, described in %.1f s" % (
    len(soup.tri_vert), len(soup.instances), len(soup.prim_runs) if soup.prim_runs is not None else len(soup.prim_shape), time.time() - t0), flush=True)
ctx = rt.Context(0)
t0 = time.time()
ctx.upload(soup)
print("rt_scene_upload %.2f s" % (time.time() - t0), ctx.bvh_info().n_nodes, "reference nodes,", ctx.bvh_info().n_wide_nodes, "wide nodes, depth", ctx.bvh_info().max_depth, flush=True)
kw = {"max_batch_samples": int(os.environ.get("C5_BATCH", "0")), "num_streams": int(os.environ.get("C5_STREAMS", "0"))}
a, ida, ta, st = ctx.render(0, timing=1, **kw)
b, idb, tb, st2 = ctx.render(0, timing=1, **kw)
print("frame: rays", list(st2.rays), "%.1f ms kernels, %.1f ms trace in %d launches, %.1f Mrays/s" % (st2.ms_kernels, st2.ms_trace, st2.trace_launches, sum(st2.rays) / st2.ms_kernels / 1e3), flush=True)
if os.environ.get("C5_QUICK"):
    sys.exit(0)
assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(ida, idb), "not deterministic"
full = a.reshape(-1, 3)
union = np.zeros_like(full)
for r in range(3):
    part, _, _, _ = ctx.render(0, want_ids=False, tile_size=32, tile_rank=r, tile_world=3, compact_output=True)
    order = rt.renderer.tile_pixel_order(nx, ny, 32, r, 3)
    union[order] = part
assert np.array_equal(union.view(np.uint32), full.view(np.uint32)), "tile union differs from the full frame"
print("hit pixels %.1f %%, deterministic, union of 3 tile ranks == full frame" % (100.0 * (ida >= 0).mean()))
