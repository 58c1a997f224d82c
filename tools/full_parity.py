"""BASELINE configs 1-4 at their FULL sizes (config 5 at a reduced instance count and 4 spp), GPU against the CPU oracle on the
same inputs and RNG: the tolerance of tests/parity_util.py, identical primary hits and identical ray counts.  The tests run
these configurations reduced so that the suite takes seconds; this is the full-size evidence, written to stdout as a
markdown table (committed as profiles/r01_full_parity.md).  Several minutes of host CPU (16 threads)."""
import importlib
import os
import sys
import time

import numpy as np

import torch  # noqa: F401  (one HIP runtime per process: torch's first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
synth = importlib.import_module("ray-tracer_amd.synth")
orc = importlib.import_module("oracle.orc")
from parity_util import compare_frames  # noqa: E402


def scene(name):
    return rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))


def main():
    cases = [("config 1  cornellbox_recursive 800x800, 1 spp, depth 6", scene("cornellbox_recursive"), {}),
             ("config 2a stand-in dragon 1M triangles 800x480, 1 spp, depth 6", synth.dragon_standin_scene(1_000_000, 800, 480), {}),
             ("config 2b scienceTree 1440x720, 1 spp, depth 6", scene("scienceTree"), {}),
             ("config 3  cornellbox_brushed_metal 800x800, 100 spp", scene("cornellbox_brushed_metal"), {"num_samples": 100}),
             ("config 3  spheres_dof 800x800, 100 spp", scene("spheres_dof"), {"num_samples": 100}),
             ("config 4  cornellbox_boxes_dynamic 750x600, 64 spp, depth 4", scene("cornellbox_boxes_dynamic"), {"num_samples": 64}),
             ("config 5  (reduced) 10 instances x 200k triangles, 1280x720, 4 spp", synth.instanced_scene(10, 200_000, 1280, 720, 4,
                                                                                                     perlin_table=rt.renderer.perlin_tables(1)[0]), {})]
    ctx = rt.Context(0)
    print("| configuration | rays | GPU ms (kernels) | oracle s (16 threads) | PSNR dB | max abs | pixels out of tolerance | values bit-equal | primary id / t mismatches |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, soup, kw in cases:
        ctx.upload(soup)
        ctx.render(0, seed=3, **kw)
        g_hdr, g_id, g_t, g_st = ctx.render(0, seed=3, timing=1, **kw)
        osc = orc.OracleScene(soup)
        t0 = time.time()
        c_hdr, c_id, c_t, c_st = osc.render(0, seed=3, nthreads=16, dynamic=True, **kw)
        cpu_s = time.time() - t0
        osc.close()
        m = compare_frames(g_hdr, c_hdr, g_id, c_id, g_t, c_t)
        assert list(g_st.rays) == list(c_st.rays), (name, list(g_st.rays), list(c_st.rays))
        print("| %s | %d | %.1f | %.1f | %.1f | %.2e | %d | %.2f %% | %d / %d |" % (name, sum(g_st.rays), g_st.ms_kernels, cpu_s, m["psnr"], m["max_abs"], m["bad_pixels"],
                                                                             100 * m["bit_equal"], m["id_mismatch"], m["t_mismatch"]), flush=True)


if __name__ == "__main__":
    main()
