"""GPU ray rates of the shipped (converted) scenes at their own resolution: a quick look for performance cliffs."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
names = sys.argv[1:] or ["cornellbox_recursive", "scienceTree", "bunny", "spheres", "cornellbox_boxes_dynamic", "cornellbox_brushed_metal"]
ctx = rt.Context(0)
for n in names:
    soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", n + ".npz"))
    ctx.upload(soup)
    spp = min(int(soup.cameras[0].get("num_samples", 1)), 16)
    for _ in range(3):
        _, _, _, st = ctx.render(0, want_ids=False, timing=1, num_samples=spp)
    rays = sum(st.rays)
    print("%-28s spp %3d rays %10d  ms_kernels %8.3f  trace %8.3f  Mrays/s %8.1f" % (n, spp, rays, st.ms_kernels, st.ms_trace, rays / st.ms_kernels / 1e3))
