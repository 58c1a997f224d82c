"""Throughput with several frames in flight: K contexts on one GPU share one built scene (rt_scene_export / rt_scene_import) and render
frames of config 2 from K host threads (rt_render blocks its caller, not the GPU).  usage: python tools/frames_in_flight.py [K ...]"""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
synth = importlib.import_module("ray-tracer_amd.synth")

ks = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]
frames = 24
soup = synth.dragon_standin_scene(1_000_000, 800, 480)
first = rt.Context(0).upload(soup)
hdr, sizes = first.export_scene_header()
bufs = [torch.empty(max(n, 1), dtype=torch.uint8, device="cuda:0") for n in sizes]
for i, n in enumerate(sizes):
    first.export_scene_arena(i, bufs[i].data_ptr())
ctxs = [first]
for _ in range(max(ks) - 1):
    ctxs.append(rt.Context(0).import_scene(hdr, [b.data_ptr() if n else 0 for b, n in zip(bufs, sizes)], soup=soup))
cd = soup.camera_desc(0)
outs = [torch.zeros((480 * 800, 3), dtype=torch.float32, device="cuda:0") for _ in ctxs]
rays = None
for c, o in zip(ctxs, outs):                       # first frame of every context: host-counted
    opts = c.make_opts(output_on_device=True, timing=1)
    st = c.render_device(cd, opts, o.data_ptr())
    rays = sum(st.rays)
ref = outs[0].cpu().numpy()
for o in outs[1:]:
    assert np.array_equal(o.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def worker(c, o, n):
    opts = c.make_opts(output_on_device=True)
    for _ in range(n):
        c.render_device(cd, opts, o.data_ptr())


for k in ks:
    per = frames // k
    for _ in range(2):                                # warm-up round, then the timed one
        th = [threading.Thread(target=worker, args=(ctxs[j], outs[j], per)) for j in range(k)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("frames in flight %d: %d frames in %.2f ms = %.3f ms per frame, %.0f Mrays/s" % (k, per * k, dt * 1e3, dt * 1e3 / (per * k), rays * per * k / dt / 1e6), flush=True)
for o in outs:
    assert np.array_equal(o.cpu().numpy().view(np.uint32), ref.view(np.uint32))
print("all frames bit-identical")
