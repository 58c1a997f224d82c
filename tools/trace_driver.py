"""Small GPU driver for profiling: uploads the config-2 scene and renders a few frames (no torch, no CPU baseline)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
synth = importlib.import_module("ray-tracer_amd.synth")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
soup = synth.dragon_standin_scene(tris, 800, 480)
ctx = rt.Context(0)
ctx.upload(soup)
streams = [int(x) for x in os.environ.get("RT_STREAMS", "0").split(",")]
side = int(os.environ.get("RT_SHADOW_STREAM", "0"))   # 1: shadow rays on a stream of their own (rt_render_opts.shadow_stream)
sizes = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("RT_SIZES", "800x480").split(",")]
for ns in streams:
  for (nx, ny) in sizes:
    if os.environ.get("RT_TRACE_DEBUG") or os.environ.get("RT_COUNT"):
        c = ctx.render(0, want_ids=False, timing=2, num_streams=ns, nx=nx, ny=ny)[3]   # counting frame (RT_TRACE_DEBUG: prints the lane-occupancy diagnostics)
        print("counted: node steps %.2f, triangle tests %.2f per ray" % (c.node_visits / c.total_rays(), c.tri_tests / c.total_rays()))
    for _ in range(frames):
        hdr, _, _, st = ctx.render(0, want_ids=False, timing=1, num_streams=ns, nx=nx, ny=ny, shadow_stream=side)
    import hashlib
    print("frame sha", hashlib.sha256(hdr.tobytes()).hexdigest()[:16], "shadow_stream", side)
    print("size", nx, ny, "streams", ns, "rays", list(st.rays), "ms_total %.3f ms_kernels %.3f ms_trace(sum) %.3f busy %.3f launches %d" %
          (st.ms_total, st.ms_kernels, st.ms_trace, st.ms_trace_busy, st.trace_launches))
