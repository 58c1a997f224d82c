#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X render hot path on BASELINE.json's config 2 (the ~1M-triangle "dragon" scene).

A "step" is one frame: ray generation, closest-hit traversal, shadow rays, shading, reflect/refract recursion to
depth 6, resolve into an HDR framebuffer that already lives in HBM (scene + BVH are uploaded before the timed region,
like the reference's own Timer excludes the BVH build, src/DefaultRenderer.cpp:47-61).  One "ray" is one closest-hit
query (the unit of AccelerationStructure::Intersect): primary + shadow + reflection + refraction.

N > 1 (torchrun, one process per GPU): the frame is cut into interleaved 32x32 tiles, rank r renders tiles with
id % N == r, and the HDR tiles are gathered on rank 0 over RCCL inside the timed region.  Scaling is WEAK: the image
resolution grows with N (same camera frustum, more pixels) so that every GPU keeps the 800x480-pixel share of config 2.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAY_BYTES = 32                 # closest-hit query: origin+time, direction+medium read (2 x 16 B)
HIT_BYTES = 32                 # closest-hit query: raw hit record written (2 x 16 B)
SHADOW_BYTES = 52              # shadow query: origin, direction+distance, shaded point+slot read (3 x 16 B), 4 B result written
NODE_BYTES = 112               # node step: 7 x 16 B of a DevNode4 (four child boxes + references); the eighth piece a step also loads, the four
                               # cone words of the exactness rule, is not counted as algorithmic
PRIM_BYTES = 48                # the 3 x 16 B of a DevPrim a triangle / sphere test reads (geometry, object, kind)
RES_BY_GPUS = {1: (800, 480), 2: (1600, 480), 4: (1600, 960), 8: (3200, 960)}
TILE = 32


def resolution_for(n_gpus):
    if n_gpus in RES_BY_GPUS:
        return RES_BY_GPUS[n_gpus]
    return 800 * n_gpus, 480


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tris", type=int, default=1_000_000, help="triangles of the stand-in mesh (config 2: 1M)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the resolution grows with N (each GPU keeps config 2's 800x480 share); strong: one fixed frame (800x480 at --spp samples) cut over the N GPUs")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: 1 = config 2 as quoted; 16 with --scaling strong)")
    ap.add_argument("--workload", choices=["config2", "config5"], default="config2",
                    help="config2: the headline workload; config5: BASELINE config 5's geometry (50 instances of the 1M-triangle mesh, 3840x2160), one fixed frame: implies --scaling strong")
    ap.add_argument("--instances", type=int, default=50, help="mesh instances of --workload config5")
    ap.add_argument("--eighth-share", action="store_true",
                    help="also time the share one GPU of eight gets of this frame (every 8th tile); off by default so that the default run launches only whole frames")
    ap.add_argument("--broadcast-scene", action="store_true",
                    help="N > 1: rank 0 builds the scene and the other ranks receive its arenas over RCCL (rt_scene_export / rt_scene_import) instead of building their own")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process group of N > 1.  nccl (= RCCL, device tensors) is the measured path; gloo is a rehearsal for boxes with fewer GPUs than "
                         "ranks: all ranks share the GPUs there are, tiles are staged through host memory, and the line is marked as a rehearsal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frames-in-flight", action="store_true", help="skip the extra measurement with two frames in flight")
    ap.add_argument("--cpu-row-step", type=int, default=1, help="CPU baseline renders every k-th row of the frame")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    rehearsal = world > 1 and args.backend == "gloo"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = torch.device("cpu") if rehearsal else dev              # where the tensors of the timing reductions live (gloo: CPU tensors only)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    rt = importlib.import_module("ray-tracer_amd")
    synth = importlib.import_module("ray-tracer_amd.synth")
    abi = rt.abi

    c5 = args.workload == "config5"
    if c5:
        args.scaling = "strong"
    strong = args.scaling == "strong"
    spp = args.spp if args.spp > 0 else (1 if c5 else (16 if strong else 1))
    if c5:
        # instances and primitive runs go to the library in compact form and are expanded in HBM (52 M triangles at the defaults)
        nx, ny = 3840, 2160
        soup = synth.instanced_scene(args.instances, args.tris, nx, ny, spp, perlin_table=rt.renderer.perlin_tables(1)[0], compact=True)
        args.no_cpu_baseline = True               # the CPU restatement needs minutes for this scene
    else:
        nx, ny = resolution_for(1 if strong else world)
        soup = synth.dragon_standin_scene(args.tris, nx, ny)
        if spp != 1:
            soup.cameras[0]["num_samples"] = spp
    ctx = rt.Context(local_rank)
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    t0 = time.time()
    if args.broadcast_scene and world > 1:
        if rank == 0:
            ctx.upload(soup)
        mg.broadcast_scene(ctx, dist, rank, 0, dev, torch, host_staging=rehearsal, soup=soup)
    else:
        ctx.upload(soup)                            # default: every rank builds its own copy, in parallel
    upload_s = time.time() - t0
    cd = soup.camera_desc(0)

    tg = mg.TileGather(nx, ny, TILE, rank, world, dev, torch, host_staging=rehearsal)                    # compact owned pixels + frame, HBM resident

    def step(opts, stats):
        tg.acquire()                                                              # the other of two tile buffers: frame i's gather may still run
        ctx.render_device(cd, opts, tg.tile_hdr.data_ptr(), None, None, stats)    # returns after the stream is idle
        # the only exchange step of the path: HDR tiles -> rank 0; point-to-point gather over xGMI
        tg.gather(dist)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    opts_count = ctx.make_opts(seed=0, tile_size=TILE, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True, timing=2)
    opts_timed = ctx.make_opts(seed=0, tile_size=TILE, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True, timing=1)
    st_count = abi.Stats()
    step(opts_count, st_count)                      # untimed: per-frame traversal counters (deterministic)
    for _ in range(args.warmup):
        step(opts_timed, abi.Stats())
    sync()
    t0 = time.perf_counter()
    st_steps = [abi.Stats() for _ in range(args.steps)]
    for i in range(args.steps):
        step(opts_timed, st_steps[i])
    sync()
    elapsed = time.perf_counter() - t0

    # N > 1, weak scaling of config 2: the same JSON line also carries the strong regime — the fixed 800x480 frame of config 2 at 1 spp
    # and at 16 spp cut over the N ranks (tiles + gather inside the timed region), next to the time one GPU alone needs for that frame
    # (every rank renders the whole frame by itself at the same time: N independent replicas, rank 0's clock counts)
    strong_obj = None
    if world > 1 and not c5 and not strong:
        strong_obj = {}
        snx, sny = RES_BY_GPUS[1]
        tg_s = mg.TileGather(snx, sny, TILE, rank, world, dev, torch, host_staging=rehearsal)
        whole = torch.zeros((snx * sny, 3), dtype=torch.float32, device=dev)
        for sspp, ssteps in ((1, 20), (16, 5)):
            scd = soup.camera_desc(0, **synth.dragon_camera(snx, sny, sspp))
            o_one = ctx.make_opts(seed=0, tile_size=TILE, tile_rank=0, tile_world=1, output_on_device=True, compact_output=True, timing=1)
            o_cut = ctx.make_opts(seed=0, tile_size=TILE, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True, timing=1)
            res = {}
            for label, o in (("one_gpu", o_one), ("cut", o_cut)):
                def sstep(st):
                    if label == "one_gpu":
                        ctx.render_device(scd, o, whole.data_ptr(), None, None, st)
                    else:
                        tg_s.acquire()
                        ctx.render_device(scd, o, tg_s.tile_hdr.data_ptr(), None, None, st)
                        tg_s.gather(dist)
                for _ in range(2):
                    sstep(abi.Stats())
                sync()
                ts = time.perf_counter()
                sts = [abi.Stats() for _ in range(ssteps)]
                for st in sts:
                    sstep(st)
                sync()
                dt = (time.perf_counter() - ts) / ssteps
                v = torch.tensor([dt, float(sum(sts[0].rays)), float(max(x.frame_passes for x in sts)), float(max(x.level_syncs for x in sts))],
                                 dtype=torch.float64, device=rdev)
                mxv = v.clone()
                dist.all_reduce(mxv, op=dist.ReduceOp.MAX)
                smv = v.clone()
                dist.all_reduce(smv, op=dist.ReduceOp.SUM)
                res[label] = (float(mxv[0]), float(smv[1]), int(mxv[2]), int(mxv[3]), float(v[0]), float(v[1]))
            t_cut, rays_cut = res["cut"][0], res["cut"][1]
            t_one, rays_one = res["one_gpu"][4], res["one_gpu"][5]          # this rank's own clock and rays (rank 0 prints)
            strong_obj["spp_%d" % sspp] = {
                "resolution": [snx, sny], "steps": ssteps, "rays_per_frame": int(rays_cut), "ms_per_frame": round(t_cut * 1e3, 4),
                "value": round(rays_cut / t_cut / 1e6, 2), "unit": "Mrays/s", "one_gpu_ms_per_frame": round(t_one * 1e3, 4),
                "one_gpu_value": round(rays_one / t_one / 1e6, 2), "speedup": round(t_one / t_cut, 3), "efficiency": round(t_one / t_cut / world, 4),
                "max_frame_passes": res["cut"][2], "max_level_syncs": res["cut"][3],
                "note": "time = slowest rank incl. the gather on rank 0; one_gpu = the whole frame on rank 0 alone while the other ranks do the same"}

    rays_rank = float(sum(st_steps[0].rays))
    vals = torch.tensor([elapsed, rays_rank, sum(s.ms_trace for s in st_steps), sum(s.trace_launches for s in st_steps),
                         float(st_count.node_visits), float(st_count.tri_tests), float(st_count.sphere_tests),
                         float(sum(st_count.rays)), float(st_count.rays[abi.RT_RAY_SHADOW]),
                         float(max(s.frame_passes for s in st_steps)), float(max(s.level_syncs for s in st_steps))], dtype=torch.float64, device=rdev)
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        total_rays = float(sm[1])
        max_passes, max_syncs = int(mx[9]), int(mx[10])
    else:
        total_rays = rays_rank
        max_passes, max_syncs = int(vals[9]), int(vals[10])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = total_rays / (elapsed / args.steps) / 1e6

    # roofline of the dominant kernel (k_trace_generic), rank 0's launches: algorithmic bytes / HIP-event time
    n_node, n_tri, n_sph = float(st_count.node_visits), float(st_count.tri_tests), float(st_count.sphere_tests)
    n_rays = float(sum(st_count.rays))
    n_shadow = float(st_count.rays[abi.RT_RAY_SHADOW])
    frame_bytes = NODE_BYTES * n_node + PRIM_BYTES * (n_tri + n_sph) + (RAY_BYTES + HIT_BYTES) * (n_rays - n_shadow) + SHADOW_BYTES * n_shadow
    launches = sum(s.trace_launches for s in st_steps) / args.steps
    trace_ms = sum(s.ms_trace for s in st_steps) / args.steps
    bytes_per_launch = frame_bytes / launches
    avg_launch_ms = trace_ms / launches
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes (tools/pmc_passes.sh + tools/traffic_json.py): only quoted when they were taken with
    # exactly this library (the file records its SHA-256) and for this workload; otherwise null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world == 1 and spp == 1 and args.tris == 1_000_000 and not c5:
        try:
            import hashlib
            tj = json.load(open(tpath))
            if tj.get("lib_sha256") == hashlib.sha256(open(rt.renderer.LIB_PATH, "rb").read()).hexdigest():
                traffic = tj.get("hbm_bytes_per_trace_launch")
        except Exception:
            traffic = None
    # the same launch time priced with a second, traversal-independent yardstick: 64 B per node visit and per primitive test of a canonical
    # binary, nearest-first, pruned traversal of the reference tree (counted by the CPU oracle, tests/golden/canonical_counts.json)
    survey = None
    cpath = os.path.join(ROOT, "tests", "golden", "canonical_counts.json")
    ckey = "config2 dragon_standin_%d %dx%d" % (args.tris, nx, ny)
    if os.path.exists(cpath) and spp == 1 and not c5:
        try:
            cc = json.load(open(cpath)).get(ckey)
            if cc and cc["rays"] == int(n_rays) and world == 1:
                sbytes = cc["bytes_per_ray"] * cc["rays"] / launches
                survey = {"bytes_per_ray": round(cc["bytes_per_ray"], 1), "nodes_per_ray": round(cc["nodes"] / cc["rays"], 2),
                          "tris_per_ray": round(cc["tris"] / cc["rays"], 2), "achieved": round(sbytes / (avg_launch_ms * 1e-3) / 1e9, 2),
                          "frac": round(sbytes / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                # flop count of the same canonical traversal (24 per node visit, 140 per triangle, 60 per sphere test)
                # against the FP32 vector peak it names (157 TFLOP/s): the path is nowhere near a compute roofline
                sflops = 24.0 * cc["nodes"] + 140.0 * cc["tris"] + 60.0 * cc["spheres"]
                survey["flops_per_ray"] = round(sflops / cc["rays"], 1)
                survey["tflops"] = round(sflops / (trace_ms * 1e-3) / 1e12, 3)
                survey["frac_fp32_valu_157"] = round(sflops / (trace_ms * 1e-3) / 1e12 / 157.0, 5)
        except Exception:
            survey = None
    # "bound": of the contract's two choices (hbm, mfma) hbm is the one that applies: there is no matrix work.  The kernel does not reach that
    # roof: measured HBM traffic is about 40 % of peak while VALU instruction issue is at 75-90 % with half the lanes active (counter passes under profiles/ and
    # a probe build that adds v_nop instructions to the loop).  `achieved` / `frac` price the ALGORITHMIC bytes; traffic_frac = measured HBM bytes / launch time / peak.
    roofline = {"bound": "hbm", "limited_by": "VALU instruction issue of the divergent wave64 traversal (about half the lanes active); measured HBM traffic is ~40 % of peak",
                "traffic_frac": (round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
                "achieved_is": "a model: counted node steps and primitive tests x the record bytes each reads (the node's cone words, a 16-byte piece only lanes "
                               "with a bound load, are not counted), not measured traffic; traffic / traffic_frac are the measured HBM bytes",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "kernel": "k_trace_generic (closest-hit rays of a level + shadow rays of the level before, one launch)", "launches_per_frame": launches,
                "avg_launch_ms": round(avg_launch_ms, 5), "algorithmic_bytes_per_launch": round(bytes_per_launch),
                "per_ray": {"nodes": round(n_node / n_rays, 2), "tris": round(n_tri / n_rays, 2), "bytes": round(frame_bytes / n_rays, 1)},
                "trace_ms_per_frame": round(trace_ms, 4), "survey_8d_formula": survey}

    cpu = None
    if not args.no_cpu_baseline and world == 1:       # rank 0 at N=1 only
        from oracle import orc
        osc = orc.OracleScene(soup)
        # one thread per core this job may use: the affinity mask, capped by OMP_NUM_THREADS (the GPU box grants 16 of the
        # machine's cores to a one-GPU job); os.cpu_count() — the whole machine — is reported next to it
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, int(os.environ.get("OMP_NUM_THREADS", avail) or avail)))
        t0 = time.perf_counter()
        _, _, _, cst = osc.render(0, nthreads=cores, dynamic=True, row_step=args.cpu_row_step, want_ids=False)
        cpu_s = time.perf_counter() - t0
        cpu = {"value": round(sum(cst.rays) / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
               "sample": "every %dth row of the %dx%d frame (%d rows, %d rays, %.1f s), oracle/rt_oracle.cpp with %d threads, rows handed out dynamically"
                         % (args.cpu_row_step, nx, ny, len(range(0, ny, args.cpu_row_step)), sum(cst.rays), cpu_s, cores)}
        # the reference's own split (DefaultRenderer.cpp:65-75): 8 threads, static row bands
        t0 = time.perf_counter()
        _, _, _, cst8 = osc.render(0, nthreads=8, dynamic=False, row_step=args.cpu_row_step, want_ids=False)
        cpu["value_8_static_bands"] = round(sum(cst8.rays) / (time.perf_counter() - t0) / 1e6, 4)
        osc.close()

    # the share one GPU of eight gets of this frame (every 8th 32x32 tile): what fixed costs leave of strong scaling
    share_ms = None
    if world == 1 and args.eighth_share:
        o8 = ctx.make_opts(seed=0, tile_size=TILE, tile_rank=0, tile_world=8, output_on_device=True, compact_output=True, timing=1)
        for _ in range(3):
            ctx.render_device(cd, o8, tg.tile_hdr.data_ptr(), None, None, abi.Stats())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            st8 = ctx.render_device(cd, o8, tg.tile_hdr.data_ptr(), None, None, abi.Stats())
        torch.cuda.synchronize()
        share_ms = {"ms_per_frame": round((time.perf_counter() - t0) * 100.0, 4), "kernels_ms": round(st8.ms_kernels, 4), "rays": int(sum(st8.rays)),
                    "level_syncs": int(st8.level_syncs)}
    # Throughput with two frames in flight (an animation, or the cameras of one scene): a second context adopts the built scene
    # (rt_scene_export / rt_scene_import, no second build) and two host threads render the same frame into buffers of their own; one
    # context's launch tails then overlap the other's launches.  Reported beside `value`, which stays one frame, start to finish.
    two_in_flight = None
    if world == 1 and not c5 and not args.no_frames_in_flight:
        import threading
        hdr, sizes = ctx.export_scene_header()
        bufs = [torch.empty(max(nb, 1), dtype=torch.uint8, device=dev) for nb in sizes]
        for ai, nb in enumerate(sizes):
            if nb:
                ctx.export_scene_arena(ai, bufs[ai].data_ptr())
        ctx2 = rt.Context(local_rank).import_scene(hdr, [b.data_ptr() if nb else 0 for b, nb in zip(bufs, sizes)], soup=soup)
        del bufs
        pair = [(ctx, torch.zeros((ny * nx, 3), dtype=torch.float32, device=dev)), (ctx2, torch.zeros((ny * nx, 3), dtype=torch.float32, device=dev))]
        o_plain = [c.make_opts(seed=0, output_on_device=True) for c, _ in pair]
        for (c, o), op in zip(pair, o_plain):
            c.render_device(cd, op, o.data_ptr())            # the imported context's first frame is host-counted
        per = max(1, args.steps // 2)

        def frames(c, o, op):
            for _ in range(per):
                c.render_device(cd, op, o.data_ptr())
        dt2 = None
        for _ in range(2):                                   # a warm-up round, then the timed one
            th = [threading.Thread(target=frames, args=(c, o, op)) for (c, o), op in zip(pair, o_plain)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
        same = bool(torch.equal(pair[0][1].view(torch.int32), pair[1][1].view(torch.int32)))
        two_in_flight = {"frames": 2 * per, "ms_per_frame": round(dt2 * 1e3 / (2 * per), 4), "value": round(total_rays * 2 * per / dt2 / 1e6, 2), "unit": "Mrays/s",
                         "frames_bit_identical": same, "note": "two contexts sharing one built scene, one host thread each; not the headline value"}
        ctx2.close()
    bvh = ctx.bvh_info()
    out = {"metric": "Mrays/s (primary+secondary, whole node)", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: gloo, %d ranks share %d GPU(s), tiles staged through host memory — not a measurement)" % (world, torch.cuda.device_count()) if rehearsal else ""),
           "config": {"workload": ("BASELINE config 5 geometry: %d instances of the stand-in %d-triangle mesh (half with Perlin bump), textured floor, depth 6, %d spp"
                                   % (args.instances, args.tris, spp)) if c5 else
                                  ("BASELINE config 2: other_dragon.xml camera/lights/materials, depth 6, %d spp, stand-in %d-triangle mesh "
                                   "(dragon_remeshed.ply is a missing blob) + mirror floor" % (spp, args.tris)),
                      "resolution": [nx, ny], "spp": spp, "rays_per_frame": int(total_rays), "tiles": "%dx%d interleaved over %d GPU(s)" % (TILE, TILE, world),
                      "bvh_nodes": int(bvh.n_nodes), "bvh_depth": int(bvh.max_depth), "scene_upload_s": round(upload_s, 2), "kernels_ms_per_frame": round(sum(s.ms_kernels for s in st_steps) / args.steps, 4),
                      "host_round_trips_per_frame": int(st_steps[-1].level_syncs),
                      "max_level_syncs": max_syncs, "max_frame_passes": max_passes, "traversal": "exact (normal-cone pruning rule): hit ids equal the reference's unpruned traversal",
                      "one_gpu_of_eight_share": share_ms, "two_frames_in_flight": two_in_flight},
           "roofline": roofline, "cpu_baseline": cpu}
    if strong_obj is not None:
        out["strong"] = strong_obj
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
