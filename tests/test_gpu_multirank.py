"""The N > 1 path end to end on the GPU: two processes (gloo, one rt_ctx each on the box's GPU) render the interleaved tiles
they own with rt_render(tile_rank, tile_world, compact output into HBM) and rank 0 gathers + de-interleaves them
(ray-tracer_amd/multigpu.py).  gloo gets CPU tensors only: the tiles are staged through host memory with synchronous copies
(on an 8-GPU node the same code runs over the nccl backend = RCCL on device tensors, bench.py --gpus N).  Rank 0 checks the
assembled frame against its own single-rank render (bit for bit) and against the CPU oracle (tolerance of parity_util)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rt = importlib.import_module("ray-tracer_amd")
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    synth = importlib.import_module("ray-tracer_amd.synth")
    nx, ny, tile = 320, 192, 32
    soup = synth.dragon_standin_scene(60000, nx, ny)
    ctx = rt.Context(0)
    ctx.upload(soup)
    cd = soup.camera_desc(0)
    tg = mg.TileGather(nx, ny, tile, rank, world, dev, torch, host_staging=True)
    opts = ctx.make_opts(seed=0, tile_size=tile, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True)
    rays = 0
    for it in range(3):                                   # three frames through the two alternating tile buffers
        buf = tg.acquire()
        buf.zero_()
        torch.cuda.synchronize()
        st = ctx.render_device(cd, opts, buf.data_ptr())
        frame = tg.gather(dist)
        rays = sum(st.rays)
    tot = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        full, fid, ft, fst = ctx.render(0)
        got = frame.numpy().reshape(ny, nx, 3)
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "gathered frame differs from the single-rank frame"
        assert int(tot[0]) == sum(fst.rays), (int(tot[0]), sum(fst.rays))
        from oracle import orc
        from parity_util import compare_frames
        c_hdr, c_id, c_t, c_st = orc.OracleScene(soup).render(0, nthreads=8, dynamic=True)
        m = compare_frames(got, c_hdr, fid, c_id, ft, c_t)
        assert list(fst.rays) == list(c_st.rays)
        print("MULTIRANK_OK", m, int(tot[0]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
''')


def test_two_ranks_render_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "MULTIRANK_OK" in out.stdout


BCAST_WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rt = importlib.import_module("ray-tracer_amd")
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    synth = importlib.import_module("ray-tracer_amd.synth")
    nx, ny, tile = 320, 192, 32
    soup = synth.dragon_standin_scene(60000, nx, ny)          # every rank knows the cameras; only rank 0 builds the scene
    ctx = rt.Context(0)
    if rank == 0:
        ctx.upload(soup)
    sizes = mg.broadcast_scene(ctx, dist, rank, 0, dev, torch, host_staging=True, soup=soup)
    assert sum(sizes) > 10_000_000 and len(sizes) == rt.abi.RT_SCENE_ARENAS
    info = ctx.bvh_info()
    cd = soup.camera_desc(0)
    tg = mg.TileGather(nx, ny, tile, rank, world, dev, torch, host_staging=True)
    opts = ctx.make_opts(seed=0, tile_size=tile, tile_rank=rank, tile_world=world, output_on_device=True, compact_output=True)
    buf = tg.acquire()
    buf.zero_()
    torch.cuda.synchronize()
    ctx.render_device(cd, opts, buf.data_ptr())
    frame = tg.gather(dist)
    nodes = torch.tensor([int(info.n_nodes)], dtype=torch.int64)
    both = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(both, nodes)
    if rank == 0:
        assert int(both[0][0]) == int(both[1][0]) > 0
        full, fid, ft, fst = ctx.render(0)
        got = frame.numpy().reshape(ny, nx, 3)
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "frame of (built tiles + imported tiles) differs from the single-rank frame"
        print("BROADCAST_OK", int(both[0][0]), sum(sizes))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
''')


def test_scene_built_once_and_broadcast(tmp_path):
    """Rank 0 builds the scene, rank 1 receives header + arenas (rt_scene_export / rt_scene_import through multigpu.broadcast_scene,
    gloo with host staging here, RCCL on device buffers on a multi-GPU node) and renders its tiles without ever calling rt_scene_upload."""
    script = tmp_path / "bworker.py"
    script.write_text(BCAST_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29543", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "BROADCAST_OK" in out.stdout


def test_bench_two_rank_line_carries_both_regimes():
    """bench.py's N > 1 code path as a two-rank gloo rehearsal on the one GPU (the timing means nothing, the line's shape does): the weak
    value plus a `strong` object for the fixed 800x480 frame at 1 and 16 spp, and the worst frame_passes / level_syncs of the timed steps."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29545", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tris", "60000",
                          "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "REHEARSAL" in j["data"] and j["config"]["resolution"] == [1600, 480]
    assert j["config"]["max_frame_passes"] == 1 and j["config"]["max_level_syncs"] == 0
    one = None
    for key in ("spp_1", "spp_16"):
        s = j["strong"][key]
        assert s["resolution"] == [800, 480] and s["rays_per_frame"] > 0 and s["value"] > 0 and 0 < s["efficiency"] < 1.5
        assert s["max_frame_passes"] == 1
        one = one or s
    assert j["strong"]["spp_16"]["rays_per_frame"] > 10 * j["strong"]["spp_1"]["rays_per_frame"]
