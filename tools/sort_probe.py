"""Upper bound of what ordering secondary rays could win: reflection rays of the config-2 stand-in scene, level by level, traced by
rt_intersect_batch in generation (pixel) order, sorted by a Morton key of the origin + direction octant, and shuffled.
Usage (GPU box): python tools/sort_probe.py"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
rt = importlib.import_module("ray-tracer_amd")
synth = importlib.import_module("ray-tracer_amd.synth")


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def timed(ctx, rays, reps=5):
    d_r = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_h = torch.empty(len(rays) * rt.abi.hit_dtype.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.intersect_device(d_r.data_ptr(), len(rays), d_h.data_ptr())
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.intersect_device(d_r.data_ptr(), len(rays), d_h.data_ptr())
        t.append(time.perf_counter() - t0)
    hits = np.frombuffer(d_h.cpu().numpy().tobytes(), rt.abi.hit_dtype)
    return min(t) * 1e3, hits


def main():
    soup = synth.dragon_standin_scene(1_000_000, 800, 480)
    ctx = rt.Context(0).upload(soup)
    cam = soup.cameras[0]
    nx, ny = cam["nx"], cam["ny"]
    l, r, b, t = [float(x) for x in cam["plane"]]
    g = cam["gaze"] / np.linalg.norm(cam["gaze"])
    u = np.cross(g, cam["up"]); u /= np.linalg.norm(u)
    v = np.cross(u, g)
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    su = l + (r - l) * (ii.reshape(-1) + 0.5) / nx
    sv = t - (t - b) * (jj.reshape(-1) + 0.5) / ny
    d = g[None, :] * float(cam["distance"]) + su[:, None] * u[None, :] + sv[:, None] * v[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(nx * ny, rt.abi.ray_dtype)
    rays["origin"] = cam["position"]
    rays["direction"] = d
    rng = np.random.RandomState(1)
    for level in range(5):
        n = len(rays)
        ms_native, hits = timed(ctx, rays)
        o = rays["origin"].astype(np.float64)
        lo, hi = o.min(0), o.max(0)
        q = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * 1023), 0, 1023).astype(np.int64)
        octant = (rays["direction"][:, 0] < 0) * 1 + (rays["direction"][:, 1] < 0) * 2 + (rays["direction"][:, 2] < 0) * 4
        for name, key in (("morton(origin)>>15+octant", ((morton3(q) >> np.uint64(15)) << np.uint64(3)) | octant.astype(np.uint64)),
                          ("octant+morton(origin)", (octant.astype(np.uint64) << np.uint64(30)) | morton3(q)),
                          ("morton(origin)", morton3(q))):
            perm = np.argsort(key, kind="stable")
            ms_sorted, _ = timed(ctx, np.ascontiguousarray(rays[perm]))
            print("level %d  %8d rays  native %.3f ms  %-28s %.3f ms (%+.1f %%)" % (level, n, ms_native, name, ms_sorted, 100 * (ms_sorted / ms_native - 1)), flush=True)
        ms_shuf, _ = timed(ctx, np.ascontiguousarray(rays[rng.permutation(n)]))
        print("level %d  shuffled %.3f ms (%+.1f %%)   %.0f Mrays/s native" % (level, ms_shuf, 100 * (ms_shuf / ms_native - 1), n / ms_native / 1e3), flush=True)
        hit = hits["shape"] >= 0
        if hit.sum() < 1000:
            break
        dd, nn, ip = rays["direction"][hit].astype(np.float32), hits["n"][hit], hits["ip"][hit]
        nn = nn / np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)
        refl = dd - 2 * (dd * nn).sum(1, keepdims=True) * nn
        nxt = np.zeros(int(hit.sum()), rt.abi.ray_dtype)
        nxt["origin"] = ip + nn * 1e-3 * np.sign(-(dd * nn).sum(1, keepdims=True))
        nxt["direction"] = refl / np.linalg.norm(refl, axis=1, keepdims=True)
        rays = nxt


if __name__ == "__main__":
    main()
