"""Times the frame post-processing kernels (rt_image_encode P3, rt_tonemap) on device-resident frames and the oracle's C++
restatements on the host.  Usage: python3 tools/image_bench.py [nx ny]..."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
orc = importlib.import_module("oracle.orc")
abi = rt.abi


def main():
    sizes = [(800, 480), (3840, 2160)]
    ctx = rt.Context(0)
    for nx, ny in sizes:
        g = torch.Generator(device="cuda:0").manual_seed(1)
        f = torch.rand((ny, nx, 3), device="cuda:0", generator=g) ** 3 * 400.0
        ptr = int(f.data_ptr())
        best_p3, best_tm, nbytes = 1e9, 1e9, 0
        for _ in range(5):
            b, ms = ctx.encode_image(ptr, abi.RT_IMG_PPM_P3, nx=nx, ny=ny, timing=True)
            best_p3, nbytes = min(best_p3, ms), len(b)
            _, _, ms = ctx.tonemap(ptr, 0.18, 1.0, 1.0, nx=nx, ny=ny, timing=True)
            best_tm = min(best_tm, ms)
        npx = nx * ny
        # algorithmic bytes: resolve reads 12 B/px, writes 3; measure + write read the 8-bit image once each; text written once
        p3_bytes = npx * (12 + 3 + 3 + 3) + nbytes
        # tone mapping (8-bit input): resolve 12+3, luminance 3+4, three select passes x 4, apply 3+12
        tm_bytes = npx * (12 + 3 + 3 + 4 + 3 * 4 + 3 + 12)
        host = f.cpu().numpy()
        img8 = orc.resolve_8bit(host)
        t0 = time.perf_counter(); txt = orc.ppm_p3(img8); t1 = time.perf_counter()
        orc.tonemap(img8.astype(np.float32), 0.18, 1.0, 1.0); t2 = time.perf_counter()
        assert txt == b.tobytes()
        print("%dx%d: P3 text %d bytes in %.3f ms (%.0f GB/s algorithmic; CPU sprintf %.1f ms); tonemap %.3f ms (%.0f GB/s; CPU %.1f ms)"
              % (nx, ny, nbytes, best_p3, p3_bytes / best_p3 / 1e6, (t1 - t0) * 1e3, best_tm, tm_bytes / best_tm / 1e6, (t2 - t1) * 1e3))


if __name__ == "__main__":
    main()
