"""profiles/traffic_latest.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh (run on the GPU box, merged back into gpurun_out/).
The file records the SHA-256 of the librt_mi355.so the counters were taken with; bench.py reports `roofline.traffic` only when that equals the
library it is running, null otherwise.  usage: python tools/traffic_json.py gpurun_out/<pmc_dir> [frames=2]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402


def lib_sha256():
    return hashlib.sha256(open(os.path.join(ROOT, "ray-tracer_amd", "csrc", "librt_mi355.so"), "rb").read()).hexdigest()


if __name__ == "__main__":
    pmc_dir = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    out, calls = pmc_summary.load(pmc_dir)
    tr = out["k_trace"]
    launches = len(calls[("k_trace", "FETCH_SIZE")])
    fetch_raw = tr["FETCH_SIZE"] * 1024.0          # gfx950: KiB
    write = tr["WRITE_SIZE"] * 1024.0
    fetch = 2.0 * fetch_raw                         # MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per request although wide requests move 128 B
    d = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each, tools/pmc_passes.sh) over tools/trace_driver.py: %d frames of config 2, %d k_trace_generic launches" % (frames, launches),
         "unit_note": "gfx950 reports FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE = TCC_EA0_RDREQ x 64 B although wide requests are 128 B, so it is doubled (an upper bound: the 128-byte node lines are such requests, the 64-byte primitive records may not be); WRITE_SIZE is exact",
         "lib_sha256": lib_sha256(),
         "fetch_bytes_per_frame_raw": int(fetch_raw / frames), "fetch_bytes_per_frame": int(fetch / frames), "write_bytes_per_frame": int(write / frames),
         "hbm_bytes_per_trace_launch": int((fetch + write) / max(1, launches))}
    if tr.get("TCC_HIT", 0) + tr.get("TCC_MISS", 0) > 0:
        d["tcc_hit_rate"] = round(tr["TCC_HIT"] / (tr["TCC_HIT"] + tr["TCC_MISS"]), 4)
    json.dump(d, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(d, indent=1))
