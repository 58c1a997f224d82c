"""Builds tests/golden/:
  ref_png_<scene>.npz   the author's published 8-bit render of a deterministic 1-spp scene (reference src/results/<scene>.png,
                        decoded with Pillow; DATA the reference holds, from an older revision of its sources) — the only
                        reference-produced outputs that exist for this path; used as a loose sanity bound on the oracle.
  oracle_frames.json    per-scene digests of the oracle's own frames (regression guard for the oracle and the loader).
usage: python tools/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
from oracle import orc  # noqa: E402

PNG_SCENES = ["cornellbox", "cornellbox_recursive", "simple", "spheres", "bunny", "scienceTree", "cube_point", "cube_directional",
              "brdf_blinnphong_original", "brdf_blinnphong_modified", "brdf_phong_original", "brdf_phong_modified", "brdf_torrancesparrow"]
DIGEST_SCENES = PNG_SCENES + ["sphere_perlin", "sphere_perlin_bump", "cube_perlin", "cube_perlin_bump"]


def main():
    from PIL import Image
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    digests = {}
    for name in DIGEST_SCENES:
        soup = rt.SceneSoup.load(os.path.join(ROOT, "scenes", name + ".npz"))
        hdr, pid, pt, st = orc.OracleScene(soup).render(0, nthreads=8, dynamic=True)
        img8 = rt.renderer.resolve_8bit(hdr)
        entry = {"shape": list(img8.shape), "sha256_8bit": hashlib.sha256(img8.tobytes()).hexdigest(),
                 "sha256_prim_id": hashlib.sha256(pid.tobytes()).hexdigest(), "hdr_sum": float(np.float64(hdr.astype(np.float64).sum())),
                 "rays": [int(x) for x in st.rays]}
        if name in PNG_SCENES:
            ref = np.asarray(Image.open("/root/reference/src/results/%s.png" % name).convert("RGB"), np.uint8)
            np.savez_compressed(os.path.join(gold, "ref_png_%s.npz" % name), rgb=ref)
            d = np.abs(img8.astype(int) - ref.astype(int)).max(axis=2)
            entry["png_differing_pixels"] = int((d > 0).sum())
            entry["png_within_1"] = float((d <= 1).mean())
            entry["png_max_abs"] = int(d.max())
        digests[name] = entry
        print(name, entry)
    json.dump(digests, open(os.path.join(gold, "oracle_frames.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
