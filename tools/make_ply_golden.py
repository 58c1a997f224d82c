"""Builds tests/golden/ref_ply_digests.json: what the oracle's PLY reader (orc_ply_read, the restatement of happly + the PLY branch of
the reference's scene parser, SceneParser.cpp:983-1023) decodes from the four PLY files the reference ships under src/scenes/ply/ —
vertex and triangle counts and SHA-256 digests of the decoded float32 positions / int32 triangles.  Digests are data about the
reference's own input files; no file text is stored.  Runs only where /root/reference is readable (the build container).
usage: python tools/make_ply_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

out = {}
for name in ("cube", "plane", "tap_tap", "tap_water_200"):
    data = open("/root/reference/src/scenes/ply/%s.ply" % name, "rb").read()
    pos, tris = orc.ply_read(data)
    out[name] = {"file_bytes": len(data), "n_vertices": int(len(pos)), "n_triangles": int(len(tris)),
                 "sha256_positions_f32": hashlib.sha256(pos.tobytes()).hexdigest(), "sha256_triangles_i32": hashlib.sha256(tris.tobytes()).hexdigest(),
                 # the three corner positions of every triangle, in face order: does not depend on how a loader numbers the vertices
                 "sha256_triangle_corners_f32": hashlib.sha256(pos[tris].tobytes()).hexdigest()}
    print(name, out[name])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ref_ply_digests.json"), "w"), indent=1, sort_keys=True)
