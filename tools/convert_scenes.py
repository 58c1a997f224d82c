"""Converts reference scene files (XML + PLY + textures, read as DATA from /root/reference/src/scenes) into the
repository's own numpy scene fixtures under scenes/*.npz, so that GPU tests / bench can run where /root/reference does
not exist.  The fixtures hold only inputs (geometry, materials, lights, cameras), never reference source code.

usage: python tools/convert_scenes.py [scene ...]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")

REF = "/root/reference/src/scenes"
DEFAULT = ["cornellbox", "cornellbox_recursive", "scienceTree", "bunny", "spheres", "simple", "cornellbox_brushed_metal", "spheres_dof",
           "cornellbox_boxes_dynamic", "cornellbox_area", "metal_glass_plates", "cube_point", "cube_directional",
           "sphere_perlin", "sphere_perlin_bump", "sphere_perlin_scale", "cube_perlin", "cube_perlin_bump",
           "brdf_blinnphong_original", "brdf_blinnphong_modified", "brdf_blinnphong_modified_normalized", "brdf_phong_original",
           "brdf_phong_modified", "brdf_phong_modified_normalized", "brdf_torrancesparrow"]


def main():
    names = sys.argv[1:] or DEFAULT
    out_dir = os.path.join(ROOT, "scenes")
    os.makedirs(out_dir, exist_ok=True)
    for n in names:
        src = os.path.join(REF, n + ".xml")
        try:
            soup = rt.load_xml(src, scene_dir=REF, perlin_tables=rt.renderer.perlin_tables)
        except Exception as e:  # missing blobs, EXR textures ...
            print("skip %-40s %s: %s" % (n, type(e).__name__, e))
            continue
        dst = os.path.join(out_dir, n + ".npz")
        soup.save(dst)
        print("%-40s prims=%-8d tris=%-8d spheres=%-3d lights=%d textures=%d depth=%d cams=%s -> %.1f KB" % (
            n, len(soup.prim_shape), len(soup.tri_vert), len(soup.sph_object), len(soup.lights), len(soup.textures), soup.max_depth,
            [(c["nx"], c["ny"], c["num_samples"]) for c in soup.cameras], os.path.getsize(dst) / 1024))


if __name__ == "__main__":
    main()
