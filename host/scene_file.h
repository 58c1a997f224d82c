// scene_file.h — loader of the flat scene file that `SceneSoup.save_raw()` (ray-tracer_amd/scene.py) writes: the arrays an
// rt_scene_desc points to, the cameras and the optional tone-map settings.  It stands where the reference's SceneParser
// stands (src/SceneParser.cpp; XML + PLY parsing is out of scope, the Python loader produces this file from the same XML).
//
//   "RTSCENE1"                                   8 bytes
//   int32 max_depth, float isect_eps, shadow_eps, background[3], ambient[3], int32 bg_texture, int32 has_tonemap, float tonemap[4]
//   12 arrays, each: uint64 element count, then the elements (ABI structs / packed floats / ints), in the order of LoadScene
//   uint64 camera count; per camera: rt_camera_desc, uint32 name length, name
#ifndef RT_HOST_SCENE_FILE_H_
#define RT_HOST_SCENE_FILE_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_mi355.hpp"

namespace rt_mi355 {

struct LoadedScene {
    SceneView view;
    std::vector<float> vert_pos, vert_uv, sph_center_radius;
    std::vector<int32_t> tri_vert, tri_object, sph_object, prim_shape;
    std::vector<rt_object> objects;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> texels;
};

namespace detail {
inline void ReadExact(FILE *f, void *dst, size_t n, const std::string &path) {
    if (n && std::fread(dst, 1, n, f) != n) throw Error(RT_ERR_INVALID, "truncated scene file " + path);
}
template <typename T>
inline void ReadArray(FILE *f, std::vector<T> &v, size_t per_element, const std::string &path) {
    uint64_t n = 0;
    ReadExact(f, &n, 8, path);
    if (n > (1ull << 34)) throw Error(RT_ERR_TOO_LARGE, "implausible array length in " + path);
    v.resize((size_t)n * per_element);
    ReadExact(f, v.data(), v.size() * sizeof(T), path);
}
}  // namespace detail

inline void LoadScene(const std::string &path, LoadedScene &s) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error(RT_ERR_INVALID, "cannot open " + path);
    try {
        char magic[8];
        detail::ReadExact(f, magic, 8, path);
        if (std::memcmp(magic, "RTSCENE1", 8) != 0) throw Error(RT_ERR_INVALID, path + " is not a scene file");
        rt_scene_desc &d = s.view.desc;
        std::memset(&d, 0, sizeof(d));
        d.abi_version = RT_ABI_VERSION;
        int32_t has_tm = 0;
        detail::ReadExact(f, &d.max_depth, 4, path);
        detail::ReadExact(f, &d.isect_eps, 4, path);
        detail::ReadExact(f, &d.shadow_eps, 4, path);
        detail::ReadExact(f, d.background, 12, path);
        detail::ReadExact(f, d.ambient, 12, path);
        detail::ReadExact(f, &d.bg_texture, 4, path);
        detail::ReadExact(f, &has_tm, 4, path);
        detail::ReadExact(f, s.view.tonemap, 16, path);
        s.view.has_tonemap = has_tm != 0;
        detail::ReadArray(f, s.vert_pos, 3, path);
        detail::ReadArray(f, s.vert_uv, 2, path);
        detail::ReadArray(f, s.tri_vert, 3, path);
        detail::ReadArray(f, s.tri_object, 1, path);
        detail::ReadArray(f, s.sph_center_radius, 4, path);
        detail::ReadArray(f, s.sph_object, 1, path);
        detail::ReadArray(f, s.objects, 1, path);
        detail::ReadArray(f, s.prim_shape, 1, path);
        detail::ReadArray(f, s.materials, 1, path);
        detail::ReadArray(f, s.lights, 1, path);
        detail::ReadArray(f, s.textures, 1, path);
        detail::ReadArray(f, s.texels, 1, path);
        if (s.vert_uv.size() / 2 != s.vert_pos.size() / 3 || s.tri_object.size() != s.tri_vert.size() / 3 || s.sph_object.size() != s.sph_center_radius.size() / 4)
            throw Error(RT_ERR_INVALID, "inconsistent array lengths in " + path);
        d.n_vertices = (int64_t)s.vert_pos.size() / 3; d.vert_pos = s.vert_pos.data(); d.vert_uv = s.vert_uv.data();
        d.n_triangles = (int64_t)s.tri_object.size(); d.tri_vert = s.tri_vert.data(); d.tri_object = s.tri_object.data();
        d.n_spheres = (int64_t)s.sph_object.size(); d.sph_center_radius = s.sph_center_radius.data(); d.sph_object = s.sph_object.data();
        d.n_objects = (int64_t)s.objects.size(); d.objects = s.objects.data();
        d.n_prims = (int64_t)s.prim_shape.size(); d.prim_shape = s.prim_shape.data();
        d.n_materials = (int32_t)s.materials.size(); d.materials = s.materials.data();
        d.n_lights = (int32_t)s.lights.size(); d.lights = s.lights.data();
        d.n_textures = (int32_t)s.textures.size(); d.textures = s.textures.data();
        d.texels = s.texels.empty() ? nullptr : s.texels.data(); d.texels_bytes = (int64_t)s.texels.size();
        uint64_t ncam = 0;
        detail::ReadExact(f, &ncam, 8, path);
        if (ncam > 4096) throw Error(RT_ERR_TOO_LARGE, "implausible camera count in " + path);
        s.view.cameras.resize((size_t)ncam);
        for (CameraView &c : s.view.cameras) {
            uint32_t len = 0;
            detail::ReadExact(f, &c.desc, sizeof(rt_camera_desc), path);
            detail::ReadExact(f, &len, 4, path);
            if (len > 4096) throw Error(RT_ERR_INVALID, "implausible camera name in " + path);
            c.image_name.resize(len);
            detail::ReadExact(f, &c.image_name[0], len, path);
        }
    } catch (...) { std::fclose(f); throw; }
    std::fclose(f);
}

}  // namespace rt_mi355
#endif
