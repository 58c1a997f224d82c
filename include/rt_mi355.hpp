// rt_mi355.hpp — C++11 host side over the C-ABI of librt_mi355.so (include/rt_mi355.h), header only.
//
// The reference's host is C++ (src/main.cpp:15-30: parse a scene, `new DefaultRenderer()`, RenderSceneIntoPPM, delete), so
// this is the same shape in the same language: two classes with the names and call signatures of the reference's plugin
// interfaces, working on a flattened scene (SceneView: the content of the reference's `Scene` after parsing) instead of the
// reference's own class graph, which cannot be linked here (a maintainer adds the same two subclasses on top of the reference's headers).
//
//   rt_mi355::Mi355Renderer      <->  RenderStrategy        (src/RenderStrategy.h:12-26, DefaultRenderer.cpp:35-162)
//   rt_mi355::Mi355Accelerator   <->  AccelerationStructure (src/AccelerationStructure.h:12-26, BVHTree.cpp:155-206)
//
// Errors: the reference's interfaces return void and never report anything; here every failed ABI call throws
// rt_mi355::Error (code = rt_status, what() = rt_last_error).  There is no CPU fallback.
#ifndef RT_MI355_HPP_
#define RT_MI355_HPP_

#include <atomic>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "rt_mi355.h"

namespace rt_mi355 {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &msg) : std::runtime_error("rt_mi355 error " + std::to_string(c) + ": " + msg), code(c) {}
};

struct CameraView {
    rt_camera_desc desc;
    std::string image_name;          // Camera::GetImageName()
};

// What the reference's Scene holds after parsing: the description arrays, the cameras and the optional <Tonemap>.
struct SceneView {
    rt_scene_desc desc;
    std::vector<CameraView> cameras;
    bool has_tonemap = false;
    float tonemap[4] = {0, 0, 0, 0}; // key, burn percent, saturation, gamma (SceneParser.cpp:162-181)
};

class Context {
public:
    explicit Context(int hip_device = 0) { const int rc = rt_ctx_create(hip_device, &h_); if (rc != RT_OK) throw Error(rc, rt_last_error(nullptr)); }
    ~Context() { rt_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    void Upload(const rt_scene_desc &scene) { Check(rt_scene_upload(h_, &scene)); }
    rt_bvh_info BvhInfo() const { rt_bvh_info i; Check(rt_scene_bvh_info(h_, &i)); return i; }

    // float frame [ny*nx*3] on the host
    std::vector<float> Render(const rt_camera_desc &cam, unsigned seed = 0, rt_stats *stats = nullptr) {
        rt_render_opts o = rt_render_opts();
        o.abi_version = RT_ABI_VERSION; o.seed = seed; o.timing = stats ? 1 : 0;
        std::vector<float> hdr((size_t)3 * cam.nx * cam.ny);
        Check(rt_render(h_, &cam, &o, hdr.data(), nullptr, nullptr, stats));
        return hdr;
    }
    // One rank's share of a frame: the interleaved tiles with (tile id % tile_world) == tile_rank, written in compact form (3 floats
    // per owned pixel, rt_tile_pixel_order's order) into DEVICE memory of this context's GPU.  What one of the reference's row-band
    // threads does (src/DefaultRenderer.cpp:65-71), with a GPU in the thread's place.
    void RenderTiles(const rt_camera_desc &cam, int tile_size, int tile_rank, int tile_world, float *d_compact_rgb, unsigned seed = 0, rt_stats *stats = nullptr) {
        rt_render_opts o = rt_render_opts();
        o.abi_version = RT_ABI_VERSION; o.seed = seed; o.timing = stats ? 1 : 0;
        o.tile_size = tile_size; o.tile_rank = tile_rank; o.tile_world = tile_world; o.output_on_device = 1; o.compact_output = 1;
        Check(rt_render(h_, &cam, &o, d_compact_rgb, nullptr, nullptr, stats));
    }
    void Intersect(const rt_ray *rays, size_t n, rt_hit *hits) const { Check(rt_intersect_batch(h_, rays, n, hits, 0, nullptr)); }

    std::vector<unsigned char> EncodeImage(const float *rgb, int nx, int ny, int format) {
        std::vector<unsigned char> out(rt_image_encode_bound(nx, ny, format));
        size_t n = 0;
        Check(rt_image_encode(h_, rgb, nx, ny, 0, format, out.data(), out.size(), &n, nullptr));
        out.resize(n);
        return out;
    }
    std::vector<float> Tonemap(const float *rgb, int nx, int ny, const float tm[4], bool input_8bit = true) {
        rt_tonemap_opts o = rt_tonemap_opts();
        o.abi_version = RT_ABI_VERSION; o.key = tm[0]; o.burn_percent = tm[1]; o.saturation = tm[2]; o.gamma = tm[3]; o.input_8bit = input_8bit ? 1 : 0;
        std::vector<float> out((size_t)3 * nx * ny);
        Check(rt_tonemap(h_, rgb, nx, ny, &o, out.data(), nullptr, nullptr));
        return out;
    }
    // body of a binary PLY (layout from the caller's header parse) -> positions [nv*3] and triangle indices [nt*3]
    void DecodePly(const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, rt_ply_layout layout, std::vector<float> &pos,
                   std::vector<int32_t> &tris) {
        layout.abi_version = RT_ABI_VERSION;
        pos.resize((size_t)layout.n_vertices * 3);
        tris.resize((size_t)layout.n_faces * 6);
        int64_t n = 0;
        Check(rt_ply_decode(h_, vertex_bytes, face_bytes, face_bytes_size, &layout, pos.data(), tris.data(), &n, nullptr));
        tris.resize((size_t)n * 3);
    }
    // the same decoder with its output left in DEVICE buffers of the caller (d_pos [nv*3] floats, d_tris [<= 2*nf*3] ints, vertex
    // indices + index_offset); returns the number of triangles.  With rt_scene_desc.geometry_on_device the mesh never visits the host.
    int64_t DecodePlyTo(const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, rt_ply_layout layout, float *d_pos, int32_t *d_tris,
                        int32_t index_offset = 0) {
        layout.abi_version = RT_ABI_VERSION;
        int64_t n = 0;
        Check(rt_ply_decode_to(h_, vertex_bytes, face_bytes, face_bytes_size, &layout, d_pos, d_tris, index_offset, &n, nullptr));
        return n;
    }
    // One build, many GPUs: the uploaded scene as an opaque header + RT_SCENE_ARENAS byte arrays (host copies here; a multi-GPU host
    // passes device buffers to rt_scene_export_arena / rt_scene_import directly and moves them with its own broadcast).
    struct SceneBytes { std::vector<unsigned char> header; std::vector<std::vector<unsigned char> > arenas; };
    SceneBytes ExportScene() {
        SceneBytes b;
        b.header.resize(rt_scene_header_bytes());
        uint64_t sizes[RT_SCENE_ARENAS];
        Check(rt_scene_export(h_, b.header.data(), sizes));
        b.arenas.resize(RT_SCENE_ARENAS);
        for (int i = 0; i < RT_SCENE_ARENAS; ++i) {
            b.arenas[i].resize((size_t)sizes[i]);
            if (sizes[i]) Check(rt_scene_export_arena(h_, i, b.arenas[i].data()));
        }
        return b;
    }
    void ImportScene(const SceneBytes &b) {
        const void *ptrs[RT_SCENE_ARENAS];
        for (int i = 0; i < RT_SCENE_ARENAS; ++i) ptrs[i] = (i < (int)b.arenas.size() && !b.arenas[i].empty()) ? b.arenas[i].data() : nullptr;
        Check(rt_scene_import(h_, b.header.data(), ptrs));
    }
    rt_ctx *handle() const { return h_; }

private:
    void Check(int rc) const { if (rc != RT_OK) throw Error(rc, rt_last_error(h_)); }
    rt_ctx *h_ = nullptr;
};

// AccelerationStructure: built from a scene's primitives, answers closest-hit queries.  The reference takes one ray per
// call; a GPU needs batches, so Intersect takes n rays ("no hit" is shape == -1 where the reference has mat == nullptr).
class Mi355Accelerator {
public:
    explicit Mi355Accelerator(const SceneView &scene, int hip_device = 0) : ctx_(hip_device) { ctx_.Upload(scene.desc); }
    // from a scene another context has already built (the reference's renderer and accelerator share one BVHTree, DefaultRenderer.cpp:29-33)
    explicit Mi355Accelerator(const Context::SceneBytes &built, int hip_device = 0) : ctx_(hip_device) { ctx_.ImportScene(built); }
    void Intersect(const rt_ray *rays, size_t n, rt_hit *out) const { ctx_.Intersect(rays, n, out); }
    void Intersect(const rt_ray &ray, rt_hit &out) const { ctx_.Intersect(&ray, 1, &out); }      // drop-in form: one launch per ray
private:
    Context ctx_;
};

// RenderStrategy: RenderSceneIntoPPM renders every camera of the scene and writes camera.image_name — the P3 text of
// Image::SaveImageAsPPM, or with <Tonemap> the tone-mapped frame as HALF EXR (Image::SaveImage, src/Image.cpp:38-52).
class Mi355Renderer {
public:
    // frames_in_flight = 2: the cameras of a scene are rendered by two contexts that share one built scene (ExportScene / ImportScene),
    // one host thread each, so that one context's launch tails overlap the other's launches.  The files are the same either way.
    explicit Mi355Renderer(int hip_device = 0, const std::string &out_dir = ".", int frames_in_flight = 1)
        : ctx_(hip_device), device_(hip_device), out_dir_(out_dir), in_flight_(frames_in_flight < 1 ? 1 : frames_in_flight) {}
    void RenderSceneIntoPPM(const SceneView &scene) {
        ctx_.Upload(scene.desc);                                   // RetrieveRenderingParamsFromScene: builds the accelerator
        const size_t n_ctx = std::min<size_t>((size_t)in_flight_, scene.cameras.size());
        if (n_ctx <= 1) {
            for (const CameraView &cam : scene.cameras) {
                const std::vector<float> hdr = ctx_.Render(cam.desc);
                SaveImage(ctx_, scene, cam, hdr.data(), out_dir_);
            }
            return;
        }
        std::vector<std::unique_ptr<Context> > extra;
        {
            const Context::SceneBytes built = ctx_.ExportScene();
            for (size_t i = 1; i < n_ctx; ++i) { extra.emplace_back(new Context(device_)); extra.back()->ImportScene(built); }
        }
        std::atomic<size_t> next(0);
        std::vector<std::string> failed(n_ctx);
        std::vector<int> code(n_ctx, RT_OK);
        std::vector<std::thread> workers;
        for (size_t w = 0; w < n_ctx; ++w)
            workers.emplace_back([&, w] {
                Context &ctx = w == 0 ? ctx_ : *extra[w - 1];
                try {
                    for (size_t i = next++; i < scene.cameras.size(); i = next++) {
                        const std::vector<float> hdr = ctx.Render(scene.cameras[i].desc);
                        SaveImage(ctx, scene, scene.cameras[i], hdr.data(), out_dir_);
                    }
                } catch (const Error &e) { code[w] = e.code; failed[w] = e.what(); }
            });
        for (std::thread &t : workers) t.join();
        for (size_t w = 0; w < n_ctx; ++w) if (code[w] != RT_OK) throw Error(code[w], failed[w]);
    }
    // Image::SaveImage for a finished float frame on the host (src/Image.cpp:38-52): P3 text, or HALF EXR after the tone mapper
    static void SaveImage(Context &ctx, const SceneView &scene, const CameraView &cam, const float *hdr, const std::string &out_dir) {
        std::vector<unsigned char> file;
        if (!scene.has_tonemap) file = ctx.EncodeImage(hdr, cam.desc.nx, cam.desc.ny, RT_IMG_PPM_P3);
        else {
            const std::vector<float> mapped = ctx.Tonemap(hdr, cam.desc.nx, cam.desc.ny, scene.tonemap);
            file = ctx.EncodeImage(mapped.data(), cam.desc.nx, cam.desc.ny, RT_IMG_EXR_HALF);
        }
        const std::string path = out_dir + "/" + cam.image_name;
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw Error(RT_ERR_INVALID, "cannot open " + path);
        const size_t w = std::fwrite(file.data(), 1, file.size(), f);
        std::fclose(f);
        if (w != file.size()) throw Error(RT_ERR_INVALID, "short write to " + path);
    }
    // One rank of a multi-GPU job (one process per GPU): the scene must be uploaded or imported already; renders this rank's tiles of
    // one camera into the caller's device buffer.  The gather that follows is rt_gather_tiles (rt_mi355_mgpu.h).
    void RenderCameraTiles(const CameraView &cam, int tile_size, int tile_rank, int tile_world, float *d_compact_rgb) {
        ctx_.RenderTiles(cam.desc, tile_size, tile_rank, tile_world, d_compact_rgb);
    }
    Context &context() { return ctx_; }
private:
    Context ctx_;
    int device_;
    std::string out_dir_;
    int in_flight_;
};

}  // namespace rt_mi355
#endif
