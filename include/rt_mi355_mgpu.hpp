// rt_mi355_mgpu.hpp — the RenderStrategy of rt_mi355.hpp over several GPUs of one process, header only, on top of the C-ABI of
// librt_mi355_mgpu.so (include/rt_mi355_mgpu.h).
//
// The reference's RenderCamera starts eight threads over row bands of the frame and joins them (src/DefaultRenderer.cpp:65-75).
// Mi355NodeRenderer keeps that shape with a GPU behind every thread: thread r renders the interleaved tiles rank r owns
// (Context::RenderTiles), the join is followed by rt_gather_tiles (RCCL send / receive to rank 0 over xGMI + de-interleave), and the
// scene all "threads" read is built once on rank 0 and broadcast (rt_scene_broadcast), the counterpart of the one Scene / BVHTree the
// reference's threads share (DefaultRenderer.cpp:29-33).  The files it writes are byte-identical to Mi355Renderer's.
#ifndef RT_MI355_MGPU_HPP_
#define RT_MI355_MGPU_HPP_

#include <memory>
#include <thread>

#include "rt_mi355.hpp"
#include "rt_mi355_mgpu.h"

namespace rt_mi355 {

class Mi355NodeRenderer {
public:
    // rank r renders on devices[r]; rank 0 gathers and writes the files
    explicit Mi355NodeRenderer(const std::vector<int> &devices, const std::string &out_dir = ".", int tile_size = 32) : out_dir_(out_dir), tile_(tile_size) {
        if (devices.empty()) throw Error(RT_ERR_INVALID, "Mi355NodeRenderer: no devices");
        for (int d : devices) ctx_.emplace_back(new Context(d));
        const int rc = rt_gather_create(devices.data(), (int)devices.size(), &g_);
        if (rc != RT_OK) throw Error(rc, rt_gather_last_error(nullptr));
    }
    ~Mi355NodeRenderer() { rt_gather_destroy(g_); }
    Mi355NodeRenderer(const Mi355NodeRenderer &) = delete;
    Mi355NodeRenderer &operator=(const Mi355NodeRenderer &) = delete;

    void RenderSceneIntoPPM(const SceneView &scene) {
        const int world = (int)ctx_.size();
        ctx_[0]->Upload(scene.desc);                                 // one build ...
        std::vector<rt_ctx *> handles;
        for (auto &c : ctx_) handles.push_back(c->handle());
        Check(rt_scene_broadcast(g_, handles.data()));                // ... every GPU gets the arenas
        for (const CameraView &cam : scene.cameras) {
            std::vector<float *> d_tiles((size_t)world, nullptr);
            float *d_frame = nullptr;
            Check(rt_gather_buffers(g_, cam.desc.nx, cam.desc.ny, tile_, d_tiles.data(), &d_frame));
            std::vector<std::string> failed((size_t)world);
            std::vector<int> code((size_t)world, RT_OK);
            std::vector<std::thread> workers;
            for (int r = 0; r < world; ++r)                           // DefaultRenderer.cpp:65-71, a GPU per thread
                workers.emplace_back([&, r] {
                    try { ctx_[(size_t)r]->RenderTiles(cam.desc, tile_, r, world, d_tiles[(size_t)r]); }
                    catch (const Error &e) { code[(size_t)r] = e.code; failed[(size_t)r] = e.what(); }
                });
            for (std::thread &t : workers) t.join();                  // DefaultRenderer.cpp:73-75
            for (int r = 0; r < world; ++r) if (code[(size_t)r] != RT_OK) throw Error(code[(size_t)r], "rank " + std::to_string(r) + ": " + failed[(size_t)r]);
            Check(rt_gather_tiles(g_, cam.desc.nx, cam.desc.ny, tile_, d_tiles.data(), d_frame));
            std::vector<float> hdr((size_t)3 * cam.desc.nx * cam.desc.ny);
            Check(rt_gather_copy_frame(g_, hdr.data()));
            Mi355Renderer::SaveImage(*ctx_[0], scene, cam, hdr.data(), out_dir_);
        }
    }
    Context &context(int rank = 0) { return *ctx_[(size_t)rank]; }
    int ranks() const { return (int)ctx_.size(); }

private:
    void Check(int rc) const { if (rc != RT_OK) throw Error(rc, rt_gather_last_error(g_)); }
    std::vector<std::unique_ptr<Context> > ctx_;
    rt_gather *g_ = nullptr;
    std::string out_dir_;
    int tile_;
};

}  // namespace rt_mi355
#endif
