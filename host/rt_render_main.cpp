// rt_render — the reference's main (src/main.cpp:15-30) over the MI355X path, in the reference's language:
//   parse the scene (here: the flat scene file, host/scene_file.h), create the renderer, RenderSceneIntoPPM, report the time.
// usage: rt_render <scene.rtscene> [out_dir] [--rays <rays.bin> <hits.bin>] [--shared-scene] [--ranks N] [--tile T]
//   --in-flight K: the cameras of the scene are rendered by K contexts sharing one built scene, one host thread each (default 1).
//   --ranks N: render every frame on GPUs 0..N-1 of this process (interleaved T x T tiles, default 32; scene built on GPU 0 and broadcast,
//              tiles gathered on GPU 0 over RCCL: include/rt_mi355_mgpu.hpp) instead of on GPU 0 alone.  Same files either way.
//   --rays: also answer a batch of AccelerationStructure::Intersect queries (rt_ray records in, rt_hit records out).
//   --shared-scene: the accelerator of --rays adopts the scene the renderer has built (rt_scene_export / rt_scene_import) instead of building it again.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene_file.h"
#include "rt_mi355_mgpu.hpp"

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <scene.rtscene> [out_dir] [--rays rays.bin hits.bin] [--shared-scene] [--in-flight K] [--ranks N] [--tile T]\n", argv[0]); return 2; }
    try {
        rt_mi355::LoadedScene scene;
        rt_mi355::LoadScene(argv[1], scene);
        std::string out_dir = ".";
        const char *rays_path = nullptr, *hits_path = nullptr;
        bool shared_scene = false;
        int ranks = 0, tile = 32, in_flight = 1;
        for (int i = 2; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--rays") && i + 2 < argc) { rays_path = argv[i + 1]; hits_path = argv[i + 2]; i += 2; }
            else if (!std::strcmp(argv[i], "--shared-scene")) shared_scene = true;
            else if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = std::atoi(argv[++i]);
            else if (!std::strcmp(argv[i], "--tile") && i + 1 < argc) tile = std::atoi(argv[++i]);
            else if (!std::strcmp(argv[i], "--in-flight") && i + 1 < argc) in_flight = std::atoi(argv[++i]);
            else out_dir = argv[i];
        }
        if (ranks > 0) {
            std::vector<int> devices;
            for (int r = 0; r < ranks; ++r) devices.push_back(r);
            rt_mi355::Mi355NodeRenderer node(devices, out_dir, tile);
            const auto t0 = std::chrono::steady_clock::now();
            node.RenderSceneIntoPPM(scene.view);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const rt_bvh_info info = node.context(ranks - 1).BvhInfo();      // the last rank's scene: received, never built there (ranks > 1)
            std::printf("rendered %zu camera(s) of %s on %d rank(s), %dx%d tiles, in %.1f ms (upload + broadcast incl.; %lld BVH nodes, depth %d)\n",
                        scene.view.cameras.size(), argv[1], ranks, tile, tile, ms, (long long)info.n_nodes, info.max_depth);
            return 0;
        }
        rt_mi355::Mi355Renderer renderer(0, out_dir, in_flight);                   // main.cpp:20  new DefaultRenderer()
        const auto t0 = std::chrono::steady_clock::now();
        renderer.RenderSceneIntoPPM(scene.view);                          // main.cpp:22
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const rt_bvh_info info = renderer.context().BvhInfo();
        std::printf("rendered %zu camera(s) of %s in %.1f ms (upload incl.; %lld BVH nodes, depth %d)\n", scene.view.cameras.size(), argv[1], ms,
                    (long long)info.n_nodes, info.max_depth);
        if (rays_path) {
            FILE *f = std::fopen(rays_path, "rb");
            if (!f) throw rt_mi355::Error(RT_ERR_INVALID, std::string("cannot open ") + rays_path);
            std::fseek(f, 0, SEEK_END);
            const long bytes = std::ftell(f);
            std::fseek(f, 0, SEEK_SET);
            std::vector<rt_ray> rays((size_t)bytes / sizeof(rt_ray));
            if (std::fread(rays.data(), sizeof(rt_ray), rays.size(), f) != rays.size()) { std::fclose(f); throw rt_mi355::Error(RT_ERR_INVALID, "short read"); }
            std::fclose(f);
            std::vector<rt_hit> hits(rays.size());
            if (shared_scene) {
                rt_mi355::Mi355Accelerator accel(renderer.context().ExportScene());
                accel.Intersect(rays.data(), rays.size(), hits.data());
            } else {
                rt_mi355::Mi355Accelerator accel(scene.view);             // AccelerationStructureFactory::CreateAccelerationStructure
                accel.Intersect(rays.data(), rays.size(), hits.data());
            }
            f = std::fopen(hits_path, "wb");
            if (!f || std::fwrite(hits.data(), sizeof(rt_hit), hits.size(), f) != hits.size()) throw rt_mi355::Error(RT_ERR_INVALID, std::string("cannot write ") + hits_path);
            std::fclose(f);
            std::printf("answered %zu intersection queries\n", rays.size());
        }
    } catch (const rt_mi355::Error &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
