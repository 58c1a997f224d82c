// rt_build.h — host-side scene flattening for the MI355X render path.
#ifndef RT_BUILD_H_
#define RT_BUILD_H_

#include <string>
#include <vector>

#include "rt_device.h"

struct HostScene {
    DevSceneParams params;
    std::vector<DevNode4> nodes;       // 4-wide collapse of the reference tree, preorder
    std::vector<DevChain> chain;       // one per reference node (index = reference node number, preorder of the build)
    std::vector<DevLeaf> leaves;       // DFS (left-to-right) order
    std::vector<DevPrim> prims;        // leaf order, duplicates inside one leaf removed
    std::vector<DevObject> objects;
    std::vector<DevVertex> verts;
    std::vector<int32_t> tri_vert;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_texture> textures;
    std::vector<uint8_t> texels;
    // reference-topology view kept for rt_host_bvh_* (tests compare it with the oracle's build)
    std::vector<int32_t> ref_prim_order;   // shape id of every reference primitive in post-partition order
    std::vector<int64_t> ref_leaf_range;   // [2*n_leaves] start,end into ref_prim_order
    std::vector<float> leaf_boxes;         // [6*n_leaves]
    rt_bvh_info info;
    float scene_scale;                     // max |coordinate| of the root box
};

// One node of the reference-topology tree as a builder hands it to the flattener.
struct RtBuildNode {
    float lo[3], hi[3];
    int32_t left, right;      // node indices, -1 for a leaf
    int32_t start, end;       // range in the primitive order
    int32_t depth;            // root = 1
};

// The reference's build on the GPU (rt_bvh_gpu.hip): same topology, order and boxes as the host builder.  pbox = n boxes of
// 6 floats (lo, hi).  Needs a current HIP device.
bool rt_bvh_gpu_build(const float *pbox, int64_t n, std::vector<int32_t> &order, std::vector<RtBuildNode> &nodes, int &max_depth, std::string &err);

// Returns 0 or a negative rt_status; `err` receives a message.  gpu_build: build the tree with rt_bvh_gpu_build when the scene
// has at least RT_GPU_BUILD_MIN_PRIMS primitives (the flattening stays on the host); the result is the same either way.
#define RT_GPU_BUILD_MIN_PRIMS 4096
int rt_build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err, int gpu_build = 0 /* 0 host, 1 auto, 2 always */);

#endif
