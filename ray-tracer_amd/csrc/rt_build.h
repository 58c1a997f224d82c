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

// Returns 0 or a negative rt_status; `err` receives a message.
int rt_build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err);

#endif
