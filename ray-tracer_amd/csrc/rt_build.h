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
    std::vector<int32_t> node_chain;   // [4 * nodes.size()]: DevChain index of each child of each wide node
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
    std::vector<uint32_t> cone_words;      // per child of every wide node: packed normal cone (rt_cone.h) ...
    std::vector<int64_t> cone_ranges;      // ... and the [start, end) range of that child's subtree in ref_prim_order
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

// The traversal arrays as the device flattener leaves them (device memory owned by the caller) + what the host needs to know.
struct RtGpuScene {
    DevNode4 *nodes = nullptr;
    DevChain *chain = nullptr;
    int32_t *node_chain = nullptr;     // [4 * n_wide]
    DevLeaf *leaves = nullptr;
    DevPrim *prims = nullptr;
    int64_t n_wide = 0, n_nodes = 0, n_leaves = 0, n_leaf_prims = 0;
    int32_t root = 0;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    int max_depth = 0, stack_bound = 0;
    DevVertex *verts = nullptr;        // set when the scene was ingested on the device (rt_ingest.hip), else null
    int32_t *tri_vert = nullptr;
    int64_t n_verts = 0, n_tris = 0;
};
// Build and flatten on the GPU: shape_rec = one DevPrim per shape, pshape[i] = shape of reference primitive i; host arrays, or
// device arrays (left untouched, still owned by the caller) when inputs_on_device is set.
// objects: the object table (host), read for the normal cones of rt_cone.h.
bool rt_bvh_gpu_build_flatten(const float *pbox, int64_t n, const DevPrim *shape_rec, int64_t n_shapes, const int32_t *pshape, const std::vector<DevObject> &objects,
                              RtGpuScene &out, std::string &err, bool inputs_on_device = false);

// Shape setup on the device (rt_ingest.hip): triangle records, smooth vertex normals, primitive world boxes — all device memory.
struct RtGpuIngest {
    float *pbox = nullptr;             // [6 * n_prims]
    DevPrim *shape_rec = nullptr;      // [n_triangles + n_spheres]
    int32_t *pshape = nullptr;         // [n_prims]
    DevVertex *verts = nullptr;        // [n_vertices]
    int32_t *tri_vert = nullptr;       // [3 * n_triangles]
    void release();
};
// RT_OK, a negative rt_status, or 1 = "use the host loops" (a vertex shared by two smooth objects).  The description must have
// been range-checked; sphere_rec holds the sphere records (host).
// n_tris_total / n_prims_total: triangle and primitive counts after the expansion of rt_instance / rt_prim_run records, which
// happens here, in HBM; with geometry_on_device the vertex and triangle arrays of the description are device memory already.
int rt_ingest_gpu(const rt_scene_desc *d, int64_t n_tris_total, int64_t n_prims_total, const std::vector<DevObject> &objects, const std::vector<DevPrim> &sphere_rec,
                  RtGpuIngest &out, std::string &err);

// Returns 0 or a negative rt_status; `err` receives a message.  gpu_build: build the tree with rt_bvh_gpu_build when the scene
// has at least RT_GPU_BUILD_MIN_PRIMS primitives (the flattening stays on the host); the result is the same either way.
#define RT_GPU_BUILD_MIN_PRIMS 4096
// device_out != nullptr (and the scene large enough): build + flatten on the GPU; nodes / chain / leaves / prims of `out` stay
// empty and *device_out receives the device arrays.
int rt_build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err, int gpu_build = 0 /* 0 host, 1 auto, 2 always */,
                        RtGpuScene *device_out = nullptr);

#endif
