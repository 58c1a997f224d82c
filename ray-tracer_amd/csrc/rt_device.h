// rt_device.h — HBM data layout shared by the host flattener (rt_build.hip) and the gfx950 kernels.
//
// The reference's binary BVH (same topology, same float boxes) is stored in two forms:
//   DevNode4  — the traversal structure: the binary tree collapsed to 4-wide nodes (a node's children are reference
//               nodes, opened greedily by surface area).  128 bytes, boxes as structure-of-arrays so that one lane
//               tests the four child boxes with the same instruction stream.  Skipping the intermediate binary nodes is
//               exact: boxes nest, and a leaf whose own box test is unambiguous provably passes every ancestor's test
//               (boxes nest, the slab formula is monotonic); only a leaf whose test is ambiguous walks DevChain[] with the reference's exact test.
//   DevChain  — one record per reference node (box + parent), for that rare walk.
//   DevPrim   — one per leaf primitive reference, stored in leaf order (a leaf is a contiguous run), 64 bytes.
// Leaves are not nodes: a child reference < 0 is ~(count << 26 | start), the leaf's run inside DevPrim[] (count = 0: the
// low bits are an index into DevLeaf[] instead, for leaves that do not fit).  DevPrim[] is stored in the reference
// tree's left-to-right DFS leaf order, so `start` is also the tie-break key of Q4.
#ifndef RT_DEVICE_H_
#define RT_DEVICE_H_

#include <stdint.h>

#include "../../include/rt_mi355.h"

#define DEV_REF_EMPTY 0x3fffffff  // unused child slot of a DevNode4
struct alignas(128) DevNode4 {
    float minx[4], miny[4], minz[4];   // child boxes (BVHNode::bbox of reference nodes, src/BVHTree.h:20)
    float maxx[4], maxy[4], maxz[4];
    int32_t ref[4];                    // >= 0: DevNode4 index, < 0: leaf reference, DEV_REF_EMPTY: no child
    uint32_t cone[4];                  // packed normal cone of each child's subtree (rt_cone.h): may it be pruned by distance?
};
// (the DevChain index of each child, only needed by the rare exact walk, lives in DevScene::node_chain[4 * node + child])
static_assert(sizeof(DevNode4) == 128, "DevNode4 must be 128 bytes");

struct alignas(32) DevChain {
    float lo[3], hi[3];                // the reference node's box
    int32_t parent;                    // reference node above it, -1 for the root
    int32_t _pad;
};
static_assert(sizeof(DevChain) == 32, "DevChain must be 32 bytes");

#define DEV_LEAF_START_BITS 26
#define DEV_LEAF_MAX_PACKED 15
struct DevLeaf {
    int32_t start, count;     // run inside DevPrim[]
};

enum { DEV_PRIM_TRIANGLE = 0, DEV_PRIM_SPHERE = 1 };

struct alignas(64) DevPrim {
    // the first 48 bytes are what the traversal reads per primitive test (three 16-byte loads)
    float p0[3];              // triangle: v0 position            | sphere: centre
    float e1[3];              // triangle: p0p1 = p0 - p1         | sphere: radius, 0, 0
    float e2[3];              // triangle: p0p2 = p0 - p2
    int32_t object;           // DevObject index (transform, material, textures, container identity)
    int32_t kind;             // DEV_PRIM_*
    int32_t shape;            // rt_hit.shape id: triangle index, or sphere index | RT_PRIM_SPHERE
    float nrm[3];             // triangle: object-space unit normal (Triangle.cpp:37)
    int32_t ill;              // 1: triangle so thin that the Cramer solution is rounding noise for every ray (informational; the
                              // traversal's pruning rule is the normal cone of rt_cone.h, which covers this case)
};
static_assert(sizeof(DevPrim) == 64, "DevPrim must be 64 bytes");

enum { DEV_OBJ_SMOOTH = 1, DEV_OBJ_HAS_XFORM = 2, DEV_OBJ_COMPOSITE = 4, DEV_OBJ_MOTION = 8, DEV_OBJ_IDENTITY = 16,
       DEV_OBJ_UNIT = 32 };   // static object whose matrix is bit for bit the unit matrix (1.0f / +0.0f): all such objects share one object-space ray

// 160 bytes; matrices in glm order m[col*4+row].
struct alignas(16) DevObject {
    float M[16];              // Transform::transposeMatrix        (object -> world)
    float inv[16];            // Transform::invTransformationMatrix = inverse(transpose(M))
    float mblur[3];
    int32_t flags;            // DEV_OBJ_*
    int32_t material, tex_color, tex_normal;
    float cone_k;             // sigma_min(A)^2 / sigma_max(A) of the linear part A of M (rt_cone.h); 1 without a transform
};
static_assert(sizeof(DevObject) == 160, "DevObject must be 160 bytes");

struct DevVertex {            // 32 bytes: what the winner's finalisation reads
    float n[3];               // smooth-shading vertex normal (Q28)
    float uv[2];
    float _pad[3];
};
static_assert(sizeof(DevVertex) == 32, "DevVertex must be 32 bytes");

#define RT_TRI_MIN_LANES 10      // run the primitive path of the traversal once this many lanes of a wave wait in a leaf (4 ... 12 within 0.5 %, 16: +1 %, 24: +6 %)
#define RT_DONATE_MIN_IDLE 32    // once the queue is empty: split the rays in flight (donation) when this many lanes of a wave are idle (28-36 measured best)
#define RT_REFILL_MIN_IDLE 16    // retire + refill (or donate) once this many lanes of a wave are idle (8 ... 48 swept: 16 best)

struct DevSceneParams {
    int32_t max_depth;
    float isect_eps, shadow_eps;
    float background[3];
    float ambient[3];
    int32_t bg_texture;
    int32_t n_lights, n_materials, n_textures;
    int32_t root;             // >= 0 DevNode4 index, < 0 leaf reference, RT_NO_ROOT when the scene is empty
    float root_min[3], root_max[3];
    float prune_err;          // absolute bound used to keep t-pruning conservative: how far a computed hit point may lie outside its primitive's box
    float prune_rel;          // the same per unit of |origin|_1
    int32_t bvh_depth;        // depth of the reference (binary) tree
    int32_t stack_bound;      // most entries a traversal of the 4-wide tree can have on its stack
    int32_t tri_min_lanes;    // run the primitive path of the traversal once this many lanes of a wave wait in a leaf
    int32_t refill_min_idle;  // retire + refill (or donate) once this many lanes of a wave are idle
    int32_t donate_min_idle;  // the same threshold once the queue is empty (donation rounds)
    int32_t donate;           // 1: idle lanes of a wave take over stack entries of its long rays once the queue is empty
    int32_t has_normal_tex;   // any object with a normal texture   } select the k_shade_pre instantiation
    int32_t has_rng;          // any area light or glossy material  }
    int32_t has_motion;       // any object with a non-zero motion-blur vector (selects the kernel instantiation)
    int32_t n_objects;
    float isect_eps_p1;       // 1 + isect_eps in float (Triangle.cpp:219): a kernel argument so that it stays in a scalar register
    float flt_max;            // FLT_MAX as a kernel argument: a literal would sit in a vector register of the traversal loop, which
                              // the register allocator spills and reloads (with a wait for every outstanding load) once per node step
    int32_t cone_off;         // tuning builds only (RT_NO_CONE): prune far children without the normal-cone test (NOT exact)
    int32_t interleave_shift; // log2 of the run of consecutive queue entries a wave takes from one place (6: the whole chunk, no interleaving)
    int32_t interleave_max_share;  // interleave only launches with at most this many rays per wave of the grid
    uint32_t node_bytes;      // size of the DevNode4 array in bytes (range of the traversal's buffer descriptor), at most 0xffffff00
    uint32_t prim_bytes;      // the same for the DevPrim array
    int32_t unit_object;      // a DEV_OBJ_UNIT object among the first objects (the traversal's LDS table), or -1
    int32_t tail_window_waves; // the traversal reserves exactly one ray per idle lane once the queue is within this many grids' worth of rays of its end (0: never)
    int32_t tl_slot;          // tuning builds only: slot of the launch-timeline record this launch fills (-1: none)
};
#define RT_NO_ROOT 0x7fffffff

// Pointers to the uploaded scene (all device memory).
struct DevScene {
    DevSceneParams p;
    const DevNode4 *nodes;
    const DevChain *chain;
    const int32_t *node_chain;    // [4 * n_wide]: DevChain index (= reference node) of each child of each DevNode4
    const DevLeaf *leaves;
    const DevPrim *prims;
    const DevObject *objects;
    const DevVertex *verts;
    const int32_t *tri_vert;      // [n_triangles][3]
    const rt_material *materials;
    const rt_light *lights;
    const rt_texture *textures;
    const uint8_t *texels;
};

#endif
