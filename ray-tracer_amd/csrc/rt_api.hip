// rt_api.hip — the C-ABI of librt_mi355.so (include/rt_mi355.h): context, scene upload, wavefront driver.
//
// Host orchestration of one frame (replaces DefaultRenderer::RenderCamera's 8 std::thread row bands,
// reference src/DefaultRenderer.cpp:59-104): the owned pixels (interleaved tiles) are cut into batches of whole
// pixels; per batch  raygen -> for level k: trace, shade_pre, trace_shadow, shade_post (children compacted into
// level k+1) -> backward sweep -> resolve.  Up to rt_render_opts.num_streams batches are in flight, each on its own
// HIP stream with its own queues: a level's launch ends with a tail (the longest traversal chains, few lanes busy)
// that the other streams' launches fill.  The only host round trip per level and stream is the 4-byte child count.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_cone.h"
#include "rt_math.h"
#include "rt_wavefront.h"

// Scratch of rt_intersect_batch_noalloc: [0, NOALLOC_HEADER_INTS) counters (word 0 = the fetch counter the persistent waves
// draw ray indices from), then the traversal-stack overflow columns.  A file-scope device array: every GPU the library runs
// on holds its own copy at its own address until the process ends; host code only ever uses the address hipGetSymbolAddress
// returns for the current device (g_noalloc_scratch[device]), never the array's name.
#define NOALLOC_HEADER_INTS 64
#define NOALLOC_OVERFLOW_INTS (8u << 20)
static __device__ __align__(256) int scratch[NOALLOC_HEADER_INTS + NOALLOC_OVERFLOW_INTS];

namespace {

std::string g_create_error;
enum { RT_MAX_DEVICES = 64 };
int *g_noalloc_scratch[RT_MAX_DEVICES];      // per device: address of `scratch` on that device (set in rt_ctx_create)
// rt_intersect_batch_noalloc shares that scratch between all its calls on a device: a call that is enqueued while an earlier one
// on ANOTHER stream has not finished would race with it for the fetch counter.  The event marks the end of the latest call.
hipEvent_t g_noalloc_done[RT_MAX_DEVICES];
hipStream_t g_noalloc_stream[RT_MAX_DEVICES];
bool g_noalloc_pending[RT_MAX_DEVICES];

#define HIP_TRY(ctx, expr)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
            return RT_ERR_HIP;                                                                                 \
        }                                                                                                      \
    } while (0)

template <typename T>
struct DevArray {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct LevelMem {
    DevArray<float4> ray_o, ray_d, hit0, hit1, hit2, hit3, node_a, node_b, node_c, color;
    DevArray<int4> ray_m;
    DevArray<int2> child;
    int count = 0;
    hipError_t ensure(size_t n) {
        hipError_t e;
        if ((e = ray_o.ensure(n)) != hipSuccess) return e;
        if ((e = ray_d.ensure(n)) != hipSuccess) return e;
        if ((e = ray_m.ensure(n)) != hipSuccess) return e;
        if ((e = hit0.ensure(n)) != hipSuccess) return e;
        if ((e = hit1.ensure(n)) != hipSuccess) return e;
        if ((e = hit2.ensure(n)) != hipSuccess) return e;
        if ((e = hit3.ensure(n)) != hipSuccess) return e;
        if ((e = node_a.ensure(n)) != hipSuccess) return e;
        if ((e = node_b.ensure(n)) != hipSuccess) return e;
        if ((e = node_c.ensure(n)) != hipSuccess) return e;
        if ((e = child.ensure(n)) != hipSuccess) return e;
        if ((e = color.ensure(n)) != hipSuccess) return e;
        return hipSuccess;
    }
    LevelBuf view() const {
        LevelBuf b;
        b.ray_o = ray_o.p; b.ray_d = ray_d.p; b.ray_m = ray_m.p; b.hit0 = hit0.p; b.hit1 = hit1.p; b.hit2 = hit2.p; b.hit3 = hit3.p;
        b.node_a = node_a.p; b.node_b = node_b.p; b.node_c = node_c.p; b.child = child.p; b.color = color.p;
        b.capacity = (int)std::min<size_t>(ray_o.cap, 0x7fffffff); b.count = count;
        return b;
    }
    void release() {
        ray_o.release(); ray_d.release(); ray_m.release(); hit0.release(); hit1.release(); hit2.release(); hit3.release();
        node_a.release(); node_b.release(); node_c.release(); child.release(); color.release();
    }
};

// One in-flight batch: its own stream, queues, counters and traversal-stack overflow area.
struct StreamLane {
    hipStream_t stream = nullptr;
    hipEvent_t ev_count = nullptr, ev_done = nullptr;
    int *h_count = nullptr;                 // pinned: child count of the level just shaded
    std::vector<LevelMem> levels;
    struct ShadowMem {
        DevArray<float4> o, d, ip, ldir;
        DevArray<int> occ;
        hipError_t ensure(size_t n) {
            hipError_t e;
            if ((e = o.ensure(n)) != hipSuccess) return e;
            if ((e = d.ensure(n)) != hipSuccess) return e;
            if ((e = ip.ensure(n)) != hipSuccess) return e;
            if ((e = ldir.ensure(n)) != hipSuccess) return e;
            return occ.ensure(n);
        }
        void release() { o.release(); d.release(); ip.release(); ldir.release(); occ.release(); }
    } sh[2];                                // shadow rays of level k live in sh[k & 1] until level k+1's launch has traced them
    DevArray<int> d_counts;                 // [child counts | fetch counters | refraction children | shadow queue lengths | fetch counters of the shadow launches] x (max_depth+2)
    DevArray<int> d_overflow;
    // Shadow rays and the lighting that needs them run beside the closest-hit chain, on a stream of their own (frames whose level
    // lengths stay on the device): the chain of a frame is trace -> surface pass -> trace of the children, the shadow rays hang off it.
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_pre, ev_post;   // surface pass of level k done (main stream) / lighting of level k done (side stream)
    DevArray<int> d_overflow_side;
    // state of the batch in flight
    bool busy = false;
    int64_t p0 = 0;
    int np = 0, k = 0;
    void release() {
        for (auto &l : levels) l.release();
        levels.clear();
        sh[0].release(); sh[1].release(); d_counts.release(); d_overflow.release(); d_overflow_side.release();
        for (hipEvent_t e : ev_pre) if (e) hipEventDestroy(e);
        for (hipEvent_t e : ev_post) if (e) hipEventDestroy(e);
        ev_pre.clear(); ev_post.clear();
        if (side) hipStreamDestroy(side);
        side = nullptr;
        if (h_count) hipHostFree(h_count);
        if (ev_count) hipEventDestroy(ev_count);
        if (ev_done) hipEventDestroy(ev_done);
        if (stream) hipStreamDestroy(stream);
        h_count = nullptr; ev_count = ev_done = nullptr; stream = nullptr;
    }
};

enum { RT_DEFAULT_STREAMS = 1, RT_MAX_STREAMS = 8 };

}  // namespace

struct rt_ctx {
    int device = 0;
    int num_cus = 256;          // compute units of `device`: the persistent grid and the stack-overflow areas of this context are sized by it
    hipStream_t stream = nullptr;
    std::string error;
    bool has_scene = false;
    HostScene host;             // kept for rt_scene_bvh_info and as the owner of table sizes
    DevScene dev;               // device pointers
    // scene arenas
    DevArray<DevNode4> d_nodes;
    DevArray<DevChain> d_chain;
    DevArray<int32_t> d_node_chain;
    DevArray<DevLeaf> d_leaves;
    DevArray<DevPrim> d_prims;
    DevArray<DevObject> d_objects;
    DevArray<DevVertex> d_verts;
    DevArray<int32_t> d_tri_vert;
    DevArray<rt_material> d_materials;
    DevArray<rt_light> d_lights;
    DevArray<rt_texture> d_textures;
    DevArray<uint8_t> d_texels;
    uint64_t arena_bytes[RT_SCENE_ARENAS] = {0};   // bytes in use of the twelve arrays above, in that order (rt_scene_export)
    // wavefront state
    std::vector<StreamLane> lanes;
    hipEvent_t ev_ready = nullptr;
    DevArray<int> d_pixels;
    int pixels_key[5] = {0, 0, 0, 0, 0};    // (nx, ny, tile size, rank, world) the device pixel list was built for
    DevArray<int> d_overflow;               // rt_intersect_batch
    DevArray<int> d_counts;                 // rt_intersect_batch fetch counter
    DevArray<unsigned long long> d_kind;    // rays per kind
    DevArray<DevCounters> d_counters;
    DevArray<float> d_hdr;
    DevArray<int> d_pid;
    DevArray<float> d_pt;
    DevArray<rt_ray> d_rays;
    DevArray<rt_hit> d_hits;
    int overflow_depth = 0;
    std::vector<double> level_ratio;        // rays per primary sample of every level in the last frame (empty: no frame of this scene yet)
    uint64_t mispredicted_frames = 0;       // frames that had to be rendered a second time because a level outgrew its predicted queue
    int wide_margin_frames = 0;             // frames still to be rendered with the wide queue margin (set after a misprediction)
    std::vector<hipEvent_t> ev_pool;        // timing events, kept between frames (creating one costs several microseconds)
    size_t ev_used = 0;
    hipError_t timing_event(hipEvent_t *out) {
        if (ev_used == ev_pool.size()) { hipEvent_t e; hipError_t rc = hipEventCreate(&e); if (rc != hipSuccess) return rc; ev_pool.push_back(e); }
        *out = ev_pool[ev_used++];
        return hipSuccess;
    }
    // frame post-processing (rt_image_encode, rt_tonemap)
    DevArray<float> d_img_in, d_tm_lum, d_tm_out, d_tm_params;
    DevArray<uint8_t> d_img8, d_img_text, d_img_tmp;
    DevArray<unsigned long long> d_img_len, d_img_off;
    // PLY decoding
    DevArray<uint8_t> d_ply_in, d_ply_tmp;
    DevArray<float> d_ply_pos, d_ply_fix;
    DevArray<int> d_ply_tri, d_ply_cnt, d_ply_flag;
    DevArray<long long> d_ply_off;
    DevArray<double> d_tm_partial;
};

static void release_all(rt_ctx *c) {
    c->d_nodes.release(); c->d_chain.release(); c->d_node_chain.release(); c->d_leaves.release(); c->d_prims.release(); c->d_objects.release(); c->d_verts.release(); c->d_tri_vert.release();
    c->d_materials.release(); c->d_lights.release(); c->d_textures.release(); c->d_texels.release();
    for (auto &l : c->lanes) l.release();
    c->lanes.clear();
    if (c->ev_ready) { hipEventDestroy(c->ev_ready); c->ev_ready = nullptr; }
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    c->ev_pool.clear(); c->ev_used = 0;
    c->d_pixels.release(); c->d_overflow.release();
    c->d_counts.release(); c->d_kind.release(); c->d_counters.release(); c->d_hdr.release(); c->d_pid.release(); c->d_pt.release();
    c->d_rays.release(); c->d_hits.release();
    c->d_img_in.release(); c->d_tm_lum.release(); c->d_tm_out.release(); c->d_tm_params.release();
    c->d_img8.release(); c->d_img_text.release(); c->d_img_tmp.release(); c->d_img_len.release(); c->d_img_off.release(); c->d_tm_partial.release();
    c->d_ply_in.release(); c->d_ply_tmp.release(); c->d_ply_pos.release(); c->d_ply_fix.release(); c->d_ply_tri.release(); c->d_ply_cnt.release(); c->d_ply_flag.release(); c->d_ply_off.release();
}

// Experiment switches of the traversal (thresholds of the persistent loop, pruning margin).  Only compiled into tuning builds
// (make TUNE=1): a release library reads no such variable.  Values are clamped to what the kernel tolerates (an idle threshold
// above 64 would never trigger a refill; a pruning scale below 1 would make pruning non-conservative) and are applied to the
// scene's defaults on every call, so they do not outlive the variable.
static void apply_tuning_knobs(rt_ctx *c) {
#ifdef RT_TUNING_KNOBS
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    DevSceneParams &p = c->dev.p;
    const DevSceneParams &d = c->host.params;
    p.tri_min_lanes = d.tri_min_lanes; p.donate = d.donate; p.refill_min_idle = d.refill_min_idle; p.donate_min_idle = d.donate_min_idle;
    p.prune_err = d.prune_err; p.prune_rel = d.prune_rel; p.cone_off = 0;
    p.interleave_shift = d.interleave_shift; p.interleave_max_share = d.interleave_max_share; p.tail_window_waves = d.tail_window_waves;
    if (const char *tw = getenv("RT_TAIL_WINDOW")) p.tail_window_waves = clampi(atoi(tw), 0, 64);
    if (const char *ig = getenv("RT_INTERLEAVE_SHIFT")) p.interleave_shift = clampi(atoi(ig), 0, 6);
    if (const char *im = getenv("RT_INTERLEAVE_MAX_SHARE")) p.interleave_max_share = clampi(atoi(im), 0, 1 << 30);
    rt_set_trace_block_cap(0);
    if (const char *sb = getenv("RT_SMALL_RAYS_PER_BLOCK")) rt_set_small_rays_per_block(atoi(sb));
    if (const char *nc = getenv("RT_NO_CONE")) p.cone_off = atoi(nc) ? 1 : 0;
    if (const char *cap = getenv("RT_TRACE_BLOCK_CAP")) rt_set_trace_block_cap(std::max(0, atoi(cap)));
    if (const char *tml = getenv("RT_TRI_MIN_LANES")) p.tri_min_lanes = clampi(atoi(tml), 1, 64);
    if (const char *dn = getenv("RT_DONATE")) p.donate = atoi(dn) ? 1 : 0;
    if (const char *ri = getenv("RT_REFILL_MIN_IDLE")) p.refill_min_idle = clampi(atoi(ri), 1, 64);
    if (const char *di = getenv("RT_DONATE_MIN_IDLE")) p.donate_min_idle = clampi(atoi(di), 1, 64);
    if (const char *ps = getenv("RT_PRUNE_SCALE")) { const float f = std::max(1.0f, (float)atof(ps)); p.prune_err = d.prune_err * f; p.prune_rel = d.prune_rel * f; }
#else
    (void)c;
#endif
}

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_ctx_create(int hip_device_id, rt_ctx **out) {
    if (!out) { g_create_error = "out is NULL"; return RT_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                         "); librt_mi355 has no CPU fallback";
        return RT_ERR_NO_DEVICE;
    }
    if (hip_device_id < 0 || hip_device_id >= n) { g_create_error = "hip_device_id out of range"; return RT_ERR_INVALID; }
    int num_cus = 256;
    e = hipSetDevice(hip_device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, hip_device_id);
        if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return RT_ERR_HIP; }
        num_cus = prop.multiProcessorCount;
        rt_set_num_cus(num_cus);
    }
    if (hip_device_id < RT_MAX_DEVICES && !g_noalloc_scratch[hip_device_id]) {     // once per device, with that device current
        int *ptr = nullptr;
        e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) { g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return RT_ERR_HIP; }
        e = hipEventCreateWithFlags(&g_noalloc_done[hip_device_id], hipEventDisableTiming);      // created here, once: the entry point itself allocates nothing
        if (e != hipSuccess) { g_create_error = std::string("hipEventCreate: ") + hipGetErrorString(e); return RT_ERR_HIP; }
        g_noalloc_scratch[hip_device_id] = ptr;
    }
    rt_ctx *c = new rt_ctx();
    c->device = hip_device_id;
    c->num_cus = num_cus;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return RT_ERR_HIP; }
    std::memset(&c->dev, 0, sizeof(c->dev));
    *out = c;
    return RT_OK;
}

void rt_ctx_destroy(rt_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); }
    release_all(c);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *rt_last_error(const rt_ctx *c) { return c ? c->error.c_str() : g_create_error.c_str(); }

#define UPLOAD(ctx, darr, hvec)                                                                                            \
    do {                                                                                                                   \
        HIP_TRY(ctx, (darr).ensure((hvec).size() ? (hvec).size() : 1));                                                    \
        if (!(hvec).empty()) HIP_TRY(ctx, hipMemcpyAsync((darr).p, (hvec).data(), (hvec).size() * sizeof((hvec)[0]), hipMemcpyHostToDevice, (ctx)->stream)); \
    } while (0)

int rt_scene_upload(rt_ctx *c, const rt_scene_desc *scene) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    c->has_scene = false;
    c->level_ratio.clear();
    if (scene && scene->abi_version == RT_ABI_VERSION && scene->n_spheres > 0 && scene->sph_object && scene->objects)
        for (int64_t i = 0; i < scene->n_spheres; ++i) {                 // Sphere.cpp:128 dereferences the transform unconditionally (Q9)
            const int32_t ob = scene->sph_object[i];
            if (ob >= 0 && ob < scene->n_objects && !(scene->objects[ob].flags & RT_OBJ_HAS_XFORM)) { c->error = "sphere object without a transform"; return RT_ERR_INVALID; }
        }
    RtGpuScene gs;
    int rc = rt_build_host_scene(scene, c->host, c->error, /*gpu_build=*/1, &gs);
    if (rc != RT_OK) return rc;
    HostScene &h = c->host;
    if (gs.prims) {       // built and flattened on the device (rt_bvh_gpu.hip): adopt the arrays
        c->d_nodes.release(); c->d_chain.release(); c->d_node_chain.release(); c->d_leaves.release(); c->d_prims.release();
        c->d_nodes.p = gs.nodes; c->d_nodes.cap = (size_t)gs.n_wide;
        c->d_node_chain.p = gs.node_chain; c->d_node_chain.cap = 4 * (size_t)gs.n_wide;
        if (!c->d_node_chain.p) HIP_TRY(c, c->d_node_chain.ensure(4));
        c->d_chain.p = gs.chain; c->d_chain.cap = (size_t)gs.n_nodes;
        c->d_leaves.p = gs.leaves; c->d_leaves.cap = (size_t)gs.n_leaves;
        c->d_prims.p = gs.prims; c->d_prims.cap = (size_t)gs.n_leaf_prims;
        if (!c->d_nodes.p) HIP_TRY(c, c->d_nodes.ensure(1));          // a one-leaf scene has no wide node
    } else {
        UPLOAD(c, c->d_nodes, h.nodes);
        UPLOAD(c, c->d_chain, h.chain);
        UPLOAD(c, c->d_node_chain, h.node_chain);
        UPLOAD(c, c->d_leaves, h.leaves);
        UPLOAD(c, c->d_prims, h.prims);
    }
    UPLOAD(c, c->d_objects, h.objects);
    if (gs.verts) {       // shape setup ran on the device as well (rt_ingest.hip)
        c->d_verts.release(); c->d_tri_vert.release();
        c->d_verts.p = gs.verts; c->d_verts.cap = (size_t)std::max<int64_t>(1, gs.n_verts);
        c->d_tri_vert.p = gs.tri_vert; c->d_tri_vert.cap = (size_t)std::max<int64_t>(1, 3 * gs.n_tris);
    } else {
        UPLOAD(c, c->d_verts, h.verts);
        UPLOAD(c, c->d_tri_vert, h.tri_vert);
    }
    UPLOAD(c, c->d_materials, h.materials);
    UPLOAD(c, c->d_lights, h.lights);
    UPLOAD(c, c->d_textures, h.textures);
    UPLOAD(c, c->d_texels, h.texels);
    DevScene &d = c->dev;
    d.p = h.params;
    {   // the traversal reads node records through a buffer descriptor with 32-bit offsets
        const uint64_t nb = (uint64_t)h.info.n_wide_nodes * sizeof(DevNode4);
        if (nb > 0xffffff00ull) { c->error = "too many BVH nodes for the traversal's 32-bit node offsets (limit 2^25 - 2 wide nodes)"; return RT_ERR_TOO_LARGE; }
        d.p.node_bytes = (uint32_t)nb;
        const uint64_t pb = (uint64_t)h.info.n_leaf_prims * sizeof(DevPrim);
        if (pb > 0xffffff00ull) { c->error = "too many primitive records for the traversal's 32-bit offsets (limit 2^26 - 4)"; return RT_ERR_TOO_LARGE; }
        d.p.prim_bytes = (uint32_t)pb;
    }
    d.nodes = c->d_nodes.p; d.chain = c->d_chain.p; d.node_chain = c->d_node_chain.p; d.leaves = c->d_leaves.p; d.prims = c->d_prims.p; d.objects = c->d_objects.p; d.verts = c->d_verts.p;
    d.tri_vert = c->d_tri_vert.p; d.materials = c->d_materials.p; d.lights = c->d_lights.p; d.textures = c->d_textures.p; d.texels = c->d_texels.p;
    // traversal stack overflow area: entries beyond the LDS part, one column per persistent thread
    c->overflow_depth = std::max(0, h.params.stack_bound + 2 - rt_trace_lds_stack_depth());
    HIP_TRY(c, c->d_overflow.ensure((size_t)std::max(1, c->overflow_depth) * rt_trace_grid_threads()));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        const bool dev_tree = gs.prims != nullptr, dev_shapes = gs.verts != nullptr;
        uint64_t *b = c->arena_bytes;
        b[0] = (dev_tree ? (uint64_t)gs.n_wide : h.nodes.size()) * sizeof(DevNode4);
        b[1] = (dev_tree ? (uint64_t)gs.n_nodes : h.chain.size()) * sizeof(DevChain);
        b[2] = (dev_tree ? (gs.node_chain ? 4 * (uint64_t)gs.n_wide : 0) : h.node_chain.size()) * sizeof(int32_t);
        b[3] = (dev_tree ? (uint64_t)gs.n_leaves : h.leaves.size()) * sizeof(DevLeaf);
        b[4] = (dev_tree ? (uint64_t)gs.n_leaf_prims : h.prims.size()) * sizeof(DevPrim);
        b[5] = h.objects.size() * sizeof(DevObject);
        b[6] = (dev_shapes ? (uint64_t)gs.n_verts : h.verts.size()) * sizeof(DevVertex);
        b[7] = (dev_shapes ? 3 * (uint64_t)gs.n_tris : h.tri_vert.size()) * sizeof(int32_t);
        b[8] = h.materials.size() * sizeof(rt_material);
        b[9] = h.lights.size() * sizeof(rt_light);
        b[10] = h.textures.size() * sizeof(rt_texture);
        b[11] = h.texels.size();
    }
    c->has_scene = true;
    return RT_OK;
}

// ---- the uploaded scene as a header + flat arenas (one build, many GPUs) ---------------------------------------------------
namespace {
struct SceneHeader {
    uint32_t magic, abi_version;
    uint32_t record_bytes[RT_SCENE_ARENAS];      // element size of every arena: a library with other record layouts must refuse the bytes
    uint64_t arena_bytes[RT_SCENE_ARENAS];
    uint64_t arena_mark[RT_SCENE_ARENAS];     // FNV-1a of each arena's first and last 256 bytes: ties the header to the arrays it was exported with
    rt_bvh_info info;
    DevSceneParams params;
};
// fingerprint of an arena that lives at `src` (device or host memory): its first and last min(256, bytes) bytes
int arena_mark(rt_ctx *c, const void *src, uint64_t bytes, uint64_t *out) {
    unsigned char buf[512];
    const size_t k = (size_t)std::min<uint64_t>(256, bytes);
    *out = 0xcbf29ce484222325ull ^ bytes;
    if (!k) return RT_OK;
    HIP_TRY(c, hipMemcpyAsync(buf, src, k, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(buf + k, (const unsigned char *)src + (bytes - k), k, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t h = *out;
    for (size_t i = 0; i < 2 * k; ++i) { h ^= buf[i]; h *= 0x100000001b3ull; }
    *out = h;
    return RT_OK;
}
const uint32_t kSceneMagic = 0x4e435352u;        // "RSCN"
const uint32_t kRecordBytes[RT_SCENE_ARENAS] = {sizeof(DevNode4), sizeof(DevChain), sizeof(int32_t), sizeof(DevLeaf), sizeof(DevPrim), sizeof(DevObject),
                                                sizeof(DevVertex), sizeof(int32_t), sizeof(rt_material), sizeof(rt_light), sizeof(rt_texture), 1};
void *arena_ptr(rt_ctx *c, int i) {
    void *p[RT_SCENE_ARENAS] = {c->d_nodes.p, c->d_chain.p, c->d_node_chain.p, c->d_leaves.p, c->d_prims.p, c->d_objects.p,
                                c->d_verts.p, c->d_tri_vert.p, c->d_materials.p, c->d_lights.p, c->d_textures.p, c->d_texels.p};
    return p[i];
}
}  // namespace

size_t rt_scene_header_bytes(void) { return sizeof(SceneHeader); }

int rt_scene_export(rt_ctx *c, void *header, uint64_t arena_bytes[RT_SCENE_ARENAS]) {
    if (!c || !header || !arena_bytes) return RT_ERR_INVALID;
    if (!c->has_scene) { c->error = "rt_scene_export: no scene uploaded"; return RT_ERR_NO_SCENE; }
    SceneHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = kSceneMagic; hd.abi_version = RT_ABI_VERSION;
    for (int i = 0; i < RT_SCENE_ARENAS; ++i) { hd.record_bytes[i] = kRecordBytes[i]; hd.arena_bytes[i] = c->arena_bytes[i]; arena_bytes[i] = c->arena_bytes[i]; }
    hd.info = c->host.info;
    hd.params = c->host.params;
    HIP_TRY(c, hipSetDevice(c->device));
    for (int i = 0; i < RT_SCENE_ARENAS; ++i) { const int rc = arena_mark(c, arena_ptr(c, i), c->arena_bytes[i], &hd.arena_mark[i]); if (rc != RT_OK) return rc; }
    std::memcpy(header, &hd, sizeof(hd));
    return RT_OK;
}

int rt_scene_export_arena(rt_ctx *c, int index, void *dst) {
    if (!c || index < 0 || index >= RT_SCENE_ARENAS) return RT_ERR_INVALID;
    if (!c->has_scene) { c->error = "rt_scene_export_arena: no scene uploaded"; return RT_ERR_NO_SCENE; }
    if (c->arena_bytes[index] == 0) return RT_OK;
    if (!dst) { c->error = "rt_scene_export_arena: dst is NULL"; return RT_ERR_INVALID; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    HIP_TRY(c, hipMemcpyAsync(dst, arena_ptr(c, index), c->arena_bytes[index], hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_scene_import(rt_ctx *c, const void *header, const void *const arenas[RT_SCENE_ARENAS]) {
    if (!c || !header || !arenas) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    SceneHeader hd;
    std::memcpy(&hd, header, sizeof(hd));
    if (hd.magic != kSceneMagic) { c->error = "rt_scene_import: not a scene header"; return RT_ERR_INVALID; }
    if (hd.abi_version != RT_ABI_VERSION) { c->error = "rt_scene_import: header of another ABI version"; return RT_ERR_ABI; }
    for (int i = 0; i < RT_SCENE_ARENAS; ++i) {
        if (hd.record_bytes[i] != kRecordBytes[i]) { c->error = "rt_scene_import: header of a library with other record layouts"; return RT_ERR_ABI; }
        if (hd.arena_bytes[i] % kRecordBytes[i] != 0 || (hd.arena_bytes[i] && !arenas[i])) { c->error = "rt_scene_import: bad arena size or NULL arena"; return RT_ERR_INVALID; }
    }
    // what the kernels index with must be consistent with the arena sizes (the bytes come from another process)
    const DevSceneParams &p = hd.params;
    const uint64_t n_wide = hd.arena_bytes[0] / sizeof(DevNode4), n_prims = hd.arena_bytes[4] / sizeof(DevPrim);
    if (p.n_lights < 0 || (uint64_t)p.n_lights * sizeof(rt_light) != hd.arena_bytes[9] || p.n_materials < 0 || (uint64_t)p.n_materials * sizeof(rt_material) != hd.arena_bytes[8] ||
        p.n_textures < 0 || (uint64_t)p.n_textures * sizeof(rt_texture) != hd.arena_bytes[10] || p.n_objects < 0 || (uint64_t)p.n_objects * sizeof(DevObject) != hd.arena_bytes[5] ||
        (p.root != RT_NO_ROOT && p.root >= 0 && (uint64_t)p.root >= n_wide) || hd.arena_bytes[2] != 4 * n_wide * sizeof(int32_t) || (p.root != RT_NO_ROOT && n_prims == 0) ||
        p.stack_bound < 0 || p.stack_bound > (1 << 20) || p.bg_texture >= p.n_textures) { c->error = "rt_scene_import: header inconsistent with its arenas"; return RT_ERR_INVALID; }
    {   // a traversal stack sized too small, a root leaf outside the primitive records or arrays of another export would be
        // out-of-bounds accesses in the kernels, not errors: refuse them here
        const uint64_t n_leaves = hd.arena_bytes[3] / sizeof(DevLeaf);
        bool ok = hd.info.n_wide_nodes >= 0 && (uint64_t)hd.info.n_wide_nodes == n_wide && hd.info.max_depth >= 0;
        // a 4-wide tree of depth w has at most (4^w - 1) / 3 nodes, and its traversal stack needs 3 w + 2 entries
        int w = 0;
        for (uint64_t cap = 0; cap < n_wide; cap = 4 * cap + 1) ++w;
        ok = ok && p.stack_bound >= 3 * w + 2;
        if (p.root != RT_NO_ROOT && p.root < 0) {
            const uint32_t u = (uint32_t)~p.root, start = u & ((1u << DEV_LEAF_START_BITS) - 1), count = u >> DEV_LEAF_START_BITS;
            ok = ok && (count ? (uint64_t)start + count <= n_prims : (uint64_t)start < n_leaves);
        }
        if (!ok) { c->error = "rt_scene_import: header inconsistent with its arenas"; return RT_ERR_INVALID; }
        for (int i = 0; i < RT_SCENE_ARENAS; ++i) {
            uint64_t mark = 0;
            const int rc = arena_mark(c, arenas[i], hd.arena_bytes[i], &mark);
            if (rc != RT_OK) return rc;
            if (mark != hd.arena_mark[i]) { c->error = "rt_scene_import: an arena is not the one this header was exported with"; return RT_ERR_INVALID; }
        }
    }
    c->has_scene = false;
    c->level_ratio.clear();
    c->host = HostScene();
    c->host.info = hd.info;
    c->host.params = hd.params;
#define RT_IMPORT(idx, darr)                                                                                                 \
    do {                                                                                                                     \
        const size_t n_el = (size_t)(hd.arena_bytes[idx] / kRecordBytes[idx]);                                               \
        HIP_TRY(c, (darr).ensure(n_el ? n_el : 1));                                                                          \
        if (n_el) HIP_TRY(c, hipMemcpyAsync((darr).p, arenas[idx], hd.arena_bytes[idx], hipMemcpyDefault, c->stream));       \
        c->arena_bytes[idx] = hd.arena_bytes[idx];                                                                           \
    } while (0)
    RT_IMPORT(0, c->d_nodes); RT_IMPORT(1, c->d_chain); RT_IMPORT(2, c->d_node_chain); RT_IMPORT(3, c->d_leaves); RT_IMPORT(4, c->d_prims); RT_IMPORT(5, c->d_objects);
    RT_IMPORT(6, c->d_verts); RT_IMPORT(7, c->d_tri_vert); RT_IMPORT(8, c->d_materials); RT_IMPORT(9, c->d_lights); RT_IMPORT(10, c->d_textures); RT_IMPORT(11, c->d_texels);
#undef RT_IMPORT
    DevScene &d = c->dev;
    d.p = hd.params;
    // the ranges of the traversal's buffer descriptors are those of the arenas that actually arrived
    if (hd.arena_bytes[0] > 0xffffff00ull || hd.arena_bytes[4] > 0xffffff00ull) { c->error = "imported scene too large for the traversal's 32-bit record offsets"; return RT_ERR_TOO_LARGE; }
    d.p.node_bytes = (uint32_t)(hd.arena_bytes[0] / sizeof(DevNode4) * sizeof(DevNode4));
    d.p.prim_bytes = (uint32_t)(hd.arena_bytes[4] / sizeof(DevPrim) * sizeof(DevPrim));
    d.nodes = c->d_nodes.p; d.chain = c->d_chain.p; d.node_chain = c->d_node_chain.p; d.leaves = c->d_leaves.p; d.prims = c->d_prims.p; d.objects = c->d_objects.p; d.verts = c->d_verts.p;
    d.tri_vert = c->d_tri_vert.p; d.materials = c->d_materials.p; d.lights = c->d_lights.p; d.textures = c->d_textures.p; d.texels = c->d_texels.p;
    c->overflow_depth = std::max(0, hd.params.stack_bound + 2 - rt_trace_lds_stack_depth());
    HIP_TRY(c, c->d_overflow.ensure((size_t)std::max(1, c->overflow_depth) * rt_trace_grid_threads()));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->has_scene = true;
    return RT_OK;
}

int rt_scene_bvh_info(const rt_ctx *c, rt_bvh_info *out) {
    if (!c || !out) return RT_ERR_INVALID;
    if (!c->has_scene) return RT_ERR_NO_SCENE;
    *out = c->host.info;
    return RT_OK;
}

// ---- host-only helpers -----------------------------------------------------------------------------
static inline bool owns_tile(int tile_id, int rank, int world) { return world <= 1 || (tile_id % world) == rank; }

static int64_t tile_walk(int nx, int ny, int ts, int rank, int world, int32_t *out) {
    if (nx <= 0 || ny <= 0) return 0;
    if (ts <= 0) ts = std::max(nx, ny);
    if (world <= 1) { rank = 0; world = 1; }
    const int tx = (nx + ts - 1) / ts, ty = (ny + ts - 1) / ts;
    int64_t n = 0;
    for (int j = 0; j < ty; ++j)
        for (int i = 0; i < tx; ++i) {
            if (!owns_tile(j * tx + i, rank, world)) continue;
            const int x1 = std::min(nx, (i + 1) * ts), y1 = std::min(ny, (j + 1) * ts);
            for (int y = j * ts; y < y1; ++y)
                for (int x = i * ts; x < x1; ++x) {
                    if (out) out[n] = y * nx + x;
                    ++n;
                }
        }
    return n;
}
int64_t rt_owned_pixel_count(int nx, int ny, int tile_size, int tile_rank, int tile_world) { return tile_walk(nx, ny, tile_size, tile_rank, tile_world, nullptr); }
int rt_tile_pixel_order(int nx, int ny, int tile_size, int tile_rank, int tile_world, int32_t *out_index) {
    if (tile_world > 1 && (tile_rank < 0 || tile_rank >= tile_world)) return RT_ERR_INVALID;
    tile_walk(nx, ny, tile_size, tile_rank, tile_world, out_index);
    return RT_OK;
}

int rt_perlin_tables(int n_tables, int32_t *out) {
    if (n_tables < 0 || (n_tables && !out)) return RT_ERR_INVALID;
    std::default_random_engine generator;                      // Scene::generator (src/Scene.h:51): default seed
    for (int t = 0; t < n_tables; ++t) {
        std::vector<int> idx;
        for (int i = 0; i < 16; ++i) idx.push_back(i);         // PerlinTextureImpl.cpp:29-32
        std::shuffle(idx.begin(), idx.end(), generator);
        for (int i = 0; i < 16; ++i) out[t * 16 + i] = idx[i];
    }
    return RT_OK;
}

int rt_resolve_8bit(const float *hdr_rgb, int64_t n_pixels, uint8_t *out_rgb) {
    if (!hdr_rgb || !out_rgb || n_pixels < 0) return RT_ERR_INVALID;
    for (int64_t i = 0; i < 3 * n_pixels; ++i) {
        // ObtainColorFromUnclampedVector (DefaultRenderer.cpp:18-27): ">255 -> 255" then float->uchar truncation.  Negative
        // or NaN inputs are undefined behaviour in the reference; they are mapped to 0 here.
        float v = hdr_rgb[i];
        out_rgb[i] = v > 255.0f ? 255 : (v > 0.0f ? (uint8_t)v : 0);
    }
    return RT_OK;
}

// ---- camera -------------------------------------------------------------------------------------------
static DevCamera make_camera(const rt_camera_desc *cd, unsigned seed) {
    DevCamera c;
    f3 gaze = normalize3(ld3(cd->gaze)), up = normalize3(ld3(cd->up)), right;   // SetupCameraCoordinateAxes, Camera.cpp:48-64
    const float corr = dot3(gaze, up);
    if (corr > -0.01f && corr < 0.01f) {
        up = normalize3(up);
        right = normalize3(cross3(gaze, up));
    } else {
        right = normalize3(cross3(gaze, up));
        up = normalize3(cross3(right, gaze));
    }
    c.pos[0] = cd->position[0]; c.pos[1] = cd->position[1]; c.pos[2] = cd->position[2];
    c.gaze[0] = gaze.x; c.gaze[1] = gaze.y; c.gaze[2] = gaze.z;
    c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
    c.right[0] = right.x; c.right[1] = right.y; c.right[2] = right.z;
    c.left = cd->left; c.top = cd->top; c.distance = cd->distance;
    c.pixel_w = (cd->right - cd->left) / cd->nx;                                   // Camera.cpp:22-23
    c.pixel_h = (cd->top - cd->bottom) / cd->ny;
    c.nx = cd->nx; c.ny = cd->ny; c.spp = cd->num_samples < 1 ? 1 : cd->num_samples;
    c.jitter_n = (int)std::sqrt((double)c.spp);                                    // PixelSampler.cpp:22-24
    if (c.jitter_n < 1) c.jitter_n = 1;
    c.jitter_w = 1.0f / c.jitter_n;
    c.aperture = cd->aperture_size; c.focus = cd->focus_distance; c.seed = seed;
    return c;
}

int rt_render(rt_ctx *c, const rt_camera_desc *cam, const rt_render_opts *opts, float *hdr_rgb, int32_t *prim_id, float *prim_t, rt_stats *stats) {
    if (!c) return RT_ERR_INVALID;
    if (!cam || !opts || !hdr_rgb) { c->error = "camera, opts and hdr_rgb are required"; return RT_ERR_INVALID; }
    if (opts->abi_version != RT_ABI_VERSION) { c->error = "rt_render_opts.abi_version mismatch"; return RT_ERR_ABI; }
    if (!c->has_scene) { c->error = "no scene uploaded"; return RT_ERR_NO_SCENE; }
    if (cam->nx <= 0 || cam->ny <= 0 || (int64_t)cam->nx * cam->ny > 0x7fffffff) { c->error = "bad image resolution"; return RT_ERR_INVALID; }
    if (opts->tile_world > 1 && (opts->tile_rank < 0 || opts->tile_rank >= opts->tile_world)) { c->error = "tile_rank outside tile_world"; return RT_ERR_INVALID; }
    const auto t_begin = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    hipStream_t s = c->stream;
    const DevCamera dc = make_camera(cam, opts->seed);
    const int spp = dc.spp;
    const int max_depth = std::max(0, c->dev.p.max_depth);
    const int nl = c->dev.p.n_lights;

    // owned pixels (interleaved tiles: tile id % tile_world == tile_rank)
    const int64_t n_owned = rt_owned_pixel_count(cam->nx, cam->ny, opts->tile_size, opts->tile_rank, opts->tile_world);
    // the pixel order only depends on (nx, ny, tile parameters): keep the device copy between frames
    const int pix_key[5] = {cam->nx, cam->ny, opts->tile_size, opts->tile_rank, opts->tile_world};
    if (std::memcmp(pix_key, c->pixels_key, sizeof(pix_key)) != 0 || !c->d_pixels.p) {
        std::vector<int32_t> pixels((size_t)n_owned);
        rt_tile_pixel_order(cam->nx, cam->ny, opts->tile_size, opts->tile_rank, opts->tile_world, pixels.data());
        HIP_TRY(c, c->d_pixels.ensure((size_t)std::max<int64_t>(1, n_owned)));
        if (n_owned) HIP_TRY(c, hipMemcpyAsync(c->d_pixels.p, pixels.data(), (size_t)n_owned * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));          // `pixels` is pageable host memory that dies here
        std::memcpy(c->pixels_key, pix_key, sizeof(pix_key));
    }

    const bool compact = opts->compact_output != 0;
    const size_t n_out = compact ? (size_t)n_owned : (size_t)cam->nx * cam->ny;
    float *d_hdr = hdr_rgb; int *d_pid = prim_id; float *d_pt = prim_t;
    if (!opts->output_on_device) {
        HIP_TRY(c, c->d_hdr.ensure(std::max<size_t>(1, 3 * n_out)));
        d_hdr = c->d_hdr.p;
        if (prim_id) { HIP_TRY(c, c->d_pid.ensure(std::max<size_t>(1, n_out))); d_pid = c->d_pid.p; }
        if (prim_t) { HIP_TRY(c, c->d_pt.ensure(std::max<size_t>(1, n_out))); d_pt = c->d_pt.p; }
    }
    if (!compact) {      // pixels this rank does not own stay zero / -1
        HIP_TRY(c, hipMemsetAsync(d_hdr, 0, 3 * n_out * sizeof(float), s));
        if (d_pid) HIP_TRY(c, hipMemsetAsync(d_pid, 0xff, n_out * sizeof(int), s));
        if (d_pt) HIP_TRY(c, hipMemsetAsync(d_pt, 0, n_out * sizeof(float), s));
    }

    apply_tuning_knobs(c);
    // streams
    int n_streams = opts->num_streams > 0 ? std::min<int>(opts->num_streams, RT_MAX_STREAMS) : RT_DEFAULT_STREAMS;
    int64_t batch_samples = opts->max_batch_samples > 0 ? opts->max_batch_samples : (8 << 20);     // a 4K frame at 1 spp in one batch: +11 % over two 4M batches
    int64_t batch_pixels = std::max<int64_t>(1, batch_samples / spp);
    if (batch_pixels * spp > 0x3fffffff) batch_pixels = 0x3fffffff / spp;
    {   // split the frame evenly over the streams (whole 256-pixel runs), never above the caller's batch limit
        int64_t share = (n_owned + n_streams - 1) / n_streams;
        share = (share + 255) / 256 * 256;
        batch_pixels = std::max<int64_t>(1, std::min(batch_pixels, share));
    }
    n_streams = (int)std::max<int64_t>(1, std::min<int64_t>(n_streams, (n_owned + batch_pixels - 1) / batch_pixels));
    if ((int)c->lanes.size() < n_streams) c->lanes.resize(n_streams);
    if (!c->ev_ready) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    const size_t n_cnt = 5 * ((size_t)max_depth + 2);
    const bool side_shadows = opts->shadow_stream == 1 && nl > 0;
    for (int j = 0; j < n_streams; ++j) {
        StreamLane &ln = c->lanes[j];
        if (side_shadows) {
            if (!ln.side) HIP_TRY(c, hipStreamCreateWithFlags(&ln.side, hipStreamNonBlocking));
            while ((int)ln.ev_pre.size() < max_depth + 2) {
                hipEvent_t a = nullptr, b = nullptr;
                HIP_TRY(c, hipEventCreateWithFlags(&a, hipEventDisableTiming));
                ln.ev_pre.push_back(a);
                HIP_TRY(c, hipEventCreateWithFlags(&b, hipEventDisableTiming));
                ln.ev_post.push_back(b);
            }
            HIP_TRY(c, ln.d_overflow_side.ensure((size_t)std::max(1, c->overflow_depth) * rt_trace_grid_threads()));
        }
        if (!ln.stream) {
            HIP_TRY(c, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&ln.ev_count, hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming));
            HIP_TRY(c, hipHostMalloc((void **)&ln.h_count, sizeof(int), hipHostMallocDefault));
        }
        if ((int)ln.levels.size() < max_depth + 2) ln.levels.resize(max_depth + 2);
        HIP_TRY(c, ln.d_counts.ensure(n_cnt));
        HIP_TRY(c, ln.d_overflow.ensure((size_t)std::max(1, c->overflow_depth) * rt_trace_grid_threads()));
        ln.busy = false;
    }
    HIP_TRY(c, c->d_kind.ensure(RT_RAY_KINDS));
    DevCounters *d_counters = nullptr;
    if (stats && opts->timing >= 2) {       // timing >= 2: also count node visits / tests (slower kernels)
        HIP_TRY(c, c->d_counters.ensure(1));
        d_counters = c->d_counters.p;
    }

    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;
    const bool timing = stats && opts->timing;
    uint64_t primary_rays = 0, trace_launches = 0, trace_rays = 0;

    auto timed_trace = [&](hipStream_t st, auto launch) -> int {
        if (timing) {
            hipEvent_t a, b;
            HIP_TRY(c, c->timing_event(&a)); HIP_TRY(c, c->timing_event(&b));
            HIP_TRY(c, hipEventRecord(a, st));
            launch();
            HIP_TRY(c, hipEventRecord(b, st));
            trace_events.push_back({a, b});
        } else launch();
        return RT_OK;
    };

    // counters of one batch: [child counts | fetch counters | refraction children | shadow queue lengths], (max_depth + 2) ints each
    auto shadow_view = [&](StreamLane &ln, int *counts, int k) -> ShadowBuf {
        StreamLane::ShadowMem &m = ln.sh[k & 1];
        ShadowBuf sb;
        sb.o = m.o.p; sb.d = m.d.p; sb.ipref = m.ip.p; sb.ldir = m.ldir.p; sb.occluded = m.occ.p;
        sb.count = counts + 3 * (max_depth + 2) + k;
        sb.capacity = (int)std::min<size_t>(m.o.cap, 0x7fffffff);
        return sb;
    };
    auto shadow_bound = [&](StreamLane &ln, int k) -> int { return nl > 0 ? ln.levels[k].count * nl : 0; };
    // After the last level: its shadow rays, its lighting, then the backward sweep and the resolve of the batch.
    // dev: the lengths of levels >= 1 are device counters (levels[k].count is then the bound they were sized with).
    auto finish_batch = [&](StreamLane &ln, int *counts, int last, bool dev) -> int {
        hipStream_t st = ln.stream;
        auto ndev = [&](int k) -> const int * { return dev && k > 0 ? counts + k : nullptr; };
        const ShadowBuf sb = shadow_view(ln, counts, last);
        const int bound = shadow_bound(ln, last);
        if (dev && side_shadows) {
            // every level's shadow rays and lighting were queued on the side stream by enqueue_level: the backward sweep needs them all
            // (the side stream runs them in level order, so its last event covers the earlier ones)
            HIP_TRY(c, hipStreamWaitEvent(st, ln.ev_post[(size_t)last], 0));
        } else {
        if (bound > 0) {
            int rc = timed_trace(st, [&] { rt_launch_trace(st, c->dev, ln.levels[last].view(), 0, nullptr, sb, bound, counts + (max_depth + 2) + last + 1, ln.d_overflow.p, d_counters); });
            if (rc != RT_OK) return rc;
            ++trace_launches;
            if (!dev) trace_rays += (uint64_t)bound;
        }
        rt_launch_shade_post(st, c->dev, dc, ln.levels[last].view(), ln.levels[last].count, ndev(last), sb);
        }
        for (int k = last; k >= 0; --k)
            rt_launch_backward(st, ln.levels[k].view(), ln.levels[k].count, ndev(k), ln.levels[k + 1].view(), (k < last) ? 1 : 0);
        rt_launch_resolve(st, dc, c->d_pixels.p, (int)ln.p0, ln.np, ln.levels[0].view(), compact ? 1 : 0, d_hdr, d_pid, d_pt);
        ln.busy = false;
        return RT_OK;
    };
    // Level ln.k: ONE trace launch for its closest-hit rays and the shadow rays of level k-1, then the lighting of level
    // k-1 and the surface pass of level k (shadow rays + children).  Host-counted mode (dev = false): leaves the lane waiting
    // for the child count, or finishes the batch at the last level.
    auto enqueue_level = [&](StreamLane &ln, int *counts, bool dev, int next_cap) -> int {
        const int k = ln.k;
        hipStream_t st = ln.stream;
        LevelMem &lm = ln.levels[k];
        const int n = lm.count;
        auto ndev = [&](int q) -> const int * { return dev && q > 0 ? counts + q : nullptr; };
        const bool split = dev && side_shadows;      // the shadow rays of this level get a launch of their own, on the side stream
        const ShadowBuf sb_prev = shadow_view(ln, counts, k > 0 ? k - 1 : 1);
        const int bound_prev = (k > 0 && !split) ? shadow_bound(ln, k - 1) : 0;
        int rc = timed_trace(st, [&] { rt_launch_trace(st, c->dev, lm.view(), n, ndev(k), sb_prev, bound_prev, counts + (max_depth + 2) + k, ln.d_overflow.p, d_counters); });
        if (rc != RT_OK) return rc;
        ++trace_launches;
        if (!dev) trace_rays += (uint64_t)n + (uint64_t)bound_prev;
        LevelMem &nx = ln.levels[k + 1];
        LevelBuf nxv;
        if (!dev) {
            const size_t nsh = (size_t)n * (size_t)std::max(1, nl);
            if (nsh > 0x3fffffff) { c->error = "shadow queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
            HIP_TRY(c, ln.sh[k & 1].ensure(nsh));
            nx.count = 0;
            if (k < max_depth) {
                if ((size_t)n * 2 > 0x3fffffff) { c->error = "ray queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
                HIP_TRY(c, nx.ensure((size_t)n * 2));
            }
            nxv = nx.view();
        } else { nxv = nx.view(); nxv.capacity = next_cap; }     // buffers were sized before the batch started
        // the surface pass of level k writes the shadow queue sh[k & 1]: the lighting of level k-2, which read it last, must be done
        if (split && k >= 2) HIP_TRY(c, hipStreamWaitEvent(st, ln.ev_post[(size_t)k - 2], 0));
        rt_launch_shade_pre(st, c->dev, dc, lm.view(), n, ndev(k), shadow_view(ln, counts, k), nxv, k, counts + k + 1, counts + 2 * (max_depth + 2) + k);
        if (split) {
            const ShadowBuf sb_k = shadow_view(ln, counts, k);
            const int bound_k = shadow_bound(ln, k);
            HIP_TRY(c, hipEventRecord(ln.ev_pre[(size_t)k], st));
            HIP_TRY(c, hipStreamWaitEvent(ln.side, ln.ev_pre[(size_t)k], 0));
            if (bound_k > 0) {
                rc = timed_trace(ln.side, [&] { rt_launch_trace(ln.side, c->dev, lm.view(), 0, nullptr, sb_k, bound_k, counts + 4 * (max_depth + 2) + k, ln.d_overflow_side.p, d_counters); });
                if (rc != RT_OK) return rc;
                ++trace_launches;
            }
            rt_launch_shade_post(ln.side, c->dev, dc, lm.view(), n, ndev(k), sb_k);
            HIP_TRY(c, hipEventRecord(ln.ev_post[(size_t)k], ln.side));
            return RT_OK;
        }
        if (!dev && k < max_depth) {
            HIP_TRY(c, hipMemcpyAsync(ln.h_count, counts + k + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipEventRecord(ln.ev_count, st));
        }
        // the ray statistics of level k and the lighting of level k-1 (which needs only this launch's shadow results) are queued
        // AFTER the count copy: they keep the GPU busy while the host makes its round trip for the child count (measured gap
        // without them: 21-27 us per level)
        // host-counted frames only: frames whose lengths stay on the device get the rays per kind from the level counters afterwards
        if (stats && !dev) rt_launch_count_kinds(st, lm.view(), n, ndev(k), shadow_view(ln, counts, k), c->d_kind.p);
        if (k > 0) rt_launch_shade_post(st, c->dev, dc, ln.levels[k - 1].view(), ln.levels[k - 1].count, ndev(k - 1), sb_prev);
        return RT_OK;
    };

    // What the levels of a batch held, as a fraction of its primary samples: the next frame sizes its queues and launches from
    // these (c->level_ratio), so that no level length has to travel to the host between raygen and resolve.
    std::vector<double> seen_ratio((size_t)max_depth + 2, 0.0);
    auto note_counts = [&](int n0, const int *level_count) {
        for (int k = 1; k <= max_depth + 1; ++k) seen_ratio[(size_t)k] = std::max(seen_ratio[(size_t)k], n0 > 0 ? (double)level_count[k] / (double)n0 : 0.0);
    };

    // One pass over the frame.  predicted = false: the host reads every level's child count before it sizes and launches the
    // next level (one round trip per level and batch).  predicted = true: sizes and launch bounds come from the previous frame
    // of this scene; lengths stay on the device; a level that outgrows its queue drops the excess children and the pass
    // returns 1 (the caller then renders the frame again, host-counted).
    uint64_t level_syncs = 0, frame_passes = 0;
    uint64_t kinds_host[RT_RAY_KINDS] = {0, 0, 0, 0};   // rays per kind of a frame whose level lengths stayed on the device
    bool kinds_on_host = false;
    // the context stream continues after every batch stream has drained; the end of the frame's kernels for the timing
    auto drain_lanes = [&]() -> hipError_t {
        for (int j = 0; j < n_streams; ++j) {
            hipError_t e = hipEventRecord(c->lanes[j].ev_done, c->lanes[j].stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, c->lanes[j].ev_done, 0);
            if (e != hipSuccess) return e;
        }
        return timing ? hipEventRecord(ev_last, s) : hipSuccess;
    };
    auto run_frame = [&](bool predicted) -> int {
        ++frame_passes;
        kinds_on_host = predicted;
        std::fill(kinds_host, kinds_host + RT_RAY_KINDS, (uint64_t)0);
        HIP_TRY(c, hipMemsetAsync(c->d_kind.p, 0, RT_RAY_KINDS * sizeof(unsigned long long), s));
        if (d_counters) {
            HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(DevCounters), s));
            if (getenv("RT_TRACE_DEBUG")) {
                static const unsigned long long one = 1;
                HIP_TRY(c, hipMemcpyAsync(&d_counters->want_hist, &one, sizeof(one), hipMemcpyHostToDevice, s));
            }
        }
        trace_events.clear();
        c->ev_used = 0;
        primary_rays = trace_launches = trace_rays = 0;
        std::fill(seen_ratio.begin(), seen_ratio.end(), 0.0);
        if (timing) { HIP_TRY(c, c->timing_event(&ev_first)); HIP_TRY(c, c->timing_event(&ev_last)); HIP_TRY(c, hipEventRecord(ev_first, s)); }
        // the batch streams start after the uploads / clears queued on the context stream
        HIP_TRY(c, hipEventRecord(c->ev_ready, s));
        for (int j = 0; j < n_streams; ++j) { HIP_TRY(c, hipStreamWaitEvent(c->lanes[j].stream, c->ev_ready, 0)); c->lanes[j].busy = false; }
        int64_t next_p0 = 0;
        if (predicted) {
            const int64_t n_batches = (n_owned + batch_pixels - 1) / batch_pixels;
            const size_t slots = (size_t)((n_batches + n_streams - 1) / n_streams);
            struct Rec { int lane, slot, n0; std::vector<int> est; };
            std::vector<Rec> recs;
            for (int j = 0; j < n_streams; ++j) HIP_TRY(c, c->lanes[j].d_counts.ensure(n_cnt * std::max<size_t>(1, slots)));
            // Batches that run side by side (one per stream) are enqueued level by level, so that no stream's launches wait for the
            // host to finish another stream's whole chain; more batches than streams: one batch after the other.
            const bool level_major = n_batches > 1 && n_batches <= (int64_t)n_streams;
            std::vector<int> lasts;
            const double margin = c->wide_margin_frames > 0 ? 1.5 : 1.0625;
            if (c->wide_margin_frames > 0) --c->wide_margin_frames;
            {   // all queues of all streams are sized for the largest batch BEFORE the first launch: nothing is allocated (and so
                // nothing synchronises the device) between ray generation and resolve
                const int64_t n0_max = std::min<int64_t>(batch_pixels, n_owned) * spp;
                double prev = (double)n0_max;
                for (int k = 0; k <= max_depth; ++k) {
                    double est = prev;
                    if (k > 0) {
                        const double r = c->level_ratio[(size_t)k];
                        if (!(r > 0.0) || prev == 0) break;
                        est = std::min(std::min(r * (double)n0_max * margin + 4096.0, 2.0 * prev), (double)0x3fffffff);
                    }
                    const size_t nsh = (size_t)est * (size_t)std::max(1, nl);
                    if (nsh > 0x3fffffff) { c->error = "shadow queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
                    for (int j = 0; j < n_streams; ++j) {
                        HIP_TRY(c, c->lanes[j].levels[(size_t)k].ensure((size_t)est));
                        HIP_TRY(c, c->lanes[j].sh[k & 1].ensure(nsh));
                    }
                    prev = (double)(int)est;
                }
            }
            for (int64_t b = 0; b < n_batches; ++b) {
                StreamLane &ln = c->lanes[(size_t)(b % n_streams)];
                const int slot = (int)(b / n_streams);
                int *counts = ln.d_counts.p + (size_t)slot * n_cnt;
                ln.p0 = next_p0;
                ln.np = (int)std::min<int64_t>(batch_pixels, n_owned - next_p0);
#ifdef RT_TUNING_KNOBS
                if (level_major && n_batches == 2 && getenv("RT_SPLIT_PCT")) {       // experiment: unequal halves (the second gets this share)
                    const int64_t second = std::max<int64_t>(256, n_owned * std::min(90, std::max(5, atoi(getenv("RT_SPLIT_PCT")))) / 100 / 256 * 256);
                    ln.np = (int)(b == 0 ? n_owned - second : n_owned - next_p0);
                }
#endif
                next_p0 += ln.np;
                const int n0 = ln.np * spp;
                primary_rays += (uint64_t)n0;
                // bounds of every level from the previous frame's ratios (+ 1/16 + 4096), never above what the level above can spawn
                Rec rec; rec.lane = (int)(b % n_streams); rec.slot = slot; rec.n0 = n0;
                rec.est.assign((size_t)max_depth + 2, 0);
                rec.est[0] = n0;
                for (int k = 1; k <= max_depth; ++k) {
                    const double r = c->level_ratio[(size_t)k];
                    if (!(r > 0.0) || rec.est[(size_t)k - 1] == 0) break;
                    const double want = r * (double)n0 * margin + 4096.0;
                    rec.est[(size_t)k] = (int)std::min<double>(std::min<double>(want, 2.0 * (double)rec.est[(size_t)k - 1]), (double)0x3fffffff);
                }
                int last = 0;
                for (int k = 0; k <= max_depth; ++k) {
                    if (rec.est[(size_t)k] == 0) break;
                    last = k;
                    HIP_TRY(c, ln.levels[(size_t)k].ensure((size_t)rec.est[(size_t)k]));
                    ln.levels[(size_t)k].count = rec.est[(size_t)k];
                    const size_t nsh = (size_t)rec.est[(size_t)k] * (size_t)std::max(1, nl);
                    if (nsh > 0x3fffffff) { c->error = "shadow queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
                    HIP_TRY(c, ln.sh[k & 1].ensure(nsh));
                }
                ln.levels[(size_t)last + 1].count = 0;
                rt_launch_raygen(ln.stream, dc, c->d_pixels.p, (int)ln.p0, n0, ln.levels[0].view());
                rt_launch_fill_int(ln.stream, counts, (int)n_cnt, 0);
                lasts.push_back(last);
                if (!level_major) {
                    for (int k = 0; k <= last; ++k) {
                        ln.k = k;
                        int rc = enqueue_level(ln, counts, true, k < last ? rec.est[(size_t)k + 1] : 0);
                        if (rc != RT_OK) return rc;
                    }
                    int rc = finish_batch(ln, counts, last, true);
                    if (rc != RT_OK) return rc;
                }
                recs.push_back(std::move(rec));
            }
            if (level_major) {
                for (int k = 0; k <= max_depth; ++k)
                    for (size_t b = 0; b < recs.size(); ++b) {
                        StreamLane &ln = c->lanes[(size_t)recs[b].lane];
                        int *counts = ln.d_counts.p + (size_t)recs[b].slot * n_cnt;
                        if (k <= lasts[b]) {
                            ln.k = k;
                            int rc = enqueue_level(ln, counts, true, k < lasts[b] ? recs[b].est[(size_t)k + 1] : 0);
                            if (rc != RT_OK) return rc;
                        }
                        if (k == lasts[b]) { int rc = finish_batch(ln, counts, lasts[b], true); if (rc != RT_OK) return rc; }
                    }
            }
            HIP_TRY(c, drain_lanes());
            // one read-back per frame: every batch's counters, after the streams have drained
            std::vector<std::vector<int>> host_counts((size_t)n_streams);
            for (int j = 0; j < n_streams; ++j) {
                host_counts[(size_t)j].resize(n_cnt * std::max<size_t>(1, slots));
                HIP_TRY(c, hipMemcpyAsync(host_counts[(size_t)j].data(), c->lanes[j].d_counts.p, host_counts[(size_t)j].size() * sizeof(int), hipMemcpyDeviceToHost, c->lanes[j].stream));
            }
            for (int j = 0; j < n_streams; ++j) HIP_TRY(c, hipStreamSynchronize(c->lanes[j].stream));
            bool overflow = false;
            for (const Rec &r : recs) {
                const int *hc = host_counts[(size_t)r.lane].data() + (size_t)r.slot * n_cnt;
                for (int k = 1; k <= max_depth + 1; ++k) overflow = overflow || hc[k] > r.est[(size_t)k];
                note_counts(r.n0, hc);
                trace_rays += (uint64_t)r.n0;
                for (int k = 1; k <= max_depth; ++k) trace_rays += (uint64_t)std::max(0, hc[k]);
                for (int k = 0; k <= max_depth; ++k) trace_rays += (uint64_t)std::max(0, hc[3 * (max_depth + 2) + k]);
                // rays per kind: a level's children are its refraction + reflection rays, its shadow queue its shadow rays
                for (int k = 0; k <= max_depth; ++k) {
                    const uint64_t refr = (uint64_t)std::max(0, hc[2 * (max_depth + 2) + k]), kids = (uint64_t)std::max(0, hc[k + 1]);
                    kinds_host[RT_RAY_REFRACT] += refr;
                    kinds_host[RT_RAY_REFLECT] += kids > refr ? kids - refr : 0;
                    kinds_host[RT_RAY_SHADOW] += (uint64_t)std::max(0, hc[3 * (max_depth + 2) + k]);
                }
            }
            return overflow ? 1 : RT_OK;
        }
        std::vector<int> cur_counts((size_t)n_streams * ((size_t)max_depth + 2), 0);
        for (;;) {
            bool any = false;
            for (int j = 0; j < n_streams; ++j) {
                StreamLane &ln = c->lanes[j];
                int *lane_counts = cur_counts.data() + (size_t)j * ((size_t)max_depth + 2);
                auto batch_done = [&]() { note_counts(ln.np * spp, lane_counts); };
                if (!ln.busy) {
                    if (next_p0 >= n_owned) continue;
                    // start the next batch on this stream
                    ln.p0 = next_p0;
                    ln.np = (int)std::min<int64_t>(batch_pixels, n_owned - next_p0);
                    next_p0 += ln.np;
                    const int n0 = ln.np * spp;
                    primary_rays += (uint64_t)n0;
                    HIP_TRY(c, ln.levels[0].ensure((size_t)n0));
                    ln.levels[0].count = n0;
                    std::fill(lane_counts, lane_counts + max_depth + 2, 0);
                    rt_launch_raygen(ln.stream, dc, c->d_pixels.p, (int)ln.p0, n0, ln.levels[0].view());
                    HIP_TRY(c, hipMemsetAsync(ln.d_counts.p, 0, n_cnt * sizeof(int), ln.stream));
                    ln.k = 0;
                    ln.busy = true;
                    int rc = enqueue_level(ln, ln.d_counts.p, false, 0);
                    if (rc == RT_OK && max_depth == 0) { rc = finish_batch(ln, ln.d_counts.p, 0, false); batch_done(); }
                    if (rc != RT_OK) return rc;
                    any = true;
                    continue;
                }
                // the lane waits for the child count of level ln.k
                any = true;
                HIP_TRY(c, hipEventSynchronize(ln.ev_count));
                ++level_syncs;
                const int child_count = *ln.h_count;
                ln.levels[ln.k + 1].count = child_count;
                lane_counts[ln.k + 1] = child_count;
                int rc;
                if (child_count == 0) { rc = finish_batch(ln, ln.d_counts.p, ln.k, false); batch_done(); }
                else {
                    ++ln.k;
                    rc = enqueue_level(ln, ln.d_counts.p, false, 0);
                    if (rc == RT_OK && ln.k == max_depth) { rc = finish_batch(ln, ln.d_counts.p, ln.k, false); batch_done(); }
                }
                if (rc != RT_OK) return rc;
            }
            if (!any) break;
        }
        HIP_TRY(c, drain_lanes());
        return RT_OK;
    };

    // rt_render_opts.level_counts = 1: the host reads every level's length (one round trip per level); 0: lengths stay on the device
    // whenever the scene has been rendered before.  After a frame whose prediction failed the next frames get a wider margin.
    const bool can_predict = c->level_ratio.size() == (size_t)max_depth + 2 && opts->level_counts != 1;
    const bool want_tl = getenv("RT_TIMELINE") != nullptr;
    if (want_tl) rt_tl_begin();
    int frc = run_frame(can_predict);
    if (want_tl) rt_tl_end();
    if (frc == 1) {
        // a level outgrew the queue its prediction had sized (new camera, other sample count ...): once more, host-counted
        for (int j = 0; j < n_streams; ++j) HIP_TRY(c, hipStreamSynchronize(c->lanes[j].stream));
        ++c->mispredicted_frames;
        c->wide_margin_frames = 8;           // a sequence whose ray counts keep growing should not render every frame twice
        frc = run_frame(false);
    }
    if (frc != RT_OK) {
        // leave no launch of a failed frame in flight: they write caller-owned buffers and the context's queues
        for (int j = 0; j < n_streams; ++j) {
            if (c->lanes[j].side) hipStreamSynchronize(c->lanes[j].side);
            if (c->lanes[j].stream) hipStreamSynchronize(c->lanes[j].stream);
        }
        hipStreamSynchronize(s);
        return frc;
    }
    c->level_ratio = seen_ratio;
    if (!opts->output_on_device) {
        HIP_TRY(c, hipMemcpyAsync(hdr_rgb, d_hdr, 3 * n_out * sizeof(float), hipMemcpyDeviceToHost, s));
        if (prim_id) HIP_TRY(c, hipMemcpyAsync(prim_id, d_pid, n_out * sizeof(int), hipMemcpyDeviceToHost, s));
        if (prim_t) HIP_TRY(c, hipMemcpyAsync(prim_t, d_pt, n_out * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    unsigned long long kinds[RT_RAY_KINDS] = {0, 0, 0, 0};
    DevCounters hc;
    std::memset(&hc, 0, sizeof(hc));
    if (stats) {
        HIP_TRY(c, hipMemcpyAsync(kinds, c->d_kind.p, sizeof(kinds), hipMemcpyDeviceToHost, s));
        if (d_counters) HIP_TRY(c, hipMemcpyAsync(&hc, d_counters, sizeof(hc), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { c->error = std::string("kernel launch failed: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays[RT_RAY_PRIMARY] = primary_rays;
        if (kinds_on_host) for (int q = 0; q < RT_RAY_KINDS; ++q) kinds[q] = kinds_host[q];
        stats->rays[RT_RAY_SHADOW] = kinds[RT_RAY_SHADOW]; stats->rays[RT_RAY_REFLECT] = kinds[RT_RAY_REFLECT]; stats->rays[RT_RAY_REFRACT] = kinds[RT_RAY_REFRACT];
        stats->node_visits = hc.node_visits; stats->box_tests = hc.box_tests; stats->tri_tests = hc.tri_tests; stats->sphere_tests = hc.sphere_tests;
        if (getenv("RT_TRACE_DEBUG")) {
            for (int q = 0; q < 2; ++q) {
                fprintf(stderr, "[rt] trace lanes %s:", q ? "shadow" : "closest");
                for (int k = 0; k < 14; ++k) fprintf(stderr, " %llu", hc.dbg[q][k]);
                fprintf(stderr, "\n[rt] steps histogram (log2 buckets):");
                for (int k = 0; k < 16; ++k) fprintf(stderr, " %llu", hc.hist[q * 16 + k]);
                fprintf(stderr, "\n[rt] ns per node step by bucket:");
                for (int k = 0; k < 16; ++k) fprintf(stderr, " %.0f", hc.hist[64 + q * 16 + k] ? 10.0 * hc.hist[32 + q * 16 + k] / hc.hist[64 + q * 16 + k] : 0.0);
                fprintf(stderr, "\n");
            }
        }
        stats->trace_launches = trace_launches;
        stats->trace_rays = stats->rays[RT_RAY_PRIMARY] + stats->rays[RT_RAY_SHADOW] + stats->rays[RT_RAY_REFLECT] + stats->rays[RT_RAY_REFRACT];
        (void)trace_rays;
        stats->level_syncs = level_syncs; stats->frame_passes = frame_passes;
        if (timing) {
            float ms = 0;
            hipEventElapsedTime(&ms, ev_first, ev_last);
            stats->ms_kernels = ms;
            double tr = 0;
            std::vector<std::pair<float, float>> iv;
            for (auto &pr : trace_events) {
                float m = 0, a = 0;
                hipEventElapsedTime(&m, pr.first, pr.second);
                hipEventElapsedTime(&a, ev_first, pr.first);
                tr += m;
                iv.push_back({a, a + m});
            }
            stats->ms_trace = tr;
            std::sort(iv.begin(), iv.end());
            double busy = 0; float cur_a = 0, cur_b = -1;
            for (auto &x : iv) {
                if (cur_b < cur_a || x.first > cur_b) { if (cur_b >= cur_a) busy += cur_b - cur_a; cur_a = x.first; cur_b = x.second; }
                else cur_b = std::max(cur_b, x.second);
            }
            if (cur_b >= cur_a) busy += cur_b - cur_a;
            stats->ms_trace_busy = busy;
        }
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return RT_OK;
}

int rt_intersect_batch(rt_ctx *c, const rt_ray *rays, size_t n, rt_hit *hits, int on_device, rt_stats *stats) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!rays || !hits)) { c->error = "bad ray batch"; return RT_ERR_INVALID; }
    if (n > RT_MAX_BATCH_RAYS) { c->error = "more than RT_MAX_BATCH_RAYS rays in one call"; return RT_ERR_TOO_LARGE; }
    if (!c->has_scene) { c->error = "no scene uploaded"; return RT_ERR_NO_SCENE; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    apply_tuning_knobs(c);
    hipStream_t s = c->stream;
    const rt_ray *dr = rays; rt_hit *dh = hits;
    if (!on_device) {
        HIP_TRY(c, c->d_rays.ensure(std::max<size_t>(1, n))); HIP_TRY(c, c->d_hits.ensure(std::max<size_t>(1, n)));
        if (n) HIP_TRY(c, hipMemcpyAsync(c->d_rays.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, s));
        dr = c->d_rays.p; dh = c->d_hits.p;
    }
    DevCounters *d_counters = nullptr;
    if (stats) {
        HIP_TRY(c, c->d_counters.ensure(1));
        HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(DevCounters), s));
        d_counters = c->d_counters.p;
    }
    hipEvent_t a = nullptr, b = nullptr;
    if (stats) { HIP_TRY(c, hipEventCreate(&a)); HIP_TRY(c, hipEventCreate(&b)); HIP_TRY(c, hipEventRecord(a, s)); }
    HIP_TRY(c, c->d_counts.ensure(4));
    HIP_TRY(c, hipMemsetAsync(c->d_counts.p, 0, sizeof(int), s));
    rt_launch_trace_batch(s, c->dev, dr, dh, (int)n, c->d_counts.p, c->d_overflow.p, d_counters);
    if (stats) HIP_TRY(c, hipEventRecord(b, s));
    if (!on_device && n) HIP_TRY(c, hipMemcpyAsync(hits, c->d_hits.p, n * sizeof(rt_hit), hipMemcpyDeviceToHost, s));
    DevCounters hc;
    std::memset(&hc, 0, sizeof(hc));
    if (stats) HIP_TRY(c, hipMemcpyAsync(&hc, d_counters, sizeof(hc), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { c->error = std::string("kernel launch failed: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays[RT_RAY_PRIMARY] = n;
        stats->node_visits = hc.node_visits; stats->box_tests = hc.box_tests; stats->tri_tests = hc.tri_tests; stats->sphere_tests = hc.sphere_tests;
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        stats->ms_kernels = stats->ms_trace = ms; stats->trace_launches = 1; stats->trace_rays = n;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    return RT_OK;
}

size_t rt_noalloc_scratch_bytes(void) { return sizeof(int) * ((size_t)NOALLOC_HEADER_INTS + NOALLOC_OVERFLOW_INTS); }

int rt_intersect_batch_noalloc(rt_ctx *c, const rt_ray *d_rays, size_t n, rt_hit *d_hits, void *hip_stream) {
    if (!c) return RT_ERR_INVALID;
    if ((n && (!d_rays || !d_hits))) { c->error = "rt_intersect_batch_noalloc: NULL ray or hit buffer"; return RT_ERR_INVALID; }
    if (n > RT_MAX_BATCH_RAYS) { c->error = "rt_intersect_batch_noalloc: more than RT_MAX_BATCH_RAYS rays in one call"; return RT_ERR_TOO_LARGE; }
    if (!c->has_scene) { c->error = "no scene uploaded"; return RT_ERR_NO_SCENE; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    int *ptr = c->device < RT_MAX_DEVICES ? g_noalloc_scratch[c->device] : nullptr;
    if (!ptr) { c->error = "rt_intersect_batch_noalloc: no scratch address for this device"; return RT_ERR_INVALID; }
    // the overflow area holds `depth` stack entries for each persistent thread: fewer blocks for a deeper tree
    const int depth = std::max(1, c->overflow_depth), tb = rt_trace_block_threads();
    const int64_t fit = (int64_t)NOALLOC_OVERFLOW_INTS / ((int64_t)depth * tb);
    if (fit < 1) { c->error = "rt_intersect_batch_noalloc: the scene's BVH is too deep for the fixed scratch (stack overflow depth " + std::to_string(depth) + ")"; return RT_ERR_TOO_LARGE; }
    const int blocks = (int)std::min<int64_t>(fit, rt_trace_resident_blocks());
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // Overlapping calls on one GPU share the scratch.  Calls on one stream are ordered; a call on another stream while the latest
    // one is still running is refused.  (A stream under graph capture executes nothing now: its replays are the caller's to order.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && g_noalloc_pending[c->device] && g_noalloc_stream[c->device] != s && hipEventQuery(g_noalloc_done[c->device]) == hipErrorNotReady) {
        c->error = "rt_intersect_batch_noalloc: an earlier call on another stream of this GPU is still running (the calls share one scratch buffer)";
        return RT_ERR_INVALID;
    }
    (void)hipGetLastError();
    if (n) {
        rt_launch_fill_int(s, ptr, 1, 0);        // the fetch counter starts at zero; the overflow columns need no clearing (written before read)
        rt_launch_trace_batch(s, c->dev, d_rays, d_hits, (int)n, ptr, ptr + NOALLOC_HEADER_INTS, nullptr, blocks, blocks * tb);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->error = std::string("kernel launch failed: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    if (!capturing && n) {
        HIP_TRY(c, hipEventRecord(g_noalloc_done[c->device], s));
        g_noalloc_stream[c->device] = s; g_noalloc_pending[c->device] = true;
    }
    if (!hip_stream) HIP_TRY(c, hipStreamSynchronize(s));
    return RT_OK;
}

// ---- frame post-processing ---------------------------------------------------------------------------
static int ppm_header(char *buf, size_t cap, int format, int nx, int ny) {
    return std::snprintf(buf, cap, "P%d\n%d %d\n255\n", format == RT_IMG_PPM_P3 ? 3 : 6, nx, ny);   // Image.cpp:59-61
}

// Everything of an EXR file before the scan-line blocks, as SaveEXRImageToMemory (reference src/tinyexr.h:12126-12266,
// 12569-12581) writes it for Tonemapper::SaveEXR's header (src/Tonemapper.cpp:34-99): magic, version, the attributes
// channels (B, G, R as HALF) / compression (NONE: InitEXRHeader zeroes it) / dataWindow / displayWindow / lineOrder /
// pixelAspectRatio / screenWindowCenter / screenWindowWidth, the terminator and the offset table (one block per row).
static std::vector<unsigned char> exr_prefix(int nx, int ny) {
    std::vector<unsigned char> m;
    auto bytes = [&](const void *p, size_t n) { const unsigned char *b = static_cast<const unsigned char *>(p); m.insert(m.end(), b, b + n); };
    auto attr = [&](const char *name, const char *type, const void *data, int len) {
        bytes(name, std::strlen(name) + 1); bytes(type, std::strlen(type) + 1); bytes(&len, 4); bytes(data, (size_t)len);
    };
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    bytes(magic, 8);
    std::vector<unsigned char> ch;
    for (const char *name : {"B", "G", "R"}) {
        const int half = 1, one = 1;                       // TINYEXR_PIXELTYPE_HALF, sampling 1
        const unsigned char plinear[4] = {0, 0, 0, 0};
        ch.insert(ch.end(), name, name + 2);
        const unsigned char *q = reinterpret_cast<const unsigned char *>(&half); ch.insert(ch.end(), q, q + 4);
        ch.insert(ch.end(), plinear, plinear + 4);
        q = reinterpret_cast<const unsigned char *>(&one); ch.insert(ch.end(), q, q + 4); ch.insert(ch.end(), q, q + 4);
    }
    ch.push_back(0);
    attr("channels", "chlist", ch.data(), (int)ch.size());
    const unsigned char comp = 0, line_order = 0;
    attr("compression", "compression", &comp, 1);
    const int window[4] = {0, 0, nx - 1, ny - 1};
    attr("dataWindow", "box2i", window, 16);
    attr("displayWindow", "box2i", window, 16);
    attr("lineOrder", "lineOrder", &line_order, 1);
    const float aspect = 1.0f, center[2] = {0.0f, 0.0f}, width = (float)nx;
    attr("pixelAspectRatio", "float", &aspect, 4);
    attr("screenWindowCenter", "v2f", center, 8);
    attr("screenWindowWidth", "float", &width, 4);
    m.push_back(0);
    const uint64_t first = (uint64_t)m.size() + 8ull * (uint64_t)ny, row = 8ull + 6ull * (uint64_t)nx;
    for (int y = 0; y < ny; ++y) { const uint64_t off = first + row * (uint64_t)y; bytes(&off, 8); }
    return m;
}

size_t rt_image_encode_bound(int nx, int ny, int format) {
    if (nx <= 0 || ny <= 0) return 0;
    const size_t npx = (size_t)nx * (size_t)ny;
    switch (format) {
    case RT_IMG_RGB8: return 3 * npx;
    case RT_IMG_PPM_P6: return 3 * npx + 40;
    case RT_IMG_PPM_P3: return 12 * npx + (size_t)ny + 40;
    case RT_IMG_EXR_HALF: return exr_prefix(nx, ny).size() + (size_t)ny * (8 + 6 * (size_t)nx);
    default: return 0;
    }
}

namespace {
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool on = false;
    hipError_t start(bool want, hipStream_t s) {
        on = want;
        if (!on) return hipSuccess;
        hipError_t e;
        if ((e = hipEventCreate(&a)) != hipSuccess) return e;
        if ((e = hipEventCreate(&b)) != hipSuccess) return e;
        return hipEventRecord(a, s);
    }
    hipError_t stop(hipStream_t s) { return on ? hipEventRecord(b, s) : hipSuccess; }
    double ms() const { float m = 0; if (on) hipEventElapsedTime(&m, a, b); return m; }
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
};
}  // namespace

int rt_image_encode(rt_ctx *c, const float *hdr_rgb, int nx, int ny, int hdr_on_device, int format, void *out_host, size_t out_capacity,
                    size_t *out_bytes, double *ms_kernels) {
    if (!c) return RT_ERR_INVALID;
    if (out_bytes) *out_bytes = 0;
    if (!hdr_rgb || !out_host || nx <= 0 || ny <= 0 || (format != RT_IMG_RGB8 && format != RT_IMG_PPM_P3 && format != RT_IMG_PPM_P6 && format != RT_IMG_EXR_HALF)) {
        c->error = "rt_image_encode: bad argument"; return RT_ERR_INVALID;
    }
    const int64_t npx = (int64_t)nx * ny;
    if (npx > 0x7fffffff / 4) { c->error = "rt_image_encode: frame larger than 2^29 pixels"; return RT_ERR_TOO_LARGE; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    hipStream_t s = c->stream;
    const float *d_hdr = hdr_rgb;
    if (!hdr_on_device) {
        HIP_TRY(c, c->d_img_in.ensure((size_t)npx * 3));
        HIP_TRY(c, hipMemcpyAsync(c->d_img_in.p, hdr_rgb, (size_t)npx * 3 * sizeof(float), hipMemcpyHostToDevice, s));
        d_hdr = c->d_img_in.p;
    }
    if (format == RT_IMG_EXR_HALF) {
        const std::vector<unsigned char> prefix = exr_prefix(nx, ny);
        const size_t body = (size_t)ny * (8 + 6 * (size_t)nx);
        if (out_bytes) *out_bytes = prefix.size() + body;
        if (out_capacity < prefix.size() + body) { c->error = "rt_image_encode: output buffer too small (see rt_image_encode_bound)"; return RT_ERR_INVALID; }
        HIP_TRY(c, c->d_img_text.ensure(body + 16));
        EventPair evx;
        HIP_TRY(c, evx.start(ms_kernels != nullptr, s));
        HIP_TRY(c, rt_img_exr_rows(s, d_hdr, nx, ny, c->d_img_text.p));
        HIP_TRY(c, evx.stop(s));
        std::memcpy(out_host, prefix.data(), prefix.size());
        HIP_TRY(c, hipMemcpyAsync(static_cast<uint8_t *>(out_host) + prefix.size(), c->d_img_text.p, body, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (ms_kernels) *ms_kernels = evx.ms();
        return RT_OK;
    }
    HIP_TRY(c, c->d_img8.ensure((size_t)npx * 3));
    char header[48];
    const int hlen = format == RT_IMG_RGB8 ? 0 : ppm_header(header, sizeof(header), format, nx, ny);
    uint8_t *out = static_cast<uint8_t *>(out_host);
    EventPair ev;
    if (format != RT_IMG_PPM_P3) {
        if (out_capacity < (size_t)hlen + (size_t)npx * 3) { c->error = "rt_image_encode: output buffer too small"; return RT_ERR_INVALID; }
        HIP_TRY(c, ev.start(ms_kernels != nullptr, s));
        HIP_TRY(c, rt_img_resolve8(s, d_hdr, npx * 3, c->d_img8.p));
        HIP_TRY(c, ev.stop(s));
        std::memcpy(out, header, (size_t)hlen);
        HIP_TRY(c, hipMemcpyAsync(out + hlen, c->d_img8.p, (size_t)npx * 3, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (out_bytes) *out_bytes = (size_t)hlen + (size_t)npx * 3;
        if (ms_kernels) *ms_kernels = ev.ms();
        return RT_OK;
    }
    const int64_t nb = rt_img_p3_blocks(npx);
    size_t scan_bytes = 0;
    HIP_TRY(c, rt_img_p3_scan_bytes(npx, &scan_bytes));
    HIP_TRY(c, c->d_img_tmp.ensure(scan_bytes + 16));
    HIP_TRY(c, c->d_img_len.ensure((size_t)nb + 1));
    HIP_TRY(c, c->d_img_off.ensure((size_t)nb + 1));
    HIP_TRY(c, c->d_img_text.ensure(12 * (size_t)npx + (size_t)ny + 16));
    HIP_TRY(c, ev.start(ms_kernels != nullptr, s));
    HIP_TRY(c, rt_img_resolve8(s, d_hdr, npx * 3, c->d_img8.p));
    HIP_TRY(c, rt_img_p3_measure(s, c->d_img8.p, npx, nx, c->d_img_len.p, c->d_img_off.p, c->d_img_tmp.p, scan_bytes));
    HIP_TRY(c, rt_img_p3_write(s, c->d_img8.p, npx, nx, c->d_img_off.p, c->d_img_text.p));
    HIP_TRY(c, ev.stop(s));
    unsigned long long body = 0;
    HIP_TRY(c, hipMemcpyAsync(&body, c->d_img_off.p + nb, sizeof(body), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (out_bytes) *out_bytes = (size_t)hlen + (size_t)body;
    if (out_capacity < (size_t)hlen + (size_t)body) { c->error = "rt_image_encode: output buffer too small (see rt_image_encode_bound)"; return RT_ERR_INVALID; }
    std::memcpy(out, header, (size_t)hlen);
    HIP_TRY(c, hipMemcpyAsync(out + hlen, c->d_img_text.p, (size_t)body, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (ms_kernels) *ms_kernels = ev.ms();
    return RT_OK;
}

int rt_tonemap(rt_ctx *c, const float *hdr_rgb, int nx, int ny, const rt_tonemap_opts *o, float *out_rgb, float *out_params, double *ms_kernels) {
    if (!c) return RT_ERR_INVALID;
    if (!o || o->abi_version != RT_ABI_VERSION) { c->error = "rt_tonemap: abi_version mismatch"; return RT_ERR_ABI; }
    if (!hdr_rgb || !out_rgb || nx <= 0 || ny <= 0) { c->error = "rt_tonemap: bad argument"; return RT_ERR_INVALID; }
    const int64_t npx = (int64_t)nx * ny;
    if (npx > 0x7fffffff / 4) { c->error = "rt_tonemap: frame larger than 2^29 pixels"; return RT_ERR_TOO_LARGE; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    hipStream_t s = c->stream;
    const float *d_hdr = hdr_rgb;
    if (!o->in_on_device) {
        HIP_TRY(c, c->d_img_in.ensure((size_t)npx * 3));
        HIP_TRY(c, hipMemcpyAsync(c->d_img_in.p, hdr_rgb, (size_t)npx * 3 * sizeof(float), hipMemcpyHostToDevice, s));
        d_hdr = c->d_img_in.p;
    }
    float *d_out = out_rgb;
    if (!o->out_on_device) { HIP_TRY(c, c->d_tm_out.ensure((size_t)npx * 3)); d_out = c->d_tm_out.p; }
    const size_t n_part = (size_t)rt_img_p3_blocks(npx);
    HIP_TRY(c, c->d_tm_lum.ensure((size_t)npx));
    HIP_TRY(c, c->d_tm_partial.ensure(n_part)); HIP_TRY(c, c->d_tm_params.ensure(2));
    HIP_TRY(c, c->d_img_tmp.ensure((n_part + RT_TM_WORK_WORDS) * sizeof(unsigned)));
    unsigned *d_pmax = reinterpret_cast<unsigned *>(c->d_img_tmp.p), *d_work = d_pmax + n_part;
    if (o->input_8bit) HIP_TRY(c, c->d_img8.ensure((size_t)npx * 3));
    EventPair ev;
    HIP_TRY(c, ev.start(ms_kernels != nullptr, s));
    const void *src = d_hdr;
    if (o->input_8bit) { HIP_TRY(c, rt_img_resolve8(s, d_hdr, npx * 3, c->d_img8.p)); src = c->d_img8.p; }   // the reference tone-maps its 8-bit Image (Q20)
    HIP_TRY(c, rt_img_tonemap(s, src, o->input_8bit ? 1 : 0, npx, o->key, o->burn_percent, o->saturation, c->d_tm_lum.p, c->d_tm_partial.p, d_pmax, d_work,
                              c->d_tm_params.p, d_out));
    HIP_TRY(c, ev.stop(s));
    if (!o->out_on_device) HIP_TRY(c, hipMemcpyAsync(out_rgb, d_out, (size_t)npx * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out_params) HIP_TRY(c, hipMemcpyAsync(out_params, c->d_tm_params.p, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (ms_kernels) *ms_kernels = ev.ms();
    return RT_OK;
}

// ---- PLY body -> positions + triangles (rt_ply.hip) -----------------------------------------------------
// Decodes into the context's device arrays d_ply_pos [3 nv] / d_ply_tri [3 n_tris]; the callers below copy them out.
static int ply_decode_device(rt_ctx *c, const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, const rt_ply_layout *L, int64_t *out_n_tris,
                             double *ms_kernels) {
    if (!L || L->abi_version != RT_ABI_VERSION) { c->error = "rt_ply_decode: abi_version mismatch"; return RT_ERR_ABI; }
    const int64_t nv = L->n_vertices, nf = L->n_faces;
    const bool ascii = L->format == RT_PLY_ASCII;
    auto size_ok = [](int v) { return v == 1 || v == 2 || v == 4; };
    if (nv < 0 || nf < 0 || !out_n_tris || (nv > 0 && !vertex_bytes) || (nf > 0 && !ascii && !face_bytes) || (L->pos_type != 0 && L->pos_type != 1) ||
        (L->format != RT_PLY_BINARY_LE && L->format != RT_PLY_ASCII) || (!ascii && (!size_ok(L->count_size) || !size_ok(L->index_size))) || L->face_pre_bytes < 0 ||
        L->face_post_bytes < 0) { c->error = "rt_ply_decode: bad argument"; return RT_ERR_INVALID; }
    if (nv > 0x7fffffffLL || nf > 0x3fffffffLL) { c->error = "rt_ply_decode: more than 2^31 vertices or 2^30 faces"; return RT_ERR_TOO_LARGE; }
    HIP_TRY(c, hipSetDevice(c->device));
    rt_set_num_cus(c->num_cus);
    hipStream_t s = c->stream;
    HIP_TRY(c, c->d_ply_pos.ensure((size_t)nv * 3 + 1));
    HIP_TRY(c, c->d_ply_tri.ensure((size_t)nf * 6 + 1));
    HIP_TRY(c, c->d_ply_cnt.ensure(2 * ((size_t)nf + 1) + 4));
    int *d_flag = c->d_ply_cnt.p + 2 * ((size_t)nf + 1);
    HIP_TRY(c, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), s));
    EventPair ev;
    int64_t n_tris = 0;
    if (ascii) {
        // the whole body text: vertex lines, then face lines (src/happly.h:2075-2095)
        const size_t n_text = face_bytes_size;
        if (n_text > 0x7ffffff0ull) { c->error = "rt_ply_decode: ascii body larger than 2 GiB"; return RT_ERR_TOO_LARGE; }
        for (int k = 0; k < 3; ++k)
            if (L->pos_offset[k] < 0 || L->pos_offset[k] >= L->vertex_stride) { c->error = "rt_ply_decode: position token outside the vertex line"; return RT_ERR_INVALID; }
        const int64_t n_lines = nv + nf;
        // every coordinate may be a number the device cannot convert exactly (20 significant digits, huge exponents, subnormals): room for all
        const int hard_cap = (int)std::min<int64_t>(3 * nv, 0x1fffffff);
        size_t scan_bytes = 0, scan2 = 0;
        HIP_TRY(c, rt_ply_ascii_scan_bytes((int64_t)n_text, &scan_bytes));
        HIP_TRY(c, rt_ply_scan_bytes(nf, &scan2));
        HIP_TRY(c, c->d_ply_tmp.ensure(std::max(scan_bytes, scan2) + 16));
        HIP_TRY(c, c->d_ply_in.ensure(n_text + 16));
        HIP_TRY(c, c->d_ply_off.ensure((size_t)n_lines + 2));
        HIP_TRY(c, c->d_ply_flag.ensure(2 * (n_text + 1) + 1 + 3 * (size_t)hard_cap));
        int *d_nl = c->d_ply_flag.p, *d_scan = d_nl + (n_text + 1), *d_hard = d_scan + (n_text + 1);
        if (n_text) HIP_TRY(c, hipMemcpyAsync(c->d_ply_in.p, vertex_bytes, n_text, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemsetAsync(c->d_ply_off.p, 0, ((size_t)n_lines + 2) * sizeof(long long), s));
        HIP_TRY(c, hipMemsetAsync(d_hard, 0, sizeof(int), s));
        HIP_TRY(c, ev.start(ms_kernels != nullptr, s));
        HIP_TRY(c, rt_ply_ascii_lines(s, c->d_ply_in.p, (int64_t)n_text, d_nl, d_scan, c->d_ply_tmp.p, scan_bytes, n_lines, c->d_ply_off.p));
        HIP_TRY(c, rt_ply_ascii_vertices(s, c->d_ply_in.p, c->d_ply_off.p, (int64_t)n_text, nv, L->pos_offset, L->pos_type, c->d_ply_pos.p, d_hard, hard_cap, d_flag));
        int *d_ntri = c->d_ply_cnt.p, *d_toff = c->d_ply_cnt.p + (nf + 1);
        HIP_TRY(c, rt_ply_ascii_faces_count(s, c->d_ply_in.p, c->d_ply_off.p, (int64_t)n_text, nv, nf, L->face_pre_bytes, d_ntri, d_toff, c->d_ply_tmp.p, scan2, d_flag));
        HIP_TRY(c, rt_ply_ascii_faces_write(s, c->d_ply_in.p, c->d_ply_off.p, (int64_t)n_text, nv, nf, L->face_pre_bytes, d_toff, c->d_ply_tri.p, d_flag));
        HIP_TRY(c, ev.stop(s));
        int total = 0, bad = 0, n_lines_found = 0, n_hard = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, d_toff + nf, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&n_lines_found, d_scan + n_text, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&n_hard, d_hard, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const bool last_open = n_text > 0 && static_cast<const char *>(vertex_bytes)[n_text - 1] != '\n';
        if ((int64_t)n_lines_found + (last_open ? 1 : 0) < n_lines) { c->error = "rt_ply_decode: ascii body has fewer lines than vertices + faces"; return RT_ERR_INVALID; }
        if (bad & 1) { c->error = "rt_ply_decode: a vertex line has fewer values than the header declares"; return RT_ERR_INVALID; }
        if (bad & 2) { c->error = "rt_ply_decode: a face line is not a count followed by non-negative integer indices"; return RT_ERR_INVALID; }
        if (n_hard > hard_cap) { c->error = "rt_ply_decode: more unconverted numbers than the file has coordinates"; return RT_ERR_INVALID; }
        if (n_hard > 0) {
            // tokens the device could not prove correct: the reference library's own conversion (operator>> of the property's type),
            // all of them in one staging array, patched into the positions by one kernel
            std::vector<int> rec(3 * (size_t)n_hard);
            HIP_TRY(c, hipMemcpyAsync(rec.data(), d_hard + 1, rec.size() * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            const char *text = static_cast<const char *>(vertex_bytes);
            std::vector<float> fixed((size_t)n_hard);
            for (int h = 0; h < n_hard; ++h) {
                size_t a = (size_t)rec[3 * h + 2], b = a;
                while (b < n_text && text[b] != ' ' && text[b] != '\n' && text[b] != '\r') ++b;
                std::istringstream iss(std::string(text + a, b - a));
                if (L->pos_type) { double v = 0; iss >> v; fixed[(size_t)h] = (float)v; } else { float v = 0; iss >> v; fixed[(size_t)h] = v; }
            }
            HIP_TRY(c, c->d_ply_fix.ensure((size_t)n_hard));
            HIP_TRY(c, hipMemcpyAsync(c->d_ply_fix.p, fixed.data(), fixed.size() * sizeof(float), hipMemcpyHostToDevice, s));
            HIP_TRY(c, rt_ply_patch(s, d_hard, c->d_ply_fix.p, n_hard, c->d_ply_pos.p));
            HIP_TRY(c, hipStreamSynchronize(s));
        }
        n_tris = total;
        *out_n_tris = n_tris;
        if (ms_kernels) *ms_kernels = ev.ms();
        return RT_OK;
    }
    const int csz = L->pos_type ? 8 : 4;
    for (int k = 0; k < 3; ++k)
        if (L->pos_offset[k] < 0 || L->pos_offset[k] + csz > L->vertex_stride) { c->error = "rt_ply_decode: position offset outside the vertex record"; return RT_ERR_INVALID; }
    const uint8_t *fb = static_cast<const uint8_t *>(face_bytes);
    auto count_at = [&](size_t pos) { unsigned v = fb[pos]; if (L->count_size >= 2) v |= (unsigned)fb[pos + 1] << 8; if (L->count_size == 4) v |= (unsigned)fb[pos + 2] << 16 | (unsigned)fb[pos + 3] << 24; return v; };
    const size_t fixed = (size_t)L->face_pre_bytes + L->count_size + L->face_post_bytes;
    unsigned k0 = 0;
    if (nf > 0) {
        if (face_bytes_size < fixed) { c->error = "rt_ply_decode: face data truncated"; return RT_ERR_INVALID; }
        k0 = count_at(L->face_pre_bytes);
    }
    const size_t vbytes = (size_t)nv * L->vertex_stride;
    HIP_TRY(c, c->d_ply_in.ensure(vbytes + face_bytes_size + 16));
    uint8_t *d_vb = c->d_ply_in.p, *d_fb = c->d_ply_in.p + vbytes;
    if (vbytes) HIP_TRY(c, hipMemcpyAsync(d_vb, vertex_bytes, vbytes, hipMemcpyHostToDevice, s));
    if (nf > 0) HIP_TRY(c, hipMemcpyAsync(d_fb, face_bytes, face_bytes_size, hipMemcpyHostToDevice, s));
    HIP_TRY(c, ev.start(ms_kernels != nullptr, s));
    HIP_TRY(c, rt_ply_vertices(s, d_vb, nv, L->vertex_stride, L->pos_offset, L->pos_type, c->d_ply_pos.p));
    // one face size for the whole file?  (the records are then at i * record size; the kernel checks every count)
    const size_t rec0 = fixed + (size_t)k0 * L->index_size;
    bool uniform = nf > 0 && rec0 <= face_bytes_size / (size_t)nf;
    if (uniform) {
        HIP_TRY(c, rt_ply_faces_uniform(s, d_fb, nf, L->face_pre_bytes, L->count_size, L->index_size, L->face_post_bytes, k0, c->d_ply_tri.p, d_flag));
        int flag = 0;
        HIP_TRY(c, hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (flag) uniform = false;
        else n_tris = k0 == 3 ? nf : (k0 == 4 ? 2 * nf : 0);
    }
    if (!uniform && nf > 0) {
        // mixed face sizes: the offsets are a dependent chain over the counts (host, one pass), the rest stays on the device
        std::vector<long long> offs((size_t)nf);
        size_t pos = 0;
        for (int64_t i = 0; i < nf; ++i) {
            if (pos + fixed > face_bytes_size) { c->error = "rt_ply_decode: face data truncated"; return RT_ERR_INVALID; }
            offs[(size_t)i] = (long long)pos;
            pos += fixed + (size_t)count_at(pos + L->face_pre_bytes) * L->index_size;
        }
        if (pos > face_bytes_size) { c->error = "rt_ply_decode: face data truncated"; return RT_ERR_INVALID; }
        size_t scan_bytes = 0;
        HIP_TRY(c, rt_ply_scan_bytes(nf, &scan_bytes));
        HIP_TRY(c, c->d_ply_tmp.ensure(scan_bytes + 16));
        HIP_TRY(c, c->d_ply_off.ensure((size_t)nf));
        HIP_TRY(c, hipMemcpyAsync(c->d_ply_off.p, offs.data(), (size_t)nf * sizeof(long long), hipMemcpyHostToDevice, s));
        int *d_ntri = c->d_ply_cnt.p, *d_toff = c->d_ply_cnt.p + (nf + 1);
        HIP_TRY(c, rt_ply_faces_mixed_count(s, d_fb, c->d_ply_off.p, nf, L->face_pre_bytes, L->count_size, d_ntri, d_toff, c->d_ply_tmp.p, scan_bytes));
        HIP_TRY(c, rt_ply_faces_mixed_write(s, d_fb, c->d_ply_off.p, d_toff, nf, L->face_pre_bytes, L->count_size, L->index_size, c->d_ply_tri.p));
        int total = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, d_toff + nf, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));      // `offs` and `total` are host memory of this call
        n_tris = total;
    }
    HIP_TRY(c, ev.stop(s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *out_n_tris = n_tris;
    if (ms_kernels) *ms_kernels = ev.ms();
    return RT_OK;
}

int rt_ply_decode(rt_ctx *c, const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, const rt_ply_layout *L, float *out_pos, int32_t *out_tris,
                  int64_t *out_n_tris, double *ms_kernels) {
    if (!c) return RT_ERR_INVALID;
    if (L && ((L->n_vertices > 0 && !out_pos) || (L->n_faces > 0 && !out_tris))) { c->error = "rt_ply_decode: bad argument"; return RT_ERR_INVALID; }
    int64_t n_tris = 0;
    const int rc = ply_decode_device(c, vertex_bytes, face_bytes, face_bytes_size, L, &n_tris, ms_kernels);
    if (rc != RT_OK) return rc;
    hipStream_t s = c->stream;
    if (L->n_vertices > 0) HIP_TRY(c, hipMemcpyAsync(out_pos, c->d_ply_pos.p, (size_t)L->n_vertices * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (n_tris > 0) HIP_TRY(c, hipMemcpyAsync(out_tris, c->d_ply_tri.p, (size_t)n_tris * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *out_n_tris = n_tris;
    return RT_OK;
}

int rt_ply_decode_to(rt_ctx *c, const void *vertex_bytes, const void *face_bytes, size_t face_bytes_size, const rt_ply_layout *L, float *d_out_pos, int32_t *d_out_tris,
                     int32_t index_offset, int64_t *out_n_tris, double *ms_kernels) {
    if (!c) return RT_ERR_INVALID;
    if (L && ((L->n_vertices > 0 && !d_out_pos) || (L->n_faces > 0 && !d_out_tris))) { c->error = "rt_ply_decode_to: bad argument"; return RT_ERR_INVALID; }
    int64_t n_tris = 0;
    const int rc = ply_decode_device(c, vertex_bytes, face_bytes, face_bytes_size, L, &n_tris, ms_kernels);
    if (rc != RT_OK) return rc;
    hipStream_t s = c->stream;
    if (L->n_vertices > 0) HIP_TRY(c, hipMemcpyAsync(d_out_pos, c->d_ply_pos.p, (size_t)L->n_vertices * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, rt_ply_offset_copy(s, c->d_ply_tri.p, d_out_tris, 3 * n_tris, index_offset));
    HIP_TRY(c, hipStreamSynchronize(s));
    *out_n_tris = n_tris;
    return RT_OK;
}

// ---- host-only BVH view for tests (no GPU needed) ------------------------------------------------------
struct rt_host_bvh { HostScene hs; std::string error; };

int rt_host_bvh_build(const rt_scene_desc *scene, rt_host_bvh **out) {
    if (!out) return RT_ERR_INVALID;
    rt_host_bvh *b = new rt_host_bvh();
    int rc = rt_build_host_scene(scene, b->hs, b->error);
    if (rc != RT_OK) { g_create_error = b->error; delete b; *out = nullptr; return rc; }
    *out = b;
    return RT_OK;
}
int rt_host_bvh_build_gpu(const rt_scene_desc *scene, int hip_device_id, rt_host_bvh **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_create_error = "no HIP device available; the GPU BVH builder has no CPU fallback"; return RT_ERR_NO_DEVICE; }
    if (hip_device_id < 0 || hip_device_id >= n || hipSetDevice(hip_device_id) != hipSuccess) { g_create_error = "hip_device_id out of range"; return RT_ERR_INVALID; }
    rt_host_bvh *b = new rt_host_bvh();
    int rc = rt_build_host_scene(scene, b->hs, b->error, /*gpu_build=*/2);
    if (rc != RT_OK) { g_create_error = b->error; delete b; return rc; }
    *out = b;
    return RT_OK;
}
int rt_host_bvh_info(const rt_host_bvh *b, rt_bvh_info *out) { if (!b || !out) return RT_ERR_INVALID; *out = b->hs.info; return RT_OK; }
int rt_host_bvh_leaves(const rt_host_bvh *b, int64_t *leaf_start_end, int32_t *prim_order, float *leaf_boxes, int64_t *dev_leaf_start_count, int32_t *dev_prim_shape) {
    if (!b) return RT_ERR_INVALID;
    const HostScene &h = b->hs;
    if (leaf_start_end) std::memcpy(leaf_start_end, h.ref_leaf_range.data(), h.ref_leaf_range.size() * sizeof(int64_t));
    if (prim_order) std::memcpy(prim_order, h.ref_prim_order.data(), h.ref_prim_order.size() * sizeof(int32_t));
    if (leaf_boxes) std::memcpy(leaf_boxes, h.leaf_boxes.data(), h.leaf_boxes.size() * sizeof(float));
    if (dev_leaf_start_count) for (size_t i = 0; i < h.leaves.size(); ++i) { dev_leaf_start_count[2 * i] = h.leaves[i].start; dev_leaf_start_count[2 * i + 1] = h.leaves[i].count; }
    if (dev_prim_shape) for (size_t i = 0; i < h.prims.size(); ++i) dev_prim_shape[i] = h.prims[i].shape;
    return RT_OK;
}
int64_t rt_host_bvh_cones(const rt_host_bvh *b, uint32_t *words, int64_t *start_end) {
    if (!b) return RT_ERR_INVALID;
    const HostScene &h = b->hs;
    if (words) std::memcpy(words, h.cone_words.data(), h.cone_words.size() * sizeof(uint32_t));
    if (start_end) std::memcpy(start_end, h.cone_ranges.data(), h.cone_ranges.size() * sizeof(int64_t));
    return (int64_t)h.cone_words.size();
}
int rt_cone_allows_prune(uint32_t word, const float *direction) {
    if (!direction) return 0;
    const f3 d = ld3(direction);
    return cone_allows_prune(word, d.x, d.y, d.z, norm3(d)) ? 1 : 0;
}
void rt_host_bvh_free(rt_host_bvh *b) { delete b; }

// sizeof table so that the Python mirror of the header can be checked without a GPU
int rt_abi_sizeof(int which) {
    switch (which) {
    case 0: return (int)sizeof(rt_material);
    case 1: return (int)sizeof(rt_light);
    case 2: return (int)sizeof(rt_texture);
    case 3: return (int)sizeof(rt_object);
    case 4: return (int)sizeof(rt_scene_desc);
    case 5: return (int)sizeof(rt_camera_desc);
    case 6: return (int)sizeof(rt_render_opts);
    case 7: return (int)sizeof(rt_stats);
    case 8: return (int)sizeof(rt_ray);
    case 9: return (int)sizeof(rt_hit);
    case 10: return (int)sizeof(rt_bvh_info);
    case 11: return (int)sizeof(rt_tonemap_opts);
    case 12: return (int)sizeof(rt_ply_layout);
    case 13: return (int)sizeof(rt_instance);
    case 14: return (int)sizeof(rt_prim_run);
    default: return -1;
    }
}

}  // extern "C"
