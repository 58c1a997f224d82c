// rt_api.hip — the C-ABI of librt_mi355.so (include/rt_mi355.h): context, scene upload, wavefront driver.
//
// Host orchestration of one frame (replaces DefaultRenderer::RenderCamera's 8 std::thread row bands,
// reference src/DefaultRenderer.cpp:59-104): the owned pixels (interleaved tiles) are cut into batches of whole
// pixels; per batch  raygen -> for level k: trace, shade_pre, trace_shadow, shade_post (children compacted into
// level k+1) -> backward sweep -> resolve.  Everything is enqueued on one HIP stream; the only host round trip
// per level is the 4-byte child count.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_math.h"
#include "rt_wavefront.h"

namespace {

std::string g_create_error;

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

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool has_scene = false;
    HostScene host;             // kept for rt_scene_bvh_info and as the owner of table sizes
    DevScene dev;               // device pointers
    // scene arenas
    DevArray<DevNode> d_nodes;
    DevArray<DevLeaf> d_leaves;
    DevArray<DevPrim> d_prims;
    DevArray<DevObject> d_objects;
    DevArray<DevVertex> d_verts;
    DevArray<int32_t> d_tri_vert;
    DevArray<rt_material> d_materials;
    DevArray<rt_light> d_lights;
    DevArray<rt_texture> d_textures;
    DevArray<uint8_t> d_texels;
    // wavefront state
    std::vector<LevelMem> levels;
    DevArray<float4> sh_o, sh_d, sh_ip;
    DevArray<int> sh_occ;
    DevArray<int> d_pixels;
    DevArray<int> d_overflow;
    DevArray<int> d_counts;                 // per-level child counters
    DevArray<unsigned long long> d_kind;    // rays per kind
    DevArray<DevCounters> d_counters;
    DevArray<float> d_hdr;
    DevArray<int> d_pid;
    DevArray<float> d_pt;
    DevArray<rt_ray> d_rays;
    DevArray<rt_hit> d_hits;
    int overflow_depth = 0;
};

static void release_all(rt_ctx *c) {
    c->d_nodes.release(); c->d_leaves.release(); c->d_prims.release(); c->d_objects.release(); c->d_verts.release(); c->d_tri_vert.release();
    c->d_materials.release(); c->d_lights.release(); c->d_textures.release(); c->d_texels.release();
    for (auto &l : c->levels) l.release();
    c->sh_o.release(); c->sh_d.release(); c->sh_ip.release(); c->sh_occ.release(); c->d_pixels.release(); c->d_overflow.release();
    c->d_counts.release(); c->d_kind.release(); c->d_counters.release(); c->d_hdr.release(); c->d_pid.release(); c->d_pt.release();
    c->d_rays.release(); c->d_hits.release();
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
    e = hipSetDevice(hip_device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    rt_ctx *c = new rt_ctx();
    c->device = hip_device_id;
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
    c->has_scene = false;
    int rc = rt_build_host_scene(scene, c->host, c->error);
    if (rc != RT_OK) return rc;
    HostScene &h = c->host;
    for (const DevPrim &p : h.prims)
        if (p.kind == DEV_PRIM_SPHERE && !(h.objects[p.object].flags & DEV_OBJ_HAS_XFORM)) { c->error = "sphere object without a transform"; return RT_ERR_INVALID; }
    UPLOAD(c, c->d_nodes, h.nodes);
    UPLOAD(c, c->d_leaves, h.leaves);
    UPLOAD(c, c->d_prims, h.prims);
    UPLOAD(c, c->d_objects, h.objects);
    UPLOAD(c, c->d_verts, h.verts);
    UPLOAD(c, c->d_tri_vert, h.tri_vert);
    UPLOAD(c, c->d_materials, h.materials);
    UPLOAD(c, c->d_lights, h.lights);
    UPLOAD(c, c->d_textures, h.textures);
    UPLOAD(c, c->d_texels, h.texels);
    DevScene &d = c->dev;
    d.p = h.params;
    d.nodes = c->d_nodes.p; d.leaves = c->d_leaves.p; d.prims = c->d_prims.p; d.objects = c->d_objects.p; d.verts = c->d_verts.p;
    d.tri_vert = c->d_tri_vert.p; d.materials = c->d_materials.p; d.lights = c->d_lights.p; d.textures = c->d_textures.p; d.texels = c->d_texels.p;
    // traversal stack overflow area: entries beyond the LDS part, one column per persistent thread
    c->overflow_depth = std::max(0, h.params.bvh_depth + 2 - rt_trace_lds_stack_depth());
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
    hipStream_t s = c->stream;
    const DevCamera dc = make_camera(cam, opts->seed);
    const int spp = dc.spp;
    const int max_depth = std::max(0, c->dev.p.max_depth);
    const int nl = c->dev.p.n_lights;

    // owned pixels (interleaved tiles)
    const int64_t n_owned = rt_owned_pixel_count(cam->nx, cam->ny, opts->tile_size, opts->tile_rank, opts->tile_world);
    std::vector<int32_t> pixels((size_t)n_owned);
    rt_tile_pixel_order(cam->nx, cam->ny, opts->tile_size, opts->tile_rank, opts->tile_world, pixels.data());
    HIP_TRY(c, c->d_pixels.ensure((size_t)std::max<int64_t>(1, n_owned)));
    if (n_owned) HIP_TRY(c, hipMemcpyAsync(c->d_pixels.p, pixels.data(), (size_t)n_owned * sizeof(int32_t), hipMemcpyHostToDevice, s));

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

    if ((int)c->levels.size() < max_depth + 2) c->levels.resize(max_depth + 2);
    HIP_TRY(c, c->d_counts.ensure(3 * ((size_t)max_depth + 2)));
    HIP_TRY(c, c->d_kind.ensure(RT_RAY_KINDS));
    HIP_TRY(c, hipMemsetAsync(c->d_kind.p, 0, RT_RAY_KINDS * sizeof(unsigned long long), s));
    DevCounters *d_counters = nullptr;
    if (stats && opts->timing >= 2) {       // timing >= 2: also count node visits / tests (slower kernels)
        HIP_TRY(c, c->d_counters.ensure(1));
        HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(DevCounters), s));
        d_counters = c->d_counters.p;
    }

    int64_t batch_samples = opts->max_batch_samples > 0 ? opts->max_batch_samples : (4 << 20);
    int64_t batch_pixels = std::max<int64_t>(1, batch_samples / spp);
    if (batch_pixels * spp > 0x3fffffff) batch_pixels = 0x3fffffff / spp;

    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;
    const bool timing = stats && opts->timing;
    if (timing) { HIP_TRY(c, hipEventCreate(&ev_first)); HIP_TRY(c, hipEventCreate(&ev_last)); HIP_TRY(c, hipEventRecord(ev_first, s)); }
    uint64_t primary_rays = 0, trace_launches = 0, trace_rays = 0;

    auto timed_trace = [&](auto launch) -> int {
        if (timing) {
            hipEvent_t a, b;
            HIP_TRY(c, hipEventCreate(&a)); HIP_TRY(c, hipEventCreate(&b));
            HIP_TRY(c, hipEventRecord(a, s));
            launch();
            HIP_TRY(c, hipEventRecord(b, s));
            trace_events.push_back({a, b});
        } else launch();
        return RT_OK;
    };

    for (int64_t p0 = 0; p0 < n_owned; p0 += batch_pixels) {
        const int np = (int)std::min<int64_t>(batch_pixels, n_owned - p0);
        const int n0 = np * spp;
        primary_rays += (uint64_t)n0;
        HIP_TRY(c, c->levels[0].ensure((size_t)n0));
        c->levels[0].count = n0;
        rt_launch_raygen(s, dc, c->d_pixels.p, (int)p0, n0, c->levels[0].view());
        HIP_TRY(c, hipMemsetAsync(c->d_counts.p, 0, 3 * ((size_t)max_depth + 2) * sizeof(int), s));
        int *fetch_closest = c->d_counts.p + (max_depth + 2), *fetch_shadow = c->d_counts.p + 2 * (max_depth + 2);
        int last = 0;
        for (int k = 0; k <= max_depth; ++k) {
            LevelMem &lm = c->levels[k];
            const int n = lm.count;
            if (n == 0) break;
            last = k;
            int rc = timed_trace([&] { rt_launch_trace(s, c->dev, lm.view(), n, fetch_closest + k, c->d_overflow.p, d_counters); });
            if (rc != RT_OK) return rc;
            ++trace_launches; trace_rays += (uint64_t)n;
            const size_t nsh = (size_t)n * (size_t)std::max(1, nl);
            if (nsh > 0x7fffffff) { c->error = "shadow queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
            HIP_TRY(c, c->sh_o.ensure(nsh)); HIP_TRY(c, c->sh_d.ensure(nsh)); HIP_TRY(c, c->sh_ip.ensure(nsh)); HIP_TRY(c, c->sh_occ.ensure(nsh));
            ShadowBuf sb;
            sb.o = c->sh_o.p; sb.d = c->sh_d.p; sb.ipref = c->sh_ip.p; sb.occluded = c->sh_occ.p; sb.capacity = (int)std::min<size_t>(c->sh_o.cap, 0x7fffffff);
            rt_launch_shade_pre(s, c->dev, dc, lm.view(), n, sb);
            if (nl > 0) {
                rc = timed_trace([&] { rt_launch_trace_shadow(s, c->dev, sb, n * nl, fetch_shadow + k, c->d_overflow.p, d_counters); });
                if (rc != RT_OK) return rc;
                ++trace_launches; trace_rays += (uint64_t)n * nl;
            }
            LevelMem &nx = c->levels[k + 1];
            nx.count = 0;
            if (k < max_depth) {
                if ((size_t)n * 2 > 0x3fffffff) { c->error = "ray queue too large; lower max_batch_samples"; return RT_ERR_TOO_LARGE; }
                HIP_TRY(c, nx.ensure((size_t)n * 2));
            }
            rt_launch_shade_post(s, c->dev, dc, lm.view(), n, sb, nx.view(), k, c->d_counts.p + k + 1, c->d_kind.p);
            if (k < max_depth) {
                int child_count = 0;
                HIP_TRY(c, hipMemcpyAsync(&child_count, c->d_counts.p + k + 1, sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipStreamSynchronize(s));
                nx.count = child_count;
            }
        }
        for (int k = last; k >= 0; --k)
            rt_launch_backward(s, c->levels[k].view(), c->levels[k].count, c->levels[k + 1].view(), (k < last) ? 1 : 0);
        rt_launch_resolve(s, dc, c->d_pixels.p, (int)p0, np, c->levels[0].view(), compact ? 1 : 0, d_hdr, d_pid, d_pt);
    }
    if (timing) HIP_TRY(c, hipEventRecord(ev_last, s));
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
        stats->rays[RT_RAY_SHADOW] = kinds[RT_RAY_SHADOW]; stats->rays[RT_RAY_REFLECT] = kinds[RT_RAY_REFLECT]; stats->rays[RT_RAY_REFRACT] = kinds[RT_RAY_REFRACT];
        stats->node_visits = hc.node_visits; stats->box_tests = hc.box_tests; stats->tri_tests = hc.tri_tests; stats->sphere_tests = hc.sphere_tests;
        stats->trace_launches = trace_launches; stats->trace_rays = trace_rays;
        if (timing) {
            float ms = 0;
            hipEventElapsedTime(&ms, ev_first, ev_last);
            stats->ms_kernels = ms;
            double tr = 0;
            for (auto &pr : trace_events) { float m = 0; hipEventElapsedTime(&m, pr.first, pr.second); tr += m; }
            stats->ms_trace = tr;
        }
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    if (ev_first) hipEventDestroy(ev_first);
    if (ev_last) hipEventDestroy(ev_last);
    for (auto &pr : trace_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    return RT_OK;
}

int rt_intersect_batch(rt_ctx *c, const rt_ray *rays, size_t n, rt_hit *hits, int on_device, rt_stats *stats) {
    if (!c) return RT_ERR_INVALID;
    if ((n && (!rays || !hits)) || n > 0x7fffffff) { c->error = "bad ray batch"; return RT_ERR_INVALID; }
    if (!c->has_scene) { c->error = "no scene uploaded"; return RT_ERR_NO_SCENE; }
    HIP_TRY(c, hipSetDevice(c->device));
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
    default: return -1;
    }
}

}  // extern "C"
