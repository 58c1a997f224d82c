// rt_gather.hip — librt_mi355_mgpu.so: the multi-GPU step of the render path (include/rt_mi355_mgpu.h).
//
// The reference renders one frame with eight threads over row bands and joins them (src/DefaultRenderer.cpp:65-75).  Here the
// "threads" are GPUs with interleaved tiles; the join is a gather of compact RGB32F tile buffers to rank 0 over RCCL (grouped
// ncclSend / ncclRecv: point to point, one xGMI link per peer) followed by one de-interleave kernel, and the scene every GPU needs
// is broadcast once from the rank that built it.  Uses only the C-ABI of librt_mi355.so for everything that concerns scenes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355_mgpu.h"

namespace {

thread_local std::string g_create_error;

struct Local {                      // one rank this process drives
    int rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    float *d_tiles = nullptr;       // rt_gather_buffers: a compact tile buffer owned by the library
    size_t tiles_floats = 0;
};

}  // namespace

struct rt_gather {
    int world = 1;
    std::vector<Local> local;       // all ranks (one process, N GPUs) or exactly one (one process per GPU)
    std::string error;
    // rank 0's state for the frame geometry it last gathered
    int key[3] = {0, 0, 0};         // nx, ny, tile size
    int *d_order = nullptr;         // pixel index of every entry of the concatenated tile buffers (rank 0's tiles first)
    float *d_stage = nullptr;       // tiles of ranks 1 .. world-1 as they arrive, in rank order
    size_t stage_floats = 0, order_ints = 0;
    std::vector<int64_t> count;     // owned pixels per rank
    float *d_frame = nullptr;       // rt_gather_buffers: rank 0's frame, owned by the library
    size_t frame_floats = 0;
    int frame_nx = 0, frame_ny = 0;
};

#define G_HIP(g, expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (g)->error = std::string(#expr) + ": " + hipGetErrorString(e_); return RT_ERR_HIP; } } while (0)
#define G_NCCL(g, expr)                                                                             \
    do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { (g)->error = std::string(#expr) + ": " + ncclGetErrorString(r_); return RT_ERR_HIP; } } while (0)

// Inside ncclGroupStart / ncclGroupEnd a failed call must not leave the group open: remember the first failure, keep going to the
// group's end, report afterwards.
#define G_NCCL_IN_GROUP(first, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess && (first) == ncclSuccess) (first) = r_; } while (0)

// out[order[i]] = in[i] for the concatenated tile buffers: entries [0, n0) come from rank 0's own buffer, the rest from the staging area
__global__ void k_deinterleave(const float *own, const float *stage, const int *order, long long n0, long long n, float *frame) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *src = i < n0 ? own + 3 * i : stage + 3 * (i - n0);
    float *dst = frame + 3 * (long long)order[i];
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

static int device_count_or_error(std::string &err) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        err = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); librt_mi355_mgpu has no CPU fallback";
        return 0;
    }
    return n;
}

extern "C" {

const char *rt_gather_last_error(const rt_gather *g) { return g ? g->error.c_str() : g_create_error.c_str(); }
int rt_gather_world(const rt_gather *g) { return g ? g->world : 0; }

int rt_gather_create(const int *hip_device_ids, int n_ranks, rt_gather **out) {
    if (!out) { g_create_error = "out is NULL"; return RT_ERR_INVALID; }
    *out = nullptr;
    if (!hip_device_ids || n_ranks < 1 || n_ranks > 64) { g_create_error = "rt_gather_create: bad device list"; return RT_ERR_INVALID; }
    const int ndev = device_count_or_error(g_create_error);
    if (!ndev) return RT_ERR_NO_DEVICE;
    for (int i = 0; i < n_ranks; ++i) {
        if (hip_device_ids[i] < 0 || hip_device_ids[i] >= ndev) { g_create_error = "rt_gather_create: device id out of range"; return RT_ERR_INVALID; }
        for (int j = 0; j < i; ++j) if (hip_device_ids[j] == hip_device_ids[i]) { g_create_error = "rt_gather_create: every rank needs a GPU of its own"; return RT_ERR_INVALID; }
    }
    rt_gather *g = new rt_gather();
    g->world = n_ranks;
    g->local.resize((size_t)n_ranks);
    std::vector<ncclComm_t> comms((size_t)n_ranks);
    const ncclResult_t r = ncclCommInitAll(comms.data(), n_ranks, hip_device_ids);
    if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); delete g; return RT_ERR_HIP; }
    for (int i = 0; i < n_ranks; ++i) {
        Local &l = g->local[(size_t)i];
        l.rank = i; l.device = hip_device_ids[i]; l.comm = comms[(size_t)i];
        hipError_t e = hipSetDevice(l.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        if (e != hipSuccess) { g_create_error = std::string("stream for a rank: ") + hipGetErrorString(e); rt_gather_destroy(g); return RT_ERR_HIP; }
    }
    *out = g;
    return RT_OK;
}

int rt_gather_unique_id(void *id_bytes) {
    if (!id_bytes) { g_create_error = "id_bytes is NULL"; return RT_ERR_INVALID; }
    static_assert(sizeof(ncclUniqueId) == RT_GATHER_ID_BYTES, "RT_GATHER_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return RT_ERR_HIP; }
    std::memcpy(id_bytes, &id, sizeof(id));
    return RT_OK;
}

int rt_gather_create_rank(const void *id_bytes, int rank, int world, int hip_device_id, rt_gather **out) {
    if (!out) { g_create_error = "out is NULL"; return RT_ERR_INVALID; }
    *out = nullptr;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) { g_create_error = "rt_gather_create_rank: bad rank / world"; return RT_ERR_INVALID; }
    const int ndev = device_count_or_error(g_create_error);
    if (!ndev) return RT_ERR_NO_DEVICE;
    if (hip_device_id < 0 || hip_device_id >= ndev) { g_create_error = "rt_gather_create_rank: device id out of range"; return RT_ERR_INVALID; }
    hipError_t e = hipSetDevice(hip_device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RT_ERR_HIP; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    rt_gather *g = new rt_gather();
    g->world = world;
    g->local.resize(1);
    Local &l = g->local[0];
    l.rank = rank; l.device = hip_device_id;
    const ncclResult_t r = ncclCommInitRank(&l.comm, world, id, rank);
    if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); delete g; return RT_ERR_HIP; }
    e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); rt_gather_destroy(g); return RT_ERR_HIP; }
    *out = g;
    return RT_OK;
}

void rt_gather_destroy(rt_gather *g) {
    if (!g) return;
    for (Local &l : g->local) {
        (void)hipSetDevice(l.device);
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
        if (l.comm) ncclCommDestroy(l.comm);
        if (l.d_tiles) (void)hipFree(l.d_tiles);
        if (l.rank == 0 && g->d_frame) (void)hipFree(g->d_frame);
        if (l.rank == 0) { if (g->d_order) (void)hipFree(g->d_order); if (g->d_stage) (void)hipFree(g->d_stage); }
    }
    delete g;
}

int rt_gather_deinterleave_host(int nx, int ny, int tile_size, int world, const float *const *tiles, float *frame_rgb) {
    if (nx <= 0 || ny <= 0 || world < 1 || !tiles || !frame_rgb) return RT_ERR_INVALID;
    std::vector<int32_t> order;
    for (int r = 0; r < world; ++r) {
        const int64_t n = rt_owned_pixel_count(nx, ny, tile_size, r, world);
        if (n < 0) return RT_ERR_INVALID;
        if (n && !tiles[r]) return RT_ERR_INVALID;
        order.resize((size_t)n);
        rt_tile_pixel_order(nx, ny, tile_size, r, world, order.data());
        for (int64_t i = 0; i < n; ++i) std::memcpy(frame_rgb + 3 * (size_t)order[(size_t)i], tiles[r] + 3 * (size_t)i, 3 * sizeof(float));
    }
    return RT_OK;
}

int rt_gather_buffers(rt_gather *g, int nx, int ny, int tile_size, float **d_tiles, float **d_frame) {
    if (!g) return RT_ERR_INVALID;
    if (nx <= 0 || ny <= 0 || !d_tiles) { g->error = "rt_gather_buffers: bad argument"; return RT_ERR_INVALID; }
    if (d_frame) *d_frame = nullptr;
    for (size_t i = 0; i < g->local.size(); ++i) {
        Local &l = g->local[i];
        const int64_t n = rt_owned_pixel_count(nx, ny, tile_size, l.rank, g->world);
        if (n < 0) { g->error = "rt_gather_buffers: bad tile geometry"; return RT_ERR_INVALID; }
        const size_t need = 3 * (size_t)std::max<int64_t>(n, 1);
        G_HIP(g, hipSetDevice(l.device));
        if (need > l.tiles_floats) {
            if (l.d_tiles) G_HIP(g, hipFree(l.d_tiles));
            l.d_tiles = nullptr; l.tiles_floats = 0;
            G_HIP(g, hipMalloc((void **)&l.d_tiles, need * sizeof(float)));
            l.tiles_floats = need;
        }
        d_tiles[i] = l.d_tiles;
        if (l.rank == 0) {
            const size_t fneed = 3 * (size_t)nx * (size_t)ny;
            if (fneed > g->frame_floats) {
                if (g->d_frame) G_HIP(g, hipFree(g->d_frame));
                g->d_frame = nullptr; g->frame_floats = 0;
                G_HIP(g, hipMalloc((void **)&g->d_frame, fneed * sizeof(float)));
                g->frame_floats = fneed;
            }
            g->frame_nx = nx; g->frame_ny = ny;
            if (d_frame) *d_frame = g->d_frame;
        }
    }
    return RT_OK;
}

int rt_gather_copy_frame(rt_gather *g, float *frame_rgb) {
    if (!g) return RT_ERR_INVALID;
    if (!frame_rgb) { g->error = "rt_gather_copy_frame: frame_rgb is NULL"; return RT_ERR_INVALID; }
    for (const Local &l : g->local)
        if (l.rank == 0) {
            if (!g->d_frame) { g->error = "rt_gather_copy_frame: no frame (rt_gather_buffers was not called)"; return RT_ERR_INVALID; }
            G_HIP(g, hipSetDevice(l.device));
            G_HIP(g, hipMemcpy(frame_rgb, g->d_frame, 3 * (size_t)g->frame_nx * (size_t)g->frame_ny * sizeof(float), hipMemcpyDeviceToHost));
            return RT_OK;
        }
    g->error = "rt_gather_copy_frame: rank 0 is not driven by this process";
    return RT_ERR_INVALID;
}

// rank 0: (re)build the pixel order of the concatenated tile buffers and the staging area for this frame geometry
static int prepare_root(rt_gather *g, const Local &root, int nx, int ny, int tile_size) {
    if (g->key[0] == nx && g->key[1] == ny && g->key[2] == tile_size && g->d_order) return RT_OK;
    g->count.assign((size_t)g->world, 0);
    int64_t total = 0;
    for (int r = 0; r < g->world; ++r) { g->count[(size_t)r] = rt_owned_pixel_count(nx, ny, tile_size, r, g->world); total += g->count[(size_t)r]; }
    if (total != (int64_t)nx * ny) { g->error = "rt_gather_tiles: tile partition does not cover the frame"; return RT_ERR_INVALID; }
    std::vector<int32_t> order((size_t)total);
    int64_t at = 0;
    for (int r = 0; r < g->world; ++r) { rt_tile_pixel_order(nx, ny, tile_size, r, g->world, order.data() + at); at += g->count[(size_t)r]; }
    G_HIP(g, hipSetDevice(root.device));
    if ((size_t)total > g->order_ints) {
        if (g->d_order) G_HIP(g, hipFree(g->d_order));
        g->d_order = nullptr; g->order_ints = 0;
        G_HIP(g, hipMalloc((void **)&g->d_order, (size_t)total * sizeof(int)));
        g->order_ints = (size_t)total;
    }
    const size_t stage = 3 * (size_t)(total - g->count[0]);
    if (stage > g->stage_floats) {
        if (g->d_stage) G_HIP(g, hipFree(g->d_stage));
        g->d_stage = nullptr; g->stage_floats = 0;
        G_HIP(g, hipMalloc((void **)&g->d_stage, std::max<size_t>(stage, 1) * sizeof(float)));
        g->stage_floats = stage;
    }
    G_HIP(g, hipMemcpyAsync(g->d_order, order.data(), (size_t)total * sizeof(int), hipMemcpyHostToDevice, root.stream));
    G_HIP(g, hipStreamSynchronize(root.stream));          // `order` is pageable host memory that dies here
    g->key[0] = nx; g->key[1] = ny; g->key[2] = tile_size;
    return RT_OK;
}

int rt_gather_tiles(rt_gather *g, int nx, int ny, int tile_size, const float *const *d_tiles, float *d_frame) {
    if (!g) return RT_ERR_INVALID;
    if (nx <= 0 || ny <= 0 || !d_tiles) { g->error = "rt_gather_tiles: bad argument"; return RT_ERR_INVALID; }
    const Local *root = nullptr;
    for (const Local &l : g->local) if (l.rank == 0) root = &l;
    if (root && !d_frame) { g->error = "rt_gather_tiles: rank 0 needs a frame buffer"; return RT_ERR_INVALID; }
    if (root) { const int rc = prepare_root(g, *root, nx, ny, tile_size); if (rc != RT_OK) return rc; }
    else {
        g->count.assign((size_t)g->world, 0);
        for (int r = 0; r < g->world; ++r) g->count[(size_t)r] = rt_owned_pixel_count(nx, ny, tile_size, r, g->world);
    }
    for (size_t i = 0; i < g->local.size(); ++i)
        if (g->count[(size_t)g->local[i].rank] > 0 && !d_tiles[i]) { g->error = "rt_gather_tiles: NULL tile buffer"; return RT_ERR_INVALID; }
    // every peer sends its compact buffer to rank 0; rank 0 posts the matching receives (one group: the transfers run side by side)
    ncclResult_t first = ncclSuccess;
    G_NCCL(g, ncclGroupStart());
    for (size_t i = 0; i < g->local.size(); ++i) {
        const Local &l = g->local[i];
        if (l.rank != 0 && g->count[(size_t)l.rank] > 0)
            G_NCCL_IN_GROUP(first, ncclSend(d_tiles[i], 3 * (size_t)g->count[(size_t)l.rank], ncclFloat, 0, l.comm, l.stream));
    }
    if (root) {
        size_t at = 0;
        for (int r = 1; r < g->world; ++r) {
            if (g->count[(size_t)r] > 0) G_NCCL_IN_GROUP(first, ncclRecv(g->d_stage + at, 3 * (size_t)g->count[(size_t)r], ncclFloat, r, root->comm, root->stream));
            at += 3 * (size_t)g->count[(size_t)r];
        }
    }
    G_NCCL(g, ncclGroupEnd());
    if (first != ncclSuccess) { g->error = std::string("rt_gather_tiles: ncclSend / ncclRecv: ") + ncclGetErrorString(first); return RT_ERR_HIP; }
    if (root) {
        size_t root_slot = 0;
        for (size_t i = 0; i < g->local.size(); ++i) if (g->local[i].rank == 0) root_slot = i;
        G_HIP(g, hipSetDevice(root->device));
        const long long n0 = g->count[0], n = (long long)nx * ny;
        hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, root->stream, d_tiles[root_slot], g->d_stage, g->d_order, n0, n, d_frame);
        G_HIP(g, hipGetLastError());
    }
    for (const Local &l : g->local) { G_HIP(g, hipSetDevice(l.device)); G_HIP(g, hipStreamSynchronize(l.stream)); }
    return RT_OK;
}

namespace {
struct DevBufs {                    // device temporaries of one broadcast: freed on every way out
    std::vector<int> device;
    std::vector<void *> ptr;
    int alloc(int dev, size_t bytes, void **out) {
        *out = nullptr;
        if (hipSetDevice(dev) != hipSuccess || hipMalloc(out, bytes) != hipSuccess) return RT_ERR_HIP;
        device.push_back(dev); ptr.push_back(*out);
        return RT_OK;
    }
    ~DevBufs() { for (size_t i = 0; i < ptr.size(); ++i) { (void)hipSetDevice(device[i]); (void)hipFree(ptr[i]); } }
};
}  // namespace

int rt_scene_broadcast(rt_gather *g, rt_ctx *const *ctxs) {
    if (!g) return RT_ERR_INVALID;
    if (!ctxs) { g->error = "rt_scene_broadcast: ctxs is NULL"; return RT_ERR_INVALID; }
    for (size_t i = 0; i < g->local.size(); ++i) if (!ctxs[i]) { g->error = "rt_scene_broadcast: NULL context"; return RT_ERR_INVALID; }
    const size_t hb = rt_scene_header_bytes();
    const size_t meta_bytes = hb + RT_SCENE_ARENAS * sizeof(uint64_t);
    std::vector<unsigned char> meta(meta_bytes, 0);
    for (size_t i = 0; i < g->local.size(); ++i)
        if (g->local[i].rank == 0) {
            uint64_t sizes[RT_SCENE_ARENAS];
            const int rc = rt_scene_export(ctxs[i], meta.data(), sizes);
            if (rc != RT_OK) { g->error = std::string("rt_scene_export: ") + rt_last_error(ctxs[i]); return rc; }
            std::memcpy(meta.data() + hb, sizes, sizeof(sizes));
        }
    // 1. header + arena sizes
    DevBufs tmp;
    std::vector<unsigned char *> d_meta(g->local.size(), nullptr);
    for (size_t i = 0; i < g->local.size(); ++i) {
        if (tmp.alloc(g->local[i].device, meta_bytes, (void **)&d_meta[i]) != RT_OK) { g->error = "rt_scene_broadcast: out of device memory"; return RT_ERR_HIP; }
        if (g->local[i].rank == 0) G_HIP(g, hipMemcpyAsync(d_meta[i], meta.data(), meta_bytes, hipMemcpyHostToDevice, g->local[i].stream));
    }
    ncclResult_t first = ncclSuccess;
    G_NCCL(g, ncclGroupStart());
    for (size_t i = 0; i < g->local.size(); ++i) G_NCCL_IN_GROUP(first, ncclBroadcast(d_meta[i], d_meta[i], meta_bytes, ncclUint8, 0, g->local[i].comm, g->local[i].stream));
    G_NCCL(g, ncclGroupEnd());
    if (first != ncclSuccess) { g->error = std::string("rt_scene_broadcast: ncclBroadcast: ") + ncclGetErrorString(first); return RT_ERR_HIP; }
    std::vector<std::vector<unsigned char> > metas(g->local.size(), std::vector<unsigned char>(meta_bytes));
    for (size_t i = 0; i < g->local.size(); ++i) {
        G_HIP(g, hipSetDevice(g->local[i].device));
        G_HIP(g, hipMemcpyAsync(metas[i].data(), d_meta[i], meta_bytes, hipMemcpyDeviceToHost, g->local[i].stream));
        G_HIP(g, hipStreamSynchronize(g->local[i].stream));
    }
    uint64_t sizes[RT_SCENE_ARENAS];
    std::memcpy(sizes, metas[0].data() + hb, sizeof(sizes));
    // 2. the arenas: rank 0 copies each out of its context into a buffer of its own GPU, every rank receives it in place
    std::vector<std::vector<void *> > bufs(g->local.size(), std::vector<void *>(RT_SCENE_ARENAS, nullptr));
    int rc_all = RT_OK;
    for (int a = 0; a < RT_SCENE_ARENAS && rc_all == RT_OK; ++a) {
        if (!sizes[a]) continue;
        for (size_t i = 0; i < g->local.size(); ++i) {
            if (tmp.alloc(g->local[i].device, (size_t)sizes[a], &bufs[i][(size_t)a]) != RT_OK) { g->error = "rt_scene_broadcast: out of device memory"; return RT_ERR_HIP; }
            if (g->local[i].rank == 0) {
                const int rc = rt_scene_export_arena(ctxs[i], a, bufs[i][(size_t)a]);
                if (rc != RT_OK) { g->error = std::string("rt_scene_export_arena: ") + rt_last_error(ctxs[i]); rc_all = rc; }
            }
        }
        if (rc_all != RT_OK) break;
        G_NCCL(g, ncclGroupStart());
        for (size_t i = 0; i < g->local.size(); ++i)
            G_NCCL_IN_GROUP(first, ncclBroadcast(bufs[i][(size_t)a], bufs[i][(size_t)a], (size_t)sizes[a], ncclUint8, 0, g->local[i].comm, g->local[i].stream));
        G_NCCL(g, ncclGroupEnd());
        if (first != ncclSuccess) { g->error = std::string("rt_scene_broadcast: ncclBroadcast: ") + ncclGetErrorString(first); return RT_ERR_HIP; }
    }
    for (size_t i = 0; i < g->local.size(); ++i) { G_HIP(g, hipSetDevice(g->local[i].device)); G_HIP(g, hipStreamSynchronize(g->local[i].stream)); }
    // 3. every other rank adopts the bytes
    for (size_t i = 0; i < g->local.size() && rc_all == RT_OK; ++i) {
        if (g->local[i].rank == 0) continue;
        const int rc = rt_scene_import(ctxs[i], metas[i].data(), bufs[i].data());
        if (rc != RT_OK) { g->error = std::string("rt_scene_import: ") + rt_last_error(ctxs[i]); rc_all = rc; }
    }
    return rc_all;
}

}  // extern "C"
