// rt_bvh_gpu.hip — the reference's BVH build (BVHTree.cpp:39-153: top-down, 4-bin SAH on the widest centroid axis,
// std::partition) executed level by level on the GPU, with the SAME topology, primitive order and boxes as the host
// builder in rt_build.hip.
//
// Why the result is identical and not merely equivalent:
//   * every per-node quantity is a min / max / count over the node's primitives (order independent) followed by the
//     node-local arithmetic of rt_bvh_split.h, which host and device share;
//   * std::partition (libstdc++, bidirectional) is a two-pointer sweep: elements already on the correct side stay, the
//     k-th misplaced element from the left is swapped with the k-th misplaced element from the right.  That pairing is
//     computed here with one prefix sum per level;
//   * a node is a contiguous range of the order array, so the left-to-right leaf order is the order of range starts.
// Only the sign of a zero box coordinate can differ (min/max of -0 and +0), which no later comparison can see.
//
// Per level: reduce boxes (wave-segmented by node) -> axis -> bin + reduce bins -> SAH -> flags -> scan -> pair -> swap ->
// reassign -> finish one-primitive children.  All kernels are HBM streaming over the order array; atomics are issued once
// per run of equal node ids inside a wave.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_bvh_split.h"
#include "rt_cone.h"

namespace {

#define GB_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); ok = false; goto done; } \
    } while (0)

// order-preserving map float <-> unsigned so that atomicMin / atomicMax work on floats
__host__ __device__ __forceinline__ unsigned f2o(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return u ^ (0x80000000u | (unsigned)((int)u >> 31));          // negative: all bits flipped; positive: sign bit set
}
__host__ __device__ __forceinline__ float o2f(unsigned u) {
    const unsigned top = (unsigned)((int)u >> 31);                   // all ones: was a positive float
    return __builtin_bit_cast(float, u ^ (0x80000000u | ~top));
}

struct Tree {                 // structure of arrays, one entry per reference node (BFS numbering)
    int *start, *end, *left, *depth;
    float *box;               // [6] lo, hi
};
struct Level {                // scratch of the nodes of the current level, index = node - base
    unsigned *all;            // [12]: lo3 (min), hi3 (max) of the boxes, then of the centroids
    int *binCount;            // [4]
    unsigned *binBox;         // [4][6]
    int *ax, *split, *mid;    // split: -2 leaf, -1 undecided, 0..3
    float *clo, *chi;
};

__global__ void k_level_init(Level lv, int cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const unsigned elo = f2o(1e9f), ehi = f2o(-1e9f);
    for (int k = 0; k < 3; ++k) { lv.all[12 * s + k] = elo; lv.all[12 * s + 3 + k] = ehi; lv.all[12 * s + 6 + k] = elo; lv.all[12 * s + 9 + k] = ehi; }
    for (int b = 0; b < 4; ++b) {
        lv.binCount[4 * s + b] = 0;
        for (int k = 0; k < 3; ++k) { lv.binBox[24 * s + 6 * b + k] = elo; lv.binBox[24 * s + 6 * b + 3 + k] = ehi; }
    }
    lv.split[s] = -1;
}

// wave64 reduction over runs of equal `key` (runs are contiguous: a node is a range of positions); the first lane of each
// run ends up with the run's min (MIN) or max
template <bool MIN>
__device__ __forceinline__ unsigned seg_reduce(unsigned v, int key) {
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_down(v, off);
        const int k2 = __shfl_down(key, off);
        if (lane + off < 64 && k2 == key) v = MIN ? (o < v ? o : v) : (o > v ? o : v);
    }
    return v;
}
__device__ __forceinline__ int seg_sum(int v, int key) {
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(v, off);
        const int k2 = __shfl_down(key, off);
        if (lane + off < 64 && k2 == key) v += o;
    }
    return v;
}

#define GB_ITEMS 8             // consecutive positions per thread in the two reduction kernels (fewer, fatter atomics)

__device__ __forceinline__ void atomic_box12(unsigned *a, const unsigned *v) {
#pragma unroll
    for (int k = 0; k < 12; ++k) { if ((k % 6) < 3) atomicMin(a + k, v[k]); else atomicMax(a + k, v[k]); }
}

// Box of the primitives and box of their centroids, per node.  A thread walks GB_ITEMS consecutive positions and keeps a
// running box while the node stays the same; completed runs go out as atomics at once, the thread's last run is first
// combined across the wave (runs of equal node are contiguous), so a wave inside one node issues a single set of atomics.
__global__ __launch_bounds__(256) void k_reduce_boxes(const float *boxes, const int *order, const int *node_of, int n, int base, Level lv) {
    const int p0 = (blockIdx.x * blockDim.x + threadIdx.x) * GB_ITEMS;
    const int lane = threadIdx.x & 63;
    int nd = -1;
    unsigned v[12];
    const unsigned elo = 0xffffffffu, ehi = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k] = elo; v[3 + k] = ehi; v[6 + k] = elo; v[9 + k] = ehi; }
    for (int j = 0; j < GB_ITEMS; ++j) {
        const int p = p0 + j;
        if (p >= n) break;
        const int ndj = node_of[p];
        if (ndj != nd) {
            if (nd >= 0) atomic_box12(lv.all + 12 * (size_t)(nd - base), v);
            nd = ndj;
#pragma unroll
            for (int k = 0; k < 3; ++k) { v[k] = elo; v[3 + k] = ehi; v[6 + k] = elo; v[9 + k] = ehi; }
        }
        if (ndj < 0) continue;
        const float *b = boxes + 6 * (size_t)order[p];
        Aabb bx; bx.lo = ld3(b); bx.hi = ld3(b + 3);
        const f3 c = centroid(bx);
        const unsigned w[12] = {f2o(bx.lo.x), f2o(bx.lo.y), f2o(bx.lo.z), f2o(bx.hi.x), f2o(bx.hi.y), f2o(bx.hi.z),
                                f2o(c.x), f2o(c.y), f2o(c.z), f2o(c.x), f2o(c.y), f2o(c.z)};
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = ((k % 6) < 3) ? (w[k] < v[k] ? w[k] : v[k]) : (w[k] > v[k] ? w[k] : v[k]);
    }
    const int prev = __shfl_up(nd, 1);
    const bool head = nd >= 0 && (lane == 0 || prev != nd);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const bool is_min = (k % 6) < 3;
        v[k] = is_min ? seg_reduce<true>(v[k], nd) : seg_reduce<false>(v[k], nd);
    }
    if (head) atomic_box12(lv.all + 12 * (size_t)(nd - base), v);
}

__device__ __forceinline__ Aabb load_obox(const unsigned *u) {
    Aabb b;
    b.lo = mk3(o2f(u[0]), o2f(u[1]), o2f(u[2])); b.hi = mk3(o2f(u[3]), o2f(u[4]), o2f(u[5]));
    return b;
}

__global__ void k_node_axis(Tree t, Level lv, int base, int cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const int nd = base + s;
    if (t.end[nd] - t.start[nd] <= 1) { lv.split[s] = -2; return; }           // already a finished leaf
    const Aabb all = load_obox(lv.all + 12 * (size_t)s), cen = load_obox(lv.all + 12 * (size_t)s + 6);
    int ax; float clo, chi;
    if (!bvh_split_axis(cen, ax, clo, chi)) {                                    // leaf: all centroids coincide
        lv.split[s] = -2; t.left[nd] = -1;
        float *b = t.box + 6 * (size_t)nd;
        b[0] = all.lo.x; b[1] = all.lo.y; b[2] = all.lo.z; b[3] = all.hi.x; b[4] = all.hi.y; b[5] = all.hi.z;
        return;
    }
    lv.ax[s] = ax; lv.clo[s] = clo; lv.chi[s] = chi;
}

// Bin of every primitive (stored for the partition) and count / box per (node, bin).  Same scheme: a thread whose
// GB_ITEMS positions lie in one node accumulates four local bins; a wave whose threads all lie in the same node combines
// them before the atomics.
__global__ __launch_bounds__(256) void k_bin(const float *boxes, const int *order, const int *node_of, unsigned char *bin_raw, int n, int base, Level lv) {
    const int p0 = (blockIdx.x * blockDim.x + threadIdx.x) * GB_ITEMS;
    const int lane = threadIdx.x & 63;
    const unsigned elo = 0xffffffffu, ehi = 0u;
    int cnt[RT_BVH_BINS] = {0, 0, 0, 0};
    unsigned bb[RT_BVH_BINS][6];
#pragma unroll
    for (int b = 0; b < RT_BVH_BINS; ++b) { bb[b][0] = bb[b][1] = bb[b][2] = elo; bb[b][3] = bb[b][4] = bb[b][5] = ehi; }
    int nd0 = -2;                   // node of the thread's first position (-1: finished leaf, -2: none seen); the local bins are its
    for (int j = 0; j < GB_ITEMS; ++j) {
        const int p = p0 + j;
        if (p >= n) break;
        int nd = node_of[p];
        if (nd >= 0 && lv.split[nd - base] == -2) nd = -1;
        if (nd0 == -2) nd0 = nd;
        if (nd < 0) continue;
        const int s = nd - base;
        const float *b = boxes + 6 * (size_t)order[p];
        Aabb bx; bx.lo = ld3(b); bx.hi = ld3(b + 3);
        const int raw = bvh_bin_raw(bx, lv.ax[s], lv.clo[s], lv.chi[s]);
        bin_raw[p] = (unsigned char)raw;
        const int bin = raw == RT_BVH_BINS ? RT_BVH_BINS - 1 : raw;
        const unsigned w[6] = {f2o(bx.lo.x), f2o(bx.lo.y), f2o(bx.lo.z), f2o(bx.hi.x), f2o(bx.hi.y), f2o(bx.hi.z)};
        if (nd == nd0) {
#pragma unroll
            for (int q = 0; q < RT_BVH_BINS; ++q)
                if (q == bin) {
                    ++cnt[q];
#pragma unroll
                    for (int k = 0; k < 6; ++k) bb[q][k] = k < 3 ? (w[k] < bb[q][k] ? w[k] : bb[q][k]) : (w[k] > bb[q][k] ? w[k] : bb[q][k]);
                }
        } else {                    // a second node inside this thread's positions: small nodes, little contention
            atomicAdd(lv.binCount + 4 * (size_t)s + bin, 1);
            unsigned *g = lv.binBox + 24 * (size_t)s + 6 * bin;
            atomicMin(g + 0, w[0]); atomicMin(g + 1, w[1]); atomicMin(g + 2, w[2]);
            atomicMax(g + 3, w[3]); atomicMax(g + 4, w[4]); atomicMax(g + 5, w[5]);
        }
    }
    // local bins belong to node nd0 (if >= 0)
    const int mine = nd0 >= 0 ? nd0 : -1;
    const int first = __shfl(mine, 0);
    if (__all(mine == first)) {
        if (first < 0) return;
        const int s = first - base;
#pragma unroll
        for (int b = 0; b < RT_BVH_BINS; ++b) {
            int c = cnt[b];
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
            if (c == 0) continue;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(bb[b][k], off); bb[b][k] = k < 3 ? (o < bb[b][k] ? o : bb[b][k]) : (o > bb[b][k] ? o : bb[b][k]); }
            if (lane == 0) {
                atomicAdd(lv.binCount + 4 * (size_t)s + b, c);
                unsigned *g = lv.binBox + 24 * (size_t)s + 6 * b;
#pragma unroll
                for (int k = 0; k < 6; ++k) { if (k < 3) atomicMin(g + k, bb[b][k]); else atomicMax(g + k, bb[b][k]); }
            }
        }
        return;
    }
    if (mine >= 0) {
        const int s = mine - base;
#pragma unroll
        for (int b = 0; b < RT_BVH_BINS; ++b) {
            if (cnt[b] == 0) continue;
            atomicAdd(lv.binCount + 4 * (size_t)s + b, cnt[b]);
            unsigned *g = lv.binBox + 24 * (size_t)s + 6 * b;
#pragma unroll
            for (int k = 0; k < 6; ++k) { if (k < 3) atomicMin(g + k, bb[b][k]); else atomicMax(g + k, bb[b][k]); }
        }
    }
}

__global__ void k_split(Tree t, Level lv, int base, int cnt, int *next_node) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt || lv.split[s] == -2) return;
    const int nd = base + s;
    const Aabb all = load_obox(lv.all + 12 * (size_t)s);
    int binCount[RT_BVH_BINS];
    Aabb binBox[RT_BVH_BINS];
    for (int b = 0; b < RT_BVH_BINS; ++b) { binCount[b] = lv.binCount[4 * (size_t)s + b]; binBox[b] = load_obox(lv.binBox + 24 * (size_t)s + 6 * b); }
    const int count = t.end[nd] - t.start[nd];
    int left_count = 0;
    const int split = bvh_choose_split(binCount, binBox, all, count, left_count);
    float *b = t.box + 6 * (size_t)nd;
    b[0] = all.lo.x; b[1] = all.lo.y; b[2] = all.lo.z; b[3] = all.hi.x; b[4] = all.hi.y; b[5] = all.hi.z;
    if (split < 0) { lv.split[s] = -2; t.left[nd] = -1; return; }
    const int c = atomicAdd(next_node, 2);
    const int m = t.start[nd] + left_count;
    lv.split[s] = split; lv.mid[s] = m;
    t.left[nd] = c;
    t.start[c] = t.start[nd]; t.end[c] = m; t.depth[c] = t.depth[nd] + 1; t.left[c] = -1;
    t.start[c + 1] = m; t.end[c + 1] = t.end[nd]; t.depth[c + 1] = t.depth[nd] + 1; t.left[c + 1] = -1;
}

__global__ __launch_bounds__(256) void k_flags(const int *node_of, const unsigned char *bin_raw, int *flag, int n, int base, Level lv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n) return;
    int f = 0;
    if (p < n) {
        const int nd = node_of[p];
        if (nd >= 0) { const int sp = lv.split[nd - base]; if (sp >= 0 && (int)bin_raw[p] <= sp) f = 1; }
    }
    flag[p] = f;                                                                 // flag[n] = 0: scan[n] is the total
}

// std::partition's pairing: the k-th "false" left of the partition point swaps with the k-th "true" counted from the right end
__global__ __launch_bounds__(256) void k_pairs(const int *node_of, const int *flag, const int *scan, int *pairL, int *pairR, int n, int base, Tree t, Level lv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int nd = node_of[p];
    if (nd < 0) return;
    const int s = nd - base;
    if (lv.split[s] < 0) return;
    const int start = t.start[nd], m = lv.mid[s];
    const int tb = scan[p] - scan[start];
    if (p < m) { if (!flag[p]) pairL[start + (p - start) - tb] = p; }
    else if (flag[p]) pairR[start + (m - start) - tb - 1] = p;
}
__global__ __launch_bounds__(256) void k_swap(const int *node_of, const int *scan, const int *pairL, const int *pairR, int *order, int n, int base, Tree t, Level lv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int nd = node_of[p];
    if (nd < 0) return;
    const int s = nd - base;
    if (lv.split[s] < 0) return;
    const int start = t.start[nd], m = lv.mid[s];
    const int npairs = (m - start) - (scan[m] - scan[start]);
    const int k = p - start;
    if (k < npairs) {
        const int a = pairL[start + k], b = pairR[start + k];
        const int oa = order[a], ob = order[b];
        order[a] = ob; order[b] = oa;
    }
}
__global__ __launch_bounds__(256) void k_assign(int *node_of, int n, int base, Tree t, Level lv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int nd = node_of[p];
    if (nd < 0) return;
    const int s = nd - base;
    if (lv.split[s] < 0) { node_of[p] = -1; return; }
    node_of[p] = p < lv.mid[s] ? t.left[nd] : t.left[nd] + 1;
}
// children of one primitive are leaves at once (BVHTree.cpp:46-51): box = the primitive's box
__global__ void k_children(Tree t, const float *boxes, const int *order, int *node_of, int base, int cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const int nd = base + s;
    if (t.end[nd] - t.start[nd] > 1) return;
    const int p = t.start[nd];
    const float *b = boxes + 6 * (size_t)order[p];
    float *o = t.box + 6 * (size_t)nd;
    for (int k = 0; k < 6; ++k) o[k] = b[k];
    node_of[p] = -1;
}
__global__ void k_fill(int *a, int n, int v, int iota) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) a[p] = iota ? p : v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Flattening on the device: leaves in left-to-right order (= order of range starts), primitive records with repeated shapes
// inside a leaf removed, packed leaf references, one DevChain per reference node, 4-wide collapse level by level.
// Mirrors the host flattener in rt_build.hip; the wide nodes are numbered breadth-first here (preorder there), which no
// result depends on.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_leaves(Tree t, int n_nodes, int *flag) {
    const int nd = blockIdx.x * blockDim.x + threadIdx.x;
    if (nd < n_nodes && t.left[nd] < 0) flag[t.start[nd]] = 1;
}
// per leaf: number of distinct shapes (a second reference to a shape inside one leaf can never win: same t, and the leaf rule needs t < best - 0.001)
__global__ __launch_bounds__(256) void k_leaf_unique(Tree t, int n_nodes, const int *order, const int *pshape, const int *leaf_at, int *uniq) {
    const int nd = blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= n_nodes || t.left[nd] >= 0) return;
    const int s = t.start[nd], e = t.end[nd];
    int c = 0;
    for (int i = s; i < e; ++i) {
        const int sh = pshape[order[i]];
        bool dup = false;
        for (int j = s; j < i && !dup; ++j) dup = pshape[order[j]] == sh;
        c += dup ? 0 : 1;
    }
    uniq[leaf_at[s]] = c;
}
__global__ __launch_bounds__(256) void k_leaf_write(Tree t, int n_nodes, const int *order, const int *pshape, const int *leaf_at, const int *dev_start,
                                                   const float4 *shape_rec, float4 *prims, DevLeaf *leaves, int *leaf_ref, int *status) {
    const int nd = blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= n_nodes || t.left[nd] >= 0) return;
    const int s = t.start[nd], e = t.end[nd];
    const int li = leaf_at[s], ds = dev_start[li];
    int c = 0;
    for (int i = s; i < e; ++i) {
        const int sh = pshape[order[i]];
        bool dup = false;
        for (int j = s; j < i && !dup; ++j) dup = pshape[order[j]] == sh;
        if (dup) continue;
        const float4 *src = shape_rec + 4 * (size_t)sh;
        float4 *dst = prims + 4 * (size_t)(ds + c);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        ++c;
    }
    leaves[li].start = ds; leaves[li].count = c;
    if (c >= 1 && c <= DEV_LEAF_MAX_PACKED && ds < (1 << DEV_LEAF_START_BITS)) leaf_ref[nd] = ~(int)(((unsigned)c << DEV_LEAF_START_BITS) | (unsigned)ds);
    else { if (li >= (1 << DEV_LEAF_START_BITS)) *status = 1; leaf_ref[nd] = ~li; }
}
// Normal cone (rt_cone.h) of every reference node of one tree level, bottom-up: a leaf merges the cones of its triangles,
// an internal node the cones of its two children (one level deeper, already done).  cone = 5 floats per node: axis, half-angle, s.
__device__ __forceinline__ NCone load_cone(const float *c) { NCone r; r.a = ld3(c); r.alpha = c[3]; r.smin = c[4]; return r; }
__global__ __launch_bounds__(256) void k_cone_level(Tree t, int base, int cnt, const int *order, const int *pshape, const DevPrim *shape_rec,
                                                   const DevObject *objects, float *cone) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const int nd = base + s;
    NCone c;
    const int l = t.left[nd];
    if (l >= 0) c = cone_merge(load_cone(cone + 5 * (size_t)l), load_cone(cone + 5 * (size_t)(l + 1)));
    else {
        c = cone_empty();
        for (int i = t.start[nd]; i < t.end[nd]; ++i) {
            const DevPrim &rec = shape_rec[pshape[order[i]]];
            if (rec.kind != DEV_PRIM_TRIANGLE) continue;
            const DevObject &ob = objects[rec.object];
            c = cone_merge(c, cone_of_triangle(ld3(rec.e1), ld3(rec.e2), (ob.flags & DEV_OBJ_HAS_XFORM) ? ob.inv : nullptr, ob.cone_k));
        }
    }
    float *o = cone + 5 * (size_t)nd;
    o[0] = c.a.x; o[1] = c.a.y; o[2] = c.a.z; o[3] = c.alpha; o[4] = c.smin;
}
__global__ __launch_bounds__(256) void k_chain(Tree t, int n_nodes, DevChain *chain) {
    const int nd = blockIdx.x * blockDim.x + threadIdx.x;
    if (nd >= n_nodes) return;
    const float *b = t.box + 6 * (size_t)nd;
    DevChain &c = chain[nd];
    c.lo[0] = b[0]; c.lo[1] = b[1]; c.lo[2] = b[2]; c.hi[0] = b[3]; c.hi[1] = b[4]; c.hi[2] = b[5];
    c._pad = 0;
    if (nd == 0) c.parent = -1;
    const int l = t.left[nd];
    if (l >= 0) { chain[l].parent = nd; chain[l + 1].parent = nd; }      // distinct fields per thread: no race
}
// one level of the 4-wide collapse: every work item is (reference node, wide node index)
__global__ __launch_bounds__(256) void k_collapse(Tree t, const int *leaf_ref, const float *cone, const int2 *work, int n_work, int2 *next_work, int *next_count,
                                                int *wide_count, DevNode4 *wide, int *node_chain) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_work) return;
    const int bnode = work[w].x, windex = work[w].y;
    int kids[4];
    int nk = 2;
    kids[0] = t.left[bnode]; kids[1] = kids[0] + 1;
    while (nk < 4) {
        int pick = -1;
        float area = -1.0f;
        for (int i = 0; i < nk; ++i)
            if (t.left[kids[i]] >= 0) {
                const float *b = t.box + 6 * (size_t)kids[i];
                Aabb bx; bx.lo = ld3(b); bx.hi = ld3(b + 3);
                const float a = surface_area(bx);
                if (a > area) { area = a; pick = i; }
            }
        if (pick < 0) break;
        const int open = kids[pick];
        for (int i = nk; i > pick + 1; --i) kids[i] = kids[i - 1];           // keep left-to-right order
        kids[pick] = t.left[open]; kids[pick + 1] = t.left[open] + 1;
        ++nk;
    }
    int n_int = 0;
    for (int i = 0; i < nk; ++i) n_int += t.left[kids[i]] >= 0 ? 1 : 0;
    int wbase = 0, qbase = 0;
    if (n_int) { wbase = atomicAdd(wide_count, n_int); qbase = atomicAdd(next_count, n_int); }
    DevNode4 &o = wide[windex];
    int k_int = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < nk) {
            const float *b = t.box + 6 * (size_t)kids[i];
            o.minx[i] = b[0]; o.miny[i] = b[1]; o.minz[i] = b[2]; o.maxx[i] = b[3]; o.maxy[i] = b[4]; o.maxz[i] = b[5];
            node_chain[4 * (size_t)windex + i] = kids[i];
            o.cone[i] = cone_pack(load_cone(cone + 5 * (size_t)kids[i]));      // may this child be pruned by distance? (rt_cone.h)
            if (t.left[kids[i]] >= 0) {
                o.ref[i] = wbase + k_int;
                next_work[qbase + k_int] = make_int2(kids[i], wbase + k_int);
                ++k_int;
            } else o.ref[i] = leaf_ref[kids[i]];
        } else {
            o.minx[i] = o.miny[i] = o.minz[i] = 0.0f; o.maxx[i] = o.maxy[i] = o.maxz[i] = 0.0f;
            o.ref[i] = DEV_REF_EMPTY; o.cone[i] = RT_CONE_WORD_NEVER; node_chain[4 * (size_t)windex + i] = -1;
        }
    }
}

template <typename T>
hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, (n ? n : 1) * sizeof(T)); }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Device state of one build.
struct GpuBuild {
    int n = 0, n_nodes = 1, max_depth = 0;
    float *d_box = nullptr; int *d_order = nullptr, *d_node_of = nullptr, *d_flag = nullptr, *d_scan = nullptr, *d_pairL = nullptr, *d_pairR = nullptr;
    unsigned char *d_bin = nullptr; int *d_next = nullptr; void *d_tmp = nullptr; size_t tmp_bytes = 0;
    Tree t = {nullptr, nullptr, nullptr, nullptr, nullptr};
    Level lv = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    bool tm = false, own_box = true;
    double t_mark = 0;
    std::vector<std::pair<int, int>> levels;      // (first node, node count) of every tree level, root level first
    void lap(const char *what) { if (tm) { const double tn = now_ms(); fprintf(stderr, "[rt gpu build] %-24s %8.2f ms\n", what, tn - t_mark); t_mark = tn; } }

    bool build(const float *pbox, int n_, std::string &err, bool box_on_device = false) {
        n = n_;
        own_box = !box_on_device;
        bool ok = true;
        tm = getenv("RT_BUILD_TIMING") != nullptr;
        t_mark = now_ms();
        const size_t cap = 2 * (size_t)n + 2;
        int base = 0, cnt = 1;
        const dim3 blk(256), grid_n((n + 1 + 255) / 256), grid_r((n + 256 * GB_ITEMS - 1) / (256 * GB_ITEMS));
        {
            if (own_box) GB_TRY(dalloc(&d_box, 6 * (size_t)n)); else d_box = const_cast<float *>(pbox);
            GB_TRY(dalloc(&d_order, (size_t)n)); GB_TRY(dalloc(&d_node_of, (size_t)n));
            GB_TRY(dalloc(&d_flag, (size_t)n + 1)); GB_TRY(dalloc(&d_scan, (size_t)n + 1)); GB_TRY(dalloc(&d_pairL, (size_t)n)); GB_TRY(dalloc(&d_pairR, (size_t)n));
            GB_TRY(dalloc(&d_bin, (size_t)n)); GB_TRY(dalloc(&d_next, 4));
            GB_TRY(dalloc(&t.start, cap)); GB_TRY(dalloc(&t.end, cap)); GB_TRY(dalloc(&t.left, cap)); GB_TRY(dalloc(&t.depth, cap)); GB_TRY(dalloc(&t.box, 6 * cap));
            GB_TRY(dalloc(&lv.all, 12 * (size_t)n)); GB_TRY(dalloc(&lv.binCount, 4 * (size_t)n)); GB_TRY(dalloc(&lv.binBox, 24 * (size_t)n));
            GB_TRY(dalloc(&lv.ax, (size_t)n)); GB_TRY(dalloc(&lv.split, (size_t)n)); GB_TRY(dalloc(&lv.mid, (size_t)n)); GB_TRY(dalloc(&lv.clo, (size_t)n)); GB_TRY(dalloc(&lv.chi, (size_t)n));
            GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag, d_scan, n + 1));
            GB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
            GB_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            lap("allocate");
            if (own_box) GB_TRY(hipMemcpyAsync(d_box, pbox, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_fill, grid_n, blk, 0, st, d_order, n, 0, 1);
            hipLaunchKernelGGL(k_fill, grid_n, blk, 0, st, d_node_of, n, 0, 0);
            {   // root
                const int root[4] = {0, n, -1, 1};
                GB_TRY(hipMemcpyAsync(t.start, &root[0], sizeof(int), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(t.end, &root[1], sizeof(int), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(t.left, &root[2], sizeof(int), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(t.depth, &root[3], sizeof(int), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(d_next, &n_nodes, sizeof(int), hipMemcpyHostToDevice, st));
                GB_TRY(hipStreamSynchronize(st));
            }
            lap("upload boxes");
            hipLaunchKernelGGL(k_children, dim3(1), dim3(64), 0, st, t, d_box, d_order, d_node_of, 0, 1);      // n == 1: the root is a leaf
            while (cnt > 0) {
                ++max_depth;
                levels.push_back({base, cnt});
                const dim3 grid_c((cnt + 255) / 256);
                hipLaunchKernelGGL(k_level_init, grid_c, blk, 0, st, lv, cnt);
                hipLaunchKernelGGL(k_reduce_boxes, grid_r, blk, 0, st, d_box, d_order, d_node_of, n, base, lv);
                hipLaunchKernelGGL(k_node_axis, grid_c, blk, 0, st, t, lv, base, cnt);
                hipLaunchKernelGGL(k_bin, grid_r, blk, 0, st, d_box, d_order, d_node_of, d_bin, n, base, lv);
                hipLaunchKernelGGL(k_split, grid_c, blk, 0, st, t, lv, base, cnt, d_next);
                hipLaunchKernelGGL(k_flags, grid_n, blk, 0, st, d_node_of, d_bin, d_flag, n, base, lv);
                GB_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flag, d_scan, n + 1, st));
                hipLaunchKernelGGL(k_pairs, grid_n, blk, 0, st, d_node_of, d_flag, d_scan, d_pairL, d_pairR, n, base, t, lv);
                hipLaunchKernelGGL(k_swap, grid_n, blk, 0, st, d_node_of, d_scan, d_pairL, d_pairR, d_order, n, base, t, lv);
                hipLaunchKernelGGL(k_assign, grid_n, blk, 0, st, d_node_of, n, base, t, lv);
                int next = 0;
                GB_TRY(hipMemcpyAsync(&next, d_next, sizeof(int), hipMemcpyDeviceToHost, st));
                GB_TRY(hipStreamSynchronize(st));
                base += cnt; cnt = next - n_nodes; n_nodes = next;
                if (cnt > 0) hipLaunchKernelGGL(k_children, dim3((cnt + 255) / 256), blk, 0, st, t, d_box, d_order, d_node_of, base, cnt);
            }
            GB_TRY(hipGetLastError());
            // the depth of a leaf created on the last level is max_depth as counted (root = 1)
            lap("levels");
        }
    done:
        return ok;
    }
    void release() {
        if (own_box) hipFree(d_box);
        hipFree(d_order); hipFree(d_node_of); hipFree(d_flag); hipFree(d_scan); hipFree(d_pairL); hipFree(d_pairR); hipFree(d_bin);
        hipFree(d_next); hipFree(d_tmp); hipFree(t.start); hipFree(t.end); hipFree(t.left); hipFree(t.depth); hipFree(t.box);
        hipFree(lv.all); hipFree(lv.binCount); hipFree(lv.binBox); hipFree(lv.ax); hipFree(lv.split); hipFree(lv.mid); hipFree(lv.clo); hipFree(lv.chi);
        if (st) hipStreamDestroy(st);
        lap("free");
    }
};

}  // namespace

// Host entry.  pbox: n boxes (lo.xyz, hi.xyz).  On success `order` holds the primitive permutation and `nodes` the tree in
// breadth-first numbering (root = 0, children of a node adjacent).
bool rt_bvh_gpu_build(const float *pbox, int64_t n64, std::vector<int32_t> &order, std::vector<RtBuildNode> &nodes, int &max_depth, std::string &err) {
    order.clear(); nodes.clear(); max_depth = 0;
    if (n64 <= 0) return true;
    GpuBuild g;
    bool ok = g.build(pbox, (int)n64, err);
    if (ok) {
        const int n = g.n, n_nodes = g.n_nodes;
        order.resize((size_t)n);
        std::vector<int> hs((size_t)n_nodes), he((size_t)n_nodes), hl((size_t)n_nodes), hd((size_t)n_nodes);
        std::vector<float> hb(6 * (size_t)n_nodes);
        {
            GB_TRY(hipMemcpyAsync(order.data(), g.d_order, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipMemcpyAsync(hs.data(), g.t.start, (size_t)n_nodes * sizeof(int), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipMemcpyAsync(he.data(), g.t.end, (size_t)n_nodes * sizeof(int), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipMemcpyAsync(hl.data(), g.t.left, (size_t)n_nodes * sizeof(int), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipMemcpyAsync(hd.data(), g.t.depth, (size_t)n_nodes * sizeof(int), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipMemcpyAsync(hb.data(), g.t.box, 6 * (size_t)n_nodes * sizeof(float), hipMemcpyDeviceToHost, g.st));
            GB_TRY(hipStreamSynchronize(g.st));
        }
        g.lap("download");
        nodes.resize((size_t)n_nodes);
        for (int i = 0; i < n_nodes; ++i) {
            RtBuildNode &b = nodes[(size_t)i];
            for (int k = 0; k < 3; ++k) { b.lo[k] = hb[6 * (size_t)i + k]; b.hi[k] = hb[6 * (size_t)i + 3 + k]; }
            b.left = hl[i]; b.right = hl[i] >= 0 ? hl[i] + 1 : -1; b.start = hs[i]; b.end = he[i]; b.depth = hd[i];
            if (hd[i] > max_depth) max_depth = hd[i];
        }
        g.lap("convert");
    }
done:
    g.release();
    return ok;
}

// Build AND flatten on the device: the scene arrays the traversal reads never exist on the host.  shape_rec: one 64-byte
// DevPrim per shape (host), pshape: shape of every reference primitive (host).  On success the caller owns out.* (hipFree).
bool rt_bvh_gpu_build_flatten(const float *pbox, int64_t n64, const DevPrim *shape_rec, int64_t n_shapes, const int32_t *pshape, const std::vector<DevObject> &objects,
                              RtGpuScene &out, std::string &err, bool inputs_on_device) {
    out = RtGpuScene();
    if (n64 <= 0) { err = "empty scene"; return false; }
    GpuBuild g;
    bool ok = g.build(pbox, (int)n64, err, inputs_on_device);
    float4 *d_shape = nullptr; int *d_pshape = nullptr, *d_uniq = nullptr, *d_dev_start = nullptr, *d_leaf_ref = nullptr, *d_cnt = nullptr;
    int2 *d_work[2] = {nullptr, nullptr};
    float *d_cone = nullptr;
    DevObject *d_objects = nullptr;
    if (ok) {
        const int n = g.n, n_nodes = g.n_nodes;
        hipStream_t st = g.st;
        const dim3 blk(256), grid_n((n + 1 + 255) / 256), grid_nodes((n_nodes + 255) / 256);
        int n_leaves = 0, n_leaf_prims = 0;
        {
            if (inputs_on_device) {
                d_shape = reinterpret_cast<float4 *>(const_cast<DevPrim *>(shape_rec)); d_pshape = const_cast<int *>(pshape);
            } else {
                GB_TRY(dalloc(&d_shape, 4 * (size_t)n_shapes)); GB_TRY(dalloc(&d_pshape, (size_t)n));
                GB_TRY(hipMemcpyAsync(d_shape, shape_rec, (size_t)n_shapes * sizeof(DevPrim), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(d_pshape, pshape, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
            }
            // leaves in order of their range start
            GB_TRY(hipMemsetAsync(g.d_flag, 0, ((size_t)n + 1) * sizeof(int), st));
            hipLaunchKernelGGL(k_mark_leaves, grid_nodes, blk, 0, st, g.t, n_nodes, g.d_flag);
            GB_TRY(hipcub::DeviceScan::ExclusiveSum(g.d_tmp, g.tmp_bytes, g.d_flag, g.d_scan, n + 1, st));      // d_scan[start] = leaf index
            GB_TRY(hipMemcpyAsync(&n_leaves, g.d_scan + n, sizeof(int), hipMemcpyDeviceToHost, st));
            GB_TRY(hipStreamSynchronize(st));
            GB_TRY(dalloc(&d_uniq, (size_t)n_leaves + 1)); GB_TRY(dalloc(&d_dev_start, (size_t)n_leaves + 1)); GB_TRY(dalloc(&d_leaf_ref, (size_t)n_nodes));
            GB_TRY(dalloc(&d_cnt, 4));
            GB_TRY(hipMemsetAsync(d_uniq, 0, ((size_t)n_leaves + 1) * sizeof(int), st));
            GB_TRY(hipMemsetAsync(d_cnt, 0, 4 * sizeof(int), st));
            hipLaunchKernelGGL(k_leaf_unique, grid_nodes, blk, 0, st, g.t, n_nodes, g.d_order, d_pshape, g.d_scan, d_uniq);
            {   // the scan temp storage was sized for n + 1 >= n_leaves + 1 items
                size_t need = 0;
                GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_uniq, d_dev_start, n_leaves + 1));
                if (need > g.tmp_bytes) { hipFree(g.d_tmp); g.d_tmp = nullptr; GB_TRY(hipMalloc(&g.d_tmp, need)); g.tmp_bytes = need; }
            }
            GB_TRY(hipcub::DeviceScan::ExclusiveSum(g.d_tmp, g.tmp_bytes, d_uniq, d_dev_start, n_leaves + 1, st));
            GB_TRY(hipMemcpyAsync(&n_leaf_prims, d_dev_start + n_leaves, sizeof(int), hipMemcpyDeviceToHost, st));
            GB_TRY(hipStreamSynchronize(st));
            GB_TRY(dalloc(&out.prims, (size_t)n_leaf_prims)); GB_TRY(dalloc(&out.leaves, (size_t)n_leaves)); GB_TRY(dalloc(&out.chain, (size_t)n_nodes));
            hipLaunchKernelGGL(k_leaf_write, grid_nodes, blk, 0, st, g.t, n_nodes, g.d_order, d_pshape, g.d_scan, d_dev_start, d_shape,
                               reinterpret_cast<float4 *>(out.prims), out.leaves, d_leaf_ref, d_cnt + 3);
            hipLaunchKernelGGL(k_chain, grid_nodes, blk, 0, st, g.t, n_nodes, out.chain);
            // normal cones, deepest level first
            GB_TRY(dalloc(&d_cone, 5 * (size_t)n_nodes)); GB_TRY(dalloc(&d_objects, objects.size()));
            if (!objects.empty()) GB_TRY(hipMemcpyAsync(d_objects, objects.data(), objects.size() * sizeof(DevObject), hipMemcpyHostToDevice, st));
            for (size_t li = g.levels.size(); li-- > 0;) {
                const int lbase = g.levels[li].first, lcnt = g.levels[li].second;
                hipLaunchKernelGGL(k_cone_level, dim3((lcnt + 255) / 256), blk, 0, st, g.t, lbase, lcnt, g.d_order, d_pshape, reinterpret_cast<const DevPrim *>(d_shape),
                                   d_objects, d_cone);
            }
            g.lap("leaves, records, chain, cones");
            // wide nodes
            const int n_internal = (n_nodes - 1) / 2;                 // a full binary tree: every internal node has two children
            int n_wide = 0, wide_depth = 0;
            if (n_internal > 0) {
                GB_TRY(dalloc(&out.nodes, (size_t)n_internal));        // at most one wide node per internal reference node
                GB_TRY(dalloc(&out.node_chain, 4 * (size_t)n_internal));
                GB_TRY(dalloc(&d_work[0], (size_t)n_internal)); GB_TRY(dalloc(&d_work[1], (size_t)n_internal));
                const int2 first = make_int2(0, 0);
                const int one = 1;
                GB_TRY(hipMemcpyAsync(d_work[0], &first, sizeof(first), hipMemcpyHostToDevice, st));
                GB_TRY(hipMemcpyAsync(d_cnt + 0, &one, sizeof(int), hipMemcpyHostToDevice, st));      // wide node count
                int n_work = 1, cur = 0;
                while (n_work > 0) {
                    ++wide_depth;
                    GB_TRY(hipMemsetAsync(d_cnt + 1, 0, sizeof(int), st));
                    hipLaunchKernelGGL(k_collapse, dim3((n_work + 255) / 256), blk, 0, st, g.t, d_leaf_ref, d_cone, d_work[cur], n_work, d_work[cur ^ 1], d_cnt + 1, d_cnt + 0, out.nodes, out.node_chain);
                    GB_TRY(hipMemcpyAsync(&n_work, d_cnt + 1, sizeof(int), hipMemcpyDeviceToHost, st));
                    GB_TRY(hipStreamSynchronize(st));
                    cur ^= 1;
                }
                GB_TRY(hipMemcpyAsync(&n_wide, d_cnt + 0, sizeof(int), hipMemcpyDeviceToHost, st));
            }
            int status = 0, root_ref = 0;
            float root_box[6];
            GB_TRY(hipMemcpyAsync(&status, d_cnt + 3, sizeof(int), hipMemcpyDeviceToHost, st));
            GB_TRY(hipMemcpyAsync(&root_ref, d_leaf_ref, sizeof(int), hipMemcpyDeviceToHost, st));
            GB_TRY(hipMemcpyAsync(root_box, g.t.box, sizeof(root_box), hipMemcpyDeviceToHost, st));
            GB_TRY(hipStreamSynchronize(st));
            GB_TRY(hipGetLastError());
            if (status) { err = "too many BVH leaves for the packed leaf reference"; ok = false; goto done; }
            out.n_wide = n_wide; out.n_nodes = n_nodes; out.n_leaves = n_leaves; out.n_leaf_prims = n_leaf_prims;
            out.root = n_internal > 0 ? 0 : root_ref;
            for (int k = 0; k < 3; ++k) { out.root_lo[k] = root_box[k]; out.root_hi[k] = root_box[3 + k]; }
            out.max_depth = g.max_depth;
            out.stack_bound = 3 * wide_depth + 2;
            g.lap("wide nodes");
        }
    }
done:
    if (!inputs_on_device) { hipFree(d_shape); hipFree(d_pshape); }
    hipFree(d_uniq); hipFree(d_dev_start); hipFree(d_leaf_ref); hipFree(d_cnt); hipFree(d_work[0]); hipFree(d_work[1]); hipFree(d_cone); hipFree(d_objects);
    if (!ok) { hipFree(out.nodes); hipFree(out.chain); hipFree(out.node_chain); hipFree(out.leaves); hipFree(out.prims); out = RtGpuScene(); }
    g.release();
    return ok;
}
