// rt_bvh_split.h — the arithmetic of one node of the reference's BVH build (BVHTree.cpp:39-153), shared by the host
// builder (rt_build.hip) and the GPU builder (rt_bvh_gpu.hip) so that both make the same decisions bit for bit.
#ifndef RT_BVH_SPLIT_H_
#define RT_BVH_SPLIT_H_

#include "rt_math.h"

struct Aabb {
    f3 lo, hi;
};

#define RT_BVH_BINS 4            // BVHTree.h:57

RT_HD Aabb empty_box() { Aabb b; b.lo = mk3(1e9f, 1e9f, 1e9f); b.hi = mk3(-1e9f, -1e9f, -1e9f); return b; }   // acmath.h:754-758
RT_HD Aabb box_of(f3 a, f3 b) { Aabb r; r.hi = max3(a, b); r.lo = min3(a, b); return r; }                       // acmath.h:760-764
RT_HD Aabb merge(const Aabb &a, const Aabb &b) { return box_of(max3(a.hi, b.hi), min3(a.lo, b.lo)); }            // acmath.h:922
RT_HD Aabb merge(const Aabb &a, f3 p) { return box_of(max3(a.hi, p), min3(a.lo, p)); }                          // acmath.h:932
RT_HD float surface_area(const Aabb &b) { f3 e = b.hi - b.lo; return 2 * (e.x * e.y + e.y * e.z + e.x * e.z); } // acmath.h:830
RT_HD f3 centroid(const Aabb &b) { return (b.hi + b.lo) * 0.5f; }
RT_HD int widest_axis(f3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }                   // MaxElementIndex acmath.h:609

// Split axis and centroid range of a node from the box of its primitives' centroids; false = the node stays a leaf
// (BVHTree.cpp:70-76: all centroids coincide on the widest axis).
RT_HD bool bvh_split_axis(const Aabb &cen, int &ax, float &clo, float &chi) {
    ax = widest_axis(cen.hi - cen.lo);
    clo = comp3(cen.lo, ax); chi = comp3(cen.hi, ax);
    return !(fabsf(chi - clo) < 0.00000001f);
}
// Bin of a primitive (ClippedPosition acmath.h:942, BVHTree.cpp:84-90): the unclamped value is what the partition compares.
RT_HD int bvh_bin_raw(const Aabb &box, int ax, float clo, float chi) {
    const float cp = (comp3(centroid(box), ax) - clo) / (chi - clo);
    return (int)(RT_BVH_BINS * cp);
}
// SAH over the four bins (BVHTree.cpp:92-125).  Returns the split bin, or -1 when the node stays a leaf (cost not below the
// primitive count).  left_count = primitives with bin <= split.
RT_HD int bvh_choose_split(const int *binCount, const Aabb *binBox, const Aabb &all, int count, int &left_count) {
    int fwdCount[RT_BVH_BINS], bwdCount[RT_BVH_BINS];
    Aabb fwdBox[RT_BVH_BINS], bwdBox[RT_BVH_BINS];
    fwdCount[0] = binCount[0]; fwdBox[0] = binBox[0];
    for (int i = 1; i < RT_BVH_BINS; ++i) { fwdBox[i] = merge(binBox[i], fwdBox[i - 1]); fwdCount[i] = binCount[i] + fwdCount[i - 1]; }
    bwdCount[RT_BVH_BINS - 1] = binCount[RT_BVH_BINS - 1]; bwdBox[RT_BVH_BINS - 1] = binBox[RT_BVH_BINS - 1];
    float cost[RT_BVH_BINS];
    const float saAll = surface_area(all);
    cost[RT_BVH_BINS - 1] = 0.125f + (fwdCount[RT_BVH_BINS - 1] * surface_area(fwdBox[RT_BVH_BINS - 1])) / saAll;
    for (int i = RT_BVH_BINS - 2; i >= 0; --i) {
        bwdBox[i] = merge(binBox[i], bwdBox[i + 1]); bwdCount[i] = binCount[i] + bwdCount[i + 1];
        cost[i] = 0.125f + (fwdCount[i] * surface_area(fwdBox[i]) + bwdCount[i + 1] * surface_area(bwdBox[i + 1])) / saAll;
    }
    float best = 1e9;
    int split = -1;
    for (int i = 0; i < RT_BVH_BINS; ++i) if (best > cost[i]) { best = cost[i]; split = i; }
    if (!(best < count)) return -1;
    left_count = fwdCount[split];
    return split;
}

#endif
