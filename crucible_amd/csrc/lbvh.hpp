// CR_BVH_LBVH -- SURVEY 8(f) row 1, "GPU LBVH": the wrapper tree built on the device instead of on the host.
//
//   1. lbvh_key_kernel     one 63-bit Morton key per primitive (21 bits per axis of the primitive-box centroid,
//                          normalised to the centroid bounds of the scene), next to its index;
//   2. hipcub radix sort   (key, index) pairs;
//   3. lbvh_topology_kernel one thread per internal node: its children by Karras' construction
//                          ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012):
//                          node i covers the keys sharing the longest common prefix around i; duplicates are
//                          told apart by their position, so the tree is a proper binary tree for any input;
//   4. (host, one O(n) pass) the node graph is numbered level by level with skip links -- the Entry layout every
//                          kernel already walks; leaves hold one primitive each;
//   5. refit_level_kernel  (refit.hpp, construction-time mode) fills in the boxes bottom-up.
//
// Same wrapper semantics as the other modes (box = union of the children, BVHWrapper::hit's walk), another
// topology: faster to build (a few milliseconds for 10^6 primitives against 0.3-0.4 s on the host), walked like any
// other tree, checked the same way (cr_export_bvh + the test suite's CPU checker).  Not the reference's tree.
#pragma once
#include "pathtrace.hpp"

namespace cr {

// Length of the common prefix of keys i and j (64 + common prefix of the positions when the keys are equal),
// -1 when j is outside [0, n).
CR_HD int lbvh_delta(const uint64_t* keys, int32_t n, int32_t i, int32_t j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __builtin_clzll(a ^ b);
    return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);   // i != j here
}

// Children of internal node i of a tree over n >= 2 sorted keys.  A child >= 0 is an internal node, a child < 0 is
// ~(position of a leaf in sorted order).
CR_HD void lbvh_children(const uint64_t* keys, int32_t n, int32_t i, int32_t& left, int32_t& right) {
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int32_t lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int32_t l = 0;
    for (int32_t t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int32_t j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int32_t s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int32_t g = i + s * d + (d < 0 ? -1 : 0);
    const int32_t lo = i < j ? i : j, hi = i < j ? j : i;
    left = lo == g ? ~g : g;
    right = hi == g + 1 ? ~(g + 1) : g + 1;
}

// 21 bits -> every third bit of a 63-bit word
CR_HD uint64_t lbvh_spread(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
CR_HD uint64_t lbvh_key(const double c[3], const double lo[3], const double inv_ext[3]) {
    uint64_t k = 0;
    for (int a = 0; a < 3; a++) {
        double u = (c[a] - lo[a]) * inv_ext[a];
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);   // also maps NaN to 0
        uint64_t q = (uint64_t)(u * 2097151.0);
        k |= lbvh_spread(q) << (2 - a);             // x is the most significant axis
    }
    return k;
}
// Centroid of a primitive's construction-time box (Sphere::new / Triangle::new boxes, keys not applied).
template <typename real> CR_HD void lbvh_centroid(const Prim<real>& p, double c[3]) {
    if (p.kind() == 0) { c[0] = (double)p.g[0]; c[1] = (double)p.g[1]; c[2] = (double)p.g[2]; return; }
    for (int a = 0; a < 3; a++) {
        const double x = (double)p.g[a], y = (double)p.g[3 + a], z = (double)p.g[6 + a];
        const double mn = x < y ? (x < z ? x : z) : (y < z ? y : z), mx = x > y ? (x > z ? x : z) : (y > z ? y : z);
        c[a] = 0.5 * (mn + mx);
    }
}

struct LbvhBounds { double lo[3], inv_ext[3]; };

#if defined(__HIPCC__)
template <typename real>
__global__ void lbvh_key_kernel(const Prim<real>* prims, int32_t n, LbvhBounds bnd, uint64_t* keys, int32_t* index) {
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    double c[3];
    lbvh_centroid(prims[i], c);
    keys[i] = lbvh_key(c, bnd.lo, bnd.inv_ext);
    index[i] = i;
}
__global__ void lbvh_topology_kernel(const uint64_t* keys, int32_t n, int32_t* children) {
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    int32_t l, r;
    lbvh_children(keys, n, i, l, r);
    children[2 * i] = l; children[2 * i + 1] = r;
}
#endif

}   // namespace cr
