// BVH refit for keyframed primitives -- SURVEY 8(f) rows 1-2, opt-in through CrRenderParams.refit_boxes.
//
// The reference computes a wrapper's box once, from the primitives' construction-time boxes
// (src/objects/bvhwrapper.rs:47-50), and never again: Hittables::update_bb reaches only leaf primitives and its
// result is not read (objects/mod.rs:141, bvhwrapper.rs:102-106 carry the author's TODO).  A primitive that a
// keyframe moves out of its construction-time box is therefore clipped.  With refit_boxes = 0 this library keeps
// those stale boxes (that is the reference's image); with refit_boxes = 1 every wrapper box is re-derived for the
// frame before the render, from the same primitives and the same timeline arithmetic:
//
//   box of a primitive over the frame's ray-time interval [ta, tb] = union of its box at ta, at tb, at every key
//   end t1 inside (ta, tb), and at every key start t0 inside (ta, tb] taken twice -- with the key active (the
//   value from t0 on) and with keys starting exactly at t0 still inactive (the value just before t0).  A keyed
//   channel is piecewise linear in t with jumps only at key starts (timeline/mod.rs:233-263), so these times
//   carry its extremes.  The box at one time is Sphere::new's / Triangle::new's (sphere.rs:29-30,
//   triangle.rs:28-35) on the evaluated position; wrapper box = tight_enclose of its children (bvh.rs:67-73).
//
// A triangle with ScaleX / ScaleY / ScaleZ keys moves along v(t) * x(t) (+ y(t)): a product of two piecewise-linear
// functions, whose extremes need not sit at those times.  For such a primitive the translate part and the scale
// part (the winning scale key and its value) are sampled at those times INDEPENDENTLY and every combination is
// united: between two consecutive sample times both parts are linear, so the vertex is a bilinear function of
// (translate time, scale time) there and its coordinates are bounded by the four corner combinations, all of which
// are among the pairs.  Conservative (a superset), never too small.
//
// On a scene without primitive keys this reproduces the construction-time boxes exactly.  The topology is never
// changed.  (The test suite's CPU checker applies the same rule to its own tree.)
#pragma once
#include "pathtrace.hpp"

namespace cr {

// combine_and_compute (timeline/mod.rs:233-263) with a switch for the left limit at a key start:
// before_start = true treats a key whose t0 equals t as not yet active.
template <typename real>
CR_HD void timeline_eval_side(const Key<real>* keys, int n, real t, bool before_start, real& x, real& y, real& z, real& w,
                              int32_t* skind = nullptr) {
    x = real(0) + x; y = real(0) + y; z = real(0) + z;
    int32_t sk = -1;
    for (int i = 0; i < n; i++) {
        Key<real> k = keys[i];
        bool started = before_start ? (k.t0 < t) : (k.t0 <= t);
        bool active = (t > k.t1) || (started && t <= k.t1);
        if (!active) continue;
        real s = clamp01((t - k.t0) / (k.t1 - k.t0));
        if (k.channel <= 2) {
            real val = k.interp ? k.a * s : k.a;
            if (k.channel == 0) x = x + val; else if (k.channel == 1) y = y + val; else z = z + val;
        } else {
            w = k.interp ? k.a + (k.b - k.a) * s : k.a;
            sk = k.channel;
        }
    }
    if (skind) *skind = sk;
}

// A sample box with a coordinate that is not a number is not united (a zero-length LERP key evaluated exactly at its
// own start gives 0/0): the rule would otherwise depend on the order of its unions.  Ray times are drawn from a
// continuum, so the instant itself carries no rays.
template <typename real> CR_HD bool box_is_number(const real blo[3], const real bhi[3]) {
    return blo[0] == blo[0] && blo[1] == blo[1] && blo[2] == blo[2] && bhi[0] == bhi[0] && bhi[1] == bhi[1] && bhi[2] == bhi[2];
}
template <typename real> CR_HD void enclose(real lo[3], real hi[3], const real blo[3], const real bhi[3]) {
    for (int a = 0; a < 3; a++) {   // Interval::tight_enclose, utils.rs:629-633
        lo[a] = lo[a] <= blo[a] ? lo[a] : blo[a];
        hi[a] = hi[a] >= bhi[a] ? hi[a] : bhi[a];
    }
}

// Box of primitive p with its translate part taken at time t and its scale part at time ts, united into lo/hi
// (ts = t, bs = before_start: the box at one time).  use_keys = false: the construction-time box (no keys applied).
template <typename real>
CR_HD void prim_box_at2(const Prim<real>& p, const Key<real>* keys, real t, bool before_start, real ts, bool bs, real lo[3], real hi[3],
                        bool use_keys = true) {
    real blo[3], bhi[3];
    const Key<real>* k = keys + p.key_first;
    const int32_t n_keys = use_keys ? p.key_count : 0;
    real v[3][3];
    for (int j = 0; j < 3; j++) {
        real w = real(1), w2 = real(1), ux = p.g[3 * j], uy = p.g[3 * j + 1], uz = p.g[3 * j + 2];
        int32_t sk;
        v[j][0] = ux; v[j][1] = uy; v[j][2] = uz;
        timeline_eval_side(k, n_keys, t, before_start, v[j][0], v[j][1], v[j][2], w);
        timeline_eval_side(k, n_keys, ts, bs, ux, uy, uz, w2, &sk);
        const V3<real> q = scale_point(sk, w2, v[j][0], v[j][1], v[j][2]);
        v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z;
    }
    for (int a = 0; a < 3; a++) {
        bhi[a] = r_fmax(v[0][a], r_fmax(v[1][a], v[2][a]));
        blo[a] = r_fmin(v[0][a], r_fmin(v[1][a], v[2][a]));
    }
    if (box_is_number(blo, bhi)) enclose(lo, hi, blo, bhi);
}

template <typename real>
CR_HD void prim_box_at(const Prim<real>& p, const Key<real>* keys, real t, bool before_start, real lo[3], real hi[3], bool use_keys = true) {
    if (p.kind() != 0) { prim_box_at2(p, keys, t, before_start, t, before_start, lo, hi, use_keys); return; }
    real blo[3], bhi[3];
    const Key<real>* k = keys + p.key_first;
    const int32_t n_keys = use_keys ? p.key_count : 0;
    {   // Sphere::new, sphere.rs:29-30; Aabb::new_from_points, bvh.rs:44-64
        real c[3] = {p.g[0], p.g[1], p.g[2]}, r = p.g[3];
        timeline_eval_side(k, n_keys, t, before_start, c[0], c[1], c[2], r);
        for (int a = 0; a < 3; a++) {
            real l = c[a] + (-r), h = c[a] + r;
            if (l <= h) { blo[a] = l; bhi[a] = h; } else { blo[a] = h; bhi[a] = l; }
        }
    }
    if (box_is_number(blo, bhi)) enclose(lo, hi, blo, bhi);
}

// The sample times of the refit rule, indexed: 0 = ta, 1 = tb, then three per key (its start with the key active,
// its start with the key not yet active, its end).  false: that sample is not part of the rule.
template <typename real>
CR_HD bool refit_sample(const Prim<real>& p, const Key<real>* keys, real ta, real tb, int32_t i, real& t, bool& before_start) {
    before_start = false;
    if (i == 0) { t = ta; return true; }
    if (i == 1) { t = tb; return true; }
    const Key<real> k = keys[p.key_first + (i - 2) / 3];
    const int32_t which = (i - 2) % 3;
    if (which < 2) { t = k.t0; before_start = which == 1; return ta < k.t0 && k.t0 <= tb; }
    t = k.t1;
    return ta < k.t1 && k.t1 < tb;
}

// Box of primitive p over ray times [ta, tb], united into lo/hi (rule in the header comment).
template <typename real>
CR_HD void prim_box_over(const Prim<real>& p, const Key<real>* keys, real ta, real tb, real lo[3], real hi[3], bool use_keys = true) {
    prim_box_at(p, keys, ta, false, lo, hi, use_keys);
    if (!use_keys || p.key_count == 0) return;
    bool scaled = false;   // ScaleX / ScaleY / ScaleZ keys: translate and scale parts sampled independently
    for (int i = 0; i < p.key_count; i++) scaled |= keys[p.key_first + i].channel >= 4;
    if (scaled) {
        const int32_t n = 2 + 3 * p.key_count;
        for (int32_t i = 0; i < n; i++) {
            real t1; bool b1;
            if (!refit_sample(p, keys, ta, tb, i, t1, b1)) continue;
            for (int32_t j = 0; j < n; j++) {
                real t2; bool b2;
                if (refit_sample(p, keys, ta, tb, j, t2, b2)) prim_box_at2(p, keys, t1, b1, t2, b2, lo, hi);
            }
        }
        return;
    }
    prim_box_at(p, keys, tb, false, lo, hi);
    for (int i = 0; i < p.key_count; i++) {
        const Key<real> k = keys[p.key_first + i];
        if (ta < k.t0 && k.t0 <= tb) {
            prim_box_at(p, keys, k.t0, false, lo, hi);
            prim_box_at(p, keys, k.t0, true, lo, hi);
        }
        if (ta < k.t1 && k.t1 < tb) prim_box_at(p, keys, k.t1, false, lo, hi);
    }
}

#if defined(__HIPCC__)
// One level of the tree, deepest level first (entries are stored level by level, children after parents):
// leaves take their primitives' boxes, inner wrappers the union of their two children, already refitted.
// use_keys = 0 gives the construction-time boxes (how the LBVH builder fills in its boxes).
template <typename real, bool ORD>
__global__ void refit_level_kernel(typename EntryOf<real, ORD>::type* entries, int32_t begin, int32_t end, const Prim<real>* prims,
                                   const Key<real>* keys, real ta, real tb, int32_t use_keys, const int32_t* leaf_runs) {
    const int32_t i = begin + (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= end) return;
    const int32_t leaf = entries[i].leaf;
    real lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = r_inf(real(0)); hi[a] = -r_inf(real(0)); }
    if (leaf >= 0) {
        int32_t first = leaf >> 1, count = (leaf & 1) + 1;
        if (leaf & kLeafRun) { first = leaf_runs[2 * (leaf & kLeafRunIndex)]; count = leaf_runs[2 * (leaf & kLeafRunIndex) + 1]; }
        if ((leaf & kLeafRun) && (leaf & kLeafPseudo)) count = 0;   // tested without a box in the reference: the box stays empty
        for (int32_t k = 0; k < count; k++) prim_box_over(prims[first + k], keys, ta, tb, lo, hi, use_keys != 0);
    } else {
        const int32_t li = ORD ? ordered_left(leaf) : -leaf;   // siblings are adjacent in the level-order array
        for (int32_t c = li; c <= li + 1; c++) {
            const int32_t cl = entries[c].leaf;
            if (cl >= 0 && (cl & kLeafRun) && (cl & kLeafPseudo)) {
                // a primitive / list beside a BVHWrapper element: its own record keeps the empty box (the reference tests it
                // without one), but the wrapper above it spans it
                const int32_t first = leaf_runs[2 * (cl & kLeafRunIndex)], count = leaf_runs[2 * (cl & kLeafRunIndex) + 1];
                real clo[3], chi[3];
                for (int a = 0; a < 3; a++) { clo[a] = r_inf(real(0)); chi[a] = -r_inf(real(0)); }
                for (int32_t k = 0; k < count; k++) prim_box_over(prims[first + k], keys, ta, tb, clo, chi, use_keys != 0);
                enclose(lo, hi, clo, chi);
                continue;
            }
            const real* cb = entries[c].b;
            const real clo[3] = {cb[0], cb[2], cb[4]}, chi[3] = {cb[1], cb[3], cb[5]};
            enclose(lo, hi, clo, chi);
        }
    }
    real* b = entries[i].b;
    b[0] = lo[0]; b[1] = hi[0]; b[2] = lo[1]; b[3] = hi[1]; b[4] = lo[2]; b[5] = hi[2];
}
#endif

}   // namespace cr
