// pathtrace.hpp -- the gfx950 path-tracing megakernel and the math it shares with
// the host-side camera set-up.  Hand-written for CDNA4: wave64, one persistent
// launch, scene staged in LDS when it fits, stackless fixed-order BVH walk,
// ballot/prefix regeneration of finished lanes.  No MFMA (branchy traversal).
//
// Reference semantics (citations relative to the Crucible tree):
//   Camera::cast_ray / ray_color / average_samples   src/camera/ray_casting.rs:64-173
//   viewport + basis math                            src/camera/rendering_compute.rs
//   BVHWrapper::hit, Aabb::hit                       src/objects/bvhwrapper.rs:96-126, bvh.rs:96-132
//   Sphere::hit, Triangle::hit                       src/objects/sphere.rs:60-105, triangle.rs:84-140
//   Materials::scatter                               src/materials/{lambertian,metal,dielectric}.rs
//   Textures::value, SkyboxImage::get_color          src/textures/*.rs, src/scene/mod.rs:37-45
//   Point3 / Color / Interval arithmetic             src/utils.rs:78-697
// Every expression keeps the reference's operand order; the build uses
// -ffp-contract=off so no FMA is formed, and IEEE div/sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cr {

#define CR_HD __host__ __device__ __forceinline__
#define CR_D __device__ __forceinline__

// ------------------------------------------------------------------ scalar traits
template <typename real> struct RealTraits;
template <> struct RealTraits<float> {
    static constexpr float eps = 1.1920928955078125e-07f;   // FLT_EPSILON (f32 twin of f64::EPSILON, triangle.rs:101)
    static constexpr float tiny = 0.0f;                     // 1e-160 rounds to 0 in f32 (utils.rs:131)
    static constexpr float pi = 3.14159265358979323846f;
};
template <> struct RealTraits<double> {
    static constexpr double eps = 2.220446049250313e-16;    // f64::EPSILON
    static constexpr double tiny = 1e-160;
    static constexpr double pi = 3.14159265358979323846;
};

CR_HD float r_sqrt(float x) { return __builtin_sqrtf(x); }
// f64 on the device: the compiler expands a correctly rounded square root into v_rsq_f64 and two Newton steps on
// g ~ sqrt(x), h ~ 1/(2 sqrt(x)), wrapped in a scale-up of inputs below 2^-767, the scale-down of the result and a select for
// 0 and +inf -- eight instructions that do nothing for an ordinary number.  When every active lane of the wave holds one (one
// scalar branch), the same ten-instruction core runs without them and returns the same bits (tests/test_gpu_sqrt.py compares
// the two on the device); any other wave takes the compiler's sequence.  The sphere test runs ~8 times per path segment.
CR_HD double r_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = (uint32_t)((unsigned long long)__builtin_bit_cast(long long, x) >> 32);
    const bool ordinary = (hi - 0x10000000u) < (0x7ff00000u - 0x10000000u);   // 2^-767 <= x < +inf
    if (__builtin_amdgcn_ballot_w64(!ordinary) == 0ull) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
#endif
    return __builtin_sqrt(x);
}
CR_HD float r_abs(float x) { return __builtin_fabsf(x); }
CR_HD double r_abs(double x) { return __builtin_fabs(x); }
CR_HD float r_floor(float x) { return __builtin_floorf(x); }
CR_HD double r_floor(double x) { return __builtin_floor(x); }
CR_HD float r_inf(float) { return __builtin_huge_valf(); }
CR_HD double r_inf(double) { return __builtin_huge_val(); }
// f64::min: the non-NaN operand when one is NaN (== fmin)
CR_HD float r_fmin(float a, float b) { return __builtin_fminf(a, b); }
CR_HD double r_fmin(double a, double b) { return __builtin_fmin(a, b); }
CR_HD float r_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
CR_HD double r_fmax(double a, double b) { return __builtin_fmax(a, b); }

// ------------------------------------------------------------------ Vec3 (utils.rs:72-340)
template <typename real> struct V3 { real x, y, z; };

template <typename real> CR_HD V3<real> mk(real x, real y, real z) { return V3<real>{x, y, z}; }
template <typename real> CR_HD V3<real> neg(V3<real> a) { return mk<real>(-a.x, -a.y, -a.z); }
template <typename real> CR_HD V3<real> add(V3<real> a, V3<real> b) { return mk<real>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename real> CR_HD V3<real> sub(V3<real> a, V3<real> b) { return add(a, neg(b)); }   // self + (-rhs), utils.rs:293-298
template <typename real> CR_HD V3<real> scale(real s, V3<real> a) { return mk<real>(s * a.x, s * a.y, s * a.z); }
template <typename real> CR_HD V3<real> divs(V3<real> a, real s) { return scale(real(1) / s, a); }   // (1.0/rhs)*self, utils.rs:335-340
template <typename real> CR_HD real len2(V3<real> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <typename real> CR_HD real dot(V3<real> a, V3<real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename real> CR_HD V3<real> cross(V3<real> a, V3<real> b) {
    return mk<real>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename real> CR_HD V3<real> unit(V3<real> a) { return divs(a, r_sqrt(len2(a))); }
template <typename real> CR_HD V3<real> reflect(V3<real> v, V3<real> n) { return sub(v, scale(real(2) * dot(v, n), n)); }   // utils.rs:149-151
template <typename real> CR_HD V3<real> refract(V3<real> v, V3<real> n, real eta) {   // utils.rs:157-163
    real cos_theta = r_fmin(dot(neg(v), n), real(1));
    V3<real> perp = scale(eta, add(v, scale(cos_theta, n)));
    V3<real> par = scale(-(r_sqrt(r_abs(real(1) - len2(perp)))), n);
    return add(perp, par);
}

template <typename real> CR_HD real clamp01(real x) { return x < real(0) ? real(0) : (x > real(1) ? real(1) : x); }   // f64::clamp

// clamped Color ops (utils.rs:445-607); a Color is a V3 with r,g,b in x,y,z
template <typename real> CR_HD V3<real> c_neg(V3<real> c) {
    real lo = c.x < c.y ? c.x : c.y; lo = lo < c.z ? lo : c.z;
    real hi = c.x > c.y ? c.x : c.y; hi = hi > c.z ? hi : c.z;
    real k = lo + hi;
    return mk<real>(r_abs(k - c.x), r_abs(k - c.y), r_abs(k - c.z));
}
template <typename real> CR_HD V3<real> c_add(V3<real> a, V3<real> b) { return mk<real>(clamp01(a.x + b.x), clamp01(a.y + b.y), clamp01(a.z + b.z)); }
template <typename real> CR_HD V3<real> c_scale(real s, V3<real> c) {   // impl Mul<Color> for f64
    V3<real> m = (s < real(0)) ? c_neg(c) : c;
    real p = r_abs(s);
    return mk<real>(clamp01(p * m.x), clamp01(p * m.y), clamp01(p * m.z));
}
template <typename real> CR_HD V3<real> c_mul(V3<real> a, V3<real> b) { return mk<real>(clamp01(a.x * b.x), clamp01(a.y * b.y), clamp01(a.z * b.z)); }
template <typename real> CR_HD V3<real> c_div(V3<real> c, real s) {
    V3<real> m = (s < real(0)) ? c_neg(c) : c;
    return c_scale(real(1) / r_abs(s), m);
}

// ------------------------------------------------------------------ RNG (DESIGN.md "RNG")
// One stream per (seed, pixel, sample): the key is SplitMix64's finaliser of those three (mix64), the draws are
// xorshift64* (Marsaglia's 12/25/27 xorshift scrambled by one multiply; Vigna, "An experimental exploration of
// Marsaglia's xorshift generators, scrambled", 2016) started from that key.  One 64-bit multiply per draw instead
// of SplitMix64's two -- 64-bit multiplies are quarter-rate here and the draws were ~10 % of the kernel.  Only the
// top 24 (f32) / 53 (f64) bits of an output are used, so f32 uniforms are truncations of the f64 ones.
constexpr uint64_t RNG_GAMMA = 0x9E3779B97F4A7C15ULL;
CR_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
CR_HD uint64_t rng_key(uint64_t seed_mixed, uint32_t pixel, uint32_t sample) {
    const uint64_t k = mix64(seed_mixed ^ (((uint64_t)pixel << 32) | (uint64_t)sample));
    return k ? k : RNG_GAMMA;   // xorshift state must not be zero
}
CR_HD uint64_t rng_next(uint64_t& s) {
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return s * 0x2545F4914F6CDD1DULL;
}
// The conversions go through 32-bit words (exact: 24 resp. 21+32 significant bits), which is much
// cheaper on the GPU than the generic u64 -> float sequence and yields the same value.
CR_HD float u01(uint64_t u, float) { return (float)(uint32_t)(u >> 40) * 0x1.0p-24f; }
CR_HD double u01(uint64_t u, double) {
    const uint64_t v = u >> 11;
    return ((double)(uint32_t)(v >> 32) * 4294967296.0 + (double)(uint32_t)v) * 0x1.0p-53;
}
template <typename real> CR_HD real rng_uniform(uint64_t& s) { return u01(rng_next(s), real(0)); }
template <typename real> CR_HD real rng_range(uint64_t& s, real lo, real hi) { return lo + (hi - lo) * rng_uniform<real>(s); }

template <typename real> CR_HD V3<real> random_unit_vector(uint64_t& s) {   // utils.rs:127-136
    for (;;) {
        real x = rng_range<real>(s, real(-1), real(1));
        real y = rng_range<real>(s, real(-1), real(1));
        real z = rng_range<real>(s, real(-1), real(1));
        V3<real> p = mk<real>(x, y, z);
        real lensq = len2(p);
        if (RealTraits<real>::tiny < lensq && lensq <= real(1)) return divs(p, r_sqrt(lensq));
    }
}

// ------------------------------------------------------------------ device scene layout
// Threaded BVH: the reference's wrapper tree (bvhwrapper.rs:46-78) with explicit links.
// A walk that goes to the left child on a box hit and to `skip` (the next wrapper in DFS
// pre-order after this subtree; n_entries = none) on a miss or after a leaf visits exactly the
// wrappers BVHWrapper::hit visits, in the same order (left subtree, then right).
// leaf < 0: inner wrapper, left child = -leaf.  leaf >= 0: span-1 or span-2 wrapper whose
// children are primitives: first = leaf >> 1, count = (leaf & 1) + 1, in leaf order.
// leaf >= 0 with kLeafRun set: a leaf that holds a HitList element; its primitives are the run
// (first, count) = leaf_runs[2 * (leaf & kLeafRunIndex)], [.. + 1] -- the list's visible objects and the wrapper's other
// child in the order BVHWrapper::hit and HitList::hit visit them (bvhwrapper.rs:108-120, hitlist.rs:55-61).
// Entries are stored level by level (BFS), so the first K entries are the top of the tree:
// scenes too large for LDS keep those K in LDS and read the rest through L2.
template <typename real> struct alignas(16) Entry {
    real b[6];      // xmin, xmax, ymin, ymax, zmin, zmax
    int32_t skip;
    int32_t leaf;
};
// CR_BVH_SAH_ORDERED: the same wrapper with one skip link per ray-direction octant, so the walk can take the
// child on the ray's side of the split first and still be stackless.  Siblings are adjacent in the level-order
// array, hence an inner wrapper stores only its left child and the split axis: leaf = -(left * 4 + axis), and the
// near child of a ray with octant bits oct (bit a set: direction[a] < 0) is left + ((oct >> axis) & 1).
// skip[oct] = the wrapper to visit after this subtree under that octant's order.  The head (b, leaf) overlays
// Entry's, so a refit or an export can read either.
template <typename real> struct alignas(16) EntryO {
    real b[6];
    int32_t unused;
    int32_t leaf;
    int32_t skip[8];
};
template <typename real, bool ORD> struct EntryOf { using type = Entry<real>; };
template <typename real> struct EntryOf<real, true> { using type = EntryO<real>; };
// f64 kernels: the f32 SCREENING copy of a wrapper -- the box rounded to f32, the same links.  The walk decides most box
// tests on this 32-byte record (half the bytes of Entry<double>) and reads the f64 box only when the f32 result is too
// close to call (screen_step below; DESIGN.md section 3.4 has the error bound).
// The links are stored the way the walk's inner loop consumes them: `skip` = BYTE offset of the wrapper to visit after a
// miss (index * 32; n_entries * 32 ends the walk) and `hit` = what a box hit leads to -- the byte offset of the left child, or
// kScreenLeaf | leaf code for a leaf wrapper.  One select picks the next offset and ONE unsigned compare (next >= n_entries * 32)
// sees both ways out of the loop; the offset is the LDS address / the 32-bit offset of a global load as it stands.
struct alignas(16) ScreenEntry {
    float b[6];
    uint32_t skip;
    uint32_t hit;
};
constexpr uint32_t kScreenLeaf = 0x80000000u;
constexpr int32_t kScreenMaxEntries = 1 << 26;   // offsets stay below bit 31
// ... and of an EntryO (CR_BVH_SAH_ORDERED): 64 bytes instead of 96.
// Links as in ScreenEntry (byte offsets into this array, index * 64): `hit` = the LEFT child's offset or kScreenLeaf | leaf
// code, `axis` = the split axis (3 for a leaf), `skip[o]` = the wrapper after a miss for a ray of direction octant o.  The
// near child is one bit-field extract and one shift-add away: hit + (((octant >> axis) & 1) << 6) -- and a leaf's axis 3 selects
// a bit no octant has, so the same two instructions leave its code alone.
struct alignas(16) ScreenEntryO {
    float b[6];
    uint32_t axis;
    uint32_t hit;
    uint32_t skip[8];
};
constexpr int32_t kScreenMaxEntriesO = 1 << 25;   // offsets stay below bit 31
template <bool ORD> struct ScreenOf { using type = ScreenEntry; };
template <> struct ScreenOf<true> { using type = ScreenEntryO; };
constexpr int32_t kLeafRun = 0x40000000;
constexpr int32_t kLeafPseudo = 0x20000000;   // with kLeafRun: the record stands for a primitive / list that BVHWrapper::hit tests without a box
constexpr int32_t kLeafRunIndex = 0x1fffffff;
CR_HD int32_t ordered_left(int32_t leaf) { return (-leaf) >> 2; }
CR_HD int32_t ordered_near(int32_t leaf, int32_t oct) { const int32_t v = -leaf; return (v >> 2) + ((oct >> (v & 3)) & 1); }

// Primitive record in leaf order.  g[0..3] sphere centre+radius, or g[0..8] a,b,c.
template <typename real> struct alignas(16) Prim {
    real g[9];
    int32_t kind_mat;   // kind in bit 0, material index above it
    int32_t key_first;
    int32_t key_count;
    CR_HD int32_t kind() const { return kind_mat & 1; }
    CR_HD int32_t mat() const { return kind_mat >> 1; }
};
template <typename real> struct alignas(16) Mat {
    real albedo[3];   // metal albedo | lambertian colour when its texture is solid | dielectric: {1/ior, r0(1/ior), r0(ior)}
    real param;       // scatter_prob | fuzz | refraction_index
    int32_t kind;
    int32_t tex;      // lambertian: texture index, or -1 when albedo[] already holds the solid colour
    real aux;         // lambertian: 1/|scatter_prob|, the factor Color / f64 multiplies by (utils.rs:599-607)
};
// Per-material / per-primitive values the reference recomputes at every hit are computed once at upload with
// the same IEEE operations: 1/radius for static spheres (Prim::g[4]; sphere.rs:97 `/ radius` = (1/radius)*v),
// 1/scatter_prob, 1/ior and Schlick's r0 = ((1-ri)/(1+ri))^2 for both orientations (dielectric.rs:21-38).
template <typename real> struct alignas(16) Tex {
    real color[3];
    real inv_scale;
    int32_t kind, even, odd, image;
};
struct ImageRef { int32_t w, h; uint32_t offset; uint32_t pad; };   // texels: RGBA8 at texels[offset + y*w + x]
template <typename real> struct Key { real t0, t1, a, b; int32_t channel, interp; };

// Camera in `real`.  The four scalars are computed in f64 at set-up as fix_viewport does
// (rendering_compute.rs:5-11,71-73) and rounded once; everything per-sample is `real`.
template <typename real> struct CamConst {
    int32_t W, H;
    real viewport_width, viewport_height, focus_dist, defocus_radius;
    int32_t defocus_on, animated;
    V3<real> from, at, vup;
    int32_t from_key_first, from_key_count, at_key_first, at_key_count;
    // static camera: the per-sample vectors, precomputed with the same expression tree
    V3<real> p00, pdu, pdv, ddu, ddv;
};

template <typename real> struct KernelArgs {
    const Entry<real>* entries;
    const Prim<real>* prims;
    const Mat<real>* mats;
    const Tex<real>* texs;
    const ImageRef* images;
    const uint32_t* texels;
    const Key<real>* keys;
    const Key<real>* cam_keys;   // this launch's camera keyframes (look_from keys, then look_at keys)
    const int32_t* leaf_runs;    // (first, count) pairs for the leaves flagged kLeafRun
    const void* screen;          // f64 SCREEN kernels: one screening record per wrapper (ScreenEntry, or ScreenEntryO for an ordered tree)
    int32_t n_entries, n_prims, n_mats, n_texs;
    int32_t lds_entries;      // entries staged in LDS (all of them, or the top levels of a large tree)
    int32_t lds_side;         // RES_TOP: materials and textures follow the entry window in LDS (they are small even when
                              // the tree is not: the teapot scenes have 2 materials), so shading reads no global tables
    int32_t sky_kind, sky_image;
    CamConst<real> cam;
    int32_t sample_begin, sample_end, samples_total, max_depth;
    uint64_t seed_mixed;      // mix64(seed + GAMMA)
    real current_time, shutter_length;
    int32_t output_sum;
    uint32_t tiles_x, tiles_y;
    uint32_t* work_counter;   // zeroed before launch
    // Sample-granular scheduling (megakernel; speed only).  sg_on = 0: a lane owns a pixel and sums its samples
    // in a register.  sg_on = 1: a work item is one (pixel, sample); 64 consecutive items are a tile of
    // 2^sg_lw x 2^sg_lh pixels times 64 >> (sg_lw + sg_lh) consecutive samples, tiles_x/tiles_y count those tiles,
    // sg_groups such groups cover a tile's samples [sample_begin, sample_end), and each finished sample's colour
    // goes to sample_buf[item * 3 .. +2], item = its work-item index: the 64 colours of a group are one contiguous run
    // written by one wave (they merge in that XCD's L2 into whole lines), and sg_finalize_kernel adds them in order.
    uint32_t sg_on, sg_lw, sg_lh, sg_groups;
    uint32_t sg_total;        // work items of the launch (tiles * sg_groups * 64)
    uint32_t sg_chunk;        // work items a wave takes per atomic on the counter (a multiple of 64)
    real* sample_buf;
    uint64_t* counters;       // [0] segments [1] node tests [2] prim tests [3] texel fetches
    real* att_stack;          // max_depth * n_threads records of 3 reals, level-major
    uint32_t n_threads;
    real* out;
    uint32_t walk_exit_lanes;   // megakernel: leave the walk once this many lanes are done walking (speed only)
    uint32_t walk_round_steps;  // wrappers a lane may step through before the wave intersects the parked leaves (speed only)
    uint32_t walk_leaf_min;     // parked lanes a leaf phase waits for while other lanes can still step (speed only; 0 = every round)
    int32_t uniform_kind;       // 0 / 1: every primitive is a sphere / a triangle (the test need not load the record's kind word); -1: mixed
    uint32_t queue_walk_waves;  // queue_kernel: how many of the workgroup's 16 waves walk (the rest shade)
    uint32_t queue_min_batch, queue_patience;   // queue_kernel: shaders wait for this many hits, at most this many polls
    // CR_SUM_RELAXED (the RELAX kernels): per-pixel fixed-point sums instead of per-sample colours.  A finished sample adds
    // round(colour * fx_scale) to three 64-bit integers of its pixel -- integer adds commute, so the image does not
    // depend on which wave finished which sample when.  Bit 63 of a sum is the NaN flag (a colour that is not a number).
    unsigned long long* fx_acc;   // [H * W * 3], zeroed before the first launch of a render
    double fx_scale;              // 2^S, S = 52 for up to 2047 samples per pixel
    uint32_t fx_lds_off;          // byte offset of the waves' LDS accumulators: 2 slots per wave, each one work tile
                                  // (2^(sg_lw + sg_lh) pixels x 3 channels) of 64-bit words
};
constexpr unsigned long long kFxNaN = 0x8000000000000000ull;
constexpr size_t fx_lds_bytes(int block, uint32_t tile_log2) { return (size_t)(block / 64) * 2 * ((size_t)3 << tile_log2) * sizeof(unsigned long long); }

// ------------------------------------------------------------------ timeline (timeline/mod.rs:233-263)
// combine_and_compute = S * T * (0,0,0,1): T's last column is the initial position plus every active translate
// value, added in list order; S is the LAST active scale transform (:249-255) -- the initial one (sphere:
// diag(1,1,1,r); others: diag(s,s,s,s), s = 1), a ScaleR key (radius), or a ScaleX / ScaleY / ScaleZ key
// (transform_builder.rs:101-346).  x,y,z come back as T's column, w as the scale value, *skind as the channel of
// the scale key that won (-1: the initial scale).
template <typename real> CR_HD void timeline_eval(const Key<real>* keys, int n, real t, real& x, real& y, real& z, real& w, int32_t* skind = nullptr) {
    x = real(0) + x; y = real(0) + y; z = real(0) + z;   // identity * initial translate
    int32_t sk = -1;
    for (int i = 0; i < n; i++) {
        Key<real> k = keys[i];
        bool active = (t > k.t1) || (k.t0 <= t && t <= k.t1);   // is_less(t) || contains(t)
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
// S * (x, y, z, 1) for a non-sphere point (a triangle vertex; the 4th component is dropped, triangle.rs:95-97).
// ScaleX is diag(v,1,1,1) and ScaleZ diag(1,1,v,1); ScaleY writes v into row 1, column 0 and leaves the diagonal
// at 1 (transform_builder.rs:228-246), so it shears y by v*x.  Rows of S that hold only a unit diagonal return the
// coordinate unchanged (1*y plus zeros).
template <typename real> CR_HD V3<real> scale_point(int32_t skind, real v, real x, real y, real z) {
    if (skind == 4) return mk<real>(v * x, y, z);
    if (skind == 5) return mk<real>(x, v * x + y, z);
    if (skind == 6) return mk<real>(x, y, v * z);
    return mk<real>(v * x, v * y, v * z);   // build_other_scaler, matrix_builder.rs:63-86
}
// One vertex timeline of a triangle (a_timeline / b_timeline / c_timeline share their keys, scene_animator.rs).
template <typename real> CR_HD V3<real> timeline_vertex(const Key<real>* keys, int n, real t, V3<real> p) {
    real w = real(1);
    int32_t sk;
    timeline_eval(keys, n, t, p.x, p.y, p.z, w, &sk);
    return scale_point(sk, w, p.x, p.y, p.z);
}

// ------------------------------------------------------------------ camera (rendering_compute.rs)
template <typename real> struct CamFrame { V3<real> from, p00, pdu, pdv, ddu, ddv; };

template <typename real> CR_HD CamFrame<real> camera_frame(const CamConst<real>& c, V3<real> from, V3<real> at) {
    CamFrame<real> f;
    f.from = from;
    V3<real> w = unit(sub(from, at));                           // w_basis :91-96
    V3<real> u = unit(cross(c.vup, w));                         // u_basis :81-83
    V3<real> v = cross(w, u);                                   // v_basis :86-88
    V3<real> vu = scale(c.viewport_width, u);                   // viewport_u :18-20
    V3<real> vv = scale(c.viewport_height, neg(v));             // viewport_v :25-28
    f.pdu = divs(vu, (real)c.W);                                // pixel_delta_u :34-36
    f.pdv = divs(vv, (real)c.H);                                // pixel_delta_v :42-44
    V3<real> ul = sub(from, scale(c.focus_dist, w));            // viewport_upperleft :51-56
    ul = sub(ul, divs(vu, real(2)));
    ul = sub(ul, divs(vv, real(2)));
    f.p00 = add(ul, scale(real(0.5), add(f.pdu, f.pdv)));       // pixel_start_location :59-61
    f.ddu = scale(c.defocus_radius, u);                         // defocus_disk_u :99-101
    f.ddv = scale(c.defocus_radius, v);                         // defocus_disk_v :104-106
    return f;
}

#if defined(__HIPCC__)
// ================================================================== device only

// Diagnostic build (-DCR_DIAG, scripts/diag only): per-lane event counts that the megakernel sums per wave and adds
// to counters[16 + i]; "wave" counts are taken by the first active lane of the wave at that point, "lane" counts by
// every active lane, so lane / (64 * wave) is the lane occupancy of that piece of code.  The product build compiles
// none of it (Diag is empty, CR_DIAG_* expand to nothing).
#ifdef CR_DIAG
enum { DG_BOX_WAVE = 0, DG_BOX_LANE, DG_PRIM_WAVE, DG_PRIM_LANE, DG_ROUND_WAVE, DG_ROUND_LANE, DG_LEAFPH_WAVE, DG_LEAFPH_LANE,
       DG_SHADE_WAVE, DG_SHADE_LANE, DG_LAMB_LANE, DG_METAL_LANE, DG_DIEL_LANE, DG_SKY_LANE, DG_RUV_WAVE, DG_RUV_LANE,
       DG_REGEN_WAVE, DG_REGEN_LANE, DG_OUTER_WAVE, DG_UNWIND_WAVE, DG_UNWIND_LANE, DG_HITSH_WAVE, DG_HITSH_LANE, DG_BAND_WAVE, DG_BAND_LANE, DG_N };
struct Diag { uint32_t v[DG_N]; };
#define CR_DIAG_LEADER() ((threadIdx.x & 63u) == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1))
#define CR_DIAG_HIT(dg, wave_i, lane_i) do { if (dg) { (dg)->v[lane_i]++; if (CR_DIAG_LEADER()) (dg)->v[wave_i]++; } } while (0)
#define CR_DIAG_LANE(dg, lane_i) do { if (dg) (dg)->v[lane_i]++; } while (0)
#else
struct Diag {};
#define CR_DIAG_HIT(dg, wave_i, lane_i) ((void)0)
#define CR_DIAG_LANE(dg, lane_i) ((void)0)
#endif

template <typename real> struct Hit {
    real t;
    int32_t prim;     // leaf-order index, -1 = miss
};

// Aabb::hit (bvh.rs:96-132) with 1/dir hoisted (same value every node).  The per-axis
// early return is folded into one final test: once max <= min holds it keeps holding,
// because new_min >= min and new_max <= max for non-NaN min/max.
template <typename real>
CR_D bool box_hit(const real* b, V3<real> o, V3<real> inv, real tmin, real tmax) {
    real t0 = (b[0] - o.x) * inv.x, t1 = (b[1] - o.x) * inv.x;
    real nmin, nmax;
    if (t0 < t1) { nmin = t0 > tmin ? t0 : tmin; nmax = t1 < tmax ? t1 : tmax; }
    else { nmin = t1 > tmin ? t1 : tmin; nmax = t0 < tmax ? t0 : tmax; }
    tmin = nmin; tmax = nmax;
    t0 = (b[2] - o.y) * inv.y; t1 = (b[3] - o.y) * inv.y;
    if (t0 < t1) { nmin = t0 > tmin ? t0 : tmin; nmax = t1 < tmax ? t1 : tmax; }
    else { nmin = t1 > tmin ? t1 : tmin; nmax = t0 < tmax ? t0 : tmax; }
    tmin = nmin; tmax = nmax;
    t0 = (b[4] - o.z) * inv.z; t1 = (b[5] - o.z) * inv.z;
    if (t0 < t1) { nmin = t0 > tmin ? t0 : tmin; nmax = t1 < tmax ? t1 : tmax; }
    else { nmin = t1 > tmin ? t1 : tmin; nmax = t0 < tmax ? t0 : tmax; }
    return !(nmax <= nmin);
}

// The same test in min/max form on (min,max) pairs, which the compiler maps onto
// v_pk_add_f32 / v_pk_mul_f32.  Identical to box_hit whenever no slab distance is NaN, i.e.
// whenever 1/dir is finite on every axis (then `t0 < t1 ? .. : ..` is min/max of two non-NaN
// numbers and `a > b ? a : b` is max; signed zeros only ever feed comparisons).  Rays with an
// infinite 1/dir component walk with box_hit instead.
template <typename real> using Pair = real __attribute__((ext_vector_type(2)));
CR_D float r_min(float a, float b) { return __builtin_fminf(a, b); }
CR_D float r_max(float a, float b) { return __builtin_fmaxf(a, b); }
CR_D double r_min(double a, double b) { return __builtin_fmin(a, b); }
CR_D double r_max(double a, double b) { return __builtin_fmax(a, b); }
template <typename real>
CR_D bool box_hit_fast(const real* b, Pair<real> ox, Pair<real> oy, Pair<real> oz, Pair<real> ix, Pair<real> iy, Pair<real> iz,
                       real tmin, real tmax) {
    Pair<real> tx = (Pair<real>{b[0], b[1]} - ox) * ix;
    Pair<real> ty = (Pair<real>{b[2], b[3]} - oy) * iy;
    Pair<real> tz = (Pair<real>{b[4], b[5]} - oz) * iz;
    real lo = r_max(r_max(r_min(tx.x, tx.y), r_min(ty.x, ty.y)), r_max(r_min(tz.x, tz.y), tmin));
    real hi = r_min(r_min(r_max(tx.x, tx.y), r_max(ty.x, ty.y)), r_min(r_max(tz.x, tz.y), tmax));
    return !(hi <= lo);
}
// The complement, for callers that combine it with other masks: Aabb::hit's `max <= min -> miss`.
// min(h, tmax) <= lo is evaluated as h <= lo || tmax <= lo: identical for every input the fast path sees (the slab
// distances are never NaN there, and a NaN tmax is ignored by either form), and one instruction shorter because
// v_min would first have to canonicalise tmax.
template <typename real>
CR_D bool box_miss_fast(const real* b, Pair<real> ox, Pair<real> oy, Pair<real> oz, Pair<real> ix, Pair<real> iy, Pair<real> iz,
                        real tmin, real tmax) {
    Pair<real> tx = (Pair<real>{b[0], b[1]} - ox) * ix;
    Pair<real> ty = (Pair<real>{b[2], b[3]} - oy) * iy;
    Pair<real> tz = (Pair<real>{b[4], b[5]} - oz) * iz;
    real lo = r_max(r_max(r_min(tx.x, tx.y), r_min(ty.x, ty.y)), r_max(r_min(tz.x, tz.y), tmin));
    real hi = r_min(r_min(r_max(tx.x, tx.y), r_max(ty.x, ty.y)), r_max(tz.x, tz.y));
    return (hi <= lo) | (tmax <= lo);
}

// Sphere::hit root search (sphere.rs:72-95): returns t or a negative number for a miss.
template <typename real>
CR_D bool sphere_t(real cx, real cy, real cz, real radius, V3<real> o, V3<real> d, real a /* |d|^2 */, real tmin, real tmax, real& t_out) {
    V3<real> oc = sub(mk<real>(cx, cy, cz), o);
    real h = dot(d, oc);
    real c = len2(oc) - radius * radius;
    real disc = h * h - a * c;
    if (disc < real(0)) return false;
    real sqrtd = r_sqrt(disc);
    real root = (h - sqrtd) / a;
    if (!(tmin < root && root < tmax)) {
        root = (h + sqrtd) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t_out = root;
    return true;
}

// Triangle::hit up to t (triangle.rs:95-123)
template <typename real>
CR_D bool triangle_t(V3<real> a, V3<real> b, V3<real> c, V3<real> o, V3<real> d, real tmin, real tmax, real& t_out) {
    V3<real> e1 = sub(b, a), e2 = sub(c, a);
    V3<real> rce2 = cross(d, e2);
    real det = dot(e1, rce2);
    if (det > -RealTraits<real>::eps && det < RealTraits<real>::eps) return false;
    real inv_det = real(1) / det;
    V3<real> s = sub(o, a);
    real u = inv_det * dot(s, rce2);
    if (!(real(0) <= u && u <= real(1))) return false;
    V3<real> sce1 = cross(s, e1);
    real v = inv_det * dot(d, sce1);
    if (v < real(0) || u + v > real(1)) return false;
    real t = inv_det * dot(e2, sce1);
    if (!(tmin < t && t < tmax)) return false;
    t_out = t;
    return true;
}

// Rust `as i32`: truncation toward zero, saturating, NaN -> 0 -- which is what v_cvt_i32_f32 / v_cvt_i32_f64 do (ISA: out-of-range
// values and infinities saturate, NaN converts to 0).  Written as the instruction: C++'s cast is undefined out of range, and the
// guarded form costs three compares and their branches per conversion (the checker texture makes three per hit).
// tests/sqrt_check.hip compares it with as_i32_reference on the device.
template <typename real> CR_HD int32_t as_i32_reference(real x) {
    if (!(x == x)) return 0;
    if (x <= real(-2147483648.0)) return INT32_MIN;
    if (x >= real(2147483647.0)) return INT32_MAX;
    return (int32_t)x;
}
CR_D int32_t as_i32(float x) { int32_t r; asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x)); return r; }
CR_D int32_t as_i32(double x) { int32_t r; asm("v_cvt_i32_f64_e32 %0, %1" : "=v"(r) : "v"(x)); return r; }
template <typename real> CR_D uint32_t as_index(real x, int32_t n) {   // `as usize` then clamp to n-1 (img_loader.rs:72-73)
    if (!(x == x) || x <= real(0)) return 0;
    if (x >= (real)n) return (uint32_t)(n - 1);
    return (uint32_t)x;
}

template <typename real>
CR_D V3<real> image_lookup(const ImageRef* images, const uint32_t* texels, int image, real u, real v, uint32_t& n_texel) {
    ImageRef im = images[image];
    u = clamp01(u);
    v = real(1) - clamp01(v);
    uint32_t i = as_index(u * (real)im.w, im.w);
    uint32_t j = as_index(v * (real)im.h, im.h);
    uint32_t px = texels[im.offset + j * (uint32_t)im.w + i];
    n_texel++;
    return mk<real>((real)(px & 255u) / real(255), (real)((px >> 8) & 255u) / real(255), (real)((px >> 16) & 255u) / real(255));
}

// ------------------------------------------------------------------ software atan2 / asin / acos
// The reference calls f64::atan2 / asin / acos (ray_casting.rs:137-138, sphere.rs:42-43): the platform libm, whose
// last-ulp behaviour differs between glibc and the device's ocml -- enough to move a texel index.  The library and
// the oracle therefore evaluate ONE documented algorithm with +, -, *, /, sqrt only (DESIGN.md "software
// trigonometry"; coefficients from scripts/gen_trig_coeffs.py), which makes the image-texture and spherical-sky
// scenes bit-exact:
//   atan:  fdlibm's argument reduction at 7/16, 11/16, 19/16, 39/16, then atan(t) = t - t*(z*A(z)), z = t*t,
//          result = hi - ((t*(z*A(z)) - lo) - t) with hi + lo = atan(0.5), pi/4, atan(1.5), pi/2;
//   asin:  |x| < 0.5: x + x*(z*S(z)), z = x*x; else pi/2 - 2*asin(sqrt((1 - |x|)/2)); acos likewise;
//   A, S:  Horner from the highest coefficient, multiply then add.  Within 1-3 ulp of the correctly rounded value.
template <typename real> struct TrigK;
template <> struct TrigK<double> {
    static CR_HD double A(double z) {
        double p = -0x1.e4167464d3de8p-7;
        p = p * z + 0x1.0f62bba6a2558p-5; p = p * z + -0x1.7001816fd063fp-5; p = p * z + 0x1.ab59b417b2d3fp-5;
        p = p * z + -0x1.e170800a46210p-5; p = p * z + 0x1.110c9ce7b0572p-4; p = p * z + -0x1.3b1375ce5bdc6p-4;
        p = p * z + 0x1.745d154c84f7ap-4; p = p * z + -0x1.c71c71bd2b8bcp-4; p = p * z + 0x1.2492492485503p-3;
        p = p * z + -0x1.99999999998c5p-3; p = p * z + 0x1.5555555555555p-2;
        return p;
    }
    static CR_HD double S(double z) {
        double p = 0x1.06c051be25377p-5;
        p = p * z + -0x1.dfdd83264a978p-6; p = p * z + 0x1.b20b9dc229eb5p-6; p = p * z + -0x1.641b6703bb104p-9;
        p = p * z + 0x1.1e6dafec868fcp-7; p = p * z + 0x1.c232290f7ae75p-8; p = p * z + 0x1.14f7ebcffc822p-7;
        p = p * z + 0x1.3fa92e3923959p-7; p = p * z + 0x1.7a8b73dc1b007p-7; p = p * z + 0x1.c99964e8e2de8p-7;
        p = p * z + 0x1.1c4ec5dfe81d9p-6; p = p * z + 0x1.6e8ba2e2f8089p-6; p = p * z + 0x1.f1c71c71dc217p-6;
        p = p * z + 0x1.6db6db6db6c75p-5; p = p * z + 0x1.3333333333334p-4; p = p * z + 0x1.5555555555555p-3;
        return p;
    }
    static constexpr double at_hi0 = 0x1.dac670561bb4fp-2, at_lo0 = 0x1.a2b7f222f65e2p-56;   // atan(0.5)
    static constexpr double at_hi1 = 0x1.921fb54442d18p-1, at_lo1 = 0x1.1a62633145c07p-55;   // pi/4
    static constexpr double at_hi2 = 0x1.f730bd281f69bp-1, at_lo2 = 0x1.007887af0cbbdp-56;   // atan(1.5)
    static constexpr double pio2_hi = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
    static constexpr double pi_hi = 0x1.921fb54442d18p+1, pi_lo = 0x1.1a62633145c07p-53;
};
template <> struct TrigK<float> {
    static CR_HD float A(float z) {
        float p = -0x1.8b0b06p-5f;
        p = p * z + 0x1.5d79d8p-4f; p = p * z + -0x1.c4f1ecp-4f; p = p * z + 0x1.248626p-3f; p = p * z + -0x1.999968p-3f;
        p = p * z + 0x1.555556p-2f;
        return p;
    }
    static CR_HD float S(float z) {
        float p = 0x1.fb7ca4p-6f;
        p = p * z + 0x1.5a80ap-7f; p = p * z + 0x1.82db24p-6f; p = p * z + 0x1.efedf8p-6f; p = p * z + 0x1.6dc0fp-5f;
        p = p * z + 0x1.33331ep-4f; p = p * z + 0x1.555556p-3f;
        return p;
    }
    static constexpr float at_hi0 = 0.46364760398864746f, at_lo0 = 5.01215868808913e-09f;
    static constexpr float at_hi1 = 0.7853981852531433f, at_lo1 = -2.1855694143368964e-08f;
    static constexpr float at_hi2 = 0.9827937483787537f, at_lo2 = -2.5131424052915463e-08f;
    static constexpr float pio2_hi = 1.5707963705062866f, pio2_lo = -4.371138828673793e-08f;
    static constexpr float pi_hi = 3.1415927410125732f, pi_lo = -8.742277657347586e-08f;
};
// atan(x) for x >= 0 (also +inf; NaN propagates)
template <typename real> CR_HD real soft_atan_pos(real x) {
    using K = TrigK<real>;
    real t, hi = real(0), lo = real(0);
    bool reduced = true;
    if (x < real(0.4375)) { t = x; reduced = false; }
    else if (x < real(0.6875)) { t = (real(2) * x - real(1)) / (real(2) + x); hi = K::at_hi0; lo = K::at_lo0; }
    else if (x < real(1.1875)) { t = (x - real(1)) / (x + real(1)); hi = K::at_hi1; lo = K::at_lo1; }
    else if (x < real(2.4375)) { t = (x - real(1.5)) / (real(1) + real(1.5) * x); hi = K::at_hi2; lo = K::at_lo2; }
    else { t = real(-1) / x; hi = K::pio2_hi; lo = K::pio2_lo; }
    const real z = t * t;
    const real s = t * (z * K::A(z));
    return reduced ? hi - ((s - lo) - t) : t - s;
}
template <typename real> CR_HD real soft_atan2(real y, real x) {
    using K = TrigK<real>;
    if (x != x || y != y) return x + y;
    const bool sx = __builtin_signbit(x), sy = __builtin_signbit(y);
    const real inf = r_inf(real(0));
    if (y == real(0)) return sx ? (sy ? -K::pi_hi : K::pi_hi) : y;
    if (x == real(0)) return sy ? -K::pio2_hi : K::pio2_hi;
    const real ax = r_abs(x), ay = r_abs(y);
    if (ax == inf) {
        if (ay == inf) { const real q = sx ? real(3) * K::at_hi1 : K::at_hi1; return sy ? -q : q; }
        const real q = sx ? K::pi_hi : real(0);
        return sy ? -q : q;
    }
    if (ay == inf) return sy ? -K::pio2_hi : K::pio2_hi;
    const real z = soft_atan_pos(ay / ax);
    const real res = sx ? K::pi_hi - (z - K::pi_lo) : z;
    return sy ? -res : res;
}
template <typename real> CR_HD real soft_asin(real x) {
    using K = TrigK<real>;
    const real ax = r_abs(x);
    if (ax < real(0.5)) { const real z = x * x; return x + x * (z * K::S(z)); }
    if (!(ax <= real(1))) return (x - x) / (x - x);
    const real t = (real(1) - ax) * real(0.5);
    const real s = r_sqrt(t);
    const real res = K::pio2_hi - (real(2) * (s + s * (t * K::S(t))) - K::pio2_lo);
    return x < real(0) ? -res : res;
}
template <typename real> CR_HD real soft_acos(real x) {
    using K = TrigK<real>;
    const real ax = r_abs(x);
    if (ax < real(0.5)) { const real z = x * x; return K::pio2_hi - (x - (K::pio2_lo - x * (z * K::S(z)))); }
    if (!(ax <= real(1))) return (x - x) / (x - x);
    if (x < real(0)) {
        const real t = (real(1) + x) * real(0.5);
        const real s = r_sqrt(t);
        const real w = (t * K::S(t)) * s - K::pio2_lo;
        return K::pi_hi - real(2) * (s + w);
    }
    const real t = (real(1) - x) * real(0.5);
    const real s = r_sqrt(t);
    return real(2) * (s + s * (t * K::S(t)));
}
template <typename real> CR_D real r_atan2(real y, real x) { return soft_atan2(y, x); }
template <typename real> CR_D real r_asin(real x) { return soft_asin(x); }
template <typename real> CR_D real r_acos(real x) { return soft_acos(x); }

// random_unit_vector (utils.rs:127-136) and random_in_unit_disk (utils.rs:110-124) for the f64 kernel, with the
// rejection test SCREENED in f32.  A wave leaves a rejection loop only when its slowest lane does (six rounds for a
// unit vector with ~36 lanes drawing), and in f64 most of a round is converting three 64-bit draws to doubles just to
// throw half of them away.  The screen uses the top 24 bits of each draw: xf = -1 + 2*(u >> 40)*2^-24 is EXACT in f32
// and within 2^-23 of the f64 coordinate, so the f32 squared length lf differs from the f64 one by less than 2e-6
// (3 * 2 * 2^-23 from the truncation, < 1e-6 from f32 rounding).  Hence
//     lf > 1 + 1e-5              =>  the reference's test `lensq <= 1` fails      -> next round, nothing converted
//     1e-5 < lf < 1 - 1e-5       =>  it passes (1e-160 < lensq <= 1)             -> leave the loop with the raw draws
// and only a candidate inside the 2e-5 band (or with lf <= 1e-5) is decided by evaluating the reference's f64
// expression on the spot -- a branch no lane takes in ~99.9 % of the rounds.  The accepted candidate is converted to
// f64 once, after the loop, with the reference's expression tree; draws consumed and results are those of the plain
// loop, bit for bit (the parity tests run every f64 scene through this).
CR_D float screen_coord(uint64_t u) {   // -1 + 2 * u01(u, float): (top24 - 2^23) * 2^-23, exact
    return (float)((int32_t)(uint32_t)(u >> 40) - (int32_t)(1 << 23)) * 0x1.0p-23f;
}
template <typename real> CR_D V3<real> random_unit_vector_dev(uint64_t& s) {
    if constexpr (!std::is_same<real, double>::value) return random_unit_vector<real>(s);
    else {
        uint64_t ux, uy, uz;
        for (;;) {
            ux = rng_next(s); uy = rng_next(s); uz = rng_next(s);
            const float fx = screen_coord(ux), fy = screen_coord(uy), fz = screen_coord(uz);
            const float lf = fx * fx + fy * fy + fz * fz;
            if (lf > 1.0f + 1e-5f) continue;
            if (lf > 1e-5f && lf < 1.0f - 1e-5f) break;
            const double x = -1.0 + 2.0 * u01(ux, 0.0), y = -1.0 + 2.0 * u01(uy, 0.0), z = -1.0 + 2.0 * u01(uz, 0.0);
            const double lensq = x * x + y * y + z * z;
            if (RealTraits<double>::tiny < lensq && lensq <= 1.0) break;
        }
        // rng_range(-1, 1) = lo + (hi - lo) * u with hi - lo = 2 exactly
        const V3<double> p = mk<double>(-1.0 + 2.0 * u01(ux, 0.0), -1.0 + 2.0 * u01(uy, 0.0), -1.0 + 2.0 * u01(uz, 0.0));
        return divs(p, r_sqrt(len2(p)));
    }
}
template <typename real> CR_D void random_in_unit_disk_dev(uint64_t& s, real& px, real& py) {
    if constexpr (!std::is_same<real, double>::value) {
        for (;;) {
            px = rng_range<real>(s, real(-1), real(1));
            py = rng_range<real>(s, real(-1), real(1));
            if (px * px + py * py + real(0) * real(0) < real(1)) break;
        }
    } else {
        uint64_t ux, uy;
        for (;;) {
            ux = rng_next(s); uy = rng_next(s);
            const float fx = screen_coord(ux), fy = screen_coord(uy);
            const float lf = fx * fx + fy * fy;
            if (lf > 1.0f + 1e-5f) continue;
            if (lf < 1.0f - 1e-5f) break;
            const double x = -1.0 + 2.0 * u01(ux, 0.0), y = -1.0 + 2.0 * u01(uy, 0.0);
            if (x * x + y * y + 0.0 * 0.0 < 1.0) break;
        }
        px = -1.0 + 2.0 * u01(ux, 0.0); py = -1.0 + 2.0 * u01(uy, 0.0);
    }
}

// Camera::cast_ray's per-sample ray (ray_casting.rs:82-105): seeds the sample's RNG stream and draws
// time, pixel offset and (with defocus) the lens point, in the reference's order.
// ANIM: the kernels for keyed primitives, which also follow a keyed camera (cam.animated, per launch).  CAMK: the
// static-primitive kernels with the camera keys compiled in -- a movie that only moves the camera keeps the static walk
// (the teapot orbit frame: +6.5 % in f64 over running it on the ANIM kernels; folding the camera code into the plain
// static kernels instead cost book1 0.9 % f64 / 1.8 % f32, so it is a variant of its own).
template <typename real, bool ANIM, bool CAMK = false>
CR_D void camera_ray(const KernelArgs<real>& A, uint32_t pix_i, uint32_t pix_j, int32_t sample, uint64_t& rng, V3<real>& ro, V3<real>& rd,
                     real& rtime) {
    const CamConst<real>& cam = A.cam;
    uint32_t pixel = pix_j * (uint32_t)cam.W + pix_i;
    rng = rng_key(A.seed_mixed, pixel, (uint32_t)sample);
    real ts = A.current_time + rng_range<real>(rng, real(0), A.shutter_length);
    real ox = rng_uniform<real>(rng) - real(0.5);   // sample_square, camera/mod.rs:368-376
    real oy = rng_uniform<real>(rng) - real(0.5);
    CamFrame<real> f;
    if ((ANIM || CAMK) && cam.animated) {
        real fx = cam.from.x, fy = cam.from.y, fz = cam.from.z, fw = real(1);
        real ax = cam.at.x, ay = cam.at.y, az = cam.at.z, aw = real(1);
        timeline_eval(A.cam_keys + cam.from_key_first, cam.from_key_count, ts, fx, fy, fz, fw);
        timeline_eval(A.cam_keys + cam.at_key_first, cam.at_key_count, ts, ax, ay, az, aw);
        f = camera_frame(cam, mk<real>(fw * fx, fw * fy, fw * fz), mk<real>(aw * ax, aw * ay, aw * az));
    } else {
        f.from = cam.from; f.p00 = cam.p00; f.pdu = cam.pdu; f.pdv = cam.pdv; f.ddu = cam.ddu; f.ddv = cam.ddv;
    }
    V3<real> ps = add(add(f.p00, scale((real)pix_i + ox, f.pdu)), scale((real)pix_j + oy, f.pdv));   // get_pixel_pos :64-68
    V3<real> orig = f.from;
    if (cam.defocus_on) {   // defocus_disk_sample :104-110, random_in_unit_disk utils.rs:110-124
        real px, py;
        random_in_unit_disk_dev<real>(rng, px, py);
        orig = add(add(f.from, scale(px, f.ddu)), scale(py, f.ddv));
    }
    ro = orig; rd = sub(ps, orig); rtime = ts;
}

// A record of a table that sits either in LDS or in global memory (RES_TOP's materials and textures, decided per
// launch): loaded word by word through an address-space-qualified pointer under a wave-uniform branch, so that the
// compiler emits ds_read / global_load rather than a flat_load through the generic pointer.
template <typename T> CR_D T load_rec(const T* p, bool in_lds) {
    static_assert(sizeof(T) % 4 == 0, "records are whole words");
    T out;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
    if (in_lds) {
        const __attribute__((address_space(3))) uint32_t* s = (const __attribute__((address_space(3))) uint32_t*)(const void*)p;
        for (size_t k = 0; k < sizeof(T) / 4; k++) o[k] = s[k];
    } else {
        const __attribute__((address_space(1))) uint32_t* s = (const __attribute__((address_space(1))) uint32_t*)(const void*)p;
        for (size_t k = 0; k < sizeof(T) / 4; k++) o[k] = s[k];
    }
    return out;
}

// What ray_color does after the closest-hit query (ray_casting.rs:122-151) for a path whose hit is
// (best_t, best) -- best < 0 is a miss.  Returns true when the path is finished (col = the colour the
// outermost ray_color call returns), false when it scattered (ro/rd replaced, depth_left decremented).
// The path's non-unit attenuations live at att_stack[(level * stack_stride + stack_slot) * 3 .. +2]: one record per
// level, so a push is one contiguous store and an unwind step one contiguous load.
// RELAX (CR_SUM_RELAXED): no stack -- *thr carries the product of the attenuations met so far (a_1 * a_2 * ... in path
// order; the same multiplies as the reference's a_1 * (a_2 * (...)), associated the other way) and the colour handed
// back is thr * sky.  Every factor is in [0, 1] or NaN, so Color's clamp (utils.rs:553-563) never changes a product.
template <typename real, bool ANIM, bool SIDE_SPLIT = false, bool RELAX = false>
CR_D bool shade(const KernelArgs<real>& A, const Prim<real>* prims, const Mat<real>* mats, const Tex<real>* texs, V3<real>& ro, V3<real>& rd,
                real rtime, uint64_t& rng, int32_t& depth_left, int32_t& stack_n, real best_t, int32_t best, uint32_t stack_stride,
                uint32_t stack_slot, uint32_t& c_tex, V3<real>& col, Diag* dg = nullptr, V3<real>* thr = nullptr) {
    CR_DIAG_HIT(dg, DG_SHADE_WAVE, DG_SHADE_LANE);
    if (best >= 0) {
        CR_DIAG_HIT(dg, DG_HITSH_WAVE, DG_HITSH_LANE);
        const Prim<real>& p = prims[best];
        V3<real> loc = add(ro, scale(best_t, rd));   // Ray::at
        V3<real> n;
        real tu = 0, tv = 0;
        // SIDE_SPLIT (RES_TOP): the two tables are in LDS or in global memory, per launch
        auto mat_at = [&](int32_t i) { return SIDE_SPLIT ? load_rec(mats + i, A.lds_side != 0) : mats[i]; };
        auto tex_at = [&](int32_t i) { return SIDE_SPLIT ? load_rec(texs + i, A.lds_side != 0) : texs[i]; };
        const Mat<real> m = mat_at(p.mat());
        bool need_uv = false;
        int32_t leaf_tex = -1;
        if (m.kind == 0 && m.tex >= 0) {   // only image textures read u,v
            int ti = m.tex;
            Tex<real> tx = tex_at(ti);
            for (int guard = 0; guard < 32 && tx.kind == 1; guard++) {   // checker_texture.rs:38-51; <= CR_MAX_CHECKER_DEPTH levels by upload
                int32_t s = (int32_t)((uint32_t)as_i32(r_floor(tx.inv_scale * loc.x)) + (uint32_t)as_i32(r_floor(tx.inv_scale * loc.y)) +
                                      (uint32_t)as_i32(r_floor(tx.inv_scale * loc.z)));
                ti = (s % 2 == 0) ? tx.even : tx.odd;
                tx = tex_at(ti);
            }
            need_uv = tx.kind == 2;
            leaf_tex = ti;
        }
        if (p.kind() == 0) {
            real g0 = p.g[0], g1 = p.g[1], g2 = p.g[2], g3 = p.g[3];
            if (ANIM && p.key_count) {
                timeline_eval(A.keys + p.key_first, p.key_count, rtime, g0, g1, g2, g3);
                n = divs(sub(loc, mk<real>(g0, g1, g2)), g3);   // sphere.rs:97
            } else n = scale(p.g[4], sub(loc, mk<real>(g0, g1, g2)));   // p.g[4] = 1/radius
            if (need_uv) {                                  // get_sphere_uv, sphere.rs:41-46
                real theta = r_acos(-n.y);
                real phi = r_atan2(-n.z, n.x) + RealTraits<real>::pi;
                tu = phi / (real(2) * RealTraits<real>::pi);
                tv = theta / RealTraits<real>::pi;
            }
        } else {
            V3<real> a = mk<real>(p.g[0], p.g[1], p.g[2]), b = mk<real>(p.g[3], p.g[4], p.g[5]), c = mk<real>(p.g[6], p.g[7], p.g[8]);
            if (ANIM && p.key_count) {
                a = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, a);
                b = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, b);
                c = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, c);
            }
            n = unit(cross(sub(b, a), sub(c, a)));          // safe_new, objects/mod.rs:76
            tu = 0; tv = 0;                                 // triangle.rs:130-131
        }
        bool front = dot(rd, n) < real(0);                  // HitRecord::new
        if (!front) n = neg(n);

        V3<real> att = mk<real>(0, 0, 0), ndir = rd;
        bool some;
        // Lambertian and Metal both begin their draws with random_unit_vector() (lambertian.rs:41,
        // metal.rs:31): one shared pass of the rejection loop serves both groups of lanes
        V3<real> ruv = mk<real>(0, 0, 0);
#ifdef CR_DIAG
        if (m.kind != 2) {   // random_unit_vector with its rounds counted
            for (;;) {
                CR_DIAG_HIT(dg, DG_RUV_WAVE, DG_RUV_LANE);
                real x = rng_range<real>(rng, real(-1), real(1)), y = rng_range<real>(rng, real(-1), real(1)), z = rng_range<real>(rng, real(-1), real(1));
                V3<real> pp = mk<real>(x, y, z);
                real lensq = len2(pp);
                if (RealTraits<real>::tiny < lensq && lensq <= real(1)) { ruv = divs(pp, r_sqrt(lensq)); break; }
            }
        }
        CR_DIAG_LANE(dg, m.kind == 0 ? DG_LAMB_LANE : (m.kind == 1 ? DG_METAL_LANE : DG_DIEL_LANE));
#else
        if (m.kind != 2) ruv = random_unit_vector_dev<real>(rng);
#endif
        if (m.kind == 0) {                                  // lambertian.rs:40-61
            V3<real> dir = add(n, ruv);
            real tol = real(1e-8);
            if (r_abs(dir.x) < tol && r_abs(dir.y) < tol && r_abs(dir.z) < tol) dir = n;
            V3<real> tc;
            if (m.tex < 0) tc = mk<real>(m.albedo[0], m.albedo[1], m.albedo[2]);
            else {
                const Tex<real> lt = tex_at(leaf_tex);
                if (need_uv) tc = image_lookup(A.images, A.texels, lt.image, tu, tv, c_tex);
                else tc = mk<real>(lt.color[0], lt.color[1], lt.color[2]);
            }
            att = c_scale(m.aux, (m.param < real(0)) ? c_neg(tc) : tc);   // tc / scatter_prob
            ndir = dir;
            some = rng_uniform<real>(rng) <= m.param;
        } else if (m.kind == 1) {                           // metal.rs:29-42
            V3<real> refl = reflect(rd, n);
            refl = add(unit(refl), scale(m.param, ruv));
            att = mk<real>(m.albedo[0], m.albedo[1], m.albedo[2]);
            ndir = refl;
            some = dot(refl, n) > real(0);
        } else {                                            // dielectric.rs:30-55
            att = mk<real>(1, 1, 1);
            real ri = front ? m.albedo[0] : m.param;
            V3<real> ud = unit(rd);
            real cos_theta = -(r_fmin(dot(ud, n), real(1)));
            real sin_theta = r_sqrt(real(1) - cos_theta * cos_theta);
            bool refl = ri * sin_theta > real(1);
            if (!refl) {
                real r0 = front ? m.albedo[1] : m.albedo[2];
                real x = real(1) - cos_theta;
                real x2 = x * x;
                real x5 = x * (x2 * x2);
                refl = (r0 + (real(1) - r0) * x5) > rng_uniform<real>(rng);
            }
            ndir = refl ? reflect(ud, n) : refract(ud, n, ri);
            some = true;
        }
        if (!some) { col = mk<real>(0, 0, 0); return true; }   // scatter None -> black
        // attenuation * ray_color(scattered): the product is formed innermost-first, so remember the
        // factor and multiply on the way back (ray_casting.rs:128).  (1,1,1) multiplies exactly and
        // need not be stored.
        if (m.kind != 2) {
            if constexpr (RELAX) { thr->x = thr->x * att.x; thr->y = thr->y * att.y; thr->z = thr->z * att.z; }
            else {
                real* rec = A.att_stack + ((size_t)stack_n * stack_stride + stack_slot) * 3;
                rec[0] = att.x; rec[1] = att.y; rec[2] = att.z;
                stack_n++;
            }
        }
        ro = loc; rd = ndir; depth_left--;
        return false;
    }
    // sky (ray_casting.rs:133-151)
    CR_DIAG_LANE(dg, DG_SKY_LANE);
    V3<real> ud = unit(rd);
    if (A.sky_kind == 1) {
        real theta = r_atan2(ud.x, ud.z);
        real phi = r_asin(ud.y);
        real u = (theta / (real(2) * RealTraits<real>::pi)) + real(0.5);
        real v = (phi / RealTraits<real>::pi) + real(0.5);
        col = image_lookup(A.images, A.texels, A.sky_image, u, v, c_tex);
    } else {
        real a = real(0.5) * (ud.y + real(1));
        col = c_add(c_scale(real(1) - a, mk<real>(1, 1, 1)), c_scale(a, mk<real>(real(0.5), real(0.7), real(1))));
    }
    if constexpr (RELAX) { col = mk<real>(thr->x * col.x, thr->y * col.y, thr->z * col.z); return true; }
    // unwind: a_1 * (a_2 * ( ... (a_n * sky))).  The factors live in global memory; four levels are fetched
    // per round trip and applied innermost-first, so the product is formed in the reference's order.
    int32_t k = stack_n - 1;
    for (; k >= 3; k -= 4) {
        CR_DIAG_HIT(dg, DG_UNWIND_WAVE, DG_UNWIND_LANE);
        V3<real> a[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const real* rec = A.att_stack + ((size_t)(k - j) * stack_stride + stack_slot) * 3;
            a[j] = mk<real>(rec[0], rec[1], rec[2]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) col = c_mul(a[j], col);
    }
    for (; k >= 0; k--) {
        CR_DIAG_HIT(dg, DG_UNWIND_WAVE, DG_UNWIND_LANE);
        const real* rec = A.att_stack + ((size_t)k * stack_stride + stack_slot) * 3;
        V3<real> a_k = mk<real>(rec[0], rec[1], rec[2]);
        col = c_mul(a_k, col);
    }
    return true;
}

// Where the scene is read from: RES_GLOBAL everything through L2; RES_LDS entries | primitives |
// materials | textures staged in LDS; RES_TOP only the first lds_entries wrappers (top levels) in LDS.
enum : int { RES_GLOBAL = 0, RES_LDS = 1, RES_TOP = 2 };

// RES_TOP reads a wrapper from the LDS window or from global memory.  A select between the two pointers compiles to
// one flat_load, which is unordered against both counters (every use waits for vmcnt(0) and lgkmcnt(0)) and pays the
// aperture check; loads through address-space-qualified pointers in the two arms of a branch compile to ds_read and
// global_load.  Measured on one box, f64: the 1M-sphere frame +8 %, the movie frame +2 %, the teapot +1 %.
template <typename T> using LdsPtr = const __attribute__((address_space(3))) T*;
template <typename T> using GlobPtr = const __attribute__((address_space(1))) T*;
template <typename real, int RES>
CR_D Entry<real> fetch_entry(const Entry<real>* lds, const Entry<real>* glob, int32_t lds_entries, int32_t idx) {
    if (RES == RES_LDS) return lds[idx];
    if (RES == RES_TOP) {
        Entry<real> e;
        if (idx < lds_entries) {
            const Entry<real>* s = lds + idx;
            LdsPtr<real> b = (LdsPtr<real>)s->b;
            for (int k = 0; k < 6; k++) e.b[k] = b[k];
            e.skip = *(LdsPtr<int32_t>)&s->skip; e.leaf = *(LdsPtr<int32_t>)&s->leaf;
        } else {
            const Entry<real>* s = glob + idx;
            GlobPtr<real> b = (GlobPtr<real>)s->b;
            for (int k = 0; k < 6; k++) e.b[k] = b[k];
            e.skip = *(GlobPtr<int32_t>)&s->skip; e.leaf = *(GlobPtr<int32_t>)&s->leaf;
        }
        return e;
    }
    return glob[idx];
}
// The ordered layout read into the same record: skip = the link of the ray's octant.
template <typename real, int RES>
CR_D Entry<real> fetch_entry_ordered(const Entry<real>* lds, const Entry<real>* glob, int32_t lds_entries, int32_t idx, int32_t oct) {
    auto rd = [&](const EntryO<real>* p) {
        const EntryO<real>& s = p[idx];
        Entry<real> e;
        for (int k = 0; k < 6; k++) e.b[k] = s.b[k];
        e.leaf = s.leaf;
        e.skip = s.skip[oct];
        return e;
    };
    if (RES == RES_LDS) return rd((const EntryO<real>*)lds);
    if (RES == RES_TOP) {
        Entry<real> e;
        if (idx < lds_entries) {
            const EntryO<real>* s = (const EntryO<real>*)lds + idx;
            LdsPtr<real> b = (LdsPtr<real>)s->b;
            for (int k = 0; k < 6; k++) e.b[k] = b[k];
            e.leaf = *(LdsPtr<int32_t>)&s->leaf; e.skip = ((LdsPtr<int32_t>)s->skip)[oct];
        } else {
            const EntryO<real>* s = (const EntryO<real>*)glob + idx;
            GlobPtr<real> b = (GlobPtr<real>)s->b;
            for (int k = 0; k < 6; k++) e.b[k] = b[k];
            e.leaf = *(GlobPtr<int32_t>)&s->leaf; e.skip = ((GlobPtr<int32_t>)s->skip)[oct];
        }
        return e;
    }
    return rd((const EntryO<real>*)glob);
}

// A kernel that holds ALL screening records in LDS (RES_LDS) rewrites the links of its copy into LDS addresses while staging
// it, so that the walk's position is the address of the next ds_read as it stands: the address the copy starts at.
template <int RES> CR_D uint32_t screen_lds_base(const void* lds) { return RES == RES_LDS ? (uint32_t)(uintptr_t)(LdsPtr<char>)lds : 0u; }
// A screening record by its byte offset, from the LDS copy or from global memory (RES_TOP: the LDS window holds the first
// lds_bytes of the array).
template <int RES>
CR_D ScreenEntry fetch_screen(const ScreenEntry* lds, const ScreenEntry* glob, uint32_t lds_bytes, uint32_t off) {
    static_assert(sizeof(ScreenEntry) == 32, "offsets are index << 5");
    typedef uint32_t Quad __attribute__((ext_vector_type(4)));   // 16-byte aligned: two ds_read_b128 / global_load_dwordx4
    if (RES == RES_LDS) {   // the staged copy holds LDS addresses (screen_lds_base): `off` is one
        ScreenEntry e;
        Quad* o = reinterpret_cast<Quad*>(&e);
        LdsPtr<Quad> s = (LdsPtr<Quad>)(uintptr_t)off;
        o[0] = s[0]; o[1] = s[1];
        return e;
    }
    if (RES == RES_TOP) {
        ScreenEntry e;
        Quad* o = reinterpret_cast<Quad*>(&e);
        if (off < lds_bytes) {
            LdsPtr<Quad> s = (LdsPtr<Quad>)(const void*)((const char*)lds + off);
            o[0] = s[0]; o[1] = s[1];
        } else {
            GlobPtr<Quad> s = (GlobPtr<Quad>)(const void*)((const char*)glob + off);
            o[0] = s[0]; o[1] = s[1];
        }
        return e;
    }
    return *(const ScreenEntry*)((const char*)glob + off);
}

// The ordered layout's screening record by its byte offset: the box, {axis, hit} and the skip link of the ray's octant
// (octoff = 32 + 4 * octant, the byte offset of skip[octant] in the record).
struct ScreenStepO { float b[6]; uint32_t axis, hit, skip; };
template <int RES>
CR_D ScreenStepO fetch_screen_ordered(const ScreenEntryO* lds, const ScreenEntryO* glob, uint32_t lds_bytes, uint32_t off, uint32_t octoff) {
    static_assert(sizeof(ScreenEntryO) == 64, "offsets are index << 6");
    typedef uint32_t Quad __attribute__((ext_vector_type(4)));
    ScreenStepO e;
    Quad q0, q1;
    if (RES == RES_LDS) {   // the staged copy holds LDS addresses: `off` is one
        LdsPtr<Quad> s = (LdsPtr<Quad>)(uintptr_t)off;
        q0 = s[0]; q1 = s[1];
        e.skip = *(LdsPtr<uint32_t>)(uintptr_t)(off + octoff);
    } else if (RES == RES_TOP && off < lds_bytes) {
        LdsPtr<Quad> s = (LdsPtr<Quad>)(const void*)((const char*)lds + off);
        q0 = s[0]; q1 = s[1];
        e.skip = *(LdsPtr<uint32_t>)(const void*)((const char*)lds + off + octoff);
    } else {
        GlobPtr<Quad> s = (GlobPtr<Quad>)(const void*)((const char*)glob + off);
        q0 = s[0]; q1 = s[1];
        e.skip = *(GlobPtr<uint32_t>)(const void*)((const char*)glob + off + octoff);
    }
    uint32_t w[8];
    __builtin_memcpy(w, &q0, 16); __builtin_memcpy(w + 4, &q1, 16);
    for (int k = 0; k < 6; k++) e.b[k] = __builtin_bit_cast(float, w[k]);
    e.axis = w[6]; e.hit = w[7];
    return e;
}

// The per-ray state of BVHWrapper::hit's walk, kept in registers so a walk can be suspended and resumed.
template <typename real> struct WalkState {
    V3<real> inv;        // 1 / direction
    real dd;             // |direction|^2: Sphere::hit's `a`, the same for every sphere of the segment
    real best_t;         // closest hit so far = the interval's max handed to the next wrapper
    int32_t best;        // leaf-order index of that primitive, -1 = none
    int32_t idx;         // next wrapper to visit; n_entries = walk finished
    bool exact_box;      // an infinite 1/dir component: Aabb::hit's compare/select form is required
    int32_t oct;         // ordered walk: bit a set when direction[a] < 0
    int32_t pending;     // a leaf wrapper whose box was hit and whose primitives are still to be tested (-1: none)
};

template <typename real> CR_D void walk_begin(WalkState<real>& w, V3<real> rd) {
    w.inv = mk<real>(real(1) / rd.x, real(1) / rd.y, real(1) / rd.z);
    // 1/dir infinite on some axis (zero or denormal component): slab distances can be NaN, where only the
    // compare/select form reproduces Aabb::hit
    w.exact_box = (r_abs(w.inv.x) == r_inf(real(0))) || (r_abs(w.inv.y) == r_inf(real(0))) || (r_abs(w.inv.z) == r_inf(real(0)));
    w.dd = len2(rd);
    w.idx = 0; w.best_t = r_inf(real(0)); w.best = -1; w.pending = -1;
    w.oct = (rd.x < real(0) ? 1 : 0) | (rd.y < real(0) ? 2 : 0) | (rd.z < real(0) ? 4 : 0);
}

// One round of the while-while walk for the lanes with `walking` set: step through wrappers in the reference's
// order (left child on a box hit, skip link on a miss) until the lane reaches a leaf wrapper, runs out of wrappers
// or has made `budget` steps (0 = unbounded); then the lanes parked on a leaf intersect its primitives together.
// Per lane this is exactly BVHWrapper::hit's sequence (bvhwrapper.rs:96-126); the round structure only decides
// when lanes wait for each other.
// ORD: the ordered layout (EntryO behind the same pointers) -- near child first, per-octant skip links.
// SCREEN kernels walk on the 32-byte ScreenEntry (64-byte ScreenEntryO) records: byte-offset links in the form the loop consumes.
// f32 kernels (unordered trees): the record holds the wrapper's own box and the test on it is Aabb::hit (EXACT below).
// f64 kernels: Aabb::hit decided on the f32 screening record wherever f32 can decide it:
// Notation: b, o, inv = an f64 box plane, the origin component and 1/direction on that axis; bf, of, if their f32
// roundings; u = 2^-24; T = (b - o) * inv; t32 = fl(fl(bf - of) * if) the f32 slab distance.  Then
//     |t32 - T| <= 1.01 u |inv| (|b| + |o|) + 3.01 u |t32|          (three roundings of inputs, two of operations)
// and, because |inv| |b| <= |T| + |inv| |o|,
//     |t32 - T| <= 4.03 u |t32| + 2.03 u |inv| |o|                   (the f64 test's own roundings, 2^-52 |T|, vanish in the slack).
// lo32 = max(nears, 0.001) and hi32 = min(fars, tmax32): an operand can decide the f64 result only if its own value lies
// within the two errors of the f32 winner, so the end's error is bounded by the same expression in |lo32| resp. |hi32|
// (to first order in u); an interval end that wins was rounded once (u |end|); the subtraction rounds once more.  With
// M = max(|lo32|, |hi32|) and Q = max over the axes of |if of|,
//     TH = 2^-20 M + 2^-21 Q + 2^-147 max |if| + 1e-35
// is at least 1.5 times the largest possible |(hi32 - lo32) - (hi - lo)| (10.1 u M + 4.06 u Q; the last two terms cover
// a box plane or origin component below the normal f32 range and a product that underflows).  Hence hi32 - lo32 < -TH proves Aabb::hit's
// `max <= min` (a miss), hi32 - lo32 > TH proves a hit, and only a lane with |hi32 - lo32| <= TH evaluates Aabb::hit in f64
// on the f64 box.  Overflow and NaN land there too (every comparison with them is false), and the round uses the screen only
// when |if| lies in [2^-100, 2^100] and |of| <= 2^100 on every axis.  The decisions -- hence the walk, the counters and the image -- are those of the f64 test.
// SCREEN is a kernel variant of its own: a kernel that carried both the f32 loop and the f64 min/max loop lost 5 % on the
// teapot frames to register pressure.  In a SCREEN kernel a ray whose
// 1/direction is infinite, or outside the f32 range above, walks with Aabb::hit's compare/select form (valid for every ray).
template <typename real, int RES, bool ANIM, bool ORD = false, bool SCREEN = false>
CR_D void walk_round(const KernelArgs<real>& A, const Entry<real>* lds_entries, const Prim<real>* prims, V3<real> ro, V3<real> rd, real rtime,
                     WalkState<real>& w, bool walking, uint32_t budget, unsigned long long& c_node, uint32_t& c_prim, Diag* dg = nullptr,
                     const void* lds_screen = nullptr) {
    // f64: decisions on the f32 copy where f32 can decide; f32 (EXACT below): the record IS the wrapper's box, in the link layout of ScreenEntry
    constexpr bool EXACT = std::is_same<real, float>::value;
    static_assert(!(SCREEN && EXACT && ORD), "the f32 kernels use ScreenEntry records for unordered trees only");
    const real tmin = real(0.001);
    const int32_t n_entries = A.n_entries;
    // SCREEN kernels keep only screening records in LDS: the rare f64 record is read from global memory
    constexpr int RES64 = SCREEN ? RES_GLOBAL : RES;
    const int32_t lds_n64 = SCREEN ? 0 : A.lds_entries;
    int32_t leaf = w.pending;
    if (walking && leaf < 0) {
        CR_DIAG_HIT(dg, DG_ROUND_WAVE, DG_ROUND_LANE);
        // SCREEN kernels: steps the f32 loop could not take -- one, for a lane it left at a box too close to call, or all of
        // them for a ray outside its range -- are made below with Aabb::hit's own compare/select form on the f64 records
        uint32_t exact_steps = w.exact_box ? 0xffffffffu : 0u;
        if constexpr (SCREEN) {
            const float ofx = (float)ro.x, ofy = (float)ro.y, ofz = (float)ro.z;
            const float ifx = (float)w.inv.x, ify = (float)w.inv.y, ifz = (float)w.inv.z;
            const float mo = r_max(r_max(__builtin_fabsf(ofx), __builtin_fabsf(ofy)), __builtin_fabsf(ofz));
            const float pmax = r_max(r_max(__builtin_fabsf(ifx), __builtin_fabsf(ify)), __builtin_fabsf(ifz));
            const float pmin = r_min(r_min(__builtin_fabsf(ifx), __builtin_fabsf(ify)), __builtin_fabsf(ifz));
            // (an f32 kernel's test on the record is Aabb::hit itself for every ray with finite 1/direction: no range to respect)
            bool screened;
            if constexpr (EXACT) screened = !w.exact_box;
            else screened = !w.exact_box && pmin >= 0x1.0p-100f && pmax <= 0x1.0p100f && mo <= 0x1.0p100f;
            if (!screened) exact_steps = 0xffffffffu;
            else {
                const float qx = __builtin_fabsf(ofx * ifx), qy = __builtin_fabsf(ofy * ify), qz = __builtin_fabsf(ofz * ifz);
                const Pair<float> fox = {ofx, ofx}, foy = {ofy, ofy}, foz = {ofz, ofz};
                const Pair<float> fix = {ifx, ifx}, fiy = {ify, ify}, fiz = {ifz, ifz};
                const float tminf = 0.001f;
                float tmaxf = (float)w.best_t;   // tmax cannot change inside the loop
                // the part of TH that does not depend on the box: 2^-21 Q, plus what rounding a box plane or an origin component
                // BELOW the normal f32 range can add (2^-149 each, times |if| <= 2^100), plus a product that underflows
                const float th0 = __builtin_fmaf(0x1.0p-21f, r_max(r_max(qx, qy), qz), __builtin_fmaf(pmax * 0x1.0p-100f, 0x1.0p-47f, 1e-35f));
                asm volatile("" : "+v"(tmaxf));   // keep it in a register: the allocator would re-convert tmax at every step
                uint32_t nodes = 0;
                // Aabb::hit of the wrapper at index `idx` decided on its screening box `b`: true = miss
                auto box_miss = [&](const float* b, int32_t idx) -> bool {
                    const Pair<float> tx = (Pair<float>{b[0], b[1]} - fox) * fix;
                    const Pair<float> ty = (Pair<float>{b[2], b[3]} - foy) * fiy;
                    const Pair<float> tz = (Pair<float>{b[4], b[5]} - foz) * fiz;
                    const float nx = r_min(tx.x, tx.y), ny = r_min(ty.x, ty.y), nz = r_min(tz.x, tz.y);
                    const float fx = r_max(tx.x, tx.y), fy = r_max(ty.x, ty.y), fz = r_max(tz.x, tz.y);
                    const float lo = r_max(r_max(nx, ny), r_max(nz, tminf));
                    // hi = r_min(r_min(fx, fy), r_min(fz, tmaxf)) and m = r_max(|lo|, |hi|), written out: the compiler re-quiets the
                    // loop-invariant tmaxf with a v_max_f32 x, x at every step (instruction selection works block by block and
                    // cannot see that it is a number), and one VALU instruction in this loop is about 1.5 % of the frame
                    float hi, m;
                    if constexpr (EXACT) {
                        nodes++;
                        // f32 kernels: `max <= min -> miss` on the same operations as box_miss_fast (min(h, tmax) <= lo there is
                        // h <= lo || tmax <= lo)
                        asm("v_min_f32_e32 %0, %1, %2\n\tv_min3_f32 %0, %3, %4, %0" : "=&v"(hi) : "v"(fz), "v"(tmaxf), "v"(fx), "v"(fy));
                        CR_DIAG_HIT(dg, DG_BOX_WAVE, DG_BOX_LANE);
                        return hi <= lo;
                    }
                    asm("v_min_f32_e32 %0, %2, %3\n\tv_min3_f32 %0, %4, %5, %0\n\tv_max_f32_e64 %1, |%6|, |%0|"
                        : "=&v"(hi), "=v"(m) : "v"(fz), "v"(tmaxf), "v"(fx), "v"(fy), "v"(lo));
                    const float th = __builtin_fmaf(0x1.0p-20f, m, th0);
                    float d = hi - lo;
                    nodes++;
                    CR_DIAG_HIT(dg, DG_BOX_WAVE, DG_BOX_LANE);
                    // too close to call in f32: Aabb::hit in f64 on the f64 box.  (The wave tests "any lane?" with a scalar branch and
                    // the rare lane overwrites d, so that the common path carries no mask bookkeeping for the merge.)
                    const bool band = !(__builtin_fabsf(d) > th);
                    if (__builtin_expect(__ballot(band) != 0ull, 0)) {
                        if (band) {
                            CR_DIAG_HIT(dg, DG_BAND_WAVE, DG_BAND_LANE);
                            const Entry<real> e = ORD ? fetch_entry_ordered<real, RES64>(lds_entries, A.entries, lds_n64, idx, w.oct)
                                                      : fetch_entry<real, RES64>(lds_entries, A.entries, lds_n64, idx);
                            d = box_hit(e.b, ro, w.inv, tmin, w.best_t) ? 1.0f : -1.0f;
                        }
                    }
                    return d < 0.0f;
                };
                // `it` is the same in every lane still in the loop (a scalar register)
                if constexpr (ORD) {
                    // near child first: the same loop on ScreenEntryO records -- the skip link is the octant's, the hit link the left
                    // child's plus one record when the ray points down the split axis
                    const uint32_t base = screen_lds_base<RES>(lds_screen);
                    const uint32_t end = base + ((uint32_t)n_entries << 6);
                    const uint32_t oct = (uint32_t)w.oct, octoff = 32u + (oct << 2);
                    uint32_t off = base + ((uint32_t)w.idx << 6), after_leaf = 0;
                    if (off < end) {
                        for (uint32_t it = 0;; it++) {
                            const ScreenStepO se = fetch_screen_ordered<RES>((const ScreenEntryO*)lds_screen, (const ScreenEntryO*)A.screen, (uint32_t)A.lds_entries << 6, off, octoff);
                            const bool miss = box_miss(se.b, (int32_t)((off - base) >> 6));
                            after_leaf = se.skip;
                            off = miss ? se.skip : se.hit + (__builtin_amdgcn_ubfe(oct, se.axis, 1u) << 6);
                            uint32_t limit = (it + 1 == budget) ? 0u : end;   // (as below)
                            asm("" : "+s"(limit));
                            if (off >= limit) break;
                        }
                        if (off & kScreenLeaf) { leaf = (int32_t)(off & ~kScreenLeaf); off = after_leaf; }
                        w.idx = (int32_t)((off - base) >> 6);
                    }
                } else {
                    // the lane's position as a byte offset into the record array (ScreenEntry): a lane leaves the loop at the
                    // end of the array or with a leaf in hand, and one compare sees both
                    const uint32_t base = screen_lds_base<RES>(lds_screen);
                    const uint32_t end = base + ((uint32_t)n_entries << 5);
                    uint32_t off = base + ((uint32_t)w.idx << 5), after_leaf = 0;
                    if (off < end) {
                        for (uint32_t it = 0;; it++) {
                            const ScreenEntry se = fetch_screen<RES>((const ScreenEntry*)lds_screen, (const ScreenEntry*)A.screen, (uint32_t)A.lds_entries << 5, off);
                            const bool miss = box_miss(se.b, (int32_t)((off - base) >> 5));
                            after_leaf = se.skip;
                            off = miss ? se.skip : se.hit;
                            // the round's last step ends it for every lane: a scalar select of the limit (hidden from the optimiser,
                            // which would turn it back into a second exit mask)
                            uint32_t limit = (it + 1 == budget) ? 0u : end;
                            asm("" : "+s"(limit));
                            if (off >= limit) break;
                        }
                        if (off & kScreenLeaf) { leaf = (int32_t)(off & ~kScreenLeaf); off = after_leaf; }
                        w.idx = (int32_t)((off - base) >> 5);
                    }
                }
                c_node += nodes;
            }
        } else if (!w.exact_box) {
            const Pair<real> ox = {ro.x, ro.x}, oy = {ro.y, ro.y}, oz = {ro.z, ro.z};
            const Pair<real> ix = {w.inv.x, w.inv.x}, iy = {w.inv.y, w.inv.y}, iz = {w.inv.z, w.inv.z};
            // `it` is the same in every lane still in the loop (a scalar register); tmax cannot change inside it
            const real tmax = w.best_t;
            uint32_t nodes = 0;
            for (uint32_t it = 0; w.idx < n_entries; it++) {
                const Entry<real> e = ORD ? fetch_entry_ordered<real, RES64>(lds_entries, A.entries, lds_n64, w.idx, w.oct)
                                          : fetch_entry<real, RES64>(lds_entries, A.entries, lds_n64, w.idx);
                nodes++;
                CR_DIAG_HIT(dg, DG_BOX_WAVE, DG_BOX_LANE);
                const bool miss = box_miss_fast(e.b, ox, oy, oz, ix, iy, iz, tmin, tmax);
                const bool inner = e.leaf < 0;
                w.idx = (inner && !miss) ? (ORD ? ordered_near(e.leaf, w.oct) : -e.leaf) : e.skip;
                if (!(miss || inner)) { leaf = e.leaf; break; }
                if (it + 1 == budget) break;
            }
            c_node += nodes;
        }
        for (; exact_steps != 0u && w.idx < n_entries; exact_steps--) {
            const Entry<real> e = ORD ? fetch_entry_ordered<real, RES64>(lds_entries, A.entries, lds_n64, w.idx, w.oct)
                                      : fetch_entry<real, RES64>(lds_entries, A.entries, lds_n64, w.idx);
            c_node++;
            bool hit = box_hit(e.b, ro, w.inv, tmin, w.best_t);
            w.idx = (hit && e.leaf < 0) ? (ORD ? ordered_near(e.leaf, w.oct) : -e.leaf) : e.skip;
            if (hit && e.leaf >= 0) { leaf = e.leaf; break; }
        }
    }
    // The parked lanes intersect their leaves together; with walk_leaf_min set, a thin group waits (parked) while other lanes
    // of the wave can still step, so that the expensive primitive test runs with more lanes.  Per lane the sequence of
    // operations is unchanged: a parked lane does nothing until its leaf is tested.
    w.pending = leaf;
    if (A.walk_leaf_min > 0u) {
        const uint32_t parked = (uint32_t)__popcll(__ballot(leaf >= 0));
        const bool can_step = __ballot(walking && leaf < 0 && w.idx < n_entries) != 0ull;
        if (parked < A.walk_leaf_min && can_step) return;
    }
    w.pending = -1;
    if (leaf >= 0) {
        CR_DIAG_HIT(dg, DG_LEAFPH_WAVE, DG_LEAFPH_LANE);
        auto test = [&](int32_t pi) {
            const Prim<real>& p = prims[pi];
            c_prim++;
            CR_DIAG_HIT(dg, DG_PRIM_WAVE, DG_PRIM_LANE);
            real t;
            bool h;
            real g0 = p.g[0], g1 = p.g[1], g2 = p.g[2], g3 = p.g[3];
            const int32_t kind = A.uniform_kind >= 0 ? A.uniform_kind : p.kind();   // a scalar test; one load fewer per primitive when it holds
            if (kind == 0) {
                if (ANIM && p.key_count) timeline_eval(A.keys + p.key_first, p.key_count, rtime, g0, g1, g2, g3);
                h = sphere_t(g0, g1, g2, g3, ro, rd, w.dd, tmin, w.best_t, t);
            } else {
                V3<real> a = mk<real>(g0, g1, g2), b = mk<real>(g3, p.g[4], p.g[5]), c = mk<real>(p.g[6], p.g[7], p.g[8]);
                if (ANIM && p.key_count) {
                    a = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, a);
                    b = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, b);
                    c = timeline_vertex(A.keys + p.key_first, p.key_count, rtime, c);
                }
                h = triangle_t(a, b, c, ro, rd, tmin, w.best_t, t);
            }
            if (h) { w.best_t = t; w.best = pi; }
        };
        // Scenes with a HitList element run the ANIM kernels (capi.hip), so the static kernels -- the headline's -- keep
        // the one-or-two-record loop alone; A.leaf_runs is null unless such an element exists (a scalar test)
        if (!ANIM || A.leaf_runs == nullptr || !(leaf & kLeafRun)) {
            const int32_t first = leaf >> 1, count = (leaf & 1) + 1;
            if constexpr (RES != RES_LDS && !ANIM) {
                // primitives in global memory: touch the second record's lines while the first is being tested, so that its own
                // loads hit L1 instead of waiting a second L2 round trip (teapot +1.5 %, 1M spheres +1.4 %)
                GlobPtr<uint32_t> nx = (GlobPtr<uint32_t>)(const void*)(prims + first + (count - 1));
                const uint32_t t0 = nx[0], t1 = nx[20];
                test(first);
                asm volatile("" :: "v"(t0), "v"(t1));
                if (count == 2) test(first + 1);
            } else
            for (int32_t k = 0; k < count; k++) test(first + k);
        } else {
            const int32_t first = A.leaf_runs[2 * (leaf & kLeafRunIndex)], count = A.leaf_runs[2 * (leaf & kLeafRunIndex) + 1];
            for (int32_t k = 0; k < count; k++) test(first + k);
        }
    }
}

// Diagnostic build: per-wave phase clocks go to counters[9..12] (regeneration, walk, shade, total).
#ifdef CR_DIAG
#define CR_DIAG_ONLY(...) __VA_ARGS__
#else
#define CR_DIAG_ONLY(...)
#endif

// TRACE: has a ray whose walk has not begun; WALK: walking (state kept across rounds of the outer loop);
// SHADE: closest hit known.
enum : int { ST_NEED_PIXEL = 0, ST_NEED_SAMPLE = 1, ST_TRACE = 2, ST_DONE = 3, ST_WALK = 4, ST_SHADE = 5 };

// Largest workgroup the kernels may be launched with.  The launch bound caps the register allocation:
// 1024 threads = 4 waves/SIMD = 128 VGPRs.  The f32 kernels need ~103; the f64 kernels sit at the cap with 10-18 VGPRs
// spilled to scratch (profiles/r03_kernel_resources.json) -- in round 1 the f64 kernel at 4 waves/SIMD with its spills
// measured 7 % faster than at 2 waves/SIMD without, and 5 waves/SIMD (<= 96 VGPRs, 78 spills) loses 32 %.
template <typename real> struct MaxBlock { static constexpr int value = 1024; };

// The persistent kernel body.  A work item is one (pixel, sample) handed out in chunks from one global counter (sample-granular
// mode, the default); the reference-order variants keep per-sample colours for the ordered sum, the RELAX variants add into
// fixed-point per-pixel sums.  (sg_on = 0, a fallback: one lane owns a pixel and sums its samples in draw order itself.)
// RELAX: CR_SUM_RELAXED -- the same paths (same draws, same walks, same counters); a finished sample's colour is
// thr * sky and goes into fixed-point per-pixel sums.  Each wave owns two LDS accumulators, one work tile (<= 16
// pixels x 3 channels) each: ds_add_u64 there, and one global atomic per word when the wave moves on to another tile
// -- about 48 global atomics per 1024 samples instead of 3 per sample (global atomics execute at the memory side, one
// request per lane when the lanes' addresses are scattered).  A straggler whose tile has already been flushed adds to
// the global sums directly.
template <typename real, int RES, bool ANIM, bool ORD, bool CAMK = false, bool RELAX = false, bool SCREEN = false>
CR_D void pathtrace_body(const KernelArgs<real>& A) {
    using EntryT = typename EntryOf<real, ORD>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef CR_HOLD_VCC
    unsigned long long vcc_hold;
    asm volatile("s_mov_b64 %0, 0" : "={vcc}"(vcc_hold));
#endif
    const Entry<real>* lds_entries = nullptr;
    const void* lds_screen = nullptr;
    const Prim<real>* prims = A.prims;
    const Mat<real>* mats = A.mats;
    const Tex<real>* texs = A.texs;
    if (RES != RES_GLOBAL) {
        auto copy = [&](const void* src, size_t off, size_t bytes) {
            const uint32_t* s = (const uint32_t*)src;
            uint32_t* d = (uint32_t*)(smem + off);
            for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
        };
        // SCREEN kernels stage the f32 screening records instead of the f64 wrappers (twice as many wrappers in the same bytes)
        using ScreenT = typename ScreenOf<ORD>::type;
        constexpr size_t window_rec = SCREEN ? sizeof(ScreenT) : sizeof(EntryT);
        copy(SCREEN ? A.screen : (const void*)A.entries, 0, (size_t)A.lds_entries * window_rec);
        lds_entries = (const Entry<real>*)smem;
        if (SCREEN) lds_screen = (const void*)smem;
        if (RES == RES_LDS) {   // the whole scene: entries | prims | mats | texs, each 16-B aligned
            size_t o1 = (((size_t)A.n_entries * window_rec + 15) & ~(size_t)15);
            size_t o2 = o1 + (((size_t)A.n_prims * sizeof(Prim<real>) + 15) & ~(size_t)15);
            size_t o3 = o2 + (((size_t)A.n_mats * sizeof(Mat<real>) + 15) & ~(size_t)15);
            copy(A.prims, o1, (size_t)A.n_prims * sizeof(Prim<real>));
            copy(A.mats, o2, (size_t)A.n_mats * sizeof(Mat<real>));
            copy(A.texs, o3, (size_t)A.n_texs * sizeof(Tex<real>));
            prims = (const Prim<real>*)(smem + o1);
            mats = (const Mat<real>*)(smem + o2);
            texs = (const Tex<real>*)(smem + o3);
            if constexpr (SCREEN) {   // links of the staged screening records become LDS addresses (fetch_screen, fetch_screen_ordered)
                __syncthreads();
                const uint32_t base = screen_lds_base<RES>(smem);
                ScreenT* rec = (ScreenT*)smem;
                for (int32_t i = (int32_t)threadIdx.x; i < A.n_entries; i += (int32_t)blockDim.x) {
                    if constexpr (ORD) { for (int k = 0; k < 8; k++) rec[i].skip[k] += base; }
                    else rec[i].skip += base;
                    if (!(rec[i].hit & kScreenLeaf)) rec[i].hit += base;
                }
            }
        } else if (A.lds_side) {   // RES_TOP: entry window | mats | texs
            size_t o2 = (((size_t)A.lds_entries * window_rec + 15) & ~(size_t)15);
            size_t o3 = o2 + (((size_t)A.n_mats * sizeof(Mat<real>) + 15) & ~(size_t)15);
            copy(A.mats, o2, (size_t)A.n_mats * sizeof(Mat<real>));
            copy(A.texs, o3, (size_t)A.n_texs * sizeof(Tex<real>));
            mats = (const Mat<real>*)(smem + o2);
            texs = (const Tex<real>*)(smem + o3);
        }
        __syncthreads();
    }

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    // RELAX: this wave's two accumulator slots and the tiles they hold (wave-uniform; kFxNoTile = empty)
    constexpr uint32_t kFxNoTile = 0xffffffffu;
    unsigned long long* fx_slots = nullptr;
    uint32_t fx_tile0 = kFxNoTile, fx_tile1 = kFxNoTile, fx_mru = 0;
    V3<real> thr = mk<real>(1, 1, 1);
    const uint32_t fx_words = 3u << (A.sg_lw + A.sg_lh);   // words per slot
    if constexpr (RELAX) {
        fx_slots = (unsigned long long*)(smem + A.fx_lds_off) + (size_t)(threadIdx.x >> 6) * 2 * fx_words;
        for (uint32_t k = lane; k < 2 * fx_words; k += 64) fx_slots[k] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    // adds a slot's sums to the global ones and empties it (whole wave; `tile` is wave-uniform)
    auto fx_flush = [&](uint32_t slot, uint32_t tile) {
        if constexpr (RELAX) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (tile != kFxNoTile) {
                const uint32_t ti = (tile % A.tiles_x) << A.sg_lw, tj = (tile / A.tiles_x) << A.sg_lh;
                for (uint32_t k = lane; k < fx_words; k += 64) {   // 48 words for the usual 4 x 4 tile: one pass
                    unsigned long long* w = fx_slots + slot * fx_words + k;
                    const unsigned long long v = *w;
                    if (v) {
                        *w = 0ull;
                        const uint32_t px = k / 3u, ch = k - px * 3u;
                        const uint32_t pi = ti + (px & ((1u << A.sg_lw) - 1u)), pj = tj + (px >> A.sg_lw);
                        unsigned long long* g = A.fx_acc + ((size_t)pj * (size_t)A.cam.W + pi) * 3 + ch;
                        if (v & ~kFxNaN) atomicAdd(g, v & ~kFxNaN);
                        if (v & kFxNaN) atomicOr(g, kFxNaN);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    };
    const uint32_t total_work = (A.sg_on || RELAX) ? A.sg_total : A.tiles_x * A.tiles_y * 64u;
    const CamConst<real>& cam = A.cam;
    // sample-granular mode: the wave's private slice [wv_next, wv_end) of the work counter (same value in all lanes)
    uint32_t wv_next = 0, wv_end = 0;
    const uint32_t SG_CHUNK = A.sg_chunk;   // items a wave takes per atomic: 1024 for long launches, fewer for short ones (launch())

    int state = ST_NEED_PIXEL;
    uint32_t pix_i = 0, pix_j = 0;
    uint32_t item = 0;   // sample-granular mode: the lane's current work item
    int32_t sample = 0;
    real acc_r = 0, acc_g = 0, acc_b = 0;
    // path state
    V3<real> ro = mk<real>(0, 0, 0), rd = mk<real>(0, 0, 1);
    real rtime = 0;
    uint64_t rng = 0;
    int32_t depth_left = 0, stack_n = 0;
    uint32_t c_seg = 0, c_prim = 0, c_tex = 0;
    unsigned long long c_node = 0;
    // walk state, kept across rounds: a lane whose walk is cut short resumes where it stopped
    WalkState<real> ws;
    ws.inv = mk<real>(0, 0, 0); ws.dd = 0; ws.best_t = 0; ws.best = -1; ws.idx = 0; ws.exact_box = false; ws.pending = -1;
    const int32_t n_entries = A.n_entries;

    Diag* dgp = nullptr;
    CR_DIAG_ONLY(Diag dg_store; for (int i = 0; i < DG_N; i++) dg_store.v[i] = 0; dgp = &dg_store;
                 unsigned long long d_t_regen = 0, d_t_trace = 0, d_t_shade = 0;
                 unsigned long long d_t0 = __builtin_readcyclecounter(); const unsigned long long d_begin = d_t0;)
    for (;;) {
        CR_DIAG_ONLY(if (CR_DIAG_LEADER()) dg_store.v[DG_OUTER_WAVE]++; d_t0 = __builtin_readcyclecounter();)
        // ---------------- regeneration: pixels
        uint64_t need = __ballot(state == ST_NEED_PIXEL);
        if (need && (A.sg_on || RELAX)) {
            // One (pixel, sample) per lane.  The wave takes SG_CHUNK consecutive items from the global counter at a
            // time and hands them to its lanes in order, so a wave stays on one tile's samples (coherent rays) and
            // the counter sees one atomic per 1024 samples.
            const uint32_t cnt = (uint32_t)__popcll(need), avail = wv_end - wv_next;
            uint32_t fresh = 0;
            if (cnt > avail) {
                const int leader = __ffsll((unsigned long long)need) - 1;
                if ((int)lane == leader) fresh = atomicAdd(A.work_counter, SG_CHUNK);
                fresh = __shfl(fresh, leader);
            }
            if (state == ST_NEED_PIXEL) {
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                // past the end of the counter's range (also when it wrapped): nothing left
                const uint32_t w = rank < avail ? wv_next + rank : fresh + (rank - avail);
                if (w >= total_work) state = ST_DONE;
                else {
                    item = w;
                    const uint32_t group = w >> 6, in = w & 63u;
                    const uint32_t tile = group / A.sg_groups, sg = group - tile * A.sg_groups;
                    const uint32_t px = in & ((1u << A.sg_lw) - 1u), py = (in >> A.sg_lw) & ((1u << A.sg_lh) - 1u);
                    const uint32_t ds = in >> (A.sg_lw + A.sg_lh);
                    pix_i = ((tile % A.tiles_x) << A.sg_lw) + px;
                    pix_j = ((tile / A.tiles_x) << A.sg_lh) + py;
                    sample = A.sample_begin + (int32_t)(sg * (64u >> (A.sg_lw + A.sg_lh)) + ds);
                    if (pix_i < (uint32_t)cam.W && pix_j < (uint32_t)cam.H && sample < A.sample_end) state = ST_NEED_SAMPLE;
                    // else: padding of an edge tile or of the last sample group, ask again next round
                }
            }
            if (cnt > avail) { wv_next = fresh + (cnt - avail); wv_end = fresh + SG_CHUNK; }
            else wv_next += cnt;
            if constexpr (RELAX) {
                // make room for the tiles this round's new items belong to (consecutive items: one or two tiles)
                const uint32_t my_tile = (pix_j >> A.sg_lh) * A.tiles_x + (pix_i >> A.sg_lw);
                uint64_t fresh_items = __ballot(state == ST_NEED_SAMPLE) & need;
                while (fresh_items) {
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_tile, __ffsll((unsigned long long)fresh_items) - 1);
                    fresh_items &= ~__ballot(my_tile == t);
                    if (t == fx_tile0) fx_mru = 0;
                    else if (t == fx_tile1) fx_mru = 1;
                    else if (fx_mru == 0) { fx_flush(1, fx_tile1); fx_tile1 = t; fx_mru = 1; }
                    else { fx_flush(0, fx_tile0); fx_tile0 = t; fx_mru = 0; }
                }
            }
        } else if (need) {
            uint32_t cnt = (uint32_t)__popcll(need);
            uint32_t base = 0;
            int leader = __ffsll((unsigned long long)need) - 1;
            if ((int)lane == leader) base = atomicAdd(A.work_counter, cnt);
            base = __shfl(base, leader);
            if (state == ST_NEED_PIXEL) {
                uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                uint32_t w = base + rank;
                if (w >= total_work) state = ST_DONE;
                else {
                    uint32_t tile = w >> 6, in = w & 63u;
                    pix_i = (tile % A.tiles_x) * 8u + (in & 7u);
                    pix_j = (tile / A.tiles_x) * 8u + (in >> 3);
                    if (pix_i < (uint32_t)cam.W && pix_j < (uint32_t)cam.H) {
                        state = ST_NEED_SAMPLE; sample = A.sample_begin; acc_r = acc_g = acc_b = 0;
                    }   // else: padding of an edge tile, ask again next round
                }
            }
        }
        if (__ballot(state != ST_DONE) == 0) break;

        // ---------------- regeneration: camera rays (cast_ray, ray_casting.rs:82-105)
        if (state == ST_NEED_SAMPLE) {
            CR_DIAG_HIT(dgp, DG_REGEN_WAVE, DG_REGEN_LANE);
            camera_ray<real, ANIM, CAMK>(A, pix_i, pix_j, sample, rng, ro, rd, rtime);
            depth_left = A.max_depth; stack_n = 0;
            if constexpr (RELAX) thr = mk<real>(1, 1, 1);
            state = ST_TRACE;
        }

        CR_DIAG_ONLY({ unsigned long long t = __builtin_readcyclecounter(); d_t_regen += t - d_t0; d_t0 = t; })
        // ---------------- closest hit (Hittables::hit on the BVH root, interval (0.001, inf))
        V3<real> col = mk<real>(0, 0, 0);   // colour returned by the innermost ray_color call
        bool finished = false;
        if (state == ST_TRACE) {
            if (depth_left == 0) finished = true;   // ray_color: depth == 0 -> black
            else {
                c_seg++;
                walk_begin(ws, rd);
                state = n_entries > 0 ? ST_WALK : ST_SHADE;
            }
        }
        // Rounds of the while-while walk (walk_round).  After each round the wave may leave the walk if enough
        // lanes are done: the stragglers keep their WalkState and resume on the next round of the outer loop, so
        // each lane still performs BVHWrapper::hit's exact sequence; only the interleaving with other lanes'
        // shading changes.
        if (__ballot(state == ST_WALK)) {
            for (;;) {
                walk_round<real, RES, ANIM, ORD, SCREEN>(A, lds_entries, prims, ro, rd, rtime, ws, state == ST_WALK, A.walk_round_steps, c_node, c_prim, dgp, lds_screen);
                if (state == ST_WALK && ws.idx >= n_entries && ws.pending < 0) state = ST_SHADE;
                const uint64_t walking = __ballot(state == ST_WALK);
                if (!walking || 64u - (uint32_t)__popcll(walking) >= A.walk_exit_lanes) break;
            }
        }
        const bool tracing = (state == ST_SHADE);

        CR_DIAG_ONLY({ unsigned long long t = __builtin_readcyclecounter(); d_t_trace += t - d_t0; d_t0 = t; })
        // ---------------- shade
        if (tracing) {
            finished = shade<real, ANIM, RES == RES_TOP, RELAX>(A, prims, mats, texs, ro, rd, rtime, rng, depth_left, stack_n, ws.best_t, ws.best,
                                         A.n_threads, gtid, c_tex, col, dgp, &thr);
            state = ST_TRACE;   // scattered: a fresh ray to walk (overwritten below when the path finished)
        }

        // ---------------- sample / pixel completion (average_samples, ray_casting.rs:154-173)
        if (RELAX && finished) {
            if constexpr (RELAX) {
                // round(c * 2^S) as an integer: c in [0, 1] (clamped Color), so c * 2^S + 2^52 lies in [2^52, 2^53], where
                // doubles are the integers -- one fused multiply-add rounds once, and the integer is the difference of
                // the bit patterns.  A colour that is not a number sets the pixel's NaN flag instead.
                const bool black = col.x == real(0) && col.y == real(0) && col.z == real(0);   // adds nothing (depth ran out, scatter None)
                if (!black) {
                    const uint32_t tile = (pix_j >> A.sg_lh) * A.tiles_x + (pix_i >> A.sg_lw);
                    const uint32_t px = ((pix_j & ((1u << A.sg_lh) - 1u)) << A.sg_lw) | (pix_i & ((1u << A.sg_lw) - 1u));
                    const real cc[3] = {col.x, col.y, col.z};
                    unsigned long long v[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const double x = (double)cc[c];
                        v[c] = (x == x) ? (unsigned long long)(__builtin_bit_cast(long long, __builtin_fma(x, A.fx_scale, 0x1.0p52)) - 0x4330000000000000ll) : kFxNaN;
                    }
                    const bool nan = (v[0] | v[1] | v[2]) >> 63;
                    unsigned long long* dst;
                    bool in_lds = true;
                    if (tile == fx_tile0) dst = fx_slots + px * 3u;
                    else if (tile == fx_tile1) dst = fx_slots + fx_words + px * 3u;
                    else { dst = A.fx_acc + ((size_t)pix_j * (size_t)cam.W + pix_i) * 3; in_lds = false; }
                    if (!nan) {
                        if (in_lds) {
                            auto d = (__attribute__((address_space(3))) unsigned long long*)dst;   // ds_add_u64, not a flat atomic
                            for (int c = 0; c < 3; c++) if (v[c]) (void)__hip_atomic_fetch_add(d + c, v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            auto d = (__attribute__((address_space(1))) unsigned long long*)dst;
                            for (int c = 0; c < 3; c++) if (v[c]) (void)__hip_atomic_fetch_add(d + c, v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        for (int c = 0; c < 3; c++) {
                            if (v[c] >> 63) atomicOr(dst + c, kFxNaN);
                            else if (v[c]) atomicAdd(dst + c, v[c]);
                        }
                    }
                }
            }
            state = ST_NEED_PIXEL;
        } else if (finished && A.sg_on) {   // the ordered sum happens in sg_finalize_kernel
            real* o = A.sample_buf + (size_t)item * 3;
            o[0] = col.x; o[1] = col.y; o[2] = col.z;
            state = ST_NEED_PIXEL;
        } else if (finished) {
            acc_r += col.x; acc_g += col.y; acc_b += col.z;
            sample++;
            if (sample >= A.sample_end) {
                size_t o = ((size_t)pix_j * (size_t)cam.W + pix_i) * 3;
                if (A.output_sum) { A.out[o] = acc_r; A.out[o + 1] = acc_g; A.out[o + 2] = acc_b; }
                else {
                    real cnt = (real)A.samples_total;
                    A.out[o] = acc_r / cnt; A.out[o + 1] = acc_g / cnt; A.out[o + 2] = acc_b / cnt;
                }
                state = ST_NEED_PIXEL;
            } else state = ST_NEED_SAMPLE;
        }
        CR_DIAG_ONLY({ unsigned long long t = __builtin_readcyclecounter(); d_t_shade += t - d_t0; d_t0 = t; })
    }

    if constexpr (RELAX) { fx_flush(0, fx_tile0); fx_flush(1, fx_tile1); }
    // flush work counters: one atomic per counter per wave
    auto wave_sum = [&](unsigned long long v) -> unsigned long long {
        unsigned long long s = v;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        return s;
    };
    unsigned long long s0 = wave_sum(c_seg), s1 = wave_sum(c_node), s2 = wave_sum(c_prim), s3 = wave_sum(c_tex);
    CR_DIAG_ONLY(
        {
            unsigned long long* c = (unsigned long long*)A.counters;
            for (int i = 0; i < DG_N; i++) { unsigned long long w = wave_sum(dg_store.v[i]); if (lane == 0) atomicAdd(&c[16 + i], w); }
            if (lane == 0) {
                atomicAdd(&c[9], d_t_regen); atomicAdd(&c[10], d_t_trace); atomicAdd(&c[11], d_t_shade);
                atomicAdd(&c[12], __builtin_readcyclecounter() - d_begin);
            }
        })
#ifdef CR_HOLD_VCC
    asm volatile("" :: "{vcc}"(vcc_hold));
#endif
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.counters[0], s0);
        atomicAdd((unsigned long long*)&A.counters[1], s1);
        atomicAdd((unsigned long long*)&A.counters[2], s2);
        atomicAdd((unsigned long long*)&A.counters[3], s3);

    }
}
// The ~70 launch parameters are read where they are used, through the kernarg segment's own address (constant address
// space: s_load, served by the scalar cache).  As a by-value parameter they were all loaded at kernel entry and stayed
// live for the whole launch: the f64 kernel spilled 97 of them to VGPR lanes (230 v_readlane / v_writelane, 20 VGPRs
// pushed to scratch); read this way it spills 6 (19, and 6).
template <typename real> CR_D const KernelArgs<real>& kernel_args() {
    return *(const KernelArgs<real>*)(const __attribute__((address_space(4))) KernelArgs<real>*)__builtin_amdgcn_kernarg_segment_ptr();
}
template <typename real, int RES, bool ANIM, bool ORD = false, bool CAMK = false, bool RELAX = false, bool SCREEN = false>
__global__ void __launch_bounds__(MaxBlock<real>::value) pathtrace_kernel(const KernelArgs<real> A) {
    // the LDS-resident f32 kernels have registers to spare and lose 1 % to the reloads: they keep the by-value parameter
    if constexpr (std::is_same<real, double>::value || RES != RES_LDS) pathtrace_body<real, RES, ANIM, ORD, CAMK, RELAX, SCREEN>(kernel_args<real>());
    else pathtrace_body<real, RES, ANIM, ORD, CAMK, RELAX, SCREEN>(A);
}

// The same kernel compiled for 6 waves per SIMD (<= 80 VGPRs, 512-thread groups), for trees far larger than the
// LDS window: there the walk waits on L2 / HBM reads and more resident waves hide more of that latency than the
// extra spills cost (1M spheres +11 %; the LDS-resident book1 and the 8K-wrapper teapot lose 3 % and stay on
// pathtrace_kernel).  RES_TOP only.
constexpr int LatencyBlock = 512;
template <typename real, bool ANIM, bool ORD = false, bool CAMK = false, bool RELAX = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(64, LatencyBlock), amdgpu_waves_per_eu(6, 6)))
pathtrace_kernel_latency(const KernelArgs<real> A) {
    pathtrace_body<real, RES_TOP, ANIM, ORD, CAMK, RELAX>(A);
}

// The screening records of a wrapper array: boxes rounded to the nearest f32 (the band of walk_round allows for either
// direction), links copied.  Run after every upload and after every refit of the f64 boxes.
template <typename real>
__global__ void __launch_bounds__(256) screen_from_entries_kernel(const Entry<real>* e, ScreenEntry* s, int32_t n) {
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const Entry<real> v = e[i];
    ScreenEntry o;
    for (int k = 0; k < 6; k++) o.b[k] = (float)v.b[k];
    o.skip = (uint32_t)v.skip << 5;
    o.hit = v.leaf < 0 ? (uint32_t)(-v.leaf) << 5 : (kScreenLeaf | (uint32_t)v.leaf);
    s[i] = o;
}

__global__ void __launch_bounds__(256) screen_from_ordered_entries_kernel(const EntryO<double>* e, ScreenEntryO* s, int32_t n) {
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const EntryO<double> v = e[i];
    ScreenEntryO o;
    for (int k = 0; k < 6; k++) o.b[k] = (float)v.b[k];
    if (v.leaf < 0) { o.axis = (uint32_t)(-v.leaf) & 3u; o.hit = (uint32_t)ordered_left(v.leaf) << 6; }
    else { o.axis = 3u; o.hit = kScreenLeaf | (uint32_t)v.leaf; }
    for (int k = 0; k < 8; k++) o.skip[k] = (uint32_t)v.skip[k] << 6;
    s[i] = o;
}

// CR_SUM_RELAXED: the fixed-point sums become the frame -- sum * 2^-S, divided by the sample count unless the raw
// sum of the shard is asked for; a set NaN flag gives NaN (the reference would have panicked in Color::new).
template <typename real>
__global__ void __launch_bounds__(256) fx_finalize_kernel(const unsigned long long* acc, real* out, size_t n, double inv_scale, double count,
                                                          int32_t output_sum) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = acc[i];
    const unsigned long long m = v & ~kFxNaN;
    // exact u64 -> f64 in two halves (each below 2^53), then one rounding in the add
    double s = ((double)(uint32_t)(m >> 32) * 4294967296.0 + (double)(uint32_t)m) * inv_scale;
    if (!output_sum) s = s / count;
    if (v & kFxNaN) s = __builtin_nan("");
    out[i] = (real)s;
}

// average_samples' running sum (ray_casting.rs:161-165) for the sample-granular mode: the batch's colours are added
// to the pixel's sum in sample order, exactly the order the pixel-owning lane uses; the last batch divides by the
// sample count (`/= count`, :168-170) or hands out the raw sum.
template <typename real>
__global__ void __launch_bounds__(256) sg_finalize_kernel(const KernelArgs<real> A, real* acc, int32_t batch_samples, int32_t first_batch,
                                                          int32_t last_batch) {
    // one thread per pixel, threads numbered tile by tile (the pixels of a tile are consecutive threads), so the
    // reads of a sample group are contiguous across the tile's threads
    const uint32_t tile_px = 1u << (A.sg_lw + A.sg_lh);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = (uint32_t)(t >> (A.sg_lw + A.sg_lh)), in_px = (uint32_t)t & (tile_px - 1u);
    if (tile >= A.tiles_x * A.tiles_y) return;
    const uint32_t pi = ((tile % A.tiles_x) << A.sg_lw) + (in_px & ((1u << A.sg_lw) - 1u));
    const uint32_t pj = ((tile / A.tiles_x) << A.sg_lh) + (in_px >> A.sg_lw);
    if (pi >= (uint32_t)A.cam.W || pj >= (uint32_t)A.cam.H) return;
    const size_t p = (size_t)pj * (size_t)A.cam.W + pi;
    const uint32_t ns = 64u >> (A.sg_lw + A.sg_lh);
    real r = 0, g = 0, b = 0;
    if (!first_batch) { r = acc[3 * p]; g = acc[3 * p + 1]; b = acc[3 * p + 2]; }
    for (int32_t s = 0; s < batch_samples; s++) {
        const uint32_t sg = (uint32_t)s / ns, ds = (uint32_t)s - sg * ns;
        const size_t item = ((size_t)tile * A.sg_groups + sg) * 64u + ((size_t)ds << (A.sg_lw + A.sg_lh)) + in_px;
        const real* c = A.sample_buf + item * 3;
        r += c[0]; g += c[1]; b += c[2];
    }
    if (last_batch) {
        if (A.output_sum) { A.out[3 * p] = r; A.out[3 * p + 1] = g; A.out[3 * p + 2] = b; }
        else {
            real cnt = (real)A.samples_total;
            A.out[3 * p] = r / cnt; A.out[3 * p + 1] = g / cnt; A.out[3 * p + 2] = b / cnt;
        }
    } else { acc[3 * p] = r; acc[3 * p + 1] = g; acc[3 * p + 2] = b; }
}

#endif   // __HIPCC__

}   // namespace cr
