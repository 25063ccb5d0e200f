/*
 * crucible_oracle.c -- CPU restatement of Crucible's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call it.  The product library (crucible_amd/csrc) never links or calls it.
 *
 * What it restates (all citations relative to /root/reference):
 *   src/camera/ray_casting.rs      cast_ray, ray_color, average_samples, Ray
 *   src/camera/rendering_compute.rs viewport / basis / defocus math
 *   src/camera/mod.rs:36-47,368-376 Viewport::new, sample_square
 *   src/objects/{mod,bvh,bvhwrapper,hitlist,sphere,triangle}.rs
 *   src/materials/{mod,lambertian,metal,dielectric}.rs, src/textures/{mod,solid_color,checker_texture,image_texture}.rs
 *   src/asset_loader/img_loader.rs:69-76, src/scene/mod.rs:37-45
 *   src/timeline/mod.rs:90-96,233-263 + transform_builder.rs closures
 *   src/utils.rs:78-697             Point3/Vec3, Color (clamped), Interval
 * It keeps the reference's structure on purpose: recursive ray_color, a boxed
 * BVH tree of wrappers whose span-1 leaves hold the same object twice, both
 * children always visited left then right, timelines evaluated at every hit.
 *
 * Compiled twice by oracle/Makefile with -ffp-contract=off:
 *   -DCR_ORACLE_F64  real = double : the reference's own scalar type
 *   -DCR_ORACLE_F32  real = float  : the same expression tree in f32, the twin of
 *                                    the library's CR_REAL_F32 mode
 *
 * Parity pinning.  The reference cannot be built here (Rust, no toolchain).  The
 * arithmetic helpers are pinned by the reference's own known-answer tests
 * (src/utils.rs:699-913, src/camera/mod.rs:378-397, src/timeline/mod.rs:266-350;
 * see tests/test_oracle_kat.py).  Nothing in the reference pins intersection, BVH,
 * scatter, texture, sky, camera-ray or whole-image results, and every random draw
 * there is unseeded, so for those this oracle is "parity unpinned": the restatement
 * itself (and fixtures generated from it, tests/golden/) is the pin.
 *
 * Deliberate, documented departures from the reference:
 *  - RNG: rand::rng() (thread-local ChaCha12, OS-seeded; rand 0.9.2, not vendored)
 *    is replaced by one seeded stream per (seed, pixel index, sample index):
 *    key = SplitMix64's finaliser of the three, draws = xorshift64* (Vigna 2016)
 *    started from that key.
 *    uniform [0,1): f64 = (u>>11)*2^-53 (rand's StandardUniform mapping),
 *    f32 = (u>>40)*2^-24; random_range(lo..hi) and (lo..=hi) = lo + (hi-lo)*u.
 *  - powi(2) = x*x; powi(5) = x*((x*x)*(x*x)) (LLVM's expansion of llvm.powi).
 *  - f32 build: f64::EPSILON -> FLT_EPSILON (triangle.rs:101); 1e-160 rounds to 0
 *    (utils.rs:131); set-up scalars (tan terms, viewport size) are computed in f64
 *    as fix_viewport does and rounded to f32 once.
 *  - Hittables::update_bb calls in BVHWrapper::hit (bvhwrapper.rs:104-106) write leaf
 *    boxes that traversal never reads; they are omitted (no observable effect).
 *  - A triangle's keyframes move its three vertices together (what
 *    Scene::translate_point(.., Local, ..) produces, scene_animator.rs).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/crucible_hip.h"

#if defined(CR_ORACLE_F32)
typedef float real;
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_FLOOR floorf
#define LIBM_ATAN2 atan2f
#define LIBM_ASIN asinf
#define LIBM_ACOS acosf
#define R_FMIN fminf
#define R_EPSILON FLT_EPSILON
#define R_INF HUGE_VALF
#define ORACLE_REAL_TYPE CR_REAL_F32
#elif defined(CR_ORACLE_F64)
typedef double real;
#define R_SQRT sqrt
#define R_FABS fabs
#define R_FLOOR floor
#define LIBM_ATAN2 atan2
#define LIBM_ASIN asin
#define LIBM_ACOS acos
#define R_FMIN fmin
#define R_EPSILON DBL_EPSILON
#define R_INF HUGE_VAL
#define ORACLE_REAL_TYPE CR_REAL_F64
#else
#error "define CR_ORACLE_F64 or CR_ORACLE_F32"
#endif

#define R(x) ((real)(x))
static const real R_PI = (real)3.14159265358979323846264338327950288; /* std::f64::consts::PI */

/* ------------------------------------------------------------------ atan2 / asin / acos
 * The reference calls f64::atan2 / asin / acos (ray_casting.rs:137-138, sphere.rs:42-43), i.e. the platform libm.
 * libm results are not bit-identical across platforms (glibc here, ocml on the device), and a last-ulp difference
 * can move a texel index.  Like the random generator, the build therefore DEFINES these three functions (DESIGN.md,
 * "software trigonometry") and this oracle and the device library each implement that definition from its text:
 *
 *   atan(x), x >= 0:   x < 7/16: t = x;  < 11/16: t = (2x-1)/(2+x), c = atan(.5);  < 19/16: t = (x-1)/(x+1), c = pi/4;
 *                      < 39/16: t = (x-1.5)/(1+1.5x), c = atan(1.5);  else t = -1/x, c = pi/2
 *                      z = t*t; s = t*(z*A(z));  result = t - s   or   c_hi - ((s - c_lo) - t)
 *   atan2(y, x):       quadrant fix-up of atan(|y|/|x|) as fdlibm's e_atan2.c does (pi - (z - pi_lo) for x < 0)
 *   asin(x):           |x| < .5: x + x*(z*S(z)), z = x*x;  else t = (1-|x|)/2, s = sqrt(t):
 *                      pio2_hi - (2*(s + s*(t*S(t))) - pio2_lo), sign of x
 *   acos(x):           |x| < .5: pio2_hi - (x - (pio2_lo - x*(z*S(z))));  x <= -.5: t = (1+x)/2: pi_hi - 2*(s + ((t*S(t))*s - pio2_lo));
 *                      x >= .5: t = (1-x)/2: 2*(s + s*(t*S(t)))
 *   A, S:              polynomials (scripts/gen_trig_coeffs.py), Horner from the highest coefficient, multiply then add.
 *
 * oracle_set_libm(1) switches to glibc's functions instead -- what the Rust binary would call on this box -- so the
 * tests can bound the difference (a few ulp per call; texel indices agree except on isolated pixels). */
static int g_use_libm = 0;
#if defined(CR_ORACLE_F64)
static const real TRIG_A[12] = {0x1.5555555555555p-2, -0x1.99999999998c5p-3, 0x1.2492492485503p-3, -0x1.c71c71bd2b8bcp-4,
                                0x1.745d154c84f7ap-4, -0x1.3b1375ce5bdc6p-4, 0x1.110c9ce7b0572p-4, -0x1.e170800a46210p-5,
                                0x1.ab59b417b2d3fp-5, -0x1.7001816fd063fp-5, 0x1.0f62bba6a2558p-5, -0x1.e4167464d3de8p-7};
static const real TRIG_S[16] = {0x1.5555555555555p-3, 0x1.3333333333334p-4, 0x1.6db6db6db6c75p-5, 0x1.f1c71c71dc217p-6,
                                0x1.6e8ba2e2f8089p-6, 0x1.1c4ec5dfe81d9p-6, 0x1.c99964e8e2de8p-7, 0x1.7a8b73dc1b007p-7,
                                0x1.3fa92e3923959p-7, 0x1.14f7ebcffc822p-7, 0x1.c232290f7ae75p-8, 0x1.1e6dafec868fcp-7,
                                -0x1.641b6703bb104p-9, 0x1.b20b9dc229eb5p-6, -0x1.dfdd83264a978p-6, 0x1.06c051be25377p-5};
#define TRIG_NA 12
#define TRIG_NS 16
static const real AT_HI[4] = {0x1.dac670561bb4fp-2, 0x1.921fb54442d18p-1, 0x1.f730bd281f69bp-1, 0x1.921fb54442d18p+0};
static const real AT_LO[4] = {0x1.a2b7f222f65e2p-56, 0x1.1a62633145c07p-55, 0x1.007887af0cbbdp-56, 0x1.1a62633145c07p-54};
static const real PI_HI = 0x1.921fb54442d18p+1, PI_LO = 0x1.1a62633145c07p-53;
#else
static const real TRIG_A[6] = {0x1.555556p-2f, -0x1.999968p-3f, 0x1.248626p-3f, -0x1.c4f1ecp-4f, 0x1.5d79d8p-4f, -0x1.8b0b06p-5f};
static const real TRIG_S[7] = {0x1.555556p-3f, 0x1.33331ep-4f, 0x1.6dc0fp-5f, 0x1.efedf8p-6f, 0x1.82db24p-6f, 0x1.5a80ap-7f, 0x1.fb7ca4p-6f};
#define TRIG_NA 6
#define TRIG_NS 7
static const real AT_HI[4] = {0.46364760398864746f, 0.7853981852531433f, 0.9827937483787537f, 1.5707963705062866f};
static const real AT_LO[4] = {5.01215868808913e-09f, -2.1855694143368964e-08f, -2.5131424052915463e-08f, -4.371138828673793e-08f};
static const real PI_HI = 3.1415927410125732f, PI_LO = -8.742277657347586e-08f;
#endif
static real trig_poly(const real* c, int n, real z) {
    real p = c[n - 1];
    for (int i = n - 2; i >= 0; i--) p = p * z + c[i];
    return p;
}
static real def_atan_pos(real x) {
    int id;
    real t;
    if (x < R(0.4375)) { id = -1; t = x; }
    else if (x < R(0.6875)) { id = 0; t = (R(2.0) * x - R(1.0)) / (R(2.0) + x); }
    else if (x < R(1.1875)) { id = 1; t = (x - R(1.0)) / (x + R(1.0)); }
    else if (x < R(2.4375)) { id = 2; t = (x - R(1.5)) / (R(1.0) + R(1.5) * x); }
    else { id = 3; t = R(-1.0) / x; }
    real z = t * t;
    real s = t * (z * trig_poly(TRIG_A, TRIG_NA, z));
    if (id < 0) return t - s;
    return AT_HI[id] - ((s - AT_LO[id]) - t);
}
static real def_atan2(real y, real x) {
    if (x != x || y != y) return x + y;
    int sx = signbit(x) != 0, sy = signbit(y) != 0;
    real pio2 = AT_HI[3];
    if (y == R(0.0)) return sx ? (sy ? -PI_HI : PI_HI) : y;
    if (x == R(0.0)) return sy ? -pio2 : pio2;
    real ax = R_FABS(x), ay = R_FABS(y);
    if (ax == R_INF) {
        real q;
        if (ay == R_INF) q = sx ? R(3.0) * AT_HI[1] : AT_HI[1];
        else q = sx ? PI_HI : R(0.0);
        return sy ? -q : q;
    }
    if (ay == R_INF) return sy ? -pio2 : pio2;
    real z = def_atan_pos(ay / ax);
    real res = sx ? PI_HI - (z - PI_LO) : z;
    return sy ? -res : res;
}
static real def_asin(real x) {
    real ax = R_FABS(x);
    if (ax < R(0.5)) { real z = x * x; return x + x * (z * trig_poly(TRIG_S, TRIG_NS, z)); }
    if (!(ax <= R(1.0))) return (x - x) / (x - x);
    real t = (R(1.0) - ax) * R(0.5);
    real s = R_SQRT(t);
    real res = AT_HI[3] - (R(2.0) * (s + s * (t * trig_poly(TRIG_S, TRIG_NS, t))) - AT_LO[3]);
    return x < R(0.0) ? -res : res;
}
static real def_acos(real x) {
    real ax = R_FABS(x);
    if (ax < R(0.5)) { real z = x * x; return AT_HI[3] - (x - (AT_LO[3] - x * (z * trig_poly(TRIG_S, TRIG_NS, z)))); }
    if (!(ax <= R(1.0))) return (x - x) / (x - x);
    if (x < R(0.0)) {
        real t = (R(1.0) + x) * R(0.5);
        real s = R_SQRT(t);
        real w = (t * trig_poly(TRIG_S, TRIG_NS, t)) * s - AT_LO[3];
        return PI_HI - R(2.0) * (s + w);
    }
    real t = (R(1.0) - x) * R(0.5);
    real s = R_SQRT(t);
    return R(2.0) * (s + s * (t * trig_poly(TRIG_S, TRIG_NS, t)));
}
static real R_ATAN2(real y, real x) { return g_use_libm ? LIBM_ATAN2(y, x) : def_atan2(y, x); }
static real R_ASIN(real x) { return g_use_libm ? LIBM_ASIN(x) : def_asin(x); }
static real R_ACOS(real x) { return g_use_libm ? LIBM_ACOS(x) : def_acos(x); }

/* ------------------------------------------------------------------ utils.rs */

typedef struct { real x, y, z; } Vec3; /* Point3 / Vec3, utils.rs:72-76 */

static Vec3 v3(real x, real y, real z) { Vec3 v = {x, y, z}; return v; }
static Vec3 v_neg(Vec3 a) { return v3(-a.x, -a.y, -a.z); }                       /* :248-257 */
static Vec3 v_add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); } /* :281-290 */
static Vec3 v_sub(Vec3 a, Vec3 b) { return v_add(a, v_neg(b)); }                  /* :293-298 self + (-rhs) */
static Vec3 v_scale(real s, Vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }      /* :301-321 */
static Vec3 v_div(Vec3 a, real s) { return v_scale(R(1.0) / s, a); }               /* :335-340 (1.0/rhs)*self */
static real v_len2(Vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }          /* :188-191 powi(2) sum */
static real v_len(Vec3 a) { return R_SQRT(v_len2(a)); }                           /* :184-186 */
static real v_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* :199-204 */
static Vec3 v_cross(Vec3 a, Vec3 b) {                                             /* :206-217 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static Vec3 v_unit(Vec3 a) { real l = v_len(a); return v_div(a, l); }             /* :220-223 */
static int v_near_zero(Vec3 a) {                                                  /* :194-197 */
    real tol = R(1e-8);
    return R_FABS(a.x) < tol && R_FABS(a.y) < tol && R_FABS(a.z) < tol;
}
static Vec3 v_reflect(Vec3 v, Vec3 n) {                                           /* :149-151 */
    return v_sub(v, v_scale(R(2.0) * v_dot(v, n), n));
}
static Vec3 v_refract(Vec3 v, Vec3 n, real etai_over_etat) {                      /* :157-163 */
    real cos_theta = R_FMIN(v_dot(v_neg(v), n), R(1.0));
    Vec3 perp = v_scale(etai_over_etat, v_add(v, v_scale(cos_theta, n)));
    Vec3 par = v_scale(-(R_SQRT(R_FABS(R(1.0) - v_len2(perp)))), n);
    return v_add(perp, par);
}

static real r_clamp(real x, real lo, real hi) { /* f64::clamp: NaN stays NaN */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

typedef struct { real r, g, b; } Color; /* utils.rs:339-342, invariant 0<=c<=1 */

static Color c3(real r, real g, real b) { Color c = {r, g, b}; return c; }
static real min2(real x, real y) { return x < y ? x : y; }   /* :462-464 */
static real max2(real x, real y) { return x > y ? x : y; }   /* :470-472 */
static Color c_neg(Color c) {                                /* :445-460 hilo complement */
    real k = min2(min2(c.r, c.g), c.b) + max2(max2(c.r, c.g), c.b);
    return c3(R_FABS(k - c.r), R_FABS(k - c.g), R_FABS(k - c.b));
}
static Color c_add(Color a, Color b) {                       /* :520-534 */
    return c3(r_clamp(a.r + b.r, 0, 1), r_clamp(a.g + b.g, 0, 1), r_clamp(a.b + b.b, 0, 1));
}
static Color c_scale(real s, Color c) {                      /* :563-579 impl Mul<Color> for f64 */
    Color m = (s < R(0.0)) ? c_neg(c) : c;
    real p = R_FABS(s);
    return c3(r_clamp(p * m.r, 0, 1), r_clamp(p * m.g, 0, 1), r_clamp(p * m.b, 0, 1));
}
static Color c_mul(Color a, Color b) {                       /* :582-596 */
    return c3(r_clamp(a.r * b.r, 0, 1), r_clamp(a.g * b.g, 0, 1), r_clamp(a.b * b.b, 0, 1));
}
static Color c_div(Color c, real s) {                        /* :599-607 */
    Color m = (s < R(0.0)) ? c_neg(c) : c;
    s = R_FABS(s);
    return c_scale(R(1.0) / s, m);
}

/* impl Display for Color, utils.rs:422-437: (255.0 * c.sqrt()) as u32.
 * `as u32` saturates: NaN -> 0, negative -> 0.  Always evaluated in f64. */
static uint32_t display_byte(double c) {
    double v = 255.0 * sqrt(c);
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

typedef struct { real min, max; } Interval; /* utils.rs:613-697 */
static int iv_contains(Interval i, real x) { return i.min <= x && x <= i.max; }
static int iv_surrounds(Interval i, real x) { return i.min < x && x < i.max; }
static int iv_is_less(Interval i, real x) { return x > i.max; }
static int iv_is_greater(Interval i, real x) { return x < i.min; }
static real iv_proportion(Interval i, real x) { return (x - i.min) / (i.max - i.min); }
static Interval iv_tight_enclose(Interval a, Interval b) {  /* :629-633 */
    Interval o;
    o.min = a.min <= b.min ? a.min : b.min;
    o.max = a.max >= b.max ? a.max : b.max;
    return o;
}

/* ------------------------------------------------------------------ RNG (see header) */

#define RNG_GAMMA 0x9E3779B97F4A7C15ULL
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
typedef struct { uint64_t s; uint64_t draws; } Rng;
static Rng rng_for_sample(uint64_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    r.s = mix64(mix64(seed + RNG_GAMMA) ^ (((uint64_t)pixel << 32) | (uint64_t)sample));
    if (r.s == 0) r.s = RNG_GAMMA;   /* xorshift state must not be zero */
    r.draws = 0;
    return r;
}
/* xorshift64* (Vigna 2016): Marsaglia's 12/25/27 xorshift, output scrambled by one multiply */
static uint64_t rng_u64(Rng* r) {
    uint64_t s = r->s;
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    r->s = s; r->draws++;
    return s * 0x2545F4914F6CDD1DULL;
}
static real rng_uniform(Rng* r) {
    uint64_t u = rng_u64(r);
#if defined(CR_ORACLE_F32)
    return (real)(u >> 40) * 0x1.0p-24f;
#else
    return (real)(u >> 11) * 0x1.0p-53;
#endif
}
static real rng_range(Rng* r, real lo, real hi) { return lo + (hi - lo) * rng_uniform(r); }

static Vec3 random_in_unit_disk(Rng* r) {   /* utils.rs:110-124 */
    for (;;) {
        real x = rng_range(r, R(-1.0), R(1.0));
        real y = rng_range(r, R(-1.0), R(1.0));
        Vec3 p = v3(x, y, R(0.0));
        if (v_len2(p) < R(1.0)) return p;
    }
}
static Vec3 random_unit_vector(Rng* r) {    /* utils.rs:127-136 */
    for (;;) {
        real x = rng_range(r, R(-1.0), R(1.0));   /* random_vec3_range: x, y, z in order, :101-108 */
        real y = rng_range(r, R(-1.0), R(1.0));
        real z = rng_range(r, R(-1.0), R(1.0));
        Vec3 p = v3(x, y, z);
        real lensq = v_len2(p);
        if (R(1e-160) < lensq && lensq <= R(1.0)) return v_div(p, R_SQRT(lensq));
    }
}

/* ------------------------------------------------------------------ timeline */

/* One Transform of a TransformTimeline (timeline/mod.rs:64-71), flattened:
 * see CrKeyframe in include/crucible_hip.h. */
typedef struct { int channel, interp; real t0, t1, a, b; } Key;

typedef struct {
    real init[4];       /* x,y,z and w (radius for spheres, 1.0 scale otherwise) */
    int n_keys;
    const Key* keys;
} Timeline;

static real key_scaled_time(const Key* k, real t) {   /* timeline/mod.rs:92 */
    Interval iv = {k->t0, k->t1};
    return r_clamp(iv_proportion(iv, t), R(0.0), R(1.0));
}
static int key_active(const Key* k, real t) {          /* timeline/mod.rs:239 */
    Interval iv = {k->t0, k->t1};
    return iv_is_less(iv, t) || iv_contains(iv, t);
}

/* S * (x, y, z, 1) for a non-sphere timeline; S = the last active scale transform (timeline/mod.rs:249-255):
 *   skind < 0      the initial build_other_scaler(s): diag(s,s,s,s)             (matrix_builder.rs:63-86)
 *   CR_KEY_SCALE_X diag(v,1,1,1)                                               (transform_builder.rs:101-179)
 *   CR_KEY_SCALE_Y identity with v in ROW 1, COLUMN 0: y' = v*x + y            (transform_builder.rs:228-246)
 *   CR_KEY_SCALE_Z diag(1,1,v,1)                                               (transform_builder.rs:262-340)
 * nalgebra forms each component as ((S_i0*x + S_i1*y) + S_i2*z) + S_i3*1; rows holding only a unit diagonal
 * give the coordinate back (1*c plus zero products), which is written as the coordinate itself here. */
static void scale_apply(int skind, real v, real x, real y, real z, real out[4]) {
    if (skind == CR_KEY_SCALE_X) { out[0] = v * x; out[1] = y; out[2] = z; out[3] = R(1.0); }
    else if (skind == CR_KEY_SCALE_Y) { out[0] = x; out[1] = v * x + y; out[2] = z; out[3] = R(1.0); }
    else if (skind == CR_KEY_SCALE_Z) { out[0] = x; out[1] = y; out[2] = v * z; out[3] = R(1.0); }
    else { out[0] = v * x; out[1] = v * y; out[2] = v * z; out[3] = v; }
}

/* TransformTimeline::combine_and_compute, timeline/mod.rs:233-263.  Translation
 * matrices multiply into a sum of offsets, added in list order starting from the
 * initial position (the t=-0.1 Omni transform is always active and first);
 * the scale is the last active scale transform.  is_sphere selects
 * build_sphere_scaler (w = radius, xyz untouched) vs the non-sphere scale
 * matrices of scale_apply.
 * side: 0 = the value at t; 1 = the left limit at a key start (a key whose t0 equals t counts as not yet
 * active) -- used by the refit rule only.  tr / sc: which parts to evaluate at this t (the refit rule samples
 * them independently): the translate part goes to xyz[0..2], the scale part to *w and *skind. */
static void timeline_parts(const Timeline* tl, real t, int side, real xyz[3], real* w, int* skind) {
    real x = R(0.0) + tl->init[0], y = R(0.0) + tl->init[1], z = R(0.0) + tl->init[2];
    real wv = tl->init[3];
    int sk = -1;
    for (int i = 0; i < tl->n_keys; i++) {
        const Key* k = &tl->keys[i];
        int active;
        if (side) {
            int started = k->t0 < t;
            active = (t > k->t1) || (started && t <= k->t1);
        } else active = key_active(k, t);
        if (!active) continue;
        real s = key_scaled_time(k, t);
        if (k->channel <= CR_KEY_TZ) {
            real val = (k->interp == CR_KEY_LERP) ? k->a * s : k->a;
            if (k->channel == CR_KEY_TX) x = x + val;
            else if (k->channel == CR_KEY_TY) y = y + val;
            else z = z + val;
        } else {   /* ScaleR | ScaleX | ScaleY | ScaleZ: `start + (x - start) * t` */
            wv = (k->interp == CR_KEY_LERP) ? k->a + (k->b - k->a) * s : k->a;
            sk = k->channel;
        }
    }
    xyz[0] = x; xyz[1] = y; xyz[2] = z; *w = wv; *skind = sk;
}
/* ---- "faithful" evaluation (bench.py's second CPU baseline): the same numbers computed the way the
 * reference computes them at EVERY hit -- each Transform is a 4x4 matrix of boxed closures (timeline/mod.rs:20-22,
 * 64-71); get_matrix_at_time calls all 16 of them, each after its own proportion + clamp (:90-96); the active
 * translate matrices are multiplied together, then with the scale matrix, then with (0,0,0,1) (:233-263), in
 * nalgebra's accumulation order.  oracle_set_faithful(1) also enables the dead update_bb calls of BVHWrapper::hit
 * (bvhwrapper.rs:104-106), the reference-count traffic of HitRecord's material clones (objects/mod.rs:53,101-103),
 * the per-thread deep clone of the world (cpu_threading.rs:44) and the mutex around every pixel hand-out (:88).
 * Results are bit-identical to the direct evaluation (tests/test_oracle_properties.py). */
static int g_faithful = 0;
typedef struct { real a, b; } ClosureEnv;
typedef struct { real (*call)(const ClosureEnv*, real); ClosureEnv env; } MatrixInfo;   /* Arc<dyn Fn(f64) -> f64> */
static real closure_const(const ClosureEnv* e, real t) { (void)t; return e->a; }          /* move |_t| x */
static real closure_times(const ClosureEnv* e, real t) { return e->a * t; }               /* move |t| x * t */
static real closure_lerp(const ClosureEnv* e, real t) { return e->a + (e->b - e->a) * t; } /* move |t| s + (x - s) * t */
typedef struct { MatrixInfo m[16]; Interval valid; } TransformM;   /* column-major like nalgebra: m[col * 4 + row] */
static void tm_identity(TransformM* tf, real t0, real t1) {
    for (int i = 0; i < 16; i++) { tf->m[i].call = closure_const; tf->m[i].env.a = (i % 5 == 0) ? R(1.0) : R(0.0); tf->m[i].env.b = R(0.0); }
    tf->valid.min = t0; tf->valid.max = t1;
}
static void tm_set(TransformM* tf, int row, int col, real (*call)(const ClosureEnv*, real), real a, real b) {
    tf->m[col * 4 + row].call = call; tf->m[col * 4 + row].env.a = a; tf->m[col * 4 + row].env.b = b;
}
static void tm_at_time(const TransformM* tf, real t, real out[16]) {   /* Transform::get_matrix_at_time */
    for (int i = 0; i < 16; i++) {
        real scaled = r_clamp(iv_proportion(tf->valid, t), R(0.0), R(1.0));
        out[i] = tf->m[i].call(&tf->m[i].env, scaled);
    }
}
static void mat4_mul(const real a[16], const real b[16], real c[16]) {   /* gemv per column, axpy per term */
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            real y = a[0 * 4 + i] * b[j * 4 + 0];
            for (int k = 1; k < 4; k++) y = a[k * 4 + i] * b[j * 4 + k] + y;
            c[j * 4 + i] = y;
        }
}
static void timeline_eval_matrices(const Timeline* tl, real t, int is_sphere, real out[4]) {
    real tm[16], m[16], tmp[16];
    for (int i = 0; i < 16; i++) tm[i] = (i % 5 == 0) ? R(1.0) : R(0.0);   /* build_identity_f64 */
    TransformM tf;
    tm_identity(&tf, R(-0.1), R(-0.1));                                    /* build_pos(start_pos), always active */
    for (int r = 0; r < 3; r++) tm_set(&tf, r, 3, closure_const, tl->init[r], R(0.0));
    tm_at_time(&tf, t, m); mat4_mul(m, tm, tmp); memcpy(tm, tmp, sizeof tm);
    for (int i = 0; i < tl->n_keys; i++) {
        const Key* k = &tl->keys[i];
        if (k->channel > CR_KEY_TZ || !key_active(k, t)) continue;
        tm_identity(&tf, k->t0, k->t1);
        tm_set(&tf, k->channel, 3, k->interp == CR_KEY_LERP ? closure_times : closure_const, k->a, R(0.0));
        tm_at_time(&tf, t, m); mat4_mul(m, tm, tmp); memcpy(tm, tmp, sizeof tm);
    }
    /* the last active scale transform: the initial scaler unless a key took over */
    tm_identity(&tf, R(-0.1), R(-0.1));
    if (is_sphere) tm_set(&tf, 3, 3, closure_const, tl->init[3], R(0.0));                   /* build_sphere_scaler */
    else for (int d = 0; d < 4; d++) tm_set(&tf, d, d, closure_const, tl->init[3], R(0.0)); /* build_other_scaler */
    for (int i = 0; i < tl->n_keys; i++) {
        const Key* k = &tl->keys[i];
        if (k->channel <= CR_KEY_TZ || !key_active(k, t)) continue;
        tm_identity(&tf, k->t0, k->t1);
        int row = k->channel == CR_KEY_RADIUS ? 3 : (k->channel == CR_KEY_SCALE_X ? 0 : (k->channel == CR_KEY_SCALE_Y ? 1 : 2));
        int col = k->channel == CR_KEY_RADIUS ? 3 : (k->channel == CR_KEY_SCALE_Y ? 0 : row);   /* ScaleY: row 1, column 0 */
        tm_set(&tf, row, col, k->interp == CR_KEY_LERP ? closure_lerp : closure_const, k->a, k->b);
    }
    real sm[16], comb[16];
    tm_at_time(&tf, t, sm);
    mat4_mul(sm, tm, comb);                                                /* scale_matrix * translate_matrix */
    for (int i = 0; i < 4; i++) {                                          /* combined * (0, 0, 0, 1) */
        real y = comb[0 * 4 + i] * R(0.0);
        y = comb[1 * 4 + i] * R(0.0) + y; y = comb[2 * 4 + i] * R(0.0) + y; y = comb[3 * 4 + i] * R(1.0) + y;
        out[i] = y;
    }
}

static void timeline_eval(const Timeline* tl, real t, int is_sphere, real out[4]) {
    if (g_faithful) { timeline_eval_matrices(tl, t, is_sphere, out); return; }
    real p[3], w;
    int sk;
    timeline_parts(tl, t, 0, p, &w, &sk);
    if (is_sphere) { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = w; }
    else scale_apply(sk, w, p[0], p[1], p[2], out);
}

/* ------------------------------------------------------------------ scene objects */

typedef struct { Interval x, y, z; } Aabb;   /* bvh.rs:19-23 */
static const Interval IV_EMPTY = {HUGE_VAL, -HUGE_VAL};

static Aabb aabb_empty(void) { Aabb b = {IV_EMPTY, IV_EMPTY, IV_EMPTY}; return b; }
static Aabb aabb_from_points(Vec3 a, Vec3 b) {  /* bvh.rs:44-64 */
    Aabb o;
    if (a.x <= b.x) { o.x.min = a.x; o.x.max = b.x; } else { o.x.min = b.x; o.x.max = a.x; }
    if (a.y <= b.y) { o.y.min = a.y; o.y.max = b.y; } else { o.y.min = b.y; o.y.max = a.y; }
    if (a.z <= b.z) { o.z.min = a.z; o.z.max = b.z; } else { o.z.min = b.z; o.z.max = a.z; }
    return o;
}
static Aabb aabb_from_boxes(Aabb a, Aabb b) {   /* bvh.rs:67-73 */
    Aabb o;
    o.x = iv_tight_enclose(a.x, b.x);
    o.y = iv_tight_enclose(a.y, b.y);
    o.z = iv_tight_enclose(a.z, b.z);
    return o;
}
static int aabb_longest_axis(Aabb b) {          /* bvh.rs:82-94, strict > */
    real sx = b.x.max - b.x.min, sy = b.y.max - b.y.min, sz = b.z.max - b.z.min;
    if (sx > sy) return (sx > sz) ? 0 : 2;
    return (sy > sz) ? 1 : 2;
}

typedef struct { Vec3 origin, direction; real tm; } Ray;   /* ray_casting.rs:15-20 */
static Vec3 ray_at(const Ray* r, real t) { return v_add(r->origin, v_scale(t, r->direction)); } /* :53-59 */

/* Aabb::hit, bvh.rs:96-132.  ray_t is the caller's clone. */
static int aabb_hit(const Aabb* b, const Ray* r, Interval ray_t) {
    for (int axis = 0; axis < 3; axis++) {
        Interval ax = axis == 0 ? b->x : (axis == 1 ? b->y : b->z);
        real o = axis == 0 ? r->origin.x : (axis == 1 ? r->origin.y : r->origin.z);
        real d = axis == 0 ? r->direction.x : (axis == 1 ? r->direction.y : r->direction.z);
        real adinv = R(1.0) / d;
        real t0 = (ax.min - o) * adinv;
        real t1 = (ax.max - o) * adinv;
        real new_min, new_max;
        if (t0 < t1) {
            new_min = t0 > ray_t.min ? t0 : ray_t.min;
            new_max = t1 < ray_t.max ? t1 : ray_t.max;
        } else {
            new_min = t1 > ray_t.min ? t1 : ray_t.min;
            new_max = t0 < ray_t.max ? t0 : ray_t.max;
        }
        ray_t.min = new_min;
        ray_t.max = new_max;
        if (ray_t.max <= ray_t.min) return 0;
    }
    return 1;
}

typedef struct {   /* CrTexture in real */
    int kind, even, odd, image;
    Color color;
    real inv_scale;
} Texture;
typedef struct {   /* CrMaterial in real */
    int kind, texture;
    Color albedo;
    real param;
} Material;
typedef struct { int w, h; uint8_t* rgb; } Image;

enum { H_SPHERE = 0, H_HITLIST = 1, H_BVH = 2, H_TRIANGLE = 3 };   /* objects/mod.rs:109-115 */

typedef struct Hittable {
    int kind;
    int mat;
    int hide;
    int member;              /* an object of a HitList element, not a scene element itself (CR_PRIM_MEMBER) */
    Timeline tl;             /* sphere: centre+radius; triangle: vertex a (b, c below) */
    real vb[3], vc[3];
    Aabb bbox;
    Aabb bbox0;              /* wrappers: the construction-time box (refit_boxes = 0 restores it) */
    int near_axis;           /* wrappers: 0 = BVHWrapper::hit's order (left, right); k+1 = CR_BVH_SAH_ORDERED: the right
                                child first when direction[k] < 0 (not in the reference; set by oracle_set_tree only) */
    struct Hittable* left;   /* BVHWrapper, bvhwrapper.rs:7-11 */
    struct Hittable* right;
    struct Hittable** objs;  /* HitList, hitlist.rs:7-10 */
    int n_objs;
} Hittable;

typedef struct {   /* HitRecord, objects/mod.rs:21-29 */
    Vec3 loc, normal;
    int mat;
    real t, u, v;
    int front_face;
} HitRecord;

typedef struct {
    uint64_t segments, node_tests, prim_tests, prim_tests_dedup, texel_fetches;
} Counters;

typedef struct Scene {
    int n_prims, n_materials, n_textures, n_images, n_keys;
    Hittable* prims;         /* the flat element list */
    Material* materials;
    Texture* textures;
    Image* images;
    Key* keys;
    int sky_kind, sky_image;
    int* mat_rc;             /* faithful mode: stands in for the Arc<Textures> reference counts */
    Hittable* world;         /* BVHWrapper::new_wrapper result */
    Hittable* pool;          /* wrapper nodes */
    int pool_used, pool_cap;
    Hittable empty_list;
} Scene;

static Aabb sphere_bbox(Vec3 c, real radius) {   /* sphere.rs:29-30 */
    Vec3 rvec = v3(radius, radius, radius);
    return aabb_from_points(v_sub(c, rvec), v_add(c, rvec));
}
static Aabb triangle_bbox(Vec3 a, Vec3 b, Vec3 c) {   /* triangle.rs:28-35,48-62 */
    Aabb o;
    o.x.max = fmax(a.x, fmax(b.x, c.x)); o.y.max = fmax(a.y, fmax(b.y, c.y)); o.z.max = fmax(a.z, fmax(b.z, c.z));
    o.x.min = fmin(a.x, fmin(b.x, c.x)); o.y.min = fmin(a.y, fmin(b.y, c.y)); o.z.min = fmin(a.z, fmin(b.z, c.z));
    return o;
}

/* HitRecord::new (unsafe), objects/mod.rs:38-61 */
static HitRecord hitrec_new(const Ray* r, Vec3 loc, Vec3 normal, real t, real u, real v, int mat) {
    HitRecord h;
    h.front_face = v_dot(r->direction, normal) < R(0.0);
    h.normal = h.front_face ? normal : v_neg(normal);
    h.loc = loc; h.t = t; h.u = u; h.v = v; h.mat = mat;
    return h;
}
/* HitRecord::safe_new, objects/mod.rs:67-90 */
static HitRecord hitrec_safe_new(const Ray* r, Vec3 loc, Vec3 normal, real t, real u, real v, int mat) {
    return hitrec_new(r, loc, v_unit(normal), t, u, v, mat);
}

static void sphere_uv(Vec3 p, real* u, real* v) {   /* sphere.rs:41-46 */
    real theta = R_ACOS(-p.y);
    real phi = R_ATAN2(-p.z, p.x) + R_PI;
    *u = phi / (R(2.0) * R_PI);
    *v = theta / R_PI;
}

static int sphere_hit(const Hittable* s, const Ray* r, Interval ray_t, HitRecord* rec) {   /* sphere.rs:60-105 */
    if (s->hide) return 0;
    real sp[4];
    timeline_eval(&s->tl, r->tm, 1, sp);
    Vec3 center = v3(sp[0], sp[1], sp[2]);
    real radius = sp[3];
    Vec3 oc = v_sub(center, r->origin);
    real a = v_len2(r->direction);
    real h = v_dot(r->direction, oc);
    real c = v_len2(oc) - radius * radius;
    real disc = h * h - a * c;
    if (disc < R(0.0)) return 0;
    real sqrtd = R_SQRT(disc);
    real root = (h - sqrtd) / a;
    if (!iv_surrounds(ray_t, root)) {
        root = (h + sqrtd) / a;
        if (!iv_surrounds(ray_t, root)) return 0;
    }
    real t = root;
    Vec3 p = ray_at(r, t);
    Vec3 n = v_div(v_sub(p, center), radius);
    real u, v;
    sphere_uv(n, &u, &v);
    *rec = hitrec_new(r, p, n, t, u, v, s->mat);
    return 1;
}

static int triangle_hit(const Hittable* tr, const Ray* r, Interval ray_t, HitRecord* rec) {   /* triangle.rs:84-140 */
    if (tr->hide) return 0;
    real pa[4], pb[4], pc[4];
    Timeline tlb = tr->tl, tlc = tr->tl;
    tlb.init[0] = tr->vb[0]; tlb.init[1] = tr->vb[1]; tlb.init[2] = tr->vb[2];
    tlc.init[0] = tr->vc[0]; tlc.init[1] = tr->vc[1]; tlc.init[2] = tr->vc[2];
    timeline_eval(&tr->tl, r->tm, 0, pa);
    timeline_eval(&tlb, r->tm, 0, pb);
    timeline_eval(&tlc, r->tm, 0, pc);
    Vec3 a = v3(pa[0], pa[1], pa[2]), b = v3(pb[0], pb[1], pb[2]), c = v3(pc[0], pc[1], pc[2]);
    Vec3 e1 = v_sub(b, a), e2 = v_sub(c, a);
    Vec3 ray_cross_e2 = v_cross(r->direction, e2);
    real det = v_dot(e1, ray_cross_e2);
    if (det > -R_EPSILON && det < R_EPSILON) return 0;
    real inv_det = R(1.0) / det;
    Vec3 s = v_sub(r->origin, a);
    real u = inv_det * v_dot(s, ray_cross_e2);
    if (!(R(0.0) <= u && u <= R(1.0))) return 0;   /* !(0.0..=1.0).contains(&u) */
    Vec3 s_cross_e1 = v_cross(s, e1);
    real v = inv_det * v_dot(r->direction, s_cross_e1);
    if (v < R(0.0) || u + v > R(1.0)) return 0;
    real t = inv_det * v_dot(e2, s_cross_e1);
    if (!iv_surrounds(ray_t, t)) return 0;
    Vec3 p = ray_at(r, t);
    Vec3 normal = v_cross(e1, e2);
    *rec = hitrec_safe_new(r, p, normal, t, R(0.0), R(0.0), tr->mat);
    return 1;
}

static int hittable_hit(Hittable* h, const Ray* r, Interval ray_t, HitRecord* rec, Counters* cn);
static int* g_mat_rc = NULL;
static __thread real g_dead_box[6];
/* Hittables::update_bb on a leaf child (objects/mod.rs:129-145): recomputes the primitive's box at the ray's time;
 * BVHWrapper::hit never reads it (bvhwrapper.rs:104-106).  Faithful mode only. */
static void dead_update_bb(const Hittable* h, real t) {
    if (h->kind == H_SPHERE) {
        real sp[4];
        timeline_eval(&h->tl, t, 1, sp);
        Aabb b = sphere_bbox(v3(sp[0], sp[1], sp[2]), sp[3]);
        g_dead_box[0] = b.x.min; g_dead_box[1] = b.x.max; g_dead_box[2] = b.y.min; g_dead_box[3] = b.y.max; g_dead_box[4] = b.z.min; g_dead_box[5] = b.z.max;
    } else if (h->kind == H_TRIANGLE) {
        real pa[4], pb[4], pc[4];
        Timeline tlb = h->tl, tlc = h->tl;
        tlb.init[0] = h->vb[0]; tlb.init[1] = h->vb[1]; tlb.init[2] = h->vb[2];
        tlc.init[0] = h->vc[0]; tlc.init[1] = h->vc[1]; tlc.init[2] = h->vc[2];
        timeline_eval(&h->tl, t, 0, pa); timeline_eval(&tlb, t, 0, pb); timeline_eval(&tlc, t, 0, pc);
        Aabb b = triangle_bbox(v3(pa[0], pa[1], pa[2]), v3(pb[0], pb[1], pb[2]), v3(pc[0], pc[1], pc[2]));
        g_dead_box[0] = b.x.min; g_dead_box[1] = b.x.max; g_dead_box[2] = b.y.min; g_dead_box[3] = b.y.max; g_dead_box[4] = b.z.min; g_dead_box[5] = b.z.max;
    } else if (h->kind == H_HITLIST) {   /* HitList::update_bb, hitlist.rs:33-42: every object's update, then their union (never read) */
        for (int i = 0; i < h->n_objs; i++) dead_update_bb(h->objs[i], t);
    }
}

/* HitList::hit, hitlist.rs:51-65 */
static int hitlist_hit(Hittable* l, const Ray* r, Interval ray_t, HitRecord* rec, Counters* cn) {
    int any = 0;
    real closest = ray_t.max;
    for (int i = 0; i < l->n_objs; i++) {
        Interval iv = {ray_t.min, closest};
        HitRecord tmp;
        if (hittable_hit(l->objs[i], r, iv, &tmp, cn)) { closest = tmp.t; *rec = tmp; any = 1; }
    }
    return any;
}

/* BVHWrapper::hit, bvhwrapper.rs:96-126 */
static int bvh_hit(Hittable* b, const Ray* r, Interval ray_t, HitRecord* rec, Counters* cn) {
    cn->node_tests++;
    if (!aabb_hit(&b->bbox, r, ray_t)) return 0;
    if (g_faithful) { dead_update_bb(b->left, r->tm); dead_update_bb(b->right, r->tm); }
    HitRecord hl, hr;
    /* A BVHWrapper element (CR_PRIM_BVH) under a leaf wrapper.  The device stores the wrapper's other child -- a primitive
     * or list, which the reference tests without any box -- as a record of its own with an empty box, and a span-1
     * wrapper's second copy of the element as an empty record: one more record visited in either case. */
    const int lb = b->left->kind == H_BVH, rb = b->right->kind == H_BVH;
    if (lb != rb || (lb && b->left == b->right)) cn->node_tests++;
    Hittable *first = b->left, *second = b->right;
    if (b->near_axis) {
        const real d = b->near_axis == 1 ? r->direction.x : (b->near_axis == 2 ? r->direction.y : r->direction.z);
        if (d < R(0.0)) { first = b->right; second = b->left; }
    }
    int hit_left = hittable_hit(first, r, ray_t, &hl, cn);
    Interval right_t = {ray_t.min, hit_left ? hl.t : ray_t.max};
    /* a span-1 wrapper holds the same object twice; the reference tests it twice (a list: all its objects twice) */
    const Counters before = *cn;
    int hit_right = hittable_hit(second, r, right_t, &hr, cn);
    if (b->left == b->right && b->left->kind != H_BVH) cn->prim_tests_dedup = before.prim_tests_dedup;
    /* the same sub-tree walked a second time finds nothing closer (every t is outside the shrunk interval): counted once */
    if (b->left == b->right && b->left->kind == H_BVH) *cn = before;
    if (hit_right) { *rec = hr; return 1; }
    if (hit_left) { *rec = hl; return 1; }
    return 0;
}

static int hittable_hit(Hittable* h, const Ray* r, Interval ray_t, HitRecord* rec, Counters* cn) {   /* objects/mod.rs:118-125 */
    int hit;
    /* a hidden object of a list answers None before anything is computed (sphere.rs:62, triangle.rs:87); not counted as a test */
    if (h->hide && (h->kind == H_SPHERE || h->kind == H_TRIANGLE)) return 0;
    switch (h->kind) {
        case H_SPHERE: cn->prim_tests++; cn->prim_tests_dedup++; hit = sphere_hit(h, r, ray_t, rec); break;
        case H_HITLIST: return hitlist_hit(h, r, ray_t, rec, cn);
        case H_BVH: return bvh_hit(h, r, ray_t, rec, cn);
        default: cn->prim_tests++; cn->prim_tests_dedup++; hit = triangle_hit(h, r, ray_t, rec); break;
    }
    if (hit && g_faithful && g_mat_rc) {   /* HitRecord { mat: mat.clone() } and rec.material() clone the Arc, both dropped later */
        __sync_fetch_and_add(&g_mat_rc[h->mat], 2);
        __sync_fetch_and_sub(&g_mat_rc[h->mat], 2);
    }
    return hit;
}

/* ------------------------------------------------------------------ BVH build */

static Hittable* pool_new(Scene* sc) {
    if (sc->pool_used == sc->pool_cap) { fprintf(stderr, "oracle: BVH pool exhausted\n"); abort(); }
    Hittable* h = &sc->pool[sc->pool_used++];
    memset(h, 0, sizeof *h);
    return h;
}

static real box_axis_min(const Hittable* h, int axis) {
    return axis == 0 ? h->bbox.x.min : (axis == 1 ? h->bbox.y.min : h->bbox.z.min);
}
/* stable merge sort on bbox min of `axis` -- sort_by(box_compare), bvhwrapper.rs:66-67,80-92 */
static void stable_sort_axis(Hittable** a, Hittable** tmp, int n, int axis) {
    if (n < 2) return;
    int m = n / 2;
    stable_sort_axis(a, tmp, m, axis);
    stable_sort_axis(a + m, tmp, n - m, axis);
    int i = 0, j = m, k = 0;
    while (i < m && j < n) {
        /* take right only when strictly Less than left: keeps equal elements in order */
        if (box_axis_min(a[j], axis) < box_axis_min(a[i], axis)) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < m) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof *a);
}

/* BVHWrapper::help_generate, bvhwrapper.rs:46-78 */
static Hittable* bvh_generate(Scene* sc, Hittable** objects, Hittable** tmp, int start, int end) {
    Aabb bbox = aabb_empty();
    for (int i = start; i < end; i++) bbox = aabb_from_boxes(bbox, objects[i]->bbox);
    int axis = aabb_longest_axis(bbox);
    int span = end - start;
    Hittable *left, *right;
    if (span == 1) { left = objects[start]; right = objects[start]; }
    else if (span == 2) { left = objects[start]; right = objects[start + 1]; }
    else {
        stable_sort_axis(objects + start, tmp, span, axis);
        int mid = start + span / 2;
        left = bvh_generate(sc, objects, tmp, start, mid);
        right = bvh_generate(sc, objects, tmp, mid, end);
    }
    Hittable* w = pool_new(sc);
    w->kind = H_BVH; w->left = left; w->right = right; w->bbox = bbox;
    return w;
}

/* BVHWrapper::new_wrapper + new_from_vec, bvhwrapper.rs:15-44 */
static void scene_build_world(Scene* sc) {
    int n_vis = 0;
    Hittable** vis = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(sc->n_prims + 1));
    Hittable** tmp = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(sc->n_prims + 1));
    /* bvhwrapper.rs:16-26: hidden spheres and triangles are dropped, a HitList stays whatever it holds */
    for (int i = 0; i < sc->n_prims; i++)
        if (!sc->prims[i].member && (sc->prims[i].kind == H_HITLIST || sc->prims[i].kind == H_BVH || !sc->prims[i].hide)) vis[n_vis++] = &sc->prims[i];
    if (n_vis == 0) {
        memset(&sc->empty_list, 0, sizeof sc->empty_list);
        sc->empty_list.kind = H_HITLIST;
        sc->world = &sc->empty_list;
    } else {
        if (!sc->pool) {
            sc->pool_cap = 2 * n_vis + 4;
            sc->pool = (Hittable*)malloc(sizeof(Hittable) * (size_t)sc->pool_cap);
            sc->pool_used = 0;
        }
        Hittable* root = bvh_generate(sc, vis, tmp, 0, n_vis);
        root->bbox = aabb_from_boxes(root->left->bbox, root->right->bbox);   /* new_from_vec :39 */
        sc->world = root;
        for (int i = 0; i < sc->pool_used; i++) sc->pool[i].bbox0 = sc->pool[i].bbox;
    }
    for (int i = 0; i < sc->n_prims; i++) sc->prims[i].bbox0 = sc->prims[i].bbox;   /* BVHWrapper elements live here */
    free(vis); free(tmp);
}

/* ------------------------------------------------------------------ textures, materials */

static Color image_pixel(const Scene* sc, int image, size_t x, size_t y, Counters* cn) {   /* img_loader.rs:69-76 */
    const Image* im = &sc->images[image];
    if (x > (size_t)(im->w - 1)) x = (size_t)(im->w - 1);
    if (y > (size_t)(im->h - 1)) y = (size_t)(im->h - 1);
    const uint8_t* p = &im->rgb[(y * (size_t)im->w + x) * 3];
    cn->texel_fetches++;
    /* img_loader.rs:37-39: pixel as f64 / 255.0 */
    return c3((real)p[0] / R(255.0), (real)p[1] / R(255.0), (real)p[2] / R(255.0));
}
static size_t as_usize(real x) {   /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(x == x) || x <= R(0.0)) return 0;
    if (x >= R(1.8446744073709552e19)) return (size_t)-1;
    return (size_t)x;
}
static int32_t as_i32(real x) {    /* Rust `as i32`: saturating, NaN -> 0 */
    if (!(x == x)) return 0;
    if (x <= R(-2147483648.0)) return INT32_MIN;
    if (x >= R(2147483647.0)) return INT32_MAX;
    return (int32_t)x;
}
/* ImageTexture::value (image_texture.rs:22-32) == SkyboxImage::get_color (scene/mod.rs:37-45) */
static Color image_lookup(const Scene* sc, int image, real u, real v, Counters* cn) {
    const Image* im = &sc->images[image];
    u = r_clamp(u, R(0.0), R(1.0));
    v = R(1.0) - r_clamp(v, R(0.0), R(1.0));
    size_t i = as_usize(u * (real)im->w);
    size_t j = as_usize(v * (real)im->h);
    return image_pixel(sc, image, i, j, cn);
}

static Color texture_value(const Scene* sc, int tex, real u, real v, Vec3 p, Counters* cn) {   /* textures/mod.rs:20-26 */
    const Texture* t = &sc->textures[tex];
    switch (t->kind) {
        case CR_TEX_SOLID: return t->color;                              /* solid_color.rs:24-28 */
        case CR_TEX_CHECKER: {                                           /* checker_texture.rs:38-51 */
            int32_t xi = as_i32(R_FLOOR(t->inv_scale * p.x));
            int32_t yi = as_i32(R_FLOOR(t->inv_scale * p.y));
            int32_t zi = as_i32(R_FLOOR(t->inv_scale * p.z));
            int32_t sum = (int32_t)((uint32_t)xi + (uint32_t)yi + (uint32_t)zi);   /* release build wraps */
            int is_even = (sum % 2) == 0;
            return texture_value(sc, is_even ? t->even : t->odd, u, v, p, cn);
        }
        default: return image_lookup(sc, t->image, u, v, cn);
    }
}

/* Materials::scatter, materials/mod.rs:23-29.  Returns 1 for Some(scattered). */
static int material_scatter(const Scene* sc, const Ray* r_in, const HitRecord* rec, Color* attenuation,
                            Ray* scattered, Rng* rng, Counters* cn) {
    const Material* m = &sc->materials[rec->mat];
    switch (m->kind) {
        case CR_MAT_LAMBERTIAN: {   /* lambertian.rs:40-61 */
            Vec3 dir = v_add(rec->normal, random_unit_vector(rng));
            if (v_near_zero(dir)) dir = rec->normal;
            scattered->origin = rec->loc; scattered->direction = dir; scattered->tm = r_in->tm;
            *attenuation = c_div(texture_value(sc, m->texture, rec->u, rec->v, rec->loc, cn), m->param);
            return rng_uniform(rng) <= m->param;
        }
        case CR_MAT_METAL: {        /* metal.rs:29-42 */
            Vec3 reflected = v_reflect(r_in->direction, rec->normal);
            reflected = v_add(v_unit(reflected), v_scale(m->param, random_unit_vector(rng)));
            scattered->origin = rec->loc; scattered->direction = reflected; scattered->tm = r_in->tm;
            *attenuation = m->albedo;
            return v_dot(scattered->direction, rec->normal) > R(0.0);
        }
        default: {                  /* dielectric.rs:30-55 */
            *attenuation = c3(R(1.0), R(1.0), R(1.0));
            real ri = rec->front_face ? R(1.0) / m->param : m->param;
            Vec3 ud = v_unit(r_in->direction);
            real cos_theta = -(R_FMIN(v_dot(ud, rec->normal), R(1.0)));   /* `-a.dot(b).min(1.0)` */
            real sin_theta = R_SQRT(R(1.0) - cos_theta * cos_theta);
            int cannot_refract = ri * sin_theta > R(1.0);
            int reflect = cannot_refract;
            if (!reflect) {   /* `||` short-circuits: the draw happens only here */
                real r0 = (R(1.0) - ri) / (R(1.0) + ri);
                r0 = r0 * r0;
                real x = R(1.0) - cos_theta;
                real x2 = x * x;
                real x5 = x * (x2 * x2);
                real reflectance = r0 + (R(1.0) - r0) * x5;   /* dielectric.rs:21-26 */
                reflect = reflectance > rng_uniform(rng);
            }
            Vec3 dir = reflect ? v_reflect(ud, rec->normal) : v_refract(ud, rec->normal, ri);
            scattered->origin = rec->loc; scattered->direction = dir; scattered->tm = r_in->tm;
            return 1;
        }
    }
}

/* ------------------------------------------------------------------ ray_color */

static Color ray_color(const Scene* sc, Ray r, uint32_t depth, Rng* rng, Counters* cn) {   /* ray_casting.rs:112-152 */
    if (depth == 0) return c3(0, 0, 0);
    Interval iv = {R(0.001), R_INF};
    HitRecord h;
    cn->segments++;
    if (hittable_hit(sc->world, &r, iv, &h, cn)) {
        Color attenuation = c3(0, 0, 0);
        Ray s;
        if (material_scatter(sc, &r, &h, &attenuation, &s, rng, cn))
            return c_mul(attenuation, ray_color(sc, s, depth - 1, rng, cn));
        return c3(0, 0, 0);
    }
    Vec3 ud = v_unit(r.direction);
    if (sc->sky_kind == CR_SKY_SPHERICAL) {
        real theta = R_ATAN2(ud.x, ud.z);
        real phi = R_ASIN(ud.y);
        real u = (theta / (R(2.0) * R_PI)) + R(0.5);
        real v = (phi / R_PI) + R(0.5);
        return image_lookup(sc, sc->sky_image, u, v, cn);
    }
    real a = R(0.5) * (ud.y + R(1.0));
    return c_add(c_scale(R(1.0) - a, c3(1, 1, 1)), c_scale(a, c3(R(0.5), R(0.7), R(1.0))));
}

/* ------------------------------------------------------------------ camera */

typedef struct {
    int W, H;
    real viewport_width, viewport_height, focus_dist, defocus_radius;
    int defocus_on;          /* defocus_angle > 0 (ray_casting.rs:97) */
    Timeline from, at;
    Vec3 vup;
    Key* key_store;
} Camera;

static Vec3 cam_from(const Camera* c, real t) { real o[4]; timeline_eval(&c->from, t, 0, o); return v3(o[0], o[1], o[2]); } /* camera/mod.rs:319-322 */
static Vec3 cam_at(const Camera* c, real t) { real o[4]; timeline_eval(&c->at, t, 0, o); return v3(o[0], o[1], o[2]); }     /* :324-327 */
/* rendering_compute.rs:78-96 */
static Vec3 w_basis(const Camera* c, real t) { return v_unit(v_sub(cam_from(c, t), cam_at(c, t))); }
static Vec3 u_basis(const Camera* c, real t) { return v_unit(v_cross(c->vup, w_basis(c, t))); }
static Vec3 v_basis(const Camera* c, real t) { return v_cross(w_basis(c, t), u_basis(c, t)); }
/* rendering_compute.rs:18-59 */
static Vec3 viewport_u(const Camera* c, real t) { return v_scale(c->viewport_width, u_basis(c, t)); }
static Vec3 viewport_v(const Camera* c, real t) { return v_scale(c->viewport_height, v_neg(v_basis(c, t))); }
static Vec3 pixel_delta_u(const Camera* c, real t) { return v_div(viewport_u(c, t), (real)c->W); }
static Vec3 pixel_delta_v(const Camera* c, real t) { return v_div(viewport_v(c, t), (real)c->H); }
static Vec3 viewport_upperleft(const Camera* c, real t) {
    Vec3 cc = cam_from(c, t);
    Vec3 a = v_sub(cc, v_scale(c->focus_dist, w_basis(c, t)));
    a = v_sub(a, v_div(viewport_u(c, t), R(2.0)));
    return v_sub(a, v_div(viewport_v(c, t), R(2.0)));
}
static Vec3 pixel_start_location(const Camera* c, real t) {
    return v_add(viewport_upperleft(c, t), v_scale(R(0.5), v_add(pixel_delta_u(c, t), pixel_delta_v(c, t))));
}
static Vec3 get_pixel_pos(const Camera* c, uint32_t i, uint32_t j, Vec3 offset, real t) {   /* :64-68 */
    Vec3 a = v_add(pixel_start_location(c, t), v_scale((real)i + offset.x, pixel_delta_u(c, t)));
    return v_add(a, v_scale((real)j + offset.y, pixel_delta_v(c, t)));
}
static Vec3 defocus_disk_sample(const Camera* c, real t, Rng* rng) {   /* :104-110 */
    Vec3 p = random_in_unit_disk(rng);
    Vec3 from = cam_from(c, t);
    Vec3 ddu = v_scale(c->defocus_radius, u_basis(c, t));   /* u_basis(t) * defocus_radius */
    Vec3 ddv = v_scale(c->defocus_radius, v_basis(c, t));
    return v_add(v_add(from, v_scale(p.x, ddu)), v_scale(p.y, ddv));
}

static void keys_to_real(const CrKeyframe* in, int n, Key* out) {
    for (int i = 0; i < n; i++) {
        out[i].channel = in[i].channel; out[i].interp = in[i].interp;
        out[i].t0 = (real)in[i].t0; out[i].t1 = (real)in[i].t1;
        out[i].a = (real)in[i].a; out[i].b = (real)in[i].b;
    }
}

static void camera_setup(Camera* c, const CrCameraDesc* d) {
    const double PI64 = 3.14159265358979323846264338327950288;
    memset(c, 0, sizeof *c);
    c->W = d->image_width; c->H = d->image_height;
    /* Radians::new_from_degrees (utils.rs:51-55), fix_viewport (rendering_compute.rs:5-11): f64 set-up */
    double vfov = d->vfov_degrees * PI64 / 180.0;
    double h = tan(vfov / 2.0);
    double vh = 2.0 * h * d->focus_dist;
    double vw = vh * ((double)d->image_width / (double)d->image_height);
    double da = d->defocus_angle_degrees * PI64 / 180.0;
    c->viewport_height = (real)vh;
    c->viewport_width = (real)vw;
    c->focus_dist = (real)d->focus_dist;
    c->defocus_on = !(da <= 0.0);
    c->defocus_radius = (real)(d->focus_dist * tan(da / 2.0));   /* rendering_compute.rs:71-73 */
    int nk = d->from_key_count + d->at_key_count;
    c->key_store = (Key*)malloc(sizeof(Key) * (size_t)(nk + 1));
    keys_to_real(d->from_keys, d->from_key_count, c->key_store);
    keys_to_real(d->at_keys, d->at_key_count, c->key_store + d->from_key_count);
    for (int k = 0; k < 3; k++) { c->from.init[k] = (real)d->look_from[k]; c->at.init[k] = (real)d->look_at[k]; }
    c->from.init[3] = R(1.0); c->at.init[3] = R(1.0);
    c->from.n_keys = d->from_key_count; c->from.keys = c->key_store;
    c->at.n_keys = d->at_key_count; c->at.keys = c->key_store + d->from_key_count;
    c->vup = v3((real)d->vup[0], (real)d->vup[1], (real)d->vup[2]);
}

/* Camera::cast_ray (ray_casting.rs:64-108) restricted to sample indices
 * [s0, s0+n): returns the running sum in draw order (average_samples :154-173 sums
 * sequentially); the caller divides. */
static void cast_ray_sum(const Scene* sc, const Camera* cam, const CrRenderParams* p, uint32_t i, uint32_t j,
                         real sum[3], Counters* cn) {
    real current_time = (real)p->frame * (R(1.0) / (real)p->frame_rate);
    real shutter_length = ((real)p->shutter_angle / R(360.0)) * (R(1.0) / (real)p->frame_rate);
    real r_tot = R(0.0), g_tot = R(0.0), b_tot = R(0.0);
    uint32_t pixel = j * (uint32_t)cam->W + i;
    for (int s = p->sample_begin; s < p->sample_begin + p->sample_count; s++) {
        Rng rng = rng_for_sample(p->seed, pixel, (uint32_t)s);
        real time_sample = current_time + rng_range(&rng, R(0.0), shutter_length);
        Vec3 cc = cam_from(cam, time_sample);
        real ox = rng_uniform(&rng) - R(0.5);            /* sample_square, camera/mod.rs:368-376 */
        real oy = rng_uniform(&rng) - R(0.5);
        Vec3 ps = get_pixel_pos(cam, i, j, v3(ox, oy, R(0.0)), time_sample);
        Vec3 orig = cam->defocus_on ? defocus_disk_sample(cam, time_sample, &rng) : cc;
        Ray ray = {orig, v_sub(ps, orig), time_sample};
        Color c = ray_color(sc, ray, (uint32_t)p->max_depth, &rng, cn);
        r_tot += c.r; g_tot += c.g; b_tot += c.b;
    }
    sum[0] = r_tot; sum[1] = g_tot; sum[2] = b_tot;
}

/* ------------------------------------------------------------------ exported API */

#define EXPORT __attribute__((visibility("default")))

EXPORT int32_t oracle_real_type(void) { return ORACLE_REAL_TYPE; }

EXPORT void oracle_scene_destroy(Scene* sc) {
    if (!sc) return;
    for (int i = 0; i < sc->n_images; i++) free(sc->images[i].rgb);
    for (int i = 0; i < sc->n_prims; i++) if (sc->prims[i].kind == H_HITLIST) free(sc->prims[i].objs);
    free(sc->images); free(sc->prims); free(sc->materials); free(sc->textures); free(sc->keys); free(sc->pool); free(sc->mat_rc);
    free(sc->empty_list.objs);
    free(sc);
}

EXPORT void oracle_set_libm(int32_t use_glibc) { g_use_libm = use_glibc != 0; }
EXPORT void oracle_set_faithful(int32_t on) { g_faithful = on != 0; }
EXPORT void oracle_trig(const real* in, int32_t n, real* atan2_out, real* asin_out, real* acos_out) {   /* in: n pairs (y, x) */
    for (int32_t i = 0; i < n; i++) {
        atan2_out[i] = R_ATAN2(in[2 * i], in[2 * i + 1]);
        asin_out[i] = R_ASIN(in[2 * i]);
        acos_out[i] = R_ACOS(in[2 * i]);
    }
}

EXPORT Scene* oracle_scene_create(const CrSceneDesc* d) {
    Scene* sc = (Scene*)calloc(1, sizeof *sc);
    sc->n_prims = d->n_prims; sc->n_materials = d->n_materials; sc->n_textures = d->n_textures;
    sc->n_images = d->n_images; sc->n_keys = d->n_keys;
    sc->sky_kind = d->sky_kind; sc->sky_image = d->sky_image;
    sc->keys = (Key*)malloc(sizeof(Key) * (size_t)(d->n_keys + 1));
    keys_to_real(d->keys, d->n_keys, sc->keys);
    sc->materials = (Material*)malloc(sizeof(Material) * (size_t)(d->n_materials + 1));
    for (int i = 0; i < d->n_materials; i++) {
        const CrMaterial* m = &d->materials[i];
        sc->materials[i].kind = m->kind; sc->materials[i].texture = m->texture;
        sc->materials[i].albedo = c3((real)m->albedo[0], (real)m->albedo[1], (real)m->albedo[2]);
        sc->materials[i].param = (real)m->param;
    }
    sc->mat_rc = (int*)calloc((size_t)d->n_materials + 1, sizeof(int));
    sc->textures = (Texture*)malloc(sizeof(Texture) * (size_t)(d->n_textures + 1));
    for (int i = 0; i < d->n_textures; i++) {
        const CrTexture* t = &d->textures[i];
        sc->textures[i].kind = t->kind; sc->textures[i].even = t->even; sc->textures[i].odd = t->odd;
        sc->textures[i].image = t->image;
        sc->textures[i].color = c3((real)t->color[0], (real)t->color[1], (real)t->color[2]);
        sc->textures[i].inv_scale = (real)t->inv_scale;
    }
    sc->images = (Image*)calloc((size_t)(d->n_images + 1), sizeof(Image));
    for (int i = 0; i < d->n_images; i++) {
        size_t n = (size_t)d->images[i].width * (size_t)d->images[i].height * 3;
        sc->images[i].w = d->images[i].width; sc->images[i].h = d->images[i].height;
        sc->images[i].rgb = (uint8_t*)malloc(n);
        memcpy(sc->images[i].rgb, d->images[i].rgb8, n);
    }
    sc->prims = (Hittable*)calloc((size_t)(d->n_prims + 1), sizeof(Hittable));
    for (int i = 0; i < d->n_prims; i++) {
        const CrPrimitive* p = &d->prims[i];
        Hittable* h = &sc->prims[i];
        h->mat = p->material; h->hide = (p->flags & CR_PRIM_HIDDEN) != 0;
        h->member = (p->flags & CR_PRIM_MEMBER) != 0;
        if (p->kind == CR_PRIM_LIST || p->kind == CR_PRIM_BVH) { h->kind = H_HITLIST; h->hide = 0; continue; }   /* second pass below */
        h->tl.n_keys = p->key_count; h->tl.keys = sc->keys + p->key_first;
        if (p->kind == CR_PRIM_SPHERE) {
            h->kind = H_SPHERE;
            for (int k = 0; k < 4; k++) h->tl.init[k] = (real)p->v[k];
            h->bbox = sphere_bbox(v3(h->tl.init[0], h->tl.init[1], h->tl.init[2]), h->tl.init[3]);
        } else {
            h->kind = H_TRIANGLE;
            for (int k = 0; k < 3; k++) { h->tl.init[k] = (real)p->v[k]; h->vb[k] = (real)p->v[3 + k]; h->vc[k] = (real)p->v[6 + k]; }
            h->tl.init[3] = R(1.0);
            h->bbox = triangle_bbox(v3(h->tl.init[0], h->tl.init[1], h->tl.init[2]), v3(h->vb[0], h->vb[1], h->vb[2]),
                                    v3(h->vc[0], h->vc[1], h->vc[2]));
        }
    }
    for (int i = 0; i < d->n_prims; i++) {   /* BVHWrapper elements (crucible_hip.h CR_PRIM_BVH): BVHWrapper::new_wrapper over the objects */
        const CrPrimitive* p = &d->prims[i];
        if (p->kind != CR_PRIM_BVH) continue;
        const int first = (int)p->v[0], count = (int)p->v[1];
        if (!sc->pool) {
            sc->pool_cap = 2 * d->n_prims + 8;
            sc->pool = (Hittable*)malloc(sizeof(Hittable) * (size_t)sc->pool_cap);
            sc->pool_used = 0;
        }
        Hittable** vis = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(count + 1));
        Hittable** tmp = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(count + 1));
        int n_vis = 0;
        for (int k = 0; k < count; k++) if (!sc->prims[first + k].hide) vis[n_vis++] = &sc->prims[first + k];   /* bvhwrapper.rs:16-26 */
        Hittable* h = &sc->prims[i];
        if (n_vis == 0) { h->kind = H_HITLIST; h->n_objs = 0; h->objs = NULL; h->bbox = aabb_empty(); }   /* :28-30 */
        else {
            Hittable* root = bvh_generate(sc, vis, tmp, 0, n_vis);
            root->bbox = aabb_from_boxes(root->left->bbox, root->right->bbox);   /* new_from_vec :39 */
            int member = h->member;
            *h = *root;            /* the element itself is the root wrapper */
            h->member = member;
        }
        free(vis); free(tmp);
    }
    for (int i = 0; i < d->n_prims; i++) {   /* HitList elements (crucible_hip.h CR_PRIM_LIST) */
        const CrPrimitive* p = &d->prims[i];
        if (p->kind != CR_PRIM_LIST) continue;
        Hittable* h = &sc->prims[i];
        const int first = (int)p->v[0], count = (int)p->v[1];
        h->objs = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(count + 1));
        h->n_objs = count;
        h->bbox = aabb_empty();   /* HitList::new: Aabb::default(), hitlist.rs:13-18 */
        for (int k = 0; k < count; k++) {
            h->objs[k] = &sc->prims[first + k];
            if (!(p->flags & CR_LIST_EMPTY_BOX)) h->bbox = aabb_from_boxes(h->bbox, h->objs[k]->bbox);   /* HitList::add, hitlist.rs:24-27 */
        }
    }
    scene_build_world(sc);
    return sc;
}

/* ------------------------------------------------------------------ refit_boxes (not in the reference)
 * CrRenderParams.refit_boxes = 1, the rule of crucible_amd/csrc/refit.hpp restated: a primitive's box over the
 * frame's ray times [ta, tb] is the union of its construction-rule box (sphere_bbox / triangle_bbox) at ta, tb,
 * every key end inside (ta, tb) and every key start inside (ta, tb] with the key active and not yet active; a
 * wrapper's box is aabb_from_boxes of its children.  The reference never recomputes wrapper boxes
 * (bvhwrapper.rs:47-50,102-106), so this mode is pinned by construction only: against the linear list
 * (oracle_use_list) it must give the same closest hits. */
/* the primitive's box with its translate part taken at (t, side) and its scale part at (ts, side_s) */
static Aabb prim_box_at2(const Hittable* h, real t, int side, real ts, int side_s) {
    real p[3], p2[3], w, w2;
    int sk, sk2;
    if (h->kind == H_SPHERE) {
        timeline_parts(&h->tl, t, side, p, &w, &sk);
        return sphere_bbox(v3(p[0], p[1], p[2]), w);
    }
    Timeline tl[3] = {h->tl, h->tl, h->tl};
    tl[1].init[0] = h->vb[0]; tl[1].init[1] = h->vb[1]; tl[1].init[2] = h->vb[2];
    tl[2].init[0] = h->vc[0]; tl[2].init[1] = h->vc[1]; tl[2].init[2] = h->vc[2];
    Vec3 v[3];
    for (int j = 0; j < 3; j++) {
        real o[4];
        timeline_parts(&tl[j], t, side, p, &w, &sk);
        timeline_parts(&tl[j], ts, side_s, p2, &w2, &sk2);
        scale_apply(sk2, w2, p[0], p[1], p[2], o);
        v[j] = v3(o[0], o[1], o[2]);
    }
    return triangle_bbox(v[0], v[1], v[2]);
}
static Aabb prim_box_at(const Hittable* h, real t, int before_start) { return prim_box_at2(h, t, before_start, t, before_start); }
/* sample i of the refit rule: 0 = ta, 1 = tb, then per key its start (active / not yet active) and its end */
static int refit_sample(const Hittable* h, real ta, real tb, int i, real* t, int* side) {
    *side = 0;
    if (i == 0) { *t = ta; return 1; }
    if (i == 1) { *t = tb; return 1; }
    const Key* k = &h->tl.keys[(i - 2) / 3];
    int which = (i - 2) % 3;
    if (which < 2) { *t = k->t0; *side = which == 1; return ta < k->t0 && k->t0 <= tb; }
    *t = k->t1;
    return ta < k->t1 && k->t1 < tb;
}
/* a sample box with a coordinate that is not a number is not united (refit.hpp box_is_number) */
static Aabb unite_sample(Aabb acc, Aabb s) {
    if (!(s.x.min == s.x.min && s.x.max == s.x.max && s.y.min == s.y.min && s.y.max == s.y.max && s.z.min == s.z.min && s.z.max == s.z.max)) return acc;
    return aabb_from_boxes(acc, s);
}
static Aabb prim_box_over(const Hittable* h, real ta, real tb) {
    if (h->kind == H_HITLIST) {   /* the visible objects' boxes, united in the list's order */
        Aabb l = aabb_empty();
        for (int i = 0; i < h->n_objs; i++) if (!h->objs[i]->hide) l = aabb_from_boxes(l, prim_box_over(h->objs[i], ta, tb));
        return l;
    }
    Aabb b = unite_sample(aabb_empty(), prim_box_at(h, ta, 0));
    if (h->tl.n_keys == 0) return b;
    int scaled = 0;
    for (int i = 0; i < h->tl.n_keys; i++) scaled |= h->tl.keys[i].channel >= CR_KEY_SCALE_X;
    if (scaled) {   /* translate and scale parts sampled independently, every pair united (refit.hpp) */
        int n = 2 + 3 * h->tl.n_keys;
        for (int i = 0; i < n; i++) {
            real t1, t2; int s1, s2;
            if (!refit_sample(h, ta, tb, i, &t1, &s1)) continue;
            for (int j = 0; j < n; j++)
                if (refit_sample(h, ta, tb, j, &t2, &s2)) b = unite_sample(b, prim_box_at2(h, t1, s1, t2, s2));
        }
        return b;
    }
    b = unite_sample(b, prim_box_at(h, tb, 0));
    for (int i = 0; i < h->tl.n_keys; i++) {
        const Key* k = &h->tl.keys[i];
        if (ta < k->t0 && k->t0 <= tb) {
            b = unite_sample(b, prim_box_at(h, k->t0, 0));
            b = unite_sample(b, prim_box_at(h, k->t0, 1));
        }
        if (ta < k->t1 && k->t1 < tb) b = unite_sample(b, prim_box_at(h, k->t1, 0));
    }
    return b;
}
static Aabb refit_rec(Hittable* h, real ta, real tb) {
    if (h->kind != H_BVH) return prim_box_over(h, ta, tb);
    Aabb l = refit_rec(h->left, ta, tb);
    Aabb r = h->right == h->left ? l : refit_rec(h->right, ta, tb);
    h->bbox = aabb_from_boxes(l, r);
    return h->bbox;
}
/* choose the boxes this render walks: refitted to [ta, tb], or the construction-time ones */
static void scene_prepare_boxes(Scene* sc, int refit, real ta, real tb) {
    if (sc->world->kind != H_BVH) return;
    for (int i = 0; i < sc->pool_used; i++) sc->pool[i].bbox = sc->pool[i].bbox0;
    for (int i = 0; i < sc->n_prims; i++) if (sc->prims[i].kind == H_BVH) sc->prims[i].bbox = sc->prims[i].bbox0;
    if (refit) (void)refit_rec(sc->world, ta, tb);
}

typedef struct {
    const Scene* sc; const Camera* cam; const CrRenderParams* p;
    real* out; int64_t pix_begin, pix_end;
    volatile int64_t* next;
    Counters cn; uint64_t nan_pixels;
} Job;

static pthread_mutex_t g_receiver = PTHREAD_MUTEX_INITIALIZER;
static void* worker(void* arg) {   /* one pixel per work item, cpu_threading.rs:85-101 */
    Job* jb = (Job*)arg;
    const int W = jb->cam->W;
    void* world_clone = NULL;
    if (g_faithful) {   /* `let mut world = world.clone()` per thread, cpu_threading.rs:44: a deep copy of every primitive */
        size_t bytes = sizeof(Hittable) * (size_t)(jb->sc->n_prims + 1) + sizeof(Key) * (size_t)(jb->sc->n_keys + 1);
        world_clone = malloc(bytes);
        memcpy(world_clone, jb->sc->prims, sizeof(Hittable) * (size_t)(jb->sc->n_prims + 1));
        memcpy((char*)world_clone + sizeof(Hittable) * (size_t)(jb->sc->n_prims + 1), jb->sc->keys, sizeof(Key) * (size_t)(jb->sc->n_keys + 1));
    }
    for (;;) {
        int64_t pix;
        if (g_faithful) {   /* receiver.lock().unwrap().recv(), cpu_threading.rs:88 */
            pthread_mutex_lock(&g_receiver);
            pix = *jb->next; *jb->next = pix + 1;
            pthread_mutex_unlock(&g_receiver);
        } else pix = __sync_fetch_and_add(jb->next, 1);
        if (pix >= jb->pix_end) break;
        uint32_t i = (uint32_t)(pix % W), j = (uint32_t)(pix / W);
        real sum[3];
        cast_ray_sum(jb->sc, jb->cam, jb->p, i, j, sum, &jb->cn);
        real* o = jb->out + (pix - jb->pix_begin) * 3;
        if (jb->p->output_sum) { o[0] = sum[0]; o[1] = sum[1]; o[2] = sum[2]; }
        else {
            real cnt = (real)jb->p->samples;   /* `/= sample_count as f64`, ray_casting.rs:168-170 */
            o[0] = sum[0] / cnt; o[1] = sum[1] / cnt; o[2] = sum[2] / cnt;
            for (int k = 0; k < 3; k++) if (!(o[k] >= R(0.0) && o[k] <= R(1.0))) { jb->nan_pixels++; break; }   /* Color::new asserts */
        }
    }
    free(world_clone);
    return NULL;
}

/* Renders pixels [pix_begin, pix_end) in row-major order (pix = j*W + i) into
 * out[(pix-pix_begin)*3..].  n_threads OS threads pull pixels from one counter. */
EXPORT int32_t oracle_render(const Scene* sc, const CrCameraDesc* cd, const CrRenderParams* p, int64_t pix_begin,
                             int64_t pix_end, real* out, int32_t n_threads, CrStats* stats) {
    if (p->real_type != ORACLE_REAL_TYPE) return CR_ERR_INVALID_ARG;
    Camera cam;
    camera_setup(&cam, cd);
    if (n_threads < 1) n_threads = 1;
    {   /* the boxes of this frame (one render at a time per scene, as the library's handle) */
        real current_time = (real)p->frame * (R(1.0) / (real)p->frame_rate);
        real shutter_length = ((real)p->shutter_angle / R(360.0)) * (R(1.0) / (real)p->frame_rate);
        scene_prepare_boxes((Scene*)sc, p->refit_boxes, current_time, current_time + shutter_length);
    }
    volatile int64_t next = pix_begin;
    g_mat_rc = sc->mat_rc;
    Job* jobs = (Job*)calloc((size_t)n_threads, sizeof(Job));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].sc = sc; jobs[t].cam = &cam; jobs[t].p = p; jobs[t].out = out;
        jobs[t].pix_begin = pix_begin; jobs[t].pix_end = pix_end; jobs[t].next = &next;
        if (n_threads > 1) pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (n_threads == 1) worker(&jobs[0]);
    else for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int t = 0; t < n_threads; t++) {
            stats->segments += jobs[t].cn.segments; stats->node_tests += jobs[t].cn.node_tests;
            stats->prim_tests += jobs[t].cn.prim_tests_dedup; stats->texel_fetches += jobs[t].cn.texel_fetches;
            stats->nan_pixels += jobs[t].nan_pixels;
        }
        stats->samples = (uint64_t)(pix_end - pix_begin) * (uint64_t)p->sample_count;
        stats->bvh_entries = sc->pool_used;
    }
    free(jobs); free(th); free(cam.key_store);
    return CR_OK;
}

/* ---- probes for unit tests / fixtures: one reference function each ---- */

EXPORT real oracle_dot(const real* a, const real* b) { return v_dot(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
EXPORT void oracle_cross(const real* a, const real* b, real* o) { Vec3 c = v_cross(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT real oracle_length(const real* a) { return v_len(v3(a[0], a[1], a[2])); }
EXPORT void oracle_neg(const real* a, real* o) { Vec3 c = v_neg(v3(a[0], a[1], a[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT void oracle_add(const real* a, const real* b, real* o) { Vec3 c = v_add(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT void oracle_unit(const real* a, real* o) { Vec3 c = v_unit(v3(a[0], a[1], a[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT void oracle_reflect(const real* v, const real* n, real* o) { Vec3 c = v_reflect(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT void oracle_refract(const real* v, const real* n, real eta, real* o) { Vec3 c = v_refract(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), eta); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
EXPORT void oracle_ray_at(const real* orig, const real* dir, real t, real* o) { Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), 0}; Vec3 c = ray_at(&r, t); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
/* Color::new validity (utils.rs:345-350): 1 if it would construct, 0 if it would panic */
EXPORT int32_t oracle_color_valid(real r, real g, real b) { return r <= 1 && g <= 1 && b <= 1 && r >= 0 && g >= 0 && b >= 0; }
EXPORT void oracle_color_display(double r, double g, double b, uint32_t* o) { o[0] = display_byte(r); o[1] = display_byte(g); o[2] = display_byte(b); }
EXPORT void oracle_color_neg(const real* c, real* o) { Color n = c_neg(c3(c[0], c[1], c[2])); o[0] = n.r; o[1] = n.g; o[2] = n.b; }
EXPORT void oracle_color_add(const real* a, const real* b, real* o) { Color n = c_add(c3(a[0], a[1], a[2]), c3(b[0], b[1], b[2])); o[0] = n.r; o[1] = n.g; o[2] = n.b; }
EXPORT void oracle_color_mul(const real* a, const real* b, real* o) { Color n = c_mul(c3(a[0], a[1], a[2]), c3(b[0], b[1], b[2])); o[0] = n.r; o[1] = n.g; o[2] = n.b; }
EXPORT void oracle_color_scale(real s, const real* c, real* o) { Color n = c_scale(s, c3(c[0], c[1], c[2])); o[0] = n.r; o[1] = n.g; o[2] = n.b; }
EXPORT void oracle_color_div(const real* c, real s, real* o) { Color n = c_div(c3(c[0], c[1], c[2]), s); o[0] = n.r; o[1] = n.g; o[2] = n.b; }
/* average_samples, ray_casting.rs:154-173 */
EXPORT void oracle_average_samples(const real* rgb, int32_t n, real* o) {
    real r = 0, g = 0, b = 0;
    for (int i = 0; i < n; i++) { r += rgb[3 * i]; g += rgb[3 * i + 1]; b += rgb[3 * i + 2]; }
    o[0] = r / (real)n; o[1] = g / (real)n; o[2] = b / (real)n;
}
EXPORT real oracle_degrees_to_radians(real d) { return d * R_PI / R(180.0); }   /* utils.rs:27-31 */
EXPORT real oracle_radians_to_degrees(real r) { return r * R(180.0) / R_PI; }   /* utils.rs:58-62 */
EXPORT real oracle_interval_size(real lo, real hi) { return hi - lo; }
EXPORT int32_t oracle_interval_contains(real lo, real hi, real x) { Interval i = {lo, hi}; return iv_contains(i, x); }
EXPORT int32_t oracle_interval_surrounds(real lo, real hi, real x) { Interval i = {lo, hi}; return iv_surrounds(i, x); }
EXPORT int32_t oracle_interval_is_greater(real lo, real hi, real x) { Interval i = {lo, hi}; return iv_is_greater(i, x); }
EXPORT int32_t oracle_interval_is_less(real lo, real hi, real x) { Interval i = {lo, hi}; return iv_is_less(i, x); }
EXPORT real oracle_interval_proportion(real lo, real hi, real x) { Interval i = {lo, hi}; return iv_proportion(i, x); }

/* combine_and_compute on an explicit key list (timeline/mod.rs:233-263) */
EXPORT void oracle_timeline_eval(const real* init4, const CrKeyframe* keys, int32_t n, int32_t is_sphere, real t, real* out4) {
    Key* k = (Key*)malloc(sizeof(Key) * (size_t)(n + 1));
    keys_to_real(keys, n, k);
    Timeline tl; memcpy(tl.init, init4, sizeof tl.init); tl.n_keys = n; tl.keys = k;
    timeline_eval(&tl, t, is_sphere, out4);
    free(k);
}

EXPORT int32_t oracle_aabb_hit(const real* box6, const real* orig, const real* dir, real tmin, real tmax) {
    Aabb b = {{box6[0], box6[1]}, {box6[2], box6[3]}, {box6[4], box6[5]}};
    Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), 0};
    Interval iv = {tmin, tmax};
    return aabb_hit(&b, &r, iv);
}

/* out: t, loc(3), normal(3), u, v, front_face  (10 reals) */
static void rec_out(const HitRecord* h, real* o) {
    o[0] = h->t; o[1] = h->loc.x; o[2] = h->loc.y; o[3] = h->loc.z; o[4] = h->normal.x; o[5] = h->normal.y; o[6] = h->normal.z;
    o[7] = h->u; o[8] = h->v; o[9] = (real)h->front_face;
}
EXPORT int32_t oracle_sphere_hit(const real* center_radius, const real* orig, const real* dir, real tmin, real tmax, real* out10) {
    Hittable s; memset(&s, 0, sizeof s); s.kind = H_SPHERE;
    memcpy(s.tl.init, center_radius, 4 * sizeof(real));
    Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), 0};
    Interval iv = {tmin, tmax}; HitRecord h;
    if (!sphere_hit(&s, &r, iv, &h)) return 0;
    rec_out(&h, out10); return 1;
}
EXPORT int32_t oracle_triangle_hit(const real* abc9, const real* orig, const real* dir, real tmin, real tmax, real* out10) {
    Hittable t; memset(&t, 0, sizeof t); t.kind = H_TRIANGLE;
    for (int k = 0; k < 3; k++) { t.tl.init[k] = abc9[k]; t.vb[k] = abc9[3 + k]; t.vc[k] = abc9[6 + k]; }
    t.tl.init[3] = R(1.0);
    Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), 0};
    Interval iv = {tmin, tmax}; HitRecord h;
    if (!triangle_hit(&t, &r, iv, &h)) return 0;
    rec_out(&h, out10); return 1;
}

/* closest hit through the scene's BVH; out10 as above, *prim = index in the desc's list */
EXPORT int32_t oracle_world_hit(Scene* sc, const real* orig, const real* dir, real tm, real tmin, real tmax, real* out10, int32_t* mat) {
    Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), tm};
    Interval iv = {tmin, tmax}; HitRecord h; Counters cn; memset(&cn, 0, sizeof cn);
    if (!hittable_hit(sc->world, &r, iv, &h, &cn)) return 0;
    rec_out(&h, out10); *mat = h.mat; return 1;
}

/* scatter with an explicit RNG key; out: attenuation(3), scattered origin(3), dir(3), draws used */
EXPORT int32_t oracle_scatter(const Scene* sc, int32_t mat, const real* orig, const real* dir, const real* rec10,
                              uint64_t seed, uint32_t pixel, uint32_t sample, real* out10) {
    Ray r = {v3(orig[0], orig[1], orig[2]), v3(dir[0], dir[1], dir[2]), 0};
    HitRecord h; h.t = rec10[0]; h.loc = v3(rec10[1], rec10[2], rec10[3]); h.normal = v3(rec10[4], rec10[5], rec10[6]);
    h.u = rec10[7]; h.v = rec10[8]; h.front_face = rec10[9] != 0; h.mat = mat;
    Rng rng = rng_for_sample(seed, pixel, sample); Counters cn; memset(&cn, 0, sizeof cn);
    Color att = c3(0, 0, 0); Ray s; memset(&s, 0, sizeof s);
    int some = material_scatter(sc, &r, &h, &att, &s, &rng, &cn);
    out10[0] = att.r; out10[1] = att.g; out10[2] = att.b;
    out10[3] = s.origin.x; out10[4] = s.origin.y; out10[5] = s.origin.z;
    out10[6] = s.direction.x; out10[7] = s.direction.y; out10[8] = s.direction.z; out10[9] = (real)rng.draws;
    return some;
}

EXPORT void oracle_texture_value(const Scene* sc, int32_t tex, real u, real v, const real* p, real* out3) {
    Counters cn; memset(&cn, 0, sizeof cn);
    Color c = texture_value(sc, tex, u, v, v3(p[0], p[1], p[2]), &cn);
    out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}

/* sky colour for a ray that missed everything (ray_casting.rs:133-151) */
EXPORT void oracle_sky(const Scene* sc, const real* dir, real* out3) {
    Scene tmp = *sc; Hittable empty; memset(&empty, 0, sizeof empty); empty.kind = H_HITLIST; tmp.world = &empty;
    Ray r = {v3(0, 0, 0), v3(dir[0], dir[1], dir[2]), 0};
    Rng rng = rng_for_sample(0, 0, 0); Counters cn; memset(&cn, 0, sizeof cn);
    Color c = ray_color(&tmp, r, 1, &rng, &cn);
    out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}

/* primary ray of (pixel i,j, sample s): out = origin(3), dir(3), time, draws used */
EXPORT void oracle_camera_ray(const CrCameraDesc* cd, const CrRenderParams* p, uint32_t i, uint32_t j, uint32_t s, real* out8) {
    Camera cam; camera_setup(&cam, cd);
    real current_time = (real)p->frame * (R(1.0) / (real)p->frame_rate);
    real shutter_length = ((real)p->shutter_angle / R(360.0)) * (R(1.0) / (real)p->frame_rate);
    Rng rng = rng_for_sample(p->seed, j * (uint32_t)cam.W + i, s);
    real ts = current_time + rng_range(&rng, R(0.0), shutter_length);
    Vec3 cc = cam_from(&cam, ts);
    real ox = rng_uniform(&rng) - R(0.5), oy = rng_uniform(&rng) - R(0.5);
    Vec3 ps = get_pixel_pos(&cam, i, j, v3(ox, oy, R(0.0)), ts);
    Vec3 orig = cam.defocus_on ? defocus_disk_sample(&cam, ts, &rng) : cc;
    Vec3 d = v_sub(ps, orig);
    out8[0] = orig.x; out8[1] = orig.y; out8[2] = orig.z; out8[3] = d.x; out8[4] = d.y; out8[5] = d.z; out8[6] = ts; out8[7] = (real)rng.draws;
    free(cam.key_store);
}

/* the RNG stream itself: first n uniforms of (seed, pixel, sample) */
EXPORT void oracle_rng_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, int32_t n, real* out) {
    Rng r = rng_for_sample(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = rng_uniform(&r);
}
EXPORT void oracle_rng_u64(uint64_t seed, uint32_t pixel, uint32_t sample, int32_t n, uint64_t* out) {
    Rng r = rng_for_sample(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = rng_u64(&r);
}

/* BVH dump in DFS pre-order for cross-checking the library's builder:
 * per wrapper: 6 reals (xmin,xmax,ymin,ymax,zmin,zmax) and the kind of
 * (left,right): -1 wrapper, else index of the primitive in the desc list. */
static int dump_rec(const Scene* sc, const Hittable* h, real* boxes, int32_t* kids, int cap, int n) {
    if (h->kind != H_BVH) return n;
    if (n < cap) {
        real* b = boxes + 6 * n;
        b[0] = h->bbox.x.min; b[1] = h->bbox.x.max; b[2] = h->bbox.y.min; b[3] = h->bbox.y.max; b[4] = h->bbox.z.min; b[5] = h->bbox.z.max;
        kids[2 * n] = h->left->kind == H_BVH ? -1 : (int32_t)(h->left - sc->prims);
        kids[2 * n + 1] = h->right->kind == H_BVH ? -1 : (int32_t)(h->right - sc->prims);
    }
    n++;
    n = dump_rec(sc, h->left, boxes, kids, cap, n);
    if (h->right != h->left) n = dump_rec(sc, h->right, boxes, kids, cap, n);
    return n;
}
/* Replace the scene's wrapper tree by a given one (the shape cr_export_bvh writes: per wrapper 6 doubles
 * xmin,xmax,ymin,ymax,zmin,zmax and (left,right): >= 0 wrapper index, < 0 ~index of a primitive in the
 * desc list; wrapper 0 is the root; split_axis NULL or per wrapper -1 | the axis of CR_BVH_SAH_ORDERED's
 * near-child-first rule).  The walk stays BVHWrapper::hit (bvhwrapper.rs:96-126) -- with split_axis, except for
 * which child goes first; only the topology and boxes are the caller's.  Used to check the library's CR_BVH_SAH mode: the reference builds
 * no such tree, so that mode is pinned only by this walk.  Returns 0, or -1 on a malformed tree. */
EXPORT int32_t oracle_set_tree(Scene* sc, const double* boxes, const int32_t* kids, const int32_t* split_axis, int32_t n) {
    if (n <= 0) return -1;
    for (int i = 0; i < 2 * n; i++) {
        if (kids[i] >= n || (kids[i] >= 0 && kids[i] <= i / 2)) return -1;   /* children come after their parent */
        if (kids[i] < 0 && ~kids[i] >= sc->n_prims) return -1;
    }
    Hittable* pool = (Hittable*)calloc((size_t)n, sizeof(Hittable));
    for (int i = 0; i < n; i++) {
        Hittable* w = &pool[i];
        const double* b = boxes + 6 * i;
        w->kind = H_BVH;
        w->bbox.x.min = (real)b[0]; w->bbox.x.max = (real)b[1];
        w->bbox.y.min = (real)b[2]; w->bbox.y.max = (real)b[3];
        w->bbox.z.min = (real)b[4]; w->bbox.z.max = (real)b[5];
        w->bbox0 = w->bbox;
        w->near_axis = (split_axis && split_axis[i] >= 0 && split_axis[i] <= 2) ? split_axis[i] + 1 : 0;
        w->left = kids[2 * i] >= 0 ? &pool[kids[2 * i]] : &sc->prims[~kids[2 * i]];
        w->right = kids[2 * i + 1] >= 0 ? &pool[kids[2 * i + 1]] : &sc->prims[~kids[2 * i + 1]];
    }
    free(sc->pool);
    sc->pool = pool; sc->pool_used = n; sc->pool_cap = n;
    sc->world = &pool[0];
    return 0;
}

/* Replace the world by the flat list of visible primitives (HitList::hit, hitlist.rs:51-65: a linear closest-hit
 * scan with no boxes at all): the ground truth a refitted tree must agree with. */
EXPORT void oracle_use_list(Scene* sc) {
    Hittable* l = &sc->empty_list;
    free(l->objs);
    memset(l, 0, sizeof *l);
    l->kind = H_HITLIST;
    l->objs = (Hittable**)malloc(sizeof(Hittable*) * (size_t)(sc->n_prims + 1));
    for (int i = 0; i < sc->n_prims; i++)   /* a list element's objects are in prims themselves */
        if (sc->prims[i].kind != H_HITLIST && sc->prims[i].kind != H_BVH && !sc->prims[i].hide) l->objs[l->n_objs++] = &sc->prims[i];
    sc->world = l;
}

EXPORT int32_t oracle_bvh_dump(const Scene* sc, real* boxes, int32_t* kids, int32_t cap) {
    return dump_rec(sc, sc->world, boxes, kids, cap, 0);
}
