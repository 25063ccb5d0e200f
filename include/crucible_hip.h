/*
 * crucible_hip.h -- C ABI of the MI355X (gfx950) path-tracing integrator that
 * stands in for Crucible's per-pixel render loop.
 *
 * The reference (kylittle/Crucible, Rust) has no FFI.  The seam this ABI
 * replaces is
 *
 *     pub fn Camera::render(&mut self, skybox: &Skybox, world: &Hittables,
 *                           fname: &str) -> Result<(), std::io::Error>
 *                                              (src/camera/mod.rs:270-317)
 *
 * called only from Scene::render_image (src/scene/mod.rs:332-347) right after
 * BVHWrapper::new_wrapper(self.elements.clone()) (src/scene/mod.rs:333).
 * Everything under that call -- cast_ray / ray_color / Hittables::hit /
 * Materials::scatter / Textures::value / the sky lookup -- runs on the GPU
 * behind the entry points below.  Scene building, asset decoding and the PPM
 * text output stay on the host side of the boundary.
 *
 * Rust-callable by construction: #[repr(C)] PODs, caller-owned memory, int32
 * status codes, no unwinding, no global state besides the opaque handle.  The
 * binding a Crucible maintainer would add is shown in INTEGRATION.md.
 *
 * All scene numbers cross the boundary as f64 (the reference's scalar type,
 * src/utils.rs:72-74).  The library computes either in f64 (CR_REAL_F64,
 * arithmetic twin of the reference) or in f32 (CR_REAL_F32, inputs rounded to
 * f32 once at upload).
 */
#ifndef CRUCIBLE_HIP_H
#define CRUCIBLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CR_ABI_VERSION 3

#if defined(__GNUC__)
#define CR_API __attribute__((visibility("default")))
#else
#define CR_API
#endif

/* ---- status codes (the reference panics or returns io::Error instead) ---- */
enum {
    CR_OK = 0,
    CR_ERR_INVALID_ARG = 1,   /* reference: assert!/panic! in constructors          */
    CR_ERR_NO_DEVICE = 2,     /* no HIP device / HIP runtime error at create         */
    CR_ERR_HIP = 3,           /* any other HIP runtime failure (see cr_last_error)   */
    CR_ERR_NO_SCENE = 4,      /* render before upload                                 */
    CR_ERR_IO = 5,            /* file open/write failed (reference: io::Error)        */
    CR_ERR_NAN = 6,           /* a pixel mean is NaN / outside [0,1]
                                 (reference: Color::new assert, src/utils.rs:345-350) */
    CR_ERR_UNSUPPORTED = 7,
    CR_ERR_PEER = 8           /* cr_group_*: another member of the group failed its render (every rank returns this or
                                 its own error, none is left waiting in the collective), or an earlier collective of
                                 this group failed and the group must be destroyed                               */
};

/* ---- scalar type the path computes in ---- */
enum { CR_REAL_F32 = 0, CR_REAL_F64 = 1 };

/* ---- scene elements: Hittables::{Sphere,Triangle,HitList} (src/objects/mod.rs:109-115) ---- */
enum { CR_PRIM_SPHERE = 0, CR_PRIM_TRIANGLE = 1, CR_PRIM_LIST = 2, CR_PRIM_BVH = 3 };
enum {
    CR_PRIM_HIDDEN = 1,      /* Sphere.hide / Triangle.hide (sphere.rs:18, triangle.rs:11)                       */
    CR_PRIM_MEMBER = 2,      /* this sphere/triangle is an object of a CR_PRIM_LIST record, not a scene element  */
    CR_LIST_EMPTY_BOX = 4    /* list built by HitList::new(vec): its box stays Aabb::default() (hitlist.rs:13-18);
                                without the flag the box is what HitList::add accumulates (hitlist.rs:24-27)     */
};

/*
 * One element of the flat scene list (Scene.elements, src/scene/mod.rs:77), in
 * list order -- the order matters because the BVH build stable-sorts it
 * (src/objects/bvhwrapper.rs:66-67).
 *   sphere   : v[0..2] = centre, v[3] = radius          (sphere.rs:26)
 *   triangle : v[0..2] = a, v[3..5] = b, v[6..8] = c    (triangle.rs:24)
 *   list     : a HitList handed to Scene::add_element (scene/mod.rs:164-166): v[0] = index of its first object
 *              in prims, v[1] = number of objects (whole numbers).  The objects are consecutive spheres/triangles
 *              flagged CR_PRIM_MEMBER, in the list's order; each belongs to exactly one list.  The BVH build
 *              treats the list as ONE object (bvhwrapper.rs:18-22 keeps it whatever it holds): a leaf wrapper
 *              that holds it walks every object in order with the shrinking interval and no box test
 *              (HitList::hit, hitlist.rs:51-65); hidden objects return no hit (sphere.rs:62, triangle.rs:87)
 *              but still count towards an add()-built box.  A list inside a list behaves exactly like its
 *              objects spliced in place (the inner box is never read), which is how the host mirrors pass it;
 *   bvh      : a BVHWrapper handed to Scene::add_element (scene/mod.rs:161-163), i.e. the result of
 *              BVHWrapper::new_wrapper(list) (bvhwrapper.rs:15-32): v[0], v[1] name its objects like a list's (spheres
 *              and triangles flagged CR_PRIM_MEMBER; hidden ones are dropped, as new_wrapper drops them).  The library
 *              rebuilds the inner tree with the reference's own algorithm (it is a function of the object list), the
 *              outer build sorts the wrapper by its root box, and BVHWrapper::hit walks into it as into any wrapper.
 *              cr_export_bvh answers CR_ERR_UNSUPPORTED for such a scene in CR_BVH_REFERENCE mode (a wrapper holding
 *              one primitive and one sub-tree has no two-children form in the exported layout).  A wrapper without
 *              a visible object is the empty list new_wrapper returns (bvhwrapper.rs:28-30).  Lists or wrappers
 *              inside a wrapper are not representable (CR_ERR_UNSUPPORTED at the mirrors).
 *              Under the opt-in CR_BVH_SAH / _ORDERED / LBVH trees a list's visible objects are ordinary
 *              primitives of the tree (a bvh record's likewise).  material, key_first, key_count are unused for both.
 * key_first/key_count select this primitive's keyframes in CrSceneDesc.keys
 * (0 keys = static; the initial transform is the v[] values themselves).
 */
typedef struct CrPrimitive {
    int32_t kind;
    int32_t material;
    int32_t flags;
    int32_t key_first;
    int32_t key_count;
    int32_t _pad;
    double v[9];
} CrPrimitive;

/* ---- materials: Materials::{Lambertian,Metal,Dielectric} (src/materials/mod.rs:16-21) ---- */
enum { CR_MAT_LAMBERTIAN = 0, CR_MAT_METAL = 1, CR_MAT_DIELECTRIC = 2 };

typedef struct CrMaterial {
    int32_t kind;
    int32_t texture;     /* Lambertian: index into textures (lambertian.rs:18)          */
    double albedo[3];    /* Metal: albedo (metal.rs:12)                                  */
    double param;        /* Lambertian: scatter_prob; Metal: fuzz; Dielectric: refraction_index */
} CrMaterial;

/* ---- textures: Textures::{SolidColor,CheckerTexture,ImageTexture} (src/textures/mod.rs:12-17) ---- */
enum { CR_TEX_SOLID = 0, CR_TEX_CHECKER = 1, CR_TEX_IMAGE = 2 };
/* Checker textures may nest (checker_texture.rs:12-13 hold arbitrary Arc<Textures>); cr_upload_scene returns
 * CR_ERR_UNSUPPORTED for a chain of more than this many checker levels. */
#define CR_MAX_CHECKER_DEPTH 32

typedef struct CrTexture {
    int32_t kind;
    int32_t even;        /* checker: texture index (checker_texture.rs:12)   */
    int32_t odd;         /* checker: texture index (checker_texture.rs:13)   */
    int32_t image;       /* image: index into images (image_texture.rs:11)   */
    double color[3];     /* solid: albedo (solid_color.rs:7)                 */
    double inv_scale;    /* checker: 1.0/scale, computed by the caller exactly as
                            CheckerTexture::new_* does (checker_texture.rs:23,31) */
} CrTexture;

/* RTWImage after image.to_rgb8() (src/asset_loader/img_loader.rs:27-46):
 * row-major, 3 bytes per texel, texel value = byte / 255.0. */
typedef struct CrImage {
    int32_t width;
    int32_t height;
    const uint8_t* rgb8;
} CrImage;

/* ---- Skybox (src/scene/mod.rs:18-25) ---- */
enum { CR_SKY_DEFAULT = 0, CR_SKY_SPHERICAL = 1 };

/*
 * One flattened keyframe of a TransformTimeline (src/timeline/mod.rs:116-120),
 * i.e. one Transform pushed by translate_{x,y,z} / scale_sphere
 * (src/timeline/transform_builder.rs).  The authoring API stays on the host;
 * only its evaluated form crosses the boundary.
 *   channel 0,1,2 : translate x,y,z.  Active when t >= t0
 *                   (valid_time.is_less(t) || contains(t), timeline/mod.rs:239).
 *                   value = a (NERP) | a * s (LERP), s = clamp((t-t0)/(t1-t0),0,1)
 *                   (timeline/mod.rs:90-96, transform_builder.rs `move |t| x * t`).
 *                   Active values are added in array order (timeline/mod.rs:243-246).
 *   channel 3     : sphere radius (scale_sphere, transform_builder.rs:17-96).  Spheres only.
 *   channel 4,5,6 : ScaleX / ScaleY / ScaleZ of a triangle (scale_x/y/z, scale_point, scale_all_uniform:
 *                   scene_animator.rs:38-229, transform_builder.rs:101-346,729).  Triangles only.
 *                   Channels 3..6 are the timeline's `scale` list: the LAST active key in array order wins and
 *                   replaces the initial scale (timeline/mod.rs:249-255); value v = a (NERP) | a + (b - a) * s
 *                   (LERP: `start + (x - start) * t`).  A vertex (x,y,z) -- after the translate keys -- becomes
 *                   ScaleX: (v*x, y, z); ScaleZ: (x, y, v*z); ScaleY: (x, v*x + y, z) -- the reference writes the
 *                   y factor into row 1, column 0 of the matrix (transform_builder.rs:228-246) and this ABI
 *                   reproduces that.  scale_point pushes X, Y, Z keys with one interval, so its Z key wins.
 *                   Keys must arrive in the timeline's list order (translate list, then scale list, each stably
 *                   sorted by start time as the reference sorts them).
 */
enum { CR_KEY_TX = 0, CR_KEY_TY = 1, CR_KEY_TZ = 2, CR_KEY_RADIUS = 3, CR_KEY_SCALE_X = 4, CR_KEY_SCALE_Y = 5,
       CR_KEY_SCALE_Z = 6 };
enum { CR_KEY_NERP = 0, CR_KEY_LERP = 1 };

typedef struct CrKeyframe {
    int32_t channel;
    int32_t interp;
    double t0, t1;
    double a, b;
} CrKeyframe;

/*
 * Which wrapper tree cr_upload_scene builds over the visible primitives.
 *   CR_BVH_REFERENCE  BVHWrapper::help_generate's median split (src/objects/bvhwrapper.rs:46-78):
 *                     the tree the reference walks, hence the parity mode and the default.
 *   CR_BVH_SAH        SURVEY 8(f) row 1: binned surface-area-heuristic topology.  Same wrapper kind
 *                     (box = union of the primitive boxes, leaves of 1-2 primitives), same walk
 *                     (BVHWrapper::hit, bvhwrapper.rs:96-126: left then right, shrinking interval),
 *                     fewer box tests.  The closest hit equals the reference tree's except where a
 *                     ray grazes a box face (Aabb::hit's `max <= min` miss, bvh.rs:96-132), so
 *                     images are not guaranteed bit-identical to CR_BVH_REFERENCE; cr_export_bvh
 *                     hands a checker the exact tree.
 *   CR_BVH_SAH_ORDERED CR_BVH_SAH's tree walked near child first: at an inner wrapper the child on the ray's
 *                     side of the split (the left one when direction[axis] >= 0) is visited before the other,
 *                     which then sees the interval already shrunk.  Not BVHWrapper::hit's order (always left
 *                     then right); the closest hit again differs from the reference tree's only on box-grazing
 *                     rays.  Stackless on the device (one skip link per direction octant).  Megakernel
 *                     pipeline only.
 *   CR_BVH_LBVH       SURVEY 8(f) row 1, "GPU LBVH": the tree is built on the device (Morton keys of the
 *                     primitive-box centroids, radix sort, Karras' topology; one primitive per leaf) in a few
 *                     milliseconds instead of the host builders' 0.3-0.4 s per million primitives.  Walked in
 *                     BVHWrapper::hit's order like CR_BVH_SAH; a lower-quality tree than SAH.
 */
enum { CR_BVH_REFERENCE = 0, CR_BVH_SAH = 1, CR_BVH_SAH_ORDERED = 2, CR_BVH_LBVH = 3 };

typedef struct CrSceneDesc {
    int32_t n_prims;
    int32_t n_materials;
    int32_t n_textures;
    int32_t n_images;
    int32_t n_keys;
    int32_t sky_kind;
    int32_t sky_image;
    int32_t bvh_mode;       /* CR_BVH_REFERENCE (0) | CR_BVH_SAH | CR_BVH_SAH_ORDERED | CR_BVH_LBVH */
    const CrPrimitive* prims;
    const CrMaterial* materials;
    const CrTexture* textures;
    const CrImage* images;
    const CrKeyframe* keys;
} CrSceneDesc;

/*
 * Camera state read by the render path (src/camera/mod.rs:66-100).  Angles are
 * passed in degrees as the reference's setters take them (set_vfov :213,
 * set_defocus_angle :250); image_height is what Viewport::new derives
 * (camera/mod.rs:37-38) and is passed explicitly so the caller's value wins.
 * look_from / look_at keyframes (cam_translate_point, scene_animator.rs) are
 * flattened like primitive keys; channels 0..2 only.
 */
typedef struct CrCameraDesc {
    int32_t image_width;
    int32_t image_height;
    double vfov_degrees;
    double defocus_angle_degrees;
    double focus_dist;
    double look_from[3];
    double look_at[3];
    double vup[3];
    int32_t from_key_count;
    int32_t at_key_count;
    const CrKeyframe* from_keys;
    const CrKeyframe* at_keys;
} CrCameraDesc;

/*
 * Per-render parameters.  samples/max_depth/frame/frame_rate/shutter_angle are
 * Camera fields (camera/mod.rs:83-99).  The reference draws every random number
 * from an unseeded thread-local generator (rand::rng(), ray_casting.rs:74); this
 * ABI replaces it with one seeded stream per (seed, pixel index, sample index)
 * (SplitMix64-derived key, xorshift64* draws) -- see DESIGN.md "RNG".
 * sample_begin/sample_count select a sub-range of the `samples` sample indices
 * (samples-per-pixel sharding across GPUs); the mean is still taken over
 * `samples` when the sums of all shards are added.
 */
/*
 * How a pixel's samples are summed and a path's attenuations multiplied (CrRenderParams.sum_order).  The paths are
 * the same in every mode -- same draws, same walks, same work counters; only the association of the products and
 * sums differs.
 *   CR_SUM_REFERENCE_ORDER  the reference's own order, bit for bit: `attenuation * ray_color(..)` multiplies
 *                           innermost-first (ray_casting.rs:128) and average_samples adds the samples in draw order
 *                           (:161-165).  The parity mode (every bit-exact test runs in it).  Costs a per-sample colour
 *                           buffer of image_width*image_height*3 reals per sample index (up to 40 GiB, see
 *                           cr_render_device) and a per-path attenuation stack.
 *   CR_SUM_RELAXED          the attenuations are multiplied in path order (a_1*a_2*...*a_n*sky: the same n
 *                           multiplies, associated left to right) and a finished sample is added to its pixel as
 *                           round(colour * 2^52) in a 64-bit integer (2^51, 2^50, ... beyond 2047 samples per pixel).
 *                           Integer adds commute, so the frame is deterministic -- two runs, or any split into
 *                           shards, give the same sums.  Per channel the mean differs from the reference order by
 *                           at most about (2 * max_depth + samples) * 2^-53: each of the two product orders rounds
 *                           max_depth times, and the reference's own sequential sum rounds once per sample where the
 *                           integer sum does not -- below 1e-13 at depth 50 and 512 samples, 1.0e-14 measured on the
 *                           headline frame (tested: <= 1e-12 against the oracle, equal counters, equal PPM bytes up to
 *                           values that sit on a byte boundary).  No per-sample buffer and no stack: 24 bytes of device
 *                           memory per pixel.
 *   CR_SUM_DEFAULT          the library's choice: CR_SUM_RELAXED (see DESIGN.md section 3.3 for the measurement);
 *                           the environment variable CRUCIBLE_SUM_ORDER=reference|relaxed overrides it.
 */
enum { CR_SUM_DEFAULT = 0, CR_SUM_REFERENCE_ORDER = 1, CR_SUM_RELAXED = 2 };

typedef struct CrRenderParams {
    int32_t samples;
    int32_t sample_begin;
    int32_t sample_count;
    int32_t max_depth;
    uint64_t seed;
    int32_t frame;
    int32_t real_type;      /* CR_REAL_F32 | CR_REAL_F64 */
    double frame_rate;
    double shutter_angle;
    int32_t output_sum;     /* 0: per-pixel mean over `samples` (average_samples,
                               ray_casting.rs:154-173); 1: raw per-pixel sum of this shard */
    int32_t refit_boxes;    /* 0: wrapper boxes stay the construction-time boxes, as in the reference
                               (bvhwrapper.rs:47-50) -- keyframed primitives are clipped where they leave them;
                               1: SURVEY 8(f) rows 1-2: boxes are re-derived on the device for this frame's
                               ray-time interval before the render (crucible_amd/csrc/refit.hpp), so moving
                               primitives are intersected wherever they are.  No effect without primitive keys. */
    int32_t sum_order;      /* CR_SUM_DEFAULT (0) | CR_SUM_REFERENCE_ORDER | CR_SUM_RELAXED */
    int32_t _reserved;
} CrRenderParams;

/*
 * Work counters of the last render (the algorithmic-bytes model of DESIGN.md):
 * segments = closest-hit queries, node_tests = BVH boxes tested,
 * prim_tests = primitive intersection tests, texel_fetches = image texels read.
 */
typedef struct CrStats {
    uint64_t samples;
    uint64_t segments;
    uint64_t node_tests;
    uint64_t prim_tests;
    uint64_t texel_fetches;
    uint64_t nan_pixels;
    double kernel_ms;       /* HIP-event time of the render kernel(s) on the handle's stream */
    double upload_ms;
    int32_t bvh_entries;
    int32_t scene_in_lds;   /* 0: scene read through L2; 1: whole scene staged in LDS; 2: top levels of the BVH in LDS */
} CrStats;

typedef struct CrHandle CrHandle;

/* ---- entry points ---- */

/* Library ABI version (CR_ABI_VERSION). */
CR_API int32_t cr_abi_version(void);

/* Create a renderer bound to one HIP device.  One handle = one caller thread at
 * a time, like the reference's single-caller Camera (camera/mod.rs:354). */
CR_API int32_t cr_create(int32_t device_id, CrHandle** out);
CR_API void cr_destroy(CrHandle* h);

/* Copies the whole description (the caller may free it on return), filters
 * hidden primitives and builds the BVH selected by scene->bvh_mode -- by default the
 * reference topology (BVHWrapper::new_wrapper, src/objects/bvhwrapper.rs:15-93) -- for
 * each scalar type on first use.  Replaces the `world`/`skybox` arguments of Camera::render. */
CR_API int32_t cr_upload_scene(CrHandle* h, const CrSceneDesc* scene);

/* Camera::render minus the file output: renders into a DEVICE buffer of
 * image_width*image_height*3 reals (f32 or f64 per params->real_type), row-major,
 * RGB interleaved.  Asynchronous on the handle's stream unless `stats` is non-NULL
 * (then it synchronises to fill the stats).
 * Device memory: besides the scene the handle keeps a per-path attenuation stack (3*max_depth reals per
 * resident lane, ~150 MB at depth 50) and a per-sample colour buffer of image_width*image_height*3 reals per
 * sample index, up to 40 GiB (CRUCIBLE_SAMPLE_BUF_MB; a render that needs more runs as consecutive sample
 * batches, CRUCIBLE_SAMPLE_GRANULAR=0 avoids the buffer at a large cost in speed).  The buffers are
 * grown on demand, reused by later renders and freed by cr_destroy. */
CR_API int32_t cr_render_device(CrHandle* h, const CrCameraDesc* cam, const CrRenderParams* params,
                         void* d_out_rgb, CrStats* stats);

/* Same, into a HOST buffer (render + device->host copy, synchronous). */
CR_API int32_t cr_render_host(CrHandle* h, const CrCameraDesc* cam, const CrRenderParams* params,
                       void* h_out_rgb, CrStats* stats);

/* The wrapper tree the device walks for `real_type`, as BVHWrapper's shape (src/objects/bvhwrapper.rs:7-11):
 * wrapper k has boxes[6k..6k+5] = xmin,xmax,ymin,ymax,zmin,zmax (exact values of `real_type`) and
 * children[2k], children[2k+1] = left, right: >= 0 another wrapper's index, < 0 the bitwise complement of a
 * primitive's index in CrSceneDesc.prims.  Wrappers are numbered in walk order (root 0, left subtree, right
 * subtree); a one-primitive wrapper names that primitive twice (bvhwrapper.rs:58-60).  split_axis (may be NULL)
 * receives, per wrapper, the axis whose direction sign picks the child visited first in CR_BVH_SAH_ORDERED mode
 * (right child first when direction[axis] < 0), or -1 where the order is always left then right.  *n_wrappers
 * receives the count; boxes/children may be NULL to query it.  Builds the tree if the scene was not rendered yet. */
CR_API int32_t cr_export_bvh(CrHandle* h, int32_t real_type, double* boxes, int32_t* children, int32_t* split_axis,
                             int32_t capacity, int32_t* n_wrappers);

/* Wait for the last render launched on this handle and return its kernel time in
 * milliseconds, measured with HIP events recorded on the handle's stream around the
 * launch (no counters are copied back). */
CR_API int32_t cr_last_kernel_ms(CrHandle* h, double* out_ms);

/* Block until the handle's stream is idle. */
CR_API int32_t cr_synchronize(CrHandle* h);

/* The hipStream_t the handle launches on (so a caller can order its own work). */
CR_API void* cr_stream(CrHandle* h);

/* PPM P3 writer reproducing Camera::render's output (camera/mod.rs:286,306-311) and
 * `impl Display for Color` (src/utils.rs:422-437): (255*sqrt(c)) as u32.  `rgb` is
 * image_width*image_height*3 host reals of `real_type` holding per-pixel means. */
CR_API int32_t cr_write_ppm(const char* path, const void* rgb, int32_t real_type,
                     int32_t image_width, int32_t image_height);

/* SURVEY 8(f) row 3 -- faster frame output than the reference's ASCII P3, same bytes per channel as
 * `impl Display for Color`: binary PPM (P6) and 8-bit RGB PNG (what BASELINE.json's north_star names). */
CR_API int32_t cr_write_ppm_binary(const char* path, const void* rgb, int32_t real_type,
                                   int32_t image_width, int32_t image_height);
CR_API int32_t cr_write_png(const char* path, const void* rgb, int32_t real_type,
                            int32_t image_width, int32_t image_height);

/* Quantise means to the bytes Display would print (3 per pixel); no file. */
CR_API int32_t cr_quantize_rgb8(const void* rgb, int32_t real_type, int64_t n_pixels, uint8_t* out);

/* Last error text of this handle (NULL handle: last create error). */
CR_API const char* cr_last_error(CrHandle* h);

/*
 * ---- several GPUs of one node: samples-per-pixel sharding + one RCCL reduce ----
 *
 * Every sample is an independent path (src/camera/ray_casting.rs:82-105) and the random stream is keyed by
 * (seed, pixel, sample), so member g of G renders sample indices cr_group_shard(samples, g, G) of EVERY pixel as raw
 * per-pixel sums (the scene is replicated), one ncclReduce(sum, f32 | f64 per real_type, count = W*H*3, root = member 0)
 * over xGMI adds them, and the root divides by `samples` (average_samples' `/= count`, ray_casting.rs:168-170).  The
 * union is the 1-GPU sample set; the image differs from the 1-GPU one only by the order of the floating-point adds
 * (a group of one member is bit-identical to cr_render_device).  This replaces the reference's worker pool
 * (src/camera/cpu_threading.rs:25-115: `thread_count` OS threads behind one mutex) across devices.
 *
 * Two ways to form a group; both end in the same cr_group_render:
 *   cr_group_create       one process drives n devices (one handle + stream per device, ncclCommInitAll) -- what a
 *                         Rust host calling this library would use;
 *   cr_group_create_rank  one process per GPU (torch.distributed.run style): rank 0 obtains cr_group_unique_id and
 *                         ships its 128 bytes to every rank by any means (ncclCommInitRank).
 * RCCL (librccl.so.1) is loaded on first use; a group of ONE member never needs it.  Movies shard whole frames
 * instead (frame f -> member f % G, src/scene/mod.rs:307-316): no collective, drive cr_group_handle(g) directly.
 */
typedef struct CrGroup CrGroup;
#define CR_GROUP_ID_BYTES 128

/* [begin, begin+count) of member `member` among `n_members`: begin = member*samples/n_members (integer division).
 * The ranges partition [0, samples); with more members than samples some are empty (they contribute zeros).
 * Pure arithmetic: callable without a GPU. */
CR_API int32_t cr_group_shard(int32_t samples, int32_t member, int32_t n_members, int32_t* begin, int32_t* count);

CR_API int32_t cr_group_create(const int32_t* device_ids, int32_t n_devices, CrGroup** out);
CR_API int32_t cr_group_unique_id(uint8_t id[CR_GROUP_ID_BYTES]);
CR_API int32_t cr_group_create_rank(int32_t device_id, int32_t rank, int32_t world_size,
                                    const uint8_t id[CR_GROUP_ID_BYTES], CrGroup** out);
CR_API void cr_group_destroy(CrGroup* g);

/* Members driven by THIS process (n_devices, or 1 in rank mode) / in the whole group / this process's first member. */
CR_API int32_t cr_group_local_size(CrGroup* g);
CR_API int32_t cr_group_size(CrGroup* g);
CR_API int32_t cr_group_rank(CrGroup* g);
CR_API CrHandle* cr_group_handle(CrGroup* g, int32_t local_member);

/* cr_upload_scene on every local member (the scene is replicated). */
CR_API int32_t cr_group_upload_scene(CrGroup* g, const CrSceneDesc* scene);

/* Camera::render across the group.  params->sample_begin/sample_count/output_sum are ignored: the group splits
 * [0, samples) itself and returns the per-pixel MEAN.  d_out_rgb: device buffer of W*H*3 reals on the ROOT member's
 * device (member 0 = device_ids[0], or rank 0); other ranks may pass NULL.  Every rank of a rank-mode group must
 * call this with the same arguments (it is a collective).  Synchronous.  stats (may be NULL): counters summed over
 * the LOCAL members, kernel_ms = the slowest local member's render, reduce_ms in CrGroupStats.
 * Failure: arguments are validated and buffers allocated on every member before anything is launched; with a
 * collective, the members then agree (a 4-byte ncclAllReduce(min) on the render streams) that every render is fine
 * before the ncclReduce is entered -- if one is not, EVERY rank returns an error (its own, or CR_ERR_PEER) and the
 * group stays usable.  If a collective call itself fails, the group's communicators are aborted and every later call
 * on it returns CR_ERR_PEER: destroy the group.  The calling thread's current HIP device is restored on return. */
typedef struct CrGroupStats {
    CrStats render;        /* local members: counters summed, kernel_ms = max */
    double reduce_ms;      /* root-side time of the ncclReduce + the divide, HIP events on the root's stream */
    int32_t members;       /* whole group */
    int32_t used_rccl;     /* 0: a one-member group rendered directly */
} CrGroupStats;
CR_API int32_t cr_group_render(CrGroup* g, const CrCameraDesc* cam, const CrRenderParams* params, void* d_out_rgb,
                               CrGroupStats* stats);
/* Same, into a HOST buffer on the root (cr_render_host's counterpart: render + reduce + device->host copy, and the
 * Color::new check of every mean -> CR_ERR_NAN).  Ranks other than the root may pass NULL. */
CR_API int32_t cr_group_render_host(CrGroup* g, const CrCameraDesc* cam, const CrRenderParams* params, void* h_out_rgb,
                                    CrGroupStats* stats);
CR_API const char* cr_group_last_error(CrGroup* g);

#ifdef __cplusplus
}
#endif
#endif /* CRUCIBLE_HIP_H */
