// crucible_render -- CLI mirror of Crucible's src/main.rs:5-79 over the HIP library.
//   --file/-f NAME  --world/-w N  [--threads/-t N (accepted, unused: the GPU replaces the pool)]
//   [--movie/-m --seconds/-s S --rate/-r R]
// Extras (not in the reference): --width, --samples, --seed, --scene-seed, --real f32|f64, --device,
// --format ppm|p6|png (frame files: the reference's ASCII P3, binary PPM, PNG),
// --sky FILE.hdr (Radiance map as the spherical skybox; world 5 = demo_images::garden_skybox needs it),
// --bvh reference|sah|ordered|lbvh (the tree cr_upload_scene builds; reference = the parity mode, default), --refit (re-derive
// the wrapper boxes per frame so keyframed primitives are not clipped; the reference does not),
// --dump-desc FILE (write the flattened scene description and exit; used by the tests to check
// this mirror against the Python one), --sum-order reference|relaxed (CrRenderParams.sum_order; default: the library's),
// --repeat N --timing (measurement: render_scene N times in this process, one JSON line of wall-clock phases each --
// the shape of the reference's criterion benchmark, benches/renderer_benchmark.rs:16-42 -- the first is the cold one).
#include "crucible.hpp"

#include <cstdio>
#include <cstring>

using namespace crucible;

static void dump_desc(const FlatScene& f, const char* path) {
    FILE* fp = fopen(path, "wb");
    if (!fp) { perror("dump"); exit(2); }
    int32_t hdr[7] = {f.desc.n_prims, f.desc.n_materials, f.desc.n_textures, f.desc.n_images, f.desc.n_keys, f.desc.sky_kind, f.desc.sky_image};
    fwrite(hdr, sizeof hdr, 1, fp);
    auto put = [fp](const void* p, size_t size, size_t n) { if (n) fwrite(p, size, n, fp); };   // an empty vector's data() may be null
    put(f.prims.data(), sizeof(CrPrimitive), f.prims.size());
    put(f.materials.data(), sizeof(CrMaterial), f.materials.size());
    put(f.textures.data(), sizeof(CrTexture), f.textures.size());
    put(f.keys.data(), sizeof(CrKeyframe), f.keys.size());
    for (const CrImage& im : f.images) {
        int32_t wh[2] = {im.width, im.height};
        fwrite(wh, sizeof wh, 1, fp);
        fwrite(im.rgb8, 1, (size_t)im.width * im.height * 3, fp);
    }
    fclose(fp);
}

int main(int argc, char** argv) {
    std::string file, dump, real = "f32";
    size_t threads = 1, world = 1, rate = 0;
    bool movie = false;
    double seconds = -1;
    long width = -1, samples = -1;
    uint64_t seed = 0xC0FFEE, scene_seed = 1;
    int device = 0;
    std::string bvh = "reference", sky, format = "ppm";
    bool refit = false, use_group = false, timing = false;
    int gpus = 1, repeat = 1;
    std::string sum_order = "default";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "-f" || a == "--file") file = next();
        else if (a == "-t" || a == "--threads") threads = strtoul(next(), nullptr, 10);
        else if (a == "-w" || a == "--world") world = strtoul(next(), nullptr, 10);
        else if (a == "-m" || a == "--movie") movie = true;
        else if (a == "-s" || a == "--seconds") seconds = atof(next());
        else if (a == "-r" || a == "--rate") rate = strtoul(next(), nullptr, 10);
        else if (a == "--width") width = atol(next());
        else if (a == "--samples") samples = atol(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 0);
        else if (a == "--scene-seed") scene_seed = strtoull(next(), nullptr, 0);
        else if (a == "--real") real = next();
        else if (a == "--device") device = atoi(next());
        else if (a == "--bvh") bvh = next();
        else if (a == "--sky") sky = next();
        else if (a == "--format") format = next();
        else if (a == "--refit") refit = true;
        else if (a == "--gpus") gpus = (int)strtol(next(), nullptr, 10);
        else if (a == "--group") use_group = true;
        else if (a == "--timing") timing = true;
        else if (a == "--repeat") repeat = std::max(1, atoi(next()));
        else if (a == "--sum-order") sum_order = next();
        else if (a == "--dump-desc") dump = next();
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (file.empty() && dump.empty()) { fprintf(stderr, "usage: crucible_render --file NAME --world N [--movie --seconds S --rate R]\n"); return 2; }
    uint32_t w = width > 0 ? (uint32_t)width : 400;
    try {
        Scene scene = [&]() {
            if (movie) {
                fprintf(stderr, "Rendering a movie!\n");
                if (rate == 0) throw std::invalid_argument("You must provide a frame rate if you are making a movie");
                if (seconds < 0) throw std::invalid_argument("You must provide seconds if you are making a movie");
                // demo_movies::{first_movie,moving_teapot} need garden.hdr (missing blob) / panic in scale_r; the movie
                // here is the book1 camera walk: the same scene API, frames through render_movie
                Scene s = demo_builder::book1_end_scene(threads, scene_seed, w, samples > 0 ? (uint32_t)samples : 50);
                s.is_movie = true; s.duration = seconds; s.frame_rate = rate; s.scene_cam.frame_rate = (double)rate;
                s.scene_cam.set_max_depth(5);
                s.cam_translate_point(Point3{3, 2, 13}, seconds, InterpolationType::LERP, TransformSpace::World, "from");
                return s;
            }
            fprintf(stderr, "Rendering an image!\n");
            switch (world) {
                case 1: return demo_builder::book1_end_scene(threads, scene_seed, w, samples > 0 ? (uint32_t)samples : 500);
                case 2: return demo_builder::checkered_spheres(threads, w, samples > 0 ? (uint32_t)samples : 500);
                case 3: return demo_builder::load_teapot(threads, w, samples > 0 ? (uint32_t)samples : 200);
                case 6: return demo_builder::scaled_teapot(threads, w, samples > 0 ? (uint32_t)samples : 200);
                case 7: return demo_builder::teapot_as_list(threads, w, samples > 0 ? (uint32_t)samples : 200);
                case 5:
                    if (sky.empty()) throw std::invalid_argument("world 5 (garden_skybox) needs --sky FILE.hdr: garden.hdr is not shipped");
                    return demo_builder::garden_skybox(threads, RTWImage::load_hdr(sky), w, samples > 0 ? (uint32_t)samples : 500);
                default:
                    fprintf(stderr, "Invalid world number. Selecting default scene\n");   // world 4 (earth) needs a JPEG decoder (third-party codec)
                    return demo_builder::book1_end_scene(threads, scene_seed, w, samples > 0 ? (uint32_t)samples : 500);
            }
        }();
        if (!sky.empty() && world != 5) scene.load_spherical_skybox(RTWImage::load_hdr(sky));
        scene.seed = seed; scene.device = device;
        scene.real_type = real == "f64" ? CR_REAL_F64 : CR_REAL_F32;
        if (bvh != "reference" && bvh != "sah" && bvh != "ordered" && bvh != "lbvh") { fprintf(stderr, "--bvh takes reference, sah, ordered or lbvh\n"); return 2; }
        scene.bvh_mode = bvh == "sah" ? CR_BVH_SAH : (bvh == "ordered" ? CR_BVH_SAH_ORDERED : (bvh == "lbvh" ? CR_BVH_LBVH : CR_BVH_REFERENCE));
        scene.refit_boxes = refit;
        scene.gpus = std::max(1, gpus); scene.use_group = use_group;
        if (format != "ppm" && format != "p6" && format != "png") { fprintf(stderr, "--format takes ppm, p6 or png\n"); return 2; }
        scene.frame_format = format;
        if (sum_order != "default" && sum_order != "reference" && sum_order != "relaxed") { fprintf(stderr, "--sum-order takes default, reference or relaxed\n"); return 2; }
        scene.sum_order = sum_order == "reference" ? CR_SUM_REFERENCE_ORDER : (sum_order == "relaxed" ? CR_SUM_RELAXED : CR_SUM_DEFAULT);
        if (!dump.empty()) { dump_desc(scene.flatten(), dump.c_str()); return 0; }
        int32_t rc = CR_OK;
        for (int rep = 0; rep < repeat && rc == CR_OK; rep++) {
            CrStats st;
            memset(&st, 0, sizeof st);
            scene.scene_cam.frame = 0;
            scene.quiet = timing;
            rc = scene.render_scene(repeat > 1 ? file + (rep ? "_" + std::to_string(rep) : "") : file, &st);
            if (rc == CR_OK && !timing) fprintf(stderr, "kernel %.3f ms, %.1f Msamples/s\n", st.kernel_ms, st.kernel_ms > 0 ? st.samples / st.kernel_ms / 1e3 : 0.0);
            if (rc == CR_OK && timing) {
                const Scene::Timing& t = scene.timing;
                const double samples = (double)scene.scene_cam.image_width * scene.scene_cam.image_height * scene.scene_cam.samples * (movie ? (double)t.frames : 1.0);
                printf("{\"run\": %d, \"width\": %u, \"height\": %u, \"samples\": %u, \"frames\": %zu, \"format\": \"%s\", \"real\": \"%s\", \"create_ms\": %.3f, \"flatten_ms\": %.3f, "
                       "\"upload_ms\": %.3f, \"bvh_build_ms\": %.3f, \"render_ms\": %.3f, \"kernel_ms\": %.3f, \"write_ms\": %.3f, \"total_ms\": %.3f, \"msamples_per_s_end_to_end\": %.1f, "
                       "\"msamples_per_s_kernel\": %.1f}\n",
                       rep, scene.scene_cam.image_width, scene.scene_cam.image_height, scene.scene_cam.samples, movie ? t.frames : (size_t)1, format.c_str(), real.c_str(), t.create_ms,
                       t.flatten_ms, t.upload_ms, t.bvh_build_ms, t.render_ms, t.kernel_ms, t.write_ms, t.total_ms, samples / t.total_ms / 1e3, t.kernel_ms > 0 ? samples / t.kernel_ms / 1e3 : 0.0);
                fflush(stdout);
            }
        }
        return rc == CR_OK ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "panic: %s\n", e.what());   // the reference panics here
        return 101;
    }
}
