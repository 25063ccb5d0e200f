// crucible.hpp -- C++ host-side mirror of Crucible's Scene / Camera builder API over the C ABI
// (include/crucible_hip.h).  The reference is Rust and this image has no Rust toolchain, so the host
// layer above the boundary is written in C++ with the reference's names, argument meaning and error
// behaviour (exceptions where the reference panics, CR_ERR_IO where it returns io::Error):
//   Scene            src/scene/mod.rs:75-347, src/scene/scene_animator.rs
//   Camera           src/camera/mod.rs:66-263
//   TransformTimeline src/timeline/mod.rs:116-231, src/timeline/transform_builder.rs
//   Sphere/Triangle  src/objects/sphere.rs:25-39, src/objects/triangle.rs:23-46
//   Materials/Textures src/materials/*.rs, src/textures/*.rs
//   load_obj         src/asset_loader/obj_loader.rs:21-143
//   demo scenes      src/demo_builder/demo_images.rs (seeded; see DESIGN.md "RNG")
// Nothing here computes pixels: Scene::flatten() produces the CrSceneDesc the library consumes and
// Scene::render_scene() is Camera::render over cr_upload_scene / cr_render_host / cr_write_ppm.
#pragma once
#include <thread>
#include <cstring>
#include "../../include/crucible_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace crucible {

struct Point3 { double x = 0, y = 0, z = 0; };
using Vec3 = Point3;

struct Color {   // Color::new asserts 0 <= c <= 1 (utils.rs:345-350)
    double r, g, b;
    Color(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {
        auto chk = [](const char* n, double v) {
            if (!(v >= 0.0 && v <= 1.0)) throw std::invalid_argument(std::string(n) + " must be between 0.0 and 1.0");
        };
        chk("R", r); chk("G", g); chk("B", b);
    }
};

// ---------------------------------------------------------------- timeline authoring
enum class InterpolationType { NERP, LERP };   // timeline/mod.rs:100-103
enum class TransformSpace { World, Local };     // timeline/mod.rs:108-111

class TransformTimeline {
    struct Tf { int channel; int ttype; double t0, t1; int interp; double a, b; int end_kind; double end_v[3]; };
    enum { TX = 0, TY = 1, TZ = 2, SR = 3, SX = 4, SY = 5, SZ = 6, OMNI = 9, END_AXIS = 0, END_INIT = 1 };
    std::vector<Tf> scale_, translate_;

    static const Tf* most_recent(const std::vector<Tf>& l, double t, int ttype) {   // helper_functions.rs:41-140
        for (auto it = l.rbegin(); it != l.rend(); ++it)
            if (t > it->t1 && (it->ttype == ttype || it->ttype == OMNI)) return &*it;
        return nullptr;
    }
    void translate_axis(int axis, double x, double keyframe, InterpolationType it, TransformSpace sp) {   // transform_builder.rs:339-717
        if (!(keyframe >= 0.0)) throw std::invalid_argument("Cannot add a keyframe before the animation start.");
        const Tf* prev = most_recent(translate_, keyframe, axis);
        if (!prev) throw std::runtime_error("Missing transform data! could not find a previous position reference");
        double prev_time = std::max(prev->t1, 0.0), standard = x;
        if (sp == TransformSpace::World) x -= (prev->end_kind == END_AXIS ? prev->end_v[0] : prev->end_v[axis]);
        Tf tf{axis, axis, it == InterpolationType::LERP ? prev_time : keyframe, keyframe,
              it == InterpolationType::LERP ? CR_KEY_LERP : CR_KEY_NERP, x, 0.0, END_AXIS, {standard, 0, 0}};
        translate_.push_back(tf);
        std::stable_sort(translate_.begin(), translate_.end(), [](const Tf& a, const Tf& b) { return a.t0 < b.t0; });
    }

public:
    Point3 start_pos;
    double start_scale = 1.0;
    bool sphere = false;

    TransformTimeline(Point3 pos = {}, double scale = 1.0, bool is_sphere = false) : start_pos(pos), start_scale(scale), sphere(is_sphere) {
        scale_.push_back(Tf{-1, OMNI, -0.1, -0.1, CR_KEY_NERP, scale, 0, END_INIT, {scale, scale, scale}});
        translate_.push_back(Tf{-1, OMNI, -0.1, -0.1, CR_KEY_NERP, 0, 0, END_INIT, {pos.x, pos.y, pos.z}});
    }
    static TransformTimeline new_sphere(Point3 pos, double radius) { return TransformTimeline(pos, radius, true); }

    void translate_x(double x, double k, InterpolationType it, TransformSpace sp) { translate_axis(TX, x, k, it, sp); }
    void translate_y(double y, double k, InterpolationType it, TransformSpace sp) { translate_axis(TY, y, k, it, sp); }
    void translate_z(double z, double k, InterpolationType it, TransformSpace sp) { translate_axis(TZ, z, k, it, sp); }
    void translate_point(Point3 p, double k, InterpolationType it, TransformSpace sp) {   // transform_builder.rs:721-731
        translate_x(p.x, k, it, sp); translate_y(p.y, k, it, sp); translate_z(p.z, k, it, sp);
    }
    void scale_sphere(double r, double keyframe, InterpolationType it) {   // transform_builder.rs:17-96
        if (!(keyframe >= 0.0)) throw std::invalid_argument("Cannot add a keyframe before the animation start.");
        const Tf* prev = most_recent(scale_, keyframe, SR);
        if (!prev) throw std::runtime_error("Missing transform data! Tried to scale radius but could not find a previous scale reference!");
        double prev_time = std::max(prev->t1, 0.0), start = prev->end_v[0];
        Tf tf = it == InterpolationType::LERP ? Tf{CR_KEY_RADIUS, SR, prev_time, keyframe, CR_KEY_LERP, start, r, END_AXIS, {r, 0, 0}}
                                              : Tf{CR_KEY_RADIUS, SR, keyframe, keyframe, CR_KEY_NERP, r, 0.0, END_AXIS, {r, 0, 0}};
        scale_.push_back(tf);
        std::stable_sort(scale_.begin(), scale_.end(), [](const Tf& a, const Tf& b) { return a.t0 < b.t0; });
    }
    // scale_x / scale_y / scale_z (transform_builder.rs:101-346) -> CR_KEY_SCALE_X/Y/Z; which matrix slot each writes
    // (ScaleY: row 1, column 0) is applied where the keys are evaluated (pathtrace.hpp scale_point)
    void scale_axis(int ttype, double x, double keyframe, InterpolationType it) {
        if (!(keyframe >= 0.0)) throw std::invalid_argument("Cannot add a keyframe before the animation start.");
        const Tf* prev = most_recent(scale_, keyframe, ttype);
        if (!prev) throw std::runtime_error("Missing transform data! Tried to scale an axis but could not find a previous scale reference!");
        double prev_time = std::max(prev->t1, 0.0), start = prev->end_v[0];
        Tf tf = it == InterpolationType::LERP ? Tf{ttype, ttype, prev_time, keyframe, CR_KEY_LERP, start, x, END_AXIS, {x, 0, 0}}
                                              : Tf{ttype, ttype, keyframe, keyframe, CR_KEY_NERP, x, 0.0, END_AXIS, {x, 0, 0}};
        scale_.push_back(tf);
        std::stable_sort(scale_.begin(), scale_.end(), [](const Tf& a, const Tf& b) { return a.t0 < b.t0; });
    }
    void scale_x(double x, double k, InterpolationType it) { scale_axis(SX, x, k, it); }
    void scale_y(double y, double k, InterpolationType it) { scale_axis(SY, y, k, it); }
    void scale_z(double z, double k, InterpolationType it) { scale_axis(SZ, z, k, it); }
    void scale_point(Point3 p, double k, InterpolationType it) {   // transform_builder.rs:729-733
        scale_x(p.x, k, it); scale_y(p.y, k, it); scale_z(p.z, k, it);
    }
    std::vector<CrKeyframe> keyframes() const {   // translate list order, then scale list order
        std::vector<CrKeyframe> out;
        for (const Tf& t : translate_) if (t.channel >= 0) out.push_back(CrKeyframe{t.channel, t.interp, t.t0, t.t1, t.a, t.b});
        for (const Tf& t : scale_) if (t.channel >= 0) out.push_back(CrKeyframe{t.channel, t.interp, t.t0, t.t1, t.a, t.b});
        return out;
    }
};

// ---------------------------------------------------------------- textures, materials
struct RTWImage {   // decoded RGB8 (img_loader.rs:17-55)
    int width = 0, height = 0;
    std::vector<uint8_t> rgb8;

    // Radiance .hdr (RGBE) files -- the format of the reference's garden.hdr skybox
    // (demo_images.rs:223-242).  The reference decodes through the `image` crate and keeps 8 bits
    // (`to_rgb8()`, img_loader.rs:24-28): float = mantissa * 2^(e - 136) (0 when e == 0), then
    // round(clamp(x, 0, 1) * 255).  The crate is not vendored, so this conversion is parity unpinned.
    // Flat, new-style (per-channel) RLE and old-style repeat-marker scanlines; -Y/+Y H +X W orientations.
    static std::shared_ptr<RTWImage> load_hdr(const std::string& path) {
        FILE* fp = fopen(path.c_str(), "rb");
        if (!fp) throw std::runtime_error("Could not open image " + path);
        std::vector<uint8_t> d;
        uint8_t chunk[65536];
        for (size_t n; (n = fread(chunk, 1, sizeof chunk, fp)) > 0;) d.insert(d.end(), chunk, chunk + n);
        fclose(fp);
        size_t pos = 0;
        auto line = [&]() { std::string l; while (pos < d.size() && d[pos] != '\n') l.push_back((char)d[pos++]); pos++; return l; };
        std::string magic = line();
        if (magic.rfind("#?", 0) != 0) throw std::runtime_error("not a Radiance file: " + path);
        for (;;) { if (pos >= d.size()) throw std::runtime_error("truncated Radiance header"); if (line().empty()) break; }
        char sy = 0, sx = 0, ay = 0, ax = 0;
        int H = 0, W = 0;
        std::string res = line();
        if (sscanf(res.c_str(), "%c%c %d %c%c %d", &sy, &ay, &H, &sx, &ax, &W) != 6 || ay != 'Y' || ax != 'X' || sx != '+' || W < 1 || H < 1)
            throw std::runtime_error("unsupported Radiance resolution line: " + res);
        // every pixel takes at least one byte of the file (a run covers 127 at most, one scanline at a time): a header that
        // promises more than the file can hold is refused before anything is allocated
        if ((uint64_t)W * (uint64_t)H > (uint64_t)d.size() * 128u) throw std::runtime_error("truncated Radiance data");
        auto im = std::make_shared<RTWImage>();
        im->width = W; im->height = H; im->rgb8.resize((size_t)W * H * 3);
        std::vector<uint8_t> sl((size_t)W * 4);
        auto need = [&](size_t n) { if (pos + n > d.size()) throw std::runtime_error("truncated Radiance data"); };
        for (int row = 0; row < H; row++) {
            need(4);
            if (W >= 8 && W < 32768 && d[pos] == 2 && d[pos + 1] == 2 && ((d[pos + 2] << 8) | d[pos + 3]) == W) {
                pos += 4;
                for (int c = 0; c < 4; c++)
                    for (int x = 0; x < W;) {
                        need(1);
                        int n = d[pos++];
                        if (n > 128) { n -= 128; need(1); if (x + n > W) throw std::runtime_error("bad Radiance run"); uint8_t v = d[pos++]; for (int k = 0; k < n; k++) sl[(size_t)(x++) * 4 + c] = v; }
                        else { if (n == 0 || x + n > W) throw std::runtime_error("bad Radiance run"); need((size_t)n); for (int k = 0; k < n; k++) sl[(size_t)(x++) * 4 + c] = d[pos++]; }
                    }
            } else {   // flat pixels, with old-style repeat markers (1,1,1,count)
                int shift = 0;
                for (int x = 0; x < W;) {
                    need(4);
                    const uint8_t* q = &d[pos]; pos += 4;
                    if (q[0] == 1 && q[1] == 1 && q[2] == 1 && x > 0) {
                        long n = (long)q[3] << shift;
                        if (x + n > W) throw std::runtime_error("bad Radiance run");
                        for (long k = 0; k < n; k++, x++) memcpy(&sl[(size_t)x * 4], &sl[(size_t)(x - 1) * 4], 4);
                        shift += 8;
                    } else { memcpy(&sl[(size_t)x * 4], q, 4); x++; shift = 0; }
                }
            }
            const int out_row = sy == '-' ? row : H - 1 - row;   // -Y: top row first
            for (int x = 0; x < W; x++) {
                const uint8_t* q = &sl[(size_t)x * 4];
                const float f = q[3] ? std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
                for (int c = 0; c < 3; c++) {
                    float v = (float)q[c] * f;
                    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                    im->rgb8[((size_t)out_row * W + x) * 3 + c] = (uint8_t)std::lround(v * 255.0f);
                }
            }
        }
        return im;
    }
};
struct Textures;
using TexturePtr = std::shared_ptr<Textures>;   // Arc<Textures>
struct Textures {
    int kind = CR_TEX_SOLID;
    Color albedo{0, 0, 0};
    double inv_scale = 0;
    TexturePtr even, odd;
    std::shared_ptr<RTWImage> image;
    static TexturePtr solid(Color c) { auto t = std::make_shared<Textures>(); t->albedo = c; return t; }   // SolidColor::new_from_color
    static TexturePtr checker(double scale, TexturePtr e, TexturePtr o) {   // CheckerTexture::new_from_textures
        auto t = std::make_shared<Textures>(); t->kind = CR_TEX_CHECKER; t->inv_scale = 1.0 / scale; t->even = e; t->odd = o; return t;
    }
    static TexturePtr checker(double scale, Color c1, Color c2) { return checker(scale, solid(c1), solid(c2)); }   // ::new_from_color
    static TexturePtr from_image(std::shared_ptr<RTWImage> im) { auto t = std::make_shared<Textures>(); t->kind = CR_TEX_IMAGE; t->image = im; return t; }
};
struct Materials;
using MaterialPtr = std::shared_ptr<Materials>;
struct Materials {
    int kind = CR_MAT_LAMBERTIAN;
    TexturePtr tex;
    Color albedo{0, 0, 0};
    double param = 0;
    static MaterialPtr lambertian(Color c, double prob) { return lambertian(Textures::solid(c), prob); }   // Lambertian::new_from_color
    static MaterialPtr lambertian(TexturePtr t, double prob) { auto m = std::make_shared<Materials>(); m->tex = t; m->param = prob; return m; }
    static MaterialPtr metal(Color c, double fuzz) {   // Metal::new, metal.rs:20-24
        if (!(fuzz <= 1.0)) throw std::invalid_argument("A metal cannot have a fuzz factor above 1.0");
        if (!(fuzz >= 0.0)) throw std::invalid_argument("A metal cannot have a fuzz factor below 0.0");
        auto m = std::make_shared<Materials>(); m->kind = CR_MAT_METAL; m->albedo = c; m->param = fuzz; return m;
    }
    static MaterialPtr dielectric(double ri) { auto m = std::make_shared<Materials>(); m->kind = CR_MAT_DIELECTRIC; m->albedo = Color(1, 1, 1); m->param = ri; return m; }
};

// ---------------------------------------------------------------- objects
struct Hittables {   // Hittables::{Sphere,Triangle,HitList}
    int kind = CR_PRIM_SPHERE;
    size_t id = 0;
    bool hide = false;
    double v[9] = {0};
    MaterialPtr mat;
    TransformTimeline timeline;
    // HitList (objects/hitlist.rs).  As a scene element (scene/mod.rs:164-166) the BVH build treats it as one object
    // whose box is what add() accumulated: new(objs) leaves Aabb::default() (:13-18), clear() keeps the old box (:20-22).
    // The C ABI can say "empty box" or "union of the objects"; a list inside a list is passed as its objects spliced
    // in place (the inner box is never read by a hit).  Other mixtures make flatten() throw.
    std::vector<Hittables> objs;
    size_t n_new = 0, n_added = 0;
    bool stale_box = false;
    // BVHWrapper::new_wrapper(list) as a scene element (bvhwrapper.rs:15-32, scene/mod.rs:161-163): the list's visible spheres
    // and triangles (the library rebuilds the tree from them); none visible -> the empty list the reference returns.
    static Hittables bvh_wrapper(const Hittables& list) {
        list.need_list();
        Hittables h; h.kind = CR_PRIM_BVH; h.id = (size_t)-1;
        for (const Hittables& o : list.objs) {
            if (o.kind != CR_PRIM_SPHERE && o.kind != CR_PRIM_TRIANGLE) throw std::invalid_argument("a BVHWrapper element may hold spheres and triangles only");
            if (!o.hide) h.objs.push_back(o);
        }
        if (h.objs.empty()) return hit_list();
        return h;
    }
    static Hittables hit_list(std::vector<Hittables> objs = {}) {   // HitList::new / HitList::default
        Hittables h; h.kind = CR_PRIM_LIST; h.id = (size_t)-1; h.objs = std::move(objs); h.n_new = h.objs.size();
        return h;
    }
    void add(Hittables o) { need_list(); objs.push_back(std::move(o)); n_added++; }   // :24-27 (the object is cloned: later edits do not reach it)
    void clear() { need_list(); stale_box = stale_box || n_added > 0; objs.clear(); n_new = 0; }
    const std::vector<Hittables>& get_objs() const { return objs; }
    bool box_is_default() const { return n_added == 0 && !stale_box; }
    bool box_is_union() const {
        if (stale_box || n_new != 0) return false;
        for (const Hittables& o : objs) if (o.kind == CR_PRIM_LIST && !o.box_is_union()) return false;
        return true;
    }
    // the spheres and triangles under this list in visiting order; *empty_box = the descriptor's CR_LIST_EMPTY_BOX
    void spliced(std::vector<const Hittables*>& out, bool* empty_box) const {
        need_list();
        collect(out);
        if (box_is_default()) *empty_box = true;
        else if (box_is_union()) *empty_box = false;
        else throw std::invalid_argument("this HitList's box is neither Aabb::default() nor the union of its objects: not representable");
    }
private:
    void need_list() const { if (kind != CR_PRIM_LIST) throw std::invalid_argument("not a HitList"); }
    void collect(std::vector<const Hittables*>& out) const {
        for (const Hittables& o : objs) { if (o.kind == CR_PRIM_LIST) o.collect(out); else out.push_back(&o); }
    }
public:
    static Hittables sphere(Point3 c, double radius, MaterialPtr m) {   // Sphere::new, sphere.rs:25-39
        if (!(radius >= 0.0)) throw std::invalid_argument("Cannot make a sphere with negative radius");
        Hittables h; h.kind = CR_PRIM_SPHERE; h.v[0] = c.x; h.v[1] = c.y; h.v[2] = c.z; h.v[3] = radius; h.mat = m;
        h.timeline = TransformTimeline::new_sphere(c, radius);
        return h;
    }
    static Hittables triangle(Point3 a, Point3 b, Point3 c, MaterialPtr m) {   // Triangle::new, triangle.rs:23-46
        Hittables h; h.kind = CR_PRIM_TRIANGLE; h.mat = m; h.timeline = TransformTimeline(a);
        double vv[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
        std::copy(vv, vv + 9, h.v);
        return h;
    }
};

inline std::string build_asset_path(const std::string& name) {   // asset_loader/mod.rs:6-41
    if (const char* d = std::getenv("ASSET_DIR")) return std::string(d) + name;
    std::string base;
    for (int up = 0; up <= 6; up++) {
        std::string p = base + "assets/" + name;
        struct stat st;
        if (stat(p.c_str(), &st) == 0) return p;
        base += "../";
    }
    throw std::runtime_error("Could not find the asset " + name);
}

inline std::vector<Hittables> load_obj(const std::string& file, double scale, Point3 shift, MaterialPtr mat) {   // obj_loader.rs:21-143
    std::string path = build_asset_path(file);
    if (path.size() < 4 || path.substr(path.size() - 4) != ".obj") throw std::invalid_argument("Expected an obj file.");
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Cannot open OBJ file.");
    std::vector<Point3> verts;
    std::vector<std::vector<long>> faces;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::vector<std::string> tok;
        for (std::string t; ss >> t;) tok.push_back(t);
        if (tok.empty()) continue;
        if (tok[0] == "v") {
            if (tok.size() != 4) throw std::invalid_argument("Invalid number of coordinates for a vertex");
            verts.push_back(Point3{std::stod(tok[1]), std::stod(tok[2]), std::stod(tok[3])});
        } else if (tok[0] == "f") {
            if (tok.size() != 4) throw std::invalid_argument("The asset loader only supports triangularized images");
            faces.push_back({std::stol(tok[1]), std::stol(tok[2]), std::stol(tok[3])});
        } else throw std::invalid_argument("Unsupported OBJ file");
    }
    for (Point3& p : verts) p = Point3{scale * p.x + shift.x, scale * p.y + shift.y, scale * p.z + shift.z};
    std::vector<Hittables> out;
    for (auto& f : faces) out.push_back(Hittables::triangle(verts.at(f[0] - 1), verts.at(f[1] - 1), verts.at(f[2] - 1), mat));
    return out;
}

// ---------------------------------------------------------------- camera
class Camera {   // Camera::new + setters, camera/mod.rs:106-263
public:
    double aspect_ratio;
    int image_width, image_height;
    double vfov_degrees = 90.0, defocus_angle_degrees = 0.0, focus_dist = 10.0;
    TransformTimeline look_from_tl, look_at_tl;
    Vec3 vup{0, 1, 0};
    uint32_t samples = 10, max_depth = 10;
    size_t thread_count;
    double frame_rate, shutter_angle;
    size_t frame = 0;

    Camera(double aspect, uint32_t width, double rate, double shutter, size_t threads)
        : aspect_ratio(aspect), image_width((int)width), thread_count(threads), frame_rate(rate), shutter_angle(shutter) {
        image_height = std::max(1, (int)(uint32_t)((double)width / aspect));   // Viewport::new, camera/mod.rs:37-38
    }
    void next_frame() { frame += 1; }
    void look_from(Point3 p) { look_from_tl = TransformTimeline(p); }
    void look_at(Point3 p) { look_at_tl = TransformTimeline(p); }
    void set_vup(Vec3 v) { vup = v; }
    void set_vfov(double deg) { vfov_degrees = deg; }
    void set_samples(uint32_t s) {
        if (!(s > 0)) throw std::invalid_argument("The camera must have a positive number of samples.");
        samples = s;
    }
    void set_max_depth(uint32_t md) { max_depth = md; }
    void set_defocus_angle(double deg) { defocus_angle_degrees = deg; }
    void set_focus_dist(double fd) { focus_dist = fd; }
    void set_threads(size_t t) { thread_count = t; }
};

// ---------------------------------------------------------------- flat scene
struct FlatScene {
    std::vector<CrPrimitive> prims;
    std::vector<CrMaterial> materials;
    std::vector<CrTexture> textures;
    std::vector<CrImage> images;
    std::vector<CrKeyframe> keys;
    std::vector<std::shared_ptr<RTWImage>> image_keep;
    CrSceneDesc desc{};
    void finish(int sky_kind, int sky_image) {
        desc = CrSceneDesc{(int32_t)prims.size(), (int32_t)materials.size(), (int32_t)textures.size(), (int32_t)images.size(),
                           (int32_t)keys.size(), sky_kind, sky_image, 0, prims.data(), materials.data(), textures.data(),
                           images.data(), keys.data()};
    }
};

class SceneRng {   // SplitMix64, uniform = (u >> 11) * 2^-53; see DESIGN.md "RNG"
    uint64_t s;
public:
    explicit SceneRng(uint64_t seed) : s(seed) {}
    uint64_t u64() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double random() { return (double)(u64() >> 11) * 0x1.0p-53; }
    double random_range(double lo, double hi) { return lo + (hi - lo) * random(); }
};

// ---------------------------------------------------------------- scene
class Scene {
    std::map<std::string, std::pair<size_t, std::string>> aliases_{{"cam", {0, "Camera"}}};   // id_vendor.rs
    size_t next_id_ = 1;
    size_t vend_id(const std::string& alias, const std::string& otype) {
        if (aliases_.count(alias)) throw std::invalid_argument("This " + otype + "'s alias collides with another name in the scene! Try changing " + alias + " to a new name.");
        aliases_[alias] = {next_id_, otype};
        return next_id_++;
    }
    size_t lookup(const std::string& alias, std::initializer_list<const char*> invalid) const {
        auto it = aliases_.find(alias);
        if (it == aliases_.end()) throw std::out_of_range("Could not find an object with the alias: `" + alias + "`. Are you sure you spelled it right?");
        for (const char* t : invalid) if (it->second.second == t) throw std::invalid_argument("this transform cannot apply to a " + it->second.second);
        return it->second.first;
    }

public:
    Camera scene_cam;
    std::vector<Hittables> elements;
    std::shared_ptr<RTWImage> skybox;   // null = Skybox::Default
    bool is_movie = false;
    double duration = 0;
    size_t frame_rate;
    uint64_t seed = 0xC0FFEE;
    int real_type = CR_REAL_F32;
    int bvh_mode = CR_BVH_REFERENCE;   // CR_BVH_SAH: the quality builder (include/crucible_hip.h)
    bool refit_boxes = false;          // CrRenderParams.refit_boxes: wrapper boxes follow keyframed primitives
    std::string frame_format = "ppm";  // "ppm" (ASCII P3, the reference's) | "p6" | "png"
    int device = 0;
    int gpus = 1;            // > 1 (or use_group): still images split their samples over devices 0..gpus-1 through cr_group_*;
    bool use_group = false;  // movies give frame f to device f % gpus (scene/mod.rs:307-316: frames are independent)

    Scene(double aspect, uint32_t width, size_t rate, double shutter, size_t threads)
        : scene_cam(aspect, width, (double)rate, shutter, threads), frame_rate(rate) {}
    static Scene new_image(double aspect, uint32_t width, size_t rate, double shutter, size_t threads) { return Scene(aspect, width, rate, shutter, threads); }
    static Scene new_movie(double aspect, uint32_t width, size_t rate, double shutter, size_t threads, double dur) {
        Scene s(aspect, width, rate, shutter, threads); s.is_movie = true; s.duration = dur; return s;
    }
    void load_default_skybox() { skybox.reset(); }
    void load_spherical_skybox(std::shared_ptr<RTWImage> im) { skybox = im; }
    void add_element(Hittables e, const std::string& alias) {   // scene/mod.rs:159-188
        if (e.kind == CR_PRIM_LIST || e.kind == CR_PRIM_BVH) { elements.push_back(std::move(e)); return; }   // :161-166: kept as it is, the alias is not registered
        e.id = vend_id(alias, e.kind == CR_PRIM_SPHERE ? "Sphere" : "Triangle");
        elements.push_back(std::move(e));
    }
    void load_asset(const std::string& path, const std::string& alias, double scale, Point3 shift, MaterialPtr mat) {   // :191-230
        size_t id = vend_id(alias, "TriangleMesh");
        for (Hittables& t : load_obj(path, scale, shift, mat)) { t.id = id; elements.push_back(std::move(t)); }
    }
    void set_visibility(const std::string& alias, bool hide) {   // :241-278
        auto it = aliases_.find(alias);
        if (it == aliases_.end()) { fprintf(stderr, "WARNING: The element `%s` does not exist. Are you sure you typed the right name?\n", alias.c_str()); return; }
        for (Hittables& e : elements) if (e.id == it->second.first) e.hide = hide;
    }
    void show_element(const std::string& a) { set_visibility(a, false); }
    void hide_element(const std::string& a) { set_visibility(a, true); }
    // scene_animator.rs
    void translate_point(Point3 p, double k, InterpolationType it, TransformSpace sp, const std::string& alias) {
        size_t id = lookup(alias, {"Camera"});
        for (Hittables& e : elements) if (e.id == id) e.timeline.translate_point(p, k, it, sp);
    }
    void scale_r(double r, double k, InterpolationType it, const std::string& alias) {   // :140-150
        size_t id = lookup(alias, {"Camera", "TriangleMesh", "Triangle"});
        for (Hittables& e : elements) if (e.id == id) e.timeline.scale_sphere(r, k, it);
    }
    // ScaleX / ScaleY / ScaleZ / ScaleAll: invalid on spheres, applied to the alias's triangles (:38-229)
    void scale_x(double x, double k, InterpolationType it, const std::string& alias) {
        size_t id = lookup(alias, {"Sphere"});
        for (Hittables& e : elements) if (e.id == id && e.kind == CR_PRIM_TRIANGLE) e.timeline.scale_x(x, k, it);
    }
    void scale_y(double y, double k, InterpolationType it, const std::string& alias) {
        size_t id = lookup(alias, {"Sphere"});
        for (Hittables& e : elements) if (e.id == id && e.kind == CR_PRIM_TRIANGLE) e.timeline.scale_y(y, k, it);
    }
    void scale_z(double z, double k, InterpolationType it, const std::string& alias) {
        size_t id = lookup(alias, {"Sphere"});
        for (Hittables& e : elements) if (e.id == id && e.kind == CR_PRIM_TRIANGLE) e.timeline.scale_z(z, k, it);
    }
    void scale_point(Point3 p, double k, InterpolationType it, const std::string& alias) {
        size_t id = lookup(alias, {"Sphere"});
        for (Hittables& e : elements) if (e.id == id && e.kind == CR_PRIM_TRIANGLE) e.timeline.scale_point(p, k, it);
    }
    void scale_all_uniform(double v, double k, InterpolationType it, const std::string& alias) { scale_point(Point3{v, v, v}, k, it, alias); }   // :217-219
    void cam_translate_point(Point3 p, double k, InterpolationType it, TransformSpace sp, const std::string& which) {   // :532-556
        if (which != "from" && which != "at") throw std::invalid_argument("alias must be 'from' or 'at'");
        (which == "from" ? scene_cam.look_from_tl : scene_cam.look_at_tl).translate_point(p, k, it, sp);
    }

    // ---- flatten to the C ABI (children before parents; materials/textures shared by pointer)
    FlatScene flatten() const {
        FlatScene f;
        std::map<const void*, int> tex_ids, mat_ids, img_ids;
        auto image_id = [&](const std::shared_ptr<RTWImage>& im) {
            auto it = img_ids.find(im.get());
            if (it != img_ids.end()) return it->second;
            int id = (int)f.images.size();
            f.images.push_back(CrImage{im->width, im->height, im->rgb8.data()});
            f.image_keep.push_back(im);
            return img_ids[im.get()] = id;
        };
        std::function<int(const TexturePtr&)> texture_id = [&](const TexturePtr& t) -> int {
            auto it = tex_ids.find(t.get());
            if (it != tex_ids.end()) return it->second;
            CrTexture rec{t->kind, -1, -1, -1, {0, 0, 0}, 0.0};
            if (t->kind == CR_TEX_SOLID) { rec.color[0] = t->albedo.r; rec.color[1] = t->albedo.g; rec.color[2] = t->albedo.b; }
            else if (t->kind == CR_TEX_CHECKER) { rec.even = texture_id(t->even); rec.odd = texture_id(t->odd); rec.inv_scale = t->inv_scale; }
            else rec.image = image_id(t->image);
            int id = (int)f.textures.size();
            f.textures.push_back(rec);
            return tex_ids[t.get()] = id;
        };
        auto material_id = [&](const MaterialPtr& m) {
            auto it = mat_ids.find(m.get());
            if (it != mat_ids.end()) return it->second;
            CrMaterial rec{m->kind, -1, {0, 0, 0}, m->param};
            if (m->kind == CR_MAT_LAMBERTIAN) rec.texture = texture_id(m->tex);
            else { rec.albedo[0] = m->albedo.r; rec.albedo[1] = m->albedo.g; rec.albedo[2] = m->albedo.b; }
            int id = (int)f.materials.size();
            f.materials.push_back(rec);
            return mat_ids[m.get()] = id;
        };
        auto emit = [&](const Hittables& e, int32_t extra_flags) {
            std::vector<CrKeyframe> ks = e.timeline.keyframes();
            CrPrimitive p{e.kind, material_id(e.mat), (e.hide ? CR_PRIM_HIDDEN : 0) | extra_flags, (int32_t)f.keys.size(), (int32_t)ks.size(), 0, {0}};
            std::copy(e.v, e.v + 9, p.v);
            f.prims.push_back(p);
            f.keys.insert(f.keys.end(), ks.begin(), ks.end());
        };
        for (const Hittables& e : elements) {
            if (e.kind == CR_PRIM_BVH) {   // the wrapper record, then its objects (crucible_hip.h CR_PRIM_BVH)
                CrPrimitive p{CR_PRIM_BVH, 0, 0, 0, 0, 0, {0}};
                p.v[0] = (double)(f.prims.size() + 1); p.v[1] = (double)e.objs.size();
                f.prims.push_back(p);
                for (const Hittables& o : e.objs) emit(o, CR_PRIM_MEMBER);
            } else if (e.kind == CR_PRIM_LIST) {   // the list record, then its objects (crucible_hip.h CR_PRIM_LIST)
                std::vector<const Hittables*> objs;
                bool empty_box = false;
                e.spliced(objs, &empty_box);
                CrPrimitive p{CR_PRIM_LIST, 0, empty_box ? CR_LIST_EMPTY_BOX : 0, 0, 0, 0, {0}};
                p.v[0] = (double)(f.prims.size() + 1); p.v[1] = (double)objs.size();
                f.prims.push_back(p);
                for (const Hittables* o : objs) emit(*o, CR_PRIM_MEMBER);
            } else emit(e, 0);
        }
        int sky_kind = CR_SKY_DEFAULT, sky_image = -1;
        if (skybox) { sky_kind = CR_SKY_SPHERICAL; sky_image = image_id(skybox); }
        f.finish(sky_kind, sky_image);
        f.desc.bvh_mode = bvh_mode;
        return f;
    }

    // movie_maker::make_mp4's ffmpeg invocation (scene/movie_maker.rs:6-33) for the frames render_movie wrote, as an
    // argument vector.  The reference runs it; this mirror hands it to the caller ("optional
    // ffmpeg hand-off"): a process that has initialised the GPU must not exec another program on this pool.
    std::vector<std::string> mp4_command(const std::string& fname, size_t padding) const {
        const std::string ext = frame_format == "png" ? ".png" : ".ppm";
        return {"ffmpeg", "-framerate", std::to_string(frame_rate), "-i", fname + "/artifacts/image%0" + std::to_string(padding) + "d" + ext,
                "-vf", "scale=trunc(iw/2)*2:trunc(ih/2)*2", "-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", "25", fname + "/movie.mp4"};
    }

    size_t compute_frame_count() const { return (size_t)std::ceil(duration * (double)frame_rate); }   // scene/mod.rs:324-330

    // ---- Camera::render over the library (scene/mod.rs:283-347)
    // Wall-clock phases of the last render_scene call (a measurement aid: the reference's one benchmark times the whole
    // of render_scene -- BVH build, render, file -- benches/renderer_benchmark.rs:16-42).
    struct Timing { double create_ms = 0, flatten_ms = 0, upload_ms = 0, bvh_build_ms = 0, render_ms = 0, write_ms = 0, total_ms = 0, kernel_ms = 0; size_t frames = 0; };
    Timing timing;
    int sum_order = CR_SUM_DEFAULT;    // CrRenderParams.sum_order
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    CrCameraDesc camera_desc(std::vector<CrKeyframe>& fk, std::vector<CrKeyframe>& ak) const {
        const Camera& c = scene_cam;
        fk = c.look_from_tl.keyframes(); ak = c.look_at_tl.keyframes();
        return CrCameraDesc{c.image_width, c.image_height, c.vfov_degrees, c.defocus_angle_degrees, c.focus_dist,
                            {c.look_from_tl.start_pos.x, c.look_from_tl.start_pos.y, c.look_from_tl.start_pos.z},
                            {c.look_at_tl.start_pos.x, c.look_at_tl.start_pos.y, c.look_at_tl.start_pos.z},
                            {c.vup.x, c.vup.y, c.vup.z}, (int32_t)fk.size(), (int32_t)ak.size(), fk.data(), ak.data()};
    }
    CrRenderParams render_params(size_t frame) const {
        const Camera& c = scene_cam;
        CrRenderParams p{(int32_t)c.samples, 0, (int32_t)c.samples, (int32_t)c.max_depth, seed, (int32_t)frame, real_type,
                         c.frame_rate, c.shutter_angle, 0, refit_boxes ? 1 : 0, sum_order, 0};
        return p;
    }
    // The render proper: W*H*3 reals of `real_type` into `buf` (sized for either scalar type), at the camera's frame
    // counter or at `frame`.
    int32_t render_frame(CrHandle* h, std::vector<double>& buf, CrStats* stats = nullptr, int64_t frame = -1) const {
        std::vector<CrKeyframe> fk, ak;
        CrCameraDesc cd = camera_desc(fk, ak);
        CrRenderParams p = render_params(frame < 0 ? (size_t)scene_cam.frame : (size_t)frame);
        buf.resize((size_t)scene_cam.image_width * scene_cam.image_height * 3);
        return cr_render_host(h, &cd, &p, buf.data(), stats);
    }
    // The file: "ppm" = the reference's ASCII P3 (camera/mod.rs:286,306-311); "p6" / "png" = SURVEY 8(f) row 3,
    // the same per-channel bytes in binary PPM / PNG.
    int32_t write_frame(const std::string& fname, const std::vector<double>& buf) const {
        const Camera& c = scene_cam;
        const std::string path = fname + (frame_format == "png" ? ".png" : ".ppm");
        int32_t rc = frame_format == "png" ? cr_write_png(path.c_str(), buf.data(), real_type, c.image_width, c.image_height)
                   : frame_format == "p6" ? cr_write_ppm_binary(path.c_str(), buf.data(), real_type, c.image_width, c.image_height)
                                          : cr_write_ppm(path.c_str(), buf.data(), real_type, c.image_width, c.image_height);
        if (rc == CR_OK && !quiet) fprintf(stderr, "Successful render! Image stored at: %s\n", path.c_str());
        return rc;
    }
    bool quiet = false;   // no "Successful render!" line per frame
    int32_t render_image(CrHandle* h, const std::string& fname, CrStats* stats = nullptr) {
        std::vector<double> buf;
        double t0 = now_ms();
        int32_t rc = render_frame(h, buf, stats);
        double t1 = now_ms();
        timing.render_ms += t1 - t0;
        if (rc == CR_OK) { rc = write_frame(fname, buf); timing.write_ms += now_ms() - t1; timing.frames++; }
        return rc;
    }
    // Camera::render across several devices: cr_group_render_host splits the sample indices, adds the per-pixel sums with
    // one RCCL reduce inside the library and hands back the mean.
    int32_t render_frame_group(CrGroup* g, std::vector<double>& buf, CrStats* stats = nullptr) const {
        std::vector<CrKeyframe> fk, ak;
        CrCameraDesc cd = camera_desc(fk, ak);
        CrRenderParams p = render_params((size_t)scene_cam.frame);
        buf.resize((size_t)scene_cam.image_width * scene_cam.image_height * 3);
        CrGroupStats gs;
        int32_t rc = cr_group_render_host(g, &cd, &p, buf.data(), &gs);
        if (stats) *stats = gs.render;
        return rc;
    }
    // render_movie (scene/mod.rs:295-322) for the frames first, first + step, ... on one handle: frame k is encoded and
    // written by a helper thread while frame k+1 renders (two buffers; SURVEY 8(f) row 3 -- the reference formats and
    // writes each frame before starting the next).  render_ms: time this thread spent inside renders; write_ms: time the
    // helper threads spent encoding and writing (overlapped, so it is not part of the wall clock unless it is the longer one).
    int32_t render_movie_frames(CrHandle* h, const std::string& fname, size_t first, size_t step, size_t frames, size_t digits,
                                CrStats* stats, Timing* tm) const {
        std::vector<double> bufs[2];
        std::thread writers[2];
        int32_t write_rc[2] = {CR_OK, CR_OK};
        double write_ms[2] = {0, 0};
        int32_t rc = CR_OK;
        size_t k = 0;
        for (size_t fr = first; rc == CR_OK && fr < frames; fr += step, k++) {
            const int slot = (int)(k & 1);
            if (writers[slot].joinable()) { writers[slot].join(); if (write_rc[slot] != CR_OK) { rc = write_rc[slot]; break; } }
            std::string num = std::to_string(fr);
            num = std::string(digits - num.size(), '0') + num;
            CrStats st;
            const double t0 = now_ms();
            rc = render_frame(h, bufs[slot], &st, (int64_t)fr);
            if (tm) { tm->render_ms += now_ms() - t0; tm->kernel_ms += st.kernel_ms; tm->frames++; }
            if (stats) *stats = st;
            if (rc != CR_OK) break;
            const std::string stem = fname + "/artifacts/image" + num;
            writers[slot] = std::thread([this, stem, slot, &bufs, &write_rc, &write_ms] {
                const double w0 = now_ms();
                write_rc[slot] = write_frame(stem, bufs[slot]);
                write_ms[slot] += now_ms() - w0;
            });
        }
        for (int s2 = 0; s2 < 2; s2++) if (writers[s2].joinable()) { writers[s2].join(); if (rc == CR_OK) rc = write_rc[s2]; }
        if (tm) tm->write_ms += write_ms[0] + write_ms[1];
        return rc;
    }
    void print_mp4_command(const std::string& fname, size_t digits) const {   // scene/mod.rs:319 would run this
        std::string cmd;
        for (const std::string& a : mp4_command(fname, digits)) cmd += (cmd.empty() ? "" : " ") + (a.find_first_of("*()") != std::string::npos ? "'" + a + "'" : a);
        fprintf(stderr, "Frames written. To assemble the movie: %s\n", cmd.c_str());
    }
    // Several devices.  A still image splits its samples through cr_group_* (one collective).  A movie needs no collective:
    // frames are independent (scene/mod.rs:307-316), so device m gets its own plain handle and its own host thread for the
    // frames m, m + n, ... -- no communicator, no RCCL.
    int32_t render_scene_group(const std::string& fname, CrStats* stats) {
        const int n = std::max(1, gpus);
        FlatScene f = flatten();
        if (!is_movie) {
            std::vector<int32_t> ids;
            for (int i = 0; i < n; i++) ids.push_back(device + i);
            CrGroup* g = nullptr;
            int32_t rc = cr_group_create(ids.data(), (int32_t)ids.size(), &g);
            if (rc != CR_OK) { fprintf(stderr, "Render failed. %s\n", cr_group_last_error(nullptr)); return rc; }
            rc = cr_group_upload_scene(g, &f.desc);
            std::vector<double> buf;
            if (rc == CR_OK) rc = render_frame_group(g, buf, stats);
            if (rc == CR_OK) rc = write_frame(fname, buf);
            if (rc != CR_OK) fprintf(stderr, "Render failed. %s\n", cr_group_last_error(g));
            cr_group_destroy(g);
            return rc;
        }
        if (mkdir(fname.c_str(), 0777) != 0 || mkdir((fname + "/artifacts").c_str(), 0777) != 0) { fprintf(stderr, "Render failed. cannot create %s\n", fname.c_str()); return CR_ERR_IO; }
        const size_t frames = compute_frame_count(), digits = std::to_string(frames).size();
        std::vector<int32_t> member_rc((size_t)n, CR_OK);
        std::vector<std::string> member_err((size_t)n);
        std::vector<Timing> member_tm((size_t)n);
        std::vector<std::thread> workers;
        for (int m = 0; m < n; m++)
            workers.emplace_back([this, m, n, frames, digits, &fname, &f, &member_rc, &member_err, &member_tm] {
                CrHandle* h = nullptr;
                int32_t rc = cr_create(device + m, &h);
                if (rc != CR_OK) { member_rc[(size_t)m] = rc; member_err[(size_t)m] = cr_last_error(nullptr); return; }
                rc = cr_upload_scene(h, &f.desc);
                if (rc == CR_OK) rc = render_movie_frames(h, fname, (size_t)m, (size_t)n, frames, digits, nullptr, &member_tm[(size_t)m]);
                if (rc != CR_OK) { member_rc[(size_t)m] = rc; member_err[(size_t)m] = cr_last_error(h); }   // the failing member's own message
                cr_destroy(h);
            });
        for (std::thread& t : workers) t.join();
        int32_t rc = CR_OK;
        for (int m = 0; m < n; m++) {
            timing.render_ms = std::max(timing.render_ms, member_tm[(size_t)m].render_ms); timing.write_ms = std::max(timing.write_ms, member_tm[(size_t)m].write_ms);
            timing.kernel_ms += member_tm[(size_t)m].kernel_ms; timing.frames += member_tm[(size_t)m].frames;
            if (member_rc[(size_t)m] != CR_OK) { fprintf(stderr, "Render failed on device %d. %s\n", device + m, member_err[(size_t)m].c_str()); if (rc == CR_OK) rc = member_rc[(size_t)m]; }
        }
        if (rc == CR_OK) print_mp4_command(fname, digits);
        return rc;
    }
    int32_t render_scene(const std::string& fname, CrStats* stats = nullptr) {
        timing = Timing();
        const double t_begin = now_ms();
        int32_t rc;
        if (gpus > 1 || use_group) rc = render_scene_group(fname, stats);
        else {
            CrHandle* h = nullptr;
            double t0 = now_ms();
            rc = cr_create(device, &h);
            timing.create_ms = now_ms() - t0;
            if (rc != CR_OK) { fprintf(stderr, "Render failed. %s\n", cr_last_error(nullptr)); return rc; }
            t0 = now_ms();
            FlatScene f = flatten();
            timing.flatten_ms = now_ms() - t0;
            t0 = now_ms();
            rc = cr_upload_scene(h, &f.desc);
            timing.upload_ms = now_ms() - t0;
            if (rc == CR_OK) {
                if (!is_movie) { CrStats st; memset(&st, 0, sizeof st); rc = render_image(h, fname, &st); timing.kernel_ms = st.kernel_ms; timing.bvh_build_ms = std::max(0.0, st.upload_ms - timing.upload_ms); if (stats) *stats = st; }
                else {   // render_movie, scene/mod.rs:295-322 (the ffmpeg hand-off is out of scope)
                    if (mkdir(fname.c_str(), 0777) != 0 || mkdir((fname + "/artifacts").c_str(), 0777) != 0) rc = CR_ERR_IO;
                    const size_t frames = compute_frame_count(), digits = std::to_string(frames).size();
                    if (rc == CR_OK) rc = render_movie_frames(h, fname, 0, 1, frames, digits, stats, &timing);
                    if (rc == CR_OK) print_mp4_command(fname, digits);
                }
            }
            if (rc != CR_OK) fprintf(stderr, "Render failed. %s\n", cr_last_error(h));
            cr_destroy(h);
        }
        timing.total_ms = now_ms() - t_begin;
        return rc;
    }
};

// ---------------------------------------------------------------- demo scenes (demo_images.rs), seeded
namespace demo_builder {

inline double clamp01(double x) { return std::min(std::max(x, 0.0), 1.0); }
inline MaterialPtr checker_ground() {
    return Materials::lambertian(Textures::checker(0.32, Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)), 1.0);
}
inline MaterialPtr small_sphere_material(SceneRng& rng, double choose_mat) {   // demo_images.rs:58-82
    if (choose_mat < 0.8) {
        double c1[3] = {rng.random(), rng.random(), rng.random()};
        double c2[3] = {rng.random(), rng.random(), rng.random()};
        return Materials::lambertian(Color(clamp01(c1[0] * c2[0]), clamp01(c1[1] * c2[1]), clamp01(c1[2] * c2[2])), 1.0);
    }
    if (choose_mat < 0.95) {
        double r = rng.random_range(0.5, 1.0), g = rng.random_range(0.5, 1.0), b = rng.random_range(0.5, 1.0);
        double fuzz = rng.random_range(0.0, 0.5);
        return Materials::metal(Color(r, g, b), fuzz);
    }
    return Materials::dielectric(1.5);
}
inline void book1_camera(Camera& cam, uint32_t samples, Point3 from) {
    cam.set_samples(samples); cam.set_max_depth(50);
    cam.look_from(from); cam.look_at(Point3{0, 0, 0});
    cam.set_vfov(20.0); cam.set_defocus_angle(0.6); cam.set_focus_dist(10.0);
}
inline Scene book1_end_scene(size_t threads, uint64_t scene_seed = 1, uint32_t image_width = 400, uint32_t samples = 500) {   // :14-109
    Scene sc = Scene::new_image(16.0 / 9.0, image_width, 24, 180.0, threads);
    book1_camera(sc.scene_cam, samples, Point3{13, 2, 3});
    sc.add_element(Hittables::sphere(Point3{0, -1000, 0}, 1000.0, checker_ground()), "ground");
    SceneRng rng(scene_seed);
    int counter = 0;
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            double choose_mat = rng.random();
            double cx = a + 0.9 * rng.random();
            double cz = b + 0.9 * rng.random();
            double dx = cx - 4.0, dy = 0.2 - 0.2, dz = cz - 0.0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
                sc.add_element(Hittables::sphere(Point3{cx, 0.2, cz}, 0.2, small_sphere_material(rng, choose_mat)), "small" + std::to_string(counter));
                counter++;
            }
        }
    sc.add_element(Hittables::sphere(Point3{0, 1, 0}, 1.0, Materials::dielectric(1.5)), "large_dielectric");
    sc.add_element(Hittables::sphere(Point3{-4, 1, 0}, 1.0, Materials::lambertian(Color(0.4, 0.2, 0.1), 1.0)), "large_lambertian");
    sc.add_element(Hittables::sphere(Point3{4, 1, 0}, 1.0, Materials::metal(Color(0.7, 0.6, 0.5), 0.0)), "large_metal");
    return sc;
}
inline Scene checkered_spheres(size_t threads, uint32_t image_width = 400, uint32_t samples = 500) {   // :112-152
    Scene sc = Scene::new_image(16.0 / 9.0, image_width, 24, 180.0, threads);
    book1_camera(sc.scene_cam, samples, Point3{13, 2, 3});
    TexturePtr checker = Textures::checker(0.32, Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9));
    sc.add_element(Hittables::sphere(Point3{0, -10, 0}, 10.0, Materials::lambertian(checker, 1.0)), "bottom_sphere");
    sc.add_element(Hittables::sphere(Point3{0, 10, 0}, 10.0, Materials::lambertian(checker, 1.0)), "top_sphere");
    return sc;
}
inline Scene load_teapot(size_t threads, uint32_t image_width = 400, uint32_t samples = 200) {   // :155-200
    Scene sc = Scene::new_image(16.0 / 9.0, image_width, 24, 180.0, threads);
    book1_camera(sc.scene_cam, samples, Point3{13, 10, 3});
    sc.load_asset("teapot.obj", "teapot", 0.5, Point3{0, 0, 0}, Materials::metal(Color(0.8, 0.3, 0.5), 0.05));
    sc.add_element(Hittables::sphere(Point3{0, -1000, 0}, 1000.0, checker_ground()), "ground");
    return sc;
}

// The teapot with every non-sphere scale builder on it (scene_animator.rs:38-229) plus a translation: what
// demo_movies::moving_teapot (demo_movies.rs:125) is after -- it calls scale_r on the mesh, which the reference's own
// type check rejects.  Keys fall inside the first frames' shutter intervals so a still image shows them.
inline Scene scaled_teapot(size_t threads, uint32_t image_width = 400, uint32_t samples = 200) {
    Scene sc = load_teapot(threads, image_width, samples);
    sc.scale_x(1.4, 0.012, InterpolationType::LERP, "teapot");
    sc.scale_y(0.25, 0.016, InterpolationType::NERP, "teapot");
    sc.translate_point(Point3{0.0, 0.4, 0.3}, 0.02, InterpolationType::LERP, TransformSpace::Local, "teapot");
    sc.scale_all_uniform(1.2, 0.05, InterpolationType::LERP, "teapot");
    sc.scale_z(0.7, 0.09, InterpolationType::LERP, "teapot");
    return sc;
}

// Not in the reference's demos: the teapot handed to the scene as ONE HitList element (what
// `scene.add_element(Hittables::HitList(load_obj(..)), ..)` gives, scene/mod.rs:164-166) next to a HitList::new(vec)
// list (empty box), a list inside a list with a hidden object, and ordinary elements.
inline Scene teapot_as_list(size_t threads, uint32_t image_width = 400, uint32_t samples = 200) {
    Scene sc = Scene::new_image(16.0 / 9.0, image_width, 24, 180.0, threads);
    book1_camera(sc.scene_cam, samples, Point3{13, 10, 3});
    Hittables pot = Hittables::hit_list();
    for (Hittables& t : load_obj("teapot.obj", 0.5, Point3{0, 0, 0}, Materials::metal(Color(0.8, 0.3, 0.5), 0.05))) pot.add(std::move(t));
    sc.add_element(std::move(pot), "teapot");
    sc.add_element(Hittables::sphere(Point3{0, -1000, 0}, 1000.0, checker_ground()), "ground");
    MaterialPtr glass = Materials::dielectric(1.5), matte = Materials::lambertian(Color(0.2, 0.4, 0.8), 1.0);
    sc.add_element(Hittables::hit_list({Hittables::sphere(Point3{2.5, 0.5, 2.0}, 0.5, glass), Hittables::sphere(Point3{3.4, 0.3, 1.2}, 0.3, matte)}), "loose");
    Hittables inner = Hittables::hit_list();
    inner.add(Hittables::sphere(Point3{-2.0, 0.4, 2.5}, 0.4, matte));
    Hittables hidden = Hittables::sphere(Point3{-2.0, 1.2, 2.5}, 0.4, glass);
    hidden.hide = true;
    inner.add(hidden);
    Hittables outer = Hittables::hit_list();
    outer.add(Hittables::sphere(Point3{-3.0, 0.5, 1.5}, 0.5, Materials::metal(Color(0.8, 0.8, 0.8), 0.0)));
    outer.add(inner);
    sc.add_element(std::move(outer), "outer");
    sc.add_element(Hittables::hit_list(), "nothing");
    // ... and a pre-built BVHWrapper as an element (scene/mod.rs:161-163)
    sc.add_element(Hittables::bvh_wrapper(Hittables::hit_list({Hittables::sphere(Point3{1.5, 0.3, 3.0}, 0.3, matte), Hittables::sphere(Point3{0.6, 0.3, 3.4}, 0.3, glass),
                                                                Hittables::sphere(Point3{-0.4, 0.3, 3.6}, 0.3, matte)})), "wrapped");
    return sc;
}

inline Scene garden_skybox(size_t threads, std::shared_ptr<RTWImage> sky, uint32_t image_width = 1920, uint32_t samples = 500) {   // :223-242
    Scene sc = Scene::new_image(16.0 / 9.0, image_width, 24, 180.0, threads);
    Camera& cam = sc.scene_cam;
    cam.set_samples(samples); cam.set_max_depth(50);
    cam.look_from(Point3{0, 0, -12}); cam.look_at(Point3{0, 0, 0});
    cam.set_vfov(40.0);
    sc.add_element(Hittables::sphere(Point3{0, 0, 0}, 2.0, Materials::metal(Color(0.8, 0.8, 0.8), 0.05)), "metal_ball");
    sc.load_spherical_skybox(sky);
    return sc;
}

}   // namespace demo_builder
}   // namespace crucible
