// capi.hip -- C ABI of include/crucible_hip.h over the gfx950 kernels in pathtrace.hpp.
// Host side of the boundary: deep-copies the scene description, filters hidden
// primitives, builds the BVH in the reference's topology, lays the scene out for the
// device, launches the persistent kernel and writes PPM P3.
//
// Nothing here falls back to a CPU renderer: without a HIP device cr_create fails.
#include "../../include/crucible_hip.h"
#include "pathtrace.hpp"
#include "wavefront.hpp"
#include "queue.hpp"
#include "refit.hpp"
#include "lbvh.hpp"
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include <zlib.h>

using namespace cr;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;      // what users address: raw + pad
    void* raw = nullptr;    // the allocation (hipMalloc aligns it to 256 bytes)
    size_t bytes = 0, pad = 0;
    // pad: bytes skipped at the front, so that p is deliberately MISaligned by that much (see entry_pad)
    hipError_t ensure(size_t n, size_t front_pad = 0) {
        if (n <= bytes && front_pad == pad) return hipSuccess;
        if (raw) (void)hipFree(raw);
        p = raw = nullptr; bytes = 0; pad = 0;
        hipError_t e = hipMalloc(&raw, n + front_pad);
        if (e == hipSuccess) { bytes = n; pad = front_pad; p = (char*)raw + front_pad; }
        return e;
    }
    void release() { if (raw) (void)hipFree(raw); p = raw = nullptr; bytes = 0; pad = 0; }
};
// Sibling wrappers are adjacent in the level-order array and start at ODD indices (1,2), (3,4), ...  Skipping one
// entry at the front of the allocation puts every pair on one 2*sizeof(Entry) boundary: the two children of a
// wrapper then share a cache line (f64: exactly one 128-byte line), so the walk's left-then-right visits touch it once.
template <typename E> constexpr size_t entry_pad() { return (sizeof(E) & (sizeof(E) - 1)) == 0 ? sizeof(E) : 0; }

template <typename real> struct DevScene {
    bool built = false;
    DevBuf entries, prims, mats, texs, keys;
    DevBuf leaf_runs;                        // (first, count) of the primitive runs that leaves holding a list name
    int32_t n_entries = 0, n_prims = 0, n_mats = 0, n_texs = 0, n_scene_keys = 0;
    size_t lds_bytes = 0;
    bool animated = false;
    bool has_triangles = false;
    bool has_spheres = false;
    bool has_leaf_runs = false;              // some leaf names its primitives through leaf_runs (a HitList element)
    bool has_bvh_elements = false;           // CR_BVH_REFERENCE over a BVHWrapper element: the records are not the reference's wrappers one to one (no export)
    bool has_lists = false;                  // the tree was built over at least one HitList element: its construction-time box
                                             // (empty, or grown over hidden objects too) is not what refit derives
    DevBuf entries_refit;                    // working copy whose boxes refit_level_kernel rewrites per frame
    DevBuf screen, screen_refit;             // f64, unordered trees: the f32 screening records of entries / entries_refit
    bool ordered = false;                    // CR_BVH_SAH_ORDERED: `entries` holds EntryO records
    size_t entry_bytes = sizeof(Entry<real>);
    std::vector<int8_t> host_axis;           // ordered: split axis per wrapper (-1 leaf), same order as host_entries
    std::vector<int32_t> level_begin;        // entries of tree level l are [level_begin[l], level_begin[l+1])
    std::vector<Entry<real>> host_entries;   // the tree over the scene's objects (for cr_export_bvh); the device copy names primitive runs
    std::vector<int32_t> leaf_desc;          // leaf-order position -> index in the caller's primitive list
    void release() { entries.release(); entries_refit.release(); screen.release(); screen_refit.release(); leaf_runs.release(); prims.release(); mats.release(); texs.release(); keys.release(); built = false; }
};

}   // namespace

struct CrHandle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 0;
    std::string error;
    // host copy of the scene description
    bool has_scene = false;
    std::vector<CrPrimitive> prims;
    std::vector<CrMaterial> materials;
    std::vector<CrTexture> textures;
    std::vector<CrKeyframe> keys;
    int32_t sky_kind = 0, sky_image = -1, bvh_mode = 0;
    // images are precision independent
    DevBuf images, texels;
    int32_t n_images = 0;
    DevScene<float> s32;
    DevScene<double> s64;
    DevBuf work_counter, counters, att_stack, out_buf;
    // Camera keyframes of a render travel in a ring of per-launch slots: a pinned host slot is filled, copied to its
    // device slot on the handle's stream and kept until that copy's event has fired, so back-to-back asynchronous
    // renders (a movie's frames) never see each other's keys.
    static constexpr int kCamSlots = 4;
    static constexpr size_t kMaxCamKeys = 512;
    void* cam_host[kCamSlots] = {};
    DevBuf cam_dev[kCamSlots];
    hipEvent_t cam_ev[kCamSlots] = {};
    int cam_next = 0, cam_pending_slot = -1;
    DevBuf sample_buf, sg_acc;   // sample-granular megakernel: per-sample colours of a batch, running sums between batches
    DevBuf fx_acc;               // CR_SUM_RELAXED: per-pixel fixed-point sums (3 x u64 per pixel)
    int screen_boxes = 1;        // f64, unordered trees: box tests decided on f32 screening records where f32 can (CRUCIBLE_SCREEN=0: never)
    int screen_lds = 1;          // ... also for scenes that sit in LDS whole (CRUCIBLE_SCREEN_LDS=0: only trees read from global memory)
    int default_sum_order = CR_SUM_RELAXED;   // what CR_SUM_DEFAULT means on this handle (CRUCIBLE_SUM_ORDER=reference|relaxed)
    // wavefront pipeline state (wavefront.hpp)
    DevBuf wf_job, wf_rng, wf_ray, wf_depth, wf_hit_t, wf_hit_prim, wf_chunk, wf_ctrl, wf_samples, wf_acc;
    uint32_t* wf_ring_host = nullptr;   // host-mapped ring the extend kernel reports its queue length into
    uint32_t* wf_ring_dev = nullptr;
    hipEvent_t wf_ev[8] = {};
    int pipeline = 0;                   // 0 = megakernel (default), 1 = wavefront kernels, 2 = LDS-queue megakernel (CRUCIBLE_PIPELINE=mega|wavefront|queue)
    int queue_walk_waves = 9, queue_min_batch = 48, queue_patience = 64;
    uint32_t wf_slots = 1u << 21;
    size_t wf_sample_bytes = (size_t)1600 << 20;
    int wf_last_iterations = 0;
    double upload_ms = 0;
    size_t lds_limit = 160 * 1024;
    // Sample-granular scheduling of the megakernel (pathtrace.hpp, KernelArgs::sg_on): on by default; the per-sample
    // colour buffer may take up to sample_buf_limit bytes (more samples than fit are rendered in batches).
    int sample_granular = 1;             // CRUCIBLE_SAMPLE_GRANULAR=0: a lane owns a pixel (no buffer)
    size_t sample_buf_limit = (size_t)40 << 30;   // CRUCIBLE_SAMPLE_BUF_MB (MI355X: 288 GB of HBM)
    int sg_chunk_override = 0;           // CRUCIBLE_SG_CHUNK: items per atomic (default: by launch size)
    uint64_t work_counter_max = 0xF0000000ull;   // work items one launch may hand out (32-bit counter); more samples run as consecutive launches
                                         // (CRUCIBLE_WORK_COUNTER_MAX: tests shrink it to reach that path on small frames)
    int sg_lw = -1, sg_lh = -1;          // CRUCIBLE_SG_TILE=WxH (powers of two, W*H <= 64); default 4x4 pixels x 4 samples
    // f32 trees with more than latency_entries wrappers run on pathtrace_kernel_latency (6 waves/SIMD) with a
    // latency_top_bytes LDS window, three 512-thread groups per CU.  CRUCIBLE_LATENCY_ENTRIES (0 = never).
    int32_t latency_entries = 0;         // (round 3: never by default -- with relaxed sums, the early touch of the leaf's second primitive and the deferred
                                         //  leaf phases the regular kernel is 6.7 % faster on the 1M-sphere tree: 1770 against 1658 Msamples/s)
    size_t latency_top_bytes = 48 * 1024;
    size_t lds_side_limit = 16 * 1024;   // RES_TOP: materials + textures join the LDS window up to this size (CRUCIBLE_LDS_SIDE_KB; 0 = never)
    size_t lds_top_bytes = 128 * 1024;  // LDS spent on the top of a tree that does not fit whole (CRUCIBLE_LDS_TOP_KB; 0 = none): 4096 32-byte records
                                        // (f32 wrappers, or the f64 kernels' screening records) -- one 1024-thread workgroup per CU has the LDS to itself; teapot +2.5 %
    bool lds_top_set = false;           // CRUCIBLE_LDS_TOP_KB given
    int blocks_per_cu_override = 0;
    int block_override = 0;
    // Wave scheduling of the walk (speed only).  -1 = chosen per scene: sphere scenes 10 / 56, scenes with triangles 8 / 40 --
    // a triangle test is ~1.5x a sphere test, so parked lanes are dearer and the sweeps (gpurun_out/exp7.txt, exp8.txt:
    // teapot +9 % in f64 and f32 at 8 / 40; book1 and the 1M-sphere scene lose 1-2 % there) favour shorter rounds and
    // an earlier exit.  CRUCIBLE_WALK_ROUND / CRUCIBLE_WALK_EXIT override.
    int walk_round_steps = -1;         // wrappers a lane may step through per round; 0 = until every walking lane found a leaf or ran out
    int walk_exit_lanes = -1;          // leave the walk phase once this many lanes are not walking (64 = wait for all)
    int walk_leaf_min = -1;            // CRUCIBLE_WALK_LEAF_MIN: parked lanes a leaf phase waits for while others can still step (0 = every round; default 8:
                                       // book1 +1.5 %, movie frame +1.5 %, 1M spheres +2.9 %, profiles/experiments/r03_leaf_min.txt)
    int last_block = 0, last_grid = 0;
    bool check_abort = false;          // the last launch was a queue kernel whose abort word has not been read yet
};

namespace {

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            (h)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                   \
            return CR_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

int32_t fail(CrHandle* h, int32_t code, const std::string& msg) {
    if (h) h->error = msg; else g_create_error = msg;
    return code;
}

template <typename real> DevScene<real>& dev_scene(CrHandle* h);
template <> DevScene<float>& dev_scene<float>(CrHandle* h) { return h->s32; }
template <> DevScene<double>& dev_scene<double>(CrHandle* h) { return h->s64; }

// ---------------------------------------------------------------- BVH build
// BVHWrapper::help_generate (src/objects/bvhwrapper.rs:46-78) emitted as a threaded
// pre-order array.  Node box = union of the range's construction-time primitive boxes
// (:47-50); axis = longest_axis with strict '>' (bvh.rs:82-94); span 1 and 2 become
// leaves without sorting (:58-63); span >= 3: stable sort by box min on the axis
// (sort_by is stable, :66-67), mid = start + span/2 (:71).
template <typename real> struct Builder {
    std::vector<real> bmin[3], bmax[3];
    std::vector<int32_t> order;
    std::vector<Entry<real>> entries;

    // Number of wrappers of a range of `span` primitives: a pure function of the span (median split),
    // so every subtree's position in the pre-order array is known before it is built and subtrees can
    // be built by independent threads.
    static int32_t tree_size(int32_t span) {
        if (span <= 2) return span > 0 ? 1 : 0;
        return 1 + tree_size(span / 2) + tree_size(span - span / 2);   // depth log2(n), two distinct spans per level
    }

    void build_root(int32_t n) {
        sizes.clear();
        entries.assign((size_t)size_of(n), Entry<real>());
        build(0, n, 0, 0);
    }

  private:
    std::map<int32_t, int32_t> sizes;
    int32_t size_of(int32_t span) {
        if (span <= 2) return span > 0 ? 1 : 0;
        auto it = sizes.find(span);
        if (it != sizes.end()) return it->second;
        int32_t v = 1 + size_of(span / 2) + size_of(span - span / 2);
        sizes[span] = v;
        return v;
    }

    void build(int32_t start, int32_t end, int32_t idx, int depth) {
        real lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = r_inf(real(0)); hi[a] = -r_inf(real(0)); }
        for (int32_t i = start; i < end; i++) {
            int32_t p = order[i];
            for (int a = 0; a < 3; a++) {   // Interval::tight_enclose, utils.rs:629-633
                lo[a] = lo[a] <= bmin[a][p] ? lo[a] : bmin[a][p];
                hi[a] = hi[a] >= bmax[a][p] ? hi[a] : bmax[a][p];
            }
        }
        real sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
        int axis = (sx > sy) ? ((sx > sz) ? 0 : 2) : ((sy > sz) ? 1 : 2);
        int32_t span = end - start;
        Entry<real> e;
        e.b[0] = lo[0]; e.b[1] = hi[0]; e.b[2] = lo[1]; e.b[3] = hi[1]; e.b[4] = lo[2]; e.b[5] = hi[2];
        e.skip = idx + 1; e.leaf = -1;
        if (span <= 2) { e.leaf = (start << 1) | (span - 1); entries[idx] = e; return; }
        const std::vector<real>& key = bmin[axis];
        std::stable_sort(order.begin() + start, order.begin() + end, [&](int32_t a, int32_t b) { return key[a] < key[b]; });
        int32_t mid = start + span / 2;
        const int32_t left_idx = idx + 1, right_idx = idx + 1 + sizes_at(span / 2);
        e.skip = idx + sizes_at(span);
        entries[idx] = e;
        if (depth < 4 && span >= (1 << 15)) {   // the two halves touch disjoint ranges of `order` and `entries`
            std::thread t([&] { build(start, mid, left_idx, depth + 1); });
            build(mid, end, right_idx, depth + 1);
            t.join();
        } else {
            build(start, mid, left_idx, depth + 1);
            build(mid, end, right_idx, depth + 1);
        }
    }
    int32_t sizes_at(int32_t span) const {   // read-only after build_root filled the table (thread-safe)
        if (span <= 2) return span > 0 ? 1 : 0;
        return sizes.at(span);
    }

};

// DFS pre-order -> level order with explicit links.  In pre-order the left child of inner entry i is
// i + 1 and `skip` already names the next wrapper after the subtree; storing the tree level by level
// (stable in DFS order within a level) puts the top of the tree first, which is what a partial LDS
// copy wants.  The walk order is unchanged: it follows the links, not the storage order.
template <typename real>
void relayout_bfs(std::vector<Entry<real>>& entries, std::vector<int32_t>& level_begin, std::vector<int8_t>* axis = nullptr) {
    const int32_t n = (int32_t)entries.size();
    level_begin.assign(1, 0);
    if (n == 0) return;
    std::vector<int32_t> level(n, 0), order_idx(n), new_of(n + 1);
    std::vector<int32_t> stack_end;   // ends (skip) of the enclosing inner wrappers
    for (int32_t i = 0; i < n; i++) {
        while (!stack_end.empty() && stack_end.back() <= i) stack_end.pop_back();
        level[i] = (int32_t)stack_end.size();
        if (entries[i].leaf < 0) stack_end.push_back(entries[i].skip);
    }
    for (int32_t i = 0; i < n; i++) order_idx[i] = i;
    std::stable_sort(order_idx.begin(), order_idx.end(), [&](int32_t a, int32_t b) { return level[a] < level[b]; });
    for (int32_t k = 0; k < n; k++) new_of[order_idx[k]] = k;
    new_of[n] = n;
    for (int32_t k = 1; k < n; k++) if (level[order_idx[k]] != level[order_idx[k - 1]]) level_begin.push_back(k);
    level_begin.push_back(n);
    std::vector<Entry<real>> out(n);
    for (int32_t k = 0; k < n; k++) {
        const int32_t i = order_idx[k];
        Entry<real> e = entries[i];
        e.skip = new_of[e.skip];
        if (e.leaf < 0) e.leaf = -new_of[i + 1];   // left child
        out[k] = e;
    }
    entries.swap(out);
    if (axis && !axis->empty()) {
        std::vector<int8_t> ax(n);
        for (int32_t k = 0; k < n; k++) ax[k] = (*axis)[order_idx[k]];
        axis->swap(ax);
    }
}

// SURVEY 8(f) row 1 -- CR_BVH_SAH: a binned surface-area-heuristic builder (16 bins per axis on the
// primitive-box centroids, all three axes tried, cost = area_L * n_L + area_R * n_R) instead of the
// reference's median split.  It emits the same wrapper array (boxes = union of the range's primitive boxes,
// leaves of one or two primitives, walked left then right with the shrinking interval), so the kernels and
// BVHWrapper::hit's semantics are unchanged; only the topology differs.  Decisions are made in f64 from the
// `real` boxes and are deterministic (stable partition, fixed tie-breaks), so cr_export_bvh reproduces the
// tree for a checker.
template <typename real> struct SahBuilder {
    const std::vector<real>* bmin;   // [3]
    const std::vector<real>* bmax;   // [3]
    std::vector<int32_t>* order;
    struct Node { real b[6]; int32_t left, right, start, end, axis; };
    std::vector<Node> nodes;
    std::atomic<int32_t> next{0};
    static constexpr int kBins = 16;

    static double area(const double lo[3], const double hi[3]) {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }

    void build_root(int32_t n) {
        nodes.assign((size_t)std::max(1, 2 * n), Node());
        next = 1;
        build(0, 0, n, 0);
    }

    void build(int32_t ni, int32_t start, int32_t end, int depth) {
        std::vector<int32_t>& ord = *order;
        Node nd;
        nd.left = nd.right = -1; nd.start = start; nd.end = end; nd.axis = 0;
        real lo[3], hi[3];
        double clo[3], chi[3];
        for (int a = 0; a < 3; a++) { lo[a] = r_inf(real(0)); hi[a] = -r_inf(real(0)); clo[a] = INFINITY; chi[a] = -INFINITY; }
        for (int32_t i = start; i < end; i++) {
            const int32_t p = ord[i];
            for (int a = 0; a < 3; a++) {
                lo[a] = lo[a] <= bmin[a][p] ? lo[a] : bmin[a][p];
                hi[a] = hi[a] >= bmax[a][p] ? hi[a] : bmax[a][p];
                const double cen = 0.5 * ((double)bmin[a][p] + (double)bmax[a][p]);
                clo[a] = std::min(clo[a], cen); chi[a] = std::max(chi[a], cen);
            }
        }
        nd.b[0] = lo[0]; nd.b[1] = hi[0]; nd.b[2] = lo[1]; nd.b[3] = hi[1]; nd.b[4] = lo[2]; nd.b[5] = hi[2];
        const int32_t span = end - start;
        if (span <= 2) { nodes[ni] = nd; return; }

        int best_axis = -1, best_plane = -1;
        double best_cost = INFINITY;
        for (int a = 0; a < 3; a++) {
            const double ext = chi[a] - clo[a];
            if (!(ext > 0.0) || !std::isfinite(ext)) continue;
            const double scale = (double)kBins / ext;
            int32_t cnt[kBins] = {0};
            double blo[kBins][3], bhi[kBins][3];
            for (int k = 0; k < kBins; k++) for (int d = 0; d < 3; d++) { blo[k][d] = INFINITY; bhi[k][d] = -INFINITY; }
            for (int32_t i = start; i < end; i++) {
                const int32_t p = ord[i];
                const double cen = 0.5 * ((double)bmin[a][p] + (double)bmax[a][p]);
                int k = (int)((cen - clo[a]) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                cnt[k]++;
                for (int d = 0; d < 3; d++) { blo[k][d] = std::min(blo[k][d], (double)bmin[d][p]); bhi[k][d] = std::max(bhi[k][d], (double)bmax[d][p]); }
            }
            double r_area[kBins];
            int32_t r_cnt[kBins];
            {   // suffix sweep: everything in bins k..end
                double l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
                int32_t c = 0;
                for (int k = kBins - 1; k >= 1; k--) {
                    if (cnt[k]) for (int d = 0; d < 3; d++) { l3[d] = std::min(l3[d], blo[k][d]); h3[d] = std::max(h3[d], bhi[k][d]); }
                    c += cnt[k];
                    r_cnt[k] = c; r_area[k] = c ? area(l3, h3) : 0.0;
                }
            }
            double l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
            int32_t c = 0;
            for (int k = 0; k + 1 < kBins; k++) {   // plane k: bins 0..k | k+1..end
                if (cnt[k]) for (int d = 0; d < 3; d++) { l3[d] = std::min(l3[d], blo[k][d]); h3[d] = std::max(h3[d], bhi[k][d]); }
                c += cnt[k];
                if (c == 0 || r_cnt[k + 1] == 0) continue;
                const double cost = area(l3, h3) * (double)c + r_area[k + 1] * (double)r_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = k; }
            }
        }
        int32_t mid;
        if (best_axis < 0) mid = start + span / 2;   // coincident centroids (or non-finite extents): split the list
        else {
            const int a = best_axis;
            const double scale = (double)kBins / (chi[a] - clo[a]);
            auto it = std::stable_partition(ord.begin() + start, ord.begin() + end, [&](int32_t p) {
                const double cen = 0.5 * ((double)bmin[a][p] + (double)bmax[a][p]);
                int k = (int)((cen - clo[a]) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= best_plane;
            });
            mid = (int32_t)(it - ord.begin());
        }
        nd.left = next.fetch_add(2);
        nd.right = nd.left + 1;
        nd.axis = best_axis < 0 ? 0 : best_axis;   // the left child holds the lower centroids along this axis
        nodes[ni] = nd;
        if (depth < 4 && span >= (1 << 15)) {   // the halves touch disjoint ranges of `order` and distinct nodes
            std::thread t([&] { build(nd.left, start, mid, depth + 1); });
            build(nd.right, mid, end, depth + 1);
            t.join();
        } else {
            build(nd.left, start, mid, depth + 1);
            build(nd.right, mid, end, depth + 1);
        }
    }

    // Node graph -> pre-order wrapper array with skip links (the layout Builder emits).
    void linearise(std::vector<Entry<real>>& out, std::vector<int8_t>& axis) const {
        out.clear(); axis.clear();
        std::vector<int32_t> stack{0}, open;   // open: pre-order indices of inner wrappers awaiting their end
        std::vector<std::pair<int32_t, int32_t>> todo;   // (node, pre-order index of the parent) -- iterative DFS
        struct Frame { int32_t node; int32_t state; int32_t idx; };
        std::vector<Frame> fr{{0, 0, -1}};
        while (!fr.empty()) {
            Frame& f = fr.back();
            const Node& nd = nodes[f.node];
            if (f.state == 0) {
                f.idx = (int32_t)out.size();
                Entry<real> e;
                for (int k = 0; k < 6; k++) e.b[k] = nd.b[k];
                e.skip = f.idx + 1; e.leaf = -1;
                if (nd.left < 0) { e.leaf = (nd.start << 1) | (nd.end - nd.start - 1); out.push_back(e); axis.push_back(-1); fr.pop_back(); continue; }
                out.push_back(e); axis.push_back((int8_t)nd.axis);
                f.state = 1;
                fr.push_back({nd.left, 0, -1});
            } else if (f.state == 1) {
                f.state = 2;
                fr.push_back({nd.right, 0, -1});
            } else {
                out[f.idx].skip = (int32_t)out.size();
                fr.pop_back();
            }
        }
    }
};

// Boxes of every wrapper of `entries` (a device copy of the tree), bottom-up by level: for the ray times [ta, tb]
// of a frame (use_keys) or the construction-time boxes (!use_keys).
template <typename real>
int32_t run_box_kernels(CrHandle* h, DevScene<real>& ds, void* entries, real ta, real tb, bool use_keys) {
    for (size_t l = ds.level_begin.size() - 1; l-- > 0;) {
        const int32_t begin = ds.level_begin[l], end = ds.level_begin[l + 1];
        if (end <= begin) continue;
        const dim3 grid((unsigned)((end - begin + 255) / 256)), block(256);
        if (ds.ordered) hipLaunchKernelGGL((refit_level_kernel<real, true>), grid, block, 0, h->stream, (EntryO<real>*)entries, begin, end,
                                           (const Prim<real>*)ds.prims.p, (const Key<real>*)ds.keys.p, ta, tb, use_keys ? 1 : 0, (const int32_t*)ds.leaf_runs.p);
        else hipLaunchKernelGGL((refit_level_kernel<real, false>), grid, block, 0, h->stream, (Entry<real>*)entries, begin, end,
                                (const Prim<real>*)ds.prims.p, (const Key<real>*)ds.keys.p, ta, tb, use_keys ? 1 : 0, (const int32_t*)ds.leaf_runs.p);
    }
    HIP_TRY(h, hipGetLastError());
    return CR_OK;
}

// CR_BVH_LBVH (lbvh.hpp): keys, sort and topology on the device; the node graph is then numbered into the
// level-order wrapper array (one primitive per leaf wrapper -- pairing sibling leaves measured slower: both
// primitives get tested on every visit; boxes are filled in later by run_box_kernels).  `order` receives the primitives' sorted order.
template <typename real>
int32_t build_lbvh(CrHandle* h, const std::vector<Prim<real>>& src, const std::vector<real>* bmin, const std::vector<real>* bmax,
                   std::vector<int32_t>& order, std::vector<Entry<real>>& entries, std::vector<int32_t>& level_begin) {
    const int32_t n = (int32_t)src.size();
    entries.clear();
    level_begin.assign(1, 0);
    if (n == 0) return CR_OK;
    LbvhBounds bnd;
    for (int a = 0; a < 3; a++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int32_t i = 0; i < n; i++) {
            const double cen = 0.5 * ((double)bmin[a][i] + (double)bmax[a][i]);
            lo = std::min(lo, cen); hi = std::max(hi, cen);
        }
        bnd.lo[a] = std::isfinite(lo) ? lo : 0.0;
        bnd.inv_ext[a] = (std::isfinite(hi - lo) && hi > lo) ? 1.0 / (hi - lo) : 0.0;
    }
    DevBuf d_src, d_keys, d_keys2, d_idx, d_idx2, d_tmp, d_children;
    auto cleanup = [&] { d_src.release(); d_keys.release(); d_keys2.release(); d_idx.release(); d_idx2.release(); d_tmp.release(); d_children.release(); };
#define LBVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(h, CR_ERR_HIP, hipGetErrorString(e_)); } } while (0)
    LBVH_TRY(d_src.ensure((size_t)n * sizeof(Prim<real>)));
    LBVH_TRY(hipMemcpyAsync(d_src.p, src.data(), (size_t)n * sizeof(Prim<real>), hipMemcpyHostToDevice, h->stream));
    LBVH_TRY(d_keys.ensure((size_t)n * 8)); LBVH_TRY(d_keys2.ensure((size_t)n * 8));
    LBVH_TRY(d_idx.ensure((size_t)n * 4)); LBVH_TRY(d_idx2.ensure((size_t)n * 4));
    LBVH_TRY(d_children.ensure((size_t)std::max(1, n - 1) * 8));
    const dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL((lbvh_key_kernel<real>), grid, block, 0, h->stream, (const Prim<real>*)d_src.p, n, bnd, (uint64_t*)d_keys.p, (int32_t*)d_idx.p);
    LBVH_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int32_t*)d_idx.p,
                                                (int32_t*)d_idx2.p, n, 0, 63, h->stream));
    LBVH_TRY(d_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const int32_t*)d_idx.p,
                                                (int32_t*)d_idx2.p, n, 0, 63, h->stream));
    if (n >= 2) {
        hipLaunchKernelGGL(lbvh_topology_kernel, dim3((unsigned)((n - 1 + 255) / 256)), block, 0, h->stream, (const uint64_t*)d_keys2.p, n, (int32_t*)d_children.p);
        LBVH_TRY(hipGetLastError());
    }
    order.resize(n);
    std::vector<int32_t> children((size_t)2 * std::max(1, n - 1));
    LBVH_TRY(hipMemcpyAsync(order.data(), d_idx2.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (n >= 2) LBVH_TRY(hipMemcpyAsync(children.data(), d_children.p, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, h->stream));
    LBVH_TRY(hipStreamSynchronize(h->stream));
#undef LBVH_TRY
    cleanup();
    // node graph -> level-order wrappers with links (what relayout_bfs would produce from a pre-order array), in one
    // breadth-first pass: a child < 0 is ~(sorted position of a primitive); siblings get adjacent indices
    const int32_t total = 2 * n - 1;
    entries.assign((size_t)total, Entry<real>());
    level_begin.assign(1, 0);
    if (n == 1) { entries[0].leaf = 0; entries[0].skip = 1; level_begin.push_back(1); return CR_OK; }
    std::vector<int32_t> ref((size_t)total);   // node reference (as in `children`) of each new index
    ref[0] = 0;
    entries[0].skip = total;
    int32_t level_first = 0, level_end = 1, next = 1;
    std::vector<char> seen((size_t)(n - 1), 0);
    while (level_first < level_end) {
        for (int32_t k = level_first; k < level_end; k++) {
            const int32_t r = ref[k];
            Entry<real>& e = entries[k];
            if (r < 0) { e.leaf = (~r) << 1; continue; }              // one primitive
            if (r >= n - 1 || seen[r] || next + 2 > total) return fail(h, CR_ERR_HIP, "LBVH: malformed topology from the device");
            seen[r] = 1;
            const int32_t cl = children[2 * r], cr = children[2 * r + 1];
            if ((cl < 0 && ~cl >= n) || (cr < 0 && ~cr >= n)) return fail(h, CR_ERR_HIP, "LBVH: malformed topology from the device");
            e.leaf = -next;
            ref[next] = cl; ref[next + 1] = cr;
            entries[next].skip = next + 1;                            // after the left subtree comes the right child
            entries[next + 1].skip = e.skip;                          // after the right subtree: whatever follows the parent
            next += 2;
        }
        level_first = level_end; level_end = next;
        level_begin.push_back(level_first);
    }
    entries.resize((size_t)next);
    for (Entry<real>& e : entries) if (e.skip == total) e.skip = next;   // "no wrapper follows" = the final count
    if (level_begin.back() != next) level_begin.push_back(next);
    return CR_OK;
}

// The f32 screening records of a (possibly refitted) f64 wrapper array, in the layout of the tree (ScreenEntry / ScreenEntryO).
int32_t make_screen(CrHandle* h, DevScene<double>& ds, const void* entries, DevBuf& out) {
    const size_t rec = ds.ordered ? sizeof(ScreenEntryO) : sizeof(ScreenEntry);
    HIP_TRY(h, out.ensure((size_t)ds.n_entries * rec, ds.ordered ? entry_pad<ScreenEntryO>() : entry_pad<ScreenEntry>()));
    const dim3 grid((unsigned)((ds.n_entries + 255) / 256));
    if (ds.ordered) hipLaunchKernelGGL(screen_from_ordered_entries_kernel, grid, dim3(256), 0, h->stream, (const EntryO<double>*)entries, (ScreenEntryO*)out.p, ds.n_entries);
    else hipLaunchKernelGGL(screen_from_entries_kernel<double>, grid, dim3(256), 0, h->stream, (const Entry<double>*)entries, (ScreenEntry*)out.p, ds.n_entries);
    HIP_TRY(h, hipGetLastError());
    return CR_OK;
}
// f32 scenes: the same boxes in ScreenEntry's link layout (the walk's inner loop reads that one), unordered trees only
int32_t make_screen(CrHandle* h, DevScene<float>& ds, const void* entries, DevBuf& out) {
    if (ds.ordered) { out.release(); return CR_OK; }
    HIP_TRY(h, out.ensure((size_t)ds.n_entries * sizeof(ScreenEntry), entry_pad<ScreenEntry>()));
    hipLaunchKernelGGL(screen_from_entries_kernel<float>, dim3((unsigned)((ds.n_entries + 255) / 256)), dim3(256), 0, h->stream, (const Entry<float>*)entries, (ScreenEntry*)out.p, ds.n_entries);
    HIP_TRY(h, hipGetLastError());
    return CR_OK;
}

template <typename real> int32_t build_dev_scene(CrHandle* h) {
    DevScene<real>& ds = dev_scene<real>(h);
    if (ds.built) return CR_OK;
    auto t_begin = std::chrono::steady_clock::now();
    const bool timing = getenv("CRUCIBLE_BUILD_TIMING") != nullptr;
    auto lap = [&](const char* what) { if (timing) fprintf(stderr, "[build] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    // The objects the BVH build sees, in list order (bvhwrapper.rs:16-26): visible spheres and triangles, and every
    // list whatever it holds.  Under the opt-in trees a list's visible objects stand in for it.
    struct Obj { int32_t desc, first, count, inner; };   // count < 0: a primitive; inner >= 0: a BVHWrapper element (index into `inners`)
    const bool ref_tree = h->bvh_mode == CR_BVH_REFERENCE;
    std::vector<Obj> objs;
    std::vector<std::vector<int32_t>> inner_members;   // per BVHWrapper element: its visible objects (descriptor indices)
    for (size_t i = 0; i < h->prims.size(); i++) {
        const CrPrimitive& p = h->prims[i];
        if (p.flags & CR_PRIM_MEMBER) continue;
        if (p.kind == CR_PRIM_LIST || p.kind == CR_PRIM_BVH) {
            const int32_t first = (int32_t)p.v[0], count = (int32_t)p.v[1];
            if (!ref_tree) { for (int32_t k = first; k < first + count; k++) if (!(h->prims[k].flags & CR_PRIM_HIDDEN)) objs.push_back({k, 0, -1, -1}); continue; }
            if (p.kind == CR_PRIM_LIST) { objs.push_back({(int32_t)i, first, count, -1}); continue; }
            std::vector<int32_t> vis;   // new_wrapper drops the hidden objects (bvhwrapper.rs:16-26)
            for (int32_t k = first; k < first + count; k++) if (!(h->prims[k].flags & CR_PRIM_HIDDEN)) vis.push_back(k);
            if (vis.empty()) { objs.push_back({(int32_t)i, first, 0, -1}); continue; }   // ... and returns an empty list for none (:28-30)
            objs.push_back({(int32_t)i, first, count, (int32_t)inner_members.size()});
            inner_members.push_back(std::move(vis));
        } else if (!(p.flags & CR_PRIM_HIDDEN)) objs.push_back({(int32_t)i, 0, -1, -1});
    }
    const int32_t n = (int32_t)objs.size();
    Builder<real> b;
    for (int a = 0; a < 3; a++) { b.bmin[a].resize(n); b.bmax[a].resize(n); }
    b.order.resize(n);
    std::vector<Prim<real>> src(n);
    bool any_keys = false, any_lists = false;
    ds.has_triangles = false; ds.has_spheres = false;
    auto make_prim = [&](const CrPrimitive& p) {
        Prim<real> q;
        memset(&q, 0, sizeof q);
        for (int k = 0; k < 9; k++) q.g[k] = (real)p.v[k];
        if (p.kind == CR_PRIM_SPHERE) q.g[4] = real(1) / q.g[3];    // 1/radius, used for the hit normal of static spheres
        q.kind_mat = (p.kind & 1) | (p.material << 1);
        q.key_first = p.key_first; q.key_count = p.key_count;
        any_keys |= p.key_count > 0;
        ds.has_triangles |= p.kind == CR_PRIM_TRIANGLE;
        ds.has_spheres |= p.kind == CR_PRIM_SPHERE;
        return q;
    };
    auto prim_box = [](const Prim<real>& q, real lo[3], real hi[3]) {
        if (q.kind() == CR_PRIM_SPHERE) {   // Sphere::new, sphere.rs:29-30; Aabb::new_from_points bvh.rs:44-64
            const real r = q.g[3];
            for (int a = 0; a < 3; a++) {
                const real l = q.g[a] + (-r), u = q.g[a] + r;
                if (l <= u) { lo[a] = l; hi[a] = u; } else { lo[a] = u; hi[a] = l; }
            }
        } else {                            // Triangle::new, triangle.rs:28-35 (f64::min/max)
            for (int a = 0; a < 3; a++) {
                hi[a] = std::fmax(q.g[a], std::fmax(q.g[3 + a], q.g[6 + a]));
                lo[a] = std::fmin(q.g[a], std::fmin(q.g[3 + a], q.g[6 + a]));
            }
        }
    };
    // BVHWrapper elements: the inner trees, by the reference's own build over their visible objects
    std::vector<Builder<real>> inners(inner_members.size());
    std::vector<std::vector<Prim<real>>> inner_src(inner_members.size());
    for (size_t w = 0; w < inners.size(); w++) {
        const std::vector<int32_t>& mem = inner_members[w];
        const int32_t m = (int32_t)mem.size();
        Builder<real>& ib = inners[w];
        for (int a = 0; a < 3; a++) { ib.bmin[a].resize(m); ib.bmax[a].resize(m); }
        ib.order.resize(m);
        inner_src[w].resize(m);
        for (int32_t k = 0; k < m; k++) {
            inner_src[w][k] = make_prim(h->prims[mem[k]]);
            real lo[3], hi[3];
            prim_box(inner_src[w][k], lo, hi);
            for (int a = 0; a < 3; a++) { ib.bmin[a][k] = lo[a]; ib.bmax[a][k] = hi[a]; }
            ib.order[k] = k;
        }
        ib.build_root(m);
    }
    for (int32_t i = 0; i < n; i++) {
        const Obj& o = objs[i];
        b.order[i] = i;
        real lo[3], hi[3];
        if (o.count < 0) {
            src[i] = make_prim(h->prims[o.desc]);
            prim_box(src[i], lo, hi);
        } else if (o.inner >= 0) {   // the wrapper's box: its root's (new_from_vec, bvhwrapper.rs:39)
            const Entry<real>& root = inners[o.inner].entries[0];
            for (int a = 0; a < 3; a++) { lo[a] = root.b[2 * a]; hi[a] = root.b[2 * a + 1]; }
        } else {   // HitList: Aabb::default() (hitlist.rs:13-18), grown by add() over every object, hidden or not (hitlist.rs:24-27)
            any_lists = true;
            for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<real>::infinity(); hi[a] = -std::numeric_limits<real>::infinity(); }
            if (!(h->prims[o.desc].flags & CR_LIST_EMPTY_BOX))
                for (int32_t k = o.first; k < o.first + o.count; k++) {
                    CrPrimitive m = h->prims[k];
                    Prim<real> q;
                    for (int j = 0; j < 9; j++) q.g[j] = (real)m.v[j];
                    q.kind_mat = m.kind & 1;
                    real ml[3], mh[3];
                    prim_box(q, ml, mh);
                    for (int a = 0; a < 3; a++) {   // Interval::tight_enclose, utils.rs:631-635
                        lo[a] = lo[a] <= ml[a] ? lo[a] : ml[a];
                        hi[a] = hi[a] >= mh[a] ? hi[a] : mh[a];
                    }
                }
        }
        for (int a = 0; a < 3; a++) { b.bmin[a][i] = lo[a]; b.bmax[a][i] = hi[a]; }
    }
    std::vector<int8_t> axis;
    struct Run { int32_t first, count; bool pseudo; };
    std::vector<Run> spliced_runs;              // scenes with a BVHWrapper element: the primitive run of every leaf record
    std::vector<Prim<real>> spliced_prims;      // ... and the primitive records in the order the runs name them
    bool spliced = false;
    ds.ordered = h->bvh_mode == CR_BVH_SAH_ORDERED;
    lap("primitive records and boxes");
    const bool lbvh = h->bvh_mode == CR_BVH_LBVH;
    if (lbvh) {
        int32_t rc = build_lbvh<real>(h, src, b.bmin, b.bmax, b.order, b.entries, ds.level_begin);
        if (rc != CR_OK) return rc;
    } else if (n > 0 && h->bvh_mode != CR_BVH_REFERENCE) {
        SahBuilder<real> sb;
        sb.bmin = b.bmin; sb.bmax = b.bmax; sb.order = &b.order;
        sb.build_root(n);
        sb.linearise(b.entries, axis);
        relayout_bfs(b.entries, ds.level_begin, &axis);
    } else if (n > 0) {
        b.build_root(n);
        if (!inners.empty()) {
            // A leaf wrapper that holds a BVHWrapper element becomes an inner record: the element's own tree is spliced in
            // as one child; a primitive or list beside it becomes a record of its own with an empty box (which the box
            // test always passes, bvh.rs:96-130 -- BVHWrapper::hit tests that child without any box), and a span-1
            // wrapper (the element twice, bvhwrapper.rs:56-58) gets an empty record as its second child: the second walk
            // of the same tree cannot find anything closer.  Every leaf names its primitive run through `runs`.
            std::vector<Entry<real>> sp;
            std::function<void(int32_t)> emit;
            auto new_run = [&](int32_t first, int32_t count, bool pseudo) { spliced_runs.push_back({first, count, pseudo}); return (int32_t)spliced_runs.size() - 1; };
            auto append_obj = [&](const Obj& o, int32_t order_pos) {
                if (o.count < 0) spliced_prims.push_back(src[order_pos]);
                else for (int32_t k = o.first; k < o.first + o.count; k++) if (!(h->prims[k].flags & CR_PRIM_HIDDEN)) spliced_prims.push_back(make_prim(h->prims[k]));
            };
            const real inf = std::numeric_limits<real>::infinity();
            auto pseudo_leaf = [&](int32_t first, int32_t count) {
                Entry<real> pe;
                for (int a = 0; a < 3; a++) { pe.b[2 * a] = inf; pe.b[2 * a + 1] = -inf; }
                pe.leaf = new_run(first, count, true);
                pe.skip = (int32_t)sp.size() + 1;
                sp.push_back(pe);
            };
            emit = [&](int32_t i) {
                const Entry<real> e = b.entries[i];
                const int32_t idx = (int32_t)sp.size();
                sp.push_back(e);
                if (e.leaf < 0) { emit(i + 1); emit(b.entries[i + 1].skip); sp[idx].skip = (int32_t)sp.size(); return; }
                const int32_t start = e.leaf >> 1, span = (e.leaf & 1) + 1;
                bool any_inner = false;
                for (int32_t k = 0; k < span; k++) any_inner |= objs[b.order[start + k]].inner >= 0;
                if (!any_inner) {
                    const int32_t first = (int32_t)spliced_prims.size();
                    for (int32_t k = 0; k < span; k++) append_obj(objs[b.order[start + k]], b.order[start + k]);
                    sp[idx].leaf = new_run(first, (int32_t)spliced_prims.size() - first, false);
                    sp[idx].skip = idx + 1;
                    return;
                }
                sp[idx].leaf = -1;
                for (int32_t k = 0; k < span; k++) {
                    const Obj& o = objs[b.order[start + k]];
                    if (o.inner < 0) {
                        const int32_t first = (int32_t)spliced_prims.size();
                        append_obj(o, b.order[start + k]);
                        pseudo_leaf(first, (int32_t)spliced_prims.size() - first);
                        continue;
                    }
                    const Builder<real>& ib = inners[o.inner];
                    const int32_t base = (int32_t)sp.size();
                    for (const Entry<real>& ie : ib.entries) {
                        Entry<real> c = ie;
                        c.skip += base;
                        if (c.leaf >= 0) {
                            const int32_t s0 = c.leaf >> 1, cnt = (c.leaf & 1) + 1, first = (int32_t)spliced_prims.size();
                            for (int32_t q = 0; q < cnt; q++) spliced_prims.push_back(inner_src[o.inner][ib.order[s0 + q]]);
                            c.leaf = new_run(first, cnt, false);
                        }
                        sp.push_back(c);
                    }
                }
                if (span == 1) pseudo_leaf((int32_t)spliced_prims.size(), 0);
                sp[idx].skip = (int32_t)sp.size();
            };
            emit(0);
            b.entries.swap(sp);
            spliced = true;
        }
        relayout_bfs(b.entries, ds.level_begin);
    }
    else ds.level_begin.assign(1, 0);
    lap("tree");
    // Primitive records in leaf order; a list contributes its visible objects in the list's order (a hidden object
    // returns no hit before anything is computed: sphere.rs:62, triangle.rs:87).
    std::vector<Prim<real>> leaf_prims;
    leaf_prims.reserve(n);
    std::vector<int32_t> first_of((size_t)n + 1);
    if (spliced) leaf_prims.swap(spliced_prims);
    else for (int32_t i = 0; i < n; i++) {
        const Obj& o = objs[b.order[i]];
        first_of[i] = (int32_t)leaf_prims.size();
        if (o.count < 0) leaf_prims.push_back(src[b.order[i]]);
        else for (int32_t k = o.first; k < o.first + o.count; k++) if (!(h->prims[k].flags & CR_PRIM_HIDDEN)) leaf_prims.push_back(make_prim(h->prims[k]));
    }
    if (!spliced) first_of[n] = (int32_t)leaf_prims.size();
    if (leaf_prims.size() >= ((size_t)1 << 29)) return fail(h, CR_ERR_INVALID_ARG, "too many primitives");
    // What the device walks: a leaf wrapper names a run of primitive records.  One or two records fit the wrapper
    // itself; a leaf that holds a list names its run through the side table (first, count).
    std::vector<Entry<real>> dev_entries;
    std::vector<int32_t> leaf_runs;
    if (spliced) {
        dev_entries = b.entries;
        for (Entry<real>& e : dev_entries) {
            if (e.leaf < 0) continue;
            const Run r = spliced_runs[e.leaf];
            if (!r.pseudo && (r.count == 1 || r.count == 2)) e.leaf = (r.first << 1) | (r.count - 1);
            else { e.leaf = kLeafRun | (r.pseudo ? kLeafPseudo : 0) | (int32_t)(leaf_runs.size() / 2); leaf_runs.push_back(r.first); leaf_runs.push_back(r.count); }
        }
    } else if (any_lists) {
        dev_entries = b.entries;
        for (Entry<real>& e : dev_entries) {
            if (e.leaf < 0) continue;
            const int32_t start = e.leaf >> 1, span = (e.leaf & 1) + 1;
            const int32_t first = first_of[start], count = first_of[start + span] - first;
            if (count == 1 || count == 2) e.leaf = (first << 1) | (count - 1);
            else { e.leaf = kLeafRun | (int32_t)(leaf_runs.size() / 2); leaf_runs.push_back(first); leaf_runs.push_back(count); }
        }
    }
    const std::vector<Entry<real>>& up_entries = (any_lists || spliced) ? dev_entries : b.entries;

    // Device texture table: only textures a non-solid lambertian can reach (a solid top-level
    // texture is folded into its material), re-indexed densely; children keep smaller indices.
    std::vector<int32_t> tex_remap(h->textures.size(), -1);
    {
        std::vector<char> live(h->textures.size(), 0);
        for (const CrMaterial& m : h->materials)
            if (m.kind == CR_MAT_LAMBERTIAN && h->textures[m.texture].kind != CR_TEX_SOLID) live[m.texture] = 1;
        for (size_t i = h->textures.size(); i-- > 0;)   // parents have larger indices than children
            if (live[i] && h->textures[i].kind == CR_TEX_CHECKER) { live[h->textures[i].even] = 1; live[h->textures[i].odd] = 1; }
        int32_t next = 0;
        for (size_t i = 0; i < live.size(); i++) if (live[i]) tex_remap[i] = next++;
    }
    std::vector<Mat<real>> mats(h->materials.size());
    for (size_t i = 0; i < mats.size(); i++) {
        const CrMaterial& m = h->materials[i];
        Mat<real>& o = mats[i];
        memset(&o, 0, sizeof o);
        o.kind = m.kind; o.param = (real)m.param; o.tex = -1;
        for (int k = 0; k < 3; k++) o.albedo[k] = (real)m.albedo[k];
        if (m.kind == CR_MAT_LAMBERTIAN) {
            const CrTexture& t = h->textures[m.texture];
            if (t.kind == CR_TEX_SOLID) for (int k = 0; k < 3; k++) o.albedo[k] = (real)t.color[k];
            else o.tex = tex_remap[m.texture];
            o.aux = real(1) / r_abs(o.param);                       // Color / f64: (1.0 / rhs.abs()) * c
        } else if (m.kind == CR_MAT_DIELECTRIC) {
            auto r0 = [](real ri) { real q = (real(1) - ri) / (real(1) + ri); return q * q; };   // dielectric.rs:21-23
            o.albedo[0] = real(1) / o.param;                        // ri for a front-face hit (dielectric.rs:33-37)
            o.albedo[1] = r0(o.albedo[0]);
            o.albedo[2] = r0(o.param);
        }
    }
    std::vector<Tex<real>> texs;
    for (size_t i = 0; i < h->textures.size(); i++) {
        if (tex_remap[i] < 0) continue;
        const CrTexture& t = h->textures[i];
        Tex<real> o;
        memset(&o, 0, sizeof o);
        o.kind = t.kind; o.image = t.image; o.inv_scale = (real)t.inv_scale;
        o.even = t.kind == CR_TEX_CHECKER ? tex_remap[t.even] : -1;
        o.odd = t.kind == CR_TEX_CHECKER ? tex_remap[t.odd] : -1;
        for (int k = 0; k < 3; k++) o.color[k] = (real)t.color[k];
        texs.push_back(o);
    }
    std::vector<Key<real>> keys(h->keys.size());
    if (!keys.empty()) memset(keys.data(), 0, keys.size() * sizeof(Key<real>));
    for (size_t i = 0; i < h->keys.size(); i++) {
        const CrKeyframe& k = h->keys[i];
        keys[i].t0 = (real)k.t0; keys[i].t1 = (real)k.t1; keys[i].a = (real)k.a; keys[i].b = (real)k.b;
        keys[i].channel = k.channel; keys[i].interp = k.interp;
    }

    auto up = [&](DevBuf& d, const void* src_p, size_t bytes, size_t front_pad = 0) -> hipError_t {
        hipError_t e = d.ensure(bytes ? bytes : 16, front_pad);
        if (e != hipSuccess) return e;
        if (bytes) return hipMemcpy(d.p, src_p, bytes, hipMemcpyHostToDevice);
        return hipSuccess;
    };
    ds.entry_bytes = ds.ordered ? sizeof(EntryO<real>) : sizeof(Entry<real>);
    if (ds.ordered) {   // per-octant skip links, parents before children (level order)
        const int32_t ne = (int32_t)b.entries.size();
        std::vector<EntryO<real>> eo((size_t)ne);
        for (int32_t i = 0; i < ne; i++) {
            for (int k = 0; k < 6; k++) eo[i].b[k] = b.entries[i].b[k];
            eo[i].unused = 0;
            const int32_t leaf = b.entries[i].leaf;
            eo[i].leaf = leaf < 0 ? -((-leaf) * 4 + axis[i]) : leaf;
        }
        if (ne > 0) for (int o = 0; o < 8; o++) eo[0].skip[o] = ne;
        for (int32_t i = 0; i < ne; i++) {
            const int32_t leaf = b.entries[i].leaf;
            if (leaf >= 0) continue;
            const int32_t left = -leaf;
            for (int o = 0; o < 8; o++) {
                const int32_t nearc = left + ((o >> axis[i]) & 1), farc = left + 1 - ((o >> axis[i]) & 1);
                eo[nearc].skip[o] = farc;
                eo[farc].skip[o] = eo[i].skip[o];
            }
        }
        HIP_TRY(h, up(ds.entries, eo.data(), eo.size() * sizeof(EntryO<real>), entry_pad<EntryO<real>>()));
    } else
    HIP_TRY(h, up(ds.entries, up_entries.data(), up_entries.size() * sizeof(Entry<real>), entry_pad<Entry<real>>()));
    HIP_TRY(h, up(ds.leaf_runs, leaf_runs.data(), leaf_runs.size() * sizeof(int32_t)));
    ds.has_leaf_runs = !leaf_runs.empty();
    ds.has_lists = any_lists;
    HIP_TRY(h, up(ds.prims, leaf_prims.data(), leaf_prims.size() * sizeof(Prim<real>)));
    HIP_TRY(h, up(ds.mats, mats.data(), mats.size() * sizeof(Mat<real>)));
    HIP_TRY(h, up(ds.texs, texs.data(), texs.size() * sizeof(Tex<real>)));
    HIP_TRY(h, up(ds.keys, keys.data(), keys.size() * sizeof(Key<real>)));
    ds.n_entries = (int32_t)b.entries.size(); ds.n_prims = (int32_t)leaf_prims.size(); ds.n_mats = (int32_t)mats.size(); ds.n_texs = (int32_t)texs.size();
    ds.n_scene_keys = (int32_t)h->keys.size();
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    ds.lds_bytes = r16(b.entries.size() * ds.entry_bytes) + r16(leaf_prims.size() * sizeof(Prim<real>)) +
                   r16(mats.size() * sizeof(Mat<real>)) + r16(texs.size() * sizeof(Tex<real>));
    ds.animated = any_keys;
    if (lbvh && ds.n_entries > 0) {   // boxes: construction-time primitive boxes, bottom-up, on the device
        int32_t rc = run_box_kernels<real>(h, ds, ds.entries.p, real(0), real(0), false);
        if (rc != CR_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(b.entries.data(), ds.entries.p, b.entries.size() * sizeof(Entry<real>), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (ds.n_entries > 0) {   // the f32 screening records of an f64 scene / the link-layout records of an f32 scene (pathtrace.hpp walk_round)
        int32_t rc = make_screen(h, ds, ds.entries.p, ds.screen);
        if (rc != CR_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else ds.screen.release();
    lap("uploads and boxes");
    ds.host_entries = b.entries;
    ds.host_axis = axis;
    ds.leaf_desc.resize(n);
    for (int32_t i = 0; i < n; i++) ds.leaf_desc[i] = objs[b.order[i]].desc;
    ds.has_bvh_elements = spliced;
    ds.built = true;
    h->upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return CR_OK;
}

template <typename real> void key_to_real(const CrKeyframe& k, Key<real>& o) {
    o.t0 = (real)k.t0; o.t1 = (real)k.t1; o.a = (real)k.a; o.b = (real)k.b; o.channel = k.channel; o.interp = k.interp;
}

template <typename real, int RES, bool ANIM, bool ORD = false, bool LATENCY = false, bool CAMK = false, bool RELAX = false, bool SCREEN = false>
int32_t launch(CrHandle* h, const KernelArgs<real>& args_in, size_t scene_lds_bytes, CrStats* stats) {
    constexpr bool LDS = RES != RES_GLOBAL || RELAX;
    static_assert(!LATENCY || RES == RES_TOP, "the 6-waves-per-SIMD entry point exists for RES_TOP only");
    KernelArgs<real> args = args_in;
    void (*kern)(const KernelArgs<real>) = pathtrace_kernel<real, RES, ANIM, ORD, CAMK, RELAX, SCREEN>;
    if constexpr (LATENCY) kern = pathtrace_kernel_latency<real, ANIM, ORD, CAMK, RELAX>;
    const int max_block = LATENCY ? LatencyBlock : MaxBlock<real>::value;
    // The work tile (sample-granular hand-out): 2^lw x 2^lh pixels times 64 >> (lw + lh) consecutive samples; by default
    // 4 x 4 x 4, wider tiles of fewer samples when fewer than 4 samples are rendered.
    const int32_t n_samples = args.sample_end - args.sample_begin;
    int tile_lw = h->sg_lw, tile_lh = h->sg_lh;
    if (tile_lw < 0) { const int ns = n_samples >= 4 ? 4 : (n_samples >= 2 ? 2 : 1); tile_lw = ns == 4 ? 2 : 3; tile_lh = ns == 1 ? 3 : 2; }
    // RELAX: the waves' accumulator slots follow the scene in LDS (2 x 384 B per wave for a 16-pixel tile); a tile too
    // large for what the scene leaves free falls back to 16 pixels (the surplus sample slots of its groups stay empty)
    const size_t fx_off = RES != RES_GLOBAL ? ((scene_lds_bytes + 15) & ~(size_t)15) : 0;
    if (RELAX && fx_off + fx_lds_bytes(max_block, (uint32_t)(tile_lw + tile_lh)) > (size_t)160 * 1024) { tile_lw = 2; tile_lh = 2; }
    auto lds_for = [&](int block) { return RELAX ? fx_off + fx_lds_bytes(block, (uint32_t)(tile_lw + tile_lh)) : scene_lds_bytes; };
    args.fx_lds_off = (uint32_t)fx_off;
    if (LDS) HIP_TRY(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(max_block)));
    // Workgroup size: the candidate that keeps the most waves resident per CU (a larger
    // workgroup shares one LDS copy of the scene among more waves); ties go to the larger.
    int block = 256, per_cu = 1, best_waves = 0;
    for (int cand : {1024, 512, 256}) {
        if (cand > max_block) continue;
        if (h->block_override > 0 && cand != h->block_override && h->block_override <= max_block) continue;
        int n = 0;
        HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, cand, LDS ? lds_for(cand) : 0));
        if (n * cand / 64 > best_waves) { best_waves = n * cand / 64; block = cand; per_cu = n; }
    }
    const size_t lds_bytes = lds_for(block);
    if (best_waves == 0) return fail(h, CR_ERR_HIP, "kernel does not fit on a CU");
    if (h->blocks_per_cu_override > 0) per_cu = h->blocks_per_cu_override;
    // Sample-granular mode: batches of samples whose colours fit the buffer; each batch is one launch of the
    // path tracer followed by the ordered sum (sg_finalize_kernel).
    const size_t npix = (size_t)args.cam.W * (size_t)args.cam.H;
    const int32_t s_begin = args.sample_begin, s_end = args.sample_end;
    int32_t batch = 0;
    if ((h->sample_granular || RELAX) && s_end > s_begin) {
        const size_t per_sample = npix * 3 * sizeof(real);
        batch = (int32_t)std::min<size_t>((size_t)(s_end - s_begin), RELAX ? (size_t)INT32_MAX : std::max<size_t>(1, h->sample_buf_limit / per_sample));
        int lw = tile_lw, lh = tile_lh;
        if (!RELAX && h->sg_lw < 0) { const int ns = batch >= 4 ? 4 : (batch >= 2 ? 2 : 1); lw = ns == 4 ? 2 : 3; lh = ns == 1 ? 3 : 2; }   // by the batch, which the buffer may have cut
        const uint32_t ns = 64u >> (lw + lh);
        args.sg_lw = (uint32_t)lw; args.sg_lh = (uint32_t)lh;
        args.tiles_x = ((uint32_t)args.cam.W + (1u << lw) - 1) >> lw;
        args.tiles_y = ((uint32_t)args.cam.H + (1u << lh) - 1) >> lh;
        const uint64_t tiles = (uint64_t)args.tiles_x * args.tiles_y;
        // the 32-bit work counter must hold tiles * groups * 64 plus one chunk per wave
        const uint64_t max_groups = h->work_counter_max / (tiles * 64);
        if (max_groups < 1) batch = 0;
        else batch = (int32_t)std::min<uint64_t>((uint64_t)batch, max_groups * ns);
        // the buffer holds one colour per work item of a batch: whole tiles and whole sample groups (edge padding included)
        if constexpr (RELAX) {
            if (batch <= 0) return fail(h, CR_ERR_UNSUPPORTED, "image too large for the 32-bit work counter");
            HIP_TRY(h, h->fx_acc.ensure(npix * 3 * sizeof(unsigned long long)));
        } else {
            auto batch_bytes = [&](int32_t b) { return (size_t)tiles * ((size_t)(b + (int32_t)ns - 1) / ns) * 64u * 3u * sizeof(real); };
            while (batch > (int32_t)ns && batch_bytes(batch) > std::max(h->sample_buf_limit, batch_bytes((int32_t)ns))) batch -= (int32_t)ns;
            if (batch > 0 && h->sample_buf.ensure(batch_bytes(batch)) != hipSuccess) { (void)hipGetLastError(); batch = 0; }
            if (batch > 0 && batch < s_end - s_begin && h->sg_acc.ensure(per_sample) != hipSuccess) { (void)hipGetLastError(); batch = 0; }
            if (batch == 0) { args.tiles_x = args_in.tiles_x; args.tiles_y = args_in.tiles_y; }   // fall back: a lane owns a pixel
        }
    }
    args.sg_on = batch > 0 ? 1u : 0u;
    const uint32_t ns = args.sg_on ? (64u >> (args.sg_lw + args.sg_lh)) : 1u;
    auto groups_of = [&](int32_t n) { return (uint32_t)((n + (int32_t)ns - 1) / (int32_t)ns); };
    uint64_t total_work = args.sg_on ? (uint64_t)args.tiles_x * args.tiles_y * groups_of(std::min(batch, s_end - s_begin)) * 64u
                                     : (uint64_t)args.tiles_x * args.tiles_y * 64u;
    uint32_t grid = (uint32_t)(h->n_cus * per_cu);
    uint64_t need_blocks = (total_work + block - 1) / block;
    if ((uint64_t)grid > need_blocks) grid = (uint32_t)need_blocks;
    if (grid < 1) grid = 1;
    args.n_threads = grid * (uint32_t)block;
    if constexpr (!RELAX) {
        size_t stack_bytes = (size_t)3 * (size_t)(args.max_depth > 0 ? args.max_depth : 1) * args.n_threads * sizeof(real);
        HIP_TRY(h, h->att_stack.ensure(stack_bytes));
        args.att_stack = (real*)h->att_stack.p;
    } else {
        // n * 2^S < 2^63 for the n samples a pixel receives in this render: S = 52 up to 2047 samples
        int lg = 0;
        while (((int64_t)(s_end - s_begin) >> (lg + 1)) > 0) lg++;
        const int S = std::min(52, 62 - lg);
        args.fx_scale = std::ldexp(1.0, S);
        args.fx_acc = (unsigned long long*)h->fx_acc.p;
        HIP_TRY(h, hipMemsetAsync(h->fx_acc.p, 0, npix * 3 * sizeof(unsigned long long), h->stream));
    }
    HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, 64 * sizeof(uint64_t), h->stream));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (!args.sg_on) {
        HIP_TRY(h, hipMemsetAsync(h->work_counter.p, 0, 4, h->stream));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), LDS ? lds_bytes : 0, h->stream, args);
        HIP_TRY(h, hipGetLastError());
    } else {
        args.sample_buf = (real*)h->sample_buf.p;
        for (int32_t b0 = s_begin; b0 < s_end; b0 += batch) {
            const int32_t b1 = std::min(s_end, b0 + batch);
            args.sample_begin = b0; args.sample_end = b1;
            args.sg_groups = groups_of(b1 - b0);
            args.sg_total = (uint32_t)((uint64_t)args.tiles_x * args.tiles_y * args.sg_groups * 64u);
            {   // 1024 items per atomic keeps the counter quiet on long launches; a short launch (a small frame, or one
                // GPU's shard of the samples) would end with whole chunks of imbalance, so a wave's chunk is at most
                // 1/128 of its share
                const uint64_t per_wave = (uint64_t)args.sg_total / std::max<uint64_t>(1, (uint64_t)grid * block / 64);
                const uint64_t c = h->sg_chunk_override > 0 ? (uint64_t)h->sg_chunk_override : std::min<uint64_t>(1024, std::max<uint64_t>(64, per_wave / 128));
                args.sg_chunk = (uint32_t)((c + 63) / 64 * 64);
            }
            HIP_TRY(h, hipMemsetAsync(h->work_counter.p, 0, 4, h->stream));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), LDS ? lds_bytes : 0, h->stream, args);
            HIP_TRY(h, hipGetLastError());
            if constexpr (RELAX) continue;   // the sums stay in fx_acc until the last batch
            const size_t fin_threads = ((size_t)args.tiles_x * args.tiles_y) << (args.sg_lw + args.sg_lh);
            hipLaunchKernelGGL((sg_finalize_kernel<real>), dim3((unsigned)((fin_threads + 255) / 256)), dim3(256), 0, h->stream, args,
                               (real*)h->sg_acc.p, b1 - b0, b0 == s_begin ? 1 : 0, b1 == s_end ? 1 : 0);
            HIP_TRY(h, hipGetLastError());
        }
        if constexpr (RELAX) {
            const size_t n = npix * 3;
            hipLaunchKernelGGL((fx_finalize_kernel<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                               (const unsigned long long*)h->fx_acc.p, args.out, n, 1.0 / args.fx_scale, (double)args.samples_total, args.output_sum);
            HIP_TRY(h, hipGetLastError());
        }
        args.sample_begin = s_begin; args.sample_end = s_end;
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->last_block = block; h->last_grid = (int)grid;
    if (stats) {
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        uint64_t c[4];
        HIP_TRY(h, hipMemcpy(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost));
        memset(stats, 0, sizeof *stats);
        stats->kernel_ms = ms;
        stats->segments = c[0]; stats->node_tests = c[1]; stats->prim_tests = c[2]; stats->texel_fetches = c[3];
        stats->samples = (uint64_t)args.cam.W * (uint64_t)args.cam.H * (uint64_t)(args.sample_end - args.sample_begin);
        stats->upload_ms = h->upload_ms;
        stats->bvh_entries = args.n_entries;
        stats->scene_in_lds = RES;
#ifdef CR_DIAG
        {
            uint64_t d[64];
            HIP_TRY(h, hipMemcpy(d, h->counters.p, sizeof d, hipMemcpyDeviceToHost));
            const char* names[] = {"box_wave", "box_lane", "prim_wave", "prim_lane", "round_wave", "round_lane", "leafph_wave", "leafph_lane",
                                   "shade_wave", "shade_lane", "lamb_lane", "metal_lane", "diel_lane", "sky_lane", "ruv_wave", "ruv_lane",
                                   "regen_wave", "regen_lane", "outer_wave", "unwind_wave", "unwind_lane", "hitsh_wave", "hitsh_lane", "band_wave", "band_lane"};
            fprintf(stderr, "[diag] block=%d grid=%u clk_regen=%llu clk_trace=%llu clk_shade=%llu clk_total=%llu", block, grid,
                    (unsigned long long)d[9], (unsigned long long)d[10], (unsigned long long)d[11], (unsigned long long)d[12]);
            for (int i = 0; i < DG_N; i++) fprintf(stderr, " %s=%llu", names[i], (unsigned long long)d[16 + i]);
            fprintf(stderr, "\n");
        }
#endif
    }
    return CR_OK;
}


// ---------------------------------------------------------------- LDS-queue megakernel (queue.hpp)
template <typename real, int RES, bool ANIM>
int32_t launch_queue(CrHandle* h, const KernelArgs<real>& args_in, size_t scene_lds_bytes, CrStats* stats) {
    KernelArgs<real> args = args_in;
    auto kern = queue_kernel<real, RES, ANIM>;
    const size_t lds_bytes = ((scene_lds_bytes + 15) & ~(size_t)15) + queue_state_bytes<real>();
    HIP_TRY(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu = 0;
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int)QK_SLOTS, lds_bytes));
    if (per_cu < 1) return fail(h, CR_ERR_HIP, "queue kernel does not fit on a CU");
    const uint32_t total_work = args.tiles_x * args.tiles_y * 64u;
    uint32_t grid = (uint32_t)(h->n_cus * per_cu);
    const uint32_t need_blocks = (total_work + QK_SLOTS - 1) / QK_SLOTS;
    if (grid > need_blocks) grid = need_blocks;
    if (grid < 1) grid = 1;
    args.n_threads = grid * QK_SLOTS;
    args.queue_walk_waves = (uint32_t)h->queue_walk_waves;
    args.queue_min_batch = (uint32_t)h->queue_min_batch; args.queue_patience = (uint32_t)h->queue_patience;
    HIP_TRY(h, h->att_stack.ensure((size_t)3 * (size_t)std::max(1, args.max_depth) * args.n_threads * sizeof(real)));
    args.att_stack = (real*)h->att_stack.p;
    HIP_TRY(h, hipMemsetAsync(h->work_counter.p, 0, 4, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, 64 * sizeof(uint64_t), h->stream));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(QK_SLOTS), lds_bytes, h->stream, args);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->last_block = (int)QK_SLOTS; h->last_grid = (int)grid; h->check_abort = true;
    if (stats) {
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        uint64_t c[5];
        HIP_TRY(h, hipMemcpy(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost));
        h->check_abort = false;
        if (c[4]) return fail(h, CR_ERR_HIP, "queue pipeline: a wave timed out waiting on an LDS queue (image incomplete)");
        memset(stats, 0, sizeof *stats);
        stats->kernel_ms = ms;
        stats->segments = c[0]; stats->node_tests = c[1]; stats->prim_tests = c[2]; stats->texel_fetches = c[3];
        stats->samples = (uint64_t)args.cam.W * (uint64_t)args.cam.H * (uint64_t)(args.sample_end - args.sample_begin);
        stats->upload_ms = h->upload_ms;
        stats->bvh_entries = args.n_entries;
        stats->scene_in_lds = RES;
    }
    return CR_OK;
}

// ---------------------------------------------------------------- wavefront pipeline driver
template <typename real, int RES, bool ANIM>
int32_t wf_extend_config(CrHandle* h, size_t lds_bytes, int& block, int& grid) {
    constexpr bool LDS = RES != RES_GLOBAL;
    auto kern = wf_extend_kernel<real, RES, ANIM>;
    if (LDS) HIP_TRY(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int best_waves = 0, per_cu = 1;
    block = 256;
    for (int cand : {1024, 512, 256}) {
        if (cand > MaxBlock<real>::value) continue;
        if (h->block_override > 0 && cand != h->block_override) continue;
        int n = 0;
        HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, cand, LDS ? lds_bytes : 0));
        if (n * cand / 64 > best_waves) { best_waves = n * cand / 64; block = cand; per_cu = n; }
    }
    if (best_waves == 0) return fail(h, CR_ERR_HIP, "extend kernel does not fit on a CU");
    if (h->blocks_per_cu_override > 0) per_cu = h->blocks_per_cu_override;
    grid = h->n_cus * per_cu;
    return CR_OK;
}

template <typename real, int RES, bool ANIM>
int32_t wf_run(CrHandle* h, WfArgs<real>& W, size_t lds_bytes, int32_t s_begin, int32_t s_count, int32_t batch_cap, CrStats* stats) {
    constexpr bool LDS = RES != RES_GLOBAL;
    int block = 256, grid = 1;
    int32_t rc = wf_extend_config<real, RES, ANIM>(h, lds_bytes, block, grid);
    if (rc != CR_OK) return rc;
    const size_t npix = (size_t)W.k.cam.W * W.k.cam.H;
    const uint32_t logic_grid = (W.n_slots + 255) / 256;
    const uint32_t fin_grid = (uint32_t)((npix + 255) / 256);
    HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, 64 * sizeof(uint64_t), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->wf_acc.p, 0, npix * 3 * sizeof(real), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->wf_job.p, 0xFF, (size_t)W.n_slots * 4, h->stream));
    // no slot may look like it holds a ray before the logic kernel gives it one: recycled device memory can hold
    // WF_PENDING from an earlier handle, and extend would walk that slot's stale ray (extra node tests, same image)
    HIP_TRY(h, hipMemsetAsync(h->wf_hit_prim.p, 0, (size_t)W.n_slots * 4, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int iterations = 0;
    const int LAG = 4, RING = 8;
    for (int32_t b0 = s_begin; b0 < s_begin + s_count; b0 += batch_cap) {
        W.batch_begin = b0;
        W.batch_samples = std::min(batch_cap, s_begin + s_count - b0);
        W.n_jobs = (uint32_t)W.batch_samples * W.total_work;
        W.last_batch = (b0 + W.batch_samples >= s_begin + s_count) ? 1 : 0;
        HIP_TRY(h, hipMemsetAsync(h->wf_ctrl.p, 0, 1024, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->wf_chunk.p, 0, ((size_t)W.n_slots / 64 + 1) * 8, h->stream));
        for (int it = 0;; it++) {
            W.ctrl_set = (uint32_t)(it & 1);
            W.ring_slot = h->wf_ring_dev + (it % RING);
            hipLaunchKernelGGL((wf_logic_kernel<real, ANIM>), dim3(logic_grid), dim3(256), 0, h->stream, W);
            hipLaunchKernelGGL((wf_extend_kernel<real, RES, ANIM>), dim3(grid), dim3(block), LDS ? lds_bytes : 0, h->stream, W);
            HIP_TRY(h, hipEventRecord(h->wf_ev[it % RING], h->stream));
            iterations++;
            if (it >= LAG) {   // lagged check: the GPU is already LAG iterations ahead, so it never waits for the host
                int k = it - LAG;
                HIP_TRY(h, hipEventSynchronize(h->wf_ev[k % RING]));
                if (h->wf_ring_host[k % RING] == 0) break;   // logic found nothing to trace and no job left: batch done
            }
        }
        hipLaunchKernelGGL((wf_finalize_kernel<real>), dim3(fin_grid), dim3(256), 0, h->stream, W);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->last_block = block; h->last_grid = grid; h->wf_last_iterations = iterations;
    if (stats) {
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        uint64_t c[4];
        HIP_TRY(h, hipMemcpy(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost));
        memset(stats, 0, sizeof *stats);
        stats->kernel_ms = ms;
        stats->segments = c[0]; stats->node_tests = c[1]; stats->prim_tests = c[2]; stats->texel_fetches = c[3];
        stats->samples = (uint64_t)npix * (uint64_t)s_count;
        stats->upload_ms = h->upload_ms;
        stats->bvh_entries = W.k.n_entries;
        stats->scene_in_lds = RES;
#ifdef CR_DIAG
        {
            uint64_t d[16];
            HIP_TRY(h, hipMemcpy(d, h->counters.p, sizeof d, hipMemcpyDeviceToHost));
            fprintf(stderr, "[diag-wf] block=%d grid=%d iterations=%d rounds=%llu walk_wave_steps=%llu leaf_lane=%llu leaf_wave=%llu clk_refill=%llu clk_walk=%llu clk_leaf=%llu clk_total=%llu lane_steps=%llu\n",
                    block, grid, iterations, (unsigned long long)d[4], (unsigned long long)d[5], (unsigned long long)d[6], (unsigned long long)d[7],
                    (unsigned long long)d[8], (unsigned long long)d[9], (unsigned long long)d[10], (unsigned long long)d[11], (unsigned long long)d[1]);
        }
#endif
    }
    return CR_OK;
}

template <typename real>
int32_t render_wavefront(CrHandle* h, const KernelArgs<real>& a, DevScene<real>& ds, bool anim, CrStats* stats) {
    WfArgs<real> W;
    memset(&W, 0, sizeof W);
    W.k = a;
    const size_t npix = (size_t)a.cam.W * a.cam.H;
    const int32_t s_begin = a.sample_begin, s_count = a.sample_end - a.sample_begin;
    W.total_work = a.tiles_x * a.tiles_y * 64u;
    // samples per batch: bounded by the per-sample colour buffer and by 32-bit job ids
    int64_t cap = (int64_t)(h->wf_sample_bytes / (npix * 3 * sizeof(real)));
    cap = std::min<int64_t>(cap, (int64_t)0xF0000000u / W.total_work);
    cap = std::max<int64_t>(1, std::min<int64_t>(cap, std::max(1, s_count)));
    const uint64_t jobs_first = (uint64_t)cap * W.total_work;
    W.n_slots = (uint32_t)std::min<uint64_t>(h->wf_slots, (jobs_first + 63) / 64 * 64);
    const size_t N = W.n_slots;
    HIP_TRY(h, h->wf_job.ensure(N * 4));
    HIP_TRY(h, h->wf_rng.ensure(N * 8));
    HIP_TRY(h, h->wf_ray.ensure(N * 7 * sizeof(real)));
    HIP_TRY(h, h->wf_depth.ensure(N * 4));
    HIP_TRY(h, h->wf_hit_t.ensure(N * sizeof(real)));
    HIP_TRY(h, h->wf_hit_prim.ensure(N * 4));
    HIP_TRY(h, h->wf_chunk.ensure((N / 64 + 1) * 8));
    HIP_TRY(h, h->wf_ctrl.ensure(1024));
    HIP_TRY(h, h->wf_samples.ensure((size_t)cap * npix * 3 * sizeof(real)));
    HIP_TRY(h, h->wf_acc.ensure(npix * 3 * sizeof(real)));
    HIP_TRY(h, h->att_stack.ensure((size_t)3 * (size_t)std::max(1, a.max_depth) * N * sizeof(real)));
    if (!h->wf_ring_host) {
        HIP_TRY(h, hipHostMalloc((void**)&h->wf_ring_host, 64, hipHostMallocMapped));
        HIP_TRY(h, hipHostGetDevicePointer((void**)&h->wf_ring_dev, h->wf_ring_host, 0));
        for (int i = 0; i < 8; i++) HIP_TRY(h, hipEventCreateWithFlags(&h->wf_ev[i], hipEventDisableTiming));
    }
    W.k.att_stack = (real*)h->att_stack.p;
    W.k.n_threads = W.n_slots;
    W.job = (uint32_t*)h->wf_job.p; W.rng = (uint64_t*)h->wf_rng.p; W.ray = (real*)h->wf_ray.p; W.depth = (int32_t*)h->wf_depth.p;
    W.hit_t = (real*)h->wf_hit_t.p; W.hit_prim = (int32_t*)h->wf_hit_prim.p; W.job_chunk = (uint32_t*)h->wf_chunk.p;
    W.ctrl = (uint32_t*)h->wf_ctrl.p; W.sample_rgb = (real*)h->wf_samples.p; W.acc = (real*)h->wf_acc.p;
    // the extend kernel stages entries | primitives when both fit, else the top levels of the tree
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t full = r16((size_t)ds.n_entries * sizeof(Entry<real>)) + r16((size_t)ds.n_prims * sizeof(Prim<real>));
    const int32_t bc = (int32_t)cap;
    if (ds.n_entries > 0 && full <= h->lds_limit) {
        W.k.lds_entries = ds.n_entries;
        return anim ? wf_run<real, RES_LDS, true>(h, W, full, s_begin, s_count, bc, stats) : wf_run<real, RES_LDS, false>(h, W, full, s_begin, s_count, bc, stats);
    }
    const int32_t top = (int32_t)std::min<size_t>((size_t)ds.n_entries, h->lds_top_bytes / sizeof(Entry<real>));
    if (top > 0) {
        W.k.lds_entries = top;
        const size_t bytes = (size_t)top * sizeof(Entry<real>);
        return anim ? wf_run<real, RES_TOP, true>(h, W, bytes, s_begin, s_count, bc, stats) : wf_run<real, RES_TOP, false>(h, W, bytes, s_begin, s_count, bc, stats);
    }
    W.k.lds_entries = 0;
    return anim ? wf_run<real, RES_GLOBAL, true>(h, W, 0, s_begin, s_count, bc, stats) : wf_run<real, RES_GLOBAL, false>(h, W, 0, s_begin, s_count, bc, stats);
}

// Picks the kernel variant: keyed primitives (ANIM), camera keys alone (CAMK) or neither, each in the reference's
// summation order or with relaxed sums (CrRenderParams.sum_order).
template <typename real, int RES, bool ORD, bool LATENCY, bool SCREEN = false>
int32_t launch_variant(CrHandle* h, const KernelArgs<real>& a, size_t lds_bytes, CrStats* stats, bool anim, bool cam_keys, bool relax) {
    if (relax) {
        if (anim) return launch<real, RES, true, ORD, LATENCY, false, true, SCREEN>(h, a, lds_bytes, stats);
        if (cam_keys) return launch<real, RES, false, ORD, LATENCY, true, true, SCREEN>(h, a, lds_bytes, stats);
        return launch<real, RES, false, ORD, LATENCY, false, true, SCREEN>(h, a, lds_bytes, stats);
    }
    if (anim) return launch<real, RES, true, ORD, LATENCY, false, false, SCREEN>(h, a, lds_bytes, stats);
    if (cam_keys) return launch<real, RES, false, ORD, LATENCY, true, false, SCREEN>(h, a, lds_bytes, stats);
    return launch<real, RES, false, ORD, LATENCY, false, false, SCREEN>(h, a, lds_bytes, stats);
}

template <typename real>
int32_t render_typed(CrHandle* h, const CrCameraDesc* cd, const CrRenderParams* p, void* d_out, CrStats* stats) {
    int32_t rc = build_dev_scene<real>(h);
    if (rc != CR_OK) return rc;
    DevScene<real>& ds = dev_scene<real>(h);
    if (p->sample_count == 0) {
        // An empty shard (more ranks than samples): the sum of no samples, and 0 / samples for the mean, are both
        // zero -- cast_ray's loop body never runs (ray_casting.rs:82).  No kernel is launched.
        const size_t bytes = (size_t)cd->image_width * (size_t)cd->image_height * 3 * sizeof(real);
        HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
        HIP_TRY(h, hipMemsetAsync(d_out, 0, bytes, h->stream));
        HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
        if (stats) {
            HIP_TRY(h, hipEventSynchronize(h->ev1));
            float ms = 0;
            HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
            memset(stats, 0, sizeof *stats);
            stats->kernel_ms = ms; stats->upload_ms = h->upload_ms; stats->bvh_entries = ds.n_entries;
        }
        return CR_OK;
    }
    KernelArgs<real> a;
    memset(&a, 0, sizeof a);
    a.entries = (const Entry<real>*)ds.entries.p; a.prims = (const Prim<real>*)ds.prims.p; a.leaf_runs = ds.has_leaf_runs ? (const int32_t*)ds.leaf_runs.p : nullptr;
    // without primitive keys the boxes would not change -- unless a HitList element's box is not its objects' union
    const bool refit = p->refit_boxes && (ds.animated || ds.has_lists) && ds.n_entries > 0;
    a.mats = (const Mat<real>*)ds.mats.p; a.texs = (const Tex<real>*)ds.texs.p;
    a.images = (const ImageRef*)h->images.p; a.texels = (const uint32_t*)h->texels.p;
    a.keys = (const Key<real>*)ds.keys.p;
    a.n_entries = ds.n_entries; a.n_prims = ds.n_prims; a.n_mats = ds.n_mats; a.n_texs = ds.n_texs;
    a.sky_kind = h->sky_kind; a.sky_image = h->sky_image;

    // camera set-up: Radians::new_from_degrees (utils.rs:51-55), fix_viewport
    // (rendering_compute.rs:5-11) and defocus_radius (:71-73) in f64, rounded once
    const double PI64 = 3.14159265358979323846264338327950288;
    CamConst<real>& c = a.cam;
    c.W = cd->image_width; c.H = cd->image_height;
    double vfov = cd->vfov_degrees * PI64 / 180.0;
    double hh = std::tan(vfov / 2.0);
    double vh = 2.0 * hh * cd->focus_dist;
    double vw = vh * ((double)cd->image_width / (double)cd->image_height);
    double da = cd->defocus_angle_degrees * PI64 / 180.0;
    c.viewport_height = (real)vh; c.viewport_width = (real)vw; c.focus_dist = (real)cd->focus_dist;
    c.defocus_on = !(da <= 0.0);
    c.defocus_radius = (real)(cd->focus_dist * std::tan(da / 2.0));
    c.from = mk<real>((real)cd->look_from[0], (real)cd->look_from[1], (real)cd->look_from[2]);
    c.at = mk<real>((real)cd->look_at[0], (real)cd->look_at[1], (real)cd->look_at[2]);
    c.vup = mk<real>((real)cd->vup[0], (real)cd->vup[1], (real)cd->vup[2]);
    int nk = cd->from_key_count + cd->at_key_count;
    c.animated = nk > 0;
    if ((size_t)nk > CrHandle::kMaxCamKeys) return fail(h, CR_ERR_UNSUPPORTED, "more than 512 camera keyframes");
    c.from_key_first = 0; c.from_key_count = cd->from_key_count;
    c.at_key_first = cd->from_key_count; c.at_key_count = cd->at_key_count;
    a.cam_keys = nullptr;
    if (nk > 0) {   // per-launch slot: never overwritten while an earlier render may still read it
        const int slot = h->cam_next;
        h->cam_next = (slot + 1) % CrHandle::kCamSlots;
        const size_t slot_bytes = CrHandle::kMaxCamKeys * sizeof(Key<double>);
        if (!h->cam_host[slot]) {   // the slot's three resources together, or none of them (a later render tries again)
            void* host = nullptr;
            hipEvent_t ev = nullptr;
            hipError_t e = hipHostMalloc(&host, slot_bytes, hipHostMallocDefault);
            if (e == hipSuccess) e = h->cam_dev[slot].ensure(slot_bytes);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) {
                if (host) (void)hipHostFree(host);
                h->cam_dev[slot].release();
                (void)hipGetLastError();
                return fail(h, CR_ERR_HIP, std::string("camera keyframe slot: ") + hipGetErrorString(e));
            }
            h->cam_host[slot] = host; h->cam_ev[slot] = ev;
        } else HIP_TRY(h, hipEventSynchronize(h->cam_ev[slot]));   // the slot's previous user has finished with it
        Key<real>* ck = (Key<real>*)h->cam_host[slot];
        for (int i = 0; i < cd->from_key_count; i++) key_to_real(cd->from_keys[i], ck[i]);
        for (int i = 0; i < cd->at_key_count; i++) key_to_real(cd->at_keys[i], ck[cd->from_key_count + i]);
        HIP_TRY(h, hipMemcpyAsync(h->cam_dev[slot].p, ck, nk * sizeof(Key<real>), hipMemcpyHostToDevice, h->stream));
        a.cam_keys = (const Key<real>*)h->cam_dev[slot].p;
        h->cam_pending_slot = slot;
    }
    {   // static camera: same expression tree the kernel would evaluate per sample
        V3<real> from = mk<real>(real(0) + c.from.x, real(0) + c.from.y, real(0) + c.from.z);
        V3<real> at = mk<real>(real(0) + c.at.x, real(0) + c.at.y, real(0) + c.at.z);
        from = scale(real(1), from); at = scale(real(1), at);   // build_other_scaler(1.0): s*x
        CamFrame<real> f = camera_frame(c, from, at);
        if (!c.animated) c.from = f.from;
        c.p00 = f.p00; c.pdu = f.pdu; c.pdv = f.pdv; c.ddu = f.ddu; c.ddv = f.ddv;
    }

    a.sample_begin = p->sample_begin; a.sample_end = p->sample_begin + p->sample_count;
    a.samples_total = p->samples; a.max_depth = p->max_depth;
    a.seed_mixed = mix64(p->seed + RNG_GAMMA);
    a.current_time = (real)p->frame * (real(1) / (real)p->frame_rate);                       // ray_casting.rs:77
    a.shutter_length = ((real)p->shutter_angle / real(360)) * (real(1) / (real)p->frame_rate);   // :79
    a.output_sum = p->output_sum;
    if (refit) {   // refit.hpp: wrapper boxes for this frame's ray times [current_time, current_time + shutter_length]
        const size_t bytes = (size_t)ds.n_entries * ds.entry_bytes;
        HIP_TRY(h, ds.entries_refit.ensure(bytes, ds.entries.pad));
        HIP_TRY(h, hipMemcpyAsync(ds.entries_refit.p, ds.entries.p, bytes, hipMemcpyDeviceToDevice, h->stream));
        { int32_t rc = run_box_kernels<real>(h, ds, ds.entries_refit.p, a.current_time, a.current_time + a.shutter_length, true); if (rc != CR_OK) return rc; }
        a.entries = (const Entry<real>*)ds.entries_refit.p;
    }
    // f64 megakernel on an unordered tree: the walk decides its box tests on the f32 screening records (half the bytes
    // per step), see walk_round (A/B in profiles/experiments/r03_screen_ab.txt).
    bool screen = h->pipeline == 0 && ds.screen.p != nullptr && ds.n_entries > 0 && h->screen_boxes && ds.n_entries < (ds.ordered ? kScreenMaxEntriesO : kScreenMaxEntries);
    if (screen) {
        a.screen = ds.screen.p;
        if (refit) {
            int32_t rc = make_screen(h, ds, ds.entries_refit.p, ds.screen_refit);
            if (rc != CR_OK) return rc;
            a.screen = ds.screen_refit.p;
        }
    }
    // a SCREEN kernel stages screening records where the others stage wrappers
    const size_t screen_rec = ds.ordered ? sizeof(ScreenEntryO) : sizeof(ScreenEntry);
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t lds_all_screen = ds.lds_bytes - r16((size_t)ds.n_entries * ds.entry_bytes) + r16((size_t)ds.n_entries * screen_rec);
    a.tiles_x = (uint32_t)(c.W + 7) / 8u; a.tiles_y = (uint32_t)(c.H + 7) / 8u;
    a.work_counter = (uint32_t*)h->work_counter.p;
    a.counters = (uint64_t*)h->counters.p;
    a.out = (real*)d_out;
    a.uniform_kind = (ds.has_spheres && !ds.has_triangles) ? 0 : ((ds.has_triangles && !ds.has_spheres) ? 1 : -1);
    a.walk_exit_lanes = (uint32_t)(h->walk_exit_lanes >= 0 ? h->walk_exit_lanes : (ds.has_triangles ? 40 : 56));
    a.walk_round_steps = (uint32_t)(h->walk_round_steps >= 0 ? h->walk_round_steps : (ds.has_triangles ? 8 : 10));
    a.walk_leaf_min = h->pipeline == 0 ? (uint32_t)(h->walk_leaf_min >= 0 ? h->walk_leaf_min : 8) : 0u;   // the other pipelines test a leaf in the round that found it
    a.sg_on = 0; a.sg_lw = a.sg_lh = 3; a.sg_groups = 0; a.sg_total = 0; a.sample_buf = nullptr;   // set by launch()

    // the ANIM kernels also carry the decode of leaves that hold a HitList element (pathtrace.hpp walk_round)
    const bool anim = ds.animated || ds.has_leaf_runs;   // keyed primitives (the ANIM kernels also follow a keyed camera)
    const bool cam_keys = c.animated;                     // camera keys alone: the static kernels' CAMK variant
    // relaxed sums exist in the megakernel; the alternative pipelines are reference-order cross-checks
    const int sum_order = p->sum_order == CR_SUM_DEFAULT ? (h->pipeline == 0 ? h->default_sum_order : CR_SUM_REFERENCE_ORDER) : p->sum_order;
    if (sum_order == CR_SUM_RELAXED && h->pipeline != 0) return fail(h, CR_ERR_UNSUPPORTED, "CR_SUM_RELAXED is implemented by the megakernel pipeline only");
    const bool relax = sum_order == CR_SUM_RELAXED;
    const size_t fx_need = relax ? fx_lds_bytes(MaxBlock<real>::value, 4) : 0;   // the relaxed sums' slots share the LDS
    const bool screen_lds = screen && h->screen_lds && lds_all_screen + fx_need <= h->lds_limit;
    const bool plain_lds = ds.lds_bytes + fx_need <= h->lds_limit;
    if (ds.ordered) {   // near-child-first walk: megakernel only
        if (h->pipeline != 0) return fail(h, CR_ERR_UNSUPPORTED, "CR_BVH_SAH_ORDERED is implemented by the megakernel pipeline only");
        constexpr bool f32 = std::is_same<real, float>::value;   // the double kernel needs far more than 80 VGPRs: it halves there
        if (ds.n_entries > 0 && (plain_lds || screen_lds)) {
            a.lds_entries = ds.n_entries;
            if constexpr (!f32) if (screen_lds) return launch_variant<real, RES_LDS, true, false, true>(h, a, lds_all_screen, stats, anim, cam_keys, relax);
            a.screen = nullptr;
            return launch_variant<real, RES_LDS, true, false>(h, a, ds.lds_bytes, stats, anim, cam_keys, relax);
        }
        const bool latency = f32 && h->latency_entries > 0 && ds.n_entries > h->latency_entries;
        const size_t window_rec = screen ? sizeof(ScreenEntryO) : sizeof(EntryO<real>);
        const int32_t top = (int32_t)std::min<size_t>((size_t)ds.n_entries, (latency ? h->latency_top_bytes : h->lds_top_bytes) / window_rec);
        if (top > 0) {
            a.lds_entries = top;
            const size_t bytes = (size_t)top * window_rec;
            if constexpr (f32) if (latency) return launch_variant<real, RES_TOP, true, true>(h, a, bytes, stats, anim, cam_keys, relax);
            if constexpr (!f32) if (screen) return launch_variant<real, RES_TOP, true, false, true>(h, a, bytes, stats, anim, cam_keys, relax);
            return launch_variant<real, RES_TOP, true, false>(h, a, bytes, stats, anim, cam_keys, relax);
        }
        a.lds_entries = 0;
        if constexpr (!f32) if (screen) return launch_variant<real, RES_GLOBAL, true, false, true>(h, a, 0, stats, anim, cam_keys, relax);
        return launch_variant<real, RES_GLOBAL, true, false>(h, a, 0, stats, anim, cam_keys, relax);
    }
    if (h->pipeline == 1) return render_wavefront<real>(h, a, ds, anim || cam_keys, stats);
    if (h->pipeline == 2) {   // LDS-queue megakernel when scene + slot arrays fit in LDS, else the plain megakernel below
        const size_t budget = 160 * 1024, state = queue_state_bytes<real>();
        if (ds.n_entries > 0 && ds.lds_bytes + 16 + state <= budget) {
            a.lds_entries = ds.n_entries;
            return (anim || cam_keys) ? launch_queue<real, RES_LDS, true>(h, a, ds.lds_bytes, stats) : launch_queue<real, RES_LDS, false>(h, a, ds.lds_bytes, stats);
        }
        if (ds.n_entries > 0 && state + 16 * 1024 <= budget) {
            const size_t top_bytes = std::min(h->lds_top_bytes, (budget - state - 64) & ~(size_t)1023);
            a.lds_entries = (int32_t)std::min<size_t>((size_t)ds.n_entries, top_bytes / sizeof(Entry<real>));
            const size_t bytes = (size_t)a.lds_entries * sizeof(Entry<real>);
            return (anim || cam_keys) ? launch_queue<real, RES_TOP, true>(h, a, bytes, stats) : launch_queue<real, RES_TOP, false>(h, a, bytes, stats);
        }
    }
    if (ds.n_entries > 0 && (plain_lds || screen_lds)) {
        a.lds_entries = ds.n_entries;
        if (screen_lds) return launch_variant<real, RES_LDS, false, false, true>(h, a, lds_all_screen, stats, anim, cam_keys, relax);
        a.screen = nullptr;
        return launch_variant<real, RES_LDS, false, false>(h, a, ds.lds_bytes, stats, anim, cam_keys, relax);
    }
    constexpr bool f32 = std::is_same<real, float>::value;   // the double kernel needs far more than 80 VGPRs: it halves there
    const bool latency = f32 && h->latency_entries > 0 && ds.n_entries > h->latency_entries;
    const size_t window_rec = screen ? sizeof(ScreenEntry) : sizeof(Entry<real>);   // a window of screening records holds twice the wrappers
    const int32_t top = (int32_t)std::min<size_t>((size_t)ds.n_entries, (latency ? h->latency_top_bytes : h->lds_top_bytes) / window_rec);
    if (top > 0) {   // large scene: the top levels of the tree in LDS, everything else through L2
        a.lds_entries = top;
        size_t bytes = (size_t)top * window_rec;
        // materials and textures ride along when they are small (the tree can be large with two materials)
        const size_t side = (((size_t)ds.n_mats * sizeof(Mat<real>) + 15) & ~(size_t)15) + (((size_t)ds.n_texs * sizeof(Tex<real>) + 15) & ~(size_t)15);
        // (the 6-waves-per-SIMD entry point runs three 512-thread groups per CU: window, side tables and the relaxed sums' slots of all three share 160 KB)
        const size_t side_cap = latency ? (((size_t)160 * 1024 / 3 - 16 > bytes + fx_lds_bytes(LatencyBlock, 4)) ? (size_t)160 * 1024 / 3 - 16 - bytes - fx_lds_bytes(LatencyBlock, 4) : 0) : h->lds_side_limit;
        if (side <= std::min(h->lds_side_limit, side_cap)) { a.lds_side = 1; bytes = ((bytes + 15) & ~(size_t)15) + side; }
        if constexpr (f32) if (latency) return launch_variant<real, RES_TOP, false, true>(h, a, bytes, stats, anim, cam_keys, relax);
        if (screen) return launch_variant<real, RES_TOP, false, false, true>(h, a, bytes, stats, anim, cam_keys, relax);
        return launch_variant<real, RES_TOP, false, false>(h, a, bytes, stats, anim, cam_keys, relax);
    }
    a.lds_entries = 0;
    if (screen) return launch_variant<real, RES_GLOBAL, false, false, true>(h, a, 0, stats, anim, cam_keys, relax);
    return launch_variant<real, RES_GLOBAL, false, false>(h, a, 0, stats, anim, cam_keys, relax);
}

int32_t validate_render(CrHandle* h, const CrCameraDesc* cam, const CrRenderParams* p) {
    if (!h) return CR_ERR_INVALID_ARG;
    if (!cam || !p) return fail(h, CR_ERR_INVALID_ARG, "null camera or params");
    if (!h->has_scene) return fail(h, CR_ERR_NO_SCENE, "cr_render before cr_upload_scene");
    if (cam->image_width < 1 || cam->image_height < 1) return fail(h, CR_ERR_INVALID_ARG, "image size must be positive");
    if ((int64_t)cam->image_width * cam->image_height > (int64_t)1 << 26) return fail(h, CR_ERR_INVALID_ARG, "image too large");
    if (p->samples < 1) return fail(h, CR_ERR_INVALID_ARG, "The camera must have a positive number of samples.");   // camera/mod.rs:235-238
    if (p->sample_begin < 0 || p->sample_count < 0 || p->sample_begin + p->sample_count > p->samples)
        return fail(h, CR_ERR_INVALID_ARG, "sample range outside [0, samples)");
    if (p->max_depth < 0) return fail(h, CR_ERR_INVALID_ARG, "max_depth must be >= 0");
    if (p->real_type != CR_REAL_F32 && p->real_type != CR_REAL_F64) return fail(h, CR_ERR_INVALID_ARG, "unknown real_type");
    if (p->sum_order != CR_SUM_DEFAULT && p->sum_order != CR_SUM_REFERENCE_ORDER && p->sum_order != CR_SUM_RELAXED) return fail(h, CR_ERR_INVALID_ARG, "unknown sum_order");
    if (!(p->frame_rate > 0)) return fail(h, CR_ERR_INVALID_ARG, "frame_rate must be positive");
    if ((cam->from_key_count > 0 && !cam->from_keys) || (cam->at_key_count > 0 && !cam->at_keys) || cam->from_key_count < 0 || cam->at_key_count < 0)
        return fail(h, CR_ERR_INVALID_ARG, "camera keyframe array missing");
    for (int i = 0; i < cam->from_key_count + cam->at_key_count; i++) {   // cam_translate_* only (scene_animator.rs)
        const CrKeyframe& k = i < cam->from_key_count ? cam->from_keys[i] : cam->at_keys[i - cam->from_key_count];
        if (k.channel < CR_KEY_TX || k.channel > CR_KEY_TZ || (k.interp != CR_KEY_NERP && k.interp != CR_KEY_LERP))
            return fail(h, CR_ERR_INVALID_ARG, "camera keyframes are translations (channels 0..2)");
    }
    return CR_OK;
}

size_t real_size(int32_t real_type) { return real_type == CR_REAL_F64 ? sizeof(double) : sizeof(float); }

uint32_t display_byte(double c) {   // impl Display for Color, utils.rs:422-437: (255.0 * c.sqrt()) as u32
    double v = 255.0 * std::sqrt(c);
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

}   // namespace

// cr_export_bvh: the wrapper tree the device walks, re-expressed as the reference's BVHWrapper tree (each
// wrapper = box + left/right child) in walk order.  A leaf wrapper of one primitive holds it twice, as the
// reference's span-1 wrappers do (bvhwrapper.rs:58-60).
template <typename real>
int32_t export_bvh(CrHandle* h, double* boxes, int32_t* children, int32_t* split_axis, int32_t capacity, int32_t* n_out) {
    int32_t rc = build_dev_scene<real>(h);
    if (rc != CR_OK) return rc;
    const DevScene<real>& ds = dev_scene<real>(h);
    if (ds.has_bvh_elements) return fail(h, CR_ERR_UNSUPPORTED, "cr_export_bvh: the scene holds a BVHWrapper element (CR_BVH_REFERENCE): its records are not two-children wrappers");
    const std::vector<Entry<real>>& E = ds.host_entries;
    *n_out = (int32_t)E.size();
    if (!boxes || !children || capacity < (int32_t)E.size()) return E.empty() || (!boxes && !children) ? CR_OK : fail(h, CR_ERR_INVALID_ARG, "cr_export_bvh: capacity too small");
    if (E.empty()) return CR_OK;
    struct Frame { int32_t entry, out, state; };
    std::vector<Frame> fr{{0, -1, 0}};
    int32_t n = 0;
    while (!fr.empty()) {
        Frame& f = fr.back();
        const Entry<real>& e = E[f.entry];
        if (f.state == 0) {
            f.out = n++;
            for (int k = 0; k < 6; k++) boxes[6 * f.out + k] = (double)e.b[k];
            if (split_axis) split_axis[f.out] = (ds.ordered && e.leaf < 0) ? (int32_t)ds.host_axis[f.entry] : -1;
            if (e.leaf >= 0) {
                const int32_t first = e.leaf >> 1, count = (e.leaf & 1) + 1;
                children[2 * f.out] = ~ds.leaf_desc[first];
                children[2 * f.out + 1] = ~ds.leaf_desc[first + count - 1];
                fr.pop_back();
                continue;
            }
            f.state = 1;
            children[2 * f.out] = n;                 // the left child is exported next
            const int32_t left = -e.leaf;
            fr.push_back({left, -1, 0});
        } else if (f.state == 1) {
            f.state = 2;
            children[2 * f.out + 1] = n;
            const int32_t right = E[-e.leaf].skip;   // the wrapper after the left subtree
            fr.push_back({right, -1, 0});
        } else fr.pop_back();
    }
    return CR_OK;
}


extern "C" {

int32_t cr_abi_version(void) { return CR_ABI_VERSION; }

int32_t cr_create(int32_t device_id, CrHandle** out) {
    if (!out) return fail(nullptr, CR_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1) return fail(nullptr, CR_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_dev) return fail(nullptr, CR_ERR_INVALID_ARG, "device_id out of range");
    CrHandle* h = new CrHandle();
    h->device = device_id;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        delete h;
        return CR_ERR_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    h->n_cus = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = h->work_counter.ensure(16)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = h->counters.ensure(64 * sizeof(uint64_t))) != hipSuccess) return bail("hipMalloc", e);
    if (const char* s = getenv("CRUCIBLE_LDS_LIMIT")) h->lds_limit = (size_t)atol(s);
    if (const char* s = getenv("CRUCIBLE_SAMPLE_GRANULAR")) h->sample_granular = atoi(s) != 0;
    if (const char* s = getenv("CRUCIBLE_SCREEN")) h->screen_boxes = atoi(s) != 0;
    if (const char* s = getenv("CRUCIBLE_SCREEN_LDS")) h->screen_lds = atoi(s) != 0;
    if (const char* s = getenv("CRUCIBLE_SUM_ORDER")) h->default_sum_order = strcmp(s, "reference") == 0 ? CR_SUM_REFERENCE_ORDER : CR_SUM_RELAXED;
    if (const char* s = getenv("CRUCIBLE_SAMPLE_BUF_MB")) h->sample_buf_limit = (size_t)std::max(0L, atol(s)) << 20;
    if (const char* s = getenv("CRUCIBLE_SG_CHUNK")) h->sg_chunk_override = std::max(0, atoi(s));
    if (const char* s = getenv("CRUCIBLE_WORK_COUNTER_MAX")) h->work_counter_max = std::min<uint64_t>(0xF0000000ull, (uint64_t)std::max(64LL, atoll(s)));
    if (const char* s = getenv("CRUCIBLE_SG_TILE")) {
        int tw = 0, th = 0;
        if (sscanf(s, "%dx%d", &tw, &th) == 2 && tw > 0 && th > 0 && (tw & (tw - 1)) == 0 && (th & (th - 1)) == 0 && tw * th <= 64) {
            h->sg_lw = __builtin_ctz((unsigned)tw); h->sg_lh = __builtin_ctz((unsigned)th);
        }
    }
    if (const char* s = getenv("CRUCIBLE_LATENCY_ENTRIES")) h->latency_entries = (int32_t)std::max(0L, atol(s));
    if (const char* s = getenv("CRUCIBLE_LATENCY_TOP_KB")) h->latency_top_bytes = (size_t)std::max(0L, atol(s)) * 1024;
    if (const char* s = getenv("CRUCIBLE_LDS_SIDE_KB")) h->lds_side_limit = (size_t)std::max(0L, atol(s)) * 1024;
    if (const char* s = getenv("CRUCIBLE_LDS_TOP_KB")) { h->lds_top_bytes = (size_t)std::max(0L, atol(s)) * 1024; h->lds_top_set = true; }
    if (const char* s = getenv("CRUCIBLE_BLOCKS_PER_CU")) h->blocks_per_cu_override = atoi(s);
    if (const char* s = getenv("CRUCIBLE_BLOCK")) h->block_override = atoi(s);
    if (const char* s = getenv("CRUCIBLE_WALK_ROUND")) h->walk_round_steps = std::max(0, atoi(s));
    if (const char* s = getenv("CRUCIBLE_WALK_EXIT")) h->walk_exit_lanes = std::min(64, std::max(1, atoi(s)));
    if (const char* s = getenv("CRUCIBLE_WALK_LEAF_MIN")) h->walk_leaf_min = std::min(64, std::max(0, atoi(s)));
    if (const char* s = getenv("CRUCIBLE_PIPELINE")) h->pipeline = strcmp(s, "mega") == 0 ? 0 : (strcmp(s, "queue") == 0 ? 2 : 1);
    if (const char* s = getenv("CRUCIBLE_QUEUE_BATCH")) h->queue_min_batch = std::min(64, std::max(1, atoi(s)));
    if (const char* s = getenv("CRUCIBLE_QUEUE_PATIENCE")) h->queue_patience = std::max(0, atoi(s));
    if (const char* s = getenv("CRUCIBLE_QUEUE_WALKERS")) h->queue_walk_waves = std::min(15, std::max(1, atoi(s)));
    if (const char* s = getenv("CRUCIBLE_WF_SLOTS")) h->wf_slots = (uint32_t)std::max(64L, atol(s));
    if (const char* s = getenv("CRUCIBLE_WF_SAMPLE_MB")) h->wf_sample_bytes = (size_t)std::max(1L, atol(s)) << 20;
    *out = h;
    return CR_OK;
}

void cr_destroy(CrHandle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->s32.release(); h->s64.release();
    h->images.release(); h->texels.release(); h->work_counter.release(); h->counters.release();
    h->att_stack.release(); h->out_buf.release(); h->sample_buf.release(); h->sg_acc.release(); h->fx_acc.release();
    h->wf_job.release(); h->wf_rng.release(); h->wf_ray.release(); h->wf_depth.release(); h->wf_hit_t.release(); h->wf_hit_prim.release();
    h->wf_chunk.release(); h->wf_ctrl.release(); h->wf_samples.release(); h->wf_acc.release();
    for (int i = 0; i < CrHandle::kCamSlots; i++) {
        if (h->cam_host[i]) (void)hipHostFree(h->cam_host[i]);
        h->cam_dev[i].release();
        if (h->cam_ev[i]) (void)hipEventDestroy(h->cam_ev[i]);
    }
    if (h->wf_ring_host) { (void)hipHostFree(h->wf_ring_host); for (int i = 0; i < 8; i++) if (h->wf_ev[i]) (void)hipEventDestroy(h->wf_ev[i]); }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int32_t cr_upload_scene(CrHandle* h, const CrSceneDesc* s) {
    if (!h) return CR_ERR_INVALID_ARG;
    if (!s) return fail(h, CR_ERR_INVALID_ARG, "scene is null");
    if (s->n_prims < 0 || s->n_materials < 0 || s->n_textures < 0 || s->n_images < 0 || s->n_keys < 0)
        return fail(h, CR_ERR_INVALID_ARG, "negative count");
    if (s->n_prims >= (1 << 29)) return fail(h, CR_ERR_INVALID_ARG, "too many primitives");
    if ((s->n_prims > 0 && !s->prims) || (s->n_materials > 0 && !s->materials) || (s->n_textures > 0 && !s->textures) ||
        (s->n_images > 0 && !s->images) || (s->n_keys > 0 && !s->keys))
        return fail(h, CR_ERR_INVALID_ARG, "a descriptor array is null although its count is not zero");
    auto finite = [](double x) { return x == x && x != HUGE_VAL && x != -HUGE_VAL; };
    for (int i = 0; i < s->n_textures; i++) {
        const CrTexture& t = s->textures[i];
        if (t.kind < CR_TEX_SOLID || t.kind > CR_TEX_IMAGE) return fail(h, CR_ERR_INVALID_ARG, "unknown texture kind");
        // children before parents keeps the texture graph acyclic (Arc<Textures> cannot cycle either)
        if (t.kind == CR_TEX_CHECKER && (t.even < 0 || t.even >= i || t.odd < 0 || t.odd >= i))
            return fail(h, CR_ERR_INVALID_ARG, "checker sub-textures must have smaller indices");
        if (t.kind == CR_TEX_IMAGE && (t.image < 0 || t.image >= s->n_images)) return fail(h, CR_ERR_INVALID_ARG, "texture image index out of range");
        if (t.kind == CR_TEX_SOLID) for (int k = 0; k < 3; k++) if (!(t.color[k] >= 0.0 && t.color[k] <= 1.0))
            return fail(h, CR_ERR_INVALID_ARG, "colour component outside [0,1]");   // Color::new, utils.rs:345-350
    }
    {   // the device resolves a checker chain iteratively with a bound of 32 levels (pathtrace.hpp, shade)
        std::vector<int32_t> depth((size_t)s->n_textures, 0);
        for (int i = 0; i < s->n_textures; i++) {
            const CrTexture& t = s->textures[i];
            if (t.kind != CR_TEX_CHECKER) continue;
            depth[i] = 1 + std::max(depth[t.even], depth[t.odd]);
            if (depth[i] > CR_MAX_CHECKER_DEPTH) return fail(h, CR_ERR_UNSUPPORTED, "checker textures nested deeper than CR_MAX_CHECKER_DEPTH (32)");
        }
    }
    for (int i = 0; i < s->n_materials; i++) {
        const CrMaterial& m = s->materials[i];
        if (m.kind < CR_MAT_LAMBERTIAN || m.kind > CR_MAT_DIELECTRIC) return fail(h, CR_ERR_INVALID_ARG, "unknown material kind");
        if (m.kind == CR_MAT_LAMBERTIAN && (m.texture < 0 || m.texture >= s->n_textures)) return fail(h, CR_ERR_INVALID_ARG, "material texture index out of range");
        if (m.kind == CR_MAT_METAL) {
            if (!(m.param <= 1.0)) return fail(h, CR_ERR_INVALID_ARG, "A metal cannot have a fuzz factor above 1.0");   // metal.rs:21
            if (!(m.param >= 0.0)) return fail(h, CR_ERR_INVALID_ARG, "A metal cannot have a fuzz factor below 0.0");   // metal.rs:22
            for (int k = 0; k < 3; k++) if (!(m.albedo[k] >= 0.0 && m.albedo[k] <= 1.0)) return fail(h, CR_ERR_INVALID_ARG, "colour component outside [0,1]");
        }
        if (!finite(m.param)) return fail(h, CR_ERR_INVALID_ARG, "material parameter is not finite");
    }
    for (int i = 0; i < s->n_keys; i++) {
        const CrKeyframe& k = s->keys[i];
        if (k.channel < CR_KEY_TX || k.channel > CR_KEY_SCALE_Z || (k.interp != CR_KEY_NERP && k.interp != CR_KEY_LERP))
            return fail(h, CR_ERR_INVALID_ARG, "bad keyframe");
    }
    {   // lists (CR_PRIM_LIST): whole-number ranges of flagged spheres/triangles, every flagged primitive in exactly one
        std::vector<char> owned((size_t)std::max(0, s->n_prims), 0);
        for (int i = 0; i < s->n_prims; i++) {
            const CrPrimitive& p = s->prims[i];
            if (p.kind != CR_PRIM_LIST && p.kind != CR_PRIM_BVH) continue;
            if (p.flags & (CR_PRIM_MEMBER | CR_PRIM_HIDDEN)) return fail(h, CR_ERR_INVALID_ARG, "a list is a scene element: it cannot be hidden or be an object of a list");
            if (p.kind == CR_PRIM_BVH && (p.flags & CR_LIST_EMPTY_BOX)) return fail(h, CR_ERR_INVALID_ARG, "CR_LIST_EMPTY_BOX applies to lists");
            const double first = p.v[0], count = p.v[1];
            if (!(first >= 0.0 && count >= 0.0 && first == std::floor(first) && count == std::floor(count) && first + count <= (double)s->n_prims))
                return fail(h, CR_ERR_INVALID_ARG, "list object range out of bounds");
            for (int64_t k = (int64_t)first; k < (int64_t)(first + count); k++) {
                const CrPrimitive& m = s->prims[k];
                if ((m.kind != CR_PRIM_SPHERE && m.kind != CR_PRIM_TRIANGLE) || !(m.flags & CR_PRIM_MEMBER))
                    return fail(h, CR_ERR_INVALID_ARG, "a list's objects must be spheres or triangles flagged CR_PRIM_MEMBER");
                if (owned[(size_t)k]) return fail(h, CR_ERR_INVALID_ARG, "a primitive is an object of two lists");
                owned[(size_t)k] = 1;
            }
        }
        for (int i = 0; i < s->n_prims; i++)
            if ((s->prims[i].flags & CR_PRIM_MEMBER) && !owned[(size_t)i]) return fail(h, CR_ERR_INVALID_ARG, "a primitive flagged CR_PRIM_MEMBER belongs to no list");
    }
    for (int i = 0; i < s->n_prims; i++) {
        const CrPrimitive& p = s->prims[i];
        if (p.kind == CR_PRIM_LIST || p.kind == CR_PRIM_BVH) continue;
        if (p.kind != CR_PRIM_SPHERE && p.kind != CR_PRIM_TRIANGLE) return fail(h, CR_ERR_INVALID_ARG, "unknown primitive kind");
        if (p.material < 0 || p.material >= s->n_materials) return fail(h, CR_ERR_INVALID_ARG, "primitive material index out of range");
        if (p.key_count < 0 || p.key_first < 0 || p.key_first + p.key_count > s->n_keys) return fail(h, CR_ERR_INVALID_ARG, "primitive keyframe range out of bounds");
        for (int k = 0; k < p.key_count; k++) {   // the Scene API type-checks scale keys (scene_animator.rs:38-183)
            const int32_t ch = s->keys[p.key_first + k].channel;
            if (p.kind == CR_PRIM_SPHERE && ch > CR_KEY_RADIUS) return fail(h, CR_ERR_INVALID_ARG, "ScaleX/ScaleY/ScaleZ cannot apply to Spheres");
            if (p.kind == CR_PRIM_TRIANGLE && ch == CR_KEY_RADIUS) return fail(h, CR_ERR_INVALID_ARG, "ScaleR can only be applied to Spheres");
        }
        int nv = p.kind == CR_PRIM_SPHERE ? 4 : 9;
        for (int k = 0; k < nv; k++) if (!finite(p.v[k])) return fail(h, CR_ERR_INVALID_ARG, "primitive coordinate is not finite");
        if (p.kind == CR_PRIM_SPHERE && !(p.v[3] >= 0.0)) return fail(h, CR_ERR_INVALID_ARG, "Cannot make a sphere with negative radius");   // sphere.rs:26
    }
    if (s->sky_kind != CR_SKY_DEFAULT && s->sky_kind != CR_SKY_SPHERICAL) return fail(h, CR_ERR_INVALID_ARG, "unknown sky kind");
    if (s->bvh_mode < CR_BVH_REFERENCE || s->bvh_mode > CR_BVH_LBVH) return fail(h, CR_ERR_INVALID_ARG, "unknown bvh_mode");
    if (s->sky_kind == CR_SKY_SPHERICAL && (s->sky_image < 0 || s->sky_image >= s->n_images)) return fail(h, CR_ERR_INVALID_ARG, "sky image index out of range");
    for (int i = 0; i < s->n_images; i++)
        if (s->images[i].width < 1 || s->images[i].height < 1 || !s->images[i].rgb8) return fail(h, CR_ERR_INVALID_ARG, "bad image");

    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto t_begin = std::chrono::steady_clock::now();
    h->prims.assign(s->prims, s->prims + s->n_prims);
    h->materials.assign(s->materials, s->materials + s->n_materials);
    h->textures.assign(s->textures, s->textures + s->n_textures);
    h->keys.assign(s->keys, s->keys + s->n_keys);
    h->sky_kind = s->sky_kind; h->sky_image = s->sky_image; h->bvh_mode = s->bvh_mode;
    h->s32.built = false; h->s64.built = false;
    // images: RGB8 -> RGBA8 words, one flat texel array
    std::vector<ImageRef> refs(s->n_images);
    size_t total = 0;
    for (int i = 0; i < s->n_images; i++) {
        refs[i].w = s->images[i].width; refs[i].h = s->images[i].height; refs[i].offset = (uint32_t)total; refs[i].pad = 0;
        total += (size_t)s->images[i].width * s->images[i].height;
    }
    if (total >= ((size_t)1 << 32)) return fail(h, CR_ERR_INVALID_ARG, "too many texels");
    std::vector<uint32_t> texels(total ? total : 1);
    for (int i = 0; i < s->n_images; i++) {
        const uint8_t* src = s->images[i].rgb8;
        size_t n = (size_t)refs[i].w * refs[i].h;
        uint32_t* dst = texels.data() + refs[i].offset;
        for (size_t k = 0; k < n; k++) dst[k] = (uint32_t)src[3 * k] | ((uint32_t)src[3 * k + 1] << 8) | ((uint32_t)src[3 * k + 2] << 16);
    }
    HIP_TRY(h, h->images.ensure(refs.size() * sizeof(ImageRef) + 16));
    HIP_TRY(h, h->texels.ensure(texels.size() * 4));
    if (!refs.empty()) HIP_TRY(h, hipMemcpy(h->images.p, refs.data(), refs.size() * sizeof(ImageRef), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->texels.p, texels.data(), texels.size() * 4, hipMemcpyHostToDevice));
    h->n_images = s->n_images;
    h->has_scene = true;
    h->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return CR_OK;
}

int32_t cr_render_device(CrHandle* h, const CrCameraDesc* cam, const CrRenderParams* p, void* d_out, CrStats* stats) {
    int32_t rc = validate_render(h, cam, p);
    if (rc != CR_OK) return rc;
    if (!d_out) return fail(h, CR_ERR_INVALID_ARG, "output buffer is null");
    HIP_TRY(h, hipSetDevice(h->device));
    h->cam_pending_slot = -1;
    rc = p->real_type == CR_REAL_F64 ? render_typed<double>(h, cam, p, d_out, stats) : render_typed<float>(h, cam, p, d_out, stats);
    if (h->cam_pending_slot >= 0) {   // the camera-key slot is free again once everything queued so far has run
        hipError_t e = hipEventRecord(h->cam_ev[h->cam_pending_slot], h->stream);
        h->cam_pending_slot = -1;
        if (e != hipSuccess && rc == CR_OK) { h->error = std::string("hipEventRecord: ") + hipGetErrorString(e); rc = CR_ERR_HIP; }
    }
    return rc;
}

int32_t cr_render_host(CrHandle* h, const CrCameraDesc* cam, const CrRenderParams* p, void* h_out, CrStats* stats) {
    int32_t rc = validate_render(h, cam, p);
    if (rc != CR_OK) return rc;
    if (!h_out) return fail(h, CR_ERR_INVALID_ARG, "output buffer is null");
    HIP_TRY(h, hipSetDevice(h->device));
    size_t n = (size_t)cam->image_width * cam->image_height * 3;
    size_t bytes = n * real_size(p->real_type);
    HIP_TRY(h, h->out_buf.ensure(bytes));
    CrStats local;
    rc = cr_render_device(h, cam, p, h->out_buf.p, stats ? stats : &local);
    if (rc != CR_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h_out, h->out_buf.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!p->output_sum) {   // Color::new asserts 0 <= c <= 1 on every mean (ray_casting.rs:172)
        uint64_t bad = 0;
        size_t n_pix = n / 3;
        for (size_t i = 0; i < n_pix; i++) {
            bool ok = true;
            for (int k = 0; k < 3; k++) {
                double v = p->real_type == CR_REAL_F64 ? ((const double*)h_out)[3 * i + k] : (double)((const float*)h_out)[3 * i + k];
                ok = ok && (v >= 0.0 && v <= 1.0);
            }
            bad += ok ? 0 : 1;
        }
        if (stats) stats->nan_pixels = bad;
        if (bad) return fail(h, CR_ERR_NAN, "a pixel mean is NaN or outside [0,1] (the reference panics in Color::new)");
    }
    return CR_OK;
}

static int32_t check_queue_abort(CrHandle* h) {
    if (!h->check_abort) return CR_OK;
    uint64_t aborted = 0;
    HIP_TRY(h, hipMemcpy(&aborted, (uint64_t*)h->counters.p + 4, sizeof aborted, hipMemcpyDeviceToHost));
    h->check_abort = false;
    if (aborted) return fail(h, CR_ERR_HIP, "queue pipeline: a wave timed out waiting on an LDS queue (image incomplete)");
    return CR_OK;
}

int32_t cr_export_bvh(CrHandle* h, int32_t real_type, double* boxes, int32_t* children, int32_t* split_axis, int32_t capacity,
                      int32_t* n_wrappers) {
    if (!h) return CR_ERR_INVALID_ARG;
    if (!n_wrappers) return fail(h, CR_ERR_INVALID_ARG, "cr_export_bvh: null n_wrappers");
    if (!h->has_scene) return fail(h, CR_ERR_NO_SCENE, "cr_export_bvh before cr_upload_scene");
    if (real_type != CR_REAL_F32 && real_type != CR_REAL_F64) return fail(h, CR_ERR_INVALID_ARG, "unknown real_type");
    HIP_TRY(h, hipSetDevice(h->device));
    return real_type == CR_REAL_F64 ? export_bvh<double>(h, boxes, children, split_axis, capacity, n_wrappers)
                                    : export_bvh<float>(h, boxes, children, split_axis, capacity, n_wrappers);
}

int32_t cr_last_kernel_ms(CrHandle* h, double* out_ms) {
    if (!h || !out_ms) return CR_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *out_ms = ms;
    return check_queue_abort(h);
}

int32_t cr_synchronize(CrHandle* h) {
    if (!h) return CR_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_queue_abort(h);
}

void* cr_stream(CrHandle* h) { return h ? (void*)h->stream : nullptr; }

int32_t cr_quantize_rgb8(const void* rgb, int32_t real_type, int64_t n_pixels, uint8_t* out) {
    if (!rgb || !out || n_pixels < 0 || (real_type != CR_REAL_F32 && real_type != CR_REAL_F64)) return CR_ERR_INVALID_ARG;
    for (int64_t i = 0; i < n_pixels * 3; i++) {
        double v = real_type == CR_REAL_F64 ? ((const double*)rgb)[i] : (double)((const float*)rgb)[i];
        uint32_t b = display_byte(v);
        out[i] = (uint8_t)(b > 255u ? 255u : b);
    }
    return CR_OK;
}

int32_t cr_write_ppm(const char* path, const void* rgb, int32_t real_type, int32_t w, int32_t hgt) {
    if (!path || !rgb || w < 1 || hgt < 1 || (real_type != CR_REAL_F32 && real_type != CR_REAL_F64)) return CR_ERR_INVALID_ARG;
    FILE* f = fopen(path, "w");   // OpenOptions write+create+truncate, camera/mod.rs:275-279
    if (!f) return CR_ERR_IO;
    // The text of `writeln!(file, "{color}")` per pixel (camera/mod.rs:306-311, utils.rs:422-437), formatted into memory
    // rows at a time: 2 M fprintf calls per 1080p frame took 0.3 s, as long as the frame's render.
    struct Dec { char s[4]; uint8_t n; };
    static const std::vector<Dec> table = [] { std::vector<Dec> t(256); for (int v = 0; v < 256; v++) t[(size_t)v].n = (uint8_t)snprintf(t[(size_t)v].s, 4, "%d", v); return t; }();
    bool ok = fprintf(f, "P3\n%d %d\n255\n", w, hgt) > 0;   // camera/mod.rs:286
    const int64_t npix = (int64_t)w * hgt, chunk = 1 << 16;
    std::vector<char> buf((size_t)chunk * 36);   // three u32 of up to 10 digits, two blanks, a newline
    for (int64_t p0 = 0; ok && p0 < npix; p0 += chunk) {   // row-major, j outer
        char* o = buf.data();
        const int64_t p1 = std::min(npix, p0 + chunk);
        for (int64_t i = p0; i < p1; i++) {
            for (int k = 0; k < 3; k++) {
                const double c = real_type == CR_REAL_F64 ? ((const double*)rgb)[3 * i + k] : (double)((const float*)rgb)[3 * i + k];
                const uint32_t v = display_byte(c);
                if (v < 256u) { const Dec& d = table[v]; memcpy(o, d.s, 3); o += d.n; }
                else o += snprintf(o, 11, "%u", v);   // a channel above 1: not a Color the reference could hold, printed as `as u32` would
                *o++ = k == 2 ? '\n' : ' ';
            }
        }
        ok = fwrite(buf.data(), 1, (size_t)(o - buf.data()), f) == (size_t)(o - buf.data());
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? CR_OK : CR_ERR_IO;
}

int32_t cr_write_ppm_binary(const char* path, const void* rgb, int32_t real_type, int32_t w, int32_t hgt) {
    if (!path || !rgb || w < 1 || hgt < 1 || (real_type != CR_REAL_F32 && real_type != CR_REAL_F64)) return CR_ERR_INVALID_ARG;
    std::vector<uint8_t> bytes((size_t)w * hgt * 3);
    if (cr_quantize_rgb8(rgb, real_type, (int64_t)w * hgt, bytes.data()) != CR_OK) return CR_ERR_INVALID_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return CR_ERR_IO;
    bool ok = fprintf(f, "P6\n%d %d\n255\n", w, hgt) > 0 && fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = (fclose(f) == 0) && ok;
    return ok ? CR_OK : CR_ERR_IO;
}

int32_t cr_write_png(const char* path, const void* rgb, int32_t real_type, int32_t w, int32_t hgt) {
    if (!path || !rgb || w < 1 || hgt < 1 || (real_type != CR_REAL_F32 && real_type != CR_REAL_F64)) return CR_ERR_INVALID_ARG;
    const size_t row = (size_t)w * 3;
    std::vector<uint8_t> raw((row + 1) * hgt);   // filter byte 0 (None) + RGB8 per scanline
    {
        std::vector<uint8_t> bytes(row * hgt);
        if (cr_quantize_rgb8(rgb, real_type, (int64_t)w * hgt, bytes.data()) != CR_OK) return CR_ERR_INVALID_ARG;
        for (int32_t y = 0; y < hgt; y++) { raw[(row + 1) * y] = 0; memcpy(&raw[(row + 1) * y + 1], &bytes[row * y], row); }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 1) != Z_OK) return CR_ERR_IO;   // level 1: output speed matters, not size
    FILE* f = fopen(path, "wb");
    if (!f) return CR_ERR_IO;
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    bool ok = true;
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8];
        be32(hdr, len); memcpy(hdr + 4, type, 4);
        uint32_t crc = (uint32_t)crc32(0L, (const Bytef*)type, 4);
        if (len) crc = (uint32_t)crc32(crc, data, len);
        uint8_t tail[4];
        be32(tail, crc);
        ok = ok && fwrite(hdr, 1, 8, f) == 8 && (len == 0 || fwrite(data, 1, len, f) == len) && fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    ok = fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(ihdr, (uint32_t)w); be32(ihdr + 4, (uint32_t)hgt);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8-bit, colour type 2 (RGB), no interlace
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    ok = (fclose(f) == 0) && ok;
    return ok ? CR_OK : CR_ERR_IO;
}

const char* cr_last_error(CrHandle* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

}   // extern "C"

// ================================================================== groups of devices (group.hpp)
#include "group.hpp"

extern "C" {

int32_t cr_group_shard(int32_t samples, int32_t member, int32_t n_members, int32_t* begin, int32_t* count) {
    if (samples < 0 || n_members < 1 || member < 0 || member >= n_members || !begin || !count) return CR_ERR_INVALID_ARG;
    const int64_t b = (int64_t)member * samples / n_members, e = (int64_t)(member + 1) * samples / n_members;
    *begin = (int32_t)b; *count = (int32_t)(e - b);
    return CR_OK;
}

int32_t cr_group_create(const int32_t* device_ids, int32_t n_devices, CrGroup** out) {
    if (!out) return gfail(nullptr, CR_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1) return gfail(nullptr, CR_ERR_INVALID_ARG, "device list is empty");
    // CRUCIBLE_GROUP_SAME_DEVICE=1 (tests on a one-GPU box): the members may share a device; their sums are then added
    // by a plain kernel instead of RCCL, which refuses two ranks on one device.  Everything else is the real path.
    const bool same_device = getenv("CRUCIBLE_GROUP_SAME_DEVICE") != nullptr;
    for (int i = 0; i < n_devices; i++) for (int j = 0; j < i; j++)
        if (device_ids[i] == device_ids[j] && !same_device) return gfail(nullptr, CR_ERR_INVALID_ARG, "a device appears twice in the list");
    DeviceGuard guard;
    CrGroup* g = new CrGroup();
    g->same_device_sum = same_device && n_devices > 1;
    g->world = n_devices; g->first = 0;
    g->members.assign((size_t)n_devices, nullptr);
    g->partial.resize((size_t)n_devices);
    g->status.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        int32_t rc = cr_create(device_ids[i], &g->members[(size_t)i]);
        if (rc != CR_OK) { g_group_create_error = g_create_error; group_free(g); return rc; }
    }
    const bool force = getenv("CRUCIBLE_GROUP_FORCE_RCCL") != nullptr;   // tests: exercise the collective on one device
    if ((n_devices > 1 || force) && !g->same_device_sum) {
        RcclApi& api = rccl_api();
        if (!api.lib) { g_group_create_error = api.error; group_free(g); return CR_ERR_UNSUPPORTED; }
        g->comms.assign((size_t)n_devices, nullptr);
        ncclResult_t r = api.CommInitAll(g->comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitAll: ") + api.GetErrorString(r); g->comms.clear(); group_free(g); return CR_ERR_HIP; }
    }
    (void)hipSetDevice(g->members[0]->device);
    if (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) { g_group_create_error = "hipEventCreate failed"; group_free(g); return CR_ERR_HIP; }
    *out = g;
    return CR_OK;
}

int32_t cr_group_unique_id(uint8_t id[CR_GROUP_ID_BYTES]) {
    static_assert(CR_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return gfail(nullptr, CR_ERR_INVALID_ARG, "id is null");
    RcclApi& api = rccl_api();
    if (!api.lib) return gfail(nullptr, CR_ERR_UNSUPPORTED, api.error);
    ncclUniqueId u;
    ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return gfail(nullptr, CR_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    memcpy(id, u.internal, CR_GROUP_ID_BYTES);
    return CR_OK;
}

int32_t cr_group_create_rank(int32_t device_id, int32_t rank, int32_t world_size, const uint8_t id[CR_GROUP_ID_BYTES], CrGroup** out) {
    if (!out) return gfail(nullptr, CR_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return gfail(nullptr, CR_ERR_INVALID_ARG, "rank outside [0, world_size)");
    if (world_size > 1 && !id) return gfail(nullptr, CR_ERR_INVALID_ARG, "id is null");
    DeviceGuard guard;
    CrGroup* g = new CrGroup();
    g->world = world_size; g->first = rank;
    g->members.assign(1, nullptr);
    g->partial.resize(1);
    g->status.resize(1);
    int32_t rc = cr_create(device_id, &g->members[0]);
    if (rc != CR_OK) { g_group_create_error = g_create_error; group_free(g); return rc; }
    const bool force = getenv("CRUCIBLE_GROUP_FORCE_RCCL") != nullptr && id;
    if (world_size > 1 || force) {
        RcclApi& api = rccl_api();
        if (!api.lib) { g_group_create_error = api.error; group_free(g); return CR_ERR_UNSUPPORTED; }
        ncclUniqueId u;
        memcpy(u.internal, id, CR_GROUP_ID_BYTES);
        g->comms.assign(1, nullptr);
        (void)hipSetDevice(device_id);
        ncclResult_t r = api.CommInitRank(&g->comms[0], world_size, u, rank);
        if (r != ncclSuccess) { g_group_create_error = std::string("ncclCommInitRank: ") + api.GetErrorString(r); g->comms.clear(); group_free(g); return CR_ERR_HIP; }
    }
    (void)hipSetDevice(device_id);
    if (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) { g_group_create_error = "hipEventCreate failed"; group_free(g); return CR_ERR_HIP; }
    *out = g;
    return CR_OK;
}

void cr_group_destroy(CrGroup* g) { DeviceGuard guard; group_free(g); }
int32_t cr_group_local_size(CrGroup* g) { return g ? (int32_t)g->members.size() : 0; }
int32_t cr_group_size(CrGroup* g) { return g ? g->world : 0; }
int32_t cr_group_rank(CrGroup* g) { return g ? g->first : -1; }
CrHandle* cr_group_handle(CrGroup* g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->members.size()) ? g->members[(size_t)i] : nullptr; }
const char* cr_group_last_error(CrGroup* g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

int32_t cr_group_upload_scene(CrGroup* g, const CrSceneDesc* scene) {
    if (!g) return CR_ERR_INVALID_ARG;
    DeviceGuard guard;
    for (CrHandle* h : g->members) {
        int32_t rc = cr_upload_scene(h, scene);
        if (rc != CR_OK) return gfail(g, rc, h->error);
    }
    return CR_OK;
}

// Failure handling.  Nothing is launched before every local member's arguments have been validated and its buffers exist,
// so bad arguments fail the same way on every rank.  A member that fails later (its render, an allocation) does NOT leave:
// with a collective, every member first takes part in a 4-byte ncclAllReduce(min) of "my render is fine" on the render's own
// stream, and only a unanimous 1 goes on to the ncclReduce -- otherwise every rank returns an error (its own, or
// CR_ERR_PEER) and the communicator is still consistent.  Only a failing collective call itself poisons the group
// (its communicators are aborted; every later call answers CR_ERR_PEER): peers inside that collective cannot be told.
// pre_rc / pre_msg: a failure this rank met before the call (cr_group_render_host's root-side buffer); it takes part
// in the agreement like a failed render, so the other ranks are not left waiting.
static int32_t group_render_impl(CrGroup* g, const CrCameraDesc* cam, const CrRenderParams* params, void* d_out, CrGroupStats* stats,
                                 int32_t pre_rc, const char* pre_msg) {
    if (!g) return CR_ERR_INVALID_ARG;
    if (g->poisoned) return gfail(g, CR_ERR_PEER, "an earlier collective of this group failed: destroy it and create a new one");
    if (!cam || !params) return gfail(g, CR_ERR_INVALID_ARG, "null camera or params");   // the same on every rank
    const bool root_here = g->first == 0;
    DeviceGuard guard;   // the caller's current device is the caller's again on every way out
    const bool collective = !g->comms.empty();
    const bool summed = g->world > 1 || collective;
    const int local = (int)g->members.size();
    std::vector<CrRenderParams> ps((size_t)local, *params);
    // 0. arguments and buffers, before anything is launched
    int32_t local_rc = CR_OK;
    std::string local_err;
    auto note = [&](int32_t rc, const std::string& msg) { if (local_rc == CR_OK && rc != CR_OK) { local_rc = rc; local_err = msg; } };
    if (pre_rc != CR_OK) note(pre_rc, pre_msg ? pre_msg : "");
    if (root_here && !d_out) note(CR_ERR_INVALID_ARG, "the root member needs an output buffer");
    for (int i = 0; i < local; i++) {
        CrHandle* h = g->members[(size_t)i];
        int32_t rc = validate_render(h, cam, params);
        if (rc != CR_OK) { note(rc, h->error); continue; }
        if (cr_group_shard(params->samples, g->first + i, g->world, &ps[(size_t)i].sample_begin, &ps[(size_t)i].sample_count) != CR_OK) note(CR_ERR_INVALID_ARG, "samples must be >= 0");
        ps[(size_t)i].output_sum = summed ? 1 : 0;   // one member, no collective: exactly cr_render_device
    }
    const size_t n = local_rc == CR_OK ? (size_t)cam->image_width * (size_t)cam->image_height * 3 : 0;
    const bool f64 = params->real_type == CR_REAL_F64;
    const size_t bytes = n * (f64 ? sizeof(double) : sizeof(float));
    if (summed) for (int i = 0; i < local && local_rc == CR_OK; i++) {
        CrHandle* h = g->members[(size_t)i];
        if (hipSetDevice(h->device) != hipSuccess || g->partial[(size_t)i].ensure(bytes) != hipSuccess ||
            (collective && g->status[(size_t)i].ensure(sizeof(int32_t)) != hipSuccess)) { (void)hipGetLastError(); note(CR_ERR_HIP, "cannot allocate a member's buffer of per-pixel sums"); }
    }
    // 1. every local member renders its shard, asynchronously on its own stream
    const char* fail_member = getenv("CRUCIBLE_GROUP_FAIL_MEMBER");   // tests: this member's render reports a failure after it was launched
    for (int i = 0; i < local && local_rc == CR_OK; i++) {
        CrHandle* h = g->members[(size_t)i];
        int32_t rc = cr_render_device(h, cam, &ps[(size_t)i], summed ? g->partial[(size_t)i].p : d_out, nullptr);
        if (rc == CR_OK && fail_member && atoi(fail_member) == g->first + i) rc = fail(h, CR_ERR_HIP, "render failure injected by CRUCIBLE_GROUP_FAIL_MEMBER");
        if (rc != CR_OK) note(rc, h->error);
    }
    // a collective call that fails leaves peers behind inside it: nothing more can be agreed on through these communicators
    auto poison = [&](const std::string& what) {
        g->poisoned = true;
        RcclApi& api = rccl_api();
        for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i]) { (void)hipSetDevice(g->members[i]->device); if (api.CommAbort) (void)api.CommAbort(g->comms[i]); g->comms[i] = nullptr; }
        for (CrHandle* h : g->members) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
        return gfail(g, CR_ERR_HIP, what);
    };
    // 2. do all members of the whole group stand?  (min over "1 = fine")
    bool all_fine = local_rc == CR_OK;
    if (collective) {
        RcclApi& api = rccl_api();
        const int32_t mine = local_rc == CR_OK ? 1 : 0;
        bool ok = true;
        for (int i = 0; i < local && ok; i++) {
            CrHandle* h = g->members[(size_t)i];
            ok = hipSetDevice(h->device) == hipSuccess && g->status[(size_t)i].ensure(sizeof(int32_t)) == hipSuccess &&
                 hipMemcpyAsync(g->status[(size_t)i].p, &mine, sizeof mine, hipMemcpyHostToDevice, h->stream) == hipSuccess;
        }
        if (!ok) return poison("cannot stage the group's status word");
        ncclResult_t r = api.GroupStart();
        for (int i = 0; i < local && r == ncclSuccess; i++) {
            CrHandle* h = g->members[(size_t)i];
            (void)hipSetDevice(h->device);
            r = api.AllReduce(g->status[(size_t)i].p, g->status[(size_t)i].p, 1, ncclInt32, ncclMin, g->comms[(size_t)i], h->stream);
        }
        if (r == ncclSuccess) r = api.GroupEnd(); else (void)api.GroupEnd();
        if (r != ncclSuccess) return poison(std::string("ncclAllReduce of the status word: ") + api.GetErrorString(r));
        int32_t agreed = 1;
        for (int i = 0; i < local; i++) {
            CrHandle* h = g->members[(size_t)i];
            int32_t v = 0;
            if (hipSetDevice(h->device) != hipSuccess || hipMemcpyAsync(&v, g->status[(size_t)i].p, sizeof v, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) return poison("cannot read the group's status word");
            agreed = std::min(agreed, v);
        }
        all_fine = agreed == 1;
    }
    if (!all_fine) {   // every rank is here: wait for what was launched and report
        for (CrHandle* h : g->members) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
        if (local_rc != CR_OK) return gfail(g, local_rc, local_err);
        return gfail(g, CR_ERR_PEER, "another member of the group failed its render; nothing was reduced");
    }
    // 3. one reduce of the sums to the root, then the divide there
    if (summed) {
        CrHandle* root = g->members[0];
        if (root_here) { GHIP_TRY(g, hipSetDevice(root->device)); GHIP_TRY(g, hipEventRecord(g->ev0, root->stream)); }
        if (g->same_device_sum) {   // every member is on the root's device: wait for their renders, then add in member order
            for (int i = 1; i < local; i++) GHIP_TRY(g, hipStreamSynchronize(g->members[(size_t)i]->stream));
            const unsigned grid = (unsigned)((n + 255) / 256);
            for (int i = 1; i < local; i++) {
                if (f64) hipLaunchKernelGGL((group_add_kernel<double>), dim3(grid), dim3(256), 0, root->stream, (double*)g->partial[0].p, (const double*)g->partial[(size_t)i].p, n);
                else hipLaunchKernelGGL((group_add_kernel<float>), dim3(grid), dim3(256), 0, root->stream, (float*)g->partial[0].p, (const float*)g->partial[(size_t)i].p, n);
            }
            GHIP_TRY(g, hipGetLastError());
        }
        if (collective) {
            RcclApi& api = rccl_api();
            const ncclDataType_t dt = f64 ? ncclDouble : ncclFloat;
            ncclResult_t r = api.GroupStart();
            for (int i = 0; i < local && r == ncclSuccess; i++) {
                CrHandle* h = g->members[(size_t)i];
                (void)hipSetDevice(h->device);
                void* buf = g->partial[(size_t)i].p;   // in place on the root
                r = api.Reduce(buf, buf, n, dt, ncclSum, 0, g->comms[(size_t)i], h->stream);
            }
            if (r == ncclSuccess) r = api.GroupEnd(); else (void)api.GroupEnd();
            if (r != ncclSuccess) return poison(std::string("ncclReduce: ") + api.GetErrorString(r));
        }
        if (root_here) {
            GHIP_TRY(g, hipSetDevice(root->device));
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (f64) hipLaunchKernelGGL((group_mean_kernel<double>), dim3(grid), dim3(256), 0, root->stream, (const double*)g->partial[0].p, (double*)d_out, n, (double)params->samples);
            else hipLaunchKernelGGL((group_mean_kernel<float>), dim3(grid), dim3(256), 0, root->stream, (const float*)g->partial[0].p, (float*)d_out, n, (float)params->samples);
            GHIP_TRY(g, hipGetLastError());
            GHIP_TRY(g, hipEventRecord(g->ev1, root->stream));
        }
    }
    // 4. wait for every local stream; a queue-pipeline wave that gave up leaves an incomplete image behind
    for (CrHandle* h : g->members) { GHIP_TRY(g, hipSetDevice(h->device)); GHIP_TRY(g, hipStreamSynchronize(h->stream)); }
    for (CrHandle* h : g->members) { int32_t rc = check_queue_abort(h); if (rc != CR_OK) return gfail(g, rc, h->error); }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->members = g->world; stats->used_rccl = collective ? 1 : 0;
        const int64_t npix = (int64_t)cam->image_width * cam->image_height;
        for (int i = 0; i < local; i++) {
            CrStats s;
            int32_t rc = member_stats(g->members[(size_t)i], npix * ps[(size_t)i].sample_count, &s);
            if (rc != CR_OK) return gfail(g, rc, g->members[(size_t)i]->error);
            stats->render.samples += s.samples; stats->render.segments += s.segments; stats->render.node_tests += s.node_tests;
            stats->render.prim_tests += s.prim_tests; stats->render.texel_fetches += s.texel_fetches;
            stats->render.kernel_ms = std::max(stats->render.kernel_ms, s.kernel_ms);
            stats->render.upload_ms = std::max(stats->render.upload_ms, s.upload_ms);
        }
        if (root_here && summed) {
            float ms = 0;
            GHIP_TRY(g, hipSetDevice(g->members[0]->device));
            GHIP_TRY(g, hipEventElapsedTime(&ms, g->ev0, g->ev1));
            stats->reduce_ms = ms;
        }
    }
    return CR_OK;
}

int32_t cr_group_render(CrGroup* g, const CrCameraDesc* cam, const CrRenderParams* params, void* d_out, CrGroupStats* stats) {
    return group_render_impl(g, cam, params, d_out, stats, CR_OK, nullptr);
}

int32_t cr_group_render_host(CrGroup* g, const CrCameraDesc* cam, const CrRenderParams* params, void* h_out, CrGroupStats* stats) {
    if (!g) return CR_ERR_INVALID_ARG;
    if (!cam || !params) return gfail(g, CR_ERR_INVALID_ARG, "null camera or params");
    DeviceGuard guard;
    const bool root_here = g->first == 0;
    CrHandle* root = g->members[0];
    const bool sized = cam->image_width >= 1 && cam->image_height >= 1;
    const size_t n = sized ? (size_t)cam->image_width * (size_t)cam->image_height * 3 : 0;
    const size_t bytes = n * real_size(params->real_type);
    void* d_out = nullptr;
    // what only the root can get wrong goes into the agreement step, so the other ranks are not left in the collective
    int32_t pre_rc = CR_OK;
    const char* pre_msg = nullptr;
    if (root_here && !h_out) { pre_rc = CR_ERR_INVALID_ARG; pre_msg = "the root member needs an output buffer"; }
    else if (root_here && sized) {
        if (hipSetDevice(root->device) != hipSuccess || root->out_buf.ensure(bytes) != hipSuccess) { (void)hipGetLastError(); pre_rc = CR_ERR_HIP; pre_msg = "cannot allocate the root's output buffer"; }
        d_out = root->out_buf.p;
    }
    CrGroupStats local;
    int32_t rc = group_render_impl(g, cam, params, d_out, stats ? stats : &local, pre_rc, pre_msg);
    if (rc != CR_OK || !root_here) return rc;
    GHIP_TRY(g, hipSetDevice(root->device));
    GHIP_TRY(g, hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, root->stream));
    GHIP_TRY(g, hipStreamSynchronize(root->stream));
    uint64_t bad = 0;   // Color::new asserts 0 <= c <= 1 on every mean (ray_casting.rs:172)
    for (size_t i = 0; i < n / 3; i++) {
        bool ok = true;
        for (int k = 0; k < 3; k++) {
            const double v = params->real_type == CR_REAL_F64 ? ((const double*)h_out)[3 * i + k] : (double)((const float*)h_out)[3 * i + k];
            ok = ok && (v >= 0.0 && v <= 1.0);
        }
        bad += ok ? 0 : 1;
    }
    if (stats) stats->render.nan_pixels = bad;
    if (bad) return gfail(g, CR_ERR_NAN, "a pixel mean is NaN or outside [0,1] (the reference panics in Color::new)");
    return CR_OK;
}

}   // extern "C"
