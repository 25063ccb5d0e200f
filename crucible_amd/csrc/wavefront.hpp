// wavefront.hpp -- the same path tracer as pathtrace.hpp's megakernel, scheduled as a wavefront
// pipeline so that the BVH walk never waits for shading:
//
//   logic  (one thread per path slot)   shade the slot's traced ray or, when its path ended, write the
//                                       sample's colour, pull the next (pixel, sample) job and generate
//                                       its camera ray; append the slot to the ray queue
//   extend (persistent, LDS scene)      walk the BVH for queued rays; a lane whose ray is done stores
//                                       the hit and immediately pulls the next queued ray (ballot +
//                                       one aggregated atomic), so lanes stay busy
//   finalize (one thread per pixel)     add the batch's sample colours to the pixel sum IN SAMPLE ORDER
//
// Path state lives in HBM as SoA arrays indexed by slot (ray 7 reals, rng, depth/stack counters, hit);
// with 2M slots it is ~130 MB and stays in the 256 MB Infinity Cache.  Every path performs exactly the
// operations the megakernel performs for it (same camera_ray / walk / shade code, same RNG keys, same
// attenuation stack order, same sequential per-pixel sum), so results are bit-identical; only the
// scheduling differs.
#pragma once
#include "pathtrace.hpp"

namespace cr {

constexpr uint32_t WF_IDLE = 0xFFFFFFFFu;
constexpr int32_t WF_PENDING = -3;      // hit_prim value of a slot whose ray waits for extend
constexpr uint32_t WF_CHUNK = 256;      // jobs / slots claimed per global atomic (one word sustains only ~88 atomics/us)
constexpr uint32_t WF_SET = 64;         // words per control set: [0..31] ray-count shards, [32] slot cursor
constexpr uint32_t WF_JOB_CURSOR = 128;

template <typename real> struct WfArgs {
    KernelArgs<real> k;          // scene / camera / render parameters; att_stack has stride n_slots
    uint32_t n_slots;
    uint32_t total_work;         // tiles_x * tiles_y * 64 (tile-ordered pixel slots incl. edge padding)
    uint32_t n_jobs;             // jobs of this batch: batch_samples * total_work, job = s_local * total_work + w
    int32_t batch_begin;         // first sample index of the batch
    int32_t batch_samples;
    uint32_t ctrl_set;           // which of the two {queue count, queue head} pairs this iteration uses
    uint32_t* job;               // [n_slots] job id or WF_IDLE
    uint64_t* rng;               // [n_slots]
    real* ray;                   // [7][n_slots]: ox oy oz dx dy dz time
    int32_t* depth;              // [n_slots] depth_left | stack_n << 16
    real* hit_t;                 // [n_slots]
    int32_t* hit_prim;           // [n_slots]
    uint32_t* job_chunk;         // [n_slots/64][2] each logic wave's private job range {next, end}
    uint32_t* ctrl;              // two sets of {32 sharded ray counts, slot cursor} at [0..63] / [64..127]; [128] job cursor
    uint32_t* ring_slot;         // host-mapped word: extend reports the queue length it saw
    real* sample_rgb;            // [batch_samples][W*H][3]
    real* acc;                   // [W*H][3] running per-pixel sums
    int32_t last_batch;
};

#if defined(__HIPCC__)

template <typename real, bool ANIM>
__global__ void __launch_bounds__(256) wf_logic_kernel(const WfArgs<real> W) {
    const KernelArgs<real>& A = W.k;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = slot < W.n_slots;
    const uint32_t N = W.n_slots;
    uint32_t* my_set = &W.ctrl[WF_SET * W.ctrl_set];
    if (slot < WF_SET) W.ctrl[WF_SET * (W.ctrl_set ^ 1u) + slot] = 0;   // reset the other set for the next iteration

    V3<real> ro = mk<real>(0, 0, 0), rd = mk<real>(0, 0, 1);
    real rtime = 0;
    uint64_t rng = 0;
    int32_t depth_left = 0, stack_n = 0;
    uint32_t c_seg = 0, c_tex = 0;
    bool need_job = false, enqueue = false;
    uint32_t jb = WF_IDLE;
    if (live) jb = W.job[slot];
    if (live && jb != WF_IDLE) {
        // this slot's ray came back from extend: the rest of ray_color for it
        ro = mk<real>(W.ray[slot], W.ray[N + slot], W.ray[2 * N + slot]);
        rd = mk<real>(W.ray[3 * (size_t)N + slot], W.ray[4 * (size_t)N + slot], W.ray[5 * (size_t)N + slot]);
        if (ANIM) rtime = W.ray[6 * (size_t)N + slot];
        rng = W.rng[slot];
        int32_t dp = W.depth[slot];
        depth_left = dp & 0xFFFF; stack_n = dp >> 16;
        V3<real> col = mk<real>(0, 0, 0);
        bool finished = shade<real, ANIM>(A, A.prims, A.mats, A.texs, ro, rd, rtime, rng, depth_left, stack_n, W.hit_t[slot], W.hit_prim[slot],
                                          N, slot, c_tex, col);
        if (!finished && depth_left == 0) { finished = true; col = mk<real>(0, 0, 0); }   // the next ray_color call returns black: product is 0
        if (finished) {
            uint32_t s_local = jb / W.total_work, w = jb % W.total_work;
            uint32_t tile = w >> 6, in = w & 63u;
            uint32_t pi = (tile % A.tiles_x) * 8u + (in & 7u), pj = (tile / A.tiles_x) * 8u + (in >> 3);
            size_t o = ((size_t)s_local * ((size_t)A.cam.W * A.cam.H) + ((size_t)pj * A.cam.W + pi)) * 3;
            W.sample_rgb[o] = col.x; W.sample_rgb[o + 1] = col.y; W.sample_rgb[o + 2] = col.z;
            need_job = true;
        } else enqueue = true;
    } else if (live) need_job = true;

    // pull (pixel, sample) jobs: sample-major, pixels in 8x8-tile order, so a wave's new rays are neighbours.
    // Each wave owns a private range of job ids and claims the next WF_CHUNK with one atomic when it runs dry.
    const uint32_t wid = slot >> 6;
    uint32_t cn = 0, ce = 0;
    if (live) { cn = W.job_chunk[2 * wid]; ce = W.job_chunk[2 * wid + 1]; }
    cn = __shfl(cn, 0); ce = __shfl(ce, 0);
    bool exhausted = (cn == WF_IDLE);
    for (;;) {
        uint64_t need = __ballot(need_job);
        if (!need) break;
        if (cn >= ce) {
            if (exhausted) { if (need_job) { jb = WF_IDLE; need_job = false; } break; }
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&W.ctrl[WF_JOB_CURSOR], WF_CHUNK);
            base = __shfl(base, 0);
            if (base >= W.n_jobs) { exhausted = true; cn = ce = WF_IDLE; continue; }
            cn = base; ce = base + WF_CHUNK < W.n_jobs ? base + WF_CHUNK : W.n_jobs;
        }
        const uint32_t avail = ce - cn, rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
        if (need_job && rank < avail) {
            uint32_t j = cn + rank;
            uint32_t s_local = j / W.total_work, w = j % W.total_work;
            uint32_t tile = w >> 6, in = w & 63u;
            uint32_t pi = (tile % A.tiles_x) * 8u + (in & 7u), pj = (tile / A.tiles_x) * 8u + (in >> 3);
            if (pi < (uint32_t)A.cam.W && pj < (uint32_t)A.cam.H) {
                if (A.max_depth == 0) {   // ray_color(depth 0) is black without tracing
                    size_t o = ((size_t)s_local * ((size_t)A.cam.W * A.cam.H) + ((size_t)pj * A.cam.W + pi)) * 3;
                    W.sample_rgb[o] = 0; W.sample_rgb[o + 1] = 0; W.sample_rgb[o + 2] = 0;
                } else {
                    camera_ray<real, ANIM>(A, pi, pj, W.batch_begin + (int32_t)s_local, rng, ro, rd, rtime);
                    depth_left = A.max_depth; stack_n = 0;
                    jb = j; need_job = false; enqueue = true;
                }
            }   // else: padding of an edge tile, ask again
        }
        const uint32_t want = (uint32_t)__popcll(need);
        cn += want < avail ? want : avail;
    }
    if (live && lane == 0) { W.job_chunk[2 * wid] = cn; W.job_chunk[2 * wid + 1] = ce; }
    if (live) W.job[slot] = jb;
    if (enqueue) {
        W.ray[slot] = ro.x; W.ray[N + slot] = ro.y; W.ray[2 * (size_t)N + slot] = ro.z;
        W.ray[3 * (size_t)N + slot] = rd.x; W.ray[4 * (size_t)N + slot] = rd.y; W.ray[5 * (size_t)N + slot] = rd.z;
        if (ANIM) W.ray[6 * (size_t)N + slot] = rtime;
        W.rng[slot] = rng;
        W.depth[slot] = depth_left | (stack_n << 16);
        W.hit_prim[slot] = WF_PENDING;
        c_seg = 1;
    }
    // rays handed to extend this iteration (sharded: the host only needs "any?")
    uint64_t em = __ballot(enqueue);
    if (em && lane == 0) atomicAdd(&my_set[blockIdx.x & 31u], (uint32_t)__popcll(em));
    unsigned long long s0 = c_seg, s3 = c_tex;
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s3 += __shfl_down(s3, off); }
    if (lane == 0 && (s0 | s3)) {
        if (s0) atomicAdd((unsigned long long*)&A.counters[0], s0);
        if (s3) atomicAdd((unsigned long long*)&A.counters[3], s3);
    }
}

template <typename real, int RES, bool ANIM>
__global__ void __launch_bounds__(MaxBlock<real>::value) wf_extend_kernel(const WfArgs<real> W) {
    const KernelArgs<real>& A = W.k;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Entry<real>* lds_entries = nullptr;
    const Prim<real>* prims = A.prims;
    uint32_t* my_set = &W.ctrl[WF_SET * W.ctrl_set];
    uint32_t q_count = 0;
    for (int i = 0; i < 32; i++) q_count += my_set[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *W.ring_slot = q_count;
    if (q_count == 0) return;
    if (RES != RES_GLOBAL) {   // entries (all or the top levels) and, when they fit, the primitives: shading does not run here
        auto copy = [&](const void* src, size_t off, size_t bytes) {
            const uint32_t* s = (const uint32_t*)src;
            uint32_t* d = (uint32_t*)(smem + off);
            for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
        };
        copy(A.entries, 0, (size_t)A.lds_entries * sizeof(Entry<real>));
        lds_entries = (const Entry<real>*)smem;
        if (RES == RES_LDS) {
            size_t o1 = (((size_t)A.n_entries * sizeof(Entry<real>) + 15) & ~(size_t)15);
            copy(A.prims, o1, (size_t)A.n_prims * sizeof(Prim<real>));
            prims = (const Prim<real>*)(smem + o1);
        }
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t N = W.n_slots;
    uint32_t* cursor = &my_set[32];
    uint32_t cn = 0, ce = 0;   // this wave's private range of slots to scan
    bool drained = false;
    const int32_t n_entries = A.n_entries;

    bool has_ray = false;
    uint32_t slot = 0;
    V3<real> ro = mk<real>(0, 0, 0), rd = mk<real>(0, 0, 1);
    real rtime = 0;
    WalkState<real> ws;
    ws.inv = mk<real>(0, 0, 0); ws.dd = 0; ws.best_t = 0; ws.best = -1; ws.idx = 0; ws.exact_box = false; ws.pending = -1;
    uint32_t c_prim = 0;
    unsigned long long c_node = 0;

    for (;;) {
        // ---- refill: lanes without a ray scan the wave's private slot range for pending rays; the range is
        // topped up WF_CHUNK slots at a time with one atomic
        for (;;) {
            uint64_t need = __ballot(!has_ray);
            if (!need || (drained && cn >= ce)) break;
            if (cn >= ce) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, WF_CHUNK);
                base = __shfl(base, 0);
                if (base >= N) { drained = true; break; }
                cn = base; ce = base + WF_CHUNK < N ? base + WF_CHUNK : N;
            }
            const uint32_t avail = ce - cn, rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            if (!has_ray && rank < avail) {
                slot = cn + rank;
                if (W.hit_prim[slot] == WF_PENDING) {
                    ro = mk<real>(W.ray[slot], W.ray[N + slot], W.ray[2 * (size_t)N + slot]);
                    rd = mk<real>(W.ray[3 * (size_t)N + slot], W.ray[4 * (size_t)N + slot], W.ray[5 * (size_t)N + slot]);
                    if (ANIM) rtime = W.ray[6 * (size_t)N + slot];
                    walk_begin(ws, rd);
                    has_ray = true;
                }
            }
            const uint32_t want = (uint32_t)__popcll(need);
            cn += want < avail ? want : avail;
        }
        if (__ballot(has_ray) == 0) break;
        walk_round<real, RES, ANIM>(A, lds_entries, prims, ro, rd, rtime, ws, has_ray, 0u, c_node, c_prim);
        // ---- a ray with no wrappers left is done: hand its hit to the logic kernel
        if (has_ray && ws.idx >= n_entries) {
            W.hit_t[slot] = ws.best_t;
            W.hit_prim[slot] = ws.best;
            has_ray = false;
        }
    }
    unsigned long long s1 = c_node, s2 = c_prim;
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.counters[1], s1);
        atomicAdd((unsigned long long*)&A.counters[2], s2);
    }
}

// average_samples' running sum (ray_casting.rs:161-165): samples of the batch are added in sample order.
template <typename real>
__global__ void __launch_bounds__(256) wf_finalize_kernel(const WfArgs<real> W) {
    const KernelArgs<real>& A = W.k;
    const size_t npix = (size_t)A.cam.W * A.cam.H;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    real r = W.acc[3 * p], g = W.acc[3 * p + 1], b = W.acc[3 * p + 2];
    for (int32_t s = 0; s < W.batch_samples; s++) {
        const real* c = W.sample_rgb + ((size_t)s * npix + p) * 3;
        r += c[0]; g += c[1]; b += c[2];
    }
    if (W.last_batch) {
        if (A.output_sum) { A.out[3 * p] = r; A.out[3 * p + 1] = g; A.out[3 * p + 2] = b; }
        else {
            real cnt = (real)A.samples_total;
            A.out[3 * p] = r / cnt; A.out[3 * p + 1] = g / cnt; A.out[3 * p + 2] = b / cnt;
        }
    } else { W.acc[3 * p] = r; W.acc[3 * p + 1] = g; W.acc[3 * p + 2] = b; }
}

#endif   // __HIPCC__
}   // namespace cr
