// queue.hpp -- the wavefront megakernel: ONE persistent launch whose waves specialise and hand paths to each
// other through LDS queues.
//
//   workgroup = 1024 threads = 16 waves on one CU, owning 1024 path slots whose whole state lives in LDS
//   walker waves   pull slot ids from the ray queue (a lane that finishes its ray pushes the hit and refills in
//                  the same round: an LDS round trip, not a shading pass), run walk_round, push to the hit queue
//   shader waves   pull 64 hits at a time from the hit queue, so shading always runs on full waves: the rest
//                  of ray_color for each (shade / sky + unwind), sample and pixel bookkeeping, the next camera
//                  ray; push the slot back to the ray queue
//
// Each slot renders one pixel at a time, all of its samples in draw order (the reference's sequential sum), and
// every path performs exactly the operations the plain megakernel performs for it (same camera_ray / walk_round /
// shade code): results are bit-identical, only the scheduling differs.
//
// Queues are rings of 1024 tagged entries (tag = lap of the position, so a stale entry is never mistaken for a
// new one and no cell is ever reset); producers reserve positions with one wave-aggregated LDS atomic, consumers
// claim positions with a bounded CAS and spin (bounded) on the tag.  Every wait is bounded: on a timeout the
// workgroup raises `abort`, all its waves leave, and the host reports an error instead of hanging the GPU.
#pragma once
#include "pathtrace.hpp"

namespace cr {

constexpr uint32_t QK_SLOTS = 1024;
constexpr uint32_t QK_LOG = 10;
constexpr uint32_t QK_SPIN_LIMIT = 1u << 24;   // idle polls (~60 ns each) before a wave gives up: ~1 s, far beyond any legitimate wait
enum : int { QC_RQ_HEAD = 0, QC_RQ_TAIL = 1, QC_HQ_HEAD = 2, QC_HQ_TAIL = 3, QC_DONE = 4, QC_ABORT = 5, QC_WORDS = 16 };

// LDS bytes of the slot arrays + rings + control words (the scene part is added by the caller)
template <typename real> constexpr size_t queue_state_bytes() {
    return QK_SLOTS * (7 * sizeof(real) /*ray*/ + sizeof(real) + 4 /*hit*/ + 8 /*rng*/ + 12 /*pix sample depth*/ + 3 * sizeof(real) /*acc*/) +
           2 * QK_SLOTS * 4 + QC_WORDS * 4;
}

#if defined(__HIPCC__)

CR_D uint32_t q_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

// producers: one LDS atomic per wave reserves positions, then each lane publishes its tagged entry
CR_D void ring_push(uint32_t* ring, uint32_t* tail, bool pred, uint32_t slot, uint32_t lane) {
    const uint64_t m = __ballot(pred);
    if (!m) return;
    __threadfence_block();   // the slot's state is written before its id becomes visible
    uint32_t base = 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    if ((int)lane == leader) base = atomicAdd(tail, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (pred) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        __atomic_store_n(&ring[pos & (QK_SLOTS - 1)], ((pos >> QK_LOG) << QK_LOG) | slot, __ATOMIC_RELAXED);
    }
}

// consumers: lane 0 claims up to `want` positions that producers have at least reserved
CR_D uint32_t ring_claim(uint32_t* head, const uint32_t* tail, uint32_t want, uint32_t lane, uint32_t& base, uint32_t at_least = 1) {
    uint32_t take = 0, b = 0;
    if (lane == 0) {
        for (int tries = 0; tries < 4; tries++) {
            const uint32_t h = q_load(head), t = q_load(tail);
            const int32_t avail = (int32_t)(t - h);
            if (avail < (int32_t)at_least) break;
            const uint32_t k = want < (uint32_t)avail ? want : (uint32_t)avail;
            if (atomicCAS(head, h, h + k) == h) { take = k; b = h; break; }
        }
    }
    base = __shfl(b, 0);
    return __shfl(take, 0);
}

// wait (bounded) until the producer of position `pos` has published its entry
CR_D bool ring_read(const uint32_t* ring, uint32_t pos, uint32_t& slot, uint32_t* ctrl) {
    const uint32_t tag = pos >> QK_LOG;
    for (uint32_t spin = 0; spin < QK_SPIN_LIMIT; spin++) {
        const uint32_t e = q_load(&ring[pos & (QK_SLOTS - 1)]);
        if ((e >> QK_LOG) == tag) { slot = e & (QK_SLOTS - 1); __threadfence_block(); return true; }
        if (q_load(&ctrl[QC_ABORT])) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __atomic_store_n(&ctrl[QC_ABORT], 1u, __ATOMIC_RELAXED);
    return false;
}

template <typename real, int RES, bool ANIM>
__global__ void __launch_bounds__(1024) queue_kernel(const KernelArgs<real> A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- scene (as in pathtrace_kernel)
    const Entry<real>* lds_entries = nullptr;
    const Prim<real>* prims = A.prims;
    const Mat<real>* mats = A.mats;
    const Tex<real>* texs = A.texs;
    size_t off = 0;
    auto copy = [&](const void* src, size_t o, size_t bytes) {
        const uint32_t* s = (const uint32_t*)src;
        uint32_t* d = (uint32_t*)(smem + o);
        for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
    };
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    if (RES != RES_GLOBAL) {
        copy(A.entries, 0, (size_t)A.lds_entries * sizeof(Entry<real>));
        lds_entries = (const Entry<real>*)smem;
        off = r16((size_t)A.lds_entries * sizeof(Entry<real>));
        if (RES == RES_LDS) {
            size_t o1 = off, o2 = o1 + r16((size_t)A.n_prims * sizeof(Prim<real>)), o3 = o2 + r16((size_t)A.n_mats * sizeof(Mat<real>));
            copy(A.prims, o1, (size_t)A.n_prims * sizeof(Prim<real>));
            copy(A.mats, o2, (size_t)A.n_mats * sizeof(Mat<real>));
            copy(A.texs, o3, (size_t)A.n_texs * sizeof(Tex<real>));
            prims = (const Prim<real>*)(smem + o1); mats = (const Mat<real>*)(smem + o2); texs = (const Tex<real>*)(smem + o3);
            off = o3 + r16((size_t)A.n_texs * sizeof(Tex<real>));
        }
    }
    // ---- path slots, rings, control
    constexpr uint32_t S = QK_SLOTS;
    uint64_t* s_rng = (uint64_t*)(smem + off); off += S * 8;
    real* s_ray = (real*)(smem + off); off += 7 * S * sizeof(real);        // ox oy oz dx dy dz time
    real* s_acc = (real*)(smem + off); off += 3 * S * sizeof(real);
    real* s_hit_t = (real*)(smem + off); off += S * sizeof(real);
    int32_t* s_hit_prim = (int32_t*)(smem + off); off += S * 4;
    uint32_t* s_pix = (uint32_t*)(smem + off); off += S * 4;               // i | j << 16
    int32_t* s_sample = (int32_t*)(smem + off); off += S * 4;
    int32_t* s_depth = (int32_t*)(smem + off); off += S * 4;               // depth_left | stack_n << 16
    uint32_t* ring_r = (uint32_t*)(smem + off); off += S * 4;
    uint32_t* ring_h = (uint32_t*)(smem + off); off += S * 4;
    uint32_t* ctrl = (uint32_t*)(smem + off);
    ring_r[threadIdx.x] = 0xFFFFFFFFu; ring_h[threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x < QC_WORDS) ctrl[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t total_work = A.tiles_x * A.tiles_y * 64u;
    const uint32_t stack_stride = A.n_threads;
    const int32_t n_entries = A.n_entries;
    const CamConst<real>& cam = A.cam;
    uint32_t c_seg = 0, c_prim = 0, c_tex = 0;
    unsigned long long c_node = 0;

    // The shader side of one slot.  `have_hit`: the slot's ray came back from a walker (else the slot is new and
    // only needs a pixel).  Handles ray_color after the closest-hit query, average_samples' running sum, the
    // pixel hand-out and the next camera ray; leaves the slot queued for a walker or retired.
    auto shade_slots = [&](bool act, uint32_t slot, bool have_hit) {
        V3<real> ro = mk<real>(0, 0, 0), rd = mk<real>(0, 0, 1), col = mk<real>(0, 0, 0);
        real rtime = 0, acc_r = 0, acc_g = 0, acc_b = 0;
        uint64_t rng = 0;
        int32_t depth_left = 0, stack_n = 0, sample = 0;
        uint32_t pix_i = 0, pix_j = 0;
        bool finished = false, enqueue = false, need_pixel = act && !have_hit, need_sample = false, retired = false;
        if (act && have_hit) {
            ro = mk<real>(s_ray[slot], s_ray[S + slot], s_ray[2 * S + slot]);
            rd = mk<real>(s_ray[3 * S + slot], s_ray[4 * S + slot], s_ray[5 * S + slot]);
            if (ANIM) rtime = s_ray[6 * S + slot];
            rng = s_rng[slot];
            const int32_t dp = s_depth[slot];
            depth_left = dp & 0xFFFF; stack_n = dp >> 16;
            const uint32_t px = s_pix[slot];
            pix_i = px & 0xFFFFu; pix_j = px >> 16;
            sample = s_sample[slot];
            acc_r = s_acc[slot]; acc_g = s_acc[S + slot]; acc_b = s_acc[2 * S + slot];
            finished = shade<real, ANIM>(A, prims, mats, texs, ro, rd, rtime, rng, depth_left, stack_n, s_hit_t[slot], s_hit_prim[slot],
                                         stack_stride, blockIdx.x * S + slot, c_tex, col);
            if (!finished && depth_left == 0) { finished = true; col = mk<real>(0, 0, 0); }   // the next ray_color call returns black
            if (!finished) enqueue = true;
        }
        for (;;) {
            if (__ballot(finished || need_pixel || need_sample) == 0) break;
            if (finished) {   // average_samples (ray_casting.rs:154-173): samples are added in draw order
                acc_r += col.x; acc_g += col.y; acc_b += col.z;
                sample++; finished = false;
                if (sample >= A.sample_end) {
                    const size_t o = ((size_t)pix_j * (size_t)cam.W + pix_i) * 3;
                    if (A.output_sum) { A.out[o] = acc_r; A.out[o + 1] = acc_g; A.out[o + 2] = acc_b; }
                    else {
                        const real cnt = (real)A.samples_total;
                        A.out[o] = acc_r / cnt; A.out[o + 1] = acc_g / cnt; A.out[o + 2] = acc_b / cnt;
                    }
                    need_pixel = true;
                } else need_sample = true;
            }
            const uint64_t np = __ballot(need_pixel);
            if (np) {   // next pixel, 8x8-tile order, one wave-aggregated atomic
                uint32_t base = 0;
                const int leader = __ffsll((unsigned long long)np) - 1;
                if ((int)lane == leader) base = atomicAdd(A.work_counter, (uint32_t)__popcll(np));
                base = __shfl(base, leader);
                if (need_pixel) {
                    const uint32_t w = base + (uint32_t)__popcll(np & ((1ull << lane) - 1ull));
                    if (w >= total_work) { need_pixel = false; retired = true; }
                    else {
                        const uint32_t tile = w >> 6, in = w & 63u;
                        pix_i = (tile % A.tiles_x) * 8u + (in & 7u);
                        pix_j = (tile / A.tiles_x) * 8u + (in >> 3);
                        if (pix_i < (uint32_t)cam.W && pix_j < (uint32_t)cam.H) {
                            need_pixel = false; need_sample = true; sample = A.sample_begin; acc_r = acc_g = acc_b = 0;
                        }   // else: padding of an edge tile, ask again
                    }
                }
            }
            if (need_sample) {
                camera_ray<real, ANIM>(A, pix_i, pix_j, sample, rng, ro, rd, rtime);
                depth_left = A.max_depth; stack_n = 0; need_sample = false;
                if (depth_left == 0) { finished = true; col = mk<real>(0, 0, 0); }   // ray_color: depth == 0 -> black
                else enqueue = true;
            }
        }
        if (enqueue) {
            s_ray[slot] = ro.x; s_ray[S + slot] = ro.y; s_ray[2 * S + slot] = ro.z;
            s_ray[3 * S + slot] = rd.x; s_ray[4 * S + slot] = rd.y; s_ray[5 * S + slot] = rd.z;
            if (ANIM) s_ray[6 * S + slot] = rtime;
            s_rng[slot] = rng;
            s_depth[slot] = depth_left | (stack_n << 16);
            s_pix[slot] = pix_i | (pix_j << 16);
            s_sample[slot] = sample;
            s_acc[slot] = acc_r; s_acc[S + slot] = acc_g; s_acc[2 * S + slot] = acc_b;
            c_seg++;
        }
        const uint64_t rm = __ballot(retired);
        if (rm && lane == (uint32_t)(__ffsll((unsigned long long)rm) - 1)) atomicAdd(&ctrl[QC_DONE], (uint32_t)__popcll(rm));
        ring_push(ring_r, &ctrl[QC_RQ_TAIL], enqueue, slot, lane);
    };

    // ---- every slot starts by asking for a pixel
    shade_slots(true, threadIdx.x, false);
    __syncthreads();

    if (wave < A.queue_walk_waves) {
        // =================== walker
        bool has_ray = false;
        uint32_t slot = 0, idle = 0;
        V3<real> ro = mk<real>(0, 0, 0), rd = mk<real>(0, 0, 1);
        real rtime = 0;
        WalkState<real> ws;
        ws.inv = mk<real>(0, 0, 0); ws.dd = 0; ws.best_t = 0; ws.best = -1; ws.idx = 0; ws.exact_box = false; ws.pending = -1;
        for (;;) {
            const uint64_t need = __ballot(!has_ray);
            if (need) {
                uint32_t base = 0;
                const uint32_t take = ring_claim(&ctrl[QC_RQ_HEAD], &ctrl[QC_RQ_TAIL], (uint32_t)__popcll(need), lane, base);
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (!has_ray && rank < take && ring_read(ring_r, base + rank, slot, ctrl)) {
                    ro = mk<real>(s_ray[slot], s_ray[S + slot], s_ray[2 * S + slot]);
                    rd = mk<real>(s_ray[3 * S + slot], s_ray[4 * S + slot], s_ray[5 * S + slot]);
                    if (ANIM) rtime = s_ray[6 * S + slot];
                    walk_begin(ws, rd);
                    has_ray = true;
                }
            }
            if (__ballot(has_ray) == 0) {
                if (q_load(&ctrl[QC_DONE]) >= S || q_load(&ctrl[QC_ABORT])) break;
                if (++idle > QK_SPIN_LIMIT) { __atomic_store_n(&ctrl[QC_ABORT], 1u, __ATOMIC_RELAXED); break; }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            idle = 0;
            walk_round<real, RES, ANIM>(A, lds_entries, prims, ro, rd, rtime, ws, has_ray, A.walk_round_steps, c_node, c_prim);
            const bool fin = has_ray && ws.idx >= n_entries;
            if (fin) { s_hit_t[slot] = ws.best_t; s_hit_prim[slot] = ws.best; has_ray = false; }
            ring_push(ring_h, &ctrl[QC_HQ_TAIL], fin, slot, lane);
        }
    } else {
        // =================== shader
        // shade full waves: wait for a batch of hits, but never longer than a few polls (the tail has few paths left)
        uint32_t idle = 0, patience = 0;
        for (;;) {
            uint32_t base = 0;
            const uint32_t take = ring_claim(&ctrl[QC_HQ_HEAD], &ctrl[QC_HQ_TAIL], 64u, lane, base, patience < A.queue_patience ? A.queue_min_batch : 1u);
            patience = take ? 0 : patience + 1;
            if (take == 0) {
                if (q_load(&ctrl[QC_DONE]) >= S || q_load(&ctrl[QC_ABORT])) break;
                if (++idle > QK_SPIN_LIMIT) { __atomic_store_n(&ctrl[QC_ABORT], 1u, __ATOMIC_RELAXED); break; }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            idle = 0;
            uint32_t slot = 0;
            bool act = lane < take;
            if (act) act = ring_read(ring_h, base + lane, slot, ctrl);
            shade_slots(act, slot, true);
        }
    }

    // ---- flush work counters (and the abort flag) once per wave
    auto wave_sum = [&](unsigned long long v) -> unsigned long long {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        return v;
    };
    const unsigned long long s0 = wave_sum(c_seg), s1 = wave_sum(c_node), s2 = wave_sum(c_prim), s3 = wave_sum(c_tex);
    if (lane == 0) {
        atomicAdd((unsigned long long*)&A.counters[0], s0);
        atomicAdd((unsigned long long*)&A.counters[1], s1);
        atomicAdd((unsigned long long*)&A.counters[2], s2);
        atomicAdd((unsigned long long*)&A.counters[3], s3);
        if (q_load(&ctrl[QC_ABORT])) atomicAdd((unsigned long long*)&A.counters[4], 1ull);
    }
}

#endif   // __HIPCC__
}   // namespace cr
