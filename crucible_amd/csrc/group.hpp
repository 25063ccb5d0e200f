// group.hpp -- several GPUs of one node behind the C ABI: samples-per-pixel sharding, one
// ncclReduce(sum) of the W*H*3 per-pixel sums over xGMI, the divide by `samples` on the root.  Included by capi.hip
// (one translation unit).  RCCL is loaded with dlopen on first use: the library has no link-time dependency on it,
// and a group of one member never touches it.
//
// What it replaces in the reference: the worker pool of src/camera/cpu_threading.rs:25-115 (thread_count OS threads
// pulling pixels behind one mutex) -- here the unit handed out is a range of sample indices per device, because
// samples are the independent unit (ray_casting.rs:82-105) and a per-pixel split would be load-imbalanced.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; every call goes through the dlopen'ed table below

#include <mutex>

namespace {

struct RcclApi {
    void* lib = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* override_path = getenv("CRUCIBLE_RCCL_LIB");
        const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            api.error = dlerror();
        }
        if (!api.lib) { api.error = "cannot load RCCL: " + api.error; return; }
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(api.lib, name); if (!p) { ok = false; api.error = std::string("RCCL symbol missing: ") + name; } return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    });
    return api;
}

thread_local std::string g_group_create_error;

// Test-only stand-in for the collective when several members share ONE device (CRUCIBLE_GROUP_SAME_DEVICE=1: RCCL
// refuses two ranks on a device): acc += part, member by member in index order.
template <typename real>
__global__ void __launch_bounds__(256) group_add_kernel(real* acc, const real* part, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] + part[i];
}

template <typename real>
__global__ void __launch_bounds__(256) group_mean_kernel(const real* sum, real* out, size_t n, real cnt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sum[i] / cnt;   // `/= sample_count as f64`, ray_casting.rs:168-170 -- sg_finalize_kernel's operation
}

}   // namespace

struct CrGroup {
    std::vector<CrHandle*> members;    // driven by this process
    std::vector<ncclComm_t> comms;     // one per local member; empty: no collective (one-member group)
    bool same_device_sum = false;      // tests: members share a device and their sums are added by group_add_kernel
    std::vector<DevBuf> partial;       // per local member: W*H*3 raw sums of its shard
    std::vector<DevBuf> status;        // per local member: the 4-byte "my render is fine" word of the agreement step
    bool poisoned = false;             // a collective call failed: peers may still be inside it, the communicators are gone
    int first = 0;                     // group-wide index of members[0]
    int world = 1;                     // members in the whole group
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // on the root member's stream: reduce + divide
    std::string error;
};

namespace {

// Restores the calling thread's current HIP device when a cr_group_* entry point returns (they visit every member's).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

int32_t gfail(CrGroup* g, int32_t code, const std::string& msg) {
    if (g) g->error = msg; else g_group_create_error = msg;
    return code;
}

#define NCCL_TRY(g, api, expr)                                                                                   \
    do {                                                                                                         \
        ncclResult_t _r = (expr);                                                                                \
        if (_r != ncclSuccess) return gfail(g, CR_ERR_HIP, std::string(#expr) + ": " + (api).GetErrorString(_r)); \
    } while (0)
#define GHIP_TRY(g, expr)                                                                                        \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) return gfail(g, CR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

void group_free(CrGroup* g) {
    if (!g) return;
    if (!g->comms.empty()) {
        RcclApi& api = rccl_api();
        for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i]) { (void)hipSetDevice(g->members[i]->device); (void)api.CommDestroy(g->comms[i]); }
    }
    for (size_t i = 0; i < g->members.size(); i++) {
        if (!g->members[i]) continue;
        (void)hipSetDevice(g->members[i]->device);
        if (i < g->partial.size()) g->partial[i].release();
        if (i < g->status.size()) g->status[i].release();
        if (i == 0) { if (g->ev0) (void)hipEventDestroy(g->ev0); if (g->ev1) (void)hipEventDestroy(g->ev1); }
        cr_destroy(g->members[i]);
    }
    delete g;
}

// Counters and kernel time of the member's last (asynchronous) render, read after its stream went idle.
int32_t member_stats(CrHandle* h, int64_t samples, CrStats* st) {
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    uint64_t c[4] = {0, 0, 0, 0};
    if (samples > 0) HIP_TRY(h, hipMemcpy(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost));
    memset(st, 0, sizeof *st);
    st->kernel_ms = ms; st->samples = (uint64_t)samples;
    st->segments = c[0]; st->node_tests = c[1]; st->prim_tests = c[2]; st->texel_fetches = c[3];
    st->upload_ms = h->upload_ms;
    return CR_OK;
}

}   // namespace
