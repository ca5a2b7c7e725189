// rnde_comm.hip -- the gradient collective behind the C ABI (include/rnde.h: rnde_comm_*).
//
// The reference is single-process; data parallelism over the 8 GPUs of a node puts exactly ONE collective per
// training step between Tracker.gradient and update_parameters! (reference experiments/mnist_node.jl:229-233,
// src/utils.jl:149-156): a sum of the flat gradient (166,418 fp32 for MNIST-NODE).  A Julia (or any non-Python) caller gets it
// from librnde.so itself: RCCL over xGMI, on the caller's HIP stream, no torch anywhere.  RCCL is bound at run time
// (dlopen "librccl.so.1"): librnde.so has no link-time dependency on it, single-GPU users never load it, and inside a process
// that already holds an RCCL (PyTorch's) the same instance is reused.
#include "../../include/rnde.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>

namespace {

typedef struct { char internal[128]; } nccl_unique_id;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL that is ALREADY in the process wins (PyTorch bundles its own copy: two instances in one process would each keep their own
        // topology, proxy threads and IPC handles): first the loaded image under any of its names (RTLD_NOLOAD), then whatever exports
        // ncclAllReduce globally; only then is the library loaded by name.  rnde_comm_library() says which file was bound.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (a.lib) break;
        }
        if (!a.lib && dlsym(RTLD_DEFAULT, "ncclAllReduce")) a.lib = dlopen(nullptr, RTLD_NOW);
        if (!a.lib) {
            for (const char* name : names) {
                a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a.lib) break;
            }
        }
        if (!a.lib) { a.err = std::string("cannot load RCCL: ") + dlerror(); return; }
        auto sym = [&](const char* s) { void* p = dlsym(a.lib, s); if (!p && a.err.empty()) a.err = std::string("RCCL symbol missing: ") + s; return p; };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    });
    return a;
}

thread_local std::string g_comm_err;

}  // namespace

extern "C" const char* rnde_comm_library(void) {
    static std::string path;
    RcclApi& a = api();
    if (!a.AllReduce) return a.err.c_str();
    Dl_info info;
    path = (dladdr((void*)a.AllReduce, &info) && info.dli_fname) ? info.dli_fname : "(unknown)";
    return path.c_str();
}

namespace {

__global__ void rnde_scale_kernel(float* __restrict__ v, long long n, float s) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) v[i] *= s;
}

// ---- ranks that live in ONE process on ONE device (rnde_comm_create_local_group), each driven by its own host thread on its own
// stream.  No kernel ever waits for another stream's kernel (two streams of a process may share a hardware queue, where a spinning
// kernel would hold back the very kernel it is waiting for): the ranks meet on the HOST, the device side is ordered by events.
//   publish kernel (buf -> this rank's slot) ; record ev_pub ; host barrier "every rank has enqueued its publish of exchange k" ;
//   wait for the other ranks' ev_pub ; sum kernel (slots in rank order -> buf) ; record ev_done.
// Two slot sets alternate; before a slot is rewritten (exchange k + 2) the stream waits for every rank's ev_done of exchange k, which
// were recorded before those ranks could pass the host barrier of exchange k + 1.
constexpr int kLocalMaxCount = 8192;
constexpr int kLocalMaxWorld = 64;
struct LocalShared {
    float* data = nullptr;        // [2][world][kLocalMaxCount]
    int world = 0, device = 0;
    std::atomic<int> refs{0};
    std::mutex mu;
    std::condition_variable cv;
    unsigned published[kLocalMaxWorld] = {0};
    bool failed = false;
    hipEvent_t ev_pub[2][kLocalMaxWorld] = {{nullptr}}, ev_done[2][kLocalMaxWorld] = {{nullptr}};
};
__global__ __launch_bounds__(256) void rnde_local_publish_kernel(const float* __restrict__ buf, int n, float* __restrict__ mine) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) mine[i] = buf[i];
}
__global__ __launch_bounds__(256) void rnde_local_sum_kernel(float* __restrict__ buf, int n, const float* __restrict__ slots, int world) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float s = 0.f;
        for (int r = 0; r < world; ++r) s += slots[(size_t)r * kLocalMaxCount + i];
        buf[i] = s;
    }
}

}  // namespace

struct rnde_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    LocalShared* loc = nullptr;   // non-null: a rank of an in-process group
    unsigned seq = 0;
    std::string err;
};

extern "C" const char* rnde_comm_last_error(const rnde_comm* c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

extern "C" rnde_status rnde_comm_unique_id(uint8_t id_out[RNDE_COMM_ID_BYTES]) {
    RcclApi& a = api();
    if (!a.err.empty()) { g_comm_err = a.err; return RNDE_ERR_HIP; }
    nccl_unique_id id;
    const int r = a.GetUniqueId(&id);
    if (r != kNcclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + a.GetErrorString(r); return RNDE_ERR_HIP; }
    static_assert(sizeof(id) == RNDE_COMM_ID_BYTES, "unique id size");
    std::memcpy(id_out, &id, sizeof(id));
    return RNDE_OK;
}

extern "C" rnde_status rnde_comm_create(const uint8_t id[RNDE_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, rnde_comm** out) {
    *out = nullptr;
    RcclApi& a = api();
    if (!a.err.empty()) { g_comm_err = a.err; return RNDE_ERR_HIP; }
    if (world < 1 || rank < 0 || rank >= world) { g_comm_err = "rank / world out of range"; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return RNDE_ERR_NO_DEVICE; }
    rnde_comm* c = new rnde_comm();
    c->rank = rank; c->world = world; c->device = device;
    nccl_unique_id uid;
    std::memcpy(&uid, id, sizeof(uid));
    const int r = a.CommInitRank(&c->comm, world, uid, rank);
    if (r != kNcclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + a.GetErrorString(r); delete c; return RNDE_ERR_HIP; }
    *out = c;
    return RNDE_OK;
}

extern "C" rnde_status rnde_comm_create_local_group(int32_t world, int32_t device, rnde_comm** out) {
    if (!out || world < 1 || world > kLocalMaxWorld) { g_comm_err = "world: 1..64"; return RNDE_ERR_BAD_ARG; }
    for (int r = 0; r < world; ++r) out[r] = nullptr;
    if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return RNDE_ERR_NO_DEVICE; }
    LocalShared* L = new LocalShared();
    L->world = world; L->device = device;
    bool ok = hipMalloc((void**)&L->data, (size_t)2 * world * kLocalMaxCount * 4) == hipSuccess;
    for (int sl = 0; sl < 2 && ok; ++sl)
        for (int r = 0; r < world && ok; ++r)
            ok = hipEventCreateWithFlags(&L->ev_pub[sl][r], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&L->ev_done[sl][r], hipEventDisableTiming) == hipSuccess;
    if (!ok) { g_comm_err = "device allocation failed"; if (L->data) (void)hipFree(L->data); delete L; return RNDE_ERR_HIP; }
    for (int r = 0; r < world; ++r) {
        rnde_comm* c = new rnde_comm();
        c->rank = r; c->world = world; c->device = device; c->loc = L;
        L->refs++;
        out[r] = c;
    }
    return RNDE_OK;
}

extern "C" void rnde_comm_destroy(rnde_comm* c) {
    if (!c) return;
    if (c->comm) (void)api().CommDestroy(c->comm);
    if (c->loc && --c->loc->refs == 0) {
        for (int sl = 0; sl < 2; ++sl)
            for (int r = 0; r < c->loc->world; ++r) { if (c->loc->ev_pub[sl][r]) (void)hipEventDestroy(c->loc->ev_pub[sl][r]); if (c->loc->ev_done[sl][r]) (void)hipEventDestroy(c->loc->ev_done[sl][r]); }
        (void)hipFree(c->loc->data);
        delete c->loc;
    }
    delete c;
}

extern "C" int32_t rnde_comm_world(const rnde_comm* c) { return c ? c->world : 0; }

// Did an all-reduce of this communicator give up waiting for a rank?  (In-process groups; RCCL reports its failures from the enqueue call.)
extern "C" rnde_status rnde_comm_health(rnde_comm* c) {
    if (!c || !c->loc) return RNDE_OK;
    std::lock_guard<std::mutex> g(c->loc->mu);
    if (c->loc->failed) { c->err = "an all-reduce gave up waiting for a rank (every rank must make the same calls, each from its own host thread)"; return RNDE_ERR_HIP; }
    return RNDE_OK;
}

// In-place sum over the ranks of `n` floats, then (mean != 0) a scale by 1 / world -- both on `stream`, asynchronous.
extern "C" rnde_status rnde_comm_allreduce(rnde_comm* c, float* buf_dev, int64_t n, int32_t mean, void* stream) {
    if (!c || !buf_dev || n < 0) return RNDE_ERR_BAD_ARG;
    if (n == 0) return RNDE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return RNDE_ERR_HIP; }
    if (c->loc) {
        LocalShared* L = c->loc;
        if (n > kLocalMaxCount) { c->err = "in-process group: at most 8192 floats per all-reduce"; return RNDE_ERR_BAD_ARG; }
        const unsigned seq = ++c->seq;
        const int slot = (int)(seq & 1u);
        const unsigned blocks = (unsigned)((n + 255) / 256);
        bool ok = true;
        if (seq > 2) for (int r = 0; r < L->world && ok; ++r) ok = hipStreamWaitEvent(s, L->ev_done[slot][r], 0) == hipSuccess;   // the slot's previous use has been read by everyone
        float* mine = L->data + ((size_t)slot * L->world + c->rank) * kLocalMaxCount;
        hipLaunchKernelGGL(rnde_local_publish_kernel, dim3(blocks), dim3(256), 0, s, buf_dev, (int)n, mine);
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(L->ev_pub[slot][c->rank], s) == hipSuccess;
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->published[c->rank] = seq;
            L->cv.notify_all();
            const bool all = L->cv.wait_for(lk, std::chrono::seconds(20), [&] {
                if (L->failed) return true;
                for (int r = 0; r < L->world; ++r) if (L->published[r] < seq) return false;
                return true;
            });
            if (!all || L->failed) { L->failed = true; L->cv.notify_all(); c->err = "in-process group: a rank did not reach the all-reduce (every rank must make the same calls, each from its own host thread)"; return RNDE_ERR_HIP; }
        }
        for (int r = 0; r < L->world && ok; ++r) if (r != c->rank) ok = hipStreamWaitEvent(s, L->ev_pub[slot][r], 0) == hipSuccess;
        hipLaunchKernelGGL(rnde_local_sum_kernel, dim3(blocks), dim3(256), 0, s, buf_dev, (int)n, L->data + (size_t)slot * L->world * kLocalMaxCount, L->world);
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(L->ev_done[slot][c->rank], s) == hipSuccess;
        if (!ok) { c->err = "in-process all-reduce: enqueue failed"; return RNDE_ERR_HIP; }
    } else {
        const int r = api().AllReduce(buf_dev, buf_dev, (size_t)n, kNcclFloat32, kNcclSum, c->comm, s);
        if (r != kNcclSuccess) { c->err = std::string("ncclAllReduce: ") + api().GetErrorString(r); return RNDE_ERR_HIP; }
    }
    if (mean && c->world > 1) {
        hipLaunchKernelGGL(rnde_scale_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, buf_dev, (long long)n, 1.0f / (float)c->world);
        if (hipGetLastError() != hipSuccess) { c->err = "scale kernel launch failed"; return RNDE_ERR_HIP; }
    }
    return RNDE_OK;
}
