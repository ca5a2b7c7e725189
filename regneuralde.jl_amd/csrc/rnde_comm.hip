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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

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
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;   // (one-shot path: ships the window handles)
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    std::string path;      // file the symbols were bound from (set once, inside the call_once below: rnde_comm_library() returns it unchanged)
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
        a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
        Dl_info info;
        a.path = (a.AllReduce && dladdr((void*)a.AllReduce, &info) && info.dli_fname) ? info.dli_fname : "(unknown)";
    });
    return a;
}

thread_local std::string g_comm_err;

}  // namespace

extern "C" const char* rnde_comm_library(void) {      // (immutable after the first call of api(): safe from any number of host threads)
    RcclApi& a = api();
    return a.AllReduce ? a.path.c_str() : a.err.c_str();
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


// ---- one-shot all-reduce over peer-mapped windows (the gradient message is 0.67 MB -- latency bound, so every rank
// reads the N - 1 peers' copies directly over its 7 xGMI links in ONE kernel instead of walking a ring).  Every rank owns a WINDOW in its
// own HBM ([flags | 2 data slots]), exported with hipIpcGetMemHandle and mapped by every peer.  One kernel per all-reduce and rank:
//   a. block w copies its chunk of `buf` into this rank's slot (seq & 1), fences to system scope, and stores `seq` into flag
//      (slot, this rank, w) of EVERY peer's window (remote 4-byte stores: the peers poll local memory);
//   b. block w waits until flags (slot, p, w) of its own window carry `seq` for every peer p, acquires at system scope;
//   c. block w sums its chunk over the ranks' slots IN RANK ORDER (every rank computes the same bits) and writes `buf`.
// Block w touches chunk w on every rank, so per-block flags are the whole synchronisation -- no grid-wide meeting.  Two slots alternate:
// a rank rewrites a slot at seq + 2, which it reaches only after every peer has signalled seq + 1, i.e. finished reading seq.
// A wait gives up after kPeerTimeoutTicks (20 s) of the 100 MHz wall clock and raises the window's `fail` word (rnde_comm_health) rather than
// hanging the queue.  Opt-in (RNDE_ONESHOT=1 with rnde_comm_create, or rnde_comm_create_peers with handles the caller exchanged):
// RCCL stays the default until the path has been measured on N > 1 GPUs.
constexpr int kPeerMaxWorld = 16;
constexpr int kPeerBlocks = 64;                       // workgroups per all-reduce at most (0.67 MB: 41.6 K float4 = 2.5 per thread and peer)
constexpr long long kPeerCap = 256 * 1024;            // floats per slot (1 MB); larger buffers go in pieces (or to RCCL when the communicator has one)
constexpr size_t kPeerFlagBytes = 8192;               // [2 slots][kPeerMaxWorld][kPeerBlocks] unsigned = 8 KB, then one page of padding
constexpr size_t kPeerDataOffset = 16384;
constexpr size_t kPeerWindowBytes = kPeerDataOffset + 2 * (size_t)kPeerCap * 4;
constexpr long long kPeerTimeoutTicks = 20LL * 100000000;  // 20 s (start-up skew between ranks is the long case; RCCL would wait for ever)

struct PeerView {
    float* data[kPeerMaxWorld];        // rank r's data slots as mapped in THIS process (own rank: the local pointer)
    unsigned* flags[kPeerMaxWorld];    // rank r's flag array
    unsigned* fail;                    // local: raised when a wait gave up
    unsigned* fail_host;               // the same in pinned host memory (mapped): the next enqueue sees it without a device round trip
    int world, rank;
};

template <bool ALIGNED>
__global__ __launch_bounds__(256) void rnde_peer_allreduce_kernel(PeerView V, float* __restrict__ buf, long long n, long long chunk, unsigned seq, float scale, long long timeout_ticks) {
    const int w = blockIdx.x, tid = threadIdx.x, slot = (int)(seq & 1u);
    const long long lo = (long long)w * chunk, hi = lo + chunk < n ? lo + chunk : n;     // chunk is a multiple of 4: lo is 16-byte aligned in the windows
    const long long len = hi - lo, nv = ALIGNED ? len / 4 : 0;
    float* mine = V.data[V.rank] + (size_t)slot * kPeerCap;
    // a. publish
    if (ALIGNED) for (long long i = tid; i < nv; i += 256) ((float4*)(mine + lo))[i] = ((const float4*)(buf + lo))[i];
    for (long long i = lo + 4 * nv + tid; i < hi; i += 256) mine[i] = buf[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the stores above are in memory before the flags below
    __syncthreads();
    if (tid < V.world && tid != V.rank)
        __hip_atomic_store(V.flags[tid] + ((size_t)slot * kPeerMaxWorld + V.rank) * kPeerBlocks + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // b. wait for the peers' chunk w
    if (tid < V.world && tid != V.rank) {
        const unsigned* f = V.flags[V.rank] + ((size_t)slot * kPeerMaxWorld + tid) * kPeerBlocks + w;
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (wall_clock64() - t0 > timeout_ticks) { atomicExch(V.fail, 1u); __hip_atomic_store(V.fail_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // a wait gave up (now or in an earlier all-reduce of this communicator: the flag protocol is out of step from then on): the chunk is
    // poisoned instead of summed from whatever the slots hold -- an update computed from it is NaN, not a silently un-reduced gradient
    if (__hip_atomic_load(V.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        for (long long i = lo + tid; i < hi; i += 256) buf[i] = __builtin_nanf("");
        return;
    }
    // c. sum in rank order
    const float* src[kPeerMaxWorld];
#pragma unroll
    for (int r = 0; r < kPeerMaxWorld; ++r) src[r] = r < V.world ? V.data[r] + (size_t)slot * kPeerCap : nullptr;
    if (ALIGNED) for (long long i = tid; i < nv; i += 256) {
        float4 v[kPeerMaxWorld];
#pragma unroll
        for (int r = 0; r < kPeerMaxWorld; ++r) if (r < V.world) v[r] = ((const float4*)(src[r] + lo))[i];
        float4 a = v[0];
#pragma unroll
        for (int r = 1; r < kPeerMaxWorld; ++r) if (r < V.world) { a.x += v[r].x; a.y += v[r].y; a.z += v[r].z; a.w += v[r].w; }
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
        ((float4*)(buf + lo))[i] = a;
    }
    for (long long i = lo + 4 * nv + tid; i < hi; i += 256) {
        float a = src[0][i];
        for (int r = 1; r < V.world; ++r) a += src[r][i];
        buf[i] = a * scale;
    }
}

struct PeerState {
    void* window = nullptr;            // this rank's window (own allocation)
    void* mapped[kPeerMaxWorld] = {nullptr};   // the peers' windows as opened here
    PeerView view{};
    const char* kind = "";
    unsigned* fail_host = nullptr;     // pinned, mapped
};

}  // namespace

struct rnde_comm_window {
    void* base = nullptr;
    int device = 0;
    const char* kind = "";
};

namespace {

rnde_status window_alloc(int device, rnde_comm_window** out, uint8_t* handle_out, std::string& err) {
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return RNDE_ERR_NO_DEVICE; }
    rnde_comm_window* w = new rnde_comm_window();
    w->device = device;
    // uncached at the device's L2 first (what the peers read is what was stored), then fine-grained, then plain device memory (the
    // kernel's system-scope fences write back / invalidate either way)
    if (hipExtMallocWithFlags(&w->base, kPeerWindowBytes, hipDeviceMallocUncached) == hipSuccess) w->kind = "uncached";
    else if (hipExtMallocWithFlags(&w->base, kPeerWindowBytes, hipDeviceMallocFinegrained) == hipSuccess) w->kind = "fine-grained";
    else if (hipMalloc(&w->base, kPeerWindowBytes) == hipSuccess) w->kind = "coarse-grained";
    else { (void)hipGetLastError(); err = "window allocation failed"; delete w; return RNDE_ERR_HIP; }
    (void)hipGetLastError();
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == RNDE_COMM_WINDOW_BYTES, "ipc handle size");
    if (hipMemset(w->base, 0, kPeerDataOffset) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, w->base) != hipSuccess) {
        err = std::string("window export failed: ") + hipGetErrorString(hipGetLastError());
        (void)hipFree(w->base); delete w; return RNDE_ERR_HIP;
    }
    std::memcpy(handle_out, &h, sizeof(h));
    *out = w;
    return RNDE_OK;
}

}  // namespace

struct rnde_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    LocalShared* loc = nullptr;   // non-null: a rank of an in-process group
    PeerState* peer = nullptr;    // non-null: peer-mapped windows for the one-shot all-reduce
    unsigned seq = 0, peer_seq = 0;
    bool failed = false;              // sticky: a one-shot all-reduce timed out; every later call of this communicator fails
    long long timeout_ticks = 0;      // one-shot wait limit (100 MHz ticks)
    hipStream_t last_stream = nullptr; bool have_stream = false; hipEvent_t order_ev = nullptr;   // the one-shot protocol needs its launches ordered: see peer_allreduce
    std::string err, path;
};

namespace {

// Map the peers' windows (handles in rank order; this rank's own entry is not opened) and take ownership of `win`.
rnde_status peers_attach(rnde_comm* c, rnde_comm_window* win, const uint8_t* handles, std::string& err) {
    if (c->world > kPeerMaxWorld) { err = "one-shot all-reduce: at most 16 ranks"; return RNDE_ERR_BAD_ARG; }
    PeerState* P = new PeerState();
    P->window = win->base; P->kind = win->kind;
    for (int r = 0; r < c->world; ++r) {
        void* base = win->base;
        if (r != c->rank) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handles + (size_t)r * RNDE_COMM_WINDOW_BYTES, sizeof(h));
            if (hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                err = std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(hipGetLastError());
                for (int q = 0; q < r; ++q) if (P->mapped[q]) (void)hipIpcCloseMemHandle(P->mapped[q]);
                delete P; return RNDE_ERR_HIP;
            }
            P->mapped[r] = base;
        }
        P->view.flags[r] = (unsigned*)base;
        P->view.data[r] = (float*)((char*)base + kPeerDataOffset);
    }
    P->view.fail = (unsigned*)((char*)win->base + kPeerFlagBytes);     // (in the padding page behind the flags)
    if (hipHostMalloc((void**)&P->fail_host, 64, hipHostMallocMapped) != hipSuccess) {
        err = "pinned allocation failed";
        for (int q = 0; q < c->world; ++q) if (P->mapped[q]) (void)hipIpcCloseMemHandle(P->mapped[q]);
        delete P; return RNDE_ERR_HIP;
    }
    *P->fail_host = 0;
    P->view.fail_host = P->fail_host;
    P->view.world = c->world; P->view.rank = c->rank;
    c->peer = P;
    c->path = std::string("one-shot over peer-mapped windows (") + win->kind + " device memory, hipIpc)";
    delete win;
    return RNDE_OK;
}

// "a one-shot all-reduce timed out" as text, with the limit that was in force
std::string peer_timeout_message(const rnde_comm* c) {
    char t[64];
    snprintf(t, sizeof t, "%.3g s", (double)c->timeout_ticks / 1e8);
    return std::string("one-shot all-reduce: a wait for a peer's chunk gave up after ") + t + " (every rank must make the same calls in the same order); the communicator is "
           "failed for good, the buffers of that all-reduce and of every later one hold NaN";
}

rnde_status peer_allreduce(rnde_comm* c, float* buf, long long n, float scale, hipStream_t s) {
    static const long long ticks = getenv("RNDE_ONESHOT_TIMEOUT_MS") ? std::max(1LL, atoll(getenv("RNDE_ONESHOT_TIMEOUT_MS"))) * 100000 : kPeerTimeoutTicks;   // (tests)
    c->timeout_ticks = ticks;
    // sticky failure: a time-out of an EARLIER all-reduce (the kernel raised the pinned word) fails this and every later call -- the training
    // loop meets it at its next enqueue, before another update is computed from poisoned gradients
    if (c->failed || *(volatile unsigned*)c->peer->fail_host != 0u) { c->failed = true; c->err = peer_timeout_message(c); return RNDE_ERR_HIP; }
    // The slot-reuse argument (a rank rewrites a slot at seq + 2 only after every peer has finished reading seq) holds when kernel seq + 1 of a
    // rank cannot start before its kernel seq has finished: true inside one stream.  A call on ANOTHER stream is ordered behind the previous
    // one with an event (the coupled controller enqueues on the node's stream, the gradient reducer on the caller's).
    if (c->have_stream && c->last_stream != s) {
        if (!c->order_ev && hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming) != hipSuccess) { c->err = "one-shot all-reduce: event creation failed"; return RNDE_ERR_HIP; }
        if (hipEventRecord(c->order_ev, c->last_stream) != hipSuccess || hipStreamWaitEvent(s, c->order_ev, 0) != hipSuccess) { c->err = "one-shot all-reduce: cannot order the call behind the previous stream"; return RNDE_ERR_HIP; }
    }
    c->last_stream = s; c->have_stream = true;
    for (long long off = 0; off < n; off += kPeerCap) {
        const long long m = std::min<long long>(kPeerCap, n - off);
        const int G = (int)std::max<long long>(1, std::min<long long>(kPeerBlocks, (m + 1023) / 1024));
        const long long chunk = (((m + G - 1) / G) + 3) & ~3LL;
        const unsigned seq = ++c->peer_seq;
        float* b = buf + off;
        if (((uintptr_t)b & 15) == 0) hipLaunchKernelGGL(rnde_peer_allreduce_kernel<true>, dim3(G), dim3(256), 0, s, c->peer->view, b, m, chunk, seq, scale, ticks);
        else hipLaunchKernelGGL(rnde_peer_allreduce_kernel<false>, dim3(G), dim3(256), 0, s, c->peer->view, b, m, chunk, seq, scale, ticks);
        if (hipGetLastError() != hipSuccess) { c->err = "one-shot all-reduce: launch failed"; return RNDE_ERR_HIP; }
    }
    return RNDE_OK;
}

}  // namespace

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
    // ncclCommInitRank blocks until every rank has arrived.  A rank that never does (a crashed peer, an unreachable fabric, a wrong id) must not
    // hang the caller for good: at world > 1 the call runs on a helper thread and the caller waits a BOUNDED time for it (RNDE_COMM_INIT_TIMEOUT_S,
    // default 180 s; 0 = wait for ever).  On a time-out the helper is left behind (it still owns its own copy of everything it touches) and the call
    // fails loudly with the limit in the message.
    int r = kNcclSuccess;
    double limit_s = 180.0;
    if (const char* e = getenv("RNDE_COMM_INIT_TIMEOUT_S")) limit_s = atof(e);
    if (world == 1 || limit_s <= 0.0) r = a.CommInitRank(&c->comm, world, uid, rank);
    else {
        struct InitJob { std::mutex mu; std::condition_variable cv; bool done = false; int rc = 0; nccl_comm_t comm = nullptr; };
        auto job = std::make_shared<InitJob>();
        RcclApi* ap = &a;
        std::thread([job, ap, world, uid, rank, device] {
            nccl_comm_t cm = nullptr;
            int rc = hipSetDevice(device) == hipSuccess ? ap->CommInitRank(&cm, world, uid, rank) : -1;
            std::lock_guard<std::mutex> lk(job->mu);
            job->rc = rc; job->comm = cm; job->done = true;
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return job->done; })) {
            char t[160];
            snprintf(t, sizeof t, "ncclCommInitRank (rank %d of %d) did not return within %.3g s (RNDE_COMM_INIT_TIMEOUT_S): a rank is absent or the fabric is unreachable", rank, world, limit_s);
            g_comm_err = t; delete c; return RNDE_ERR_HIP;
        }
        r = job->rc; c->comm = job->comm;
    }
    if (r != kNcclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + (r == -1 ? "hipSetDevice failed on the helper thread" : a.GetErrorString(r)); delete c; return RNDE_ERR_HIP; }
    c->path = "RCCL (ncclAllReduce)";
    const char* one = getenv("RNDE_ONESHOT");
    if (one && one[0] == '1' && world > 1) {
        // the windows' handles travel through the communicator that was just built: one 64-byte all-gather
        rnde_comm_window* win = nullptr;
        uint8_t mine[RNDE_COMM_WINDOW_BYTES];
        std::string err;
        rnde_status st = window_alloc(device, &win, mine, err);
        uint8_t* dev = nullptr;
        std::string all((size_t)world * RNDE_COMM_WINDOW_BYTES, '\0');
        if (st == RNDE_OK && !a.AllGather) { err = "RCCL symbol missing: ncclAllGather"; st = RNDE_ERR_HIP; }
        if (st == RNDE_OK && (hipMalloc((void**)&dev, (size_t)(world + 1) * RNDE_COMM_WINDOW_BYTES) != hipSuccess ||
                              hipMemcpy(dev, mine, RNDE_COMM_WINDOW_BYTES, hipMemcpyHostToDevice) != hipSuccess)) { err = "handle staging failed"; st = RNDE_ERR_HIP; }
        if (st == RNDE_OK) {
            const int g = a.AllGather(dev, dev + RNDE_COMM_WINDOW_BYTES, RNDE_COMM_WINDOW_BYTES, /*ncclInt8*/ 0, c->comm, nullptr);
            if (g != kNcclSuccess || hipStreamSynchronize(nullptr) != hipSuccess ||
                hipMemcpy(&all[0], dev + RNDE_COMM_WINDOW_BYTES, all.size(), hipMemcpyDeviceToHost) != hipSuccess) { err = "ncclAllGather of the window handles failed"; st = RNDE_ERR_HIP; }
        }
        if (dev) (void)hipFree(dev);
        if (st == RNDE_OK) st = peers_attach(c, win, (const uint8_t*)all.data(), err);
        if (st != RNDE_OK) {   // asked for and not available: say so rather than fall back silently
            g_comm_err = "RNDE_ONESHOT=1: " + err;
            if (win) { (void)hipFree(win->base); delete win; }
            (void)a.CommDestroy(c->comm); delete c; return st;
        }
    }
    *out = c;
    return RNDE_OK;
}

extern "C" rnde_status rnde_comm_window_create(int32_t device, rnde_comm_window** win_out, uint8_t handle_out[RNDE_COMM_WINDOW_BYTES]) {
    if (!win_out || !handle_out) return RNDE_ERR_BAD_ARG;
    return window_alloc(device, win_out, handle_out, g_comm_err);
}

extern "C" void rnde_comm_window_destroy(rnde_comm_window* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    (void)hipFree(w->base);
    delete w;
}

extern "C" rnde_status rnde_comm_create_peers(rnde_comm_window* win, const uint8_t* handles, int32_t rank, int32_t world, rnde_comm** out) {
    if (!out) return RNDE_ERR_BAD_ARG;
    *out = nullptr;
    if (!win || !handles || world < 1 || world > kPeerMaxWorld || rank < 0 || rank >= world) { g_comm_err = "window / handles / rank / world (1..16) out of range"; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(win->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return RNDE_ERR_NO_DEVICE; }
    rnde_comm* c = new rnde_comm();
    c->rank = rank; c->world = world; c->device = win->device;
    const rnde_status st = peers_attach(c, win, handles, g_comm_err);
    if (st != RNDE_OK) { delete c; return st; }
    *out = c;
    return RNDE_OK;
}

extern "C" const char* rnde_comm_path(const rnde_comm* c) { return c ? c->path.c_str() : ""; }

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
        c->path = "in-process group (host meeting, events)";
        L->refs++;
        out[r] = c;
    }
    return RNDE_OK;
}

extern "C" void rnde_comm_destroy(rnde_comm* c) {
    if (!c) return;
    if (c->comm) (void)api().CommDestroy(c->comm);
    if (c->peer) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (int r = 0; r < c->world; ++r) if (c->peer->mapped[r]) (void)hipIpcCloseMemHandle(c->peer->mapped[r]);
        (void)hipFree(c->peer->window);
        if (c->peer->fail_host) (void)hipHostFree(c->peer->fail_host);
        if (c->order_ev) (void)hipEventDestroy(c->order_ev);
        delete c->peer;
    }
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
    if (c && c->peer) {
        unsigned f = 0;
        if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(&f, c->peer->view.fail, 4, hipMemcpyDeviceToHost) != hipSuccess) { c->err = "one-shot all-reduce: cannot read the window"; return RNDE_ERR_HIP; }
        if (f || c->failed) { c->failed = true; if (!c->timeout_ticks) c->timeout_ticks = kPeerTimeoutTicks; c->err = peer_timeout_message(c); return RNDE_ERR_HIP; }
        return RNDE_OK;
    }
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
    } else if (c->peer && (n <= kPeerCap || !c->comm)) {
        return peer_allreduce(c, buf_dev, n, (mean && c->world > 1) ? 1.0f / (float)c->world : 1.0f, s);   // (the scale rides in the kernel)
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
