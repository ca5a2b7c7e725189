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
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
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

__global__ void rnde_scale_kernel(float* __restrict__ v, long long n, float s) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) v[i] *= s;
}

}  // namespace

struct rnde_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
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

extern "C" void rnde_comm_destroy(rnde_comm* c) {
    if (!c) return;
    if (c->comm) (void)api().CommDestroy(c->comm);
    delete c;
}

extern "C" int32_t rnde_comm_world(const rnde_comm* c) { return c ? c->world : 0; }

// In-place sum over the ranks of `n` floats, then (mean != 0) a scale by 1 / world -- both on `stream`, asynchronous.
extern "C" rnde_status rnde_comm_allreduce(rnde_comm* c, float* buf_dev, int64_t n, int32_t mean, void* stream) {
    if (!c || !buf_dev || n < 0) return RNDE_ERR_BAD_ARG;
    if (n == 0) return RNDE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return RNDE_ERR_HIP; }
    const int r = api().AllReduce(buf_dev, buf_dev, (size_t)n, kNcclFloat32, kNcclSum, c->comm, s);
    if (r != kNcclSuccess) { c->err = std::string("ncclAllReduce: ") + api().GetErrorString(r); return RNDE_ERR_HIP; }
    if (mean && c->world > 1) {
        hipLaunchKernelGGL(rnde_scale_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, buf_dev, (long long)n, 1.0f / (float)c->world);
        if (hipGetLastError() != hipSuccess) { c->err = "scale kernel launch failed"; return RNDE_ERR_HIP; }
    }
    return RNDE_OK;
}
