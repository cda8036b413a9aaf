// rt_comm.hip — the frame reduce of multi-GPU rendering (include/rt_hip.h: rt_comm_*, rt_reduce).
//
// The reference is one process with a thread pool (main.cpp:954-1005); its multi-GPU lift (SURVEY 8 e) shards sample
// passes or image tiles over GPUs with NO data-path collective and closes the frame with ONE sum of the fp32 radiance
// buffers into rank 0 over RCCL / xGMI: 24.9 MB at 1080p, 99.5 MB at 4K.  xGMI is point-to-point (7 links per GPU): a
// reduce to one root moves 7 buffers into it over 7 independent links, ~0.65 ms at 4K against seconds of rendering.
//
// RCCL is loaded at run time (dlopen), not linked: a process that has torch loaded gets torch's own librccl (the same
// soname resolves to the copy already mapped), a plain C++ host gets ROCm's.  Single-GPU use never touches it.
//
// Two ways to make a communicator:
//   rt_comm_create        one rank per process or thread, ranks meet through a 128-byte id (ncclCommInitRank)
//   rt_comm_create_local  all ranks in this process, one per listed device (ncclCommInitAll)
// A local communicator whose device list names one GPU more than once cannot be an RCCL communicator (RCCL refuses
// duplicate GPUs).  It is a REHEARSAL communicator: the ranks meet at a host barrier and the root adds the other ranks'
// buffers with a kernel, in rank order — for exercising the multi-rank control path on a one-GPU box, nothing else.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

int rt_fail_message(int code, const std::string& msg); // rt_hip.hip: sets the calling thread's rt_last_error text

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl* rccl() { // loaded once; nullptr-lib + error text when it cannot be
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); // the copy this process already has (torch's), if any
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            const char* e = dlerror(); // one call: dlerror() hands the message out once and clears it
            r.error = std::string("cannot load librccl: ") + (e ? e : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p) {
                ok = false;
                r.error = std::string("librccl lacks ") + name;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) r.lib = nullptr;
    });
    return &r;
}

// Rehearsal communicator (ranks sharing a GPU): what the ranks of one rt_comm_create_local call have in common.
struct SharedDevice {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<float*> bufs;
    size_t n = 0;
    int status = RT_OK;       // of the round being gathered
    int done_status = RT_OK;  // of the round that finished last: what that round's waiters read (a faster rank may already
                              // have opened the next round and reset `status`)
};

__global__ __launch_bounds__(256) void k_add_into(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

} // namespace

struct rt_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;                 // RCCL communicator (null for a rehearsal communicator)
    std::shared_ptr<SharedDevice> shared;      // rehearsal communicator
    hipStream_t stream = nullptr;
};

#define COMM_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return rt_fail_message(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int rt_comm_unique_id(uint8_t* id128) {
    if (!id128) return rt_fail_message(RT_ERR_INVALID, "rt_comm_unique_id: null argument");
    Rccl* r = rccl();
    if (!r->lib) return rt_fail_message(RT_ERR_HIP, "rt_comm_unique_id: " + r->error);
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return rt_fail_message(RT_ERR_HIP, std::string("ncclGetUniqueId: ") + r->GetErrorString(rc));
    static_assert(sizeof(id) == RT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return RT_OK;
}

int rt_comm_create(const uint8_t* id128, int rank, int world, int device, rt_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return rt_fail_message(RT_ERR_INVALID, "rt_comm_create: bad argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rt_fail_message(RT_ERR_NO_DEVICE, "rt_comm_create: no HIP device");
    if (device < 0 || device >= n) return rt_fail_message(RT_ERR_INVALID, "rt_comm_create: device index out of range");
    Rccl* r = rccl();
    if (!r->lib) return rt_fail_message(RT_ERR_HIP, "rt_comm_create: " + r->error);
    COMM_HIP(hipSetDevice(device));
    rt_comm* c = new rt_comm;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) {
        delete c;
        return rt_fail_message(RT_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        r->CommDestroy(c->comm);
        delete c;
        return rt_fail_message(RT_ERR_HIP, "rt_comm_create: hipStreamCreate failed");
    }
    *out = c;
    return RT_OK;
}

int rt_comm_create_local(const int* devices, int n_devices, rt_comm** out) {
    if (!devices || !out || n_devices < 1) return rt_fail_message(RT_ERR_INVALID, "rt_comm_create_local: bad argument");
    for (int i = 0; i < n_devices; i++) out[i] = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rt_fail_message(RT_ERR_NO_DEVICE, "rt_comm_create_local: no HIP device");
    std::set<int> distinct;
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= n) return rt_fail_message(RT_ERR_INVALID, "rt_comm_create_local: device index out of range");
        distinct.insert(devices[i]);
    }
    std::vector<rt_comm*> made;
    auto undo = [&]() {
        for (rt_comm* c : made) rt_comm_destroy(c);
        for (int i = 0; i < n_devices; i++) out[i] = nullptr;
    };
    if ((int)distinct.size() == n_devices) { // one GPU per rank: RCCL
        Rccl* r = rccl();
        if (!r->lib) return rt_fail_message(RT_ERR_HIP, "rt_comm_create_local: " + r->error);
        std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
        const ncclResult_t rc = r->CommInitAll(comms.data(), n_devices, devices);
        if (rc != ncclSuccess) return rt_fail_message(RT_ERR_HIP, std::string("ncclCommInitAll: ") + r->GetErrorString(rc));
        for (int i = 0; i < n_devices; i++) {
            rt_comm* c = new rt_comm;
            c->rank = i;
            c->world = n_devices;
            c->device = devices[i];
            c->comm = comms[(size_t)i];
            made.push_back(c);
            if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
                for (int j = i + 1; j < n_devices; j++) r->CommDestroy(comms[(size_t)j]);
                undo();
                return rt_fail_message(RT_ERR_HIP, "rt_comm_create_local: hipStreamCreate failed");
            }
            out[i] = c;
        }
        return RT_OK;
    }
    if (distinct.size() != 1) return rt_fail_message(RT_ERR_INVALID, "rt_comm_create_local: a device may be listed twice only when ALL ranks share one device (rehearsal)");
    auto shared = std::make_shared<SharedDevice>();
    shared->world = n_devices;
    shared->bufs.assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; i++) {
        rt_comm* c = new rt_comm;
        c->rank = i;
        c->world = n_devices;
        c->device = devices[i];
        c->shared = shared;
        made.push_back(c);
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            undo();
            return rt_fail_message(RT_ERR_HIP, "rt_comm_create_local: hipStreamCreate failed");
        }
        out[i] = c;
    }
    return RT_OK;
}

int rt_comm_rank(const rt_comm* c) { return c ? c->rank : -1; }
int rt_comm_world(const rt_comm* c) { return c ? c->world : 0; }

int rt_reduce(rt_comm* c, float* buf, size_t n, int root) {
    if (!c || !buf || root < 0 || root >= c->world) return rt_fail_message(RT_ERR_INVALID, "rt_reduce: bad argument");
    if (n == 0) return RT_OK;
    COMM_HIP(hipSetDevice(c->device));
    if (c->comm) { // sum of every rank's buffer into the root's, in place, then this rank's part of the operation has finished
        Rccl* r = rccl();
        const ncclResult_t rc = r->Reduce(buf, buf, n, ncclFloat32, ncclSum, root, c->comm, c->stream);
        if (rc != ncclSuccess) return rt_fail_message(RT_ERR_HIP, std::string("ncclReduce: ") + r->GetErrorString(rc));
        COMM_HIP(hipStreamSynchronize(c->stream));
        return RT_OK;
    }
    // rehearsal: all ranks on one GPU.  Everyone registers its buffer; the last to arrive (whoever that is) does the
    // root's additions in rank order on its own stream, then releases the others.
    SharedDevice& sh = *c->shared;
    std::unique_lock<std::mutex> lock(sh.mu);
    const uint64_t gen = sh.generation;
    if (sh.arrived == 0) {
        sh.n = n;
        sh.status = RT_OK;
    } else if (sh.n != n) {
        sh.status = RT_ERR_INVALID;
    }
    sh.bufs[(size_t)c->rank] = buf;
    if (++sh.arrived == sh.world) {
        if (sh.status == RT_OK) {
            for (int k = 0; k < sh.world && sh.status == RT_OK; k++) {
                if (k == root) continue;
                hipLaunchKernelGGL(k_add_into, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sh.bufs[(size_t)root], sh.bufs[(size_t)k], n);
                if (hipGetLastError() != hipSuccess) sh.status = RT_ERR_HIP;
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) sh.status = RT_ERR_HIP;
        }
        sh.done_status = sh.status;
        sh.arrived = 0;
        sh.generation++;
        sh.cv.notify_all();
    } else {
        // (a waiter leaves with the mutex held before any rank can finish the NEXT round: that round needs this rank too)
        sh.cv.wait(lock, [&] { return sh.generation != gen; });
    }
    const int status = sh.done_status;
    lock.unlock();
    if (status == RT_ERR_INVALID) return rt_fail_message(status, "rt_reduce: ranks passed different element counts");
    if (status != RT_OK) return rt_fail_message(status, "rt_reduce: the shared-device add failed");
    return RT_OK;
}

// Device frames for hosts that are not compiled with HIP (the C++ mirror's multi-GPU Renderer): plain wrappers.
int rt_buffer_alloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return rt_fail_message(RT_ERR_INVALID, "rt_buffer_alloc: bad argument");
    *out = nullptr;
    COMM_HIP(hipSetDevice(device));
    COMM_HIP(hipMalloc(out, bytes));
    return RT_OK;
}
int rt_buffer_download(int device, const void* src, void* host_dst, size_t bytes) {
    if (!src || !host_dst) return rt_fail_message(RT_ERR_INVALID, "rt_buffer_download: bad argument");
    COMM_HIP(hipSetDevice(device));
    COMM_HIP(hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}
int rt_buffer_zero(int device, void* p, size_t bytes) {
    if (!p) return rt_fail_message(RT_ERR_INVALID, "rt_buffer_zero: bad argument");
    COMM_HIP(hipSetDevice(device));
    COMM_HIP(hipMemset(p, 0, bytes));
    COMM_HIP(hipDeviceSynchronize());
    return RT_OK;
}
void rt_buffer_free(int device, void* p) {
    if (!p) return;
    (void)hipSetDevice(device);
    (void)hipFree(p);
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm) rccl()->CommDestroy(c->comm);
    delete c;
}

} // extern "C"
