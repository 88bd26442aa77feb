// RCCL behind the C ABI: the two collectives a sharded map needs -- one broadcast of the shared lookup tables, one
// gather of the per-observation result rows -- plus the max-over-ranks / barrier a benchmark brackets its timed region with.
// One communicator per process (one process per GPU, xGMI underneath); nothing in the fit loop communicates.
//
// librccl.so is loaded on first use (dlopen), not linked: a single-GPU caller of libhipdrt.so never maps it.  The rendezvous
// (how rank 0's ncclUniqueId reaches the other ranks) is the host layer's business: hipdrt_comm_unique_id hands out 128 opaque
// bytes, hipdrt_comm_create takes them back on every rank (hybrid-drt_amd/mapping/dist.py passes them through a file that
// rank 0 writes next to MASTER_PORT; any out-of-band channel does).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>

#include <rccl/rccl.h>

#include "common.hpp"

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char* why = dlerror();           // (a second call would return NULL: dlerror clears its message)
            r.err = std::string("librccl.so not found: ") + (why ? why : "");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p && r.err.empty()) r.err = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

#define RCCL_READY(R)                                                          \
    Rccl* R = rccl();                                                          \
    if (!R->err.empty()) { hipdrt::set_error(R->err); return HIPDRT_E_HIP; }

#define NCCL_CHECK(R, expr)                                                                                     \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            hipdrt::set_error(std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(_r) : "?"));   \
            return HIPDRT_E_HIP;                                                                                \
        }                                                                                                       \
    } while (0)

}  // namespace

struct hipdrt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    hipdrt::DevBuf send, recv;          // staging of the host-buffer entry points (grown on demand)
};

#define COMM_CATCH                                                                                                \
    catch (const std::bad_alloc&) { hipdrt::set_error("out of host memory"); return HIPDRT_E_HIP; }                \
    catch (const std::exception& e) { hipdrt::set_error(std::string("internal error: ") + e.what()); return HIPDRT_E_HIP; } \
    catch (...) { hipdrt::set_error("internal error"); return HIPDRT_E_HIP; }

extern "C" {

int hipdrt_comm_unique_id(char* id128) try {
    HIPDRT_REQUIRE(id128, "NULL pointer");
    RCCL_READY(R);
    ncclUniqueId id;
    NCCL_CHECK(R, R->GetUniqueId(&id));
    static_assert(sizeof(id) == HIPDRT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_comm_create(int device, int rank, int world, const char* id128, hipdrt_comm** out) try {
    HIPDRT_REQUIRE(id128 && out, "NULL pointer");
    HIPDRT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "0 <= rank < world");
    RCCL_READY(R);
    HIPDRT_CHECK(hipSetDevice(device)); (void)hipGetLastError();
    auto* c = new hipdrt_comm();
    c->rank = rank; c->world = world; c->device = device;
    hipdrt::ensure_stream_pool(device);            // the fits' streams get their hardware queues first
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; hipdrt::set_error(hipGetErrorString(e)); return HIPDRT_E_HIP; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        hipdrt::set_error(std::string("ncclCommInitRank: ") + R->GetErrorString(r));
        (void)hipStreamDestroy(c->stream);
        delete c;
        return HIPDRT_E_HIP;
    }
    *out = c;
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_comm_destroy(hipdrt_comm* c) try {
    if (!c) return HIPDRT_OK;
    Rccl* R = rccl();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && R->CommDestroy) (void)R->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_comm_info(hipdrt_comm* c, int* rank, int* world, int* device) try {
    HIPDRT_REQUIRE(c, "NULL pointer");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return HIPDRT_OK;
} COMM_CATCH

// ---- device buffers in, device buffers out ----------------------------------------------------------------------------------
int hipdrt_comm_broadcast_dev(hipdrt_comm* c, double* dev_buf, long long count, int root) try {
    HIPDRT_REQUIRE(c && dev_buf && count >= 0 && root >= 0 && root < c->world, "arguments");
    RCCL_READY(R);
    HIPDRT_CHECK(hipSetDevice(c->device)); (void)hipGetLastError();
    if (count == 0) return HIPDRT_OK;
    NCCL_CHECK(R, R->Broadcast(dev_buf, dev_buf, (size_t)count, ncclFloat64, root, c->comm, c->stream));
    HIPDRT_CHECK(hipStreamSynchronize(c->stream));
    return HIPDRT_OK;
} COMM_CATCH

// every rank sends `count` doubles; `root` receives world x count (rank r's block at r * count), the others' recv may be NULL:
// a true gather -- one ncclSend per rank and world ncclRecv on the root inside ONE group, i.e. one collective on the wire
int hipdrt_comm_gather_dev(hipdrt_comm* c, const double* dev_send, long long count, double* dev_recv, int root) try {
    HIPDRT_REQUIRE(c && count >= 0 && root >= 0 && root < c->world, "arguments");
    HIPDRT_REQUIRE(count == 0 || dev_send, "send buffer is NULL");
    HIPDRT_REQUIRE(c->rank != root || count == 0 || dev_recv, "the root needs a receive buffer");
    RCCL_READY(R);
    HIPDRT_CHECK(hipSetDevice(c->device)); (void)hipGetLastError();
    if (count == 0) return HIPDRT_OK;
    NCCL_CHECK(R, R->GroupStart());
    ncclResult_t r = R->Send(dev_send, (size_t)count, ncclFloat64, root, c->comm, c->stream);
    if (r == ncclSuccess && c->rank == root) {
        for (int p = 0; p < c->world && r == ncclSuccess; ++p)
            r = R->Recv(dev_recv + (size_t)p * count, (size_t)count, ncclFloat64, p, c->comm, c->stream);
    }
    ncclResult_t g = R->GroupEnd();
    NCCL_CHECK(R, r);
    NCCL_CHECK(R, g);
    HIPDRT_CHECK(hipStreamSynchronize(c->stream));
    return HIPDRT_OK;
} COMM_CATCH

// ---- numpy in, numpy out: the same through the communicator's own staging buffers ---------------------------------------------
int hipdrt_comm_broadcast(hipdrt_comm* c, double* host_buf, long long count, int root) try {
    HIPDRT_REQUIRE(c && host_buf && count >= 0 && root >= 0 && root < c->world, "arguments");
    if (count == 0) return HIPDRT_OK;
    HIPDRT_CHECK(hipSetDevice(c->device)); (void)hipGetLastError();
    const size_t bytes = (size_t)count * sizeof(double);
    if (c->send.bytes < bytes) HIPDRT_CHECK(c->send.alloc(bytes));
    if (c->rank == root) HIPDRT_CHECK(hipMemcpyAsync(c->send.p, host_buf, bytes, hipMemcpyHostToDevice, c->stream));
    int rc = hipdrt_comm_broadcast_dev(c, c->send.d(), count, root);
    if (rc) return rc;
    if (c->rank != root) {
        HIPDRT_CHECK(hipMemcpyAsync(host_buf, c->send.p, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPDRT_CHECK(hipStreamSynchronize(c->stream));
    }
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_comm_gather(hipdrt_comm* c, const double* host_send, long long count, double* host_recv, int root) try {
    HIPDRT_REQUIRE(c && count >= 0 && root >= 0 && root < c->world, "arguments");
    HIPDRT_REQUIRE(count == 0 || host_send, "send buffer is NULL");
    HIPDRT_REQUIRE(c->rank != root || count == 0 || host_recv, "the root needs a receive buffer");
    if (count == 0) return HIPDRT_OK;
    HIPDRT_CHECK(hipSetDevice(c->device)); (void)hipGetLastError();
    const size_t bytes = (size_t)count * sizeof(double);
    if (c->send.bytes < bytes) HIPDRT_CHECK(c->send.alloc(bytes));
    if (c->rank == root && c->recv.bytes < bytes * c->world) HIPDRT_CHECK(c->recv.alloc(bytes * c->world));
    HIPDRT_CHECK(hipMemcpyAsync(c->send.p, host_send, bytes, hipMemcpyHostToDevice, c->stream));
    int rc = hipdrt_comm_gather_dev(c, c->send.d(), count, c->rank == root ? c->recv.d() : nullptr, root);
    if (rc) return rc;
    if (c->rank == root) {
        HIPDRT_CHECK(hipMemcpyAsync(host_recv, c->recv.p, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
        HIPDRT_CHECK(hipStreamSynchronize(c->stream));
    }
    return HIPDRT_OK;
} COMM_CATCH

// max over the ranks of one double, in every rank (the benchmark's "slowest rank" time); with value == NULL a plain barrier
int hipdrt_comm_allreduce_max(hipdrt_comm* c, double* value) try {
    HIPDRT_REQUIRE(c, "NULL pointer");
    RCCL_READY(R);
    HIPDRT_CHECK(hipSetDevice(c->device)); (void)hipGetLastError();
    if (c->send.bytes < sizeof(double)) HIPDRT_CHECK(c->send.alloc(sizeof(double)));
    double v = value ? *value : 0.0;
    HIPDRT_CHECK(hipMemcpyAsync(c->send.p, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCL_CHECK(R, R->AllReduce(c->send.p, c->send.p, 1, ncclFloat64, ncclMax, c->comm, c->stream));
    HIPDRT_CHECK(hipMemcpyAsync(&v, c->send.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPDRT_CHECK(hipStreamSynchronize(c->stream));
    if (value) *value = v;
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_comm_barrier(hipdrt_comm* c) { return hipdrt_comm_allreduce_max(c, nullptr); }

// device memory for callers that keep matrices resident (the *_dev entry points take such pointers)
int hipdrt_device_alloc(hipdrt_ctx* ctx, long long bytes, void** out) try {
    HIPDRT_REQUIRE(ctx && out && bytes >= 0, "arguments");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    HIPDRT_CHECK(hipMalloc(out, (size_t)(bytes > 0 ? bytes : 8)));
    return HIPDRT_OK;
} COMM_CATCH

// 0 when device `device` exists and is a gfx950 part; creates nothing on it (no context, no stream)
int hipdrt_device_probe(int device) try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); return HIPDRT_E_NODEVICE; }
    hipDeviceProp_t prop;
    HIPDRT_CHECK(hipGetDeviceProperties(&prop, device));
    return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 ? HIPDRT_OK : HIPDRT_E_NODEVICE;
} COMM_CATCH

// every stream of the context's device drained (hipDeviceSynchronize: what a benchmark brackets its timed region with)
int hipdrt_device_synchronize(hipdrt_ctx* ctx) try {
    HIPDRT_REQUIRE(ctx, "NULL pointer");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    HIPDRT_CHECK(hipDeviceSynchronize());
    return HIPDRT_OK;
} COMM_CATCH

int hipdrt_device_free(hipdrt_ctx* ctx, void* ptr) try {
    HIPDRT_REQUIRE(ctx, "NULL pointer");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    if (ptr) HIPDRT_CHECK(hipFree(ptr));
    return HIPDRT_OK;
} COMM_CATCH

}  // extern "C"
