// rccl_double.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that moves collectives between processes sharing ONE GPU box
// through a POSIX shared-memory segment (device -> host staging -> device).  RCCL refuses two ranks on one device
// ("invalid usage" in ncclCommInitRank), so on a one-GPU box the receiving half of the fan-out -- ggrs_hip_fanout_sync_confirmed
// on rank != 0, adopt, per-rank branch lists, the gathered table -- could never run.  With GGRS_RCCL_LIB pointing here,
// libggrs_hip.so's fan-out code runs unchanged at world size 2 and only the transport is replaced.  It implements exactly the
// entry points the library resolves: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclBroadcast, ncclAllGather,
// ncclGetErrorString, ncclCommCount, ncclCommUserRank.  Collectives are host-synchronous (they drain the stream they are given).
//
// build: g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_double.cpp -o tests/cpp/_build/librccl_double.so -L/opt/rocm/lib -lamdhip64 -lrt
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

constexpr size_t STAGE_BYTES = 256u << 20;
struct Shared {
    std::atomic<int> arrived;       // sense-reversing barrier
    std::atomic<int> generation;
    char pad[248];
    unsigned char data[1];          // STAGE_BYTES of staging
};
struct Comm { Shared* sh; int rank, size; char name[64]; };
typedef Comm* ncclComm_t;

static size_t dtype_bytes(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: case ncclBfloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
static void barrier(Comm* c) {
    const int gen = c->sh->generation.load(std::memory_order_acquire);
    if (c->sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->size) {
        c->sh->arrived.store(0, std::memory_order_relaxed);
        c->sh->generation.fetch_add(1, std::memory_order_release);
    } else {
        while (c->sh->generation.load(std::memory_order_acquire) == gen) sched_yield();
    }
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (rccl_double)" : "rccl_double: shared-memory transport error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/ggrs_rccl_double_%d_%ld", (int)getpid(), (long)random());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const size_t bytes = sizeof(Shared) + STAGE_BYTES;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }      // a fresh segment is zero-filled: counters start at 0
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm{(Shared*)p, rank, nranks, {0}};
    strncpy(c->name, id.internal, sizeof c->name - 1);
    barrier(c);                                    // everyone is attached
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->sh, sizeof(Shared) + STAGE_BYTES);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->size; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t stream) {
    const size_t bytes = count * dtype_bytes(t);
    if (bytes > STAGE_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && hipMemcpy(c->sh->data, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (c->rank != root && hipMemcpy(recv, c->sh->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && recv != send && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t stream) {
    const size_t bytes = sendcount * dtype_bytes(t);
    if (bytes * (size_t)c->size > STAGE_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (hipMemcpy(recv, c->sh->data, bytes * (size_t)c->size, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    return ncclSuccess;
}

}  // extern "C"
