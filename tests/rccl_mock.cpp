// rccl_mock.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl that lets several processes SHARING ONE GPU run the
// collectives of toast_hip_comm_* (toast_amd/csrc/comm.cpp).  RCCL itself refuses two ranks on one device ("Duplicate
// GPU detected"), so on a single-GPU box the library's multi-rank logic -- pixel shards, padded reduce-scatter /
// all-gather, owner-computes kernels on real shards, the PCG's dot products summed over the ranks -- could otherwise
// only run with one rank.  The mock stages every collective through a POSIX shared-memory segment: each rank copies
// its send buffer to its slot (after synchronising the stream), a barrier, every rank reduces / gathers what it needs
// on the host and copies the result to its receive buffer, a barrier.  Same results as RCCL up to the order of the
// floating-point sums (ranks are added in rank order); none of its performance.  Selected with
// TOAST_HIP_RCCL_LIB=<path of this library> (comm.cpp opens that instead of librccl).  Never shipped, never timed.
//
//   hipcc -x hip --offload-arch=gfx950 -O2 -fPIC -shared tests/rccl_mock.cpp -o tests/librccl_mock.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

typedef struct MockComm * ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

}  // extern "C"

namespace {

constexpr size_t kSlotBytes = size_t(96) << 20;   // per rank; the tests' maps are far smaller

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int n_ranks;
};

}  // namespace

struct MockComm {
    int n_ranks = 0;
    int rank = -1;
    Header * head = nullptr;
    char * slots = nullptr;   // n_ranks x kSlotBytes
    size_t bytes = 0;
    char name[128];
};

namespace {

void barrier(MockComm * c) {
    Header * h = c->head;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == c->n_ranks) {
        h->arrived.store(0);
        h->generation.fetch_add(1);
    } else {
        while (h->generation.load() == gen) usleep(50);
    }
}

size_t elt_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

template <typename T>
void reduce_into(T * acc, const T * x, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) {
        if (op == ncclSum) acc[i] = acc[i] + x[i];
        else if (op == ncclMax) acc[i] = (x[i] > acc[i]) ? x[i] : acc[i];
        else if (op == ncclMin) acc[i] = (x[i] < acc[i]) ? x[i] : acc[i];
        else acc[i] = acc[i] * x[i];
    }
}

void reduce_any(void * acc, const void * x, size_t n, ncclDataType_t t, ncclRedOp_t op) {
    switch (t) {
        case ncclFloat64: reduce_into((double *)acc, (const double *)x, n, op); break;
        case ncclFloat32: reduce_into((float *)acc, (const float *)x, n, op); break;
        case ncclInt64: reduce_into((int64_t *)acc, (const int64_t *)x, n, op); break;
        case ncclUint64: reduce_into((uint64_t *)acc, (const uint64_t *)x, n, op); break;
        case ncclInt32: reduce_into((int32_t *)acc, (const int32_t *)x, n, op); break;
        case ncclUint32: reduce_into((uint32_t *)acc, (const uint32_t *)x, n, op); break;
        case ncclUint8: reduce_into((uint8_t *)acc, (const uint8_t *)x, n, op); break;
        default: reduce_into((int8_t *)acc, (const int8_t *)x, n, op); break;
    }
}

// every rank's `bytes` of send data -> its slot
ncclResult_t publish(MockComm * c, const void * d_send, size_t bytes, hipStream_t st) {
    if (bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        return ncclUnhandledCudaError;
    }
    barrier(c);
    return ncclSuccess;
}

ncclResult_t deliver(MockComm * c, void * d_recv, const void * host, size_t bytes) {
    const hipError_t e = hipMemcpy(d_recv, host, bytes, hipMemcpyHostToDevice);
    barrier(c);      // nobody overwrites a slot before everybody has read it
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int * v) {
    *v = 1;     // "mock"
    return ncclSuccess;
}

const char * ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "mock: HIP error";
        case ncclSystemError: return "mock: shared memory error";
        default: return "mock: invalid argument (buffer larger than the mock's 96 MB slot?)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId * id) {
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/toast_rccl_mock_%d_%ld", (int)getpid(), (long)random());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t * out, int n_ranks, ncclUniqueId id, int rank) {
    MockComm * c = new MockComm;
    c->n_ranks = n_ranks;
    c->rank = rank;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->bytes = sizeof(Header) + 4096 + (size_t)n_ranks * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd < 0) usleep(500);
        }
        if (fd < 0) return ncclSystemError;
        // wait until rank 0 has sized the segment
        for (int tries = 0; tries < 20000; ++tries) {
            const off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)c->bytes) break;
            usleep(500);
        }
    }
    void * p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->head = static_cast<Header *>(p);
    c->slots = static_cast<char *>(p) + 4096;
    if (rank == 0) c->head->n_ranks = n_ranks;     // (a fresh segment is zero-filled: counters start at 0)
    c->head->attached.fetch_add(1);
    while (c->head->attached.load() < n_ranks) usleep(200);
    barrier(c);
    if (rank == 0) shm_unlink(c->name);            // everybody has it mapped
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (c == nullptr) return ncclSuccess;
    munmap(c->head, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void * send, void * recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
    const size_t bytes = count * elt_size(t);
    const ncclResult_t r = publish(c, send, bytes, st);
    if (r != ncclSuccess) return r;
    std::vector<char> acc(c->slots, c->slots + bytes);                       // rank 0's data, then the others in order
    for (int k = 1; k < c->n_ranks; ++k) reduce_any(acc.data(), c->slots + (size_t)k * kSlotBytes, count, t, op);
    return deliver(c, recv, acc.data(), bytes);
}

ncclResult_t ncclReduceScatter(const void * send, void * recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op,
                               ncclComm_t c, hipStream_t st) {
    const size_t piece = recvcount * elt_size(t);
    const ncclResult_t r = publish(c, send, piece * c->n_ranks, st);
    if (r != ncclSuccess) return r;
    const size_t off = (size_t)c->rank * piece;
    std::vector<char> acc(c->slots + off, c->slots + off + piece);
    for (int k = 1; k < c->n_ranks; ++k) reduce_any(acc.data(), c->slots + (size_t)k * kSlotBytes + off, recvcount, t, op);
    return deliver(c, recv, acc.data(), piece);
}

ncclResult_t ncclAllGather(const void * send, void * recv, size_t sendcount, ncclDataType_t t, ncclComm_t c,
                           hipStream_t st) {
    const size_t piece = sendcount * elt_size(t);
    const ncclResult_t r = publish(c, send, piece, st);
    if (r != ncclSuccess) return r;
    std::vector<char> all(piece * c->n_ranks);
    for (int k = 0; k < c->n_ranks; ++k) std::memcpy(all.data() + (size_t)k * piece, c->slots + (size_t)k * kSlotBytes, piece);
    return deliver(c, recv, all.data(), all.size());
}

ncclResult_t ncclBroadcast(const void * send, void * recv, size_t count, ncclDataType_t t, int root, ncclComm_t c,
                           hipStream_t st) {
    const size_t bytes = count * elt_size(t);
    const ncclResult_t r = publish(c, send, bytes, st);
    if (r != ncclSuccess) return r;
    std::vector<char> data(c->slots + (size_t)root * kSlotBytes, c->slots + (size_t)root * kSlotBytes + bytes);
    return deliver(c, recv, data.data(), bytes);
}

}  // extern "C"
