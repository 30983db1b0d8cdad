// comm.cpp -- the process' RCCL communicator behind the C ABI.
//
// One process per GPU.  The collectives of the map-making path (sum of the noise-weighted map over the detector
// shards, the scalar sums of the PCG, the union of the hit submaps) are enqueued on the stream the kernels run on, so a
// PCG iteration needs no host synchronisation around them: kernel -> collective -> kernel is plain stream order.
//
// Reference semantics: PixelData.sync_allreduce (src/toast/pixels.py:710-780: every process ends with the sum) and
// PixelData.sync_alltoallv (src/toast/pixels.py:942-967: every submap goes to ONE owner, the owner runs `local_func`
// on it, the result goes back to every holder).  With one process per GPU all ranks hold the same local submaps
// (the union of the hit submaps), so "owner of a pixel" = the rank whose contiguous pixel shard contains it:
// reduce-scatter (owners receive the sum) -> per-pixel kernel on the owned shard -> all-gather (results go back).
//
// librccl is opened at run time (dlopen) the first time a communicator is asked for: a single-GPU process needs no
// RCCL at all, and a process that already carries an RCCL (PyTorch-ROCm bundles one) keeps using that one copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include "../../include/toast_hip.h"
#include "runtime.hpp"

namespace {

using namespace toast_hip;

struct Rccl {
    void * handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclGetVersion) get_version = nullptr;
};

template <typename F>
void resolve(void * h, const char * name, F & fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (fn == nullptr) throw Error(TOAST_HIP_ERR_DEVICE, std::string("HipComm:  librccl has no symbol ") + name);
}

Rccl & rccl() {
    static Rccl ready;
    if (ready.handle != nullptr) return ready;
    Rccl r;      // published only when every entry point has been found (a later call must not see half a table)
    // TOAST_HIP_RCCL_LIB names the library outright (a site's own RCCL build; the tests' shared-memory stand-in that
    // lets several ranks share one GPU, tests/rccl_mock.cpp): no search, and a failure to open it is an error
    if (const char * forced = std::getenv("TOAST_HIP_RCCL_LIB"); forced != nullptr && forced[0] != '\0') {
        r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (r.handle == nullptr) {
            const char * e = dlerror();
            throw Error(TOAST_HIP_ERR_DEVICE,
                        std::string("HipComm:  cannot open TOAST_HIP_RCCL_LIB=") + forced + " (" + (e ? e : "?") + ")");
        }
    }
    // the copy this process already carries first (RTLD_NOLOAD), then the loader's search path, then ROCm's
    const char * loaded[] = {"librccl.so", "librccl.so.1"};
    for (const char * n : loaded) {
        if (r.handle == nullptr) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char * n : names) {
        if (r.handle == nullptr) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (r.handle == nullptr) {
        const char * e = dlerror();
        throw Error(TOAST_HIP_ERR_DEVICE, std::string("HipComm:  cannot open librccl (") + (e ? e : "?") + ")");
    }
    resolve(r.handle, "ncclGetUniqueId", r.get_unique_id);
    resolve(r.handle, "ncclCommInitRank", r.comm_init_rank);
    resolve(r.handle, "ncclCommDestroy", r.comm_destroy);
    resolve(r.handle, "ncclAllReduce", r.all_reduce);
    resolve(r.handle, "ncclReduceScatter", r.reduce_scatter);
    resolve(r.handle, "ncclAllGather", r.all_gather);
    resolve(r.handle, "ncclBroadcast", r.broadcast);
    resolve(r.handle, "ncclGetErrorString", r.error_string);
    resolve(r.handle, "ncclGetVersion", r.get_version);
    ready = r;
    return ready;
}

void check(ncclResult_t rc, const char * what) {
    if (rc == ncclSuccess) return;
    std::ostringstream o;
    o << "HipComm:  " << what << " failed: " << rccl().error_string(rc);
    throw Error(TOAST_HIP_ERR_DEVICE, o.str());
}

ncclComm_t g_comm = nullptr;
int g_size = 0;
int g_rank = -1;

ncclComm_t comm() {
    if (g_comm == nullptr) {
        throw Error(TOAST_HIP_ERR_DEVICE, "HipComm:  no communicator, call toast_hip_comm_init() first");
    }
    return g_comm;
}

struct DType {
    ncclDataType_t nccl;
    size_t bytes;
};

DType dtype_of(int code) {
    switch (code) {
        case TOAST_HIP_COMM_F64: return {ncclFloat64, 8};
        case TOAST_HIP_COMM_F32: return {ncclFloat32, 4};
        case TOAST_HIP_COMM_I64: return {ncclInt64, 8};
        case TOAST_HIP_COMM_I32: return {ncclInt32, 4};
        case TOAST_HIP_COMM_U8: return {ncclUint8, 1};
        default: fail_arg("HipComm:  unknown data type code");
    }
}

ncclRedOp_t op_of(int code) {
    switch (code) {
        case TOAST_HIP_COMM_SUM: return ncclSum;
        case TOAST_HIP_COMM_MAX: return ncclMax;
        case TOAST_HIP_COMM_MIN: return ncclMin;
        default: fail_arg("HipComm:  unknown reduction code");
    }
}

inline hipStream_t as_stream(void * s) { return static_cast<hipStream_t>(s); }

// Pixel shards: rank r owns pixels [r * per, min((r + 1) * per, n_px)), per = ceil(n_px / size).
struct Shard {
    int64_t per;     // pixels per rank (the last ranks may own fewer, or none)
    int64_t first;   // first owned pixel
    int64_t count;   // owned pixels (>= 0)
    bool even;       // n_px == per * size: the collectives run in place on the map
};

Shard shard_of(int64_t n_px, int size, int rank) {
    Shard s;
    s.per = (n_px + size - 1) / size;
    s.first = (int64_t)rank * s.per;
    const int64_t last = (s.first + s.per < n_px) ? s.first + s.per : n_px;
    s.count = (last > s.first) ? last - s.first : 0;
    if (s.first > n_px) s.first = n_px;
    s.even = (s.per * size == n_px);
    return s;
}

Shard shard_of(int64_t n_px) { return shard_of(n_px, g_size, g_rank); }

void peer_release();      // (mode "peer", below: its exchange buffers go before the communicator does)
bool peer_error_pending();
// mode "peer*": a sum of doubles goes through the exchange buffers like the maps do (so that the headline step of
// bench.py A/Bs the exchange, not only the owner-computes pass); anything else: false, the caller uses RCCL
bool peer_allreduce(void * d_buf, int64_t count, int dtype, int op, hipStream_t st);

// `work` = the buffer the two ring halves run on: the map itself when the pixels divide evenly, else a zero-padded
// scratch copy of per * size pixels (kScratchCommA / B of the manager's grow-only scratch buffers).
double * padded_copy(const double * d_map, int64_t n_px, int64_t nv, const Shard & s, int slot, hipStream_t st) {
    const size_t bytes = (size_t)s.per * g_size * nv * sizeof(double);
    double * w = static_cast<double *>(Manager::get().scratch(slot, bytes));
    TH_HIP(hipMemcpyAsync(w, d_map, (size_t)n_px * nv * sizeof(double), hipMemcpyDeviceToDevice, st));
    const size_t pad = bytes - (size_t)n_px * nv * sizeof(double);
    if (pad > 0) TH_HIP(hipMemsetAsync(w + n_px * nv, 0, pad, st));
    return w;
}

void all_gather_pixels(double * d_data, int64_t n_px, int64_t nv, const Shard & s, int slot, hipStream_t st) {
    if (s.even) {
        check(rccl().all_gather(d_data + s.first * nv, d_data, (size_t)(s.per * nv), ncclFloat64, comm(), st),
              "ncclAllGather");
        return;
    }
    double * w = padded_copy(d_data, n_px, nv, s, slot, st);
    check(rccl().all_gather(w + (int64_t)g_rank * s.per * nv, w, (size_t)(s.per * nv), ncclFloat64, comm(), st),
          "ncclAllGather");
    TH_HIP(hipMemcpyAsync(d_data, w, (size_t)n_px * nv * sizeof(double), hipMemcpyDeviceToDevice, st));
}

}  // namespace

extern "C" {

int toast_hip_comm_available(void) {
    // 1 when librccl can be opened and has every entry point this file uses; never throws, creates nothing.  Lets all
    // ranks agree BEFORE the collective toast_hip_comm_init (a rank that cannot load RCCL would leave the others
    // waiting inside ncclCommInitRank).
    try {
        (void)rccl();
        return 1;
    } catch (...) {
        return 0;
    }
}

int toast_hip_comm_unique_id(void * id128) {
    return guarded([&] {
        if (id128 == nullptr) fail_arg("HipComm:  unique id buffer is null");
        static_assert(sizeof(ncclUniqueId) == TOAST_HIP_COMM_ID_BYTES, "unique id size");
        ncclUniqueId id;
        check(rccl().get_unique_id(&id), "ncclGetUniqueId");
        std::memcpy(id128, &id, sizeof(id));
    });
}

int toast_hip_comm_init(const void * id128, int n_ranks, int rank) {
    return guarded([&] {
        if (id128 == nullptr) fail_arg("HipComm:  unique id buffer is null");
        if (n_ranks < 1 || rank < 0 || rank >= n_ranks) fail_arg("HipComm:  need 0 <= rank < n_ranks");
        if (g_comm != nullptr) {
            throw Error(TOAST_HIP_ERR_DEVICE, "HipComm:  a communicator already exists, call toast_hip_comm_destroy() first");
        }
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ncclComm_t c = nullptr;
        check(rccl().comm_init_rank(&c, n_ranks, id, rank), "ncclCommInitRank");
        g_comm = c;
        g_size = n_ranks;
        g_rank = rank;
    });
}

int toast_hip_comm_info(int * n_ranks, int * rank, int * rccl_version) {
    return guarded([&] {
        if (n_ranks) *n_ranks = g_comm ? g_size : 0;
        if (rank) *rank = g_comm ? g_rank : -1;
        if (rccl_version) {
            *rccl_version = 0;
            if (g_comm) check(rccl().get_version(rccl_version), "ncclGetVersion");
        }
    });
}

int toast_hip_comm_destroy(void) {
    return guarded([&] {
        if (g_comm == nullptr) return;
        const bool gave_up = peer_error_pending();
        peer_release();
        ncclComm_t c = g_comm;
        g_comm = nullptr;
        g_size = 0;
        g_rank = -1;
        check(rccl().comm_destroy(c), "ncclCommDestroy");
        if (gave_up) {
            throw Error(TOAST_HIP_ERR_DEVICE, "HipComm:  mode 'peer:flags': a wait for a peer's flag gave up in the last reduction "
                                              "before the communicator was destroyed; its maps were not valid");
        }
    });
}

int toast_hip_comm_allreduce_dev(void * d_buf, int64_t count, int dtype, int op, void * stream) {
    return guarded([&] {
        if (count <= 0) return;
        const DType t = dtype_of(dtype);
        if (peer_allreduce(d_buf, count, dtype, op, as_stream(stream))) return;
        check(rccl().all_reduce(d_buf, d_buf, (size_t)count, t.nccl, op_of(op), comm(), as_stream(stream)), "ncclAllReduce");
    });
}

int toast_hip_comm_broadcast_dev(void * d_buf, int64_t count, int dtype, int root, void * stream) {
    return guarded([&] {
        if (count <= 0) return;
        const DType t = dtype_of(dtype);
        check(rccl().broadcast(d_buf, d_buf, (size_t)count, t.nccl, root, comm(), as_stream(stream)), "ncclBroadcast");
    });
}

int toast_hip_comm_reduce_scatter_dev(const void * d_send, void * d_recv, int64_t recv_count, int dtype, int op,
                                      void * stream) {
    return guarded([&] {
        if (recv_count <= 0) return;
        const DType t = dtype_of(dtype);
        check(rccl().reduce_scatter(d_send, d_recv, (size_t)recv_count, t.nccl, op_of(op), comm(), as_stream(stream)),
              "ncclReduceScatter");
    });
}

int toast_hip_comm_all_gather_dev(const void * d_send, void * d_recv, int64_t send_count, int dtype, void * stream) {
    return guarded([&] {
        if (send_count <= 0) return;
        const DType t = dtype_of(dtype);
        check(rccl().all_gather(d_send, d_recv, (size_t)send_count, t.nccl, comm(), as_stream(stream)), "ncclAllGather");
    });
}

int toast_hip_comm_shard_of(int64_t n_px, int n_ranks, int rank, int64_t * first, int64_t * count, int64_t * per_rank) {
    return guarded([&] {
        if (n_px < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) fail_arg("HipComm:  need n_px >= 0 and 0 <= rank < n_ranks");
        const Shard s = shard_of(n_px, n_ranks, rank);
        if (first) *first = s.first;
        if (count) *count = s.count;
        if (per_rank) *per_rank = s.per;
    });
}

int toast_hip_comm_pixel_shard(int64_t n_px, int64_t * first, int64_t * count) {
    return guarded([&] {
        (void)comm();
        const Shard s = shard_of(n_px);
        if (first) *first = s.first;
        if (count) *count = s.count;
    });
}

// ------------------------------------------------------------------------------------
// map <- [C .] sum over ranks (map): the middle of every PCG iteration and the finalisation of a binned map.
// Four ways to do it, switchable at run time so that a multi-GPU run can compare them on its own data
// (TOAST_HIP_COMM_MODE / toast_hip_comm_set_mode; bench.py reports all of them per N):
//   owner      (default) reduce-scatter of the pixel shards, cov_apply_diag on the owned shard, all-gather -- all on the
//              caller's stream.  2 (N-1)/N map volumes over the links, the multiplication done once per pixel.
//              (Round 4 also had "sliced:S", the same in S pixel slices on two side streams of this ONE communicator.  RCCL
//              orders the operations of a communicator, so the reduce-scatter of slice k + 1 never ran under the all-gather
//              of slice k, and two ring phases on the same links would not gain from it anyway: removed.)
//   allreduce  one all-reduce of the whole map, then every rank multiplies the whole map (what sync_allreduce +
//              covariance_apply do; the reference's default, pixels.py:710-780).
//   peer       no RCCL on the data path: xGMI is a full mesh of point-to-point links, so every rank WRITES the seven
//              foreign slices of its map straight into their owners' inboxes (memory opened through hipIpc handles, all
//              links busy at once), owners add the inboxes in rank order, multiply, and every rank READS the seven
//              finished slices from their owners.  (N-1)/N map volumes per direction and rank, each link carrying 1/N of
//              the map per phase instead of a ring's N-1 steps over one link; the sum has a fixed order.  RCCL only
//              provides the two barriers (a one-word all-reduce each) and the exchange of the handles.
//   peer:flags the same with the barriers done by the ranks themselves: after its phase every rank stores the number of
//              the reduction into a word of every peer's (uncached, hipIpc-opened) flag block, and a one-workgroup kernel
//              on the stream waits until all peers' words have arrived.  No library call per reduction at all.
// reduce = 0: the map is already the same on all ranks (the reference's covariance_apply(use_alltoallv=True),
// covariance.py:224-306): owners apply, results are gathered (allreduce mode: every rank applies, no communication).
}  // extern "C"

namespace {

struct CommMode {
    int kind = 0;      // 0 owner, 2 allreduce, 3 peer
    int slices = 0;    // peer: 1 = the barriers are device flags
};
CommMode g_mode;
bool g_mode_read = false;

CommMode parse_mode(const char * text) {
    CommMode m;
    const std::string v(text ? text : "");
    if (v.empty() || v == "owner") return m;
    if (v == "allreduce") {
        m.kind = 2;
        return m;
    }
    if (v == "peer" || v == "peer:flags") {
        m.kind = 3;
        m.slices = (v == "peer") ? 0 : 1;      // 1: the two barriers are device flags, not RCCL calls
        return m;
    }
    fail_arg("HipComm:  unknown mode '" + v + "' (owner | allreduce | peer[:flags])");
}

const CommMode & mode() {
    if (!g_mode_read) {
        g_mode = parse_mode(std::getenv("TOAST_HIP_COMM_MODE"));
        g_mode_read = true;
    }
    return g_mode;
}


// owner-computes pass over the pixels [px0, px0 + n) of the map, on `st`
void reduce_apply_range(int64_t px0, int64_t n, int64_t nnz, const double * d_cov, double * d_map, int reduce, int slot,
                        hipStream_t st) {
    if (n <= 0) return;
    const Shard s = shard_of(n);
    const int64_t ncov = nnz * (nnz + 1) / 2;
    double * part = d_map + px0 * nnz;
    double * work = part;
    if (!s.even) work = padded_copy(part, n, nnz, s, slot, st);
    double * mine = work + (int64_t)g_rank * s.per * nnz;
    if (reduce) {
        check(rccl().reduce_scatter(work, mine, (size_t)(s.per * nnz), ncclFloat64, ncclSum, comm(), st), "ncclReduceScatter");
    }
    if (d_cov != nullptr && s.count > 0) {
        const int rc = toast_hip_cov_apply_diag_dev(1, s.count, nnz, d_cov + (px0 + s.first) * ncov, mine, st);
        if (rc != 0) throw Error(rc, toast_hip_last_error());
    }
    check(rccl().all_gather(mine, work, (size_t)(s.per * nnz), ncclFloat64, comm(), st), "ncclAllGather");
    if (work != part) TH_HIP(hipMemcpyAsync(part, work, (size_t)n * nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
}


// ------------------------------------------------------------------------------------ mode "peer"
// Every rank owns one driver allocation [out | inbox]: `out` holds the finished slice this rank owns (cap_v doubles),
// `inbox` one slot per rank for their contributions to that slice.  All ranks open all allocations (hipIpc), so a kernel
// can address any of them.  One reduction, everything on the caller's stream:
//   push     slice s (s != me) of my map -> inbox of rank s, slot me            (remote writes over all links at once)
//   barrier  (every contribution has landed: kernel boundaries + a one-word all-reduce)
//   sum      my slice <- rank 0's + rank 1's + ... in that order (mine read from the map), then C . slice, copy to `out`
//   barrier  (every owner's `out` is complete)
//   pull     slice s (s != me) of my map <- out of rank s                        (remote reads over all links at once)
// Nothing of reduction e + 1 can overtake reduction e: a peer writes my inbox again only after its pull, which follows the
// second barrier, which I enter after my sum; and I rewrite `out` only after the first barrier of e + 1, which every
// peer enters after its pull of e.
constexpr int kPeerMax = 16;

struct PeerTable {
    double * p[kPeerMax];
};

// Everything another agent writes or reads goes through system-scope accesses (sc0 sc1: past the caches that are not
// coherent between agents), on top of the kernel boundaries and barriers between the phases: every value is used once,
// so there is nothing for a cache to give.
__device__ inline double peer_load(const double * p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void peer_store(double * p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// W = 8: one double per lane and access, system-scope atomic loads / stores (above).  W = 16 (TOAST_HIP_COMM_PEER_WIDTH=16 /
// toast_hip_comm_set_peer_width): two doubles per lane as ONE ordinary 16-byte access (non-temporal: streamed past the
// caches) -- the request width every other kernel of the library uses; what orders it against the other agents are the
// kernel boundaries and the barriers between the phases.  Needs even slices (per_v % 2 == 0) and 16-byte aligned maps;
// anything else takes W = 8.  Same values either way (plain copies; the sum keeps its rank order).
typedef double peer_d2 __attribute__((ext_vector_type(2)));
template <int W>
struct PeerAccess;
template <>
struct PeerAccess<8> {
    typedef double T;
    static __device__ inline T load_remote(const T * p) { return peer_load(p); }
    static __device__ inline void store_remote(T * p, T v) { peer_store(p, v); }
};
template <>
struct PeerAccess<16> {
    typedef peer_d2 T;
    static __device__ inline T load_remote(const T * p) { return __builtin_nontemporal_load(p); }
    static __device__ inline void store_remote(T * p, T v) { __builtin_nontemporal_store(v, p); }
};

template <int W>
__global__ void __launch_bounds__(256) k_peer_push(const double * __restrict__ map, PeerTable inbox, int rank, int64_t per_v,
                                                   int64_t n_v) {
    typedef typename PeerAccess<W>::T T;
    constexpr int E = W / 8;
    const int s = blockIdx.y;          // owner of the slice
    if (s == rank) return;
    const int64_t first = (int64_t)s * per_v;
    int64_t cnt = n_v - first;
    if (cnt <= 0) return;
    if (cnt > per_v) cnt = per_v;
    const T * src = reinterpret_cast<const T *>(map + first);
    T * dst = reinterpret_cast<T *>(inbox.p[s] + (int64_t)rank * per_v);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t whole = cnt / E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < whole; i += stride) {
        PeerAccess<W>::store_remote(dst + i, src[i]);
    }
    if (E > 1 && (cnt % E) != 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        peer_store(inbox.p[s] + (int64_t)rank * per_v + cnt - 1, map[first + cnt - 1]);
    }
    // W = 16: ordinary stores -- a system-scope release pushes them past this agent's L2 before the kernel ends and the
    // barrier tells the owner (the 8-byte form does that per access)
    if (W == 16) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

template <int W>
__global__ void __launch_bounds__(256) k_peer_sum(double * __restrict__ mine, const double * inbox, int size, int rank,
                                                  int64_t per_v, int64_t cnt) {
    typedef typename PeerAccess<W>::T T;
    constexpr int E = W / 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t whole = cnt / E;
    T * m = reinterpret_cast<T *>(mine);
    // W = 16: ordinary loads of an inbox that OTHER agents wrote -- a system-scope acquire drops whatever line of it this
    // agent's caches still hold from the previous reduction (nothing may rest on the buffer staying uncached)
    if (W == 16) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < whole; i += stride) {
        T acc = (rank == 0) ? m[i] : PeerAccess<W>::load_remote(reinterpret_cast<const T *>(inbox) + i);
        for (int p = 1; p < size; ++p) {
            acc += (p == rank) ? m[i] : PeerAccess<W>::load_remote(reinterpret_cast<const T *>(inbox + (int64_t)p * per_v) + i);
        }
        m[i] = acc;
    }
    if (E > 1 && (cnt % E) != 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = cnt - 1;
        double acc = (rank == 0) ? mine[i] : peer_load(inbox + i);
        for (int p = 1; p < size; ++p) acc += (p == rank) ? mine[i] : peer_load(inbox + (int64_t)p * per_v + i);
        mine[i] = acc;
    }
}

template <int W>
__global__ void __launch_bounds__(256) k_peer_pull(double * __restrict__ map, PeerTable out, int rank, int64_t per_v,
                                                   int64_t n_v) {
    typedef typename PeerAccess<W>::T T;
    constexpr int E = W / 8;
    const int s = blockIdx.y;
    if (s == rank) return;
    const int64_t first = (int64_t)s * per_v;
    int64_t cnt = n_v - first;
    if (cnt <= 0) return;
    if (cnt > per_v) cnt = per_v;
    const T * src = reinterpret_cast<const T *>(out.p[s]);
    T * dst = reinterpret_cast<T *>(map + first);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t whole = cnt / E;
    if (W == 16) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (as in k_peer_sum: the owners' `out` is foreign memory)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < whole; i += stride) {
        dst[i] = PeerAccess<W>::load_remote(src + i);
    }
    if (E > 1 && (cnt % E) != 0 && blockIdx.x == 0 && threadIdx.x == 0) map[first + cnt - 1] = peer_load(out.p[s] + cnt - 1);
}

// a wait that gave up (k_peer_wait) leaves garbage in the map: make it unmistakable -- NaN in the first values -- for
// whoever uses the result before the host has seen the error word
__global__ void k_peer_poison(double * __restrict__ map, int64_t n_v, const int * error) {
    if (__hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) return;
    const int64_t i = threadIdx.x;
    if (i < n_v) map[i] = __builtin_nan("");
}

struct FlagTable {
    unsigned long long * p[kPeerMax];
};

// flag block of a rank: word [slot][source rank], slot 0 = "my contributions have landed", slot 1 = "my slice is finished"
__global__ void k_peer_signal(FlagTable f, int size, int rank, int slot, unsigned long long epoch) {
    const int p = threadIdx.x;
    if (p >= size || p == rank) return;
    __threadfence_system();
    __hip_atomic_store(f.p[p] + slot * kPeerMax + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane per peer; gives up after `limit` ticks of the 100 MHz clock (a rank that never arrives must not park a kernel
// on the GPU for ever): the host sees *error != 0 at its next call and raises
__global__ void k_peer_wait(const unsigned long long * mine, int size, int rank, int slot, unsigned long long epoch,
                            long long limit, int * error) {
    const int p = threadIdx.x;
    if (p < size && p != rank) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(mine + slot * kPeerMax + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > limit) {
                __hip_atomic_store(error, 1 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __threadfence_system();
}

struct PeerExchange {
    unsigned long long * flags[kPeerMax] = {};   // everybody's flag block (uncached memory), flags[g_rank] is mine
    unsigned long long epoch = 0;                // number of the last barrier pair handed out
    int * h_error = nullptr;                     // host-mapped word the waiting kernel writes on a time-out
    long long wait_ticks = 0;
    int64_t cap_v = 0;                 // doubles per slot
    char * base = nullptr;             // my allocation
    char * peer[kPeerMax] = {};        // everybody's (peer[g_rank] == base)
    unsigned char * d_words = nullptr; // device words: [0, 64 * kPeerMax) the handles, then the barrier / agreement word
    int64_t reductions = 0, establishments = 0;
    bool fine = false;                 // the exchange buffer is fine-grained memory (TOAST_HIP_COMM_PEER_MEM=fine)
    int width = 0;                     // bytes per lane and access of push / sum / pull: 8 or 16 (0: not read yet)
    size_t out_bytes() const { return ((size_t)cap_v * sizeof(double) + 255) & ~(size_t)255; }
    double * out_of(int r) const { return reinterpret_cast<double *>(peer[r]); }
    double * inbox_of(int r) const { return reinterpret_cast<double *>(peer[r] + out_bytes()); }
};
PeerExchange g_peer;

void peer_barrier(hipStream_t st) {
    int * word = reinterpret_cast<int *>(g_peer.d_words + 64 * kPeerMax);
    check(rccl().all_reduce(word, word, 1, ncclInt32, ncclMax, comm(), st), "ncclAllReduce (barrier)");
}

bool peer_agree(bool ok, hipStream_t st);

// barrier `slot` of reduction `epoch` by flags: tell every peer, wait for every peer
void peer_flag_barrier(int slot, unsigned long long epoch, hipStream_t st) {
    FlagTable f;
    for (int r = 0; r < kPeerMax; ++r) f.p[r] = r < g_size ? g_peer.flags[r] : nullptr;
    hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, st, f, g_size, g_rank, slot, epoch);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, g_peer.flags[g_rank], g_size, g_rank, slot, epoch,
                       g_peer.wait_ticks, g_peer.h_error);
    TH_HIP(hipGetLastError());
}

void peer_check_error() {
    if (g_peer.h_error != nullptr && *g_peer.h_error != 0) {
        const int who = *g_peer.h_error - 1;
        *g_peer.h_error = 0;
        throw Error(TOAST_HIP_ERR_DEVICE, "HipComm:  mode 'peer:flags': rank " + std::to_string(g_rank) +
                                              " waited in vain for the flag of rank " + std::to_string(who) +
                                              " (TOAST_HIP_COMM_PEER_TIMEOUT_MS); the maps of this reduction are not valid");
    }
}

// collective, once per communicator: every rank's flag block, opened by everybody
void peer_establish_flags(hipStream_t st) {
    if (g_peer.flags[g_rank] != nullptr) return;
    const char * t = std::getenv("TOAST_HIP_COMM_PEER_TIMEOUT_MS");
    const double ms = (t != nullptr && t[0] != '\0') ? std::atof(t) : 60000.0;
    g_peer.wait_ticks = (long long)(ms * 1.0e5);          // 100 MHz
    hipIpcMemHandle_t all[kPeerMax];
    std::memset(all, 0, sizeof(all));
    void * mine = nullptr;
    bool ok = hipExtMallocWithFlags(&mine, 4096, hipDeviceMallocUncached) == hipSuccess;
    if (ok) ok = hipMemsetAsync(mine, 0, 4096, st) == hipSuccess;
    if (ok) ok = hipIpcGetMemHandle(&all[g_rank], mine) == hipSuccess;
    if (ok && g_peer.h_error == nullptr) {
        ok = hipHostMalloc(reinterpret_cast<void **>(&g_peer.h_error), sizeof(int), hipHostMallocMapped) == hipSuccess;
        if (ok) *g_peer.h_error = 0;
    }
    (void)hipGetLastError();
    TH_HIP(hipMemcpyAsync(g_peer.d_words + 64 * g_rank, &all[g_rank], 64, hipMemcpyHostToDevice, st));
    check(rccl().all_gather(g_peer.d_words + 64 * g_rank, g_peer.d_words, 64, ncclUint8, comm(), st), "ncclAllGather (handles)");
    TH_HIP(hipMemcpyAsync(all, g_peer.d_words, 64 * (size_t)g_size, hipMemcpyDeviceToHost, st));
    TH_HIP(hipStreamSynchronize(st));
    ok = peer_agree(ok, st);
    if (ok) {
        g_peer.flags[g_rank] = static_cast<unsigned long long *>(mine);
        for (int r = 0; r < g_size && ok; ++r) {
            if (r == g_rank) continue;
            void * p = nullptr;
            ok = hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
            g_peer.flags[r] = ok ? static_cast<unsigned long long *>(p) : nullptr;
        }
        (void)hipGetLastError();
        ok = peer_agree(ok, st);
    }
    if (!ok) {
        for (int r = 0; r < g_size; ++r) {
            if (r != g_rank && g_peer.flags[r] != nullptr) (void)hipIpcCloseMemHandle(g_peer.flags[r]);
            g_peer.flags[r] = nullptr;
        }
        peer_barrier(st);
        TH_HIP(hipStreamSynchronize(st));
        if (mine != nullptr) (void)hipFree(mine);
        throw Error(TOAST_HIP_ERR_DEVICE,
                    "HipComm:  mode 'peer:flags': the ranks could not open each other's flag blocks (uncached memory over "
                    "hipIpc); use TOAST_HIP_COMM_MODE=peer or owner");
    }
}

void peer_release_flags() {
    if (g_peer.flags[g_rank < 0 ? 0 : g_rank] == nullptr) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < g_size; ++r) {
        if (r != g_rank && g_peer.flags[r] != nullptr) (void)hipIpcCloseMemHandle(g_peer.flags[r]);
    }
    if (g_comm != nullptr && g_peer.d_words != nullptr) {
        peer_barrier(nullptr);
        (void)hipStreamSynchronize(nullptr);
    }
    (void)hipFree(g_peer.flags[g_rank]);
    for (int r = 0; r < kPeerMax; ++r) g_peer.flags[r] = nullptr;
}

// all ranks: does everybody say yes?  (a rank that failed to open a handle must not leave the others inside a barrier)
bool peer_agree(bool ok, hipStream_t st) {
    int * word = reinterpret_cast<int *>(g_peer.d_words + 64 * kPeerMax) + 1;
    int v = ok ? 1 : 0;
    TH_HIP(hipMemcpyAsync(word, &v, sizeof(int), hipMemcpyHostToDevice, st));
    check(rccl().all_reduce(word, word, 1, ncclInt32, ncclMin, comm(), st), "ncclAllReduce (agreement)");
    TH_HIP(hipMemcpyAsync(&v, word, sizeof(int), hipMemcpyDeviceToHost, st));
    TH_HIP(hipStreamSynchronize(st));
    return v == 1;
}

// collective: close what is open, free what is mine
void peer_teardown(hipStream_t st) {
    if (g_peer.base == nullptr) return;
    TH_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < g_size; ++r) {
        if (r != g_rank && g_peer.peer[r] != nullptr) (void)hipIpcCloseMemHandle(g_peer.peer[r]);
        g_peer.peer[r] = nullptr;
    }
    if (g_comm != nullptr && g_peer.d_words != nullptr) {
        peer_barrier(st);              // nobody frees what somebody still has open
        TH_HIP(hipStreamSynchronize(st));
    }
    (void)hipFree(g_peer.base);
    g_peer.base = nullptr;
    g_peer.cap_v = 0;
}

// collective: every rank arrives with the same need_v (the map is replicated)
void peer_establish(int64_t need_v, hipStream_t st) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    if (g_peer.d_words == nullptr) {
        TH_HIP(hipMalloc(reinterpret_cast<void **>(&g_peer.d_words), 64 * kPeerMax + 64));
        TH_HIP(hipMemsetAsync(g_peer.d_words, 0, 64 * kPeerMax + 64, st));
    }
    peer_teardown(st);
    g_peer.cap_v = (need_v + 31) / 32 * 32;
    const size_t bytes = g_peer.out_bytes() * (size_t)(1 + g_size);
    hipIpcMemHandle_t all[kPeerMax];
    // TOAST_HIP_COMM_PEER_MEM = coarse (default: an ordinary hipMalloc; visibility between the agents rests on the
    // system-scope accesses / fences of the kernels and on the kernel boundaries around the barriers) | fine
    // (hipDeviceMallocFinegrained: coherent between agents by construction, slower for the owner's own passes) -- so that
    // the first run on several GPUs can A/B the memory kind instead of debugging it (VERDICT round 5, item 7 a)
    g_peer.fine = false;
    if (const char * e = std::getenv("TOAST_HIP_COMM_PEER_MEM")) g_peer.fine = std::strcmp(e, "fine") == 0;
    bool ok = g_peer.fine
                  ? hipExtMallocWithFlags(reinterpret_cast<void **>(&g_peer.base), bytes, hipDeviceMallocFinegrained) == hipSuccess
                  : hipMalloc(reinterpret_cast<void **>(&g_peer.base), bytes) == hipSuccess;
    if (!ok) g_peer.base = nullptr;
    std::memset(all, 0, sizeof(all));
    if (ok) ok = hipIpcGetMemHandle(&all[g_rank], g_peer.base) == hipSuccess;
    (void)hipGetLastError();
    TH_HIP(hipMemcpyAsync(g_peer.d_words + 64 * g_rank, &all[g_rank], 64, hipMemcpyHostToDevice, st));
    check(rccl().all_gather(g_peer.d_words + 64 * g_rank, g_peer.d_words, 64, ncclUint8, comm(), st), "ncclAllGather (handles)");
    TH_HIP(hipMemcpyAsync(all, g_peer.d_words, 64 * (size_t)g_size, hipMemcpyDeviceToHost, st));
    TH_HIP(hipStreamSynchronize(st));
    ok = peer_agree(ok, st);           // everybody has an allocation and a handle
    if (ok) {
        g_peer.peer[g_rank] = g_peer.base;
        for (int r = 0; r < g_size && ok; ++r) {
            if (r == g_rank) continue;
            void * p = nullptr;
            ok = hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess) == hipSuccess;
            g_peer.peer[r] = ok ? static_cast<char *>(p) : nullptr;
        }
        (void)hipGetLastError();
        ok = peer_agree(ok, st);
    }
    if (!ok) {
        peer_teardown(st);
        throw Error(TOAST_HIP_ERR_DEVICE,
                    "HipComm:  mode 'peer': the ranks could not open each other's exchange buffers (hipIpc); use "
                    "TOAST_HIP_COMM_MODE=owner");
    }
    ++g_peer.establishments;
}

bool peer_error_pending() {
    if (g_peer.h_error == nullptr) return false;
    (void)hipDeviceSynchronize();
    const bool e = *g_peer.h_error != 0;
    *g_peer.h_error = 0;
    return e;
}

void peer_release() {
    peer_release_flags();
    peer_teardown(nullptr);
    if (g_peer.d_words != nullptr) (void)hipFree(g_peer.d_words);
    g_peer.d_words = nullptr;
}

void peer_reduce_apply(int64_t n_px, int64_t nnz, const double * d_cov, double * d_map, int reduce, bool flags,
                       hipStream_t st) {
    if (g_size > kPeerMax) fail_arg("HipComm:  mode 'peer' serves at most 16 ranks");
    peer_check_error();
    const Shard s = shard_of(n_px);
    const int64_t ncov = nnz * (nnz + 1) / 2;
    const int64_t per_v = s.per * nnz, n_v = n_px * nnz, cnt_v = s.count * nnz;
    double * mine = d_map + s.first * nnz;
    if (g_size == 1) {
        if (d_cov != nullptr) {
            const int rc = toast_hip_cov_apply_diag_dev(1, n_px, nnz, d_cov, d_map, st);
            if (rc != 0) throw Error(rc, toast_hip_last_error());
        }
        return;
    }
    if (per_v > g_peer.cap_v || g_peer.base == nullptr) peer_establish(per_v, st);
    if (flags) peer_establish_flags(st);
    const unsigned long long epoch = ++g_peer.epoch;
    const unsigned gx = (unsigned)std::min<int64_t>((per_v + 255) / 256, 4096);
    const dim3 grid(gx > 0 ? gx : 1, (unsigned)g_size);
    if (g_peer.width == 0) {
        const char * e = std::getenv("TOAST_HIP_COMM_PEER_WIDTH");
        g_peer.width = (e != nullptr && std::atoi(e) == 16) ? 16 : 8;
    }
    const bool wide = g_peer.width == 16 && (per_v % 2) == 0 && (reinterpret_cast<uintptr_t>(d_map) % 16) == 0;
    PeerTable tab;
    if (reduce) {
        for (int r = 0; r < kPeerMax; ++r) tab.p[r] = r < g_size ? g_peer.inbox_of(r) : nullptr;
        if (wide) hipLaunchKernelGGL(k_peer_push<16>, grid, dim3(256), 0, st, d_map, tab, g_rank, per_v, n_v);
        else hipLaunchKernelGGL(k_peer_push<8>, grid, dim3(256), 0, st, d_map, tab, g_rank, per_v, n_v);
        TH_HIP(hipGetLastError());
    }
    // The first barrier is passed in EVERY call, also when nothing is summed (reduce = 0): it is what keeps a fast rank
    // from rewriting its `out` below while a slower peer still pulls the previous call's result from it (ADVICE round 4:
    // without it a reduce = 0 call after any other call could hand that peer slices of two different maps).
    if (flags) peer_flag_barrier(0, epoch, st);
    else peer_barrier(st);
    if (reduce && cnt_v > 0) {
        const unsigned gs = (unsigned)std::min<int64_t>((cnt_v + 255) / 256, 8192);
        if (wide) {
            hipLaunchKernelGGL(k_peer_sum<16>, dim3(gs), dim3(256), 0, st, mine, g_peer.inbox_of(g_rank), g_size, g_rank, per_v, cnt_v);
        } else {
            hipLaunchKernelGGL(k_peer_sum<8>, dim3(gs), dim3(256), 0, st, mine, g_peer.inbox_of(g_rank), g_size, g_rank, per_v, cnt_v);
        }
        TH_HIP(hipGetLastError());
    }
    if (cnt_v > 0) {
        if (d_cov != nullptr) {
            const int rc = toast_hip_cov_apply_diag_dev(1, s.count, nnz, d_cov + s.first * ncov, mine, st);
            if (rc != 0) throw Error(rc, toast_hip_last_error());
        }
        TH_HIP(hipMemcpyAsync(g_peer.out_of(g_rank), mine, (size_t)cnt_v * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    if (flags) peer_flag_barrier(1, epoch, st);
    else peer_barrier(st);
    for (int r = 0; r < kPeerMax; ++r) tab.p[r] = r < g_size ? g_peer.out_of(r) : nullptr;
    if (wide) hipLaunchKernelGGL(k_peer_pull<16>, grid, dim3(256), 0, st, d_map, tab, g_rank, per_v, n_v);
    else hipLaunchKernelGGL(k_peer_pull<8>, grid, dim3(256), 0, st, d_map, tab, g_rank, per_v, n_v);
    if (flags && g_peer.h_error != nullptr) {
        hipLaunchKernelGGL(k_peer_poison, dim3(1), dim3(64), 0, st, d_map, n_v, g_peer.h_error);
    }
    TH_HIP(hipGetLastError());
    ++g_peer.reductions;
}

bool peer_allreduce(void * d_buf, int64_t count, int dtype, int op, hipStream_t st) {
    const CommMode & m = mode();
    // (maps, not the PCG's scalar sums: those stay one small RCCL call)
    if (m.kind != 3 || dtype != TOAST_HIP_COMM_F64 || op != TOAST_HIP_COMM_SUM || g_size < 2 || g_size > kPeerMax ||
        count < (int64_t)4096 * g_size) {
        return false;
    }
    peer_reduce_apply(count, 1, nullptr, static_cast<double *>(d_buf), 1, m.slices != 0, st);
    return true;
}

}  // namespace

extern "C" {

int toast_hip_comm_set_mode(const char * text) {
    return guarded([&] {
        const CommMode m = parse_mode(text);
        g_mode = m;
        g_mode_read = true;
        peer_check_error();     // a wait of the mode that is being left may have given up in its last reduction
    });
}

int toast_hip_comm_set_peer_width(int bytes) {
    return guarded([&] {
        if (bytes != 8 && bytes != 16) fail_arg("HipComm:  the peer exchange moves 8 or 16 bytes per lane");
        g_peer.width = bytes;
    });
}

// The error word of mode "peer:flags" (a wait that gave up), raised where the host next synchronises with the stream the
// reduction ran on -- not only at the start of the next reduction, which may never come (the last binned map, the last
// PCG iteration): toast_hip_comm_check(), toast_hip_comm_set_mode(), toast_hip_comm_destroy().
int toast_hip_comm_check(void * stream) {
    return guarded([&] {
        if (g_peer.h_error == nullptr) return;
        TH_HIP(hipStreamSynchronize(as_stream(stream)));
        peer_check_error();
    });
}

int toast_hip_comm_peer_stats(int64_t * reductions, int64_t * establishments, int64_t * exchange_bytes) {
    return guarded([&] {
        if (reductions) *reductions = g_peer.reductions;
        if (establishments) *establishments = g_peer.establishments;
        if (exchange_bytes) *exchange_bytes = g_peer.base ? (int64_t)(g_peer.out_bytes() * (size_t)(1 + g_size)) : 0;
    });
}

int toast_hip_comm_peer_mem(int * fine_grained) {
    return guarded([&] {
        if (fine_grained) *fine_grained = (g_peer.base != nullptr && g_peer.fine) ? 1 : 0;
    });
}

int toast_hip_comm_get_mode(char * text, size_t len) {
    return guarded([&] {
        const CommMode & m = mode();
        const std::string v = m.kind == 0   ? "owner"
                              : m.kind == 2 ? "allreduce"
                                            : (m.slices ? "peer:flags" : "peer");
        if (text == nullptr || len < v.size() + 1) fail_arg("HipComm:  mode buffer too small");
        std::memcpy(text, v.c_str(), v.size() + 1);
    });
}

int toast_hip_comm_map_reduce_apply_dev(int64_t n_px, int64_t nnz, const double * d_cov, double * d_map, int reduce,
                                        void * stream) {
    return guarded([&] {
        if (n_px <= 0) return;
        if (nnz <= 0) fail_arg("nnz must be positive");
        (void)comm();
        hipStream_t st = as_stream(stream);
        const CommMode & m = mode();
        if (m.kind == 2) {
            if (reduce) {
                check(rccl().all_reduce(d_map, d_map, (size_t)(n_px * nnz), ncclFloat64, ncclSum, comm(), st), "ncclAllReduce");
            }
            if (d_cov != nullptr) {
                const int rc = toast_hip_cov_apply_diag_dev(1, n_px, nnz, d_cov, d_map, stream);
                if (rc != 0) throw Error(rc, toast_hip_last_error());
            }
            return;
        }
        if (m.kind == 3) {
            peer_reduce_apply(n_px, nnz, d_cov, d_map, reduce, m.slices != 0, st);
            return;
        }
        reduce_apply_range(0, n_px, nnz, d_cov, d_map, reduce, Manager::kScratchCommA, st);
    });
}

// In-place inverse of the per-pixel blocks, owner computes (covariance_invert(use_alltoallv=True), covariance.py:34-131):
// every rank decomposes its pixel shard of the (replicated) matrix, then the shards are gathered; same for the
// condition-number map when given.
int toast_hip_comm_cov_invert_dev(int64_t n_px, int64_t nnz, double * d_cov, double * d_rcond, double threshold,
                                  int invert, void * stream) {
    return guarded([&] {
        if (n_px <= 0) return;
        (void)comm();
        hipStream_t st = as_stream(stream);
        const Shard s = shard_of(n_px);
        const int64_t ncov = nnz * (nnz + 1) / 2;
        if (s.count > 0) {
            const int rc = toast_hip_cov_eigendecompose_diag_dev(1, s.count, nnz, d_cov + s.first * ncov,
                                                                 d_rcond ? d_rcond + s.first : nullptr, threshold, invert,
                                                                 stream);
            if (rc != 0) throw Error(rc, toast_hip_last_error());
        }
        if (invert) all_gather_pixels(d_cov, n_px, ncov, s, Manager::kScratchCommA, st);
        if (d_rcond != nullptr) all_gather_pixels(d_rcond, n_px, 1, s, Manager::kScratchCommB, st);
    });
}

// cov1 <- cov1 . cov2 per pixel, owner computes (covariance_multiply(use_alltoallv=True), covariance.py:134-221).
int toast_hip_comm_cov_mult_dev(int64_t n_px, int64_t nnz, double * d_cov1, const double * d_cov2, void * stream) {
    return guarded([&] {
        if (n_px <= 0) return;
        (void)comm();
        hipStream_t st = as_stream(stream);
        const Shard s = shard_of(n_px);
        const int64_t ncov = nnz * (nnz + 1) / 2;
        if (s.count > 0) {
            const int rc = toast_hip_cov_mult_diag_dev(1, s.count, nnz, d_cov1 + s.first * ncov, d_cov2 + s.first * ncov,
                                                       stream);
            if (rc != 0) throw Error(rc, toast_hip_last_error());
        }
        all_gather_pixels(d_cov1, n_px, ncov, s, Manager::kScratchCommA, st);
    });
}

}  // extern "C"
