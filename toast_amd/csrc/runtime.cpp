// runtime.cpp -- see runtime.hpp.  Reference counterpart of the memory manager:
// /root/reference/src/toast/_libtoast/accelerator.cpp:233-766 (OmpManager).
#include "runtime.hpp"

#include <chrono>
#include <condition_variable>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <cstdio>
#include <cstdlib>
#include <list>
#include <mutex>
#include <vector>

namespace toast_hip {

namespace {
thread_local std::string g_last_error;
}

void set_last_error(const std::string & msg) { g_last_error = msg; }

[[noreturn]] void fail_arg(const std::string & msg) { throw Error(TOAST_HIP_ERR_ARG, msg); }

// ------------------------------------------------------------------ chunks
int chunk_size() {
    // Tuning knob for experiments (profiles/): samples per workgroup, default kChunk.
    static const int v = [] {
        const char * e = std::getenv("TOAST_HIP_CHUNK");
        const int c = e ? std::atoi(e) : 0;
        return (c >= 64 && c <= (1 << 20)) ? c : kChunk;
    }();
    return v;
}

namespace {
int g_det_major = -1;   // -1: environment
}

bool det_major_grid() {
    // Experiment switch: detector-major instead of time-major workgroup order.
    if (g_det_major < 0) {
        const char * e = std::getenv("TOAST_HIP_DET_MAJOR");
        g_det_major = (e && e[0] == '1') ? 1 : 0;
    }
    return g_det_major == 1;
}

void set_det_major_grid(int on) { g_det_major = on ? 1 : 0; }

namespace {
int g_pair = -1;   // -1: not read yet
}
bool pair_detectors() {
    // TOAST_HIP_PAIR=0 disables the detector-pair kernels; toast_hip_set_tuning("pair", v) at run time.
    if (g_pair < 0) {
        const char * e = std::getenv("TOAST_HIP_PAIR");
        g_pair = (e && e[0] == '0') ? 0 : 1;
    }
    return g_pair != 0;
}
void set_pair_detectors(int on) { g_pair = on ? 1 : 0; }

namespace {
int g_vec2 = -1;   // -1: not read yet
}
bool vec2_lanes() {
    // Two consecutive samples per lane (16-byte lane accesses) in scan_map / build_noise_weighted / noise_weight
    // (DESIGN.md §4 "16 bytes per lane"); TOAST_HIP_VEC2=0 selects the one-sample-per-lane kernels;
    // toast_hip_set_tuning("vec2", v) at run time.
    if (g_vec2 < 0) {
        const char * e = std::getenv("TOAST_HIP_VEC2");
        g_vec2 = (e && e[0] == '0') ? 0 : 1;
    }
    return g_vec2 != 0;
}
void set_vec2_lanes(int on) { g_vec2 = on ? 1 : 0; }

std::vector<Chunk> make_chunks(const toast_hip_interval * ivl, int64_t n_view, int64_t n_samp) {
    std::vector<Chunk> out;
    const int64_t kc = chunk_size();
    for (int64_t v = 0; v < n_view; ++v) {
        const int64_t first = ivl[v].first;
        const int64_t last = ivl[v].last;
        if (first < 0 || last > n_samp) {
            std::ostringstream o;
            o << "interval " << v << " = [" << first << ", " << last << ") is outside the "
              << n_samp << " samples of the buffers";
            fail_arg(o.str());
        }
        for (int64_t s = first; s < last; s += kc) {
            const int64_t n = (last - s < kc) ? (last - s) : kc;
            out.push_back(Chunk{s, (int32_t)n, (int32_t)v});
        }
    }
    return out;
}

// ------------------------------------------------------------------ parameter blocks
size_t ParamBlock::push(const void * src, size_t bytes) {
    size_t off = (host_.size() + 15) & ~size_t(15);
    host_.resize(off + bytes);
    if (bytes) std::memcpy(host_.data() + off, src, bytes);
    return off;
}

namespace {

struct CachedBlock {
    std::vector<char> host;
    char * dev;
    int device;
};

// Most-recently-used list of uploaded blocks; bounded in bytes.  Device storage is carved out of
// 8 MB slabs from the manager's arena (bump allocation): committing a new block costs one async copy, never a
// hipMalloc or a synchronisation on the launch path.  When the cache is full everything is dropped at once
// (one device synchronisation per ~256 MB of distinct parameter blocks).
std::list<CachedBlock> g_blocks;
size_t g_block_bytes = 0;
constexpr size_t kBlockCacheBytes = size_t(256) << 20;
constexpr size_t kSlabBytes = size_t(8) << 20;
struct Slab {
    char * base;
    size_t bytes;
    size_t used;
    int device;
};
std::vector<Slab> g_slabs;
std::mutex g_block_mutex;

char * slab_alloc(size_t bytes, int device) {
    const size_t need = (bytes + 255) & ~size_t(255);
    for (auto & s : g_slabs) {
        if (s.device == device && s.used + need <= s.bytes) {
            char * p = s.base + s.used;
            s.used += need;
            return p;
        }
    }
    Slab s{nullptr, need > kSlabBytes ? need : kSlabBytes, 0, device};
    void * p = Manager::get().device_alloc(s.bytes);
    if (p == nullptr) throw Error(TOAST_HIP_ERR_MEMORY, "HipManager:  parameter block storage, allocation failed");
    s.base = static_cast<char *>(p);
    s.used = need;
    g_slabs.push_back(s);
    return s.base;
}

void drop_all_blocks() {
    // Blocks still referenced by queued kernels must outlive them.
    (void)hipDeviceSynchronize();
    for (auto & s : g_slabs) Manager::get().device_free(s.base);
    g_slabs.clear();
    g_blocks.clear();
    g_block_bytes = 0;
}

}  // namespace

void drop_param_blocks() {
    std::lock_guard<std::mutex> lock(g_block_mutex);
    drop_all_blocks();
}

const char * ParamBlock::commit(hipStream_t stream) {
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_block_mutex);
    for (auto it = g_blocks.begin(); it != g_blocks.end(); ++it) {
        if (it->device == dev && it->host.size() == host_.size() &&
            std::memcmp(it->host.data(), host_.data(), host_.size()) == 0) {
            g_blocks.splice(g_blocks.begin(), g_blocks, it);
            return g_blocks.front().dev;
        }
    }
    if (!g_blocks.empty() && g_block_bytes + host_.size() > kBlockCacheBytes) drop_all_blocks();
    CachedBlock blk;
    blk.device = dev;
    blk.host = host_;
    blk.dev = slab_alloc(host_.size() ? host_.size() : 16, dev);
    // The cached host copy outlives the asynchronous copy.
    g_blocks.push_front(std::move(blk));
    g_block_bytes += host_.size();
    if (!host_.empty()) copy_to_device(g_blocks.front().dev, g_blocks.front().host.data(), host_.size(), stream);
    return g_blocks.front().dev;
}

// ------------------------------------------------------------------ bounce ring
namespace {

struct BounceRing {
    static constexpr int kSlots = 2;
    static constexpr size_t kSlotBytes = size_t(4) << 20;
    char * slot[kSlots] = {nullptr, nullptr};
    hipEvent_t done[kSlots] = {nullptr, nullptr};
    bool busy[kSlots] = {false, false};
    int device = -1;
    int next = 0;
    std::mutex mutex;

    void prepare() {
        int dev = 0;
        TH_HIP(hipGetDevice(&dev));
        if (slot[0] != nullptr && dev == device) return;
        release();
        for (int k = 0; k < kSlots; ++k) {
            void * p = nullptr;
            TH_HIP(hipHostMalloc(&p, kSlotBytes, hipHostMallocDefault));
            slot[k] = static_cast<char *>(p);
            TH_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            busy[k] = false;
        }
        device = dev;
        next = 0;
    }
    void release() {
        for (int k = 0; k < kSlots; ++k) {
            if (done[k] != nullptr) {
                if (busy[k]) (void)hipEventSynchronize(done[k]);
                (void)hipEventDestroy(done[k]);
            }
            if (slot[k] != nullptr) (void)hipHostFree(slot[k]);
            slot[k] = nullptr;
            done[k] = nullptr;
            busy[k] = false;
        }
    }
    // wait until the device has finished with slot k
    void wait(int k) {
        if (busy[k]) {
            TH_HIP(hipEventSynchronize(done[k]));
            busy[k] = false;
        }
    }
};

BounceRing g_bounce;

// Host -> device for the ring's slots by a KERNEL that reads the page-locked slot over PCIe, instead of a DMA command: the
// DMA engine serves its queue in order, so a few KB of kernel parameters enqueued behind an upload of gigabytes (the
// timestream parts of ops.NoiseFilter on the upload stream) would reach the device -- and release the kernels that wait
// for them -- only after the whole upload (profiles/r04_f).  A kernel's loads share the link with the DMA packet by packet.
__global__ void __launch_bounds__(256) k_slot_to_device(char * __restrict__ dst, const char * __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const size_t n16 = n >> 4;
        const uint4 * s = reinterpret_cast<const uint4 *>(src);
        uint4 * d = reinterpret_cast<uint4 *>(dst);
        for (size_t i = i0; i < n16; i += stride) d[i] = s[i];
        for (size_t i = (n16 << 4) + i0; i < n; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = i0; i < n; i += stride) dst[i] = src[i];
    }
}

bool bounce_by_kernel() {
    static const bool v = [] {
        const char * e = std::getenv("TOAST_HIP_BOUNCE_KERNEL");
        return !(e != nullptr && e[0] == '0');
    }();
    return v;
}

bool bounce_enabled() {
    static const bool v = [] {
        const char * e = std::getenv("TOAST_HIP_BOUNCE");
        return !(e != nullptr && e[0] == '0');
    }();
    return v;
}

}  // namespace

void copy_to_device(void * dev, const void * host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    if (!bounce_enabled()) {
        TH_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream));
        TH_HIP(hipStreamSynchronize(stream));
        return;
    }
    std::lock_guard<std::mutex> lock(g_bounce.mutex);
    g_bounce.prepare();
    const char * src = static_cast<const char *>(host);
    char * dst = static_cast<char *>(dev);
    for (size_t off = 0; off < bytes; off += BounceRing::kSlotBytes) {
        const size_t n = (bytes - off < BounceRing::kSlotBytes) ? bytes - off : BounceRing::kSlotBytes;
        const int k = g_bounce.next;
        g_bounce.next = (k + 1) % BounceRing::kSlots;
        g_bounce.wait(k);
        std::memcpy(g_bounce.slot[k], src + off, n);
        if (bounce_by_kernel()) {
            const unsigned blocks = (unsigned)std::min<size_t>((n / 16 + 255) / 256 + 1, 1024);
            hipLaunchKernelGGL(k_slot_to_device, dim3(blocks), dim3(256), 0, stream, dst + off, g_bounce.slot[k], n);
            TH_HIP(hipGetLastError());
        } else {
            TH_HIP(hipMemcpyAsync(dst + off, g_bounce.slot[k], n, hipMemcpyHostToDevice, stream));
        }
        TH_HIP(hipEventRecord(g_bounce.done[k], stream));
        g_bounce.busy[k] = true;
    }
}

void copy_to_host(void * host, const void * dev, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    if (!bounce_enabled()) {
        TH_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
        TH_HIP(hipStreamSynchronize(stream));
        return;
    }
    std::lock_guard<std::mutex> lock(g_bounce.mutex);
    g_bounce.prepare();
    char * dst = static_cast<char *>(host);
    const char * src = static_cast<const char *>(dev);
    const size_t nchunk = (bytes + BounceRing::kSlotBytes - 1) / BounceRing::kSlotBytes;
    auto chunk_bytes = [&](size_t c) {
        const size_t off = c * BounceRing::kSlotBytes;
        return (bytes - off < BounceRing::kSlotBytes) ? bytes - off : BounceRing::kSlotBytes;
    };
    // slot of chunk c = (first + c) % kSlots; chunk c + 1 is in flight while chunk c is copied out
    const int first = g_bounce.next;
    auto issue = [&](size_t c) {
        const int k = (first + (int)(c % BounceRing::kSlots)) % BounceRing::kSlots;
        g_bounce.wait(k);
        TH_HIP(hipMemcpyAsync(g_bounce.slot[k], src + c * BounceRing::kSlotBytes, chunk_bytes(c), hipMemcpyDeviceToHost,
                              stream));
        TH_HIP(hipEventRecord(g_bounce.done[k], stream));
        g_bounce.busy[k] = true;
    };
    issue(0);
    for (size_t c = 0; c < nchunk; ++c) {
        const int k = (first + (int)(c % BounceRing::kSlots)) % BounceRing::kSlots;
        if (c + 1 < nchunk) issue(c + 1);
        g_bounce.wait(k);
        std::memcpy(dst + c * BounceRing::kSlotBytes, g_bounce.slot[k], chunk_bytes(c));
    }
    g_bounce.next = (first + (int)(nchunk % BounceRing::kSlots)) % BounceRing::kSlots;
}

static bool trace_enabled();

// ------------------------------------------------------------------ manager
Manager & Manager::get() {
    static Manager m;
    return m;
}

size_t pin_threshold() {
    // TOAST_HIP_PIN_HOST_MB: page-lock host buffers of at least this many MiB when they are
    // registered (default 16; 0 disables pinning).
    static const size_t v = [] {
        const char * e = std::getenv("TOAST_HIP_PIN_HOST_MB");
        const long mb = e ? std::atol(e) : 16;
        return (mb <= 0) ? ~size_t(0) : (size_t)mb << 20;
    }();
    return v;
}

// Grow-only scratch buffers owned by the manager (so: per process = per device, released by
// clear()): blocks of the arena.  A failed growth leaves the slot empty -- never a dangling pointer.
void * Manager::scratch(int slot, size_t bytes) { return scratch_impl(slot, bytes, false, nullptr); }

void * Manager::scratch(int slot, size_t bytes, hipStream_t user) { return scratch_impl(slot, bytes, true, user); }

void * Manager::scratch_impl(int slot, size_t bytes, bool named, hipStream_t user) {
    // keyed by the CURRENT device: the device-pointer entry points (toast_hip_*_dev) serve callers
    // that own their device memory and never went through assign_device()
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    const auto key = std::make_pair(dev, slot);
    auto & s = scratch_[key];
    auto known = scratch_user_.find(key);
    const bool all_named = (known == scratch_user_.end()) ? true : known->second.first;
    const hipStream_t last = (known == scratch_user_.end()) ? nullptr : known->second.second;
    scratch_user_[key] = std::make_pair(all_named && named, user);
    if (bytes <= s.second && s.first != nullptr) {
        // a different stream takes the buffer over: its kernels must come after the previous user's
        if (named && all_named && known != scratch_user_.end() && last != user) TH_HIP(hipStreamSynchronize(last));
        return s.first;
    }
    if (s.first != nullptr) {
        if (all_named && known != scratch_user_.end()) {
            TH_HIP(hipStreamSynchronize(last));
        } else {
            TH_HIP(hipDeviceSynchronize());
        }
        void * old = s.first;
        s.first = nullptr;
        s.second = 0;
        device_free(old);
    }
    void * p = device_alloc(bytes ? bytes : 16);
    if (p == nullptr) {
        std::ostringstream o;
        o << "HipManager:  scratch buffer of " << bytes << " bytes on device " << dev
          << ", allocation failed";
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    s.first = p;
    s.second = bytes;
    return p;
}

void Manager::clear() {
    // Everything goes back to the ARENA, nothing to the driver: the next phase, solve or benchmark of this process
    // finds its memory where the last one left it (toast_hip_accel_release_cached returns the slabs).
    if (!table_.empty() || !scratch_.empty()) (void)hipDeviceSynchronize();
    for (auto & kv : scratch_) {
        if (kv.second.first) device_free(kv.second.first);
    }
    scratch_.clear();
    scratch_user_.clear();
    if (upload_stream_ != nullptr) (void)hipStreamSynchronize(upload_stream_);
    for (auto & kv : table_) {
        for (hipEvent_t ev : kv.second.part_done) (void)hipEventDestroy(ev);
        kv.second.part_done.clear();
        unpin(kv.first, kv.second);
        if (kv.second.owned) device_free(kv.second.dev);
    }
    table_.clear();
    owned_bytes_ = 0;
    ++generation_;
}

int Manager::device() {
    if (!assigned_) {
        throw Error(TOAST_HIP_ERR_DEVICE,
                    "HipManager:  device not yet assigned, call assign_device() first");
    }
    return device_;
}

void Manager::require_device() {
    if (device() < 0) {
        throw Error(TOAST_HIP_ERR_DEVICE,
                    "HipManager:  no gfx950 device is assigned to this process (disabled or "
                    "none visible); libtoast_hip has no host implementation");
    }
    TH_HIP(hipSetDevice(device_));
}

Manager::Entry & Manager::lookup(const void * host, size_t nbytes, const char * name,
                                 const char * what) {
    auto it = table_.find(host);
    std::ostringstream o;
    if (it == table_.end()) {
        o << "HipManager:  host ptr " << host << " (name='" << (name ? name : "NA")
          << "') is not present- cannot " << what;
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    if (it->second.nbytes != nbytes) {
        o << "HipManager:  on " << what << ", host ptr " << host << " (name='" << it->second.name
          << "') has " << it->second.nbytes << " bytes instead of " << nbytes;
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    return it->second;
}

int Manager::present(const void * host, size_t nbytes) {
    if (device() < 0) return 0;
    auto it = table_.find(host);
    if (it == table_.end()) return 0;
    if (it->second.nbytes != nbytes) {
        // accelerator.cpp:672-685
        std::ostringstream o;
        o << "HipManager:  host ptr " << host << " is present, but has " << it->second.nbytes
          << " bytes instead of " << nbytes;
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    return 1;
}

// One allocation function for everything the library keeps on the device.
//   TOAST_HIP_ALLOC=arena        (default) ranges of slabs taken once from the driver (arena.hpp)
//   TOAST_HIP_ALLOC=plain        one hipMalloc / hipFree per block (experiments; what rounds 1-2 did)
//   TOAST_HIP_ARENA_SLAB_GB      size of a slab taken on demand (default 16; a larger request gets a slab of its own size)
//   TOAST_HIP_ARENA_RESERVE_GB   taken at assign_device (default: its mem_gb argument / processes per device)
// Blocks below 1 MB live in an arena of their own (64 MB slabs, 512-byte granule) so that they cannot split the
// ranges the timestream-sized blocks need; the large arena works in 2 MB granules (the driver's large-fragment size).
namespace {
struct AllocPolicy {
    bool plain = false;
    size_t slab_bytes = size_t(16) << 30;
};
const AllocPolicy & alloc_policy() {
    static const AllocPolicy pol = [] {
        AllocPolicy a;
        const char * e = std::getenv("TOAST_HIP_ALLOC");
        if (e != nullptr && std::string(e) == "plain") a.plain = true;
        const char * g = std::getenv("TOAST_HIP_ARENA_SLAB_GB");
        if (g != nullptr && std::atof(g) > 0.0) a.slab_bytes = (size_t)(std::atof(g) * 1073741824.0);
        return a;
    }();
    return pol;
}
constexpr size_t kSmallBlock = size_t(1) << 20;
Arena & big_arena() {
    static Arena a(size_t(2) << 20, alloc_policy().slab_bytes, hip_backend(), size_t(1) << 30);
    return a;
}
// written timestreams (device_alloc(streamed = true)): slabs of 8 GB built from 1 GB chunks of two HBM zones
constexpr size_t kStreamBlock = size_t(1) << 30;
Arena & stream_arena() {
    static Arena a(size_t(2) << 20, size_t(8) << 30, hip_interleaved_backend(), size_t(1) << 30);
    return a;
}
bool stream_arena_enabled() {
    static const bool on = [] {
        const char * e = std::getenv("TOAST_HIP_ARENA_INTERLEAVE");
        return !(e != nullptr && e[0] == '0');
    }();
    return on;
}
Arena & small_arena() {
    static Arena a(512, size_t(64) << 20);
    return a;
}
struct DirectStats {
    int64_t mallocs = 0;
    double malloc_ms = 0.0, max_malloc_ms = 0.0;
} g_direct;
}  // namespace

// The streamed slab is built where the process has time for it: on a thread of its own, started by assign_device
// (a caller that only speaks the reference's accel_* API gets the zone placement, and no operator
// waits for the search).  Whoever needs the slab -- a streamed / scatter block, statistics, a trim -- waits for it.
struct Builder {
    std::mutex m;
    std::condition_variable cv;
    bool running = false;
    bool at_exit = false;
} g_builder;

void builder_wait() {
    std::unique_lock<std::mutex> lock(g_builder.m);
    g_builder.cv.wait(lock, [] { return !g_builder.running; });
}

void builder_start(size_t bytes, int dev) {
    builder_wait();
    {
        std::lock_guard<std::mutex> lock(g_builder.m);
        g_builder.running = true;
        if (!g_builder.at_exit) {
            g_builder.at_exit = true;
            std::atexit([] { builder_wait(); });       // nothing of it may run into the runtime's own exit
        }
    }
    std::thread([bytes, dev] {
        hipStream_t st = nullptr;
        if (hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
            try {
                (void)stream_arena().reserve(bytes, st, true);
            } catch (...) {
            }
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(g_builder.m);
            g_builder.running = false;
        }
        g_builder.cv.notify_all();
    }).detach();
}

// Where blocks go inside an interleaved slab (pattern P Q Q P ..., vmm_slot_other):
// a scatter target inside ONE run of chunks of one class (other = the Q chunks) ...
size_t place_in_class(const char * base, size_t lo, size_t hi, size_t need, bool other) {
    size_t chunk = 0;
    if (!vmm_slab_layout(base, &chunk) || chunk == 0) return SIZE_MAX;       // (a slab that fell back to a plain hipMalloc)
    for (size_t k = lo / chunk; k * chunk < hi; ++k) {
        if (vmm_slot_other(k) != other) continue;
        size_t run_end = k;
        while (vmm_slot_other(run_end) == other) ++run_end;
        const size_t a = std::max(lo, k * chunk), b = std::min(hi, run_end * chunk);
        if (b >= a + need) return a;
        k = run_end - 1;
    }
    return SIZE_MAX;
}

// Which class of a slab's chunks the scatter targets belong into.  A slab built against the ends of the read-mostly
// slab (vmm_slab.cpp, n_ref > 0): the Q chunks, by construction.  A slab that assign_device built before any array
// existed knows two classes but not which of them the arrays will share a zone with: decided when the first scatter
// target is asked for -- half a chunk of either class (free memory: the pass rewrites what it reads) against the first
// and the last GB of the read-mostly slab; the class that runs slower with a reference lies in its zone.  Preferred: the
// class that is clear of both ends; second: clear of the START (the arrays that feed the A^T scatter are created first).
// The decision lives in the slab's own record (vmm_slab.cpp: vmm_slab_scatter_class, under its mutex, forgotten when
// the slab is given back -- a later slab at the same address starts undecided).
int scatter_class_of(const char * base, hipStream_t st) {
    const int known = vmm_slab_scatter_class(base);
    if (known >= 0) return known;
    size_t chunk = 0;
    int n_ref = 0;
    if (!vmm_slab_layout(base, &chunk, &n_ref)) return 1;
    if (n_ref > 0) {
        vmm_slab_set_scatter_class(base, 1);
        return 1;
    }
    const size_t half = chunk / 2;
    ZoneRefs refs = zone_references_take(half);
    if (refs.last == nullptr) return 1;            // no read-mostly arrays yet: nothing to keep away from (not cached)
    void * cand[2] = {stream_arena().alloc_placed(half, [](const char * b, size_t lo, size_t hi, size_t need) {
                          return place_in_class(b, lo, hi, need, false);
                      }),
                      stream_arena().alloc_placed(half, [](const char * b, size_t lo, size_t hi, size_t need) {
                          return place_in_class(b, lo, hi, need, true);
                      })};
    int pick = 1;
    if (cand[0] != nullptr && cand[1] != nullptr) {
        void * ref[2] = {refs.first, refs.last};
        const int nr = (refs.first != nullptr && refs.first != refs.last) ? 2 : 1;
        if (nr == 1) ref[0] = refs.last;
        bool same[2][2] = {{false, false}, {false, false}};      // [class][reference]
        try {
            for (int k = 0; k < nr; ++k) {
                double rate[2];
                for (int c = 0; c < 2; ++c) {
                    void * two[2] = {ref[k], cand[c]};
                    const double ms = probe_stream_split_ms(two, 2, half, st);
                    rate[c] = ms > 0.0 ? 1.0 / ms : 0.0;
                }
                if (rate[0] < rate[1] * (1.0 - 0.035)) same[0][k] = true;
                if (rate[1] < rate[0] * (1.0 - 0.035)) same[1][k] = true;
            }
            const bool clear0 = !same[0][0] && !same[0][1], clear1 = !same[1][0] && !same[1][1];
            if (clear1) pick = 1;
            else if (clear0) pick = 0;
            else pick = same[1][0] ? 0 : 1;          // neither is clear of both ends: away from the start (reference 0 = first GB)
        } catch (const Error &) {
            pick = 1;
        }
        vmm_slab_set_scatter_class(base, pick);
        if (trace_enabled()) {
            std::fprintf(stderr, "[toast_hip] scatter targets of the slab at %p go to its %s chunks (P same zone as start/end: %d/%d, Q: %d/%d)\n",
                         (const void *)base, pick ? "Q" : "P", (int)same[0][0], (int)same[0][1], (int)same[1][0], (int)same[1][1]);
        }
    }
    for (void * c : cand) {
        if (c != nullptr) (void)stream_arena().release(c);
    }
    zone_references_release(refs);
    return pick;
}

// ... and a streamed block that is too small to cover both zones wherever it lies (<= 2 chunks): astride a P | Q boundary
size_t place_astride(const char * base, size_t lo, size_t hi, size_t need) {
    size_t chunk = 0;
    if (!vmm_slab_layout(base, &chunk) || chunk == 0) return SIZE_MAX;
    for (size_t k = lo / chunk + 1; k * chunk < hi; ++k) {
        if (vmm_slot_other(k) == vmm_slot_other(k - 1)) continue;          // boundary between slots k - 1 and k
        const size_t edge = k * chunk, half = need / 2;
        size_t a = edge > half ? edge - half : 0;
        if (a < lo) a = lo;
        if (a + need > hi) a = hi > need ? hi - need : SIZE_MAX;
        if (a != SIZE_MAX && a >= lo && a < edge && a + need > edge) return a;
    }
    return SIZE_MAX;
}

AllocStats alloc_stats() {
    builder_wait();
    AllocStats o;
    for (Arena * a : {&big_arena(), &small_arena(), &stream_arena()}) {
        const ArenaStats s = a->stats();
        o.slabs += s.slabs;
        o.slab_bytes += s.slab_bytes;
        o.used_bytes += s.used_bytes;
        o.peak_used_bytes += s.peak_used_bytes;
        o.slab_mallocs += s.slab_mallocs;
        o.slab_frees += s.slab_frees;
        o.malloc_ms += s.malloc_ms;
        o.max_malloc_ms = s.max_malloc_ms > o.max_malloc_ms ? s.max_malloc_ms : o.max_malloc_ms;
        o.touch_ms += s.touch_ms;
        o.allocs += s.allocs;
        o.releases += s.releases;
        o.failed += s.failed;
    }
    o.direct_mallocs = g_direct.mallocs;
    o.malloc_ms += g_direct.malloc_ms;
    o.max_malloc_ms = g_direct.max_malloc_ms > o.max_malloc_ms ? g_direct.max_malloc_ms : o.max_malloc_ms;
    return o;
}

void Manager::assign_device(int node_procs, int node_rank, double mem_gb, bool disabled) {
    // accelerator.cpp:236-246
    if (node_procs < 1 || node_rank < 0) {
        throw Error(TOAST_HIP_ERR_ARG,
                    "HipManager:  must have at least one process per node with a rank >= 0");
    }
    if (node_rank >= node_procs) {
        throw Error(TOAST_HIP_ERR_ARG, "HipManager:  node rank must be < number of node procs");
    }
    clear();
    int n_dev = 0;
    if (!disabled) {
        if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
    }
    if (n_dev == 0) {
        if (device_ >= 0) drop_arenas();
        device_ = -1;
    } else {
        // accelerator.cpp:276-281: ceil(node_procs / n_dev) processes share a device
        int per = node_procs / n_dev;
        if (n_dev * per < node_procs) per += 1;
        const int dev = node_rank / per;
        if (dev != device_) drop_arenas();    // (slabs belong to the device they were taken on)
        device_ = dev;
        TH_HIP(hipSetDevice(device_));
        vmm_set_device_share(per);
        // accelerator.cpp:296-300 (dormant upstream): this process' share of `mem_gb` becomes the pool.  Here the
        // pool can grow past it (a slab per request that does not fit), so the number is a reservation, not a limit:
        // what it covers is taken from the driver -- and touched -- now instead of inside the first operators.
        // TOAST_HIP_ARENA_RESERVE_GB overrides the argument (0 = reserve nothing).
        double gb = mem_gb / (double)per;
        if (const char * e = std::getenv("TOAST_HIP_ARENA_RESERVE_GB")) gb = std::atof(e);
        if (gb > 0.0) reserve((size_t)(gb * 1073741824.0));
        // ... and the slab for the blocks that sweeps read AND write (timestreams) or scatter into (maps, amplitudes),
        // built from chunks of two HBM zones (vmm_slab.cpp): a quarter of a real reservation + 2 GB -- 16 of the ~76 B per
        // detector-sample that a map-making run holds are timestreams that get written (workflows/mapmaker_pcg.py) --, a
        // sixteenth of the free memory with the reference's token mem_gb; between 8 and 48 GB.
        // TOAST_HIP_ARENA_STREAM_GB overrides (0: none).  Built on a thread of its own unless TOAST_HIP_ARENA_BUILDER=0:
        // device_alloc waits for it when it needs it.
        double sgb = -1.0;
        if (const char * e = std::getenv("TOAST_HIP_ARENA_STREAM_GB")) sgb = std::atof(e);
        if (sgb < 0.0 && per > 1) {
            // processes that share a device get no slab by default: the search creates chunks beyond the slab itself for a
            // moment, and several of them side by side are what makes a neighbour's hipMalloc fail;
            // TOAST_HIP_ARENA_STREAM_GB / toast_hip_arena_reserve_streamed still build one
            sgb = 0.0;
        }
        if (sgb < 0.0 && stream_arena_enabled() && !alloc_policy().plain) {
            size_t f = 0, t = 0;
            if (hipMemGetInfo(&f, &t) != hipSuccess) f = 0;
            sgb = (gb >= 24.0) ? gb / 4.0 + 2.0 : (double)f / 1073741824.0 / 16.0 / (double)per;
            if (sgb < 8.0) sgb = 8.0;
            if (sgb > 48.0) sgb = 48.0;
            if (sgb * 1073741824.0 > (double)f * 0.5) sgb = 0.0;     // a device that is nearly full: nothing
        }
        if (sgb > 0.0 && stream_arena_enabled() && !alloc_policy().plain && stream_arena().largest_free() < (size_t)(sgb * 1073741824.0)) {
            const char * b = std::getenv("TOAST_HIP_ARENA_BUILDER");
            if (b != nullptr && b[0] == '0') reserve((size_t)(sgb * 1073741824.0), true);
            else builder_start((size_t)(sgb * 1073741824.0), device_);
        }
    }
    assigned_ = true;
}

void * Manager::device_alloc(size_t nbytes, int kind) {
    if (nbytes == 0) nbytes = 16;
    // The slabs belong to the device they were taken on (assign_device drops them on a change).  A caller of the *_dev
    // entry points that switches devices on its own gets driver blocks of the device that is current.
    int cur = device_;
    if (device_ >= 0 && hipGetDevice(&cur) != hipSuccess) cur = device_;
    if (alloc_policy().plain || (device_ >= 0 && cur != device_)) {
        void * p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipMalloc(&p, nbytes);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        ++g_direct.mallocs;
        g_direct.malloc_ms += ms;
        if (ms > g_direct.max_malloc_ms) g_direct.max_malloc_ms = ms;
        return p;
    }
    if (kind != kBlockDefault && stream_arena_enabled()) builder_wait();
    if (kind != kBlockDefault && stream_arena_enabled() && stream_arena().capacity() > 0) {
        // Streamed and scatter blocks come from what has been RESERVED for them (assign_device does, by default;
        // toast_hip_arena_reserve_streamed, TOAST_HIP_ARENA_STREAM_GB): building an interleaved slab takes 0.3 s and more,
        // which is set-up work -- never something an operator pays for in passing.  Without a reservation they are
        // ordinary blocks.
        void * p = nullptr;
        if (kind == kBlockStreamed && nbytes >= kStreamBlock) {
            // more than two chunks: in both zones wherever it lies; up to two: astride a boundary
            if (nbytes <= 2 * kStreamBlock) p = stream_arena().alloc_placed(nbytes, place_astride);
            if (p == nullptr) p = stream_arena().alloc(nbytes, stream_, false);
        }
        if (kind == kBlockScatter && nbytes >= kSmallBlock) {
            // (the class of every slab is settled first: scatter_class_of allocates from the arena itself)
            for (const char * b : stream_arena().slab_bases()) (void)scatter_class_of(b, stream_);
            p = stream_arena().alloc_placed(nbytes, [](const char * b, size_t lo, size_t hi, size_t need) {
                return place_in_class(b, lo, hi, need, vmm_slab_scatter_class(b) != 0);
            });
        }
        if (p != nullptr) return p;
    }
    Arena & a = nbytes < kSmallBlock ? small_arena() : big_arena();
    void * p = a.alloc(nbytes, stream_);
    if (p == nullptr) {
        // the driver refused a slab: slabs that hold nothing may be what is in the way (a 16 GB default slab with one
        // block in it does not move, but empty ones do); then exactly what is asked for
        if (big_arena().trim() + small_arena().trim() + stream_arena().trim() > 0) p = a.alloc(nbytes, stream_);
    }
    if (p != nullptr && trace_enabled() && nbytes >= kSmallBlock) {
        const ArenaStats s = a.stats();
        std::fprintf(stderr, "[toast_hip] arena alloc   %10.3f MB -> %p  (%lld slabs, %.1f of %.1f GB in use)\n", nbytes / 1.0e6, p,
                     (long long)s.slabs, s.used_bytes / 1.0e9, s.slab_bytes / 1.0e9);
    }
    return p;
}

void Manager::device_free(void * p) {
    if (p == nullptr) return;
    if (big_arena().release(p) || small_arena().release(p) || stream_arena().release(p)) return;
    (void)hipFree(p);
}

void Manager::wait_for_builder() { builder_wait(); }

size_t Manager::release_cached() {
    builder_wait();
    (void)hipDeviceSynchronize();
    return big_arena().trim() + small_arena().trim() + stream_arena().trim();
}

size_t Manager::cached_bytes() const {
    return big_arena().free_bytes() + small_arena().free_bytes() + stream_arena().free_bytes();
}

void Manager::reserve(size_t bytes, bool streamed) {
    if (alloc_policy().plain || bytes == 0) return;
    if (streamed && !stream_arena_enabled()) return;
    // never more than 90 % of what the device has free right now: a reservation is a hint, not worth a failure
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return;
    if (streamed) {
        builder_wait();
        // streamed blocks are few and large and each has to fit ONE range: the reservation is "a free range of `bytes`"
        if (bytes > f / 10 * 9) return;
        (void)stream_arena().reserve(bytes, stream_, true);
        return;
    }
    Arena & arena = big_arena();
    const size_t have = arena.capacity();
    if (bytes <= have) return;
    size_t want = bytes - have;
    if (want > f / 10 * 9) want = f / 10 * 9;
    if (want < (size_t(64) << 20)) return;
    (void)arena.reserve(have + want, stream_);
}

ZoneRefs zone_references_take(size_t bytes) {
    // From the slabs the read-mostly arrays live in -- only if they exist already (no new slab for this): the first
    // block the arena hands out, and the one at the end of its largest free range (a filler block in between, given back
    // at once).  At the time a streamed slab is built -- right after the read-mostly reservation -- that is the first and
    // the last GB of the slab.
    ZoneRefs r;
    // (a token reservation -- the reference's mem_gb = 1 -- says nothing about where the arrays will live)
    if (alloc_policy().plain || big_arena().capacity() < (size_t(8) << 30)) return r;
    hipStream_t st = Manager::get().stream();
    // (the far one straight from the END of the largest free range: round 5 reached it with a filler block that took
    //  nearly the whole range for a moment -- an allocation racing with it on another thread could end up in a NEW slab)
    r.first = big_arena().alloc(bytes, st, false);
    r.last = big_arena().alloc_at_end(bytes);
    if (r.last == nullptr) {
        r.last = r.first;       // room for one reference only
    }
    return r;
}

bool read_mostly_free_range(const char ** base, size_t * lo, size_t * hi) {
    if (alloc_policy().plain || big_arena().capacity() < (size_t(8) << 30)) return false;
    const char * b0 = nullptr;
    size_t lo0 = 0, hi0 = 0;
    // (a placement that never places: every free range passes by)
    (void)big_arena().alloc_placed(size_t(2) << 20, [&](const char * b, size_t l, size_t h, size_t) {
        if (h - l > hi0 - lo0) {
            b0 = b;
            lo0 = l;
            hi0 = h;
        }
        return SIZE_MAX;
    });
    if (b0 == nullptr) return false;
    *base = b0;
    *lo = lo0;
    *hi = hi0;
    return true;
}

void * read_mostly_take_at(const char * base, size_t offset, size_t bytes) {
    return big_arena().alloc_placed(bytes, [&](const char * b, size_t l, size_t h, size_t need) {
        return (b == base && l <= offset && offset + need <= h) ? offset : SIZE_MAX;
    });
}

void read_mostly_release(void * p) {
    if (p != nullptr) (void)big_arena().release(p);
}

void zone_references_release(const ZoneRefs & r) {
    if (r.first != nullptr) (void)big_arena().release(r.first);
    if (r.last != nullptr && r.last != r.first) (void)big_arena().release(r.last);
}

void Manager::drop_arenas() {
    builder_wait();
    (void)hipDeviceSynchronize();
    drop_param_blocks();
    big_arena().destroy();
    small_arena().destroy();
    stream_arena().destroy();
}

void * Manager::create(const void * host, size_t nbytes, const char * name, int kind) {
    require_device();
    auto it = table_.find(host);
    if (it != table_.end()) {
        // accelerator.cpp:339-347
        std::ostringstream o;
        o << "HipManager:  on create, host ptr " << host << " with " << nbytes
          << " bytes (name='" << it->second.name << "') is already present with "
          << it->second.nbytes << " bytes on device " << device_;
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    const double t0 = trace_begin();
    void * dev = nullptr;
    // TOAST_HIP_MEM_LIMIT_MB: cap on the bytes this manager may hold (the role of the reference's
    // per-process pool size, accelerator.cpp:262-300); exceeding it fails like an exhausted device,
    // which is what lets callers exercise their eviction paths on a 288 GB part.
    static const size_t limit = [] {
        const char * s = std::getenv("TOAST_HIP_MEM_LIMIT_MB");
        return (s != nullptr && std::atol(s) > 0) ? (size_t)std::atol(s) << 20 : (size_t)0;
    }();
    hipError_t e = hipErrorOutOfMemory;
    if (limit == 0 || owned_bytes_ + nbytes <= limit) {
        dev = device_alloc(nbytes, kind);
        e = (dev != nullptr) ? hipSuccess : hipErrorOutOfMemory;
    }
    if (e != hipSuccess || dev == nullptr) {
        (void)hipGetLastError();
        std::ostringstream o;
        o << "HipManager:  on create, host ptr " << host << " with " << nbytes << " bytes (name='"
          << (name ? name : "NA") << "') on device " << device_ << ", allocation failed";
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    Entry ent{dev, nbytes, name ? name : "NA", true};
    table_[host] = ent;
    owned_bytes_ += nbytes;
    ++generation_;
    trace("create", ent.name, nbytes, t0);
    return dev;
}

// Page-lock large host buffers at their first transfer: update_device / update_host then run at
// PCIe speed instead of through the driver's bounce buffers (pageable copies measured 5-10x
// slower).  Buffers that are produced and consumed on the device never pay for the pinning
// (nor for touching their host pages at all).
void Manager::pin_for_transfer(const void * host, Entry & e) {
    if (e.host_registered || e.pin_failed || e.nbytes < pin_threshold()) return;
    if (hipHostRegister(const_cast<void *>(host), e.nbytes, hipHostRegisterDefault) == hipSuccess) {
        e.host_registered = true;
    } else {
        (void)hipGetLastError();
        e.pin_failed = true;
    }
}

void Manager::unpin(const void * host, Entry & e) {
    if (!e.host_registered) return;
    if (e.pin_ends.empty()) {
        (void)hipHostUnregister(const_cast<void *>(host));
    } else {
        size_t off = 0;
        for (size_t end : e.pin_ends) {
            if (end > off) (void)hipHostUnregister(const_cast<char *>(static_cast<const char *>(host)) + off);
            off = end;
        }
        e.pin_ends.clear();
    }
    e.host_registered = false;
}

void Manager::copy_pinned(const Entry & e, void * dev, void * host, size_t off, size_t len, bool to_device, hipStream_t st) {
    auto piece = [&](size_t a, size_t b) {
        if (b <= a) return;
        char * h = static_cast<char *>(host) + a;
        char * d = static_cast<char *>(dev) + a;
        if (to_device) TH_HIP(hipMemcpyAsync(d, h, b - a, hipMemcpyHostToDevice, st));
        else TH_HIP(hipMemcpyAsync(h, d, b - a, hipMemcpyDeviceToHost, st));
    };
    const size_t stop = off + len;
    size_t at = off;
    for (size_t end : e.pin_ends) {
        if (end <= at) continue;
        if (at >= stop) break;
        piece(at, end < stop ? end : stop);
        at = end < stop ? end : stop;
    }
    piece(at, stop);
}

static bool trace_enabled() {
    static int v = -1;
    if (v < 0) {
        const char * s = std::getenv("TOAST_HIP_TRACE");
        v = (s != nullptr && s[0] != '\0' && s[0] != '0') ? 1 : 0;
    }
    return v == 1;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double Manager::trace_begin() { return trace_enabled() ? now_s() : 0.0; }

static bool call_trace_enabled() {
    static int v = -1;
    if (v < 0) {
        const char * s = std::getenv("TOAST_HIP_TRACE");
        v = (s != nullptr && std::atoi(s) >= 2) ? 1 : 0;
    }
    return v == 1;
}

double call_trace_begin() noexcept { return call_trace_enabled() ? now_s() : -1.0; }

void call_trace_end(const char * fn, double t0) noexcept {
    static double origin = t0;
    const double tb = now_s();
    (void)hipDeviceSynchronize();
    const double t1 = now_s();
    std::fprintf(stderr, "[toast_hip] call %10.2f ms  +%9.3f ms  %s (host %.3f ms)\n", (t0 - origin) * 1e3,
                 (t1 - t0) * 1e3, fn, (tb - t0) * 1e3);
}

void Manager::trace(const char * what, const std::string & name, size_t nbytes, double t0) {
    if (!trace_enabled()) return;
    const double dt = now_s() - t0;
    std::fprintf(stderr, "[toast_hip] %-13s %-28s %10.3f MB %8.2f ms %7.2f GB/s\n", what, name.c_str(),
                 nbytes / 1.0e6, dt * 1e3, dt > 0 ? nbytes / dt / 1e9 : 0.0);
}

void Manager::adopt(const void * host, size_t nbytes, void * device, const char * name) {
    require_device();
    if (table_.count(host)) {
        std::ostringstream o;
        o << "HipManager:  on adopt, host ptr " << host << " is already present";
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    Entry ent{device, nbytes, name ? name : "NA", false};
    table_[host] = ent;
    ++generation_;
}

void Manager::reset(const void * host, size_t nbytes, const char * name) {
    require_device();
    Entry & e = lookup(host, nbytes, name, "reset data");
    const double t0 = trace_begin();
    TH_HIP(hipMemsetAsync(e.dev, 0, nbytes, stream_));
    trace("reset", e.name, nbytes, t0);   // (enqueue time only: the fill itself is asynchronous)
}

void Manager::update_device(const void * host, size_t nbytes, const char * name) {
    require_device();
    Entry & e = lookup(host, nbytes, name, "update device");
    const double t0 = trace_begin();
    pin_for_transfer(host, e);
    if (e.host_registered) {
        // page-locked source: direct DMA; the call returns once the source has been consumed
        copy_pinned(e, e.dev, const_cast<void *>(host), 0, nbytes, true, stream_);
        TH_HIP(hipStreamSynchronize(stream_));
    } else {
        copy_to_device(e.dev, host, nbytes, stream_);
    }
    trace("update_device", e.name, nbytes, t0);
}

void Manager::update_device_parts(const void * host, size_t nbytes, const char * name, const size_t * part_end,
                                  int n_parts) {
    require_device();
    Entry & e = lookup(host, nbytes, name, "update device");
    if (n_parts < 1 || part_end == nullptr || part_end[n_parts - 1] != nbytes) {
        fail_arg("update_device_parts: the last part must end at the size of the buffer");
    }
    if (!e.part_done.empty()) fail_arg("update_device_parts: an upload of this buffer is still in flight");
    const double t0 = trace_begin();
    // Page-locking costs ~3.5 ms per GB of host time.  A buffer that is not page-locked yet is locked RANGE BY RANGE, each
    // right before its part is enqueued, so that locking part k + 1 runs while part k crosses PCIe.  Ranges end on 4 KB
    // boundaries past their part's end (two registrations never share a page); copies are split where ranges meet.
    const bool lock_in_parts = !e.host_registered && !e.pin_failed && n_parts > 1 && nbytes >= pin_threshold() &&
                               std::getenv("TOAST_HIP_PIN_IN_PARTS_OFF") == nullptr;
    if (!lock_in_parts) pin_for_transfer(host, e);
    if (!lock_in_parts && !e.host_registered) {
        // pageable source (small buffer, or page-locking failed): the synchronous path through the bounce ring
        copy_to_device(e.dev, host, nbytes, stream_);
        trace("update_device", e.name, nbytes, t0);
        return;
    }
    if (upload_stream_ == nullptr) TH_HIP(hipStreamCreateWithFlags(&upload_stream_, hipStreamNonBlocking));
    // the device block may still be in use by earlier work of the default stream (a reset, a probe pass)
    hipEvent_t before;
    TH_HIP(hipEventCreateWithFlags(&before, hipEventDisableTiming));
    TH_HIP(hipEventRecord(before, stream_));
    TH_HIP(hipStreamWaitEvent(upload_stream_, before, 0));
    (void)hipEventDestroy(before);
    char * hbytes = const_cast<char *>(static_cast<const char *>(host));
    size_t off = 0, locked_to = 0;
    for (int k = 0; k < n_parts; ++k) {
        if (part_end[k] < off || part_end[k] > nbytes) fail_arg("update_device_parts: part ends must increase");
        if (lock_in_parts && part_end[k] > locked_to) {
            size_t end = part_end[k];
            if (k < n_parts - 1) {
                const uintptr_t a = (reinterpret_cast<uintptr_t>(hbytes) + end + 4095) & ~uintptr_t(4095);
                end = (size_t)(a - reinterpret_cast<uintptr_t>(hbytes));
            }
            if (end > nbytes || k == n_parts - 1) end = nbytes;
            if (hipHostRegister(hbytes + locked_to, end - locked_to, hipHostRegisterDefault) != hipSuccess) {
                // give the ranges back and finish through the bounce ring
                (void)hipGetLastError();
                TH_HIP(hipStreamSynchronize(upload_stream_));
                e.host_registered = !e.pin_ends.empty();
                unpin(host, e);
                e.pin_failed = true;
                for (hipEvent_t ev : e.part_done) (void)hipEventDestroy(ev);
                e.part_done.clear();
                if (nbytes > off) copy_to_device(static_cast<char *>(e.dev) + off, hbytes + off, nbytes - off, stream_);
                trace("update_device", e.name, nbytes, t0);
                return;
            }
            e.pin_ends.push_back(end);
            locked_to = end;
        }
        const size_t len = part_end[k] - off;
        if (len > 0) copy_pinned(e, e.dev, hbytes, off, len, true, upload_stream_);
        hipEvent_t ev;
        TH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        TH_HIP(hipEventRecord(ev, upload_stream_));
        e.part_done.push_back(ev);
        off = part_end[k];
    }
    if (lock_in_parts) e.host_registered = true;
    trace("upload_parts", e.name, nbytes, t0);   // (enqueue time: page-locking + launches; the copies run on)
}

void Manager::update_device_wait(const void * host, int part, hipStream_t stream) {
    auto it = table_.find(host);
    if (it == table_.end()) fail_arg("update_device_wait: host pointer is not registered");
    Entry & e = it->second;
    if (e.part_done.empty()) return;   // the upload took the synchronous path: everything is there
    if (part < 0 || part >= (int)e.part_done.size()) fail_arg("update_device_wait: no such part");
    TH_HIP(hipStreamWaitEvent(stream, e.part_done[(size_t)part], 0));
}

int Manager::update_device_arrived(const void * host, int part) {
    auto it = table_.find(host);
    if (it == table_.end()) fail_arg("update_device_arrived: host pointer is not registered");
    Entry & e = it->second;
    if (e.part_done.empty()) return 1;
    if (part < 0 || part >= (int)e.part_done.size()) fail_arg("update_device_arrived: no such part");
    const hipError_t rc = hipEventQuery(e.part_done[(size_t)part]);
    if (rc == hipSuccess) return 1;
    (void)hipGetLastError();
    return 0;
}

void Manager::update_device_finish(const void * host) {
    auto it = table_.find(host);
    if (it == table_.end()) fail_arg("update_device_finish: host pointer is not registered");
    Entry & e = it->second;
    if (e.part_done.empty()) return;
    TH_HIP(hipStreamSynchronize(upload_stream_));
    for (hipEvent_t ev : e.part_done) (void)hipEventDestroy(ev);
    e.part_done.clear();
}

void Manager::update_host(void * host, size_t nbytes, const char * name) {
    require_device();
    Entry & e = lookup(host, nbytes, name, "update host");
    const double t0 = trace_begin();
    pin_for_transfer(host, e);
    if (e.host_registered) {
        copy_pinned(e, e.dev, host, 0, nbytes, false, stream_);
        TH_HIP(hipStreamSynchronize(stream_));
    } else {
        copy_to_host(host, e.dev, nbytes, stream_);
    }
    trace("update_host", e.name, nbytes, t0);
}

void Manager::remove(const void * host, size_t nbytes, const char * name) {
    require_device();
    Entry & e = lookup(host, nbytes, name, "delete");
    const double t0 = trace_begin();
    if (!e.part_done.empty()) update_device_finish(host);   // an upload in parts that nobody waited for
    TH_HIP(hipStreamSynchronize(stream_));
    unpin(host, e);
    if (e.owned) {
        device_free(e.dev);
        owned_bytes_ -= (e.nbytes <= owned_bytes_) ? e.nbytes : owned_bytes_;
    }
    trace("delete", e.name, nbytes, t0);
    table_.erase(host);
    ++generation_;
}

void * Manager::find(const void * host) {
    auto it = table_.find(host);
    return (it == table_.end()) ? nullptr : it->second.dev;
}

void * Manager::device_ptr(const void * host) {
    void * p = find(host);
    if (p == nullptr) {
        // accelerator.hpp:127-133
        std::ostringstream o;
        o << "HipManager:  host ptr " << host << " is not present- cannot get device pointer";
        throw Error(TOAST_HIP_ERR_MEMORY, o.str());
    }
    return p;
}

void Manager::dump() {
    std::printf("HipManager: device %d, %zu buffers\n", device_, table_.size());
    for (auto & kv : table_) {
        std::printf("  host %p -> dev %p  %zu bytes  '%s'\n", kv.first, kv.second.dev,
                    kv.second.nbytes, kv.second.name.c_str());
    }
    std::fflush(stdout);
}

// ------------------------------------------------------------------ staging
void * Staging::resolve(void * host, size_t bytes, bool upload, bool download, bool force_temp) {
    if (host == nullptr) return nullptr;
    if (accel_ && !force_temp) return Manager::get().device_ptr(host);
    void * dev = nullptr;
    // stream-ordered pool allocation: no device-wide synchronisation per staged call
    TH_HIP(hipMallocAsync(&dev, bytes ? bytes : 16, stream_));
    // Large buffers: page-lock the host range for the call (DMA at PCIe speed; the bounce ring is bound by the host
    // memcpy, ~10 GB/s) and release it in finish(), before the caller can free the memory.
    bool registered = false;
    if (bytes >= pin_threshold() && (upload || download)) {
        if (hipHostRegister(host, bytes, hipHostRegisterDefault) == hipSuccess) {
            registered = true;
        } else {
            (void)hipGetLastError();
        }
    }
    temps_.push_back(Temp{host, dev, bytes, download, registered});
    if (upload && bytes) {
        if (registered) {
            TH_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream_));
        } else {
            copy_to_device(dev, host, bytes, stream_);
        }
    }
    return dev;
}

void Staging::finish() {
    if (finished_) return;
    finished_ = true;
    if (temps_.empty()) return;
    for (auto & t : temps_) {
        if (t.download && t.bytes) {
            if (t.registered) {
                TH_HIP(hipMemcpyAsync(t.host, t.dev, t.bytes, hipMemcpyDeviceToHost, stream_));
            } else {
                copy_to_host(t.host, t.dev, t.bytes, stream_);
            }
        }
    }
    for (auto & t : temps_) (void)hipFreeAsync(t.dev, stream_);
    TH_HIP(hipStreamSynchronize(stream_));
    for (auto & t : temps_) {
        if (t.registered) (void)hipHostUnregister(t.host);
    }
    temps_.clear();
}

Staging::~Staging() {
    if (!finished_) {
        // error path: make sure nothing queued still uses the temporaries, then free them
        for (auto & t : temps_) (void)hipFreeAsync(t.dev, stream_);
        (void)hipStreamSynchronize(stream_);
        for (auto & t : temps_) {
            if (t.registered) (void)hipHostUnregister(t.host);
        }
    }
}

}  // namespace toast_hip

// ------------------------------------------------------------------ C ABI: runtime part
using namespace toast_hip;

extern "C" {

const char * toast_hip_last_error(void) { return g_last_error.c_str(); }

const char * toast_hip_version(void) { return "toast_hip 0.1 (gfx950)"; }

// Raw device blocks from the manager's arena (bench.py, the solver's packed pointing cache, experiments): flags -1
// (and the older -2) = Manager::device_alloc; 0 = a plain hipMalloc; > 0 = hipExtMallocWithFlags flags.
int toast_hip_arena_stats(toast_hip_arena_stats_t * out) {
    return guarded([&] {
        if (out == nullptr) fail_arg("toast_hip_arena_stats: out is NULL");
        const AllocStats a = alloc_stats();
        out->slabs = a.slabs;
        out->slab_bytes = a.slab_bytes;
        out->used_bytes = a.used_bytes;
        out->peak_used_bytes = a.peak_used_bytes;
        out->slab_mallocs = a.slab_mallocs;
        out->slab_frees = a.slab_frees;
        out->malloc_ms = a.malloc_ms;
        out->max_malloc_ms = a.max_malloc_ms;
        out->touch_ms = a.touch_ms;
        out->allocs = a.allocs;
        out->releases = a.releases;
        out->direct_mallocs = a.direct_mallocs;
        out->failed = a.failed;
        const VmmSlabStats v = vmm_slab_stats();
        out->interleaved_slabs = v.slabs;
        out->chunks = v.chunks;
        out->chunks_other_zone = v.chunks_other_zone;
        out->chunks_created = v.created;
        out->interleave_ms = v.build_ms;
        out->same_zone_tbs = v.same_zone_tbs;
        out->chunks_other_wanted = v.chunks_other_wanted;
        out->searches = v.searches;
        out->searches_exhausted = v.searches_exhausted;
        out->searches_capped_ms = v.searches_capped_ms;
        out->probes = v.probes;
        out->probes_by_clock = v.probes_by_clock;
        out->create_ms_per_chunk = v.create_ms_per_chunk;
        out->search_ms = v.search_ms;
        out->slabs_third_zone = v.slabs_third_zone;
        out->read_mostly_zones = v.read_mostly_zones;
    });
}

int toast_hip_arena_placement_status(int * placement_ok, int * search_exhausted, int64_t * chunks_other_zone,
                                     int64_t * chunks_other_wanted) {
    return guarded([&] {
        Manager::wait_for_builder();
        const VmmSlabStats v = vmm_slab_stats();
        // ok: at least one interleaved slab stands and every slot that the pattern gives to the other zone holds a chunk
        // that measured clear of the read-mostly slabs
        if (placement_ok) *placement_ok = (v.slabs > 0 && v.chunks_other_zone >= v.chunks_other_wanted) ? 1 : 0;
        if (search_exhausted) *search_exhausted = v.searches_exhausted > 0 ? 1 : 0;
        if (chunks_other_zone) *chunks_other_zone = v.chunks_other_zone;
        if (chunks_other_wanted) *chunks_other_wanted = v.chunks_other_wanted;
    });
}

int toast_hip_arena_reserve(size_t bytes) {
    // (on the CURRENT device: the device-pointer entry points serve callers that never went through assign_device)
    return guarded([&] { Manager::get().reserve(bytes); });
}

int toast_hip_arena_reserve_streamed(size_t bytes) {
    return guarded([&] { Manager::get().reserve(bytes, true); });
}

int toast_hip_arena_block_zone(const void * device_ptr, size_t bytes, int * interleaved, int * chunks_own_zone,
                               int * chunks_other_zone) {
    return guarded([&] {
        Manager::wait_for_builder();
        int in = 0, own = 0, other = 0;
        size_t off = 0, chunk = 0;
        if (stream_arena().slab_offset(device_ptr, &off) &&
            vmm_slab_layout(static_cast<const char *>(device_ptr) - off, &chunk) && chunk > 0) {
            in = 1;
            const size_t last = off + (bytes ? bytes - 1 : 0);
            for (size_t k = off / chunk; k <= last / chunk; ++k) (vmm_slot_other(k) ? other : own) += 1;
        }
        if (interleaved) *interleaved = in;
        if (chunks_own_zone) *chunks_own_zone = own;
        if (chunks_other_zone) *chunks_other_zone = other;
    });
}

// The sub-allocation logic on host memory (no device needed): `n_ops` random allocations / releases of 1 .. max_block
// bytes against an arena with the given granule and slab size; after every step the bookkeeping is checked, live blocks
// are checked for overlap through a byte pattern, and at the end everything is released and every slab must be one
// free range again.  0 = sound; otherwise the first inconsistency is in toast_hip_last_error().
int toast_hip_arena_zone_threshold(const double * rates, int n, double level, double * threshold) {
    return guarded([&] {
        if (rates == nullptr || threshold == nullptr || n < 0) fail_arg("toast_hip_arena_zone_threshold: null argument");
        *threshold = vmm_zone_threshold(rates, n, level);
    });
}

int toast_hip_arena_selftest(uint64_t seed, int n_ops, size_t granule, size_t slab_bytes, size_t max_block) {
    return guarded([&] {
        if (granule == 0 || slab_bytes < granule || max_block == 0) fail_arg("toast_hip_arena_selftest: bad sizes");
        Arena a(granule, slab_bytes, host_backend());
        struct Live { unsigned char * p; size_t n; unsigned char tag; };
        std::vector<Live> live;
        uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
        auto rnd = [&] {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            return (uint64_t)(st >> 24);
        };
        auto verify = [&](const Live & b) {
            for (size_t i = 0; i < b.n; i += (b.n > 4096 ? b.n / 64 : 1)) {
                if (b.p[i] != b.tag) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: a live block was overwritten (overlap)");
            }
            if (b.p[b.n - 1] != b.tag) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: a live block was overwritten (overlap)");
        };
        for (int op = 0; op < n_ops; ++op) {
            const bool do_alloc = live.empty() || (rnd() % 100) < 55;
            if (do_alloc) {
                // a mix of sizes: mostly small, some close to a slab, a few above the slab size
                const uint64_t r = rnd() % 100;
                size_t n = 1 + (size_t)(rnd() % max_block);
                if (r < 50) n = 1 + n % (max_block / 16 + 1);
                unsigned char * p = nullptr;
                const size_t stripe = 8 * granule;
                if (r >= 90 && n <= stripe) {
                    // a striped block: inside one stripe of the wanted parity, or refused (never a new slab)
                    const int parity = (int)(rnd() & 1);
                    p = static_cast<unsigned char *>(a.alloc_striped(n, stripe, parity));
                    if (p == nullptr) continue;
                    size_t off = 0;
                    if (!a.slab_offset(p, &off) || off / stripe != (off + n - 1) / stripe || (int)((off / stripe) & 1) != parity) {
                        throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: a striped block crosses its stripe or has the wrong parity");
                    }
                } else {
                    p = static_cast<unsigned char *>(a.alloc(n, nullptr));
                }
                if (p == nullptr) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: host allocation failed");
                const unsigned char tag = (unsigned char)(1 + rnd() % 255);
                std::memset(p, tag, n);
                live.push_back(Live{p, n, tag});
            } else {
                const size_t k = (size_t)(rnd() % live.size());
                verify(live[k]);
                if (!a.release(live[k].p)) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: release refused a live block");
                if (a.release(live[k].p)) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: a block was released twice");
                live[k] = live.back();
                live.pop_back();
            }
            const std::string bad = a.check();
            if (!bad.empty()) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: " + bad);
        }
        for (const Live & b : live) {
            verify(b);
            if (!a.release(b.p)) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: release refused a live block");
        }
        const std::string bad = a.check();
        if (!bad.empty()) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: " + bad);
        if (a.free_bytes() != a.capacity()) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: bytes still in use after the last release");
        const ArenaStats s = a.stats();
        if (a.largest_free() * (size_t)s.slabs < a.capacity() && s.slabs == 1) {
            throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: free ranges were not merged");
        }
        const size_t held = a.capacity();
        if (a.trim() != held || a.capacity() != 0) throw Error(TOAST_HIP_ERR_MEMORY, "arena selftest: trim left slabs behind");
    });
}

// Time (ms) of one read + write pass over [p, p + bytes) with the timestream kernels' access pattern (1024 rows in
// flight); the better of two passes.  For placement experiments on ranges of arena blocks (tools/exp_arena_regions.py).
int toast_hip_probe_stream(void * p, size_t bytes, double * ms) {
    return guarded([&] {
        *ms = probe_stream_ms(p, bytes, Manager::get().stream());
    });
}

int toast_hip_exp_vmm_pair_matrix(int n_phys, int n_slots, double * out) {
    return guarded([&] { vmm_pair_matrix(n_phys, n_slots, out, Manager::get().stream()); });
}

int toast_hip_probe_stream_split(void * const * bases, int nb, size_t bytes_each, double * ms) {
    return guarded([&] { *ms = probe_stream_split_ms(bases, nb, bytes_each, Manager::get().stream()); });
}

int toast_hip_probe_byte_mix_dev(const int64_t * d_pixels, const double * d_weights, const double * d_tod, double * d_out,
                                 int64_t n_det, int64_t n_samp, void * stream) {
    return guarded([&] {
        probe_byte_mix(d_pixels, d_weights, d_tod, d_out, n_det, n_samp, static_cast<hipStream_t>(stream));
    });
}

int toast_hip_accel_mem_info(size_t * free_bytes, size_t * total_bytes) {
    return guarded([&] {
        Manager::get().require_device();
        size_t f = 0, t = 0;
        TH_HIP(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f + Manager::get().cached_bytes();    // (free ranges of the slabs can be had at once)
        if (total_bytes) *total_bytes = t;
    });
}

int toast_hip_accel_release_cached(void) {
    return guarded([&] { Manager::get().release_cached(); });
}

int toast_hip_device_malloc(size_t nbytes, int flags, void ** out) {
    return guarded([&] {
        void * p = nullptr;
        if (flags < 0) {
            p = Manager::get().device_alloc(nbytes, flags == -3 ? Manager::kBlockStreamed
                                                    : flags == -4 ? Manager::kBlockScatter : Manager::kBlockDefault);
            if (p == nullptr) throw Error(TOAST_HIP_ERR_MEMORY, "HipManager:  device_malloc, allocation failed");
        } else if (flags == 0) {
            TH_HIP(hipMalloc(&p, nbytes));
        } else {
            TH_HIP(hipExtMallocWithFlags(&p, nbytes, (unsigned)flags));
        }
        *out = p;
    });
}

// Run-time tuning switches for experiments and tests: "det_major" = 0 / 1, "pair" = 0 / 1.
int toast_hip_set_tuning(const char * key, int value) {
    return guarded([&] {
        if (std::string(key) == "det_major") {
            set_det_major_grid(value);
        } else if (std::string(key) == "pair") {
            set_pair_detectors(value);
        } else if (std::string(key) == "vec2") {
            set_vec2_lanes(value);
        } else {
            fail_arg(std::string("unknown tuning key ") + key);
        }
    });
}

// EXPERIMENT (tools/exp_alloc_flags.py): a virtual range backed by `chunk_mb`-sized physical allocations
// mapped in a shuffled order (flags: 100 = in order, 101 = shuffled).  Never freed.
int toast_hip_device_malloc_vmm(size_t nbytes, int chunk_mb, int shuffled, void ** out) {
    return guarded([&] {
        int dev = 0;
        TH_HIP(hipGetDevice(&dev));
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        TH_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t chunk = (size_t)chunk_mb << 20;
        if (chunk < gran) chunk = gran;
        chunk = (chunk + gran - 1) / gran * gran;
        const size_t n = (nbytes + chunk - 1) / chunk;
        void * va = nullptr;
        TH_HIP(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) TH_HIP(hipMemCreate(&h[i], chunk, &prop, 0));
        std::vector<size_t> perm(n);
        for (size_t i = 0; i < n; ++i) perm[i] = i;
        if (shuffled) {
            uint64_t st = 0x9E3779B97F4A7C15ull;
            for (size_t i = n - 1; i > 0; --i) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                const size_t j = (size_t)((st >> 33) % (i + 1));
                std::swap(perm[i], perm[j]);
            }
        }
        for (size_t i = 0; i < n; ++i) {
            TH_HIP(hipMemMap((char *)va + i * chunk, chunk, 0, h[perm[i]], 0));
        }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        TH_HIP(hipMemSetAccess(va, n * chunk, &acc, 1));
        *out = va;
    });
}

int toast_hip_device_free(void * p) {
    return guarded([&] {
        if (p == nullptr) return;
        // nothing enqueued may still be using the range when somebody else takes it over
        TH_HIP(hipStreamSynchronize(Manager::get().stream()));
        Manager::get().device_free(p);
    });
}
int toast_hip_device_release(void * p, size_t /*nbytes*/) { return toast_hip_device_free(p); }

int toast_hip_accel_generation(uint64_t * generation) {
    return toast_hip::guarded([&] { *generation = toast_hip::Manager::get().generation(); });
}

int toast_hip_accel_enabled(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

int toast_hip_accel_assign_device(int node_procs, int node_rank, double mem_gb, int disabled) {
    return guarded([&] { Manager::get().assign_device(node_procs, node_rank, mem_gb, disabled != 0); });
}

int toast_hip_accel_get_device(int * device) {
    return guarded([&] { *device = Manager::get().device(); });
}

int toast_hip_accel_present(const void * host, size_t nbytes, int * present) {
    return guarded([&] { *present = Manager::get().present(host, nbytes); });
}

int toast_hip_accel_create(const void * host, size_t nbytes, const char * name) {
    return guarded([&] { Manager::get().create(host, nbytes, name); });
}

int toast_hip_accel_create_kind(const void * host, size_t nbytes, const char * name, int kind) {
    return guarded([&] {
        if (kind < 0 || kind > 2) fail_arg("toast_hip_accel_create_kind: kind must be 0 (default), 1 (streamed) or 2 (scatter target)");
        Manager::get().create(host, nbytes, name, kind);
    });
}

int toast_hip_accel_adopt(const void * host, size_t nbytes, void * device, const char * name) {
    return guarded([&] { Manager::get().adopt(host, nbytes, device, name); });
}

int toast_hip_accel_reset(const void * host, size_t nbytes, const char * name) {
    return guarded([&] { Manager::get().reset(host, nbytes, name); });
}

int toast_hip_accel_update_device(const void * host, size_t nbytes, const char * name) {
    return guarded([&] { Manager::get().update_device(host, nbytes, name); });
}

int toast_hip_accel_update_device_parts(const void * host, size_t nbytes, const char * name, const size_t * part_end,
                                        int n_parts) {
    return guarded([&] { Manager::get().update_device_parts(host, nbytes, name, part_end, n_parts); });
}

int toast_hip_accel_update_device_wait(const void * host, int part, void * stream) {
    return guarded([&] { Manager::get().update_device_wait(host, part, static_cast<hipStream_t>(stream)); });
}

int toast_hip_accel_update_device_arrived(const void * host, int part, int * arrived) {
    return guarded([&] {
        const int a = Manager::get().update_device_arrived(host, part);
        if (arrived) *arrived = a;
    });
}

int toast_hip_accel_update_device_finish(const void * host) {
    return guarded([&] { Manager::get().update_device_finish(host); });
}

int toast_hip_accel_update_host(void * host, size_t nbytes, const char * name) {
    return guarded([&] { Manager::get().update_host(host, nbytes, name); });
}

int toast_hip_accel_delete(const void * host, size_t nbytes, const char * name) {
    return guarded([&] { Manager::get().remove(host, nbytes, name); });
}

int toast_hip_accel_device_ptr(const void * host, void ** device) {
    return guarded([&] { *device = Manager::get().device_ptr(host); });
}

int toast_hip_accel_dump(void) {
    return guarded([&] { Manager::get().dump(); });
}

int toast_hip_set_stream(void * stream) {
    return guarded([&] { Manager::get().set_stream(static_cast<hipStream_t>(stream)); });
}

int toast_hip_synchronize(void) {
    return guarded([&] { TH_HIP(hipStreamSynchronize(Manager::get().stream())); });
}

}  // extern "C"
