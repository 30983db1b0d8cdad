// runtime.cpp -- see runtime.hpp.  Reference counterpart of the memory manager:
// /root/reference/src/toast/_libtoast/accelerator.cpp:233-766 (OmpManager).
#include "runtime.hpp"

#include <chrono>
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
// 8 MB slabs (bump allocation): committing a new block costs one async copy, never a hipMalloc or
// a synchronisation on the launch path.  When the cache is full everything is dropped at once
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
    void * p = nullptr;
    TH_HIP(hipMalloc(&p, s.bytes));
    s.base = static_cast<char *>(p);
    s.used = need;
    g_slabs.push_back(s);
    return s.base;
}

void drop_all_blocks() {
    // Blocks still referenced by queued kernels must outlive them.
    TH_HIP(hipDeviceSynchronize());
    for (auto & s : g_slabs) (void)hipFree(s.base);
    g_slabs.clear();
    g_blocks.clear();
    g_block_bytes = 0;
}

}  // namespace

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
        TH_HIP(hipMemcpyAsync(dst + off, g_bounce.slot[k], n, hipMemcpyHostToDevice, stream));
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

// Released blocks of at least 64 MB are kept for the next create of exactly that size instead of
// going back to the driver: the map-maker allocates and frees the same multi-GB temporaries in
// every phase (temp_RHS, the ApplyAmplitudes timestream, per-batch quaternions), and hipMalloc costs
// up to ~27 ms per GB on some boxes (0.3 ms on others).  TOAST_HIP_ALLOC_CACHE_MB caps the bytes
// held (default 32768, 0 disables); the cache is emptied when an allocation fails and on clear().
void * Manager::take_cached(size_t nbytes) {
    for (size_t i = 0; i < free_blocks_.size(); ++i) {
        // (candidates that the placement policy measured as slow are only handed out by device_alloc, as a last resort)
        if (free_blocks_[i].second == nbytes && free_blocks_[i].slow_tbs <= 0.0) {
            void * p = free_blocks_[i].first;
            free_blocks_.erase(free_blocks_.begin() + (long)i);
            cached_bytes_ -= nbytes;
            return p;
        }
    }
    return nullptr;
}

bool Manager::keep_cached(void * dev, size_t nbytes, double slow_tbs) {
    static const size_t cap = [] {
        const char * s = std::getenv("TOAST_HIP_ALLOC_CACHE_MB");
        return (size_t)((s != nullptr) ? std::atol(s) : 32768) << 20;
    }();
    if (nbytes < ((size_t)64 << 20) || cap == 0) return false;
    while (!free_blocks_.empty() && cached_bytes_ + nbytes > cap) {   // oldest first
        (void)hipFree(free_blocks_.front().first);
        cached_bytes_ -= free_blocks_.front().second;
        free_blocks_.erase(free_blocks_.begin());
    }
    if (cached_bytes_ + nbytes > cap) return false;
    free_blocks_.push_back(FreeBlock{dev, nbytes, slow_tbs});
    cached_bytes_ += nbytes;
    return true;
}

void Manager::flush_cached() {
    for (auto & b : free_blocks_) (void)hipFree(b.first);
    free_blocks_.clear();
    cached_bytes_ = 0;
}

// Grow-only scratch buffers owned by the manager (so: per process = per device, released by
// clear()).  A failed growth leaves the slot empty -- never a dangling pointer -- and is retried once
// after the cache of released blocks has been given back to the driver.
void * Manager::scratch(int slot, size_t bytes) {
    // keyed by the CURRENT device: the device-pointer entry points (toast_hip_*_dev) serve callers
    // that own their device memory and never went through assign_device()
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    auto & s = scratch_[std::make_pair(dev, slot)];
    if (bytes <= s.second && s.first != nullptr) return s.first;
    if (s.first != nullptr) {
        TH_HIP(hipDeviceSynchronize());
        void * old = s.first;
        s.first = nullptr;
        s.second = 0;
        TH_HIP(hipFree(old));
    }
    void * p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if ((e != hipSuccess || p == nullptr) && !free_blocks_.empty()) {
        (void)hipGetLastError();
        flush_cached();
        e = hipMalloc(&p, bytes ? bytes : 16);
    }
    if (e != hipSuccess || p == nullptr) {
        (void)hipGetLastError();
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
    flush_cached();
    for (auto & kv : scratch_) {
        if (kv.second.first) (void)hipFree(kv.second.first);
    }
    scratch_.clear();
    if (upload_stream_ != nullptr) (void)hipStreamSynchronize(upload_stream_);
    for (auto & kv : table_) {
        for (hipEvent_t ev : kv.second.part_done) (void)hipEventDestroy(ev);
        kv.second.part_done.clear();
        unpin(kv.first, kv.second);
        if (kv.second.owned) (void)hipFree(kv.second.dev);
    }
    table_.clear();
    owned_bytes_ = 0;
    ++generation_;
}

void Manager::assign_device(int node_procs, int node_rank, double /*mem_gb*/, bool disabled) {
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
        device_ = -1;
    } else {
        // accelerator.cpp:276-281: ceil(node_procs / n_dev) processes share a device
        int per = node_procs / n_dev;
        if (n_dev * per < node_procs) per += 1;
        device_ = node_rank / per;
        TH_HIP(hipSetDevice(device_));
    }
    assigned_ = true;
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

// One allocation function for everything large the manager owns.
//
// Placement policy.  The time-major kernels stream one piece of each of ~1000 detector rows at a time, and how fast an
// allocation streams under that pattern depends on where the driver placed it: 5.05 or 5.99 TB/s, for the lifetime of
// the allocation (profiles/r02_d_placement_experiments.txt); the headline moves by 12 % with it.
// Only measuring an allocation tells.  So blocks of the size class that shows the two levels (1 GB .. 8 GB: the
// timestreams and the pixel numbers; the 17.7 GB weight buffers show a 2.5 % spread and are not probed) are chosen
// among up to K candidates: each candidate gets one row-parallel read + write pass with the kernels' own access
// pattern (k_probe_stream, after a first-touch pass); the first one that streams at the fast level is taken, otherwise
// the fastest of the K.  Candidates are held until the choice is made (a freed slow region would simply be handed out
// again) and the losers are then released.  Cost: one ~5 ms probe per block on a box that hands out fast memory, K of
// them on one that does not.
//   TOAST_HIP_ALLOC=probe[:K]   the policy with K candidates (default: probe:8; a candidate costs ~2 ms, and with ~40 %
//                               of them at the fast level four were not enough in one process out of ten)
//   TOAST_HIP_ALLOC=plain       one hipMalloc per block, no probing
//   TOAST_HIP_ALLOC=contiguous  hipDeviceMallocContiguous for blocks >= 256 MB (experiment: always the slow level)
// Returns nullptr on failure.
namespace {
struct AllocPolicy {
    bool contiguous = false;
    int probe_k = 8;
    double accept_tbs = 5.65;   // read + write bytes / probe time: between the two levels (5.05-5.4 and 5.9-6.0)
    size_t max_bytes = size_t(8) << 30;   // TOAST_HIP_ALLOC_PROBE_MAX_GB
    double budget_ms = 60.0;              // TOAST_HIP_ALLOC_BUDGET_MS: what a candidate may cost (two dearer ones in a row end the search)
    size_t hold_bytes = size_t(24) << 30;  // TOAST_HIP_ALLOC_HOLD_GB: slow candidates kept allocated between searches
    size_t min_bytes = size_t(1) << 30;    // TOAST_HIP_ALLOC_PROBE_MIN_MB
};
const AllocPolicy & alloc_policy() {
    static const AllocPolicy pol = [] {
        AllocPolicy a;
        const char * e = std::getenv("TOAST_HIP_ALLOC");
        if (e != nullptr) {
            const std::string v(e);
            if (v == "contiguous") {
                a.contiguous = true;
                a.probe_k = 0;
            } else if (v == "plain") {
                a.probe_k = 0;
            } else if (v.rfind("probe", 0) == 0) {
                const char * c = std::strchr(e, ':');
                const int k = c ? std::atoi(c + 1) : 8;
                a.probe_k = k < 2 ? 2 : (k > 32 ? 32 : k);
            }
        }
        const char * t = std::getenv("TOAST_HIP_ALLOC_ACCEPT_TBS");
        if (t != nullptr && std::atof(t) > 0.0) a.accept_tbs = std::atof(t);
        const char * b = std::getenv("TOAST_HIP_ALLOC_BUDGET_MS");
        if (b != nullptr && std::atof(b) > 0.0) a.budget_ms = std::atof(b);
        const char * h = std::getenv("TOAST_HIP_ALLOC_HOLD_GB");
        if (h != nullptr && std::atol(h) >= 0) a.hold_bytes = (size_t)std::atol(h) << 30;
        const char * n = std::getenv("TOAST_HIP_ALLOC_PROBE_MIN_MB");
        if (n != nullptr && std::atol(n) >= 16) a.min_bytes = (size_t)std::atol(n) << 20;
        const char * m = std::getenv("TOAST_HIP_ALLOC_PROBE_MAX_GB");
        if (m != nullptr && std::atol(m) > 0) a.max_bytes = (size_t)std::atol(m) << 30;
        return a;
    }();
    return pol;
}
AllocStats g_alloc_stats;
}  // namespace

const AllocStats & alloc_stats() { return g_alloc_stats; }

void * Manager::device_alloc(size_t nbytes) {
    const AllocPolicy & pol = alloc_policy();
    const size_t lo = pol.min_bytes, hi = pol.max_bytes;
    if (pol.probe_k > 0 && nbytes >= lo && nbytes <= hi) {
        std::vector<void *> cand;
        std::vector<double> tbs;
        size_t best = 0;
        // A candidate normally costs ~5 ms (hipMalloc, first-touch pass, timed pass).  On a box whose memory is still
        // being cleared after another process hipMalloc can take ~200 ms per 5.9 GB, candidate after candidate: the
        // search stops when two candidates in a row have each cost more than `budget_ms`.  (One slow call says little:
        // the driver also stalls a SINGLE hipMalloc for seconds when it runs out of cleared memory, and the next ones
        // are quick again -- counting that against the block would leave it with the one candidate.)  A cap on the
        // whole search bounds the worst case.
        const auto t_start = std::chrono::steady_clock::now();
        auto ms_since = [](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        };
        int slow_in_a_row = 0;
        for (int k = 0; k < pol.probe_k; ++k) {
            if (k > 0 && (slow_in_a_row >= 2 || ms_since(t_start) > 40.0 * pol.budget_ms)) {
                ++g_alloc_stats.budget_stops;
                break;
            }
            const auto t_cand = std::chrono::steady_clock::now();
            void * c = nullptr;
            if (hipMalloc(&c, nbytes) != hipSuccess) {
                (void)hipGetLastError();
                break;   // memory is short: make do with the candidates so far
            }
            const double malloc_ms = ms_since(t_cand);
            g_alloc_stats.malloc_ms += malloc_ms;
            if (malloc_ms > g_alloc_stats.max_malloc_ms) g_alloc_stats.max_malloc_ms = malloc_ms;
            const double ms = probe_stream_ms(c, nbytes, stream_);
            cand.push_back(c);
            tbs.push_back(ms > 0.0 ? 2.0 * (double)nbytes / ms / 1.0e9 : 0.0);
            g_alloc_stats.probe_ms += ms;
            ++g_alloc_stats.candidates;
            slow_in_a_row = (ms_since(t_cand) > pol.budget_ms) ? slow_in_a_row + 1 : 0;
            if (tbs.back() > tbs[best]) best = cand.size() - 1;
            if (tbs.back() >= pol.accept_tbs) break;
        }
        // Slow candidates of EARLIER searches for this size are still allocated (below: they sit in the cache of
        // released blocks, which keeps the driver from handing the same ranges out again): if none of the new ones is
        // fast either, the best of all of them is taken.
        const size_t n_new = cand.size();
        if (cand.empty() || tbs[best] < pol.accept_tbs) {
            for (const FreeBlock & b : free_blocks_) {
                if (b.second == nbytes && b.slow_tbs > 0.0) {
                    cand.push_back(b.first);
                    tbs.push_back(b.slow_tbs);
                    if (tbs.back() > tbs[best]) best = cand.size() - 1;
                }
            }
        }
        if (!cand.empty()) {
            ++g_alloc_stats.probed_blocks;
            if (tbs[best] >= pol.accept_tbs) ++g_alloc_stats.fast_blocks;
            g_alloc_stats.last_tbs = tbs[best];
            if (trace_enabled()) {
                std::string line;
                for (size_t k = 0; k < cand.size(); ++k) {
                    line += (k ? " " : "") + std::to_string(tbs[k]) + (k >= n_new ? "(held)" : "");
                }
                std::fprintf(stderr, "[toast_hip] probe         %.1f MB: %s TB/s, kept #%zu\n", nbytes / 1.0e6, line.c_str(),
                             best);
            }
            void * chosen = cand[best];
            if (best >= n_new) {
                // a held one: out of the cache
                for (size_t i = 0; i < free_blocks_.size(); ++i) {
                    if (free_blocks_[i].first == chosen) {
                        free_blocks_.erase(free_blocks_.begin() + (long)i);
                        cached_bytes_ -= nbytes;
                        break;
                    }
                }
                ++g_alloc_stats.held_reused;
            }
            // The losers stay allocated in the cache of released blocks, marked with their rate (a rate of 0 would make
            // them ordinary released blocks): at most `hold_bytes` of them, inside that cache's own cap, and given back
            // with the rest of the cache when memory runs short.
            size_t held = 0;
            for (const FreeBlock & b : free_blocks_) held += (b.slow_tbs > 0.0) ? b.second : 0;
            for (size_t k = 0; k < n_new; ++k) {
                if (k == best) continue;
                if (held + nbytes <= pol.hold_bytes && keep_cached(cand[k], nbytes, tbs[k] > 0.0 ? tbs[k] : 1.0e-3)) {
                    held += nbytes;
                } else {
                    (void)hipFree(cand[k]);
                }
            }
            return chosen;
        }
    }
    void * p = nullptr;
    if (pol.contiguous && nbytes >= (size_t(256) << 20)) {
        if (hipExtMallocWithFlags(&p, nbytes, hipDeviceMallocContiguous) == hipSuccess && p != nullptr) return p;
        (void)hipGetLastError();
        p = nullptr;
    }
    if (hipMalloc(&p, nbytes ? nbytes : 16) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void * Manager::create(const void * host, size_t nbytes, const char * name) {
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
        dev = take_cached(nbytes);
        if (dev != nullptr) {
            e = hipSuccess;
        } else {
            dev = device_alloc(nbytes);
            e = (dev != nullptr) ? hipSuccess : hipErrorOutOfMemory;
            if (dev == nullptr && !free_blocks_.empty()) {
                // the cache of released blocks is holding the memory: give it back and retry
                flush_cached();
                dev = device_alloc(nbytes);
                e = (dev != nullptr) ? hipSuccess : hipErrorOutOfMemory;
            }
        }
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
    (void)hipHostUnregister(const_cast<void *>(host));
    e.host_registered = false;
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
        TH_HIP(hipMemcpyAsync(e.dev, host, nbytes, hipMemcpyHostToDevice, stream_));
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
    pin_for_transfer(host, e);
    if (!e.host_registered) {
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
    size_t off = 0;
    for (int k = 0; k < n_parts; ++k) {
        if (part_end[k] < off || part_end[k] > nbytes) fail_arg("update_device_parts: part ends must increase");
        const size_t len = part_end[k] - off;
        if (len > 0) {
            TH_HIP(hipMemcpyAsync(static_cast<char *>(e.dev) + off, static_cast<const char *>(host) + off, len,
                                  hipMemcpyHostToDevice, upload_stream_));
        }
        hipEvent_t ev;
        TH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        TH_HIP(hipEventRecord(ev, upload_stream_));
        e.part_done.push_back(ev);
        off = part_end[k];
    }
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
        TH_HIP(hipMemcpyAsync(host, e.dev, nbytes, hipMemcpyDeviceToHost, stream_));
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
        if (!keep_cached(e.dev, e.nbytes)) TH_HIP(hipFree(e.dev));
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

// Raw device allocations with the manager's policy (experiments, bench.py): flags as in
// hipExtMallocWithFlags (0 default, 4 hipDeviceMallocContiguous), -1 = the manager's own policy.
int toast_hip_alloc_stats(int64_t * probed_blocks, int64_t * fast_blocks, int64_t * candidates, double * probe_ms,
                          double * last_tbs) {
    return guarded([&] {
        const AllocStats & a = alloc_stats();
        if (probed_blocks) *probed_blocks = a.probed_blocks;
        if (fast_blocks) *fast_blocks = a.fast_blocks;
        if (candidates) *candidates = a.candidates;
        if (probe_ms) *probe_ms = a.probe_ms;
        if (last_tbs) *last_tbs = a.last_tbs;
    });
}

int toast_hip_accel_mem_info(size_t * free_bytes, size_t * total_bytes) {
    return guarded([&] {
        Manager::get().require_device();
        size_t f = 0, t = 0;
        TH_HIP(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f + Manager::get().cached_bytes();    // (what the cache holds can be had back)
        if (total_bytes) *total_bytes = t;
    });
}

int toast_hip_accel_release_cached(void) {
    return guarded([&] { Manager::get().release_cached(); });
}

int toast_hip_alloc_stats_ex(double * malloc_ms, double * max_malloc_ms, int64_t * budget_stops, int64_t * held_reused,
                             int64_t * held_bytes) {
    return guarded([&] {
        const AllocStats & a = alloc_stats();
        if (malloc_ms) *malloc_ms = a.malloc_ms;
        if (max_malloc_ms) *max_malloc_ms = a.max_malloc_ms;
        if (budget_stops) *budget_stops = a.budget_stops;
        if (held_reused) *held_reused = a.held_reused;
        if (held_bytes) *held_bytes = (int64_t)Manager::get().held_slow_bytes();
    });
}

int toast_hip_device_malloc(size_t nbytes, int flags, void ** out) {
    return guarded([&] {
        void * p = nullptr;
        if (flags < 0) {
            // -2: a block of exactly this size that the manager kept when it was released, if there is one
            if (flags == -2) p = Manager::get().cached_block(nbytes);
            if (p == nullptr) p = Manager::get().device_alloc(nbytes);
            if (p == nullptr) {
                Manager::get().release_cached();       // the cache may be what holds the memory
                p = Manager::get().device_alloc(nbytes);
            }
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
    return guarded([&] { TH_HIP(hipFree(p)); });
}
int toast_hip_device_release(void * p, size_t nbytes) {
    return guarded([&] {
        if (p == nullptr) return;
        if (!Manager::get().keep_block(p, nbytes)) TH_HIP(hipFree(p));
    });
}

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
