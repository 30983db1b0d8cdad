// runtime.hpp -- host-side runtime of libtoast_hip: error plumbing, device selection, the
// host-pointer -> device-pointer memory manager (counterpart of the reference's OmpManager,
// /root/reference/src/toast/_libtoast/accelerator.hpp:73-159) and the per-call parameter
// block cache.  HIP only (gfx950); there is no host fallback anywhere in this library.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/toast_hip.h"
#include "arena.hpp"

namespace toast_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string & m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string & msg);

#define TH_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::ostringstream o_;                                                            \
            o_ << "HIP error " << hipGetErrorName(e_) << " (" << hipGetErrorString(e_)        \
               << ") at " << __FILE__ << ":" << __LINE__ << " in " #expr;                     \
            throw ::toast_hip::Error(TOAST_HIP_ERR_DEVICE, o_.str());                         \
        }                                                                                     \
    } while (0)

// TOAST_HIP_TRACE=2: every entry point is followed by a device synchronisation and logged with its wall time and its
// start time (stderr), i.e. a serialised timeline of the calls.  call_trace_begin() returns < 0 when this is off.
double call_trace_begin() noexcept;
void call_trace_end(const char * fn, double t0) noexcept;

// Wrap the body of an extern "C" entry point.
template <typename F>
int guarded(F && f, const char * fn = __builtin_FUNCTION()) noexcept {
    try {
        const double t0 = call_trace_begin();
        f();
        if (t0 >= 0) call_trace_end(fn, t0);
        return TOAST_HIP_OK;
    } catch (const Error & e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception & e) {
        set_last_error(e.what());
        return TOAST_HIP_ERR_ARG;
    } catch (...) {
        set_last_error("unknown error");
        return TOAST_HIP_ERR_ARG;
    }
}

[[noreturn]] void fail_arg(const std::string & msg);

// ------------------------------------------------------------------ work decomposition
// A chunk is a run of <= kChunk consecutive samples inside one interval; a workgroup
// processes one (chunk, detector) pair.  16 bytes.
constexpr int kChunk = 1024;
struct Chunk {
    int64_t first;   // first sample
    int32_t count;   // number of samples (1..kChunk)
    int32_t view;    // interval index
};

// ------------------------------------------------------------------ parameter blocks
// Small per-call arrays (index arrays, chunk list, per-detector scalars) are packed into
// one host block, deduplicated against recently used blocks and uploaded at most once:
// a PCG loop re-issuing the same call pays no host->device traffic after the first pass.
class ParamBlock {
public:
    // Append `bytes` from `src` aligned to 16; returns the byte offset inside the block.
    size_t push(const void * src, size_t bytes);
    template <typename T>
    size_t push_vec(const std::vector<T> & v) {
        return push(v.data(), v.size() * sizeof(T));
    }
    // Upload (or find cached) on `stream`; returns the device base pointer.
    const char * commit(hipStream_t stream);

private:
    std::vector<char> host_;
};

size_t pin_threshold();
void drop_param_blocks();   // forget the cached parameter blocks (their storage is about to go away)
// kernels.hip: time of a read + write pass over a new device block with the timestream kernels' access pattern
double probe_stream_ms(void * block, size_t bytes, hipStream_t stream);
// ... with the 1024 rows dealt round-robin to nb <= 4 separate ranges of bytes_each (placement experiments)
// (timed by the device's constant-rate clock inside the kernel; *used_clock = false: HIP events had to do)
// the byte mix of scan_map (out != nullptr) / build_noise_weighted (out == nullptr) as plain streams in the launch shape of the
// *_v2 kernels (kernels.hip k_probe_byte_mix); asynchronous on st
void probe_byte_mix(const int64_t * pixels, const double * weights, const double * tod, double * out, int64_t n_det,
                    int64_t n_samp, hipStream_t st);
double probe_stream_split_ms(void * const * bases, int nb, size_t bytes_each, hipStream_t stream, bool * used_clock = nullptr);
// Counters of Manager::device_alloc: both arenas (arena.hpp) together, plus what went around them.
struct AllocStats {
    int64_t slabs = 0;             // slabs held now
    size_t slab_bytes = 0;
    size_t used_bytes = 0;         // bytes in live blocks
    size_t peak_used_bytes = 0;
    int64_t slab_mallocs = 0;      // hipMalloc calls for slabs so far
    int64_t slab_frees = 0;
    double malloc_ms = 0.0;        // wall time inside hipMalloc (slabs and direct blocks)
    double max_malloc_ms = 0.0;    // ... the longest single call (a driver still clearing memory stalls one for seconds)
    double touch_ms = 0.0;
    int64_t allocs = 0;            // blocks handed out
    int64_t releases = 0;
    int64_t direct_mallocs = 0;    // blocks that went straight to hipMalloc (TOAST_HIP_ALLOC=plain)
    int64_t failed = 0;
};
AllocStats alloc_stats();

// Rank-interleaved slabs (vmm_slab.cpp): nullptr when switched off, too small, or the driver refuses.
void * vmm_slab_take(size_t bytes, hipStream_t stream);
// Device ranges of `bytes` at the START and at the END of the read-mostly slabs' free space, for the slab builder to
// measure chunks against (runtime.cpp; nullptr: none to be had -- the slab's first chunk is the reference then);
// borrowed from the arena, given back with zone_references_release.
struct ZoneRefs {
    void * first = nullptr;
    void * last = nullptr;
};
ZoneRefs zone_references_take(size_t bytes);
void zone_references_release(const ZoneRefs & r);
// For the zone survey of the read-mostly range (vmm_slab.cpp): the largest free range of the arena the read-mostly arrays
// come from -- where they are going to lie, in address order -- (false: no such slab yet, or a token reservation), a block
// of it at a given offset (nullptr when somebody else's allocation got there first), and its return.
bool read_mostly_free_range(const char ** base, size_t * lo, size_t * hi);
void * read_mostly_take_at(const char * base, size_t offset, size_t bytes);
void read_mostly_release(void * p);
bool vmm_slab_give(void * p);       // false: not one of them
double vmm_zone_threshold(const double * rates, int n, double level);      // the zone search's class threshold on given rates
struct VmmSlabStats {
    int64_t slabs = 0;               // interleaved slabs alive
    int64_t chunks = 0;              // chunks mapped into slabs so far
    int64_t chunks_other_zone = 0;   // ... of which in a zone other than their slab's first chunk
    int64_t chunks_other_wanted = 0; // ... of which the pattern P Q Q P asks for (half of `chunks`)
    int64_t created = 0;             // chunks created (the surplus was released)
    int64_t probes = 0;
    double build_ms = 0.0;
    double same_zone_tbs = 0.0;      // the last slab's reference rate (two halves of one chunk)
    // how the candidate searches ended: the search is budgeted in PROBES
    // (TOAST_HIP_ARENA_SEARCH_PROBES) and candidate bytes (TOAST_HIP_ARENA_SEARCH_GB, at most half of the free memory of
    // this process' share of the device); TOAST_HIP_ARENA_SEARCH_MS is a hard cap only
    int64_t searches = 0;            // slabs whose chunks were searched for
    int64_t searches_exhausted = 0;  // ... that ran out of a budget before both classes were full
    int64_t searches_capped_ms = 0;  // ... that hit the hard cap in ms (a subset of the exhausted ones)
    int64_t probes_by_clock = 0;     // probes timed by the device clock (the rest: HIP events)
    int64_t read_mostly_zones = 0;   // zones the last survey found in the read-mostly range (0: none surveyed)
    int64_t slabs_third_zone = 0;    // slabs whose BOTH chunk classes are clear of the read-mostly slab (two other zones)
    double create_ms_per_chunk = 0.0;   // the last search's average hipMemCreate time per chunk (0.1: clean memory; 20-50: the
                                        // driver is still clearing what another process returned)
    double search_ms = 0.0;          // wall time of the searches (part of build_ms)
};
VmmSlabStats vmm_slab_stats();
// slot k of an interleaved slab holds a chunk of the OTHER zone (pattern P Q Q P P Q Q P ...)
inline bool vmm_slot_other(size_t k) { return (((k + 1) >> 1) & 1) != 0; }
// is `base` the start of an interleaved slab?  its chunk size
// n_ref: 0 when the slab's classes are relative to its own first chunk (built before any read-mostly slab existed)
bool vmm_slab_layout(const void * base, size_t * chunk, int * n_ref = nullptr);
// which class of a slab's chunks its scatter targets go to: -1 not decided yet, 0 = P (even slots), 1 = Q; forgotten with the slab
int vmm_slab_scatter_class(const void * base);
void vmm_slab_set_scatter_class(const void * base, int cls);
// processes that share this device (assign_device): the transient candidate search is capped by this process' share
void vmm_set_device_share(int per);
void vmm_pair_matrix(int n_phys, int n_slots, double * out, hipStream_t st);   // experiment, vmm_slab.cpp

// Transfers between pageable application memory and the device, through a page-locked bounce ring owned by the library
// (two 4 MB slots, copy of slot k overlapped with the host memcpy of slot k+1).  The HIP runtime never sees pageable
// application memory: handing it such a pointer makes the driver register the range for device access for the duration
// of the copy, and a later change of that mapping by the allocator (free, heap growth) stalls the NEXT device operation
// of the process by 20-30 ms (profiles/r02_j).  TOAST_HIP_BOUNCE=0 restores direct hipMemcpyAsync calls.
//   copy_to_device: returns when the source has been consumed (the device copy is stream-ordered).
//   copy_to_host:   returns when the destination holds the data.
void copy_to_device(void * dev, const void * host, size_t bytes, hipStream_t stream);
void copy_to_host(void * host, const void * dev, size_t bytes, hipStream_t stream);
int chunk_size();
bool det_major_grid();
void set_det_major_grid(int on);
bool pair_detectors();
void set_pair_detectors(int on);
bool vec2_lanes();
void set_vec2_lanes(int on);
// NaN Q / U weights where the reference's formulation produces them (hpix_math.hpp: stokes_cs2alpha): the default;
// TOAST_HIP_STOKES_REFERENCE_NAN=0 / toast_hip_set_stokes_reference_nan(0) selects the finite form.
bool stokes_reference_nan();
// TOAST_HIP_DETERMINISTIC=1 / toast_hip_set_deterministic(): order-deterministic A^T scatter
// (deterministic.hip) instead of the atomic kernels.
bool deterministic_mode();
void deterministic_scatter(int mode, const int64_t * d_g2l, double * d_out, int64_t n_pix_submap, int64_t nnz,
                           const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index,
                           const double * d_weights, const int32_t * data_index, const double * d_det_data,
                           const int32_t * flag_index, const uint8_t * d_det_flags, int use_d,
                           const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
                           const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
                           int use_s, uint8_t shared_flag_mask, hipStream_t st);
std::vector<Chunk> make_chunks(const toast_hip_interval * ivl, int64_t n_view, int64_t n_samp);

// ------------------------------------------------------------------ memory manager
class Manager {
public:
    static Manager & get();

    void assign_device(int node_procs, int node_rank, double mem_gb, bool disabled);
    int device();               // throws if not yet assigned; -1 when disabled
    void require_device();      // throws unless a GPU is assigned and usable; hipSetDevice
    int present(const void * host, size_t nbytes);
    // kind: what the array is to the kernels (device_alloc)
    void * create(const void * host, size_t nbytes, const char * name, int kind = 0);
    void adopt(const void * host, size_t nbytes, void * device, const char * name);
    void reset(const void * host, size_t nbytes, const char * name);
    void update_device(const void * host, size_t nbytes, const char * name);
    // Upload in parts on a stream of its own: returns once the copies are ENQUEUED (part k ends at byte part_end[k]);
    // update_device_wait makes `stream` wait for part k (no host blocking), update_device_finish waits for all of
    // them.  Lets kernels work on the first rows while the later ones are still crossing PCIe.
    void update_device_parts(const void * host, size_t nbytes, const char * name, const size_t * part_end, int n_parts);
    void update_device_wait(const void * host, int part, hipStream_t stream);
    int update_device_arrived(const void * host, int part);
    void update_device_finish(const void * host);
    void update_host(void * host, size_t nbytes, const char * name);
    void remove(const void * host, size_t nbytes, const char * name);
    void * device_ptr(const void * host);   // throws if absent
    void * find(const void * host);         // nullptr if absent
    void dump();
    void clear();
    // Grow-only device scratch (slot = kScratch*): FFT work buffers, reduction results.
    // `user`: the stream the caller's kernels on this buffer run on.  When every use of a slot has named its stream, growing
    // the slot waits for that stream only -- not for the whole device, which would also wait for an upload of gigabytes
    // that is crossing PCIe on the upload stream (ops.NoiseFilter; profiles/r04_f).
    void * scratch(int slot, size_t bytes);
    void * scratch(int slot, size_t bytes, hipStream_t user);
    // Every device block of the library comes from here and goes back through device_free: a range of an arena slab
    // (arena.hpp), or -- TOAST_HIP_ALLOC=plain -- one hipMalloc / hipFree per block.  nullptr on failure.
    // `kind` says what the block is to the kernels, which decides WHERE in HBM it should live (vmm_slab.cpp: the 288 GB
    // behave as three zones; reads and writes that meet in one zone slow each other down):
    //   kBlockDefault  read-mostly arrays (pixels, weights, flags, the packed cache): plain slabs -- one zone, "P"
    //   kBlockStreamed a timestream that sweeps read AND write (scan_map, noise_weight, template projections): a slab
    //                  whose 1 GB chunks alternate between zone P and another zone, so that its ~1000 rows in flight are
    //                  spread over two zones (6.1 instead of 5.1 TB/s for such sweeps)
    //   kBlockScatter  the target of a scatter with atomics (maps, amplitude vectors): inside ONE chunk of the other
    //                  zone, away from the streams that feed the scatter (build_noise_weighted 5.2 instead of 5.8 ms)
    // Streamed / scatter blocks come from what has been reserved for them (reserve(bytes, true)); without a
    // reservation, and below 1 GB for streamed blocks, they are default blocks.
    static constexpr int kBlockDefault = 0, kBlockStreamed = 1, kBlockScatter = 2;
    void * device_alloc(size_t nbytes, int kind = 0);
    // p from device_alloc (anything else is handed to hipFree).  The caller guarantees that nothing still queued uses it
    // on a stream other than stream(): the next owner's work is ordered after it on that stream.
    void device_free(void * p);
    // slabs without live blocks go back to the driver; returns the bytes
    size_t release_cached();
    // free bytes inside the slabs (what device_alloc can serve without the driver)
    size_t cached_bytes() const;
    // take `bytes` for the arena now (one slab, touched once): assign_device(mem_gb), TOAST_HIP_ARENA_RESERVE_GB
    void reserve(size_t bytes, bool streamed = false);
    void drop_arenas();   // device change: every slab goes back to the driver
    // the streamed slab that assign_device started building on a thread of its own is complete (no-op otherwise)
    static void wait_for_builder();

    uint64_t generation() const { return generation_; }
    hipStream_t stream() const { return stream_; }
    void set_stream(hipStream_t s) { stream_ = s; }

    static constexpr int kScratchFftTime = 0, kScratchFftFreq = 1, kScratchDot = 2, kScratchFftWork = 3,
                         kScratchSort = 4, kScratchFftImpulse = 5, kScratchCommA = 6, kScratchCommB = 7, kScratchStatus = 8;

private:
    std::map<std::pair<int, int>, std::pair<void *, size_t>> scratch_;   // (device, slot) -> (ptr, bytes)
    // (device, slot) -> (every use so far named its stream, the last one named)
    std::map<std::pair<int, int>, std::pair<bool, hipStream_t>> scratch_user_;
    void * scratch_impl(int slot, size_t bytes, bool named, hipStream_t user);
    struct Entry {
        void * dev;
        size_t nbytes;
        std::string name;
        bool owned = true;
        bool host_registered = false;
        bool pin_failed = false;
        std::vector<size_t> pin_ends;        // page-locked in several ranges (upload in parts): their end offsets; empty = one range
        std::vector<hipEvent_t> part_done;   // update_device_parts: one event per enqueued part
    };
    hipStream_t upload_stream_ = nullptr;   // non-blocking: does not synchronise with the default stream
    void pin_for_transfer(const void * host, Entry & e);
    // hipMemcpyAsync of [off, off + len) of a page-locked entry, split where its page-locked ranges meet
    static void copy_pinned(const Entry & e, void * dev, void * host, size_t off, size_t len, bool to_device, hipStream_t st);
    static void unpin(const void * host, Entry & e);
    static double trace_begin();
    static void trace(const char * what, const std::string & name, size_t nbytes, double t0);
    Entry & lookup(const void * host, size_t nbytes, const char * name, const char * what);

    std::unordered_map<const void *, Entry> table_;
    bool assigned_ = false;
    size_t owned_bytes_ = 0;
    uint64_t generation_ = 0;
    int device_ = -1;
    hipStream_t stream_ = nullptr;
};

// Resolve one array of a host-level call.  use_accel: device copy must be registered.
// Otherwise: stage a temporary device copy (optionally uploading the host contents) that is
// copied back (if `out`) and released by finish().
class Staging {
public:
    Staging(bool use_accel, hipStream_t stream) : accel_(use_accel), stream_(stream) {}
    ~Staging();
    template <typename T>
    T * in(const T * host, size_t count) {
        return static_cast<T *>(resolve(const_cast<T *>(host), count * sizeof(T), true, false));
    }
    template <typename T>
    T * inout(T * host, size_t count) {
        return static_cast<T *>(resolve(host, count * sizeof(T), true, true));
    }
    template <typename T>
    T * out(T * host, size_t count) {
        // outputs are partially written (intervals only): keep the host contents elsewhere
        return static_cast<T *>(resolve(host, count * sizeof(T), true, true));
    }
    // Always staged (never looked up in the manager), e.g. hit_submaps.
    template <typename T>
    T * temp_inout(T * host, size_t count) {
        return static_cast<T *>(resolve(host, count * sizeof(T), true, true, true));
    }
    // a small per-call host array that is never part of the registered data (uploaded to a temporary even with
    // use_accel)
    template <typename T>
    const T * temp_in(const T * host, size_t count) {
        return static_cast<const T *>(resolve(const_cast<T *>(host), count * sizeof(T), true, false, true));
    }
    void finish();  // copy outputs back, free temporaries, synchronise when anything was staged

private:
    void * resolve(void * host, size_t bytes, bool upload, bool download, bool force_temp = false);
    struct Temp {
        void * host;
        void * dev;
        size_t bytes;
        bool download;
        bool registered;   // host range page-locked for the duration of the call (large buffers)
    };
    bool accel_;
    hipStream_t stream_;
    std::vector<Temp> temps_;
    bool finished_ = false;
};

}  // namespace toast_hip
