// arena.hpp -- the device memory arena of libtoast_hip.
//
// Reference counterpart: OmpPoolResource, /root/reference/src/toast/_libtoast/accelerator.cpp:13-230 (one pool of
// `mem_gb` per process taken at assign_device, :233-306 -- dormant upstream: the constructor call is commented out).
// Same role, different shape: the pool here is a set of SLABS taken from the driver with one hipMalloc each, touched
// once, and never given back while the process computes; every device block the library hands out -- registered
// arrays, temporaries, the packed pointing cache, FFT work space, parameter-block storage, scratch -- is a range of a
// slab.  After set-up no operator phase calls hipMalloc or hipFree: on this driver a hipMalloc costs 0.3 .. 300 ms
// per GB depending on how much freed memory it still has to clear, a first touch 12 .. 17 ms per GB, and a hipFree
// synchronises the device.
//
// Sub-allocation: address-ordered free ranges per slab, best fit over all slabs (smallest range that holds the request;
// ties: lowest address), split on allocation, merged with both neighbours on release.  Two instances with different
// granules keep the kilobyte-sized blocks (interval lists, per-detector scalars) out of the ranges the gigabyte-sized
// ones need: Manager::device_alloc sends requests below 1 MB to the small arena.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace toast_hip {

struct ArenaStats {
    int64_t slabs = 0;            // slabs currently held
    size_t slab_bytes = 0;        // ... their bytes
    size_t used_bytes = 0;        // bytes in live blocks (granule-rounded)
    size_t peak_used_bytes = 0;
    int64_t slab_mallocs = 0;     // hipMalloc calls so far (slabs ever taken)
    int64_t slab_frees = 0;       // slabs given back (trim / destroy)
    double malloc_ms = 0.0;       // wall time inside those hipMalloc calls
    double max_malloc_ms = 0.0;   // ... the longest one
    double touch_ms = 0.0;        // host time spent in the first-touch fills (waited for: blocks go to callers on any stream)
    int64_t allocs = 0;           // blocks handed out
    int64_t releases = 0;
    int64_t failed = 0;           // requests that could not be served
};

// Where slabs come from.  The product uses hip_backend(); host_backend() (plain malloc, no fill) exists so that the
// sub-allocation logic can be exercised without a GPU (toast_hip_arena_selftest, tests/test_capi_load.py).
struct ArenaBackend {
    void * (*take)(size_t bytes, hipStream_t st);             // nullptr on failure (st: where measuring passes run)
    void (*give)(void * p);
    void (*touch)(void * p, size_t bytes, hipStream_t st);    // first-touch fill on st, complete on return (may be a no-op)
};
const ArenaBackend & hip_backend();          // hipMalloc
const ArenaBackend & hip_interleaved_backend();   // rank-interleaved slabs (vmm_slab.cpp) when they can be had, else hipMalloc
const ArenaBackend & host_backend();

class Arena {
public:
    // granule: every block is a multiple of it and aligned to it; slab_default: size of a slab taken on demand when
    // the request is smaller (requests above it get a slab of their own size)
    // slab_round: slab sizes are multiples of it (the large arena: 1 GB, the chunk size of rank-interleaved slabs)
    Arena(size_t granule, size_t slab_default, const ArenaBackend & backend = hip_backend(), size_t slab_round = 0)
        : granule_(granule), slab_default_(slab_default), slab_round_(slab_round ? slab_round : granule), be_(backend) {}
    ~Arena() = default;   // (process exit: the driver reclaims the slabs; nothing may call HIP from a static destructor)

    // nullptr when neither a free range nor a new slab can hold the request.  `stream`: where the first-touch fill
    // of a new slab runs (waited for before the block is handed out: its user may work on any stream).
    // grow = false: only from the free ranges of the slabs already held
    void * alloc(size_t nbytes, hipStream_t stream, bool grow = true);
    // A block that lies inside ONE stripe of width `stripe` whose index (offset / stripe within its slab) has the given
    // parity: the slabs of the stream arena alternate between two HBM zones chunk by chunk (vmm_slab.cpp), and a map
    // -- the target of a scatter -- belongs into the zone the read-only arrays are NOT in.  Never grows the arena;
    // nullptr when no such range is free.
    void * alloc_striped(size_t nbytes, size_t stripe, int parity);
    // A block whose position inside a free range the caller chooses: place(slab base, lo, hi, need) returns the offset
    // (lo <= offset, offset + need <= hi; offsets are relative to the slab) or SIZE_MAX when [lo, hi) has no place for
    // it.  Used for blocks that belong into particular chunks of a zone-interleaved slab (runtime.cpp: scatter targets
    // inside a run of chunks of the other zone, small streamed blocks astride a zone boundary).  Never grows the arena.
    void * alloc_placed(size_t nbytes, const std::function<size_t(const char * base, size_t lo, size_t hi, size_t need)> & place);
    // A block at the END of the largest free range (never grows the arena; nullptr when no range holds it): the far
    // reference of the zone measurement (runtime.cpp zone_references_take), without a filler block in between.
    void * alloc_at_end(size_t nbytes);
    // false when p is not a live block of this arena
    bool release(void * p);
    bool owns(const void * p) const;
    // offset of an address inside its slab; false when it lies in none
    bool slab_offset(const void * p, size_t * off) const;
    // make the capacity (free + used) at least `bytes` by taking ONE more slab for the difference; false if the
    // driver refuses.  contiguous: make the LARGEST FREE RANGE at least `bytes` instead (a slab of that size when no
    // range is) -- for arenas that serve a few large blocks each of which has to fit one range
    bool reserve(size_t bytes, hipStream_t stream, bool contiguous = false);
    // give slabs without live blocks back to the driver; returns the bytes released
    size_t trim();
    // forget everything (all slabs are freed, live blocks included): device change / tests
    void destroy();
    size_t free_bytes() const;
    size_t capacity() const;
    std::vector<const char *> slab_bases() const;
    size_t largest_free() const;
    ArenaStats stats() const;
    // consistency of the bookkeeping (ranges tile every slab, free ranges are merged, counters add up); "" if sound
    std::string check() const;
    void set_slab_default(size_t b) { slab_default_ = b; }
    size_t slab_default() const { return slab_default_; }

private:
    struct Slab {
        char * base = nullptr;
        size_t bytes = 0;
        std::map<size_t, size_t> free;   // offset -> length of the free ranges (address order, never adjacent)
        std::map<size_t, size_t> live;   // offset -> length of the blocks handed out
    };
    // A new slab is taken from the driver -- and touched, and for an interleaved slab measured chunk by chunk: 0.05 .. 8 s --
    // with only grow_mutex_ held, then adopted under mutex_: release / owns / free_bytes / alloc from existing ranges on
    // other threads do not wait for it.
    bool grow_by(size_t bytes, hipStream_t stream);
    void * carve_best(size_t need);       // best fit over the free ranges (mutex_ held); nullptr when none holds `need`
    size_t round_up(size_t n) const { return (n + granule_ - 1) / granule_ * granule_; }
    size_t round_slab(size_t n) const { return (n + slab_round_ - 1) / slab_round_ * slab_round_; }

    size_t granule_;
    size_t slab_default_;
    size_t slab_round_;
    ArenaBackend be_;
    std::vector<Slab> slabs_;
    ArenaStats st_;
    mutable std::mutex mutex_;
    std::mutex grow_mutex_;      // one slab at a time is being taken (never held together with a wait for mutex_'s holder)
};

}  // namespace toast_hip
