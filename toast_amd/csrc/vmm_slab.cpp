// vmm_slab.cpp -- zone-interleaved slabs for the device arena (arena.hpp).
//
// Measured on MI355X (profiles/r04_a_arena_and_hbm_zones.txt): the 288 GB of HBM3E behave as THREE zones of 96 GB
// (12-high stacks = three ranks of four dies behind every channel).  Writes and reads that meet in one zone slow each
// other down: a sweep that reads AND writes ~1000 rows of a timestream at once -- scan_map, noise_weight, the template
// projections -- runs at 5.05 TB/s when the timestream lies inside one zone and at 6.1-6.2 TB/s when its rows are spread
// over two; a scatter (build_noise_weighted) takes 5.84 ms when the map it adds to lies in the zone of the arrays it
// streams and 5.22 ms when it lies in another.  Read-only sweeps do not care.  A plain hipMalloc is physically contiguous,
// so a timestream-sized block sits inside one zone unless it happens to straddle a boundary: the "slow allocations" of
// rounds 1-3, which the round-3 placement policy could only pick among.
//
// Here a slab is a virtual range built from separately created 1 GB physical chunks (hipMemCreate / hipMemMap), mapped
// so that chunks of two zones ALTERNATE along the range -- even slots: the zone of the arena's read-mostly slabs ("P"),
// odd slots: another zone.  A streamed block of 2 GB or more then has its rows in both zones wherever the arena puts it
// (caller-visible layouts, the reference's [detector][sample] arrays, stay as they are), and a scatter target is
// placed inside one odd chunk (Arena::alloc_striped).  Only those two kinds of block live here: read-only sweeps are
// ~5 % slower on chunk-mapped memory than on a plain slab, for any chunk size (r04_a section 6).
//
// The zone of a chunk cannot be asked for, so it is measured: each new chunk gets one read + write pass together with a
// reference range (1 GB borrowed from the read-mostly slabs; the slab's own first chunk when there are none yet), rows
// dealt alternately to the two (probe_stream_split_ms, ~1 ms); the pass runs at the slow level when both lie in one
// zone.  Every chunk is measured at an address of its own (section 5: a chunk mapped where another one sat a moment
// before shows the earlier chunk's level).  Chunks are created until both classes have filled their half of the slots
// (the driver hands out one zone after the other, so this can mean creating many more chunks than the slab needs; the
// surplus is released before the function returns), or until `TOAST_HIP_ARENA_SEARCH_GB` (default 128) of surplus
// have been looked at -- then the remaining slots take what there is.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "runtime.hpp"

namespace toast_hip {

namespace {

struct VmmSlab {
    char * base = nullptr;
    size_t bytes = 0;
    size_t chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;   // in mapping order
};

std::map<void *, VmmSlab> g_vmm;
std::mutex g_vmm_mutex;
VmmSlabStats g_vmm_stats;

struct VmmPolicy {
    bool on = true;
    size_t chunk = size_t(1) << 30;
    size_t min_slab = size_t(4) << 30;
    size_t search = size_t(128) << 30;
    double gap = 0.035;      // two rates this far apart (relative) belong to different levels
    double search_ms = 500.0;   // TOAST_HIP_ARENA_SEARCH_MS: what the candidate search may cost
};
const VmmPolicy & policy() {
    static const VmmPolicy p = [] {
        VmmPolicy v;
        if (const char * e = std::getenv("TOAST_HIP_ARENA_INTERLEAVE")) v.on = !(e[0] == '0');
        if (const char * e = std::getenv("TOAST_HIP_ARENA_CHUNK_MB")) {
            if (std::atol(e) >= 64) v.chunk = (size_t)std::atol(e) << 20;
        }
        if (const char * e = std::getenv("TOAST_HIP_ARENA_INTERLEAVE_MIN_GB")) {
            if (std::atof(e) > 0.0) v.min_slab = (size_t)(std::atof(e) * 1073741824.0);
        }
        if (const char * e = std::getenv("TOAST_HIP_ARENA_SEARCH_GB")) {
            if (std::atof(e) >= 0.0) v.search = (size_t)(std::atof(e) * 1073741824.0);
        }
        if (const char * e = std::getenv("TOAST_HIP_ARENA_SEARCH_MS")) {
            if (std::atof(e) >= 0.0) v.search_ms = std::atof(e);
        }
        if (const char * e = std::getenv("TOAST_HIP_ARENA_ZONE_GAP")) {
            if (std::atof(e) > 0.0) v.gap = std::atof(e);
        }
        return v;
    }();
    return p;
}

double ms_since(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}

bool map_chunk(char * va, size_t chunk, hipMemGenericAllocationHandle_t h, int dev) {
    if (hipMemMap(va, chunk, 0, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, chunk, &acc, 1) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemUnmap(va, chunk);
        return false;
    }
    return true;
}

}  // namespace

VmmSlabStats vmm_slab_stats() {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    return g_vmm_stats;
}

namespace {

void register_slab(char * base, size_t n, size_t chunk, const std::vector<hipMemGenericAllocationHandle_t> & slot,
                   size_t other, size_t created, size_t probes, double level, std::chrono::steady_clock::time_point t_start,
                   const char * how) {
    VmmSlab s;
    s.base = base;
    s.bytes = n * chunk;
    s.chunk = chunk;
    s.handles = slot;
    {
        std::lock_guard<std::mutex> lock(g_vmm_mutex);
        g_vmm[base] = s;
        ++g_vmm_stats.slabs;
        g_vmm_stats.chunks += (int64_t)n;
        g_vmm_stats.chunks_other_zone += (int64_t)other;
        g_vmm_stats.created += (int64_t)created;
        g_vmm_stats.probes += (int64_t)probes;
        g_vmm_stats.build_ms += ms_since(t_start);
        g_vmm_stats.same_zone_tbs = level / 1.0e9;
    }
    if (const char * e = std::getenv("TOAST_HIP_TRACE")) {
        if (e[0] != '\0' && e[0] != '0') {
            std::fprintf(stderr, "[toast_hip] vmm slab      %zu chunks of %zu MB at %p (%s): %zu in the other zone, %zu created, %.1f ms\n",
                         n, chunk >> 20, (void *)base, how, other, created, ms_since(t_start));
        }
    }
}

}  // namespace

void * vmm_slab_take(size_t bytes, hipStream_t st) {
    const VmmPolicy & pol = policy();
    if (!pol.on || bytes < pol.min_slab) return nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
        (void)hipGetLastError();
        return nullptr;
    }
    const size_t chunk = (pol.chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    if (n < 4) return nullptr;
    // the final range
    void * va = nullptr;
    if (hipMemAddressReserve(&va, n * chunk, chunk, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    char * base = static_cast<char *>(va);
    // Every candidate is measured at an address of its own: a pass over a chunk mapped where ANOTHER chunk sat a moment
    // ago showed the level of the earlier chunk (profiles/r04_a section 5) -- addresses are never reused here.
    char * own_va = nullptr;
    const size_t own_n = n + pol.search / chunk + 1;
    {
        void * t = nullptr;
        if (hipMemAddressReserve(&t, own_n * chunk, chunk, nullptr, 0) == hipSuccess) own_va = static_cast<char *>(t);
    }
    if (own_va == nullptr) {
        (void)hipGetLastError();
        (void)hipMemAddressFree(va, n * chunk);
        return nullptr;
    }
    // Candidates: every chunk created, with the rate of its pass together with chunk 0 (which sits in slot 0 from the
    // start).  The classes are decided on the whole set, not chunk by chunk: the levels move a few per cent from box to
    // box and with the clocks, their ratio (~1.13 for 1 GB ranges) does not.
    struct Cand {
        hipMemGenericAllocationHandle_t h;
        double rate;
    };
    // What the chunks are measured against: a range in the zone of the read-mostly arrays when the arena has one (the
    // even slots then hold chunks of THAT zone, the odd slots chunks of another: scatter targets go into odd chunks,
    // Manager::device_alloc), else the slab's own first chunk.
    char * ext_ref = static_cast<char *>(zone_reference_take(chunk));
    const bool external = ext_ref != nullptr;
    std::vector<Cand> cand;
    hipMemGenericAllocationHandle_t first;
    bool have_first = false, failed = false;
    size_t probes = 0;
    // (without an external reference chunk 0 is of class A by definition and sits in slot 0 from the start)
    const size_t want_b = n / 2, want_a = n - want_b - (external ? 0 : 1);
    const size_t max_create = n + pol.search / chunk;
    double thr = 1.0e300;
    auto classify = [&] {
        // Two levels ~13 % apart, each a few per cent wide: the threshold is the middle of the largest gap between
        // neighbouring rates, once that gap is wider than anything a single level shows (3.5 %).  (A quantile does not
        // work: either class can be the small one.)
        thr = 1.0e300;
        if (cand.size() < 4) return;
        std::vector<double> r;
        for (const Cand & c : cand) r.push_back(c.rate);
        std::sort(r.begin(), r.end());
        double best = 0.0;
        for (size_t i = 1; i < r.size(); ++i) {
            const double gap = (r[i] - r[i - 1]) / r[i];
            if (gap > best && gap >= pol.gap) {
                best = gap;
                thr = 0.5 * (r[i] + r[i - 1]);
            }
        }
    };
    auto counts = [&](size_t & na, size_t & nb) {
        na = nb = 0;
        for (const Cand & c : cand) (c.rate > thr ? nb : na) += 1;
    };
    const auto t_search = std::chrono::steady_clock::now();
    while (1 + cand.size() < max_create) {
        // the search for a second zone is worth half a second, not more: on a box whose memory the driver is still
        // clearing every chunk costs 20-30 ms, and the slab is built from what there is by then
        if (1 + cand.size() >= n && ms_since(t_search) > pol.search_ms) break;
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;   // the device is full: make do with what has been created
        }
        if (!have_first && !external) {
            if (!map_chunk(base, chunk, h, dev)) {
                (void)hipMemRelease(h);
                failed = true;
                break;
            }
            first = h;
            have_first = true;
            continue;
        }
        char * where = own_va + cand.size() * chunk;
        if (!map_chunk(where, chunk, h, dev)) {
            (void)hipMemRelease(h);
            failed = true;
            break;
        }
        void * two[2] = {external ? ext_ref : base, where};
        double ms = 0.0;
        try {
            ms = probe_stream_split_ms(two, 2, chunk, st);     // (best of three passes: the first one touches the chunk)
        } catch (const Error &) {
            // a failed launch or event: nothing may leak -- this chunk, and everything collected so far below
            (void)hipMemUnmap(where, chunk);
            (void)hipMemRelease(h);
            failed = true;
            break;
        }
        ++probes;
        (void)hipMemUnmap(where, chunk);
        cand.push_back(Cand{h, ms > 0.0 ? 4.0 * (double)chunk / ms : 0.0});
        classify();
        size_t na, nb;
        counts(na, nb);
        if (na >= want_a && nb >= want_b) break;
    }
    zone_reference_release(ext_ref);
    const size_t created = cand.size() + (have_first ? 1 : 0);
    size_t have_a = 0, have_b = 0;
    std::vector<hipMemGenericAllocationHandle_t> slot(n);
    std::vector<char> filled(n, 0);
    if (have_first) {
        slot[0] = first;
        filled[0] = 1;
        have_a = 1;
    }
    if (!failed && ((!have_first && !external) || created < n)) failed = true;   // not enough memory for the slab
    if (!failed) {
        // even slots: class A, odd slots: the others; what one class cannot fill, the other does -- the LAST created
        // first: the driver hands out one zone after the other, so a late chunk is the most likely to differ from chunk 0
        std::vector<size_t> ia, ib;
        for (size_t k = 0; k < cand.size(); ++k) (cand[k].rate > thr ? ib : ia).push_back(k);
        std::vector<size_t> even, odd;      // candidates for the even slots 2, 4, ... and the odd slots 1, 3, ...
        const size_t take_a = std::min(ia.size(), want_a), take_b = std::min(ib.size(), want_b);
        even.assign(ia.begin(), ia.begin() + (long)take_a);
        odd.assign(ib.begin(), ib.begin() + (long)take_b);
        for (size_t j = ia.size(); j > take_a && odd.size() < want_b; --j) odd.push_back(ia[j - 1]);
        for (size_t j = ib.size(); j > take_b && even.size() < want_a; --j) even.push_back(ib[j - 1]);
        std::vector<char> used(cand.size(), 0);
        if (even.size() < want_a || odd.size() < want_b) failed = true;
        const size_t shift = have_first ? 1 : 0;      // even[] starts at slot 2 when chunk 0 already sits in slot 0
        for (size_t k = shift; k < n && !failed; ++k) {
            const size_t pick = (k % 2 == 1) ? odd[k / 2] : even[k / 2 - shift];
            used[pick] = 1;
            if (!map_chunk(base + k * chunk, chunk, cand[pick].h, dev)) {
                failed = true;
                break;
            }
            slot[k] = cand[pick].h;
            filled[k] = 1;
            (cand[pick].rate > thr ? have_b : have_a) += 1;
        }
        if (failed) {
            for (size_t k = 0; k < n; ++k) {
                if (filled[k]) (void)hipMemUnmap(base + k * chunk, chunk);
            }
            for (const Cand & c : cand) (void)hipMemRelease(c.h);
            if (have_first) (void)hipMemRelease(first);
            (void)hipMemAddressFree(va, n * chunk);
            (void)hipMemAddressFree(own_va, own_n * chunk);
            return nullptr;
        }
        for (size_t j = 0; j < cand.size(); ++j) {
            if (!used[j]) (void)hipMemRelease(cand[j].h);
        }
    } else {
        if (have_first) {
            (void)hipMemUnmap(base, chunk);
            (void)hipMemRelease(first);
        }
        for (const Cand & c : cand) (void)hipMemRelease(c.h);
        (void)hipMemAddressFree(va, n * chunk);
        (void)hipMemAddressFree(own_va, own_n * chunk);
        return nullptr;
    }
    (void)hipMemAddressFree(own_va, own_n * chunk);
    const double self_rate = thr < 1.0e299 ? thr : 0.0;     // (reported: the level that separates the classes)
    if (const char * e = std::getenv("TOAST_HIP_TRACE")) {
        if (e[0] != '\0' && e[0] != '0') {
            std::string line;
            for (const Cand & c : cand) {
                char buf[32];
                std::snprintf(buf, sizeof buf, " %.2f", c.rate / 1.0e9);
                line += buf;
            }
            std::fprintf(stderr, "[toast_hip] vmm rates (TB/s, creation order; threshold %.2f):%s\n", thr < 1.0e299 ? thr / 1.0e9 : 0.0,
                         line.c_str());
        }
    }
    register_slab(base, n, chunk, slot, have_b, created, probes, self_rate, t_start, "chunk-by-chunk search");
    return base;
}

// EXPERIMENT (tools/exp_vmm_matrix.py, profiles/r04_a): is the slow / fast level of a pair of 1 GB ranges a property of
// the PHYSICAL chunks or of the VIRTUAL addresses they are mapped at?  n_phys chunks are created; chunk 0 stays at slot 0
// of a reserved range; for every other chunk j and every slot v in [1, n_slots) the chunk is mapped at slot v, the pair
// (slot 0, slot v) gets one split pass, and the chunk is unmapped again.  out[j * n_slots + v] = TB/s.
void vmm_pair_matrix(int n_phys, int n_slots, double * out, hipStream_t st) {
    int dev = 0;
    TH_HIP(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    const size_t chunk = size_t(1) << 30;
    void * va = nullptr;
    TH_HIP(hipMemAddressReserve(&va, (size_t)n_slots * chunk, chunk, nullptr, 0));
    char * base = static_cast<char *>(va);
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)n_phys);
    for (int j = 0; j < n_phys; ++j) TH_HIP(hipMemCreate(&h[(size_t)j], chunk, &prop, 0));
    if (!map_chunk(base, chunk, h[0], dev)) fail_arg("vmm_pair_matrix: map failed");
    for (int j = 1; j < n_phys; ++j) {
        for (int v = 1; v < n_slots; ++v) {
            char * where = base + (size_t)v * chunk;
            if (!map_chunk(where, chunk, h[(size_t)j], dev)) fail_arg("vmm_pair_matrix: map failed");
            void * two[2] = {base, where};
            const double ms = probe_stream_split_ms(two, 2, chunk, st);
            out[(size_t)j * (size_t)n_slots + (size_t)v] = ms > 0.0 ? 4.0 * (double)chunk / ms / 1.0e9 : 0.0;
            (void)hipMemUnmap(where, chunk);
        }
    }
    (void)hipMemUnmap(base, chunk);
    for (auto hh : h) (void)hipMemRelease(hh);
    (void)hipMemAddressFree(va, (size_t)n_slots * chunk);
    std::fprintf(stderr, "[toast_hip] vmm_pair_matrix: range at %p\n", va);
}

size_t vmm_slab_size(void * p) {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    auto it = g_vmm.find(p);
    return it == g_vmm.end() ? 0 : it->second.bytes;
}

bool vmm_slab_give(void * p) {
    VmmSlab s;
    {
        std::lock_guard<std::mutex> lock(g_vmm_mutex);
        auto it = g_vmm.find(p);
        if (it == g_vmm.end()) return false;
        s = it->second;
        g_vmm.erase(it);
        --g_vmm_stats.slabs;
    }
    for (size_t k = 0; k < s.handles.size(); ++k) {
        (void)hipMemUnmap(s.base + k * s.chunk, s.chunk);
        (void)hipMemRelease(s.handles[k]);
    }
    (void)hipMemAddressFree(s.base, s.bytes);
    return true;
}

}  // namespace toast_hip
