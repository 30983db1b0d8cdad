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
// in the pattern P Q Q P P Q Q P ... along the range -- "P": the zone of the arena's read-mostly slabs, "Q": another zone
// (vmm_slot_other: slots 1, 2, 5, 6, ...).  A streamed block of more than 2 GB then has its rows in both zones wherever
// the arena puts it (caller-visible layouts, the reference's [detector][sample] arrays, stay as they are), a smaller one
// is laid astride a P | Q boundary, and a scatter target of up to 2 GB is placed inside one Q Q run (round 4 alternated
// single chunks: the 1.2 GB map of Nside 2048 IQU straddled two zones; Manager::device_alloc).  Only those two kinds of block live here: read-only sweeps are
// ~5 % slower on chunk-mapped memory than on a plain slab, for any chunk size (r04_a section 6).
//
// The zone of a chunk cannot be asked for, so it is measured: each new chunk gets one read + write pass together with a
// reference range, rows dealt alternately to the two (probe_stream_split_ms, ~2 ms); the pass runs at the slow level when
// both lie in one zone.  The references are the FIRST and the LAST GB of the read-mostly slab (borrowed from the arena;
// the slab's own first chunk when there is none yet): a plain slab of ~50 GB straddles a zone boundary more often than
// not, and the odd slots take chunks that are clear of BOTH ends -- with three zones there is one that holds neither.
// (Against the first GB alone the maps shared their zone with the slab's upper part in two processes of three, against
// the last GB alone with its lower part: profiles/r04_a section 8.)  Every chunk is measured at an address of its own
// (section 5: a chunk mapped where another one sat a moment before shows the earlier chunk's level).  Chunks are created
// until both classes have filled their half of the slots (the surplus is released before the function returns), within
// a budget counted in PROBES (`TOAST_HIP_ARENA_SEARCH_PROBES`, default 320) and surplus bytes (`TOAST_HIP_ARENA_SEARCH_GB`,
// default 128, at most half of the free memory of this process' share of the device) -- then the remaining slots take what
// there is, late chunks first.  Rounds 4-5 boxed the search into 500 ms of WALL time: a process that started while the
// driver was still clearing its predecessor's memory (25-50 ms per hipMemCreate instead of 0.1), or under a profiler,
// ran out of the box with 1 chunk of the other zone and lost 12 % of scan_map (profiles/r05_f section 2 against the
// un-profiled line).  Now nothing that the host's clock sees decides the placement: the passes are timed by the device's
// own constant-rate clock inside the probe kernel, classes are gaps between the rates of ONE search, the budget counts
// probes, and `TOAST_HIP_ARENA_SEARCH_MS` (default 8000) is only a hard cap whose use is reported
// (toast_hip_arena_placement_status, toast_hip_arena_stats_t.searches_capped_ms).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "runtime.hpp"

namespace toast_hip {

namespace {

struct VmmSlab {
    char * base = nullptr;
    size_t bytes = 0;
    size_t chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;   // in mapping order
    int n_ref = 0;      // 0: the classes are relative to the slab's own first chunk (no read-mostly slab existed), else to it
    int scatter_class = -1;   // which class its scatter targets go to (runtime.cpp scatter_class_of): -1 not decided, 0 P, 1 Q
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
    double search_ms = 8000.0;  // TOAST_HIP_ARENA_SEARCH_MS: hard cap on the candidate search (reported when it is hit)
    long search_probes = 320;   // TOAST_HIP_ARENA_SEARCH_PROBES: the search's budget, counted in measuring passes
    bool third = true;          // TOAST_HIP_ARENA_THIRD_ZONE=0 turns off: both chunk classes from zones that hold NEITHER end of the read-mostly slab
    size_t spacer = size_t(8) << 30;   // TOAST_HIP_ARENA_SPACER_GB: plain block that steps over a run of useless chunks (0: none)
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
        if (const char * e = std::getenv("TOAST_HIP_ARENA_SEARCH_PROBES")) {
            if (std::atol(e) >= 0) v.search_probes = std::atol(e);
        }
        if (const char * e = std::getenv("TOAST_HIP_ARENA_THIRD_ZONE")) v.third = (e[0] != '0');
        if (const char * e = std::getenv("TOAST_HIP_ARENA_SPACER_GB")) {
            if (std::atof(e) >= 0.0) v.spacer = (size_t)(std::atof(e) * 1073741824.0);
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

// processes sharing this device (Manager::assign_device): the transient search may use half of the free memory of ONE share
int g_share = 1;

// TEST HOOK (tests/test_gpu_accel.py): TOAST_HIP_ARENA_TEST_SLOW_MS = host milliseconds added to every chunk creation and
// every measuring pass of a search -- a box whose driver is still clearing memory, a profiler, a loaded host.  The
// placement must come out the same (read at every search, not once: the test sets it per slab).
double test_slow_ms() {
    const char * e = std::getenv("TOAST_HIP_ARENA_TEST_SLOW_MS");
    return (e != nullptr && std::atof(e) > 0.0) ? std::atof(e) : 0.0;
}
void test_slow(double ms) {
    if (ms > 0.0) std::this_thread::sleep_for(std::chrono::microseconds((long long)(ms * 1000.0)));
}

// threshold between "same zone" and "another zone" over a set of measured rates: the middle of the largest relative gap
// between neighbours once that gap is wider than anything one level shows; with one level only, that level against `level`
// (the rate of a pass over the two halves of the reference itself): 0 = all "other", 1e300 = all "same"
double gap_threshold(std::vector<double> r, double level, double min_gap) {
    if (r.empty()) return 1.0e300;
    std::sort(r.begin(), r.end());
    double best = 0.0, thr = -1.0;
    for (size_t i = 1; i < r.size(); ++i) {
        const double gap = (r[i] - r[i - 1]) / r[i];
        // (what lies above a gap must be ABOVE the level a range shows with itself: one slow outlier below a single level
        //  -- 4.85 under 5.13 5.13 5.16 5.17, all of them in the reference's zone -- is not a second level; taken for one, it
        //  put the maps into the zone of every array the kernels read: build_noise_weighted 5.89 ms, profiles/r06_f section 6)
        if (gap > best && gap >= min_gap && r[i] > (1.0 + min_gap) * level) {
            best = gap;
            thr = 0.5 * (r[i] + r[i - 1]);
        }
    }
    if (thr > 0.0) return thr;
    return (r[r.size() / 2] > (1.0 + 1.7 * min_gap) * level) ? 0.0 : 1.0e300;
}

// TOAST_HIP_ARENA_SURVEY_REFS=0: the survey only gates the third-zone split; the chunks are measured against the two ENDS
bool survey_refs_enabled() {
    static const bool on = [] {
        const char * e = std::getenv("TOAST_HIP_ARENA_SURVEY_REFS");
        return !(e != nullptr && e[0] == '0');
    }();
    return on;
}

// ---- the zone survey of a range (round 6, profiles/r06_f) ------------------------------------------------------------------
// K blocks of `chunk` bytes spread over a range are sorted into HBM zones: every block against the first one (one read +
// write pass each: two levels ~13 % apart), the ones in another zone against the last of them.
struct ZoneSurvey {
    std::vector<int> label;      // per block: 0 / 1 / 2, -1 = could not be had
    int zones = 0;
    std::string map;             // X / Y / Z / ? per block, address order
};

std::vector<size_t> survey_offsets(size_t lo, size_t hi, size_t chunk) {
    const size_t len = hi - lo, gr = size_t(2) << 20;
    int K = (int)(len / (size_t(4) << 30)) + 1;
    K = K < 3 ? 3 : (K > 16 ? 16 : K);
    std::vector<size_t> at((size_t)K);
    for (int k = 0; k < K; ++k) at[(size_t)k] = (lo + (size_t)((double)(len - chunk) * k / (K - 1))) / gr * gr;
    at[0] = (lo + gr - 1) / gr * gr;
    return at;
}

template <class Take, class Release, class Pass>
ZoneSurvey survey_zones(int K, Take take, Release release, Pass pass, size_t chunk, double gap) {
    struct Borrowed {
        void * p;
        Release & give;
        ~Borrowed() {
            if (p != nullptr) give(p);
        }
    };
    ZoneSurvey zs;
    zs.label.assign((size_t)K, -1);
    Borrowed start{take(0), release};
    if (start.p == nullptr) return zs;
    zs.label[0] = 0;
    const double level0 = pass(start.p, static_cast<char *>(start.p) + chunk / 2, chunk / 2);
    std::vector<double> r0((size_t)K, -1.0), valid;
    for (int k = 1; k < K; ++k) {
        Borrowed b{take(k), release};
        if (b.p == nullptr) continue;
        r0[(size_t)k] = pass(start.p, b.p, chunk);
        valid.push_back(r0[(size_t)k]);
    }
    const double thr0 = gap_threshold(valid, level0, gap);
    std::vector<int> other;
    for (int k = 1; k < K; ++k) {
        if (r0[(size_t)k] < 0.0) continue;
        if (r0[(size_t)k] > thr0) other.push_back(k);
        else zs.label[(size_t)k] = 0;
    }
    if (!other.empty()) {
        // the blocks of another zone than the first one's: one zone or two?  Against the LAST of them that can be had
        int k1 = -1;
        void * p1 = nullptr;
        for (size_t j = other.size(); j-- > 0 && p1 == nullptr;) {
            p1 = take(other[j]);
            if (p1 != nullptr) k1 = other[j];
        }
        Borrowed rep1{p1, release};
        if (rep1.p != nullptr) {
            zs.label[(size_t)k1] = 1;
            const double level1 = pass(rep1.p, static_cast<char *>(rep1.p) + chunk / 2, chunk / 2);
            std::vector<double> r1((size_t)K, -1.0), valid1;
            for (int k : other) {
                if (k == k1) continue;
                Borrowed b{take(k), release};
                if (b.p == nullptr) continue;
                r1[(size_t)k] = pass(rep1.p, b.p, chunk);
                valid1.push_back(r1[(size_t)k]);
            }
            const double thr1 = gap_threshold(valid1, level1, gap);
            for (int k : other) {
                if (k != k1 && r1[(size_t)k] >= 0.0) zs.label[(size_t)k] = r1[(size_t)k] > thr1 ? 2 : 1;
            }
        }
    }
    bool seen[3] = {false, false, false};
    for (int k = 0; k < K; ++k) {
        const int z = zs.label[(size_t)k];
        if (z >= 0) seen[z] = true;
        zs.map += z < 0 ? '?' : (char)('X' + z);
    }
    zs.zones = (int)seen[0] + (int)seen[1] + (int)seen[2];
    return zs;
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

struct SearchOutcome {
    size_t want_other = 0;
    size_t probes_clock = 0;
    bool exhausted = false, capped = false, third = false;
    int zones = 0;               // zones the survey found in the read-mostly range (0: no survey)
    std::string zone_map;        // ... one letter per surveyed block
    double create_ms = 0.0, search_ms = 0.0;
};

void register_slab(char * base, size_t n, size_t chunk, const std::vector<hipMemGenericAllocationHandle_t> & slot,
                   size_t other, size_t created, size_t probes, double level, std::chrono::steady_clock::time_point t_start,
                   const char * how, int n_ref, const SearchOutcome & so) {
    VmmSlab s;
    s.base = base;
    s.bytes = n * chunk;
    s.chunk = chunk;
    s.handles = slot;
    s.n_ref = n_ref;
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
        g_vmm_stats.chunks_other_wanted += (int64_t)so.want_other;
        ++g_vmm_stats.searches;
        g_vmm_stats.searches_exhausted += so.exhausted ? 1 : 0;
        g_vmm_stats.searches_capped_ms += so.capped ? 1 : 0;
        g_vmm_stats.slabs_third_zone += so.third ? 1 : 0;
        if (so.zones > 0) g_vmm_stats.read_mostly_zones = so.zones;
        g_vmm_stats.probes_by_clock += (int64_t)so.probes_clock;
        g_vmm_stats.create_ms_per_chunk = created > 0 ? so.create_ms / (double)created : 0.0;
        g_vmm_stats.search_ms += so.search_ms;
    }
    if (const char * e = std::getenv("TOAST_HIP_TRACE")) {
        if (e[0] != '\0' && e[0] != '0') {
            std::fprintf(stderr, "[toast_hip] vmm slab      %zu chunks of %zu MB at %p (%s): %zu of %zu wanted in the other zone, %zu created "
                                 "(%.2f ms each), %zu probes (%zu by the device clock), %.1f ms%s%s\n",
                         n, chunk >> 20, (void *)base, how, other, so.want_other, created, created ? so.create_ms / (double)created : 0.0, probes,
                         so.probes_clock, ms_since(t_start), so.exhausted ? ", search budget exhausted" : "", so.capped ? " (hard cap in ms)" : "");
            if (!so.zone_map.empty()) {
                std::fprintf(stderr, "[toast_hip] vmm survey    zones of the read-mostly range, a block every ~4 GB: %s\n", so.zone_map.c_str());
            }
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
    // Every candidate is measured at an address of its own: a pass over a chunk mapped where ANOTHER chunk sat a moment
    // ago showed the level of the earlier chunk (profiles/r04_a section 5) -- addresses are never reused here.
    // surplus candidates: TOAST_HIP_ARENA_SEARCH_GB, but never more than half of what this process' share of the device has
    // free beyond the slab itself (the transient chunks of one process' search must not make the
    // allocations of torch or of the processes it shares the device with fail)
    size_t search_bytes = pol.search;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t mine = free_b / (size_t)(g_share > 0 ? g_share : 1);
            const size_t room = mine > n * chunk ? (mine - n * chunk) / 2 : 0;
            if (search_bytes > room) search_bytes = room;
        } else {
            (void)hipGetLastError();
        }
    }
    const size_t own_n = n + search_bytes / chunk + 1;
    const double slow_ms = test_slow_ms();
    SearchOutcome so;
    void * own_res = nullptr;
    if (hipMemAddressReserve(&own_res, own_n * chunk, chunk, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    char * own_va = static_cast<char *>(own_res);

    // What the chunks are measured against.  With read-mostly slabs in place: their FIRST and their LAST GB -- a plain slab
    // of ~50 GB straddles a boundary of the 96 GB zones more often than not, and the scatter targets (odd slots) belong
    // into a zone that holds NEITHER end of it (three zones: there is one).  Measured against the first GB alone, the maps
    // shared their zone with the slab's upper part (the packed cache: 5.5 instead of 5.1 ms for its sweeps in two processes
    // of three); against the last GB alone, with its lower part (build_noise_weighted 5.9 instead of 5.1 ms).  Without such
    // slabs: the slab's own first chunk.
    ZoneRefs ext = zone_references_take(chunk);
    const int n_ref = ext.last != nullptr ? (ext.first != nullptr && ext.first != ext.last ? 2 : 1) : 0;
    struct Cand {
        hipMemGenericAllocationHandle_t h;
        char * at;              // where it is mapped during the search
        double r[2];            // rate of its pass with reference 0 (the slab's last GB / chunk 0) and 1 (the first GB); < 0: not measured
        double r2;              // rate of its pass with the ANCHOR, the first chunk that is clear of both references (third-zone split)
    };
    std::vector<Cand> cand;
    std::vector<void *> spacers;
    size_t spacer_bytes = 0;
    bool failed = false;
    size_t probes = 0;
    auto pass = [&](void * x, void * y, size_t each) {
        void * two[2] = {x, y};
        bool by_clock = false;
        test_slow(slow_ms);
        const double ms = probe_stream_split_ms(two, 2, each, st, &by_clock);     // (best of three passes: the first one touches the chunk)
        ++probes;
        so.probes_clock += by_clock ? 1 : 0;
        return ms > 0.0 ? 4.0 * (double)each / ms : 0.0;
    };
    // threshold between "same zone as the reference" and "another zone": the middle of the largest gap between
    // neighbouring rates once that gap is wider than anything one level shows (the two levels are ~13 % apart, each a few per
    // cent wide; a quantile does not work: either class can be the small one); with one level only, that level against
    // the rate of a pass over the two halves of the reference itself
    auto threshold = [&](int k, double level) {
        std::vector<double> r;
        for (const Cand & c : cand) {
            if (c.r[k] >= 0.0) r.push_back(c.r[k]);
        }
        if (r.empty()) return 1.0e300;
        std::sort(r.begin(), r.end());
        return gap_threshold(r, level, pol.gap);      // (0: all "other", 1e300: all "same")
    };
    size_t want_odd = 0;                         // slots of the other zone ("odd": round 4's name for them)
    for (size_t k = 0; k < n; ++k) want_odd += vmm_slot_other(k) ? 1 : 0;
    const size_t want_even = n - want_odd;
    const size_t max_create = n + search_bytes / chunk;
    so.want_other = want_odd;
    bool full = false;
    char * ref[2] = {nullptr, nullptr};
    double level[2] = {0.0, 0.0}, thr[2] = {1.0e300, 1.0e300};
    std::vector<size_t> cls_odd, cls_even;       // candidates for the odd slots (other than every reference) and the rest
    std::vector<size_t> grp_a, grp_b;            // third-zone split of cls_odd: same zone as the anchor / the other one
    bool third_done = false;
    bool prefer_end_zone = false;               // even slots: chunks of reference 0's zone first (see the survey below)
    bool refs_by_weight = false;                // the references are blocks of the two heaviest zones of the range, not its ends
    std::vector<void *> survey_held;            // ... held until the search is over
    try {
        if (n_ref >= 1) {
            ref[0] = static_cast<char *>(ext.last);
            if (n_ref == 2) ref[1] = static_cast<char *>(ext.first);
            for (int k = 0; k < n_ref; ++k) level[k] = pass(ref[k], ref[k] + chunk / 2, chunk / 2);
        }
        // The two ends say nothing about the middle: the driver hands a 56 GB slab out in runs of several zones, and on a box
        // whose memory is fragmented the third-zone split below -- both chunk classes out of the ends' zone -- put the map
        // and the written timestreams into the zones of the slab's MIDDLE, where half of the arrays lay
        // (build_noise_weighted 5.52 instead of 5.27 ms, scan_map 6.32 instead of 6.15: profiles/r06_f section 5).  A block
        // every ~4 GB of the range is sorted into zones first; the split is for a range that is ONE zone throughout.
        int survey_zones_n = 0;
        std::string survey_map;
        // (its ~K + 3 passes come out of the search's budget: no survey -- and no split -- on a budget that is smaller than that)
        if (pol.third && n_ref == 2 && pol.search_probes >= 64) {
            const char * rb = nullptr;
            size_t rlo = 0, rhi = 0;
            // (the two references are live blocks at its ends: the free range lies between them)
            if (read_mostly_free_range(&rb, &rlo, &rhi) && rhi - rlo >= 3 * chunk) {
                const std::vector<size_t> at = survey_offsets(rlo, rhi, chunk);
                const ZoneSurvey zs = survey_zones(
                    (int)at.size(), [&](int k) { return read_mostly_take_at(rb, at[(size_t)k], chunk); },
                    [](void * b) { read_mostly_release(b); }, pass, chunk, pol.gap);
                survey_zones_n = zs.zones;
                survey_map = zs.map;
                // The front half of the range -- where the arrays come first -- in the zone of its START, its END in another
                // one: the even slots ("the rest") then belong into the END's zone, not into the one all the readers are in
                // (scan_map 6.32 instead of 6.15 ms when they fell into it)
                const size_t K = zs.label.size();
                bool front_one = K >= 4 && zs.label[K - 1] > 0;
                for (size_t k = 0; k < (K + 1) / 2 && front_one; ++k) front_one = zs.label[k] == 0;
                prefer_end_zone = front_one;
                // Two or three zones in the range: the references become one block of each of the two HEAVIEST zones -- the
                // arrays are placed in address order, block k of K counts K - k -- instead of the two ends: "clear of both"
                // is then the lightest zone of the range (or one it does not touch), and the even slots prefer the second
                // heaviest (reference 0) to the heaviest.  (XXXXXXZZZYYYX: the ends are both X, "clear of X" was Z or Y as
                // they came, the rest X -- the zone of nearly every array: scan_map 6.28-6.30 ms.)
                // (ranges of up to 64 GB: on a 112 GB range -- the cfg-4 shard, a block every 7 GB -- the two ends did better,
                //  64.2-64.5 against 62.2-62.6 G in alternating processes, and the search for the rarest zone took 2.5-4.3 s)
                if (zs.zones >= 2 && survey_refs_enabled() && rhi - rlo <= (size_t(64) << 30)) {
                    double w[3] = {0.0, 0.0, 0.0};
                    int first_of[3] = {-1, -1, -1};
                    for (size_t k = 0; k < K; ++k) {
                        const int z = zs.label[k];
                        if (z < 0) continue;
                        w[z] += (double)(K - k);
                        if (first_of[z] < 0) first_of[z] = (int)k;
                    }
                    int order[3] = {0, 1, 2};
                    std::sort(order, order + 3, [&](int a, int b) { return w[a] > w[b]; });
                    void * heavy = read_mostly_take_at(rb, at[(size_t)first_of[order[0]]], chunk);
                    void * second = read_mostly_take_at(rb, at[(size_t)first_of[order[1]]], chunk);
                    if (heavy != nullptr && second != nullptr) {
                        survey_held = {heavy, second};
                        ref[1] = static_cast<char *>(heavy);
                        ref[0] = static_cast<char *>(second);
                        for (int k = 0; k < 2; ++k) level[k] = pass(ref[k], ref[k] + chunk / 2, chunk / 2);
                        prefer_end_zone = true;          // (reference 0 = the lighter of the two: first choice of the even slots)
                        refs_by_weight = true;
                    } else {
                        read_mostly_release(heavy);
                        read_mostly_release(second);
                    }
                }
            }
        }
        so.zones = survey_zones_n;
        so.zone_map = survey_map;
        const auto t_search = std::chrono::steady_clock::now();
        size_t streak = 0, last_odd = 0, last_even = 0;
        // Third-zone split (default; TOAST_HIP_ARENA_THIRD_ZONE=0 turns it off; profiles/r06_e).  When the read-mostly slab lies inside ONE zone X
        // -- both of its ends slow against the same chunks -- the chunks that are clear of it belong to the two OTHER zones,
        // and a written timestream whose rows are spread over THOSE two shares a zone with none of the streams the sweeps
        // read (P = X puts half of its rows next to pixels, weights and the read timestream: scan_map 6.33 instead of 6.08 ms).
        // The clear chunks are told apart by one more pass each against the first of them (the anchor).
        size_t anchor = SIZE_MAX, last_a = 0, last_b = 0, last_pref = 0;
        double level2 = 0.0;
        while (cand.size() < max_create) {
            // the budget is counted in measuring passes; wall time is a hard cap only (and reported): on a box whose memory the
            // driver is still clearing every chunk costs 20-50 ms instead of 0.1, and a search boxed into wall time gave up
            // there with the written timestreams in one zone
            if (cand.size() >= n && (long)probes >= pol.search_probes) break;
            if (cand.size() >= n && ms_since(t_search) > pol.search_ms) {
                so.capped = true;
                break;
            }
            // The driver hands out one zone after the other in runs of 4-12 GB, and a class that is still short may lie
            // 30 GB further on: that far in chunks of 1 GB is most of a second on a box that is still clearing its memory.
            // A plain hipMalloc is not (~1.5 ms per GB): when the last three chunks brought nothing that is still needed,
            // 8 GB of spacer step over the run (released below with the surplus chunks).
            if (pol.spacer > 0 && cand.size() >= n && streak >= 3 && spacer_bytes + pol.spacer <= search_bytes) {
                size_t free_b = 0, total_b = 0;
                void * sp = nullptr;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > pol.spacer + (size_t(16) << 30) &&
                    hipMalloc(&sp, pol.spacer) == hipSuccess) {
                    spacers.push_back(sp);
                    spacer_bytes += pol.spacer;
                    streak = 0;
                    continue;
                }
                (void)hipGetLastError();
                streak = 0;      // (no room: go on chunk by chunk)
            }
            hipMemGenericAllocationHandle_t h;
            const auto t_create = std::chrono::steady_clock::now();
            test_slow(slow_ms);
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
                (void)hipGetLastError();
                break;   // the device is full: make do with what has been created
            }
            so.create_ms += ms_since(t_create);
            char * where = own_va + cand.size() * chunk;
            if (!map_chunk(where, chunk, h, dev)) {
                (void)hipMemRelease(h);
                failed = true;
                break;
            }
            cand.push_back(Cand{h, where, {-1.0, -1.0}, -1.0});
            Cand & c = cand.back();
            if (n_ref == 0 && cand.size() == 1) {
                // no read-mostly slab yet: this chunk is the reference (and of the reference's class by definition)
                ref[0] = where;
                level[0] = pass(where, where + chunk / 2, chunk / 2);
                continue;       // (r[0] stays "not measured": never clear of itself, not part of the threshold)
            }
            c.r[0] = pass(ref[0], where, chunk);
            thr[0] = threshold(0, level[0]);
            if (n_ref == 2) {
                // the second reference only for chunks that are clear of the first (the others are "same" already)
                for (Cand & d : cand) {
                    if (d.r[1] < 0.0 && d.r[0] > thr[0]) d.r[1] = pass(ref[1], d.at, chunk);
                }
                thr[1] = threshold(1, level[1]);
            }
            cls_odd.clear();
            cls_even.clear();
            for (size_t i = 0; i < cand.size(); ++i) {
                const bool clear0 = cand[i].r[0] >= 0.0 && cand[i].r[0] > thr[0];
                const bool clear1 = n_ref < 2 || (cand[i].r[1] >= 0.0 && cand[i].r[1] > thr[1]);
                ((clear0 && clear1) ? cls_odd : cls_even).push_back(i);
            }
            bool third_open = false;
            if (pol.third && n_ref == 2 && (long)probes < pol.search_probes / 2 && cand.size() < n + 96) {
                // a chunk that is clear of the slab's end but not of its start: the slab straddles two zones, the chunks
                // that are clear of both belong to ONE zone -- nothing to split
                bool straddles = survey_zones_n != 1;        // (the range between the ends changes zone, or could not be surveyed)
                for (const Cand & d : cand) straddles |= (d.r[0] > thr[0] && d.r[1] >= 0.0 && !(d.r[1] > thr[1]));
                if (!straddles) {
                    third_open = true;
                    if (anchor == SIZE_MAX && !cls_odd.empty()) {
                        anchor = cls_odd[0];
                        level2 = pass(cand[anchor].at, cand[anchor].at + chunk / 2, chunk / 2);
                    }
                    if (anchor != SIZE_MAX && std::find(cls_odd.begin(), cls_odd.end(), anchor) == cls_odd.end()) {
                        // the thresholds moved and the anchor is no longer clear of the slab: start over with another one
                        anchor = SIZE_MAX;
                        for (Cand & d : cand) d.r2 = -1.0;
                        grp_a.clear();
                        grp_b.clear();
                        if (!cls_odd.empty()) {
                            anchor = cls_odd[0];
                            level2 = pass(cand[anchor].at, cand[anchor].at + chunk / 2, chunk / 2);
                        }
                    }
                    if (anchor != SIZE_MAX) {
                        for (size_t i : cls_odd) {
                            if (i != anchor && cand[i].r2 < 0.0) cand[i].r2 = pass(cand[anchor].at, cand[i].at, chunk);
                        }
                        std::vector<double> r2;
                        for (size_t i : cls_odd) {
                            if (i != anchor && cand[i].r2 >= 0.0) r2.push_back(cand[i].r2);
                        }
                        const double thr2 = gap_threshold(r2, level2, pol.gap);
                        grp_a.assign(1, anchor);
                        grp_b.clear();
                        for (size_t i : cls_odd) {
                            if (i == anchor || cand[i].r2 < 0.0) continue;
                            (cand[i].r2 > thr2 ? grp_b : grp_a).push_back(i);
                        }
                        if ((grp_a.size() >= want_even && grp_b.size() >= want_odd) ||
                            (grp_b.size() >= want_even && grp_a.size() >= want_odd)) {
                            third_done = true;
                            full = true;
                            break;
                        }
                    }
                }
            }
            // (reference 0 is the END of the range: "not clear of it" = in its zone)
            size_t pref_even = 0;
            if (prefer_end_zone) {
                for (size_t i : cls_even) pref_even += (cand[i].r[0] >= 0.0 && !(cand[i].r[0] > thr[0])) ? 1 : 0;
            }
            const bool want_pref = prefer_end_zone && (long)probes < pol.search_probes / 2 && cand.size() < n + 96;
            if (!third_open && cls_odd.size() >= want_odd && cls_even.size() >= want_even &&
                (!want_pref || pref_even >= want_even)) {
                full = true;
                break;
            }
            // did this chunk add to a class that is still short?
            bool useful = (cls_odd.size() > last_odd && last_odd < want_odd) ||
                          (cls_even.size() > last_even && last_even < want_even) ||
                          (want_pref && pref_even > last_pref && last_pref < want_even);
            last_pref = pref_even;
            if (third_open) {
                useful = (grp_a.size() > last_a) || (grp_b.size() > last_b) || cls_odd.size() > last_odd;
                last_a = grp_a.size();
                last_b = grp_b.size();
            }
            streak = useful ? 0 : streak + 1;
            last_odd = cls_odd.size();
            last_even = cls_even.size();
        }
        if (!full) full = cls_odd.size() >= want_odd && cls_even.size() >= want_even;   // (the third-zone attempt ran out: the two-class slab)
        so.search_ms = ms_since(t_search);
    } catch (const Error &) {
        failed = true;       // a failed launch or event: nothing may leak
    }
    so.exhausted = !full;
    for (void * sp : spacers) (void)hipFree(sp);
    zone_references_release(ext);
    for (void * b : survey_held) read_mostly_release(b);
    for (const Cand & c : cand) (void)hipMemUnmap(c.at, chunk);
    (void)hipMemAddressFree(own_res, own_n * chunk);
    const size_t created = cand.size();
    if (failed || created < n) {
        for (const Cand & c : cand) (void)hipMemRelease(c.h);
        return nullptr;      // (the caller falls back to a plain hipMalloc)
    }
    // Odd slots: the class that is clear of every reference; even slots: the rest.  What one class cannot fill, the other
    // does -- the LAST created first: the driver tends to hand out one zone after the other, so a late chunk is the most
    // likely to differ from the early ones.
    std::vector<size_t> odd, even;
    size_t take_o = std::min(cls_odd.size(), want_odd), take_e = std::min(cls_even.size(), want_even);
    if (third_done) {
        // both classes from zones that hold neither end of the read-mostly slab (the larger group serves the even slots)
        const bool a_even = grp_a.size() >= want_even && grp_b.size() >= want_odd;
        const std::vector<size_t> & ge = a_even ? grp_a : grp_b;
        const std::vector<size_t> & go = a_even ? grp_b : grp_a;
        even.assign(ge.begin(), ge.begin() + (long)want_even);
        odd.assign(go.begin(), go.begin() + (long)want_odd);
        take_o = want_odd;
        take_e = want_even;
        cls_even = ge;       // (nothing is topped up below: both lists are full)
        cls_odd = go;
    } else {
        if (prefer_end_zone) {
            std::stable_partition(cls_even.begin(), cls_even.end(),
                                  [&](size_t i) { return cand[i].r[0] >= 0.0 && !(cand[i].r[0] > thr[0]); });
        }
        odd.assign(cls_odd.begin(), cls_odd.begin() + (long)take_o);
        even.assign(cls_even.begin(), cls_even.begin() + (long)take_e);
    }
    if (odd.size() < want_odd) {
        // not enough chunks clear of both ends (the search ran out of time): second best for the scatter targets is the
        // zone of the slab's END -- the arrays that feed the A^T scatter are created first and sit at its start
        std::vector<char> taken(cand.size(), 0);
        for (size_t i : odd) taken[i] = 1;
        for (size_t i : even) taken[i] = 1;
        for (int round = 0; round < 2 && odd.size() < want_odd; ++round) {
            for (size_t j = cls_even.size(); j > take_e && odd.size() < want_odd; --j) {
                const size_t i = cls_even[j - 1];
                const bool at_end = !(cand[i].r[0] >= 0.0 && cand[i].r[0] > thr[0]);
                if (taken[i] || (round == 0 && !at_end)) continue;
                odd.push_back(i);
                taken[i] = 1;
            }
        }
    }
    for (size_t j = cls_odd.size(); j > take_o && even.size() < want_even; --j) even.push_back(cls_odd[j - 1]);
    void * va = nullptr;
    if (odd.size() < want_odd || even.size() < want_even ||
        hipMemAddressReserve(&va, n * chunk, chunk, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        for (const Cand & c : cand) (void)hipMemRelease(c.h);
        return nullptr;
    }
    char * base = static_cast<char *>(va);
    std::vector<hipMemGenericAllocationHandle_t> slot(n);
    std::vector<char> used(cand.size(), 0);
    size_t next_odd = 0, next_even = 0;
    for (size_t k = 0; k < n && !failed; ++k) {
        const size_t pick = vmm_slot_other(k) ? odd[next_odd++] : even[next_even++];
        used[pick] = 1;
        slot[k] = cand[pick].h;
        if (!map_chunk(base + k * chunk, chunk, slot[k], dev)) {
            for (size_t j = 0; j < k; ++j) (void)hipMemUnmap(base + j * chunk, chunk);
            failed = true;
        }
    }
    if (failed) {
        (void)hipMemAddressFree(va, n * chunk);
        for (const Cand & c : cand) (void)hipMemRelease(c.h);
        return nullptr;
    }
    for (size_t j = 0; j < cand.size(); ++j) {
        if (!used[j]) (void)hipMemRelease(cand[j].h);
    }
    if (const char * e = std::getenv("TOAST_HIP_TRACE")) {
        if (e[0] != '\0' && e[0] != '0') {
            std::string line;
            for (const Cand & c : cand) {
                char buf[48];
                if (n_ref == 2) {
                    std::snprintf(buf, sizeof buf, " %.2f/%.2f", c.r[0] / 1.0e9, c.r[1] < 0.0 ? 0.0 : c.r[1] / 1.0e9);
                } else {
                    std::snprintf(buf, sizeof buf, " %.2f", c.r[0] / 1.0e9);
                }
                line += buf;
            }
            std::fprintf(stderr, "[toast_hip] vmm rates vs %d reference(s) (TB/s, creation order; thresholds %.2f %.2f; %zu spacer(s) of %zu GB):%s\n", n_ref,
                         thr[0] < 1.0e299 ? thr[0] / 1.0e9 : 0.0, thr[1] < 1.0e299 ? thr[1] / 1.0e9 : 0.0, spacers.size(), pol.spacer >> 30, line.c_str());
        }
    }
    so.third = third_done;
    register_slab(base, n, chunk, slot, take_o, created, probes, level[0], t_start,
                  third_done ? "both classes clear of the read-mostly slab: two other zones"
                             : (refs_by_weight ? "against the two heaviest zones of the read-mostly range"
                                : n_ref == 2 ? "against both ends of the read-mostly slab"
                                           : (n_ref == 1 ? "against the read-mostly slab" : "against its first chunk")),
                  n_ref, so);
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

bool vmm_slab_layout(const void * base, size_t * chunk, int * n_ref) {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    auto it = g_vmm.find(const_cast<void *>(base));
    if (it == g_vmm.end()) return false;
    if (chunk != nullptr) *chunk = it->second.chunk;
    if (n_ref != nullptr) *n_ref = it->second.n_ref;
    return true;
}

int vmm_slab_scatter_class(const void * base) {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    auto it = g_vmm.find(const_cast<void *>(base));
    return it == g_vmm.end() ? -1 : it->second.scatter_class;
}

void vmm_slab_set_scatter_class(const void * base, int cls) {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    auto it = g_vmm.find(const_cast<void *>(base));
    if (it != g_vmm.end()) it->second.scatter_class = cls;
}

void vmm_set_device_share(int per) {
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    g_share = per > 0 ? per : 1;
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

// the class threshold of the zone search on given rates (no device): toast_hip_arena_zone_threshold, tests/test_capi_load.py
double vmm_zone_threshold(const double * rates, int n, double level) {
    return gap_threshold(std::vector<double>(rates, rates + (n > 0 ? n : 0)), level, policy().gap);
}

}  // namespace toast_hip
