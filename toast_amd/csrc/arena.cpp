// arena.cpp -- see arena.hpp.
#include "arena.hpp"
#include "runtime.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <sstream>

namespace toast_hip {

namespace {

double ms_since(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}

void * hip_take(size_t bytes, hipStream_t) {
    void * p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void hip_give(void * p) { (void)hipFree(p); }

void hip_touch(void * p, size_t bytes, hipStream_t st) {
    // One write of every page: what the first kernel to use a fresh block would otherwise pay inside an operator
    // (12 .. 17 ms per GB on this driver against 0.2 ms per GB for a fill of touched memory).  The host WAITS for it
    // (ADVICE round 4): blocks of the slab are handed to callers that work on other streams -- torch's, the upload and
    // communication streams of the library, any hipStreamNonBlocking one -- and a fill still in flight on this stream could
    // zero a parameter block or a scratch range after its owner wrote it.  Taking a slab is set-up work either way.
    if (hipMemsetAsync(p, 0, bytes, st) == hipSuccess) (void)hipStreamSynchronize(st);
}

void * host_take(size_t bytes, hipStream_t) { return std::malloc(bytes); }
void host_give(void * p) { std::free(p); }
void host_touch(void *, size_t, hipStream_t) {}

}  // namespace

const ArenaBackend & hip_backend() {
    static const ArenaBackend b{hip_take, hip_give, hip_touch};
    return b;
}

namespace {
void * hip_take_interleaved(size_t bytes, hipStream_t st) {
    void * p = vmm_slab_take(bytes, st);
    return p != nullptr ? p : hip_take(bytes, st);
}
void hip_give_interleaved(void * p) {
    if (!vmm_slab_give(p)) (void)hipFree(p);
}
}  // namespace

const ArenaBackend & hip_interleaved_backend() {
    static const ArenaBackend b{hip_take_interleaved, hip_give_interleaved, hip_touch};
    return b;
}

const ArenaBackend & host_backend() {
    static const ArenaBackend b{host_take, host_give, host_touch};
    return b;
}

bool Arena::grow_by(size_t bytes, hipStream_t stream) {
    // (grow_mutex_ held by the caller, mutex_ NOT)
    bytes = round_slab(bytes);
    const auto t0 = std::chrono::steady_clock::now();
    void * p = be_.take(bytes, stream);
    const double ms = ms_since(t0);
    if (p == nullptr) return false;
    const auto t1 = std::chrono::steady_clock::now();
    be_.touch(p, bytes, stream);
    const double touch_ms = ms_since(t1);
    std::lock_guard<std::mutex> lock(mutex_);
    st_.malloc_ms += ms;
    if (ms > st_.max_malloc_ms) st_.max_malloc_ms = ms;
    ++st_.slab_mallocs;
    st_.touch_ms += touch_ms;
    Slab s;
    s.base = static_cast<char *>(p);
    s.bytes = bytes;
    s.free[0] = bytes;
    slabs_.push_back(std::move(s));
    ++st_.slabs;
    st_.slab_bytes += bytes;
    return true;
}

void * Arena::carve_best(size_t need) {
    // best fit over all slabs
    Slab * best_slab = nullptr;
    std::map<size_t, size_t>::iterator best;
    for (Slab & s : slabs_) {
        for (auto it = s.free.begin(); it != s.free.end(); ++it) {
            if (it->second < need) continue;
            if (best_slab == nullptr || it->second < best->second) {
                best_slab = &s;
                best = it;
                if (it->second == need) break;
            }
        }
        if (best_slab != nullptr && best->second == need) break;
    }
    if (best_slab == nullptr) return nullptr;
    const size_t off = best->first, len = best->second;
    best_slab->free.erase(best);
    if (len > need) best_slab->free[off + need] = len - need;
    best_slab->live[off] = need;
    st_.used_bytes += need;
    if (st_.used_bytes > st_.peak_used_bytes) st_.peak_used_bytes = st_.used_bytes;
    ++st_.allocs;
    return best_slab->base + off;
}

void * Arena::alloc(size_t nbytes, hipStream_t stream, bool grow) {
    const size_t need = round_up(nbytes ? nbytes : 1);
    // (a range that another thread's release or slab opened up between the rounds is taken; a range that another
    //  thread's alloc took between "adopted" and "carved" costs one more round)
    for (int round = 0; round < 4; ++round) {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            if (void * p = carve_best(need)) return p;
            if (!grow) return nullptr;
        }
        std::lock_guard<std::mutex> growing(grow_mutex_);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            if (void * p = carve_best(need)) return p;      // somebody else's slab arrived while this thread waited
        }
        // a new slab: the default size when the request is smaller; when the driver refuses that, what is asked for
        const size_t want = need > slab_default_ ? need : round_up(slab_default_);
        if (!grow_by(want, stream) && !(want > need && grow_by(need, stream))) break;
    }
    std::lock_guard<std::mutex> lock(mutex_);
    ++st_.failed;
    return nullptr;
}

void * Arena::alloc_at_end(size_t nbytes) {
    const size_t need = round_up(nbytes ? nbytes : 1);
    std::lock_guard<std::mutex> lock(mutex_);
    Slab * at = nullptr;
    std::map<size_t, size_t>::iterator big;
    for (Slab & s : slabs_) {
        for (auto it = s.free.begin(); it != s.free.end(); ++it) {
            if (it->second >= need && (at == nullptr || it->second > big->second)) {
                at = &s;
                big = it;
            }
        }
    }
    if (at == nullptr) return nullptr;
    const size_t off = big->first, len = big->second;
    if (len == need) at->free.erase(big);
    else big->second = len - need;
    at->live[off + len - need] = need;
    st_.used_bytes += need;
    if (st_.used_bytes > st_.peak_used_bytes) st_.peak_used_bytes = st_.used_bytes;
    ++st_.allocs;
    return at->base + off + len - need;
}

void * Arena::alloc_striped(size_t nbytes, size_t stripe, int parity) {
    const size_t need = round_up(nbytes ? nbytes : 1);
    if (stripe == 0 || need > stripe) return nullptr;
    std::lock_guard<std::mutex> lock(mutex_);
    for (Slab & s : slabs_) {
        for (auto it = s.free.begin(); it != s.free.end(); ++it) {
            const size_t off = it->first, len = it->second;
            // stripes of the wanted parity that overlap [off, off + len)
            for (size_t k = off / stripe; k * stripe < off + len; ++k) {
                if ((int)(k & 1) != (parity & 1)) continue;
                const size_t lo = std::max(off, k * stripe), hi = std::min(off + len, (k + 1) * stripe);
                if (hi < lo + need) continue;
                s.free.erase(it);
                if (lo > off) s.free[off] = lo - off;
                if (off + len > lo + need) s.free[lo + need] = off + len - (lo + need);
                s.live[lo] = need;
                st_.used_bytes += need;
                if (st_.used_bytes > st_.peak_used_bytes) st_.peak_used_bytes = st_.used_bytes;
                ++st_.allocs;
                return s.base + lo;
            }
        }
    }
    return nullptr;
}

void * Arena::alloc_placed(size_t nbytes,
                           const std::function<size_t(const char * base, size_t lo, size_t hi, size_t need)> & place) {
    const size_t need = round_up(nbytes ? nbytes : 1);
    std::lock_guard<std::mutex> lock(mutex_);
    for (Slab & s : slabs_) {
        for (auto it = s.free.begin(); it != s.free.end(); ++it) {
            const size_t off = it->first, len = it->second;
            if (len < need) continue;
            size_t lo = place(s.base, off, off + len, need);
            if (lo == SIZE_MAX) continue;
            lo = (lo + granule_ - 1) / granule_ * granule_;
            if (lo < off || lo + need > off + len) continue;
            s.free.erase(it);
            if (lo > off) s.free[off] = lo - off;
            if (off + len > lo + need) s.free[lo + need] = off + len - (lo + need);
            s.live[lo] = need;
            st_.used_bytes += need;
            if (st_.used_bytes > st_.peak_used_bytes) st_.peak_used_bytes = st_.used_bytes;
            ++st_.allocs;
            return s.base + lo;
        }
    }
    return nullptr;
}

bool Arena::release(void * p) {
    std::lock_guard<std::mutex> lock(mutex_);
    const char * c = static_cast<const char *>(p);
    for (Slab & s : slabs_) {
        if (c < s.base || c >= s.base + s.bytes) continue;
        const size_t off0 = (size_t)(c - s.base);
        auto it = s.live.find(off0);
        if (it == s.live.end()) return false;
        size_t off = off0, len = it->second;
        s.live.erase(it);
        st_.used_bytes -= len;
        ++st_.releases;
        // merge with the free range that ends where this one starts, and with the one that starts where it ends
        auto nx = s.free.lower_bound(off);
        if (nx != s.free.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {
                off = pv->first;
                len += pv->second;
                s.free.erase(pv);
            }
        }
        if (nx != s.free.end() && off + len == nx->first) {
            len += nx->second;
            s.free.erase(nx);
        }
        s.free[off] = len;
        return true;
    }
    return false;
}

bool Arena::owns(const void * p) const {
    std::lock_guard<std::mutex> lock(mutex_);
    const char * c = static_cast<const char *>(p);
    for (const Slab & s : slabs_) {
        if (c >= s.base && c < s.base + s.bytes) return s.live.count((size_t)(c - s.base)) != 0;
    }
    return false;
}

bool Arena::slab_offset(const void * p, size_t * off) const {
    std::lock_guard<std::mutex> lock(mutex_);
    const char * c = static_cast<const char *>(p);
    for (const Slab & s : slabs_) {
        if (c >= s.base && c < s.base + s.bytes) {
            *off = (size_t)(c - s.base);
            return true;
        }
    }
    return false;
}

bool Arena::reserve(size_t bytes, hipStream_t stream, bool contiguous) {
    auto enough = [&] {     // (mutex_ held) 0: no; else: nothing to do
        if (contiguous) {
            for (const Slab & s : slabs_) {
                for (const auto & kv : s.free) {
                    if (kv.second >= bytes) return true;
                }
            }
            return false;
        }
        return st_.slab_bytes >= bytes;
    };
    size_t want = 0;
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (enough()) return true;
    }
    std::lock_guard<std::mutex> growing(grow_mutex_);
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (enough()) return true;
        want = contiguous ? round_up(bytes) : round_up(bytes - st_.slab_bytes);
    }
    return grow_by(want, stream);
}

size_t Arena::trim() {
    std::lock_guard<std::mutex> lock(mutex_);
    size_t freed = 0;
    for (size_t i = 0; i < slabs_.size();) {
        if (slabs_[i].live.empty()) {
            be_.give(slabs_[i].base);
            freed += slabs_[i].bytes;
            st_.slab_bytes -= slabs_[i].bytes;
            --st_.slabs;
            ++st_.slab_frees;
            slabs_.erase(slabs_.begin() + (long)i);
        } else {
            ++i;
        }
    }
    return freed;
}

void Arena::destroy() {
    std::lock_guard<std::mutex> lock(mutex_);
    for (Slab & s : slabs_) {
        be_.give(s.base);
        ++st_.slab_frees;
    }
    slabs_.clear();
    st_.slabs = 0;
    st_.slab_bytes = 0;
    st_.used_bytes = 0;
}

size_t Arena::free_bytes() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return st_.slab_bytes - st_.used_bytes;
}

std::vector<const char *> Arena::slab_bases() const {
    std::lock_guard<std::mutex> lock(mutex_);
    std::vector<const char *> out;
    for (const Slab & s : slabs_) out.push_back(s.base);
    return out;
}

size_t Arena::capacity() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return st_.slab_bytes;
}

size_t Arena::largest_free() const {
    std::lock_guard<std::mutex> lock(mutex_);
    size_t m = 0;
    for (const Slab & s : slabs_) {
        for (const auto & kv : s.free) m = kv.second > m ? kv.second : m;
    }
    return m;
}

ArenaStats Arena::stats() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return st_;
}

std::string Arena::check() const {
    std::lock_guard<std::mutex> lock(mutex_);
    std::ostringstream o;
    size_t used = 0, total = 0;
    for (size_t i = 0; i < slabs_.size(); ++i) {
        const Slab & s = slabs_[i];
        total += s.bytes;
        // walk both maps in address order: the ranges must tile [0, bytes) and no two free ranges may touch
        auto f = s.free.begin();
        auto l = s.live.begin();
        size_t pos = 0;
        bool last_free = false;
        while (f != s.free.end() || l != s.live.end()) {
            const bool take_free = (l == s.live.end()) || (f != s.free.end() && f->first < l->first);
            const size_t off = take_free ? f->first : l->first;
            const size_t len = take_free ? f->second : l->second;
            if (off != pos) {
                o << "slab " << i << ": gap or overlap at offset " << pos << " (next range starts at " << off << ")";
                return o.str();
            }
            if (len == 0 || len % granule_ != 0 || off % granule_ != 0) {
                o << "slab " << i << ": range " << off << "+" << len << " is not a multiple of the granule";
                return o.str();
            }
            if (take_free && last_free) {
                o << "slab " << i << ": adjacent free ranges at offset " << off << " were not merged";
                return o.str();
            }
            last_free = take_free;
            if (!take_free) used += len;
            pos = off + len;
            if (take_free) ++f; else ++l;
        }
        if (pos != s.bytes) {
            o << "slab " << i << ": ranges end at " << pos << " of " << s.bytes;
            return o.str();
        }
    }
    if (used != st_.used_bytes || total != st_.slab_bytes || (int64_t)slabs_.size() != st_.slabs) {
        o << "counters: used " << st_.used_bytes << " vs " << used << ", slab bytes " << st_.slab_bytes << " vs " << total;
        return o.str();
    }
    return "";
}

}  // namespace toast_hip
