// r3dm_context_pool.hpp -- process-wide pool of library contexts per device, shared by the C++ faces that the reference calls
// WITHOUT a handle: ArrayMatcher_r3dm (r3dm_array_matcher.hpp) and Regard3DFeatures (regard3d_features.hpp).
//
// The reference's entry points are static functions / plain objects called from many host threads at once
// (SearchNeighbours from an OpenMP loop, /root/reference/src/R3DComputeMatches.cpp:465; detectAndExtract from CPUs+1
// worker threads, /root/reference/src/threads/R3DFeaturesThread.cpp:58-77).  A library context drives one HIP stream and
// owns its scratch, so each such call leases a context for its duration: concurrent calls run on different streams and
// overlap on the GPU, callers beyond the pool size wait for a free one.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "r3dm.h"

namespace r3d_amd {

namespace detail {

// Contexts of one device, shared by every adapter of the process.  Created on demand, never destroyed (a static destructor
// would run after the HIP runtime's own teardown).
class ContextPool {
public:
    static constexpr int kPoolSize = 4;
    // number of contexts the pool may create (1 .. 64); takes effect for contexts not yet created
    void setLimit(int n) { std::lock_guard<std::mutex> lock(mu_); limit_ = n < 1 ? 1 : (n > 64 ? 64 : n); cv_.notify_all(); }
    int limit() { std::lock_guard<std::mutex> lock(mu_); return limit_; }
    static ContextPool& of(int device)
    {
        static std::mutex mu;
        static std::map<int, ContextPool*> pools;
        std::lock_guard<std::mutex> lock(mu);
        ContextPool*& p = pools[device];
        if (!p) p = new ContextPool(device);
        return *p;
    }
    r3dm_ctx* acquire()
    {
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            if (!free_.empty()) { r3dm_ctx* c = free_.back(); free_.pop_back(); return c; }
            if (created_ < limit_) {
                r3dm_ctx* c = nullptr;
                if (r3dm_create(device_, &c) == R3DM_OK) { ++created_; all_.push_back(c); return c; }
                if (created_ == 0) return nullptr;             // no usable GPU: the adapter reports failure, it never falls back
            }
            cv_.wait(lock);
        }
    }
    void release(r3dm_ctx* c)
    {
        { std::lock_guard<std::mutex> lock(mu_); free_.push_back(c); }
        cv_.notify_one();
    }
    // copies + re-layouts made by all contexts of the pool (r3dm_stats.n_views_staged): Build = 1, every search = 1 (its queries)
    uint64_t viewsStaged()
    {
        std::lock_guard<std::mutex> lock(mu_);
        uint64_t n = 0;
        for (r3dm_ctx* c : all_) { r3dm_stats s; if (r3dm_get_stats(c, &s) == R3DM_OK) n += s.n_views_staged; }
        return n;
    }
    int created() { std::lock_guard<std::mutex> lock(mu_); return created_; }

private:
    explicit ContextPool(int device) : device_(device) {}
    int device_;
    int created_ = 0;
    int limit_ = kPoolSize;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<r3dm_ctx*> free_, all_;
};

struct ContextLease {
    explicit ContextLease(ContextPool& p) : pool(p), ctx(p.acquire()) {}
    ~ContextLease() { if (ctx) pool.release(ctx); }
    ContextLease(const ContextLease&) = delete;
    ContextLease& operator=(const ContextLease&) = delete;
    ContextPool& pool;
    r3dm_ctx* ctx;
};

}  // namespace detail

}  // namespace r3d_amd
