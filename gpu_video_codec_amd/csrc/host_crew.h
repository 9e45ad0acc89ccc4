/*
 * host_crew.h -- PRIVATE to the library: the host threads that move a large frame between the caller's ordinary (pageable)
 * planes and the context's page-locked staging ring, so that the reference-shaped call (hevc_deblocking_filter on a frame
 * the caller got from malloc / fread, as main.cu:112-133 does) is bound by the host link and not by one core's memcpy
 * (10 GB/s of staging against a 53 GB/s link).  What the reference does at this point: a row-by-row copy into
 * cudaMallocHost planes inside ReadYuvFrame's constructor (gpu.cu:1092-1133), single-threaded and outside its timers.
 *
 * One producer (the thread inside hevc_deblocking_filter) hands out row-range copies; the crew and the producer itself
 * (while it waits) execute them.  Between calls the crew sleeps on a condition variable; inside a call it polls, because a
 * futex wake-up (30-60 us) is a tenth of the whole call.  No HIP call is ever made from a crew thread.
 */
#pragma once
#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace dbkh {

/* completion of a group of jobs (one strip, one direction) */
struct CopyGroup {
    std::atomic<int> pending{0};
    std::atomic<int64_t> done_ns{0}; /* steady clock when the last job of the group ended */
    std::atomic<int64_t> first_ns{0}; /* ... when the first job of the group began */
};

struct CopyJob {
    uint8_t *dst;
    const uint8_t *src;
    size_t dpitch, spitch, row_bytes;
    unsigned rows;
    CopyGroup *group;
    bool to_device = false; /* dst is HBM seen through the PCIe BAR (write-combining): streaming stores always, and one read of the last
                               byte written at the end -- a read cannot pass the posted writes ahead of it, so when it returns they have landed */
};

inline int64_t crew_now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* rows of `row_bytes` from src to dst; a range that is contiguous on both sides goes as one block.  Blocks of 256 KiB and
 * more are written with non-temporal stores when stream_stores is set: the destination (the staging ring on the way in, the
 * caller's plane on the way out) is not read again by this core, so the write-allocate read of every destination line and the
 * eviction of the source from the cache are avoided (tools/ubench/host_stage.hip measures both forms on the box). */
inline void copy_rows(const CopyJob &j, bool stream_stores)
{
    const bool always = j.to_device;
    auto block = [&](uint8_t *d, const uint8_t *s, size_t n) {
        if (always || (stream_stores && n >= ((size_t)256 << 10))) { /* SSE2: part of every x86-64 */
            const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
            if (head && head <= n) { std::memcpy(d, s, head); d += head; s += head; n -= head; }
            size_t i = 0;
            for (; i + 64 <= n; i += 64) {
                const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 16));
                const __m128i c = _mm_loadu_si128((const __m128i *)(s + i + 32)), e = _mm_loadu_si128((const __m128i *)(s + i + 48));
                _mm_stream_si128((__m128i *)(d + i), a);
                _mm_stream_si128((__m128i *)(d + i + 16), b);
                _mm_stream_si128((__m128i *)(d + i + 32), c);
                _mm_stream_si128((__m128i *)(d + i + 48), e);
            }
            if (i < n) std::memcpy(d + i, s + i, n - i);
            return;
        }
        std::memcpy(d, s, n);
    };
    if (j.rows == 0 || j.row_bytes == 0) return;
    if (j.dpitch == j.row_bytes && j.spitch == j.row_bytes)
        block(j.dst, j.src, (size_t)j.rows * j.row_bytes);
    else if (always)
        for (unsigned r = 0; r < j.rows; r++) block(j.dst + (size_t)r * j.dpitch, j.src + (size_t)r * j.spitch, j.row_bytes);
    else
        for (unsigned r = 0; r < j.rows; r++) std::memcpy(j.dst + (size_t)r * j.dpitch, j.src + (size_t)r * j.spitch, j.row_bytes);
    _mm_sfence(); /* streaming stores leave the core's write-combining buffers before the job counts as done */
    if (j.to_device) {
        const volatile uint8_t *last = j.dst + (size_t)(j.rows - 1) * j.dpitch + j.row_bytes - 1;
        (void)*last;
    }
}

class StageCrew {
public:
    /* two lanes of jobs: 0 = copies in (towards the GPU: the ring, or HBM through the BAR, where two cores already saturate the
     * link), 1 = copies out (results back into the caller's planes).  The first min(2, ceil(workers / 2)) crew threads look at lane 0
     * first, the others at lane 1 first; an idle thread takes from either.  So copies-in run ahead continuously while copies-out of
     * finished strips are not queued behind them. */
    enum { LANE_IN = 0, LANE_OUT = 1 };

    /* cpus: where the crew may run (the CPUs next to the GPU's host bridge), or NULL for no restriction */
    StageCrew(unsigned workers, bool stream_stores, const cpu_set_t *cpus = nullptr) : stream_(stream_stores)
    {
        for (auto &ln : lane_) {
            ln.ring = std::vector<Slot>(kRing);
            for (uint64_t i = 0; i < kRing; i++) ln.ring[i].seq.store(i, std::memory_order_relaxed);
        }
        const unsigned in_first = workers <= 1 ? workers : ((workers + 1) / 2 < 2 ? (workers + 1) / 2 : 2);
        for (unsigned i = 0; i < workers; i++) {
            th_.emplace_back([this, i, in_first] { worker(i < in_first ? LANE_IN : LANE_OUT); });
            if (cpus) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof(cpu_set_t), cpus); /* best effort */
        }
    }
    ~StageCrew()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
            active_.store(0, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    StageCrew(const StageCrew &) = delete;
    StageCrew &operator=(const StageCrew &) = delete;

    unsigned workers() const { return (unsigned)th_.size(); }
    bool stream_stores() const { return stream_; }

    /* a call begins: the crew starts polling.  Every begin() is paired with an end() on every path out of the call. */
    void begin()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            active_.store(1, std::memory_order_release);
        }
        cv_.notify_all();
    }
    /* a call ends: every submitted job has been waited for (wait()), the crew goes back to sleep */
    void end() { active_.store(0, std::memory_order_release); }

    /* producer only.  The group's pending count has been set by the producer before its first job is submitted. */
    void submit(const CopyJob &j, int lane = LANE_IN)
    {
        Lane &ln = lane_[lane];
        const uint64_t t = ln.tail;
        Slot &sl = ln.ring[t % kRing];
        while (sl.seq.load(std::memory_order_acquire) != t) /* ring full: do some of the work ourselves */
            if (!run_one(lane)) _mm_pause();
        sl.job = j;
        sl.seq.store(t + 1, std::memory_order_release);
        ln.tail = t + 1;
    }
    /* producer only: executes one queued job itself, if there is one */
    bool help() { return run_one(LANE_IN) || run_one(LANE_OUT); }
    /* producer only: executes jobs itself until the group is complete */
    void wait(CopyGroup &g)
    {
        while (g.pending.load(std::memory_order_acquire) > 0)
            if (!help()) _mm_pause();
    }

private:
    static constexpr uint64_t kRing = 256;
    struct Slot {
        std::atomic<uint64_t> seq{0};
        CopyJob job{};
        Slot() = default;
        Slot(const Slot &o) : seq(o.seq.load(std::memory_order_relaxed)), job(o.job) {} /* construction of the rings only */
    };
    struct Lane {
        std::vector<Slot> ring;
        std::atomic<uint64_t> head{0};
        uint64_t tail = 0; /* the producer's alone */
    };

    /* bounded queue with a sequence number per slot (one producer, any number of consumers): slot t % kRing holds job t
     * when its seq is t + 1 and is free for job t + kRing once its seq is t + kRing */
    bool run_one(int lane)
    {
        Lane &ln = lane_[lane];
        uint64_t h = ln.head.load(std::memory_order_relaxed);
        for (;;) {
            Slot &sl = ln.ring[h % kRing];
            const int64_t dif = (int64_t)(sl.seq.load(std::memory_order_acquire) - (h + 1));
            if (dif < 0) return false; /* job h has not been published: the queue is empty */
            if (dif > 0) { h = ln.head.load(std::memory_order_relaxed); continue; } /* somebody else took job h */
            if (!ln.head.compare_exchange_weak(h, h + 1, std::memory_order_relaxed)) continue;
            const CopyJob j = sl.job;
            sl.seq.store(h + kRing, std::memory_order_release);
            int64_t zero = 0;
            j.group->first_ns.compare_exchange_strong(zero, crew_now_ns(), std::memory_order_relaxed);
            copy_rows(j, stream_);
            /* every job leaves its end time (the latest stays) before the count goes down: whoever sees zero sees the group's end */
            const int64_t t = crew_now_ns();
            for (int64_t seen = j.group->done_ns.load(std::memory_order_relaxed);
                 seen < t && !j.group->done_ns.compare_exchange_weak(seen, t, std::memory_order_relaxed);) {}
            j.group->pending.fetch_sub(1, std::memory_order_acq_rel);
            return true;
        }
    }
    void worker(int first_lane)
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return quit_ || active_.load(std::memory_order_acquire) != 0; });
                if (quit_) return;
            }
            while (active_.load(std::memory_order_acquire) != 0)
                if (!run_one(first_lane) && !run_one(first_lane ^ 1)) _mm_pause();
        }
    }

    const bool stream_;
    Lane lane_[2];
    std::atomic<int> active_{0};
    bool quit_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> th_;
};

} /* namespace dbkh */
