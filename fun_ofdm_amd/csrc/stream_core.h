// stream_core.h -- the host-side machinery of the stream engine (foa_stream_*), free of any GPU call so that it can run
// under ThreadSanitizer / AddressSanitizer on a CPU (tests/cpp/stream_core_test.cpp, tools/run_sanitizers.sh).
//
// Three kinds of threads around one stream:
//   * the CALLER pushes samples.  A push only assigns its samples a place in the staging buffer of the batch being filled
//     (splitting at batch boundaries) and queues narrowing tasks; with no helper threads it narrows in line.  A push that
//     hands over OWNERSHIP of its buffer (process_samples takes its vector by value, so the chain may keep it) returns
//     without touching a sample: the helpers narrow it later and release it.
//   * `narrow_threads` HELPERS narrow double -> float (or copy float) task by task into page-locked staging memory.
//   * one SUBMITTER owns the GPU handle while the stream is open: when every sample of the oldest unsubmitted batch has
//     been narrowed it submits the batch (backend: carry copy, H2D, pre-sync, decode, D2H queued), and it collects finished
//     batches in order into the queue the caller takes payloads from.
// Ordering rests on two counters per staging slot -- samples assigned (caller only) and samples narrowed (atomic, release /
// acquire) -- and on the slot rotation: the caller may run at most kSlots - 1 batches ahead of the submitter.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#endif

namespace foa {

struct StreamReady { std::vector<uint8_t> bytes; std::vector<uint32_t> len; };

// What the core needs from the GPU side (stream_engine.h) or from a test double.
//   float *staging(int slot)                         page-locked buffer of `batch` float2
//   int submit(int slot, int64_t n_new, bool final, uint64_t *handle)      queue the batch (may block on the GPU); 0 = ok
//   int collect(uint64_t handle, bool wait, StreamReady *out)             1 = done (payloads in *out), 0 = not yet, < 0 error
template <typename Backend>
class StreamCore {
public:
    static constexpr int kSlots = 4;
    typedef void (*release_fn)(void *);

    StreamCore(Backend *be, int64_t batch, int narrow_threads) : be_(be), B_(batch), ring_(kRing)
    {
        for (auto &x : need_) x.store(-1);
        for (size_t i = 0; i < kRing; i++) ring_[i].seq.store(i, std::memory_order_relaxed);
        for (int i = 0; i < narrow_threads; i++) helpers_.emplace_back([this] { helper_loop(); });
        submitter_ = std::thread([this] { submitter_loop(); });
        keep_threads_near_caller();
    }
    ~StreamCore()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); }
        cv_work_.notify_all(); cv_sub_.notify_all(); cv_ready_.notify_all(); cv_room_.notify_all();
        for (auto &t : helpers_) t.join();
        if (submitter_.joinable()) submitter_.join();
        // buffers nobody narrowed (the stream was torn down early) are still released
        Task t;
        while (try_pop(t)) drop_owner(t.owner);
    }
    StreamCore(const StreamCore &) = delete;
    StreamCore &operator=(const StreamCore &) = delete;

    // n samples (interleaved re, im) of T = float or double.  release != nullptr: the buffer is the engine's until
    // release(ctx) is called (from a helper or from this thread); otherwise the samples are consumed before the call returns.
    template <typename T>
    int push(const T *iq, size_t n, release_fn release, void *ctx)
    {
        if (finished_) { if (release) release(ctx); return -5; }
        if (int e = error()) { if (release) release(ctx); return e; }
        Owner *own = nullptr;
        if (release) { own = new Owner; own->refs.store(1, std::memory_order_relaxed); own->release = release; own->ctx = ctx; }
        const bool defer = own != nullptr && !helpers_.empty();
        int rc = 0;
        while (n && !rc) {
            if (fill_ == 0 && (rc = wait_for_slot())) break;
            const int slot = (int)(batch_ % kSlots);
            const size_t take = (size_t)std::min<int64_t>((int64_t)n, B_ - fill_);
            Task t;
            t.src = iq; t.is_double = sizeof(T) == sizeof(double); t.dst = be_->staging(slot) + 2 * fill_; t.n = take; t.slot = slot; t.owner = own; t.landed = nullptr;
            if (defer) {
                own->refs.fetch_add(1, std::memory_order_relaxed);
                if (!try_push(t)) run_task(t, true);           // the queue is full: the helpers are behind, narrow this one here
            } else if (!helpers_.empty() && take >= 65536) {
                split_and_wait(t);
            } else {
                run_task(t, false);
            }
            fill_ += (int64_t)take; pushed_ += (int64_t)take;
            iq += 2 * take; n -= take;
            if (fill_ == B_) close_batch(false);
        }
        drop_owner(own);                                       // the caller's own reference
        return rc;
    }

    int flush()
    {
        if (finished_) return 0;
        if (fill_ == 0) { if (int rc = wait_for_slot()) return rc; }
        finished_ = true;
        close_batch(true);                                     // also when empty: the frames after the last cut are still undecoded
        return error();
    }

    // 1: *out = the payloads of the oldest finished batch; 0: nothing finished (wait: and nothing outstanding); < 0: error
    int take(bool wait, StreamReady *out)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!ready_.empty()) { *out = std::move(ready_.front()); ready_.pop_front(); return 1; }
            if (error_) return error_;
            if (!wait || closed_ == collected_) return 0;      // every closed batch has been collected (open samples need a flush)
            cv_ready_.wait(lk);
        }
    }
    int64_t pushed() const { return pushed_; }
    int64_t batches_closed() const { return closed_load(); }
    int error() { std::lock_guard<std::mutex> lk(m_); return error_; }

private:
    // The helpers stream the caller's samples (16 B in, 8 B out per sample): on a two-socket host they are several times
    // faster when they run on the cores that share the caller's last-level cache than when the scheduler spreads them over
    // both sockets (EPYC 9575F, 220 M samples: 2.1 against 1.2-1.6 Gsample/s with four helpers).  So the engine's threads are
    // confined to the block of eight consecutive CPUs the creating thread runs on (one core complex where CPUs are numbered
    // core by core), within the process's own affinity mask.  FOA_STREAM_AFFINITY=0 leaves them to the scheduler.
    void keep_threads_near_caller()
    {
#if defined(__linux__)
        const char *e = getenv("FOA_STREAM_AFFINITY");
        if (e && e[0] == '0') return;
        const int cpu = sched_getcpu();
        cpu_set_t have, want;
        if (cpu < 0 || sched_getaffinity(0, sizeof have, &have) != 0) return;
        CPU_ZERO(&want);
        int n = 0;
        for (int c = cpu & ~7; c < (cpu & ~7) + 8; c++)
            if (c < CPU_SETSIZE && CPU_ISSET(c, &have)) { CPU_SET(c, &want); n++; }
        if (n < 2) return;
        for (auto &t : helpers_) (void)pthread_setaffinity_np(t.native_handle(), sizeof want, &want);
        (void)pthread_setaffinity_np(submitter_.native_handle(), sizeof want, &want);
#endif
    }
    struct Owner { std::atomic<int> refs; release_fn release; void *ctx; };
    struct Task { const void *src; bool is_double; float *dst; size_t n; int slot; Owner *owner; std::atomic<int64_t> *landed; };
    struct Cell { std::atomic<uint64_t> seq; Task task; };
    static constexpr size_t kRing = 1 << 14;

    int64_t closed_load() const { return closed_; }
    static void drop_owner(Owner *o)
    {
        if (o && o->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) { o->release(o->ctx); delete o; }
    }
#if defined(__x86_64__)
    __attribute__((target("avx2"))) static void narrow_avx2(float *dst, const double *src, size_t n2)
    {
        for (size_t i = 0; i < n2; i++) dst[i] = (float)src[i];        // (vcvtpd2ps on four doubles at a time)
    }
#endif
    static void narrow(const Task &t)
    {
        if (!t.is_double) { memcpy(t.dst, t.src, t.n * 8); return; }
        const double *s = (const double *)t.src;
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) { narrow_avx2(t.dst, s, 2 * t.n); return; }
#endif
        for (size_t i = 0; i < 2 * t.n; i++) t.dst[i] = (float)s[i];
    }
    void run_task(const Task &t, bool from_queue)
    {
        narrow(t);
        if (from_queue) drop_owner(t.owner);
        std::atomic<int64_t> *landed = t.landed;               // (t may be a copy of a queue entry: the counter outlives it)
        const int64_t now = done_[t.slot].fetch_add((int64_t)t.n, std::memory_order_acq_rel) + (int64_t)t.n;
        if (landed) landed->fetch_add((int64_t)t.n, std::memory_order_release);
        // the submitter sleeps (with a time-out) until the oldest closed batch is complete: nudge it when a batch's last sample lands
        const int64_t need = need_[t.slot].load(std::memory_order_acquire);
        if (need >= 0 && now >= need) cv_sub_.notify_all();
    }
    // a large push without ownership: the helpers share it, the caller waits until it is through
    void split_and_wait(const Task &t)
    {
        const size_t piece = 16384;
        std::atomic<int64_t> landed(0);
        for (size_t o = 0; o < t.n; o += piece) {
            Task p = t;
            p.src = (const char *)t.src + o * (t.is_double ? 16 : 8); p.dst = t.dst + 2 * o; p.n = std::min(piece, t.n - o); p.owner = nullptr; p.landed = &landed;
            if (o == 0 || !try_push(p)) run_task(p, false);
        }
        // help, then wait: every piece of this push must have landed before the caller's buffer is given back
        Task q;
        while (landed.load(std::memory_order_acquire) < (int64_t)t.n) {
            if (try_pop(q)) run_task(q, true);
            else std::this_thread::yield();
        }
    }
    // Task queue: a bounded ring with a sequence number per cell (D. Vyukov's MPMC queue, used with one producer -- the
    // caller -- and many consumers).  No lock and no system call on the producer's path: at 4096-sample pushes the caller
    // has about two microseconds per call.
    bool try_push(const Task &t)
    {
        Cell &c = ring_[tail_ & (kRing - 1)];
        if (c.seq.load(std::memory_order_acquire) != tail_) return false;             // full
        c.task = t;
        c.seq.store(tail_ + 1, std::memory_order_release);
        tail_++;
        return true;
    }
    bool try_pop(Task &t)
    {
        uint64_t pos = head_.load(std::memory_order_relaxed);
        for (;;) {
            Cell &c = ring_[pos & (kRing - 1)];
            const uint64_t seq = c.seq.load(std::memory_order_acquire);
            const int64_t dif = (int64_t)seq - (int64_t)(pos + 1);
            if (dif == 0) {
                if (head_.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
                    t = c.task;
                    c.seq.store(pos + kRing, std::memory_order_release);
                    return true;
                }
            } else if (dif < 0) {
                return false;                                                          // empty
            } else {
                pos = head_.load(std::memory_order_relaxed);
            }
        }
    }
    void helper_loop()
    {
        int idle = 0;
        Task t;
        for (;;) {
            if (try_pop(t)) { run_task(t, true); idle = 0; continue; }
            if (stop_.load(std::memory_order_acquire)) return;
            if (++idle < 20000) { cpu_relax(); continue; }                             // poll for roughly 100 us, then doze
            std::unique_lock<std::mutex> lk(m_);
            if (!stop_.load()) cv_work_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(500));
            idle = 19000;
        }
    }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // the staging slot of the batch about to be filled is free once the batch kSlots before it has been submitted
    int wait_for_slot()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_room_.wait(lk, [this] { return stop_.load() || error_ || batch_ - submitted_ < kSlots; });
        if (error_) return error_;
        done_[batch_ % kSlots].store(0, std::memory_order_relaxed);
        need_[batch_ % kSlots].store(-1, std::memory_order_release);
        return 0;
    }
    void close_batch(bool final)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            need_[batch_ % kSlots].store(fill_, std::memory_order_release);
            final_[batch_ % kSlots] = final;
            closed_ = batch_ + 1;
        }
        cv_sub_.notify_all();
        batch_++;
        fill_ = 0;
    }
    void submitter_loop()
    {
        std::deque<uint64_t> flight;
        for (;;) {
            int slot = -1;
            int64_t n_new = 0;
            bool final = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                for (;;) {
                    if (stop_.load()) return;
                    if (submitted_ < closed_) {
                        const int s = (int)(submitted_ % kSlots);
                        const int64_t need = need_[s].load(std::memory_order_acquire);
                        if (done_[s].load(std::memory_order_acquire) >= need) { slot = s; n_new = need; final = final_[s]; break; }
                    }
                    // (timed: a helper's nudge can fall between the test above and the wait)
                    cv_sub_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(200));   // (system clock: pthread_cond_timedwait, which ThreadSanitizer knows)
                    if (!flight.empty()) break;                      // nothing to submit yet: see whether a batch in flight has finished
                }
            }
            int rc = 0;
            if (slot >= 0) {
                // at most kSlots - 1 batches in flight: the one about to go out takes the last device buffer
                while (!rc && (int)flight.size() >= kSlots - 1) rc = collect_one(flight, true) < 0 ? -1 : 0;
                uint64_t h = 0;
                if (!rc) rc = be_->submit(slot, n_new, final, &h);
                if (!rc) flight.push_back(h);
                { std::lock_guard<std::mutex> lk(m_); if (rc && !error_) error_ = rc < 0 ? rc : -3; submitted_++; }
                cv_room_.notify_all();
                cv_ready_.notify_all();
            }
            // hand finished batches over as they complete (polling: a batch that becomes ready to submit must not wait for the GPU)
            while (!flight.empty() && collect_one(flight, false) > 0) {}
        }
    }
    int collect_one(std::deque<uint64_t> &flight, bool wait)
    {
        StreamReady r;
        const int rc = be_->collect(flight.front(), wait, &r);
        if (rc == 0) return 0;
        std::lock_guard<std::mutex> lk(m_);
        if (rc < 0) { if (!error_) error_ = rc; flight.pop_front(); collected_++; cv_ready_.notify_all(); return rc; }
        flight.pop_front();
        ready_.push_back(std::move(r));
        collected_++;
        cv_ready_.notify_all();
        return 1;
    }

    Backend *be_;
    const int64_t B_;
    // caller-side state
    int64_t batch_ = 0, fill_ = 0, pushed_ = 0;
    bool finished_ = false;
    // shared
    std::mutex m_;
    std::condition_variable cv_work_, cv_sub_, cv_ready_, cv_room_;
    std::vector<Cell> ring_;
    uint64_t tail_ = 0;                              // producer (caller) only
    std::atomic<uint64_t> head_{ 0 };
    std::deque<StreamReady> ready_;
    std::atomic<int64_t> done_[kSlots] = {};
    std::atomic<int64_t> need_[kSlots];
    bool final_[kSlots] = {};
    int64_t closed_ = 0, submitted_ = 0, collected_ = 0;
    int error_ = 0;
    std::atomic<bool> stop_{ false };
    std::vector<std::thread> helpers_;
    std::thread submitter_;
};

}  // namespace foa
