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

#include <algorithm>
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
#include <sys/prctl.h>
#include <stdio.h>
#include <stdlib.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace foa {

struct StreamReady { std::vector<uint8_t> bytes; std::vector<uint32_t> len; };

// What the core needs from the GPU side (stream_engine.h) or from a test double.
//   float *staging(int slot)                         page-locked buffer of `batch` float2
//   int stage(int slot, int64_t n_new, bool final)    queue the upload of the batch's samples and whatever else can run before the batch
//                                                     is submitted (never blocks); 0 = ok
//   bool uploaded(int slot)                           all of that is through
//   int submit(int slot, int64_t n_new, bool final, uint64_t *handle)      queue the batch's kernels (may block on the GPU); 0 = ok;
//                                                     the staging slot is free again when it returns
//   int collect(uint64_t handle, bool wait, StreamReady *out)             1 = done (payloads in *out), 0 = not yet, < 0 error
// A batch is STAGED the moment its last sample is narrowed and SUBMITTED once its upload is through, so the upload of batch
// k+1 runs under the kernels of batch k instead of in front of its own (the submitter used to wait out 0.6 ms of PCIe per 4 Mi
// samples inside every submit: a third of its time).
template <typename Backend>
class StreamCore {
public:
    static constexpr int kSlots = 12;                // at most; a core rotates through slots_ of them (constructor)
    typedef void (*release_fn)(void *);

    StreamCore(Backend *be, int64_t batch, int narrow_threads, int slots = 6, int spin_us = 0)
        : be_(be), B_(batch), spin_ns_((int64_t)spin_us * 1000), slots_(slots < 2 ? 2 : slots > kSlots ? kSlots : slots), ring_(kRing)
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
        publish();
        Task t;
        while (try_pop(t)) drop_owner(t.owner);
        release_parked();
        if (stats_on_)
            fprintf(stderr, "foa_stream: caller ms in push %.1f (%lld pushes, of which waiting for a staging slot %.1f; %lld tasks narrowed by the caller); "
                            "%d helpers: %lld tasks, %.1f ms busy in total\n", st_push_ns_ * 1e-6, (long long)st_pushes_, st_wait_slot_ns_ * 1e-6,
                    (long long)st_inline_, (int)helpers_.size(), (long long)st_helper_tasks_.load(), st_helper_ns_.load() * 1e-6);
        if (stats_on_ && !st_lat_[4].empty()) {
            // medians: a harness's warm-up (batches pushed as fast as they are taken) and the first launches are not the stream's steady state
            auto med = [](std::vector<int64_t> &v) { if (v.empty()) return 0.0; std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return v[v.size() / 2] * 1e-3; };
            fprintf(stderr, "foa_stream: a batch's way through the submitter, median us over %zu batches: closed -> staging starts %.1f (its last sample narrowed %.1f), staging calls %.1f, "
                            "upload + pre-sync until seen %.1f, decode calls %.1f, decode until collected %.1f\n", st_lat_[4].size(),
                    med(st_lat_[0]), med(st_lat_[5]), med(st_lat_[1]), med(st_lat_[2]), med(st_lat_[3]), med(st_lat_[4]));
        }
    }
    StreamCore(const StreamCore &) = delete;
    StreamCore &operator=(const StreamCore &) = delete;

    // n samples (interleaved re, im) of T = float or double.  release != nullptr: the buffer is the engine's until
    // release(ctx) is called (from a helper or from this thread); otherwise the samples are consumed before the call returns.
    template <typename T>
    int push(const T *iq, size_t n, release_fn release, void *ctx)
    {
        const int64_t st0 = stats_on_ ? now_ns() : 0;
        caller_seq_.fetch_add(1, std::memory_order_relaxed);
        release_parked();
        const int rc = push_impl(iq, n, release, ctx);
        if (stats_on_) { st_push_ns_ += now_ns() - st0; st_pushes_++; }
        return rc;
    }
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    template <typename T>
    int push_impl(const T *iq, size_t n, release_fn release, void *ctx)
    {
        if (finished_) { if (release) release(ctx); return -5; }
        if (int e = error_flag_.load(std::memory_order_acquire)) { if (release) release(ctx); return e; }
        Owner *own = nullptr;
        if (release) {
            own = new Owner; own->refs.store(1, std::memory_order_relaxed); own->release = release; own->ctx = ctx;
            own->big = n * (sizeof(T) * 2) >= kBigOwner;
        }
        const bool defer = own != nullptr && !helpers_.empty();
        int rc = 0;
        while (n && !rc) {
            if (fill_ == 0 && (rc = wait_for_slot())) break;
            const int slot = (int)(batch_ % slots_);
            const size_t take = (size_t)std::min<int64_t>((int64_t)n, B_ - fill_);
            Task t;
            t.src = iq; t.is_double = sizeof(T) == sizeof(double); t.dst = be_->staging(slot) + 2 * fill_; t.n = take; t.slot = slot; t.owner = own; t.landed = nullptr;
            if (defer) {
                // in slices of kPiece samples, each with its own reference to the buffer: a call of a million samples is then narrowed by
                // every helper at once instead of by one (round 3: calls of >= 65536 samples ran at the rate of ONE helper), and the
                // caller still returns without touching a sample
                for (size_t o = 0; o < take; o += kPiece) {
                    Task p = t;
                    p.src = (const char *)t.src + o * (2 * sizeof(T)); p.dst = t.dst + 2 * o; p.n = std::min(kPiece, take - o);
                    own->refs.fetch_add(1, std::memory_order_relaxed);
                    if (!try_push(p)) { st_inline_++; run_task(p, true); }       // the queue is full: the helpers are behind, narrow this one here
                }
            } else if (!helpers_.empty() && take >= 65536) {
                split_and_wait(t);
            } else {
                run_task(t, false);
            }
            fill_ += (int64_t)take; pushed_ += (int64_t)take;
            iq += 2 * take; n -= take;
            if (fill_ == B_) close_batch(false);
        }
        if (rc) publish();                                     // (an error path must not sit on queued buffers)
        drop_owner(own);                                       // the caller's own reference
        return rc;
    }

    int flush()
    {
        caller_seq_.fetch_add(1, std::memory_order_relaxed);
        release_parked();
        if (finished_) return 0;
        publish();
        if (fill_ == 0) { if (int rc = wait_for_slot()) return rc; }
        finished_ = true;
        close_batch(true);                                     // also when empty: the frames after the last cut are still undecoded
        return error();
    }

    // 1: *out = the payloads of the oldest finished batch; 0: nothing finished (wait: and nothing outstanding); < 0: error
    int take(bool wait, StreamReady *out)
    {
        // (A poll -- take(false) -- does not count as "the caller is calling": a producer that waits for its pool of buffers to come back
        // by polling would otherwise keep the submitter from ever publishing its last < kPublish cells: it publishes leftovers only when
        // no call that could have done so came in between two of its wake-ups.  The poll itself publishes nothing -- the receive chain
        // polls after every push and must not undo the publishing stride.  Seen as "a polling producer got 0 of 5 buffers back" in
        // tests/cpp/stream_core_test.cpp when the box was busy; publish() and release_parked() are safe from both threads at once.)
        if (wait) caller_seq_.fetch_add(1, std::memory_order_relaxed);
        release_parked();
        if (wait) publish();                                   // (a caller about to sleep must not sit on unpublished tasks)
        // (the common call -- "anything finished?" after every push -- answers from one atomic, without the lock)
        if (!wait && ready_n_.load(std::memory_order_acquire) == 0) return error_flag_.load(std::memory_order_acquire);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!ready_.empty()) { *out = std::move(ready_.front()); ready_.pop_front(); ready_n_.fetch_sub(1, std::memory_order_acq_rel); return 1; }
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
    // (Measured in round 3 and dropped: one core complex PER helper on the caller's NUMA node, from sysfs -- a task then takes a
    // helper 7.2 us instead of 4.0, the buffers it releases belong to an allocator arena another complex owns, and the rate
    // falls from 2.2 to 1.7 Gsample/s; four helpers on one complex already move 100 GB/s.)
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
        const int b0 = cpu & ~7;
        for (int c = b0; c < b0 + 8; c++)
            if (c < CPU_SETSIZE && CPU_ISSET(c, &have)) { CPU_SET(c, &want); n++; }
        if (n < 2) return;
        // more than four helpers: every second one on the neighbouring block of eight (the same socket: blocks of 64 CPUs), so that two
        // core complexes' links to memory carry the samples; FOA_STREAM_AFFINITY=1 keeps everything on the caller's block
        cpu_set_t want2;
        CPU_ZERO(&want2);
        int n2 = 0;
        const int b1 = ((b0 + 8) / 64 == b0 / 64) ? b0 + 8 : b0 - 8;
        const bool two = helpers_.size() > 4 && !(e && e[0] == '1') && b1 >= 0;
        if (two)
            for (int c = b1; c < b1 + 8; c++)
                if (c < CPU_SETSIZE && CPU_ISSET(c, &have)) { CPU_SET(c, &want2); n2++; }
        for (size_t i = 0; i < helpers_.size(); i++) {
            const cpu_set_t &w = (two && n2 >= 2 && (i & 1)) ? want2 : want;
            (void)pthread_setaffinity_np(helpers_[i].native_handle(), sizeof w, &w);
        }
        (void)pthread_setaffinity_np(submitter_.native_handle(), sizeof want, &want);
#endif
    }
    struct Owner { std::atomic<int> refs; release_fn release; void *ctx; Owner *next; bool big; };
    // A buffer of kBigOwner bytes or more is the allocator's own mapping (glibc: mmap from 128 KB), no arena is involved in freeing it, and
    // giving it back costs a munmap -- a quarter of a microsecond per page: 0.1-0.3 s for the 3.5 GB of a 220 M-sample capture handed over
    // in calls of 65536 samples or more.  Parked for the caller (below) that was the caller's time, and the reason why such calls ran at
    // 0.64-0.70 Gsample/s in round 3 against 4 with calls of 4096; the helper that drops the last reference releases such a buffer itself.
    static constexpr size_t kBigOwner = 128 * 1024;
    static constexpr size_t kPiece = 16384;          // samples per narrowing task of a large push
    struct Task { const void *src; bool is_double; float *dst; size_t n; int slot; Owner *owner; std::atomic<int64_t> *landed; };
    struct Cell { std::atomic<uint64_t> seq; Task task; };
    static constexpr size_t kRing = 1 << 14;

    int64_t closed_load() const { return closed_; }
    // A buffer is given back by the thread that handed it over wherever possible: a helper that drops the last reference only parks the
    // owner on a list the caller empties at its next push / flush / take (the buffers are the caller's allocator's: released from another
    // core complex every free takes that arena's lock across the fabric -- with helpers on two complexes that was most of a task's time).
    static void release_now(Owner *o) { o->release(o->ctx); delete o; }
    void drop_owner(Owner *o, bool on_caller = true)
    {
        if (!o || o->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        if (on_caller || o->big) { release_now(o); return; }
        Owner *h = parked_.load(std::memory_order_relaxed);
        do { o->next = h; } while (!parked_.compare_exchange_weak(h, o, std::memory_order_release, std::memory_order_relaxed));
    }
    void release_parked()
    {
        if (parked_.load(std::memory_order_relaxed) == nullptr) return;
        Owner *o = parked_.exchange(nullptr, std::memory_order_acquire);
        while (o) { Owner *n = o->next; release_now(o); o = n; }
    }
#if defined(__x86_64__)
    __attribute__((target("avx2"))) static void narrow_avx2(float *dst, const double *src, size_t n2)
    {
        for (size_t i = 0; i < n2; i++) dst[i] = (float)src[i];        // (vcvtpd2ps on four doubles at a time)
    }
    // Sixteen doubles -> one 64-byte line of floats (vcvtpd2ps rounds to nearest even like the scalar conversion), stored
    // NON-TEMPORALLY: the staging buffer is written once and then read by the GPU's DMA engine, so a cached store would first
    // fetch every line it is about to overwrite (8 of the 32 bytes per sample this loop moves) and evict the caller's samples.
    __attribute__((target("avx512f,avx512dq"))) static void narrow_avx512(float *dst, const double *src, size_t n2)
    {
        size_t i = 0;
        while (i < n2 && ((uintptr_t)(dst + i) & 63)) { dst[i] = (float)src[i]; i++; }
        for (; i + 16 <= n2; i += 16) {
            const __m256 lo = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)), hi = _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8));
            _mm512_stream_ps(dst + i, _mm512_insertf32x8(_mm512_castps256_ps512(lo), hi, 1));
        }
        for (; i < n2; i++) dst[i] = (float)src[i];
        _mm_sfence();                                                  // the counter that publishes these samples is bumped next
    }
#endif
    static void narrow(const Task &t)
    {
        if (!t.is_double) { memcpy(t.dst, t.src, t.n * 8); return; }
        const double *s = (const double *)t.src;
#if defined(__x86_64__)
        static const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && !getenv("FOA_STREAM_NO_AVX512");
        if (avx512) { narrow_avx512(t.dst, s, 2 * t.n); return; }
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) { narrow_avx2(t.dst, s, 2 * t.n); return; }
#endif
        for (size_t i = 0; i < 2 * t.n; i++) t.dst[i] = (float)s[i];
    }
    void run_task(const Task &t, bool from_queue, bool on_caller = true)
    {
        narrow(t);
        if (from_queue) drop_owner(t.owner, on_caller);
        std::atomic<int64_t> *landed = t.landed;               // (t may be a copy of a queue entry: the counter outlives it)
        const int64_t now = done_[t.slot].fetch_add((int64_t)t.n, std::memory_order_acq_rel) + (int64_t)t.n;
        if (landed) landed->fetch_add((int64_t)t.n, std::memory_order_release);
        // the submitter sleeps (with a time-out) until the oldest closed batch is complete: nudge it when a batch's last sample lands
        const int64_t need = need_[t.slot].load(std::memory_order_acquire);
        if (need >= 0 && now >= need) { if (stats_on_) st_landed_[t.slot].store(now_ns(), std::memory_order_relaxed); cv_sub_.notify_all(); }
    }
    // a large push without ownership: the helpers share it, the caller waits until it is through
    void split_and_wait(const Task &t)
    {
        const size_t piece = kPiece;
        std::atomic<int64_t> landed(0);
        for (size_t o = 0; o < t.n; o += piece) {
            Task p = t;
            p.src = (const char *)t.src + o * (t.is_double ? 16 : 8); p.dst = t.dst + 2 * o; p.n = std::min(piece, t.n - o); p.owner = nullptr; p.landed = &landed;
            if (o == 0 || !try_push(p)) run_task(p, false);
        }
        // help, then wait: every piece of this push must have landed before the caller's buffer is given back
        publish();
        Task q;
        while (landed.load(std::memory_order_acquire) < (int64_t)t.n) {
            if (try_pop(q)) run_task(q, true);
            else std::this_thread::yield();
        }
    }
    // Task queue: a bounded ring with a sequence number per cell (D. Vyukov's MPMC queue, used with one producer -- the
    // caller -- and many consumers).  No lock and no system call on the producer's path: at 4096-sample pushes the caller
    // has about two microseconds per call.
    // The sequence number that makes a cell visible is stored for kPublish cells at a time: idle helpers all poll the cell at the head
    // of the queue, i.e. the very line the caller writes next, and with the helpers on other core complexes every such store first
    // has to pull the line back from them -- at 4096-sample pushes that was most of the caller's two microseconds.  Whatever is
    // written but not yet published goes out when a batch closes, on flush() and on every error path; samples of an OPEN batch may
    // therefore sit unnarrowed (and owned buffers unreleased) until the next few pushes, or, when the caller makes no further call at all,
    // until the submitter thread notices (within half a millisecond; tests/cpp/stream_core_test.cpp, pooled producer).
    static constexpr uint64_t kPublish = 16;
    bool try_push(const Task &t)
    {
        const uint64_t tail = tail_.load(std::memory_order_relaxed);                                   // (the caller is the only writer)
        Cell &c = ring_[tail & (kRing - 1)];
        if (c.seq.load(std::memory_order_acquire) != tail) { publish(); return false; }              // full
        c.task = t;
        tail_.store(tail + 1, std::memory_order_release);
        if (tail + 1 - pub_.load(std::memory_order_relaxed) >= kPublish) publish();
        return true;
    }
    // Makes every written cell visible.  The caller does it every kPublish cells and wherever it is about to wait; the SUBMITTER does it
    // too, when cells have sat unpublished over two of its wake-ups while the caller made no call (a producer with a bounded pool of
    // buffers waits for them to come back without calling anything: round 3's core then held on to its last < kPublish buffers for
    // good).  Hence the claim: a range [pub_, tail_) belongs to whoever moves pub_ over it.
    void publish()
    {
        const uint64_t t = tail_.load(std::memory_order_acquire);
        uint64_t p = pub_.load(std::memory_order_relaxed);
        while (p < t) {
            if (!pub_.compare_exchange_weak(p, t, std::memory_order_acq_rel, std::memory_order_relaxed)) continue;
            for (; p < t; p++) ring_[p & (kRing - 1)].seq.store(p + 1, std::memory_order_release);
        }
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
            if (try_pop(t)) {
                const int64_t h0 = stats_on_ ? now_ns() : 0;
                run_task(t, true, false);
                if (stats_on_) { st_helper_ns_.fetch_add(now_ns() - h0, std::memory_order_relaxed); st_helper_tasks_.fetch_add(1, std::memory_order_relaxed); }
                idle = 0;
                continue;
            }
            if (stop_.load(std::memory_order_acquire)) return;
            if (++idle < 2500) { for (int k = 0; k < 8; k++) cpu_relax(); continue; }  // poll (sparsely: every poll shares the line the caller writes next) for roughly 100 us, then doze
            std::unique_lock<std::mutex> lk(m_);
            if (!stop_.load()) cv_work_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(500));
            idle = 2300;
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
        publish();
        const int64_t w0 = stats_on_ ? now_ns() : 0;
        std::unique_lock<std::mutex> lk(m_);
        cv_room_.wait(lk, [this] { return stop_.load() || error_ || batch_ - submitted_ < slots_; });
        if (stats_on_) st_wait_slot_ns_ += now_ns() - w0;
        if (error_) return error_;
        done_[batch_ % slots_].store(0, std::memory_order_relaxed);
        need_[batch_ % slots_].store(-1, std::memory_order_release);
        return 0;
    }
    void close_batch(bool final)
    {
        publish();
        {
            std::lock_guard<std::mutex> lk(m_);
            need_[batch_ % slots_].store(fill_, std::memory_order_release);
            final_[batch_ % slots_] = final;
            if (stats_on_) st_closed_[batch_ % slots_] = now_ns();
            closed_ = batch_ + 1;
        }
        cv_sub_.notify_all();
        batch_++;
        fill_ = 0;
    }
    struct Staged { int slot; int64_t n_new; bool final; int64_t t_staged; };
    void submitter_loop()
    {
#ifdef __linux__
        // timed waits of this thread end when they are due: the default timer slack (50 us) is more than the waits themselves
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
#endif
        std::deque<uint64_t> flight;
        std::deque<Staged> staged;
        int64_t staged_n = 0;                                        // batches staged so far (this thread only)
        bool tidy = false, had_left = false;
        uint64_t seen_seq = 0;
        int64_t last_active = now_ns();                              // a batch was staged, submitted or collected (spin_ns_)
        for (;;) {
            int slot = -1;
            int64_t n_new = 0;
            bool final = false;
            {
                std::unique_lock<std::mutex> lk(m_);
                for (;;) {
                    if (stop_.load()) return;
                    if (staged_n < closed_) {
                        const int s = (int)(staged_n % slots_);
                        const int64_t need = need_[s].load(std::memory_order_acquire);
                        if (done_[s].load(std::memory_order_acquire) >= need) { slot = s; n_new = need; final = final_[s]; break; }
                    }
                    // (timed: a helper's nudge can fall between the test above and the wait; shorter while an upload or a batch is in flight)
                    const bool busy = !staged.empty() || !flight.empty();
                    // A stream of SMALL batches does not let this thread sleep while it is live: every batch passes it three times (staged,
                    // submitted once its alignment count is on the host, collected), and a timed sleep in front of each -- with whatever the
                    // host adds to a wake-up -- is the tail of the payload latency: 4 Ki-sample batches at 20 Msample/s, twelve buffers,
                    // p90 / p99 3.3-3.6 / 4.4-4.7 ms sleeping against 1.03-1.64 / 1.23-4.1 polling, three runs each
                    // (profiles/r06_latency_stages.txt).  It polls -- a turn of this loop is about a microsecond -- until nothing has
                    // happened for spin_ns_, then sleeps as above.
                    if (spin_ns_ > 0 && now_ns() - last_active < spin_ns_) {
                        lk.unlock();
                        for (int k = 0; k < 32; k++) cpu_relax();
                        lk.lock();
                    } else
                    cv_sub_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(busy ? 20 : 200));   // (system clock: pthread_cond_timedwait, which ThreadSanitizer knows)
                    // the caller's leftovers: cells it has written but not published, buffers parked for it to release.  Both are the
                    // caller's to deal with at its next call -- unless there is none: leftovers seen at two wake-ups in a row with no call
                    // in between (>= 40 us) are dealt with here
                    const uint64_t cs = caller_seq_.load(std::memory_order_relaxed);
                    const bool left = pub_.load(std::memory_order_relaxed) != tail_.load(std::memory_order_relaxed) || parked_.load(std::memory_order_relaxed) != nullptr;
                    tidy = left && had_left && cs == seen_seq;
                    had_left = left; seen_seq = cs;
                    if (busy || tidy) break;                         // nothing to stage yet: see whether an upload or a batch in flight has finished
                }
            }
            if (tidy) { publish(); release_parked(); tidy = false; }
            int rc = 0;
            if (slot >= 0) {
                last_active = now_ns();
                // a device buffer is written again kSlots batches later: at most kSlots - 1 batches staged or in flight, the one
                // about to be staged included
                while (!rc && (int)(staged.size() + flight.size()) >= slots_ - 1) {
                    if (flight.empty()) rc = submit_front(staged, flight);
                    else rc = collect_one(flight, true) < 0 ? -1 : 0;
                }
                const int64_t s0 = stats_on_ ? now_ns() : 0;
                if (!rc) rc = be_->stage(slot, n_new, final);
                staged_n++;
                const int64_t s1 = stats_on_ ? now_ns() : 0;
                if (stats_on_) { st_lat_[0].push_back(s0 - st_closed_[slot]); st_lat_[1].push_back(s1 - s0); st_lat_[5].push_back(std::max<int64_t>(0, st_landed_[slot].load(std::memory_order_relaxed) - st_closed_[slot])); st_landed_[slot].store(0, std::memory_order_relaxed); }
                if (!rc) staged.push_back(Staged{ slot, n_new, final, s1 });
                else fail_batch(rc);
            }
            // the oldest staged batch goes out once its samples are on the device
            const size_t in_hand = staged.size() + 2 * flight.size();
            while (!staged.empty() && be_->uploaded(staged.front().slot)) (void)submit_front(staged, flight);
            // hand finished batches over as they complete (polling: a batch that becomes ready to stage must not wait for the GPU)
            while (!flight.empty() && collect_one(flight, false) > 0) {}
            if (spin_ns_ > 0 && staged.size() + 2 * flight.size() != in_hand) last_active = now_ns();
        }
    }
    // a batch that could not be staged or submitted still counts as submitted (the caller waits for its staging slot) and as
    // collected (take() waits for every closed batch)
    void fail_batch(int rc)
    {
        { std::lock_guard<std::mutex> lk(m_); if (!error_) { error_ = rc < 0 ? rc : -3; error_flag_.store(error_, std::memory_order_release); } submitted_++; collected_++; }
        cv_room_.notify_all();
        cv_ready_.notify_all();
    }
    int submit_front(std::deque<Staged> &staged, std::deque<uint64_t> &flight)
    {
        const Staged s = staged.front();
        staged.pop_front();
        uint64_t h = 0;
        const int64_t u0 = stats_on_ ? now_ns() : 0;
        const int rc = be_->submit(s.slot, s.n_new, s.final, &h);
        if (rc) { fail_batch(rc); return rc; }
        flight.push_back(h);
        if (stats_on_) { const int64_t u1 = now_ns(); st_lat_[2].push_back(u0 - s.t_staged); st_lat_[3].push_back(u1 - u0); st_submitted_.push_back(u1); }
        { std::lock_guard<std::mutex> lk(m_); submitted_++; }
        cv_room_.notify_all();
        cv_ready_.notify_all();
        return 0;
    }
    int collect_one(std::deque<uint64_t> &flight, bool wait)
    {
        StreamReady r;
        const int rc = be_->collect(flight.front(), wait, &r);
        if (rc == 0) return 0;
        std::lock_guard<std::mutex> lk(m_);
        if (rc < 0) { if (!error_) { error_ = rc; error_flag_.store(rc, std::memory_order_release); } flight.pop_front(); collected_++; cv_ready_.notify_all(); if (!st_submitted_.empty()) st_submitted_.pop_front(); return rc; }
        flight.pop_front();
        if (stats_on_ && !st_submitted_.empty()) { st_lat_[4].push_back(now_ns() - st_submitted_.front()); st_submitted_.pop_front(); }
        ready_.push_back(std::move(r));
        ready_n_.fetch_add(1, std::memory_order_acq_rel);
        collected_++;
        cv_ready_.notify_all();
        return 1;
    }

    Backend *be_;
    const int64_t B_;
    const int64_t spin_ns_;                          // the submitter polls instead of sleeping until nothing has happened for this long (0: it sleeps)
    const int slots_;                                // staging slots (= the backend's device buffers) in rotation
    // caller-side state
    int64_t batch_ = 0, fill_ = 0, pushed_ = 0;
    bool finished_ = false;
    // shared
    std::mutex m_;
    std::condition_variable cv_work_, cv_sub_, cv_ready_, cv_room_;
    std::vector<Cell> ring_;
    std::atomic<uint64_t> tail_{ 0 }, pub_{ 0 };     // cells written (the caller alone writes it) / cells made visible (whoever claims the range)
    std::atomic<uint64_t> caller_seq_{ 0 };          // bumped by every push / flush / take: lets the submitter see that the caller has gone quiet
    std::atomic<uint64_t> head_{ 0 };
    std::deque<StreamReady> ready_;
    std::atomic<int64_t> done_[kSlots] = {};
    std::atomic<int64_t> need_[kSlots];
    bool final_[kSlots] = {};
    int64_t closed_ = 0, submitted_ = 0, collected_ = 0;
    int error_ = 0;
    std::atomic<int> error_flag_{ 0 };               // = error_, readable without the lock
public:
    // where the time goes (FOA_STREAM_STATS; ns / counts): caller inside push(), helpers narrowing, tasks the caller narrowed itself
    // because the queue was full, time the caller waited for a staging slot
    const bool stats_on_ = getenv("FOA_STREAM_STATS") != nullptr;
    int64_t st_push_ns_ = 0, st_wait_slot_ns_ = 0, st_inline_ = 0, st_pushes_ = 0;
    std::atomic<int64_t> st_helper_ns_{ 0 }, st_helper_tasks_{ 0 };
    int64_t st_closed_[kSlots] = {};
    std::vector<int64_t> st_lat_[6];                 // per batch, ns: closed -> staging, staging calls, upload until seen, decode calls, decode until collected, closed -> narrowed
    std::atomic<int64_t> st_landed_[kSlots] = {};     // a batch's way through the submitter (closed_ under the lock; the rest submitter only)
    std::deque<int64_t> st_submitted_;
private:
    std::atomic<int> ready_n_{ 0 };                  // = ready_.size(), readable without the lock
    std::atomic<Owner *> parked_{ nullptr };         // owners whose last reference a helper dropped: released by the caller
    std::atomic<bool> stop_{ false };
    std::vector<std::thread> helpers_;
    std::thread submitter_;
};

}  // namespace foa
