// stream_core_test.cpp -- the threading core of the stream engine (fun_ofdm_amd/csrc/stream_core.h) against a backend double:
// every sample pushed -- in calls of every size, as float or double, borrowed or handed over -- must reach its batch in
// order, every handed-over buffer must be released exactly once, batches must come back in order.  CPU only; run under
// ThreadSanitizer and AddressSanitizer + UBSan by tools/run_sanitizers.sh.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../fun_ofdm_amd/csrc/stream_core.h"

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

struct FakeGpu {
    int64_t B;
    static constexpr int kBufs = foa::StreamCore<FakeGpu>::kSlots;
    std::vector<float> stage_buf[kBufs];
    std::vector<float> dev[kBufs];                 // what stage() uploaded, per slot
    int upload_polls[kBufs] = {};
    std::vector<float> received;                   // the stream as the "device" saw it
    struct Fl { uint64_t h; int64_t n; int polls; };
    std::deque<Fl> flight;
    uint64_t next = 1;
    int fail_at = -1, submits = 0;
    explicit FakeGpu(int64_t b) : B(b) { for (auto &s : stage_buf) s.assign(2 * b, -1.0f); }
    float *staging(int slot) { return stage_buf[slot].data(); }
    int stage(int slot, int64_t n_new, bool)
    {
        dev[slot].assign(stage_buf[slot].begin(), stage_buf[slot].begin() + 2 * n_new);
        std::fill(stage_buf[slot].begin(), stage_buf[slot].end(), -1.0f);   // a sample that arrives late would be lost
        upload_polls[slot] = 0;
        return 0;
    }
    bool uploaded(int slot) { return ++upload_polls[slot] > 2; }            // "still copying" a couple of times
    int submit(int slot, int64_t n_new, bool final, uint64_t *handle)
    {
        if (submits++ == fail_at) return -3;
        if ((int64_t)dev[slot].size() != 2 * n_new) return -2;              // submitted without (or with another batch's) upload
        received.insert(received.end(), dev[slot].begin(), dev[slot].end());
        dev[slot].clear();
        if (!final && n_new != B) return -1;
        flight.push_back(Fl{ next, n_new, 0 });
        *handle = next++;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        return 0;
    }
    int collect(uint64_t h, bool wait, foa::StreamReady *out)
    {
        if (flight.empty() || flight.front().h != h) return -5;
        if (!wait && ++flight.front().polls < 3) return 0;                  // "not finished yet" a few times
        out->len.push_back(8);
        const int64_t n = flight.front().n;
        out->bytes.resize(8);
        memcpy(out->bytes.data(), &n, 8);
        flight.pop_front();
        return 1;
    }
};

static std::atomic<int> g_released(0);
static void release_vec(void *p) { g_released++; delete (std::vector<double> *)p; }

static void run(int64_t B, int helpers, size_t total, unsigned seed, size_t max_push, int slots = 6, int spin_us = 0)
{
    std::mt19937 rng(seed);
    std::vector<double> src(2 * total);
    for (size_t i = 0; i < src.size(); i++) src[i] = (double)(float)((int)(rng() % 2000001) - 1000000) / 1024.0;
    std::vector<float> srcf(src.begin(), src.end());
    FakeGpu gpu(B);
    std::vector<int64_t> batches;
    int handed = 0;
    g_released = 0;
    {
        foa::StreamCore<FakeGpu> core(&gpu, B, helpers, slots, spin_us);
        auto drain = [&](bool wait) {
            foa::StreamReady r;
            while (core.take(wait, &r) == 1) { int64_t n; memcpy(&n, r.bytes.data(), 8); batches.push_back(n); r = foa::StreamReady(); }
        };
        size_t o = 0;
        while (o < total) {
            const size_t n = std::min(total - o, (size_t)(1 + rng() % max_push));
            const int kind = rng() % 3;
            int rc;
            if (kind == 0) rc = core.push(srcf.data() + 2 * o, n, nullptr, nullptr);
            else if (kind == 1) rc = core.push(src.data() + 2 * o, n, nullptr, nullptr);
            else {
                auto *v = new std::vector<double>(src.begin() + 2 * o, src.begin() + 2 * (o + n));
                handed++;
                rc = core.push(v->data(), n, release_vec, v);
            }
            CHECK(rc == 0, "push failed: %d", rc);
            o += n;
            if (rng() % 4 == 0) drain(false);
        }
        CHECK(core.flush() == 0, "flush failed");
        CHECK(core.push(srcf.data(), 1, nullptr, nullptr) != 0, "a push after the flush must fail");
        drain(true);
        CHECK(core.pushed() == (int64_t)total, "pushed %lld", (long long)core.pushed());
    }
    CHECK(g_released == handed, "released %d of %d handed-over buffers", (int)g_released, handed);
    CHECK(gpu.received.size() == 2 * total, "the device saw %zu of %zu samples", gpu.received.size() / 2, total);
    size_t bad = 0;
    for (size_t i = 0; i < std::min(gpu.received.size(), srcf.size()); i++) bad += gpu.received[i] != srcf[i];
    CHECK(bad == 0, "%zu sample values differ", bad);
    const size_t want_batches = total / B + 1;
    CHECK(batches.size() == want_batches, "%zu batches came back, expected %zu", batches.size(), want_batches);
    for (size_t k = 0; k < batches.size(); k++) CHECK(batches[k] == (k + 1 < want_batches ? B : (int64_t)(total % B)), "batch %zu has %lld samples", k, (long long)batches[k]);
    printf("B %lld, %d helpers, %d slots, %zu samples, pushes up to %zu: %zu batches, %d buffers handed over\n", (long long)B, helpers, slots, total, max_push, batches.size(), handed);
}

int main()
{
    run(4096, 0, 100000, 1, 3000);
    run(4096, 3, 300000, 2, 9000);
    run(65536, 4, 2000000, 3, 300000);      // pushes larger than a batch, large borrowed pushes shared with the helpers
    run(8192, 2, 500000, 4, 100);           // many tiny pushes
    run(5000, 1, 65000, 5, 20000);          // total a multiple of the batch: the final batch is empty
    run(4096, 3, 400000, 6, 9000, 12);      // twelve slots in rotation (streams of small batches)
    run(4096, 1, 100000, 7, 5000, 3);       // three
    run(4096, 2, 300000, 8, 6000, 12, 300); // a submitter that polls for 300 us after the last batch before it sleeps
    {   // an error on the submitter thread reaches the caller
        FakeGpu gpu(4096);
        gpu.fail_at = 2;
        foa::StreamCore<FakeGpu> core(&gpu, 4096, 2);
        std::vector<float> z(2 * 4096, 0.0f);
        int rc = 0;
        for (int i = 0; i < 50 && !rc; i++) rc = core.push(z.data(), 4096, nullptr, nullptr);
        if (!rc) rc = core.flush();
        foa::StreamReady r;
        int t;
        while ((t = core.take(true, &r)) == 1) {}
        CHECK(rc == -3 || t == -3, "the backend's error was not reported (push/flush %d, take %d)", rc, t);
    }
    {   // a producer with a bounded POOL of buffers, smaller than the queue's publishing stride: it hands over what it has
        // and then waits for buffers to come back WITHOUT calling into the engine -- they must come back all the same
        FakeGpu gpu(1 << 20);                       // one open batch, never full: nothing closes it
        foa::StreamCore<FakeGpu> core(&gpu, 1 << 20, 2);
        g_released = 0;
        for (int i = 0; i < 8; i++) {
            auto *v = new std::vector<double>(2 * 4096, 0.5);
            CHECK(core.push(v->data(), 4096, release_vec, v) == 0, "push failed");
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (g_released < 8 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(3)) std::this_thread::sleep_for(std::chrono::microseconds(200));
        const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
        CHECK(g_released == 8, "a pooled producer got %d of 8 buffers back while it made no call (%.0f ms)", (int)g_released, ms);
        printf("pooled producer: 8 of 8 handed-over buffers back after %.2f ms without a call\n", ms);
        // ... and one that polls take(false) meanwhile
        g_released = 0;
        for (int i = 0; i < 5; i++) {
            auto *v = new std::vector<double>(2 * 1000, 0.5);
            CHECK(core.push(v->data(), 1000, release_vec, v) == 0, "push failed");
        }
        foa::StreamReady r;
        const auto t1 = std::chrono::steady_clock::now();
        while (g_released < 5 && std::chrono::steady_clock::now() - t1 < std::chrono::seconds(3)) { (void)core.take(false, &r); std::this_thread::sleep_for(std::chrono::microseconds(50)); }
        CHECK(g_released == 5, "a polling producer got %d of 5 buffers back", (int)g_released);
        CHECK(core.flush() == 0, "flush failed");
        while (core.take(true, &r) == 1) r = foa::StreamReady();
    }
    {   // handed-over buffers far larger than a narrowing task (a call of a million samples): sliced, every sample in place, released once
        const int64_t B = 300000;
        const size_t total = 2500000, chunk = 1 << 20;
        std::vector<float> want(2 * total);
        FakeGpu gpu(B);
        g_released = 0;
        int handed = 0;
        {
            foa::StreamCore<FakeGpu> core(&gpu, B, 3);
            foa::StreamReady r;
            for (size_t o = 0; o < total; o += chunk) {
                const size_t n = std::min(chunk, total - o);
                auto *v = new std::vector<double>(2 * n);
                for (size_t i = 0; i < 2 * n; i++) { (*v)[i] = (double)(float)((2 * o + i) % 100003) / 8.0; want[2 * o + i] = (float)(*v)[i]; }
                handed++;
                CHECK(core.push(v->data(), n, release_vec, v) == 0, "push failed");
                while (core.take(false, &r) == 1) r = foa::StreamReady();
            }
            CHECK(core.flush() == 0, "flush failed");
            while (core.take(true, &r) == 1) r = foa::StreamReady();
        }
        CHECK(g_released == handed, "released %d of %d large buffers", (int)g_released, handed);
        CHECK(gpu.received == want, "samples of the large handed-over calls differ");
        printf("large handed-over calls: %d buffers of up to %zu samples, all in place\n", handed, chunk);
    }
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
