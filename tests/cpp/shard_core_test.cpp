// shard_core_test.cpp -- one stream dealt over N devices (fun_ofdm_amd/csrc/shard_core.h) against device doubles: batch k must go to
// device k mod N with exactly the C samples before it as its carry, the look-ahead of batch k must be queued only after batch k-1's has
// finished and with the chain state that one reported (and as soon as it has: not behind the decode call of the batch before), payloads
// must come back in stream order whatever the devices' speeds.  CPU only; run
// under ThreadSanitizer and AddressSanitizer + UBSan by tools/run_sanitizers.sh.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <unistd.h>

#include "../../fun_ofdm_amd/csrc/shard_core.h"

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

// what every device double reports to
struct World {
    int64_t B, C, L;
    std::vector<float> stream;                      // the samples as pushed (interleaved re, im), for checking carries
    std::mutex m;
    int64_t uploads = 0, selects_queued = 0, selects_done = 0, decodes = 0;
    int64_t next_select_batch = 0;                  // the chain: selections must be queued in batch order, each after the previous finished
    foa::ChainState last = { 0, 1.0, 0.0 };
    int64_t eager = 0;                              // look-aheads queued before the batch before them had been decoded (the chain runs ahead of the decode calls)
    std::vector<int> device_of_batch;
    int bad = 0;
    int64_t fail_upload_at = -1;                    // the upload of this batch fails (the chain must step over it, nothing may hang)
    std::set<int64_t> failed;
};

struct FakeDev {
    World *w; int id; int n_dev;
    struct Buf { int64_t batch = -1, n_new = 0, start = 0, n_eff = 0; int polls = 0; bool selected = false; foa::ChainState in = { 0, 0, 0 }; };
    Buf buf[6];
    std::deque<std::pair<uint64_t, int64_t> > flight;      // (handle, batch)
    uint64_t next = 1;
    int slow;                                       // polls until a selection / a batch reports done
    FakeDev(World *world, int i, int n, int slowness) : w(world), id(i), n_dev(n), slow(slowness) {}

    int upload(int slot, const float *carry, const float *batch, int64_t n_new, int64_t start)
    {
        std::lock_guard<std::mutex> lk(w->m);
        const int64_t k = w->uploads++;
        w->device_of_batch.push_back(id);
        if (id != (int)(k % n_dev)) { w->bad++; printf("batch %lld went to device %d\n", (long long)k, id); }
        if (k == w->fail_upload_at) { w->failed.insert(k); buf[slot] = Buf(); return -7; }
        // the carry: the C samples before the batch (zeros before the stream's start); the batch: the next n_new samples of the stream
        const int64_t b0 = start + w->C;            // stream index of the batch's first sample
        for (int64_t i = 0; i < w->C; i++) {
            const int64_t x = start + i;
            const float re = x < 0 ? 0.0f : w->stream[2 * x], im = x < 0 ? 0.0f : w->stream[2 * x + 1];
            if (carry[2 * i] != re || carry[2 * i + 1] != im) { w->bad++; printf("batch %lld: carry sample %lld differs\n", (long long)k, (long long)i); break; }
        }
        for (int64_t i = 0; i < n_new; i++)
            if (batch[2 * i] != w->stream[2 * (b0 + i)] || batch[2 * i + 1] != w->stream[2 * (b0 + i) + 1]) { w->bad++; printf("batch %lld: sample %lld differs\n", (long long)k, (long long)i); break; }
        buf[slot] = Buf();
        buf[slot].batch = k; buf[slot].n_new = n_new; buf[slot].start = start;
        return 0;
    }
    int select(int slot, int64_t n_eff, bool final, const foa::ChainState &in)
    {
        std::lock_guard<std::mutex> lk(w->m);
        Buf &b = buf[slot];
        while (w->failed.count(w->next_select_batch)) { w->next_select_batch++; w->selects_done++; }      // (a batch that was never uploaded has no look-ahead)
        if (b.batch != w->next_select_batch || w->selects_done != b.batch) { w->bad++; printf("look-ahead of batch %lld queued out of turn (%lld done)\n", (long long)b.batch, (long long)w->selects_done); }
        if (in.lo_abs != w->last.lo_abs || in.c != w->last.c || in.s != w->last.s) { w->bad++; printf("batch %lld got the wrong chain state\n", (long long)b.batch); }
        // tags are final up to kShardSettle before the buffer's end (all of it when the stream is over)
        if (n_eff != w->C + b.n_new - (final ? 0 : foa::kShardSettle)) { w->bad++; printf("batch %lld: n_eff %lld\n", (long long)b.batch, (long long)n_eff); }
        // (behind a batch that was lost an undecided frame may begin before the buffer: the stream is in error by then)
        if (in.lo_abs < b.start && w->failed.empty()) { w->bad++; printf("batch %lld: the first undecided alignment lies before its buffer\n", (long long)b.batch); }
        if (w->decodes < b.batch) w->eager++;
        b.n_eff = n_eff; b.in = in;
        w->selects_queued++;
        w->next_select_batch++;
        return 0;
    }
    int selected(int slot, foa::ChainState *out)
    {
        Buf &b = buf[slot];
        if (++b.polls < slow) return 0;
        std::lock_guard<std::mutex> lk(w->m);
        // every third batch decides no frame (the phasor passes through unchanged); the others move on to somewhere in the last L samples
        // they could decide (an undecided frame is at most L long)
        if (b.batch % 3 == 2) { *out = b.in; out->lo_abs = std::max(b.in.lo_abs, b.start + b.n_eff - w->L); }      // (a frame stays undecided for L samples at most)
        else { out->lo_abs = std::max(b.in.lo_abs, b.start + b.n_eff - (int64_t)((b.batch * 7919) % (w->L + 1))); out->c = 0.001 * (double)(b.batch + 1); out->s = (double)id; }
        w->last = *out;
        w->selects_done++;
        b.selected = true;
        return 1;
    }
    int decode(int slot, int64_t n_new, uint64_t *handle)
    {
        std::lock_guard<std::mutex> lk(w->m);
        Buf &b = buf[slot];
        if (!b.selected || n_new != b.n_new) { w->bad++; printf("batch %lld decoded before its selection was through\n", (long long)b.batch); }
        w->decodes++;
        flight.push_back(std::make_pair(next, b.batch));
        *handle = next++;
        return 0;
    }
    int collect(uint64_t h, bool wait, foa::StreamReady *out)
    {
        if (flight.empty() || flight.front().first != h) return -5;
        static thread_local int polls = 0;
        if (!wait && ++polls % (slow + 1)) return 0;
        const int64_t k = flight.front().second;
        out->len.push_back(8);
        out->bytes.resize(8);
        memcpy(out->bytes.data(), &k, 8);
        flight.pop_front();
        return 1;
    }
};

static void run(int n_dev, int64_t B, int64_t C, int64_t L, size_t total, unsigned seed, size_t max_push, int helpers)
{
    std::mt19937 rng(seed);
    World w;
    w.B = B; w.C = C; w.L = L;
    w.stream.resize(2 * total);
    for (size_t i = 0; i < w.stream.size(); i++) w.stream[i] = (float)((int)(rng() % 20001) - 10000) / 64.0f;
    std::vector<FakeDev *> devs;
    for (int i = 0; i < n_dev; i++) devs.push_back(new FakeDev(&w, i, n_dev, 1 + (int)(rng() % 5)));
    std::vector<std::vector<float> > stage(6, std::vector<float>(2 * B)), carry(6, std::vector<float>(2 * C));
    float *sp[6], *cp[6];
    for (int i = 0; i < 6; i++) { sp[i] = stage[i].data(); cp[i] = carry[i].data(); }
    std::vector<int64_t> order;
    {
        foa::ShardBackend<FakeDev> be(devs, B, C, L, sp, cp);
        foa::StreamCore<foa::ShardBackend<FakeDev> > core(&be, B, helpers);
        auto drain = [&](bool wait) {
            foa::StreamReady r;
            while (core.take(wait, &r) == 1) { int64_t k; memcpy(&k, r.bytes.data(), 8); order.push_back(k); r = foa::StreamReady(); }
        };
        size_t o = 0;
        while (o < total) {
            const size_t n = std::min(total - o, (size_t)(1 + rng() % max_push));
            CHECK(core.push(w.stream.data() + 2 * o, n, nullptr, nullptr) == 0, "push failed");
            o += n;
            if (rng() % 3 == 0) drain(false);
        }
        CHECK(core.flush() == 0, "flush failed");
        drain(true);
    }
    const size_t want = total / B + 1;
    CHECK(order.size() == want, "%zu batches came back, expected %zu", order.size(), want);
    for (size_t k = 0; k < order.size(); k++) CHECK(order[k] == (int64_t)k, "batch %lld came back in place %zu", (long long)order[k], k);
    CHECK(w.bad == 0, "%d violations", w.bad);
    CHECK(w.uploads == (int64_t)want && w.selects_done == (int64_t)want && w.decodes == (int64_t)want, "uploads %lld, selections %lld, decodes %lld of %zu",
          (long long)w.uploads, (long long)w.selects_done, (long long)w.decodes, want);
    for (auto *d : devs) delete d;
    printf("%d devices, B %lld, C %lld, %zu samples, pushes up to %zu, %d helpers: %zu batches in order, %lld look-aheads queued ahead of the decode call before them\n", n_dev,
           (long long)B, (long long)C, total, max_push, helpers, order.size(), (long long)w.eager);
}

// The upload of one batch fails: every call must come back (push / flush with the error, the destructor at all) -- the chain of
// look-aheads steps over the batch instead of waiting for a selection that is never queued -- and the batches in front of it still come
// out in order.  A watchdog turns a hang into a failure.
static void run_upload_failure(int n_dev, int64_t fail_at, unsigned seed, bool one_push = false)
{
    const int64_t B = 4096, C = 1000, L = 600;
    const size_t total = 40 * (size_t)B;
    std::mt19937 rng(seed);
    World w;
    w.B = B; w.C = C; w.L = L; w.fail_upload_at = fail_at;
    w.stream.resize(2 * total);
    for (size_t i = 0; i < w.stream.size(); i++) w.stream[i] = (float)((int)(rng() % 20001) - 10000) / 64.0f;
    std::vector<FakeDev *> devs;
    for (int i = 0; i < n_dev; i++) devs.push_back(new FakeDev(&w, i, n_dev, 1 + (int)(rng() % 5)));
    std::vector<std::vector<float> > stage(6, std::vector<float>(2 * B)), carry(6, std::vector<float>(2 * C));
    float *sp[6], *cp[6];
    for (int i = 0; i < 6; i++) { sp[i] = stage[i].data(); cp[i] = carry[i].data(); }
    std::vector<int64_t> order;
    int push_rc = 0, flush_rc = 0;
    alarm(60);
    {
        foa::ShardBackend<FakeDev> be(devs, B, C, L, sp, cp);
        foa::StreamCore<foa::ShardBackend<FakeDev> > core(&be, B, 1);
        foa::StreamReady r;
        size_t o = 0;
        while (o < total && !push_rc) {
            const size_t n = one_push ? total : std::min(total - o, (size_t)(1 + rng() % 9000));      // (one push: batches behind the lost one are closed before it is staged)
            push_rc = core.push(w.stream.data() + 2 * o, n, nullptr, nullptr);
            o += n;
            while (core.take(false, &r) == 1) { int64_t k; memcpy(&k, r.bytes.data(), 8); order.push_back(k); r = foa::StreamReady(); }
        }
        flush_rc = core.flush();
        while (core.take(true, &r) == 1) { int64_t k; memcpy(&k, r.bytes.data(), 8); order.push_back(k); r = foa::StreamReady(); }
        CHECK(core.error() != 0, "the core does not report the failed upload");
    }
    alarm(0);
    CHECK(push_rc != 0 || flush_rc != 0, "neither push nor flush reported the failed upload of batch %lld", (long long)fail_at);
    for (size_t k = 0; k < order.size(); k++) CHECK(order[k] == (int64_t)k + (order[k] > fail_at ? 1 : 0), "batch %lld came back in place %zu", (long long)order[k], k);
    CHECK(w.bad == 0, "%d violations", w.bad);
    for (auto *d : devs) delete d;
    printf("%d devices, upload of batch %lld fails: push %d, flush %d, %zu batches came back, nothing hung\n", n_dev, (long long)fail_at, push_rc, flush_rc, order.size());
}

// ... and the backend alone, the way the core drives it when every slot is staged (a forced submit): the batch behind a lost one must
// get its look-ahead and its submit must return.
static void run_forced_submit_behind_a_lost_batch(int n_dev)
{
    const int64_t B = 4096, C = 1000, L = 600;
    World w;
    w.B = B; w.C = C; w.L = L; w.fail_upload_at = 1;
    w.stream.assign(2 * 6 * (size_t)B, 0.25f);
    std::vector<FakeDev *> devs;
    for (int i = 0; i < n_dev; i++) devs.push_back(new FakeDev(&w, i, n_dev, 3));
    std::vector<std::vector<float> > stage(6, std::vector<float>(2 * B)), carry(6, std::vector<float>(2 * C));
    float *sp[6], *cp[6];
    for (int i = 0; i < 6; i++) { sp[i] = stage[i].data(); cp[i] = carry[i].data(); }
    foa::ShardBackend<FakeDev> be(devs, B, C, L, sp, cp);
    int rcs[4];
    for (int k = 0; k < 4; k++) { for (int64_t i = 0; i < 2 * B; i++) sp[k][i] = 0.25f; rcs[k] = be.stage(k, B, false); }
    CHECK(rcs[0] == 0 && rcs[1] != 0 && rcs[2] == 0 && rcs[3] == 0, "stage: %d %d %d %d", rcs[0], rcs[1], rcs[2], rcs[3]);
    alarm(20);
    uint64_t h = 0;
    CHECK(be.submit(0, B, false, &h) == 0, "batch 0");
    CHECK(be.submit(2, B, false, &h) == 0, "batch 2, behind the lost one");      // (round 5's chain never got past batch 1: this call span for ever)
    CHECK(be.submit(3, B, false, &h) == 0, "batch 3");
    alarm(0);
    CHECK(w.bad == 0, "%d violations", w.bad);
    for (auto *d : devs) delete d;
    printf("%d devices, backend alone: submits behind a lost batch return\n", n_dev);
}

int main()
{
    run(1, 4096, 1000, 600, 100000, 1, 3000, 0);
    run(2, 4096, 1000, 600, 300000, 2, 9000, 2);
    run(8, 4096, 6000, 4000, 500000, 3, 20000, 3);          // a carry longer than a batch: it spans several batches
    run(8, 65536, 20000, 9000, 3000000, 4, 300000, 4);      // config 4's device count
    run(3, 5000, 700, 300, 65000, 5, 100, 1);               // total a multiple of the batch: the final batch is empty
    run_upload_failure(1, 3, 6);
    run_upload_failure(2, 0, 7);
    run_upload_failure(3, 7, 8);
    run_upload_failure(8, 5, 9);
    run_forced_submit_behind_a_lost_batch(1);
    run_forced_submit_behind_a_lost_batch(3);
    run_upload_failure(1, 0, 10, true);
    run_upload_failure(2, 1, 11, true);
    run_upload_failure(3, 2, 12, true);
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
