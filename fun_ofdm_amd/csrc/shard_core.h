// shard_core.h -- one stream, several devices: the dealing and ordering logic of foa_shard_* (include/fun_ofdm_amd.h), free of any GPU
// call so that it runs against device doubles on a CPU (tests/cpp/shard_core_test.cpp, under ThreadSanitizer with the rest of the core).
//
// ShardBackend<Dev> is a Backend of StreamCore (stream_core.h) like the single-device StreamGpu (stream_engine.h): the core's caller /
// helper / submitter threads, its staging slots and its in-order hand-over of finished batches are the same.  What changes is where a
// batch goes: batch k of the stream is uploaded to, pre-synchronised on and decoded by device k mod N, each device through its own
// receiver handle (its own streams, work sets and job slots), so N devices work on N consecutive batches at once and the payloads
// still come out in stream order.
//
// Two things cross batches in the single-device engine, and both had stayed on the device there:
//   * the CARRY: buffer k = the last C samples before batch k + the batch (stream_engine.h).  Here the host keeps those C samples in
//     page-locked memory (one carry per staging slot, built when the batch before is staged) and every buffer is two host-to-device
//     copies -- no device ever reads another device's memory;
//   * the CHAIN STATE: where the first alignment not decided yet begins, and the phasor timing_sync left in force before it
//     (timing_sync.cpp:113-125) -- 24 bytes.  It is a chain through the batches in stream order, hence through the devices in turn: batch
//     k's look-ahead (which of the buffer's alignments can be decided with the samples there are: stream_engine.h) is queued as soon as
//     batch k-1's has finished and handed its state to the host, which passes it on -- on a stream of its own on the device, behind
//     nothing but the buffer's own pre-sync.  Everything before it (upload, the frame_detector / timing_sync kernels over the whole
//     buffer) and everything after it (the decode call) runs without waiting for any other device.
// Dev (one device) provides, all non-blocking unless said otherwise:
//   int  upload(int slot, const float *carry, const float *batch, int64_t n_new, int64_t start)   H2D of carry (C samples) + batch into the
//                                                     device buffer of `slot`, then the pre-sync kernels over it; start = stream index of the buffer's first sample
//   int  select(int slot, int64_t n_eff, bool final, const ChainState &in)   behind the pre-sync: decide, from in.lo_abs on, the alignments that can be
//                                                     decided with the buffer's first n_eff samples (final: all of them)
//   int  selected(int slot, ChainState *out)          1: through (*out = the state after this batch), 0: not yet, < 0 error
//   int  decode(int slot, int64_t n_new, uint64_t *handle)    queue the decode of the alignments selected (may block on the device's own pipeline)
//   int  collect(uint64_t handle, bool wait, StreamReady *out)    1 done, 0 not yet, < 0 error
#pragma once

#include <cstring>
#include <deque>
#include <vector>

#include "stream_core.h"

namespace foa {

struct ChainState { int64_t lo_abs; double c, s; };         // (= StreamState of the device code, foa_common.h)
constexpr int64_t kShardSettle = 192;                       // (= kStreamSettle: tags this close to a buffer's end are not final)

template <typename Dev>
class ShardBackend {
public:
    static constexpr int kSlots = 6;               // = StreamCore<>::kSlots (asserted where both are instantiated)

    // devs: one per device, in dealing order; staging / carry: page-locked buffers every device can read (kSlots x batch float2, kSlots x carry float2)
    ShardBackend(std::vector<Dev *> devs, int64_t batch, int64_t carry, int64_t longest, float *const *staging, float *const *carry_bufs)
        : devs_(std::move(devs)), B_(batch), C_(carry), L_(longest)
    {
        for (int i = 0; i < kSlots; i++) { stage_[i] = staging[i]; carry_[i] = carry_bufs[i]; }
        memset(carry_[0], 0, (size_t)C_ * 8);      // silence before the stream
    }
    int n_devices() const { return (int)devs_.size(); }
    int64_t batches() const { return n_submitted_; }
    int device_of_batch(int64_t k) const { return (int)(k % (int64_t)devs_.size()); }

    float *staging(int slot) { return stage_[slot]; }

    int stage(int slot, int64_t n_new, bool final)
    {
        const int64_t k = n_staged_;
        Info &in = info_[slot];
        in = Info();
        in.batch = k; in.dev = device_of_batch(k); in.n_new = n_new; in.final = final;
        const int64_t pushed = staged_samples_ + n_new;
        const int64_t start = pushed - n_new - C_;                      // stream index of the buffer's first sample
        in.n_eff = final ? C_ + n_new : C_ + n_new - kShardSettle;      // the buffer's tags are final up to here
        // the carry of the NEXT batch, while this batch's samples are still in their staging slot (the slot is the caller's again once
        // this batch has been submitted): the last C samples of (this carry ++ this batch)
        float *next = carry_[(slot + 1) % kSlots];
        if (n_new >= C_) {
            memcpy(next, stage_[slot] + 2 * (n_new - C_), (size_t)C_ * 8);
        } else {
            memcpy(next, carry_[slot] + 2 * n_new, (size_t)(C_ - n_new) * 8);
            memcpy(next + 2 * (C_ - n_new), stage_[slot], (size_t)n_new * 8);
        }
        const int rc = devs_[in.dev]->upload(slot, carry_[slot], stage_[slot], n_new, start);
        in.staged = rc == 0;
        staged_samples_ = pushed; n_staged_++;
        // A batch whose upload failed never gets a look-ahead: the chain steps over it (the core reports the error and does not submit
        // it, so its slot record may be gone by the time the chain gets there -- hence a list of its own), the state passes through as
        // if the batch had decided nothing, and the batches behind it still get their turn, so that a forced submit, flush and destroy
        // all come back instead of waiting for a selection that is never queued.
        if (rc != 0) { in.failed = in.sel_done = true; skipped_.push_back(k); }
        advance_chain();
        return rc;
    }

    // The chain: the look-ahead of batch chain_next_ is queued the moment that batch is staged and the batch before has handed its state on
    // -- not when the core gets round to asking about it (the core asks about the OLDEST staged batch only, i.e. after the batch before has
    // been submitted: with a decode call's worth of host work in between, every batch paid that on the chain).
    void advance_chain()
    {
        for (;;) {
            while (!skipped_.empty() && skipped_.front() <= chain_next_) { if (skipped_.front() == chain_next_) chain_next_++; skipped_.pop_front(); }
            int slot = -1;
            for (int i = 0; i < kSlots; i++) if (info_[i].batch == chain_next_ && info_[i].staged) { slot = i; break; }
            if (slot < 0) return;
            Info &in = info_[slot];
            if (in.sel_done) return;                                    // (cannot happen: chain_next_ moves on when a selection is through)
            if (!in.sel_queued) {
                if (devs_[in.dev]->select(slot, in.n_eff, in.final, state_) != 0) { in.sel_done = true; in.failed = true; chain_next_ = in.batch + 1; continue; }   // (submit reports it)
                in.sel_queued = true;
                return;
            }
            ChainState out;
            const int r = devs_[in.dev]->selected(slot, &out);
            if (r == 0) return;
            if (r < 0) in.failed = true;
            else state_ = out;
            in.sel_done = true;
            chain_next_ = in.batch + 1;
        }
    }

    bool uploaded(int slot)
    {
        advance_chain();
        return info_[slot].sel_done;
    }

    int submit(int slot, int64_t n_new, bool final, uint64_t *handle)
    {
        (void)final;
        Info &in = info_[slot];
        while (!uploaded(slot)) std::this_thread::yield();              // (the core calls submit once uploaded() said yes; a forced submit -- all slots staged -- waits here)
        if (in.failed) return -3;
        uint64_t h = 0;
        const int rc = devs_[in.dev]->decode(slot, n_new, &h);
        if (rc) return rc;
        flight_.push_back(Flight{ next_handle_, in.dev, h });
        *handle = next_handle_++;
        n_submitted_++;
        return 0;
    }

    int collect(uint64_t handle, bool wait, StreamReady *out)
    {
        if (flight_.empty() || flight_.front().handle != handle) return -5;
        const Flight f = flight_.front();
        const int rc = devs_[f.dev]->collect(f.dev_handle, wait, out);
        if (rc != 0) flight_.pop_front();
        return rc;
    }

private:
    struct Info { int64_t batch = -1, n_new = 0, n_eff = 0; int dev = 0; bool staged = false, final = false, sel_queued = false, sel_done = false, failed = false; };
    struct Flight { uint64_t handle; int dev; uint64_t dev_handle; };
    std::vector<Dev *> devs_;
    const int64_t B_, C_, L_;
    float *stage_[kSlots], *carry_[kSlots];
    Info info_[kSlots];
    int64_t n_staged_ = 0, n_submitted_ = 0, staged_samples_ = 0, chain_next_ = 0;
    ChainState state_ = { 0, 1.0, 0.0 };           // nothing decided yet; timing_sync's m_phase_acc before the first frame: 0
    std::deque<Flight> flight_;
    std::deque<int64_t> skipped_;                  // batches whose upload failed, in stream order: the chain steps over them
    uint64_t next_handle_ = 1;
};

}  // namespace foa
