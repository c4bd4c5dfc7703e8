// blocks.hpp -- C++ host side of the drop-in: fun_ofdm's plug-in surface over the C ABI (fun_ofdm_amd.h).
//
//   fun_amd::receiver_chain      same call as fun::receiver_chain::process_samples (src/receiver_chain.h:56)
//   fun_amd::fft_symbols         fun::block<tagged_sample, tagged_vector<64>>      (src/fft_symbols.h)
//   fun_amd::channel_est         fun::block<tagged_vector<64>, tagged_vector<64>>  (src/channel_est.h)
//   fun_amd::phase_tracker       fun::block<tagged_vector<64>, tagged_vector<48>>  (src/phase_tracker.h)
//   fun_amd::frame_decoder       fun::block<tagged_vector<48>, std::vector<unsigned char>> (src/frame_decoder.h)
//   fun_amd::rx_backend          fun::block<tagged_sample, std::vector<unsigned char>>: the four above fused into one device call
//
// Each adaptor keeps the block's control state (tags, counters, partially collected frames) on the host exactly as
// the reference block does and sends the arithmetic of one work() call to the GPU in as few calls as possible.
// They are meant for swapping single stages inside an otherwise unchanged reference chain; for throughput use
// receiver_chain (or the batch C entry point), which keeps everything from the raw samples to the PSDU on the device.
//
// Inside the reference tree include the reference's block.h / tagged_vector.h FIRST: this header then uses those
// types as they are.  Stand-alone (this repository's tests) it declares layout-identical ones.
//
// Errors: the reference has no error channel (failed frames are dropped silently).  A failing GPU call throws
// std::runtime_error from work() / process_samples(); there is no CPU fallback.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include "../fun_ofdm_amd.h"

// ---- the reference's plug-in contract, declared only if its own headers are not already in --------------------
#ifndef BLOCK_H
#define BLOCK_H
#define BUFFER_MAX 65536
namespace fun {
// src/block.h:36-60: a named unit of work
class block_base {
public:
    explicit block_base(std::string block_name) : name(block_name) {}
    virtual ~block_base() {}
    virtual void work() = 0;
    std::string name;
};
// src/block.h:68-112: work() consumes all of input_buffer and replaces output_buffer; the chain swaps the
// buffers of neighbouring blocks between calls
template <typename I, typename O>
class block : public block_base {
public:
    explicit block(std::string block_name) : block_base(block_name)
    {
        input_buffer.reserve(BUFFER_MAX);
        output_buffer.reserve(BUFFER_MAX);
    }
    virtual void work() = 0;
    std::vector<I> input_buffer;
    std::vector<O> output_buffer;
};
}  // namespace fun
#endif

#ifndef TAGGED_VECTOR_H
#define TAGGED_VECTOR_H
namespace fun {
// src/tagged_vector.h:25-34
enum vector_tag { NONE, STS_START, STS_END, LTS_START, LTS1, LTS2, START_OF_FRAME };
// src/tagged_vector.h:43-76 (N complex doubles then the tag)
template <int N>
struct tagged_vector {
    std::complex<double> samples[N];
    vector_tag tag;
    tagged_vector(vector_tag _tag = NONE) { tag = _tag; }
};
// src/tagged_vector.h:82-95
struct tagged_sample {
    std::complex<double> sample;
    vector_tag tag;
    tagged_sample() { tag = NONE; }
};
}  // namespace fun
#endif

namespace fun_amd {

inline void check(int rc, const char *what)
{
    if (rc != FOA_OK) throw std::runtime_error(std::string(what) + ": " + foa_last_error());
}

// One GPU receiver handle, created on first use (work() runs on a thread the chain created, src/receiver_chain.cpp:65).
class device_handle {
public:
    explicit device_handle(int device = 0) : device_(device), rx_(nullptr) {}
    ~device_handle() { if (rx_) foa_rx_destroy(rx_); }
    device_handle(const device_handle &) = delete;
    device_handle &operator=(const device_handle &) = delete;
    foa_rx *get()
    {
        if (!rx_) check(foa_rx_create(&rx_, device_), "foa_rx_create");
        return rx_;
    }
private:
    int device_;
    foa_rx *rx_;
};

// ---------------------------------------------------------------------------------------------------------------
// fft_symbols (src/fft_symbols.cpp:33-79): tag-driven cyclic-prefix removal on the host, all FFTs of the call on the GPU
// ---------------------------------------------------------------------------------------------------------------
class fft_symbols : public fun::block<fun::tagged_sample, fun::tagged_vector<64> > {
public:
    explicit fft_symbols(int device = 0) : block("fft_symbols"), dev_(device), offset_(0) {}
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(0);
        for (size_t x = 0; x < input_buffer.size(); x++) {
            if (input_buffer[x].tag == fun::LTS1) {            // start of a frame: flush a vector in progress
                if (offset_ > 15) output_buffer.push_back(current_);
                current_.tag = fun::LTS_START;
                offset_ = 16;
            }
            if (input_buffer[x].tag == fun::LTS2) offset_ = 16;   // no cyclic prefix between the two LTS
            if (offset_ > 15) current_.samples[offset_ - 16] = input_buffer[x].sample;
            if (++offset_ == 80) {
                output_buffer.push_back(current_);
                current_.tag = fun::NONE;
                offset_ = 0;
            }
        }
        const size_t n = output_buffer.size();
        if (n == 0) return;
        std::vector<double> v(n * 128);
        for (size_t i = 0; i < n; i++) memcpy(&v[i * 128], output_buffer[i].samples, 128 * sizeof(double));
        check(foa_fft_forward_f64(dev_.get(), v.data(), n), "foa_fft_forward_f64");
        for (size_t i = 0; i < n; i++) memcpy(output_buffer[i].samples, &v[i * 128], 128 * sizeof(double));
    }
private:
    device_handle dev_;
    fun::tagged_vector<64> current_;
    int offset_;
};

// ---------------------------------------------------------------------------------------------------------------
// channel_est (src/channel_est.cpp:36-85)
// ---------------------------------------------------------------------------------------------------------------
class channel_est : public fun::block<fun::tagged_vector<64>, fun::tagged_vector<64> > {
public:
    explicit channel_est(int device = 0) : block("channel_est"), dev_(device), est_(128, 0.0), lts_flag_(0), frame_start_(false)
    {
        for (int j = 0; j < 64; j++) est_[2 * j] = 1.0;           // initial estimate 1+0j (channel_est.cpp:22)
    }
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(0);
        // pass 1: control flow.  Collect the LTS pairs that complete in this call and, for every vector that is
        // forwarded, which estimate applies: 0 = the one carried in, k = the k-th pair completed in this call.
        std::vector<double> pairs;                                 // k pairs x 2 x 64 complex
        std::vector<int32_t> which;
        std::vector<size_t> src;
        int cur = 0;
        for (size_t i = 0; i < input_buffer.size(); i++) {
            if (input_buffer[i].tag == fun::LTS_START) lts_flag_ = 1;
            if (lts_flag_ > 0) {
                if (lts_flag_ == 1) {
                    memcpy(lts1_, input_buffer[i].samples, sizeof lts1_);
                } else {
                    const size_t o = pairs.size();
                    pairs.resize(o + 256);
                    memcpy(&pairs[o], lts1_, sizeof lts1_);
                    memcpy(&pairs[o + 128], input_buffer[i].samples, sizeof lts1_);
                    cur = (int)(pairs.size() / 256);
                }
                if (++lts_flag_ == 3) { lts_flag_ = 0; frame_start_ = true; }
            } else {
                fun::tagged_vector<64> sym;
                if (frame_start_) { sym.tag = fun::START_OF_FRAME; frame_start_ = false; }
                output_buffer.push_back(sym);
                which.push_back(cur);
                src.push_back(i);
            }
        }
        const size_t np = pairs.size() / 256;
        std::vector<double> hinv((np + 1) * 128);
        memcpy(hinv.data(), est_.data(), 128 * sizeof(double));
        if (np) check(foa_channel_estimate_f64(dev_.get(), pairs.data(), hinv.data() + 128, np), "foa_channel_estimate_f64");
        if (np) memcpy(est_.data(), hinv.data() + np * 128, 128 * sizeof(double));
        // pass 2: equalise everything that is forwarded, one GPU call
        const size_t n = output_buffer.size();
        if (n == 0) return;
        std::vector<double> v(n * 128);
        for (size_t i = 0; i < n; i++) memcpy(&v[i * 128], input_buffer[src[i]].samples, 128 * sizeof(double));
        check(foa_equalize_f64(dev_.get(), v.data(), n, hinv.data(), np + 1, which.data()), "foa_equalize_f64");
        for (size_t i = 0; i < n; i++) memcpy(output_buffer[i].samples, &v[i * 128], 128 * sizeof(double));
    }
private:
    device_handle dev_;
    std::vector<double> est_;        // m_chan_est as (re,im) pairs
    double lts1_[128];
    int lts_flag_;
    bool frame_start_;
};

// ---------------------------------------------------------------------------------------------------------------
// phase_tracker (src/phase_tracker.cpp:70-104)
// ---------------------------------------------------------------------------------------------------------------
class phase_tracker : public fun::block<fun::tagged_vector<64>, fun::tagged_vector<48> > {
public:
    explicit phase_tracker(int device = 0) : block("phase_tracker"), dev_(device), symbol_count_(0) {}
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        const size_t n = input_buffer.size();
        output_buffer.resize(n);
        std::vector<double> v(n * 128), o(n * 96);
        std::vector<int32_t> cnt(n);
        for (size_t i = 0; i < n; i++) {
            if (input_buffer[i].tag == fun::START_OF_FRAME) symbol_count_ = 0;
            cnt[i] = symbol_count_++;
            memcpy(&v[i * 128], input_buffer[i].samples, 128 * sizeof(double));
        }
        check(foa_phase_track_f64(dev_.get(), v.data(), cnt.data(), n, o.data()), "foa_phase_track_f64");
        for (size_t i = 0; i < n; i++) {
            memcpy(output_buffer[i].samples, &o[i * 96], 96 * sizeof(double));
            output_buffer[i].tag = input_buffer[i].tag;
        }
    }
private:
    device_handle dev_;
    int symbol_count_;
};

// ---------------------------------------------------------------------------------------------------------------
// frame_decoder (src/frame_decoder.cpp:45-91 over ppdu::decode_header / decode_data, src/ppdu.cpp:168-295)
// ---------------------------------------------------------------------------------------------------------------
class frame_decoder : public fun::block<fun::tagged_vector<48>, std::vector<unsigned char> > {
public:
    explicit frame_decoder(int device = 0) : block("frame_decoder"), dev_(device), sample_count_(0), copied_(0), rate_(0), length_(0) {}
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(0);
        const size_t n = input_buffer.size();
        // all SIGNAL symbols of this call in one GPU call (a header's outcome does not depend on decoder state)
        std::vector<size_t> sof;
        for (size_t x = 0; x < n; x++) if (input_buffer[x].tag == fun::START_OF_FRAME) sof.push_back(x);
        std::vector<foa_frame_result> hdr(sof.size());
        if (!sof.empty()) {
            std::vector<double> c(sof.size() * 96);
            for (size_t i = 0; i < sof.size(); i++) memcpy(&c[i * 96], input_buffer[sof[i]].samples, 96 * sizeof(double));
            check(foa_decode_header_f64(dev_.get(), c.data(), sof.size(), hdr.data()), "foa_decode_header_f64");
        }
        // the reference's state machine; completed frames are queued and decoded together afterwards
        std::vector<double> car;
        std::vector<uint64_t> off(1, 0);
        std::vector<foa_frame_result> fr;
        size_t h = 0;
        for (size_t x = 0; x < n; x++) {
            if (copied_ < sample_count_) {
                const size_t o = frame_.size();
                frame_.resize(o + 96);
                memcpy(&frame_[o], input_buffer[x].samples, 96 * sizeof(double));
                copied_ += 48;
            }
            if (copied_ >= sample_count_ && sample_count_ != 0) {
                car.insert(car.end(), frame_.begin(), frame_.end());
                off.push_back(car.size() / 2);
                foa_frame_result r;
                r.status = 0; r.rate = rate_; r.length = length_; r.num_symbols = sample_count_ / 48;
                fr.push_back(r);
                sample_count_ = 0;
            }
            if (input_buffer[x].tag == fun::START_OF_FRAME) {
                const foa_frame_result &r = hdr[h++];
                if (r.status != FOA_ST_OK) continue;                 // ppdu.cpp:187-203: parity / rate check failed
                rate_ = r.rate; length_ = r.length; sample_count_ = r.num_symbols * 48; copied_ = 0;
                frame_.clear();
            }
        }
        if (fr.empty()) return;
        std::vector<unsigned char> psdu(fr.size() * 4096);
        check(foa_decode_data_f64(dev_.get(), car.data(), off.data(), fr.size(), fr.data(), psdu.data(), 4096), "foa_decode_data_f64");
        for (size_t i = 0; i < fr.size(); i++)
            if (fr[i].status == FOA_ST_OK) output_buffer.push_back(std::vector<unsigned char>(psdu.begin() + i * 4096, psdu.begin() + i * 4096 + fr[i].length));
    }
private:
    device_handle dev_;
    int sample_count_, copied_, rate_, length_;
    std::vector<double> frame_;       // carriers of the frame being collected
};

// An alignment less than 64 samples behind another takes its vectors one symbol late (the earlier one's LTS2 tag falls into its first LTS
// window: fun_ofdm_amd.h), so the device has to see the two together: an alignment that waits for more of the stream keeps the (dead)
// alignments of its pile-up waiting with it.  pos(i): LTS1 stream index of pending alignment i; done: leading alignments decided.
template <typename Pending, typename Pos>
inline size_t keep_pile_together(const Pending &pending, size_t done, Pos pos)
{
    while (done > 0 && done < pending.size() && pos(pending[done]) - pos(pending[done - 1]) < 64) done--;
    return done;
}
inline int64_t pile_pos_(int64_t p) { return p; }
inline size_t keep_pile_together(const std::deque<int64_t> &pending, size_t done) { return keep_pile_together(pending, done, pile_pos_); }

// ---------------------------------------------------------------------------------------------------------------
// rx_backend: fft_symbols + channel_est + phase_tracker + frame_decoder as ONE block for a chain that keeps the reference's
// own frame_detector and timing_sync threads: add_block(new fun_amd::rx_backend()) behind them instead of the four blocks
// (src/receiver_chain.cpp:29-51).  It consumes the tagged, rotated samples timing_sync hands on (src/timing_sync.cpp:98-125: LTS1 /
// LTS2 tags, every sample multiplied by the phasor in force) and makes ONE device call per work(): the fused kernels on those very
// doubles (foa_rx_decode_frames_f64_host).  Payloads come out in stream order, each by the call that brings its frame's last sample.
// ---------------------------------------------------------------------------------------------------------------
class rx_backend : public fun::block<fun::tagged_sample, std::vector<unsigned char> > {
public:
    explicit rx_backend(int device = 0) : block("rx_backend"), dev_(device), base_(0), need_end_(0) {}
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(0);
        const size_t n = input_buffer.size(), o = buf_.size();
        const int64_t pos0 = base_ + (int64_t)(o / 2);
        buf_.resize(o + 2 * n);
        for (size_t x = 0; x < n; x++) {
            buf_[o + 2 * x] = input_buffer[x].sample.real(); buf_[o + 2 * x + 1] = input_buffer[x].sample.imag();
            if (input_buffer[x].tag == fun::LTS1) pending_.push_back(pos0 + (int64_t)x);       // (LTS2 follows 64 samples on: timing_sync.cpp:105-106)
        }
        const int64_t avail = pos0 + (int64_t)n;
        // Every pending alignment goes to the device with the stream as far as it has come: an alignment ends where the next one's LTS1
        // re-aligns fft_symbols (it is linked to it: a frame cut short there fills on with what follows, fft_symbols.cpp:41-50 /
        // frame_decoder.cpp:52-88) or where the samples end for now.  FOA_ST_TRUNCATED = ran into that end: it and what is behind it
        // wait for the next call; everything else is final.  (The newest alignment alone, known to need more: no device call.)
        if (!pending_.empty() && !(pending_.size() == 1 && need_end_ > avail)) {
            const size_t m = pending_.size();
            std::vector<foa_frame_desc> d(m);
            std::vector<int64_t> e(m);
            for (size_t i = 0; i < m; i++) {
                d[i].lts1_pos = pending_[i] - base_; d[i].rot_start = d[i].lts1_pos;
                d[i].c = 1.0; d[i].s = 0.0; d[i].c_prev = 1.0; d[i].s_prev = 0.0;      // (not applied: the samples are rotated already)
                e[i] = (i + 1 < m ? pending_[i + 1] : avail) - base_;
            }
            std::vector<unsigned char> psdu(m * 4096);
            std::vector<foa_frame_result> res(m);
            check(foa_rx_decode_frames_f64_host(dev_.get(), buf_.data(), buf_.size() / 2, d.data(), e.data(), m, psdu.data(), 4096, res.data()),
                  "foa_rx_decode_frames_f64_host");
            size_t done = 0;
            for (; done < m; done++) {
                if (res[done].status != FOA_ST_TRUNCATED) continue;
                need_end_ = done + 1 == m ? pending_[done] + (res[done].rate >= 0 ? 144 + 80 * (int64_t)res[done].num_symbols + 64 : 208) : 0;
                break;
            }
            if (done == m) need_end_ = 0;
            done = keep_pile_together(pending_, done);
            for (size_t i = 0; i < done; i++)
                if (res[i].status == FOA_ST_OK) output_buffer.push_back(std::vector<unsigned char>(psdu.begin() + i * 4096, psdu.begin() + i * 4096 + res[i].length));
            pending_.erase(pending_.begin(), pending_.begin() + done);
        }
        // samples in front of the oldest pending alignment are never looked at again
        const int64_t keep_from = pending_.empty() ? avail : pending_.front();
        if (keep_from > base_) {
            buf_.erase(buf_.begin(), buf_.begin() + 2 * (size_t)(keep_from - base_));
            base_ = keep_from;
        }
    }
private:
    device_handle dev_;
    std::vector<double> buf_;          // the rotated stream from stream index base_ on, interleaved re, im
    int64_t base_;
    std::deque<int64_t> pending_;      // LTS1 positions (stream index) not decided yet
    int64_t need_end_;                 // the newest alignment, alone: the stream index its frame needs (0: unknown)
};

// ---------------------------------------------------------------------------------------------------------------
// receiver_chain: process_samples() with everything after the host-side pre-sync on the device (fused kernels)
// ---------------------------------------------------------------------------------------------------------------
class receiver_chain {
public:
    // async_batch_calls = 0: every call decodes what has completed and returns it (synchronous GPU call inside).
    // async_batch_calls = k > 0: a stream front end that must keep up with the air.  A lone frame takes the
    // GPU about a millisecond however short it is (its trellis is one serial chain), a batch of thousands hardly longer; so
    // every k-th call submits everything decodable as ONE asynchronous batch (foa_rx_submit_host: H2D, compute and D2H of
    // consecutive batches overlap) and every call hands out the payloads of the batches that have finished -- in stream
    // order, a few calls late, like the reference's five-call latency.  Same payloads as the synchronous mode.
    // device_batch_samples = B > 0: the whole of process_samples() on the device (foa_stream_*, SURVEY 8f #1 + #3): samples are
    // only narrowed to float on the host (narrow_threads helpers for large calls); every B samples one batch is uploaded,
    // pre-synchronised (frame_detector + timing_sync kernels), decoded and its payloads fetched, asynchronously; a call
    // returns the payloads of the batches that have finished, in stream order (latency: one batch; flush() at the end of a
    // capture).  Same payloads as the other modes.  This is the mode for sample rates far above real time.
    explicit receiver_chain(int device = 0, int async_batch_calls = 0, size_t device_batch_samples = 0, int narrow_threads = 0)
        : dev_(device), sync_(nullptr), base_(0), batch_calls_(async_batch_calls), calls_(0), stream_(nullptr), device_batch_(device_batch_samples),
          narrow_threads_(narrow_threads)
    {
        if (device_batch_ == 0) check(foa_sync_create(&sync_), "foa_sync_create");
    }
    // SEVERAL devices (BASELINE config 4 behind process_samples): device mode with batch k of the stream on devices[k mod n] (foa_shard_*,
    // include/fun_ofdm_amd.h) -- n consecutive batches in work at once, one receiver handle per entry of the list (a device may be listed
    // twice), payloads in stream order as ever.  Same payloads as every other mode.
    receiver_chain(const std::vector<int> &devices, size_t device_batch_samples, int narrow_threads = 0)
        : dev_(devices.empty() ? 0 : devices[0]), sync_(nullptr), base_(0), batch_calls_(0), calls_(0), stream_(nullptr), device_batch_(device_batch_samples),
          narrow_threads_(narrow_threads), devices_(devices)
    {
        if (devices_.empty() || device_batch_ == 0) throw std::invalid_argument("receiver_chain: a device list needs at least one device and a batch size");
    }
    ~receiver_chain()
    {
        try { while (!jobs_.empty()) collect_front(true, nullptr); } catch (...) {}
        if (stream_) foa_stream_destroy(stream_);
        if (shard_) foa_shard_destroy(shard_);
        if (sync_) foa_sync_destroy(sync_);
    }
    receiver_chain(const receiver_chain &) = delete;
    receiver_chain &operator=(const receiver_chain &) = delete;

    // One line of the reference depends on how ITS caller cuts the stream into calls: timing_sync.cpp:99 drops a frame whose STS_END is the
    // first sample a call walks over (include/fun_ofdm_amd.h, foa_sync_set_call).  This chain decides it for the call size of the
    // reference's own receiver, 4096 samples (receiver.h:16), whatever the sizes process_samples() is called with; a caller that replaces a
    // reference fed with another call size says so here before the first call (0: never drop a frame for that reason).
    void set_reference_call_size(long long samples)
    {
        if (!devices_.empty()) throw std::runtime_error("receiver_chain: the multi-device mode decides by the reference's own call size (4096)");
        if (sync_) check(foa_sync_set_call(sync_, samples), "foa_sync_set_call");
        check(foa_rx_set_option(dev_.get(), "sync_call", samples), "foa_rx_set_option");
    }

    // Device mode: the longest frame this stream will hold (samples, first preamble sample to last data sample, + 192); see option
    // "stream_longest" in include/fun_ofdm_amd.h -- the latency knob of a receiver that knows its traffic.  Before the first call.
    void set_stream_longest(long long samples)
    {
        if (!devices_.empty()) throw std::runtime_error("receiver_chain: the multi-device mode keeps the format's longest frame");
        check(foa_rx_set_option(dev_.get(), "stream_longest", samples), "foa_rx_set_option");
    }

    // The highest rate this stream's frames carry, as data bits per OFDM symbol (24 = 6 Mbps .. 216 = 54 Mbps, the default): option
    // "max_dbps" in include/fun_ofdm_amd.h -- the device's work sets are sized for it (a capture of low-rate frames needs a ninth of the
    // default per call), and a frame beyond the promise is dropped like a frame that fails its CRC.  Before the first call.
    void set_max_rate_bits(int dbps)
    {
        if (!devices_.empty()) throw std::runtime_error("receiver_chain: the multi-device mode keeps the default (any rate)");
        check(foa_rx_set_option(dev_.get(), "max_dbps", dbps), "foa_rx_set_option");
    }

    // Brings the device handle -- and, in device mode, the stream engine with its threads, page-locked buffers and work sets -- into being
    // NOW instead of under the first process_samples() call: creating a handle takes ~0.2 s (the runtime loads the library's kernels), in
    // which a live radio's samples would pile up.  After the set_* options (they must precede the engine); idempotent.  (The reference's
    // chain builds its blocks in its constructor, src/receiver_chain.cpp:32-74; here the options come in between.)
    void prepare()
    {
        if (device_batch_ == 0) { (void)dev_.get(); return; }
        if (stream_over_) return;                                       // after a flush: the next call starts a new stream
        if (!devices_.empty()) { if (!shard_) check(foa_shard_create(devices_.data(), (int)devices_.size(), device_batch_, narrow_threads_, &shard_), "foa_shard_create"); }
        else if (!stream_) check(foa_stream_create(dev_.get(), device_batch_, narrow_threads_, &stream_), "foa_stream_create");
    }

    // Same signature and meaning as fun::receiver_chain::process_samples (src/receiver_chain.cpp:106-126): feed the
    // next chunk of the 20 MS/s stream, get the payloads of the frames that completed, in stream order.  A frame is
    // returned by the call that delivers its last sample (the reference returns it five calls later).
    std::vector<std::vector<unsigned char> > process_samples(std::vector<std::complex<double> > samples)
    {
        if (device_batch_ > 0) return process_device(samples, false);
        if (batch_calls_ > 0) return process_async(samples, false);
        std::vector<std::vector<unsigned char> > out;
        if (samples.empty()) return out;
        append(samples);
        // Every pending alignment is handed to the device with the stream as far as its tags are FINAL (timing_sync may still place an
        // LTS1 up to 8 samples before the point it has settled): an alignment ends at the next one's LTS1 -- it is linked to it, and a
        // frame cut short there fills on with what follows, as in the reference (fun_ofdm_amd.h) -- or at that horizon.  Whatever comes
        // back FOA_ST_TRUNCATED ran into the horizon: it and everything behind it wait for more of the stream; everything else is final.
        const int64_t hz = horizon(false);
        // (the newest alignment alone, known to need more: no GPU call per chunk while a long frame comes in)
        if (pending_.size() == 1 && pending_[0].need_end > hz) { trim(); return out; }
        size_t take = 0;
        while (take < pending_.size() && pending_[take].d.lts1_pos < hz) take++;
        if (take == 0) { trim(); return out; }
        std::vector<foa_frame_desc> rel(take);
        std::vector<int64_t> rel_end(take);
        for (size_t i = 0; i < take; i++) {
            rel[i] = pending_[i].d;
            rel[i].lts1_pos -= base_; rel[i].rot_start -= base_;
            rel_end[i] = (i + 1 < pending_.size() ? pending_[i + 1].d.lts1_pos : hz) - base_;
        }
        std::vector<unsigned char> psdu(take * 4096);
        std::vector<foa_frame_result> res(take);
        check(foa_rx_decode_frames_host(dev_.get(), buf_.data(), (size_t)(hz - base_), rel.data(), rel_end.data(), take, psdu.data(), 4096, res.data()),
              "foa_rx_decode_frames_host");
        size_t done = 0;
        for (; done < take; done++) {
            if (res[done].status != FOA_ST_TRUNCATED) continue;
            // not complete yet: if it is the newest alignment, remember how far its frame reaches and try again then
            if (done + 1 == pending_.size())
                pending_[done].need_end = pending_[done].d.lts1_pos + (res[done].rate >= 0 ? 144 + 80 * (int64_t)res[done].num_symbols + 64 : 208);
            break;
        }
        done = keep_pile_together(pending_, done, entry_pos);
        for (size_t i = 0; i < done; i++)
            if (res[i].status == FOA_ST_OK) out.push_back(std::vector<unsigned char>(psdu.begin() + i * 4096, psdu.begin() + i * 4096 + res[i].length));
        pending_.erase(pending_.begin(), pending_.begin() + done);
        trim();
        return out;
    }

    // Asynchronous mode: everything that can still be decoded, waited for (end of a capture; a radio never ends).
    std::vector<std::vector<unsigned char> > flush()
    {
        if (device_batch_ > 0) { std::vector<std::complex<double> > none; return process_device(none, true); }
        if (batch_calls_ > 0) return process_async(std::vector<std::complex<double> >(), true);
        return std::vector<std::vector<unsigned char> >();
    }

private:
    struct entry {
        foa_frame_desc d;
        int64_t need_end;     // stream index the frame needs before it can be decoded (0: SIGNAL not decoded yet)
        explicit entry(const foa_frame_desc &x) : d(x), need_end(0) {}
    };
    static int64_t entry_pos(const entry &e) { return e.d.lts1_pos; }
    struct job { uint64_t ticket; size_t n_frames; bool final; std::vector<entry> sent; };     // sent: the alignments it decodes, stream-absolute
    static const int64_t kLongestFrame = 320 + 80 * 1369 + 160;       // samples: 4095 bytes at 6 Mbps, plus timing_sync's look-ahead

    // the device consumes complex<float> (BASELINE north_star); the pre-sync decides on the doubles it was given
    void append(const std::vector<std::complex<double> > &samples)
    {
        const size_t n = samples.size(), o = buf_.size();
        if (n == 0) return;
        buf_.resize(o + 2 * n);
        for (size_t i = 0; i < n; i++) { buf_[o + 2 * i] = (float)samples[i].real(); buf_[o + 2 * i + 1] = (float)samples[i].imag(); }
        std::vector<foa_frame_desc> found(n / 300 + 8);
        size_t got = 0;
        check(foa_sync_push_f64(sync_, reinterpret_cast<const double *>(samples.data()), n, found.data(), found.size(), &got), "foa_sync_push_f64");
        for (;;) {
            for (size_t i = 0; i < got; i++) pending_.push_back(entry(found[i]));
            if (got < found.size()) break;
            check(foa_sync_push_f64(sync_, nullptr, 0, found.data(), found.size(), &got), "foa_sync_push_f64");
        }
    }
    // Stream index up to which the tags are final: timing_sync has looked at everything before `settled` and may still place an LTS1 up
    // to 8 samples before it (at the end of a capture: everything there is).
    int64_t horizon(bool final) const
    {
        const int64_t avail = base_ + (int64_t)(buf_.size() / 2);
        return final ? avail : std::min(avail, foa_sync_settled(sync_) - 8);
    }
    // oldest batch in flight -> payloads (returns false if it is not finished and wait is false).  A frame that ran into the end of what
    // was known when its batch went out (FOA_ST_TRUNCATED; only a frame cut short by a later LTS1 and filling on beyond the horizon can)
    // is decoded again with more of the stream: it and everything sent behind it go back to the head of the pending list, in order.
    bool collect_front(bool wait, std::vector<std::vector<unsigned char> > *out)
    {
        job &j = jobs_.front();
        std::vector<unsigned char> psdu(j.n_frames * 4096);
        std::vector<foa_frame_result> res(j.n_frames);
        const int rc = foa_rx_collect(dev_.get(), j.ticket, wait ? 1 : 0, psdu.data(), res.data());
        if (rc < 0) check(rc, "foa_rx_collect");
        if (rc == 0) return false;
        size_t good = j.n_frames;
        if (!j.final) {
            for (size_t i = 0; i < j.n_frames; i++) if (res[i].status == FOA_ST_TRUNCATED) { good = i; break; }
            good = keep_pile_together(j.sent, good, entry_pos);
        }
        if (out)
            for (size_t i = 0; i < good; i++)
                if (res[i].status == FOA_ST_OK) out->push_back(std::vector<unsigned char>(psdu.begin() + i * 4096, psdu.begin() + i * 4096 + res[i].length));
        if (good < j.n_frames && out) {
            std::vector<entry> back(j.sent.begin() + good, j.sent.end());
            jobs_.pop_front();
            while (!jobs_.empty()) {                                   // what is in flight behind it was decided without it: decode it again too
                job &k = jobs_.front();
                std::vector<unsigned char> p2(k.n_frames * 4096);
                std::vector<foa_frame_result> r2(k.n_frames);
                const int rc2 = foa_rx_collect(dev_.get(), k.ticket, 1, p2.data(), r2.data());
                if (rc2 < 0) check(rc2, "foa_rx_collect");
                back.insert(back.end(), k.sent.begin(), k.sent.end());
                jobs_.pop_front();
            }
            pending_.insert(pending_.begin(), back.begin(), back.end());
            return true;
        }
        jobs_.pop_front();
        return true;
    }
    std::vector<std::vector<unsigned char> > process_async(const std::vector<std::complex<double> > &samples, bool final)
    {
        std::vector<std::vector<unsigned char> > out;
        append(samples);
        while (!jobs_.empty() && collect_front(false, &out)) {}
        calls_++;
        if (final || calls_ % batch_calls_ == 0) {
            const int64_t hz = horizon(final);
            // An alignment goes out once its extent is final: the next alignment is known (fft_symbols re-aligns there), or so much
            // stream has passed that neither its frame nor a later tag can reach back into it.  (Its length is not known here -- SIGNAL
            // is decoded on the device -- so the newest alignment of a burst waits for that.)  The alignments behind those go along as
            // CONTEXT (foa_rx_submit_host_ctx): a frame cut short by the next LTS1 may fill on with their vectors.
            size_t take = 0, known = 0;
            while (known < pending_.size() && pending_[known].d.lts1_pos < hz) known++;
            for (; take < known; take++) {
                const bool has_next = take + 1 < pending_.size();
                if (!has_next && !final && hz - pending_[take].d.lts1_pos < kLongestFrame) break;
            }
            if (!final) take = keep_pile_together(pending_, take, entry_pos);      // (a pile-up goes out in one batch)
            if (take > 0) {
                // the batch only needs the samples from just before its first alignment to the horizon
                const int64_t lo = std::max(base_, std::min(pending_[0].d.lts1_pos, pending_[0].d.rot_start) - 16);
                std::vector<foa_frame_desc> rel(known);
                std::vector<int64_t> rel_end(known);
                for (size_t i = 0; i < known; i++) {
                    rel[i] = pending_[i].d;
                    rel[i].lts1_pos -= lo; rel[i].rot_start -= lo;
                    rel_end[i] = (i + 1 < pending_.size() ? pending_[i + 1].d.lts1_pos : hz) - lo;
                }
                while (jobs_.size() >= 6) collect_front(true, &out);
                if (pending_.size() >= take && !pending_.empty() && pending_[0].d.lts1_pos - lo == rel[0].lts1_pos) {     // (a collect above may have put alignments back: next round then)
                    job j;
                    j.n_frames = take; j.final = final;
                    j.sent.assign(pending_.begin(), pending_.begin() + take);
                    check(foa_rx_submit_host_ctx(dev_.get(), buf_.data() + 2 * (lo - base_), (size_t)(hz - lo), rel.data(), rel_end.data(), take, known - take, 4096, &j.ticket),
                          "foa_rx_submit_host_ctx");
                    jobs_.push_back(j);
                    pending_.erase(pending_.begin(), pending_.begin() + take);
                }
            }
        }
        if (final) {
            while (!jobs_.empty()) collect_front(true, &out);
            if (!pending_.empty()) {                                     // put back by the last collects: one more, final round
                std::vector<std::vector<unsigned char> > rest = process_async(std::vector<std::complex<double> >(), true);
                out.insert(out.end(), rest.begin(), rest.end());
            }
        }
        trim();
        return out;
    }
    // everything on the device: push, then hand out whatever has finished (final: flush and wait for all of it)
    static void release_vector(void *p) { delete static_cast<std::vector<std::complex<double> > *>(p); }
    std::vector<std::vector<unsigned char> > process_device(std::vector<std::complex<double> > &samples, bool final)
    {
        std::vector<std::vector<unsigned char> > out;
        const bool multi = !devices_.empty();
        if (stream_over_) {                                             // a new stream after a flush
            if (stream_) { foa_stream_destroy(stream_); stream_ = nullptr; }
            if (shard_) { foa_shard_destroy(shard_); shard_ = nullptr; }
            stream_over_ = false;
        }
        if (multi && !shard_) check(foa_shard_create(devices_.data(), (int)devices_.size(), device_batch_, narrow_threads_, &shard_), "foa_shard_create");
        if (!multi && !stream_) check(foa_stream_create(dev_.get(), device_batch_, narrow_threads_, &stream_), "foa_stream_create");
        if (!samples.empty()) {
            // process_samples owns its argument (by value, src/receiver_chain.h:56): hand the buffer to the engine instead of
            // narrowing it here -- its helper threads do that while the caller fetches the next chunk
            std::vector<std::complex<double> > *own = new std::vector<std::complex<double> >(std::move(samples));
            if (multi) check(foa_shard_push_f64_owned(shard_, reinterpret_cast<const double *>(own->data()), own->size(), &receiver_chain::release_vector, own), "foa_shard_push_f64_owned");
            else check(foa_stream_push_f64_owned(stream_, reinterpret_cast<const double *>(own->data()), own->size(), &receiver_chain::release_vector, own),
                       "foa_stream_push_f64_owned");
        }
        if (final) { if (multi) check(foa_shard_flush(shard_), "foa_shard_flush"); else check(foa_stream_flush(stream_), "foa_stream_flush"); }
        for (;;) {
            size_t n = 0, bytes = 0;
            const int rc = multi ? foa_shard_ready(shard_, final ? 1 : 0, &n, &bytes) : foa_stream_ready(stream_, final ? 1 : 0, &n, &bytes);
            if (rc < 0) check(rc, multi ? "foa_shard_ready" : "foa_stream_ready");
            if (rc == 0) break;
            take_bytes_.resize(bytes ? bytes : 1);
            take_len_.resize(n ? n : 1);
            if (multi) check(foa_shard_take(shard_, take_bytes_.data(), take_len_.data()), "foa_shard_take");
            else check(foa_stream_take(stream_, take_bytes_.data(), take_len_.data()), "foa_stream_take");
            size_t o = 0;
            for (size_t i = 0; i < n; i++) {
                out.push_back(std::vector<unsigned char>(take_bytes_.begin() + o, take_bytes_.begin() + o + take_len_[i]));
                o += take_len_[i];
            }
        }
        if (final) stream_over_ = true;               // (its page-locked buffers are released by the next call or the destructor)
        return out;
    }
    // drop samples nothing can refer to any more: before the oldest pending alignment, and before what a future
    // alignment could reach back to (timing_sync places LTS1 at most 160+8 samples before the point it has reached)
    void trim()
    {
        if (!sync_) return;
        const int64_t settled = foa_sync_settled(sync_);
        int64_t keep_from = settled - 400;
        if (!pending_.empty()) keep_from = std::min(keep_from, std::min(pending_.front().d.lts1_pos, pending_.front().d.rot_start) - 16);
        if (!jobs_.empty() && !jobs_.front().sent.empty())               // (a batch in flight may have to be decoded again: collect_front)
            keep_from = std::min(keep_from, std::min(jobs_.front().sent.front().d.lts1_pos, jobs_.front().sent.front().d.rot_start) - 16);
        if (keep_from > base_) {
            const size_t drop = (size_t)(keep_from - base_);
            if (drop * 2 >= buf_.size()) { base_ += (int64_t)(buf_.size() / 2); buf_.clear(); }
            else { buf_.erase(buf_.begin(), buf_.begin() + 2 * drop); base_ = keep_from; }
        }
    }
    device_handle dev_;
    foa_sync *sync_;
    std::vector<float> buf_;          // interleaved samples from stream index base_ on
    int64_t base_;
    std::deque<entry> pending_;
    int batch_calls_;
    long calls_;
    std::deque<job> jobs_;
    foa_stream *stream_;              // device mode
    foa_shard *shard_ = nullptr;      // device mode over several devices
    std::vector<int> devices_;
    bool stream_over_ = false;
    size_t device_batch_;
    int narrow_threads_;
    std::vector<unsigned char> take_bytes_;
    std::vector<uint32_t> take_len_;
};

// ---------------------------------------------------------------------------------------------------------------
// The caller of the boundary: fun::receiver without the radio.
// ---------------------------------------------------------------------------------------------------------------
// What usrp::get_samples(num_samples, buffer) (src/usrp.cpp:125-130) is to fun::receiver: fills `buffer` with the next
// num_samples complex<double> samples.  Returning false ends the stream (a radio never does).
class sample_source {
public:
    virtual ~sample_source() {}
    virtual bool get_samples(int num_samples, std::vector<std::complex<double> > &buffer) = 0;
};

// Samples held in memory (tests, replays).
class vector_source : public sample_source {
public:
    explicit vector_source(std::vector<std::complex<double> > samples) : s_(std::move(samples)), pos_(0) {}
    bool get_samples(int num_samples, std::vector<std::complex<double> > &buffer) override
    {
        if (pos_ >= s_.size()) return false;
        buffer.assign((size_t)num_samples, std::complex<double>(0, 0));        // the last call is padded with silence
        const size_t n = std::min((size_t)num_samples, s_.size() - pos_);
        std::copy(s_.begin() + pos_, s_.begin() + pos_ + n, buffer.begin());
        pos_ += n;
        return true;
    }
private:
    std::vector<std::complex<double> > s_;
    size_t pos_;
};

// Raw interleaved I/Q files: "fc32" = complex<float> (GNU Radio / UHD rx_samples_to_file default),
// "fc64" = complex<double> (UHD's wire format for this reference, src/usrp.cpp:46).
class file_source : public sample_source {
public:
    file_source(const std::string &path, const std::string &format) : f_(std::fopen(path.c_str(), "rb")), fc64_(format == "fc64")
    {
        if (!f_) throw std::runtime_error("cannot open " + path);
        if (format != "fc32" && format != "fc64") throw std::runtime_error("format must be fc32 or fc64");
    }
    ~file_source() override { if (f_) std::fclose(f_); }
    file_source(const file_source &) = delete;
    file_source &operator=(const file_source &) = delete;
    bool get_samples(int num_samples, std::vector<std::complex<double> > &buffer) override
    {
        buffer.assign((size_t)num_samples, std::complex<double>(0, 0));
        size_t got;
        if (fc64_) {
            got = std::fread(buffer.data(), sizeof(std::complex<double>), (size_t)num_samples, f_);
        } else {
            tmp_.resize((size_t)num_samples);
            got = std::fread(tmp_.data(), sizeof(std::complex<float>), (size_t)num_samples, f_);
            for (size_t i = 0; i < got; i++) buffer[i] = std::complex<double>(tmp_[i].real(), tmp_[i].imag());
        }
        return got > 0;
    }
private:
    std::FILE *f_;
    bool fc64_;
    std::vector<std::complex<float> > tmp_;
};

// fun::receiver (src/receiver.h:35-112, src/receiver.cpp:27-78) over a sample_source instead of the USRP: a thread pulls
// NUM_RX_SAMPLES at a time, runs them through receiver_chain::process_samples() and hands the packets -- possibly none
// -- to the callback after every call, exactly like receiver_chain_loop(); pause() returns once the loop is parked
// between two iterations, resume() lets it run again (the reference's binary semaphore).  Unlike a radio a source
// can end: the loop then stops and finished() turns true.
class receiver {
public:
    typedef void (*callback_t)(std::vector<std::vector<unsigned char> > packets);
    receiver(callback_t callback, sample_source *source, int device = 0, int num_rx_samples = 4096, int async_batch_calls = 0,
             size_t device_batch_samples = 0, int narrow_threads = 0)
        : callback_(callback), source_(source), chain_(device, async_batch_calls, device_batch_samples, narrow_threads), n_(num_rx_samples), token_(true),
          stop_(false), finished_(false)
    {
        thread_ = std::thread(&receiver::receiver_chain_loop, this);
    }
    // the same over several devices (receiver_chain's device-list mode)
    receiver(callback_t callback, sample_source *source, const std::vector<int> &devices, int num_rx_samples, size_t device_batch_samples, int narrow_threads = 0)
        : callback_(callback), source_(source), chain_(devices, device_batch_samples, narrow_threads), n_(num_rx_samples), token_(true), stop_(false), finished_(false)
    {
        thread_ = std::thread(&receiver::receiver_chain_loop, this);
    }
    ~receiver()
    {
        stop_ = true;
        resume();                                   // a paused loop must be able to see stop_
        if (thread_.joinable()) thread_.join();
    }
    receiver(const receiver &) = delete;
    receiver &operator=(const receiver &) = delete;

    void pause() { take(); }                        // sem_wait(&m_pause)
    void resume() { give(); }                       // sem_post(&m_pause)
    bool finished() const { return finished_; }
    void wait_finished()
    {
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return finished_.load(); });
    }

private:
    void take()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return token_; });
        token_ = false;
    }
    void give()
    {
        { std::lock_guard<std::mutex> lk(m_); token_ = true; }
        cv_.notify_one();
    }
    void receiver_chain_loop()
    {
        std::vector<std::complex<double> > samples((size_t)n_);
        while (!stop_) {
            take();                                 // block while the receiver is paused
            const bool more = !stop_ && source_->get_samples(n_, samples);
            if (more) {
                callback_(chain_.process_samples(samples));
            } else if (!stop_) {
                // end of the source: one chunk of silence lets timing_sync settle on the frames that are already complete
                samples.assign((size_t)std::max(n_, 512), std::complex<double>(0, 0));
                callback_(chain_.process_samples(samples));
                std::vector<std::vector<unsigned char> > rest = chain_.flush();          // asynchronous chain: what is still in flight
                if (!rest.empty()) callback_(rest);
            }
            give();
            if (!more) break;
        }
        { std::lock_guard<std::mutex> lk(m_); finished_ = true; }
        done_cv_.notify_all();
    }
    callback_t callback_;
    sample_source *source_;
    receiver_chain chain_;
    int n_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    bool token_;
    std::atomic<bool> stop_, finished_;
    std::thread thread_;
};

}  // namespace fun_amd
