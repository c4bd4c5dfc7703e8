// stub_abi.cpp -- a stand-in for libfun_ofdm_amd.so WITHOUT a GPU, for CPU sanitizer runs of the host-side code
// (tools/run_sanitizers.sh): include/fun_ofdm_amd/blocks.hpp (receiver_chain, receiver, sources) and
// fun_ofdm_amd/csrc/sync_host.h (the streaming pre-sync behind foa_sync_*).
// TEST INFRASTRUCTURE: the decode entry points are answered by the oracle (oracle/fo_oracle.c); nothing in the product
// links this file.  The foa_sync_* functions are the real host code (SyncHost), wrapped exactly as csrc/rx_sync.hip wraps it.
#include <cmath>
#include <complex>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../fun_ofdm_amd/csrc/sync_host.h"
extern "C" {
#include "fo_oracle.h"
}

static_assert(sizeof(foa_frame_desc) == sizeof(fo_frame_desc) && sizeof(foa_frame_result) == sizeof(fo_frame_result), "oracle and ABI structs differ");

struct foa_rx {
    uint64_t next_ticket = 1;
    struct job { std::vector<uint8_t> psdu; std::vector<foa_frame_result> res; };
    std::map<uint64_t, job> jobs;
};
struct foa_sync {
    foa::SyncHost impl;
    std::vector<foa_frame_desc> pending;
};

template <typename T>
static int sync_push(foa_sync *s, const T *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out)
{
    if (!s || (n && !iq) || (cap && !out) || !n_out) return FOA_E_INVALID;
    s->impl.push(iq, n, s->pending);
    size_t k = s->pending.size() < cap ? s->pending.size() : cap;
    if (k) memcpy(out, s->pending.data(), k * sizeof(foa_frame_desc));
    s->pending.erase(s->pending.begin(), s->pending.begin() + k);
    *n_out = k;
    return FOA_OK;
}

extern "C" {

int foa_version(void) { return FOA_VERSION; }
const char *foa_last_error(void) { return "stub"; }
int foa_rx_create(foa_rx **out, int) { *out = new foa_rx(); return FOA_OK; }
void foa_rx_destroy(foa_rx *rx) { delete rx; }

int foa_rx_decode_frames_host(foa_rx *, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                              uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    if (n_frames == 0) return FOA_OK;
    fo_decode_batch_f32(iq, (int64_t)n_samples, (const fo_frame_desc *)descs, ends, n_frames, psdu, slot_bytes, (fo_frame_result *)results, 2);
    return FOA_OK;
}

int foa_rx_submit_host_ctx(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                           size_t n_context, size_t slot_bytes, uint64_t *ticket)
{
    if (rx->jobs.size() >= 8) return FOA_E_STATE;
    foa_rx::job j;
    j.psdu.assign(n_frames * slot_bytes, 0);
    j.res.resize(n_frames);
    if (n_frames)
        fo_decode_batch_v2_f32(iq, (int64_t)n_samples, (const fo_frame_desc *)descs, ends, n_frames, n_context, j.psdu.data(), slot_bytes, (fo_frame_result *)j.res.data());
    *ticket = rx->next_ticket++;
    rx->jobs[*ticket] = std::move(j);
    return FOA_OK;
}
int foa_rx_submit_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                       size_t slot_bytes, uint64_t *ticket)
{
    return foa_rx_submit_host_ctx(rx, iq, n_samples, descs, ends, n_frames, 0, slot_bytes, ticket);
}

int foa_rx_collect(foa_rx *rx, uint64_t ticket, int, uint8_t *psdu, foa_frame_result *results)
{
    auto it = rx->jobs.find(ticket);
    if (it == rx->jobs.end()) return FOA_E_INVALID;
    memcpy(psdu, it->second.psdu.data(), it->second.psdu.size());
    memcpy(results, it->second.res.data(), it->second.res.size() * sizeof(foa_frame_result));
    rx->jobs.erase(it);
    return 1;
}

int foa_sync_create(foa_sync **out) { *out = new foa_sync(); return FOA_OK; }
void foa_sync_destroy(foa_sync *s) { delete s; }
int foa_sync_push_f32(foa_sync *s, const float *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out) { return sync_push(s, iq, n, out, cap, n_out); }
int foa_sync_push_f64(foa_sync *s, const double *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out) { return sync_push(s, iq, n, out, cap, n_out); }
int64_t foa_sync_settled(const foa_sync *s) { return s ? s->impl.settled() : 0; }
int foa_sync_set_call(foa_sync *s, int64_t call) { if (!s) return FOA_E_INVALID; s->impl.set_call(call); return FOA_OK; }
int foa_rx_set_option(foa_rx *, const char *, int64_t) { return FOA_OK; }

// foa_stream_*: the wrapper logic of fun_amd::receiver_chain's device mode is what runs here; the engine itself (batches on the
// GPU) is replaced by "collect everything, pre-sync on the host and decode with the oracle at the flush"
struct foa_stream {
    std::vector<float> iq;
    std::vector<std::vector<uint8_t> > out;
    bool flushed = false, delivered = false;
};
int foa_stream_create(foa_rx *, size_t batch, int, foa_stream **out) { if (batch < 4096) return FOA_E_INVALID; *out = new foa_stream(); return FOA_OK; }
void foa_stream_destroy(foa_stream *s) { delete s; }
int foa_stream_push_f32(foa_stream *s, const float *iq, size_t n) { if (s->flushed) return FOA_E_STATE; s->iq.insert(s->iq.end(), iq, iq + 2 * n); return FOA_OK; }
int foa_stream_push_f64(foa_stream *s, const double *iq, size_t n)
{
    if (s->flushed) return FOA_E_STATE;
    for (size_t i = 0; i < 2 * n; i++) s->iq.push_back((float)iq[i]);
    return FOA_OK;
}
int foa_stream_push_f64_owned(foa_stream *s, const double *iq, size_t n, void (*release)(void *), void *ctx)
{
    const int rc = foa_stream_push_f64(s, iq, n);
    release(ctx);
    return rc;
}
int foa_stream_flush(foa_stream *s)
{
    if (s->flushed) return FOA_OK;
    s->flushed = true;
    const size_t n = s->iq.size() / 2;
    std::vector<foa_frame_desc> d(n / 300 + 64);
    const size_t m = fo_find_alignments_f32(s->iq.data(), (int64_t)n, (fo_frame_desc *)d.data(), d.size());
    std::vector<int64_t> ends(m);
    for (size_t i = 0; i < m; i++) ends[i] = i + 1 < m ? d[i + 1].lts1_pos : (int64_t)n;
    std::vector<uint8_t> psdu(m * 4096 + 1);
    std::vector<foa_frame_result> res(m + 1);
    if (m) fo_decode_batch_f32(s->iq.data(), (int64_t)n, (const fo_frame_desc *)d.data(), ends.data(), m, psdu.data(), 4096, (fo_frame_result *)res.data(), 2);
    for (size_t i = 0; i < m; i++)
        if (res[i].status == FOA_ST_OK) s->out.push_back(std::vector<uint8_t>(psdu.begin() + i * 4096, psdu.begin() + i * 4096 + res[i].length));
    return FOA_OK;
}
int foa_stream_ready(foa_stream *s, int, size_t *n_payloads, size_t *n_bytes)
{
    *n_payloads = 0; *n_bytes = 0;
    if (!s->flushed || s->delivered) return 0;
    *n_payloads = s->out.size();
    for (auto &p : s->out) *n_bytes += p.size();
    return 1;
}
int foa_stream_take(foa_stream *s, uint8_t *payloads, uint32_t *lengths)
{
    if (!s->flushed || s->delivered) return FOA_E_STATE;
    size_t o = 0;
    for (size_t i = 0; i < s->out.size(); i++) { if (!s->out[i].empty()) memcpy(payloads + o, s->out[i].data(), s->out[i].size()); lengths[i] = (uint32_t)s->out[i].size(); o += s->out[i].size(); }
    s->delivered = true;
    return FOA_OK;
}
int foa_stream_stats(const foa_stream *, uint64_t out[8]) { memset(out, 0, 8 * sizeof(uint64_t)); return FOA_OK; }

// foa_shard_*: the same stand-in behind the multi-device entry points (what is exercised here is blocks.hpp's device-list mode)
struct foa_shard { foa_stream st; int n; };
int foa_shard_create(const int *devices, int n, size_t batch, int, foa_shard **out) { if (!devices || n < 1 || batch < 4096) return FOA_E_INVALID; *out = new foa_shard(); (*out)->n = n; return FOA_OK; }
void foa_shard_destroy(foa_shard *s) { delete s; }
int foa_shard_devices(const foa_shard *s) { return s->n; }
int foa_shard_push_f32(foa_shard *s, const float *iq, size_t n) { return foa_stream_push_f32(&s->st, iq, n); }
int foa_shard_push_f64(foa_shard *s, const double *iq, size_t n) { return foa_stream_push_f64(&s->st, iq, n); }
int foa_shard_push_f64_owned(foa_shard *s, const double *iq, size_t n, void (*release)(void *), void *ctx) { return foa_stream_push_f64_owned(&s->st, iq, n, release, ctx); }
int foa_shard_flush(foa_shard *s) { return foa_stream_flush(&s->st); }
int foa_shard_ready(foa_shard *s, int w, size_t *np, size_t *nb) { return foa_stream_ready(&s->st, w, np, nb); }
int foa_shard_take(foa_shard *s, uint8_t *p, uint32_t *l) { return foa_stream_take(&s->st, p, l); }
int foa_shard_stats(const foa_shard *, uint64_t out[8], uint64_t *, int) { memset(out, 0, 8 * sizeof(uint64_t)); return FOA_OK; }
const char *foa_rx_notes(foa_rx *) { return ""; }

// the per-stage entry points, answered by the oracle's functions of the same stage (the adaptors' HOST logic -- tags, counters, frames in
// progress -- is what a CPU run of them exercises)
typedef std::complex<double> cd;
int foa_fft_forward_f64(foa_rx *, double *v, size_t n)
{
    for (size_t i = 0; i < n; i++) fo_fft64((fo_c64 *)(v + 128 * i));
    return FOA_OK;
}
int foa_channel_estimate_f64(foa_rx *, const double *pairs, double *hinv, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fo_channel_est *ce = fo_channel_est_new();
        fo_tagged_vec64 v[2], out[2];
        for (int w = 0; w < 2; w++) { memcpy(v[w].samples, pairs + (2 * i + w) * 128, sizeof v[w].samples); v[w].tag = w == 0 ? FO_LTS_START : FO_NONE; v[w]._pad = 0; }
        fo_channel_est_work(ce, v, 2, out);
        memcpy(hinv + i * 128, fo_channel_est_state(ce), 64 * sizeof(fo_c64));
        fo_channel_est_free(ce);
    }
    return FOA_OK;
}
int foa_equalize_f64(foa_rx *, double *v, size_t n, const double *hinv, size_t n_hinv, const int32_t *idx)
{
    for (size_t i = 0; i < n; i++) {
        if (idx[i] < 0 || (size_t)idx[i] >= n_hinv) return FOA_E_INVALID;
        cd *x = (cd *)(v + 128 * i);
        const cd *h = (const cd *)(hinv + 128 * (size_t)idx[i]);
        for (int j = 0; j < 64; j++) x[j] = h[j] * x[j];                     // channel_est.cpp:77-81
    }
    return FOA_OK;
}
int foa_phase_track_f64(foa_rx *, const double *v, const int32_t *count, size_t n, double *out48)
{
    const double *pol = fo_polarity();
    const int *didx = fo_data_subcarriers(), *pidx = fo_pilot_subcarriers();
    static const double sgn[4] = { 1.0, 1.0, 1.0, -1.0 };
    for (size_t i = 0; i < n; i++) {                                         // phase_tracker.cpp:83-99
        const cd *x = (const cd *)(v + 128 * i);
        cd pe(0, 0);
        for (int p = 0; p < 4; p++) {
            const int pilot = (int)(sgn[p] * pol[count[i] % 127]);
            pe += (x[pidx[p]] * std::conj(cd((double)pilot, 0.0))) / 4.0;
        }
        const double angle = std::atan2(pe.imag(), pe.real());
        const cd rot(std::cos(-angle), std::sin(-angle));
        cd *o = (cd *)(out48 + 96 * i);
        for (int s2 = 0; s2 < 48; s2++) o[s2] = x[didx[s2]] * rot;
    }
    return FOA_OK;
}
int foa_decode_header_f64(foa_rx *, const double *c48, size_t n, foa_frame_result *res)
{
    for (size_t i = 0; i < n; i++) {
        int rate = -1, length = 0, nsym = 0;
        const int ok = fo_decode_header((const fo_c64 *)(c48 + 96 * i), &rate, &length, &nsym);
        res[i].status = ok ? FOA_ST_OK : FOA_ST_HEADER_FAIL; res[i].rate = ok ? rate : -1; res[i].length = ok ? length : 0; res[i].num_symbols = ok ? nsym : 0;
    }
    return FOA_OK;
}
int foa_decode_data_f64(foa_rx *, const double *car, const uint64_t *off, size_t n, foa_frame_result *res, uint8_t *psdu, size_t slot)
{
    for (size_t i = 0; i < n; i++) {
        std::vector<uint8_t> pay(4096);
        const int ok = fo_decode_data((const fo_c64 *)(car + 2 * off[i]), res[i].rate, res[i].length, pay.data(), nullptr, nullptr);
        res[i].status = ok ? FOA_ST_OK : FOA_ST_CRC_FAIL;
        if (ok && (size_t)res[i].length <= slot) memcpy(psdu + i * slot, pay.data(), (size_t)res[i].length);
    }
    return FOA_OK;
}
int foa_rx_decode_frames_f64_host(foa_rx *, const double *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                                  uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    if (n_frames == 0) return FOA_OK;
    fo_decode_batch_v2_f64(iq, (int64_t)n_samples, (const fo_frame_desc *)descs, ends, n_frames, 0, psdu, slot_bytes, (fo_frame_result *)results);
    return FOA_OK;
}

}  // extern "C"
