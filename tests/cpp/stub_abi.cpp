// stub_abi.cpp -- a stand-in for libfun_ofdm_amd.so WITHOUT a GPU, for CPU sanitizer runs of the host-side code
// (tools/run_sanitizers.sh): include/fun_ofdm_amd/blocks.hpp (receiver_chain, receiver, sources) and
// fun_ofdm_amd/csrc/sync_host.h (the streaming pre-sync behind foa_sync_*).
// TEST INFRASTRUCTURE: the decode entry points are answered by the oracle (oracle/fo_oracle.c); nothing in the product
// links this file.  The foa_sync_* functions are the real host code (SyncHost), wrapped exactly as foa_rx.hip wraps it.
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

int foa_rx_submit_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                       size_t slot_bytes, uint64_t *ticket)
{
    if (rx->jobs.size() >= 8) return FOA_E_STATE;
    foa_rx::job j;
    j.psdu.assign(n_frames * slot_bytes, 0);
    j.res.resize(n_frames);
    foa_rx_decode_frames_host(rx, iq, n_samples, descs, ends, n_frames, j.psdu.data(), slot_bytes, j.res.data());
    *ticket = rx->next_ticket++;
    rx->jobs[*ticket] = std::move(j);
    return FOA_OK;
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

// the per-block adaptors are not part of this run; their entry points only have to link
int foa_fft_forward_f64(foa_rx *, double *, size_t) { return FOA_E_NO_DEVICE; }
int foa_channel_estimate_f64(foa_rx *, const double *, double *, size_t) { return FOA_E_NO_DEVICE; }
int foa_equalize_f64(foa_rx *, double *, size_t, const double *, size_t, const int32_t *) { return FOA_E_NO_DEVICE; }
int foa_phase_track_f64(foa_rx *, const double *, const int32_t *, size_t, double *) { return FOA_E_NO_DEVICE; }
int foa_decode_header_f64(foa_rx *, const double *, size_t, foa_frame_result *) { return FOA_E_NO_DEVICE; }
int foa_decode_data_f64(foa_rx *, const double *, const uint64_t *, size_t, foa_frame_result *, uint8_t *, size_t) { return FOA_E_NO_DEVICE; }

}  // extern "C"
