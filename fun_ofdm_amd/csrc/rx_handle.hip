// rx_handle.hip -- the receiver handle of include/fun_ofdm_amd.h: creation, options, the constant tables, kernel timings,
// taps and the issue probe.  The kernels of the receive path live in the other units (rx_handle.h).
#include <algorithm>
#include <cmath>
#include <complex>

#include "rx_handle.h"
#include "sync_host.h"          // make_lts_time_conj, make_preamble (table generators shared with the host pre-sync)
#include "probe_kernels.h"

using namespace foa;

namespace {
thread_local std::string g_err;
}

int foa::fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
const std::string &foa::last_error_text() { return g_err; }

// ---- host-built constant tables -------------------------------------------------------------------
void foa::build_tables(DeviceTables *t)
{
    memset(t, 0, sizeof *t);
    // rates.h:52-196
    static const int rows[kNumRates][5] = {
        { 0xD, 48, 24, 1, 0 }, { 0xE, 48, 32, 1, 1 }, { 0xF, 48, 36, 1, 2 }, { 0x5, 96, 48, 2, 0 }, { 0x6, 96, 64, 2, 1 }, { 0x7, 96, 72, 2, 2 },
        { 0x9, 192, 96, 4, 0 }, { 0xA, 192, 128, 4, 1 }, { 0xB, 192, 144, 4, 2 }, { 0x1, 288, 192, 6, 1 }, { 0x3, 288, 216, 6, 2 } };
    for (int r = 0; r < kNumRates; r++) {
        RateRow &x = t->rates[r];
        x.rate_field = rows[r][0]; x.cbps = rows[r][1]; x.dbps = rows[r][2]; x.bpsc = rows[r][3]; x.punct = rows[r][4];
        // qam.h:35-51: NumBits = bits per axis, power 1.0 for BPSK else 0.5 (modulator.cpp:117-157)
        int nb = x.bpsc == 1 ? 1 : x.bpsc / 2;
        double power = x.bpsc == 1 ? 1.0 : 0.5;
        int nn = 1 << (nb - 1), sum2 = (4 * nn * nn * nn - nn) / 3;
        double sf = std::sqrt(power * (double)nn / (double)sum2);
        x.numbits = nb;
        x.scale_d = (double)(1 << (8 - nb)) / sf;
    }
    for (int k = 0; k < 64; k++) {
        double a = -2.0 * M_PI * (double)k / 64.0;
        t->tw_re[k] = std::cos(a); t->tw_im[k] = std::sin(a);
    }
    t->tw_re[0] = 1; t->tw_im[0] = 0; t->tw_re[16] = 0; t->tw_im[16] = -1; t->tw_re[32] = -1; t->tw_im[32] = 0; t->tw_re[48] = 0; t->tw_im[48] = 1;
    // 802.11a-1999 17.3.3 long training sequence L(-26..26); preamble.h:363 stores it at index k+32
    static const signed char lts[53] = { 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                         1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1 };
    for (int i = 0; i < 53; i++) t->lts_freq[i + 6] = lts[i];
    // pilot polarity p_0..126 (17.3.5.9): scrambler sequence for the all-ones seed, 0 -> +1, 1 -> -1
    int st = 0x7F;
    for (int i = 0; i < 127; i++) {
        int fb = ((st >> 6) ^ (st >> 3)) & 1;
        st = ((st << 1) & 0x7E) | fb;
        t->polarity[i] = fb ? -1 : 1;
    }
    // phase_tracker.cpp:37-50
    int n = 0;
    for (int s = 0; s < 64; s++) {
        t->data_index[s] = -1;
        if (s < 6 || s > 58 || s == 32) t->carrier_kind[s] = 0;
        else if (s == 11 || s == 25 || s == 39 || s == 53) t->carrier_kind[s] = 2;
        else { t->carrier_kind[s] = 1; t->data_index[s] = (int8_t)n++; }
    }
    // ppdu.cpp:256-264: per-byte feedback bit of the 7-bit LFSR seeded with 93 (period 127)
    st = 93;
    for (int i = 0; i < 128; i++) {
        int fb = ((st >> 6) & 1) ^ ((st >> 3) & 1);
        t->scramble[i] = (uint8_t)fb;
        st = ((st << 1) & 0x7E) | fb;
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        t->crc_table[i] = c;
    }
    std::complex<double> ltc[64];
    foa::make_lts_time_conj(ltc);
    for (int i = 0; i < 64; i++) { t->lts_conj_re[i] = ltc[i].real(); t->lts_conj_im[i] = ltc[i].imag(); }
    std::complex<double> pre[320];
    foa::make_preamble(pre);
    for (int i = 0; i < 320; i++) { t->preamble_re[i] = pre[i].real(); t->preamble_im[i] = pre[i].imag(); }
    // qam.h:110-125 with NumBits = 3 (fewer bits = a prefix of the same loop); |pt| >= 320 gives the value of +-320
    for (int p = -320; p <= 320; p++) {
        uint32_t pt = (uint32_t)p, word = 0;
        int flip = 1, amp = 128;
        for (int i = 0; i < 3; i++) {
            int v = (int)((uint32_t)flip * pt + 128u);
            word |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * i);
            int bit = ((int)pt < 0) ? -1 : 1;
            pt -= (uint32_t)(bit * amp);
            flip = -bit;
            amp >>= 1;
        }
        t->qam_lut[p + 320] = word;
    }
    for (int r = 0; r < kNumRates; r++)
        for (int c = 0; c < t->rates[r].cbps; c++) {
            const int w = c % 48, dd = 48 * (c / 48) + 16 * (w % 3) + w / 3, punct = t->rates[r].punct;     // interleaver.h:66-75 inverse
            static const int k34[4] = { 0, 1, 3, 5 }, k23[3] = { 0, 2, 3 };
            t->sym_pos[r][c] = (uint16_t)(punct == 2 ? 6 * (dd >> 2) + k34[dd & 3] : punct == 1 ? 4 * (dd / 3) + k23[dd % 3] : dd);
        }
}

extern "C" {

int foa_version(void) { return FOA_VERSION; }
const char *foa_last_error(void) { return g_err.c_str(); }

int foa_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(FOA_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

const char *foa_rx_notes(foa_rx *rx) { return rx ? rx->notes.c_str() : ""; }

int foa_rx_create(foa_rx **out, int device)
{
    if (!out) return fail(FOA_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FOA_E_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(FOA_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(FOA_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    foa_rx *rx = new foa_rx();
    rx->device = device;
    // Six streams want six hardware queues that do not get in each other's way.  Two things decide that, both measured (tools/probe_queues.hip,
    // tools/probe_pipes.hip, profiles/r05_ab_stream_layout_host_queues.txt):
    //  * the runtime keeps a pool of GPU_MAX_HW_QUEUES hardware queues (default 4, read when it starts) PER STREAM PRIORITY and hands a new
    //    stream the least used one once the pool is full.  Streams of the normal priority share the pool with whatever the host made;
    //  * a process's hardware queues go round the GPU's four dispatch pipes in the order they were made (queue k on pipe k mod 4), and a grid
    //    that does not fit on the machine at once -- the forward pass -- keeps its pipe busy until its last workgroup has started: another
    //    queue on that pipe waits.  Two lanes, the stitch stream and the copy stream on two pipes run at 24-29 Gsample/s instead of 37,
    //    which is what round 4's layout (six normal streams) did in a host that had made two or three streams of its own.
    // So: no stream at the normal level.  The four lanes sit at the LOW level (the decode yields to the host's own normal-priority work), the
    // stitch / CRC stream and the copy / pre-sync stream at the HIGH one (short kernels that gate the next call), and they are made in the
    // order lane 1, stitch, copy, lane 2, lane 3, lane 4: four queues in a row are on four different pipes wherever the row starts.  Same
    // speed with 0 .. 3 queues made by the host before, with the runtime's default of 4 queues and with 8; small batches no longer need
    // GPU_MAX_HW_QUEUES=8 (1 000 frames per call: 9.2 -> 11.0 Gsample/s at the default).
    {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));         // (numerically greatest <= least)
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream, hipStreamNonBlocking, least));
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream2, hipStreamNonBlocking, greatest));
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream3, hipStreamNonBlocking, greatest));
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream4, hipStreamNonBlocking, least));
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream5, hipStreamNonBlocking, least));
        HIP_TRY(hipStreamCreateWithPriority(&rx->stream6, hipStreamNonBlocking, least));
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (q && atoi(q) < 4) {                                          // (also set but empty = 1) fewer queues per pool than lanes
            rx->max_depth = 2;
            rx->notes += "GPU_MAX_HW_QUEUES is set below 4: the four lanes of small decode calls would share hardware queues, so such calls keep 2 loops in "
                         "flight instead of 4; leave the variable unset (the runtime's default of 4 per priority level is enough). ";
        }
    }
    HIP_TRY(hipEventCreateWithFlags(&rx->in_ready, hipEventDisableTiming));
    for (auto &ws : rx->sets) {
        for (auto &e : ws.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ws.walk_done, hipEventDisableTiming));
    }
    DeviceTables tab;
    build_tables(&tab);
    int rc;
    if ((rc = upload_tables_decode(tab)) || (rc = upload_tables_sync(tab)) || (rc = upload_tables_stage(tab)) || (rc = upload_tables_tx(tab))) {
        foa_rx_destroy(rx);
        return rc;
    }
    *out = rx;
    return FOA_OK;
}

void foa_rx_destroy(foa_rx *rx)
{
    if (!rx) return;
    if (rx->open_stream) stream_shutdown(rx->open_stream);       // (joins the engine's threads: they use the handle; the owner still frees the shell)
    (void)hipSetDevice(rx->device);
    (void)drain(rx);
    for (auto &ws : rx->sets) {
        ws.release_all();
        for (auto &e : ws.ev) if (e) (void)hipEventDestroy(e);
        if (ws.done) (void)hipEventDestroy(ws.done);
        if (ws.walk_done) (void)hipEventDestroy(ws.walk_done);
    }
    if (rx->in_ready) (void)hipEventDestroy(rx->in_ready);
    if (rx->tx_done) (void)hipEventDestroy(rx->tx_done);
    if (rx->join) (void)hipStreamDestroy(rx->join);
    if (rx->sy_done) (void)hipEventDestroy(rx->sy_done);
    if (rx->sy_pin) (void)hipHostFree(rx->sy_pin);
    for (auto &j : rx->jobs) {
        j.dev.release();
        if (j.pin) (void)hipHostFree(j.pin);
        if (j.done) (void)hipEventDestroy(j.done);
    }
    rx->scratch.release();
    rx->sy_flags.release(); rx->sy_cnt.release(); rx->sy_off.release(); rx->sy_keep.release(); rx->sy_n.release(); rx->sy_x.release(); rx->sy_cand.release();
    if (rx->stream) (void)hipStreamDestroy(rx->stream);
    if (rx->stream2) (void)hipStreamDestroy(rx->stream2);
    if (rx->stream3) (void)hipStreamDestroy(rx->stream3);
    if (rx->stream4) (void)hipStreamDestroy(rx->stream4);
    if (rx->stream5) (void)hipStreamDestroy(rx->stream5);
    if (rx->stream6) (void)hipStreamDestroy(rx->stream6);
    delete rx;
}

int foa_rx_reserve(foa_rx *rx, size_t n_samples, size_t n_frames)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    WorkSet *keep = rx->w;
    int rc = FOA_OK;
    for (auto &ws : rx->sets) {                                        // every work set (all but the first only matter when pipelining)
        rx->w = &ws;
        if ((rc = workspace(rx, n_samples, n_frames))) break;
        if (!rx->pipeline) break;
    }
    rx->w = keep;
    return rc;
}

int foa_rx_set_option(foa_rx *rx, const char *name, int64_t value)
{
    if (!rx || !name) return fail(FOA_E_INVALID, "NULL argument");
    if (!strcmp(name, "tb_segment")) {
        if (value < 96 || value > 3072 || value % 96) return fail(FOA_E_INVALID, "tb_segment must be a multiple of 96 in [96, 3072]");
        rx->tb_segment = (int)value;
        rx->tb_segment_set = true;
        return FOA_OK;
    }
    if (!strcmp(name, "tb_overlap")) {
        if (value < 0 || value > 3072 || value % 96) return fail(FOA_E_INVALID, "tb_overlap must be a multiple of 96 in [0, 3072]");
        rx->tb_overlap = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "max_dbps")) {
        if (value < 24 || value > 216) return fail(FOA_E_INVALID, "max_dbps must lie in [24, 216] (data bits per OFDM symbol of the highest rate: rates.h:52-196)");
        rx->max_dbps = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "pipeline")) { int rc0 = drain(rx); if (rc0) return rc0; rx->pipeline = value != 0; return FOA_OK; }
    if (!strcmp(name, "record_eq")) { rx->record_eq = value != 0; return FOA_OK; }
    if (!strcmp(name, "record_soft")) { rx->record_soft = value != 0; return FOA_OK; }
    if (!strcmp(name, "depth")) {
        if (value != 0 && (value < 2 || value > 4)) return fail(FOA_E_INVALID, "depth must be 0 (by grid size), 2, 3 or 4");
        rx->depth = (int)value;
        if (value > rx->max_depth && rx->notes.find("option depth") == std::string::npos)
            rx->notes += "option depth exceeds what the hardware queues the runtime started with can run side by side: lanes will share queues. ";
        return FOA_OK;
    }
    if (!strcmp(name, "sync_call")) {
        if (value != 0 && value <= 160) return fail(FOA_E_INVALID, "sync_call must be 0 (decide as one call over the whole stream) or > 160 (timing_sync.cpp:55)");
        if (rx->open_stream) return fail(FOA_E_STATE, "sync_call cannot change while a stream engine is open on the handle (its submitter thread reads it)");
        rx->sync_call = value;
        return FOA_OK;
    }
    if (!strcmp(name, "stream_longest")) {
        if (value != 0 && (value < 1024 || value > 110592)) return fail(FOA_E_INVALID, "stream_longest must be 0 (any frame: 110 592 samples) or lie in [1024, 110592]");
        if (rx->open_stream) return fail(FOA_E_STATE, "stream_longest is read when a stream is created");
        rx->stream_longest = value;
        return FOA_OK;
    }
    if (!strcmp(name, "sync_origin")) {
        if (value < 0) return fail(FOA_E_INVALID, "sync_origin is a stream index (>= 0)");
        if (rx->open_stream) return fail(FOA_E_STATE, "sync_origin cannot change while a stream engine is open on the handle");
        rx->sync_origin = value;
        return FOA_OK;
    }
    return fail(FOA_E_INVALID, "unknown option '%s'", name);
}

void *foa_rx_stream(foa_rx *rx) { return rx ? (void *)rx->stream : nullptr; }

int foa_rx_sync(foa_rx *rx)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    return drain(rx);
}

// ---- device-side ordering against the caller's own streams --------------------------------------------------------------------------
// The library's streams are its own (several, at other priorities than the host's): what orders them against a caller's stream is an
// event.  Nothing here makes the host wait.

int foa_rx_after(foa_rx *rx, void *event)
{
    if (!rx || !event) return fail(FOA_E_INVALID, "NULL argument");
    if (rx->open_stream) return fail(FOA_E_STATE, "a stream engine owns this handle");
    if (rx->after.size() >= 16) return fail(FOA_E_STATE, "16 events are already waiting for the next call");
    rx->after.push_back((hipEvent_t)event);
    return FOA_OK;
}

// the stream that joins the library's streams for a caller's event: made when first asked for (a handle that never orders against
// a caller's stream keeps the queue layout of foa_rx_create), at the high priority level (it only ever carries waits and one record)
static int join_stream(foa_rx *rx)
{
    if (!rx->join) HIP_TRY(hipStreamCreateWithPriority(&rx->join, hipStreamNonBlocking, high_priority()));
    return FOA_OK;
}

// Everything queued so far has READ what it reads of the caller's buffers (samples, descriptors, ends; payloads and frames of the transmit
// calls) once `event` completes: the front ends of the decode calls in flight -- their data-symbol kernel is the last reader -- the pre-sync
// and the transmit kernels.  Cheap: nothing is flushed, the deferred chain-back of a pipelined call stays deferred.
int foa_rx_record_consumed(foa_rx *rx, void *event)
{
    if (!rx || !event) return fail(FOA_E_INVALID, "NULL argument");
    if (rx->open_stream) return fail(FOA_E_STATE, "a stream engine owns this handle");
    HIP_TRY(enter_device(rx->device));
    if (!rx->pipeline) {                                               // calls in line: everything is on the handle's one stream
        HIP_TRY(hipEventRecord((hipEvent_t)event, rx->stream));
        return FOA_OK;
    }
    { int rc = join_stream(rx); if (rc) return rc; }
    WorkSet *w = rx->w;
    for (int i = 0; i < kSets - 1 && w && w->used; i++, w = w->before) HIP_TRY(hipStreamWaitEvent(rx->join, w->ev[3], 0));      // (one per lane and more: complete ones cost nothing)
    if (rx->sy_used) HIP_TRY(hipStreamWaitEvent(rx->join, rx->sy_done, 0));
    if (rx->tx_used) HIP_TRY(hipStreamWaitEvent(rx->join, rx->tx_done, 0));
    HIP_TRY(hipEventRecord((hipEvent_t)event, rx->join));
    return FOA_OK;
}

// Everything queued so far is COMPLETE once `event` completes: PSDU slots and results of every decode call, descriptors of the pre-sync,
// frames and samples of the transmit calls.  A pipelined call's chain-back and finish, which would otherwise wait for the next call's
// front end to be queued (rx_handle.h), are queued now.
int foa_rx_record_done(foa_rx *rx, void *event)
{
    if (!rx || !event) return fail(FOA_E_INVALID, "NULL argument");
    if (rx->open_stream) return fail(FOA_E_STATE, "a stream engine owns this handle");
    HIP_TRY(enter_device(rx->device));
    { int rc = flush_pending(rx, nullptr); if (rc) return rc; }
    if (!rx->pipeline) {
        HIP_TRY(hipEventRecord((hipEvent_t)event, rx->stream));
        return FOA_OK;
    }
    { int rc = join_stream(rx); if (rc) return rc; }
    WorkSet *w = rx->w;
    for (int i = 0; i < kSets - 1 && w && w->used; i++, w = w->before) HIP_TRY(hipStreamWaitEvent(rx->join, w->done, 0));
    if (rx->sy_used) HIP_TRY(hipStreamWaitEvent(rx->join, rx->sy_done, 0));
    if (rx->tx_used) HIP_TRY(hipStreamWaitEvent(rx->join, rx->tx_done, 0));
    HIP_TRY(hipStreamWaitEvent(rx->join, rx->in_ready, 0));            // (copies of a host-pointer entry point, if one was ever queued)
    HIP_TRY(hipEventRecord((hipEvent_t)event, rx->join));
    return FOA_OK;
}

int foa_rx_decode_frames_dev_after(foa_rx *rx, void *inputs_ready, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                                   size_t n_frames, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    if (inputs_ready) { int rc = foa_rx_after(rx, inputs_ready); if (rc) return rc; }
    const int rc = foa_rx_decode_frames_dev(rx, d_iq, n_samples, d_descs, d_ends, n_frames, d_psdu, slot_bytes, d_results);
    if (rx) rx->after.clear();                                         // (a call refused before it queued anything leaves nothing behind)
    return rc;
}

int foa_rx_sync_dev_begin_after(foa_rx *rx, void *inputs_ready, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap)
{
    if (inputs_ready) { int rc = foa_rx_after(rx, inputs_ready); if (rc) return rc; }
    const int rc = foa_rx_sync_dev_begin(rx, d_iq, n_samples, d_descs, d_ends, cap);
    if (rx) rx->after.clear();
    return rc;
}

int foa_rx_wait_age(foa_rx *rx, int age)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (age < 0 || age > 4) return fail(FOA_E_INVALID, "age must lie in 0 .. 4");
    WorkSet *w = rx->w;
    if (age == 1 && !rx->pipeline) w = rx->prev;                        // calls in line: the same set again
    else for (int i = 0; i < age && w; i++) w = w->before;
    if (!w || !w->used || (age > 0 && w == rx->w)) return FOA_OK;       // no such call: nothing to wait for
    if (rx->pending.valid && rx->pending.w == w) { int rc = flush_pending(rx, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventSynchronize(w->done));
    return FOA_OK;
}

int foa_rx_wait_previous(foa_rx *rx) { return foa_rx_wait_age(rx, 1); }

static int kernel_ms_of(foa_rx *rx, WorkSet *w, float out_ms[6])
{
    if (!w || !w->have_timing) return fail(FOA_E_STATE, "no such decode call has been made on this handle");
    if (rx->pending.valid && rx->pending.w == w) { int rc = flush_pending(rx, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventSynchronize(w->ev[4]));
    HIP_TRY(hipEventSynchronize(w->ev[5]));
    for (int i = 0; i < 3; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], w->ev[i], w->ev[i + 1]));
    // forward pass.  Pipelined, it has its own start event: consecutive forward passes overlap by design (two streams), so this is
    // the launch's own duration, like a kernel trace reports it, not the step's share.
    if (w->piped) HIP_TRY(hipEventElapsedTime(&out_ms[3], w->ev[7], w->ev[5]));
    else HIP_TRY(hipEventElapsedTime(&out_ms[3], w->ev[3], w->ev[5]));
    // chain-back + descramble + CRC; on the pipelined path from where the walk is queued behind its forward pass
    if (w->piped) HIP_TRY(hipEventElapsedTime(&out_ms[4], w->ev[6], w->ev[4]));
    else HIP_TRY(hipEventElapsedTime(&out_ms[4], w->ev[5], w->ev[4]));
    HIP_TRY(hipEventElapsedTime(&out_ms[5], w->ev[0], w->ev[4]));      // whole call, first kernel to last (includes the deferral)
    return FOA_OK;
}

int foa_rx_last_kernel_ms(foa_rx *rx, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    return kernel_ms_of(rx, rx->w, out_ms);
}

int foa_rx_prev_kernel_ms(foa_rx *rx, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    return kernel_ms_of(rx, rx->prev, out_ms);
}

int foa_rx_kernel_ms_age(foa_rx *rx, int age, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    if (age < 0 || age > 4) return fail(FOA_E_INVALID, "age must lie in 0 .. 4");
    WorkSet *w = rx->w;
    for (int i = 0; i < age && w; i++) w = w->before;                // the pipelined calls link their work sets
    if (age > 0 && (!w || w == rx->w)) return fail(FOA_E_STATE, "no decode call of that age (calls must be pipelined)");
    return kernel_ms_of(rx, w, out_ms);
}

int foa_rx_forward_spacing_ms(foa_rx *rx, int age, float out[3])
{
    if (!rx || !out) return fail(FOA_E_INVALID, "NULL argument");
    if (age < 1 || age > 3) return fail(FOA_E_INVALID, "age must lie in 1 .. 3");
    WorkSet *w = rx->w;
    for (int i = 0; i < age && w; i++) w = w->before;
    WorkSet *b = w ? w->before : nullptr;
    if (!w || !b || !w->piped || !b->piped || w == rx->w) return fail(FOA_E_STATE, "no two pipelined decode calls of that age");
    HIP_TRY(hipEventSynchronize(w->ev[5]));
    HIP_TRY(hipEventSynchronize(b->ev[5]));                              // (the call before ran on ANOTHER lane: nothing orders its pass before w's)
    HIP_TRY(hipEventElapsedTime(&out[0], b->ev[7], w->ev[7]));          // start to start
    HIP_TRY(hipEventElapsedTime(&out[1], w->ev[7], b->ev[5]));          // > 0: the earlier pass was still running when this one started
    HIP_TRY(hipEventElapsedTime(&out[2], w->ev[7], w->ev[5]));          // this pass's own duration
    return FOA_OK;
}

int foa_rx_probe_issue(foa_rx *rx, double out[6])
{
    if (!rx || !out) return fail(FOA_E_INVALID, "NULL argument");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, rx->device));
    const int n_simd = prop.multiProcessorCount * 4, W = 8, nw = n_simd * W, window_k = 600;      // ~0.26 ms per launch
    DevBuf<unsigned long long> buf;
    int rc = buf.ensure((size_t)3 * nw);
    if (rc) return rc;
    std::vector<unsigned long long> h((size_t)3 * nw);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; kind++) {
        float ms = 0;
        for (int it = 0; it < 3; it++) {                       // the first launches bring the clock to where a busy chip holds it
            HIP_TRY(hipEventRecord(e0, rx->stream));
            if (kind == 0) hipLaunchKernelGGL(k_probe_issue<0>, dim3(nw / 4), dim3(256), 0, rx->stream, buf.p, 7u, window_k);
            else hipLaunchKernelGGL(k_probe_issue<1>, dim3(nw / 4), dim3(256), 0, rx->stream, buf.p, 7u, window_k);
            HIP_TRY(hipEventRecord(e1, rx->stream));
            HIP_TRY(hipStreamSynchronize(rx->stream));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        }
        HIP_TRY(hipMemcpy(h.data(), buf.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double ticks = 0, instr = 0;
        for (int i = 0; i < nw; i++) { ticks += (double)h[3 * i]; instr += (double)h[3 * i + 1] * 64.0; }
        const double window = ticks / nw;                      // shader clocks every wave was issuing for
        out[3 * kind + 0] = window / (instr / n_simd);         // SIMD clocks per wave64 instruction
        out[3 * kind + 1] = window / (ms * 1e6);               // GHz sustained during the launch
        out[3 * kind + 2] = instr / (ms * 1e-3);               // wave-instructions per second, whole chip
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    buf.release();
    return FOA_OK;
}

// Host-to-device rate the way the stream engines move samples (stream_engine.h): hipMemcpyAsync out of page-locked hipHostMalloc staging
// on the library's copy stream, `in_flight` pieces of `piece_bytes` queued at once, `rounds` times -- what 8 bytes per sample of a
// process_samples() capture can reach on this host, whatever the kernels do.
int foa_rx_probe_h2d(foa_rx *rx, size_t piece_bytes, int in_flight, int rounds, double *gbytes_per_s)
{
    if (!rx || !gbytes_per_s || piece_bytes == 0 || in_flight < 1 || in_flight > 16 || rounds < 1) return fail(FOA_E_INVALID, "bad argument");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    uint8_t *host = nullptr;
    DevBuf<uint8_t> dev;
    int rc = dev.ensure(piece_bytes * (size_t)in_flight);
    if (rc) return rc;
    HIP_TRY(hipHostMalloc((void **)&host, piece_bytes * (size_t)in_flight, hipHostMallocDefault));
    memset(host, 1, piece_bytes * (size_t)in_flight);
    hipStream_t st = rx->stream3;
    auto pass = [&]() -> int {
        for (int i = 0; i < in_flight; i++) HIP_TRY(hipMemcpyAsync(dev.p + (size_t)i * piece_bytes, host + (size_t)i * piece_bytes, piece_bytes, hipMemcpyHostToDevice, st));
        return FOA_OK;
    };
    if ((rc = pass())) { (void)hipHostFree(host); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < rounds && !rc; r++) rc = pass();
    if (!rc) { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) rc = fail(FOA_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipHostFree(host);
    dev.release();
    if (rc) return rc;
    *gbytes_per_s = (double)piece_bytes * in_flight * rounds / s / 1e9;
    return FOA_OK;
}

int foa_rx_get_taps(foa_rx *rx, size_t n_frames, double *hinv, double *eq, size_t eq_cap, uint64_t *eq_off, uint8_t *soft, size_t soft_cap,
                    uint64_t *soft_off)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames != rx->last_frames || n_frames == 0) return fail(FOA_E_STATE, "n_frames does not match the last decode call");
    if (eq && !rx->record_eq) return fail(FOA_E_STATE, "set option record_eq=1 before the decode call to get eq");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    std::vector<FrameInfo> info(n_frames);
    HIP_TRY(hipMemcpy(info.data(), rx->w->info.p, n_frames * sizeof(FrameInfo), hipMemcpyDeviceToHost));
    if (hinv) HIP_TRY(hipMemcpy(hinv, rx->w->hinv.p, n_frames * 64 * sizeof(double2), hipMemcpyDeviceToHost));
    size_t eo = 0, so = 0;
    for (size_t f = 0; f < n_frames; f++) {
        const FrameInfo &fi = info[f];
        if (eq_off) eq_off[f] = eo;
        if (soft_off) soft_off[f] = so;
        if (fi.rate < 0) continue;
        const int nsym = fi.nsym > 0 ? fi.nsym : 0;
        if (eq) {
            if (eo + (size_t)(1 + nsym) * 48 > eq_cap) return fail(FOA_E_INVALID, "eq_cap too small");
            HIP_TRY(hipMemcpy(eq + 2 * eo, rx->w->eq_sig.p + f * 48, 48 * sizeof(double2), hipMemcpyDeviceToHost));
            if (nsym) HIP_TRY(hipMemcpy(eq + 2 * (eo + 48), rx->w->eq_data.p + (size_t)fi.sym_off * 48, (size_t)nsym * 48 * sizeof(double2), hipMemcpyDeviceToHost));
        }
        eo += (size_t)(1 + nsym) * 48;
        const size_t sb = nsym ? (size_t)2 * fi.nsteps : 0;
        if (soft && sb) {
            if (so + sb > soft_cap) return fail(FOA_E_INVALID, "soft_cap too small");
            HIP_TRY(hipMemcpy(soft + so, rx->w->sp.p + fi.dec_off, sb, hipMemcpyDeviceToHost));
        }
        so += sb;
    }
    if (eq_off) eq_off[n_frames] = eo;
    if (soft_off) soft_off[n_frames] = so;
    return FOA_OK;
}

int foa_rx_get_decisions(foa_rx *rx, size_t frame, uint64_t *out, size_t cap, size_t *n_steps)
{
    if (!rx || !out || !n_steps) return fail(FOA_E_INVALID, "NULL argument");
    if (frame >= rx->last_frames) return fail(FOA_E_STATE, "frame index beyond the last decode call");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    FrameInfo fi;
    HIP_TRY(hipMemcpy(&fi, rx->w->info.p + frame, sizeof fi, hipMemcpyDeviceToHost));
    const size_t n = fi.nsym > 0 ? (size_t)fi.nsteps : 0;
    *n_steps = n;
    if (n > cap) return fail(FOA_E_INVALID, "cap too small (%zu steps)", n);
    if (n) HIP_TRY(hipMemcpy(out, rx->w->dec.p + fi.dec_off, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FOA_OK;
}

}  // extern "C"
