// frontend_kernels.h -- the first two stages of a decode call, for gfx950 (wave64):
//   k_header      per alignment: the two LTS windows -> channel estimate (channel_est.cpp:44-58), the SIGNAL window -> equalised,
//                 derotated, demapped, decoded (fft_symbols.cpp:33-79 + fft.cpp:50-59, channel_est.cpp:77-81, phase_tracker.cpp:70-104,
//                 ppdu.cpp:168-218) -> one FrameInfo record.  One wave per alignment, lane = FFT point.
//   k_scan_*      exclusive scans over the records: where each frame's symbols, soft pairs, decisions and chain-back segments live
//                 in the call's work set, and the symbol -> frame and segment -> frame maps.
// The data symbols themselves are frontend_q4.h.
#pragma once

#include "signal_decode.h"

namespace foa {

// =================================================================================================
// K1: per alignment: LTS1 + LTS2 + SIGNAL.  Channel estimate -> hinv, SIGNAL decode -> FrameInfo.
// One wave (64 threads) per block, one block per alignment; n_total = the call's alignments, context included.
// What the alignment's extent gives fft_symbols to emit (fft_symbols.cpp:41-73) is worked out here too: K complete symbol windows
// from SIGNAL on, and the partly filled vector it pushes when the NEXT alignment's LTS1 arrives past a window's cyclic prefix.
// =================================================================================================
// S: float2 (the raw stream; the descriptor's phasors rotate it) or double2 (a stream timing_sync has rotated already: phasors 1).
// Alignment f of [0, n_total) by one wave; lds / dem / decs: the block's LDS.
template <typename S>
__device__ __forceinline__ void header_alignment(const S *__restrict__ iq, const foa_frame_desc *__restrict__ descs, const int64_t *__restrict__ ends,
                                                 int64_t n_samples, int f, int n_lead, int n_total, FrameInfo *__restrict__ info, double2 *__restrict__ hinv,
                                                 double2 *__restrict__ eq_tap, cpx *lds, uint8_t *dem, uint64_t *decs)
{
    const int lane = threadIdx.x;
    const foa_frame_desc d = descs[f];
    const int64_t e_raw = ends[f], end = min(e_raw, n_samples), p = d.lts1_pos;      // nothing is read beyond the stream, whatever the caller's ends say
    // linked: the stream goes on into the next alignment of the call (its LTS1 tag is where this one's samples end)
    const bool link = f + 1 < n_total && e_raw <= n_samples && e_raw == descs[f + 1].lts1_pos;
    // late: an LTS1 less than 64 samples behind an earlier one of the same stream.  That alignment's LTS2 tag (timing_sync.cpp:105-106) falls
    // inside this one's first LTS window and restarts the vector there (fft_symbols.cpp:53-56), this alignment's own LTS2 tag restarts it
    // again, so the first complete vector -- still tagged LTS_START -- is the window at p + 64; the one 80 samples on is taken as the second
    // LTS vector (channel_est.cpp:44-58) and START_OF_FRAME goes to the window behind that: the grid of an alignment at q = p + 80 whose first
    // LTS window sits 16 samples early.  (descs / ends may be looked at down to index -n_lead: the stream engines' earlier tags.)
    bool late = false;
    for (int i = f - 1; i >= -n_lead; i--) {
        const int64_t ei = ends[i], dp = p - descs[i].lts1_pos;
        if (ei != descs[i + 1].lts1_pos || ei > n_samples || dp < 0 || dp >= 64) break;
        if (dp > 0) { late = true; break; }
    }
    const int64_t q = late ? p + 80 : p;
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0; fi.dec_off = 0; fi.seg_off = 0;
    fi.hdr_nsym = 0; fi.nvec = 0; fi.fresh = -1; fi.flags = (link ? kInfoLink : 0) | (late ? kInfoLate : 0); fi.spec_off = 0; fi.n_own = 0; fi.pad_ = 0;
    // complete windows [q + 144 + 80 k, + 64) that end by `end`; the window in progress when the next LTS1 arrives is pushed if it has
    // got past its cyclic prefix (m_offset > 15, fft_symbols.cpp:46): its first mo - 16 samples are its own
    const int64_t K = end >= q + 208 ? (end - (q + 208)) / 80 + 1 : 0;
    const int mo = end >= q + 128 ? (int)((end - (q + 128)) % 80) : 0;
    const bool has_part = link && mo > 15;
    if (p < 0 || end < q + 128 || K + (has_part ? 1 : 0) == 0) {
        // an LTS window is cut off, or not even part of a SIGNAL vector exists: no vector, no START_OF_FRAME from this alignment
        fi.status = link ? FOA_ST_SUPERSEDED : FOA_ST_TRUNCATED;
        if (lane == 0) info[f] = fi;
        return;
    }
    fi.nvec = (int32_t)min(K + (has_part ? 1 : 0), (int64_t)0x7FFFFFFF);
    fi.fresh = has_part ? mo - 16 : -1;
    const int s = lane_subcarrier(lane);
    // channel_est.cpp:44-58: est = sum over the two LTS of (LTS_FREQ_DOMAIN / Y) / 2
    cpx est = { 0.0, 0.0 };
    const cpx ref = { (double)g_tab.lts_freq[s], 0.0 };
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const int64_t w0 = w == 0 ? (late ? p + 64 : p) : q + 64;
        cpx y = fft64_lane(load_rotated(iq, w0 + lane, d), lds, lane);
        cpx r = cdiv(ref, y);
        est.x += r.x / 2.0;
        est.y += r.y / 2.0;
    }
    hinv[(size_t)f * 64 + s] = make_double2(est.x, est.y);
    // SIGNAL: equalise (channel_est.cpp:77-81), pilot phase with polarity[0] (phase_tracker.cpp:74-99).  If the next LTS1 cuts the
    // SIGNAL window itself (K = 0), the vector pushed holds its first `fresh` samples and, behind them, what the vector held before:
    // the second LTS window (fft_symbols.cpp:46-50)
    const int64_t sig_idx = (K == 0 && lane >= fi.fresh) ? q + 64 + lane : q + 144 + lane;
    cpx y = fft64_lane(load_rotated(iq, sig_idx, d), lds, lane);
    cpx z = pilot_derotate(cmul(est, y), (int)g_tab.polarity[0]);
    const int di = g_tab.data_index[s];
    if (eq_tap && di >= 0) eq_tap[(size_t)f * 48 + di] = make_double2(z.x, z.y);   // SIGNAL tap: one row per alignment
    // ppdu.cpp:171-175: BPSK demap, deinterleave
    if (di >= 0) {
        uint8_t b;
        qam_decode(z.x, 1, 128.0, &b);
        dem[deinterleaved_pos(di)] = b;
    }
    __syncthreads();
    int rate, length, nsym;
    decode_signal_bits(dem, decs, lane, rate, length, nsym);
    if (lane == 0) {
        if (rate >= 0) {
            fi.rate = rate; fi.length = length; fi.hdr_nsym = nsym;
            fi.n_own = (int32_t)min((int64_t)nsym, max(K - 1, (int64_t)0));
            if (nsym <= K - 1) { fi.status = FOA_ST_CRC_FAIL; fi.nsym = nsym; fi.nsteps = nsym * g_tab.rates[rate].dbps; }   // pending until the CRC is checked
            else if (link) { fi.status = FOA_ST_TRUNCATED; fi.flags |= kInfoCross; }      // needs vectors beyond its own complete windows: the scan decides
            else fi.status = FOA_ST_TRUNCATED;                                            // the samples end first
        }
        info[f] = fi;
    }
}

template <typename S>
__global__ __launch_bounds__(64) void k_header(const S *__restrict__ iq, const foa_frame_desc *__restrict__ descs,
                                               const int64_t *__restrict__ ends, int64_t n_samples, int n_lead, int n_total,
                                               FrameInfo *__restrict__ info, double2 *__restrict__ hinv, double2 *__restrict__ eq_tap)
{
    __shared__ cpx lds[64];
    __shared__ uint8_t dem[48];
    __shared__ uint64_t decs[24];
    const int f = blockIdx.x;
    if (f >= n_total) return;
    header_alignment(iq, descs, ends, n_samples, f, n_lead, n_total, info, hinv, eq_tap, lds, dem, decs);
}

// ---- the stream engines' look-ahead (stream_engine.h): which alignments of a batch buffer can be decided with the samples there are? ----
// The same header work over alignments range[0] .. range[1] - 1 of a buffer whose alignment structure is final up to sample n_samples
// (device-side range: the grid is an upper bound), into the engine's own records ...
__global__ __launch_bounds__(64) void k_header_range(const float2 *__restrict__ iq, const foa_frame_desc *__restrict__ descs, const int64_t *__restrict__ ends,
                                                     int64_t n_samples, const int32_t *__restrict__ range, FrameInfo *__restrict__ info,
                                                     double2 *__restrict__ hinv)
{
    __shared__ cpx lds[64];
    __shared__ uint8_t dem[48];
    __shared__ uint64_t decs[24];
    const int f = range[0] + (int)blockIdx.x;
    if (f >= range[1]) return;
    header_alignment(iq, descs, ends, n_samples, f, 0, range[1], info, hinv, (double2 *)nullptr, lds, dem, decs);      // (descs[0] is the buffer's first tag)
}

// =================================================================================================
// K2: exclusive scans over frames -> sym_off / spec_off / dec_off / seg_off, and the symbol / segment maps.
// =================================================================================================

// Three small kernels instead of one block: a single CU's memory pipeline (one 64-line request per wave instruction)
// made the one-block version the 50 us tail of a 2 ms call.  One thread per frame throughout.
constexpr int kScanBlock = 256;

struct ScanQ { int64_t v[4]; };          // data symbols, of which special (SpecSym entries), per-step words (soft pairs / decisions / decoded), chain-back segments

// A frame that is longer than its alignment's own complete windows and linked to what follows (kInfoCross): frame_decoder copies the
// vectors behind its SIGNAL wherever they come from -- its alignment's partial vector, the next alignment's SIGNAL vector, that
// alignment's symbols if its SIGNAL is invalid, and so on -- and drops the frame when a VALID SIGNAL arrives before the last of them
// (frame_decoder.cpp:52-88).  Walks the alignments behind f: FOA_ST_CRC_FAIL = the hdr_nsym vectors are all there (decode it),
// FOA_ST_SUPERSEDED = a valid SIGNAL intervenes, FOA_ST_TRUNCATED = the linked alignments end first.  The records of other alignments
// are read field by field and only in fields k_header wrote and nothing rewrites (nvec, rate, flags).
__device__ __forceinline__ int cross_walk(const FrameInfo *info, int f)
{
    const int nsym = info[f].hdr_nsym;
    int64_t c = info[f].nvec - 1;                                   // vectors behind the SIGNAL vector that the frame has so far
    int g = f;
    while (c < nsym) {
        if (!(info[g].flags & kInfoLink)) return FOA_ST_TRUNCATED;
        g++;
        const int nv = info[g].nvec;
        if (nv == 0) continue;
        if (info[g].rate >= 0 && c + 1 < nsym) return FOA_ST_SUPERSEDED;     // (a SIGNAL vector that is the frame's LAST vector completes it first)
        c += nv;
    }
    return FOA_ST_CRC_FAIL;
}

__device__ __forceinline__ ScanQ scan_quantities(const FrameInfo *info, int f, int n_frames, int seg_steps, int *cross_status = nullptr)
{
    ScanQ q = { { 0, 0, 0, 0 } };
    if (f < n_frames) {
        int nsym = info[f].nsym, nsteps = nsym > 0 ? info[f].nsteps : 0, spec = 0;
        if (info[f].flags & kInfoCross) {
            const int st = cross_walk(info, f);
            if (cross_status) *cross_status = st;
            if (st == FOA_ST_CRC_FAIL) { nsym = info[f].hdr_nsym; nsteps = nsym * g_tab.rates[info[f].rate].dbps; spec = nsym - info[f].n_own; }
        }
        q.v[0] = nsym; q.v[1] = spec; q.v[2] = dec_words(nsteps); q.v[3] = tb_segments(nsteps, seg_steps);
    }
    return q;
}

// exclusive scan over the block's threads; totals in tot[] (valid in every thread)
__device__ __forceinline__ ScanQ block_exclusive_scan(const ScanQ &q, int64_t (&tot)[4], int64_t (*part)[kScanBlock / 64])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ScanQ r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t x = q.v[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) part[i][wv] = x;
        r.v[i] = x - q.v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t run = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; w++) {
            if (w == wv) r.v[i] += run;
            run += part[i][w];
        }
        tot[i] = run;
    }
    return r;
}

// pass 1: per-block sums -> blk[4][n_blocks]
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const FrameInfo *__restrict__ info, int n_frames, int seg_steps, int64_t *__restrict__ blk)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    int64_t tot[4];
    block_exclusive_scan(scan_quantities(info, blockIdx.x * kScanBlock + threadIdx.x, n_frames, seg_steps), tot, part);
    if (threadIdx.x < 4) blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = tot[threadIdx.x];
}

// pass 2: exclusive scan of the block sums in place, grand totals -> totals[0..2] and totals[4]  (one block)
__global__ __launch_bounds__(1024) void k_scan_blocks(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    if (threadIdx.x == 0) totals[3] = sym_cap;            // read by the data-symbol kernels next to totals[0]
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = 0; i < 4; i++) {
        int64_t carry = 0;
        for (int base = 0; base < n_blocks; base += 1024) {
            const int j = base + tid;
            const int64_t v = j < n_blocks ? blk[(size_t)i * n_blocks + j] : 0;
            int64_t x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            __syncthreads();
            if (lane == 63) wsum[wv] = x;
            __syncthreads();
            int64_t before = 0, all = 0;
            for (int w = 0; w < 16; w++) {
                if (w < wv) before += wsum[w];
                all += wsum[w];
            }
            if (j < n_blocks) blk[(size_t)i * n_blocks + j] = carry + before + x - v;
            carry += all;
        }
        if (tid == 0) totals[i < 3 ? i : 4] = carry;          // totals[3] is the symbol capacity (above)
    }
}

// pass 2 for up to 64 x 64 blocks (a million frames) as ONE wave: under a forward pass a 1024-thread block waits for sixteen
// free wave slots on one CU (5.8 us alone, 100 us on average and up to 750 us in the pipelined bench), a lone wave for one
__global__ __launch_bounds__(64) void k_scan_blocks_w(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    const int lane = threadIdx.x, per = (n_blocks + 63) / 64, lo = lane * per, hi = min(lo + per, n_blocks);
    if (lane == 0) totals[3] = sym_cap;
    for (int i = 0; i < 4; i++) {
        int64_t s = 0;
        for (int j = lo; j < hi; j++) s += blk[(size_t)i * n_blocks + j];
        int64_t x = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        int64_t run = x - s;                                   // exclusive prefix of this lane's run
        for (int j = lo; j < hi; j++) { const int64_t v = blk[(size_t)i * n_blocks + j]; blk[(size_t)i * n_blocks + j] = run; run += v; }
        if (lane == 63) totals[i < 3 ? i : 4] = x;
    }
}

// pass 3: offsets into the frame records (frames that do not fit are marked FOA_ST_NO_SPACE), the symbol -> frame and segment -> frame
// maps (so that the data-symbol and chain-back kernels find their frame without a search; frames are short: <= 1368 symbols), and for
// the rare frame that fills on beyond its own alignment the table of where each of those symbols comes from.
// sym2frame[w] = f: symbol w is window k = w - sym_off + 1 of frame f's own alignment; -1: unused slot; <= -2: entry -2 - value of spec.
// frame f's share of pass 3, given its quantities and where its symbols (a), special symbols (sp0), per-step words (c) and segments (d) begin
__device__ __forceinline__ void scan_apply_frame(FrameInfo *__restrict__ info, int f, const ScanQ &q, int cross, int64_t a, int64_t sp0, int64_t c, int64_t d, int64_t sym_cap,
                                                 int64_t dec_cap, int64_t seg_cap, int32_t *__restrict__ sym2frame, int32_t *__restrict__ seg2frame, SpecSym *__restrict__ spec)
{
    const int nsym = (int)q.v[0], nspec = (int)q.v[1];
    info[f].seg_off = (int32_t)d;
    const int ns = (int)q.v[3];
    if (cross >= 0 && cross != FOA_ST_CRC_FAIL) { info[f].status = cross; return; }      // never completes: superseded, or the samples end first
    if (nsym > 0 && (a + nsym > sym_cap || c + q.v[2] > dec_cap || sp0 + nspec > sym_cap)) {
        // keeps its slots in the numbering; they are marked unused (as far as the maps reach)
        info[f].status = FOA_ST_NO_SPACE; info[f].nsteps = 0; info[f].nsym = 0;
        for (int k = 0; k < nsym; k++) if (a + k < sym_cap) sym2frame[a + k] = -1;
        for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = -1;
        return;
    }
    info[f].sym_off = (int32_t)a; info[f].dec_off = c;
    const int n_plain = nsym - nspec;
    for (int k = 0; k < n_plain; k++) sym2frame[a + k] = f;
    for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = f;
    if (cross == FOA_ST_CRC_FAIL) {
        // the frame decodes: its record becomes a plain pending one, its symbols beyond the own complete windows get their sources
        info[f].status = FOA_ST_CRC_FAIL; info[f].nsym = nsym; info[f].nsteps = nsym * g_tab.rates[info[f].rate].dbps; info[f].spec_off = (int32_t)sp0;
        int src = f, k = n_plain + 1;                                   // next vector of alignment src (vector 0 is its SIGNAL)
        for (int i = 0; i < nspec; i++) {
            while (k >= info[src].nvec) { src++; k = 0; }               // (alignments without vectors are skipped over)
            const int fr = info[src].fresh;
            SpecSym e;
            e.frame = f; e.src = src; e.k = k; e.fresh = ((fr >= 0 && k == info[src].nvec - 1) ? fr : 64) | ((info[src].flags & kInfoLate) ? 256 : 0);
            spec[sp0 + i] = e;
            sym2frame[a + n_plain + i] = -2 - (int32_t)(sp0 + i);
            k++;
        }
    }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_apply(FrameInfo *__restrict__ info, int n_frames, int64_t sym_cap,
                                                           int64_t dec_cap, int seg_steps, int64_t seg_cap, const int64_t *__restrict__ blk,
                                                           int32_t *__restrict__ sym2frame, int32_t *__restrict__ seg2frame, SpecSym *__restrict__ spec)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    const int f = blockIdx.x * kScanBlock + threadIdx.x;
    int cross = -1;
    const ScanQ q = scan_quantities(info, f, n_frames, seg_steps, &cross);
    int64_t tot[4];
    const ScanQ ex = block_exclusive_scan(q, tot, part);
    if (f >= n_frames) return;
    scan_apply_frame(info, f, q, cross, ex.v[0] + blk[blockIdx.x], ex.v[1] + blk[(size_t)gridDim.x + blockIdx.x], ex.v[2] + blk[(size_t)2 * gridDim.x + blockIdx.x],
                     ex.v[3] + blk[(size_t)3 * gridDim.x + blockIdx.x], sym_cap, dec_cap, seg_cap, sym2frame, seg2frame, spec);
}

// the three passes as ONE launch for a call of up to kScanBlock alignments (a stream engine's small batch: every launch it does not make
// is 4-5 us of its submitter thread and one dependent kernel less between a frame's last sample and its payload)
__global__ __launch_bounds__(kScanBlock) void k_scan_small(FrameInfo *__restrict__ info, int n_frames, int64_t sym_cap, int64_t dec_cap, int seg_steps, int64_t seg_cap,
                                                           int64_t *__restrict__ totals, int32_t *__restrict__ sym2frame, int32_t *__restrict__ seg2frame,
                                                           SpecSym *__restrict__ spec)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    const int f = threadIdx.x;
    int cross = -1;
    const ScanQ q = scan_quantities(info, f, n_frames, seg_steps, &cross);
    int64_t tot[4];
    const ScanQ ex = block_exclusive_scan(q, tot, part);
    if (f == 0) { totals[0] = tot[0]; totals[1] = tot[1]; totals[2] = tot[2]; totals[3] = sym_cap; totals[4] = tot[3]; }
    if (f >= n_frames) return;
    scan_apply_frame(info, f, q, cross, ex.v[0], ex.v[1], ex.v[2], ex.v[3], sym_cap, dec_cap, seg_cap, sym2frame, seg2frame, spec);
}

// ... and the first of them that cannot: FOA_ST_TRUNCATED is "the samples ended before this alignment's frame was complete", i.e. it and
// everything behind it wait for the next batch (unless the stream is over: final).  One wave.
// state: { stream index of the STS_END sample of the first alignment not decided yet, phasor in force before it } -- carried from batch to
// batch on the device.  sel: { STS_END candidates, alignments found, first alignment of the batch, how many it decides, context behind them }.
__global__ __launch_bounds__(64) void k_stream_resolve(const FrameInfo *__restrict__ info, const foa_frame_desc *__restrict__ descs, const int32_t *__restrict__ range,
                                                       const int32_t *__restrict__ sy_n, int64_t start_abs, int64_t hz_abs, int final, StreamState *__restrict__ state,
                                                       int32_t *__restrict__ sel)
{
    const int lane = threadIdx.x, i0 = range[0], i1 = range[1];
    int first = i1;
    if (!final)
        for (int a = i0 + lane; a < i1; a += 64) {
            int st = info[a].status;
            if (info[a].flags & kInfoCross) st = cross_walk(info, a);
            if (st == FOA_ST_TRUNCATED) { first = a; break; }
        }
#pragma unroll
    for (int o = 32; o; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (lane == 0) {
        sel[0] = sy_n[0]; sel[1] = sy_n[3]; sel[2] = i0; sel[3] = first - i0; sel[4] = i1 - first;
        state->lo_abs = first < i1 ? start_abs + descs[first].rot_start : hz_abs;
        if (first > i0) { state->c = descs[first - 1].c; state->s = descs[first - 1].s; }
    }
}

}  // namespace foa
