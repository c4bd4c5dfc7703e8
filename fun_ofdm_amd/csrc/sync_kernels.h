// sync_kernels.h -- frame_detector + timing_sync on the device.
//
// What the host restatement (sync_host.h) does sample by sample, restructured for a GPU:
//   k_sync_flags     frame_detector.cpp:51-66: lag-16 autocorrelation / power over a 16-sample window, threshold 0.9.
//                    The reference keeps running sums (sum -= old; sum += new, circular_accumulator.h:88-95) whose
//                    rounding errors drift for ever; here every sample's window is a sum of its own sixteen terms, so values agree
//                    to ~1e-15 relative and decisions differ only when the normalised correlation lies within that
//                    distance of the threshold (and on exactly-zero input, where the reference's leftovers decide;
//                    there this version gives the clean 0/0 = NaN -> "below").
//   k_sync_sts_end   frame_detector.cpp:67-84 is local once the flags exist: STS_END sits on the first sample that is
//                    below threshold after >= 16 consecutive samples above it.  Bit tricks on 32-sample words,
//                    ordered compaction of the candidates through a per-block count + scan.
//   k_sync_lts       timing_sync.cpp:69-113 per candidate: 96 x 64-tap cross-correlation with the LTS (same summation
//                    order as the reference), the five strongest peaks above 0.9, the 64-apart test against the
//                    strongest, the phase of timing_sync.cpp:113.
//   k_sync_keep/emit tags written by an earlier hit can overwrite a later STS_END (timing_sync.cpp:105-106); drop those,
//                    compact in stream order, chain the "previous phasor", derive the per-alignment end.
#pragma once

#include "device_math.h"
#include "sync_host.h"

namespace foa {

constexpr int kSyncBlockWords = 256;  // flag words per block of k_sync_sts_end (8192 samples)

struct SyncCand {
    int64_t x;            // stream index of the STS_END sample
    int64_t lts1_pos;     // valid if found
    double c, s;
    int32_t found, pad;
};

constexpr int kFlagSamples = 1024;   // samples per block of k_sync_flags
constexpr int kFlagGroupBytes = 144; // LDS bytes per group of 16 samples (128 + 16 of padding): a lane stride of 144 B serves ds_read_b128 without bank conflicts

// flags[w] bit i = (|corr_sum| / pow_sum > 0.9) at sample 32*w + i.
// The window of sample i holds the products x[j] conj(x[j-16]) and |x[j]|^2 of positions i-15 .. i, and a product belongs to
// sixteen windows.  One wave takes 1024 samples and every lane owns SIXTEEN CONSECUTIVE windows: those that start in its group of
// 16 product slots (slot s = position base - 15 + s).  Such a window is the tail of the lane's own group plus the head of the next
// one, so the lane forms the products of the two groups, the running tail sums of the first (15 additions per quantity), the
// running head sums of the second (14) and adds them pairwise (15): 2.75 additions per window and quantity where a direct sum of
// sixteen takes 15.  No subtraction is involved -- every window is still a plain sum of its own sixteen terms, in a fixed order, so
// nothing drifts and a huge sample is forgotten the moment it leaves the window (unlike the reference's running accumulator) -- and
// the sums agree with any other order of the same terms to ~1e-15 relative.
// The samples reach the lanes through LDS: the wave loads its 1024 + 47 positions once, coalesced, and each lane reads back the 48
// it needs (the products themselves would be three doubles per position: three times the LDS, a third of the resident waves).
// |S| / P > 0.9 is decided on S.S against 0.81 P.P wherever that is clear by a margin of 1e-9 (the rounding of hypot and of the
// division is 1e-15); only the rest takes hypot() and the division.  A lane's sixteen verdicts leave as one 16-bit store.
// The rare window the quick test cannot call: summed again term by term, oldest first, the way circular_accumulator.h:88-95 takes its
// samples in -- a product (or a power term) that is NaN counts as zero there (`if(sample != sample) sample = 0`, for a complex sample:
// either part), so a NaN in the stream costs the reference the two products it is part of and nothing else, and this stage likewise.
// (An INFINITE sample is another matter: the reference's running sum turns NaN when it leaves the window and stays NaN for good;
// here the window is simply not above threshold while it holds one.)
__device__ __forceinline__ bool sync_flag_slow(const uint8_t *raw, int lane, int r)
{
#pragma clang fp contract(off)
    auto sample = [&](int t) -> cpx {                                        // t = 0 .. 47: position base - 31 + 16 lane + t
        return widen(*(const float2 *)(raw + (lane + (t >> 4)) * kFlagGroupBytes + (t & 15) * 8));
    };
    cpx S = { 0.0, 0.0 };
    double P = 0.0;
    for (int m = 0; m < 16; m++) {
        const cpx a = sample(16 + r + m), b = sample(r + m);
        double pr = a.x * b.x + a.y * b.y, pi = a.y * b.x - a.x * b.y, pw = a.x * a.x + a.y * a.y;
        if (pr != pr || pi != pi) { pr = 0.0; pi = 0.0; }
        if (pw != pw) pw = 0.0;
        S.x += pr; S.y += pi; P += pw;
    }
    return hypot(S.x, S.y) / P > 0.9;                                        // 0/0 and NaN/x are not above
}

__global__ __launch_bounds__(64, 3) void k_sync_flags(const float2 *__restrict__ iq, int64_t n, uint32_t *__restrict__ flags, int64_t n_words)
{
#pragma clang fp contract(off)
    constexpr int kGroups = kFlagSamples / 16 + 2;                           // group g = positions base - 31 + 16 g .. + 15; lane l reads groups l, l+1, l+2
    constexpr int kRounds = (kGroups * 16 + 63) / 64;
    __shared__ __attribute__((aligned(16))) uint8_t raw[(kRounds * 4 + 1) * kFlagGroupBytes];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kFlagSamples;
    {
        float2 v[kRounds];
#pragma unroll
        for (int m = 0; m < kRounds; m++) {
            const int64_t j = base - 31 + lane + 64 * m;
            v[m] = (j >= 0 && j < n) ? iq[j] : float2{ 0.f, 0.f };
        }
#pragma unroll
        for (int m = 0; m < kRounds; m++) {
            const int rel = lane + 64 * m;
            *(float2 *)(raw + (rel >> 4) * kFlagGroupBytes + (rel & 15) * 8) = v[m];
        }
    }
    __syncthreads();
    const float4 *g = (const float4 *)(raw + lane * kFlagGroupBytes);       // 9 float4 per group (the ninth is padding)
    // products of slot r of the own group: x = g1[r], 16 back = g0[r]; of the next group: x = g2[r], 16 back = g1[r]
    float2 g1[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float4 w = g[9 + k];
        g1[2 * k] = float2{ w.x, w.y }; g1[2 * k + 1] = float2{ w.z, w.w };
    }
    double T[3][16];                                                         // T[q][r]: slots r .. 15 of the own group
    {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int r = 15; r >= 0; r--) {
            const float4 u = g[r / 2];                                       // (group 0 is read as it is used: its sixteen samples need not all be live)
            const cpx a = widen(g1[r]), b = widen((r & 1) ? float2{ u.z, u.w } : float2{ u.x, u.y });
            const double p0 = a.x * b.x + a.y * b.y, p1 = a.y * b.x - a.x * b.y, p2 = a.x * a.x + a.y * a.y;      // a * conj(b), |a|^2
            if (r == 15) { t0 = p0; t1 = p1; t2 = p2; }
            else { t0 = p0 + t0; t1 = p1 + t1; t2 = p2 + t2; }
            T[0][r] = t0; T[1][r] = t1; T[2][r] = t2;
        }
    }
    uint32_t mask = 0, slow = 0;
    const int64_t i0 = base + 16 * lane;
    double h0 = 0.0, h1 = 0.0, h2 = 0.0;                                     // slots 0 .. r-1 of the next group
#pragma unroll
    for (int r = 0; r < 16; r++) {
        double Sx = T[0][r], Sy = T[1][r], P = T[2][r];
        if (r > 0) {
            float2 xa;
            if ((r - 1) % 2 == 0) { const float4 u = g[18 + (r - 1) / 2]; xa = float2{ u.x, u.y }; }
            else { const float4 u = g[18 + (r - 1) / 2]; xa = float2{ u.z, u.w }; }
            const cpx a = widen(xa), b = widen(g1[r - 1]);
            const double p0 = a.x * b.x + a.y * b.y, p1 = a.y * b.x - a.x * b.y, p2 = a.x * a.x + a.y * a.y;
            if (r == 1) { h0 = p0; h1 = p1; h2 = p2; }
            else { h0 = h0 + p0; h1 = h1 + p1; h2 = h2 + p2; }
            Sx += h0; Sy += h1; P += h2;
        }
        const double q = Sx * Sx + Sy * Sy, pp = P * P;
        bool above;
        if (q > pp * (0.81 * (1.0 + 1e-9))) above = true;                   // (the margins are constants: two multiplies per window less than 0.81 P.P scaled twice)
        else if (q < pp * (0.81 * (1.0 - 1e-9)) || P == 0.0) above = false; // no power: 0/0 or NaN/0, never above
        else { above = false; slow |= 1u << r; }                             // too close to call, or not finite: settled below
        if (above && i0 + r < n) mask |= 1u << r;
    }
    while (slow) {                                                           // (rare: kept out of the loop above, whose registers it would claim)
        const int r = __ffs(slow) - 1;
        slow &= slow - 1;
        if (sync_flag_slow(raw, lane, r) && i0 + r < n) mask |= 1u << r;
    }
    const int64_t h16 = i0 >> 4;                                             // half-word of the flag array (little endian: bit i of word w = sample 32 w + i)
    if (h16 < 2 * n_words) ((uint16_t *)flags)[h16] = (uint16_t)mask;
}

// STS_END candidates of one block of flag words: count (pass 0) or write in order at offsets[block] (pass 1)
__global__ __launch_bounds__(kSyncBlockWords) void k_sync_sts_end(const uint32_t *__restrict__ flags, int64_t n_words, int pass,
                                                                   int32_t *__restrict__ block_count, const int32_t *__restrict__ block_off,
                                                                   int64_t *__restrict__ cand_x, int32_t cap)
{
    __shared__ int32_t cnt[kSyncBlockWords];
    const int t = threadIdx.x;
    const int64_t w = (int64_t)blockIdx.x * kSyncBlockWords + t;
    uint32_t ends = 0;
    if (w < n_words) {
        const uint64_t cur = flags[w], prev = w > 0 ? flags[w - 1] : 0u;
        const uint64_t B = (cur << 32) | prev;
        uint64_t u = B & (B << 1);          // b[n] & b[n-1]
        u &= u << 2;
        u &= u << 4;
        u &= u << 8;                        // u[n] = b[n-15..n] all set
        ends = (uint32_t)((~B & (u << 1)) >> 32);   // below threshold now, the 16 before all above
    }
    cnt[t] = __popc(ends);
    __syncthreads();
    // exclusive scan of the 256 counts (serial over 256 by one lane would do; log-step keeps it short)
    for (int o = 1; o < kSyncBlockWords; o <<= 1) {
        int v = t >= o ? cnt[t - o] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    if (pass == 0) {
        if (t == kSyncBlockWords - 1) block_count[blockIdx.x] = cnt[t];
        return;
    }
    int pos = block_off[blockIdx.x] + cnt[t] - __popc(ends);
    while (ends) {
        const int i = __ffs(ends) - 1;
        ends &= ends - 1;
        if (pos < cap) cand_x[pos] = w * 32 + i;
        pos++;
    }
}

// exclusive scan of a block's 1024 per-thread values (in LDS part[]); returns the thread's offset, total in *total
__device__ __forceinline__ int block1024_exclusive_scan(int v, int32_t *part, int *total)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) part[wv] = x;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int pv = part[w];
        if (w < wv) before += pv;
        all += pv;
    }
    *total = all;
    return before + x - v;
}

// exclusive scan of block counts; total -> out_total[0].  ONE WAVE: the stage runs under the forward pass of a decode call in flight, where a
// 1024-thread block waits for sixteen free wave slots on one CU -- 240-320 us in the pipelined loop for 8 us of work (a kernel trace of
// BASELINE config 2 with the pre-sync in the loop, profiles/r06_presync_timeline.txt; the decode call's own scan learnt this in round 3).
// 64 consecutive counts per turn (one coalesced load), four turns' loads in flight.
__global__ __launch_bounds__(64) void k_sync_scan(const int32_t *__restrict__ cnt, int n_blocks, int32_t *__restrict__ off, int32_t *__restrict__ out_total)
{
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n_blocks; base += 256) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = base + 64 * u + lane; v[u] = i < n_blocks ? cnt[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int x = v[u];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
            const int i = base + 64 * u + lane;
            if (i < n_blocks) off[i] = carry + x - v[u];
            carry += __shfl(x, 63);
        }
    }
    if (lane == 0) out_total[0] = carry;
}

// (value, index) of the lane with the largest (value, then index); lanes with v < 0 never win
__device__ __forceinline__ void wave_argmax(double &v, int &p)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int op = __shfl_xor(p, o);
        if (ov > v || (ov == v && op > p)) { v = ov; p = op; }
    }
}

// timing_sync.cpp:69-113 for one STS_END candidate per wave
// origin: stream index of iq[0]; call: the reference receiver's call size to decide timing_sync.cpp:99 by (sync_host.h), 0 = one call
__global__ __launch_bounds__(64) void k_sync_lts(const float2 *__restrict__ iq, int64_t n, const int64_t *__restrict__ cand_x, const int32_t *__restrict__ n_cand,
                                                  int32_t cap, SyncCand *__restrict__ out, int64_t origin, int64_t call)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x, nc = min(*n_cand, cap);
    auto at = [&](int64_t i) -> cpx { return (i >= 0 && i < n) ? widen(iq[i]) : cpx{ 0.0, 0.0 }; };
    // the candidate count lives on the device: a fixed grid strides over it, so the host need not learn it in between
    __shared__ double2 win[160];                          // samples x .. x+159 of the candidate, widened once
    __shared__ double win_pw[160];                        // ... and their |a|^2, rounded as the reference's sum takes it in: every position's power sum adds the same 64 terms
    for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int64_t x = cand_x[c];
    __syncthreads();                                      // (one wave) the window of the candidate before is no longer read
    for (int i = lane; i < 160; i += 64) { const cpx a = at(x + i); win[i] = make_double2(a.x, a.y); win_pw[i] = a.x * a.x + a.y * a.y; }
    __syncthreads();
    // corr_norm for p = x + lane and (lanes < 32) p = x + 64 + lane
    double v[2] = { -1.0, -1.0 };
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && lane >= 32) break;
        const int p = 64 * h + lane;                      // relative to x
        cpx corr = { 0.0, 0.0 };
        double power = 0.0;
#pragma unroll 8
        for (int s = 0; s < 64; s++) {
            const double2 aw = win[p + s];
            const cpx a = { aw.x, aw.y };
            const cpx m = cmul(a, cpx{ g_tab.lts_conj_re[s], g_tab.lts_conj_im[s] });
            corr.x += m.x; corr.y += m.y;
            power += win_pw[p + s];
        }
        const double cn = hypot(corr.x, corr.y) / power;
        if (cn > 0.9) v[h] = cn;                         // NaN (no power) fails the test like in the reference
    }
    // five strongest, in the reference's order (descending value, then descending position)
    int pk[5];
    int npk = 0;
    for (int r = 0; r < 5; r++) {
        double bv = v[0];
        int bp = lane;
        if (v[1] > bv || (v[1] == bv && v[1] >= 0.0)) { bv = v[1]; bp = 64 + lane; }
        wave_argmax(bv, bp);
        if (bv < 0.0) break;
        pk[npk++] = bp;
        if (bp == lane) v[0] = -1.0;
        if (bp == 64 + lane) v[1] = -1.0;
    }
    SyncCand o;
    o.x = x; o.found = 0; o.lts1_pos = 0; o.c = 1.0; o.s = 0.0; o.pad = 0;
    for (int t = 0; t < npk; t++) {
        if (abs(pk[0] - pk[t]) != 64) continue;
        // positions are relative to x; the reference's working buffer starts 160 samples before the stream
        const int64_t lts_offset = x + min(pk[0], pk[t]) - 32;
        // timing_sync.cpp:99: the LTS guard interval would start before the working buffer of the call that walks over x
        if (sync_call_index(origin + x, call) + min(pk[0], pk[t]) - 32 < 0) break;
        const cpx a = at(lts_offset + 159);
        const cpx m = cmul(a, cpx{ g_tab.lts_conj_re[63], g_tab.lts_conj_im[63] });
        const double phase = atan2(m.y, m.x);            // timing_sync.cpp:113
        o.found = 1; o.lts1_pos = lts_offset + 24; o.c = cos(phase); o.s = sin(phase);
        break;
    }
    if (lane == 0) out[c] = o;
    }
}

// keep[c] = found and not overwritten by the tags of one of the four candidates before it; block_count[b] = kept in block b
__global__ __launch_bounds__(256) void k_sync_keep(const SyncCand *__restrict__ cand, const int32_t *__restrict__ n_cand, int32_t cap, int32_t *__restrict__ keep,
                                                   int32_t *__restrict__ block_count)
{
    __shared__ int32_t wsum[4];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = min(*n_cand, cap);
    int k = 0;
    if (c < nc) {
        k = cand[c].found;
        for (int b = 1; b <= 4 && c - b >= 0 && k; b++) {
            const SyncCand &e = cand[c - b];
            if (e.found && (cand[c].x == e.lts1_pos || cand[c].x == e.lts1_pos + 64)) k = 0;
        }
        keep[c] = k;
    }
    const int in_wave = __popcll(__ballot(k != 0));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = in_wave;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ordered compaction of the kept candidates -> descriptors.  block_off[] = exclusive scan of k_sync_keep's counts,
// *n_kept = their total.  A kept candidate finds its neighbours in the kept sequence by walking keep[] (nearly every
// candidate is kept, so the walks are one or two steps): the phasor before it (timing_sync.cpp:113 applies the rotation
// found at one LTS until the next) and the start of the alignment after it, which is where its own ends.
__global__ __launch_bounds__(256) void k_sync_emit(const SyncCand *__restrict__ cand, const int32_t *__restrict__ keep, const int32_t *__restrict__ n_cand,
                                                   int32_t cap, const int32_t *__restrict__ block_off, const int32_t *__restrict__ n_kept, int64_t n_samples,
                                                   foa_frame_desc *__restrict__ descs, int64_t *__restrict__ ends, int32_t desc_cap)
{
    __shared__ int32_t wsum[4];
    const int c = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nc = min(*n_cand, cap);
    const int k = c < nc ? keep[c] : 0;
    const uint64_t m = __ballot(k != 0);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    if (!k) return;
    int pos = block_off[blockIdx.x] + __popcll(m & ((1ull << lane) - 1));
    for (int w = 0; w < wv; w++) pos += wsum[w];
    if (pos >= desc_cap) return;
    const int total = min(*n_kept, desc_cap);
    foa_frame_desc d;
    d.lts1_pos = cand[c].lts1_pos; d.rot_start = cand[c].x; d.c = cand[c].c; d.s = cand[c].s;
    d.c_prev = 1.0; d.s_prev = 0.0;
    for (int j = c - 1; j >= 0; j--)
        if (keep[j]) { d.c_prev = cand[j].c; d.s_prev = cand[j].s; break; }
    descs[pos] = d;
    int64_t e = n_samples;
    if (pos + 1 < total)
        for (int j = c + 1; j < nc; j++)
            if (keep[j]) { e = cand[j].lts1_pos; break; }
    ends[pos] = e;
}

// Stream engines (stream_engine.h): the alignments the pre-sync found over a batch's buffer that are still to be decided and whose tags
// are final -- STS_END sample (rot_start) in [lo, hz), buffer-relative, lo from the state the batch before left on the device -- and the
// phasor timing_sync had in force before the first of them (the phasor of the last alignment decided so far).  One wave; the descriptors
// are in stream order, so the range is [#{rot_start < lo}, #{rot_start < hz}).
// A small batch buffer filled by ONE kernel: the carry (the last `carry` samples of the buffer before: device memory; null = silence before the
// stream) and the batch itself, read straight out of the engine's page-locked staging memory.  hipMemcpyAsync from host memory goes to a DMA
// engine, and the hand-over between that engine and the compute queue the pre-sync runs on was ~25 us for a 64-KB copy
// (profiles/r06_latency_stages.txt, section 6): more than the copy and the pre-sync's first kernels together.
__global__ __launch_bounds__(256) void k_stream_fill(float2 *__restrict__ dst, const float2 *__restrict__ carry_src, int64_t carry, const float2 *__restrict__ host_src, int64_t n_new)
{
    const int64_t n = carry + n_new;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = i < carry ? (carry_src ? carry_src[i] : float2{ 0.f, 0.f }) : host_src[i - carry];
}

__global__ __launch_bounds__(64) void k_stream_range(foa_frame_desc *__restrict__ descs, const int32_t *__restrict__ sy_n, int32_t cap, int64_t start_abs, int64_t hz_abs,
                                                     const StreamState *__restrict__ state, int32_t *__restrict__ range)
{
    const int lane = threadIdx.x;
    const int found = min(sy_n[3], cap);
    const int64_t lo = state->lo_abs - start_abs, hz = hz_abs - start_abs;
    int below_lo = 0, below_hz = 0;
    for (int i = lane; i < found; i += 64) {
        const int64_t x = descs[i].rot_start;
        below_lo += x < lo;
        below_hz += x < hz;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { below_lo += __shfl_xor(below_lo, o); below_hz += __shfl_xor(below_hz, o); }
    if (lane == 0) {
        below_hz = max(below_hz, below_lo);
        if (below_hz > below_lo) { descs[below_lo].c_prev = state->c; descs[below_lo].s_prev = state->s; }
        range[0] = below_lo; range[1] = below_hz;
    }
}

}  // namespace foa
