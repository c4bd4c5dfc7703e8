// viterbi_tb.h -- chain-back of the forward pass's decisions (viterbi.cpp:108-146), one LANE per segment, then descrambler
// (ppdu.cpp:256-264), CRC-32 (ppdu.cpp:267-279) and payload copy (ppdu.cpp:283-285), one lane per frame.
//
// With the decisions transposed (viterbi_fwd.h: u16 [block of 16 data steps][63 - slot], complemented bits), the address of the
// bit a path needs depends on the path, so the serial walk of one frame cannot be fed ahead of time.  Instead the walk is cut
// into SEGMENTS of S data steps, one lane per segment, all segments of all frames at once (k_tb_walk).  A lane does not know the
// state at the top of its segment, so it starts L steps higher in an arbitrary state and discards those bits; survivor paths
// merge quickly, so after L steps it is on the true path -- almost always.  "Almost" is not bit-exact, so each lane records the
// state it assumed at its top boundary (e) and the state it reached at its bottom boundary (s); k_tb_finish checks
// e(k) == s(k+1) down each frame (the top segment starts from the true terminal state, so by induction every segment that passes
// is the true path) and re-walks a segment from the proven state when the check fails.  That keeps the result identical to the
// serial chain-back whatever L is; L only trades overlap work against the frequency of re-walks (tests run with L = 0, where
// nearly every segment is re-walked).
#pragma once

#include "viterbi_fwd.h"

namespace foa {

constexpr int kTbBlockBytes = 8 * 1024;      // LDS of one 16-step decision block of the wave's 64 lanes: [lane / 8][lane % 8] x 128 B
constexpr int kTbRing = 3;                   // blocks resident per wave: one being walked, two streaming in
constexpr int kTbMaxSeg = 3072;              // largest segment length (LDS of the re-walk path: 8 B per step)

__device__ __forceinline__ void write_result(foa_frame_result *res, const FrameInfo &fi, int status)
{
    foa_frame_result r;
    r.status = status; r.rate = fi.rate; r.length = fi.length;
    r.num_symbols = fi.hdr_nsym;
    *res = r;
}

// descramble + CRC-32 + payload copy of one frame per lane
struct FinishTables { uint32_t crc[1024]; uint32_t scr[128]; };

__device__ __forceinline__ void finish_tables_init(FinishTables &t, int tid, int nthreads)
{
    for (int i = tid; i < 256; i += nthreads) {
        const uint32_t t0 = g_tab.crc_table[i];
        const uint32_t t1 = (t0 >> 8) ^ g_tab.crc_table[t0 & 0xFFu];
        const uint32_t t2 = (t1 >> 8) ^ g_tab.crc_table[t1 & 0xFFu];
        const uint32_t t3 = (t2 >> 8) ^ g_tab.crc_table[t2 & 0xFFu];
        t.crc[i] = t0; t.crc[256 + i] = t1; t.crc[512 + i] = t2; t.crc[768 + i] = t3;
    }
    for (int i = tid; i < 127; i += nthreads) {
        uint32_t m = 0;
        for (int b = 0; b < 4; b++) m |= (uint32_t)g_tab.scramble[(4 * i + b) % 127] << (8 * b);
        t.scr[i] = m;
    }
}

// LDS of the wave-cooperative finish (one per 64-frame wave)
struct FinishWave {
    uint32_t tile[64][17];         // 16 words of each of the wave's 64 frames (row stride 17: one bank per lane)
    int64_t base[64];              // each frame's offset (in words) into the decoded buffer
    int nwords[64], ncopy[64];     // words to descramble; payload bytes to copy (0 unless the CRC matched)
};

// descramble (one LFSR bit per byte, ppdu.cpp:256-264) + CRC over service+payload (ppdu.cpp:267-271) + payload copy
// (ppdu.cpp:283-285), one frame per lane.  Whole words go through four table look-ups that do not depend on each other
// (slicing-by-4); the scrambler's 127-byte period makes a 127-word table of descrambling masks.
// A lane that walks its own frame's words touches a different cache line than every other lane with each load and
// store, which is what this step used to spend its time on.  So memory is moved by the wave as a whole: four lanes
// per frame fetch 64 contiguous bytes of each of 16 frames per instruction into an LDS tile, each lane then works on
// its own row, and the descrambled words (and later the payload) leave the same way.  Wave-uniform control flow.
__device__ __forceinline__ void finish_crc_psdu(const FinishTables &t, FinishWave &fw, const FrameInfo &fi, bool live, int f, int n_frames,
                                                uint32_t *__restrict__ decoded, uint8_t *__restrict__ psdu, size_t slot_bytes,
                                                foa_frame_result *__restrict__ results)
{
    const int lane = threadIdx.x & 63, sub = lane >> 2, pc = lane & 3;
    const int len = fi.length, ncrc = live ? 2 + len : 0, nwords = live ? (ncrc + 4 + 3) / 4 : 0;
    int maxw = nwords;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxw = max(maxw, __shfl_xor(maxw, o));
    fw.base[lane] = live ? decoded_word_off(fi.dec_off) : 0;
    fw.nwords[lane] = nwords;
    wave_lds_sync();
    uint32_t crc = 0xFFFFFFFFu, given = 0;
    const int full = ncrc >> 2;                                            // words that lie entirely inside the CRC range
    int qs = 0;                                                            // q mod 127
    // in: frame 16 r + sub, words q0 + 4 pc .. + 3 (the regions are 256-byte aligned and padded past their last word);
    // the next 16-word chunk is fetched into registers while the current one is worked on
    uint4 pre[4];
    auto fetch = [&](int q0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            pre[r] = make_uint4(0u, 0u, 0u, 0u);
            if (q0 + 4 * pc < fw.nwords[fr]) pre[r] = *(const uint4 *)(decoded + fw.base[fr] + q0 + 4 * pc);
        }
    };
    fetch(0);
    for (int q0 = 0; q0 < maxw; q0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            fw.tile[fr][4 * pc] = pre[r].x; fw.tile[fr][4 * pc + 1] = pre[r].y; fw.tile[fr][4 * pc + 2] = pre[r].z; fw.tile[fr][4 * pc + 3] = pre[r].w;
        }
        if (q0 + 16 < maxw) fetch(q0 + 16);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int q = q0 + i;
            const uint32_t scr = t.scr[qs];
            qs = qs == 126 ? 0 : qs + 1;
            if (q < nwords) {
                const uint32_t d = fw.tile[lane][i] ^ scr;
                fw.tile[lane][i] = d;
                if (q < full) {
                    const uint32_t c = crc ^ d;
                    crc = t.crc[768 + (c & 0xFFu)] ^ t.crc[512 + ((c >> 8) & 0xFFu)] ^ t.crc[256 + ((c >> 16) & 0xFFu)] ^ t.crc[c >> 24];
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int x = 4 * q + b;
                        const uint32_t byte = (d >> (8 * b)) & 0xFFu;
                        if (x < ncrc) crc = t.crc[(crc ^ byte) & 0xFFu] ^ (crc >> 8);
                        else if (x < ncrc + 4) given |= byte << (8 * (x - ncrc));
                    }
                }
            }
        }
        wave_lds_sync();
        // out: the descrambled words, same pieces (words of a piece beyond the frame's last one are pad bits, descrambled or not)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            if (q0 + 4 * pc < fw.nwords[fr])
                *(uint4 *)(decoded + fw.base[fr] + q0 + 4 * pc) =
                    make_uint4(fw.tile[fr][4 * pc], fw.tile[fr][4 * pc + 1], fw.tile[fr][4 * pc + 2], fw.tile[fr][4 * pc + 3]);
        }
        wave_lds_sync();
    }
    const bool ok = live && (crc ^ 0xFFFFFFFFu) == given;
    const bool fits = (size_t)len <= slot_bytes;                           // a payload longer than the caller's slot is reported, never cut short
    if (f < n_frames) write_result(&results[f], fi, live ? (ok ? (fits ? FOA_ST_OK : FOA_ST_NO_SPACE) : FOA_ST_CRC_FAIL) : fi.status);

    // payload = descrambled bytes [2, 2+len) (ppdu.cpp:283-285), only for frames whose CRC matched: 16 bytes per lane,
    // payload bytes 64 c + 16 pc .. + 15 of frame 16 r + sub = bytes 2 .. 17 of the five words from 16 c + 4 pc on
    const int ncopy = ok && fits ? len : 0;
    fw.ncopy[lane] = ncopy;
    int maxc = ncopy;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
    __threadfence();                                                       // the words were stored by other lanes of this wave
    wave_lds_sync();
    const int f0 = f - lane;                                               // the wave's first frame
    for (int c = 0; 64 * c < maxc; c++) {
        const int y = 64 * c + 16 * pc;
        uint4 a[4];
        uint32_t a4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {                                      // all loads of the trip first
            const int fr = 16 * r + sub;
            // unconditional: the words exist for every frame of the wave (a frame that is not copied starts at offset 0
            // and the buffer is far longer than one PSDU), and four loads in flight beat four round trips
            const uint32_t *src = decoded + fw.base[fr] + 16 * c + 4 * pc;
            a[r] = *(const uint4 *)src;
            a4[r] = src[4];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub, nc = fw.ncopy[fr];
            if (y >= nc) continue;
            const uint4 v = make_uint4((a[r].x >> 16) | (a[r].y << 16), (a[r].y >> 16) | (a[r].z << 16), (a[r].z >> 16) | (a[r].w << 16),
                                       (a[r].w >> 16) | (a4[r] << 16));
            uint8_t *dst = psdu + (size_t)(f0 + fr) * slot_bytes + y;
            if (y + 16 <= nc && (((uintptr_t)dst) & 15) == 0) {
                *(uint4 *)dst = v;
            } else {
                const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (y + i < nc) dst[i] = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// chain-back, one lane per segment
// ---------------------------------------------------------------------------------------------------------
// LDS holds a ring of three 16-step decision blocks of every lane: [ring slot][lane / 8][lane % 8][slot] u16, whole
// 128-byte lines.  Eight consecutive lanes of an LDS-DMA instruction fetch the eight 16-byte pieces of ONE line (one
// block of one segment), so every instruction moves eight full lines.  The lane's LDS byte offset carries the
// complemented slot index in bits 1-6; the ring slot is an immediate offset.  A step is then: read 16 bits, shift the
// step's bit to its place, v_bfi it in.  24 KB per wave let six waves share a CU, which is what the kernel's speed
// hangs on: a lane's walk is one dependent chain of about 200 clocks per step, so time = steps per lane x rounds.
__device__ __forceinline__ uint32_t tb_lane_base(int lane) { return (uint32_t)(lane >> 3) * 1024u + (uint32_t)(lane & 7) * 128u; }
__device__ __forceinline__ uint32_t tb_gather(uint32_t a) { return (a >> 1) & 63u; }

// 16 steps of block KK (0..5) of a 96-step unit, i.e. data steps 96u + 16 KK + 15 .. 96u + 16 KK, decoded bits into w[]
// (data bit 96u+i at bit 31-(i mod 32) of w[i/32]: the reference's MSB-first byte order once the word is byte-swapped).
template <int KK>
__device__ __forceinline__ void tb_walk_block(const uint8_t *tb, uint32_t &a, uint32_t (&w)[3])
{
#pragma unroll
    for (int jj = 15; jj >= 0; jj--) {
        const int ju = 16 * KK + jj, q = 5 - ju % 6, pos = q + 1;            // unit starts are multiples of 96: phase = ju mod 6
        const uint32_t v = *(const uint16_t *)(tb + a + (KK % kTbRing) * kTbBlockBytes);
        const uint32_t tmp = pos >= jj ? v << (pos - jj) : v >> (jj - pos);
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(a) : "s"(1u << pos), "v"(tmp));
        if (ju % 6 == 0) {
            // the six bits just written are the six data bits of this group, complemented; p bit i = data bit 5-i
            const int o = ju, wi = o >> 5, r = o & 31;
            const uint32_t p = tb_gather(a) ^ 63u;
            if (r <= 26) {
                w[wi] |= p << (26 - r);
            } else {
                w[wi] |= p >> (r - 26);
                w[wi + 1] |= p << (58 - r);
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_tb_walk(const FrameInfo *__restrict__ info, const int32_t *__restrict__ seg2frame,
                                                const int64_t *__restrict__ totals, const uint64_t *__restrict__ dec,
                                                uint32_t *__restrict__ decoded, uint16_t *__restrict__ tb_state, int S, int L)
{
    __shared__ __attribute__((aligned(16))) uint8_t tb[kTbRing * kTbBlockBytes];
    const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
    const int n_seg = (int)totals[4];
    if (blockIdx.x * 64 >= n_seg) return;
    const int f = g < n_seg ? seg2frame[g] : -1;
    const bool live = f >= 0;
    FrameInfo fi;
    fi.nsteps = 0; fi.dec_off = 0; fi.seg_off = 0;
    if (live) fi = info[f];
    const int k = g - fi.seg_off, N = fi.nsteps - 6;
    const int n_lo = k * S, n_own = min(n_lo + S, N), n_hi = min(n_lo + S + L, N);
    const int cnt = live ? (n_hi - n_lo + kChunk3 - 1) / kChunk3 : 0;      // 48-step chunks this lane walks, numbered from its bottom
    const int own = live ? (n_own - n_lo + kChunk3 - 1) / kChunk3 : 0;     // of which the lowest `own` are its own
    const uint8_t *src = live ? (const uint8_t *)(dec + fi.dec_off) + (size_t)(n_lo / kChunk3) * 384 : (const uint8_t *)dec;
    uint32_t *out = decoded + decoded_word_off(fi.dec_off) + n_lo / 32;
    int cmax = cnt;
#pragma unroll
    for (int o = 32; o; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    uint32_t a = tb_lane_base(lane) + (63u << 1);                          // state 0; true at the frame's end, a guess elsewhere
    uint32_t e = 63u, w[3] = { 0u, 0u, 0u };

    // Fetch roles: in the instruction for lane group gi, this lane moves piece lane % 8 of a block of segment lane
    // 8 gi + lane / 8.  A segment lane that has no chunk i gets its chunk 0 again (always inside its region).
    const uint8_t *fsrc[8];
    int fcnt[8];
#pragma unroll
    for (int gi = 0; gi < 8; gi++) {
        const int from = 8 * gi + (lane >> 3);
        fsrc[gi] = (const uint8_t *)__shfl((unsigned long long)(uintptr_t)src, from) + (lane & 7) * 16;
        fcnt[gi] = __shfl(cnt, from);
    }
    // block bk (0..2) of chunk i of every segment lane -> ring slot bk
    auto fetch = [&](int i, int bk) {
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&tb[bk * kTbBlockBytes];
#pragma unroll
        for (int gi = 0; gi < 8; gi++) {
            const uint8_t *p = fsrc[gi] + (size_t)(i < fcnt[gi] ? i : 0) * 384 + 128 * bk;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(p), "s"(lds0 + (uint32_t)gi * 1024u)
                         : "memory");
        }
    };
    // Block KK of unit u is block KK % 3 of chunk 2u + KK / 3.  While it is walked the two blocks below it are in flight
    // or landed, and the block two below is requested as soon as the ring slot above is free (the block walked before).
#define FOA_TB_BLOCK(KK)                                                                                   \
    {                                                                                                      \
        constexpr int below = (KK) - 2;           /* in-unit index of the block to request, may be negative */ \
        if (below >= 0) fetch(2 * u + below / 3, below % 3);                                               \
        else if (u > 0) fetch(2 * (u - 1) + (below + 6) / 3, (below + 6) % 3);                             \
        if (below >= 0 || u > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                         \
        else if ((KK) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_wave_barrier();                                                                   \
        if ((KK) % 3 == 2 && 2 * u + (KK) / 3 == own - 1) e = tb_gather(a);                                \
        if (2 * u + (KK) / 3 < cnt) tb_walk_block<(KK)>(tb, a, w);                                         \
        __builtin_amdgcn_wave_barrier();                                                                   \
    }
    const int U = (cmax - 1) / 2;                                           // top unit
    fetch(2 * U + 1, 2);
    fetch(2 * U + 1, 1);
    for (int u = U; u >= 0; u--) {
        FOA_TB_BLOCK(5) FOA_TB_BLOCK(4) FOA_TB_BLOCK(3) FOA_TB_BLOCK(2) FOA_TB_BLOCK(1) FOA_TB_BLOCK(0)
        // own is even except in a frame's last segment, whose chunks above `own` lie beyond the frame's end (zeros)
        if (2 * u < own) {
            out[3 * u] = __builtin_bswap32(w[0]); out[3 * u + 1] = __builtin_bswap32(w[1]); out[3 * u + 2] = __builtin_bswap32(w[2]);
        }
        w[0] = w[1] = w[2] = 0u;
    }
#undef FOA_TB_BLOCK
    if (live) tb_state[g] = (uint16_t)(e | (tb_gather(a) << 8));
}

// Serial walk of data steps n_hi-1 .. n_lo (n_lo a multiple of 32) of one frame from state pbar, every lane of the
// wave with the same arguments (decisions staged through lds by the whole wave; lane 0 writes the decoded words).
// Returns the state at n_lo.  Only used when a segment's assumed start state turned out wrong.
__device__ __noinline__ uint32_t tb_rewalk(const uint16_t *__restrict__ d16, int n_lo, int n_hi, uint32_t pbar, uint32_t *__restrict__ out,
                                           uint16_t *lds, int lane)
{
    const int b_lo = n_lo >> 4, nblk = ((n_hi + 15) >> 4) - b_lo;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nblk * 64; i += 64) lds[i] = d16[(size_t)b_lo * 64 + i];
    wave_lds_sync();
    uint32_t word = 0;
    for (int n = n_hi - 1; n >= n_lo; n--) {
        const uint32_t s = (lds[((n >> 4) - b_lo) * 64 + pbar] >> (n & 15)) & 1u;
        const int q = 5 - n % 6;
        pbar = (pbar & ~(1u << q)) | (s << q);
        word |= (s ^ 1u) << (8 * ((n & 31) >> 3) + 7 - (n & 7));          // data bit n, MSB-first bytes in a little-endian word
        if ((n & 31) == 0) {
            if (lane == 0) out[n >> 5] = word;
            word = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return pbar;
}

// One lane per frame: stitch the segments (re-walking the rare one whose start state was wrong), then descramble,
// CRC and payload copy.
__global__ __launch_bounds__(64) void k_tb_finish(const FrameInfo *__restrict__ info, int n_frames, const uint64_t *__restrict__ dec,
                                                  uint32_t *__restrict__ decoded, const uint16_t *__restrict__ tb_state, int S,
                                                  uint8_t *__restrict__ psdu, size_t slot_bytes, foa_frame_result *__restrict__ results)
{
    __shared__ FinishTables tabs;
    __shared__ FinishWave fwave;
    __shared__ uint16_t rw[kTbMaxSeg / 16 * 64];
    const int lane = threadIdx.x, f = blockIdx.x * 64 + lane;
    finish_tables_init(tabs, lane, 64);
    __syncthreads();
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0; fi.dec_off = 0;
    fi.seg_off = 0; fi.hdr_nsym = 0;
    if (f < n_frames) fi = info[f];
    const bool live = f < n_frames && fi.nsym > 0;
    const int N = live ? fi.nsteps - 6 : 0, nseg = live ? tb_segments(fi.nsteps, S) : 0;

    int maxseg = nseg;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxseg = max(maxseg, __shfl_xor(maxseg, o));
    uint32_t s_next = nseg > 0 ? (uint32_t)(tb_state[fi.seg_off + nseg - 1] >> 8) : 0u;    // state at the bottom of the top segment
    for (int k = maxseg - 2; k >= 0; k--) {
        const bool has = k < nseg - 1;
        const uint32_t st = has ? tb_state[fi.seg_off + k] : 0u;
        uint32_t s_k = st >> 8;
        uint64_t redo = __ballot(has && (st & 0xFFu) != s_next);           // assumed start state != proven one
        while (redo) {
            const int l = __ffsll((unsigned long long)redo) - 1;
            redo &= redo - 1;
            const int64_t off = __shfl(fi.dec_off, l);
            const int n_l = __shfl(N, l);
            const uint32_t sn = __shfl(s_next, l);
            const uint32_t r = tb_rewalk((const uint16_t *)(dec + off), k * S, min(k * S + S, n_l), sn, decoded + decoded_word_off(off), rw, lane);
            if (lane == l) s_k = r;
        }
        if (has) s_next = s_k;
    }
    __threadfence();                                                       // re-walked words were written by lane 0
    if (psdu == nullptr) return;                                           // foa_conv_decode: the decoded bits are the result
    finish_crc_psdu(tabs, fwave, fi, live, f, n_frames, decoded, psdu, slot_bytes, results);
}

}  // namespace foa
