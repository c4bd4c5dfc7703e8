// viterbi_v4.h -- K=7 Viterbi forward pass, FOUR states per lane and four frames per wave.
//
// Same recursion, same exact u8-saturating semantics in biased packed u16 (viterbi_v2.h / viterbi_v3.h; viterbi.cpp:208-457), a
// different map of the trellis onto the wave.  In viterbi_v3.h a lane holds ONE state (of two frames, in the halves of a register), so
// every trellis step pairs two lanes and pays for an exchange (2.0 instructions, 8.7 clocks of a 43-clock step) and a frame pair
// owns a whole wave: one v_readfirstlane per step for the renormalisation test, one six-level reduction per event.  Here
//
//   * a frame is a ROW of 16 lanes (a DPP row), a wave carries four frames; a lane holds four states of its frame in two registers
//     V0 = (slot 0 | slot 1 << 16), V1 = (slot 2 | slot 3 << 16): position = (lane coordinates c0..c3, register bit g, half bit h);
//   * the add-compare-select of a step is six packed instructions for the lane's two butterflies whatever the step:
//         E = min(X + W, Y + ~W),  O = min(X + ~W, Y + W)         X = (two low predecessors), Y = (their partners i + 32),
//     W = (m of butterfly 0, m of butterfly 1), ~W = 63 - W: both from one 8-byte LDS read of the lane's entry of the step's table;
//   * one step in five pairs the lane's own two registers (X = V0, Y = V1: nothing moves); the other four pair it with the lane
//     16-lane-row-xor 8, 7, 2, 1 away: the lower lane of a pair takes the butterflies of both lanes' V0, the upper lane those of both
//     lanes' V1, so ONE register goes each way -- three DPP moves (two of them bank-masked) for xor 8 / xor 7 (row_ror:8,
//     row_half_mirror), two moves and two selects for the two quad steps: 2.8 instructions per step FOR FOUR FRAMES against 4.0;
//   * the results are NOT written back in place: the lane keeps all four new states of its two butterflies (E in V0, O in V1), so the
//     map of state labels onto positions is a bit permutation that changes with the step.  The schedule below has period 5
//     (tools/viterbi_v4_model.py finds it and checks every claim made here against the oracle's scalar model):
//         phase   pairs on          after the step, label bit of (c0, c1, c2, c3, g, h)
//           0     g  (in lane)      2 3 4 5 0 1
//           1     c3 (xor 8)        3 4 5 1 0 2
//           2     c2 (xor 7)        4 5 1 2 0 3
//           3     c1 (xor 2)        5 1 2 3 0 4
//           4     c0 (xor 1)        1 2 3 4 5 0        (+ one 16-bit transpose of (E, O): the new low bit goes to h, not g)
//     with lane-in-row = c0 ^ 2 c1 ^ 7 c2 ^ 8 c3.  Label 0 is position 0 under every linear map: state 0 is lane 0 of the row, low half of V0;
//   * the renormalisation test of all four frames is ONE compare (lanes 0, 16, 32, 48 of a v_cmp mask), an event reduces inside the
//     rows (four DPP minima, no v_readlane, no trip to the scalar unit for the amount) for all four frames at once and subtracts under
//     an EXEC mask of the rows that are due;
//   * decisions: the sign bytes of the two packed differences (bits 8..15 of a half all equal its sign, viterbi_v3.h) are gathered by one
//     v_perm_b32 and filed by one v_bfi_b32: four instructions for the sixteen decisions of four frames, a dword per lane per 8 steps.
//
// Decision memory (v4 layout), per frame 8 bytes per data step as before: [8-step sub-block][16 dwords]; the dword of the lane with
// coordinate index x = c0 | c1 << 1 | c2 << 2 | c3 << 3 sits at index 15 - x, the byte of slot s = 2 g + h at byte 3 - s, the bit of data
// step n at bit n & 7 of that byte; 1 = "survivor came from the LOW predecessor" (all as viterbi_v3.h: a chain-back that keeps the
// complemented position copies the bit it reads).  Bits at or beyond a frame's last step, up to the end of its last 40-step chunk, are 1.
#pragma once

#include "viterbi_v3.h"

namespace foa {

constexpr int kChunk4 = 40;                   // data steps per chunk: 8 periods of 5 phases, 5 sub-blocks of 8 steps
#ifndef FOA_FWD4_WAVES
#define FOA_FWD4_WAVES 4
#endif
constexpr int kFwd4Waves = FOA_FWD4_WAVES;    // waves (groups of four frames) per workgroup

// Branchtab class (viterbi.cpp:86-91) of the butterfly in the LOW half of the canonical X, two bits per lane-in-row, by phase; the
// butterfly in the high half has class ^ a(phase): kA4, two bits per phase.  (tools/viterbi_v4_model.py prints both.)
__device__ constexpr uint32_t kCls4[5] = { 0x5a5af0f0u, 0xc369963cu, 0x6c6c6c6cu, 0xd82727d8u, 0x11bbee44u };
constexpr uint32_t kA4 = 0x2f1u;
// label bit of position bits (c0, c1, c2, c3, g, h) AFTER the step of phase ph: where the decision of a new state is filed
__host__ __device__ constexpr int v4_label_bit(int ph, int posbit)
{
    constexpr int t[5][6] = { { 2, 3, 4, 5, 0, 1 }, { 3, 4, 5, 1, 0, 2 }, { 4, 5, 1, 2, 0, 3 }, { 5, 1, 2, 3, 0, 4 }, { 1, 2, 3, 4, 5, 0 } };
    return t[ph][posbit];
}
__host__ __device__ constexpr int v4_cidx(int l) { return ((l ^ (l >> 2)) & 1) | ((((l >> 1) ^ (l >> 2)) & 1) << 1) | (l & 12); }   // c0 | c1<<1 | c2<<2 | c3<<3

template <int CTRL, int BANK>
__device__ __forceinline__ uint32_t dpp4(uint32_t old, uint32_t src)       // lanes of the banks in BANK: src of the lane CTRL names; the others: old
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK, false);
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp4_all(uint32_t src)                   // every lane (no "old" to keep: saves the copy update_dpp makes for it)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t sel4(uint32_t a, uint32_t b, uint64_t mask)      // mask ? b : a, the mask a scalar pair (4 clocks; on VCC it is 16)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}

// the canonical operands of phase PH: X = the two low predecessors this lane works on, Y = their partners
template <int PH>
__device__ __forceinline__ void fwd4_operands(uint32_t V0, uint32_t V1, uint32_t &X, uint32_t &Y)
{
    if constexpr (PH == 0) { X = V0; Y = V1; }
    else if constexpr (PH == 1) {                         // xor 8: the upper lane is banks 2, 3
        const uint32_t T = dpp4_all<0x128>(V0);     // row_ror:8
        X = dpp4<0x128, 0xC>(V0, V1);
        Y = dpp4<0xE4, 0x3>(V1, T);
    } else if constexpr (PH == 2) {                       // xor 7: the upper lane is banks 1, 3
        const uint32_t T = dpp4_all<0x141>(V0);     // row_half_mirror
        X = dpp4<0x141, 0xA>(V0, V1);
        Y = dpp4<0xE4, 0x5>(V1, T);
    } else if constexpr (PH == 3) {                       // xor 2: upper lane where c1 = b1 ^ b2 is set
        const uint32_t T0 = dpp4_all<0x4E>(V0), T1 = dpp4_all<0x4E>(V1);
        X = sel4(V0, T1, 0x3C3C3C3C3C3C3C3Cull);
        Y = sel4(T0, V1, 0x3C3C3C3C3C3C3C3Cull);
    } else {                                              // xor 1: upper lane where c0 = b0 ^ b2 is set
        const uint32_t T0 = dpp4_all<0xB1>(V0), T1 = dpp4_all<0xB1>(V1);
        X = sel4(V0, T1, 0x5A5A5A5A5A5A5A5Aull);
        Y = sel4(T0, V1, 0x5A5A5A5A5A5A5A5Aull);
    }
}

// Renormalisation (viterbi.cpp:314-332) of the rows whose bit is set in `due` (bits 0, 16, 32, 48: state 0 of that frame exceeds 210):
// the row's smallest metric comes off all of its states.
__device__ __forceinline__ void fwd4_renorm(uint32_t &V0, uint32_t &V1, uint64_t due)
{
#if FOA_RN_PRIO
    __builtin_amdgcn_s_setprio(FOA_RN_PRIO);
#endif
    uint32_t t = pk_min(V0, V1);
    asm("v_pk_min_u16 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 1\n\t"                       // both halves: the lane's smallest
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      // (equal halves stay equal under a 32-bit minimum)
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(t));
    const uint32_t adj = t - kBias2;
    const uint32_t lo = ((uint32_t)due & 0x00010001u) * 0xFFFFu, hi = ((uint32_t)(due >> 32) & 0x00010001u) * 0xFFFFu;
    const uint64_t rows = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b64 exec, %2\n\tv_pk_sub_u16 %0, %0, %3\n\tv_pk_sub_u16 %1, %1, %3\n\ts_mov_b64 exec, -1"
                 : "+v"(V0), "+v"(V1) : "s"(rows), "v"(adj));
#if FOA_RN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// One trellis step of phase PH for the wave's four frames.  J >= 0: data step J of the chunk, decisions filed at bit J & 7 of acc; J < 0: no decision.
template <int PH, int J>
__device__ __forceinline__ void fwd4_step(uint32_t &V0, uint32_t &V1, const uint2 w, uint32_t &acc)
{
    uint32_t X, Y;
    fwd4_operands<PH>(V0, V1, X, Y);
    const uint32_t W = w.x, Wb = w.y;
    const uint32_t AE = pk_add_sat(X, W), BE = pk_add_sat(Y, Wb), AO = pk_add_sat(X, Wb), BO = pk_add_sat(Y, W);
    uint32_t E = pk_min(AE, BE), O = pk_min(AO, BO);
    // the test: state 0 = lane 0 of each row, low half of E (in phase 4 too: the transpose below leaves E's low half where it is)
    uint64_t due;
    asm volatile("v_cmp_lt_u16_e64 %0, %1, %2" : "=s"(due) : "s"(kRenormThr), "v"(E));
    if constexpr (J >= 0) {
        // upper predecessor wins ties: survivor = low predecessor iff A < B iff the 16-bit difference is negative (bits 8..15 of the half).
        // Sign bytes to bytes 3 - slot, slot = 2 g + h of the NEW state: E -> g = 0, O -> g = 1, half -> h; in phase 4 E -> h = 0, O -> h = 1, half -> g.
        constexpr uint32_t sel = PH == 4 ? 0x01050307u : 0x01030507u;
        constexpr uint32_t m = 0x01010101u << (J & 7);
        uint32_t d0, d1;
        asm volatile("v_pk_sub_u16 %1, %3, %4\n\tv_pk_sub_u16 %2, %5, %6\n\tv_perm_b32 %1, %2, %1, %7\n\tv_bfi_b32 %0, %8, %1, %0"
                     : "+v"(acc), "=&v"(d0), "=&v"(d1) : "v"(AE), "v"(BE), "v"(AO), "v"(BO), "s"(sel), "s"(m), "s"(due));
    }
    if constexpr (PH == 4) {
        V0 = __builtin_amdgcn_perm(O, E, 0x05040100u);       // (e of butterfly 0, o of butterfly 0)
        V1 = __builtin_amdgcn_perm(O, E, 0x07060302u);       // (e of butterfly 1, o of butterfly 1)
    } else { V0 = E; V1 = O; }
    if (__builtin_expect((due & 0x0001000100010001ull) != 0ull, 0)) fwd4_renorm(V0, V1, due);
}

// five steps, one of each phase, on staging entries E0 .. E0+4 (phase = entry mod 5)
struct Fwd4Lane { uint32_t ofs[5]; };       // LDS byte offset of this lane's entry inside its row's table, by phase

template <int E0, int J0, typename Flush>
__device__ __forceinline__ void fwd4_group(uint32_t &V0, uint32_t &V1, const uint8_t *bml, const Fwd4Lane &c, uint32_t &acc, const Flush &flush)
{
    auto inc = [&](int e, uint32_t ofs) { return *(const uint2 *)(bml + 32 * e + ofs); };
    const uint2 w0 = inc(E0 + 0, c.ofs[0]), w1 = inc(E0 + 1, c.ofs[1]), w2 = inc(E0 + 2, c.ofs[2]), w3 = inc(E0 + 3, c.ofs[3]), w4 = inc(E0 + 4, c.ofs[4]);
    fwd4_step<0, J0 + 0>(V0, V1, w0, acc);
    if constexpr (((J0 + 0) & 7) == 7) flush((J0 + 0) >> 3);
    fwd4_step<1, J0 + 1>(V0, V1, w1, acc);
    if constexpr (((J0 + 1) & 7) == 7) flush((J0 + 1) >> 3);
    fwd4_step<2, J0 + 2>(V0, V1, w2, acc);
    if constexpr (((J0 + 2) & 7) == 7) flush((J0 + 2) >> 3);
    fwd4_step<3, J0 + 3>(V0, V1, w3, acc);
    if constexpr (((J0 + 3) & 7) == 7) flush((J0 + 3) >> 3);
    fwd4_step<4, J0 + 4>(V0, V1, w4, acc);
    if constexpr (((J0 + 4) & 7) == 7) flush((J0 + 4) >> 3);
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(64 * kFwd4Waves) void k_viterbi_fwd4(const FrameInfo *__restrict__ info, int n_frames, const uint16_t *__restrict__ sp,
                                                                  uint64_t *__restrict__ dec, int64_t cap)
{
    __shared__ __attribute__((aligned(16))) uint8_t bml_all[kFwd4Waves][4 * kChunk4 * 32];      // per row a chunk's 40 entries of 32 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, l = lane & 15;
    const int f0 = 4 * (blockIdx.x * kFwd4Waves + wave), f = f0 + row;
    if (f0 >= n_frames) return;
    FrameInfo fi = info[f < n_frames ? f : f0];
    const int T_r = (f < n_frames && fi.nsym > 0) ? fi.nsteps : 0;
    const int N_r = max(T_r - 6, 0), Ntop_r = (N_r + kChunk4 - 1) / kChunk4 * kChunk4;
    int N = 0, Nmin = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < 4; r++) { const int v = __builtin_amdgcn_readlane(N_r, 16 * r); N = max(N, v); Nmin = min(Nmin, v); }
    if (N == 0) return;
    // scalar base: the region of the wave's first frame that has one; a lane adds its own frame's distance from it (rows without a frame: none).
    // Loads and stores are buffer instructions as in viterbi_v3.h; a load beyond the buffers' end (the look-ahead of the last frame) returns zero.
    int r_first = 3;
#pragma unroll
    for (int r = 3; r >= 0; r--) if (__builtin_amdgcn_readlane(T_r, 16 * r) > 0) r_first = r;
    const int64_t off0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(fi.dec_off >> 32), 16 * r_first) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)fi.dec_off, 16 * r_first));
    const int rel = T_r > 0 ? (int)(fi.dec_off - off0) : 0;                 // (frames of a wave are consecutive: this is small)
    constexpr int kRsrcFlags = 0x00020000;
    const uint64_t left = (uint64_t)(cap - off0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(sp + off0), 0, (int)(uint32_t)min(2 * left, (uint64_t)0xFFFFFFFFull), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(dec + off0), 0, (int)(uint32_t)min(8 * left, (uint64_t)0xFFFFFFFFull), kRsrcFlags);
    const int voff_ld = 2 * (rel + l);                                      // soft pair of step t0 + l (+16, +32) of this row's frame
    const int voff_st = 8 * rel + 4 * (15 - v4_cidx(l));

    uint8_t *bml = bml_all[wave];
    const uint32_t bml_row = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)bml + (uint32_t)row * (kChunk4 * 32);
    Fwd4Lane c;
#pragma unroll
    for (int ph = 0; ph < 5; ph++) c.ofs[ph] = (uint32_t)row * (kChunk4 * 32) + 8u * ((kCls4[ph] >> (2 * l)) & 3u);

    // Soft pairs (s0 | s1 << 8, 2 bytes per step, as the front end left them) of steps t0 + l, t0 + 16 + l, t0 + 32 + l of the row's frame, fetched
    // one chunk ahead; put() turns them into the step's table: for each Branchtab class c the pair { W, W ^ 0x003F003F }, W = (m[c], m[c ^ a(phase)])
    // with m[c] the branch metric of viterbi.cpp:242-247 for class c (viterbi_v3.h spells the four out).
    uint32_t pa[3] = { 0u, 0u, 0u };
    auto get = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 3; i++) pa[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff_ld + 32 * i, 2 * t0, 0);
    };
    // entry j of the table gets phase (j + shift) mod 5
    auto put = [&](int cnt, int shift) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int j = 16 * i + l;
            if (j < cnt) {
                const uint32_t s0 = pa[i] & 255u, s1 = pa[i] >> 8;
                const uint32_t P = s0 + s1, D = s0 - s1;
                const uint32_t u01 = (D << 16) + P + 0x01000001u;            // (P + 1, D + 256)
                const uint32_t u23 = 0x01FF0100u - ((P << 16) + D);          // (256 - D, 511 - P)
                const uint32_t M01 = (u01 >> 3) & 0x003F003Fu, M23 = (u23 >> 3) & 0x003F003Fu;
                const uint32_t a = (kA4 >> (2 * ((j + shift) % 5))) & 3u;
                const uint32_t K = a * 0x02020000u;
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const uint32_t addr = bml_row + 32u * (uint32_t)j;
#pragma unroll
                for (uint32_t cc = 0; cc < 4; cc += 2) {
                    const uint32_t w0 = __builtin_amdgcn_perm(M23, M01, (0x01000100u + cc * 0x02020202u) ^ K);
                    const uint32_t w1 = __builtin_amdgcn_perm(M23, M01, (0x01000100u + (cc + 1) * 0x02020202u) ^ K);
                    const u32x4 v = { w0, w0 ^ 0x003F003Fu, w1, w1 ^ 0x003F003Fu };
                    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(8 * cc) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_lds_sync();
    };

    uint32_t V0 = l == 0 ? kBias2 + 0x003F0000u : kBias2 + 0x003F003Fu, V1 = kBias2 + 0x003F003Fu;       // viterbi.cpp:71-78: state 0 = 0, the others 63
    uint32_t acc = 0u;
    auto noflush = [](int) {};
    // trellis steps 0..5 carry no data bit (viterbi.cpp:131-142): phases 4, 0, 1, 2, 3, 4, so that data step 0 has phase 0
    get(0);
    put(6, 4);
    get(6);
    {
        auto inc = [&](int e, uint32_t ofs) { return *(const uint2 *)(bml + 32 * e + ofs); };
        fwd4_step<4, -1>(V0, V1, inc(0, c.ofs[4]), acc);
        fwd4_step<0, -1>(V0, V1, inc(1, c.ofs[0]), acc);
        fwd4_step<1, -1>(V0, V1, inc(2, c.ofs[1]), acc);
        fwd4_step<2, -1>(V0, V1, inc(3, c.ofs[2]), acc);
        fwd4_step<3, -1>(V0, V1, inc(4, c.ofs[3]), acc);
        fwd4_step<4, -1>(V0, V1, inc(5, c.ofs[4]), acc);
    }
    // A sub-block's dword leaves as soon as its eight steps are through -- except a chunk's last, which is stored one chunk LATE, behind the loads
    // of the chunk after (vector loads and stores share one counter: viterbi_v3.h).
    auto store_sub = [&](int n0, uint32_t w) {             // n0: first data step of the sub-block
        if (n0 < Ntop_r) {
            uint32_t word = w;
            if (n0 + 8 > Nmin) {                           // (wave-uniform: only near a frame's end)
                const int v = min(max(N_r - n0, 0), 8);
                word |= ((0xFFu << v) & 0xFFu) * 0x01010101u;
            }
            __builtin_amdgcn_raw_buffer_store_b32(word, rd, voff_st, 8 * n0, 0);
        }
    };
    uint32_t late = 0u;
    int n_chunk = 0;
    auto flush = [&](int sb) {
        if (sb < 4) store_sub(n_chunk + 8 * sb, acc);
        else late = acc;
    };
    (void)noflush;
    for (int n0 = 0; n0 < N; n0 += kChunk4) {
        put(kChunk4, 0);
        get(n0 + kChunk4 + 6);
        if (n0 > 0) store_sub(n0 - 8, late);
        n_chunk = n0;
        fwd4_group<0, 0>(V0, V1, bml, c, acc, flush);   fwd4_group<5, 5>(V0, V1, bml, c, acc, flush);   fwd4_group<10, 10>(V0, V1, bml, c, acc, flush);
        fwd4_group<15, 15>(V0, V1, bml, c, acc, flush); fwd4_group<20, 20>(V0, V1, bml, c, acc, flush); fwd4_group<25, 25>(V0, V1, bml, c, acc, flush);
        fwd4_group<30, 30>(V0, V1, bml, c, acc, flush); fwd4_group<35, 35>(V0, V1, bml, c, acc, flush);
    }
    store_sub((N - 1) / kChunk4 * kChunk4 + 32, late);
}

// v4 decision layout -> viterbi_v3.h's (u16 [16-step block][63 - slot]), so that k_tb_walk / k_tb_finish read it unchanged: bring-up and cross-check only.
__global__ __launch_bounds__(64) void k_dec4_to_dec3(const FrameInfo *__restrict__ info, int n_frames, const uint64_t *__restrict__ dec4, uint64_t *__restrict__ dec3)
{
    const int f = blockIdx.y, p = threadIdx.x;
    if (f >= n_frames) return;
    const FrameInfo fi = info[f];
    if (fi.nsym <= 0) return;
    const int N = fi.nsteps - 6, Ntop = (N + kChunk3 - 1) / kChunk3 * kChunk3;
    const uint8_t *src = (const uint8_t *)(dec4 + fi.dec_off);
    uint16_t *dst = (uint16_t *)(dec3 + fi.dec_off);
    for (int b = blockIdx.x; 16 * b < Ntop; b += gridDim.x) {
        uint32_t word = 0;
        for (int j = 0; j < 16; j++) {
            const int n = 16 * b + j;
            uint32_t bit = 1u;
            if (n < N) {
                const int s = rotl6(p, (n + 7) % 6), ph = n % 5;                 // the label viterbi_v3.h has in slot p after data step n
                int u[6];
                for (int k = 0; k < 6; k++) u[k] = (s >> v4_label_bit(ph, k)) & 1;
                const int x = u[0] | (u[1] << 1) | (u[2] << 2) | (u[3] << 3), slot = 2 * u[4] + u[5];
                bit = (src[64 * (n >> 3) + 4 * (15 - x) + (3 - slot)] >> (n & 7)) & 1u;
            }
            word |= bit << j;
        }
        dst[(size_t)b * 64 + 63 - p] = (uint16_t)word;
    }
}

inline void launch_fwd4(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec, size_t cap)
{
    hipLaunchKernelGGL(k_viterbi_fwd4, dim3(((nf + 3) / 4 + kFwd4Waves - 1) / kFwd4Waves), dim3(64 * kFwd4Waves), 0, st, info, nf, sp, dec, (int64_t)cap);
}
inline void launch_dec4_to_dec3(hipStream_t st, const FrameInfo *info, int nf, const uint64_t *dec4, uint64_t *dec3)
{
    hipLaunchKernelGGL(k_dec4_to_dec3, dim3(16, (unsigned)nf), dim3(64), 0, st, info, nf, dec4, dec3);
}

}  // namespace foa
