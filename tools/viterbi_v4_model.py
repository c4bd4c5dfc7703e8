#!/usr/bin/env python3
"""Model of the forward pass of fun_ofdm_amd/csrc/viterbi_v4.h -- four states per lane, a frame per 16-lane row, results NOT written back in
place -- and of a chain-back over its decision layout, checked against the oracle's scalar Viterbi (oracle/fo_oracle.c fo_viterbi_forward,
viterbi.cpp:208-457): the decisions of every step, mapped back to state labels, and the decoded bits must be identical.

    python3 tools/viterbi_v4_model.py            random / saturating / noisy blocks against the oracle
    python3 tools/viterbi_v4_model.py tables     the constants viterbi_v4.h hard-codes (class of the low butterfly per lane and phase, a(phase), label maps)
    python3 tools/viterbi_v4_model.py cycles     the search that found the period-5 schedule (one step in five without an exchange)

Positions: lane coordinates c0..c3 (lane-in-row = c0 ^ 2 c1 ^ 7 c2 ^ 8 c3: the xor masks single DPP moves reach), register bit g, half bit h.
A step pairs the positions that differ in the position bit holding label bit 5; kinds of step:
  G   pairs on g: X = V0, Y = V1, E -> V0, O -> V1                          (nothing moves; g takes the new low label bit)
  EO  pairs on a lane coordinate: the lower lane works on both lanes' V0, the upper on both lanes' V1; E -> V0, O -> V1
      (g takes the new low bit, the lane coordinate inherits g's label)
  D   as EO, then (E, O) transposed so that h takes the new low bit, g inherits h's label, the lane coordinate g's
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pyoracle as po

MASK = [1, 2, 7, 8]                 # xor mask of lane coordinate c0..c3
PHASES = [('G', None), ('EO', 3), ('EO', 2), ('EO', 1), ('D', 0)]
LAB0 = [1, 2, 3, 4, 5, 0]           # labels of (c0,c1,c2,c3,g,h) before phase 0

def par(x): return bin(x).count('1') & 1
def cls(i): return (par((2 * i) & 121) << 1) | par((2 * i) & 91)
def lane_of(c): 
    l = 0
    for k in range(4):
        if c[k]: l ^= MASK[k]
    return l
COORDS = {}
for c0 in range(2):
    for c1 in range(2):
        for c2 in range(2):
            for c3 in range(2):
                COORDS[lane_of((c0, c1, c2, c3))] = (c0, c1, c2, c3)
assert len(COORDS) == 16

def label(lab, lane, g, h):
    c = COORDS[lane] + (g, h)
    return sum(c[k] << lab[k] for k in range(6))

def metrics4(s0, s1):
    out = []
    for c in range(4):
        b0, b1 = (c >> 1) * 255, (c & 1) * 255
        out.append(((((s0 ^ b0) + (s1 ^ b1) + 1) >> 1) >> 2) & 63)
    return out

def forward(sym, nsteps, phase0=0, record=None):
    """returns (decisions by label [nsteps] as python ints (oracle format), decisions by position [t][lane][slot], labels list per t (AFTER step t))"""
    V = np.full((16, 2, 2), 63, np.int64); V[0, 0, 0] = 0
    lab = list(LAB0)
    # rotate the cycle so that trellis step 0 has phase phase0
    for ph in range(phase0):
        lab = advance(lab, PHASES[ph])
    run = 2 * (nsteps // 2)
    dec_lab = [0] * nsteps; dec_pos = []; labs = []
    for t in range(run):
        ph = (phase0 + t) % 5
        kind, cq = PHASES[ph]
        m = metrics4(int(sym[2 * t]), int(sym[2 * t + 1]))
        assert lab.index(5) == (4 if kind == 'G' else cq), (lab, kind, cq)
        newV = np.zeros_like(V); newlab = advance(lab, PHASES[ph]); dp = np.zeros((16, 4), np.int64); dl = 0
        for lane in range(16):
            if kind == 'G':
                X, Y = V[lane, 0], V[lane, 1]; xl = [label(lab, lane, 0, k) for k in range(2)]; yl = [label(lab, lane, 1, k) for k in range(2)]
            else:
                side = COORDS[lane][cq]; p = lane ^ MASK[cq]
                if side == 0:
                    X, Y = V[lane, 0], V[p, 0]; xl = [label(lab, lane, 0, k) for k in range(2)]; yl = [label(lab, p, 0, k) for k in range(2)]
                else:
                    X, Y = V[p, 1], V[lane, 1]; xl = [label(lab, p, 1, k) for k in range(2)]; yl = [label(lab, lane, 1, k) for k in range(2)]
            for k in range(2):
                i = xl[k]; assert i < 32 and yl[k] == i + 32
                c = cls(i); mm = m[c]; mc = 63 - mm
                a0, a1 = min(255, X[k] + mm), min(255, Y[k] + mc)
                b0, b1 = min(255, X[k] + mc), min(255, Y[k] + mm)
                de, e = (1, a1) if a1 <= a0 else (0, a0)
                do, o = (1, b1) if b1 <= b0 else (0, b0)
                if kind == 'D': ge, he, go, ho = k, 0, k, 1
                else: ge, he, go, ho = 0, k, 1, k
                newV[lane, ge, he] = e; newV[lane, go, ho] = o
                assert label(newlab, lane, ge, he) == 2 * i and label(newlab, lane, go, ho) == 2 * i + 1
                dp[lane, 2 * ge + he] = de; dp[lane, 2 * go + ho] = do
                dl |= (de << (2 * i)) | (do << (2 * i + 1))
        V = newV; lab = newlab
        if V[0, 0, 0] > 210: V = V - V.min()
        dec_lab[t] = dl; dec_pos.append(dp); labs.append(list(lab))
    return dec_lab, dec_pos, labs, V, lab

def advance(lab, phase):
    kind, cq = phase
    new = [x + 1 for x in lab]
    if kind == 'G': new[4] = 0
    elif kind == 'EO': new[4] = 0; new[cq] = lab[4] + 1
    else: new[5] = 0; new[4] = lab[5] + 1; new[cq] = lab[4] + 1
    assert sorted(new) == list(range(6)), (lab, phase, new)
    return new

def chainback(dec_pos, nsteps, data_bits, phase0=0):
    """walk in position space: pos = (lane, g, h); returns bits (data bit n from decision at trellis step n+6)"""
    lane, g, h = 0, 0, 0                      # state 0 at the end
    bits = np.zeros(data_bits, np.uint8)
    for t in range(data_bits + 6 - 1, 5, -1):
        d = int(dec_pos[t][lane, 2 * g + h])
        bits[t - 6] = d
        kind, cq = PHASES[(phase0 + t) % 5]
        c = list(COORDS[lane])
        if kind == 'G': g = d
        elif kind == 'EO': g_old = c[cq]; c[cq] = d; g = g_old
        else: h_old = g; g_old = c[cq]; c[cq] = d; g = g_old; h = h_old
        lane = lane_of(c)
    return bits



def check_against_oracle(n_blocks=12, seed=1, max_pairs=400):
    rng = np.random.default_rng(seed)
    for it in range(n_blocks):
        nb = int(rng.integers(10, max_pairs)) * 2
        n = nb + 6
        mode = it % 4
        if mode == 0:
            s = rng.integers(0, 256, 2 * n, dtype=np.uint8)
        elif mode == 1:
            s = rng.choice(np.array([0, 255, 127], np.uint8), 2 * n)
        else:
            d = rng.integers(0, 256, (nb + 13) // 8 + 1, dtype=np.uint8)
            e = po.conv_encode(d, nb).astype(float) * 255
            s = np.clip(e + rng.normal(0, 60 * mode, e.size), 0, 255).astype(np.uint8)
        ref_dec, _, _ = po.viterbi_forward(s, n)
        for ph0 in (0, 4):                       # (the kernel runs trellis step 0 in phase 4, so that data step 0 has phase 0)
            dl, dp, labs, V, lab = forward(s, n, ph0)
            assert all(int(ref_dec[t]) == dl[t] for t in range(n)), it
            bits = chainback(dp, n, nb, ph0)
            assert np.array_equal(bits, np.unpackbits(po.conv_decode(s, nb))[:nb]), (it, ph0)
    return n_blocks


def tables():
    lab = list(LAB0)
    A, C, LAM = [], [], []
    for ph in range(5):
        kind, cq = PHASES[ph]
        a_all, cl = set(), []
        for lane in range(16):
            if kind == 'G':
                xl = [label(lab, lane, 0, k) for k in range(2)]
            else:
                side = COORDS[lane][cq]
                p = lane ^ MASK[cq]
                xl = [label(lab, lane, 0, k) for k in range(2)] if side == 0 else [label(lab, p, 1, k) for k in range(2)]
            a_all.add(cls(xl[0]) ^ cls(xl[1]))
            cl.append(cls(xl[0]))
        assert len(a_all) == 1                   # the high butterfly's class differs from the low one's by the same a in every lane
        A.append(a_all.pop())
        C.append(cl)
        lab = advance(lab, PHASES[ph])
        LAM.append(list(lab))
    assert lab == LAB0                           # period 5
    print("kCls4 =", ", ".join(hex(sum(c << (2 * l) for l, c in enumerate(row))) for row in C))
    print("kA4   =", hex(sum(a << (2 * p) for p, a in enumerate(A))), " (a per phase:", A, ")")
    print("label bit of (c0, c1, c2, c3, g, h) AFTER the step of each phase:")
    for row in LAM:
        print("   ", row)
    print("coordinate index of lane-in-row:", [sum(COORDS[l][k] << k for k in range(4)) for l in range(16)])


def cycles():
    """Label schedules: a state is the label bit of each position bit; a free step needs g or h at label 5.  Lane steps reset g (EO) or h (D)."""
    import itertools

    def step(st, choice):
        q = st.index(5)
        if q >= 4:
            new = [x + 1 for x in st]
            new[q] = 0
            return tuple(new)
        new = [x + 1 for x in st]
        if choice == 'EO':
            new[4] = 0
            new[q] = st[4] + 1
        else:
            new[5] = 0
            new[4] = st[5] + 1
            new[q] = st[4] + 1
        return tuple(new)

    for perm in itertools.permutations(range(6)):
        st = tuple(perm)
        if st.index(5) < 4:
            continue
        for choices in itertools.product(('EO', 'D'), repeat=4):
            s, seq, k, ok = st, [], 0, True
            for n in range(5):
                q = s.index(5)
                if (q >= 4) != (n == 0):
                    ok = False
                    break
                if q >= 4:
                    s = step(s, None)
                    seq.append('free(%s)' % 'gh'[q - 4])
                else:
                    s = step(s, choices[k])
                    seq.append(choices[k] + str(q))
                    k += 1
            if ok and s == st:
                print(st, ' '.join(seq))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'tables':
        tables()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cycles':
        cycles()
    else:
        print(check_against_oracle(), "blocks: decisions of every step and decoded bits equal the oracle's")
