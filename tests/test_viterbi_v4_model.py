"""tools/viterbi_v4_model.py -- the position / label bookkeeping csrc/viterbi_v4.h hard-codes (a cross-check kernel) -- against the oracle's scalar
Viterbi: the decisions of every step mapped back to state labels, and the decoded bits of a chain-back over the kernel's decision layout."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_model_of_the_four_states_per_lane_forward_pass_equals_the_oracle():
    import viterbi_v4_model as m
    assert m.check_against_oracle(n_blocks=8, seed=5, max_pairs=150) == 8


def test_tables_of_the_kernel_are_the_models(capsys):
    import re
    import viterbi_v4_model as m
    m.tables()
    out = capsys.readouterr().out
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fun_ofdm_amd", "csrc", "viterbi_v4.h")).read()
    cls = re.search(r"kCls4 = (.*)", out).group(1).split(", ")
    for c in cls:
        assert c + "u" in src, c
    assert re.search(r"kA4   = (0x[0-9a-f]+)", out).group(1) + "u" in src
