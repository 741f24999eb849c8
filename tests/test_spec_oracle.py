"""The restated cabac::encoder emits the byte string the H.264 standard defines.

Mirrors the reference's disabled test (test/arithmetic_code.cpp:66-91): encode with
cabac::encoder over a few contexts, decode with an independent CABAC decoder, compare.
There the decoder is libavcodec's ff_get_cabac; here it is oracle/spec_cabac.c, written
from H.264 9.3.3.2 / 9.3.4.2.
"""
import numpy as np

import oracle_lib


def test_encoder_equals_standard_and_roundtrips(oracle):
    rng = np.random.default_rng(2024)
    dropped_differs = 0
    for t in range(1500):
        n = int(rng.integers(0, 600))
        nctx = int(rng.integers(1, 64))
        recs, states = oracle_lib.random_cabac_stream(rng, n, nctx)
        got = oracle.cabac_encode(recs, states)
        assert got[2] == 0
        assert oracle.spec_cabac_encode(recs, states) == got
        bins, final = oracle.spec_cabac_decode(got[0], recs, states)
        assert np.array_equal(bins, recs & 1)
        assert final == got[1]
        dropped_differs += oracle.drop_stop_byte(got[0]) != got[0]
    # roughly one stream in eight ends in a bare stop byte
    assert 0 < dropped_differs < 1500


def test_five_context_case_of_the_reference_test(oracle):
    # test/arithmetic_code.cpp:51-72: five contexts with random bias, states start at 0
    rng = np.random.default_rng(5)
    prob = rng.integers(0, 100, 5)
    ctx = rng.integers(0, 5, 20000)
    bits = (rng.integers(0, 100, 20000) > prob[ctx]).astype(np.uint16)
    recs = np.concatenate([(bits | (ctx << 1)).astype(np.uint16), np.array([1 | (1025 << 1)], dtype=np.uint16)])
    states = np.zeros(0x400, dtype=np.uint8)
    data, final, st = oracle.cabac_encode(recs, states)
    assert st == 0
    bins, final2 = oracle.spec_cabac_decode(data, recs, states)
    assert np.array_equal(bins, recs & 1) and final == final2
