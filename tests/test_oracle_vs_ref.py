"""Pin the oracle (oracle/avr_oracle.c) against the reference's own range coder.

oracle/_ref is /root/reference/arithmetic_code.h compiled unmodified (oracle/ref_harness.cpp);
these tests run wherever that library exists (it is built here, and travels prebuilt to the
GPU box).  They follow the reference's only active unit test, test/arithmetic_code.cpp:93-111
(random bits, p = 1/2, encode -> decode -> compare), and extend it to the two instantiations
recode.cpp and cabac_code.h use.
"""
import numpy as np
import pytest

import oracle_lib


def test_half_coder_matches_reference(oracle, ref):
    # test/arithmetic_code.cpp:93-111 with fixed seeds instead of time(nullptr) (:49)
    for seed, n in enumerate((0, 1, 2, 15, 16, 17, 1000, 4097, 50000)):
        bins = np.random.default_rng(seed).integers(0, 2, n).astype(np.uint8)
        want = ref.half_encode(bins)
        assert oracle.half_encode(bins) == want
        assert np.array_equal(oracle.half_decode(want, n), bins)
        assert np.array_equal(ref.half_decode(want, n), bins)


@pytest.mark.parametrize("adaptive", [False, True])
def test_recoded_coder_matches_reference(oracle, ref, adaptive):
    rng = np.random.default_rng(7 + adaptive)
    for _ in range(150):
        n = int(rng.integers(0, 1500))
        recs = oracle_lib.random_range_stream(rng, n, adaptive=adaptive)
        want, st = ref.range_encode(recs)
        got, st2 = oracle.range_encode(recs)
        assert st == 0 and st2 == 0
        assert got == want
        # decoder, including reads past the end of the bytes (arithmetic_code.h:283-285)
        extra = np.concatenate([recs, np.full(40, (3 << 1) | (5 << 8), dtype=np.uint16)])
        assert np.array_equal(oracle.range_decode(want, extra), ref.range_decode(want, extra))
        assert np.array_equal(oracle.range_decode(want, recs), recs & 1)


def test_recoded_zero_probability_is_reported(oracle, ref):
    # pos = 0 makes range_of_1 zero: coding a 1 must fail like arithmetic_code.h:116-118 throws
    recs = np.array([1 | (0 << 1) | (5 << 8)], dtype=np.uint16)
    _, st_ref = ref.range_encode(recs)
    _, st = oracle.range_encode(recs)
    assert st_ref == 1 and st == oracle_lib.ctypes.c_int(1).value


def test_cabac_instantiation_matches_reference_coder(oracle, ref):
    rng = np.random.default_rng(11)
    for _ in range(400):
        n = int(rng.integers(0, 800))
        nctx = int(rng.integers(1, 200))
        recs, states = oracle_lib.random_cabac_stream(rng, n, nctx, terminate=bool(rng.integers(0, 2)))
        want = ref.cabac_encode(recs, states)
        assert oracle.cabac_encode(recs, states) == want
