"""Oracle against the committed golden vectors (tests/golden/*.npz).

The vectors were produced by the reference's own arithmetic_code.h (see
tests/golden/make_golden.py); this file needs neither /root/reference nor oracle/_ref.
"""
import os

import numpy as np

import oracle_lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_g1_half(oracle):
    g = load("g1_half.npz")
    for i in range(5):
        n = int(g[f"n_{i}"])
        bins = np.unpackbits(g[f"bins_{i}"])[:n]
        want = g[f"bytes_{i}"].tobytes()
        assert oracle.half_encode(bins) == want
        assert np.array_equal(oracle.half_decode(want, n), bins)


def test_g2_range(oracle):
    g = load("g2_range.npz")
    for i in range(int(g["n_cases"])):
        recs = g[f"recs_{i}"]
        want = g[f"bytes_{i}"].tobytes()
        got, st = oracle.range_encode(recs)
        assert st == 0 and got == want, f"case {i}"
        extra = np.concatenate([recs, np.full(64, (1 << 1) | (1 << 8), dtype=np.uint16)])
        dec = oracle.range_decode(want, extra)
        assert np.array_equal(dec, np.unpackbits(g[f"decoded_past_end_{i}"])[:extra.size]), f"case {i}"


def test_g3_cabac(oracle):
    g = load("g3_cabac.npz")
    for i in range(int(g["n_cases"])):
        got, final, st = oracle.cabac_encode(g[f"recs_{i}"], g[f"states_{i}"])
        assert st == 0, f"case {i}"
        assert got == g[f"bytes_{i}"].tobytes(), f"case {i}"
        assert final == g[f"final_{i}"].tobytes(), f"case {i}"


def test_g3_spec_encoder_agrees_where_the_slice_is_terminated(oracle):
    """Second oracle (H.264 9.3.4.2 bit-serial encoder) on the same vectors."""
    g = load("g3_cabac.npz")
    checked = 0
    for i in range(int(g["n_cases"])):
        recs = g[f"recs_{i}"]
        if recs.size == 0 or recs[-1] != (1 | (1025 << 1)):
            continue        # the standard only defines the byte string of a terminated slice
        got, final, st = oracle.spec_cabac_encode(recs, g[f"states_{i}"])
        assert st == 0 and got == g[f"bytes_{i}"].tobytes() and final == g[f"final_{i}"].tobytes(), f"case {i}"
        bins, _ = oracle.spec_cabac_decode(got, recs, g[f"states_{i}"])
        assert np.array_equal(bins, recs & 1), f"case {i}"
        checked += 1
    assert checked >= 12


def test_g4_tail_cases(oracle):
    """recode.cpp:1508-1512 and :1354-1360 on golden CABAC outputs: both endings, both parities."""
    g = load("g3_cabac.npz")
    seen = set()
    for i in range(int(g["n_cases"])):
        raw = g[f"bytes_{i}"].tobytes()
        dropped = oracle.drop_stop_byte(raw)
        assert dropped == (raw[:-1] if raw.endswith(b"\x80") else raw)
        seen.add((raw.endswith(b"\x80"), len(dropped) & 1))
        for parity in (0, 1):
            patched = oracle.tail_patch(dropped, parity, 0xAB)
            if parity != (len(dropped) & 1):
                assert patched == dropped + b"\xab"
            elif dropped:
                assert patched == dropped[:-1] + b"\xab"
        assert oracle.tail_patch(dropped, -1, 0xAB) == dropped
    assert len(seen) == 4, seen
