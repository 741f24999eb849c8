#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own range coder.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py

Every expected output below comes from oracle/_ref/libavr_ref.so, i.e. from
/root/reference/arithmetic_code.h compiled unmodified (oracle/ref_harness.cpp,
oracle/Makefile).  For the CABAC cases the reference coder is driven through the CABAC
layer restated in ref_harness.cpp, because cabac_code.h needs a libavcodec header the
reference snapshot does not contain.  The files hold data only: inputs (bins / records /
initial states) and the bytes, bins and final states the reference produced.

Groups:
  g1_half.npz    arithmetic_code<uint64,uint16>, p = 1/2 (test/arithmetic_code.cpp:93-111)
  g2_range.npz   recoded_code = arithmetic_code<uint64,uint8> with (pos,neg) records,
                 encode and decode (incl. reading past the end of the bytes)
  g3_cabac.npz   arithmetic_code<uint32,uint16,0x200> + CABAC layer: the disabled regression
                 input of test/arithmetic_code.cpp:16-34, carry chains, empty slice, all
                 states x range quarters, random mixes
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402


def rec(bin_, sel):
    return np.uint16(bin_ | (sel << 1))


def main():
    ref = oracle_lib.load_ref()
    assert ref is not None, "needs /root/reference to build oracle/_ref"

    # ---------------------------------------------------------------- G1
    g1 = {}
    for i, n in enumerate((0, 1, 7, 1000, 100000)):
        rng = np.random.default_rng(0xA100 + i)
        bins = rng.integers(0, 2, n).astype(np.uint8)
        data = ref.half_encode(bins)
        assert np.array_equal(ref.half_decode(data, n), bins)
        g1[f"bins_{i}"] = np.packbits(bins)
        g1[f"n_{i}"] = np.int64(n)
        g1[f"bytes_{i}"] = np.frombuffer(data, dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g1_half.npz"), **g1)

    # ---------------------------------------------------------------- G2
    g2 = {}
    cases = []
    for i, (n, adaptive) in enumerate(((0, False), (1, False), (50, False), (5000, False), (3000, True), (20000, True))):
        rng = np.random.default_rng(0xA200 + i)
        cases.append(oracle_lib.random_range_stream(rng, n, adaptive=adaptive))
    # extremes: most skewed estimator both ways, long runs (carry chains of 0xFF bytes)
    cases.append(np.array([1 | (0x5f << 1) | (1 << 8)] * 4000, dtype=np.uint16))      # always the likely symbol
    cases.append(np.array([0 | (0x5f << 1) | (1 << 8)] * 300, dtype=np.uint16))       # always the unlikely one
    cases.append(np.array([1 | (1 << 1) | (0x5f << 8)] * 300, dtype=np.uint16))
    cases.append(np.array([(i & 1) | (1 << 1) | (1 << 8) for i in range(999)], dtype=np.uint16))
    for i, recs in enumerate(cases):
        data, status = ref.range_encode(recs)
        assert status == 0
        extra = np.concatenate([recs, np.full(64, (1 << 1) | (1 << 8), dtype=np.uint16)])   # read past the end
        dec = ref.range_decode(data, extra)
        assert np.array_equal(dec[:recs.size], recs & 1)
        g2[f"recs_{i}"] = recs
        g2[f"bytes_{i}"] = np.frombuffer(data, dtype=np.uint8)
        g2[f"decoded_past_end_{i}"] = np.packbits(dec)
    g2["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "g2_range.npz"), **g2)

    # ---------------------------------------------------------------- G3
    g3 = {}
    ccases = []
    # (a) test/arithmetic_code.cpp:16-34: every put() there uses a fresh `state` byte, so each bin gets
    # its own context initialised to the listed value; terminate(false) after the first bin; 16 zero
    # bins at state 0; terminate(true).
    states = [15, 17, 106, 28, 16, 0, 10, 26, 33, 22, 35, 58, 44, 0, 0, 1, 3, 5]
    bits = [1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1]
    recs = [rec(bits[0], 0), rec(0, 1025)] + [rec(bits[i], i) for i in range(1, 18)]
    recs += [rec(0, 18 + i) for i in range(16)] + [rec(1, 1025)]
    ccases.append((np.array(recs, dtype=np.uint16), np.array(states + [0] * 16, dtype=np.uint8)))
    # (b) empty slice: terminate only; and no records at all (finish from the destructor)
    ccases.append((np.array([rec(1, 1025)], dtype=np.uint16), np.zeros(4, dtype=np.uint8)))
    ccases.append((np.zeros(0, dtype=np.uint16), np.zeros(4, dtype=np.uint8)))
    # (c) carry chains: all-MPS at pStateIdx 62, bypass-only ones and zeros, alternating
    ccases.append((np.array([rec(1, 0)] * 5000 + [rec(1, 1025)], dtype=np.uint16), np.array([125], dtype=np.uint8)))
    ccases.append((np.array([rec(0, 0)] * 5000 + [rec(1, 1025)], dtype=np.uint16), np.array([124], dtype=np.uint8)))
    ccases.append((np.array([rec(1, 1024)] * 3000 + [rec(1, 1025)], dtype=np.uint16), np.zeros(1, dtype=np.uint8)))
    ccases.append((np.array([rec(0, 1024)] * 3000 + [rec(1, 1025)], dtype=np.uint16), np.zeros(1, dtype=np.uint8)))
    ccases.append((np.array([rec(i & 1, 1024) for i in range(3001)] + [rec(1, 1025)], dtype=np.uint16), np.zeros(1, dtype=np.uint8)))
    # all-LPS: the worst case for output size (AVR "output sizing")
    ccases.append((np.array([rec(0, 0)] * 2000 + [rec(1, 1025)], dtype=np.uint16), np.array([125], dtype=np.uint8)))
    ccases.append((np.array([rec(1, 1025)] * 1, dtype=np.uint16), np.zeros(0, dtype=np.uint8)))
    # (d) every state x every range quarter: one context per state value, visited in random order
    rng = np.random.default_rng(0xA300)
    sel = rng.integers(0, 126, 60000)
    bins = rng.integers(0, 2, 60000)
    byp = rng.random(60000) < 0.3                       # bypass bins move the range quarter around
    recs = np.where(byp, (bins | (1024 << 1)), (bins | (sel << 1))).astype(np.uint16)
    ccases.append((np.concatenate([recs, [rec(1, 1025)]]).astype(np.uint16), np.arange(126, dtype=np.uint8)))
    # (e) random mixes of several sizes, incl. full 1024-state tables and streams without a final terminate
    for i, (n, nctx, term) in enumerate(((5, 3, True), (100, 16, True), (4000, 64, True), (30000, 460, True),
                                          (20000, 1024, True), (777, 40, False))):
        rng = np.random.default_rng(0xA310 + i)
        ccases.append(oracle_lib.random_cabac_stream(rng, n, nctx, terminate=term))
    for i, (recs, st) in enumerate(ccases):
        data, final, status = ref.cabac_encode(recs, st)
        assert status == 0
        g3[f"recs_{i}"] = recs
        g3[f"states_{i}"] = st
        g3[f"bytes_{i}"] = np.frombuffer(data, dtype=np.uint8)
        g3[f"final_{i}"] = np.frombuffer(final, dtype=np.uint8)
    g3["n_cases"] = np.int64(len(ccases))
    np.savez_compressed(os.path.join(HERE, "g3_cabac.npz"), **g3)

    for f in ("g1_half.npz", "g2_range.npz", "g3_cabac.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
