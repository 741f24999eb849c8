"""K2p (csrc/avr_k2p.h: the recoded range coder in three passes -- range recurrence, per-chunk coding into byte
sums, carries + finish) emulated on the CPU by tests/k2p_emul.cpp with the very functions the kernels run, against
the oracle.  Chunk sizes down to one bin force every boundary case (a chunk that emits nothing, left-overs of many
chunks on the same positions, carries across chunk ends)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avrecode-ms_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "k2p_emul.cpp")
SO = os.path.join(ROOT, "tests", "_k2p_emul.so")


@pytest.fixture(scope="module")
def emul():
    deps = [SRC, os.path.join(CSRC, "avr_k2p.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, "-o", SO, SRC], check=True)
    lib = ctypes.CDLL(SO)
    lib.k2p_emul_encode.restype = ctypes.c_size_t
    return lib


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load_oracle()


def emul_encode(emul, recs, chunk):
    recs = np.ascontiguousarray(recs, np.uint16)
    out = np.zeros(recs.size + 32, np.uint8)
    info = (ctypes.c_uint32 * 4)()
    n = emul.k2p_emul_encode(recs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(recs.size), ctypes.c_uint32(chunk),
                             out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size), info)
    return (None if n >= 2**63 else out[:n].tobytes()), list(info)


@pytest.mark.parametrize("chunk", [1, 2, 7, 64, 1024])
def test_chunked_range_coder_equals_the_oracle(emul, oracle, chunk):
    rng = np.random.default_rng(40 + chunk)
    for k in range(60):
        n = int(rng.integers(0, 5000))
        recs = oracle_lib.random_range_stream(rng, n, adaptive=bool(k % 3))
        want, status = oracle.range_encode(recs)
        got, info = emul_encode(emul, recs, chunk)
        assert status == 0 and got == want, f"stream {k} n={n} chunk={chunk} info={info}"
        assert info[3] == n                                  # the double-precision pass 1 walked every bin


def test_chunked_range_coder_extremes(emul, oracle):
    """Certain bins (pos or neg 0: nothing is emitted for as long as they last), the most lopsided estimators the update
    rule reaches, long runs of one value, and the empty stream."""
    def rec(b, pos, neg):
        return b | (pos << 1) | (neg << 8)
    cases = [
        np.zeros(0, np.uint16),
        np.array([rec(1, 0x5f, 1)] * 3000, np.uint16),                       # always the likely value: a byte every ~50 bins
        np.array([rec(0, 0x5f, 1)] * 700, np.uint16),                        # always the unlikely one: almost a byte per bin
        np.array([rec(1, 9, 0)] * 2500 + [rec(0, 1, 1)] * 10, np.uint16),    # certain bins: no output for 2500 bins
        np.array([rec(i & 1, 1, 1) for i in range(4000)], np.uint16),        # a bit per bin
        np.array([rec(0, 0x5f, 1), rec(1, 1, 0x5f)] * 900, np.uint16),
    ]
    for chunk in (1, 3, 256, 1024):
        for k, recs in enumerate(cases):
            want, status = oracle.range_encode(recs)
            got, info = emul_encode(emul, recs, chunk)
            assert status == 0 and got == want, f"case {k} chunk={chunk} info={info}"


def test_double_precision_range_recurrence_is_exact(emul):
    """Pass 1's double-precision form (range_step_fp, csrc/avr_k2p.h) against the 64-bit integer form over millions of bins:
    every (pos, neg) pair from 1 to 127 -- beyond what the estimator update of recode.cpp:1037-1052 can reach -- in random
    order, runs of the most lopsided pairs, and the adaptive streams; then records with pos or neg 0, which it must hand over."""
    emul.k2p_emul_fp_walk.restype = ctypes.c_size_t
    def walk(recs):
        recs = np.ascontiguousarray(recs, np.uint16)
        return emul.k2p_emul_fp_walk(recs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(recs.size)), recs.size
    rng = np.random.default_rng(7)
    n = 4_000_000
    pos, neg = rng.integers(1, 128, n), rng.integers(1, 128, n)
    for p1 in (0.5, 0.02, 0.98):
        got, size = walk((rng.random(n) < p1).astype(np.uint16) | (pos << 1) | (neg << 8))
        assert got == size
    lop = np.where(rng.random(n) < 0.5, 1 | (1 << 1) | (127 << 8), 0 | (127 << 1) | (1 << 8)).astype(np.uint16)     # 7 bits a bin
    assert walk(lop)[0] == n
    for k in range(20):
        recs = oracle_lib.random_range_stream(rng, 50_000, adaptive=True)
        assert walk(recs)[0] == recs.size
    # neg = 0 and bin 0: the new range is range mod pos, below 2^7 -- not this form's business
    odd = np.array([0 | (3 << 1) | (4 << 8)] * 40 + [0 | (9 << 1) | (0 << 8)] + [1 | (3 << 1) | (4 << 8)] * 40, np.uint16)
    got, size = walk(odd)
    assert got == size + 1 + 40


def test_zero_probability_bin_is_reported(emul, oracle):
    recs = np.array([0 | (3 << 1) | (4 << 8)] * 100 + [1 | (0 << 1) | (9 << 8)] + [0 | (3 << 1) | (4 << 8)] * 50, np.uint16)
    assert oracle.range_encode(recs)[1] == 1
    assert emul_encode(emul, recs, 16)[0] is None
