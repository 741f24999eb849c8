"""K1p (intra-slice parallel CABAC encode) on the CPU: the product's per-lane functions
(avrecode-ms_amd/csrc/avr_k1p.h, the code the HIP kernels wrap) are compiled for the host by
tests/k1p_emul.cpp and run chunk by chunk, then compared with the oracle.  This checks the
decomposition itself -- stretch summaries, the 4->4 chain, digit sums, carries, finish() --
without a GPU; tests/test_gpu_k1p.py checks the kernels."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "k1p_emul.cpp")
SO = os.path.join(ROOT, "tests", "_k1p_emul.so")
CSRC = os.path.join(ROOT, "avrecode-ms_amd", "csrc")


@pytest.fixture(scope="module")
def emul():
    deps = [SRC, os.path.join(CSRC, "avr_k1p.h"), os.path.join(CSRC, "avr_tables.h"), os.path.join(CSRC, "avr_div.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, "-o", SO, SRC], check=True)
    lib = ctypes.CDLL(SO)
    lib.k1p_emul_encode_resolved.restype = ctypes.c_size_t
    lib.div_emul_check.restype = ctypes.c_uint64
    lib.div_emul_check.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    return lib


def aligned(n):
    a = np.zeros(n + 64, np.uint8)
    o = (-a.ctypes.data) % 16
    return a[o:o + n + 16]


def k1p(emul, recs, states):
    """(bytes, final_states, info) or (None, None, rc)."""
    P = oracle_lib.ptr
    recs = np.ascontiguousarray(recs, dtype=np.uint16)
    st = np.array(states, dtype=np.uint8, copy=True)
    res = aligned(recs.size + 32)
    m = ctypes.c_size_t(0)
    rc = emul.k1p_emul_resolve(P(recs), ctypes.c_size_t(recs.size), P(st), ctypes.c_size_t(st.size), P(res), ctypes.byref(m))
    if rc:
        return None, None, rc
    out = np.zeros(recs.size + 64, np.uint8)
    info = np.zeros(8, np.uint32)
    n = emul.k1p_emul_encode_resolved(P(res), ctypes.c_size_t(m.value), P(out), ctypes.c_size_t(out.size), P(info))
    return out[:n].tobytes(), st.tobytes(), info


def test_golden_vectors(emul):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3_cabac.npz"), allow_pickle=False)
    for i in range(int(g["n_cases"])):
        data, final, info = k1p(emul, g[f"recs_{i}"], g[f"states_{i}"])
        assert info[3] == 0
        assert data == g[f"bytes_{i}"].tobytes() and final == g[f"final_{i}"].tobytes(), f"case {i}"


def test_random_streams_many_chunks(emul, oracle):
    rng = np.random.default_rng(31)
    for t in range(250):
        n = int(rng.integers(0, 12000))
        nctx = int(rng.integers(1, 120))
        recs, states = oracle_lib.random_cabac_stream(rng, n, nctx, terminate=bool(t % 3))
        if t % 8 == 0:
            states[:] = rng.integers(0, 128, nctx)          # includes pStateIdx 63 (codes 254/255)
        want = oracle.cabac_encode(recs, states)
        data, final, info = k1p(emul, recs, states)
        assert info[3] == 0 and (data, final) == want[:2], f"stream {t} n={n} info={info}"


def test_hard_shapes(emul, oracle):
    rng = np.random.default_rng(32)
    n = 40000

    def mk(bins, sels):
        return (np.asarray(bins, np.uint16) | (np.asarray(sels, np.uint16) << 1)).astype(np.uint16)
    cases = [
        (mk(np.arange(n) & 1, np.zeros(n)), [0]),                                        # alternating bins, one context
        (mk(rng.integers(0, 2, n), np.zeros(n)), [60]),                                  # coin flips in one context
        (mk(rng.integers(0, 2, n), np.where(rng.random(n) < 0.97, 1024, 0)), [10]),      # almost all bypass
        (mk((rng.random(n) < 0.002).astype(int), np.zeros(n)), [124]),                   # an LPS every ~500 bins
        (mk((rng.random(n) < 0.5).astype(int), rng.integers(0, 3, n)), [125, 0, 63]),
    ]
    for k, (recs, st) in enumerate(cases):
        st = np.array(st, np.uint8)
        want = oracle.cabac_encode(recs, st)
        data, final, info = k1p(emul, recs, st)
        assert info[3] == 0 and (data, final) == want[:2], f"case {k} info={info}"


def test_streams_without_lps_are_declined_not_miscoded(emul):
    # no coded LPS for more than 16 chunks: the scheme flags the slice (the kernels then hand it to
    # the serial kernel); it must never return wrong bytes silently
    n = 20000
    recs = (np.ones(n, np.uint16) | (1024 << 1)).astype(np.uint16)                      # bypass only
    _, _, info = k1p(emul, recs, np.zeros(1, np.uint8))
    assert info[3] == 1
    recs = np.ones(50000, np.uint16)                                                      # all MPS at pStateIdx 62
    _, _, info = k1p(emul, recs, np.array([125], np.uint8))
    assert info[3] == 1


def test_synthetic_slices(emul, oracle, avr):
    for w, n in ((2, 1), (4, 2), (5, 30)):
        cfg = avr.synth_config(w, 1000, 0)
        nb = np.zeros(n, np.uint32)
        avr.lib().avr_synth_count_host(ctypes.byref(cfg), 0, n, nb.ctypes.data)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum((nb.astype(np.uint64) + 7) // 8 * 8)
        recs = np.zeros(int(off[-1]), np.uint16)
        st = np.zeros(n * cfg.n_states, np.uint8)
        avr.lib().avr_synth_generate_host(ctypes.byref(cfg), 0, n, off.ctypes.data, recs.ctypes.data, st.ctypes.data)
        for i in range(n):
            r = recs[int(off[i]):int(off[i]) + int(nb[i])]
            s = st[i * cfg.n_states:(i + 1) * cfg.n_states]
            want = oracle.cabac_encode(r, s)
            data, final, info = k1p(emul, r, s)
            assert info[3] == 0 and (data, final) == want[:2], f"workload {w} slice {i}"


def test_state_transitions_are_monotone(avr):
    """What phase A's speculation rests on: in the order (62,MPS 0) < ... < (0,0) < (0,1) < ... < (62,1)
    both CABAC transition functions (cabac_code.h:43-47) are non-decreasing, so two state chains that
    have met stay together and bracket every chain started between them."""
    _, mlps = avr.cabac_tables()

    def sigma(s):
        return (s >> 1) if s & 1 else -1 - (s >> 1)

    def step(s, b):
        return mlps[128 + s] if b == (s & 1) else mlps[127 - s]
    states = sorted(range(126), key=sigma)                       # pStateIdx <= 62
    assert sigma(124) == min(map(sigma, states)) and sigma(125) == max(map(sigma, states))
    for b in (0, 1):
        images = [sigma(step(s, b)) for s in states]
        assert images == sorted(images), b
        assert all(step(s, b) < 126 for s in states)             # pStateIdx 63 is never entered
    for s in (126, 127):                                         # ... and never left
        assert step(s, 0) == s and step(s, 1) == s


def test_fp64_division_of_the_recoded_coder_is_exact(emul):
    """k_range_encode's range / total (recode.cpp:826) goes through the FP64 pipe (csrc/avr_div.h); the same function on
    the CPU against the integer divide: 200 000 dividends below 2^63 (edges, exact multiples, all-ones low words) times
    every divisor 1..255."""
    assert emul.div_emul_check(7, 200000) == 0
