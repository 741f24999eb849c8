"""ctypes access to oracle/ (the CPU checker) for tests, smoke() and bench.py's cpu_baseline.

Nothing in the product imports this module.
"""
import ctypes
import os
import subprocess
from ctypes import c_int, c_size_t, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libavr_ref.so")
REFERENCE = "/root/reference"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    return any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(target) for s in sources)


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in
            ("avr_oracle.c", "spec_cabac.c", "avr_oracle.h", "avr_oracle_tables.h", "avr_oracle_coder.inc")]
    if _stale(ORACLE_SO, srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)
    return ORACLE_SO


def build_ref():
    """Build oracle/_ref from the reference's own header when /root/reference is present."""
    srcs = [os.path.join(ORACLE_DIR, "ref_harness.cpp"), os.path.join(ORACLE_DIR, "avr_oracle_tables.h")]
    if os.path.isdir(REFERENCE) and _stale(REF_SO, srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)
    return REF_SO if os.path.exists(REF_SO) else None


def ptr(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


class Oracle:
    def __init__(self, path):
        L = self.L = ctypes.CDLL(path)
        for name in ("avr_oracle_half_encode", "avr_oracle_range_encode", "avr_oracle_cabac_encode",
                     "avr_spec_cabac_encode", "avr_oracle_drop_stop_byte", "avr_oracle_tail_patch"):
            getattr(L, name).restype = c_size_t
        L.avr_oracle_probability.restype = ctypes.c_uint64
        L.avr_oracle_range_decoder_new.restype = c_void_p

    # -- K1
    def _cabac(self, fn, recs, states, cap=None):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        st = np.array(states, dtype=np.uint8, copy=True)
        cap = cap if cap is not None else recs.size + 64
        out = np.zeros(cap, dtype=np.uint8)
        status = c_int(0)
        n = fn(ptr(recs), c_size_t(recs.size), ptr(st), c_size_t(st.size), ptr(out), c_size_t(cap), ctypes.byref(status))
        return out[:min(n, cap)].tobytes(), st.tobytes(), status.value

    def cabac_encode(self, recs, states, cap=None):
        """(bytes, final_states, status) from the restated cabac::encoder."""
        return self._cabac(self.L.avr_oracle_cabac_encode, recs, states, cap)

    def spec_cabac_encode(self, recs, states, cap=None):
        """Same from the H.264 9.3.4.2 bit-serial encoder."""
        return self._cabac(self.L.avr_spec_cabac_encode, recs, states, cap)

    def spec_cabac_decode(self, data, recs, states):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        st = np.array(states, dtype=np.uint8, copy=True)
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        bins = np.zeros(recs.size, dtype=np.uint8)
        rc = self.L.avr_spec_cabac_decode(ptr(buf), c_size_t(len(data)), ptr(recs), c_size_t(recs.size), ptr(st),
                                          c_size_t(st.size), ptr(bins))
        assert rc == 0
        return bins, st.tobytes()

    # -- K2
    def range_encode(self, recs, cap=None):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        cap = cap if cap is not None else recs.size + 64
        out = np.zeros(cap, dtype=np.uint8)
        status = c_int(0)
        n = self.L.avr_oracle_range_encode(ptr(recs), c_size_t(recs.size), ptr(out), c_size_t(cap), ctypes.byref(status))
        return out[:min(n, cap)].tobytes(), status.value

    def range_decode(self, data, recs):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        bins = np.zeros(recs.size, dtype=np.uint8)
        self.L.avr_oracle_range_decode(ptr(buf), c_size_t(len(data)), ptr(recs), c_size_t(recs.size), ptr(bins))
        return bins

    # -- p = 1/2 test coder
    def half_encode(self, bins):
        bins = np.ascontiguousarray(bins, dtype=np.uint8)
        out = np.zeros(bins.size // 8 + 64, dtype=np.uint8)
        status = c_int(0)
        n = self.L.avr_oracle_half_encode(ptr(bins), c_size_t(bins.size), ptr(out), c_size_t(out.size), ctypes.byref(status))
        return out[:n].tobytes()

    def half_decode(self, data, n):
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        bins = np.zeros(n, dtype=np.uint8)
        self.L.avr_oracle_half_decode(ptr(buf), c_size_t(len(data)), c_size_t(n), ptr(bins))
        return bins

    def tables(self):
        a, b = np.zeros(512, np.uint8), np.zeros(256, np.uint8)
        self.L.avr_oracle_cabac_tables(ptr(a), ptr(b))
        return a.tobytes(), b.tobytes()

    def drop_stop_byte(self, data):
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        return data[:self.L.avr_oracle_drop_stop_byte(ptr(buf), c_size_t(len(data)))]

    def tail_patch(self, data, parity, last_byte):
        buf = np.zeros(len(data) + 1, dtype=np.uint8)
        buf[:len(data)] = np.frombuffer(data, dtype=np.uint8)
        n = self.L.avr_oracle_tail_patch(ptr(buf), c_size_t(len(data)), c_int(parity), ctypes.c_uint8(last_byte))
        return buf[:n].tobytes()

    # -- threaded batch (bench cpu_baseline and large parity checks)
    def encode_batch(self, kind, recs, off, init_states, n_states, threads=1):
        """recs: flat uint16; off: uint64[n+1] record offsets. Returns (list of bytes, status array)."""
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        nb = (off[1:] - off[:-1]).astype(np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        out_off[1:] = np.cumsum(nb + 16)
        out = np.zeros(int(out_off[-1]), dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.int32)
        st = np.ascontiguousarray(init_states, dtype=np.uint8) if init_states is not None else None
        rc = self.L.avr_oracle_encode_batch(c_int(kind), ptr(recs), ptr(off), c_size_t(n), ptr(st), c_size_t(n_states),
                                            ptr(out), ptr(out_off), ptr(out_len), ptr(status), c_int(threads))
        assert rc == 0
        return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)], status


class Ref:
    """oracle/_ref: the reference's arithmetic_code.h compiled as is (see oracle/ref_harness.cpp)."""

    def __init__(self, path):
        L = self.L = ctypes.CDLL(path)
        for name in ("ref_half_encode", "ref_range_encode", "ref_cabac_encode", "ref_model_range_encode"):
            getattr(L, name).restype = c_size_t

    def half_encode(self, bins):
        bins = np.ascontiguousarray(bins, dtype=np.uint8)
        out = np.zeros(bins.size // 8 + 64, dtype=np.uint8)
        n = self.L.ref_half_encode(ptr(bins), c_size_t(bins.size), ptr(out), c_size_t(out.size))
        return out[:n].tobytes()

    def half_decode(self, data, n):
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        bins = np.zeros(n, dtype=np.uint8)
        self.L.ref_half_decode(ptr(buf), c_size_t(len(data)), c_size_t(n), ptr(bins))
        return bins

    def range_encode(self, recs):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        out = np.zeros(recs.size + 64, dtype=np.uint8)
        status = c_int(0)
        n = self.L.ref_range_encode(ptr(recs), c_size_t(recs.size), ptr(out), c_size_t(out.size), ctypes.byref(status))
        return out[:n].tobytes(), status.value

    def range_decode(self, data, recs):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        bins = np.zeros(recs.size, dtype=np.uint8)
        self.L.ref_range_decode(ptr(buf), c_size_t(len(data)), ptr(recs), c_size_t(recs.size), ptr(bins))
        return bins

    def cabac_encode(self, recs, states):
        recs = np.ascontiguousarray(recs, dtype=np.uint16)
        st = np.array(states, dtype=np.uint8, copy=True)
        out = np.zeros(recs.size + 64, dtype=np.uint8)
        status = c_int(0)
        n = self.L.ref_cabac_encode(ptr(recs), c_size_t(recs.size), ptr(st), c_size_t(st.size), ptr(out),
                                    c_size_t(out.size), ctypes.byref(status))
        return out[:n].tobytes(), st.tobytes(), status.value


_oracle = None
_ref = False


def load_oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle(build_oracle())
    return _oracle


def load_ref():
    global _ref
    if _ref is False:
        path = build_ref()
        _ref = Ref(path) if path else None
    return _ref


# ---------------------------------------------------------------- seeded stream makers shared by tests

def random_cabac_stream(rng, n, n_ctx, p_bypass=0.2, p_term0=0.03, terminate=True):
    """(recs uint16[n(+1)], init_states uint8[n_ctx]) with skewed per-context bin statistics."""
    sel = rng.integers(0, n_ctx, n)
    kind = rng.random(n)
    sel = np.where(kind < p_bypass, 1024, sel)
    sel = np.where((kind >= p_bypass) & (kind < p_bypass + p_term0), 1025, sel)
    bias = rng.random(n_ctx) ** 2
    bias = np.where(rng.random(n_ctx) < 0.5, bias, 1 - bias)
    p1 = np.where(sel < 1024, bias[np.minimum(sel, n_ctx - 1)], 0.5)
    bins = (rng.random(n) < p1).astype(np.uint16)
    bins = np.where(sel == 1025, 0, bins)
    recs = (bins | (sel << 1)).astype(np.uint16)
    if terminate:
        recs = np.concatenate([recs, np.array([1 | (1025 << 1)], dtype=np.uint16)])
    states = rng.integers(0, 126, n_ctx).astype(np.uint8)
    return recs, states


def random_range_stream(rng, n, adaptive=True, n_keys=40):
    """uint16 K2 records; adaptive=True drives {pos,neg} with the estimator update of recode.cpp:1037-1052."""
    if not adaptive:
        pos = rng.integers(1, 0x60, n)
        neg = np.minimum(rng.integers(1, 0x60, n), 0x60 - pos)
        neg = np.maximum(neg, 1)
        bins = (rng.random(n) < pos / (pos + neg)).astype(np.uint16)
        return (bins | (pos << 1) | (neg << 8)).astype(np.uint16)
    keys = rng.integers(0, n_keys, n)
    bias = rng.random(n_keys) ** 2
    bins = (rng.random(n) < bias[keys]).astype(np.uint16)
    limit = np.where(rng.random(n_keys) < 0.3, 0x50, 0x60)       # both halving thresholds
    pos = np.ones(n_keys, dtype=np.int64)
    neg = np.ones(n_keys, dtype=np.int64)
    recs = np.zeros(n, dtype=np.uint16)
    for i in range(n):
        k = keys[i]
        recs[i] = bins[i] | (pos[k] << 1) | (neg[k] << 8)
        if bins[i]:
            pos[k] += 1
        else:
            neg[k] += 1
        if pos[k] + neg[k] > limit[k]:
            pos[k] = (pos[k] + 1) // 2
            neg[k] = (neg[k] + 1) // 2
    return recs
