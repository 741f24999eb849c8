#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch API (host records in, host bytes out) -- never bench.py's `value`.

    python tools/e2e_batch.py [--workload 2] [--slices 128]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avrecode_ms_amd as avr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", type=int, default=2)
ap.add_argument("--slices", type=int, default=128)
args = ap.parse_args()
L = avr.lib()
cfg = avr.synth_config(args.workload, 1000, 0)
n = args.slices
nb = np.zeros(n, np.uint32)
L.avr_synth_count_host(ctypes.byref(cfg), 0, n, nb.ctypes.data)
off = np.zeros(n + 1, np.uint64)
off[1:] = np.cumsum((nb.astype(np.uint64) + 7) // 8 * 8)
recs = np.zeros(int(off[-1]), np.uint16)
st = np.zeros(n * cfg.n_states, np.uint8)
L.avr_synth_generate_host(ctypes.byref(cfg), 0, n, off.ctypes.data, recs.ctypes.data, st.ctypes.data)
# the same slices as resolved codes (what an adapter that tracks *state records): one byte per bin
_, mlps = avr.cabac_tables()
mlps = np.frombuffer(mlps, np.uint8)


def resolve(r, states):
    state = states.astype(np.int64).copy()
    out = np.empty(r.size, np.uint8)
    for j in range(r.size):
        b_, sel = int(r[j]) & 1, int(r[j]) >> 1
        if sel < 1024:
            s_ = int(state[sel])
            out[j] = 255 - ((b_ ^ s_) & 1) if s_ >= 126 else (s_ << 1) | b_
            state[sel] = mlps[127 - s_] if b_ != (s_ & 1) else mlps[128 + s_]
        else:
            out[j] = (252 | b_) if sel == 1024 else 255 - b_
    return out


with avr.Batch(0, n, int(nb.sum()) + 8) as b:
    for rep in range(3):
        b.reset()
        t0 = time.perf_counter()
        for i in range(n):
            b.add_slice_cabac(recs[int(off[i]):int(off[i]) + int(nb[i])], st[i * cfg.n_states:(i + 1) * cfg.n_states])
        t1 = time.perf_counter()
        b.run()
        t2 = time.perf_counter()
        out_bytes = sum(len(b.get(i)[0]) for i in range(n))
        t = b.timings()
        print(f"rep {rep}: {n} slices, {int(nb.sum())} bins, {out_bytes} H.264 bytes | add (host memcpy) {1e3*(t1-t0):.1f} ms, "
              f"run {1e3*(t2-t1):.1f} ms [h2d {t['h2d_ms']:.2f} plan/pack {t['pack_ms']:.2f} encode {t['encode_ms']:.2f} d2h {t['d2h_ms']:.2f}] "
              f"-> {out_bytes/(t2-t1)/1e9:.2f} GB/s of H.264 through avr_batch_run")

    if os.environ.get("E2E_CODES", "1") != "0":
        m = min(n, 8)                                      # resolving in Python is slow: a few slices, repeated
        codes = [resolve(recs[int(off[i]):int(off[i]) + int(nb[i])], st[i * cfg.n_states:(i + 1) * cfg.n_states]) for i in range(m)]
        want = [b.get(i)[0] for i in range(m)]
        with avr.Batch(0, n, int(nb[:m].max()) * n + 64 * n) as bc:
            for rep in range(3):
                bc.reset()
                for i in range(n):
                    bc.add_codes(codes[i % m])
                t1 = time.perf_counter()
                bc.run()
                t2 = time.perf_counter()
                out_bytes = sum(len(bc.get(i)[0]) for i in range(n))
                t = bc.timings()
                assert all(bc.get(i)[0] == want[i] for i in range(m))
                print(f"codes rep {rep}: {n} slices | run {1e3*(t2-t1):.1f} ms [h2d {t['h2d_ms']:.2f} encode {t['encode_ms']:.2f} "
                      f"d2h {t['d2h_ms']:.2f}] -> {out_bytes/(t2-t1)/1e9:.2f} GB/s of H.264 through avr_batch_run from resolved codes")
