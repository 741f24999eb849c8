#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch API (host records in, host bytes out) -- never bench.py's `value`.

    python tools/e2e_batch.py [--workload 2] [--slices 128]          (E2E_OBJECTS=3, E2E_ROUNDS=30, E2E_CODES=1)

Batches are submitted in turn over a few batch objects (avr_batch_submit / avr_batch_wait) and, for comparison, one at
a time (avr_batch_run); outputs are checked against each other.
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


def pipeline(batches, rounds, label, out_bytes):
    """submit / wait over the batch objects in turn (each holds its slices in pinned memory already, as a recorder that
    appends through avr_batch_reserve_slice leaves them): H2D of one batch under the kernels and D2H of the others."""
    for b in batches:                                      # first run of an object: asks the device for the context count
        b.submit(); b.wait()
    k = len(batches)
    t0 = time.perf_counter()
    for i in range(rounds):
        if i >= k:
            batches[i % k].wait()
        batches[i % k].submit()
    for i in range(rounds, rounds + k):
        batches[i % k].wait()
    t = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(rounds):
        batches[i % k].submit(); batches[i % k].wait()
    ts = time.perf_counter() - t1
    print(f"{label}: {rounds} batches of {n} slices over {k} batch objects | in turn {1e3*t/rounds:.2f} ms per batch = "
          f"{rounds*out_bytes/t/1e9:.2f} GB/s of H.264 | one at a time {1e3*ts/rounds:.2f} ms = {rounds*out_bytes/ts/1e9:.2f} GB/s | "
          f"last run: {batches[0].run_info()} {batches[0].timings()}")


K = int(os.environ.get("E2E_OBJECTS", "3"))
ROUNDS = int(os.environ.get("E2E_ROUNDS", "30"))
bs = [avr.Batch(0, n, int(nb.sum()) + 8) for _ in range(K)]
t0 = time.perf_counter()
for b in bs:
    for i in range(n):
        b.add_slice_cabac(recs[int(off[i]):int(off[i]) + int(nb[i])], st[i * cfg.n_states:(i + 1) * cfg.n_states])
t1 = time.perf_counter()
print(f"host fill (memcpy into pinned memory, one thread): {K * recs.nbytes / (t1 - t0) / 1e9:.1f} GB/s of records")
bs[0].run()
want_all = [bs[0].get(i)[0] for i in range(n)]
out_bytes = sum(len(x) for x in want_all)
bs[0].submit(); bs[0].wait()
assert all(bs[0].get(i)[0] == want_all[i] for i in range(n))
pipeline(bs, ROUNDS, "records (2 B per bin)", out_bytes)
assert all(bs[1].get(i)[0] == want_all[i] for i in range(n))
for b in bs:
    b.close()

if os.environ.get("E2E_CODES", "1") != "0":
    m = min(n, 8)                                          # resolving in Python is slow: a few slices, repeated
    codes = [resolve(recs[int(off[i]):int(off[i]) + int(nb[i])], st[i * cfg.n_states:(i + 1) * cfg.n_states]) for i in range(m)]
    bc = [avr.Batch(0, n, int(nb[:m].max()) * n + 64 * n) for _ in range(K)]
    for b in bc:
        for i in range(n):
            b.add_codes(codes[i % m])
    bc[0].run()
    assert all(bc[0].get(i)[0] == want_all[i] for i in range(m))
    out_c = sum(len(bc[0].get(i)[0]) for i in range(n))
    pipeline(bc, ROUNDS, "resolved codes (1 B per bin)", out_c)
    assert all(bc[1].get(i)[0] == want_all[i] for i in range(m))
    for b in bc:
        b.close()
