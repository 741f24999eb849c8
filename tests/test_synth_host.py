"""Synthetic bin streams (csrc/avr_synth.h), host side: determinism, validity, and the stream
statistics SURVEY.md 8(d) asks for (bins per bit, bypass share)."""
import ctypes

import numpy as np
import pytest


def gen(avr, workload, n_slices, kind, scale, first=0):
    L = avr.lib()
    cfg = avr.synth_config(workload, scale, first)
    nb = np.zeros(n_slices, dtype=np.uint32)
    assert L.avr_synth_count_host(ctypes.byref(cfg), kind, n_slices, nb.ctypes.data) == 0
    off = np.zeros(n_slices + 1, dtype=np.uint64)
    off[1:] = np.cumsum((nb.astype(np.uint64) + 7) // 8 * 8)
    recs = np.zeros(int(off[-1]), dtype=np.uint16)
    states = np.zeros(n_slices * cfg.n_states, dtype=np.uint8)
    assert L.avr_synth_generate_host(ctypes.byref(cfg), kind, n_slices, off.ctypes.data, recs.ctypes.data,
                                     states.ctypes.data) == 0
    return cfg, nb, off, recs, states


@pytest.mark.parametrize("workload,scale", [(2, 10), (3, 10), (4, 20), (5, 1000)])
def test_streams_are_valid_and_realistic(avr, oracle, workload, scale):
    cfg, nb, off, recs, states = gen(avr, workload, 24, avr.KIND_CABAC, scale)
    assert states.max() < 126
    total_bins = total_bytes = bypass = 0
    for i in range(24):
        r = recs[int(off[i]):int(off[i]) + int(nb[i])]
        sel = r >> 1
        assert ((sel < cfg.n_states) | (sel == 1024) | (sel == 1025)).all()
        assert r[-1] == (1 | (1025 << 1)) and not ((sel[:-1] == 1025) & (r[:-1] & 1 == 1)).any()
        st = states[i * cfg.n_states:(i + 1) * cfg.n_states]
        data, _, status = oracle.cabac_encode(r, st)
        assert status == 0
        total_bins += r.size
        total_bytes += len(data)
        bypass += int((sel == 1024).sum())
    bins_per_bit = total_bins / (8 * total_bytes)
    assert 0.9 < bins_per_bit < 1.8, bins_per_bit
    assert 0.08 < bypass / total_bins < 0.30, bypass / total_bins


def test_slices_depend_only_on_seed_and_index(avr):
    # what lets ranks shard a workload with no exchange: slice k is the same whoever generates it
    _, nb_a, off_a, recs_a, st_a = gen(avr, 4, 8, avr.KIND_CABAC, 10, first=0)
    cfg, nb_b, off_b, recs_b, st_b = gen(avr, 4, 4, avr.KIND_CABAC, 10, first=4)
    assert np.array_equal(nb_a[4:], nb_b)
    for i in range(4):
        a = recs_a[int(off_a[4 + i]):int(off_a[4 + i]) + int(nb_a[4 + i])]
        b = recs_b[int(off_b[i]):int(off_b[i]) + int(nb_b[i])]
        assert np.array_equal(a, b)
    assert np.array_equal(st_a[4 * cfg.n_states:], st_b)


def test_range_records_follow_the_estimator_rule(avr, oracle):
    # K2 records carry the {pos,neg} of recode.cpp:1064 updated as recode.cpp:1037-1052
    cfg, nb, off, recs, _ = gen(avr, 5, 4, avr.KIND_RANGE, 1000)
    _, nb_c, off_c, recs_c, _ = gen(avr, 5, 4, avr.KIND_CABAC, 1000)
    assert np.array_equal(nb, nb_c)
    for i in range(4):
        r = recs[int(off[i]):int(off[i]) + int(nb[i])]
        c = recs_c[int(off_c[i]):int(off_c[i]) + int(nb_c[i])]
        assert np.array_equal(r & 1, c & 1)
        est = {}
        for k in range(r.size):
            sel = int(c[k] >> 1)
            pos, neg = est.get(sel, (1, 1))
            assert (int(r[k]) >> 1) & 0x7f == pos and (int(r[k]) >> 8) & 0x7f == neg
            if r[k] & 1:
                pos += 1
            else:
                neg += 1
            if pos + neg > 0x60:
                pos, neg = (pos + 1) // 2, (neg + 1) // 2
            est[sel] = (pos, neg)
        data, status = oracle.range_encode(r)
        assert status == 0
        assert np.array_equal(oracle.range_decode(data, r), r & 1)
