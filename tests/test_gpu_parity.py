"""GPU parity: the HIP kernels, called through the C ABI, against the oracle.

Bit-exact is the bar (integer/byte work).  Small cases compare every byte with the oracle and
with the committed golden vectors; full-size cases use size-independent properties (decode
round trip, idempotence, sampled byte equality, a checksum of checksums).
"""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TERM1 = np.uint16(1 | (1025 << 1))


def run_cabac_batch(avr, slices):
    """slices: list of (recs, states) with equal len(states). Returns [(bytes, final_states, status)]."""
    total = sum(len(r) for r, _ in slices)
    with avr.Batch(0, max(len(slices), 1), total + 8) as b:
        for r, s in slices:
            b.add_slice_cabac(r, s)
        b.run()
        return [(*b.get(i)[:1], b.get_states(i), b.get(i)[1]) for i in range(len(slices))]


def run_range_batch(avr, slices):
    total = sum(len(r) for r in slices)
    with avr.Batch(0, max(len(slices), 1), total + 8) as b:
        for r in slices:
            b.add_slice_range(r)
        b.run()
        return [b.get(i) for i in range(len(slices))]


def pad_states(st, n):
    out = np.zeros(n, dtype=np.uint8)
    out[:len(st)] = st
    return out


# ------------------------------------------------------------------ golden vectors

@pytest.mark.parametrize("form", ["shipped", "words8", "fwd"])
def test_cabac_golden_vectors(avr, hooks, form):
    """(form: the shipped kernel, its output in 16-byte stores; its measured variants with 8-byte stores and with four state bytes
    read ahead -- the golden vectors hold the long carry chains)"""
    if form == "words8":
        hooks(k1_words8=1)
    elif form == "fwd":
        hooks(k1_fwd=1)
    g = np.load(os.path.join(GOLD, "g3_cabac.npz"), allow_pickle=False)
    n = int(g["n_cases"])
    ns = max(len(g[f"states_{i}"]) for i in range(n))
    slices = [(g[f"recs_{i}"], pad_states(g[f"states_{i}"], ns)) for i in range(n)]
    res = run_cabac_batch(avr, slices)
    for i, (data, final, st) in enumerate(res):
        k = len(g[f"states_{i}"])
        assert st == 0, f"case {i}"
        assert data == g[f"bytes_{i}"].tobytes(), f"case {i}"
        assert final[:k] == g[f"final_{i}"].tobytes(), f"case {i}"


def test_range_golden_vectors(avr):
    g = np.load(os.path.join(GOLD, "g2_range.npz"), allow_pickle=False)
    n = int(g["n_cases"])
    res = run_range_batch(avr, [g[f"recs_{i}"] for i in range(n)])
    for i, (data, st) in enumerate(res):
        assert st == 0 and data == g[f"bytes_{i}"].tobytes(), f"case {i}"


# ------------------------------------------------------------------ random, ragged, edge cases

@pytest.mark.parametrize("form", ["ref", "norm", "lds", "ref-lds", "fwd", "words8"])
@pytest.mark.parametrize("n_states", [4, 64, 460, 1024])
def test_cabac_random_ragged(avr, oracle, n_states, form, hooks):
    """form: the one-lane-per-slice coder in normalised form with the digits taken every fourth bin in step across the wave
    (shipped), and as the reference writes it (test hook k1_form_ref: a measured variant, in the test build only)."""
    if form == "ref":
        hooks(k1_form_ref=1)
    elif form == "lds":                                      # digits staged in LDS, 16-byte rows found by ballot: the north star's emitter on the shipped form (CabacLaneNS)
        hooks(k1_emit_lds=2)
    elif form == "fwd":                                      # four state bytes read ahead, an earlier bin of the group forwarding its successor (CabacLaneN::bin4: a measured variant)
        hooks(k1_fwd=1)
    elif form == "words8":                                   # the output in 8-byte stores (rounds 1-3); shipped: 16-byte ones, two words at a time (ByteWriter::put16_even_pair)
        hooks(k1_words8=1)
    elif form == "ref-lds":                                  # ... and on the form as the reference writes it (CabacLaneS)
        hooks(k1_emit_lds=1)
    rng = np.random.default_rng(100 + n_states)
    slices = []
    for i in range(200):                          # 3+ tiles, lengths from 0 to a few thousand
        n = int(rng.integers(0, 3000)) if i % 7 else int(rng.integers(0, 9))
        slices.append(oracle_lib.random_cabac_stream(rng, n, n_states, terminate=(i % 5 != 0)))
    slices.append((np.zeros(0, dtype=np.uint16), np.zeros(n_states, dtype=np.uint8)))      # empty slice
    slices.append((np.array([TERM1]), np.zeros(n_states, dtype=np.uint8)))                 # terminate only
    res = run_cabac_batch(avr, slices)
    for i, ((recs, st), got) in enumerate(zip(slices, res)):
        assert got == oracle.cabac_encode(recs, st), f"slice {i} n={len(recs)}"



def to_records8(recs):
    """Two-byte K1 records (selectors below 126, bypass, terminate) as the one-byte records of AVR_KIND_CABAC8."""
    sel = (recs >> 1).astype(np.int64)
    sel8 = np.where(sel == 1024, 126, np.where(sel == 1025, 127, sel))
    assert ((sel8 >= 0) & (sel8 < 128)).all()
    return ((sel8 << 1) | (recs & 1)).astype(np.uint8)


@pytest.mark.parametrize("n_states,long_slices", [(5, False), (126, False), (86, True)])
def test_cabac8_one_byte_records_code_what_two_byte_records_code(avr, oracle, n_states, long_slices):
    """AVR_KIND_CABAC8 (the record of recode.cpp:1442-1481 in one byte: bin | dense selector << 1): the same slices through
    avr_batch_add_slice_cabac8 and through avr_batch_add_slice_cabac give the same bytes, final states and statuses, and both equal
    the oracle -- for batches the one-lane-per-slice kernel takes (many short slices, empty and terminate-only ones among them) and for
    one the intra-slice parallel kernels take (few long ones); a selector that names no context of the slice is a bad record."""
    rng = np.random.default_rng(800 + n_states)
    if long_slices:
        slices = [oracle_lib.random_cabac_stream(rng, int(rng.integers(20000, 60000)), n_states) for _ in range(9)]
    else:
        slices = [oracle_lib.random_cabac_stream(rng, int(rng.integers(0, 2500)) if i % 6 else int(rng.integers(0, 9)), n_states,
                                                 terminate=(i % 4 != 0)) for i in range(150)]
        slices.append((np.zeros(0, dtype=np.uint16), np.zeros(n_states, dtype=np.uint8)))
        slices.append((np.array([TERM1], dtype=np.uint16), np.zeros(n_states, dtype=np.uint8)))
    total = sum(len(r) for r, _ in slices) + 8
    with avr.Batch(0, len(slices) + 1, total + 64) as b8, avr.Batch(0, len(slices) + 1, total + 64) as b16:
        for r, st in slices:
            b8.add_slice_cabac8(to_records8(r), st)
            b16.add_slice_cabac(r, st)
        b8.run(); b16.run()
        assert b8.run_info()["chunked"] == b16.run_info()["chunked"] == int(long_slices)
        for i, (r, st) in enumerate(slices):
            want = oracle.cabac_encode(r, st)
            got8 = (b8.get(i)[0], b8.get_states(i), b8.get(i)[1])
            assert got8 == want == (b16.get(i)[0], b16.get_states(i), b16.get(i)[1]), f"slice {i} n={len(r)}"
    if n_states < 126:                                       # selector n_states: neither a context of the slice nor bypass / terminate
        bad = np.array([(n_states << 1) | 1, (127 << 1) | 1], dtype=np.uint8)
        with avr.Batch(0, 4, len(slices[0][0]) + 64) as b:
            b.add_slice_cabac8(bad, np.zeros(n_states, dtype=np.uint8))
            b.add_slice_cabac8(to_records8(slices[0][0]), slices[0][1])
            b.run()
            assert b.get(0)[1] == 3                          # AVR_SLICE_BAD_RECORD
            assert (b.get(1)[0], b.get(1)[1]) == (oracle.cabac_encode(*slices[0])[0], 0)


@pytest.mark.parametrize("stride", [1, 16, 1000003])
def test_cabac_census_sample_and_hand_back(avr, oracle, stride, hooks):
    """The one-lane-per-slice kernel renumbers the batch's contexts from a SAMPLE of the records (every 16th chunk); a
    slice with a bin in a context the sample missed comes back from the first launch as 'retry' and is coded by the
    second, unrenumbered one.  Same bytes, final states and statuses whatever the sample saw: the full census (1), the
    shipped stride (16: the 1024-context streams here have many contexts that occur once or twice, so a good part of
    the slices takes the second launch), and a stride that samples next to nothing (every slice handed back)."""
    hooks(census_stride=stride, k1_path=1)
    rng = np.random.default_rng(77)
    slices = []
    for i in range(300):
        n = int(rng.integers(0, 4000)) if i % 7 else int(rng.integers(0, 9))
        recs, st = oracle_lib.random_cabac_stream(rng, n, 1024, terminate=(i % 5 != 0))
        sel = recs >> 1
        recs = np.where((sel >= 900) & (sel < 1024), ((sel % 900) << 1) | (recs & 1), recs).astype(np.uint16)   # contexts 900..1023: unused ...
        if i % 40 == 5 and n > 10:                     # ... but for single bins in a few slices, which no 1-in-16 sample is likely to see
            recs[n // 2] = np.uint16(((1000 + i // 40) << 1) | (i & 1))
            recs[n - 3] = np.uint16((1023 << 1) | 1)
        slices.append((recs, st))
    res = run_cabac_batch(avr, slices)
    for i, ((recs, st), got) in enumerate(zip(slices, res)):
        assert got == oracle.cabac_encode(recs, st), f"slice {i} n={len(recs)}"


def test_range_random_ragged(avr, oracle):
    rng = np.random.default_rng(77)
    slices = [oracle_lib.random_range_stream(rng, int(rng.integers(0, 2500)), adaptive=bool(i % 2)) for i in range(150)]
    slices.append(np.zeros(0, dtype=np.uint16))
    res = run_range_batch(avr, slices)
    for i, (recs, got) in enumerate(zip(slices, res)):
        assert got == oracle.range_encode(recs), f"slice {i} n={len(recs)}"
        assert np.array_equal(oracle.range_decode(got[0], recs), recs & 1)


def test_error_statuses_match_the_reference_errors(avr, oracle):
    ok = np.array([0 | (2 << 1), TERM1], dtype=np.uint16)
    after_finish = np.array([TERM1, 0 | (1024 << 1)], dtype=np.uint16)        # a bin after put_terminate(1)
    bad_ctx = np.array([1 | (7 << 1), TERM1], dtype=np.uint16)                # context 7 with n_states = 4
    bad_sel = np.array([1 | (1030 << 1)], dtype=np.uint16)
    st = np.zeros(4, dtype=np.uint8)
    res = run_cabac_batch(avr, [(ok, st), (after_finish, st), (bad_ctx, st), (bad_sel, st)])
    assert [r[2] for r in res] == [avr.SLICE_OK, avr.SLICE_BAD_RECORD, avr.SLICE_BAD_RECORD, avr.SLICE_BAD_RECORD]
    assert [oracle.cabac_encode(r, st)[2] for r in (ok, after_finish, bad_ctx, bad_sel)] == [0, 3, 3, 3]
    # arithmetic_code.h:116-118 "emitted a zero-probability symbol": pos = 0 and the bin is 1
    zero = np.array([1 | (0 << 1) | (9 << 8)], dtype=np.uint16)
    fine = np.array([0 | (0 << 1) | (9 << 8)], dtype=np.uint16)
    res = run_range_batch(avr, [zero, fine])
    assert res[0][1] == avr.SLICE_ZERO_PROB and oracle.range_encode(zero)[1] == 1
    assert res[1] == oracle.range_encode(fine)


def test_api_argument_errors(avr):
    with avr.Batch(0, 2, 64) as b:
        b.add_slice_cabac(np.array([TERM1]), np.zeros(8, dtype=np.uint8))
        with pytest.raises(avr.AvrError, match="same n_states"):
            b.add_slice_cabac(np.array([TERM1]), np.zeros(4, dtype=np.uint8))
        with pytest.raises(avr.AvrError, match="one kind"):
            b.add_slice_range(np.array([3 | (1 << 8)], dtype=np.uint16))
        b.add_slice_cabac(np.array([TERM1]), np.zeros(8, dtype=np.uint8))
        with pytest.raises(avr.AvrError, match="max_slices"):
            b.add_slice_cabac(np.array([TERM1]), np.zeros(8, dtype=np.uint8))
        with pytest.raises(avr.AvrError, match="not 2\\*pStateIdx"):
            avr.Batch(0, 1, 8).add_slice_cabac(np.array([TERM1]), np.full(8, 200, dtype=np.uint8))
        b.run()
        assert b.get(0)[0] == b.get(1)[0] == b"\xfe\x80"      # end_of_slice alone: 7 one bits, 0, stop bit
        b.reset()
        b.add_slice_range(np.array([3 | (1 << 8)], dtype=np.uint16))
        b.run()
        assert b.timings()["encode_ms"] > 0


def test_api_submit_wait_errors(avr):
    """The order the pipelined calls must come in, and what each refuses."""
    with avr.Batch(0, 4, 64) as b:
        with pytest.raises(avr.AvrError, match="not been submitted"):
            b.wait()
        with pytest.raises(avr.AvrError, match="unknown kind"):
            b.reserve(7, 4)
        b.add_slice_cabac(np.array([TERM1]), np.zeros(8, dtype=np.uint8))
        with pytest.raises(avr.AvrError, match="one kind"):
            b.reserve(avr.KIND_RANGE, 4)
        b.submit()
        for call in (b.submit, b.reset, lambda: b.get(0), lambda: b.add_slice_cabac(np.array([TERM1]), np.zeros(8, dtype=np.uint8)),
                     lambda: b.reserve(avr.KIND_CABAC, 1, np.zeros(8, dtype=np.uint8))):
            with pytest.raises(avr.AvrError):
                call()                                         # in flight: nothing but avr_batch_wait
        b.wait()
        b.wait()                                               # a second wait is a no-op
        assert b.get(0)[0] == b"\xfe\x80"
        with pytest.raises(avr.AvrError, match="already ran"):
            b.run()                                            # avr_batch_run keeps its contract; submit may come again
        b.submit()
        b.wait()
        assert b.get(0)[0] == b"\xfe\x80"
    with avr.Batch(0, 4, 64) as b:                             # an empty batch goes through both calls
        b.submit()
        b.wait()


# ------------------------------------------------------------------ synthetic workloads, device resident

def host_synth(avr, workload, n_slices, kind, scale, first=0):
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


@pytest.mark.parametrize("workload,scale,n_slices", [(2, 5, 130), (3, 5, 100), (4, 10, 200), (5, 1000, 1000)])
@pytest.mark.parametrize("kind", [0, 1])
def test_device_synth_and_encode_match_host_and_oracle(avr, oracle, workload, scale, n_slices, kind):
    import torch
    w = avr.DeviceWorkload.synth(workload, n_slices, kind, 0, scale)
    cfg, nb, off, recs, states = host_synth(avr, workload, n_slices, kind, scale)
    assert np.array_equal(w.n_bins.cpu().numpy().astype(np.uint32), nb)
    if kind == 0:
        assert np.array_equal(w.init_states.cpu().numpy(), states)
    # device tiles == host records packed by the plan
    order, tile_off = w.order.cpu().numpy(), w.tile_off.cpu().numpy()
    tiles = w.tiles.cpu().numpy().view(np.uint16).reshape(-1, 8)           # [chunk*64 + lane][8 records]
    for g in range(0, n_slices, max(1, n_slices // 37)):
        s = int(order[g])
        mine = tiles[int(tile_off[g // 64]) + (g % 64):int(tile_off[g // 64 + 1]):64]
        n_chunks = (int(nb[s]) + 7) // 8
        want = recs[int(off[s]):int(off[s]) + n_chunks * 8].copy()
        want[int(nb[s]):] = avr.NOP_CABAC if kind == 0 else avr.NOP_RANGE       # chunk padding = no-op records
        assert np.array_equal(mine[:n_chunks], want.reshape(-1, 8)), f"slot {g} slice {s}"
        assert (mine[n_chunks:] == (avr.NOP_CABAC if kind == 0 else avr.NOP_RANGE)).all()
    w.encode()
    got, status = w.results()
    assert not any(status)
    want, st = oracle.encode_batch(kind, *compact(recs, off, nb), states if kind == 0 else None, cfg.n_states if kind == 0 else 0, threads=8)
    assert not st.any()
    assert got == want
    # from-host upload + device pack gives the same tiles, and the slice-major kernel the same bytes
    lst = [recs[int(off[i]):int(off[i]) + int(nb[i])] for i in range(n_slices)]
    st_list = [states[i * cfg.n_states:(i + 1) * cfg.n_states] for i in range(n_slices)] if kind == 0 else None
    w2 = avr.DeviceWorkload.from_host(kind, lst, st_list, 0)
    assert torch.equal(w2.tiles, w.tiles) and torch.equal(w2.order, w.order)
    w2.encode_slice_major()
    assert w2.results()[0] == want
    if kind == 0:
        w2.encode()
        assert w2.results()[0] == want
        assert torch.equal(w2.final_states, w.final_states)


def compact(recs, off, nb):
    """Drop the per-slice padding: flat records + exact uint64 offsets for the oracle batch call."""
    parts = [recs[int(off[i]):int(off[i]) + int(nb[i])] for i in range(len(nb))]
    o = np.zeros(len(nb) + 1, dtype=np.uint64)
    o[1:] = np.cumsum(nb.astype(np.uint64))
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint16)), o


# ------------------------------------------------------------------ full size (BASELINE.json configs)

def digest(chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(hashlib.sha256(c).digest())
    return h.hexdigest()


@pytest.mark.parametrize("workload,n_slices", [(2, 512), (5, 65536), (5, 1048576)])
def test_full_size_properties(avr, oracle, workload, n_slices):
    """Config 2 and config 5 at their own size (and a 64Ki-slice cut of config 5): statuses, idempotence, sampled byte
    equality with the oracle, decode round trip of the samples, and a checksum of checksums over
    every slice against the threaded oracle."""
    w = avr.DeviceWorkload.synth(workload, n_slices, 0, 0, 1000)
    w.encode()
    got, status = w.results()
    assert not any(status)
    w.out.zero_()
    w.encode()
    again, _ = w.results()
    assert again == got                                    # idempotent: same inputs, same bytes
    sample = sorted(set(np.random.default_rng(9).integers(0, n_slices, 6).tolist() + [0, n_slices - 1]))
    for s in sample:
        cfg, nb, off, recs, states = host_synth(avr, workload, 1, 0, 1000, first=s)
        r = recs[:int(nb[0])]
        want = oracle.cabac_encode(r, states)
        assert got[s] == want[0], f"slice {s}"
        bins, _ = oracle.spec_cabac_decode(got[s], r, states)
        assert np.array_equal(bins, r & 1)                 # encode -> decode round trip
    # whole batch against the oracle through a checksum of per-slice checksums
    if workload == 5:
        cfg, nb, off, recs, states = host_synth(avr, workload, n_slices, 0, 1000)
        want, st = oracle.encode_batch(0, *compact(recs, off, nb), states, cfg.n_states, threads=16)
        assert not st.any()
        assert digest(got) == digest(want)
    # the "H.264 bytes" of the metric are the coded bytes; bins per bit stays in the realistic band
    total = sum(len(x) for x in got)
    assert 0.9 < w.total_bins / (8 * total) < 1.8


@pytest.mark.parametrize("n_slices", [65536, 1048576])
def test_full_size_properties_compress_direction(avr, oracle, n_slices):
    """Config 5 at its own size (and a 64Ki-slice cut) through K2, the recoded range coder, one lane per slice: statuses,
    idempotence, sampled byte equality with the oracle and the reference decoder's round trip, and a checksum of checksums over
    every slice against the threaded oracle."""
    w = avr.DeviceWorkload.synth(5, n_slices, avr.KIND_RANGE, 0, 1000)
    w.encode()
    got, status = w.results()
    assert not any(status)
    w.out.zero_()
    w.encode()
    again, _ = w.results()
    assert again == got
    for s in sorted(set(np.random.default_rng(19).integers(0, n_slices, 6).tolist() + [0, n_slices - 1])):
        cfg, nb, off, recs, _ = host_synth(avr, 5, 1, avr.KIND_RANGE, 1000, first=s)
        r = recs[:int(nb[0])]
        want, st = oracle.range_encode(r)
        assert st == 0 and got[s] == want, f"slice {s}"
        assert np.array_equal(oracle.range_decode(got[s], r), r & 1)
    cfg, nb, off, recs, _ = host_synth(avr, 5, n_slices, avr.KIND_RANGE, 1000)
    want, st = oracle.encode_batch(avr.KIND_RANGE, *compact(recs, off, nb), None, 0, threads=16)
    assert not st.any()
    assert digest(got) == digest(want)


def test_one_batch_over_several_devices(avr, oracle):
    """avr_multi_*: one batch sharded by greedy LPT over a device list (here the same GPU named three times: three
    sub-batches, each with its own host thread, stream and staging), results gathered by slice index."""
    rng = np.random.default_rng(12)
    slices = [oracle_lib.random_cabac_stream(rng, int(rng.lognormal(7.5, 1.0)), 120, terminate=bool(i % 4)) for i in range(90)]
    want = [oracle.cabac_encode(r, s) for r, s in slices]
    with avr.MultiBatch([0, 0, 0], len(slices), sum(len(r) for r, _ in slices) + 8) as m:
        for r, s in slices:
            m.add_slice_cabac(r, s)
        m.run()
        for i in range(len(slices)):
            assert m.get(i) == (want[i][0], 0), f"slice {i}"
        load = m.load()
        owners = [m.placement(i) for i in range(len(slices))]
    total = sum(len(r) for r, _ in slices)
    assert sum(load) == total and set(owners) == {0, 1, 2}
    assert max(load) - min(load) <= max(len(r) for r, _ in slices)       # LPT: no device is ahead by more than one slice
    from avrecode_ms_amd.sharding import lpt_assign
    assert owners == lpt_assign([len(r) for r, _ in slices], 3)           # the same plan bench.py's helper computes
    # range records (the compress direction) through the same sharding
    rr = [oracle_lib.random_range_stream(rng, int(rng.integers(0, 3000))) for _ in range(40)]
    with avr.MultiBatch([0, 0], len(rr), sum(len(r) for r in rr) + 8) as m:
        for r in rr:
            m.add_slice_range(r)
        m.run()
        for i, r in enumerate(rr):
            assert m.get(i) == oracle.range_encode(r), f"range slice {i}"


def test_batch_submit_wait_in_turn(avr, oracle):
    """Two batch objects used in turn (avr_batch_submit / avr_batch_wait): every round's bytes, final states and
    statuses equal the oracle's.  The rounds are chosen to walk the size guess through its cases: the first run of an
    object asks the device for the context count; later runs are sized by the previous count; round 3 uses MORE
    contexts than any before (one-lane-per-slice path: the extra contexts are handed back to the second launch);
    rounds 4 to 6 are few long slices (intra-slice parallel path), 6 with more contexts than its object has seen (the
    guess is too small: avr_batch_wait runs the batch again).  Slices go in through the zero-copy call on odd rounds."""
    rng = np.random.default_rng(2024)
    def round_slices(k):
        if k < 4:
            n_ctx = [40, 40, 40, 400][k]
            return [oracle_lib.random_cabac_stream(rng, int(rng.integers(0, 1500)), n_ctx) for _ in range(150)], 460
        n_ctx = 30 if k == 4 else 200
        return [oracle_lib.random_cabac_stream(rng, 20000 + 1000 * i, n_ctx) for i in range(4)], 460
    rounds = []
    for k in range(7):                                 # round 6: long slices, 200 contexts, on the object whose last count was 30
        sl, ns = round_slices(min(k, 5))
        sl = [(r, np.concatenate([s, np.zeros(ns - len(s), np.uint8)])) for r, s in sl]
        rounds.append(sl)
    bs = [avr.Batch(0, 200, 200000), avr.Batch(0, 200, 200000)]
    try:
        def fill(b, sl, zero_copy):
            b.reset()
            for r, s in sl:
                if zero_copy:
                    _, view = b.reserve(avr.KIND_CABAC, len(r), s)
                    view[:] = r
                else:
                    b.add_slice_cabac(r, s)
        infos = {}
        def check(b, sl, k):
            infos[k] = b.run_info()
            for i, (r, s) in enumerate(sl):
                data, status = b.get(i)
                assert (data, b.get_states(i), status) == oracle.cabac_encode(r, s), f"round {k} slice {i}"
        fill(bs[0], rounds[0], False)
        bs[0].submit()
        with pytest.raises(avr.AvrError):
            bs[0].get(0)                          # in flight: nothing to hand out yet
        with pytest.raises(avr.AvrError):
            bs[0].reset()
        for k in range(1, len(rounds)):
            fill(bs[k % 2], rounds[k], k % 2 == 1)
            bs[k % 2].submit()
            bs[(k - 1) % 2].wait()
            check(bs[(k - 1) % 2], rounds[k - 1], k - 1)
        bs[(len(rounds) - 1) % 2].wait()
        check(bs[(len(rounds) - 1) % 2], rounds[-1], len(rounds) - 1)
        assert [infos[k]["chunked"] for k in range(7)] == [0, 0, 0, 0, 1, 1, 1]
        assert infos[0]["rows_guessed"] == 0 and infos[1]["rows_guessed"] == 0          # first run of each object asks the device
        assert infos[2]["rows_guessed"] > 0 and infos[3]["rows_guessed"] < 100 < infos[3]["contexts_seen"]   # handed back, not run again
        assert [infos[k]["ran_again"] & 1 for k in range(7)] == [0, 0, 0, 0, 0, 0, 1]
        # the same batch again, as it is
        b = bs[(len(rounds) - 1) % 2]
        b.submit()
        b.wait()
        check(b, rounds[-1], "again")
    finally:
        for b in bs:
            b.close()


def test_batch_submit_wait_codes_and_range(avr, oracle):
    """The other two kinds through submit / wait and the zero-copy add: K2 records, and resolved codes (made here from
    records the way a recorder that tracks *state would: AVR_CODE_* of include/avrecode_ms_amd.h) in both shapes of
    batch -- many short slices (one lane per slice) and few long ones (phases B-D of the chunked path)."""
    rng = np.random.default_rng(5)
    _, mlps = avr.cabac_tables()
    mlps = np.frombuffer(mlps, np.uint8)

    def resolve(r, states):
        state = [int(x) for x in states]
        out = np.empty(len(r), np.uint8)
        for j, rec in enumerate(r.tolist()):
            b_, sel = rec & 1, rec >> 1
            if sel < 1024:
                s_ = state[sel]
                out[j] = 255 - ((b_ ^ s_) & 1) if s_ >= 126 else (s_ << 1) | b_
                state[sel] = int(mlps[127 - s_] if b_ != (s_ & 1) else mlps[128 + s_])
            else:
                out[j] = (252 | b_) if sel == 1024 else 255 - b_
        return out
    for lengths in ([int(rng.integers(0, 800)) for _ in range(60)], [9000, 12000, 17000]):
        streams = [oracle_lib.random_cabac_stream(rng, n, 40) for n in lengths]
        with avr.Batch(0, 100, 200000) as b:
            for r, s in streams:
                _, view = b.reserve(avr.KIND_CABAC_CODES, len(r))
                view[:] = resolve(r, s)
            b.submit()
            b.wait()
            for i, (r, s) in enumerate(streams):
                data, status = b.get(i)
                assert (data, status) == (oracle.cabac_encode(r, s)[0], 0), f"codes slice {i} n={len(r)}"
    ranges = [oracle_lib.random_range_stream(rng, int(rng.integers(0, 1500))) for _ in range(70)]
    with avr.Batch(0, 100, 200000) as b:
        for r in ranges:
            _, view = b.reserve(avr.KIND_RANGE, len(r))
            view[:] = r
        b.submit()
        b.wait()
        for i, r in enumerate(ranges):
            assert b.get(i) == oracle.range_encode(r), f"K2 slice {i}"


def test_batch_second_pass_from_wait(avr, oracle):
    """Few long slices, one of them with single bins in contexts nobody else uses: the sampled census misses them, the
    chunk sort sets the slice aside, and it takes the second pass -- inside the run when the run asks the device for the
    context count (first run of a batch object), from avr_batch_wait when the run was sized by a guess (second run)."""
    rng = np.random.default_rng(31)
    ns = 200
    slices = []
    for i in range(5):
        r, s = oracle_lib.random_cabac_stream(rng, 30000 + 3000 * i, 50)
        slices.append((r, np.concatenate([s, rng.integers(0, 126, ns - 50).astype(np.uint8)])))
    slices[3][0][12345] = np.uint16((199 << 1) | 1)
    slices[3][0][20001] = np.uint16((77 << 1) | 0)
    with avr.Batch(0, 8, 400000) as b:
        for r, s in slices:
            b.add_slice_cabac(r, s)
        for run in range(2):
            b.submit()
            b.wait()
            info = b.run_info()
            assert info["chunked"] == 1
            assert (info["rows_guessed"] > 0) == (run == 1)
            if run == 1:
                assert info["ran_again"] & 2               # the second pass came from avr_batch_wait
            for i, (r, s) in enumerate(slices):
                data, status = b.get(i)
                assert (data, b.get_states(i), status) == oracle.cabac_encode(r, s), f"run {run} slice {i}"
