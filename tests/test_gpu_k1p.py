"""GPU: the intra-slice parallel kernels (K1p, avr_cabac_encode_chunked_device) give the same
bytes, lengths, statuses and final states as the one-lane-per-slice kernel and the oracle."""
import ctypes

import numpy as np
import pytest

import oracle_lib
from test_gpu_parity import compact, host_synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,scale,n_slices", [(2, 40, 70), (4, 100, 40), (5, 1000, 300), (3, 30, 50)])
def test_chunked_equals_serial_and_oracle(avr, oracle, workload, scale, n_slices):
    import torch
    w = avr.DeviceWorkload.synth(workload, n_slices, 0, 0, scale)
    w.encode()
    serial, st_serial = w.results()
    fs_serial = w.final_states.clone()
    w.out.zero_(); w.out_len.zero_(); w.final_states.zero_()
    w.encode_chunked()
    chunked, st_chunked = w.results()
    assert not any(st_serial) and not any(st_chunked)
    assert chunked == serial
    assert torch.equal(w.final_states, fs_serial)
    cfg, nb, off, recs, states = host_synth(avr, workload, n_slices, 0, scale)
    want, _ = oracle.encode_batch(0, *compact(recs, off, nb), states, cfg.n_states, threads=8)
    assert chunked == want


def test_chunked_random_and_declined_slices(avr, oracle):
    rng = np.random.default_rng(55)
    slices = []
    for i in range(40):
        n = int(rng.integers(0, 30000))
        slices.append(oracle_lib.random_cabac_stream(rng, n, 64, terminate=bool(i % 4)))
    # shapes where the two extreme state chains of phase A never meet inside a segment (replay path)
    n = 9000

    def mk(bins, sels):
        return (np.asarray(bins, np.uint16) | (np.asarray(sels, np.uint16) << 1)).astype(np.uint16)
    slices.append((mk(np.arange(n) & 1, np.zeros(n)), np.zeros(64, np.uint8)))                 # alternating, one context
    slices.append((mk(rng.integers(0, 2, n), np.zeros(n)), np.full(64, 60, np.uint8)))          # coin flips, one context
    slices.append((mk(rng.integers(0, 2, 3 * n), rng.integers(0, 2, 3 * n)), np.full(64, 127, np.uint8)))   # pStateIdx 63
    slices.append((mk((np.arange(4 * n) % 3 == 0), np.arange(4 * n) % 2), np.array([124, 125] * 32, np.uint8)))
    # slices the scheme declines (no coded LPS for > 16 chunks): must come back right via the serial kernel
    slices.append(((np.ones(40000, np.uint16) | (1024 << 1)).astype(np.uint16), np.zeros(64, np.uint8)))
    slices.append((np.ones(60000, np.uint16), np.full(64, 125, np.uint8)))
    slices.append((np.zeros(0, np.uint16), np.zeros(64, np.uint8)))
    # a bin after put_terminate(1)
    bad = np.array([1 | (1025 << 1), 0], dtype=np.uint16)
    slices.append((bad, np.zeros(64, np.uint8)))
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()
    got, status = w.results()
    fs = w.final_states.cpu().numpy().reshape(len(slices), 64)
    for i, (r, s) in enumerate(slices[:-1]):
        want = oracle.cabac_encode(r, s)
        assert status[i] == 0 and got[i] == want[0] and fs[i].tobytes() == want[1], f"slice {i} n={len(r)}"
    assert status[-1] == avr.SLICE_BAD_RECORD


@pytest.mark.parametrize("n_ctx", [1, 2, 3, 5, 17, 200, 513, 1024])
def test_chunked_every_context_count_and_length_class(avr, oracle, n_ctx):
    """The sort's key width, the LDS state tables and the per-chunk entry rows all depend on the number
    of contexts; slice lengths sit on and around the chunk (1024) and sort-block (4096) boundaries."""
    rng = np.random.default_rng(1000 + n_ctx)
    lengths = [1, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 12288, 20001]
    slices = []
    for i, n in enumerate(lengths):
        recs, st = oracle_lib.random_cabac_stream(rng, n, n_ctx, p_bypass=0.15 if i % 2 else 0.0, terminate=bool(i % 3))
        if i % 4 == 0:
            st[:] = rng.integers(0, 128, n_ctx)             # includes pStateIdx 63
        slices.append((recs, st))
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()
    got, status = w.results()
    fs = w.final_states.cpu().numpy().reshape(len(slices), -1)
    for i, (r, s) in enumerate(slices):
        want = oracle.cabac_encode(r, s)
        assert status[i] == 0 and got[i] == want[0], f"slice {i} n={len(r)}"
        assert fs[i][:n_ctx].tobytes() == want[1], f"final states of slice {i}"


@pytest.mark.parametrize("waves", [1, 2, 3, 7, 8])
def test_chunked_sort_in_workgroups_of_any_size(avr, oracle, hooks, waves):
    """k_k1p_local sizes its workgroups by what the CU's LDS takes (eight waves with the contexts of a 1080p stream, one with a
    thousand): every size gives the same bytes (test hook local_waves), with the 16-bit selector table (up to 500 contexts)
    and the 32-bit one."""
    hooks(local_waves=waves)
    rng = np.random.default_rng(2000 + waves)
    slices = []
    for i, n_ctx in enumerate([86, 86, 40, 500, 501, 86][:6 if waves < 3 else 3]):     # (501 contexts: 129 KiB a wave, one or two waves only)
        recs, st = oracle_lib.random_cabac_stream(rng, 9000 + 1500 * i, n_ctx, p_bypass=0.1, terminate=True)
        slices.append((recs, st, n_ctx))
    for n_ctx in sorted({c for _, _, c in slices}):
        group = [(r, s) for r, s, c in slices if c == n_ctx]
        w = avr.DeviceWorkload.from_host(0, [r for r, _ in group], [s for _, s in group], 0)
        w.encode_chunked()
        got, status = w.results()
        for i, (r, s) in enumerate(group):
            want = oracle.cabac_encode(r, s)
            assert status[i] == 0 and got[i] == want[0], f"{n_ctx} contexts, slice {i}"


def test_batch_api_from_resolved_codes(avr, oracle):
    """avr_batch_add_slice_codes: the adapter resolves (symbol, *state) itself (cabac_code.h:33, 43-47) and
    ships one byte per bin; no state arrays, half the bytes over PCIe, phase A skipped."""
    lps, mlps = avr.cabac_tables()
    rng = np.random.default_rng(404)
    slices, codes = [], []
    for i in range(24):
        n = int(rng.integers(0, 40000)) if i else 0
        recs, st = oracle_lib.random_cabac_stream(rng, n, int(rng.integers(1, 300)), terminate=bool(i % 5))
        if i % 6 == 0:
            st[:] = rng.integers(0, 128, st.size)
        state = [int(x) for x in st]
        out = np.zeros(recs.size, np.uint8)
        for j, r in enumerate(recs):                        # what a hook adapter does per bin
            b, sel = int(r) & 1, int(r) >> 1
            if sel < 1024:
                s = state[sel]
                out[j] = 255 - ((b ^ s) & 1) if s >= 126 else (s << 1) | b                       # AVR_CODE_CONTEXT
                state[sel] = mlps[127 - s] if b != (s & 1) else mlps[128 + s]                  # cabac_code.h:43-47
            else:
                out[j] = (252 | b) if sel == 1024 else 255 - b                                   # AVR_CODE_BYPASS / _TERMINATE
        slices.append((recs, st)); codes.append(out)
    b = avr.Batch(0, len(codes), sum(c.size for c in codes) + 64)
    for c in codes:
        b.add_codes(c)
    b.run()
    for i, (r, s) in enumerate(slices):
        data, status = b.get(i)
        assert status == 0 and data == oracle.cabac_encode(r, s)[0], f"slice {i}"
    with pytest.raises(avr.AvrError):
        b.get_states(0)                                      # there are no state arrays in this kind
    b.close()


def test_chunked_very_long_slices(avr, oracle):
    """More than 1024 chunks per slice: the per-slice kernels (B2, D) go through several tiles."""
    rng = np.random.default_rng(2024)
    slices = [oracle_lib.random_cabac_stream(rng, n, 40) for n in (1_300_000, 2_100_001, 5)]
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()
    got, status = w.results()
    for i, (r, s) in enumerate(slices):
        want = oracle.cabac_encode(r, s)
        assert status[i] == 0 and got[i] == want[0], f"slice {i}"


def test_chunked_full_size_config2_sampled(avr, oracle):
    w = avr.DeviceWorkload.synth(2, 512, 0, 0, 1000)
    w.encode_chunked()
    got, status = w.results()
    assert not any(status)
    for s in (0, 17, 255, 511):
        cfg, nb, off, recs, states = host_synth(avr, 2, 1, 0, 1000, first=s)
        want = oracle.cabac_encode(recs[:int(nb[0])], states)
        assert got[s] == want[0], f"slice {s}"
    again = w.results()[0]
    w.out.zero_()
    w.encode_chunked()
    assert w.results()[0] == again == got


def test_chunked_full_size_config2_every_slice(avr, oracle):
    """The headline path (BASELINE.json configs[1] through encode_chunked(), i.e. K1p) with EVERY one of its 512 slices against the
    oracle -- cabac_code.h:33-67 on arithmetic_code.h, threaded over the host's cores -- through a checksum of checksums, and the
    final context states of every slice beside the bytes (the twin of test_range_chunked_full_size's check of the compress direction)."""
    import hashlib
    n_slices = 512
    w = avr.DeviceWorkload.synth(2, n_slices, avr.KIND_CABAC, 0, 1000)
    w.encode_chunked()
    got, status = w.results()
    assert not any(status)
    cfg, nbh, off, recs, states = host_synth(avr, 2, n_slices, avr.KIND_CABAC, 1000)
    assert np.array_equal(nbh, w.n_bins.cpu().numpy())
    parts = [recs[int(off[i]):int(off[i]) + int(nbh[i])] for i in range(n_slices)]
    roff = np.zeros(n_slices + 1, np.uint64)
    roff[1:] = np.cumsum(nbh.astype(np.uint64))
    want, st = oracle.encode_batch(avr.KIND_CABAC, np.concatenate(parts), roff, states, cfg.n_states, threads=16)
    assert not st.any()
    dig = lambda chunks: hashlib.sha256(b"".join(hashlib.sha256(c).digest() for c in chunks)).hexdigest()
    assert [len(x) for x in got] == [len(x) for x in want]
    assert dig(got) == dig(want)


def test_batch_api_takes_the_chunked_path_for_long_slices(avr, oracle, hooks):
    rng = np.random.default_rng(77)
    slices = [oracle_lib.random_cabac_stream(rng, int(rng.integers(15000, 40000)), 200) for _ in range(12)]
    slices.append((np.array([0 | (300 << 1), 1 | (1025 << 1)], dtype=np.uint16), np.zeros(200, np.uint8)))   # bad selector
    want = [oracle.cabac_encode(r, s) for r, s in slices]
    for force in (None, "serial", "chunked"):
        if force:
            hooks(k1_path={"serial": 1, "chunked": 2}[force])
        with avr.Batch(0, len(slices), sum(len(r) for r, _ in slices) + 8) as b:
            for r, s in slices:
                b.add_slice_cabac(r, s)
            b.run()
            for i in range(len(slices) - 1):
                data, status = b.get(i)
                assert (data, b.get_states(i), status) == want[i], f"path {force} slice {i}"
            assert b.get(len(slices) - 1)[1] == avr.SLICE_BAD_RECORD


def test_two_stage_form_resolved_codes(avr, oracle):
    """Stage 1 (state resolution) and stage 2 (arithmetic coding from resolved codes) on their own."""
    import torch
    w = avr.DeviceWorkload.synth(2, 40, 0, 0, 60)
    w.encode()
    want, _ = w.results()
    codes = w.resolve()
    # codes are (state << 1 | bin) etc.: check a slice against a CPU state walk
    nb = w.n_bins.cpu().numpy()
    res_off = w._plan["tensors"]["res_off"].cpu().numpy()
    cfg, nbh, off, recs, states = host_synth(avr, 2, 40, 0, 60)
    _, mlps = avr.cabac_tables()
    host = codes.cpu().numpy()
    for s in (0, 39):
        st = states[s * cfg.n_states:(s + 1) * cfg.n_states].copy()
        r = recs[int(off[s]):int(off[s]) + int(nb[s])]
        exp = np.zeros(r.size, np.uint8)
        for i, rec in enumerate(r):
            b, sel = int(rec) & 1, int(rec) >> 1
            if sel < 1024:
                exp[i] = (st[sel] << 1) | b
                st[sel] = mlps[128 + st[sel]] if b == (st[sel] & 1) else mlps[127 - st[sel]]
            else:
                exp[i] = (252 | b) if sel == 1024 else 255 - b
        assert np.array_equal(host[int(res_off[s]):int(res_off[s]) + r.size], exp), f"slice {s}"
    w.out.zero_(); w.out_len.zero_()
    w.encode_resolved(codes)
    got, status = w.results()
    assert not any(status) and got == want
    # a stream with no LPS at all (bypass only): stage 2 has no serial kernel to fall back to and walks it
    n = 30000
    byp = (np.arange(n) & 1).astype(np.uint8) | 252
    wl = avr.DeviceWorkload.from_host(0, [((np.arange(n) & 1) | (1024 << 1)).astype(np.uint16)], [np.zeros(4, np.uint8)], 0)
    plan = wl._chunk_plan()
    codes2 = torch.zeros(plan["plan"].res_total + 64, dtype=torch.uint8, device="cuda")
    base = (codes2.data_ptr() + 255) // 256 * 256 - codes2.data_ptr()
    codes2[base:base + n] = torch.from_numpy(byp).cuda()
    codes2[base + n:base + n + 32] = 252
    wl.encode_resolved(codes2[base:])
    got2, st2 = wl.results()
    assert st2 == [0] and got2[0] == oracle.cabac_encode(((np.arange(n) & 1) | (1024 << 1)).astype(np.uint16), np.zeros(4, np.uint8))[0]


@pytest.mark.parametrize("workload,scale,n_slices", [(5, 1000, 2000), (2, 30, 100)])
def test_dense_context_ids_change_nothing_but_the_numbering(avr, oracle, workload, scale, n_slices):
    import torch
    w = avr.DeviceWorkload.synth(workload, n_slices, 0, 0, scale)
    w.encode()
    want, _ = w.results()
    fs = w.final_states.clone()
    full_n = w.n_states
    w.densify()
    assert w.n_states < full_n
    w.out.zero_(); w.out_len.zero_()
    w.encode()
    got, status = w.results()
    assert not any(status) and got == want
    assert torch.equal(w.final_states_full(), fs)
    w.out.zero_(); w.out_len.zero_()
    w.encode_chunked()
    got, status = w.results()
    assert not any(status) and got == want
    assert torch.equal(w.final_states_full(), fs)


def _codes_of(avr, recs, st):
    """What a hook adapter records per bin: AVR_CODE_CONTEXT / _BYPASS / _TERMINATE (include/avrecode_ms_amd.h)."""
    _, mlps = avr.cabac_tables()
    state = [int(x) for x in st]
    out = np.zeros(recs.size, np.uint8)
    for j, r in enumerate(recs):
        b, sel = int(r) & 1, int(r) >> 1
        if sel < 1024:
            s = state[sel]
            out[j] = 255 - ((b ^ s) & 1) if s >= 126 else (s << 1) | b
            state[sel] = mlps[127 - s] if b != (s & 1) else mlps[128 + s]
        else:
            out[j] = (252 | b) if sel == 1024 else 255 - b
    return out


def test_serial_codes_kernel_equals_stage2_and_oracle(avr, oracle):
    """k_cabac_encode_codes (one lane per slice from resolved codes) against phases B-D and the oracle."""
    w = avr.DeviceWorkload.synth(2, 70, 0, 0, 40)
    w.encode()
    want, _ = w.results()
    codes = w.resolve()
    w.out.zero_(); w.out_len.zero_()
    w.encode_codes_serial(codes)
    got, status = w.results()
    assert not any(status) and got == want
    cfg, nb, off, recs, states = host_synth(avr, 2, 70, 0, 40)
    assert got == oracle.encode_batch(0, *compact(recs, off, nb), states, cfg.n_states, threads=8)[0]


@pytest.mark.parametrize("every", [1, 3])
def test_phase_d_hand_over_is_coded_by_the_serial_kernels(avr, oracle, hooks, every):
    """Phase D hands a slice whose carries it does not resolve in parallel to a serial kernel.  The pattern (a
    carry >= 2 into a 33-digit segment ffff...fffe) does not occur in practice, so the test hook
    k1p_force_retry_every (test build of the library only) makes phase D hand over every n-th slice: the bytes must still be the oracle's, from
    records (k_cabac_encode) and from resolved codes (k_cabac_encode_codes), with status 0 everywhere."""
    hooks(k1p_force_retry_every=every)
    rng = np.random.default_rng(31 + every)
    slices = [oracle_lib.random_cabac_stream(rng, int(rng.integers(1, 30000)), 120, terminate=bool(i % 3)) for i in range(20)]
    want = [oracle.cabac_encode(r, s) for r, s in slices]
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()                                       # records: phase A .. D, then k_cabac_encode
    got, status = w.results()
    assert not any(status) and got == [x[0] for x in want]
    codes = w.resolve()
    w.out.zero_(); w.out_len.zero_()
    w.encode_resolved(codes)                                 # codes: phases B .. D, then k_cabac_encode_codes
    got, status = w.results()
    assert not any(status) and got == [x[0] for x in want]
    # and through the batch API (what the host decompressor uses)
    hooks(k1p_force_retry_every=every, k1_path=2)
    with avr.Batch(0, len(slices), sum(len(r) for r, _ in slices) + 64) as b:
        for r, s in slices:
            b.add_codes(_codes_of(avr, r, s))
        b.run()
        for i in range(len(slices)):
            assert b.get(i) == (want[i][0], 0), f"slice {i}"


def test_batch_of_many_short_slices_from_codes(avr, oracle):
    """10 000 short slices of resolved codes: the batch takes the one-lane-per-slice kernel (no per-chunk
    machinery), inside the limits the caller declared (max_bins = the exact total)."""
    rng = np.random.default_rng(9)
    slices = [oracle_lib.random_cabac_stream(rng, int(rng.integers(0, 40)), 30, terminate=bool(i % 2)) for i in range(10000)]
    total = sum(len(r) for r, _ in slices)
    with avr.Batch(0, len(slices), total) as b:
        for r, s in slices:
            b.add_codes(_codes_of(avr, r, s))
        b.run()
        for i, (r, s) in enumerate(slices):
            assert b.get(i) == (oracle.cabac_encode(r, s)[0], 0), f"slice {i}"


@pytest.mark.parametrize("workload,n_slices", [(3, 4096), (4, 16384)])
def test_chunked_full_size_configs_3_and_4_sampled(avr, oracle, workload, n_slices):
    """BASELINE.json configs[2] and configs[3] at their own size through the intra-slice parallel kernels: every status 0,
    the same bytes on a second run, and byte equality with the oracle (and a decode round trip) on a seeded sample of
    slices -- the whole batch is 2.8 / 4.0 G bins, more than the CPU checker does inside a test; bench.py byte-compares
    the leading slices it times the CPU on (profiles/r02_bench_w3.json, _w4.json)."""
    w = avr.DeviceWorkload.synth(workload, n_slices, 0, 0, 1000)
    w.encode_chunked()
    got, status = w.results()
    assert not any(status)
    lens = [len(x) for x in got]
    w.out.zero_()
    w.encode_chunked()
    again, _ = w.results()
    assert again == got
    nb = w.n_bins.cpu().numpy()
    # every slice of the first and of the last tile of 64 (where a batch's grid begins and ends), 24 random ones, the longest and the shortest
    sample = sorted(set(np.random.default_rng(workload).integers(0, n_slices, 24).tolist() + list(range(64)) + list(range(n_slices - 64, n_slices))
                        + [int(nb.argmax()), int(nb.argmin())]))
    for s in sample:
        cfg, nbh, off, recs, states = host_synth(avr, workload, 1, 0, 1000, first=s)
        assert int(nbh[0]) == int(nb[s])
        r = recs[:int(nbh[0])]
        want = oracle.cabac_encode(r, states)
        assert got[s] == want[0], f"slice {s}"
        if s in sample[:4]:
            bins, _ = oracle.spec_cabac_decode(got[s], r, states)
            assert np.array_equal(bins, r & 1)
    assert 0.9 < w.total_bins / (8 * sum(lens)) < 1.8


@pytest.mark.parametrize("stride", [1, 16, 4099])
def test_chunked_census_sample_second_pass_and_validation(avr, oracle, stride, hooks):
    """The intra-slice parallel path renumbers the batch's contexts from a sample of the records; the chunk sort, which
    looks every record up, sets aside the slices with a bin in a context the sample missed, and those take a second
    pass with every record counted (stride 4099: the sample is next to nothing, so every slice with a context bin does;
    stride 1: no sampling).  The same kernel is where records are validated now: a selector the slice does not have, a
    record with a bit above its selector, put_terminate(1) anywhere but last -- each in the middle of a long slice."""
    hooks(census_stride=stride)
    rng = np.random.default_rng(91)
    ns = 300
    slices = []
    for i in range(12):
        r, s = oracle_lib.random_cabac_stream(rng, 9000 + 2500 * i, 60, terminate=bool(i % 3))
        slices.append((r, np.concatenate([s, rng.integers(0, 126, ns - 60).astype(np.uint8)])))
    for i, ctx in ((2, 299), (7, 150), (9, 61)):       # single bins in contexts nobody else uses, far apart
        slices[i][0][4000 + 777 * i] = np.uint16((ctx << 1) | (i & 1))
    good = len(slices)
    def spoiled(at, value):
        r, s = oracle_lib.random_cabac_stream(rng, 12000, 60)
        r[at] = np.uint16(value)
        return r, np.concatenate([s, np.zeros(ns - 60, np.uint8)])
    slices.append(spoiled(6001, (ns << 1) | 1))            # the first selector past the slice's contexts
    slices.append(spoiled(11000, (1027 << 1)))             # no selector at all
    slices.append(spoiled(3, 0x8000 | (5 << 1)))           # a bit above the selector
    slices.append(spoiled(5000, 1 | (1025 << 1)))          # put_terminate(1) with bins behind it (the stream's own comes last)
    slices.append(spoiled(11999, 1 | (1025 << 1)))         # ... right before the last record, which is one too
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()
    got, status = w.results()
    fs = w.final_states.cpu().numpy().reshape(len(slices), ns)
    for i, (r, s) in enumerate(slices[:good]):
        want = oracle.cabac_encode(r, s)
        assert status[i] == 0 and got[i] == want[0] and fs[i].tobytes() == want[1], f"slice {i} n={len(r)}"
    assert list(status[good:]) == [avr.SLICE_BAD_RECORD] * 5


@pytest.mark.parametrize("stride", [0, 4099])
def test_device_calls_sized_by_a_guess_of_the_context_count(avr, oracle, hooks, stride):
    """avr_cabac_encode_chunked_device_hinted / _tiles_device_hinted (what bench.py's steps call): sized by the count a previous run
    reported nothing waits for the device, and the bytes are the same; the one-lane-per-slice call is exact whatever the guess; the
    chunked call reports a guess that was too small (counts[0] > rows_hint: run again) and the slices it left for a second pass
    (stride 4099: the census sees next to nothing, so that is every slice), which avr_cabac_encode_chunked_second_pass_device codes."""
    import torch
    if stride:
        hooks(census_stride=stride)
    rng = np.random.default_rng(311)
    ns = 120
    slices = []
    for i in range(20):
        r, s = oracle_lib.random_cabac_stream(rng, 9000 + 1700 * i, 70, terminate=bool(i % 3))
        slices.append((r, np.concatenate([s, rng.integers(0, 126, ns - 70).astype(np.uint8)])))
    if stride:                                             # single bins in contexts nobody else uses: the sample misses them
        for i, ctx in ((2, 119), (7, 100), (9, 71)):
            slices[i][0][4000 + 777 * i] = np.uint16((ctx << 1) | (i & 1))
    want = [oracle.cabac_encode(r, s) for r, s in slices]

    def check(w, what):
        got, status = w.results()
        fs = w.final_states.cpu().numpy().reshape(len(slices), ns)
        for i in range(len(slices)):
            assert status[i] == 0 and got[i] == want[i][0] and fs[i].tobytes() == want[i][1], f"{what}: slice {i}"
        w.out.zero_(); w.out_len.zero_(); w.final_states.zero_()

    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    # the intra-slice parallel path: no guess, the count's guess, a guess that is too small
    w.encode_chunked(); torch.cuda.synchronize()
    first = w.settle()
    assert first["hint"] == 0 and 60 <= first["rows"] <= 70 and w.rows_hint == min(ns, first["rows"] + 8) and not first["redone"]
    check(w, "asked")
    w.encode_chunked(); torch.cuda.synchronize()
    left = int(w._counts[1])                               # slices the call left for a second pass
    assert (left >= 1) if stride else (left == 0)
    second = w.settle()
    assert second["hint"] == w.rows_hint and second["rows"] == first["rows"] and second["redone"] == bool(left)
    check(w, "guessed")
    if not stride:
        w.rows_hint = 5
        w.encode_chunked(); torch.cuda.synchronize()
        assert int(w._counts[0]) == first["rows"] > 5                 # what the caller sees: the outputs are not valid
        third = w.settle()
        assert third["redone"] and third["hint"] == 5 and third["rows"] == first["rows"]
        check(w, "guess too small, run again")
    # the one-lane-per-slice path: exact whatever the guess (slices the guess did not fit go through its second launch)
    for guess in (0, 5, ns):
        w.rows_hint = guess
        w.encode(); torch.cuda.synchronize()
        assert not w.settle()["redone"]
        check(w, f"one lane per slice, guess {guess}")


@pytest.mark.parametrize("stride", [0, 4099])
@pytest.mark.parametrize("parts", [2, 3, 8])
def test_chunked_batch_as_parts_on_streams_of_their_own(avr, oracle, hooks, parts, stride):
    """avr_cabac_encode_chunked_device_parts: the batch cut into parts of consecutive slices, each with its own plan, workspace and
    stream, gives the bytes, final states and statuses of the one call -- asked, sized by the count of the run before, sized too small
    (every part is run again), and with slices left for a second pass (stride 4099 and single bins in contexts nobody else uses)."""
    import torch
    if stride:
        hooks(census_stride=stride)
    rng = np.random.default_rng(411 + parts)
    ns = 150
    slices = []
    for i in range(37):
        r, s = oracle_lib.random_cabac_stream(rng, int(rng.integers(0, 4)) * 5000 + 300 * i, 90, terminate=bool(i % 3))
        slices.append((r, np.concatenate([s, rng.integers(0, 126, ns - 90).astype(np.uint8)])))
    if stride:
        for i, ctx in ((5, 149), (20, 120), (33, 91)):
            if len(slices[i][0]) > 700:
                slices[i][0][600 + 7 * i] = np.uint16((ctx << 1) | (i & 1))
    want = [oracle.cabac_encode(r, s) for r, s in slices]
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    assert 2 <= w.set_parts(parts) <= parts

    def check(what):
        got, status = w.results()
        fs = w.final_states.cpu().numpy().reshape(len(slices), ns)
        for i in range(len(slices)):
            assert status[i] == 0 and got[i] == want[i][0] and fs[i].tobytes() == want[i][1], f"{what}: slice {i}"
        w.out.zero_(); w.out_len.zero_(); w.final_states.zero_()

    w.encode_chunked(); torch.cuda.synchronize()
    first = w.settle()
    assert first["hint"] == 0 and first["parts"] == w.n_parts and not first["redone"]
    check("asked")
    w.encode_chunked(); torch.cuda.synchronize()
    second = w.settle()
    assert second["hint"] == w.rows_hint and second["rows"] == first["rows"]
    check("guessed")
    if not stride:
        assert not second["redone"]
        w.rows_hint = 3
        w.encode_chunked(); torch.cuda.synchronize()
        third = w.settle()
        assert third["redone"] and third["hint"] == 3 and third["rows"] == first["rows"]
        check("guess too small, every part again")


# ------------------------------------------------------------------ K2p: the recoded range coder in three passes

@pytest.mark.parametrize("pass1", ["wave", "lane", "both"])
@pytest.mark.parametrize("seg_len", [0, 1, 3])
def test_range_chunked_random_and_extremes(avr, oracle, hooks, seg_len, pass1):
    """avr_range_encode_chunked_device against the oracle: ragged lengths around the chunk size, adaptive and fixed
    estimators, certain bins (no output for thousands of bins), the most lopsided estimators, empty slices, a record with
    neg 0 (the range collapses to a few bits: the double-precision walk hands the slice to the integer one), and a
    zero-probability bin in the middle of a slice (status, like arithmetic_code.h:116-118).  seg_len 1 / 3 (test hook): the
    passes run segment by segment on two streams, as they do for long slices, with a segment boundary at every (third) chunk.
    pass1: the range recurrence by a wave per slice (what a batch of up to 1 024 slices gets), by a lane per slice (test hook
    k2p_wave=2) and by both in one launch, the longest slices a wave each (3: what larger batches get; here every slice is long --
    the full-size run of config 3 below has both kinds)."""
    lane = {"k2p_wave": 2} if pass1 == "lane" else {"k2p_wave": 3} if pass1 == "both" else {}
    hooks(k2p_seg_len=seg_len, **lane)
    rng = np.random.default_rng(17)
    def rec(b, pos, neg):
        return b | (pos << 1) | (neg << 8)
    slices = [oracle_lib.random_range_stream(rng, n, adaptive=bool(i % 3))
              for i, n in enumerate([0, 1, 7, 8, 9, 1023, 1024, 1025, 2047, 2048, 4097, 30000, 12345, 50000, 3, 20000])]
    slices += [np.array([rec(1, 0x5f, 1)] * 9000, np.uint16), np.array([rec(0, 0x5f, 1)] * 5000, np.uint16),
               np.array([rec(1, 9, 0)] * 6000 + [rec(0, 1, 1)] * 40, np.uint16), np.array([rec(i & 1, 1, 1) for i in range(7000)], np.uint16)]
    collapse = oracle_lib.random_range_stream(rng, 9000)
    collapse[4321] = np.uint16(rec(0, 77, 0))                # bin 0 with neg 0: the new range is range mod 77
    slices.append(collapse)
    zero = oracle_lib.random_range_stream(rng, 5000)
    zero[2500] = np.uint16(1 | (0 << 1) | (9 << 8))
    slices.append(zero)
    w = avr.DeviceWorkload.from_host(1, slices, None, 0)
    w.encode_chunked()
    got, status = w.results()
    for i, r in enumerate(slices[:-1]):
        want, st = oracle.range_encode(r)
        assert st == 0 and status[i] == 0 and got[i] == want, f"slice {i} n={len(r)}"
    assert status[-1] == avr.SLICE_ZERO_PROB and oracle.range_encode(zero)[1] == 1
    # the same slices through the batch API (which picks the three-pass form by the batch's shape: force it)
    hooks(k2p_seg_len=seg_len, k1_path=2, **lane)
    with avr.Batch(0, len(slices), sum(len(r) for r in slices) + 8) as b:
        for r in slices:
            b.add_slice_range(r)
        b.run()
        for i, r in enumerate(slices[:-1]):
            assert b.get(i) == (got[i], 0), f"slice {i}"
        assert b.get(len(slices) - 1)[1] == avr.SLICE_ZERO_PROB


def test_range_chunked_ragged_batch_long_slices_by_waves(avr, oracle):
    """More slices than SIMDs, a few of them far longer than the rest: pass 1 walks a lane per slice and, in the same launch, the
    longest slices by a wave each (k_k2p_ranges_hybrid; which ones: k_k2p_threshold's power-of-two cut).  Every slice against the
    oracle -- the long ones, the ones just below the cut, empty ones."""
    rng = np.random.default_rng(23)
    lengths = [int(x) for x in rng.integers(0, 2500, 1100)]
    for i, n in zip((3, 97, 500, 777, 1023, 1099), (150000, 70000, 33000, 16384 + 5, 8191, 65536)):
        lengths[i] = n
    lengths[10] = 0
    slices = [oracle_lib.random_range_stream(rng, n, adaptive=bool(i % 3)) for i, n in enumerate(lengths)]
    w = avr.DeviceWorkload.from_host(1, slices, None, 0)
    w.encode_chunked()
    got, status = w.results()
    for i, r in enumerate(slices):
        want, st = oracle.range_encode(r)
        assert st == 0 and status[i] == 0 and got[i] == want, f"slice {i} n={len(r)}"


def test_range_chunked_config2_cut_equals_the_serial_kernel(avr, oracle):
    """A 24-slice cut of BASELINE.json's configs[1] in the compress direction: the three-pass path and the one-lane-per-slice
    kernel give the same bytes, and sampled slices equal the oracle's."""
    w = avr.DeviceWorkload.synth(2, 24, avr.KIND_RANGE, 0, 1000)
    w.encode()
    serial, st0 = w.results()
    w.out.zero_()
    w.encode_chunked()
    chunked, st1 = w.results()
    assert not any(st0) and not any(st1) and chunked == serial
    for s in (0, 11, 23):
        cfg, nbh, off, recs, _ = host_synth(avr, 2, 1, avr.KIND_RANGE, 1000, first=s)
        assert chunked[s] == oracle.range_encode(recs[:int(nbh[0])])[0]


@pytest.mark.parametrize("workload,n_slices", [(2, 512), (3, 4096), (4, 16384)])
def test_range_chunked_full_size(avr, oracle, workload, n_slices):
    """The compress direction (K2, three-pass form) at the full size of BASELINE.json configs[1], [2] and [3] under the
    driver: every status 0, the same bytes on a second run, byte equality with the oracle on a seeded sample of slices (the
    longest and the shortest among them) with the reference decoder's round trip, and -- config 2, where the whole batch is
    310 M bins -- every one of the 512 slices against the threaded oracle through a checksum of checksums."""
    w = avr.DeviceWorkload.synth(workload, n_slices, avr.KIND_RANGE, 0, 1000)
    w.encode_chunked()
    got, status = w.results()
    assert not any(status)
    w.out.zero_()
    w.encode_chunked()
    again, _ = w.results()
    assert again == got
    nb = w.n_bins.cpu().numpy()
    sample = sorted(set(np.random.default_rng(100 + workload).integers(0, n_slices, 20).tolist() + [0, n_slices - 1, int(nb.argmax()), int(nb.argmin())]))
    for k, s in enumerate(sample):
        cfg, nbh, off, recs, _ = host_synth(avr, workload, 1, avr.KIND_RANGE, 1000, first=s)
        assert int(nbh[0]) == int(nb[s])
        r = recs[:int(nbh[0])]
        want, st = oracle.range_encode(r)
        assert st == 0 and got[s] == want, f"slice {s}"
        if k < 4:
            assert np.array_equal(oracle.range_decode(got[s], r), r & 1)         # encode -> decode round trip (arithmetic_code.h:218-288)
    if workload == 2:
        import hashlib
        cfg, nbh, off, recs, _ = host_synth(avr, workload, n_slices, avr.KIND_RANGE, 1000)
        parts = [recs[int(off[i]):int(off[i]) + int(nbh[i])] for i in range(n_slices)]
        roff = np.zeros(n_slices + 1, np.uint64)
        roff[1:] = np.cumsum(nbh.astype(np.uint64))
        want, st = oracle.encode_batch(avr.KIND_RANGE, np.concatenate(parts), roff, None, 0, threads=16)
        assert not st.any()
        dig = lambda chunks: hashlib.sha256(b"".join(hashlib.sha256(c).digest() for c in chunks)).hexdigest()
        assert dig(got) == dig(want)
    assert 0.9 < w.total_bins / (8 * sum(len(x) for x in got)) < 1.8


def test_range_chunked_short_region_is_an_overflow_not_a_spill(avr, oracle):
    """A slice whose output region is too small for the byte sums of passes 2 and 3 comes back AVR_SLICE_OVERFLOW and its
    neighbours' bytes are the oracle's: the sums of one slice never spill into the next one's (k_k2p_fits)."""
    rng = np.random.default_rng(5)
    slices = [oracle_lib.random_range_stream(rng, 20000) for _ in range(5)]
    w = avr.DeviceWorkload.from_host(1, slices, None, 0)
    # slice 2's region cut to a few bytes: rebuild out_off with the same total
    import torch
    off = w.out_off.cpu().numpy().astype(np.int64)
    off[3:] -= int(off[3] - off[2]) - 64
    w.out_off.copy_(torch.from_numpy(off))
    w.encode_chunked()
    got, status = w.results()
    assert status[2] == avr.SLICE_OVERFLOW
    for i in (0, 1, 3, 4):
        assert status[i] == 0 and got[i] == oracle.range_encode(slices[i])[0], f"slice {i}"


@pytest.mark.parametrize("mode", ["segments", "whole", "redo3"])
def test_context_chains_in_segments(avr, oracle, hooks, mode):
    """Phase A's per-context state chains cut into segments (k_k1p_chain_seg / _fix: walks from both extreme states, exact because
    the state machine is monotone) against the oracle, with the cases the scheme has to get right by construction: a context with
    a handful of bins per segment (carried through segments as a bit string), a hot context (the walks meet within a chunk or two),
    a context with 70 bins of one value per segment -- the walks do not meet on that, the bit string carries it --, a context parked at pStateIdx 63, contexts that stop occurring half-way through, and slices
    of one segment's length or less.  mode: the shipped path; the start-to-end walk alone (test hook); the summaries of every third
    pair distrusted, so that its lanes walk the earlier segments again chunk by chunk, as they do past a segment with more than 128
    bins whose walks did not meet."""
    if mode == "whole":
        hooks(chain_whole=1)
    elif mode == "redo3":
        hooks(chain_force_redo=3, chain_segments=1)
    else:
        hooks(chain_segments=1)
    rng = np.random.default_rng(606)
    slices = []
    for n in (65536, 131072 + 777, 40000, 8 * 1024, 5 * 1024 + 3, 300000):
        recs, st = oracle_lib.random_cabac_stream(rng, n, 30)
        sel = recs >> 1
        recs = np.where(sel == 5, (6 << 1) | (recs & 1), recs).astype(np.uint16)            # context 5: only the bins placed below
        recs = np.where((sel == 9) & (np.arange(recs.size) > n // 2), (10 << 1) | (recs & 1), recs).astype(np.uint16)   # 9 stops half-way
        seg = max(1, -(-((n + 1023) // 1024) // 8)) * 1024                                     # bins per segment of the chains
        for lo in range(0, n, seg):
            at = lo + np.sort(rng.choice(min(seg, n - lo), size=min(70, n - lo), replace=False))
            recs[at] = np.uint16((5 << 1) | 1)
        rare = rng.choice(n, size=12, replace=False)                                          # context 20: a dozen bins in the whole slice
        recs = np.where((recs >> 1) == 20, (21 << 1) | (recs & 1), recs).astype(np.uint16)
        recs[rare] = (np.uint16(20 << 1) | (rng.integers(0, 2, 12).astype(np.uint16)))
        st = st.copy()
        st[7] = 126                                                                           # parked at pStateIdx 63 (valMPS 0): its bins are all 0 below
        recs = np.where((recs >> 1) == 7, np.uint16(7 << 1), recs).astype(np.uint16)
        slices.append((recs, st))
    w = avr.DeviceWorkload.from_host(0, [r for r, _ in slices], [s for _, s in slices], 0)
    w.encode_chunked()
    got, status = w.results()
    final = w.final_states.cpu().numpy().reshape(len(slices), -1)
    for i, (r, s) in enumerate(slices):
        want = oracle.cabac_encode(r, s)
        assert status[i] == 0 and got[i] == want[0], f"slice {i} ({len(r)} bins) mode {mode}"
        assert final[i][:len(s)].tobytes() == want[1], f"final states of slice {i} mode {mode}"
