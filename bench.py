#!/usr/bin/env python3
"""bench.py -- H.264 bytes/s of the arithmetic re-encode hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 2|3|4|5] [--kind cabac|range]

A step is one pass of the hot path over one batch of synthetic slices that is already
resident in HBM (wave-interleaved tiles, generated on the device from seeded streams,
SURVEY.md 8(d)).  Default workload: BASELINE.json configs[1] -- 512 slices of a 1080p30
clip, 1 slice per frame -- on every rank (weak scaling: rank r owns slices
[r*512, (r+1)*512) of the seeded stream space; no data-path collective).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (the encode kernel): algorithmic bytes per launch / average
                launch duration (HIP events on the launching stream) against HBM peak
  cpu_baseline  the same coding done on the host CPU cores over a bounded sample of the same
                workload: oracle/_ref (the reference's own arithmetic_code.h) when it has been
                built, else the oracle restatement; GPU output is byte-compared with it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
DEFAULT_SLICES = {2: 512, 3: 4096, 4: 16384, 5: 1 << 20}
K1P_KERNELS = ("K1p: k_k1p_{census,densemap,local,chain_seg + chain_fix (up to 49 152 (slice, context) pairs, long slices: config 2) or "
               "ctxchain (beyond: configs 3, 4),replay,b2,c,d} + the idle serial fallback (one step = all of them; largest: k_k1p_local "
               "and k_k1p_replay, about equal)")
WORKLOAD_NAME = {
    2: "config2: 1080p30 CABAC clip, 1 slice/frame, 512 frames (synthetic, 8160 macroblocks per slice)",
    3: "config3: 16 files x 256 slices, log-normal slice sizes (synthetic)",
    4: "config4: 4K60, 8 slices/frame, 16384 slices (synthetic, 4080 MB/slice)",
    5: "config5: residual-only streams, 1M slices x 64 4x4 blocks (synthetic)",
}


def socket0_cpus():
    """The CPUs of one socket (the lowest `physical id` of /proc/cpuinfo) among those this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    try:
        sock, cur = {}, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cur = int(line.split(":")[1])
            elif line.startswith("physical id") and cur is not None:
                sock[cur] = int(line.split(":")[1])
        ids = sorted({sock[c] for c in allowed if c in sock})
        if ids:
            return [c for c in allowed if sock.get(c) == ids[0]], len(ids)
    except (OSError, ValueError):
        pass
    return allowed, 1


def host_sample(avr, workload, kind, n_slices, first_slice, max_bins):
    """The first m slices of the workload generated on the host: (m, n_bins, compact records of `kind`, offsets, init states, cfg)."""
    import ctypes
    import numpy as np
    L = avr.lib()
    cfg = avr.synth_config(workload, 1000, first_slice)
    nb_all = np.zeros(n_slices, dtype=np.uint32)
    L.avr_synth_count_host(ctypes.byref(cfg), kind, n_slices, nb_all.ctypes.data)
    cum = np.cumsum(nb_all.astype(np.int64))
    m = int(min(n_slices, max(1, np.searchsorted(cum, max_bins))))
    nb = nb_all[:m]
    off = np.zeros(m + 1, dtype=np.uint64)
    off[1:] = np.cumsum((nb.astype(np.uint64) + 7) // 8 * 8)

    def gen(k):
        recs = np.zeros(int(off[-1]), dtype=np.uint16)
        states = np.zeros(m * cfg.n_states, dtype=np.uint8)
        L.avr_synth_generate_host(ctypes.byref(cfg), k, m, off.ctypes.data, recs.ctypes.data, states.ctypes.data)
        # compact copy without the padding, so that off[i+1] - off[i] is the slice length
        parts = [recs[int(off[i]):int(off[i]) + int(nb[i])] for i in range(m)]
        return (np.concatenate(parts) if parts else np.zeros(0, np.uint16)), states
    roff = np.zeros(m + 1, dtype=np.uint64)
    roff[1:] = np.cumsum(nb.astype(np.uint64))
    return m, nb, gen, roff, cfg


def cpu_baseline(avr, workload, kind, n_slices, first_slice, gpu_bytes_of, budget_s=20.0):
    """Time the CPU checker on a bounded sample, on the cores of ONE socket, and byte-compare the GPU output with it.

    kind "reference" (oracle/_ref, built from the reference's own arithmetic_code.h): K1 = ref_cabac_encode, the reference
    coder under the restated CABAC layer; K2 = ref_model_range_encode, the reference coder driven the way recode.cpp drives
    it -- std::function probability over a std::map<tuple<const void*, int, int>, estimator> (recode.cpp:823-827, 1037-1052).
    The build's own C restatement (oracle/, "port") is timed beside it."""
    import ctypes
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle = oracle_lib.load_oracle()
    ref = oracle_lib.load_ref()
    P = oracle_lib.ptr
    all_cpus = sorted(os.sched_getaffinity(0))
    cpus, sockets = socket0_cpus()
    cores = len(cpus)
    rate = 40e6 if kind == avr.KIND_CABAC else 15e6         # rough bins/s per thread of the reference-kind coder
    m, nb, gen, roff, cfg = host_sample(avr, workload, kind, n_slices, first_slice, int(budget_s * rate * cores))
    recs, states = gen(kind)                                 # what the GPU codes
    keyed = gen(avr.KIND_CABAC)[0] if kind == avr.KIND_RANGE else recs   # (bin, selector): the reference model's input
    out_off = np.zeros(m + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(nb.astype(np.uint64) + 16)
    out = np.zeros(int(out_off[-1]), dtype=np.uint8)
    out_len = np.zeros(m, dtype=np.uint32)
    status = np.zeros(m, dtype=np.int32)
    is_cabac = kind == avr.KIND_CABAC

    def run_port(threads, count=m):
        rc = oracle.L.avr_oracle_encode_batch(
            ctypes.c_int(kind), P(recs), P(roff), ctypes.c_size_t(count), P(states if is_cabac else None),
            ctypes.c_size_t(cfg.n_states if is_cabac else 0), P(out), P(out_off), P(out_len), P(status), ctypes.c_int(threads))
        assert rc == 0

    def ref_range(lo, hi):
        if hi <= lo:
            return
        at = lambda a, k, w: ctypes.c_void_p(a.ctypes.data + w * k)
        if is_cabac:
            ref.L.ref_cabac_encode_batch(P(keyed), at(roff, lo, 8), ctypes.c_size_t(hi - lo), at(states, lo * cfg.n_states, 1),
                                         ctypes.c_size_t(cfg.n_states), P(out), at(out_off, lo, 8), at(out_len, lo, 4))
        else:
            ref.L.ref_model_range_encode_batch(P(keyed), at(roff, lo, 8), ctypes.c_size_t(hi - lo), P(out), at(out_off, lo, 8),
                                               at(out_len, lo, 4))

    def run_ref(threads, count=m):
        bounds = np.linspace(0, count, threads * 4 + 1).astype(int)
        with ThreadPoolExecutor(threads) as ex:              # ctypes calls release the GIL
            list(ex.map(lambda k: ref_range(int(bounds[k]), int(bounds[k + 1])), range(threads * 4)))

    def best_of(runner, threads):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            runner(threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            if dt > budget_s / 2:
                break
        return best

    def result():
        return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(m)]

    res = {}
    try:
        os.sched_setaffinity(0, cpus)                        # one socket: north_star's "single-socket CPU reference"
        t_port = best_of(run_port, cores)
        port_bytes = result()
        total = sum(len(b) for b in port_bytes)
        if ref is not None:
            t_ref = best_of(run_ref, cores)
            cpu_bytes = result()
        else:
            t_ref, cpu_bytes = None, port_bytes
        # one thread (the reference itself is single-threaded, recode.cpp:129) on a smaller cut
        m1 = max(1, min(m, int(np.searchsorted(np.cumsum(nb.astype(np.int64)), 30e6)) + 1))
        bytes1 = sum(len(b) for b in cpu_bytes[:m1])
        t0 = time.perf_counter()
        (ref_range(0, m1) if ref is not None else run_port(1, m1))
        dt1 = time.perf_counter() - t0
        t_all = None
        if sockets > 1:                                      # the whole host, for the record
            os.sched_setaffinity(0, all_cpus)
            t_all = best_of(run_ref if ref is not None else run_port, len(all_cpus))
    finally:
        os.sched_setaffinity(0, all_cpus)
    gpu = gpu_bytes_of(m)
    parity = "bit-exact" if gpu == cpu_bytes and gpu == port_bytes else "MISMATCH"
    cpu_model = "unknown"
    try:
        names = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        cpu_model = names[0] if names else "unknown"
    except OSError:
        pass
    kind_name, best = ("reference", t_ref) if ref is not None else ("port", t_port)
    res.update({
        "value": total / best, "unit": "bytes/s", "cores": cores, "sockets_used": 1, "kind": kind_name,
        "cpu_model": cpu_model, "sockets_in_host": sockets,
        "sample": f"first {m} of {n_slices} slices of the same workload ({int(nb.astype(np.int64).sum())} bins, "
                  f"{total} H.264 bytes), best of <=3, one slice per task, threads pinned to the {cores} CPUs of socket 0",
        "port_value": total / t_port,
        "single_thread_value": bytes1 / dt1,
        "all_sockets_value": (total / t_all) if t_all else None, "all_sockets_cores": len(all_cpus) if t_all else None,
        "parity_vs_gpu": parity, "parity_slices": m,
    })
    return res


def e2e_block(avr, workload, kind, n_slices, first_slice, device, rounds=12, objects=3, records8=False):
    """PCIe-inclusive rate of the batch API (what test.cpp:52-54 of the reference reports: end-to-end bytes per second): host
    records in pinned memory in, host bytes out, `objects` avr_batch objects used in turn (submit / wait), so that one
    batch's H2D runs under another's kernels.  Never `value`: the bench line's value has its inputs resident in HBM."""
    import numpy as np
    m, nb, gen, roff, cfg = host_sample(avr, workload, kind, n_slices, first_slice, 80_000_000)
    m = min(m, 16384)
    recs, states = gen(kind)
    total_bins = int(nb[:m].astype(np.int64).sum())
    n_ctx8 = None
    if records8:
        # what the recorder of INTEGRATION.md ships when a file has at most 126 contexts: ids dense by first appearance, one byte a record
        sel = (recs[:int(roff[m])] >> 1).astype(np.int64)
        ctx, first = np.unique(sel[sel < 1024], return_index=True)
        ctx = ctx[np.argsort(first)]
        n_ctx8 = int(ctx.size)
        if n_ctx8 > avr.MAX_STATES8:
            return {"skipped": f"{n_ctx8} contexts: more than one-byte records can name"}
        lut = np.full(1026, 255, dtype=np.int64)
        lut[ctx] = np.arange(n_ctx8)
        lut[1024], lut[1025] = avr.SEL8_BYPASS, avr.SEL8_TERMINATE
        recs = ((lut[sel] << 1) | (recs[:int(roff[m])] & 1)).astype(np.uint8)
        states = np.ascontiguousarray(states.reshape(-1, cfg.n_states)[:, ctx]).reshape(-1)
    ns = n_ctx8 if records8 else cfg.n_states
    bs = [avr.Batch(device, m, total_bins + 8) for _ in range(objects)]
    try:
        for b in bs:
            for i in range(m):
                r = recs[int(roff[i]):int(roff[i + 1])]
                if records8:
                    b.add_slice_cabac8(r, states[i * ns:(i + 1) * ns])
                elif kind == avr.KIND_CABAC:
                    b.add_slice_cabac(r, states[i * ns:(i + 1) * ns])
                else:
                    b.add_slice_range(r)
        for b in bs:                                         # first run of an object (it asks the device for the context count)
            b.submit(); b.wait()
        out_bytes = sum(len(bs[0].get(i)[0]) for i in range(m))
        t0 = time.perf_counter()
        for i in range(rounds):
            if i >= objects:
                bs[i % objects].wait()
            bs[i % objects].submit()
        for i in range(rounds, rounds + objects):
            bs[i % objects].wait()
        t = time.perf_counter() - t0
        tm = bs[0].timings()
    finally:
        for b in bs:
            b.close()
    return {"value": rounds * out_bytes / t, "unit": "bytes/s", "ms_per_batch": 1e3 * t / rounds, "slices_per_batch": m,
            "bins_per_batch": total_bins, "h264_bytes_per_batch": out_bytes, "batch_objects": objects, "rounds": rounds,
            "input": (f"one-byte records (bin | dense selector << 1, {n_ctx8} contexts) in pinned host memory (1 B per bin)" if records8
                      else "uint16 records in pinned host memory (2 B per bin)"), "output": "coded bytes in host memory",
            "h2d_ms_alone": tm["h2d_ms"], "note": "PCIe-inclusive; never the bench line's value"}


def launch_ranks(n):
    """`python bench.py --gpus N ...` with no launcher around it: N ranks of this same command line under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1 at a port that is free now).  Returns the child's exit code; rank 0's JSON
    line goes to this process's stdout, everything else the ranks print to stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for out_line in child.stdout:                            # only rank 0 prints to stdout: the one JSON line
        sys.stdout.write(out_line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--part-weights", default="", help="experiments: the parts' shares of the batch's chunks, e.g. 40,60 (default: equal)")
    ap.add_argument("--parts", type=int, default=0,
                    help="the intra-slice parallel K1 path: the batch as this many parts of consecutive slices at once, each on a stream of "
                         "its own (avr_cabac_encode_chunked_device_parts); 0 = by the batch's size (three for a batch of a few rounds of "
                         "workgroups, one for a large one), 1 = one call")
    ap.add_argument("--kind", default="cabac", choices=["cabac", "range"])
    ap.add_argument("--slices", type=int, default=0, help="slices per rank (default: the configuration's own count)")
    ap.add_argument("--path", default="auto", choices=["auto", "serial", "chunked"],
                    help="K1 mapping: one lane per slice, or the intra-slice parallel kernels (auto: chunked "
                         "when the batch has too few slices to fill the chip)")
    ap.add_argument("--records", default="bins", choices=["bins", "resolved"],
                    help="bins: (bin, context) records + state tables, the full K1 (default); resolved: time only the "
                         "arithmetic-coding stage of K1p from (bin, state) codes resolved beforehand")
    ap.add_argument("--full-context-table", action="store_true",
                    help="keep the contexts as numbered by the stream (default: renumber the batch onto the contexts it uses)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank codes its own copy of the workload's slice count (default); strong: ONE batch of the "
                         "workload's slice count is split over the ranks (contiguous ranges of near-equal bin totals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive block (host records in, host bytes out)")
    ap.add_argument("--test-hook", action="append", default=[], metavar="NAME=VALUE",
                    help="MEASUREMENT ONLY: run on the -DAVR_TEST_HOOKS build of the library with this hook set (csrc/avr_internal.h); "
                         "the line says so in config.test_hooks.  The product library has no such switches")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL; gloo lets several ranks rehearse on one GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # Started plainly with --gpus N: this process becomes the launcher and nothing else.  It has touched neither torch nor the
            # GPU (a process that has initialised the GPU must not start the ranks by exec), starts N ranks as a child through
            # torch.distributed.run, relays what they print and leaves with the child's exit code.
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher and the flag disagree "
                         "(one rank per GPU; start it as `python bench.py --gpus N` or through torch.distributed.run with --nproc-per-node N)")

    import torch
    import torch.distributed as dist
    import avrecode_ms_amd as avr
    from avrecode_ms_amd.sharding import balanced_ranges, reduce_timing, shard_first_slice

    hooks_cm = None
    if args.test_hook:
        hooks_cm = avr.test_hooks(**{h.split("=")[0]: int(h.split("=")[1]) for h in args.test_hook})
        hooks_cm.__enter__()                                 # for the rest of the process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or avr.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":                       # rehearsal: ranks may share a GPU
        local_rank %= torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    kind = avr.KIND_CABAC if args.kind == "cabac" else avr.KIND_RANGE

    n_slices = args.slices or DEFAULT_SLICES[args.workload]
    first = shard_first_slice(rank, n_slices)
    batch_slices = n_slices
    if args.scaling == "strong" and world > 1:
        # One batch, sharded: every rank counts the bins of all slices on its own GPU (cheap next to generating them; same
        # counts on every rank, so the same plan with nothing exchanged) and takes one contiguous range of the split.
        import ctypes
        cfg_all = avr.synth_config(args.workload, 1000, 0)
        nb_all = torch.zeros(n_slices, dtype=torch.int32, device=dev)
        avr._check(avr.lib().avr_synth_count_device(local_rank, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                           ctypes.byref(cfg_all), kind, n_slices, nb_all.data_ptr()))
        bounds = balanced_ranges(nb_all.cpu().numpy(), world)
        first, n_slices = bounds[rank], bounds[rank + 1] - bounds[rank]
    w = avr.DeviceWorkload.synth(args.workload, n_slices, kind, local_rank, 1000, first)
    declared_states = w.n_states

    path = args.path
    if path == "auto":        # one lane per slice needs >= ~64 slices per SIMD-wave-slot to fill 256 CUs
        path = "chunked" if (n_slices <= 32768 and w.total_bins // max(n_slices, 1) >= 8192) else "serial"
    # The step contains everything a batch needs: both K1 paths renumber the batch onto the contexts it uses inside the
    # call (the intra-slice parallel kernels in their census pass; the one-lane-per-slice kernel through a census of its
    # own and a look-up as records are loaded), nothing happens before the first step.
    prepass_ms = 0.0
    if args.full_context_table:
        os.environ["AVR_NO_DENSE"] = "1"
    step = w.encode_chunked if path == "chunked" else w.encode
    weights = [float(x) for x in args.part_weights.split(",")] if args.part_weights else None
    n_parts = (w.set_parts(len(weights) if weights else args.parts, weights)
               if (path == "chunked" and kind == avr.KIND_CABAC and args.records != "resolved") else 1)
    if args.records == "resolved":
        if kind != avr.KIND_CABAC:
            raise SystemExit("--records resolved applies to the CABAC kernel")
        path = "chunked-stage2"
        codes = w.resolve()                  # untimed: the recorder is assumed to have produced the codes
        step = lambda: w.encode_resolved(codes)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # K1 is sized by the context count the previous run of the workload reported (avr_cabac_encode_*_device_hinted, the way avr_batch
    # runs from its second batch on): the first warm-up step asks the device and waits, the steps after it wait for nothing; the
    # count every timed step reported is looked at after the timed region (settle) and a guess that was too small fails the line.
    settle = getattr(w, "settle", None) if kind == avr.KIND_CABAC and args.records != "resolved" else None
    for i in range(args.warmup):
        step()
        if settle and i == 0:
            torch.cuda.synchronize(dev)
            settle()
    sync_all()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()               # the kernels are launched on torch's current stream
        step()
        ends[i].record()
    sync_all()
    # the per-batch renumbering of the serial path (measured once above) belongs to every step
    elapsed = time.perf_counter() - t0 + 1e-3 * prepass_ms * args.steps
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / max(args.steps, 1) + prepass_ms

    failed = False
    sized = settle() if settle else None
    if sized and sized["redone"]:
        raise SystemExit("bench.py: a timed step was sized by a context count that was too small and had to be run again: %r" % (sized,))
    out_bytes = w.output_bytes()
    status_bad = int((w.status != 0).sum().item())
    t_max, total_bytes = reduce_timing(dist if world > 1 else None, elapsed, out_bytes * args.steps, red_dev)

    if rank == 0:
        algo = w.algorithmic_bytes()
        if args.records == "resolved":       # 1 byte per bin in, no state tables
            algo = w.total_bins + out_bytes + 16 * n_slices
        achieved = algo / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per step by PMC: not measured by this run -- the figure of the committed rocprofv3 --pmc passes over this
        # very command (profiles/pmc_traffic.json, made by tools/collect_profiles.py), or null
        # (`traffic`: FETCH_SIZE doubled throughout, the guide's gfx950 rule and an upper bound here; `traffic_calibrated`: with the
        # factors measured on known byte counts in this library's access patterns, profiles/r03_hbm_counter_calibration.txt)
        traffic, traffic_cal, traffic_source = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.test_hook:
            try:
                t = json.load(open(tpath)).get(f"{args.kind}_{path}_w{args.workload}_s{n_slices}", {})
                # the counters belong to the library they were collected on: a figure of another build is not this run's traffic
                if t.get("library_sha256") == avr.library_sha256():
                    traffic, traffic_cal = t.get("hbm_bytes_per_launch"), t.get("hbm_bytes_per_launch_calibrated")
                    traffic_source = "profiles/pmc_traffic.json" if traffic is not None else None
                elif t:
                    traffic_source = "none: profiles/pmc_traffic.json holds this command's counters for another build of the library"
            except Exception:
                traffic = None
        line = {
            "metric": "H.264 bytes/s recompressed (CABAC re-encode of recorded bin streams, bit-exact vs CPU)",
            "value": total_bytes / t_max, "unit": "bytes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / max(args.steps, 1),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "u32" if kind == avr.KIND_CABAC else "u64", "data": "synthetic",
            "config": {"workload": WORKLOAD_NAME[args.workload], "kernel": "K1 cabac_encode" if kind == avr.KIND_CABAC else "K2 range_encode",
                       "slices_per_gpu": n_slices, "batch_slices": batch_slices if args.scaling == "strong" else n_slices * world,
                       "bins_per_gpu": w.total_bins, "h264_bytes_per_gpu": out_bytes,
                       "h264_bytes_all_gpus": total_bytes // max(args.steps, 1),
                       "n_states": w.n_states, "n_states_declared": declared_states, "test_hooks": args.test_hook or None,
                       "layout": "slice-major" if path == "chunked" else "wave-interleaved tiles", "path": path, "records": args.records, "parallelism": f"slice-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_calibrated": traffic_cal, "traffic_source": traffic_source,
                         "kernel": (K1P_KERNELS if path == "chunked" else
                                    "k_k1_census (1-in-16 sample) + k_k1p_densemap + k_cabac_encode<tiled> + its hand-back launch (one step = all of them)")
                         if kind == avr.KIND_CABAC else
                         ("K2p: k_k2p_ranges_wave (up to 1 024 slices: a wave per slice) or k_k2p_ranges_fp (a lane per slice) -- the range recurrence: the wall -- + its idle hand-over + k_k2p_fits + "
                          "k_k2p_zero + k_k2p_code + k_k2p_finish"
                          if path == "chunked" else "k_range_encode<tiled>"),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo,
                         "bins_per_s": w.total_bins / (kernel_ms * 1e-3)},
            "slice_status_errors": status_bad,
            "parts": n_parts,
            "sized_by": (None if not sized else {"context_rows_guessed": sized["hint"], "context_rows_needed": sized["rows"],
                                                 "note": "0 guessed = the step asked the device and waited (no warm-up step before it)"}),
            "prepass_ms_in_step": prepass_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_bytes_of(m):
                return w.results()[0][:m]
            line["cpu_baseline"] = cpu_baseline(avr, args.workload, kind, n_slices, first, gpu_bytes_of)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        if world == 1 and not args.no_e2e and args.records == "bins":
            line["e2e"] = e2e_block(avr, args.workload, kind, n_slices, first, local_rank)
            if kind == avr.KIND_CABAC:                           # the same batches as one-byte records (AVR_KIND_CABAC8): half the bytes over PCIe
                line["e2e_cabac8"] = e2e_block(avr, args.workload, kind, n_slices, first, local_rank, records8=True)
        print(json.dumps(line), flush=True)
        if status_bad or line.get("cpu_baseline", {}).get("parity_vs_gpu") == "MISMATCH":
            failed = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: GPU output differs from the CPU checker, or a slice came back with an error status")


if __name__ == "__main__":
    main()
