#!/usr/bin/env python3
"""Turn one tools/gpu_final_r4.sh run (gpurun_out/TAG/, parts a, b, c) into the committed summaries under profiles/:
   python tools/collect_profiles.py TAG ROUND        e.g.  fin3 r03
kernel-trace stats and the PMC counter files are copied as they are (kernels of this library only for
the counters); pmc_traffic.json is what bench.py's roofline.traffic reads."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(hits, key=os.path.getmtime)                       # (a directory may hold an earlier run's file too: the latest)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("avr::", "")


SQ_STEPS = 2         # ... and the SQ counter pass with --steps 1 --warmup 1
PMC_STEPS = 4        # the counter passes run with --steps 3 --warmup 1: four identical steps


def per_kernel(path, counter, steps_key="avr::"):
    """counter value per STEP, per kernel of this library: the sum over all launches of the run / the run's steps
    (a kernel launched twice per step -- the coder and its hand-back launch -- counts twice, one made once per process
    -- the state-walk table -- a quarter)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "avr::" in r["Kernel_Name"]
            and "synth" not in r["Kernel_Name"] and "context_" not in r["Kernel_Name"] and "states_permute" not in r["Kernel_Name"]
            and "pack_tiles" not in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / PMC_STEPS for k, v in by.items()}


LANE_CHUNK_READERS = ("k_k1p_local", "k_k1p_replay", "k_k1p_chain", "k_k1p_ctxchain", "k_k2p_code", "k_k2p_ranges", "k_cabac_encode", "k_range_encode")
traffic = {}
sys.path.insert(0, ROOT)
import avrecode_ms_amd as _avr                                   # the library the counters were collected on (the tree is as it was pushed)
LIB_SHA = _avr.library_sha256()
for w, key, label in ((2, "cabac_chunked_w2_s512", "K1p pipeline, all launches of one step"),
                      (3, "cabac_chunked_w3_s4096", "K1p pipeline (config 3: 4 096 ragged slices), all launches of one step"),
                      (4, "cabac_chunked_w4_s16384", "K1p pipeline (config 4: 16 384 slices), all launches of one step"),
                      (5, "cabac_serial_w5_s1048576", "k_k1_census + k_k1p_densemap + k_cabac_encode<tiled>: all launches of one step"),
                      ("5lds", "cabac_serial_w5_s1048576_test_hook_k1_emit_lds", "the same with the reference-form coder and the LDS-row emitter (test build, hook k1_emit_lds): a measured variant"),
                      ("5ref", "cabac_serial_w5_s1048576_test_hook_k1_form_ref", "the same with the coder as cabac_code.h writes it (test build, hook k1_form_ref): round 2's shipped form")):
    if not glob.glob(os.path.join(src, f"w{w}_stats")):
        continue
    shutil.copy(one(f"w{w}_stats/**/*kernel_stats.csv"), os.path.join(dst, f"{rnd}_w{w}_kernel_stats.csv"))
    for what in ("fetch", "write"):
        rows = [r for r in csv.DictReader(open(one(f"w{w}_{what}/**/*counter_collection.csv"))) if "avr::" in r["Kernel_Name"]]
        with open(os.path.join(dst, f"{rnd}_w{w}_pmc_{what}.csv"), "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
    fetch = per_kernel(one(f"w{w}_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(one(f"w{w}_write/**/*counter_collection.csv"), "WRITE_SIZE")
    fsum, wsum = sum(fetch.values()), sum(write.values())
    # profiles/r03_hbm_counter_calibration.txt: a lane that walks a chunk of its own reads 1.2 .. 1.5 x what FETCH_SIZE says, not 2 x
    fcal = sum(v * (1.5 if any(k.startswith(n) for n in LANE_CHUNK_READERS) else 2.0) for k, v in fetch.items())
    traffic[key] = {
        "kernel": label, "library_sha256": LIB_SHA, "FETCH_SIZE_KB_raw_sum": fsum, "WRITE_SIZE_KB_raw_sum": wsum,
        "fetch_bytes_corrected_x2": fsum * 1024 * 2, "write_bytes": wsum * 1024,
        "hbm_bytes_per_launch": fsum * 1024 * 2 + wsum * 1024,
        "fetch_bytes_calibrated": fcal * 1024, "hbm_bytes_per_launch_calibrated": fcal * 1024 + wsum * 1024,
        "per_kernel_FETCH_SIZE_KB": fetch, "per_kernel_WRITE_SIZE_KB": write,
        "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), all launches of the run / its 4 steps; hbm_bytes_per_launch: "
                "FETCH_SIZE doubled throughout (the guide's gfx950 wide-read rule, an upper bound here); ..._calibrated: x1.5 for "
                "the kernels that read a chunk per lane, x2 for the rest (profiles/r03_hbm_counter_calibration.txt)"}
    # issue / wait counters of the same workload, one line per kernel
    sq = one(f"w{w}_sq/**/*counter_collection.csv")
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(sq)):
        if "avr::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]
    with open(os.path.join(dst, f"{rnd}_w{w}_pmc_sq.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel"] + names)
        for k in sorted({k[0] for k in agg}):
            wr.writerow([k] + ["%.6g" % (agg[(k, c)] / SQ_STEPS) for c in names])   # per step: all launches of the run / its steps
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
for d in ("w2k2", "w4k2"):                               # kernel stats of the compress direction
    hits = glob.glob(os.path.join(src, f"{d}_stats/**/*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(max(hits, key=os.path.getmtime), os.path.join(dst, f"{rnd}_{d}_kernel_stats.csv"))
benches = sorted(os.path.basename(p)[len("bench_"):-len(".json")] for p in glob.glob(os.path.join(src, "bench_*.json")))
for b in benches:
    shutil.copy(os.path.join(src, f"bench_{b}.json"), os.path.join(dst, f"{rnd}_bench_{b}.json"))
for extra in ("cli_timing.txt", "ubench_issue_rate.txt", "ubench_lds_chain.txt", "ubench_read_patterns.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{rnd}_{extra}"))
reh = os.path.join(ROOT, "gpurun_out", "rehearsal_2ranks_w4.json")
if os.path.exists(reh):
    shutil.copy(reh, os.path.join(dst, f"{rnd}_rehearsal_2ranks_w4.json"))
for key, t in traffic.items():
    print(key, "HBM bytes per step: %.3f GB" % (t["hbm_bytes_per_launch"] / 1e9))
for b in benches:
    p = os.path.join(dst, f"{rnd}_bench_{b}.json")
    if os.path.exists(p):
        j = json.loads(open(p).read().strip().splitlines()[-1])
        print(b, "%.3f ms  %.2f GB/s  frac %.4f" % (j["ms_per_step"], j["value"] / 1e9, j["roofline"]["frac"]),
              ("cpu %.0f MB/s x%.1f %s" % (j["cpu_baseline"]["value"] / 1e6, j["gpu_over_cpu"], j["cpu_baseline"].get("parity_vs_gpu")))
              if "cpu_baseline" in j else "")
