#!/usr/bin/env python3
"""The 'Measured' table of DESIGN.md from profiles/<round>_bench_*.json (markdown on stdout)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = [("w2", "config 2 K1p, 512 × 605 K bins (the headline)"), ("w3", "config 3 K1p, 4 096 ragged slices"),
        ("w4", "config 4 K1p, 16 384 × 243 K"), ("w5", "config 5 K1 serial, 1 Mi × 1.5 K"),
        ("w2_resolved", "config 2, stage 2 alone (resolved codes in)"), ("w2_s128", "config 2 slices, 128 of them"),
        ("w2_s128_whole_chains", "… with whole-slice chains (test hook)"), ("w5_ref_form", "config 5 K1, the coder as cabac_code.h writes it (test hook; round 2's form)"), ("w5_lds_rows", "… with the LDS-row emitter (test hook)"), ("w5_norm_lds_rows", "config 5 K1, the shipped form with the LDS-row emitter (test hook k1_emit_lds=2)"),
        ("w2_k2", "config 2 K2p (compress)"), ("w3_k2", "config 3 K2p"), ("w4_k2", "config 4 K2p"), ("w5_k2", "config 5 K2 serial")]
print("| workload | ms / step | GB/s of H.264 | roofline frac (algorithmic bytes / 8 TB/s) | CPU one socket: reference / restatement, MB/s | GPU ÷ CPU | e2e GB/s (PCIe in and out) |")
print("|---|---|---|---|---|---|---|")
for key, label in rows:
    p = os.path.join(ROOT, "profiles", f"{rnd}_bench_{key}.json")
    if not os.path.exists(p):
        continue
    j = json.loads(open(p).read().strip().splitlines()[-1])
    c, e = j.get("cpu_baseline") or {}, j.get("e2e") or {}
    cpu = f"{c['value'] / 1e6:.0f} / {c.get('port_value', 0) / 1e6:.0f} ({c.get('parity_vs_gpu')})" if c else "—"
    ratio = f"{j['gpu_over_cpu']:.0f} × / {j['value'] / c['port_value']:.0f} ×" if c and c.get("port_value") else "—"
    print(f"| {label} | {j['ms_per_step']:.3f} | {j['value'] / 1e9:.2f} | {j['roofline']['frac']:.4f} | {cpu} | {ratio} | {e.get('value', 0) / 1e9:.2f} |" if e else
          f"| {label} | {j['ms_per_step']:.3f} | {j['value'] / 1e9:.2f} | {j['roofline']['frac']:.4f} | {cpu} | {ratio} | — |")
