"""Multi-GPU path on CPU: world_size-2 gloo run of the sharding and timing plumbing bench.py uses.

The path shards by slice with no data-path collective; the only collectives are
the barrier and the max-over-ranks / sum-over-ranks reductions of the measurement itself.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes, json, os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    import avrecode_ms_amd as avr
    from avrecode_ms_amd.sharding import shard_first_slice, reduce_timing

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank = 6
    first = shard_first_slice(rank, per_rank)
    cfg = avr.synth_config(4, 10, first)
    nb = np.zeros(per_rank, dtype=np.uint32)
    assert avr.lib().avr_synth_count_host(ctypes.byref(cfg), 0, per_rank, nb.ctypes.data) == 0
    t_max, units = reduce_timing(dist, 1.0 + rank, int(nb.sum()), torch.device("cpu"))
    if rank == 0:
        print(json.dumps({{"t_max": t_max, "units": units, "mine": nb.tolist(), "world": world}}))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding(tmp_path, avr):
    import ctypes
    import json
    import numpy as np
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    # single-process view of the same 12 slices
    cfg = avr.synth_config(4, 10, 0)
    nb = np.zeros(12, dtype=np.uint32)
    assert avr.lib().avr_synth_count_host(ctypes.byref(cfg), 0, 12, nb.ctypes.data) == 0
    assert res["world"] == 2
    assert res["mine"] == nb[:6].tolist()                 # rank 0 owns slices 0..5, rank 1 owns 6..11
    assert res["units"] == int(nb.sum())                  # sum over ranks == the unsharded workload
    assert res["t_max"] == 2.0                            # max over ranks of the per-rank time


STRONG_WORKER = textwrap.dedent("""
    import ctypes, json, os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    import avrecode_ms_amd as avr
    from avrecode_ms_amd.sharding import balanced_ranges, lpt_assign

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 400
    cfg = avr.synth_config(3, 4, 0)                       # config 3: log-normal slice lengths, the imbalance case
    nb = np.zeros(total, dtype=np.uint32)
    assert avr.lib().avr_synth_count_host(ctypes.byref(cfg), 0, total, nb.ctypes.data) == 0
    bounds = balanced_ranges(nb, world)                   # every rank computes the same plan from the same counts
    lo, hi = bounds[rank], bounds[rank + 1]
    owner = lpt_assign(nb, world)
    mine_lpt = [i for i in range(total) if owner[i] == rank]
    out = [None] * world
    dist.all_gather_object(out, dict(rank=rank, lo=lo, hi=hi, bins=int(nb[lo:hi].sum()), lpt=mine_lpt,
                                     lpt_bins=int(nb[mine_lpt].sum()), all_bins=int(nb.sum())))
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()
""")


def test_two_rank_strong_split_covers_everything_and_balances(tmp_path, avr):
    """Strong scaling of one batch (BASELINE.json configs[3]: NAL batches sharded across the GPUs of a node): the union of
    the ranks' shards is the unsharded slice set, and the ranks' bin totals are within 2 % of each other -- for the
    contiguous split bench.py --scaling strong uses and for the LPT plan of avr_multi_run."""
    import json
    script = tmp_path / "strong.py"
    script.write_text(STRONG_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29618", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert [r["rank"] for r in res] == [0, 1]
    assert res[0]["lo"] == 0 and res[0]["hi"] == res[1]["lo"] and res[1]["hi"] == 400          # contiguous, disjoint, complete
    total = res[0]["all_bins"]
    assert res[0]["bins"] + res[1]["bins"] == total
    assert abs(res[0]["bins"] - res[1]["bins"]) <= 0.02 * total
    assert sorted(res[0]["lpt"] + res[1]["lpt"]) == list(range(400))                            # LPT: a partition too
    assert abs(res[0]["lpt_bins"] - res[1]["lpt_bins"]) <= 0.02 * total


def test_lpt_and_balanced_ranges_small_cases():
    from avrecode_ms_amd.sharding import balanced_ranges, lpt_assign
    assert lpt_assign([5, 1, 1, 1, 1, 1], 2) == [0, 1, 1, 1, 1, 1]
    assert lpt_assign([], 3) == []
    assert balanced_ranges([1, 1, 1, 1], 2) == [0, 2, 4]
    assert balanced_ranges([10, 1, 1], 2) == [0, 1, 3]
    assert balanced_ranges([], 2) == [0, 0, 0]
    b = balanced_ranges([3] * 10, 4)
    assert b[0] == 0 and b[-1] == 10 and all(x <= y for x, y in zip(b, b[1:]))


@pytest.mark.gpu
def test_two_ranks_on_the_hip_path_code_what_one_rank_codes(tmp_path):
    """The N > 1 launch the driver uses -- torch.distributed.run, one process per rank -- on the HIP path, rehearsed on the
    one GPU of the test box: `bench.py --gpus 2 --backend gloo --scaling strong` as a FRESH child process (both ranks on GPU 0,
    gloo for the measurement's barrier and reductions; on a multi-GPU node the backend is RCCL and each rank has its own
    device, same code otherwise).  One batch of 2048 slices of config 4 is split over the two ranks by bins; the bytes the two
    ranks code must add up to what a single rank codes for the same 2048 slices, with no slice in error.  A rehearsal of the
    launch, the sharding and the reductions -- not a scaling measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--workload", "4", "--slices", "2048", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", "strong"]
                         + common, capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    line = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["slice_status_errors"] == 0
    assert line["config"]["batch_slices"] == 2048 and 0 < line["config"]["slices_per_gpu"] < 2048
    assert line["config"]["h264_bytes_all_gpus"] == single["config"]["h264_bytes_per_gpu"]        # the shards are the batch
    # ... and the same started plainly, the way the driver starts its N = 1 line: `python bench.py --gpus 2` launches its own ranks
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scaling", "strong"] + common,
                           capture_output=True, text=True, env={k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
                           timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    own = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert own["n_gpus"] == 2 and own["slice_status_errors"] == 0
    assert own["config"]["h264_bytes_all_gpus"] == single["config"]["h264_bytes_per_gpu"]
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):                                   # kept for profiles/ (tools/collect_profiles.py)
        with open(os.path.join(out, "rehearsal_2ranks_w4.json"), "w") as f:
            f.write(json.dumps(line) + "\n")


def test_bench_refuses_a_world_size_that_is_not_its_gpus_flag():
    """`bench.py --gpus N` under a launcher that started another number of ranks must not print a line for the wrong N
    (BASELINE.json's metric is quoted at 1/2/4/8 GPUs): it leaves non-zero before touching torch or the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_with_gpus_flag_launches_that_many_ranks_itself():
    """Started plainly with --gpus 2 and no launcher around it, bench.py starts two ranks (torch.distributed.run) and hands on their
    exit code.  No GPU here: each rank leaves with the no-GPU message, and so does the launcher -- non-zero, nothing on stdout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                          # also on a GPU box this test stays a launch test
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--backend", "gloo"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-1500:]
