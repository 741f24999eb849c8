"""Multi-GPU path on CPU: world_size-2 gloo run of the sharding and timing plumbing bench.py uses.

The path shards by slice with no data-path collective; the only collectives are
the barrier and the max-over-ranks / sum-over-ranks reductions of the measurement itself.
"""
import os
import subprocess
import sys
import textwrap

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
