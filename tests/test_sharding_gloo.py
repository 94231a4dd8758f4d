"""CPU: the N>1 path (one process per shard, env_offset sharding, barrier + max-time reduction) on gloo, world_size 2."""
import os
import subprocess
import sys

from helpers import REPO


def test_two_rank_gloo_shards_concatenate_to_the_whole():
    import emu_harness

    emu_harness.build()  # build once, before two ranks race for it
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(REPO, "tests", "tools", "shard_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARD_OK" in out.stdout
