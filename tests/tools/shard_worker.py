"""
Worker of tests/test_sharding_gloo.py: one rank of a world_size-N gloo job.  Each rank steps its shard of one
logical vector of envs (env_offset = rank * per_rank) -- the N>1 layout of bench.py -- on the CPU emulation of the
kernels, then the per-step frame CRCs / rewards are all-gathered and rank 0 checks them against a single-handle run.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "emu")):
    sys.path.insert(0, p)

import emu_harness  # noqa: E402
from helpers import action_stream, rollout  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank, steps = 3, 40
    total = per_rank * world
    acts = action_stream(total, steps, seed=12)
    lo = rank * per_rank
    env = emu_harness.EmuEnv(per_rank, "coinrun", rand_seed=31, env_offset=lo)
    dist.barrier()
    mine = rollout(env, [a[lo:lo + per_rank] for a in acts])
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    ok = True
    for k in ("crc", "rew", "first", "level_seed"):
        x = torch.from_numpy(mine[k].astype(np.float64))
        parts = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        if rank == 0:
            whole = rollout(emu_harness.EmuEnv(total, "coinrun", rand_seed=31), acts) if k == "crc" else whole  # noqa: F821
            got = np.concatenate([p.numpy() for p in parts], axis=1)
            ok = ok and np.array_equal(got, whole[k].astype(np.float64))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("SHARD_OK" if ok else "SHARD_MISMATCH", flush=True)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
