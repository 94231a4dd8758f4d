"""
GPU (one is enough): the two multi-GPU launch paths of bench.py, executed end to end on the one visible device.

  * one process per GPU under torch.distributed.run (the driver's form): --dry-multi puts every rank on device 0 and lets gloo
    carry the barrier and the MAX reduction, everything else -- env_offset sharding, the per-rank handles, the JSON line -- is the
    code a real N-GPU run executes (reference analogue: the env index -> seed derivation of src/vecgame.cpp:301-314);
  * one handle over G devices (--devices-in-process G, the num_devices option) with PROCGEN_AMD_FAKE_DEVICES=1.

Both must report n_gpus = 2 and their sharding, and the shards' observation CRCs must be those of the corresponding slices of ONE
single-device handle stepped with the same actions: sharding changes where envs run, not what they do.  No scaling number is
expected from either.
"""
import json
import os
import socket
import subprocess
import sys
import zlib

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, STEPS, WARM = 2048, 12, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _single_handle_crcs(acts, slices):
    from procgen_amd import ProcgenGym3Env

    n = acts.shape[1]
    env = ProcgenGym3Env(n, "coinrun", rand_seed=23)
    env.observe()
    for a in acts:
        env.act(a)
    _, ob, _ = env.observe()
    out = [zlib.crc32(ob["rgb"][s].tobytes()) for s in slices]
    env.close()
    return out


@pytest.mark.gpu
def test_one_process_per_gpu_path_runs_with_two_ranks_on_one_device():
    line = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                  "bench.py", "--gpus", "2", "--dry-multi", "--shard-crc", "--num-envs", str(N), "--steps", str(STEPS), "--warmup", str(WARM),
                  "--no-cpu-baseline", "--steady-warmup", "0"])
    assert line["n_gpus"] == 2 and line["steps"] == STEPS and line["warmup"] == WARM and line["scaling"] == "weak"
    assert line["config"]["sharding"] == "env_offset shards x2, no collective" and line["config"]["num_envs_per_gpu"] == N and "dry_multi" in line["config"]
    assert line["value"] > 0 and abs(line["value"] - 2 * N * STEPS / (line["ms_per_step"] * 1e-3 * STEPS)) / line["value"] < 1e-3  # whole-job rate over the MAX time
    # rank r steps envs [r N, (r + 1) N) of the logical vector with the actions RandomState(r) draws
    acts = np.concatenate([np.random.RandomState(r).randint(0, 15, size=(WARM + STEPS, N), dtype=np.int32) for r in range(2)], axis=1)
    assert line["shard_crc"] == _single_handle_crcs(acts, [slice(0, N), slice(N, 2 * N)])


@pytest.mark.gpu
def test_one_handle_over_devices_path_runs_on_fake_devices():
    line = _line([sys.executable, "bench.py", "--devices-in-process", "2", "--shard-crc", "--num-envs", str(N), "--steps", str(STEPS), "--warmup", str(WARM),
                  "--no-cpu-baseline", "--steady-warmup", "0"], env={"PROCGEN_AMD_FAKE_DEVICES": "1"})
    assert line["n_gpus"] == 2 and line["config"]["num_envs_per_gpu"] == N
    assert line["config"]["sharding"] == "one handle, num_devices=2 contiguous index ranges, no collective"
    acts = np.random.RandomState(0).randint(0, 15, size=(WARM + STEPS, 2 * N), dtype=np.int32)
    assert line["shard_crc"] == _single_handle_crcs(acts, [slice(0, N), slice(N, 2 * N)])


@pytest.mark.gpu
def test_one_process_per_gpu_path_with_eight_ranks():
    """The driver's N = 8 launch (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8), every rank on device 0: eight
    env_offset shards of one logical vector, barrier and MAX reduction over eight ranks, one JSON line from rank 0.  Shard r's
    observation CRC is that of envs [r n, (r + 1) n) of one single-device handle stepped with the same actions."""
    n, ranks = 512, 8
    line = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                  "bench.py", "--gpus", str(ranks), "--dry-multi", "--shard-crc", "--num-envs", str(n), "--steps", str(STEPS), "--warmup", str(WARM),
                  "--no-cpu-baseline", "--steady-warmup", "0"])
    assert line["n_gpus"] == ranks and line["config"]["sharding"] == f"env_offset shards x{ranks}, no collective" and line["config"]["num_envs_per_gpu"] == n
    assert abs(line["value"] - ranks * n * STEPS / (line["ms_per_step"] * 1e-3 * STEPS)) / line["value"] < 1e-3
    acts = np.concatenate([np.random.RandomState(r).randint(0, 15, size=(WARM + STEPS, n), dtype=np.int32) for r in range(ranks)], axis=1)
    assert line["shard_crc"] == _single_handle_crcs(acts, [slice(r * n, (r + 1) * n) for r in range(ranks)])
