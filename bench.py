#!/usr/bin/env python
"""
bench.py -- env steps/sec of the MI355X stepper on BASELINE.json's metric.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one libenv_act + libenv_observe over every env of the workload (BASELINE.json configs[1]:
coinrun, num_envs=65536 per GPU, uniform random actions from RandomState(0), rand_seed=23, default options).
Observations stay resident in HBM (host_observations=0; rew/first/info are landed on the host every step, the
4 B/env action upload is inside the timed region).  With N GPUs every rank steps its own 65536 envs of one
logical vector of N*65536 (env_offset = rank*65536, no collective on the data path) -> weak scaling.

Extra objects on the JSON line:
  roofline     : algorithmic bytes (12306 B per env-step) / mean device time of one step's kernels
                 (HIP events on the library's stream, procgen_amd_time_steps) against the 8 TB/s HBM peak.
  cpu_baseline : the compiled reference (oracle/_ref; best of a sweep over its worker pool sizes and over independent
                 processes) or the plain-C oracle port (1 core) timed on a bounded sample of the same workload on this
                 box's host cores (rank 0, N=1 only).
  host_landed  : the PCIe-inclusive rate of the same loop through the unmodified libenv ABI (observations copied into
                 the caller's host array every step) -- reported beside `value`, never as it.
  value        : measured in the STEADY STATE since round 5: before the W warm-up and K timed steps, --steady-warmup steps (default 1500:
                 past coinrun's 1000-step timeout) are run untimed, so that episodes are desynchronised and entity tables / reset rate are
                 at their long-run level -- the regime a trainer sees.  `steady_state` holds the resets per env-step and the envs per LDS
                 arena tier there; `cold_start` the first steps after the synchronized reset (what rounds 1-4 quoted; --steady-warmup 0
                 makes `value` that again).
  roofline     : kernel_ms_per_step comes from HIP events around every libenv_act of the timed loop itself (the same K steps as
                 ms_per_step), dominant_kernel from events around each render launch on its own stream.

  --dry-multi    runs the N > 1 launch path without N GPUs: every rank uses device 0, gloo carries the barrier and the MAX
                 reduction (tests/test_multi_gpu_paths.py).  No scaling number is meant by it.
  --shard-crc    adds "shard_crc": per rank the CRC32 of its envs' last observations (host copy after the timed region).
"""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch initialises HIP: see INTEGRATION.md section 5 (joint handles: one stream per game)
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

ENVS_PER_GPU = 65536
ALGO_BYTES_PER_ENV_STEP = 12288 + 4 + 1 + 9 + 4  # obs + reward + first + info + action
HBM_PEAK_GBS = 8000.0


def _ref_rate(game, n, num_threads, seconds):
    """env steps/sec of the compiled reference on n envs (this process)."""
    import ref_env

    env = ref_env.make_ref_env(n, game, rand_seed=23, num_threads=num_threads)
    rng = np.random.RandomState(0)
    env.observe()
    for _ in range(3):
        env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
        env.observe()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
        env.observe()
        steps += 1
    dt = time.perf_counter() - t0
    env.close()
    return n * steps / dt, steps


def cpu_baseline(game="coinrun", budget_s=24.0):
    """Reported baseline, not the optimisation target: the best of a small sweep over how the reference can use this box's
    host cores -- its own worker pool (num_threads in {4, 16, 64}) on one 1024-env vector, and P independent processes
    (P in {64, 128, cores}) each stepping its own vector inline (num_threads = 0), which avoids the pool's mutex round trip per env."""
    import ref_env

    if not ref_env.available():
        import oracle_env

        n = 256
        env = oracle_env.OracleEnv(n, game, rand_seed=23)
        rng = np.random.RandomState(0)
        env.observe()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < min(budget_s, 15.0):
            env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
            env.observe()
            steps += 1
        dt = time.perf_counter() - t0
        env.close()
        return {"value": round(n * steps / dt, 1), "unit": "env steps/sec", "cores": 1, "kind": "port",
                "sample": f"{game} num_envs={n}, {steps} steps, random actions, plain-C oracle port, single thread"}
    cores = os.cpu_count() or 1
    slot = budget_s / 8.0
    tried = {}
    for nt in sorted({4, 16, 64}):
        if nt <= cores:
            tried[f"pool num_threads={nt}"] = (_ref_rate(game, 1024, nt, slot)[0], nt)
    # P independent processes x inline stepping, P over the box's core count (every core busy at P = cores)
    import subprocess

    code = (f"import sys; sys.path.insert(0, {os.path.join(REPO, 'oracle')!r}); sys.path.insert(0, {REPO!r}); import bench; "
            f"print(bench._ref_rate({game!r}, 64, 0, {slot})[0])")
    for procs in sorted({min(cores, 64), min(cores, 128), cores}):
        ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
        rates = []
        for p_ in ps:
            out, _ = p_.communicate()
            try:
                rates.append(float(out.strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
        if rates:
            tried[f"{len(rates)} processes x num_threads=0 (64 envs each)"] = (sum(rates), len(rates))
    best = max(tried, key=lambda k: tried[k][0])
    return {"value": round(tried[best][0], 1), "unit": "env steps/sec", "cores": tried[best][1], "kind": "reference",
            "sample": f"{game}, random actions, compiled reference C++/Qt (oracle/_ref), best of a sweep with ~{slot:.0f} s per point on a {cores}-core host: {best}",
            "sweep": {k: round(v[0], 1) for k, v in tried.items()}}


def measure_traffic(game, n, pre_rollout):
    """HBM bytes per step of THIS build on THIS box, measured now: two short `rocprofv3 --pmc` passes of this same script (FETCH_SIZE, then
    WRITE_SIZE: the TCC takes one of them per pass, MI355X_MICROARCH.md), outside the timed region, each over the pre-rollout plus a few
    steps.  Per step: every kernel's sum divided by the steps the run drew, counted by what the frame kernels wrote (WRITE_SIZE is exact for
    the observation store: 12288 B per env-frame, the only store of those kernels).  Returns (upper, raw, detail) in bytes -- raw = FETCH +
    WRITE as reported; upper = raw + FETCH again, since gfx950's FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads (same
    guide) -- or (None, None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, None, "unavailable: no rocprofv3 on this box"
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable, os.path.join(REPO, "bench.py"), "--steps", "8", "--warmup", "2",
                   "--game", game, "--num-envs", str(n), "--steady-warmup", str(pre_rollout), "--no-cpu-baseline", "--no-host-landed", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, None, f"unavailable: rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            c = sqlite3.connect(dbs[0])
            T = {row[0].rsplit("_0000", 1)[0]: row[0] for row in c.execute("select name from sqlite_master where type='table'")}
            q = (f"select s.kernel_name, sum(e.value) from {T['rocpd_pmc_event']} e join {T['rocpd_info_pmc']} p on e.pmc_id=p.id "
                 f"join {T['rocpd_kernel_dispatch']} d on e.event_id=d.event_id join {T['rocpd_info_kernel_symbol']} s on d.kernel_id=s.id "
                 f"where p.name='{counter}' group by s.kernel_name")
            sums[counter] = {k: float(v) for k, v in c.execute(q)}
            c.close()
        except Exception as ex:  # noqa: BLE001 -- a profiler hiccup must not cost the bench line
            return None, None, f"unavailable: {type(ex).__name__}: {str(ex)[:200]}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    frame_kernels = [k for k in sums["WRITE_SIZE"] if any(t in k for t in ("6renderI", "6rasterI", "11render_listI", "render<", "raster<", "render_list<"))]
    steps = sum(sums["WRITE_SIZE"][k] for k in frame_kernels) * 1024.0 / (n * 12288.0)
    if steps < 1:
        return None, None, "unavailable: no frame kernel in the counter pass"
    fetch = sum(sums["FETCH_SIZE"].values()) * 1024.0 / steps
    write = sum(sums["WRITE_SIZE"].values()) * 1024.0 / steps
    short = lambda k: k.split("(")[0].replace("pgamd::", "").replace("void ", "")[:60]
    per_kernel = {short(k): {"fetch_MB": round(sums["FETCH_SIZE"].get(k, 0.0) * 1024.0 / steps / 1e6, 1), "write_MB": round(sums["WRITE_SIZE"].get(k, 0.0) * 1024.0 / steps / 1e6, 1)}
                  for k in sorted(set(sums["FETCH_SIZE"]) | set(sums["WRITE_SIZE"]), key=lambda k: -(sums["FETCH_SIZE"].get(k, 0.0) + sums["WRITE_SIZE"].get(k, 0.0)))[:8]}
    return fetch * 2 + write, fetch + write, {"steps_in_pass": round(steps, 1), "fetch_MB_per_step": round(fetch / 1e6, 1), "write_MB_per_step": round(write / 1e6, 1), "per_kernel": per_kernel}


KERNEL_POLICY = {"bigfish": "BigFish", "bossfight": "BossFight", "caveflyer": "CaveFlyerT<1600, 2>", "chaser": "Chaser", "climber": "Climber", "coinrun": "CoinRun",
                 "dodgeball": "Dodgeball", "fruitbot": "FruitBot", "heist": "Heist", "jumper": "Jumper", "leaper": "Leaper", "maze": "Maze", "miner": "Miner",
                 "ninja": "Ninja", "plunder": "Plunder", "starpilot": "StarPilot"}  # the policy class a game's kernels are instantiated with (procgen_amd/csrc/game_*.h)
ALL_GAMES = ["bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot", "heist", "jumper", "leaper", "maze", "miner",
             "ninja", "plunder", "starpilot"]  # reference procgen/env.py ENV_NAMES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num-envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (default = BASELINE configs[1])")
    ap.add_argument("--game", default="coinrun")
    ap.add_argument("--host-landed", action="store_true", help="also land observations on the host (PCIe-inclusive rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-landed", action="store_true", help="skip the host_landed leg (profiling runs: its handle's launches would mix into the per-kernel averages)")
    ap.add_argument("--devices-in-process", type=int, default=1,
                    help="single-process mode: ONE libenv handle of devices x num-envs envs sharded over that many GPUs of this process (num_devices option; no torch.distributed)")
    ap.add_argument("--steady-warmup", type=int, default=1500, help="untimed pre-rollout before the W warm-up and K timed steps (0: measure the cold start, as rounds 1-4 did)")
    ap.add_argument("--dry-multi", action="store_true", help="N > 1 ranks on ONE GPU (device 0 for every rank, gloo): exercises the launch path only")
    ap.add_argument("--shard-crc", action="store_true", help="report the CRC32 of every rank's last observations")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (they re-run this script; also what those passes themselves run with)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    device = 0 if args.dry_multi else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_multi:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from procgen_amd import ProcgenGym3Env

    n = args.num_envs
    if args.game == "all16":  # BASELINE configs[4]'s shape: env n plays names[n % 16] (reference src/vecgame.cpp:295-310)
        args.game = ",".join(ALL_GAMES)
    joint = "," in args.game
    D = args.devices_in_process
    if D > 1:
        assert world == 1, "--devices-in-process is the single-process mode"
        n = n * D  # weak scaling: the per-GPU share stays num-envs
        env = ProcgenGym3Env(n, args.game, rand_seed=23, extra_options={"num_devices": D, "host_observations": bool(args.host_landed)})
    else:
        env = ProcgenGym3Env(n, args.game, rand_seed=23, extra_options={
            "device_id": device, "env_offset": rank * n, "host_observations": bool(args.host_landed)})
    single = not joint and D == 1  # single-part handle: the event-timing and tier-count hooks exist
    if single:
        env._lib.procgen_amd_kernel_timing.restype = C.c_double
        env._lib.procgen_amd_kernel_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        env._lib.procgen_amd_tier_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int)]

    display_list = False  # the game's frames are drawn by prep -> raster kernels: the dominant kernel is raster<Game>
    if single and hasattr(env._lib, "procgen_amd_display_list_frames"):
        env._lib.procgen_amd_display_list_frames.restype = C.c_int
        display_list = bool(env._lib.procgen_amd_display_list_frames(env._handle, (C.c_int * 2)()))

    def render_kernel_window(acts, steps=20):
        """the dominant kernel's own duration: events around each of its launches on the stream it is launched on, over a short window of
        its own right behind the timed loop (four more event records per step are kept out of the timed steps)"""
        if not single:
            return None
        env._lib.procgen_amd_kernel_timing(env._handle, 2, None, None)
        for t in range(min(steps, len(acts))):
            env.act(acts[t])
            env.observe()
        rinfo = (C.c_double * 2)()
        env._lib.procgen_amd_kernel_timing(env._handle, 0, None, rinfo)
        return [rinfo[0], rinfo[1], min(steps, len(acts))]

    def timed_loop(acts, warm, steps):
        """`warm` untimed steps, then exactly `steps` timed ones between two barriers; the device time of a step's kernels (HIP events
        around every libenv_act's launch sequence, procgen_amd_kernel_timing) comes from the SAME steps.  Returns (wall seconds -- MAX over
        ranks --, device ms per step, None, episode resets seen)."""
        for t in range(warm):
            env.act(acts[t])
            env.observe()
        barrier()
        if single:
            env._lib.procgen_amd_kernel_timing(env._handle, 1, None, None)
        resets = 0
        t0 = time.perf_counter()
        for t in range(warm, warm + steps):
            env.act(acts[t])
            _, _, first = env.observe()
            resets += int(np.count_nonzero(first))
        barrier()
        dt = time.perf_counter() - t0
        if single:
            nsteps = C.c_int(0)
            kms = env._lib.procgen_amd_kernel_timing(env._handle, 0, C.byref(nsteps), None)
            assert nsteps.value == steps, (nsteps.value, steps)
            render = None
        else:  # a joint / sharded handle overlaps its parts' kernels: the wall time of the step stands in
            kms, render = dt / steps * 1e3, None
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.dry_multi else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, kms, render, resets

    rng = np.random.RandomState(rank)
    env.observe()
    # (1) cold start, a side object: the first steps after the synchronized reset of every env (no resets yet, almost every env in the
    #     smallest arena tier) -- what rounds 1-4 reported as `value`
    cold = None
    pre = args.steady_warmup
    if pre > 0:
        cs = min(args.steps, 100)
        cacts = rng.randint(0, 15, size=(args.warmup + cs, n), dtype=np.int32)
        cdt, ckms, _, _ = timed_loop(cacts, args.warmup, cs)
        cold = {"value": round(n * world * cs / cdt, 1), "unit": "env steps/sec", "ms_per_step": round(cdt / cs * 1e3, 4), "steps": cs, "warmup": args.warmup,
                "kernel_ms_per_step": round(ckms, 4), "roofline_frac": round(ALGO_BYTES_PER_ENV_STEP * n / (ckms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "the first steps after the synchronized reset of all envs: no resets, entity tables small (rounds 1-4 quoted this as `value`)"}
        # (2) the rollout a trainer is in: past coinrun's 1000-step timeout every episode has ended at least once, the envs are
        #     desynchronised, trail-heavy envs sit in the larger arena tiers and the reset rate is at its long-run level
        srng = np.random.RandomState(1000 + rank)
        done = args.warmup + cs
        for t in range(max(pre - done, 0)):
            env.act(srng.randint(0, 15, size=(n,), dtype=np.int32))
        env.observe()
    # (3) the measurement: W warm-up steps, then exactly K timed steps
    acts = rng.randint(0, 15, size=(args.warmup + args.steps, n), dtype=np.int32)
    tiers = None
    if single:
        tiers = (C.c_int * 3)()
        env._lib.procgen_amd_tier_counts(env._handle, tiers)
    dt, kernel_ms, _, resets = timed_loop(acts, args.warmup, args.steps)
    shard_crc = None
    if args.shard_crc:  # the CRC32 of this rank's last observations (its shard of the logical vector), gathered on rank 0
        import zlib

        from procgen_amd import torch_view

        mine = [zlib.crc32(v.ob.cpu().numpy().tobytes()) for v in torch_view.device_views(env)]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            shard_crc = [c for g in gathered for c in g]
        else:
            shard_crc = mine

    render_info = render_kernel_window(acts)  # (behind the CRCs of the timed loop's last observations)
    # the same loop with the observations landed in the caller's (pinned) host array through the unmodified libenv ABI:
    # the PCIe-inclusive rate (never `value`), on a bounded number of steps
    host_landed = None
    env.close()
    if not joint and not args.host_landed and not args.no_host_landed and world == 1 and D == 1:
        # a handle made the way an unmodified gym3 caller makes it (host_observations is the default): large handles then step in four
        # launch chunks and land each chunk's slice while the next ones still draw (libenv_hip.cpp)
        henv = ProcgenGym3Env(n, args.game, rand_seed=23, extra_options={"device_id": device, "env_offset": rank * n})
        hl_steps = min(args.steps, 40)
        henv.observe()
        for t in range(3):
            henv.act(acts[t])
            henv.observe()
        t1 = time.perf_counter()
        for t in range(hl_steps):
            henv.act(acts[t])
            henv.observe()
        hl_dt = time.perf_counter() - t1
        henv.close()
        host_landed = {"value": round(n * hl_steps / hl_dt, 1), "unit": "env steps/sec", "ms_per_step": round(hl_dt / hl_steps * 1e3, 4), "steps": hl_steps,
                       "note": "a second handle made with host observations, as gym3 makes it: frames copied D2H into the caller's registered host array every step (libenv ABI unmodified), first steps after its reset; PCIe Gen5 x16 bounds this at ~5.1 M steps/s per GPU"}

    if rank == 0:
        # HBM bytes per launch (= one step), measured by this run: two rocprofv3 --pmc passes of this same script on this box, outside the
        # timed region (measure_traffic); null with the reason when the profiler is not there or a pass fails
        traffic = traffic_raw = traffic_detail = None
        traffic_source = "not measured (--no-traffic, a joint / sharded handle or N > 1)"
        if not args.no_traffic and world == 1 and D == 1 and not joint:
            traffic, traffic_raw, traffic_detail = measure_traffic(args.game, n, args.steady_warmup)
            if traffic is None:
                traffic_source, traffic_detail = traffic_detail, None
            else:
                traffic, traffic_raw = round(traffic), round(traffic_raw)
                traffic_source = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this script (pre-rollout + 10 steps each), all kernels, per step; "
                                  f"upper bound = WRITE + 2 x FETCH (gfx950 FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md); raw {traffic_raw}")
        total_steps = n * world * args.steps
        value = total_steps / dt
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": f"env steps/sec (whole node), {args.game} num_envs={n} random actions",
            "value": round(value, 1), "unit": "env steps/sec", "n_gpus": world * D, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+i32 game state, u8 pixels", "data": "synthetic (uniform random actions, procedurally generated levels)",
            "config": {"workload": f"{args.game} num_envs={n} per GPU x {world} GPU(s), random-action rollout, distribution_mode=hard, "
                                   f"observations {'landed on host (PCIe inclusive)' if args.host_landed else 'resident in HBM'}",
                       "num_envs_per_gpu": n // D, "sharding": (f"one handle, num_devices={D} contiguous index ranges, no collective" if D > 1 else f"env_offset shards x{world}, no collective")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source, "traffic_detail": traffic_detail,
                         "launch": "one step = exactly what libenv_act enqueues (step grids + list kernels, render kernels) over all envs of this GPU; HIP events on the library's stream around every libenv_act of the timed loop itself",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "kernel_ms_per_step": round(kernel_ms, 4), "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP},
        }
        if render_info is not None and render_info[1] > 0:
            gname = KERNEL_POLICY.get(args.game, args.game)
            line["roofline"]["dominant_kernel"] = {
                "name": f"pgamd::raster<{gname}>" if display_list else f"pgamd::render<{gname}, false>", "avg_us": round(render_info[0] * 1e3, 1), "launches_per_step": round(render_info[1], 2),
                "source": f"HIP events around each launch of the kernel on the stream it is launched on, {render_info[2]} steps right behind the timed ones (procgen_amd_kernel_timing); its launches overlap the step kernels of the other chunk",
                "achieved_GBs_observation_write": round(12288 * n / max(render_info[1], 1e-9) / (render_info[0] * 1e-3) / 1e9, 1)}
        if pre > 0:
            line["config"]["pre_rollout_steps"] = max(pre, args.warmup + min(args.steps, 100))
            line["config"]["regime"] = "steady state: timed after the pre-rollout (episodes desynchronised, reset rate and arena tiers at their long-run level)"
            line["steady_state"] = {"resets_per_env_step": round(resets / (n * args.steps), 6),
                                    "arena_tier_envs": {"tier0": tiers[0], "tier1": tiers[1], "tier2": tiers[2]} if tiers is not None else None,
                                    "note": "`value` IS the steady-state figure since round 5 (the verdict of round 4 used it); `cold_start` keeps the earlier rounds' regime"}
        if cold is not None:
            line["cold_start"] = cold
        if joint:
            line["roofline"]["launch"] = "one step = the step + render kernels of all games of the joint handle (wall time of the step, kernels of different games overlap)"
        if host_landed is not None:
            line["host_landed"] = host_landed
        if shard_crc is not None:
            line["shard_crc"] = shard_crc
        if args.dry_multi:
            line["config"]["dry_multi"] = "every rank on device 0, gloo: launch-path check, not a scaling measurement"
        if world == 1 and D == 1 and not args.no_cpu_baseline and not joint:
            line["cpu_baseline"] = cpu_baseline(args.game)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
