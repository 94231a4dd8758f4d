"""
CPU: the env -> (device shard, game) map of a multi-device / joint-game libenv handle (procgen_amd/csrc/shard_map.h).
Contiguous index ranges per device, each a multiple of the number of joint games, so that env n plays names[n % K]
whatever the sharding (reference src/vecgame.cpp:295-310).
"""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "procgen_amd", "csrc")


@pytest.fixture(scope="module")
def sm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sm") / "libshardmap.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, os.path.join(HERE, "tools", "shard_map_probe.cpp"), "-o", out])
    return C.CDLL(out)


@pytest.mark.parametrize("n,g,k", [(65536, 1, 1), (262144, 8, 1), (131072, 8, 16), (16384, 1, 16), (96, 2, 3), (64, 4, 1)])
def test_parts_partition_the_envs(sm, n, g, k):
    assert sm.sm_valid(n, g, k)
    per_part = n // (g * k)
    seen = set()
    for part in range(g * k):
        dev, game = sm.sm_device_of_part(n, g, k, part), sm.sm_game_of_part(n, g, k, part)
        prev = -1
        for i in (range(per_part) if per_part <= 64 else list(range(8)) + list(range(per_part - 8, per_part))):
            e = sm.sm_env_of(n, g, k, part, i)
            assert dev * (n // g) <= e < (dev + 1) * (n // g), "a device owns one contiguous index range"
            assert e % k == game, "env n plays names[n % K] whatever the sharding"
            assert sm.sm_part_of(n, g, k, e) == part and sm.sm_index_in_part(n, g, k, e) == i
            assert e > prev
            prev = e
            seen.add(e)
    if per_part <= 64:
        assert seen == set(range(n))


def test_shards_must_be_whole_multiples_of_the_joint_games(sm):
    assert not sm.sm_valid(100, 8, 1) and not sm.sm_valid(48, 2, 16) and sm.sm_valid(64, 2, 16)
