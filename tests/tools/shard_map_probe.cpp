// TEST HARNESS: exposes procgen_amd/csrc/shard_map.h (host-only) to tests/test_shard_map.py through ctypes.
#include "shard_map.h"
using pgamd::ShardMap;
extern "C" {
int sm_valid(int n, int g, int k) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.valid(); }
int sm_env_of(int n, int g, int k, int part, int i) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.env_of(part, i); }
int sm_part_of(int n, int g, int k, int env) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.part_of(env); }
int sm_index_in_part(int n, int g, int k, int env) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.index_in_part(env); }
int sm_device_of_part(int n, int g, int k, int part) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.device_of_part(part); }
int sm_game_of_part(int n, int g, int k, int part) { ShardMap m; m.num_envs = n; m.num_devices = g; m.num_games = k; return m.game_of_part(part); }
}
