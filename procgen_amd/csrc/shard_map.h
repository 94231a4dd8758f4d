// shard_map.h -- which part of a libenv handle owns which env.
//
// A handle of N envs is cut into G device shards of contiguous global indices [g * N / G, (g + 1) * N / G) (reference
// analogue: the worker pool inside one VecGame, src/vecgame.cpp:196-205,378-435).  With a comma
// separated env_name of K games env n plays names[n % K] (src/vecgame.cpp:295-310), so every shard is a multiple of K envs
// and the game of an env does not depend on the sharding.  Part (g, k) = game k on device g holds the envs
// base_g + k + K * i, i = 0 .. N / (G * K) - 1, in that order.  Host-only, no HIP: unit-tested on the CPU.
#pragma once

namespace pgamd {

struct ShardMap {
    int num_envs = 0, num_devices = 1, num_games = 1;
    bool valid() const { return num_envs > 0 && num_devices > 0 && num_games > 0 && num_envs % (num_devices * num_games) == 0; }
    int parts() const { return num_devices * num_games; }
    int envs_per_device() const { return num_envs / num_devices; }
    int envs_per_part() const { return num_envs / (num_devices * num_games); }
    int part_index(int device, int game) const { return device * num_games + game; }
    int device_of_part(int part) const { return part / num_games; }
    int game_of_part(int part) const { return part % num_games; }
    // global index of env i of a part, and the inverse
    int env_of(int part, int i) const { return device_of_part(part) * envs_per_device() + game_of_part(part) + num_games * i; }
    int part_of(int env) const { return part_index(env / envs_per_device(), (env % envs_per_device()) % num_games); }
    int index_in_part(int env) const { return (env % envs_per_device()) / num_games; }
    // global index of a part's env 0 (the seed stream position, "env_offset"), its stride is num_games
    int first_env(int part) const { return env_of(part, 0); }
};

}  // namespace pgamd
