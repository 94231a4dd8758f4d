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

// Launch slot of sorted position p among the `count` envs of one render launch chunk (kernels.hip render_order_scatter): workgroup j of
// a launch runs on XCD j mod 8 and each XCD has its own L2, so XCD x takes the x-th eighth of the sorted sequence.  With q = count / 8 and
// r = count % 8 the XCDs x < r take q + 1 positions and the others q: a bijection of [0, count) for ANY count (round 5's form,
// (p % q) * 8 + p / q, was one only for multiples of 8: up to 7 envs of a chunk kept stale frames, advisor finding).  Plain integer code,
// compiled for the host and the device alike.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int render_order_slot(int p, int count) {
    const int q = count >> 3, r = count & 7;
    const int head = r * (q + 1);  // positions of the XCDs that take q + 1
    if (p < head) return (p % (q + 1)) * 8 + p / (q + 1);
    return q > 0 ? ((p - head) % q) * 8 + r + (p - head) / q : p;
}

}  // namespace pgamd
