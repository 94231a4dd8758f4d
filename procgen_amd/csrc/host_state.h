// host_state.h -- host-side construction of the initial per-env state (the work of the VecGame constructor,
// reference src/vecgame.cpp:284-331): env n's level_seed_rand_gen is seeded with the n-th raw draw of
// RandGen(rand_seed), independent of num_envs -- which is what lets a shard of envs [offset, offset+n) on one
// GPU reproduce exactly the envs of a larger single-process run.
#pragma once
#include <cstdint>
#include <vector>

#include "pg_defs.h"

namespace pgamd {

struct HostMT {  // std::mt19937 (libstdc++ bits/random.tcc): seed, twist, temper
    uint32_t mt[MT_N];
    int idx;
    void seed(int s) {
        mt[0] = (uint32_t)s;
        for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = MT_N;
    }
    uint32_t next() {
        if (idx >= MT_N) {
            for (int k = 0; k < MT_N; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % MT_N] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t z = mt[idx++];
        z ^= (z >> 11);
        z ^= (z << 7) & 0x9d2c5680u;
        z ^= (z << 15) & 0xefc60000u;
        z ^= (z >> 18);
        return z;
    }
};

// Fills hdr[n] and rng[n][MT_SLOTS][MT_STRIDE] for the envs env_offset + n * env_stride (n < num_envs) of the global
// index space (stride > 1: the envs of one game of a joint handle, reference src/vecgame.cpp:309-310).
template <class Game>
inline void init_env_state(int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng) {
    HostMT seedgen;
    seedgen.seed(rand_seed);
    for (int k = 0; k < env_offset; k++) seedgen.next();
    EnvHdr proto;
    Game::construct(proto);
    for (int n = 0; n < num_envs; n++) {
        hdr[n] = proto;
        uint32_t *st = rng + (size_t)n * MT_SLOTS * MT_STRIDE;
        for (int k = 0; k < MT_SLOTS * MT_STRIDE; k++) st[k] = 0;
        HostMT lvl;
        lvl.seed((int)seedgen.next());  // games[n]->level_seed_rand_gen.seed(game_level_seed_gen.randint()), vecgame.cpp:314
        for (int k = 1; k < env_stride; k++) seedgen.next();  // the draws of the other games' envs in between
        for (int k = 0; k < MT_N; k++) st[MT_STRIDE + k] = lvl.mt[k];
    }
}

// level_seed_low/high: reference src/vecgame.cpp:284-293
inline void level_seed_range(int num_levels, int start_level, int *low, int *high) {
    *low = 0;
    *high = 0;
    if (num_levels == 0) {
        *low = 0;
        *high = INT32_MAX;
    } else if (num_levels > 0) {
        *low = start_level;
        *high = start_level + num_levels;
    }
}

}  // namespace pgamd
