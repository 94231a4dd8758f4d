// kernels.hip -- gfx950 kernels of the vectorized stepper.  One workgroup = one 64-lane wavefront = one env.
//
//   step_small<Game> : grid = num_envs; LDS arena sized for Game::ENT_CAP_SMALL entities; skips envs routed to
//                      the large kernel.
//   step_big<Game>   : fixed grid that walks the list of envs whose entity table may exceed the small arena
//                      (LDS arena for Game::ENT_CAP_BIG entities).
//   render<Game>     : grid = num_envs, 256 threads: four band-waves per env rasterize 16 rows each into a 4 KB
//                      LDS band and store the RGB888 observation (pg_render.h).
// The step kernels run Env<Game,CAP>::run (pg_env.h): HBM -> LDS staging, Game::step / reset + level generation,
// state write-back.
#include <hip/hip_runtime.h>

#include "games.h"
#include "pg_render.h"
#include "kernels.h"

namespace pgamd {

template <class Game>
__global__ __launch_bounds__(64) void step_small(DevCtx d, int mode) {
    __shared__ Lds<Game, Game::ENT_CAP_SMALL> lds;
    const int env = (int)blockIdx.x;
    if (mode != 0 && d.hdr[env].big) return;
    Env<Game, Game::ENT_CAP_SMALL> e(d, env, &lds);
    e.run(mode);
}

template <class Game>
__global__ __launch_bounds__(64) void step_big(DevCtx d, int mode) {
    __shared__ Lds<Game, Game::ENT_CAP_BIG> lds;
    const int count = *d.big_count;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        const int env = d.big_list[k];
        if (!d.hdr[env].big) continue;  // set_state replaced this env with a small one after the list was built
        Env<Game, Game::ENT_CAP_BIG> e(d, env, &lds);
        e.run(mode);
        __syncthreads();
    }
}

template <class Game>
__global__ __launch_bounds__(256) void render(DevCtx d, int env_base) {
    __shared__ uint32_t fb[NUM_BANDS][BAND_ROWS * RES_W];
    const int band = (int)(threadIdx.x >> 6);
    Renderer<Game> r(d, env_base + (int)blockIdx.x, band, fb[band]);
    r.render_band();
}

// The two step kernels touch disjoint envs, so the (few, slow, low-occupancy) large-arena envs run on a side
// stream concurrently with the small-arena grid; the render kernel joins both.
template <class Game>
static hipError_t launch_game(const DevCtx &d, int mode, const LaunchStreams &ls) {
    if (mode != 0) {
        hipError_t e = hipEventRecord(ls.fork, ls.main);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(ls.side, ls.fork, 0);
        if (e != hipSuccess) return e;
        int big_grid = d.num_envs < 4096 ? d.num_envs : 4096;
        hipLaunchKernelGGL(step_big<Game>, dim3(big_grid), dim3(64), 0, ls.side, d, mode);
        e = hipEventRecord(ls.join, ls.side);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(step_small<Game>, dim3(d.num_envs), dim3(64), 0, ls.main, d, mode);
    if (mode != 0) {
        hipError_t e = hipStreamWaitEvent(ls.main, ls.join, 0);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(render<Game>, dim3(d.num_envs), dim3(256), 0, ls.main, d, 0);
    return hipGetLastError();
}

hipError_t launch_step(int game_id, const DevCtx &d, int mode, const LaunchStreams &ls) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return launch_game<Game>(d, mode, ls);
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return hipErrorInvalidValue;
    }
}

// re-renders one env (after set_state)
hipError_t launch_render_one(int game_id, const DevCtx &d, int env, hipStream_t stream) {
    switch (game_id) {
#define PG_X(Game)                                                                       \
    case Game::GAME_ID:                                                                  \
        hipLaunchKernelGGL(render<Game>, dim3(1), dim3(256), 0, stream, d, env);         \
        return hipGetLastError();
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return hipErrorInvalidValue;
    }
}

bool game_supported(int game_id) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return true;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return false;
    }
}

int game_small_cap(int game_id) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return Game::ENT_CAP_SMALL;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return 0;
    }
}

void game_limits(int game_id, int *ent_cap_hbm, int *grid_bytes) {
    *ent_cap_hbm = 0;
    *grid_bytes = 0;
    switch (game_id) {
#define PG_X(Game)                                                                        \
    case Game::GAME_ID:                                                                   \
        *ent_cap_hbm = Game::ENT_CAP_BIG;                                                 \
        *grid_bytes = (int)((Game::MAX_CELLS * sizeof(Game::cell_t) + 15) & ~(size_t)15); \
        break;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: break;
    }
}

void game_init_state(int game_id, int num_envs, int rand_seed, int env_offset, EnvHdr *hdr, uint32_t *rng) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: init_env_state<Game>(num_envs, rand_seed, env_offset, hdr, rng); break;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: break;
    }
}

}  // namespace pgamd
