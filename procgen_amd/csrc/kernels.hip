// kernels.hip -- dispatch on game id over the per-game kernel objects (kernels_game.hip, one translation unit per game).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include "pg_bgpaint.h"
#include "pg_env.h"
#include "pg_math.h"
#include "shard_map.h"
#include "wave.h"

namespace pgamd {

#define PG_GAME_NAMES(X) X(CoinRun) X(BigFish) X(Maze) X(Climber) X(Miner) X(StarPilot) X(FruitBot) X(Leaper) X(Plunder) X(Heist) X(Ninja) X(Dodgeball) X(BossFight) X(Chaser) X(CaveFlyer) X(Jumper) X(CaveFlyerMemory)
#define PG_X(Game) const GameEntry *game_entry_##Game();
PG_GAME_NAMES(PG_X)
#undef PG_X
static const GameEntry *find(int game_id) {
    static const GameEntry *const entries[] = {
#define PG_X(Game) game_entry_##Game(),
        PG_GAME_NAMES(PG_X)
#undef PG_X
    };
    for (const GameEntry *e : entries)
        if (e->game_id == game_id) return e;
    return nullptr;
}

hipError_t launch_step(int game_id, const DevCtx &d, int mode, const LaunchStreams &ls) {
    const GameEntry *e = find(game_id);
    return e ? e->launch(d, mode, ls) : hipErrorInvalidValue;
}
hipError_t launch_render_one(int game_id, const DevCtx &d, int env, hipStream_t stream, int count) {
    const GameEntry *e = find(game_id);
    return e ? e->render_one(d, env, count, stream) : hipErrorInvalidValue;
}
hipError_t launch_render_human(int game_id, const DevCtx &d, int env_base, int count, hipStream_t stream) {
    const GameEntry *e = find(game_id);
    return e ? e->render_human(d, env_base, count, stream) : hipErrorInvalidValue;
}
int first_chunk_envs(int num_envs, int first_pct) {
    if (first_pct <= 0 || first_pct >= 100 || num_envs < 4096) return 0;
    const int first = (int)((long long)num_envs * first_pct / 100) / TILE_ENVS * TILE_ENVS;
    return (first > 0 && first < num_envs) ? first : 0;
}
int chunk_envs_for(int num_envs, int chunks) {
    const int nchunk = (chunks > 1 && num_envs >= 4096) ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1;
    const int per = ((num_envs + nchunk - 1) / nchunk + TILE_ENVS - 1) / TILE_ENVS * TILE_ENVS;
    return per > 0 ? per : TILE_ENVS;
}
bool game_supported(int game_id) { return find(game_id) != nullptr; }
int game_tier_for(int game_id, int slots_needed) {
    const GameEntry *e = find(game_id);
    if (!e) return 0;
    return slots_needed <= e->cap_t0 ? 0 : (slots_needed <= e->cap_t1 ? 1 : 2);
}
void game_limits(int game_id, int *ent_cap_hbm, int *grid_bytes) {
    const GameEntry *e = find(game_id);
    *ent_cap_hbm = e ? e->cap_t2 : 0;
    *grid_bytes = e ? e->grid_bytes : 0;
}
void game_init_state(int game_id, int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng) {
    const GameEntry *e = find(game_id);
    if (e) e->init_state(num_envs, rand_seed, env_offset, env_stride, hdr, rng);
}

int game_host_tables(int game_id, const GameOptions &opt, uint32_t *out, int max_words) {
    const GameEntry *e = find(game_id);
    return e ? e->host_tables(opt, out, max_words) : 0;
}

bool game_split_reset(int game_id) {
    const GameEntry *e = find(game_id);
    return e ? e->split_reset : false;
}
hipError_t launch_render_slow(int game_id, const DevCtx &d, int env_base, int count, int chunk, hipStream_t stream) {
    const GameEntry *e = find(game_id);
    return e ? e->render_slow(d, env_base, count, chunk, stream) : hipErrorInvalidValue;
}
int game_frame_rec_words(int game_id) {
    const GameEntry *e = find(game_id);
    return e ? e->frame_rec_words : 0;
}
bool (*game_use_block_asset(int game_id))(int) {
    const GameEntry *e = find(game_id);
    return e ? e->use_block_asset : nullptr;
}

// ---- use_generated_assets: the backgrounds of the episodes that began this step (pg_bgpaint.h), one wave per env of the chunk;
// an env without a request leaves at once
__global__ __launch_bounds__(64) void paint_backgrounds(DevCtx d, int env_base) {
    __shared__ BgPaintLds lds;
    const int env = env_base + (int)blockIdx.x;
    const int skip = d.bg_req[2 * env + 1];
    if (skip < 0) return;
    int err = 0;
    paint_background(d.gen_bg + (size_t)env * GEN_BG_WORDS, d.bg_req[2 * env], skip, &lds, &err);
    if (threadIdx.x == 0) {
        d.bg_req[2 * env + 1] = -1;
        if (err) {
            pg_report_error(d, env, pg_error_word(PGE_ASSERT, __LINE__), ERR_KIND_BGPAINT, 0, 0);
        }
    }
}
hipError_t launch_paint_backgrounds(const DevCtx &d, int env_base, int count, hipStream_t stream) {
    if (!d.gen_bg || count <= 0) return hipSuccess;
    hipLaunchKernelGGL(paint_backgrounds, dim3(count), dim3(64), 0, stream, d, env_base);
    return hipGetLastError();
}

// ---- launch order of the render kernel by background image (PROCGEN_AMD_RENDER_ORDER; libenv_hip.cpp VecGame::rebuild_render_order) ----
// A counting sort of one launch chunk's envs [base, base + count) by EnvHdr::background_index, on the device: histogram, exclusive scan,
// scatter.  Sorted position p goes to launch slot render_order_slot(p, count) (shard_map.h): workgroup j of a launch runs on XCD j mod 8,
// each XCD has its own L2, so XCD x draws the x-th eighth of the sorted sequence -- a few images, one after the other -- instead of all 62
// at once.  The order within an image is whatever the atomics give: envs are independent, any permutation draws the same frames.
constexpr int RO_BINS = MAX_BACKGROUNDS;
__global__ __launch_bounds__(256) void render_order_hist(const EnvHdr *hdr, int base, int count, int *hist) {
    __shared__ int h[RO_BINS];
    if (threadIdx.x < RO_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < count) {
        int b = hdr[base + i].background_index;
        b = b < 0 ? 0 : (b >= RO_BINS ? RO_BINS - 1 : b);
        atomicAdd(&h[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < RO_BINS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void render_order_scan(int *hist) {  // counts -> first sorted position of each image (128 values: one thread)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < RO_BINS; b++) {
            const int c = hist[b];
            hist[b] = acc;
            acc += c;
        }
    }
}
__global__ __launch_bounds__(256) void render_order_scatter(const EnvHdr *hdr, int base, int count, int *cursor, int *order) {
    __shared__ int cnt[RO_BINS], start[RO_BINS];
    if (threadIdx.x < RO_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    int b = 0, local = 0;
    if (i < count) {
        b = hdr[base + i].background_index;
        b = b < 0 ? 0 : (b >= RO_BINS ? RO_BINS - 1 : b);
        local = atomicAdd(&cnt[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < RO_BINS && cnt[threadIdx.x]) start[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]);  // one range per (block, image)
    __syncthreads();
    if (i < count) {
        const int p = start[b] + local;
        order[base + render_order_slot(p, count)] = base + i;  // (shard_map.h: a bijection for any count)
    }
}
hipError_t launch_render_order(const DevCtx &d, int base, int count, int *scratch, int *order, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(scratch, 0, RO_BINS * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int blocks = (count + 255) / 256;
    hipLaunchKernelGGL(render_order_hist, dim3(blocks), dim3(256), 0, stream, d.hdr, base, count, scratch);
    hipLaunchKernelGGL(render_order_scan, dim3(1), dim3(64), 0, stream, scratch);
    hipLaunchKernelGGL(render_order_scatter, dim3(blocks), dim3(256), 0, stream, d.hdr, base, count, scratch, order);
    return hipGetLastError();
}

// ---- device math self-tests (procgen_amd_selftest_*, include/procgen_amd.h): the exact device functions the game
// policies call, run over caller-chosen inputs so that a test can sweep a whole input domain against the host libm ----
__global__ void selftest_bigfish_radius_kernel(const float *r01, float *out, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // game_bigfish.h game_step (reference src/games/bigfish.cpp:84): FISH_MAX_R = 2, FISH_MIN_R = .25
    if (i < n) out[i] = (float)((double)(2.0f - .25f) * pg_pow((double)r01[i], 1.4) + (double).25f);
}
__global__ void selftest_sincos_kernel(uint32_t first_bits, int n, double *out_sin, double *out_cos) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) {
        const double x = (double)__builtin_bit_cast(float, first_bits + (uint32_t)i);  // the games pass float angles
        out_sin[i] = pg_sin_d(x);
        out_cos[i] = pg_cos_d(x);
    }
}
__global__ void selftest_sincos_scaled_kernel(const uint32_t *bits, int n, double scale, float *out_sin, float *out_cos) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) {
        const double x = (double)__builtin_bit_cast(float, bits[i]);
        out_sin[i] = (float)(pg_sin_d(x) * scale);  // the shape of every call site: float(trig(double(float angle)) * double(float speed))
        out_cos[i] = (float)(pg_cos_d(x) * scale);
    }
}
hipError_t selftest_sincos_scaled(const uint32_t *d_bits, int n, double scale, float *d_sin, float *d_cos) {
    hipLaunchKernelGGL(selftest_sincos_scaled_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_bits, n, scale, d_sin, d_cos);
    return hipGetLastError();
}
hipError_t selftest_bigfish_radius(const float *d_in, float *d_out, int n) {
    hipLaunchKernelGGL(selftest_bigfish_radius_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n);
    return hipGetLastError();
}
hipError_t selftest_sincos(uint32_t first_bits, int n, double *d_sin, double *d_cos) {
    hipLaunchKernelGGL(selftest_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, first_bits, n, d_sin, d_cos);
    return hipGetLastError();
}

}  // namespace pgamd
