/*
 * procgen_amd.h -- extension entry points of the HIP libenv.so, next to the gym3 libenv ABI (libenv.h).
 *
 * The reference sanctions extra exported functions on the libenv handle: gym3's CEnv takes `c_func_defs`
 * and calls them through `call_c_func` (reference procgen/env.py:128-136,145 -- that is how get_state /
 * set_state are reached).  These hooks use the same mechanism; none of the seven libenv calls changes.
 *
 * Extension OPTIONS accepted by libenv_make (all optional; the reference library would reject them as
 * "unused options", reference src/vecoptions.cpp:34-38, so only pass them to this library):
 *   "device_id"          int32  HIP device ordinal (default: $LOCAL_RANK mod device count, else 0)
 *   "env_offset"         int32  global index of env 0 of this handle; env n then reproduces env
 *                               (env_offset + n) of a single big handle (seed derivation of reference
 *                               src/vecgame.cpp:301-314 is per global index) -- used to shard one
 *                               logical vector of envs over several GPUs / processes
 *   "host_observations"  uint8  1 (default): libenv_observe lands `ob` in the caller's host buffers, as the
 *                               ABI requires; 0: observations stay in HBM (read them through
 *                               procgen_amd_device_buffers / procgen_amd_part_buffers), rew/first/info are
 *                               still landed on the host
 *   "num_devices"        int32  one handle sharded over that many GPUs of this node (0 = all visible ones;
 *                               default 1, or $PROCGEN_AMD_NUM_DEVICES): device g owns the contiguous global
 *                               index range [g * N / G, (g + 1) * N / G); with a comma separated env_name of
 *                               K games each range is a multiple of K and env n plays names[n % K] whatever
 *                               the sharding (reference src/vecgame.cpp:295-310).  No collective: every
 *                               device lands its slice of the caller's buffers.
 *
 * Reference options this library implements with kernels of their own (nothing to pass beyond what procgen/env.py passes):
 *   "render_human" (render_mode="rgb_array"): a fourth info tensor "rgb" uint8 [512][512][3], the antialiased frame of
 *   reference src/vecgame.cpp:270-282,363-376, drawn on the device behind every step and landed in the caller's info
 *   buffers; "use_generated_assets": AssetGen sprites / backgrounds (reference src/assetgen.cpp).
 */
#ifndef PROCGEN_AMD_H
#define PROCGEN_AMD_H

#include "libenv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device-side (HBM) views of the boundary buffers.  Valid until libenv_close; contents are valid after
 * libenv_observe returns (the producing stream has been joined) or after synchronizing `stream`. */
struct procgen_amd_buffers {
    int device_id;
    int num_envs;
    void *stream;                 /* hipStream_t the step kernels run on */
    uint8_t *ob;                  /* [num_envs][64][64][3] RGB888 */
    float *rew;                   /* [num_envs] */
    uint8_t *first;               /* [num_envs] */
    int32_t *prev_level_seed;     /* [num_envs] */
    uint8_t *prev_level_complete; /* [num_envs] */
    int32_t *level_seed;          /* [num_envs] */
    int32_t *action;              /* [num_envs] actions of the step in flight (written by libenv_act) */
};

/* fills *out; returns 0.  Single-game, single-device handles only (one part); see procgen_amd_part_buffers. */
LIBENV_API int procgen_amd_device_buffers(libenv_env *handle, struct procgen_amd_buffers *out);

/* A handle is G device shards x K games = G * K parts (one per game per device; a single-game single-device handle
 * is one part).  Part p's device arrays hold its envs densely: its env i is global env  first_env + i * env_stride
 * (env_stride = K).  This is the reader of the observations of joint / multi-device handles with
 * host_observations = 0 (BASELINE configs[3] / [4] in device-resident mode). */
struct procgen_amd_part {
    struct procgen_amd_buffers buffers; /* device_id, num_envs (of the part), stream, ob, rew, ... as above */
    int first_env;                      /* global index of the part's env 0 */
    int env_stride;                     /* global indices between two consecutive envs of the part */
    char game[LIBENV_MAX_NAME_LEN];     /* the game the part's envs play */
};
/* returns the number of parts; fills out[0 .. min(max_parts, parts)) when out is not NULL */
LIBENV_API int procgen_amd_part_buffers(libenv_env *handle, struct procgen_amd_part *out, int max_parts);
/* switch the D2H landing of observations on/off after construction (same meaning as the option).  The handle keeps the launch shape it
 * was made with: a handle of >= 32 768 envs made WITH host observations steps in four launch chunks (each chunk's frames landed while the
 * next are drawn) and stays at four after the landing is switched off -- ~15 % slower kernels than the two chunks a handle made without
 * host observations uses; make the handle with the mode it will mostly run in. */
LIBENV_API void procgen_amd_set_host_observations(libenv_env *handle, int enable);
/* Runs `steps` steps back to back entirely on the device (actions: [steps][num_envs] int32 on the HOST, or
 * NULL to repeat the last actions) and returns the mean device time of one step's kernels in milliseconds,
 * measured with HIP events on the library's own stream.  Used by bench.py for the roofline figure. */
LIBENV_API double procgen_amd_time_steps(libenv_env *handle, int steps, const int32_t *actions_or_null);

/* Device time of the steps of the CALLER's own libenv_act / libenv_observe loop.  enable = 1: from now on every libenv_act brackets its
 * kernels with two HIP events on the library's stream (counters reset); enable = 2: also every launch of the render kernel -- the dominant
 * kernel -- with two events on the stream it is launched on (four more event records per step: a short window of its own, not the timed one);
 * enable = 0: stop.  Every call returns the mean device milliseconds per step of the steps timed since the last enabling call and their number
 * in *steps_out (may be NULL); render_out (may be NULL): [0] the mean duration in milliseconds of one render launch, [1] its launches per
 * step.  bench.py times its measured loop this way, so that the device time and the wall time of a step come from the same steps.
 * Single-part handles. */
LIBENV_API double procgen_amd_kernel_timing(libenv_env *handle, int enable, int *steps_out, double *render_out);

/* How many envs of the coming step each LDS arena tier of the step kernel owns (single-part handles): out[0..2] = tier 0, 1, 2.
 * Tiers 1 / 2 are the envs whose entity table may outgrow the smaller arena; the split drifts with the
 * horizon of a rollout (trails, spawned objects), which is why bench.py's steady_state object reports it.  Returns num_envs. */
LIBENV_API int procgen_amd_tier_counts(libenv_env *handle, int *out);

/* The render kernel's launch order (every few steps a counting sort on the device re-maps the render workgroups to envs
 * by background image, per launch chunk).  out[slot] = the env workgroup `slot` draws; returns the entries written (num_envs), 0 when the
 * handle launches in env order or max_envs is too small.  chunk_out (may be NULL): [0] = envs of the first launch chunk, [1] = launch chunks.
 * Single-part handles.  selftest_render_order_slots (host only, no device): out[p] = the launch slot of sorted position p within a chunk of
 * `count` envs -- a permutation of [0, count) for every count. */
LIBENV_API int procgen_amd_render_order(libenv_env *handle, int *out, int max_envs, int *chunk_out);
LIBENV_API void procgen_amd_selftest_render_order_slots(int count, int *out);

/* set_state for envs [first, first + count) in one call: state j is the bytes data[offsets[j], offsets[j + 1]) -- the streams get_state /
 * procgen_amd_get_states produce.  Same result as `count` set_state calls (reference procgen/env.py:148-153, src/vecgame.cpp:447-456: the
 * reference loops per env); consecutive envs are uploaded block-wise and redrawn by one render launch. */
LIBENV_API void procgen_amd_set_states(libenv_env *handle, int first, int count, const char *data, const long long *offsets);

/* Display-list games (the frame is drawn by prep -> raster kernels, and by the full renderer for the frames the short
 * path cannot draw).  out[0] = envs whose current frame came from its record, out[1] = envs whose frame the full renderer drew.  Returns 1,
 * or 0 for a handle that renders with one kernel (out untouched).  Single-part handles. */
LIBENV_API int procgen_amd_display_list_frames(libenv_env *handle, int *out);

/* Device math self-tests (no handle, current HIP device; host pointers in and out): the exact device functions the game
 * kernels call, over caller-chosen inputs, so that a test can sweep a whole input domain against the host libm.
 *   bigfish_radius: out[i] = the fish radius bigfish computes from the rand01() draw r01[i] (pow, reference
 *                   src/games/bigfish.cpp:84)
 *   sincos:         sin / cos (pg_math.h) of the float with bit pattern first_bits + i, as doubles */
LIBENV_API void procgen_amd_selftest_bigfish_radius(const float *r01, float *out, int n);
LIBENV_API void procgen_amd_selftest_sincos(uint32_t first_bits, int n, double *out_sin, double *out_cos);
/*   sincos_scaled:  float(sin(x) * scale), float(cos(x) * scale) for the floats x with the given bit patterns -- the shape of
 *                   every call site that feeds game state (bullet / thrust velocities) */
LIBENV_API void procgen_amd_selftest_sincos_scaled(const uint32_t *bits, int n, double scale, float *out_sin, float *out_cos);

/* get_state of the envs [first, first + count) in one call: the states are packed back to back into data[0, capacity), state k at
 * data[offsets[k], offsets[k + 1]) (offsets has count + 1 entries).  Returns the number of states that fit (>= 1); call again from
 * first + that for the rest.  Same bytes as get_state; what env.get_state() of the Python mirror uses (the reference's loop,
 * procgen/env.py:138-146, crosses cffi once per env with a 1 MiB buffer each). */
LIBENV_API int procgen_amd_get_states(libenv_env *handle, int first, int count, char *data, long long capacity, long long *offsets);

/* reference src/vecgame.cpp:437-457 (declared to cffi by reference procgen/env.py:132-135) */
LIBENV_API int get_state(libenv_env *handle, int env_idx, char *data, int length);
LIBENV_API void set_state(libenv_env *handle, int env_idx, char *data, int length);

#ifdef __cplusplus
}
#endif
#endif
