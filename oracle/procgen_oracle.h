/*
 * procgen_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's hot path
 * (libenv_act -> Game::step -> BasicAbstractGame physics -> reset/level-gen -> 64x64 render).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (procgen_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref/libenv.so, built from the
 * unmodified sources) through the committed fixtures in tests/golden/ -- the reference's own tests
 * hold no value-level golden vectors (reference procgen/env_test.py, state_test.py are
 * self-consistency tests only).
 */
#ifndef PROCGEN_ORACLE_H
#define PROCGEN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PgoVec PgoVec;

/* Options mirror the libenv option names (reference src/vecgame.cpp:183-190, src/game.cpp:42-75). */
typedef struct {
    int rand_seed;
    int num_levels;
    int start_level;
    int distribution_mode; /* 0 easy, 1 hard, 2 extreme, 10 memory */
    int center_agent;
    int use_backgrounds;
    int use_monochrome_assets;
    int restrict_themes;
    int use_generated_assets; /* must be 0 */
    int paint_vel_info;
    int use_sequential_levels;
    int debug_mode;
} PgoOptions;

/* Image registry shared by all envs (stands in for images_load(), reference src/resources.cpp:30).
 * The harness asks which files a game needs and hands over decoded pixels (0xAARRGGBB; sprites
 * premultiplied ARGB32, backgrounds RGB32). */
int pgo_game_id(const char *name);
int pgo_num_images(int game_id);
const char *pgo_image_name(int game_id, int idx);
int pgo_image_is_background(int game_id, int idx);
void pgo_set_image(int game_id, int idx, int w, int h, const uint32_t *px); /* copies */

PgoVec *pgo_make(int game_id, int num_envs, const PgoOptions *opt);
/* oracle env k = env (env_offset + k * env_stride) of the logical vector (a shard, or the envs of one game of a joint handle) */
PgoVec *pgo_make_strided(int game_id, int num_envs, const PgoOptions *opt, int env_offset, int env_stride);
void pgo_free(PgoVec *v);

/* Initial reset + first observation of every env (reference src/vecgame.cpp:346-357). */
void pgo_init(PgoVec *v);
/* One Game::step per env (reference src/game.cpp:120-155). actions[num_envs]. */
void pgo_step(PgoVec *v, const int32_t *actions);
/* Outputs of the last init/step. rgb is [num_envs][64][64][3]. */
void pgo_observe(PgoVec *v, uint8_t *rgb, float *rew, uint8_t *first, int32_t *prev_level_seed,
                 uint8_t *prev_level_complete, int32_t *level_seed);

/* Debug/state view used by the tests: entity count and a flat dump of one env
 * (fields in the order of reference src/entity.cpp:90-137, 31 words each). */
int pgo_num_entities(PgoVec *v, int env);
void pgo_dump_entities(PgoVec *v, int env, int32_t *out_words);
void pgo_dump_grid(PgoVec *v, int env, int32_t *out, int *w, int *h);
/* Miscellaneous scalars: out[0]=cur_time out[1]=rand_gen draws since seed (mod 2^31) ... */
void pgo_dump_scalars(PgoVec *v, int env, int32_t *out16);
/* test hook: the Qt 5.9 ellipse restatement (midpoint or path route) on a cleared 64 x 64 canvas; out[64*64]: 0 untouched,
 * 1 brush, 2 pen */
void pgo_test_draw_ellipse(double x, double y, double w, double h, int pen, int brush, uint8_t *out);
/* test hooks: AssetGen (use_generated_assets): the 64 x 64 sprite of an object type of a game (ARGB words), and a 500 x 500
 * background painted from a generator seeded with `seed` */
void pgo_test_generated_asset(int game_id, int type, uint32_t *out4096);
void pgo_test_generated_background(int seed, uint32_t *out250000);
void pgo_dump_background(PgoVec *v, int env, uint32_t *out250000); /* use_generated_assets: env's current background canvas */

#ifdef __cplusplus
}
#endif
#endif
