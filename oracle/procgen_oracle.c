/*
 * procgen_oracle.c -- TEST INFRASTRUCTURE ONLY (see procgen_oracle.h).
 *
 * Plain-C restatement of the reference hot path, one function per reference function, each citing
 * the reference file:line it follows ("BAG" = reference procgen/src/basic-abstract-game.cpp).
 * Floating point follows the C++ promotion rules of the reference expressions exactly (double
 * literals promote, results narrow on assignment); build with -ffp-contract=off (no FMA), matching
 * the reference's PyPI-wheel flags (-march=ivybridge, reference procgen/CMakeLists.txt:28-31).
 *
 * Games restated so far: coinrun, bigfish, maze (with MazeGen::generate_maze / place_objects), climber, miner,
 * starpilot, fruitbot, leaper, plunder, heist (with MazeGen::generate_maze_with_doors), ninja, dodgeball, bossfight, chaser (with MazeGen::generate_maze_no_dead_ends), caveflyer (with RoomGenerator), jumper (compass drawn with Qt's midpoint ellipse and cosmetic line).
 */
#include "procgen_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* constants: reference src/object-ids.h, src/game.h:23-26, BAG:6-20                             */
#define RES_W 64
#define RES_H 64
#define INVALID_OBJ (-1)
#define PLAYER 0
#define SPACE 100
#define WALL_OBJ 51
#define EXPLOSION 54
#define EXPLOSION5 58
#define TRAIL 59
#define USE_ASSET_THRESHOLD 100
#define MAX_ASSETS 100
#define MAX_IMAGE_THEMES 10

static const float PI_F = 3.14159265358979323846264338327950288f; /* src/cpp-utils.h:12 */
#define MAXVTHETA (15 * PI_F / 180) /* BAG:6 */
#define MIXRATEROT 0.5f             /* BAG:7 */
static const float POS_EPS = -0.001f;   /* BAG:10 */
static const float RENDER_EPS = 0.02f;  /* BAG:14 */

enum { GAME_BIGFISH = 0, GAME_BOSSFIGHT = 1, GAME_CAVEFLYER = 2, GAME_CHASER = 3, GAME_CLIMBER = 4, GAME_COINRUN = 5, GAME_DODGEBALL = 6, GAME_FRUITBOT = 7, GAME_HEIST = 8, GAME_JUMPER = 9, GAME_LEAPER = 10, GAME_MAZE = 11, GAME_MINER = 12, GAME_NINJA = 13, GAME_PLUNDER = 14, GAME_STARPILOT = 15 };

/* jumper.cpp:11-27 */
#define JP_GOAL 1
#define JP_SPIKE 2
#define JP_CAVEWALL 6
#define JP_CAVEWALL_TOP 7
#define JP_PLAYER_JUMP 9
#define JP_PLAYER_LEFT1 10
#define JP_PLAYER_LEFT2 11
#define JP_PLAYER_RIGHT1 12
#define JP_PLAYER_RIGHT2 13
#define JP_MAZE_SCALE 3
#define JP_JUMP_COOLDOWN 3

/* caveflyer.cpp:9-21 */
#define CF_GOAL 1
#define CF_OBSTACLE 2
#define CF_TARGET 3
#define CF_PLAYER_BULLET 4
#define CF_ENEMY 5
#define CF_CAVEWALL 8
#define CF_EXHAUST 9
#define CF_MARKER 1003

/* chaser.cpp:10-23 */
#define CH_LARGE_ORB 2
#define CH_ENEMY_WEAK 3
#define CH_ENEMY_EGG 4
#define CH_MAZE_WALL 5
#define CH_ENEMY 6
#define CH_ENEMY2 7
#define CH_ENEMY3 8
#define CH_MARKER 1001
#define CH_ORB 1002
#define CH_ORB_REWARD 0.04f
#define CH_ORB_DIM 0.3f
#define INVALID_IDX (-2)

/* bossfight.cpp:8-31 */
#define BF2_PLAYER_BULLET 1
#define BF2_BOSS 2
#define BF2_SHIELDS 3
#define BF2_ENEMY_BULLET 4
#define BF2_LASER_TRAIL 5
#define BF2_REFLECTED_BULLET 6
#define BF2_BARRIER 7
#define BF2_BOSS_R 3.0f
#define BF2_BOTTOM_MARGIN 6
#define BF2_BOSS_VEL_TIMEOUT 20
#define BF2_BOSS_DAMAGED_TIMEOUT 40

/* dodgeball.cpp:8-25 */
#define DB_LAVA_WALL 1
#define DB_PLAYER_BALL 3
#define DB_ENEMY 4
#define DB_DOOR 5
#define DB_ENEMY_BALL 6
#define DB_DOOR_OPEN 7
#define DB_DUST_CLOUD 8
#define DB_OOB_WALL 10
#define DB_ENEMY_VEL 0.05f
#define DB_BALL_V_ROT (PI_F * 0.23f)

/* ninja.cpp:9-21 */
#define NJ_GOAL 1
#define NJ_BOMB 6
#define NJ_THROWING_STAR 7
#define NJ_PLAYER_JUMP 9
#define NJ_PLAYER_RIGHT1 12
#define NJ_PLAYER_RIGHT2 13
#define NJ_FIRE 14
#define NJ_WALL_MID 20

/* heist.cpp:10-15, object-ids.h */
#define HS_LOCKED_DOOR 1
#define HS_KEY 2
#define HS_EXIT 9
#define HS_KEY_ON_RING 11
#define EXIT_OBJ 52
#define AGENT_OBJ 53
#define DOOR_OBJ 200
#define KEY_OBJ 300

/* plunder.cpp:8-15 */
#define PL_PLAYER_BULLET 1
#define PL_TARGET_LEGEND 2
#define PL_TARGET_BACKGROUND 3
#define PL_PANEL 6
#define PL_SHIP 7

/* leaper.cpp:6-21 */
#define LP_LOG 1
#define LP_ROAD 2
#define LP_WATER 3
#define LP_CAR 4
#define LP_FINISH_LINE 5
#define LP_MONSTER_RADIUS 0.25f
#define LP_LOG_RADIUS 0.45f
#define LP_NSTEP 5
static const float LP_MAX_SPEED = (float)(2 / (LP_NSTEP - 1.0));
#define LP_VEL_DECAY (LP_MAX_SPEED / LP_NSTEP)

/* fruitbot.cpp:8-24 */
#define FB_BARRIER 1
#define FB_OUT_OF_BOUNDS_WALL 2
#define FB_PLAYER_BULLET 3
#define FB_BAD_OBJ 4
#define FB_GOOD_OBJ 7
#define FB_LOCKED_DOOR 10
#define FB_LOCK 11
#define FB_PRESENT 12
#define FB_KEY_DURATION 8
#define FB_DOOR_ASPECT_RATIO 3.25f

/* starpilot.cpp:6-27 */
#define SP_V_SCALE (2.0f / 5.0f)
#define SP_BULLET_PLAYER 1
#define SP_BULLET2 2
#define SP_BULLET3 3
#define SP_FLYER 4
#define SP_METEOR 5
#define SP_CLOUD 6
#define SP_TURRET 7
#define SP_FAST_FLYER 8
#define SP_FINISH_LINE 9
#define SP_SHOOTER_WIN_TIME 500
#define SP_NUM_BASIC_OBJECTS 9
#define SP_NUM_SHIP_THEMES 7
#define SP_MAX_SPAWNERS 256

/* miner ids: reference src/games/miner.cpp:11-19 */
#define MN_BOULDER 1
#define MN_DIAMOND 2
#define MN_MOVING_BOULDER 3
#define MN_MOVING_DIAMOND 4
#define MN_ENEMY 5
#define MN_EXIT 6
#define MN_DIRT 9
#define MN_OOB_WALL 10

/* climber ids: reference src/games/climber.cpp:12-28 */
#define CL_COIN 1
#define CL_ENEMY 5
#define CL_ENEMY1 6
#define CL_ENEMY2 7
#define CL_PLAYER_JUMP 9
#define CL_PLAYER_RIGHT1 12
#define CL_PLAYER_RIGHT2 13
#define CL_WALL_MID 15
#define CL_WALL_TOP 16
#define CL_ENEMY_BARRIER 19
#define CL_PATROL_RANGE 4.0f

/* maze ids: reference src/games/maze.cpp:8 */
#define MZ_GOAL 2
#define MAZE_OFFSET 1 /* src/mazegen.h:14 */

/* bigfish ids: reference src/games/bigfish.cpp:8-18 */
#define BF_FISH 2
#define BF_FISH_MIN_R .25f
#define BF_FISH_MAX_R 2.0f
#define BF_FISH_QUOTA 30

/* coinrun ids: reference src/games/coinrun.cpp:11-31 */
#define CR_GOAL 1
#define CR_SAW 2
#define CR_SAW2 3
#define CR_ENEMY 5
#define CR_ENEMY1 6
#define CR_ENEMY2 7
#define CR_PLAYER_JUMP 9
#define CR_PLAYER_RIGHT1 12
#define CR_PLAYER_RIGHT2 13
#define CR_WALL_MID 15
#define CR_WALL_TOP 16
#define CR_LAVA_MID 17
#define CR_LAVA_TOP 18
#define CR_ENEMY_BARRIER 19
#define CR_CRATE 20

static void fatal(const char *msg) {
    fprintf(stderr, "procgen_oracle: %s\n", msg);
    exit(1);
}

/* ------------------------------------------------------------------------------------------- */
/* RandGen over std::mt19937: reference src/randgen.cpp:6-31,90-98; libstdc++ bits/random.tcc   */
typedef struct {
    uint32_t mt[624];
    int idx;
    int seeded;
    int64_t draws;
} Rng;

static void rng_seed(Rng *r, int seed) { /* randgen.cpp:95-98; mersenne_twister_engine::seed */
    r->mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
    r->seeded = 1;
    r->draws = 0;
}

static uint32_t rng_u32(Rng *r) {
    if (!r->seeded) fatal("RandGen used before seed (randgen.cpp:7)");
    if (r->idx >= 624) {
        uint32_t *mt = r->mt;
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        r->idx = 0;
    }
    uint32_t z = r->mt[r->idx++];
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    r->draws++;
    return z;
}

static int rng_randint(Rng *r, int low, int high) { /* randgen.cpp:6-11 */
    uint32_t x = rng_u32(r);
    uint32_t range = (uint32_t)(high - low);
    return (int)((uint32_t)low + (x % range));
}
static int rng_randn(Rng *r, int high) { /* randgen.cpp:13-17 */
    uint32_t x = rng_u32(r);
    return (int)(x % (uint32_t)high);
}
static float rng_rand01(Rng *r) { /* randgen.cpp:19-23 */
    uint32_t x = rng_u32(r);
    return (float)((double)x / 4294967296.0);
}
static int rng_randint_raw(Rng *r) { return (int)rng_u32(r); } /* randgen.cpp:90-93 */

/* ------------------------------------------------------------------------------------------- */
/* Entity: reference src/entity.h:7-48, src/entity.cpp:8-82                                      */
typedef struct {
    float x, y, vx, vy, rx, ry;
    int type, image_type, image_theme, render_z;
    int will_erase, collides_with_entities;
    float collision_margin, rotation, vrot;
    int is_reflected, fire_time, spawn_time, life_time, expire_time, use_abs_coords;
    float friction;
    int smart_step, avoids_collisions, auto_erase;
    float alpha, health, theta, grow_rate, alpha_decay, climber_spawn_x;
} Ent;

static void ent_init(Ent *e, float x, float y, float vx, float vy, float rx, float ry, int type) { /* entity.cpp:11-51 */
    memset(e, 0, sizeof(*e));
    e->x = x; e->y = y; e->vx = vx; e->vy = vy; e->rx = rx; e->ry = ry;
    e->type = type;
    e->image_type = type;
    e->image_theme = 0;
    e->collision_margin = 0.0f;
    e->rotation = 0.0f;
    e->vrot = 0.0f;
    e->alpha = 1.0f;
    e->grow_rate = 1.0f;
    e->alpha_decay = 1.0f;
    e->fire_time = -1;
    e->spawn_time = -1;
    e->expire_time = -1;
    e->life_time = 0;
    e->health = 1;
    e->theta = -100;
    e->friction = 1;
    e->auto_erase = 1;
    if (type == EXPLOSION) {
        e->grow_rate = 1.4f;
        e->expire_time = 4;
    } else if (type == TRAIL) {
        e->grow_rate = 1.05f;
        e->alpha_decay = 0.8f;
    }
}

static void ent_step(Ent *e) { /* entity.cpp:57-82 */
    if (!e->smart_step) {
        e->x += e->vx;
        e->y += e->vy;
    }
    e->rotation += e->vrot;
    e->vx *= e->friction;
    e->vy *= e->friction;
    e->life_time += 1;
    if (e->expire_time > 0 && e->life_time > e->expire_time) e->will_erase = 1;
    if (e->type == EXPLOSION) {
        if (e->image_type < EXPLOSION5) e->image_type++;
    }
    e->rx *= e->grow_rate;
    e->ry *= e->grow_rate;
    e->alpha = e->alpha_decay * e->alpha;
}

/* ------------------------------------------------------------------------------------------- */
/* images                                                                                         */
typedef struct {
    int w, h;
    uint32_t *px;
    int generic; /* a generated asset: QImage::Format_ARGB32, which Qt draws through its generic span route (see draw_image_generic) */
} Img;

#define MAX_GAME_IMAGES 256
typedef struct {
    int n;
    char names[MAX_GAME_IMAGES][96];
    int is_bg[MAX_GAME_IMAGES];
    Img img[MAX_GAME_IMAGES];
    /* asset table: (type, theme) -> image index, built from asset_for_type */
    int type_num_themes[MAX_ASSETS];
    int type_theme_img[MAX_ASSETS][MAX_IMAGE_THEMES];
    int n_bg;
    int bg_img[128];
    int built;
} GameAssets;

static GameAssets g_assets[16];

static int assets_add(GameAssets *a, const char *name, int is_bg) {
    for (int i = 0; i < a->n; i++)
        if (a->is_bg[i] == is_bg && strcmp(a->names[i], name) == 0) return i;
    if (a->n >= MAX_GAME_IMAGES) fatal("too many images");
    strncpy(a->names[a->n], name, 95);
    a->is_bg[a->n] = is_bg;
    return a->n++;
}

static void assets_type(GameAssets *a, int type, const char *name) {
    int t = a->type_num_themes[type];
    /* names beyond MAX_IMAGE_THEMES count towards asset_num_themes (BAG:114-116) but can never be drawn (fassert BAG:888) */
    if (t < MAX_IMAGE_THEMES) a->type_theme_img[type][t] = assets_add(a, name, 0);
    a->type_num_themes[type] = t + 1;
}

/* reference src/resources.cpp:848-903 + :950-953 (space backgrounds appended) */
static const char *PLATFORM_BGS[] = {
    "platform_backgrounds/alien_bg.png", "platform_backgrounds/another_world_bg.png", "platform_backgrounds/back_cave.png",
    "platform_backgrounds/caverns.png", "platform_backgrounds/cyberpunk_bg.png", "platform_backgrounds/parallax_forest.png",
    "platform_backgrounds/scifi_bg.png", "platform_backgrounds/scifi2_bg.png", "platform_backgrounds/living_tissue_bg.png",
    "platform_backgrounds/airadventurelevel1.png", "platform_backgrounds/airadventurelevel2.png",
    "platform_backgrounds/airadventurelevel3.png", "platform_backgrounds/airadventurelevel4.png",
    "platform_backgrounds/cave_background.png", "platform_backgrounds/blue_desert.png", "platform_backgrounds/blue_grass.png",
    "platform_backgrounds/blue_land.png", "platform_backgrounds/blue_shroom.png", "platform_backgrounds/colored_desert.png",
    "platform_backgrounds/colored_grass.png", "platform_backgrounds/colored_land.png", "platform_backgrounds/colored_shroom.png",
    "platform_backgrounds/landscape1.png", "platform_backgrounds/landscape2.png", "platform_backgrounds/landscape3.png",
    "platform_backgrounds/landscape4.png", "platform_backgrounds/battleback1.png", "platform_backgrounds/battleback2.png",
    "platform_backgrounds/battleback3.png", "platform_backgrounds/battleback4.png", "platform_backgrounds/battleback5.png",
    "platform_backgrounds/battleback6.png", "platform_backgrounds/battleback7.png", "platform_backgrounds/battleback8.png",
    "platform_backgrounds/battleback9.png", "platform_backgrounds/battleback10.png", "platform_backgrounds/sunrise.png",
    "platform_backgrounds_2/beach1.png", "platform_backgrounds_2/beach2.png", "platform_backgrounds_2/beach3.png",
    "platform_backgrounds_2/beach4.png", "platform_backgrounds_2/fantasy1.png", "platform_backgrounds_2/fantasy2.png",
    "platform_backgrounds_2/fantasy3.png", "platform_backgrounds_2/fantasy4.png", "platform_backgrounds_2/candy1.png",
    "platform_backgrounds_2/candy2.png", "platform_backgrounds_2/candy3.png", "platform_backgrounds_2/candy4.png",
    /* space_backgrounds, reference src/resources.cpp:829-845 */
    "space_backgrounds/deep_space_01.png", "space_backgrounds/spacegen_01.png", "space_backgrounds/milky_way_01.png",
    "space_backgrounds/ez_space_lite_01.png", "space_backgrounds/meyespace_v1_01.png", "space_backgrounds/eye_nebula_01.png",
    "space_backgrounds/deep_sky_01.png", "space_backgrounds/space_nebula_01.png", "space_backgrounds/Background-1.png",
    "space_backgrounds/Background-2.png", "space_backgrounds/Background-3.png", "space_backgrounds/Background-4.png",
    "space_backgrounds/parallax-space-backgound.png"};

static void lower_copy(char *dst, const char *src) {
    for (; *src; src++, dst++) *dst = (char)((*src >= 'A' && *src <= 'Z') ? (*src + 32) : *src);
    *dst = 0;
}

static void assets_topdown_backgrounds(GameAssets *a) { /* topdown_backgrounds, reference src/resources.cpp:900-911 */
    static const char *TOPDOWN[] = {"topdown_backgrounds/floortiles.png", "topdown_backgrounds/backgrounddetailed1.png",
                                    "topdown_backgrounds/backgrounddetailed2.png", "topdown_backgrounds/backgrounddetailed3.png",
                                    "topdown_backgrounds/backgrounddetailed4.png", "topdown_backgrounds/backgrounddetailed5.png",
                                    "topdown_backgrounds/backgrounddetailed6.png", "topdown_backgrounds/backgrounddetailed7.png",
                                    "topdown_backgrounds/backgrounddetailed8.png"};
    a->n_bg = 9;
    for (int i = 0; i < 9; i++) a->bg_img[i] = assets_add(a, TOPDOWN[i], 1);
}

static void assets_build(int game_id) {
    GameAssets *a = &g_assets[game_id];
    if (a->built) return;
    a->built = 1;
    char buf[128], lc[32];
    /* BAG:416-430 reserved assets */
    assets_type(a, EXPLOSION, "misc_assets/explosion1.png");
    assets_type(a, EXPLOSION + 1, "misc_assets/explosion2.png");
    assets_type(a, EXPLOSION + 2, "misc_assets/explosion3.png");
    assets_type(a, EXPLOSION + 3, "misc_assets/explosion4.png");
    assets_type(a, EXPLOSION + 4, "misc_assets/explosion5.png");
    assets_type(a, TRAIL, "misc_assets/iconCircle_white.png");
    if (game_id == GAME_COINRUN) { /* coinrun.cpp:33-35,72-121 */
        static const char *ENEMIES[] = {"slimeBlock", "slimePurple", "slimeBlue", "slimeGreen", "mouse", "snail", "ladybug", "wormGreen", "wormPink"};
        static const char *COLORS[] = {"Beige", "Blue", "Green", "Pink", "Yellow"};
        static const char *GROUNDS[] = {"Dirt", "Grass", "Planet", "Sand", "Snow", "Stone"};
        static const int ptypes[4] = {PLAYER, CR_PLAYER_JUMP, CR_PLAYER_RIGHT1, CR_PLAYER_RIGHT2};
        static const char *pnames[4] = {"stand", "jump", "walk1", "walk2"};
        for (int k = 0; k < 4; k++)
            for (int c = 0; c < 5; c++) {
                snprintf(buf, sizeof buf, "kenney/Players/128x256/%s/alien%s_%s.png", COLORS[c], COLORS[c], pnames[k]);
                assets_type(a, ptypes[k], buf);
            }
        for (int e = 0; e < 9; e++) {
            snprintf(buf, sizeof buf, "kenney/Enemies/%s.png", ENEMIES[e]);
            assets_type(a, CR_ENEMY1, buf);
        }
        for (int e = 0; e < 9; e++) {
            snprintf(buf, sizeof buf, "kenney/Enemies/%s_move.png", ENEMIES[e]);
            assets_type(a, CR_ENEMY2, buf);
        }
        assets_type(a, CR_GOAL, "kenney/Items/coinGold.png");
        for (int g = 0; g < 6; g++) {
            lower_copy(lc, GROUNDS[g]);
            snprintf(buf, sizeof buf, "kenney/Ground/%s/%sMid.png", GROUNDS[g], lc);
            assets_type(a, CR_WALL_TOP, buf);
        }
        for (int g = 0; g < 6; g++) {
            lower_copy(lc, GROUNDS[g]);
            snprintf(buf, sizeof buf, "kenney/Ground/%s/%sCenter.png", GROUNDS[g], lc);
            assets_type(a, CR_WALL_MID, buf);
        }
        assets_type(a, CR_LAVA_TOP, "kenney/Tiles/lavaTop_low.png");
        assets_type(a, CR_LAVA_MID, "kenney/Tiles/lava.png");
        assets_type(a, CR_SAW, "kenney/Enemies/sawHalf.png");
        assets_type(a, CR_SAW2, "kenney/Enemies/sawHalf_move.png");
        assets_type(a, CR_CRATE, "kenney/Tiles/boxCrate.png");
        assets_type(a, CR_CRATE, "kenney/Tiles/boxCrate_double.png");
        assets_type(a, CR_CRATE, "kenney/Tiles/boxCrate_single.png");
        assets_type(a, CR_CRATE, "kenney/Tiles/boxCrate_warning.png");
        /* coinrun.cpp:60-62 load_background_images: platform_backgrounds */
        a->n_bg = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0]));
        for (int i = 0; i < a->n_bg; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[i], 1);
    } else if (game_id == GAME_BIGFISH) { /* bigfish.cpp:33-46 */
        assets_type(a, PLAYER, "misc_assets/fishTile_072.png");
        assets_type(a, BF_FISH, "misc_assets/fishTile_074.png");
        assets_type(a, BF_FISH, "misc_assets/fishTile_078.png");
        assets_type(a, BF_FISH, "misc_assets/fishTile_080.png");
        /* water_backgrounds, reference src/resources.cpp:921-932 */
        static const char *WATER[] = {"water_backgrounds/water1.png", "water_backgrounds/water2.png", "water_backgrounds/water3.png",
                                      "water_backgrounds/water4.png", "water_backgrounds/underwater1.png", "water_backgrounds/underwater2.png",
                                      "water_backgrounds/underwater3.png"};
        a->n_bg = 7;
        for (int i = 0; i < 7; i++) a->bg_img[i] = assets_add(a, WATER[i], 1);
    } else if (game_id == GAME_CLIMBER) { /* climber.cpp:42-88 */
        static const char *PCOL[] = {"Blue", "Green", "Grey", "Red"};
        static const int ptypes[4] = {PLAYER, CL_PLAYER_JUMP, CL_PLAYER_RIGHT1, CL_PLAYER_RIGHT2};
        static const char *pnames[4] = {"stand", "walk4", "walk1", "walk2"};
        for (int k = 0; k < 4; k++)
            for (int c = 0; c < 4; c++) {
                snprintf(buf, sizeof buf, "platformer/player%s_%s.png", PCOL[c], pnames[k]);
                assets_type(a, ptypes[k], buf);
            }
        assets_type(a, CL_WALL_TOP, "platformer/tileBlue_05.png");
        assets_type(a, CL_WALL_TOP, "platformer/tileGreen_05.png");
        assets_type(a, CL_WALL_TOP, "platformer/tileYellow_06.png");
        assets_type(a, CL_WALL_TOP, "platformer/tileBrown_06.png");
        assets_type(a, CL_WALL_MID, "platformer/tileBlue_08.png");
        assets_type(a, CL_WALL_MID, "platformer/tileGreen_08.png");
        assets_type(a, CL_WALL_MID, "platformer/tileYellow_09.png");
        assets_type(a, CL_WALL_MID, "platformer/tileBrown_09.png");
        assets_type(a, CL_ENEMY1, "platformer/enemySwimming_1.png");
        assets_type(a, CL_ENEMY2, "platformer/enemySwimming_2.png");
        assets_type(a, CL_COIN, "platformer/yellowCrystal.png");
        a->n_bg = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0]));
        for (int i = 0; i < a->n_bg; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[i], 1);
    } else if (game_id == GAME_MINER) { /* miner.cpp:41-55 */
        assets_type(a, PLAYER, "misc_assets/robot_greenDrive1.png");
        assets_type(a, MN_BOULDER, "misc_assets/elementStone007.png");
        assets_type(a, MN_DIAMOND, "misc_assets/gemBlue.png");
        assets_type(a, MN_EXIT, "misc_assets/window.png");
        assets_type(a, MN_DIRT, "misc_assets/dirt.png");
        assets_type(a, MN_OOB_WALL, "misc_assets/tile_bricksGrey.png");
        a->n_bg = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0]));
        for (int i = 0; i < a->n_bg; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[i], 1);
    } else if (game_id == GAME_JUMPER) { /* jumper.cpp:50-80 */
        assets_type(a, PLAYER, "misc_assets/bunny2_ready.png");
        assets_type(a, JP_SPIKE, "misc_assets/spikeMan_stand.png");
        assets_type(a, JP_GOAL, "misc_assets/carrot.png");
        assets_type(a, JP_PLAYER_JUMP, "misc_assets/bunny2_jump.png");
        assets_type(a, JP_PLAYER_RIGHT1, "misc_assets/bunny2_walk1.png");
        assets_type(a, JP_PLAYER_RIGHT2, "misc_assets/bunny2_walk2.png");
        assets_type(a, JP_PLAYER_LEFT1, "misc_assets/bunny2_walk1.png");
        assets_type(a, JP_PLAYER_LEFT2, "misc_assets/bunny2_walk2.png");
        assets_type(a, JP_CAVEWALL_TOP, "platformer/tileBlue_05.png");
        assets_type(a, JP_CAVEWALL_TOP, "platformer/tileGreen_05.png");
        assets_type(a, JP_CAVEWALL_TOP, "platformer/tileYellow_06.png");
        assets_type(a, JP_CAVEWALL_TOP, "platformer/tileBrown_06.png");
        assets_type(a, JP_CAVEWALL, "platformer/tileBlue_08.png");
        assets_type(a, JP_CAVEWALL, "platformer/tileGreen_08.png");
        assets_type(a, JP_CAVEWALL, "platformer/tileYellow_09.png");
        assets_type(a, JP_CAVEWALL, "platformer/tileBrown_09.png");
        a->n_bg = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0]));
        for (int i = 0; i < a->n_bg; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[i], 1);
    } else if (game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:35-53 */
        assets_type(a, CF_GOAL, "misc_assets/ufoGreen2.png");
        assets_type(a, CF_OBSTACLE, "misc_assets/meteorBrown_big1.png");
        assets_type(a, CF_TARGET, "misc_assets/ufoRed2.png");
        assets_type(a, CF_PLAYER_BULLET, "misc_assets/laserBlue02.png");
        assets_type(a, CF_ENEMY, "misc_assets/enemyShipBlue4.png");
        assets_type(a, PLAYER, "misc_assets/playerShip1_red.png");
        assets_type(a, CF_CAVEWALL, "misc_assets/groundA.png");
        assets_type(a, CF_EXHAUST, "misc_assets/towerDefense_tile295.png");
        int n_platform = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0])) - 13; /* space_backgrounds */
        a->n_bg = 13;
        for (int i = 0; i < 13; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[n_platform + i], 1);
    } else if (game_id == GAME_CHASER) { /* chaser.cpp:51-73 */
        assets_type(a, PLAYER, "misc_assets/enemyFloating_1b.png");
        assets_type(a, CH_ENEMY, "misc_assets/enemyFlying_1.png");
        assets_type(a, CH_ENEMY2, "misc_assets/enemyFlying_2.png");
        assets_type(a, CH_ENEMY3, "misc_assets/enemyFlying_3.png");
        assets_type(a, CH_LARGE_ORB, "misc_assets/yellowCrystal.png");
        assets_type(a, CH_ENEMY_WEAK, "misc_assets/enemyWalking_1b.png");
        assets_type(a, CH_ENEMY_EGG, "misc_assets/enemySpikey_1b.png");
        assets_type(a, CH_MAZE_WALL, "misc_assets/tileStone_slope.png");
        a->n_bg = 1; /* topdown_simple_backgrounds, reference src/resources.cpp:913-918 */
        a->bg_img[0] = assets_add(a, "topdown_backgrounds/floortiles.png", 1);
    } else if (game_id == GAME_BOSSFIGHT) { /* bossfight.cpp:77-107 */
        assets_type(a, PLAYER, "misc_assets/playerShip1_blue.png");
        assets_type(a, PLAYER, "misc_assets/playerShip1_green.png");
        assets_type(a, PLAYER, "misc_assets/playerShip2_orange.png");
        assets_type(a, PLAYER, "misc_assets/playerShip3_red.png");
        assets_type(a, BF2_BOSS, "misc_assets/enemyShipBlack1.png");
        assets_type(a, BF2_BOSS, "misc_assets/enemyShipBlue2.png");
        assets_type(a, BF2_BOSS, "misc_assets/enemyShipGreen3.png");
        assets_type(a, BF2_BOSS, "misc_assets/enemyShipRed4.png");
        for (int t = 0; t < 2; t++) {
            int ty = t == 0 ? BF2_ENEMY_BULLET : BF2_PLAYER_BULLET;
            assets_type(a, ty, "misc_assets/laserGreen14.png");
            assets_type(a, ty, "misc_assets/laserRed11.png");
            assets_type(a, ty, "misc_assets/laserBlue09.png");
        }
        assets_type(a, BF2_SHIELDS, "misc_assets/shield2.png");
        for (int i = 1; i <= 4; i++) {
            snprintf(buf, sizeof buf, "misc_assets/spaceMeteors_00%d.png", i);
            assets_type(a, BF2_BARRIER, buf);
        }
        for (int i = 1; i <= 4; i++) {
            snprintf(buf, sizeof buf, "misc_assets/meteorGrey_big%d.png", i);
            assets_type(a, BF2_BARRIER, buf);
        }
        int n_platform = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0])) - 13; /* space_backgrounds */
        a->n_bg = 13;
        for (int i = 0; i < 13; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[n_platform + i], 1);
    } else if (game_id == GAME_DODGEBALL) { /* dodgeball.cpp:49-88 */
        assets_type(a, PLAYER, "misc_assets/character12.png");
        assets_type(a, DB_PLAYER_BALL, "misc_assets/ball_soccer1.png");
        for (int i = 1; i <= 11; i++) {
            snprintf(buf, sizeof buf, "misc_assets/character%d.png", i);
            assets_type(a, DB_ENEMY, buf);
        }
        assets_type(a, DB_DOOR, "misc_assets/blockRed.png");
        assets_type(a, DB_ENEMY_BALL, "misc_assets/ball_soccer2.png");
        assets_type(a, DB_DOOR_OPEN, "misc_assets/blockGreen.png");
        assets_type(a, DB_LAVA_WALL, "misc_assets/tileStone_slope2.png");
        assets_type(a, DB_OOB_WALL, "misc_assets/tileStone_slope2.png");
        for (int i = 1; i <= 9; i++) {
            snprintf(buf, sizeof buf, "misc_assets/spaceEffect%d.png", i);
            assets_type(a, DB_DUST_CLOUD, buf);
        }
        assets_topdown_backgrounds(a);
    } else if (game_id == GAME_NINJA) { /* ninja.cpp:45-75 */
        assets_type(a, NJ_WALL_MID, "misc_assets/tile_bricksGrey.png");
        assets_type(a, NJ_WALL_MID, "misc_assets/tile_bricksGrown.png");
        assets_type(a, NJ_WALL_MID, "misc_assets/tile_bricksRed.png");
        for (int i = 1; i <= 6; i++) {
            snprintf(buf, sizeof buf, "platformer/shroom%d.png", i);
            assets_type(a, NJ_GOAL, buf);
        }
        assets_type(a, PLAYER, "platformer/zombie_idle.png");
        assets_type(a, NJ_PLAYER_JUMP, "platformer/zombie_jump.png");
        assets_type(a, NJ_PLAYER_RIGHT1, "platformer/zombie_walk1.png");
        assets_type(a, NJ_PLAYER_RIGHT2, "platformer/zombie_walk2.png");
        assets_type(a, NJ_BOMB, "misc_assets/bomb.png");
        assets_type(a, NJ_THROWING_STAR, "misc_assets/saw.png");
        assets_type(a, NJ_FIRE, "misc_assets/bomb.png");
        a->n_bg = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0]));
        for (int i = 0; i < a->n_bg; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[i], 1);
    } else if (game_id == GAME_HEIST) { /* heist.cpp:41-57 */
        assets_type(a, WALL_OBJ, "kenney/Ground/Dirt/dirtCenter.png");
        assets_type(a, HS_EXIT, "misc_assets/gemYellow.png");
        assets_type(a, PLAYER, "misc_assets/spaceAstronauts_008.png");
        assets_type(a, HS_KEY, "misc_assets/keyBlue.png");
        assets_type(a, HS_KEY, "misc_assets/keyGreen.png");
        assets_type(a, HS_KEY, "misc_assets/keyRed.png");
        assets_type(a, HS_LOCKED_DOOR, "misc_assets/lock_blue.png");
        assets_type(a, HS_LOCKED_DOOR, "misc_assets/lock_green.png");
        assets_type(a, HS_LOCKED_DOOR, "misc_assets/lock_red.png");
        assets_topdown_backgrounds(a);
    } else if (game_id == GAME_PLUNDER) { /* plunder.cpp:45-63 */
        for (int i = 1; i <= 6; i++) {
            snprintf(buf, sizeof buf, "misc_assets/ship_%d.png", i);
            assets_type(a, PL_SHIP, buf);
        }
        assets_type(a, PL_PLAYER_BULLET, "misc_assets/cannonBall.png");
        assets_type(a, PL_PANEL, "misc_assets/panel_wood.png");
        assets_type(a, PL_TARGET_BACKGROUND, "misc_assets/target_red2.png");
        /* water_surface_backgrounds, reference src/resources.cpp:933-940 */
        static const char *WS[] = {"water_backgrounds/water1.png", "water_backgrounds/water2.png", "water_backgrounds/water3.png", "water_backgrounds/water4.png"};
        a->n_bg = 4;
        for (int i = 0; i < 4; i++) a->bg_img[i] = assets_add(a, WS[i], 1);
    } else if (game_id == GAME_LEAPER) { /* leaper.cpp:40-66 */
        assets_type(a, LP_ROAD, "misc_assets/roadTile6b.png");
        assets_type(a, LP_WATER, "misc_assets/terrainTile6.png");
        assets_type(a, LP_CAR, "misc_assets/car_yellow_5.png");
        assets_type(a, LP_CAR, "misc_assets/car_black_1.png");
        assets_type(a, LP_CAR, "misc_assets/car_blue_2.png");
        assets_type(a, LP_CAR, "misc_assets/car_green_3.png");
        assets_type(a, LP_CAR, "misc_assets/car_red_4.png");
        assets_type(a, LP_LOG, "misc_assets/elementWood044.png");
        assets_type(a, PLAYER, "misc_assets/frog1.png");
        assets_type(a, PLAYER, "misc_assets/frog2.png");
        assets_type(a, PLAYER, "misc_assets/frog4.png");
        assets_type(a, PLAYER, "misc_assets/frog6.png");
        assets_type(a, PLAYER, "misc_assets/frog7.png");
        assets_type(a, LP_FINISH_LINE, "misc_assets/finish2.png");
        assets_topdown_backgrounds(a);
    } else if (game_id == GAME_FRUITBOT) { /* fruitbot.cpp:42-78 */
        assets_type(a, PLAYER, "misc_assets/robot_3Dblue.png");
        assets_type(a, FB_BARRIER, "misc_assets/tileStone_slope.png");
        assets_type(a, FB_OUT_OF_BOUNDS_WALL, "misc_assets/tileStone_slope.png");
        assets_type(a, FB_PLAYER_BULLET, "misc_assets/keyRed2.png");
        for (int i = 1; i <= 6; i++) {
            snprintf(buf, sizeof buf, "misc_assets/food%d.png", i);
            assets_type(a, FB_BAD_OBJ, buf);
        }
        for (int i = 1; i <= 6; i++) {
            snprintf(buf, sizeof buf, "misc_assets/fruit%d.png", i);
            assets_type(a, FB_GOOD_OBJ, buf);
        }
        assets_type(a, FB_LOCKED_DOOR, "misc_assets/fenceYellow.png");
        assets_type(a, FB_LOCK, "misc_assets/lockRed2.png");
        for (int i = 1; i <= 3; i++) {
            snprintf(buf, sizeof buf, "misc_assets/present%d.png", i);
            assets_type(a, FB_PRESENT, buf);
        }
        assets_topdown_backgrounds(a);
    } else if (game_id == GAME_STARPILOT) { /* starpilot.cpp:59-106 */
        assets_type(a, PLAYER, "misc_assets/playerShip2_blue.png");
        assets_type(a, SP_BULLET_PLAYER, "misc_assets/towerDefense_tile295.png");
        assets_type(a, SP_BULLET2, "misc_assets/towerDefense_tile296.png");
        assets_type(a, SP_BULLET3, "misc_assets/towerDefense_tile297.png");
        for (int t = 0; t < 2; t++)
            for (int i = 1; i <= 7; i++) {
                snprintf(buf, sizeof buf, "misc_assets/spaceShips_00%d.png", i);
                assets_type(a, t == 0 ? SP_FLYER : SP_FAST_FLYER, buf);
            }
        for (int i = 1; i <= 4; i++) {
            snprintf(buf, sizeof buf, "misc_assets/spaceMeteors_00%d.png", i);
            assets_type(a, SP_METEOR, buf);
        }
        for (int i = 1; i <= 4; i++) {
            snprintf(buf, sizeof buf, "misc_assets/meteorGrey_big%d.png", i);
            assets_type(a, SP_METEOR, buf);
        }
        for (int i = 1; i <= 9; i++) {
            snprintf(buf, sizeof buf, "misc_assets/spaceEffect%d.png", i);
            assets_type(a, SP_CLOUD, buf);
        }
        assets_type(a, SP_TURRET, "misc_assets/spaceStation_018.png");
        assets_type(a, SP_TURRET, "misc_assets/spaceStation_019.png");
        for (int i = 1; i <= 4; i++) {
            snprintf(buf, sizeof buf, "misc_assets/spaceRockets_00%d.png", i);
            assets_type(a, SP_FINISH_LINE, buf);
        }
        /* starpilot.cpp:55-57 load_background_images: space_backgrounds (resources.cpp:829-845) */
        int n_platform = (int)(sizeof(PLATFORM_BGS) / sizeof(PLATFORM_BGS[0])) - 13;
        a->n_bg = 13;
        for (int i = 0; i < 13; i++) a->bg_img[i] = assets_add(a, PLATFORM_BGS[n_platform + i], 1);
    } else if (game_id == GAME_MAZE) { /* maze.cpp:26-38 */
        assets_type(a, WALL_OBJ, "kenney/Ground/Sand/sandCenter.png");
        assets_type(a, MZ_GOAL, "misc_assets/cheese.png");
        assets_type(a, PLAYER, "kenney/Enemies/mouse_move.png");
        assets_topdown_backgrounds(a);
    } else {
        fatal("game not restated in the oracle");
    }
}

int pgo_game_id(const char *name) {
    if (strcmp(name, "coinrun") == 0) return GAME_COINRUN;
    if (strcmp(name, "bigfish") == 0) return GAME_BIGFISH;
    if (strcmp(name, "maze") == 0) return GAME_MAZE;
    if (strcmp(name, "climber") == 0) return GAME_CLIMBER;
    if (strcmp(name, "miner") == 0) return GAME_MINER;
    if (strcmp(name, "starpilot") == 0) return GAME_STARPILOT;
    if (strcmp(name, "fruitbot") == 0) return GAME_FRUITBOT;
    if (strcmp(name, "leaper") == 0) return GAME_LEAPER;
    if (strcmp(name, "plunder") == 0) return GAME_PLUNDER;
    if (strcmp(name, "heist") == 0) return GAME_HEIST;
    if (strcmp(name, "ninja") == 0) return GAME_NINJA;
    if (strcmp(name, "dodgeball") == 0) return GAME_DODGEBALL;
    if (strcmp(name, "bossfight") == 0) return GAME_BOSSFIGHT;
    if (strcmp(name, "chaser") == 0) return GAME_CHASER;
    if (strcmp(name, "caveflyer") == 0) return GAME_CAVEFLYER;
    if (strcmp(name, "jumper") == 0) return GAME_JUMPER;
    return -1;
}
int pgo_num_images(int game_id) {
    assets_build(game_id);
    return g_assets[game_id].n;
}
const char *pgo_image_name(int game_id, int idx) {
    assets_build(game_id);
    return g_assets[game_id].names[idx];
}
int pgo_image_is_background(int game_id, int idx) {
    assets_build(game_id);
    return g_assets[game_id].is_bg[idx];
}
void pgo_set_image(int game_id, int idx, int w, int h, const uint32_t *px) {
    assets_build(game_id);
    Img *im = &g_assets[game_id].img[idx];
    free(im->px);
    im->w = w;
    im->h = h;
    im->px = (uint32_t *)malloc((size_t)w * h * 4);
    memcpy(im->px, px, (size_t)w * h * 4);
}

/* ------------------------------------------------------------------------------------------- */
/* Game state: reference src/game.h:62-126, src/basic-abstract-game.h:110-160, coinrun.cpp:38-47 */
#define MAX_ENTS 2048
#define MAX_GRID (64 * 64)

typedef struct {
    int game_id;
    GameAssets *assets;
    PgoOptions opt;
    /* Game */
    int level_seed_low, level_seed_high;
    Rng level_seed_rand_gen, rand_gen;
    float reward;
    int done, level_complete;
    int action, timeout;
    int current_level_seed, prev_level_seed, episodes_remaining, episode_done;
    int last_reward_timer;
    float last_reward;
    int default_action;
    int cur_time;
    int grid_step;
    float total_reward;
    uint32_t render_buf[RES_W * RES_H];
    /* BAG */
    Ent pool[MAX_ENTS];
    int pool_free[MAX_ENTS], n_free;
    int ents[MAX_ENTS]; /* entity list (indices into pool), in vector order */
    int n_ents;
    int agent; /* pool index; stays valid after erase (shared_ptr semantics) */
    int background_index;
    Img gen_bg; /* use_generated_assets: this episode's background (BAG:58-63,769-773) */
    float bg_tile_ratio, bg_pct_x;
    int last_move_action, move_action, special_action;
    float mixrate, maxspeed, max_jump;
    float action_vx, action_vy, action_vrot;
    float center_x, center_y;
    int random_agent_start, has_useful_vel_info, step_rand_int;
    int main_width, main_height, out_of_bounds_object;
    float unit, view_dim, x_off, y_off, visibility, min_visibility;
    int grid_w, grid_h;
    int grid[MAX_GRID];
    int center_agent; /* options.center_agent: some games overwrite it in game_reset (bigfish.cpp:64) */
    /* BigFish: bigfish.cpp:22-23 */
    int fish_eaten;
    float r_inc;
    /* Climber: climber.cpp:32-38 (has_support, facing_right, wall_theme, gravity, air_control shared with CoinRun below) */
    int coin_quota, coins_collected;
    /* MinerGame: miner.cpp:23 */
    int diamonds_remaining;
    /* MazeGame: maze.cpp:12-14 */
    int maze_dim, world_dim;
    /* Jumper: jumper.cpp:31-39 (has_support, facing_right, wall_theme shared) */
    int goal; /* pool id */
    int jump_count, jump_delta, jump_time;
    float compass_dim;
    /* ChaserGame: chaser.cpp:27-36 (maze_dim shared with MazeGame; free_cells / is_space_vec follow from the grid) */
    int eat_timeout, egg_timeout, eat_time, total_enemies, total_orbs, orbs_collected;
    /* BossfightGame: bossfight.cpp:35-61 (last_fire_time shared) */
    int boss, shields; /* pool ids */
    int attack_modes[8], n_attack_modes;
    int time_to_swap, invulnerable_duration, vulnerable_duration, num_rounds, round_num, round_health;
    int boss_vel_timeout, curr_vel_timeout, attack_mode, player_laser_theme, boss_laser_theme, damaged_until_time;
    int shields_are_up, barriers_moves_right;
    float base_fire_prob, boss_bullet_vel, barrier_vel, barrier_spawn_prob, rand_pct, rand_fire_pct, rand_pct_x, rand_pct_y;
    /* DodgeballGame: dodgeball.cpp:29-36 (min_dim, last_fire_time shared with FruitBot) */
    float hard_min_dim, ball_vscale, ball_r;
    int num_enemies, enemy_fire_delay;
    /* Ninja: ninja.cpp:25-32 (has_support, facing_right, last_fire_time, wall_theme, gravity, air_control shared) */
    float jump_charge, jump_charge_inc;
    /* HeistGame: heist.cpp:19-22 (world_dim shared with MazeGame below) */
    int num_keys, has_keys[4];
    /* PlunderGame: plunder.cpp:19-31 (last_fire_time shared with FruitBot below) */
    int lane_directions[5], target_bools[6], image_permutation[6];
    float lane_vels[5];
    int num_lanes, num_current_ship_types, targets_hit, target_quota;
    float juice_left, r_scale, spawn_prob, legend_r, min_agent_x;
    /* LeaperGame: leaper.cpp:29-33 */
    int bottom_road_y, bottom_water_y, goal_y, n_road_lanes, n_water_lanes;
    float road_lane_speeds[8], water_lane_speeds[8];
    /* FruitBotGame: fruitbot.cpp:28-30 */
    float min_dim, bullet_vscale;
    int last_fire_time;
    /* StarPilotGame: starpilot.cpp:34-50 */
    Ent spawners[SP_MAX_SPAWNERS];
    int n_spawners;
    float hp_vs[SP_NUM_BASIC_OBJECTS], hp_healths[SP_NUM_BASIC_OBJECTS], hp_bullet_r[SP_NUM_BASIC_OBJECTS];
    float hp_object_r[SP_NUM_BASIC_OBJECTS], hp_object_prob_weight[SP_NUM_BASIC_OBJECTS];
    float total_prob_weight, hp_slow_v, hp_weapon_bullet_dist, hp_spawn_right_threshold;
    int hp_min_enemy_delta_t, hp_max_group_size, hp_max_enemy_delta_t;
    float char_dim; /* BAG:24 */
    /* CoinRun */
    float last_agent_y;
    int wall_theme, has_support, facing_right, is_on_crate;
    float gravity, air_control;
} Game;

struct PgoVec {
    int n;
    Game *games;
};

/* ---- entity list helpers (std::vector<std::shared_ptr<Entity>> semantics) ---- */
static int pool_alloc(Game *g) {
    if (g->n_free <= 0) fatal("entity pool exhausted");
    return g->pool_free[--g->n_free];
}
static Ent *push_entity(Game *g, float x, float y, float vx, float vy, float rx, float ry, int type) { /* BAG:566-576 */
    int id = pool_alloc(g);
    ent_init(&g->pool[id], x, y, vx, vy, rx, ry, type);
    if (g->n_ents >= MAX_ENTS) fatal("entity list overflow");
    g->ents[g->n_ents++] = id;
    return &g->pool[id];
}
static void ents_clear(Game *g) {
    g->n_ents = 0;
    g->n_free = 0;
    for (int i = MAX_ENTS - 1; i >= 0; i--) g->pool_free[g->n_free++] = i;
}

/* ---- grid: reference src/grid.h, BAG:125-131,167-223 ---- */
static int grid_contains(const Game *g, int x, int y) { return 0 <= y && y < g->grid_h && 0 <= x && x < g->grid_w; }
static int get_obj(const Game *g, int x, int y) { /* BAG:180-185 */
    if (!grid_contains(g, x, y)) return g->out_of_bounds_object;
    return g->grid[y * g->grid_w + x];
}
static void set_obj(Game *g, int x, int y, int v) { /* grid.h:54-57 */
    if (!grid_contains(g, x, y)) fatal("fassert grid.contains (grid.h:55)");
    g->grid[y * g->grid_w + x] = v;
}
static void fill_elem(Game *g, int x, int y, int dx, int dy, int elem) { /* BAG:125-131 (elem is a char there) */
    for (int j = 0; j < dx; j++)
        for (int k = 0; k < dy; k++) set_obj(g, x + j, y + k, elem);
}
static int get_obj_from_floats(const Game *g, float i, float j) { /* BAG:167-174 */
    if (i < 0) return g->out_of_bounds_object;
    if (j < 0) return g->out_of_bounds_object;
    return get_obj(g, (int)floor(i), (int)floor(j));
}

/* ---- collision predicates ---- */
static int has_collision(const Ent *e1, const Ent *e2, float margin) { /* BAG:1145-1150 */
    float threshold_x = (e1->rx + e2->rx) + margin;
    float threshold_y = (e1->ry + e2->ry) + margin;
    return (fabsf(e1->x - e2->x) < threshold_x) && (fabsf(e1->y - e2->y) < threshold_y);
}
static int is_out_of_bounds(const Game *g, const Ent *e) { /* BAG:1068-1084 */
    if (e->x + e->rx < 0) return 1;
    if (e->y + e->ry < 0) return 1;
    if (e->x - e->rx > g->main_width) return 1;
    if (e->y - e->ry > g->main_height) return 1;
    return 0;
}
static int has_agent_collision(const Game *g, const Ent *e) { /* BAG:1126-1131 */
    if (e->type == PLAYER) return 0;
    return has_collision(e, &g->pool[g->agent], e->collision_margin);
}

/* ---- per-game hooks (virtuals of BasicAbstractGame) ---- */
static int cr_is_wall(int t) { return t == CR_WALL_MID || t == CR_WALL_TOP; }
static int cr_is_lava(int t) { return t == CR_LAVA_MID || t == CR_LAVA_TOP; }

static int hook_is_blocked(const Game *g, const Ent *src, int target, int is_horizontal) {
    (void)is_horizontal;
    if (g->game_id == GAME_NINJA && target == NJ_WALL_MID) { /* ninja.cpp:142-156 */
        if (src->type == PLAYER) return 1;
        if (src->type == NJ_THROWING_STAR) { /* throwing stars stick to walls */
            ((Ent *)src)->vx = 0;
            ((Ent *)src)->vy = 0;
            return 1;
        }
    }
    if (target == WALL_OBJ) return 1; /* BAG:485-492 */
    if (target == g->out_of_bounds_object) return 1;
    if (g->game_id == GAME_COINRUN || g->game_id == GAME_CLIMBER) { /* coinrun.cpp:204-211, climber.cpp:136-143 */
        if (src->type == PLAYER && cr_is_wall(target)) return 1;
    }
    if (g->game_id == GAME_CHASER && target == CH_MAZE_WALL) return 1; /* chaser.cpp:90-95 */
    if (g->game_id == GAME_CAVEFLYER && src->type == PLAYER && target == CF_CAVEWALL) return 1; /* caveflyer.cpp:87-94 */
    if (g->game_id == GAME_JUMPER && src->type == PLAYER && (target == JP_CAVEWALL || target == JP_CAVEWALL_TOP)) return 1; /* jumper.cpp:108-115 */
    if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:84-86 */
        if (src->type == PLAYER && target == FB_OUT_OF_BOUNDS_WALL) return 1;
    }
    if (g->game_id == GAME_MINER) { /* miner.cpp:57-64 */
        if (src->type == PLAYER && (target == MN_BOULDER || target == MN_MOVING_BOULDER || target == MN_OOB_WALL)) return 1;
    }
    return 0;
}
static int hook_is_blocked_ents(Game *g, const Ent *src, const Ent *target, int is_horizontal) {
    if (g->game_id == GAME_HEIST && target->type == HS_LOCKED_DOOR) return !g->has_keys[target->image_theme]; /* heist.cpp:63-68 */
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:187-202 */
        if (target->type == CR_CRATE && !is_horizontal) {
            const Ent *agent = &g->pool[g->agent];
            if (agent->vy >= 0) return 0;
            if (g->action_vy < 0) return 0;
            if (g->last_agent_y < (target->y + target->ry + agent->ry)) return 0;
            g->is_on_crate = 1;
            return 1;
        }
    }
    return hook_is_blocked(g, src, target->type, is_horizontal); /* BAG:494-496 */
}
static int hook_will_reflect(const Game *g, int src, int target) {
    if (g->game_id == GAME_COINRUN || g->game_id == GAME_CLIMBER) /* coinrun.cpp:140-142, climber.cpp:110-112 (same ids) */
        return (src == CR_ENEMY && (cr_is_wall(target) || target == CR_ENEMY_BARRIER));
    if (g->game_id == GAME_CAVEFLYER) /* caveflyer.cpp:127-129 */
        return (src == CF_ENEMY && (target == CF_CAVEWALL || target == g->out_of_bounds_object));
    if (g->game_id == GAME_DODGEBALL) /* dodgeball.cpp:98-100 */
        return (src == DB_ENEMY && (target == DB_LAVA_WALL || target == g->out_of_bounds_object));
    if (g->game_id == GAME_FRUITBOT) /* fruitbot.cpp:80-82 */
        return (src == FB_BAD_OBJ && (target == FB_BARRIER || target == WALL_OBJ));
    if (g->game_id == GAME_MINER) /* miner.cpp:66-68 */
        return (src == MN_ENEMY && (target == MN_BOULDER || target == MN_DIAMOND || target == MN_MOVING_BOULDER || target == MN_MOVING_DIAMOND || target == g->out_of_bounds_object));
    return 0; /* BAG:498-500 */
}
static int sp_is_lethal(int type) { /* starpilot.cpp:341-345 */
    return type == SP_FLYER || type == SP_FAST_FLYER || type == SP_BULLET2 || type == SP_BULLET3 || type == SP_TURRET || type == SP_METEOR;
}
static int sp_is_destructible(int type) { /* starpilot.cpp:347-349 */
    return type == SP_FLYER || type == SP_FAST_FLYER || type == SP_TURRET || type == SP_METEOR;
}
static void hook_handle_agent_collision(Game *g, Ent *obj) {
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:123-131 */
        if (obj->type == CR_ENEMY) g->done = 1;
        else if (obj->type == CR_SAW) g->done = 1;
    } else if (g->game_id == GAME_MINER) { /* miner.cpp:70-82 */
        if (obj->type == MN_ENEMY) {
            g->done = 1;
        } else if (obj->type == MN_EXIT) {
            if (g->diamonds_remaining == 0) {
                g->reward += 10.0f;
                g->level_complete = 1;
                g->done = 1;
            }
        }
    } else if (g->game_id == GAME_CLIMBER) { /* climber.cpp:90-100 */
        if (obj->type == CL_ENEMY) {
            g->done = 1;
        } else if (obj->type == CL_COIN) {
            g->reward += 1.0f;
            g->coins_collected += 1;
            obj->will_erase = 1;
        }
    } else if (g->game_id == GAME_JUMPER) { /* jumper.cpp:82-92 */
        if (obj->type == JP_GOAL) {
            g->reward += 10.0f;
            g->level_complete = 1;
            g->done = 1;
        } else if (obj->type == JP_SPIKE) {
            g->done = 1;
        }
    } else if (g->game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:55-69 */
        if (obj->type == CF_GOAL) {
            g->reward += 10.0f;
            g->level_complete = 1;
            g->done = 1;
        } else if (obj->type == CF_OBSTACLE || obj->type == CF_ENEMY || obj->type == CF_TARGET) {
            g->done = 1;
        }
    } else if (g->game_id == GAME_CHASER) { /* chaser.cpp:121-135 */
        if (obj->type == CH_LARGE_ORB) {
            g->eat_time = g->cur_time;
            g->reward += CH_ORB_REWARD;
            obj->will_erase = 1;
        } else if (obj->type == CH_ENEMY) {
            if (g->cur_time - g->eat_time < g->eat_timeout) obj->will_erase = 1;
            else g->done = 1;
        }
    } else if (g->game_id == GAME_BOSSFIGHT) { /* bossfight.cpp:109-120 */
        if (obj->type == BF2_BOSS) g->done = 1;
        else if (obj->type == BF2_BARRIER) g->done = 1;
        if (obj->type == BF2_ENEMY_BULLET) g->done = 1;
    } else if (g->game_id == GAME_DODGEBALL) { /* dodgeball.cpp:102-118 */
        if (obj->type == DB_ENEMY) {
            g->done = 1;
        } else if (obj->type == DB_ENEMY_BALL) {
            g->done = 1;
        } else if (obj->type == DB_DOOR) {
            if (g->num_enemies == 0) {
                g->done = 1;
                g->reward += 10.0f;
                g->level_complete = 1;
            }
        } else if (obj->type == DB_LAVA_WALL) {
            g->done = 1;
        }
    } else if (g->game_id == GAME_NINJA) { /* ninja.cpp:77-87 */
        if (obj->type == EXPLOSION) {
            g->done = 1;
        } else if (obj->type == NJ_GOAL) {
            g->reward += 10.0f;
            g->level_complete = 1;
            g->done = 1;
        }
    } else if (g->game_id == GAME_HEIST) { /* heist.cpp:77-93 */
        if (obj->type == HS_EXIT) {
            g->done = 1;
            g->reward = 10.0f;
            g->level_complete = 1;
        } else if (obj->type == HS_KEY) {
            obj->will_erase = 1;
            g->has_keys[obj->image_theme] = 1;
        } else if (obj->type == HS_LOCKED_DOOR) {
            if (g->has_keys[obj->image_theme]) obj->will_erase = 1;
        }
    } else if (g->game_id == GAME_LEAPER) { /* leaper.cpp:77-85 */
        const Ent *agent = &g->pool[g->agent];
        if (obj->type == LP_CAR) {
            g->done = 1;
        } else if (obj->type == LP_FINISH_LINE && agent->vx == 0 && agent->vy == 0) {
            g->reward += 10.0f;
            g->done = 1;
            g->level_complete = 1;
        }
    } else if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:96-118 */
        if (obj->type == FB_BARRIER) {
            g->done = 1;
        } else if (obj->type == FB_BAD_OBJ) {
            g->reward += -4.0f;
            obj->will_erase = 1;
        } else if (obj->type == FB_LOCKED_DOOR) {
            g->done = 1;
        } else if (obj->type == FB_GOOD_OBJ) {
            g->reward += 1.0f;
            obj->will_erase = 1;
        } else if (obj->type == FB_PRESENT) {
            g->reward += 10.0f;
            g->done = 1;
            g->level_complete = 1;
        }
    } else if (g->game_id == GAME_STARPILOT) { /* starpilot.cpp:126-136 */
        if (obj->type == SP_FINISH_LINE) {
            g->done = 1;
            g->reward += 10.0f;
            g->level_complete = 1;
        } else if (sp_is_lethal(obj->type)) {
            g->done = 1;
        }
    } else if (g->game_id == GAME_BIGFISH) { /* bigfish.cpp:48-62 */
        Ent *agent = &g->pool[g->agent];
        if (obj->type == BF_FISH) {
            if (obj->rx > agent->rx) {
                g->done = 1;
            } else {
                g->reward += 1.0f; /* POSITIVE_REWARD is an int constant 1 */
                obj->will_erase = 1;
                agent->rx += g->r_inc;
                agent->ry += g->r_inc;
                g->fish_eaten += 1;
            }
        }
    }
}
static void hook_handle_grid_collision(Game *g, Ent *obj, int type, int i, int j) {
    if (g->game_id == GAME_NINJA) { /* ninja.cpp:89-107 */
        if (obj->type == PLAYER) {
            if (type == NJ_FIRE) g->done = 1;
            else if (type == NJ_BOMB) g->done = 1;
        } else if (obj->type == NJ_THROWING_STAR) {
            if (type == NJ_BOMB) {
                obj->will_erase = 1;
                set_obj(g, i, j, SPACE);
                push_entity(g, (float)(i + .5), (float)(j + .5), 0, 0, (float).5, (float).5, EXPLOSION);
            }
            if (type == NJ_WALL_MID) obj->will_erase = 1;
        }
    }
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:144-154 */
        if (obj->type == PLAYER) {
            if (type == CR_GOAL) {
                g->reward += 10.0f;
                g->done = 1;
                g->level_complete = 1;
            } else if (cr_is_lava(type)) {
                g->done = 1;
            }
        }
    }
}
static void bf2_prepare_boss(Game *g) { /* bossfight.cpp:194-201 */
    g->shields_are_up = 1;
    g->curr_vel_timeout = g->boss_vel_timeout;
    g->time_to_swap = g->invulnerable_duration;
    g->attack_mode = g->attack_modes[g->round_num % g->n_attack_modes];
    g->pool[g->boss].vx = 0;
    g->pool[g->boss].vy = 0;
}
static void hook_handle_collision(Game *g, Ent *src, Ent *target) { /* BAG:398 */
    if (g->game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:96-125 */
        if (target->type == CF_PLAYER_BULLET) {
            int erase_bullet = 0;
            if (src->type == CF_TARGET) {
                src->health -= 1;
                erase_bullet = 1;
                if (src->health <= 0 && !src->will_erase) {
                    float r = (float)(.5 * src->rx);
                    push_entity(g, src->x, src->y, 0, 0, r, r, EXPLOSION);
                    src->will_erase = 1;
                    g->reward += 3.0f;
                }
            } else if (src->type == CF_OBSTACLE || src->type == CF_ENEMY || src->type == CF_GOAL) {
                erase_bullet = 1;
            }
            if (erase_bullet && !target->will_erase) {
                target->will_erase = 1;
                float r = (float)(.5 * target->rx);
                Ent *ex = push_entity(g, target->x, target->y, 0, 0, r, r, EXPLOSION);
                ex->vx = src->vx;
                ex->vy = src->vy;
            }
        }
    }
    if (g->game_id == GAME_BOSSFIGHT) { /* bossfight.cpp:129-192 */
        if (src->type == BF2_PLAYER_BULLET) {
            int will_erase = 0;
            if (target->type == BF2_SHIELDS) {
                if (g->shields_are_up) {
                    src->type = BF2_REFLECTED_BULLET;
                    float theta = (float)(PI_F * (1.25 + .5 * g->rand_pct));
                    src->vy = (float)(1 * sin((double)theta) * .5);
                    src->vx = (float)(1 * cos((double)theta) * .5);
                    src->expire_time = 4;
                    src->life_time = 0;
                    src->alpha_decay = 0.8f;
                }
            } else if (target->type == BF2_BOSS) {
                if (!g->shields_are_up) {
                    target->health -= 1;
                    will_erase = 1;
                    if ((int)target->health % g->round_health == 0) {
                        g->reward += 1.0f;
                        if (target->health == 0) {
                            g->done = 1;
                            g->reward += 10.0f;
                            g->level_complete = 1;
                        } else {
                            g->round_num++;
                            bf2_prepare_boss(g);
                            g->curr_vel_timeout = BF2_BOSS_DAMAGED_TIMEOUT;
                            g->damaged_until_time = g->cur_time + BF2_BOSS_DAMAGED_TIMEOUT;
                        }
                    }
                }
            }
            if (will_erase && !src->will_erase) {
                src->will_erase = 1;
                float r = (float)(.5 * src->rx);
                float tvx = target->vx, tvy = target->vy;
                Ent *ex = push_entity(g, src->x, src->y, 0, 0, r, r, EXPLOSION);
                ex->vx = tvx;
                ex->vy = tvy;
            }
        } else if (src->type == BF2_BARRIER) {
            if (target->type == BF2_ENEMY_BULLET || target->type == BF2_PLAYER_BULLET) {
                target->will_erase = 1;
                float r = (float)(.5 * target->rx);
                push_entity(g, target->x, target->y, 0, 0, r, r, EXPLOSION);
            } else if (target->type == BF2_LASER_TRAIL) {
                target->will_erase = 1;
            }
            if (src->health <= 0) {
                if (!src->will_erase) {
                    float r = (float)(.5 * src->rx);
                    Ent *ex = push_entity(g, src->x, src->y, 0, 0, r, r, EXPLOSION);
                    ex->vx = src->vx;
                    ex->vy = src->vy;
                }
                src->will_erase = 1;
            }
        }
    }
    if (g->game_id == GAME_DODGEBALL) { /* dodgeball.cpp:120-151 */
        if (target->type == DB_PLAYER_BALL) {
            if (src->type == DB_LAVA_WALL) {
                target->will_erase = 1;
            } else if (src->type == DB_ENEMY) {
                src->health -= 1;
                target->will_erase = 1;
                if (src->health <= 0 && !src->will_erase) {
                    src->will_erase = 1;
                    g->reward += 2.0f;
                    Ent *ent = push_entity(g, src->x, src->y, 0, 0, src->rx, src->rx, DB_DUST_CLOUD); /* spawn_child BAG:225-231, match_vel = false */
                    ent->vrot = PI_F / 0.3f;
                    ent->grow_rate = 1.0f / 1.2f;
                    ent->expire_time = 4;
                    ent->alpha_decay = 0.9f;
                    ent->image_theme = g->step_rand_int % g->assets->type_num_themes[ent->image_type]; /* choose_step_random_theme BAG:1043-1046 */
                }
            }
        } else if (target->type == DB_ENEMY_BALL) {
            if (src->type == DB_LAVA_WALL) target->will_erase = 1;
        }
    }
    if (g->game_id == GAME_PLUNDER) { /* plunder.cpp:87-109 */
        if (src->type == PL_PLAYER_BULLET) {
            if (target->type == PL_SHIP) {
                target->will_erase = 1;
                src->will_erase = 1;
                if (g->target_bools[target->image_theme]) {
                    g->targets_hit += 1;
                    g->reward += 1.0f;
                    g->juice_left += 0.1f;
                } else {
                    g->juice_left -= 0.1f;
                }
            } else if (target->type == PL_PANEL) {
                src->will_erase = 1;
            }
            if (target->will_erase) {
                float r = (float)(.5 * target->rx);
                push_entity(g, target->x, target->y, target->vx / 2, target->vy / 2, r, r, EXPLOSION);
            }
        }
    }
    if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:120-138 */
        if (src->type == FB_PLAYER_BULLET) {
            if (target->type == FB_BARRIER) {
                src->will_erase = 1;
            } else if (target->type == FB_LOCK) {
                src->will_erase = 1;
                target->will_erase = 1;
                for (int k = 0; k < g->n_ents; k++) {
                    Ent *ent = &g->pool[g->ents[k]];
                    if (ent->type == FB_LOCKED_DOOR && fabs((double)(ent->y - target->y)) < 1) {
                        ent->will_erase = 1;
                        break;
                    }
                }
            }
        }
    }
    if (g->game_id == GAME_STARPILOT) { /* starpilot.cpp:138-146 */
        if (src->type == SP_BULLET_PLAYER && target->type != SP_CLOUD && sp_is_destructible(target->type)) {
            src->will_erase = 1;
            target->health -= 1;
            float sx = src->x, sy = src->y, tvx = target->vx, tvy = target->vy, r = (float)(.5 * src->rx);
            push_entity(g, sx, sy, tvx, tvy, r, r, EXPLOSION); /* add_entity BAG:571-575; pool slots never move */
        }
    }
}

/* ---- physics: BAG:240-372 ---- */
static int sub_step(Game *g, Ent *obj, float _vx, float _vy, int depth);

static double sign_d(double x) { return x > 0 ? +1 : (x == 0 ? 0 : -1); } /* src/cpp-utils.h:43-45 */

static int push_obj(Game *g, Ent *src, Ent *target, int is_horizontal, int depth) { /* BAG:240-268 */
    float rsum = is_horizontal ? (src->rx + target->rx) : (src->ry + target->ry);
    float delx = target->x - src->x;
    float dely = target->y - src->y;
    float t_vx = 0, t_vy = 0;
    if (is_horizontal) t_vx = (float)(src->x + sign_d(delx) * rsum - target->x);
    else t_vy = (float)(src->y + sign_d(dely) * rsum - target->y);
    int block = 0;
    if (depth < 5) block = sub_step(g, target, t_vx, t_vy, depth + 1);
    if (is_horizontal) target->vx = 0;
    else target->vy = 0;
    return block;
}

static int sub_step(Game *g, Ent *obj, float _vx, float _vy, int depth) { /* BAG:270-372 */
    if (obj->will_erase) return 0;
    float ny = obj->y + _vy;
    float nx = obj->x + _vx;
    float margin = 0.98f;
    int is_horizontal = _vx != 0;
    int block = 0, reflect = 0;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            int type2 = get_obj_from_floats(g, nx + obj->rx * margin * (2 * i - 1), ny + obj->ry * margin * (2 * j - 1));
            block = block || hook_is_blocked(g, obj, type2, is_horizontal);
            reflect = reflect || hook_will_reflect(g, obj->type, type2);
        }
    if (reflect) {
        if (is_horizontal) {
            float delta;
            if (_vx < 0) delta = (float)(ceil(nx - obj->rx) - (nx - obj->rx));
            else delta = (float)(floor(nx + obj->rx) - (nx + obj->rx));
            obj->vx = -1 * obj->vx;
            nx = nx + 2 * delta;
        } else {
            float delta;
            if (_vy < 0) delta = (float)(ceil(ny - obj->ry) - (ny - obj->ry));
            else delta = (float)(floor(ny + obj->ry) - (ny + obj->ry));
            obj->vy = -1 * obj->vy;
            ny = ny + 2 * delta;
        }
    } else if (block) {
        if (is_horizontal) {
            if (g->grid_step) nx = obj->x;
            else nx = (float)(_vx > 0 ? (floor(nx + obj->rx) - obj->rx) : (ceil(nx - obj->rx) + obj->rx));
        } else {
            if (g->grid_step) ny = obj->y;
            else ny = (float)(_vy > 0 ? (floor(ny + obj->ry) - obj->ry) : (ceil(ny - obj->ry) + obj->ry));
        }
    }
    obj->x = nx;
    obj->y = ny;
    int block2 = 0;
    for (int i = g->n_ents - 1; i >= 0; i--) {
        Ent *m = &g->pool[g->ents[i]];
        if (m == obj || m->will_erase) continue;
        int curr_block = 0;
        if (has_collision(obj, m, POS_EPS)) {
            if (hook_is_blocked_ents(g, obj, m, is_horizontal)) {
                curr_block = 1;
            } else if (hook_will_reflect(g, obj->type, m->type)) {
                if (is_horizontal) {
                    float delx = m->x - obj->x;
                    float rsum = m->rx + obj->rx;
                    obj->x += _vx > 0 ? -2 * (rsum - delx) : 2 * (rsum + delx);
                    obj->vx = -1 * obj->vx;
                } else {
                    float dely = m->y - obj->y;
                    float rsum = m->ry + obj->ry;
                    obj->y += _vy > 0 ? -2 * (rsum - dely) : 2 * (rsum + dely);
                    obj->vy = -1 * obj->vy;
                }
            }
            if (curr_block) push_obj(g, m, obj, is_horizontal, depth);
        }
        block2 = block2 || curr_block;
    }
    return block || block2;
}

static void basic_step_object(Game *g, Ent *obj) { /* BAG:593-656 */
    if (obj->will_erase) return;
    int num_sub_steps;
    if (g->grid_step) {
        num_sub_steps = 1;
    } else {
        /* sqrt resolves to the double overload in this TU (checked in the reference object code) */
        num_sub_steps = (int)(4 * sqrt((double)(obj->vx * obj->vx + obj->vy * obj->vy)));
        if (num_sub_steps < 4) num_sub_steps = 4;
    }
    float pct = (float)(1.0 / num_sub_steps);
    float cmp = fabsf(obj->vx) - fabsf(obj->vy);
    int step_x_first = cmp == 0 ? g->step_rand_int % 2 == 0 : (cmp > 0);
    if (obj->type == PLAYER) {
        if (g->action_vx != 0) step_x_first = 1;
        if (g->action_vy != 0) step_x_first = 0;
    }
    float vx_pct = 0, vy_pct = 0;
    for (int s = 0; s < num_sub_steps; s++) {
        int block_x, block_y;
        if (step_x_first) {
            block_x = sub_step(g, obj, obj->vx * pct, 0, 0);
            block_y = sub_step(g, obj, 0, obj->vy * pct, 0);
        } else {
            block_y = sub_step(g, obj, 0, obj->vy * pct, 0);
            block_x = sub_step(g, obj, obj->vx * pct, 0, 0);
        }
        if (!block_x) vx_pct += 1;
        if (!block_y) vy_pct += 1;
        if (block_x && block_y) break;
    }
    vx_pct = vx_pct / num_sub_steps;
    vy_pct = vy_pct / num_sub_steps;
    obj->vx *= vx_pct;
    obj->vy *= vy_pct;
}

static void check_grid_collisions(Game *g, Ent *ent) { /* BAG:145-165 */
    float ax = ent->x, ay = ent->y, arx = ent->rx, ary = ent->ry;
    int min_x = (int)(ax - (arx + POS_EPS));
    int max_x = (int)(ax + (arx + POS_EPS));
    int min_y = (int)(ay - (ary + POS_EPS));
    int max_y = (int)(ay + (ary + POS_EPS));
    for (int x = min_x; x <= max_x; x++)
        for (int y = min_y; y <= max_y; y++) {
            int grid_type = get_obj_from_floats(g, (float)x, (float)y);
            if (grid_type != SPACE) hook_handle_grid_collision(g, ent, grid_type, x, y);
        }
}

static void erase_if_needed(Game *g) { /* BAG:748-756 */
    for (int i = g->n_ents - 1; i >= 0; i--) {
        int id = g->ents[i];
        Ent *e = &g->pool[id];
        if (e->will_erase || (e->auto_erase && is_out_of_bounds(g, e))) {
            for (int k = i; k < g->n_ents - 1; k++) g->ents[k] = g->ents[k + 1];
            g->n_ents--;
            if (id != g->agent) g->pool_free[g->n_free++] = id;
        }
    }
}

/* ---- coinrun control: coinrun.cpp:156-173,447-472 ---- */
static float clip_abs(float x, float y) { /* cpp-utils.h:47-53 */
    if (x > y) return y;
    if (x < -y) return -y;
    return x;
}

static void hook_set_action_xy(Game *g, int move_act) {
    g->action_vx = (float)(move_act / 3 - 1);
    g->action_vy = (float)(move_act % 3 - 1);
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:451-472 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) g->facing_right = 1;
        if (g->action_vx < 0) g->facing_right = 0;
        int obj_below_1 = get_obj_from_floats(g, (float)(agent->x - (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int obj_below_2 = get_obj_from_floats(g, (float)(agent->x + (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int s1 = cr_is_wall(obj_below_1) || obj_below_1 == g->out_of_bounds_object;
        int s2 = cr_is_wall(obj_below_2) || obj_below_2 == g->out_of_bounds_object;
        g->has_support = (g->is_on_crate || s1 || s2) && agent->vy == 0;
        g->is_on_crate = 0;
        if (g->action_vy == 1) {
            if (!g->has_support) g->action_vy = 0;
        }
    } else if (g->game_id == GAME_NINJA) { /* ninja.cpp:318-347 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vy < 0) g->action_vy = 0;
        if (g->action_vx > 0) g->facing_right = 1;
        if (g->action_vx < 0) g->facing_right = 0;
        int obj_below_1 = get_obj_from_floats(g, (float)(agent->x - (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int obj_below_2 = get_obj_from_floats(g, (float)(agent->x + (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int s1 = obj_below_1 == NJ_WALL_MID || obj_below_1 == g->out_of_bounds_object;
        int s2 = obj_below_2 == NJ_WALL_MID || obj_below_2 == g->out_of_bounds_object;
        g->has_support = s1 || s2;
        if (g->has_support && g->action_vy == 1) {
            g->action_vy = 1;
            g->jump_charge += g->jump_charge_inc;
            if (g->jump_charge > 1) g->jump_charge = 1;
        } else {
            g->action_vy = 0;
        }
        if (!g->has_support) g->jump_charge = 0;
    } else if (g->game_id == GAME_CLIMBER) { /* climber.cpp:268-288 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vy < 0) g->action_vy = 0;
        if (g->action_vx > 0) g->facing_right = 1;
        if (g->action_vx < 0) g->facing_right = 0;
        int obj_below_1 = get_obj_from_floats(g, (float)(agent->x - (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int obj_below_2 = get_obj_from_floats(g, (float)(agent->x + (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int s1 = cr_is_wall(obj_below_1) || obj_below_1 == g->out_of_bounds_object;
        int s2 = cr_is_wall(obj_below_2) || obj_below_2 == g->out_of_bounds_object;
        g->has_support = s1 || s2;
        if (g->has_support && g->action_vy == 1) g->action_vy = 1;
        else g->action_vy = 0;
    } else if (g->game_id == GAME_JUMPER) { /* jumper.cpp:398-430 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vy < 0) g->action_vy = 0;
        if (g->action_vx > 0) g->facing_right = 1;
        if (g->action_vx < 0) g->facing_right = 0;
        int obj_below_1 = get_obj_from_floats(g, (float)(agent->x - (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        int obj_below_2 = get_obj_from_floats(g, (float)(agent->x + (agent->rx - .01)), (float)(agent->y - (agent->ry + .01)));
        g->jump_delta = 0;
        int s1 = obj_below_1 == JP_CAVEWALL || obj_below_1 == JP_CAVEWALL_TOP || obj_below_1 == g->out_of_bounds_object;
        int s2 = obj_below_2 == JP_CAVEWALL || obj_below_2 == JP_CAVEWALL_TOP || obj_below_2 == g->out_of_bounds_object;
        g->has_support = s1 || s2;
        if (g->has_support) g->jump_count = 2;
        if (g->action_vy == 1 && g->jump_count > 0 && (g->cur_time - g->jump_time > JP_JUMP_COOLDOWN)) {
            g->jump_count -= 1;
            g->jump_delta = -1;
        } else {
            g->action_vy = 0;
        }
        if (g->action_vy > 0) g->jump_time = g->cur_time;
        g->action_vrot = 0;
    } else if (g->game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:264-285 */
        const Ent *agent = &g->pool[g->agent];
        float acceleration = (float)(move_act % 3 - 1);
        if (acceleration < 0) acceleration *= 0.33f;
        float theta = -1 * agent->rotation + PI_F / 2;
        if (acceleration > 0) {
            float ax = agent->x, ay = agent->y, arx = agent->rx, ary = agent->ry;
            float r = (float)(.5 * arx);
            Ent *exhaust = push_entity(g, (float)(ax - arx * cos((double)theta)), (float)(ay - ary * sin((double)theta)), 0, 0, r, r, CF_EXHAUST);
            exhaust->expire_time = 4;
            exhaust->rotation = -1 * theta - PI_F / 2;
            exhaust->grow_rate = (float)1.25;
            exhaust->alpha_decay = 0.8f;
        }
        g->action_vy = (float)(acceleration * sin((double)theta));
        g->action_vx = (float)(acceleration * cos((double)theta));
        g->action_vrot = (float)(move_act / 3 - 1);
    } else if (g->game_id == GAME_PLUNDER) { /* plunder.cpp:111-115 */
        g->action_vy = 0;
        g->action_vrot = 0;
    } else if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:162-166 */
        g->action_vy = 0.2f;
        g->action_vrot = 0;
    } else {
        g->action_vrot = 0; /* BAG:658-662 */
        if (g->game_id == GAME_MAZE || g->game_id == GAME_MINER) { /* maze.cpp:99-103, miner.cpp:98-102 */
            if (g->action_vx != 0) g->action_vy = 0;
        }
    }
}

static void lp_decay_vel(float *vel) { /* leaper.cpp:208-214, sign() :23-25 */
    float x = (float)(1.0 * *vel);
    float vel_sign = x > 0 ? +1 : (x == 0 ? 0 : -1);
    *vel = (float)(fabs((double)*vel) - LP_VEL_DECAY);
    if (*vel < 0) *vel = 0;
    *vel = *vel * vel_sign;
}
static void hook_update_agent_velocity(Game *g) {
    Ent *agent = &g->pool[g->agent];
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:156-173 */
        float mixrate_x = g->has_support ? g->mixrate : (g->mixrate * g->air_control);
        agent->vx = (1 - mixrate_x) * agent->vx + mixrate_x * g->maxspeed * g->action_vx;
        if (fabsf(agent->vx) < mixrate_x * g->maxspeed) agent->vx = 0;
        if (g->action_vy > 0) {
            agent->vy = g->max_jump;
        } else {
            if (g->has_support) agent->vy = (float)(agent->vy + .2 * g->action_vy);
        }
        if (!(g->has_support && g->action_vy > 0)) {
            agent->vy -= g->gravity;
            agent->vy = clip_abs(agent->vy, g->max_jump);
        }
    } else if (g->game_id == GAME_JUMPER) { /* jumper.cpp:94-100 */
        float v_scale = 1.0;
        agent->vx = (1 - g->mixrate) * agent->vx + g->mixrate * g->maxspeed * g->action_vx * v_scale;
        if (g->action_vy != 0) agent->vy = g->maxspeed * g->action_vy * 2;
    } else if (g->game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:71-78, BAG:502-504,681-684 */
        float v_scale = 1.0;
        agent->vx = (float)(agent->vx + g->mixrate * g->maxspeed * g->action_vx * v_scale * .2);
        agent->vy = (float)(agent->vy + g->mixrate * g->maxspeed * g->action_vy * v_scale * .2);
        agent->vx = (float)(.9 * agent->vx);
        agent->vy = (float)(.9 * agent->vy);
    } else if (g->game_id == GAME_CHASER) { /* chaser.cpp:79-88 */
        if (g->action_vx != 0) agent->vx = g->maxspeed * g->action_vx;
        if (g->action_vy != 0) agent->vy = g->maxspeed * g->action_vy;
        agent->vx = (float)(sign_d(agent->vx) * g->maxspeed);
        agent->vy = (float)(sign_d(agent->vy) * g->maxspeed);
    } else if (g->game_id == GAME_NINJA) { /* ninja.cpp:109-124 */
        float mixrate_x = g->has_support ? g->mixrate : (g->mixrate * g->air_control);
        agent->vx = (1 - mixrate_x) * agent->vx + mixrate_x * g->maxspeed * g->action_vx;
        if (g->action_vy < 1 && g->jump_charge > 0) {
            agent->vy = g->jump_charge * g->max_jump;
            g->jump_charge = 0;
        }
        if (!g->has_support) {
            if (agent->vy > -2) agent->vy -= g->gravity;
        }
    } else if (g->game_id == GAME_CLIMBER) { /* climber.cpp:114-126 */
        float mixrate_x = g->has_support ? g->mixrate : (g->mixrate * g->air_control);
        agent->vx = (1 - mixrate_x) * agent->vx + mixrate_x * g->maxspeed * g->action_vx;
        if (g->action_vy > 0) agent->vy = g->max_jump;
        if (!g->has_support) {
            if (agent->vy > -2) agent->vy -= g->gravity;
        }
    } else if (g->game_id == GAME_LEAPER) { /* leaper.cpp:216-231 */
        if (agent->vx == 0 && agent->vy == 0) {
            if (g->action_vx != 0) {
                agent->vx = g->maxspeed * g->action_vx;
                agent->image_theme = 1;
                agent->rotation = (agent->vx > 0 ? 1 : -1) * PI_F / 2;
            } else if (g->action_vy != 0) {
                agent->vy = g->maxspeed * g->action_vy;
                agent->image_theme = 1;
                agent->rotation = agent->vy > 0 ? 0 : PI_F;
            }
        }
        lp_decay_vel(&agent->vx);
        lp_decay_vel(&agent->vy);
    } else { /* BAG:669-684 */
        float v_scale = 1.0f;
        agent->vx = (1 - g->mixrate) * agent->vx;
        agent->vy = (1 - g->mixrate) * agent->vy;
        agent->vx += g->mixrate * g->maxspeed * g->action_vx * v_scale;
        agent->vy += g->mixrate * g->maxspeed * g->action_vy * v_scale;
        agent->vx = (float)(.9 * agent->vx);
        agent->vy = (float)(.9 * agent->vy);
    }
}

static void step_entities(Game *g) { /* BAG:1086-1098 (count captured before the loop) */
    int entities_count = g->n_ents;
    for (int i = entities_count - 1; i >= 0; i--) {
        Ent *ent = &g->pool[g->ents[i]];
        if (ent->smart_step) basic_step_object(g, ent);
        ent_step(ent);
    }
}

/* ---- BasicAbstractGame::game_step: BAG:686-746 ---- */
static void bag_game_step(Game *g) {
    g->step_rand_int = rng_randint(&g->rand_gen, 0, 1000000);
    g->move_action = g->action % 9;
    g->special_action = 0;
    if (g->action >= 9) {
        g->special_action = g->action - 8;
        g->move_action = 4;
    }
    if (g->move_action != 4) g->last_move_action = g->move_action;
    g->action_vrot = 0;
    g->action_vx = 0;
    g->action_vy = 0;
    hook_set_action_xy(g, g->move_action);
    Ent *agent = &g->pool[g->agent];
    if (g->grid_step) {
        agent->vx = g->action_vx;
        agent->vy = g->action_vy;
    } else {
        hook_update_agent_velocity(g);
        agent->vrot = MIXRATEROT * agent->vrot;
        agent->vrot += MIXRATEROT * MAXVTHETA * g->action_vrot;
    }
    step_entities(g);
    for (int i = g->n_ents - 1; i >= 0; i--) { /* BAG:719-741 */
        Ent *ent = &g->pool[g->ents[i]];
        if (has_agent_collision(g, ent)) hook_handle_agent_collision(g, ent);
        if (ent->collides_with_entities) {
            for (int j = g->n_ents - 1; j >= 0; j--) {
                if (i == j) continue;
                Ent *ent2 = &g->pool[g->ents[j]];
                if (has_collision(ent, ent2, ent->collision_margin) && !ent->will_erase && !ent2->will_erase) hook_handle_collision(g, ent, ent2);
            }
        }
        if (ent->smart_step) check_grid_collisions(g, ent);
    }
    erase_if_needed(g);
    g->done = g->done || is_out_of_bounds(g, &g->pool[g->agent]);
}

static void choose_random_theme(Game *g, Ent *ent);
static void match_aspect_ratio(Game *g, Ent *ent);
static int hook_preserve_type_themes(const Game *g, int type);
static void mn_game_step_tail(Game *g);
static void sp_game_step_tail(Game *g);
static void db_game_step_tail(Game *g);
static void bf2_game_step_tail(Game *g);
static void ch_game_step_tail(Game *g);
static void face_direction(Ent *e, float dx, float dy, float rotation_offset);
static int has_any_collision(const Game *g, const Ent *e1, float margin);
static void lp_spawn_entities(Game *g);

/* ---- per-game game_step: coinrun.cpp:474-498, bigfish.cpp:80-107 ---- */
static void game_step(Game *g) {
    if (g->game_id == GAME_LEAPER) { /* leaper.cpp:241-244 */
        Ent *agent = &g->pool[g->agent];
        if (agent->image_theme >= 1) agent->image_theme = (agent->image_theme + 1) % LP_NSTEP;
    }
    bag_game_step(g);
    if (g->game_id == GAME_COINRUN) {
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 0;
        if (g->action_vx < 0) agent->is_reflected = 1;
        for (int i = g->n_ents - 1; i >= 0; i--) {
            Ent *ent = &g->pool[g->ents[i]];
            if (ent->type == CR_ENEMY) {
                float ty = (float)(ent->y - ent->ry * .5);
                float tx = ent->x;
                Ent *trail = push_entity(g, tx, ty, 0, 0.01f, 0.3f, 0.2f, TRAIL);
                ent = &g->pool[g->ents[i]];
                trail->expire_time = 8;
                trail->alpha = (float).5;
                ent->image_type = g->cur_time / 5 % 2 == 0 ? CR_ENEMY1 : CR_ENEMY2;
                ent->is_reflected = ent->vx > 0;
            } else if (ent->type == CR_SAW) {
                ent->image_type = g->cur_time % 2 == 0 ? CR_SAW : CR_SAW2;
            }
        }
        g->last_agent_y = agent->y;
    } else if (g->game_id == GAME_BIGFISH) { /* bigfish.cpp:83-107 */
        if (rng_randn(&g->rand_gen, 10) == 1) {
            float ent_r = (float)((BF_FISH_MAX_R - BF_FISH_MIN_R) * pow((double)rng_rand01(&g->rand_gen), 1.4) + BF_FISH_MIN_R);
            float ent_y = rng_rand01(&g->rand_gen) * (g->main_height - 2 * ent_r);
            float moves_right = rng_rand01(&g->rand_gen) < .5;
            float ent_vx = (float)((.15 + rng_rand01(&g->rand_gen) * .25) * (moves_right ? 1 : -1));
            float ent_x = moves_right ? -1 * ent_r : g->main_width + ent_r;
            Ent *ent = push_entity(g, ent_x, ent_y, ent_vx, 0, ent_r, ent_r, BF_FISH);
            choose_random_theme(g, ent);
            match_aspect_ratio(g, ent);
            ent->is_reflected = !moves_right;
        }
        if (g->fish_eaten >= BF_FISH_QUOTA) {
            g->done = 1;
            g->reward += 10.0f; /* COMPLETION_BONUS */
            g->level_complete = 1;
        }
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 0;
        if (g->action_vx < 0) agent->is_reflected = 1;
    } else if (g->game_id == GAME_MINER) {
        mn_game_step_tail(g);
    } else if (g->game_id == GAME_STARPILOT) {
        sp_game_step_tail(g);
    } else if (g->game_id == GAME_DODGEBALL) {
        db_game_step_tail(g);
    } else if (g->game_id == GAME_BOSSFIGHT) {
        bf2_game_step_tail(g);
    } else if (g->game_id == GAME_CHASER) {
        ch_game_step_tail(g);
    } else if (g->game_id == GAME_JUMPER) { /* jumper.cpp:432-449 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 0;
        if (g->action_vx < 0) agent->is_reflected = 1;
        if (fabs((double)agent->vx) + fabs((double)agent->vy) > .05) {
            Ent *trail = push_entity(g, agent->x, (float)(agent->y - agent->ry * .5), 0, 0.01f, 0.3f, 0.2f, TRAIL);
            trail->expire_time = 8;
            trail->alpha = (float).5;
        }
        agent = &g->pool[g->agent];
        if (agent->vy > -2) agent->vy -= 0.15f;
    } else if (g->game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:287-324 */
        if (g->special_action == 1) {
            const Ent *agent = &g->pool[g->agent];
            float theta = -1 * agent->rotation + PI_F / 2;
            float vx = (float)cos((double)theta);
            float vy = (float)sin((double)theta);
            float rot = agent->rotation;
            Ent *nb = push_entity(g, agent->x, agent->y, vx, vy, 0.1f, 0.25f, CF_PLAYER_BULLET);
            nb->expire_time = 10;
            nb->rotation = rot;
        }
        for (int k = g->n_ents - 1; k >= 0; k--) {
            Ent *ent = &g->pool[g->ents[k]];
            if (ent->type == CF_ENEMY) face_direction(ent, ent->vx, ent->vy, -1 * PI_F / 2);
            if (ent->type != CF_PLAYER_BULLET) continue;
            int found_wall = 0;
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    int type2 = get_obj_from_floats(g, ent->x + ent->rx * (2 * i - 1), ent->y + ent->ry * (2 * j - 1));
                    found_wall = found_wall || type2 == CF_CAVEWALL;
                }
            if (found_wall) {
                ent->will_erase = 1;
                float r = (float)(.5 * ent->rx);
                push_entity(g, ent->x, ent->y, 0, 0, r, r, EXPLOSION);
            }
        }
        erase_if_needed(g);
    } else if (g->game_id == GAME_NINJA) { /* ninja.cpp:349-383 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 0;
        if (g->action_vx < 0) agent->is_reflected = 1;
        if (g->special_action > 0 && (g->cur_time - g->last_fire_time) >= 3) {
            float theta = 0;
            float bullet_vel = 1;
            if (g->special_action == 1) theta = 0;
            else if (g->special_action == 2) theta = PI_F / 4;
            else if (g->special_action == 3) theta = PI_F / 2;
            else if (g->special_action == 4) theta = -1 * PI_F / 4;
            if (agent->is_reflected) theta = PI_F - theta;
            Ent *nb = push_entity(g, agent->x, agent->y, (float)(bullet_vel * cos((double)theta)), (float)(bullet_vel * sin((double)theta)), (float).25, (float).25, NJ_THROWING_STAR);
            nb->collides_with_entities = 1;
            nb->expire_time = 15;
            nb->smart_step = 1;
            g->last_fire_time = g->cur_time;
        }
    } else if (g->game_id == GAME_HEIST) { /* heist.cpp:196-200 */
        face_direction(&g->pool[g->agent], g->action_vx, g->action_vy, 0);
    } else if (g->game_id == GAME_PLUNDER) { /* plunder.cpp:186-239 */
        g->juice_left -= 0.0015f;
        if (rng_rand01(&g->rand_gen) < g->spawn_prob) {
            float ent_r = g->r_scale;
            int lane = rng_randn(&g->rand_gen, g->num_lanes);
            float ent_y = (float)((lane * .11 + .4) * (g->main_height / 2 - ent_r) + g->main_height / 2);
            float moves_right = (float)g->lane_directions[lane];
            float ent_vx = g->lane_vels[lane] * (moves_right ? 1 : -1);
            Ent m;
            ent_init(&m, 0, ent_y, ent_vx, 0, ent_r, ent_r, PL_SHIP);
            m.image_type = PL_SHIP;
            m.image_theme = g->image_permutation[rng_randn(&g->rand_gen, g->num_current_ship_types)];
            match_aspect_ratio(g, &m);
            m.x = moves_right ? -1 * ent_r : (g->main_width + ent_r);
            m.is_reflected = !moves_right;
            if (!has_any_collision(g, &m, 0)) {
                int id = pool_alloc(g);
                g->pool[id] = m;
                g->ents[g->n_ents++] = id;
            }
        }
        Ent *agent = &g->pool[g->agent];
        if (g->special_action == 1 && (g->cur_time - g->last_fire_time) >= 3) {
            Ent *nb = push_entity(g, agent->x, agent->y, 0, 1, (float).25, (float).25, PL_PLAYER_BULLET);
            nb->collides_with_entities = 1;
            nb->expire_time = 50;
            g->last_fire_time = g->cur_time;
            g->juice_left -= 0.02f;
        }
        if (g->juice_left <= 0) g->done = 1;
        else if (g->juice_left >= 1) g->juice_left = 1;
        if (g->targets_hit >= g->target_quota) {
            g->done = 1;
            g->reward += 10.0f;
            g->level_complete = 1;
        }
        if (agent->x < g->min_agent_x) agent->x = g->min_agent_x;
    } else if (g->game_id == GAME_LEAPER) { /* leaper.cpp:246-275 */
        lp_spawn_entities(g);
        Ent *agent = &g->pool[g->agent];
        int standing_on_log = 0;
        float log_vx = 0.0;
        float margin = -1 * agent->rx;
        for (int k = 0; k < g->n_ents; k++) {
            const Ent *m = &g->pool[g->ents[k]];
            if (m->type == LP_LOG && has_collision(agent, m, margin)) {
                standing_on_log = 1;
                log_vx = m->vx;
            }
        }
        if (get_obj(g, (int)agent->x, (int)agent->y) == LP_WATER) {
            if (!standing_on_log && agent->vx == 0 && agent->vy == 0) g->done = 1;
        }
        if (standing_on_log) agent->x += log_vx;
        if (is_out_of_bounds(g, agent)) g->done = 1;
    } else if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:251-262 */
        if (g->special_action == 1 && (g->cur_time - g->last_fire_time) >= FB_KEY_DURATION) {
            const Ent *agent = &g->pool[g->agent];
            float vx = 0, vy = 1;
            Ent *nb = push_entity(g, agent->x, agent->y, vx * g->bullet_vscale, vy * g->bullet_vscale, (float).25, (float).25, FB_PLAYER_BULLET);
            nb->expire_time = FB_KEY_DURATION;
            nb->collides_with_entities = 1;
            g->last_fire_time = g->cur_time;
        }
    } else if (g->game_id == GAME_CLIMBER) { /* climber.cpp:290-316 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 0;
        if (g->action_vx < 0) agent->is_reflected = 1;
        for (int i = g->n_ents - 1; i >= 0; i--) {
            Ent *ent = &g->pool[g->ents[i]];
            if (ent->type == CL_ENEMY) {
                if (ent->x > ent->climber_spawn_x + CL_PATROL_RANGE) ent->vx = (float)(-1 * fabs((double)ent->vx));
                else if (ent->x < ent->climber_spawn_x - CL_PATROL_RANGE) ent->vx = (float)fabs((double)ent->vx);
                ent->image_type = g->cur_time / 5 % 2 == 0 ? CL_ENEMY1 : CL_ENEMY2;
                ent->is_reflected = ent->vx < 0;
            }
        }
        if (g->coin_quota == g->coins_collected) {
            g->done = 1;
            g->reward += 10.0f;
            g->level_complete = 1;
        }
    } else if (g->game_id == GAME_MAZE) { /* maze.cpp:105-124 */
        Ent *agent = &g->pool[g->agent];
        if (g->action_vx > 0) agent->is_reflected = 1;
        if (g->action_vx < 0) agent->is_reflected = 0;
        int ix = (int)agent->x, iy = (int)agent->y;
        if (get_obj(g, ix, iy) == MZ_GOAL) {
            set_obj(g, ix, iy, SPACE);
            g->reward += 10.0f;
            g->level_complete = 1;
        }
        g->done = g->reward > 0;
    }
}

/* ---- miner helpers: index-based grid access BAG:198-203,217-219, miner.cpp:94-96,207-245 ---- */
static int get_obj_idx(const Game *g, int idx) {
    if (!(0 <= idx && idx < g->grid_w * g->grid_h)) return g->out_of_bounds_object;
    return g->grid[idx];
}
static void set_obj_idx(Game *g, int idx, int v) {
    if (!(idx < g->grid_w * g->grid_h) || idx < 0) fatal("fassert index < w * h (grid.h:60)");
    g->grid[idx] = v;
}
static int mn_agent_index(const Game *g) { return (int)g->pool[g->agent].y * g->main_width + (int)g->pool[g->agent].x; }
static int mn_is_free(const Game *g, int idx) { return get_obj_idx(g, idx) == SPACE && (mn_agent_index(g) != idx); }
static int mn_is_round(int t) { return t == MN_BOULDER || t == MN_MOVING_BOULDER || t == MN_DIAMOND || t == MN_MOVING_DIAMOND; }
static int mn_stationary(int t) { return t == MN_MOVING_DIAMOND ? MN_DIAMOND : (t == MN_MOVING_BOULDER ? MN_BOULDER : t); }
static int mn_moving(int t) { return t == MN_DIAMOND ? MN_MOVING_DIAMOND : (t == MN_BOULDER ? MN_MOVING_BOULDER : t); }

static void mn_game_step_tail(Game *g) { /* miner.cpp:247-307 */
    Ent *agent = &g->pool[g->agent];
    if (g->action_vx > 0) agent->is_reflected = 0;
    if (g->action_vx < 0) agent->is_reflected = 1;
    { /* handle_push miner.cpp:232-245 */
        int agent_idx = mn_agent_index(g);
        int agentx = agent_idx % g->main_width;
        if (g->action_vx == 1 && (agent->vx == 0) && (agentx < g->main_width - 2) && get_obj_idx(g, agent_idx + 1) == MN_BOULDER && get_obj_idx(g, agent_idx + 2) == SPACE) {
            set_obj_idx(g, agent_idx + 1, SPACE);
            set_obj_idx(g, agent_idx + 2, MN_BOULDER);
            agent->x += 1;
        } else if (g->action_vx == -1 && (agent->vx == 0) && (agentx > 1) && get_obj_idx(g, agent_idx - 1) == MN_BOULDER && get_obj_idx(g, agent_idx - 2) == SPACE) {
            set_obj_idx(g, agent_idx - 1, SPACE);
            set_obj_idx(g, agent_idx - 2, MN_BOULDER);
            agent->x -= 1;
        }
    }
    int agent_obj = get_obj(g, (int)agent->x, (int)agent->y);
    if (agent_obj == MN_DIAMOND) g->reward += 1.0f; /* DIAMOND_REWARD is an int constant 1 */
    if (agent_obj == MN_DIRT || agent_obj == MN_DIAMOND) set_obj(g, (int)agent->x, (int)agent->y, SPACE);
    int main_area = g->main_width * g->main_height;
    int diamonds_count = 0;
    for (int idx = 0; idx < main_area; idx++) {
        int obj = get_obj_idx(g, idx);
        int obj_x = idx % g->main_width;
        int agent_idx = (int)((agent->y - .5) * g->main_width + (agent->x - .5));
        int stat_type = mn_stationary(obj);
        if (stat_type == MN_DIAMOND) diamonds_count++;
        if (obj == MN_BOULDER || obj == MN_MOVING_BOULDER || obj == MN_DIAMOND || obj == MN_MOVING_DIAMOND) {
            int below_idx = idx - g->main_width;
            int obj2 = get_obj_idx(g, below_idx);
            int agent_is_below = agent_idx == below_idx;
            if (obj2 == SPACE && !agent_is_below) {
                set_obj_idx(g, idx, SPACE);
                set_obj_idx(g, below_idx, mn_moving(obj));
            } else if (agent_is_below && (obj == MN_MOVING_BOULDER || obj == MN_MOVING_DIAMOND)) {
                g->done = 1;
            } else if (mn_is_round(obj2) && obj_x > 0 && mn_is_free(g, idx - 1) && mn_is_free(g, idx - g->main_width - 1)) {
                set_obj_idx(g, idx, SPACE);
                set_obj_idx(g, idx - 1, mn_stationary(obj));
            } else if (mn_is_round(obj2) && obj_x < g->main_width - 1 && mn_is_free(g, idx + 1) && mn_is_free(g, idx - g->main_width + 1)) {
                set_obj_idx(g, idx, SPACE);
                set_obj_idx(g, idx + 1, stat_type);
            } else {
                set_obj_idx(g, idx, stat_type);
            }
        }
    }
    g->diamonds_remaining = diamonds_count;
    /* the ENEMY loop of miner.cpp:299-305 never runs: no ENEMY entity is ever created */
}

/* ---- MazeGen: reference src/mazegen.cpp (Kruskal over std::set cell sets) ----
 * The std::set objects only carry connectivity, so they are restated as one label per cell; walls.erase keeps the
 * vector order, restated with an order-preserving removal. */
#define MG_MAX_DIM 33
typedef struct {
    int maze_dim, array_dim;
    int grid[MG_MAX_DIM * MG_MAX_DIM]; /* Grid<int>, index y*array_dim+x */
    int label[MG_MAX_DIM * MG_MAX_DIM]; /* cell_sets_idxs */
    int is_free[MG_MAX_DIM * MG_MAX_DIM]; /* free_cell_set membership */
    int free_cells[MG_MAX_DIM * MG_MAX_DIM];
    int num_free_cells;
} MazeGen;

static void mg_set_free_cell(MazeGen *m, int x, int y) { /* mazegen.cpp:26-34 */
    m->grid[(y + MAZE_OFFSET) * m->array_dim + x + MAZE_OFFSET] = SPACE;
    int cell = m->maze_dim * y + x;
    if (!m->is_free[cell]) {
        m->free_cells[m->num_free_cells] = cell;
        m->is_free[cell] = 1;
        m->num_free_cells += 1;
    }
}

static void mg_generate_maze(MazeGen *m, Rng *r) { /* mazegen.cpp:112-187 */
    int md = m->maze_dim, ad = m->array_dim;
    for (int i = 0; i < ad * ad; i++) m->grid[i] = WALL_OBJ;
    m->grid[MAZE_OFFSET * ad + MAZE_OFFSET] = 0;
    m->num_free_cells = 0;
    memset(m->is_free, 0, sizeof(m->is_free));
    for (int i = 0; i < md * md; i++) m->label[i] = i;
    static int wx1[1024], wy1[1024], wx2[1024], wy2[1024];
    int nw = 0;
    for (int i = 1; i < md; i += 2)
        for (int j = 0; j < md; j += 2)
            if (i > 0 && i < md - 1) { wx1[nw] = i - 1; wy1[nw] = j; wx2[nw] = i + 1; wy2[nw] = j; nw++; }
    for (int i = 0; i < md; i += 2)
        for (int j = 1; j < md; j += 2)
            if (j > 0 && j < md - 1) { wx1[nw] = i; wy1[nw] = j - 1; wx2[nw] = i; wy2[nw] = j + 1; nw++; }
    while (nw > 0) {
        int n = rng_randn(r, nw);
        int x1 = wx1[n], y1 = wy1[n], x2 = wx2[n], y2 = wy2[n];
        int s0_idx = m->label[md * y1 + x1];
        int s1_idx = m->label[md * y2 + x2];
        int x0 = (x1 + x2) / 2, y0 = (y1 + y2) / 2;
        int center = md * y0 + x0;
        int can_remove = (m->grid[(y0 + MAZE_OFFSET) * ad + x0 + MAZE_OFFSET] == WALL_OBJ) && (s0_idx != s1_idx);
        if (can_remove) {
            mg_set_free_cell(m, x1, y1);
            mg_set_free_cell(m, x0, y0);
            mg_set_free_cell(m, x2, y2);
            for (int i = 0; i < md * md; i++)
                if (m->label[i] == s0_idx) m->label[i] = s1_idx;
            m->label[center] = s1_idx;
        }
        for (int k = n; k < nw - 1; k++) { wx1[k] = wx1[k + 1]; wy1[k] = wy1[k + 1]; wx2[k] = wx2[k + 1]; wy2[k] = wy2[k + 1]; }
        nw--;
    }
}

static void mg_place_objects(MazeGen *m, Rng *r, int start_obj, int num_objs) { /* mazegen.cpp:292-306 */
    for (int j = 0; j < num_objs; j++) {
        int k = rng_randn(r, m->num_free_cells);
        while (m->free_cells[k] == -1 || m->free_cells[k] == 0) k = rng_randn(r, m->num_free_cells);
        int coin_cell = m->free_cells[k];
        m->free_cells[k] = -1;
        m->grid[(coin_cell / m->maze_dim + MAZE_OFFSET) * m->array_dim + coin_cell % m->maze_dim + MAZE_OFFSET] = start_obj + j;
    }
}

static int mg_get_obj(const MazeGen *m, int idx) { /* mazegen.cpp:36-46 */
    int x = idx % m->array_dim, y = idx / m->array_dim;
    if (x <= 0 || x >= m->array_dim - 1) return INVALID_OBJ;
    if (y <= 0 || y >= m->array_dim - 1) return INVALID_OBJ;
    return m->grid[y * m->array_dim + x];
}
static int mg_get_neighbors(const MazeGen *m, int idx, int type, int *out) { /* mazegen.cpp:48-66: order (-1,0) (0,-1) (0,1) (1,0) */
    int x = idx % m->array_dim, y = idx / m->array_dim, n = 0;
    static const int DX[4] = {-1, 0, 0, 1}, DY[4] = {0, -1, 1, 0};
    for (int k = 0; k < 4; k++) {
        int n_idx = (y + DY[k]) * m->array_dim + (x + DX[k]);
        if (mg_get_obj(m, n_idx) == type) out[n++] = n_idx;
    }
    return n;
}
/* std::set<int> restated as membership flags; iteration in ascending cell order == the set's order */
static int mg_expand_to_type(const MazeGen *m, const unsigned char *s0, unsigned char *s1, int type) { /* mazegen.cpp:68-99 */
    int nc = m->array_dim * m->array_dim;
    static unsigned char curr[MG_MAX_DIM * MG_MAX_DIM], next[MG_MAX_DIM * MG_MAX_DIM];
    memcpy(curr, s0, (size_t)nc);
    for (;;) {
        int any = 0;
        for (int i = 0; i < nc; i++) any |= curr[i];
        if (!any) break;
        memset(next, 0, (size_t)nc);
        for (int elem = 0; elem < nc; elem++) {
            if (!curr[elem]) continue;
            int tgt[4], adj[4];
            int nt = mg_get_neighbors(m, elem, type, tgt);
            int na = mg_get_neighbors(m, elem, SPACE, adj);
            for (int k = 0; k < na; k++) {
                int j = adj[k];
                if (!s0[j] && !s1[j]) {
                    next[j] = 1;
                    s1[j] = 1;
                }
            }
            if (nt > 0) return tgt[0];
        }
        memcpy(curr, next, (size_t)nc);
    }
    return -1;
}
static void mg_generate_maze_no_dead_ends(MazeGen *m, Rng *r) { /* mazegen.cpp:189-209 */
    mg_generate_maze(m, r);
    int nc = m->array_dim * m->array_dim, adj[4], wl[4];
    for (int i = 0; i < nc; i++) {
        if (mg_get_obj(m, i) == SPACE) {
            if (mg_get_neighbors(m, i, SPACE, adj) == 1) {
                int nw = mg_get_neighbors(m, i, WALL_OBJ, wl);
                if (nw > 0) m->grid[wl[rng_randn(r, nw)]] = SPACE;
            }
        }
    }
}
static void mg_generate_maze_with_doors(MazeGen *m, Rng *r, int num_doors) { /* mazegen.cpp:211-290 */
    mg_generate_maze(m, r);
    int nc = m->array_dim * m->array_dim;
    static int forks[MG_MAX_DIM * MG_MAX_DIM], chosen[MG_MAX_DIM * MG_MAX_DIM], space_cells[MG_MAX_DIM * MG_MAX_DIM];
    int nforks = 0, tmp[4];
    for (int i = 0; i < nc; i++)
        if (mg_get_obj(m, i) == SPACE && mg_get_neighbors(m, i, SPACE, tmp) > 2) forks[nforks++] = i;
    int nchosen = 0;
    if (num_doors > nforks) { /* RandGen::choose_n randgen.cpp:49-69 */
        for (int i = 0; i < nforks; i++) chosen[nchosen++] = forks[i];
    } else {
        int nrem = nforks;
        while (nchosen < num_doors) {
            int idx = rng_randn(r, nrem);
            chosen[nchosen++] = forks[idx];
            for (int k = idx; k < nrem - 1; k++) forks[k] = forks[k + 1];
            nrem--;
        }
    }
    num_doors = nchosen;
    for (int i = 0; i < nchosen; i++) m->grid[chosen[i]] = DOOR_OBJ;
    int agent_cell;
    {
        int ns = 0;
        for (int i = 0; i < nc; i++)
            if (mg_get_obj(m, i) == SPACE) space_cells[ns++] = i;
        if (ns <= 0) fatal("fassert elems.size() > 0 (randgen.cpp:44)");
        do {
            agent_cell = space_cells[rng_randn(r, ns)];
        } while (mg_get_neighbors(m, agent_cell, DOOR_OBJ, tmp) > 0);
        m->grid[agent_cell] = AGENT_OBJ;
    }
    static unsigned char s0[MG_MAX_DIM * MG_MAX_DIM], s1[MG_MAX_DIM * MG_MAX_DIM];
    memset(s0, 0, (size_t)nc);
    s0[agent_cell] = 1;
    for (int door_num = 0; door_num < num_doors + 1; door_num++) {
        memset(s1, 0, (size_t)nc);
        int found_door = -1;
        if (door_num < num_doors) {
            found_door = mg_expand_to_type(m, s0, s1, DOOR_OBJ);
            if (found_door < 0) fatal("grid.set_index(-1) (mazegen.cpp:262)");
            m->grid[found_door] = DOOR_OBJ + door_num + 1;
            for (int i = 0; i < nc; i++) s0[i] |= s1[i];
        }
        mg_expand_to_type(m, s0, s1, -999);
        int ns = 0;
        for (int i = 0; i < nc; i++)
            if (s1[i]) space_cells[ns++] = i;
        if (ns <= 0) fatal("fassert space_cells.size() > 0 (mazegen.cpp:275)");
        int key_cell = space_cells[rng_randn(r, ns)];
        m->grid[key_cell] = door_num == num_doors ? EXIT_OBJ : (KEY_OBJ + door_num + 1);
        for (int i = 0; i < nc; i++) s0[i] |= s1[i];
        if (found_door >= 0) s0[found_door] = 1;
    }
}

/* ---- level generation ---- */
static void choose_random_theme(Game *g, Ent *ent) { /* BAG:1038-1041 */
    ent->image_theme = rng_randn(&g->rand_gen, g->assets->type_num_themes[ent->image_type]);
}

/* asset_aspect_ratios[img_idx] is filled by initialize_asset_if_necessary from the image of the MASKED theme (BAG:82-86,114):
   with restrict_themes every theme of a type has the aspect ratio of theme 0 */
static int aspect_theme(const Game *g, const Ent *ent) {
    return (g->opt.restrict_themes && !hook_preserve_type_themes(g, ent->image_type)) ? 0 : ent->image_theme;
}
static void match_aspect_ratio(Game *g, Ent *ent) { /* BAG:1014-1023 (match_width = true), aspect from BAG:114 */
    if (!g->opt.use_generated_assets && g->assets->type_num_themes[ent->image_type] <= ent->image_theme) fatal("asset theme out of range");
    const Img *im = &g->assets->img[g->assets->type_theme_img[ent->image_type][aspect_theme(g, ent)]];
    float aspect = (float)(im->w * 1.0 / im->h);
    ent->ry = ent->rx / aspect;
}

static void match_aspect_ratio_h(Game *g, Ent *ent) { /* BAG:1014-1023 (match_width = false) */
    if (!g->opt.use_generated_assets && g->assets->type_num_themes[ent->image_type] <= ent->image_theme) fatal("asset theme out of range");
    const Img *im = &g->assets->img[g->assets->type_theme_img[ent->image_type][aspect_theme(g, ent)]];
    float aspect = (float)(im->w * 1.0 / im->h);
    ent->rx = ent->ry * aspect;
}

static float rand_pos(Game *g, float r, float min, float max) { /* BAG:1100-1108 */
    if (!(min <= max)) fatal("fassert min <= max (BAG:1101)");
    if (max - min <= 2 * r) return (max + min) / 2;
    float range = max - min;
    return (range - 2 * r) * rng_rand01(&g->rand_gen) + r + min;
}

static void face_direction(Ent *e, float dx, float dy, float rotation_offset) { /* entity.cpp:84-88; atan2f by overload */
    if (dx != 0 || dy != 0) e->rotation = -1 * atan2f(dy, dx) + rotation_offset;
}

static int has_any_collision(const Game *g, const Ent *e1, float margin) { /* BAG:1114-1124 */
    for (int i = g->n_ents - 1; i >= 0; i--) {
        const Ent *ent = &g->pool[g->ents[i]];
        if (!ent->avoids_collisions && has_collision(e1, ent, margin)) return 1;
    }
    return 0;
}
static void reposition(Game *g, Ent *ent, float x, float y, float w, float h, int check_collisions) { /* BAG:541-560 */
    float rx = ent->rx, ry = ent->ry;
    ent->x = rand_pos(g, rx, x, x + w);
    ent->y = rand_pos(g, ry, y, y + h);
    int count = 0;
    while ((has_agent_collision(g, ent) || (check_collisions && has_any_collision(g, ent, 0))) && (count < 100)) {
        ent->x = rand_pos(g, rx, x, x + w);
        ent->y = rand_pos(g, ry, y, y + h);
        count++;
    }
}
static Ent *spawn_entity_rxy(Game *g, float rx, float ry, int type, float x, float y, float w, float h, int check_collisions) { /* BAG:511-519 */
    int id = pool_alloc(g);
    Ent *ent = &g->pool[id];
    ent_init(ent, 0, 0, 0, 0, rx, ry, type);
    reposition(g, ent, x, y, w, h, check_collisions);
    if (g->n_ents >= MAX_ENTS) fatal("entity list overflow");
    g->ents[g->n_ents++] = id;
    return ent;
}
static void spawn_entities(Game *g, int num, float r, int type, float x, float y, float w, float h) { /* BAG:585-589 */
    for (int i = 0; i < num; i++) spawn_entity_rxy(g, r, r, type, x, y, w, h, 1);
}
static void fit_aspect_ratio(Game *g, Ent *ent) { /* BAG:1025-1036 */
    if (!g->opt.use_generated_assets && g->assets->type_num_themes[ent->image_type] <= ent->image_theme) fatal("asset theme out of range");
    const Img *im = &g->assets->img[g->assets->type_theme_img[ent->image_type][aspect_theme(g, ent)]];
    float ar = (float)(im->w * 1.0 / im->h);
    if (ar > 1) ent->ry = ent->rx / ar;
    else ent->rx = ent->ry * ar;
}

/* ---- RoomGenerator: reference src/roomgen.cpp (std::set<int> = membership flags, ascending iteration) ---- */
static int rg_to_grid_idx(const Game *g, int x, int y) { return grid_contains(g, x, y) ? y * g->grid_w + x : -2; } /* BAG:187-192 */
static int rg_get_obj_idx(const Game *g, int idx) { return (0 <= idx && idx < g->grid_w * g->grid_h) ? g->grid[idx] : g->out_of_bounds_object; } /* BAG:198-203 */
static void rg_update(Game *g) { /* roomgen.cpp:22-37; count_neighbors :3-20 (3x3 block incl. the cell) */
    static int next_cells[MAX_GRID];
    int n = g->grid_w * g->grid_h;
    for (int idx = 0; idx < n; idx++) {
        int x = idx % g->grid_w, y = idx / g->grid_w, cnt = 0;
        for (int i = -1; i <= 1; i++)
            for (int j = -1; j <= 1; j++) cnt += get_obj(g, x + i, y + j) == WALL_OBJ;
        next_cells[idx] = cnt >= 5 ? WALL_OBJ : SPACE;
    }
    for (int idx = 0; idx < n; idx++) g->grid[idx] = next_cells[idx];
}
static int rg_build_room(const Game *g, int idx, unsigned char *room) { /* roomgen.cpp:39-70; returns the room's size */
    static int queue[4 * MAX_GRID];
    int head = 0, tail = 0, size = 0;
    if (rg_get_obj_idx(g, idx) != SPACE) return 0;
    queue[tail++] = idx;
    while (head < tail) {
        int curr = queue[head++];
        if (rg_get_obj_idx(g, curr) != SPACE) continue;
        int x = curr % g->grid_w, y = curr / g->grid_w;
        static const int DI[4] = {-1, 0, 0, 1}, DJ[4] = {0, -1, 1, 0};
        for (int k = 0; k < 4; k++) {
            int next = rg_to_grid_idx(g, x + DI[k], y + DJ[k]);
            if (next >= 0 && !room[next] && g->grid[next] == SPACE) {
                queue[tail++] = next;
                room[next] = 1;
                size++;
            }
        }
    }
    return size;
}
static int rg_find_path(const Game *g, int src, int dst, int *path) { /* roomgen.cpp:72-126; returns the path length */
    static unsigned char covered[MAX_GRID];
    static int expanded[2 * MAX_GRID], parents[2 * MAX_GRID], tmp[2 * MAX_GRID];
    int n = g->grid_w * g->grid_h, ne = 0, np = 0;
    memset(covered, 0, (size_t)n);
    if (rg_get_obj_idx(g, src) != SPACE) return 0;
    expanded[ne] = src;
    parents[ne++] = -1;
    int search_idx = 0;
    while (search_idx < ne) {
        int curr = expanded[search_idx];
        if (curr == dst) break;
        int x = curr % g->grid_w, y = curr / g->grid_w;
        static const int DI[4] = {-1, 0, 0, 1}, DJ[4] = {0, -1, 1, 0};
        for (int k = 0; k < 4; k++) {
            int next = rg_to_grid_idx(g, x + DI[k], y + DJ[k]);
            if (next >= 0 && !covered[next] && g->grid[next] == SPACE) {
                expanded[ne] = next;
                parents[ne++] = search_idx;
                covered[next] = 1;
            }
        }
        search_idx++;
    }
    if (search_idx < ne && expanded[search_idx] == dst) {
        int nt = 0;
        while (search_idx >= 0) {
            tmp[nt++] = expanded[search_idx];
            search_idx = parents[search_idx];
        }
        for (int j = nt - 1; j >= 0; j--) path[np++] = tmp[j];
    }
    return np;
}
static void rg_expand_room(const Game *g, unsigned char *set, int n_loops) { /* roomgen.cpp:150-182 */
    static unsigned char curr[MAX_GRID], next[MAX_GRID];
    int n = g->grid_w * g->grid_h;
    memcpy(curr, set, (size_t)n);
    for (int loop = 0; loop < n_loops; loop++) {
        memset(next, 0, (size_t)n);
        for (int c = 0; c < n; c++) {
            if (!curr[c]) continue;
            if (g->grid[c] != SPACE) continue;
            int x = c % g->grid_w, y = c / g->grid_w;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++)
                    if (i != 0 || j != 0) {
                        int nx = rg_to_grid_idx(g, x + i, y + j);
                        if (nx >= 0 && !set[nx] && g->grid[nx] == SPACE) {
                            set[nx] = 1;
                            next[nx] = 1;
                        }
                    }
        }
        memcpy(curr, next, (size_t)n);
    }
}

/* ---- Jumper: jumper.cpp:180-396 ---- */
static int jp_is_space_on_ground(const Game *g, int x, int y) { /* jumper.cpp:180-187 */
    if (get_obj(g, x, y) != SPACE) return 0;
    if (get_obj(g, x, y + 1) != SPACE) return 0;
    int below_obj = get_obj(g, x, y - 1);
    return below_obj == JP_CAVEWALL || below_obj == g->out_of_bounds_object;
}
static int jp_is_left_wall(const Game *g, int x, int y) { return get_obj(g, x, y) == JP_CAVEWALL && get_obj(g, x + 1, y) == SPACE; }  /* :193-195 */
static int jp_is_right_wall(const Game *g, int x, int y) { return get_obj(g, x, y) == JP_CAVEWALL && get_obj(g, x - 1, y) == SPACE; } /* :197-199 */
static void jp_pre_reset(Game *g) { /* jumper.cpp:219-231: before BasicAbstractGame::game_reset */
    if (g->opt.distribution_mode == 0) {
        g->visibility = 12;
        g->compass_dim = 3;
    } else {
        g->visibility = 16;
        g->compass_dim = 2;
    }
    if (g->opt.distribution_mode == 10) g->timeout = 2000;
}
static void jp_game_reset(Game *g) { /* jumper.cpp:233-388 */
    static MazeGen mg;
    int n = g->grid_w * g->grid_h, mw = g->main_width;
    g->out_of_bounds_object = WALL_OBJ;
    g->wall_theme = rng_randn(&g->rand_gen, 4);
    g->jump_count = 0;
    g->jump_delta = 0;
    g->jump_time = 0;
    g->has_support = 0;
    g->facing_right = 1;
    int maze_dim = g->main_width / JP_MAZE_SCALE;
    mg.maze_dim = maze_dim;
    mg.array_dim = maze_dim + 2;
    mg_generate_maze_no_dead_ends(&mg, &g->rand_gen);
    for (int i = 0; i < n; i++) {
        int obj = mg.grid[((i / mw) / JP_MAZE_SCALE + 1) * mg.array_dim + (i % mw) / JP_MAZE_SCALE + 1];
        float prob = obj == WALL_OBJ ? (float).8 : (float).2;
        g->grid[i] = rng_rand01(&g->rand_gen) < prob ? WALL_OBJ : SPACE;
    }
    for (int it = 0; it < 2; it++) rg_update(g);
    for (int i = 0; i < g->main_width; i++) {
        set_obj(g, i, 0, JP_CAVEWALL);
        set_obj(g, i, g->main_height - 1, JP_CAVEWALL);
    }
    for (int i = 0; i < g->main_height; i++) {
        set_obj(g, 0, i, JP_CAVEWALL);
        set_obj(g, g->main_width - 1, i, JP_CAVEWALL);
    }
    static unsigned char best_room[MAX_GRID], all_rooms[MAX_GRID], next_room[MAX_GRID], wide_path[MAX_GRID];
    static int cells[MAX_GRID], goal_path[2 * MAX_GRID];
    {
        memset(all_rooms, 0, (size_t)n);
        memset(best_room, 0, (size_t)n);
        int best_size = -1;
        for (int i = 0; i < n; i++)
            if (g->grid[i] == SPACE && !all_rooms[i]) {
                memset(next_room, 0, (size_t)n);
                int sz = rg_build_room(g, i, next_room);
                for (int c = 0; c < n; c++) all_rooms[c] |= next_room[c];
                if (sz > best_size) {
                    best_size = sz;
                    memcpy(best_room, next_room, (size_t)n);
                }
            }
        if (best_size <= 0) fatal("fassert best_room.size() > 0 (jumper.cpp:275)");
    }
    int nc = 0;
    for (int i = 0; i < n; i++) g->grid[i] = JP_CAVEWALL;
    for (int i = 0; i < n; i++)
        if (best_room[i]) {
            g->grid[i] = SPACE;
            cells[nc++] = i;
        }
    int goal_cell = cells[rng_randn(&g->rand_gen, nc)];
    nc = 0;
    for (int i = 0; i < n; i++)
        if (jp_is_space_on_ground(g, i % mw, i / mw)) cells[nc++] = i;
    if (nc <= 0) fatal("fassert elems.size() > 0 (randgen.cpp:44)");
    int agent_cell = cells[rng_randn(&g->rand_gen, nc)];
    int npath = rg_find_path(g, agent_cell, goal_cell, goal_path);
    if (g->opt.distribution_mode != 10) {
        memset(wide_path, 0, (size_t)n);
        for (int k = 0; k < npath; k++) wide_path[goal_path[k]] = 1;
        rg_expand_room(g, wide_path, 4);
        for (int i = 0; i < n; i++) g->grid[i] = wide_path[i] ? SPACE : JP_CAVEWALL;
    }
    push_entity(g, (float)((goal_cell % mw) + .5), (float)((goal_cell / mw) + .5), 0, 0, (float).5, (float).5, JP_GOAL);
    g->goal = g->ents[g->n_ents - 1];
    float spike_prob = g->opt.distribution_mode == 10 ? 0 : (float).2;
    for (int i = 0; i < n; i++) {
        int x = i % mw, y = i / mw;
        if (jp_is_space_on_ground(g, x, y) && (jp_is_space_on_ground(g, x - 1, y) && jp_is_space_on_ground(g, x + 1, y))) {
            if (rng_rand01(&g->rand_gen) < spike_prob) set_obj(g, x, y, JP_SPIKE);
        }
    }
    for (int i = 0; i < n; i++) {
        int x = i % mw, y = i / mw;
        if (jp_is_left_wall(g, x, y) && jp_is_left_wall(g, x, y + 1) && jp_is_left_wall(g, x, y + 2)) set_obj(g, x, y + rng_randn(&g->rand_gen, 3), SPACE);
        if (jp_is_right_wall(g, x, y) && jp_is_right_wall(g, x, y + 1) && jp_is_right_wall(g, x, y + 2)) set_obj(g, x, y + rng_randn(&g->rand_gen, 3), SPACE);
    }
    Ent *agent = &g->pool[g->agent];
    agent->x = (float)((agent_cell % mw) + .5);
    agent->y = (agent_cell / mw) + agent->ry;
    for (int i = 0; i < n; i++)
        if (g->grid[i] == JP_SPIKE) {
            g->grid[i] = SPACE;
            float spike_ry = 0.4f, spike_rx = 0.23f;
            push_entity(g, (float)((i % mw) + .5), (i / mw) + spike_ry, 0, 0, spike_rx, spike_ry, JP_SPIKE);
        }
    for (int i = 0; i < n; i++) {
        int x = i % mw, y = i / mw;
        if (get_obj(g, x, y) == JP_CAVEWALL && get_obj(g, x, y + 1) == SPACE) set_obj(g, x, y, JP_CAVEWALL_TOP); /* is_top_wall :189-191 */
    }
    agent = &g->pool[g->agent];
    agent->rx = 0.254f;
    agent->ry = 0.4f;
    g->out_of_bounds_object = JP_CAVEWALL;
}

/* ---- CaveFlyer: caveflyer.cpp:131-262 ---- */
static void rng_simple_choose(Rng *r, int n, int k, int *chosen);
static void cf_game_reset(Game *g) {
    int n = g->grid_w * g->grid_h;
    g->out_of_bounds_object = WALL_OBJ;
    for (int i = 0; i < n; i++) g->grid[i] = rng_rand01(&g->rand_gen) < .5 ? WALL_OBJ : SPACE;
    for (int it = 0; it < 4; it++) rg_update(g);
    static unsigned char best_room[MAX_GRID], all_rooms[MAX_GRID], next_room[MAX_GRID], wide_path[MAX_GRID];
    static int free_cells[MAX_GRID], goal_path[2 * MAX_GRID], sel[MAX_GRID];
    { /* find_best_room roomgen.cpp:128-148 */
        memset(all_rooms, 0, (size_t)n);
        memset(best_room, 0, (size_t)n);
        int best_size = -1;
        for (int i = 0; i < n; i++)
            if (g->grid[i] == SPACE && !all_rooms[i]) {
                memset(next_room, 0, (size_t)n);
                int sz = rg_build_room(g, i, next_room);
                for (int c = 0; c < n; c++) all_rooms[c] |= next_room[c];
                if (sz > best_size) {
                    best_size = sz;
                    memcpy(best_room, next_room, (size_t)n);
                }
            }
        if (best_size <= 0) fatal("fassert best_room.size() > 0 (caveflyer.cpp:160)");
    }
    int nfree = 0;
    for (int i = 0; i < n; i++) {
        g->grid[i] = WALL_OBJ;
    }
    for (int i = 0; i < n; i++)
        if (best_room[i]) {
            g->grid[i] = SPACE;
            free_cells[nfree++] = i;
        }
    rng_simple_choose(&g->rand_gen, nfree, 2, sel);
    int agent_cell = free_cells[sel[0]], goal_cell = free_cells[sel[1]];
    Ent *agent = &g->pool[g->agent];
    agent->x = (float)((agent_cell % g->main_width) + .5);
    agent->y = (float)((agent_cell / g->main_width) + .5);
    Ent *goal = push_entity(g, (float)((goal_cell % g->main_width) + .5), (float)((goal_cell / g->main_width) + .5), 0, 0, (float).5, (float).5, CF_GOAL);
    goal->collides_with_entities = 1;
    int npath = rg_find_path(g, agent_cell, goal_cell, goal_path);
    if (g->opt.distribution_mode != 10) {
        memset(wide_path, 0, (size_t)n);
        for (int k = 0; k < npath; k++) wide_path[goal_path[k]] = 1;
        rg_expand_room(g, wide_path, 4);
        for (int i = 0; i < n; i++) g->grid[i] = wide_path[i] ? SPACE : WALL_OBJ;
    }
    for (int it = 0; it < 4; it++) {
        rg_update(g);
        for (int k = 0; k < npath; k++) g->grid[goal_path[k]] = SPACE;
    }
    for (int k = 0; k < npath; k++) g->grid[goal_path[k]] = CF_MARKER;
    nfree = 0;
    for (int i = 0; i < n; i++) {
        if (g->grid[i] == SPACE) free_cells[nfree++] = i;
        else if (g->grid[i] == WALL_OBJ) g->grid[i] = CF_CAVEWALL;
    }
    int chunk_size = nfree / 80;
    int num_objs = 3 * chunk_size;
    rng_simple_choose(&g->rand_gen, nfree, num_objs, sel);
    for (int i = 0; i < num_objs; i++) {
        int val = free_cells[sel[i]];
        float ex_ = (float)((val % g->main_width) + .5), ey_ = (float)((val / g->main_width) + .5);
        if (i < chunk_size) {
            Ent *e = push_entity(g, ex_, ey_, 0, 0, (float).5, (float).5, CF_OBSTACLE);
            e->collides_with_entities = 1;
        } else if (i < 2 * chunk_size) {
            Ent *e = push_entity(g, ex_, ey_, 0, 0, (float).5, (float).5, CF_TARGET);
            e->health = 5;
            e->collides_with_entities = 1;
        } else {
            Ent *e = push_entity(g, ex_, ey_, 0, 0, (float).5, (float).5, CF_ENEMY);
            double mag = .1 * rng_rand01(&g->rand_gen) + .1; /* left operand of the product draws first (checked against oracle/_ref) */
            int sgn = rng_randn(&g->rand_gen, 2) * 2 - 1;
            float vel = (float)(mag * sgn);
            if (rng_rand01(&g->rand_gen) < .5) e->vx = vel;
            else e->vy = vel;
            e->smart_step = 1;
            e->collides_with_entities = 1;
        }
    }
    for (int i = 0; i < n; i++)
        if (g->grid[i] == CF_MARKER) g->grid[i] = SPACE;
    g->out_of_bounds_object = CF_CAVEWALL;
    g->visibility = g->opt.distribution_mode == 0 ? 10.0f : 16.0f;
}

/* ---- Chaser: chaser.cpp:137-390 ---- */
static int to_grid_idx(const Game *g, int x, int y) { /* BAG:187-192 */
    if (!grid_contains(g, x, y)) return INVALID_IDX;
    return y * g->grid_w + x;
}
static void ch_spawn_egg(Game *g, int enemy_cell) { /* chaser.cpp:270-273 */
    Ent *egg = push_entity(g, (float)((enemy_cell % g->maze_dim) + .5), (float)((enemy_cell / g->maze_dim) + .5), 0, 0, (float).5, (float).5, CH_ENEMY_EGG);
    egg->health = (float)g->egg_timeout;
}
/* RandGen::simple_choose randgen.cpp:71-88 */
static void rng_simple_choose(Rng *r, int n, int k, int *chosen) {
    if (!(k <= n)) fatal("fassert k <= n (randgen.cpp:75)");
    for (int i = 0; i < k; i++) {
        int next = rng_randn(r, n), dup;
        do {
            dup = 0;
            for (int q = 0; q < i; q++) dup |= chosen[q] == next;
            if (dup) next = rng_randn(r, n);
        } while (dup);
        chosen[i] = next;
    }
}
static void ch_pre_reset(Game *g) { /* chaser.cpp:141-160: sets maze_dim before BasicAbstractGame::game_reset */
    int dm = g->opt.distribution_mode;
    if (dm == 0) { g->maze_dim = 11; g->total_enemies = 3; }
    else if (dm == 1) { g->maze_dim = 13; g->total_enemies = 3; }
    else if (dm == 2) { g->maze_dim = 19; g->total_enemies = 5; }
    else fatal("fassert(false) chaser.cpp:156");
}
static void ch_game_reset(Game *g) { /* chaser.cpp:137-264 */
    static MazeGen mg;
    int dm = g->opt.distribution_mode;
    int extra_orb_sign = dm == 0 ? 0 : (dm == 1 ? -1 : 1);
    g->center_agent = 0;
    Ent *agent = &g->pool[g->agent];
    agent->rx = (float).5;
    agent->ry = (float).5;
    g->eat_time = -1 * g->eat_timeout;
    fill_elem(g, 0, 0, g->main_width, g->main_height, CH_MAZE_WALL);
    mg.maze_dim = g->maze_dim;
    mg.array_dim = g->maze_dim + 2;
    mg_generate_maze_no_dead_ends(&mg, &g->rand_gen);
    static int quadrants[4][MAX_GRID], free_cells[MAX_GRID], sel[16];
    int nq[4] = {0, 0, 0, 0}, orbs_for_quadrant[4];
    int extra_quad = rng_randn(&g->rand_gen, 4);
    for (int i = 0; i < 4; i++) orbs_for_quadrant[i] = 1 + (i == extra_quad ? extra_orb_sign : 0);
    int md = g->maze_dim;
    for (int i = 0; i < md; i++)
        for (int j = 0; j < md; j++) {
            int obj = mg.grid[(j + MAZE_OFFSET) * mg.array_dim + i + MAZE_OFFSET];
            set_obj(g, i, j, obj == WALL_OBJ ? CH_MAZE_WALL : obj);
            if (obj == SPACE) {
                int idx = j * md + i;
                int quad_idx = (i >= md / 2.0 ? 1 : 0) * 2 + (j >= md / 2.0 ? 1 : 0);
                quadrants[quad_idx][nq[quad_idx]++] = idx;
            }
        }
    for (int i = 0; i < 4; i++) {
        rng_simple_choose(&g->rand_gen, nq[i], orbs_for_quadrant[i], sel);
        for (int k = 0; k < orbs_for_quadrant[i]; k++) {
            int cell = quadrants[i][sel[k]];
            push_entity(g, (float)((cell % g->main_width) + .5), (float)((cell / g->main_width) + .5), 0, 0, 0.4f, 0.4f, CH_LARGE_ORB); /* spawn_entity_at_idx BAG:577-583 */
            g->grid[cell] = CH_MARKER;
        }
    }
    int nfree = 0;
    for (int i = 0; i < g->grid_w * g->grid_h; i++)
        if (g->grid[i] == SPACE) free_cells[nfree++] = i;
    rng_simple_choose(&g->rand_gen, nfree, 1 + g->total_enemies, sel);
    int start = free_cells[sel[0]];
    agent = &g->pool[g->agent];
    agent->x = (float)((start % md) + .5);
    agent->y = (float)((start / md) + .5);
    for (int i = 0; i < g->total_enemies; i++) {
        int cell = free_cells[sel[i + 1]];
        g->grid[cell] = CH_MARKER;
        ch_spawn_egg(g, cell);
    }
    for (int k = 0; k < nfree; k++) g->grid[free_cells[k]] = CH_ORB;
    g->total_orbs = nfree;
    g->orbs_collected = 0;
    for (int i = 0; i < g->grid_w * g->grid_h; i++)
        if (g->grid[i] == CH_MARKER) g->grid[i] = SPACE;
}
static void ch_game_step_tail(Game *g) { /* chaser.cpp:301-390 */
    int num_enemies = 0;
    Ent *agent = &g->pool[g->agent];
    int can_eat = g->cur_time - g->eat_time < g->eat_timeout;
    float default_enemy_speed = (float).5;
    float vscale = can_eat ? (float)(default_enemy_speed * .5) : default_enemy_speed;
    int mw = g->main_width;
    for (int j = g->n_ents - 1; j >= 0; j--) {
        Ent *ent = &g->pool[g->ents[j]];
        if (ent->type == CH_ENEMY_EGG) {
            num_enemies++;
            ent->health -= 1;
            if (ent->health == 0) {
                ent->will_erase = 1;
                Ent *enemy = push_entity(g, ent->x, ent->y, 0, 0, (float).5, (float).5, CH_ENEMY); /* spawn_child BAG:225-231 */
                enemy->smart_step = 1;
            }
        } else if (ent->type == CH_ENEMY) {
            num_enemies++;
            float x = (float)(ent->x - .5);
            float y = (float)(ent->y - .5);
            int dist_scale = can_eat ? -1 : 1;
            int enemy_idx = to_grid_idx(g, (int)x, (int)y);
            int agent_idx = to_grid_idx(g, (int)agent->x, (int)agent->y);
            int is_at_junction = fabsf(x - roundf(x)) + fabsf(y - roundf(y)) < .01;
            int be_agressive = g->step_rand_int % 2 == 0;
            if ((ent->vx == 0 && ent->vy == 0) || is_at_junction) {
                int adj_elems[4], na = 0, space_neighbors[4], ns = 0;
                int prev_idx = to_grid_idx(g, (int)(x - sign_d(ent->vx)), (int)(y - sign_d(ent->vy)));
                { /* get_adjacent chaser.cpp:280-299: order (-1,0) (0,-1) (0,1) (1,0) */
                    int ex_ = enemy_idx % mw, ey_ = enemy_idx / mw;
                    static const int DI[4] = {-1, 0, 0, 1}, DJ[4] = {0, -1, 1, 0};
                    for (int k = 0; k < 4; k++) {
                        int nb = to_grid_idx(g, ex_ + DI[k], ey_ + DJ[k]);
                        if (nb != INVALID_IDX) adj_elems[na++] = nb;
                    }
                }
                int min_dist = 2 * mw;
                for (int k = 0; k < na; k++) {
                    int adj = adj_elems[k];
                    if (g->grid[adj] != CH_MAZE_WALL && adj != prev_idx) {
                        int md_ = (abs((adj % mw) - (agent_idx % mw)) + abs((adj / mw) - (agent_idx / mw))) * dist_scale;
                        if (be_agressive) {
                            if (md_ < min_dist) {
                                min_dist = md_;
                                ns = 0;
                                space_neighbors[ns++] = adj;
                            } else if (md_ == min_dist) {
                                space_neighbors[ns++] = adj;
                            }
                        } else {
                            space_neighbors[ns++] = adj;
                        }
                    }
                }
                if (ns == 0) fatal("chaser: modulo by zero (no free neighbour)");
                int neighbor = space_neighbors[(unsigned)g->step_rand_int % (unsigned)ns];
                int nx = neighbor % mw, ny = neighbor / mw;
                ent->vx = (nx - x) * vscale;
                ent->vy = (ny - y) * vscale;
            }
        }
    }
    if (num_enemies < g->total_enemies) {
        int nfree = 0;
        for (int i = 0; i < g->grid_w * g->grid_h; i++) nfree += g->grid[i] != CH_MAZE_WALL;
        int selected_idx = (int)((unsigned)g->step_rand_int % (unsigned)nfree), cell = -1;
        for (int i = 0, k = 0; i < g->grid_w * g->grid_h; i++)
            if (g->grid[i] != CH_MAZE_WALL) {
                if (k == selected_idx) { cell = i; break; }
                k++;
            }
        ch_spawn_egg(g, cell);
    }
    agent = &g->pool[g->agent];
    int agent_idx = (int)agent->y * g->main_width + (int)agent->x; /* get_agent_index BAG:176-178 */
    if (get_obj_idx(g, agent_idx) == CH_ORB) {
        set_obj_idx(g, agent_idx, SPACE);
        g->reward += CH_ORB_REWARD;
        g->orbs_collected += 1;
    }
    if (g->orbs_collected == g->total_orbs) {
        g->reward += 10.0f;
        g->level_complete = 1;
        g->done = 1;
    }
}

/* ---- Bossfight: bossfight.cpp:203-414 ---- */
static void reposition_agent(Game *g);
static void bf2_spawn_barriers(Game *g) { /* bossfight.cpp:318-336 */
    int num_barriers = rng_randn(&g->rand_gen, 3) + 1;
    for (int i = 0; i < num_barriers; i++) {
        const Ent *agent = &g->pool[g->agent];
        float barrier_r = 0.6f;
        float min_barrier_y = (float)(2 * agent->ry + barrier_r + .5);
        float ent_y = rng_rand01(&g->rand_gen) * (BF2_BOTTOM_MARGIN - min_barrier_y - barrier_r) + min_barrier_y;
        float ent_x = rng_rand01(&g->rand_gen) * (g->main_width - 2 * barrier_r) + barrier_r;
        Ent m;
        ent_init(&m, ent_x, ent_y, 0, 0, barrier_r, barrier_r, BF2_BARRIER);
        choose_random_theme(g, &m);
        match_aspect_ratio(g, &m);
        m.health = 3;
        m.collides_with_entities = 1;
        if (!has_any_collision(g, &m, 0)) {
            int id = pool_alloc(g);
            g->pool[id] = m;
            g->ents[g->n_ents++] = id;
        }
    }
}
static void bf2_game_reset(Game *g) { /* bossfight.cpp:203-262 */
    g->damaged_until_time = 0;
    g->last_fire_time = 0;
    g->boss_bullet_vel = g->opt.distribution_mode == 0 ? (float).5 : (float).75;
    int max_extra_invulnerable = g->opt.distribution_mode == 0 ? 1 : 3;
    g->center_agent = 0;
    Ent *boss = push_entity(g, (float)(g->main_width / 2), (float)(g->main_height / 2), 0, 0, BF2_BOSS_R, BF2_BOSS_R, BF2_BOSS);
    g->boss = g->ents[g->n_ents - 1];
    choose_random_theme(g, boss);
    match_aspect_ratio(g, boss);
    push_entity(g, boss->x, boss->y, 0, 0, (float)(1.2 * boss->rx), (float)(1.2 * boss->ry), BF2_SHIELDS);
    g->shields = g->ents[g->n_ents - 1];
    g->boss_vel_timeout = BF2_BOSS_VEL_TIMEOUT;
    g->base_fire_prob = 0.1f;
    g->round_health = rng_randn(&g->rand_gen, 9) + 1;
    g->num_rounds = 1 + rng_randn(&g->rand_gen, 5);
    g->invulnerable_duration = 2 + rng_randn(&g->rand_gen, max_extra_invulnerable + 1);
    g->vulnerable_duration = 500;
    boss->health = (float)(g->round_health * g->num_rounds);
    choose_random_theme(g, &g->pool[g->agent]);
    g->player_laser_theme = rng_randn(&g->rand_gen, 3);
    g->boss_laser_theme = rng_randn(&g->rand_gen, 3);
    g->n_attack_modes = 0;
    for (int i = 0; i < g->num_rounds; i++) g->attack_modes[g->n_attack_modes++] = rng_randn(&g->rand_gen, 4);
    g->round_num = 0;
    bf2_prepare_boss(g);
    Ent *agent = &g->pool[g->agent];
    agent->rx = (float).75;
    match_aspect_ratio(g, agent);
    reposition_agent(g);
    agent->y = agent->ry;
    g->barrier_vel = 0.1f;
    g->barriers_moves_right = rng_rand01(&g->rand_gen) > .5; /* randbool randgen.cpp:25-27 */
    g->barrier_spawn_prob = 0.025f;
    bf2_spawn_barriers(g);
}
static void bf2_boss_fire(Game *g, float bullet_r, float vel, float theta) { /* bossfight.cpp:264-269 */
    const Ent *boss = &g->pool[g->boss];
    Ent *nb = push_entity(g, boss->x, boss->y, (float)(vel * cos((double)theta)), (float)(vel * sin((double)theta)), bullet_r, bullet_r, BF2_ENEMY_BULLET);
    nb->image_theme = g->boss_laser_theme;
    nb->expire_time = 50;
    nb->vrot = PI_F / 8;
}
static void bf2_game_step_tail(Game *g) { /* bossfight.cpp:338-414 */
    Ent *boss = &g->pool[g->boss], *shields = &g->pool[g->shields], *agent = &g->pool[g->agent];
    shields->x = boss->x;
    shields->y = boss->y;
    g->rand_pct = rng_rand01(&g->rand_gen);
    g->rand_fire_pct = rng_rand01(&g->rand_gen);
    g->rand_pct_x = rng_rand01(&g->rand_gen);
    g->rand_pct_y = rng_rand01(&g->rand_gen);
    if (g->curr_vel_timeout <= 0) {
        float dest_x = g->rand_pct_x * (g->main_width - 2 * BF2_BOSS_R) + BF2_BOSS_R;
        float dest_y = g->rand_pct_y * (g->main_height - 2 * BF2_BOSS_R - BF2_BOTTOM_MARGIN) + BF2_BOSS_R + BF2_BOTTOM_MARGIN;
        boss->vx = (dest_x - boss->x) / g->boss_vel_timeout;
        boss->vy = (dest_y - boss->y) / g->boss_vel_timeout;
        g->curr_vel_timeout = g->boss_vel_timeout;
        if (g->time_to_swap > 0) {
            g->time_to_swap -= 1;
        } else {
            if (g->shields_are_up) g->time_to_swap = g->vulnerable_duration;
            else g->time_to_swap = g->invulnerable_duration;
            g->shields_are_up = !g->shields_are_up;
        }
    } else {
        g->curr_vel_timeout -= 1;
    }
    if (g->special_action == 1 && (g->cur_time - g->last_fire_time) >= 3) {
        Ent *nb = push_entity(g, agent->x, agent->y, 0, 1, (float).25, (float).25, BF2_PLAYER_BULLET);
        nb->image_theme = g->player_laser_theme;
        nb->collides_with_entities = 1;
        nb->expire_time = 25;
        g->last_fire_time = g->cur_time;
    }
    int ct = g->cur_time;
    float bv = g->boss_bullet_vel;
    if (g->damaged_until_time >= ct) { /* damaged_mode :309-315 */
        if (ct % 3 == 0) {
            float pos_x = boss->x + (2 * g->rand_pct_x - 1) * boss->rx;
            float pos_y = boss->y + (2 * g->rand_pct_y - 1) * boss->ry;
            push_entity(g, pos_x, pos_y, 0, 0, (float).75, (float).75, EXPLOSION);
        }
    } else if (g->shields_are_up) { /* active_attack :317-327 */
        if (g->attack_mode == 0) {
            if (ct % 8 == 0)
                for (int i = 0; i < 5; i++) bf2_boss_fire(g, (float).5, bv, (float)(PI_F * 1.5 + (i - 2) * PI_F / 8));
        } else if (g->attack_mode == 1) {
            int dt = 5;
            if (ct % dt == 0) {
                int k = ct / dt;
                k = abs(8 - (k % 16));
                for (int i = 0; i < 4; i++) bf2_boss_fire(g, (float).5, bv, (float)(PI_F * (1.25 + .5 * k / 8.0) + i * PI_F / 2));
            }
        } else if (g->attack_mode == 2) {
            if (ct % 10 == 0) {
                int num_bullets = 8;
                float offset = g->rand_pct * 2 * PI_F;
                for (int i = 0; i < num_bullets; i++) {
                    float theta = 2 * PI_F / num_bullets * i + offset;
                    bf2_boss_fire(g, (float).5, bv, theta);
                }
            }
        } else if (g->attack_mode == 3) {
            if (ct % 4 == 0) bf2_boss_fire(g, (float).5, bv, PI_F * (1 + g->rand_pct));
        }
    } else { /* passive_attack_mode :271-275 */
        if (g->rand_fire_pct < g->base_fire_prob) bf2_boss_fire(g, (float).5, bv, PI_F * (1 + g->rand_pct));
    }
    for (int i = g->n_ents - 1; i >= 0; i--) {
        Ent *ent = &g->pool[g->ents[i]];
        if (ent->type == BF2_ENEMY_BULLET) {
            float v_trail = (float).5;
            Ent *trail = push_entity(g, ent->x, ent->y, ent->vx * v_trail, ent->vy * v_trail, ent->rx, ent->ry, BF2_LASER_TRAIL);
            trail->alpha_decay = 0.7f;
            trail->image_type = BF2_ENEMY_BULLET;
            trail->image_theme = g->boss_laser_theme;
            trail->vrot = ent->vrot;
            trail->rotation = ent->rotation;
            trail->expire_time = 8;
        }
    }
}

/* ---- Dodgeball: dodgeball.cpp:157-448 ---- */
typedef struct { float x, y, w, h; } DbRoom; /* QRectF built from float expressions: the doubles hold float values */
static DbRoom db_rooms[64];
static int db_nrooms;
static void db_add_room(Game *g, float x, float y, float w, float h) { /* dodgeball.cpp:157-164 */
    float rw = w, rh = h;
    if ((rw >= g->min_dim || rh >= g->min_dim) && (rw >= g->hard_min_dim) && (rh >= g->hard_min_dim)) {
        if (db_nrooms >= 64) fatal("room list overflow");
        DbRoom r = {x, y, w, h};
        db_rooms[db_nrooms++] = r;
    }
}
static void db_split_room(Game *g, DbRoom room, float thickness) { /* dodgeball.cpp:166-225 */
    int will_split_width = rng_rand01(&g->rand_gen) < .5;
    int choice2 = rng_rand01(&g->rand_gen) < .5;
    if (room.w < g->min_dim) will_split_width = 0;
    if (room.h < g->min_dim) will_split_width = 1;
    float rx = room.x, ry = room.y, rw = room.w, rh = room.h;
    float gap = (float)(.25 * (rng_randn(&g->rand_gen, 3) + 1));
    float pct = 1 - gap;
    if (!will_split_width) {
        float wy, wh, remy;
        if (choice2) {
            wy = ry;
            remy = ry + pct * rh;
            wh = pct * rh;
        } else {
            wy = ry + (1 - pct) * rh;
            remy = ry;
            wh = pct * rh;
        }
        push_entity(g, rx + rw / 2, wy + wh / 2, 0, 0, thickness, wh / 2, DB_LAVA_WALL);
        float nextw = rw / 2 - thickness;
        db_add_room(g, rx, wy, nextw, wh);
        db_add_room(g, rx + rw / 2 + thickness, wy, nextw, wh);
        db_add_room(g, rx, remy, rw, rh - wh);
    } else {
        float wx, ww, remx;
        if (choice2) {
            wx = rx;
            remx = rx + pct * rw;
            ww = pct * rw;
        } else {
            wx = rx + (1 - pct) * rw;
            remx = rx;
            ww = pct * rw;
        }
        push_entity(g, wx + ww / 2, ry + rh / 2, 0, 0, ww / 2, thickness, DB_LAVA_WALL);
        float nexth = rh / 2 - thickness;
        db_add_room(g, wx, ry, ww, nexth);
        db_add_room(g, wx, ry + rh / 2 + thickness, ww, nexth);
        db_add_room(g, remx, ry, rw - ww, rh);
    }
}
static void db_choose_vel(Game *g, Ent *ent) { /* dodgeball.cpp:227-239 */
    float vel = DB_ENEMY_VEL * (rng_randn(&g->rand_gen, 2) * 2 - 1);
    if (rng_randn(&g->rand_gen, 2) == 0) {
        ent->vx = vel;
        ent->vy = 0;
    } else {
        ent->vy = vel;
        ent->vx = 0;
    }
    ent->spawn_time = rng_randn(&g->rand_gen, 50) + 25;
}
static void reposition_agent(Game *g);
static void db_game_reset(Game *g) { /* dodgeball.cpp:261-371 */
    g->center_agent = g->opt.distribution_mode == 10;
    g->last_fire_time = 0;
    db_nrooms = 0;
    DbRoom all = {0, 0, (float)g->main_width, (float)g->main_height};
    db_rooms[db_nrooms++] = all;
    int dm = g->opt.distribution_mode;
    Ent *agent = &g->pool[g->agent];
    float thickness = 0.3f, enemy_r = (float).5, exit_r = (float).75;
    g->ball_r = (float).25;
    g->ball_vscale = (float).25;
    int num_iterations = 0, max_extra_enemies = 3;
    if (dm == 0) {
        num_iterations = 2;
        thickness *= 2; enemy_r *= 2; g->ball_r *= 2; g->ball_vscale *= 2;
        g->maxspeed = (float).75;
        agent->rx = 1; agent->ry = 1;
        exit_r *= 2;
    } else if (dm == 1) {
        num_iterations = 4;
        thickness = (float)(thickness * 1.5); enemy_r = (float)(enemy_r * 1.5); g->ball_r = (float)(g->ball_r * 1.5); g->ball_vscale = (float)(g->ball_vscale * 1.5);
        g->maxspeed = (float).5;
        agent->rx = (float).75; agent->ry = (float).75;
    } else if (dm == 2) {
        num_iterations = 8;
        g->maxspeed = (float).25;
    } else if (dm == 10) {
        num_iterations = 16;
        thickness = (float)(thickness * 1.5); enemy_r = (float)(enemy_r * 1.5); g->ball_r = (float)(g->ball_r * 1.5); g->ball_vscale = (float)(g->ball_vscale * 1.5);
        g->maxspeed = (float).5;
        agent->rx = (float).75; agent->ry = (float).75;
        max_extra_enemies = 16;
    } else {
        fatal("fassert(false) dodgeball.cpp:310");
    }
    g->hard_min_dim = (float)(4 * agent->rx + 2 * thickness + .5);
    g->min_dim = (float)(agent->rx * 8 + .5);
    for (int it = 0; it < num_iterations; it++) {
        if (db_nrooms == 0) break;
        int idx = rng_randn(&g->rand_gen, db_nrooms);
        DbRoom room = db_rooms[idx];
        for (int k = idx; k < db_nrooms - 1; k++) db_rooms[k] = db_rooms[k + 1];
        db_nrooms--;
        db_split_room(g, room, thickness);
    }
    float border_r = 0;
    float doorlen = 2 * exit_r;
    int exit_wall_choice = rng_randn(&g->rand_gen, 4);
    float mw = (float)g->main_width, mh = (float)g->main_height;
    if (exit_wall_choice == 0) spawn_entity_rxy(g, doorlen / 2, exit_r, DB_DOOR, 2 * border_r, 2 * border_r, mw - 4 * border_r, 2 * exit_r, 1);
    else if (exit_wall_choice == 1) spawn_entity_rxy(g, doorlen / 2, exit_r, DB_DOOR, 2 * border_r, mh - 2 * border_r - 2 * exit_r, mw - 4 * border_r, 2 * exit_r, 1);
    else if (exit_wall_choice == 2) spawn_entity_rxy(g, exit_r, doorlen / 2, DB_DOOR, 2 * border_r, 2 * border_r, 2 * exit_r, mh - 4 * border_r, 1);
    else if (exit_wall_choice == 3) spawn_entity_rxy(g, exit_r, doorlen / 2, DB_DOOR, mw - 2 * border_r - 2 * exit_r, 2 * border_r, 2 * exit_r, mh - 4 * border_r, 1);
    reposition_agent(g);
    g->num_enemies = rng_randn(&g->rand_gen, max_extra_enemies + 1) + 3;
    spawn_entities(g, g->num_enemies, enemy_r, DB_ENEMY, 0, 0, mw, mh);
    int enemy_theme = rng_randn(&g->rand_gen, 7);
    for (int k = 0; k < g->n_ents; k++) {
        Ent *ent = &g->pool[g->ents[k]];
        if (ent->type == DB_ENEMY) {
            ent->image_theme = enemy_theme;
            ent->health = 1;
            ent->spawn_time = 0;
            ent->fire_time = 10;
            ent->collides_with_entities = 1;
            ent->smart_step = 1;
            db_choose_vel(g, ent);
            face_direction(ent, ent->vx, ent->vy, 0);
        } else if (ent->type == DB_LAVA_WALL) {
            ent->collides_with_entities = 1;
        }
    }
    face_direction(&g->pool[g->agent], 1, 0, 0);
}
static void db_fire_ball(Game *g, Ent *ent, float vx, float vy) { /* dodgeball.cpp:373-378 */
    Ent *nb = push_entity(g, ent->x, ent->y, vx * g->ball_vscale, vy * g->ball_vscale, g->ball_r, g->ball_r, DB_ENEMY_BALL);
    ent->fire_time = g->cur_time + rng_randn(&g->rand_gen, 4);
    nb->vrot = DB_BALL_V_ROT;
    nb->expire_time = 50;
}
static void db_game_step_tail(Game *g) { /* dodgeball.cpp:380-448 */
    Ent *agent = &g->pool[g->agent];
    float vx = (float)(g->last_move_action / 3 - 1);
    float vy = (float)(g->last_move_action % 3 - 1);
    face_direction(agent, vx, vy, 0);
    if (g->special_action == 1 && (g->cur_time - g->last_fire_time) >= 7) {
        Ent *nb = push_entity(g, agent->x, agent->y, vx * g->ball_vscale, vy * g->ball_vscale, g->ball_r, g->ball_r, DB_PLAYER_BALL);
        nb->collides_with_entities = 1;
        nb->expire_time = 50;
        nb->vrot = DB_BALL_V_ROT;
        g->last_fire_time = g->cur_time;
    }
    g->num_enemies = 0;
    for (int i = g->n_ents - 1; i >= 0; i--) {
        Ent *ent = &g->pool[g->ents[i]];
        if (ent->type == DB_ENEMY) {
            g->num_enemies++;
            if (ent->spawn_time == 0) db_choose_vel(g, ent);
            else ent->spawn_time -= 1;
            int can_fire = (g->cur_time - ent->fire_time) >= g->enemy_fire_delay;
            if (can_fire) {
                float dx = ent->x - agent->x;
                float dy = ent->y - agent->y;
                float bvelx = (float)(ent->x < agent->x ? 1 : -1);
                float bvely = (float)(ent->y < agent->y ? 1 : -1);
                if (fabs((double)dx) < 1) {
                    db_fire_ball(g, ent, 0, bvely);
                    ent->vx = 0;
                    ent->vy = bvely * DB_ENEMY_VEL;
                } else if (fabs((double)dy) < 1) {
                    db_fire_ball(g, ent, bvelx, 0);
                    ent->vx = bvelx * DB_ENEMY_VEL;
                    ent->vy = 0;
                }
            }
            face_direction(ent, ent->vx, ent->vy, 0);
        } else if (ent->type == DB_PLAYER_BALL || ent->type == DB_ENEMY_BALL) {
            if (ent->x < ent->rx || ent->x > (g->main_width - ent->rx)) ent->will_erase = 1;
            else if (ent->y < ent->ry || ent->y > (g->main_height - ent->ry)) ent->will_erase = 1;
        }
    }
    erase_if_needed(g);
}

/* ---- Ninja: ninja.cpp:179-316 ---- */
static void nj_fill_ground_block(Game *g, int x, int y, int dx, int dy) { /* fill_block_top ninja.cpp:179-188 with fill == top */
    if (dy <= 0) return;
    fill_elem(g, x, y, dx, dy - 1, NJ_WALL_MID);
    fill_elem(g, x, y + dy - 1, dx, 1, NJ_WALL_MID);
}
static void nj_generate_coin_to_the_right(Game *g, int difficulty) { /* ninja.cpp:197-285 */
    int min_gap = difficulty - 1;
    int min_plat_w = 1;
    int inc_dy = 4;
    if (g->opt.distribution_mode == 0) {
        min_gap -= 1;
        if (min_gap < 0) min_gap = 0;
        min_plat_w = 3;
        inc_dy = 2;
    }
    float bomb_prob = (float)(.25 * (difficulty - 1));
    int max_gap_inc = difficulty == 1 ? 1 : 2;
    int num_sections = rng_randn(&g->rand_gen, difficulty) + difficulty;
    int start_x = 5;
    int curr_x = start_x;
    int curr_y = g->main_height / 2;
    int min_y = curr_y;
    int w = g->main_width;
    float _max_dy = g->max_jump * g->max_jump / (2 * g->gravity);
    int max_dy = (int)(_max_dy - .5);
    int prev_x, prev_y;
    nj_fill_ground_block(g, 0, 0, start_x, curr_y);
    fill_elem(g, 0, curr_y + 8, start_x, g->main_height - curr_y - 8, NJ_WALL_MID);
    for (int i = 0; i < num_sections; i++) {
        prev_x = curr_x;
        prev_y = curr_y;
        int num_edges = rng_randn(&g->rand_gen, 2) + 1;
        int max_y = -1;
        int last_edge_y = -1;
        for (int j = 0; j < num_edges; j++) {
            curr_x = prev_x + j;
            if (curr_x + 15 >= w) break;
            curr_y = prev_y;
            int dy = rng_randn(&g->rand_gen, inc_dy) + 1 + (int)(difficulty / 3);
            if (dy > max_dy) dy = max_dy;
            if (curr_y >= g->main_height - 15) dy *= -1;
            else if (curr_y >= 5 && rng_rand01(&g->rand_gen) < .4) dy *= -1;
            curr_y += dy;
            if (curr_y < 3) curr_y = 3;
            if (abs(curr_y - last_edge_y) <= 1) curr_y = last_edge_y + 2;
            int dx = min_plat_w + rng_randn(&g->rand_gen, 3);
            nj_fill_ground_block(g, curr_x, curr_y - 1, dx, 1);
            curr_x += dx;
            curr_x += min_gap + rng_randn(&g->rand_gen, max_gap_inc + 1);
            if (curr_y > max_y) max_y = curr_y;
            if (curr_y < min_y) min_y = curr_y;
            last_edge_y = curr_y;
        }
        if (rng_rand01(&g->rand_gen) < bomb_prob) set_obj(g, rng_randn(&g->rand_gen, curr_x - prev_x + 1) + prev_x, max_y + 2, NJ_BOMB);
        int ceiling_height = 11;
        int ceiling_start = max_y - 1 + ceiling_height;
        nj_fill_ground_block(g, prev_x, ceiling_start, curr_x - prev_x, g->main_height - ceiling_start);
    }
    Ent *ent = push_entity(g, (float)(curr_x + .5), (float)(curr_y + .5), 0, 0, (float).5, (float).5, NJ_GOAL);
    choose_random_theme(g, ent);
    nj_fill_ground_block(g, curr_x, curr_y - 1, 1, 1);
    fill_elem(g, curr_x, curr_y + 6, 1, g->main_height - curr_y - 6, NJ_WALL_MID);
    int fire_y = min_y - 2;
    if (fire_y < 1) fire_y = 1;
    nj_fill_ground_block(g, start_x, 0, g->main_width - start_x, fire_y);
    fill_elem(g, start_x, fire_y, g->main_width - start_x, 1, NJ_FIRE);
    fill_elem(g, curr_x + 1, 0, g->main_width - curr_x - 1, g->main_height, NJ_WALL_MID);
}
static void nj_game_reset(Game *g) { /* ninja.cpp:287-316 */
    Ent *agent = &g->pool[g->agent];
    g->gravity = 0.2f;
    g->max_jump = 1.5;
    g->air_control = 0.15f;
    g->maxspeed = (float).5;
    g->has_support = 0;
    g->facing_right = 1;
    g->jump_charge = 0;
    g->jump_charge_inc = (float).25;
    g->visibility = 16;
    agent->rx = (float).5;
    agent->ry = (float).5;
    agent->x = 1 + agent->rx;
    agent->y = g->main_height / 2 + agent->ry;
    if (g->opt.distribution_mode == 0) {
        g->max_jump = (float)1.25;
        g->jump_charge_inc = 1;
        g->visibility = 10;
    }
    int max_difficulty = 3;
    int difficulty = rng_randn(&g->rand_gen, max_difficulty) + 1;
    g->last_fire_time = 0;
    g->wall_theme = rng_randn(&g->rand_gen, 3);
    fill_elem(g, 0, 0, g->main_width, 1, NJ_WALL_MID); /* init_floor_and_walls ninja.cpp:190-195 */
    fill_elem(g, 0, 0, 1, g->main_height, NJ_WALL_MID);
    fill_elem(g, g->main_width - 1, 0, 1, g->main_height, NJ_WALL_MID);
    fill_elem(g, 0, g->main_height - 1, g->main_width, 1, NJ_WALL_MID);
    nj_generate_coin_to_the_right(g, difficulty);
}

/* ---- Heist: heist.cpp:112-194 ---- */
static Ent *spawn_entity_rxy(Game *g, float rx, float ry, int type, float x, float y, float w, float h, int check_collisions);
static void hs_game_reset(Game *g) {
    static MazeGen mg;
    int min_maze_dim = 5;
    int max_diff = (g->world_dim - min_maze_dim) / 2;
    int difficulty = rng_randn(&g->rand_gen, max_diff + 1);
    g->center_agent = g->opt.distribution_mode == 10;
    if (g->opt.distribution_mode == 10) g->num_keys = rng_randn(&g->rand_gen, 4);
    else g->num_keys = difficulty + rng_randn(&g->rand_gen, 2);
    if (g->num_keys > 3) g->num_keys = 3;
    for (int i = 0; i < 4; i++) g->has_keys[i] = 0;
    int maze_dim = difficulty * 2 + min_maze_dim;
    float maze_scale = (float)(g->main_height / (g->world_dim * 1.0));
    Ent *agent = &g->pool[g->agent];
    agent->rx = (float)(.375 * maze_scale);
    agent->ry = (float)(.375 * maze_scale);
    float r_ent = maze_scale / 2;
    mg.maze_dim = maze_dim;
    mg.array_dim = maze_dim + 2;
    mg_generate_maze_with_doors(&mg, &g->rand_gen, g->num_keys);
    agent->x = -1;
    agent->y = -1;
    int off_x = rng_randn(&g->rand_gen, g->world_dim - maze_dim + 1);
    int off_y = rng_randn(&g->rand_gen, g->world_dim - maze_dim + 1);
    for (int i = 0; i < g->grid_w * g->grid_h; i++) g->grid[i] = WALL_OBJ;
    for (int i = 0; i < maze_dim; i++) {
        for (int j = 0; j < maze_dim; j++) {
            int x = off_x + i, y = off_y + j;
            int obj = mg.grid[(j + MAZE_OFFSET) * mg.array_dim + i + MAZE_OFFSET];
            float obj_x = (float)((x + .5) * maze_scale);
            float obj_y = (float)((y + .5) * maze_scale);
            if (obj != WALL_OBJ) set_obj(g, x, y, SPACE);
            if (obj >= KEY_OBJ) {
                float r = (float)(.375 * maze_scale);
                Ent *ent = spawn_entity_rxy(g, r, r, HS_KEY, maze_scale * x, maze_scale * y, maze_scale, maze_scale, 1);
                ent->image_theme = obj - KEY_OBJ - 1;
                match_aspect_ratio(g, ent);
            } else if (obj >= DOOR_OBJ) {
                Ent *ent = push_entity(g, obj_x, obj_y, 0, 0, r_ent, r_ent, HS_LOCKED_DOOR);
                ent->image_theme = obj - DOOR_OBJ - 1;
            } else if (obj == EXIT_OBJ) {
                float r = (float)(.375 * maze_scale);
                Ent *ent = spawn_entity_rxy(g, r, r, HS_EXIT, maze_scale * x, maze_scale * y, maze_scale, maze_scale, 1);
                match_aspect_ratio(g, ent);
            } else if (obj == AGENT_OBJ) {
                g->pool[g->agent].x = obj_x;
                g->pool[g->agent].y = obj_y;
            }
        }
    }
    float ring_key_r = 0.03f;
    for (int i = 0; i < g->num_keys; i++) {
        Ent *ent = push_entity(g, (float)(1 - ring_key_r * (2 * i + 1.25)), (float)(ring_key_r * .75), 0, 0, ring_key_r, ring_key_r, HS_KEY_ON_RING);
        ent->image_theme = i;
        ent->image_type = HS_KEY;
        ent->rotation = PI_F / 2;
        ent->render_z = 1;
        ent->use_abs_coords = 1;
        match_aspect_ratio(g, ent);
    }
}

/* ---- Plunder: plunder.cpp:117-184 ---- */
static void reposition_agent(Game *g) { /* BAG:521-539 */
    Ent *agent = &g->pool[g->agent];
    int count = 0, coll;
    do {
        agent->x = rng_rand01(&g->rand_gen) * (g->main_width - 2 * agent->rx) + agent->rx;
        agent->y = rng_rand01(&g->rand_gen) * (g->main_height - 2 * agent->ry) + agent->ry;
        count++;
        coll = 0;
        for (int k = 0; k < g->n_ents && !coll; k++) coll = has_agent_collision(g, &g->pool[g->ents[k]]);
    } while (coll && (count < 100));
}
static void pl_game_reset(Game *g) {
    Ent *agent = &g->pool[g->agent];
    agent->image_type = PL_SHIP;
    g->juice_left = 1;
    g->targets_hit = 0;
    g->target_quota = 20;
    g->spawn_prob = 0.06f;
    g->r_scale = g->opt.distribution_mode == 0 ? 1.5f : 1.0f;
    int num_total_ship_types = 6;
    g->num_lanes = 5;
    { /* RandGen::choose_n randgen.cpp:49-69 with n == elems.size() */
        int rem[6], nrem = 6, nch = 0;
        for (int i = 0; i < 6; i++) rem[i] = i;
        while (nch < num_total_ship_types) {
            int idx = rng_randn(&g->rand_gen, nrem);
            g->image_permutation[nch++] = rem[idx];
            for (int k = idx; k < nrem - 1; k++) rem[k] = rem[k + 1];
            nrem--;
        }
    }
    g->num_current_ship_types = 2;
    for (int i = 0; i < num_total_ship_types; i++) g->target_bools[i] = 0;
    for (int i = 0; i < g->num_current_ship_types / 2; i++) g->target_bools[g->image_permutation[i]] = 1;
    for (int i = 0; i < g->num_lanes; i++) {
        g->lane_directions[i] = rng_rand01(&g->rand_gen) < .5;
        g->lane_vels[i] = (float)(.15 + .1 * rng_rand01(&g->rand_gen));
    }
    int num_panels = g->opt.distribution_mode == 0 ? 0 : rng_randn(&g->rand_gen, 4);
    float panel_width = 1.2f;
    for (int i = 0; i < num_panels; i++)
        spawn_entity_rxy(g, panel_width, (float).5, PL_PANEL, 0, (float)(.25 * g->main_height), (float)g->main_width, (float)(.25 * g->main_height), 1);
    float key_scale = 1.5;
    g->legend_r = 2;
    push_entity(g, g->legend_r, g->legend_r, 0, 0, g->legend_r, g->legend_r, PL_TARGET_BACKGROUND);
    float lr = g->r_scale * key_scale;
    Ent *ent = push_entity(g, g->legend_r, g->legend_r, 0, 0, lr, lr, PL_TARGET_LEGEND);
    ent->image_theme = g->image_permutation[0];
    ent->image_type = PL_SHIP;
    match_aspect_ratio(g, ent);
    ent->rotation = PI_F / 2;
    g->last_fire_time = 0;
    g->center_agent = 0;
    agent = &g->pool[g->agent];
    agent->rx = g->r_scale;
    agent->rotation = -1 * PI_F / 2;
    agent->image_theme = g->image_permutation[rng_randn(&g->rand_gen, g->num_current_ship_types / 2) + g->num_current_ship_types / 2];
    match_aspect_ratio(g, agent);
    reposition_agent(g);
    agent->y = 1 + agent->ry;
    g->min_agent_x = 2 * g->legend_r + agent->rx;
    if (agent->x < g->min_agent_x) agent->x = g->min_agent_x;
}

/* ---- Leaper: leaper.cpp:100-206 ---- */
static void lp_spawn_entities(Game *g) { /* leaper.cpp:177-206 */
    for (int lane = 0; lane < g->n_road_lanes; lane++) {
        float speed = g->road_lane_speeds[lane];
        float spawn_prob = (float)(fabs((double)speed) / 6.0);
        if (rng_rand01(&g->rand_gen) < spawn_prob) {
            float x = speed > 0 ? (-1 * LP_MONSTER_RADIUS) : (g->main_width + LP_MONSTER_RADIUS);
            Ent m;
            ent_init(&m, x, (float)(g->bottom_road_y + lane + 0.5), speed, 0, 2 * LP_MONSTER_RADIUS, LP_MONSTER_RADIUS, LP_CAR);
            choose_random_theme(g, &m);
            if (speed < 0) m.rotation = PI_F;
            if (!has_any_collision(g, &m, 0)) {
                int id = pool_alloc(g);
                g->pool[id] = m;
                g->ents[g->n_ents++] = id;
            }
        }
    }
    for (int lane = 0; lane < g->n_water_lanes; lane++) {
        float speed = g->water_lane_speeds[lane];
        float spawn_prob = (float)(fabs((double)speed) / 2.0);
        if (rng_rand01(&g->rand_gen) < spawn_prob) {
            float x = speed > 0 ? (-1 * LP_LOG_RADIUS) : (g->main_width + LP_LOG_RADIUS);
            Ent m;
            ent_init(&m, x, (float)(g->bottom_water_y + lane + 0.5), speed, 0, LP_LOG_RADIUS, LP_LOG_RADIUS, LP_LOG);
            if (!has_any_collision(g, &m, 0)) {
                int id = pool_alloc(g);
                g->pool[id] = m;
                g->ents[g->n_ents++] = id;
            }
        }
    }
}
static float lp_rand_sign(Game *g) { return rng_rand01(&g->rand_gen) < 0.5 ? 1.0f : -1.0f; } /* leaper.cpp:95-101 */
static float rng_randrange(Rng *r, float low, float high) { return rng_rand01(r) * (high - low) + low; } /* randgen.cpp:29-31 */
static int lp_choose_extra_space(Game *g) { return g->opt.distribution_mode == 0 ? 0 : rng_randn(&g->rand_gen, 2); } /* leaper.cpp:118-120 */
static void lp_game_reset(Game *g) { /* leaper.cpp:122-175 */
    g->center_agent = 0;
    Ent *agent = &g->pool[g->agent];
    agent->y = agent->ry;
    float min_car_speed = 0.05f, max_car_speed = 0.2f, min_log_speed = 0.05f, max_log_speed = 0.1f;
    if (g->opt.distribution_mode == 0) {
        min_car_speed = 0.03f; max_car_speed = 0.12f; min_log_speed = 0.025f; max_log_speed = 0.075f;
    } else if (g->opt.distribution_mode == 2) {
        min_car_speed = 0.1f; max_car_speed = 0.3f; min_log_speed = 0.1f; max_log_speed = 0.2f;
    }
    g->bottom_road_y = lp_choose_extra_space(g) + 1;
    int max_diff = g->opt.distribution_mode == 0 ? 3 : 4;
    int difficulty = rng_randn(&g->rand_gen, max_diff + 1);
    int extra_lane_option = g->opt.distribution_mode == 0 ? 0 : rng_randn(&g->rand_gen, 4);
    int num_road_lanes = difficulty + (extra_lane_option == 2 ? 1 : 0);
    g->n_road_lanes = 0;
    for (int lane = 0; lane < num_road_lanes; lane++) {
        /* rand_sign() * randrange(): both operands draw; the reference build evaluates the left operand first (checked against oracle/_ref) */
        float sgn = lp_rand_sign(g);
        float mag = rng_randrange(&g->rand_gen, min_car_speed, max_car_speed);
        g->road_lane_speeds[g->n_road_lanes++] = sgn * mag;
        fill_elem(g, 0, g->bottom_road_y + lane, g->main_width, 1, LP_ROAD);
    }
    g->bottom_water_y = g->bottom_road_y + num_road_lanes + lp_choose_extra_space(g) + 1;
    g->n_water_lanes = 0;
    int num_water_lanes = difficulty + (extra_lane_option == 3 ? 1 : 0);
    int curr_sign = (int)lp_rand_sign(g);
    for (int lane = 0; lane < num_water_lanes; lane++) {
        g->water_lane_speeds[g->n_water_lanes++] = curr_sign * rng_randrange(&g->rand_gen, min_log_speed, max_log_speed);
        curr_sign *= -1;
        fill_elem(g, 0, g->bottom_water_y + lane, g->main_width, 1, LP_WATER);
    }
    g->goal_y = g->bottom_water_y + num_water_lanes + 1;
    float lim = g->main_width / (min_car_speed < min_log_speed ? min_car_speed : min_log_speed);
    for (int i = 0; i < lim; i++) {
        lp_spawn_entities(g);
        step_entities(g);
    }
    push_entity(g, (float)(g->main_width / 2.0), (float)(g->goal_y - .5), 0, 0, (float)(g->main_width / 2.0), (float).5, LP_FINISH_LINE);
}

/* ---- FruitBot: fruitbot.cpp:168-249 ---- */
static void fb_add_walls(Game *g, float ry, int use_door, float min_pct) { /* fruitbot.cpp:168-201 */
    float rw = (float)g->main_width;
    float wall_ry = 0.3f;
    float lock_rx = (float).25;
    float lock_ry = 0.45f;
    float pct = (float)(min_pct + .2 * rng_rand01(&g->rand_gen));
    if (use_door) {
        pct += 0.1f;
        float lock_pct_w = 2 * lock_rx / g->main_width;
        float door_pct_w = (wall_ry * 2 * FB_DOOR_ASPECT_RATIO) / g->main_width;
        int num_doors = (int)ceil((double)((pct - 2 * lock_pct_w) / door_pct_w));
        pct = 2 * lock_pct_w + door_pct_w * num_doors;
    }
    float gapw = pct * rw;
    float w1 = rng_rand01(&g->rand_gen) * (rw - gapw);
    float w2 = rw - w1 - gapw;
    push_entity(g, w1 / 2, ry, 0, 0, w1 / 2, wall_ry, FB_BARRIER);
    push_entity(g, rw - w2 / 2, ry, 0, 0, w2 / 2, wall_ry, FB_BARRIER);
    if (use_door) {
        int is_on_right = rng_randn(&g->rand_gen, 2);
        float lock_x = w1 + lock_rx + is_on_right * (gapw - 2 * lock_rx);
        float door_x = w1 + gapw / 2 - (is_on_right * 2 - 1) * lock_rx;
        push_entity(g, door_x, ry, 0, 0, gapw / 2 - lock_rx, wall_ry, FB_LOCKED_DOOR);
        push_entity(g, lock_x, ry - lock_ry + wall_ry, 0, 0, lock_rx, lock_ry, FB_LOCK);
    }
}
static void fb_game_reset(Game *g) { /* fruitbot.cpp:203-249 */
    g->last_fire_time = 0;
    int min_sep = 4, num_walls = 10, object_group_size = 6, buf_h = 4;
    float door_prob = (float).125;
    float min_pct = (float).1;
    if (g->opt.distribution_mode == 0) {
        num_walls = 5;
        object_group_size = 2;
        door_prob = 0;
        min_pct = (float).2;
    }
    int partition[16] = {0}; /* RandGen::partition randgen.cpp:33-41 */
    int px = g->main_height - min_sep * num_walls - buf_h;
    for (int i = 0; i < px; i++) partition[rng_randn(&g->rand_gen, num_walls)] += 1;
    int curr_h = 0;
    for (int k = 0; k < num_walls; k++) {
        int dy = min_sep + partition[k];
        curr_h += dy;
        int use_door = (dy > 5) && rng_rand01(&g->rand_gen) < door_prob;
        fb_add_walls(g, (float)curr_h, use_door, min_pct);
    }
    Ent *agent = &g->pool[g->agent];
    agent->y = agent->ry;
    int num_good = rng_randn(&g->rand_gen, 10) + 10;
    int num_bad = rng_randn(&g->rand_gen, 10) + 10;
    for (int i = 0; i < g->main_width; i++) {
        Ent *present = push_entity(g, (float)(i + .5), (float)(g->main_height - .5), 0, 0, (float).5, (float).5, FB_PRESENT);
        choose_random_theme(g, present);
    }
    spawn_entities(g, num_good, (float).5, FB_GOOD_OBJ, 0, 0, (float)g->main_width, (float)g->main_height);
    spawn_entities(g, num_bad, (float).5, FB_BAD_OBJ, 0, 0, (float)g->main_width, (float)g->main_height);
    for (int k = 0; k < g->n_ents; k++) {
        Ent *ent = &g->pool[g->ents[k]];
        if (ent->type == FB_GOOD_OBJ || ent->type == FB_BAD_OBJ) {
            ent->image_theme = rng_randn(&g->rand_gen, object_group_size);
            fit_aspect_ratio(g, ent);
        }
    }
    g->pool[g->agent].rotation = -1 * PI_F / 2;
}

/* ---- StarPilot: starpilot.cpp:148-443 ---- */
static void sp_init_hps(Game *g) { /* starpilot.cpp:148-227 */
    float scale = 1;
    for (int i = 0; i < SP_NUM_BASIC_OBJECTS; i++) {
        g->hp_vs[i] = 1;
        g->hp_healths[i] = 0;
        g->hp_object_prob_weight[i] = 1;
        g->hp_object_r[i] = scale / 2;
    }
    float default_bullet_r = (float)(scale / 2.5);
    int dm = g->opt.distribution_mode;
    if (dm == 0) {
        g->hp_object_prob_weight[SP_METEOR] = 0;
        g->hp_object_prob_weight[SP_CLOUD] = 0;
        g->hp_object_prob_weight[SP_TURRET] = 0;
        g->hp_object_prob_weight[SP_FAST_FLYER] = 0;
        g->hp_vs[SP_FLYER] = (float).75;
        g->hp_vs[SP_BULLET2] = (float)1.25;
        g->hp_healths[SP_TURRET] = 5;
        g->hp_healths[SP_FLYER] = 2;
        g->hp_healths[SP_FAST_FLYER] = 1;
        g->maxspeed = (float)0.75;
    } else if (dm == 1) {
        g->hp_vs[SP_BULLET2] = 2;
        g->hp_healths[SP_TURRET] = 5;
        g->hp_healths[SP_FLYER] = 2;
        g->hp_healths[SP_FAST_FLYER] = 1;
        g->maxspeed = (float)0.75;
    } else if (dm == 2) {
        g->hp_vs[SP_BULLET2] = 2;
        g->hp_healths[SP_TURRET] = 10;
        g->hp_healths[SP_FLYER] = 5;
        g->hp_healths[SP_FAST_FLYER] = 2;
        g->maxspeed = (float)0.5;
        default_bullet_r = scale / 5;
    } else {
        fatal("fassert(false) starpilot.cpp:188");
    }
    for (int i = 0; i < SP_NUM_BASIC_OBJECTS; i++) g->hp_bullet_r[i] = default_bullet_r;
    g->hp_healths[SP_METEOR] = 500;
    g->hp_vs[SP_FAST_FLYER] = (float)1.5;
    g->hp_vs[SP_BULLET_PLAYER] = 2;
    g->hp_vs[SP_BULLET3] = 2;
    g->hp_object_r[SP_TURRET] = scale * 2;
    g->hp_object_r[SP_METEOR] = scale * 2;
    g->hp_object_r[SP_CLOUD] = scale * 2;
    g->hp_object_prob_weight[SP_FLYER] = 3;
    g->hp_slow_v = (float).5;
    g->hp_max_group_size = 5;
    g->hp_weapon_bullet_dist = 3;
    g->hp_min_enemy_delta_t = 10;
    g->hp_max_enemy_delta_t = g->hp_min_enemy_delta_t + 20;
    g->hp_spawn_right_threshold = 0.9f;
    g->hp_object_prob_weight[SP_BULLET_PLAYER] = 0;
    g->hp_object_prob_weight[SP_BULLET2] = 0;
    g->hp_object_prob_weight[SP_BULLET3] = 0;
    g->total_prob_weight = 0;
    for (int i = 2; i < SP_NUM_BASIC_OBJECTS; i++) g->total_prob_weight += g->hp_object_prob_weight[i];
}

static void sp_choose_random_theme_ent(Game *g, Ent *ent) { ent->image_theme = rng_randn(&g->rand_gen, g->assets->type_num_themes[ent->image_type]); }

static void sp_add_spawners(Game *g) { /* starpilot.cpp:229-325 */
    int t = 1 + rng_randint(&g->rand_gen, g->hp_min_enemy_delta_t, g->hp_max_enemy_delta_t);
    int can_spawn_left = g->opt.distribution_mode != 0;
    for (int i = 0; t <= SP_SHOOTER_WIN_TIME; i++) {
        int group_size = 1;
        float start_weight = rng_rand01(&g->rand_gen) * g->total_prob_weight;
        float curr_weight = start_weight;
        int type;
        for (type = 2; type < SP_NUM_BASIC_OBJECTS; type++) {
            curr_weight -= g->hp_object_prob_weight[type];
            if (curr_weight <= 0) break;
        }
        if (type >= SP_NUM_BASIC_OBJECTS) type = SP_NUM_BASIC_OBJECTS - 1;
        float r = g->hp_object_r[type];
        int flyer_theme = 0;
        if (type == SP_FLYER || type == SP_FAST_FLYER) {
            group_size = rng_randint(&g->rand_gen, 0, g->hp_max_group_size) + 1;
            flyer_theme = rng_randn(&g->rand_gen, SP_NUM_SHIP_THEMES);
        }
        float y_pos = rand_pos(g, r, 0, (float)g->main_height);
        for (int j = 0; j < group_size; j++) {
            int spawn_time = t + j * 5;
            int fire_time = rng_randint(&g->rand_gen, 10, 100);
            float k = 2 * PI_F / 4;
            float theta = (float)((rng_rand01(&g->rand_gen) - .5) * k);
            float v_scale = g->hp_vs[type];
            if (rng_randint(&g->rand_gen, 0, 2) == 1) theta = 0;
            float health = g->hp_healths[type];
            if (type == SP_METEOR || type == SP_CLOUD) {
                theta = 0;
                v_scale = g->hp_slow_v;
                fire_time = -1;
            } else if (type == SP_TURRET) {
                theta = 0;
                v_scale = g->hp_slow_v;
                fire_time = rng_randint(&g->rand_gen, 20, 30);
            }
            v_scale *= SP_V_SCALE;
            float vx = (float)(-1 * cos((double)theta) * v_scale);
            float vy = (float)(sin((double)theta) * v_scale);
            int spawn_right = 1;
            float x_pos;
            if (type == SP_FLYER || type == SP_FAST_FLYER) {
                if (rng_rand01(&g->rand_gen) > g->hp_spawn_right_threshold && can_spawn_left) spawn_right = 0;
            }
            if (spawn_right) {
                x_pos = g->main_width + r;
            } else {
                x_pos = -r;
                vx *= -1;
            }
            if (g->n_spawners >= SP_MAX_SPAWNERS) fatal("spawner list overflow");
            Ent *sp = &g->spawners[g->n_spawners++];
            ent_init(sp, x_pos, y_pos, vx, vy, r, r, type);
            sp->fire_time = fire_time;
            sp->spawn_time = spawn_time;
            sp->health = health;
            if (type == SP_CLOUD) {
                sp->render_z = 1;
                sp_choose_random_theme_ent(g, sp);
            } else if (type == SP_METEOR) {
                sp_choose_random_theme_ent(g, sp);
            } else if (type == SP_FLYER || type == SP_FAST_FLYER) {
                sp->image_theme = flyer_theme;
                sp->rotation = ((vx > 0) ? -1 : 1) * PI_F / 2;
            } else if (type == SP_TURRET) {
                sp_choose_random_theme_ent(g, sp);
                match_aspect_ratio(g, sp);
            }
        }
        t += rng_randint(&g->rand_gen, g->hp_min_enemy_delta_t, g->hp_max_enemy_delta_t);
    }
}

/* std::sort(spawners.begin(), spawners.end(), spawn_cmp) -- starpilot.cpp:334, spawn_cmp :29-31.
 * Third-party algorithm restated: libstdc++ (GCC 11) bits/stl_algo.h std::__sort = introsort
 * (median-of-3 to first, unguarded partition, threshold 16) + final insertion sort.  The order of
 * equal spawn_times depends on it.  The heapsort fallback (depth limit 2*floor(log2 n)) is not
 * restated; the oracle stops if it would be taken. */
static int sp_cmp(const Ent *x, const Ent *y) { return x->spawn_time > y->spawn_time; }
static void sp_swap(Ent *a, Ent *b) { Ent t = *a; *a = *b; *b = t; }
static void sp_unguarded_linear_insert(Ent *base, int last) {
    Ent val = base[last];
    int next = last - 1;
    while (sp_cmp(&val, &base[next])) {
        base[last] = base[next];
        last = next;
        next--;
    }
    base[last] = val;
}
static void sp_insertion_sort(Ent *base, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; i++) {
        if (sp_cmp(&base[i], &base[first])) {
            Ent val = base[i];
            for (int k = i; k > first; k--) base[k] = base[k - 1];
            base[first] = val;
        } else {
            sp_unguarded_linear_insert(base, i);
        }
    }
}
static void sp_introsort_loop(Ent *base, int first, int last, int depth_limit) {
    while (last - first > 16) {
        if (depth_limit == 0) fatal("std::sort heapsort fallback not restated");
        --depth_limit;
        int mid = first + (last - first) / 2;
        int a = first + 1, b = mid, c = last - 1, result = first;
        if (sp_cmp(&base[a], &base[b])) {
            if (sp_cmp(&base[b], &base[c])) sp_swap(&base[result], &base[b]);
            else if (sp_cmp(&base[a], &base[c])) sp_swap(&base[result], &base[c]);
            else sp_swap(&base[result], &base[a]);
        } else if (sp_cmp(&base[a], &base[c])) sp_swap(&base[result], &base[a]);
        else if (sp_cmp(&base[b], &base[c])) sp_swap(&base[result], &base[c]);
        else sp_swap(&base[result], &base[b]);
        int lo = first + 1, hi = last;
        for (;;) {
            while (sp_cmp(&base[lo], &base[first])) ++lo;
            --hi;
            while (sp_cmp(&base[first], &base[hi])) --hi;
            if (!(lo < hi)) break;
            sp_swap(&base[lo], &base[hi]);
            ++lo;
        }
        sp_introsort_loop(base, lo, last, depth_limit);
        last = lo;
    }
}
static void sp_sort_spawners(Game *g) {
    int n = g->n_spawners;
    if (n == 0) return;
    int lg = 0;
    while ((1 << (lg + 1)) <= n) lg++;
    sp_introsort_loop(g->spawners, 0, n, lg * 2);
    if (n > 16) {
        sp_insertion_sort(g->spawners, 0, 16);
        for (int i = 16; i != n; i++) sp_unguarded_linear_insert(g->spawners, i);
    } else {
        sp_insertion_sort(g->spawners, 0, n);
    }
}

static void sp_game_step_tail(Game *g) { /* starpilot.cpp:363-430 */
    int is_firing = g->special_action != 0;
    for (int i = g->n_ents - 1; i >= 0; i--) {
        Ent *m = &g->pool[g->ents[i]];
        if (m->type == PLAYER) continue;
        int should_fire = 0; /* starpilot.cpp:351-361 */
        if (m->fire_time > 0) {
            if (m->type == SP_TURRET) should_fire = (g->cur_time - m->spawn_time) % m->fire_time == 0;
            else should_fire = g->cur_time - m->spawn_time == m->fire_time;
        }
        if (should_fire) {
            const Ent *agent = &g->pool[g->agent];
            int bullet_type = m->type == SP_TURRET ? SP_BULLET3 : SP_BULLET2;
            float bullet_r = g->hp_bullet_r[m->type];
            float b_vx = agent->x - m->x;
            float b_vy = agent->y - m->y;
            float bv_scale = (float)(g->hp_vs[bullet_type] * SP_V_SCALE / sqrt((double)(b_vx * b_vx + b_vy * b_vy)));
            b_vx = b_vx * bv_scale;
            b_vy = b_vy * bv_scale;
            Ent *nb = push_entity(g, m->x, m->y, b_vx, b_vy, bullet_r, bullet_r, bullet_type);
            face_direction(nb, b_vx, b_vy, -1 * PI_F / 2);
        }
        if (m->health <= 0 && sp_is_destructible(m->type) && !m->will_erase) {
            float r = (float)(.5 * m->rx);
            push_entity(g, m->x, m->y, m->vx, m->vy, r, r, EXPLOSION); /* spawn_child BAG:225-231 */
            g->reward += 1.0f;
            m->will_erase = 1;
        }
    }
    while (g->n_spawners > 0 && g->cur_time == g->spawners[g->n_spawners - 1].spawn_time) {
        int id = pool_alloc(g);
        g->pool[id] = g->spawners[g->n_spawners - 1];
        if (g->n_ents >= MAX_ENTS) fatal("entity list overflow");
        g->ents[g->n_ents++] = id;
        g->n_spawners--;
    }
    float bullet_r = g->hp_bullet_r[PLAYER];
    if (is_firing) {
        const Ent *agent = &g->pool[g->agent];
        float theta = g->special_action == 2 ? PI_F : 0;
        float v_scale = g->hp_vs[SP_BULLET_PLAYER] * SP_V_SCALE;
        float vx = (float)(cos((double)theta) * v_scale);
        float vy = (float)(sin((double)theta) * v_scale);
        float x_off = (float)(agent->rx * cos((double)theta));
        Ent *bullet = push_entity(g, agent->x + x_off, agent->y, vx, vy, bullet_r, bullet_r, SP_BULLET_PLAYER);
        bullet->collides_with_entities = 1;
        face_direction(bullet, vx, vy, 0);
        bullet->rotation -= PI_F / 2;
    }
    if (g->cur_time == SP_SHOOTER_WIN_TIME) {
        Ent *finish = push_entity(g, (float)g->main_width, (float)(g->main_height / 2), -1 * g->hp_slow_v * SP_V_SCALE, 0, 2, (float)(g->main_height / 2), SP_FINISH_LINE);
        choose_random_theme(g, finish);
        match_aspect_ratio_h(g, finish);
        finish->x = g->main_width + finish->rx;
    }
}

static void cr_fill_block_top(Game *g, int x, int y, int dx, int dy, int fill, int top) { /* coinrun.cpp:227-231 */
    if (!(dy > 0)) fatal("fassert dy > 0 (coinrun.cpp:228)");
    fill_elem(g, x, y, dx, dy - 1, fill);
    fill_elem(g, x, y + dy - 1, dx, 1, top);
}
static void cr_fill_ground_block(Game *g, int x, int y, int dx, int dy) { cr_fill_block_top(g, x, y, dx, dy, CR_WALL_MID, CR_WALL_TOP); }
static void cr_fill_lava_block(Game *g, int x, int y, int dx, int dy) { cr_fill_block_top(g, x, y, dx, dy, CR_LAVA_MID, CR_LAVA_TOP); }
static void cr_create_saw_enemy(Game *g, int x, int y) { /* coinrun.cpp:248-250 */
    push_entity(g, (float)(x + .5), (float)(y + .5), 0, 0, (float).5, (float).5, CR_SAW);
}
static void cr_create_enemy(Game *g, int x, int y) { /* coinrun.cpp:252-258 */
    float vx = (float)(.15 * (rng_randn(&g->rand_gen, 2) * 2 - 1));
    Ent *ent = push_entity(g, (float)(x + .5), (float)(y + .5), vx, 0, (float).5, (float).5, CR_ENEMY);
    ent->smart_step = 1;
    ent->image_type = CR_ENEMY1;
    ent->render_z = 1;
    choose_random_theme(g, ent);
}
static void cr_create_crate(Game *g, int x, int y) { /* coinrun.cpp:260-263 */
    Ent *ent = push_entity(g, (float)(x + .5), (float)(y + .5), 0, 0, (float).5, (float).5, CR_CRATE);
    choose_random_theme(g, ent);
}

static void cr_generate_coin_to_the_right(Game *g) { /* coinrun.cpp:265-414 */
    Rng *r = &g->rand_gen;
    int max_difficulty = 3;
    int dif = rng_randn(r, max_difficulty) + 1;
    int num_sections = rng_randn(r, dif) + dif;
    int curr_x = 5, curr_y = 1;
    int pit_threshold = dif;
    int danger_type = rng_randn(r, 3);
    int allow_pit = (g->opt.debug_mode & (1 << 1)) == 0;
    int allow_crate = (g->opt.debug_mode & (1 << 2)) == 0;
    int allow_dy = (g->opt.debug_mode & (1 << 3)) == 0;
    int w = g->main_width;
    float _max_dy = g->max_jump * g->max_jump / (2 * g->gravity);
    float _max_dx = g->maxspeed * 2 * g->max_jump / g->gravity;
    int max_dy = (int)(_max_dy - .5);
    int max_dx = (int)(_max_dx - .5);
    int allow_monsters = 1;
    if (g->opt.distribution_mode == 0) allow_monsters = 0;
    for (int section_idx = 0; section_idx < num_sections; section_idx++) {
        if (curr_x + 15 >= w) break;
        int dy = rng_randn(r, 4) + 1 + (int)(dif / 3);
        if (!allow_dy) dy = 0;
        if (dy > max_dy) dy = max_dy;
        if (curr_y >= 20) dy *= -1;
        else if (curr_y >= 5 && rng_randn(r, 2) == 1) dy *= -1;
        int dx = rng_randn(r, 2 * dif) + 3 + (int)(dif / 3);
        curr_y += dy;
        if (curr_y < 1) curr_y = 1;
        int use_pit = allow_pit && (dx > 7) && (curr_y > 3) && (rng_randn(r, 20) >= pit_threshold);
        if (use_pit) {
            int x1 = rng_randn(r, 3) + 1;
            int x2 = rng_randn(r, 3) + 1;
            int pit_width = dx - x1 - x2;
            if (pit_width > max_dx) {
                pit_width = max_dx;
                x2 = dx - x1 - pit_width;
            }
            cr_fill_ground_block(g, curr_x, 0, x1, curr_y);
            cr_fill_ground_block(g, curr_x + dx - x2, 0, x2, curr_y);
            int lava_height = rng_randn(r, curr_y - 3) + 1;
            if (danger_type == 0) {
                cr_fill_lava_block(g, curr_x + x1, 1, pit_width, lava_height);
            } else if (danger_type == 1) {
                for (int ei = 0; ei < pit_width; ei++) cr_create_saw_enemy(g, curr_x + x1 + ei, 1);
            } else if (danger_type == 2) {
                for (int ei = 0; ei < pit_width; ei++) cr_create_enemy(g, curr_x + x1 + ei, 1);
            }
            if (pit_width > 4) {
                int x3, w1;
                if (pit_width == 5) {
                    x3 = 1 + rng_randn(r, 2);
                    w1 = 1 + rng_randn(r, 2);
                } else if (pit_width == 6) {
                    x3 = 2 + rng_randn(r, 2);
                    w1 = 1 + rng_randn(r, 2);
                } else {
                    x3 = 2 + rng_randn(r, 2);
                    int x4 = 2 + rng_randn(r, 2);
                    w1 = pit_width - x3 - x4;
                }
                cr_fill_ground_block(g, curr_x + x1 + x3, curr_y - 1, w1, 1);
            }
        } else {
            cr_fill_ground_block(g, curr_x, 0, dx, curr_y);
            int ob1_x = -1, ob2_x = -1;
            if (rng_randn(r, 10) < (2 * dif) && dx > 3) {
                ob1_x = curr_x + rng_randn(r, dx - 2) + 1;
                cr_create_saw_enemy(g, ob1_x, curr_y);
            }
            if (rng_randn(r, 10) < dif && dx > 3 && (max_dx >= 4) && allow_monsters) {
                ob2_x = curr_x + rng_randn(r, dx - 2) + 1;
                cr_create_enemy(g, ob2_x, curr_y);
            }
            if (allow_crate) {
                for (int i = 0; i < 2; i++) {
                    int crate_x = curr_x + rng_randn(r, dx - 2) + 1;
                    if (rng_randn(r, 2) == 1 && ob1_x != crate_x && ob2_x != crate_x) {
                        int pile_height = rng_randn(r, 3) + 1;
                        for (int j = 0; j < pile_height; j++) cr_create_crate(g, crate_x, curr_y + j);
                    }
                }
            }
        }
        if (!cr_is_wall(get_obj(g, curr_x - 1, curr_y))) set_obj(g, curr_x - 1, curr_y, CR_ENEMY_BARRIER);
        curr_x += dx;
        set_obj(g, curr_x, curr_y, CR_ENEMY_BARRIER);
    }
    set_obj(g, curr_x, curr_y, CR_GOAL);
    cr_fill_ground_block(g, curr_x, 0, 1, curr_y);
    fill_elem(g, curr_x + 1, 0, g->main_width - curr_x - 1, g->main_height, CR_WALL_MID);
}

static void cl_generate_platforms(Game *g) { /* climber.cpp:171-228 */
    Rng *r = &g->rand_gen;
    int difficulty = rng_randn(r, 3);
    int min_platforms = difficulty * difficulty + 1;
    int max_platforms = (difficulty + 1) * (difficulty + 1) + 1;
    int num_platforms = rng_randn(r, max_platforms - min_platforms + 1) + min_platforms;
    g->coin_quota = 0;
    g->coins_collected = 0;
    int curr_x = rng_randn(r, g->main_width - 4) + 2;
    int curr_y = 0;
    int margin_x = 3;
    float enemy_prob = g->opt.distribution_mode == 0 ? (float).2 : (float).5;
    for (int i = 0; i < num_platforms; i++) {
        int max_dy = (int)(g->max_jump * g->max_jump / (2 * g->gravity)); /* choose_delta_y climber.cpp:164-169 */
        int min_dy = 3;
        int delta_y = rng_randn(r, max_dy - min_dy + 1) + min_dy;
        int can_spawn_enemy = (curr_x >= margin_x) && (curr_x <= g->main_width - margin_x);
        if (can_spawn_enemy && (rng_rand01(r) < enemy_prob)) {
            /* the two draws sit in different arguments of one call: g++ (x86-64) evaluates them right to left,
             * i.e. the velocity sign first (pinned against the compiled reference) */
            float evx = (float)(.15 * (rng_randn(r, 2) * 2 - 1));
            float ey = (float)(curr_y + rng_randn(r, 2) + 2 + .5);
            Ent *ent = push_entity(g, (float)(curr_x + .5), ey, evx, 0, (float).5, (float).5, CL_ENEMY);
            ent->image_type = CL_ENEMY1;
            ent->smart_step = 1;
            ent->climber_spawn_x = (float)(curr_x + .5);
            match_aspect_ratio(g, ent);
        }
        curr_y += delta_y;
        int plat_len = 2 + rng_randn(r, 10);
        int vx = rng_randn(r, 2) * 2 - 1;
        if (curr_x < margin_x) vx = 1;
        if (curr_x > g->main_width - margin_x) vx = -1;
        int candidates[16], nc = 0;
        for (int j = 0; j < plat_len; j++) {
            int nx = curr_x + (j + 1) * vx;
            if (nx <= 0 || nx >= g->main_width - 1) break;
            candidates[nc++] = nx;
            set_obj(g, nx, curr_y, CL_WALL_TOP);
        }
        if (rng_rand01(r) < .5 || i == num_platforms - 1) {
            if (nc <= 0) fatal("fassert elems.size() > 0 (randgen.cpp:43)");
            int coin_x = candidates[rng_randn(r, nc)];
            push_entity(g, (float)(coin_x + .5), (float)(curr_y + 1.5), 0, 0, 0.3f, 0.3f, CL_COIN);
            g->coin_quota += 1;
        }
        if (nc <= 0) fatal("fassert elems.size() > 0 (randgen.cpp:43)");
        curr_x = candidates[rng_randn(r, nc)];
    }
}

static void ag_generate_resource(Rng *rng, uint32_t *px, int w, int h, int num_recurse, int blotch_scale, int is_rect);
static void bag_game_reset(Game *g) { /* BAG:758-797 */
    if (g->game_id == GAME_MINER) { /* choose_world_dim miner.cpp:116-129 */
        int dm = g->opt.distribution_mode;
        if (dm == 0) g->main_width = g->main_height = 10;
        else if (dm == 1) g->main_width = g->main_height = 20;
        else if (dm == 10) g->main_width = g->main_height = 35;
    }
    if (g->game_id == GAME_CHASER) g->main_width = g->main_height = g->maze_dim; /* choose_world_dim chaser.cpp:132-135 */
    if (g->game_id == GAME_JUMPER) { /* choose_world_dim jumper.cpp:201-217 */
        int dm = g->opt.distribution_mode;
        int wd = dm == 1 ? 40 : (dm == 10 ? 45 : 20);
        g->main_width = g->main_height = wd;
    }
    if (g->game_id == GAME_CAVEFLYER) { /* choose_world_dim caveflyer.cpp:131-146 */
        int dm = g->opt.distribution_mode;
        int wd = dm == 0 ? 30 : (dm == 1 ? 40 : (dm == 10 ? 60 : 20));
        g->main_width = g->main_height = wd;
    }
    if (g->game_id == GAME_DODGEBALL) { /* choose_world_dim dodgeball.cpp:250-259 */
        int wd = g->opt.distribution_mode == 10 ? 40 : 20;
        g->main_width = g->main_height = wd;
    }
    if (g->game_id == GAME_HEIST) { /* choose_world_dim heist.cpp:95-110 */
        int dm = g->opt.distribution_mode;
        if (dm == 0) g->world_dim = 9;
        else if (dm == 1) g->world_dim = 13;
        else if (dm == 10) g->world_dim = 23;
        g->maxspeed = (float).75;
        g->main_width = g->main_height = g->world_dim;
    }
    if (g->game_id == GAME_LEAPER) { /* choose_world_dim leaper.cpp:103-116 */
        int wd = 20;
        if (g->opt.distribution_mode == 0) wd = 9;
        else if (g->opt.distribution_mode == 1) wd = 15;
        g->main_width = g->main_height = wd;
    }
    if (g->game_id == GAME_FRUITBOT) { /* choose_world_dim fruitbot.cpp:152-160 */
        g->main_width = g->opt.distribution_mode == 0 ? 10 : 20;
        g->main_height = 60;
    }
    if (g->game_id == GAME_CLIMBER) { /* choose_world_dim climber.cpp:230-233 */
        g->main_width = g->opt.distribution_mode == 0 ? 16 : 20;
        g->main_height = 64;
    }
    if (g->game_id == GAME_MAZE) { /* choose_world_dim maze.cpp:40-53 */
        int dm = g->opt.distribution_mode;
        if (dm == 0) g->world_dim = 15;
        else if (dm == 1) g->world_dim = 25;
        else if (dm == 10) g->world_dim = 31;
        g->main_width = g->world_dim;
        g->main_height = g->world_dim;
    }
    if (!(g->main_width > 0 && g->main_height > 0)) fatal("fassert main dims (BAG:760)");
    g->bg_pct_x = rng_rand01(&g->rand_gen);
    g->grid_w = g->main_width;
    g->grid_h = g->main_height;
    memset(g->grid, 0, sizeof(int) * (size_t)(g->grid_w * g->grid_h));
    g->background_index = rng_randn(&g->rand_gen, g->assets->n_bg);
    if (g->opt.use_generated_assets) { /* BAG:769-773: AssetGen bggen(&rand_gen) paints this episode's 500 x 500 RGB32 background */
        if (!g->gen_bg.px) {
            g->gen_bg.px = (uint32_t *)malloc(sizeof(uint32_t) * 500 * 500);
            g->gen_bg.w = g->gen_bg.h = 500;
            g->gen_bg.generic = 0;
        }
        ag_generate_resource(&g->rand_gen, g->gen_bg.px, 500, 500, 1, 50, 1);
    }
    ents_clear(g);
    float ax, ay;
    float a_r = 0.4f;
    if (g->random_agent_start) {
        ax = rng_rand01(&g->rand_gen) * (g->main_width - 2 * a_r) + a_r;
        ay = rng_rand01(&g->rand_gen) * (g->main_height - 2 * a_r) + a_r;
    } else {
        ax = a_r;
        ay = a_r;
    }
    Ent *agent = push_entity(g, ax, ay, 0, 0, a_r, a_r, PLAYER);
    g->agent = g->ents[g->n_ents - 1];
    agent->smart_step = 1;
    agent->render_z = 1;
    erase_if_needed(g);
    fill_elem(g, 0, 0, g->main_width, g->main_height, SPACE);
}

static void game_reset(Game *g) {
    if (g->game_id == GAME_CHASER) ch_pre_reset(g);
    if (g->game_id == GAME_JUMPER) jp_pre_reset(g);
    bag_game_reset(g);
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:416-445 */
        Ent *agent = &g->pool[g->agent];
        g->gravity = 0.2f;
        g->max_jump = 1.5;
        g->air_control = 0.15f;
        g->maxspeed = (float).5;
        g->has_support = 0;
        g->facing_right = 1;
        if (g->opt.distribution_mode == 0) {
            agent->image_theme = 0;
            g->wall_theme = 0;
            g->background_index = 0;
        } else {
            choose_random_theme(g, agent);
            g->wall_theme = rng_randn(&g->rand_gen, 6);
        }
        agent->rx = (float).5;
        agent->ry = 0.5787f;
        agent->x = 1 + agent->rx;
        agent->y = 1 + agent->ry;
        g->last_agent_y = agent->y;
        g->is_on_crate = 0;
        /* init_floor_and_walls coinrun.cpp:241-246 */
        fill_elem(g, 0, 0, g->main_width, 1, CR_WALL_TOP);
        fill_elem(g, 0, 0, 1, g->main_height, CR_WALL_MID);
        fill_elem(g, g->main_width - 1, 0, 1, g->main_height, CR_WALL_MID);
        fill_elem(g, 0, g->main_height - 1, g->main_width, 1, CR_WALL_MID);
        cr_generate_coin_to_the_right(g);
    } else if (g->game_id == GAME_FRUITBOT) {
        fb_game_reset(g);
    } else if (g->game_id == GAME_LEAPER) {
        lp_game_reset(g);
    } else if (g->game_id == GAME_PLUNDER) {
        pl_game_reset(g);
    } else if (g->game_id == GAME_HEIST) {
        hs_game_reset(g);
    } else if (g->game_id == GAME_NINJA) {
        nj_game_reset(g);
    } else if (g->game_id == GAME_DODGEBALL) {
        db_game_reset(g);
    } else if (g->game_id == GAME_BOSSFIGHT) {
        bf2_game_reset(g);
    } else if (g->game_id == GAME_CHASER) {
        ch_game_reset(g);
    } else if (g->game_id == GAME_CAVEFLYER) {
        cf_game_reset(g);
    } else if (g->game_id == GAME_JUMPER) {
        jp_game_reset(g);
    } else if (g->game_id == GAME_STARPILOT) { /* starpilot.cpp:327-339 */
        g->center_agent = 0;
        sp_init_hps(g);
        g->n_spawners = 0;
        sp_add_spawners(g);
        sp_sort_spawners(g);
        Ent *agent = &g->pool[g->agent];
        agent->rotation = PI_F / 2;
        choose_random_theme(g, agent);
    } else if (g->game_id == GAME_BIGFISH) { /* bigfish.cpp:64-81 */
        Ent *agent = &g->pool[g->agent];
        g->center_agent = 0;
        g->fish_eaten = 0;
        float start_r = (float).5;
        if (g->opt.distribution_mode == 0) start_r = 1;
        g->r_inc = (BF_FISH_MAX_R - start_r) / BF_FISH_QUOTA;
        agent->rx = start_r;
        agent->ry = start_r;
        agent->y = 1 + agent->ry;
    } else if (g->game_id == GAME_MINER) { /* miner.cpp:131-205 */
        Ent *agent = &g->pool[g->agent];
        agent->rx = (float).5;
        agent->ry = (float).5;
        int main_area = g->main_height * g->main_width;
        g->center_agent = g->opt.distribution_mode == 10;
        g->grid_step = 1;
        float diamond_pct = 12 / 400.0f;
        float boulder_pct = 80 / 400.0f;
        int num_diamonds = (int)(diamond_pct * main_area);
        int num_boulders = (int)(boulder_pct * main_area);
        int k = num_diamonds + num_boulders + 1;
        static int obj_idxs[2048];
        static unsigned char used[2048];
        memset(used, 0, sizeof(used));
        for (int i = 0; i < k; i++) { /* RandGen::simple_choose randgen.cpp:70-88 */
            int next = rng_randn(&g->rand_gen, main_area);
            while (used[next]) next = rng_randn(&g->rand_gen, main_area);
            obj_idxs[i] = next;
            used[next] = 1;
        }
        int agent_x = obj_idxs[0] % g->main_width;
        int agent_y = obj_idxs[0] / g->main_width;
        agent->x = (float)(agent_x + .5);
        agent->y = (float)(agent_y + .5);
        for (int i = 0; i < main_area; i++) set_obj_idx(g, i, MN_DIRT);
        for (int i = 0; i < num_diamonds; i++) set_obj_idx(g, obj_idxs[i + 1], MN_DIAMOND);
        for (int i = 0; i < num_boulders; i++) set_obj_idx(g, obj_idxs[i + 1 + num_diamonds], MN_BOULDER);
        static unsigned char was_dirt[2048]; /* get_cells_with_type(DIRT) snapshot */
        for (int i = 0; i < main_area; i++) was_dirt[i] = g->grid[i] == MN_DIRT;
        set_obj(g, (int)agent->x, (int)agent->y, SPACE);
        for (int i = -1; i <= 1; i++)
            for (int j = -1; j <= 1; j++) {
                int ox = agent_x + i, oy = agent_y + j;
                if (get_obj(g, ox, oy) == MN_BOULDER) set_obj(g, ox, oy, MN_DIRT);
            }
        int ncand = 0;
        static int cand[2048];
        for (int cell = 0; cell < main_area; cell++) {
            if (!was_dirt[cell]) continue;
            int above_obj = get_obj_idx(g, cell + g->main_width);
            if (above_obj == MN_DIRT || above_obj == g->out_of_bounds_object) cand[ncand++] = cell;
        }
        if (ncand <= 0) fatal("fassert exit_candidates.size() > 0 (miner.cpp:196)");
        int exit_cell = cand[rng_randn(&g->rand_gen, ncand)];
        set_obj_idx(g, exit_cell, SPACE);
        Ent *ex = push_entity(g, (float)((exit_cell % g->main_width) + .5), (float)((exit_cell / g->main_width) + .5), 0, 0, (float).5, (float).5, MN_EXIT);
        ex->render_z = -1;
    } else if (g->game_id == GAME_CLIMBER) { /* climber.cpp:235-255 */
        Ent *agent = &g->pool[g->agent];
        g->gravity = 0.2f;
        g->max_jump = 1.5;
        g->air_control = 0.15f;
        g->maxspeed = (float).5;
        g->has_support = 0;
        g->facing_right = 1;
        agent->rx = (float).5;
        agent->ry = (float).5;
        agent->x = 1 + agent->rx;
        agent->y = 1 + agent->ry;
        choose_random_theme(g, agent);
        g->wall_theme = rng_randn(&g->rand_gen, 4);
        fill_elem(g, 0, 0, g->main_width, 1, CL_WALL_TOP);
        fill_elem(g, 0, 0, 1, g->main_height, CL_WALL_MID);
        fill_elem(g, g->main_width - 1, 0, 1, g->main_height, CL_WALL_MID);
        fill_elem(g, 0, g->main_height - 1, g->main_width, 1, CL_WALL_MID);
        cl_generate_platforms(g);
    } else if (g->game_id == GAME_MAZE) { /* maze.cpp:55-97 */
        static MazeGen mg;
        Ent *agent = &g->pool[g->agent];
        g->grid_step = 1;
        g->maze_dim = rng_randn(&g->rand_gen, (g->world_dim - 1) / 2) * 2 + 3;
        int margin = (g->world_dim - g->maze_dim) / 2;
        mg.maze_dim = g->maze_dim;
        mg.array_dim = g->maze_dim + 2;
        g->center_agent = g->opt.distribution_mode == 10;
        agent->rx = (float).5;
        agent->ry = (float).5;
        agent->x = (float)(margin + .5);
        agent->y = (float)(margin + .5);
        mg_generate_maze(&mg, &g->rand_gen);
        mg_place_objects(&mg, &g->rand_gen, MZ_GOAL, 1);
        for (int i = 0; i < g->main_width * g->main_height; i++) g->grid[i] = WALL_OBJ;
        for (int i = 0; i < g->maze_dim; i++)
            for (int j = 0; j < g->maze_dim; j++) set_obj(g, margin + i, margin + j, mg.grid[(j + MAZE_OFFSET) * mg.array_dim + i + MAZE_OFFSET]);
        if (margin > 0) {
            for (int i = 0; i < g->maze_dim + 2; i++) {
                set_obj(g, margin - 1, margin + i - 1, WALL_OBJ);
                set_obj(g, margin + g->maze_dim, margin + i - 1, WALL_OBJ);
                set_obj(g, margin + i - 1, margin - 1, WALL_OBJ);
                set_obj(g, margin + i - 1, margin + g->maze_dim, WALL_OBJ);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Rendering: replaces QPainter on a 64x64 Format_RGB32 buffer (reference src/game.cpp:77-91).   */
/* Third-party arithmetic restated: Qt5Gui 5.9.7 raster engine, non-antialiased, SourceOver:     */
/*   fillRect(QRectF)      -> [qRound(L),qRound(L+W)) x [qRound(T),qRound(T+H))                  */
/*   drawImage(QRectF,img) -> qt_scale_image_32 (qblendfunctions_p.h): 16.16 fixed-point nearest */
/*                            sampling, premultiplied SourceOver, const alpha = int(opacity*256) */
/* Pinned against the compiled reference's frames (tests/golden) and PyQt5 5.9.7 probes.         */
static int q_round(double d) { /* qglobal.h qRound(double), Qt 5.9 */
    return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1);
}
static uint32_t byte_mul(uint32_t x, uint32_t a) { /* qdrawhelper_p.h BYTE_MUL */
    uint32_t t = (x & 0xff00ffu) * a;
    t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
    t &= 0xff00ffu;
    x = ((x >> 8) & 0xff00ffu) * a;
    x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
    x &= 0xff00ff00u;
    return x | t;
}

typedef struct { double x, y, w, h; } RectD;

static void fill_rect(uint32_t *dst, RectD r, uint32_t color) {
    int x1 = q_round(r.x), x2 = q_round(r.x + r.w), y1 = q_round(r.y), y2 = q_round(r.y + r.h);
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; } /* toNormalizedFillRect (qpaintengine_raster.cpp) */
    if (y2 < y1) { int t = y1; y1 = y2; y2 = t; }
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    if (x2 > RES_W) x2 = RES_W;
    if (y2 > RES_H) y2 = RES_H;
    for (int y = y1; y < y2; y++)
        for (int x = x1; x < x2; x++) dst[y * RES_W + x] = color;
}

static void blend_px(uint32_t *d, uint32_t s, int io, uint32_t ca) { /* Blend_ARGB32_on_ARGB32_Source(AndConst)Alpha */
    if (io != 256) s = byte_mul(s, ca);
    *d = s + byte_mul(*d, 255u - (s >> 24));
}
static int opacity_to_io(float opacity) { /* QRasterPaintEngine: intOpacity = int(opacity * 256), opacity clamped by QPainter::setOpacity */
    double o = opacity;
    if (o < 0) o = 0;
    if (o > 1) o = 1;
    return (int)(o * 256);
}

/* qt_scale_image_32bit (qblendfunctions_p.h, Qt 5.9.7).  tr.w / tr.h may be negative (a 180 degree
 * rotation reaches this function as a negative scale through qt_mapRect_non_normalizing). */
static void draw_image_scaled(uint32_t *dst, const Img *src, int mirrored, RectD tr, float opacity) {
    if (!src->px) fatal("image not provided to the oracle");
    double sx = tr.w / (double)src->w;
    double sy = tr.h / (double)src->h;
    int ix = (int)(65536 / sx);
    int iy = (int)(65536 / sy);
    double right = tr.x + tr.w, bottom = tr.y + tr.h;
    int tx1 = q_round(tr.x), tx2 = q_round(right), ty1 = q_round(tr.y), ty2 = q_round(bottom);
    if (tx2 < tx1) { int t = tx1; tx1 = tx2; tx2 = t; }
    if (ty2 < ty1) { int t = ty1; ty1 = ty2; ty2 = t; }
    if (tx1 < 0) tx1 = 0;
    if (ty1 < 0) ty1 = 0;
    if (tx2 > RES_W) tx2 = RES_W;
    if (ty2 > RES_H) ty2 = RES_H;
    int w = tx2 - tx1, h = ty2 - ty1;
    if (w <= 0 || h <= 0) return;
    /* Qt 5.9: qCeil(..) - 1 for positive scales (newer Qt uses qFloor(..) + 1; pinned with
     * tests/tools/qt_drawimage_probe.py); qFloor(..) + 1 from the far edge for negative scales
     * (pinned with tests/tools/qt_rotate_probe.py). */
    uint32_t basex, srcy;
    if (sx < 0) basex = (uint32_t)src->w * 65536u + (uint32_t)((int)floor((tx1 + 0.5 - right) * ix) + 1);
    else basex = (uint32_t)((int)ceil((tx1 + 0.5 - tr.x) * ix) - 1);
    if (sy < 0) srcy = (uint32_t)src->h * 65536u + (uint32_t)((int)floor((ty1 + 0.5 - bottom) * iy) + 1);
    else srcy = (uint32_t)((int)ceil((ty1 + 0.5 - tr.y) * iy) - 1);
    if ((int)(srcy >> 16) >= src->h && iy < 0) { srcy += (uint32_t)iy; --h; }
    if ((int)(basex >> 16) >= src->w && ix < 0) { basex += (uint32_t)ix; --w; }
    int yend = (int)((srcy + (uint32_t)iy * (uint32_t)(h - 1)) >> 16);
    if (yend < 0 || yend >= src->h) --h;
    int xend = (int)((basex + (uint32_t)ix * (uint32_t)(w - 1)) >> 16);
    if (xend < 0 || xend >= src->w) --w;
    int io = opacity_to_io(opacity);
    uint32_t ca = (uint32_t)((io * 255) >> 8);
    for (int y = 0; y < h; y++) {
        const uint32_t *srow = src->px + (size_t)(srcy >> 16) * src->w;
        uint32_t srcx = basex;
        uint32_t *drow = dst + (ty1 + y) * RES_W + tx1;
        for (int x = 0; x < w; x++) {
            int sxp = (int)(srcx >> 16);
            blend_px(&drow[x], srow[mirrored ? (src->w - 1 - sxp) : sxp], io, ca);
            srcx += (uint32_t)ix;
        }
        srcy += (uint32_t)iy;
    }
}

/* ---- QPainter::drawImage(QRectF, QImage) for a source in QImage::Format_ARGB32 (not premultiplied): the reference's
 * generated assets (BAG:102-107).  qScaleFunctions / qTransformFunctions have no entry for that source format, so
 * QRasterPaintEngine::drawImage (qpaintengine_raster.cpp, Qt 5.9.7) takes its generic route:
 *   coverage: QRasterizer::rasterizeLine(a, b, h / w), a / b = the mapped midpoints of the rect's left / right edge, after
 *             its own line clipping against the widened clip rect.  Axis-aligned lines: the pixel box
 *             [int(left + .5), int(right - .5)] x [int(top + .5), int(bottom - .5)] of midpoint -+ half extent (doubles,
 *             clamped to the clip rect).  Other directions: the four corners pa -+ perp, pb -+ perp go to 26.6 by FLOORING and
 *             through the scan converter (pixel centres, 16.16 edge walkers, spans clipped to the frame).  A painter matrix
 *             of type TxScale (a 180 degree turn) instead fills [qRound) of the mapped rect;
 *   sampling: QSpanData::setupMatrix = inverse of (translate(1/65536, 1/65536) * matrix * translate(r.x, r.y) * scale(r.w / sw,
 *             r.h / sh)) and fetchTransformed: fx = int((m21 * cy + m11 * cx + dx) * 65536) at the span start, += int(m11 *
 *             65536) per pixel (fy likewise), source coordinate (fx >> 16, fy >> 16) clamped to the image;
 *   blend   : comp_func_SourceOver with const_alpha = (255 * intOpacity) >> 8; the generated pixels have alpha 0 or 255, so the
 *             ARGB32 -> premultiplied fetch conversion is the identity.
 * Third-party algorithm restated; pinned with tests/tools/qt_generic_image_probe.py against PyQt5 5.9.7 (20 000 unrotated
 * rects: 0 misses; 12 000 rotated draws incl. partly outside and near-axis angles: 4 single-pixel misses). */
typedef struct { double m11, m12, m21, m22, dx, dy; int type; } QtXform; /* type: 0 none, 1 translate, 2 scale, 4 rotate */
static int q_fuzzy_is_null(double v);
typedef struct { int y, x1, x2; } GSpan; /* x2 inclusive */
static int q26_equal(double p, double q) { return (int)((p - q) * 64) == 0; } /* qrasterizer.cpp q26Dot6Compare */
/* QRasterizer::rasterizeLine, not antialiased, clip rect = the 64 x 64 frame; returns the number of spans */
static int rasterize_line(double ax, double ay, double bx, double by, double width, GSpan *sp) {
    const int cw = RES_W, ch = RES_H;
    if ((ax == bx && ay == by) || width == 0) return 0;
    double pax = ax, pay = ay, pbx = bx, pby = by;
    double offx = fabs(by - ay) * width * 0.5, offy = fabs(bx - ax) * width * 0.5;
    double cl = 0 - offx, ct = 0 - offy, cr = (cw - 1) + 1 + offx, cb = (ch - 1) + 1 + offy;
#define IN_CLIP(px, py) (cl <= (px) && (px) <= cr && ct <= (py) && (py) <= cb)
    if (!IN_CLIP(pax, pay) || !IN_CLIP(pbx, pby)) {
        double t1 = 0, t2 = 1;
        const double o[2] = {pax, pay}, dd[2] = {pbx - pax, pby - pay}, low[2] = {cl, ct}, high[2] = {cr, cb};
        for (int i = 0; i < 2; i++) {
            if (dd[i] == 0) {
                if (o[i] <= low[i] || o[i] >= high[i]) return 0;
                continue;
            }
            const double d_inv = 1 / dd[i];
            double t_low = (low[i] - o[i]) * d_inv, t_high = (high[i] - o[i]) * d_inv;
            if (t_low > t_high) { double t = t_low; t_low = t_high; t_high = t; }
            if (t1 < t_low) t1 = t_low;
            if (t2 > t_high) t2 = t_high;
            if (t1 >= t2) return 0;
        }
        const double npax = pax + (pbx - pax) * t1, npay = pay + (pby - pay) * t1, npbx = pax + (pbx - pax) * t2, npby = pay + (pby - pay) * t2;
        pax = npax; pay = npay; pbx = npbx; pby = npby;
    }
#undef IN_CLIP
    {
        const double d0x = ax - bx, d0y = ay - by, w0 = d0x * d0x + d0y * d0y;
        const double dx = pax - pbx, dy = pay - pby, w = dx * dx + dy * dy;
        if (w == 0) return 0;
        width *= sqrt(w0 / w);
    }
    if (q26_equal(pay, pby)) {
        if (q26_equal(pax, pbx)) return 0;
        const double x = (pax + pbx) * 0.5, dx = fabs(pbx - pax) * 0.5, y = pay, dy = width * dx;
        pax = x; pay = y - dy;
        pbx = x; pby = y + dy;
        width = 1 / width;
    }
    int n = 0;
    if (q26_equal(pax, pbx)) {
        if (pay > pby) { double t = pax; pax = pbx; pbx = t; t = pay; pay = pby; pby = t; }
        const double dy = pby - pay, half = 0.5 * width * dy;
        double left = pax - half, right = pax + half;
        left = left < 0 ? 0 : (left > cw ? cw : left);
        right = right < 0 ? 0 : (right > cw ? cw : right);
        pay = pay < 0 ? 0 : (pay > ch ? ch : pay);
        pby = pby < 0 ? 0 : (pby > ch ? ch : pby);
        if (q26_equal(left, right) || q26_equal(pay, pby)) return 0;
        const int iTop = (int)(pay + 0.5), iBottom = pby < 0.5 ? -1 : (int)(pby - 0.5);
        const int iLeft = (int)(left + 0.5), iRight = right < 0.5 ? -1 : (int)(right - 0.5);
        for (int y = iTop; y <= iBottom; y++)
            if (iRight >= iLeft) { sp[n].y = y; sp[n].x1 = iLeft; sp[n].x2 = iRight; n++; }
        return n;
    }
    if (pay > pby) { double t = pax; pax = pbx; pbx = t; t = pay; pay = pby; pby = t; }
    const double dlx = (pbx - pax) * (0.5 * width), dly = (pby - pay) * (0.5 * width);
    const double perpx = dly, perpy = -dlx;
    double cxs[4], cys[4]; /* top, right, bottom, left */
    if (pax < pbx) {
        cxs[0] = pax + perpx; cys[0] = pay + perpy; cxs[3] = pax - perpx; cys[3] = pay - perpy;
        cxs[1] = pbx + perpx; cys[1] = pby + perpy; cxs[2] = pbx - perpx; cys[2] = pby - perpy;
    } else {
        cxs[0] = pax - perpx; cys[0] = pay - perpy; cxs[3] = pbx - perpx; cys[3] = pby - perpy;
        cxs[1] = pax + perpx; cys[1] = pay + perpy; cxs[2] = pbx + perpx; cys[2] = pby + perpy;
    }
    int qx[4], qy[4];
    for (int i = 0; i < 4; i++) {
        qx[i] = (int)floor(cxs[i] * 64.);
        qy[i] = (int)floor(cys[i] * 64.);
    }
    int cnt[RES_H], xa[RES_H], xb[RES_H];
    memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < 4; i++) { /* QScanConverter::mergeLine */
        int a_x = qx[i], a_y = qy[i], b_x = qx[(i + 1) & 3], b_y = qy[(i + 1) & 3];
        if (a_y > b_y) { int t = a_x; a_x = b_x; b_x = t; t = a_y; a_y = b_y; b_y = t; }
        int itop = (a_y + 32) >> 6, ibot = (b_y - 32) >> 6;
        if (itop < 0) itop = 0;
        if (ibot > ch - 1) ibot = ch - 1;
        if (itop > ibot) continue;
        int xfp = 32768 + a_x * 1024, slope = 0;
        if (b_x != a_x) {
            slope = (int)((b_x - a_x) / (double)(b_y - a_y) * 65536.);
            xfp += (int)(((long long)slope * (long long)((itop << 16) + 32768 - (a_y << 10))) >> 16);
        }
        for (int y = itop; y <= ibot; y++) {
            const int xi = xfp >> 16;
            if (cnt[y] == 0) xa[y] = xb[y] = xi;
            else {
                if (xi < xa[y]) xa[y] = xi;
                if (xi > xb[y]) xb[y] = xi;
            }
            cnt[y]++;
            xfp += slope;
        }
    }
    for (int y = 0; y < ch; y++)
        if (cnt[y] >= 2) {
            const int x1 = xa[y] < 0 ? 0 : xa[y], x2 = (xb[y] > cw ? cw : xb[y]) - 1;
            if (x2 >= x1) { sp[n].y = y; sp[n].x1 = x1; sp[n].x2 = x2; n++; }
        }
    return n;
}
/* p.drawImage(r, img) under painter matrix m (BAG:897-906) */
static void draw_image_generic(uint32_t *dst, const Img *src, int mirrored, const QtXform *m, RectD r, float opacity) {
    if (!(r.w > 0 && r.h > 0)) return; /* r.isEmpty() */
    if (!src->px) fatal("generated image missing");
    /* sampling matrix: copy = m; copy.translate(r.x, r.y); copy.scale(r.w / sw, r.h / sh)  (qtransform.cpp) */
    double c11 = m->m11, c12 = m->m12, c21 = m->m21, c22 = m->m22, cdx = m->dx, cdy = m->dy;
    int ctype = m->type;
    if (ctype == 0) { cdx = r.x; cdy = r.y; ctype = 1; }
    else if (ctype == 1) { cdx += r.x; cdy += r.y; }
    else if (ctype == 2) { cdx += r.x * c11; cdy += r.y * c22; }
    else { cdx += r.x * c11 + r.y * c21; cdy += r.y * c22 + r.x * c12; }
    const double scx = r.w / src->w, scy = r.h / src->h;
    if (ctype == 4) { c12 *= scx; c21 *= scy; }
    c11 *= scx;
    c22 *= scy;
    if (ctype < 2) ctype = 2;
    /* QSpanData::setupMatrix: inv = (translate(1/65536, 1/65536) * copy).inverted() */
    const double dlt = 1.0 / 65536;
    double i11, i12, i21, i22, idx, idy;
    if (ctype == 2) {
        const double p11 = 1.0 * c11, p22 = 1.0 * c22, p31 = dlt * c11 + cdx, p32 = dlt * c22 + cdy;
        i11 = 1. / p11; i22 = 1. / p22; i12 = 0; i21 = 0;
        idx = -p31 * i11; idy = -p32 * i22;
    } else {
        const double p11 = 1.0 * c11 + 0.0 * c21, p12 = 1.0 * c12 + 0.0 * c22, p21 = 0.0 * c11 + 1.0 * c21, p22 = 0.0 * c12 + 1.0 * c22;
        const double p31 = dlt * c11 + dlt * c21 + cdx, p32 = dlt * c12 + dlt * c22 + cdy;
        const double dtr = p11 * p22 - p12 * p21, dinv = 1.0 / dtr; /* QMatrix::inverted */
        i11 = p22 * dinv; i12 = -p12 * dinv; i21 = -p21 * dinv; i22 = p11 * dinv;
        idx = (p21 * p32 - p22 * p31) * dinv; idy = (p12 * p31 - p11 * p32) * dinv;
    }
    const int fdx = (int)(i11 * 65536.), fdy = (int)(i12 * 65536.);
    static GSpan sp[RES_H];
    int ns = 0;
    if (m->type == 2) { /* fillRect_normalized(QRect(qRound of the mapped rect)) */
        double x = m->m11 * r.x + m->dx, y = m->m22 * r.y + m->dy, ww = m->m11 * r.w, hh = m->m22 * r.h;
        if (ww < 0) { ww = -ww; x -= ww; }
        if (hh < 0) { hh = -hh; y -= hh; }
        int x1 = q_round(x), y1 = q_round(y), x2 = q_round(x + ww), y2 = q_round(y + hh);
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > RES_W) x2 = RES_W;
        if (y2 > RES_H) y2 = RES_H;
        for (int yy = y1; yy < y2; yy++)
            if (x2 > x1) { sp[ns].y = yy; sp[ns].x1 = x1; sp[ns].x2 = x2 - 1; ns++; }
    } else {
        const double l = r.x, t = r.y, rr = r.x + r.w, b = r.y + r.h;
        double ax = (l + l) * 0.5, ay = (t + b) * 0.5, bx = (rr + rr) * 0.5, by = (t + b) * 0.5;
        if (m->type == 1) { ax += m->dx; ay += m->dy; bx += m->dx; by += m->dy; }
        else if (m->type == 4) {
            const double tax = m->m11 * ax + m->m21 * ay + m->dx, tay = m->m12 * ax + m->m22 * ay + m->dy;
            const double tbx = m->m11 * bx + m->m21 * by + m->dx, tby = m->m12 * bx + m->m22 * by + m->dy;
            ax = tax; ay = tay; bx = tbx; by = tby;
        }
        ns = rasterize_line(ax, ay, bx, by, r.h / r.w, sp);
    }
    const int io = opacity_to_io(opacity);
    const uint32_t ca = (uint32_t)((io * 255) >> 8);
    for (int k = 0; k < ns; k++) {
        const double ccx = sp[k].x1 + 0.5, ccy = sp[k].y + 0.5;
        int fx = (int)((i21 * ccy + i11 * ccx + idx) * 65536.);
        int fy = (int)((i22 * ccy + i12 * ccx + idy) * 65536.);
        for (int x = sp[k].x1; x <= sp[k].x2; x++) {
            int px = fx >> 16, py = fy >> 16;
            px = px < 0 ? 0 : (px > src->w - 1 ? src->w - 1 : px);
            py = py < 0 ? 0 : (py > src->h - 1 ? src->h - 1 : py);
            blend_px(&dst[sp[k].y * RES_W + x], src->px[py * src->w + (mirrored ? src->w - 1 - px : px)], io, ca);
            fx += fdx;
            fy += fdy;
        }
    }
}
static void draw_image_rect(uint32_t *dst, const Img *src, int mirrored, RectD tr, float opacity) { /* untransformed painter */
    if (src->generic) {
        const QtXform ident = {1, 0, 0, 1, 0, 0, 0};
        draw_image_generic(dst, src, mirrored, &ident, tr, opacity);
    } else {
        draw_image_scaled(dst, src, mirrored, tr, opacity);
    }
}

/* qt_transform_image + qt_transform_image_rasterize (qblendfunctions_p.h, Qt 5.9.7): the path
 * drawImage takes when the painter matrix has a rotation (QTransform::type() > TxScale).
 * Vertices: x,y in device space, u,v in source pixels. */
typedef struct { double x, y, u, v; } TVert;
typedef struct {
    uint32_t *dst;
    const Img *src;
    int mirrored, io;
    uint32_t ca;
    int dudx, dvdx, dudy, dvdy, u0, v0;
} TRast;
static void transform_rasterize(const TRast *t, TVert tl, TVert bl, TVert tr, TVert br, double top_y, double bottom_y) {
    int from_y = q_round(top_y), to_y = q_round(bottom_y);
    if (from_y < 0) from_y = 0;
    if (to_y > RES_H) to_y = RES_H;
    if (from_y >= to_y) return;
    double left_slope = (bl.x - tl.x) / (bl.y - tl.y);
    double right_slope = (br.x - tr.x) / (br.y - tr.y);
    int dx_l = (int)(left_slope * 0x10000);
    int dx_r = (int)(right_slope * 0x10000);
    int x_l = (int)((tl.x + (0.5 + from_y - tl.y) * left_slope + 0.5) * 0x10000);
    int x_r = (int)((tr.x + (0.5 + from_y - tr.y) * right_slope + 0.5) * 0x10000);
    const Img *src = t->src;
    for (int y = from_y; y < to_y; y++) {
        int from_x = x_l >> 16, to_x = x_r >> 16;
        if (from_x < 0) from_x = 0;
        if (to_x > RES_W) to_x = RES_W;
        for (int x = from_x; x < to_x; x++) {
            int uu = (x * t->dudx + y * t->dudy + t->u0) >> 16;
            int vv = (x * t->dvdx + y * t->dvdy + t->v0) >> 16;
            if (uu < 0) uu = 0; /* out-of-range source coordinates are clamped to the source rect */
            if (uu > src->w - 1) uu = src->w - 1;
            if (vv < 0) vv = 0;
            if (vv > src->h - 1) vv = src->h - 1;
            blend_px(&t->dst[y * RES_W + x], src->px[(size_t)vv * src->w + (t->mirrored ? (src->w - 1 - uu) : uu)], t->io, t->ca);
        }
        x_l += dx_l;
        x_r += dx_r;
    }
}
static void draw_image_transformed(uint32_t *dst, const Img *src, int mirrored, RectD r, double m11, double m12, double m21, double m22, double dx, double dy, float opacity) {
    if (!src->px) fatal("image not provided to the oracle");
    TVert v[4];
    double L = r.x, T = r.y, R = r.x + r.w, B = r.y + r.h;
    double px[4] = {L, R, R, L}, py[4] = {T, T, B, B};
    double pu[4] = {0, (double)src->w, (double)src->w, 0}, pv[4] = {0, 0, (double)src->h, (double)src->h};
    for (int i = 0; i < 4; i++) {
        v[i].x = m11 * px[i] + m21 * py[i] + dx;
        v[i].y = m12 * px[i] + m22 * py[i] + dy;
        v[i].u = pu[i];
        v[i].v = pv[i];
    }
    int topmost = 0;
    for (int i = 1; i < 4; i++)
        if (v[i].y < v[topmost].y) topmost = i;
    TVert q[4];
    for (int i = 0; i < 4; i++) q[i] = v[(topmost + i) & 3];
    double dx1 = q[1].x - q[0].x, dy1 = q[1].y - q[0].y, dx2 = q[3].x - q[0].x, dy2 = q[3].y - q[0].y;
    if (dx1 * dy2 - dx2 * dy1 > 0) { TVert t = q[1]; q[1] = q[3]; q[3] = t; }
    TVert u = {q[1].x - q[0].x, q[1].y - q[0].y, q[1].u - q[0].u, q[1].v - q[0].v};
    TVert w = {q[2].x - q[0].x, q[2].y - q[0].y, q[2].u - q[0].u, q[2].v - q[0].v};
    double det = u.x * w.y - u.y * w.x;
    if (det == 0) return;
    double det_inv = 1 / det;
    double i11 = (u.u * w.y - u.y * w.u) * det_inv;
    double i12 = (u.x * w.u - u.u * w.x) * det_inv;
    double i21 = (u.v * w.y - u.y * w.v) * det_inv;
    double i22 = (u.x * w.v - u.v * w.x) * det_inv;
    double mdx = q[0].u - i11 * q[0].x - i12 * q[0].y;
    double mdy = q[0].v - i21 * q[0].x - i22 * q[0].y;
    TRast t;
    t.dst = dst;
    t.src = src;
    t.mirrored = mirrored;
    t.io = opacity_to_io(opacity);
    t.ca = (uint32_t)((t.io * 255) >> 8);
    t.dudx = (int)(i11 * 0x10000);
    t.dvdx = (int)(i21 * 0x10000);
    t.dudy = (int)(i12 * 0x10000);
    t.dvdy = (int)(i22 * 0x10000);
    t.u0 = (int)ceil((0.5 * i11 + 0.5 * i12 + mdx) * 0x10000) - 1;
    t.v0 = (int)ceil((0.5 * i21 + 0.5 * i22 + mdy) * 0x10000) - 1;
    if (q[1].y < q[3].y) {
        transform_rasterize(&t, q[0], q[1], q[0], q[3], q[0].y, q[1].y);
        transform_rasterize(&t, q[1], q[2], q[0], q[3], q[1].y, q[3].y);
        transform_rasterize(&t, q[1], q[2], q[3], q[2], q[3].y, q[2].y);
    } else {
        transform_rasterize(&t, q[0], q[1], q[0], q[3], q[0].y, q[3].y);
        transform_rasterize(&t, q[0], q[1], q[3], q[2], q[3].y, q[1].y);
        transform_rasterize(&t, q[1], q[2], q[3], q[2], q[1].y, q[2].y);
    }
}

/* BAG:902-906: p.translate(cx, cy); p.rotate(rotation * 180 / PI); p.drawImage(QRectF(-w/2, -h/2, w, h), img).
 * QTransform::rotate special-cases 90/180/270 degrees, otherwise sin/cos of deg2rad*a (qtransform.cpp);
 * QTransform::type() decides between the transform path and the (possibly negative) scale path. */
static int q_fuzzy_is_null(double d) { return fabs(d) <= 0.000000000001; }
static void draw_image_rotated(uint32_t *dst, const Img *src, int mirrored, RectD adjusted, float rotation, float opacity) {
    double cx = adjusted.x + adjusted.w / 2, cy = adjusted.y + adjusted.h / 2;
    double a = (double)(rotation * 180 / PI_F);
    RectD r = {-adjusted.w / 2, -adjusted.h / 2, adjusted.w, adjusted.h};
    double sina = 0, cosa = 0;
    if (a == 0) cosa = 1; /* rotate(0) returns early: identity */
    else if (a == 90. || a == -270.) sina = 1.;
    else if (a == 270. || a == -90.) sina = -1.;
    else if (a == 180.) cosa = -1.;
    else {
        double b = 0.017453292519943295769 * a;
        sina = sin(b);
        cosa = cos(b);
    }
    double m11 = cosa, m12 = sina, m21 = -sina, m22 = cosa;
    if (src->generic) { /* generated asset: the generic span route whatever the matrix type */
        QtXform m = {m11, m12, m21, m22, cx, cy, 0};
        if (!q_fuzzy_is_null(m12) || !q_fuzzy_is_null(m21)) m.type = 4;
        else if (!q_fuzzy_is_null(m11 - 1) || !q_fuzzy_is_null(m22 - 1)) m.type = 2;
        else if (!q_fuzzy_is_null(cx) || !q_fuzzy_is_null(cy)) m.type = 1;
        draw_image_generic(dst, src, mirrored, &m, r, opacity);
        return;
    }
    if (!q_fuzzy_is_null(m12) || !q_fuzzy_is_null(m21)) {
        draw_image_transformed(dst, src, mirrored, r, m11, m12, m21, m22, cx, cy, opacity);
        return;
    }
    /* TxScale or below: drawImage maps the rect with qt_mapRect_non_normalizing and scales */
    double x1, y1, x2, y2;
    if (!q_fuzzy_is_null(m11 - 1) || !q_fuzzy_is_null(m22 - 1)) { /* TxScale */
        x1 = m11 * r.x + cx;
        y1 = m22 * r.y + cy;
        x2 = m11 * (r.x + r.w) + cx;
        y2 = m22 * (r.y + r.h) + cy;
    } else if (!q_fuzzy_is_null(cx) || !q_fuzzy_is_null(cy)) { /* TxTranslate */
        x1 = r.x + cx;
        y1 = r.y + cy;
        x2 = (r.x + r.w) + cx;
        y2 = (r.y + r.h) + cy;
    } else {
        x1 = r.x;
        y1 = r.y;
        x2 = r.x + r.w;
        y2 = r.y + r.h;
    }
    RectD tr = {x1, y1, x2 - x1, y2 - y1};
    draw_image_scaled(dst, src, mirrored, tr, opacity);
}

/* BAG:840-869 */
static void tile_image(uint32_t *dst, const Img *img, int mirrored, RectD rect, float tile_ratio, float opacity) {
    if (tile_ratio != 0) {
        if (tile_ratio < 0) {
            tile_ratio = -1 * tile_ratio;
            int num_tiles = (int)(rect.h / (rect.w * tile_ratio));
            if (num_tiles < 1) num_tiles = 1;
            float tile_height = (float)(rect.h / num_tiles);
            float tile_width = (float)rect.w;
            for (int i = 0; i < num_tiles; i++) {
                RectD tr = {rect.x, rect.y + tile_height * i, tile_width, tile_height};
                draw_image_rect(dst, img, mirrored, tr, opacity);
            }
        } else {
            int num_tiles = (int)(rect.w / (rect.h * tile_ratio));
            if (num_tiles < 1) num_tiles = 1;
            float tile_width = (float)(rect.w / num_tiles);
            float tile_height = (float)rect.h;
            for (int i = 0; i < num_tiles; i++) {
                RectD tr = {rect.x + tile_width * i, rect.y, tile_width, tile_height};
                draw_image_rect(dst, img, mirrored, tr, opacity);
            }
        }
    } else {
        draw_image_rect(dst, img, mirrored, rect, opacity);
    }
}

static RectD get_screen_rect(const Game *g, float x, float y, float dx, float dy, float render_eps) { /* BAG:799-801 */
    RectD r;
    r.x = (x - render_eps) * g->unit - g->x_off;
    r.y = (g->view_dim - y - render_eps) * g->unit + g->y_off;
    r.w = (dx + 2 * render_eps) * g->unit;
    r.h = (dy + 2 * render_eps) * g->unit;
    return r;
}
static RectD adjust_rect(RectD b, RectD a) { /* src/qt-utils.h:12-19 */
    RectD r;
    r.x = b.x + b.w * a.x;
    r.y = b.y + b.h * a.y;
    r.w = b.w * a.w;
    r.h = b.h * a.h;
    return r;
}

static void prepare_for_drawing(Game *g, float rect_height) { /* BAG:819-838 */
    g->center_x = (float)(g->main_width * .5);
    g->center_y = (float)(g->main_height * .5);
    if (g->center_agent && g->game_id == GAME_CLIMBER) { /* choose_center climber.cpp:261-265 */
        g->center_x = (float)(g->main_width / 2.0);
        g->center_y = (float)(g->pool[g->agent].y + g->main_width / 2.0 - 5 * g->pool[g->agent].ry);
        g->visibility = (float)g->main_width;
    } else if (g->center_agent && g->game_id == GAME_FRUITBOT) { /* choose_center fruitbot.cpp:146-150 */
        g->center_x = (float)(g->main_width / 2.0);
        g->center_y = (float)(g->pool[g->agent].y + g->main_width / 2.0 - 2 * g->pool[g->agent].ry);
        g->visibility = (float)g->main_width;
    } else if (g->center_agent) {
        g->center_x = g->pool[g->agent].x; /* choose_center BAG:664-667 */
        g->center_y = g->pool[g->agent].y;
    } else {
        g->visibility = (float)(g->main_width > g->main_height ? g->main_width : g->main_height);
        if (g->visibility < g->min_visibility) g->visibility = g->min_visibility;
    }
    float raw_unit = 64 / g->visibility;
    g->unit = (float)(raw_unit * (rect_height / 64.0));
    g->view_dim = (float)(64.0 / raw_unit);
    g->x_off = g->unit * (g->center_x - g->view_dim / 2);
    g->y_off = g->unit * (g->center_y - g->view_dim / 2);
}

static int hook_image_for_type(const Game *g, int type) {
    if (g->game_id == GAME_MINER) { /* miner.cpp:84-92 */
        if (type == MN_MOVING_BOULDER) return MN_BOULDER;
        if (type == MN_MOVING_DIAMOND) return MN_DIAMOND;
    }
    if (g->game_id == GAME_JUMPER && type == PLAYER) { /* jumper.cpp:117-132 */
        const Ent *agent = &g->pool[g->agent];
        if (fabs((double)agent->vx) < .01 && g->action_vx == 0 && g->has_support) return PLAYER;
        if (g->facing_right) return (g->cur_time / 5 % 2 == 0 || !g->has_support) ? JP_PLAYER_RIGHT1 : JP_PLAYER_RIGHT2;
        return (g->cur_time / 5 % 2 == 0 || !g->has_support) ? JP_PLAYER_LEFT1 : JP_PLAYER_LEFT2;
    }
    if (g->game_id == GAME_CHASER && type == CH_ENEMY) { /* chaser.cpp:97-110 */
        if (g->cur_time - g->eat_time < g->eat_timeout) return CH_ENEMY_WEAK;
        int rem = (g->cur_time / 2) % 4;
        if (rem == 3) rem = 1;
        return CH_ENEMY + rem;
    }
    if (g->game_id == GAME_DODGEBALL && type == DB_DOOR) return g->num_enemies == 0 ? DB_DOOR_OPEN : DB_DOOR; /* dodgeball.cpp:90-96 */
    if (g->game_id == GAME_NINJA && type == PLAYER) { /* ninja.cpp:158-168 */
        const Ent *agent = &g->pool[g->agent];
        if (fabs((double)agent->vx) < .01 && g->action_vx == 0 && g->has_support) return PLAYER;
        return (g->cur_time / 5 % 2 == 0 || !g->has_support) ? NJ_PLAYER_RIGHT1 : NJ_PLAYER_RIGHT2;
    }
    if (g->game_id == GAME_CLIMBER) { /* climber.cpp:145-159 */
        if (type == PLAYER) {
            const Ent *agent = &g->pool[g->agent];
            if (!g->has_support) return CL_PLAYER_JUMP;
            if (fabs((double)agent->vx) < .01 && g->action_vx == 0 && g->has_support) return PLAYER;
            return (g->cur_time / 5 % 2 == 0 || !g->has_support) ? CL_PLAYER_RIGHT1 : CL_PLAYER_RIGHT2;
        } else if (type == CL_ENEMY_BARRIER) {
            return -1;
        }
    }
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:213-225 */
        if (type == PLAYER) {
            const Ent *agent = &g->pool[g->agent];
            if (fabs((double)agent->vx) < .01 && g->action_vx == 0 && g->has_support) return PLAYER;
            return (g->cur_time / 5 % 2 == 0 || !g->has_support) ? CR_PLAYER_RIGHT1 : CR_PLAYER_RIGHT2;
        } else if (type == CR_ENEMY_BARRIER) {
            return -1;
        }
    }
    return abs(type); /* BAG:438-440 */
}
static int hook_theme_for_grid_obj(const Game *g, int type) {
    if (g->game_id == GAME_NINJA && type == NJ_WALL_MID) return g->wall_theme; /* ninja.cpp:130-135 */
    if (g->game_id == GAME_JUMPER && (type == JP_CAVEWALL || type == JP_CAVEWALL_TOP)) return g->wall_theme; /* jumper.cpp:102-107 */
    if ((g->game_id == GAME_COINRUN || g->game_id == GAME_CLIMBER) && cr_is_wall(type)) return g->wall_theme; /* coinrun.cpp:133-138, climber.cpp:102-107 */
    return 0;
}
static RectD hook_adjusted_image_rect(const Game *g, int type, RectD rect) {
    if (g->game_id == GAME_LEAPER && type == PLAYER) { /* leaper.cpp:233-239 */
        RectD a = {0, -.275, 1, 1.55};
        return adjust_rect(rect, a);
    }
    if (g->game_id == GAME_COINRUN) { /* coinrun.cpp:64-70 */
        if (type == PLAYER || type == CR_PLAYER_JUMP || type == CR_PLAYER_RIGHT1 || type == CR_PLAYER_RIGHT2) {
            RectD a = {0, -.7415, 1, 1.7415};
            return adjust_rect(rect, a);
        }
    }
    return rect;
}

static int hook_preserve_type_themes(const Game *g, int type) { /* should_preserve_type_themes: leaper.cpp:91-93, plunder.cpp:83-85, heist.cpp:37-39 */
    return (g->game_id == GAME_LEAPER && type == PLAYER) || (g->game_id == GAME_PLUNDER && type == PL_SHIP) ||
           (g->game_id == GAME_HEIST && (type == HS_KEY || type == HS_LOCKED_DOOR));
}
static const Img *generated_asset(Game *g, int type);
static void draw_image(Game *g, uint32_t *dst, RectD base_rect, float rotation, int is_reflected, int base_type, int theme, float alpha, float tile_ratio) { /* BAG:877-913 */
    int img_type = hook_image_for_type(g, base_type);
    if (img_type < 0) return;
    if (g->opt.use_monochrome_assets || img_type >= USE_ASSET_THRESHOLD) {
        if (g->game_id == GAME_CHASER && img_type == CH_ORB) { /* draw_grid_obj override chaser.cpp:112-119 */
            RectD o = {base_rect.x + base_rect.w * (1 - CH_ORB_DIM) / 2, base_rect.y + base_rect.h * (1 - CH_ORB_DIM) / 2, base_rect.w * CH_ORB_DIM, base_rect.h * CH_ORB_DIM};
            fill_rect(dst, o, 0xff00ff00u);
            return;
        }
        if (img_type == SPACE) return; /* draw_grid_obj BAG:915-919 */
        if (!g->opt.use_monochrome_assets) fatal("fassert(false) color_for_type BAG:477");
        { /* color_for_type BAG:455-481 */
            int th = theme;
            if (g->opt.restrict_themes && !hook_preserve_type_themes(g, img_type)) th = 0;
            int k = 4, kcubed = k * k * k, chunk = 256 / k;
            if (!(img_type < kcubed)) fatal("fassert type < kcubed (BAG:465)");
            int new_type = (29 * (img_type + 1)) % kcubed;
            new_type = (new_type + 19 * th) % kcubed;
            uint32_t cr = (uint32_t)(chunk * (new_type / (k * k) + 1) - 1), cg = (uint32_t)(chunk * ((new_type / k) % k + 1) - 1), cb = (uint32_t)(chunk * (new_type % k + 1) - 1);
            fill_rect(dst, base_rect, 0xff000000u | (cr << 16) | (cg << 8) | cb);
        }
        return;
    }
    if (theme >= MAX_IMAGE_THEMES) fatal("fassert theme < MAX_IMAGE_THEMES (BAG:888)");
    RectD adjusted = hook_adjusted_image_rect(g, img_type, base_rect);
    int mt = theme; /* mask_theme_if_necessary BAG:450-453 (restrict_themes) */
    if (g->opt.restrict_themes && !hook_preserve_type_themes(g, img_type)) mt = 0; /* should_preserve_type_themes leaper.cpp:91-93 */
    if (!g->opt.use_generated_assets && g->assets->type_num_themes[img_type] <= mt) fatal("asset theme out of range");
    const Img *img = g->opt.use_generated_assets ? generated_asset(g, img_type) : &g->assets->img[g->assets->type_theme_img[img_type][mt]];
    if (rotation == 0) tile_image(dst, img, is_reflected, adjusted, tile_ratio, alpha);
    else draw_image_rotated(dst, img, is_reflected, adjusted, rotation, alpha);
}

static void draw_entities(Game *g, uint32_t *dst, int render_z) { /* BAG:1052-1066 */
    for (int i = 0; i < g->n_ents; i++) {
        const Ent *m = &g->pool[g->ents[i]];
        if (m->render_z != render_z) continue;
        if (g->game_id == GAME_BOSSFIGHT && m->type == BF2_SHIELDS && !g->shields_are_up) continue; /* bossfight.cpp:122-127 */
        if (g->game_id == GAME_HEIST && m->type == HS_KEY_ON_RING && !g->has_keys[m->image_theme]) continue; /* should_draw_entity heist.cpp:70-75, BAG:1055 */
        RectD r1; /* get_object_rect BAG:811-817 */
        if (m->use_abs_coords) {
            float vd = g->view_dim;
            r1.x = (vd * (m->x - m->rx)) * g->unit;
            r1.y = (vd * (m->y + m->ry)) * g->unit;
            r1.w = (2 * vd * m->rx) * g->unit;
            r1.h = (2 * vd * m->ry) * g->unit;
        } else {
            r1 = get_screen_rect(g, m->x - m->rx, m->y + m->ry, 2 * m->rx, 2 * m->ry, 0);
        }
        float tile_ratio = 0; /* get_tile_aspect_ratio BAG:409-411 */
        if (g->game_id == GAME_LEAPER && m->type == LP_FINISH_LINE) tile_ratio = 1; /* leaper.cpp:69-75 */
        if (g->game_id == GAME_DODGEBALL && m->type == DB_LAVA_WALL) tile_ratio = (float)(m->rx > m->ry ? 1 : -1); /* dodgeball.cpp:241-248 */
        if (g->game_id == GAME_FRUITBOT) { /* fruitbot.cpp:87-94 */
            if (m->type == FB_BARRIER) tile_ratio = 1;
            else if (m->type == FB_LOCKED_DOOR) tile_ratio = FB_DOOR_ASPECT_RATIO;
        }
        draw_image(g, dst, r1, m->rotation, m->is_reflected, m->image_type, m->image_theme, m->alpha, tile_ratio);
    }
}

/* QRasterPaintEngine::drawEllipse on an integer-aligned rect without antialiasing: drawEllipse_midpoint_i +
 * drawEllipsePoints (qpaintengine_raster.cpp, Qt 5.9.7), pen of width <= 1 (outline spans) and/or brush (fill spans).
 * Third-party algorithm restated; pinned with tests/tools/qt_compass_probe.py. */
typedef struct { uint32_t *dst; int w, h; int source_mode; } QtCanvas; /* source_mode: CompositionMode_Source (assetgen.cpp:158) */
static void qc_span(const QtCanvas *c, int y, int x0, int x1, uint32_t px) { /* [x0, x1) clipped, premultiplied px */
    if (y < 0 || y >= c->h) return;
    if (x0 < 0) x0 = 0;
    if (x1 > c->w) x1 = c->w;
    for (int x = x0; x < x1; x++) {
        uint32_t *d = &c->dst[y * c->w + x];
        *d = c->source_mode ? px : px + byte_mul(*d, 255u - (px >> 24));
    }
}
typedef struct { const QtCanvas *cv; int rx, ry, rw, rh, pen, brush; uint32_t pen_px, brush_px; } EllipseCtx;
static void ell_span(const EllipseCtx *c, int sx, int sy, int len, uint32_t px) { qc_span(c->cv, sy, sx, sx + len, px); }
static void ell_points(const EllipseCtx *c, int px, int py, int length) {
    if (length == 0) return;
    int midx = c->rx + (c->rw + 1) / 2, midy = c->ry + (c->rh + 1) / 2;
    int x = px + midx, y = midy - py;
    int o0x = midx + midx - x - (length - 1) - (c->rw & 1);
    int o0len = length < x - o0x ? length : x - o0x;
    int o2y = midy + midy - y - (c->rh & 1);
    if (o0x + o0len < x) {
        int f0x = o0x + o0len - 1, f0len = x - f0x > 0 ? x - f0x : 0;
        if (c->brush) {
            ell_span(c, f0x, y, f0len, c->brush_px);
            if (!(y >= o2y)) ell_span(c, f0x, o2y, f0len, c->brush_px);
        }
    }
    if (c->pen) {
        ell_span(c, o0x, y, o0len, c->pen_px);
        ell_span(c, x, y, length, c->pen_px);
        if (!(y >= o2y)) {
            ell_span(c, o0x, o2y, o0len, c->pen_px);
            ell_span(c, x, o2y, length, c->pen_px);
        }
    }
}
static void draw_ellipse_i(const QtCanvas *cv, int rx, int ry, int rw, int rh, int pen, uint32_t pen_px, int brush, uint32_t brush_px) {
    if (rw <= 0 || rh <= 0) return;
    EllipseCtx c = {cv, rx, ry, rw, rh, pen, brush, pen_px, brush_px};
    double a = rw / 2.0, b = rh / 2.0;
    double d = b * b - (a * a * b) + 0.25 * a * a;
    int x = 0, y = (rh + 1) / 2, startx = x;
    while (a * a * (2 * y - 1) > 2 * b * b * (x + 1)) {
        if (d < 0) {
            d += b * b * (2 * x + 3);
            ++x;
        } else {
            d += b * b * (2 * x + 3) + a * a * (-2 * y + 2);
            ell_points(&c, startx, y, x - startx + 1);
            startx = ++x;
            --y;
        }
    }
    ell_points(&c, startx, y, x - startx + 1);
    d = b * b * (x + 0.5) * (x + 0.5) + a * a * ((y - 1) * (y - 1) - b * b);
    int miny = rh & 1;
    while (y > miny) {
        if (d < 0) {
            d += b * b * (2 * x + 2) + a * a * (-2 * y + 3);
            ++x;
        } else {
            d += a * a * (-2 * y + 3);
        }
        --y;
        ell_points(&c, x, y, 1);
    }
}
/* QCosmeticStroker::drawLine (qcosmeticstroker.cpp, Qt 5.9.7) for a solid width-0 pen with square caps and integer
 * end points: 26.6 end points, 16.16 minor-axis walker, half-pixel cap extension at both ends. */
static int tdiv_i64(long long a, long long b) { return (int)(a / b); }
static void draw_line_cosmetic(uint32_t *dst, int X1, int Y1, int X2, int Y2, uint32_t px) {
#define PUT(xx, yy) do { if ((xx) >= 0 && (xx) < RES_W && (yy) >= 0 && (yy) < RES_H) dst[(yy) * RES_W + (xx)] = px; } while (0)
    if (X1 == X2 && Y1 == Y2) { /* QPainter::drawLine of a point with caps: one pixel */
        PUT(X1, Y1);
        return;
    }
    int x1 = X1 * 64, y1 = Y1 * 64, x2 = X2 * 64, y2 = Y2 * 64;
    int dx = abs(x2 - x1), dy = abs(y2 - y1);
    if (dx < dy) {
        if (y1 > y2) { int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
        int xinc = tdiv_i64((long long)(x2 - x1) * 65536, y2 - y1);
        int x = x1 * 1024;
        y1 -= 32; x -= xinc >> 1; y2 += 32;
        int y = (y1 + 32) >> 6, ys = (y2 + 32) >> 6, rnd = xinc > 0 ? 32 : 0;
        if (y != ys) {
            x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
            do {
                PUT(x >> 16, y);
                x += xinc;
            } while (++y < ys);
        }
    } else {
        if (!dx) return;
        if (x1 > x2) { int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
        int yinc = tdiv_i64((long long)(y2 - y1) * 65536, x2 - x1);
        int y = y1 * 1024;
        x1 -= 32; y -= yinc >> 1; x2 += 32;
        int x = (x1 + 32) >> 6, xs = (x2 + 32) >> 6, rnd = yinc > 0 ? 32 : 0;
        if (x != xs) {
            y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
            do {
                PUT(x, y >> 16);
                y += yinc;
            } while (++x < xs);
        }
    }
#undef PUT
}
/* ---- QPainter::drawEllipse(QRectF) on a rect that is NOT integer aligned, no antialiasing, identity transform (Qt 5.9.7).
 * QRasterPaintEngine::drawEllipse falls through to QPaintEngineEx::drawEllipse: qt_curves_for_arc(rect, 0, -360) gives a
 * start point and four cubics (QT_PATH_KAPPA), then draw(path) = fill with the brush + stroke with the pen.
 *   brush: QRasterPaintEngine::fill(QVectorPath): culled unless controlPointRect().toRect() -- 5.9's toRect rounds x, y, w
 *          and h separately -- intersects the device rect; QOutlineMapper::curveTo flattens every cubic with
 *          QBezier::addToPolygon (threshold .25), points become 26.6 by qRound(v * 64); QRasterizer's QScanConverter walks
 *          each line in 16.16 and samples it at the pixel centres; a row is filled from its first to its second crossing.
 *   pen  : width <= 1 => QCosmeticStroker::drawPath: calculateLastPoint on the closed path's last two points, renderCubic
 *          (<= 6 levels of subdivision), drawLine<drawPixel, NoDasher> with its duplicate / drop-out control between
 *          consecutive segments (incl. the horizontal branch repeating the vertical branch's half-step test).
 * Third-party algorithm (source not on disk); pinned with tests/tools/qt_path_probe.py against PyQt5 5.9.7: tens of
 * thousands of random, knife-edge, tiny and partly-outside rects, pen / brush / both, 0 misses.  Call sites:
 * jumper.cpp:137-142 (compass in easy mode / without center_agent), assetgen.cpp:99-105. */
#define QT_PATH_KAPPA 0.5522847498
typedef struct { double x1, y1, x2, y2, x3, y3, x4, y4; } QBez;
static void qbez_split(const QBez *b, QBez *first, QBez *second) { /* qbezier_p.h QBezier::split */
    double c = (b->x2 + b->x3) * .5;
    first->x2 = (b->x1 + b->x2) * .5;
    second->x3 = (b->x3 + b->x4) * .5;
    first->x1 = b->x1;
    second->x4 = b->x4;
    first->x3 = (first->x2 + c) * .5;
    second->x2 = (second->x3 + c) * .5;
    first->x4 = second->x1 = (first->x3 + second->x2) * .5;
    c = (b->y2 + b->y3) / 2;
    first->y2 = (b->y1 + b->y2) * .5;
    second->y3 = (b->y3 + b->y4) * .5;
    first->y1 = b->y1;
    second->y4 = b->y4;
    first->y3 = (first->y2 + c) * .5;
    second->y2 = (second->y3 + c) * .5;
    first->y4 = second->y1 = (first->y3 + second->y2) * .5;
}
#define QT_MAX_FLAT 2100 /* 4 cubics x at most 2^9 segments */
static int qbez_add_to_polygon(QBez bz, double *px, double *py, int n, double thr) { /* qbezier.cpp QBezier::addToPolygon */
    QBez beziers[10];
    int levels[10];
    beziers[0] = bz;
    levels[0] = 9;
    int top = 0;
    while (top >= 0) {
        QBez *b = &beziers[top];
        double y4y1 = b->y4 - b->y1, x4x1 = b->x4 - b->x1;
        double l = fabs(x4x1) + fabs(y4y1), d;
        if (l > 1.) {
            d = fabs(x4x1 * (b->y1 - b->y2) - y4y1 * (b->x1 - b->x2)) + fabs(x4x1 * (b->y1 - b->y3) - y4y1 * (b->x1 - b->x3));
        } else {
            d = fabs(b->x1 - b->x2) + fabs(b->y1 - b->y2) + fabs(b->x1 - b->x3) + fabs(b->y1 - b->y3);
            l = 1.;
        }
        if (d < thr * l || levels[top] == 0) {
            if (n >= QT_MAX_FLAT) fatal("flattened path overflow");
            px[n] = b->x4;
            py[n] = b->y4;
            n++;
            --top;
        } else {
            QBez whole = *b;
            qbez_split(&whole, b + 1, b);
            levels[top + 1] = --levels[top];
            ++top;
        }
    }
    return n;
}
static void qt_arc_points(RectD r, double *ax, double *ay) { /* qpainterpath.cpp qt_curves_for_arc(rect, 0, -360): 13 points */
    double x = r.x, y = r.y, w = r.w, w2 = r.w / 2, w2k = w2 * QT_PATH_KAPPA, h = r.h, h2 = r.h / 2, h2k = h2 * QT_PATH_KAPPA;
    const double X[13] = {x + w, x + w, x + w2 + w2k, x + w2, x + w2 - w2k, x, x, x, x + w2 - w2k, x + w2, x + w2 + w2k, x + w, x + w};
    const double Y[13] = {y + h2, y + h2 + h2k, y + h, y + h, y + h, y + h2 + h2k, y + h2, y + h2 - h2k, y, y, y, y + h2 - h2k, y + h2};
    for (int i = 0; i < 13; i++) { ax[i] = X[i]; ay[i] = Y[i]; }
}
static void qt_fill_ellipse_path(const QtCanvas *c, RectD r, uint32_t px) {
    double ax[13], ay[13];
    qt_arc_points(r, ax, ay);
    { /* QRasterPaintEngine::fill: controlPointRect().toRect() must intersect the device rect (QRect::intersects) */
        double l = r.x, rr = r.x + r.w, t = r.y, b = r.y + r.h;
        int x1 = q_round(l), y1 = q_round(t), x2 = x1 + q_round(rr - l) - 1, y2 = y1 + q_round(b - t) - 1;
        if (x2 == x1 - 1 && y2 == y1 - 1) return;
        if (x1 > c->w - 1 || 0 > x2 || y1 > c->h - 1 || 0 > y2) return;
    }
    static double fx[QT_MAX_FLAT + 2], fy[QT_MAX_FLAT + 2];
    int n = 0;
    fx[n] = ax[0]; fy[n] = ay[0]; n++;
    for (int k = 0; k < 4; k++) {
        QBez b = {fx[n - 1], fy[n - 1], ax[1 + 3 * k], ay[1 + 3 * k], ax[2 + 3 * k], ay[2 + 3 * k], ax[3 + 3 * k], ay[3 + 3 * k]};
        n = qbez_add_to_polygon(b, fx, fy, n, 0.25);
    }
    if (fx[n - 1] != fx[0] || fy[n - 1] != fy[0]) { fx[n] = fx[0]; fy[n] = fy[0]; n++; } /* closeSubpath */
    static int qx[QT_MAX_FLAT + 2], qy[QT_MAX_FLAT + 2];
    int min_y = 0, max_y = 0;
    for (int i = 0; i < n; i++) {
        qx[i] = q_round(fx[i] * 64);
        qy[i] = q_round(fy[i] * 64);
        if (i == 0 || qy[i] < min_y) min_y = qy[i];
        if (i == 0 || qy[i] > max_y) max_y = qy[i];
    }
    int top = (min_y + 32) >> 6, bot = (max_y - 32) >> 6; /* QRasterizer::rasterize */
    if (top < 0) top = 0;
    if (bot > c->h - 1) bot = c->h - 1;
    if (top > bot) return;
    /* a convex outline crosses a row's centre line twice: keep both crossings per row (odd-even rule) */
    int *cnt = (int *)calloc((size_t)c->h, sizeof(int)), *xa = (int *)calloc((size_t)c->h, sizeof(int)), *xb = (int *)calloc((size_t)c->h, sizeof(int));
    for (int i = 0; i + 1 < n; i++) { /* QScanConverter::mergeLine */
        int a_x = qx[i], a_y = qy[i], b_x = qx[i + 1], b_y = qy[i + 1];
        if (a_y > b_y) { int t = a_x; a_x = b_x; b_x = t; t = a_y; a_y = b_y; b_y = t; }
        int itop = (a_y + 32) >> 6, ibot = (b_y - 32) >> 6;
        if (itop < top) itop = top;
        if (ibot > bot) ibot = bot;
        if (itop > ibot) continue;
        int xfp = 32768 + a_x * 1024, slope = 0;
        if (b_x != a_x) {
            double s = (b_x - a_x) / (double)(b_y - a_y);
            slope = (int)(s * 65536.);
            xfp += (int)(((long long)slope * (long long)((itop << 16) + 32768 - (a_y << 10))) >> 16);
        }
        for (int y = itop; y <= ibot; y++) {
            int xi = xfp >> 16;
            if (cnt[y] == 0) xa[y] = xi;
            else if (cnt[y] == 1) xb[y] = xi;
            else fatal("more than two crossings on a row of an ellipse outline");
            cnt[y]++;
            xfp += slope;
        }
    }
    for (int y = top; y <= bot; y++)
        if (cnt[y] == 2) qc_span(c, y, xa[y] < xb[y] ? xa[y] : xb[y], xa[y] < xb[y] ? xb[y] : xa[y], px);
    free(cnt); free(xa); free(xb);
}
/* QCosmeticStroker (qcosmeticstroker.cpp): state carried from segment to segment */
enum { QCS_TB = 1, QCS_BT = 2, QCS_LR = 4, QCS_RL = 8 };
typedef struct { const QtCanvas *c; uint32_t px; int last_dir, last_x, last_y, last_axis_aligned; double xmin, xmax, ymin, ymax; } QCosmetic;
static int qcs_fixdiv(int x, int y) { return (int)(((long long)x << 16) / y); } /* F16Dot16FixedDiv */
static int qcs_clip_line(QCosmetic *s, double *x1, double *y1, double *x2, double *y2) { /* QCosmeticStroker::clipLine */
    if (*x1 < s->xmin) {
        if (*x2 <= s->xmin) goto clipped;
        *y1 += (*y2 - *y1) / (*x2 - *x1) * (s->xmin - *x1);
        *x1 = s->xmin;
    } else if (*x1 > s->xmax) {
        if (*x2 >= s->xmax) goto clipped;
        *y1 += (*y2 - *y1) / (*x2 - *x1) * (s->xmax - *x1);
        *x1 = s->xmax;
    }
    if (*x2 < s->xmin) {
        s->last_x = INT_MIN;
        *y2 += (*y2 - *y1) / (*x2 - *x1) * (s->xmin - *x2);
        *x2 = s->xmin;
    } else if (*x2 > s->xmax) {
        s->last_x = INT_MIN;
        *y2 += (*y2 - *y1) / (*x2 - *x1) * (s->xmax - *x2);
        *x2 = s->xmax;
    }
    if (*y1 < s->ymin) {
        if (*y2 <= s->ymin) goto clipped;
        *x1 += (*x2 - *x1) / (*y2 - *y1) * (s->ymin - *y1);
        *y1 = s->ymin;
    } else if (*y1 > s->ymax) {
        if (*y2 >= s->ymax) goto clipped;
        *x1 += (*x2 - *x1) / (*y2 - *y1) * (s->ymax - *y1);
        *y1 = s->ymax;
    }
    if (*y2 < s->ymin) {
        s->last_x = INT_MIN;
        *x2 += (*x2 - *x1) / (*y2 - *y1) * (s->ymin - *y2);
        *y2 = s->ymin;
    } else if (*y2 > s->ymax) {
        s->last_x = INT_MIN;
        *x2 += (*x2 - *x1) / (*y2 - *y1) * (s->ymax - *y2);
        *y2 = s->ymax;
    }
    return 0;
clipped:
    s->last_x = INT_MIN;
    return 1;
}
static void qcs_calculate_last_point(QCosmetic *s, double rx1, double ry1, double rx2, double ry2) {
    s->last_x = INT_MIN;
    s->last_y = INT_MIN;
    if (qcs_clip_line(s, &rx1, &ry1, &rx2, &ry2)) return;
    int x1 = (int)(rx1 * 64.), y1 = (int)(ry1 * 64.), x2 = (int)(rx2 * 64.), y2 = (int)(ry2 * 64.);
    int dx = abs(x2 - x1), dy = abs(y2 - y1);
    if (dx < dy) {
        int swapped = 0;
        if (y1 > y2) { swapped = 1; int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
        int xinc = qcs_fixdiv(x2 - x1, y2 - y1);
        int x = x1 * 1024, y = (y1 + 32) >> 6, ys = (y2 + 32) >> 6, rnd = xinc > 0 ? 32 : 0;
        if (y != ys) {
            x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
            if (swapped) {
                s->last_x = x >> 16; s->last_y = y; s->last_dir = QCS_BT;
            } else {
                s->last_x = (x + (ys - y - 1) * xinc) >> 16; s->last_y = ys - 1; s->last_dir = QCS_TB;
            }
            s->last_axis_aligned = abs(xinc) < (1 << 14);
        }
    } else {
        if (!dx) return;
        int swapped = 0;
        if (x1 > x2) { swapped = 1; int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
        int yinc = qcs_fixdiv(y2 - y1, x2 - x1);
        int y = y1 * 1024, x = (x1 + 32) >> 6, xs = (x2 + 32) >> 6, rnd = yinc > 0 ? 32 : 0;
        if (x != xs) {
            y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
            if (swapped) {
                s->last_x = x; s->last_y = y >> 16; s->last_dir = QCS_RL;
            } else {
                s->last_x = xs - 1; s->last_y = (y + (xs - x - 1) * yinc) >> 16; s->last_dir = QCS_LR;
            }
            s->last_axis_aligned = abs(yinc) < (1 << 14);
        }
    }
}
static void qcs_line(QCosmetic *s, double rx1, double ry1, double rx2, double ry2, int caps) { /* drawLine<drawPixel, NoDasher> */
    if (qcs_clip_line(s, &rx1, &ry1, &rx2, &ry2)) return;
    int x1 = (int)(rx1 * 64.), y1 = (int)(ry1 * 64.), x2 = (int)(rx2 * 64.), y2 = (int)(ry2 * 64.);
    int dx = abs(x2 - x1), dy = abs(y2 - y1);
    int last_x = s->last_x, last_y = s->last_y; /* QCosmeticStroker::Point last = stroker->lastPixel */
    const int lpx = s->last_x, lpy = s->last_y;
    if (dx < dy) {
        int dir = QCS_TB, swapped = 0;
        if (y1 > y2) {
            swapped = 1;
            int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t;
            caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
            dir = QCS_BT;
        }
        int xinc = qcs_fixdiv(x2 - x1, y2 - y1);
        int x = x1 * 1024;
        if ((s->last_dir ^ 3) == dir) caps |= swapped ? 2 : 1; /* turned around: cap towards the previous segment */
        if (caps & 1) { y1 -= 32; x -= xinc >> 1; } /* capAdjust */
        if (caps & 2) y2 += 32;
        int y = (y1 + 32) >> 6, ys = (y2 + 32) >> 6, rnd = xinc > 0 ? 32 : 0;
        if ((caps & 1) && lpy == y + 1) y++; /* "capAdjust made us round away from what calculateLastPoint gave us" */
        if (y != ys) {
            x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
            int first_x = x >> 16, first_y = y;
            last_x = (x + (ys - y - 1) * xinc) >> 16;
            last_y = ys - 1;
            if (swapped) { int t = first_x; first_x = last_x; last_x = t; t = first_y; first_y = last_y; last_y = t; }
            int axis_aligned = abs(xinc) < (1 << 14);
            if (lpx > INT_MIN) {
                if (first_x == lpx && first_y == lpy) { /* remove duplicated pixel */
                    if (swapped) --ys;
                    else { ++y; x += xinc; }
                } else if (s->last_dir != dir && ((axis_aligned && s->last_axis_aligned && lpx != first_x && lpy != first_y) || (abs(lpx - first_x) > 1 || abs(lpy - first_y) > 1))) { /* have a missing pixel, insert it */
                    if (swapped) ++ys;
                    else { --y; x -= xinc; }
                } else if (s->last_dir == dir && (abs(lpx - first_x) <= 1 && abs(lpy - first_y) > 1)) {
                    x += xinc >> 1;
                    if (swapped) last_x = x >> 16;
                    else last_x = (x + (ys - y - 1) * xinc) >> 16;
                }
            }
            s->last_dir = dir;
            s->last_axis_aligned = axis_aligned;
            do {
                qc_span(s->c, y, x >> 16, (x >> 16) + 1, s->px);
                x += xinc;
            } while (++y < ys);
        }
    } else {
        if (!dx) return;
        int dir = QCS_LR, swapped = 0;
        if (x1 > x2) {
            swapped = 1;
            int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t;
            caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
            dir = QCS_RL;
        }
        int yinc = qcs_fixdiv(y2 - y1, x2 - x1);
        int y = y1 * 1024;
        if ((s->last_dir ^ 0xc) == dir) caps |= swapped ? 2 : 1;
        if (caps & 1) { x1 -= 32; y -= yinc >> 1; }
        if (caps & 2) x2 += 32;
        int x = (x1 + 32) >> 6, xs = (x2 + 32) >> 6, rnd = yinc > 0 ? 32 : 0;
        if ((caps & 1) && lpx == x + 1) x++;
        if (x != xs) {
            y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
            int first_x = x, first_y = y >> 16;
            last_x = xs - 1;
            last_y = (y + (xs - x - 1) * yinc) >> 16;
            if (swapped) { int t = first_x; first_x = last_x; last_x = t; t = first_y; first_y = last_y; last_y = t; }
            int axis_aligned = abs(yinc) < (1 << 14);
            if (lpx > INT_MIN) {
                if (first_x == lpx && first_y == lpy) {
                    if (swapped) --xs;
                    else { ++x; y += yinc; }
                } else if (s->last_dir != dir && ((axis_aligned && s->last_axis_aligned && lpx != first_x && lpy != first_y) || (abs(lpx - first_x) > 1 || abs(lpy - first_y) > 1))) {
                    if (swapped) ++xs;
                    else { --x; y -= yinc; }
                } else if (s->last_dir == dir && (abs(lpx - first_x) <= 1 && abs(lpy - first_y) > 1)) { /* sic: the vertical branch's test (probe) */
                    y += yinc >> 1;
                    if (swapped) last_y = y >> 16;
                    else last_y = (y + (xs - x - 1) * yinc) >> 16;
                }
            }
            s->last_dir = dir;
            s->last_axis_aligned = axis_aligned;
            do {
                qc_span(s->c, y >> 16, x, x + 1, s->px);
                y += yinc;
            } while (++x < xs);
        }
    }
    s->last_x = last_x;
    s->last_y = last_y;
}
static void qcs_cubic_sub(QCosmetic *s, double *px, double *py, int level, int caps) { /* renderCubicSubdivision; points[3] = start ... points[0] = end */
    if (level) {
        double dx = px[3] - px[0], dy = py[3] - py[0];
        double len = .25 * (fabs(dx) + fabs(dy));
        if (fabs(dx * (py[0] - py[2]) - dy * (px[0] - px[2])) >= len || fabs(dx * (py[0] - py[1]) - dy * (px[0] - px[1])) >= len) {
            for (int k = 0; k < 2; k++) { /* splitCubic */
                double *p = k ? py : px, a, b, c, d;
                p[6] = p[3];
                c = p[1];
                d = p[2];
                p[1] = a = (p[0] + c) * .5;
                p[5] = b = (p[3] + d) * .5;
                c = (c + d) * .5;
                p[2] = a = (a + c) * .5;
                p[4] = b = (b + c) * .5;
                p[3] = (a + b) * .5;
            }
            --level;
            qcs_cubic_sub(s, px + 3, py + 3, level, caps & 1);
            qcs_cubic_sub(s, px, py, level, caps & 2);
            return;
        }
    }
    qcs_line(s, px[3], py[3], px[0], py[0], caps);
}
static void qt_stroke_ellipse_path(const QtCanvas *c, RectD r, uint32_t px) { /* QCosmeticStroker::drawPath on the closed 4-cubic path */
    double ax[13], ay[13];
    qt_arc_points(r, ax, ay);
    QCosmetic s = {c, px, QCS_LR, INT_MIN, INT_MIN, 0, -1., c->w + 1., -1., c->h + 1.}; /* setup(): device rect widened by one pixel */
    qcs_calculate_last_point(&s, ax[11], ay[11], ax[12], ay[12]);
    for (int k = 0; k < 4; k++) {
        double px_[3 * 6 + 4], py_[3 * 6 + 4];
        px_[3] = ax[3 * k]; py_[3] = ay[3 * k];
        px_[2] = ax[3 * k + 1]; py_[2] = ay[3 * k + 1];
        px_[1] = ax[3 * k + 2]; py_[1] = ay[3 * k + 2];
        px_[0] = ax[3 * k + 3]; py_[0] = ay[3 * k + 3];
        qcs_cubic_sub(&s, px_, py_, 6, 0);
    }
}
/* QRasterPaintEngine::drawEllipse: the midpoint algorithm when the rect is integer aligned, else the path route */
static void draw_ellipse_f(const QtCanvas *c, RectD r, int pen, uint32_t pen_px, int brush, uint32_t brush_px) { /* qpaintengine_raster.cpp QRasterPaintEngine::drawEllipse */
    if (r.w < 0) { r.x += r.w; r.w = -r.w; } /* QPainter::drawEllipse: rect.normalized() */
    if (r.h < 0) { r.y += r.h; r.h = -r.h; }
    if ((r.w > r.h ? r.w : r.h) < 32767 && r.w > 0 && r.h > 0) {
        int bx = (int)r.x, by = (int)r.y, bw = (int)r.w, bh = (int)r.h;
        if ((double)bx == r.x && (double)by == r.y && (double)bw == r.w && (double)bh == r.h) {
            draw_ellipse_i(c, bx, by, bw, bh, pen, pen_px, brush, brush_px);
            return;
        }
    }
    if (r.w == 0 && r.h == 0) return; /* qt_curves_for_arc: rect.isNull() */
    if (brush) qt_fill_ellipse_path(c, r, brush_px);
    if (pen) qt_stroke_ellipse_path(c, r, pen_px);
}

/* ---- AssetGen: reference src/assetgen.cpp (procedurally painted sprites and backgrounds, use_generated_assets) ----
 * The painter calls it makes: fillRect(QRectF, opaque QColor) = [qRound) box (fill_rect above); fillRect with alpha 200
 * = premultiplied through QRgba64 (qrgba64.h) and SourceOver; drawEllipse(QRectF) with brush c1 and a width-1 pen c2 =
 * draw_ellipse_f (midpoint or path route).  Colours are opaque, so SourceOver and (paint_shape_resource's) Source mode
 * both overwrite; a shape asset starts from transparent black.  Float / double promotions follow the reference expressions. */
typedef struct { Rng *rng; float rgb_start[3], rgb_len[3], p_rect; } ColorGen;
static void cg_roll(ColorGen *c) { /* assetgen.cpp:10-20 */
    for (int i = 0; i < 3; i++) c->rgb_len[i] = rng_rand01(c->rng);
    for (int i = 0; i < 3; i++) c->rgb_start[i] = rng_rand01(c->rng) * (1 - c->rgb_len[i]);
    c->p_rect = rng_rand01(c->rng);
}
static uint32_t cg_rand_color(ColorGen *c) { /* assetgen.cpp:22-28 -> 0xffRRGGBB */
    int ch[3];
    for (int i = 0; i < 3; i++) ch[i] = (int)(255 * (rng_rand01(c->rng) * c->rgb_len[i] + c->rgb_start[i]));
    return 0xff000000u | ((uint32_t)(ch[0] & 0xff) << 16) | ((uint32_t)(ch[1] & 0xff) << 8) | (uint32_t)(ch[2] & 0xff);
}
static void canvas_fill(const QtCanvas *cv, RectD r, uint32_t px, int over) { /* QPainter::fillRect(QRectF, QColor), no antialiasing */
    int x1 = q_round(r.x), x2 = q_round(r.x + r.w), y1 = q_round(r.y), y2 = q_round(r.y + r.h);
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; }
    if (y2 < y1) { int t = y1; y1 = y2; y2 = t; }
    if (y1 < 0) y1 = 0;
    if (y2 > cv->h) y2 = cv->h;
    if (x1 < 0) x1 = 0;
    if (x2 > cv->w) x2 = cv->w;
    for (int y = y1; y < y2; y++)
        for (int x = x1; x < x2; x++) {
            uint32_t *d = &cv->dst[y * cv->w + x];
            *d = over ? px + byte_mul(*d, 255u - (px >> 24)) : px;
        }
}
/* QColor(r, g, b, 200) as the raster engine hands it to the span filler: qPremultiply(QRgba64).toArgb32() (qrgba64.h) */
static uint32_t premul_alpha200(uint32_t rgb) {
    const uint32_t a16 = 200u * 257u;
    uint32_t out = 0;
    for (int sh = 16; sh >= 0; sh -= 8) {
        const uint32_t c16 = ((rgb >> sh) & 0xffu) * 257u;
        uint32_t x = c16 * a16;
        x = (x + (x >> 16) + 0x8000u) >> 16;  /* div_65535 */
        x += 128;                              /* div_257 */
        x = (x - (x >> 8)) >> 8;
        out |= x << sh;
    }
    uint32_t al = a16 + 128;
    al = (al - (al >> 8)) >> 8;
    return out | (al << 24);
}
typedef struct { Rng *rng; const QtCanvas *cv; } AssetGen;
static RectD ag_choose_sub_rect(AssetGen *ag, RectD rect, float min_dim, float max_dim) { /* assetgen.cpp:35-52 */
    int w = (int)rect.w, h = (int)rect.h;
    int smaller = (w > h) ? h : w;
    float del_dim = max_dim - min_dim;
    float rdx = (rng_rand01(ag->rng) * del_dim + min_dim) * smaller;
    float rdy = (rng_rand01(ag->rng) * del_dim + min_dim) * smaller;
    float rx_off = rng_rand01(ag->rng) * (w - rdx);
    float ry_off = rng_rand01(ag->rng) * (h - rdy);
    RectD d = {rx_off + rect.x, ry_off + rect.y, rdx, rdy};
    return d;
}
static void ag_paint_shape(AssetGen *ag, RectD main_rect, ColorGen *cgen) { /* assetgen.cpp:77-107 */
    int k = rng_randn(ag->rng, 10);
    int num_splits = (k * k) / 50 + 1;
    int is_horizontal = rng_rand01(ag->rng) > .5;
    /* split_rect assetgen.cpp:54-75 */
    float x = (float)main_rect.x, y = (float)main_rect.y, w = (float)main_rect.w, h = (float)main_rect.h;
    float dw = w / num_splits, dh = h / num_splits;
    int use_rect = rng_rand01(ag->rng) > .5;
    int regen_colors = rng_rand01(ag->rng) > .5;
    uint32_t c1 = cg_rand_color(cgen);
    uint32_t c2 = cg_rand_color(cgen);
    for (int i = 0; i < num_splits; i++) {
        RectD rect;
        if (is_horizontal) { rect.x = x + i * dw; rect.y = y; rect.w = dw; rect.h = h; }
        else { rect.x = x; rect.y = y + i * dh; rect.w = w; rect.h = dh; }
        if (regen_colors) {
            c1 = cg_rand_color(cgen);
            c2 = cg_rand_color(cgen);
        }
        if (use_rect) canvas_fill(ag->cv, rect, c1, 0);
        else draw_ellipse_f(ag->cv, rect, 1, c2, 1, c1);
    }
}
static void ag_paint_rect_resource(AssetGen *ag, RectD rect, int num_recurse, int blotch_scale) { /* assetgen.cpp:109-138 */
    ColorGen cgen;
    cgen.rng = ag->rng;
    cg_roll(&cgen);
    uint32_t bgcolor = cg_rand_color(&cgen);
    canvas_fill(ag->cv, rect, bgcolor, 0);
    float scale = (float)(.3 + .7 * rng_rand01(ag->rng));
    float max_rand_dim = (float)(.5 * scale);
    float min_rand_dim = (float)(.05 * scale);
    int num_blotches = rng_randint(ag->rng, blotch_scale, 2 * blotch_scale);
    float p_recurse = (float)(rng_rand01(ag->rng) * .75);
    for (int j = 0; j < num_blotches; j++) {
        RectD dst3 = ag_choose_sub_rect(ag, rect, min_rand_dim, max_rand_dim);
        if ((num_recurse > 0) && (rng_rand01(ag->rng) < p_recurse)) ag_paint_rect_resource(ag, dst3, num_recurse - 1, 10);
        else ag_paint_shape(ag, dst3, &cgen);
    }
    canvas_fill(ag->cv, rect, premul_alpha200(bgcolor), 1);
}
static RectD ag_create_bar(AssetGen *ag, RectD rect, int is_horizontal) { /* assetgen.cpp:140-155 */
    float k1 = (float)(.45 + rng_rand01(ag->rng) * .4);
    float k2 = (float)(.45 + rng_rand01(ag->rng) * .4);
    float w = (float)(rect.w * k1 * k1);
    float h = (float)(rect.h * k2 * k2);
    float pct = rng_rand01(ag->rng);
    RectD c;
    if (is_horizontal == 0) { c.x = 0; c.y = (rect.h - h) * pct; c.w = rect.w; c.h = h; }
    else { c.x = (rect.h - w) * pct; c.y = 0; c.w = w; c.h = rect.h; }
    return c;
}
static void ag_paint_shape_resource(AssetGen *ag, RectD rect) { /* assetgen.cpp:157-190 */
    ColorGen cgen;
    cgen.rng = ag->rng;
    cg_roll(&cgen);
    int horizontal_first = rng_rand01(ag->rng) > .5;
    int nbar1 = rng_randn(ag->rng, 3) / 2 + 1;
    int nbar2 = rng_randn(ag->rng, 3) / 2 + 1;
    canvas_fill(ag->cv, rect, 0u, 0); /* CompositionMode_Source, QColor(0, 0, 0, 0) */
    for (int i = 0; i < nbar1; i++) {
        RectD c1 = ag_create_bar(ag, rect, horizontal_first);
        ag_paint_shape(ag, c1, &cgen);
    }
    for (int i = 0; i < nbar2; i++) {
        RectD c2 = ag_create_bar(ag, rect, !horizontal_first);
        ag_paint_shape(ag, c2, &cgen);
    }
    int num_blotches = rng_randint(ag->rng, 1, 5);
    for (int j = 0; j < num_blotches; j++) {
        RectD d = ag_choose_sub_rect(ag, rect, 0.1f, 0.6f);
        ag_paint_shape(ag, d, &cgen);
    }
}
static void ag_generate_resource(Rng *rng, uint32_t *px, int w, int h, int num_recurse, int blotch_scale, int is_rect) { /* assetgen.cpp:192-201 */
    const QtCanvas cv = {px, w, h, 1};
    AssetGen ag = {rng, &cv};
    RectD rect = {0, 0, (double)w, (double)h};
    if (is_rect) ag_paint_rect_resource(&ag, rect, num_recurse, blotch_scale);
    else ag_paint_shape_resource(&ag, rect);
}
static int hook_use_block_asset(int game_id, int type) { /* use_block_asset overrides, BAG:404-406 */
    switch (game_id) {
    case GAME_CAVEFLYER: return type == CF_CAVEWALL;
    case GAME_CHASER: return type == CH_MAZE_WALL;
    case GAME_CLIMBER: return type == CL_WALL_MID || type == CL_WALL_TOP;
    case GAME_COINRUN: return type == CR_WALL_MID || type == CR_WALL_TOP;
    case GAME_DODGEBALL: return type == DB_LAVA_WALL || type == DB_DOOR || type == DB_DOOR_OPEN;
    case GAME_FRUITBOT: return type == FB_BARRIER || type == FB_LOCKED_DOOR || type == FB_PRESENT;
    case GAME_HEIST: return type == WALL_OBJ || type == HS_LOCKED_DOOR;
    case GAME_JUMPER: return type == JP_CAVEWALL || type == JP_CAVEWALL_TOP;
    case GAME_LEAPER: return type == LP_WATER || type == LP_ROAD;
    case GAME_NINJA: return type == NJ_WALL_MID;
    default: return 0;
    }
}
static uint32_t hash_str_uint32(const char *str) { /* src/vecgame.cpp:156-167 (FNV-1a) */
    uint32_t hash = 0x811c9dc5u;
    for (; *str; str++) {
        hash ^= (uint8_t)*str;
        hash *= 0x1000193u;
    }
    return hash;
}
static const char *game_name(int game_id) {
    static const char *all[] = {"coinrun", "bigfish", "maze", "climber", "miner", "starpilot", "fruitbot", "leaper", "plunder", "heist", "ninja", "dodgeball", "bossfight", "chaser", "caveflyer", "jumper"};
    for (int i = 0; i < 16; i++)
        if (pgo_game_id(all[i]) == game_id) return all[i];
    fatal("unknown game id");
    return "";
}
/* initialize_asset_if_necessary BAG:79-123 with use_generated_assets: a 64 x 64 image per type, seeded with fixed_asset_seed + type
 * (the same for every theme, env and episode); aspect ratio 1, one theme */
static GameAssets g_gen_assets[16];
static const Img *generated_asset(Game *g, int type) {
    GameAssets *a = &g_gen_assets[g->game_id];
    if (type < 0 || type >= MAX_ASSETS) fatal("generated asset type out of range");
    Img *im = &a->img[type];
    if (!im->px) {
        Rng asset_rand_gen;
        rng_seed(&asset_rand_gen, (int)(hash_str_uint32(game_name(g->game_id)) + (uint32_t)type));
        im->px = (uint32_t *)malloc(sizeof(uint32_t) * 64 * 64);
        ag_generate_resource(&asset_rand_gen, im->px, 64, 64, 0, 5, hook_use_block_asset(g->game_id, type));
    }
    return im;
}
static void gen_assets_init(int game_id) {
    GameAssets *a = &g_gen_assets[game_id];
    if (a->built) return;
    a->built = 1;
    a->n = MAX_ASSETS;
    for (int t = 0; t < MAX_ASSETS; t++) {
        a->type_num_themes[t] = 1;
        for (int k = 0; k < MAX_IMAGE_THEMES; k++) a->type_theme_img[t][k] = t;
        a->img[t].w = a->img[t].h = 64;
        a->img[t].generic = 1;
    }
    a->n_bg = 1;
}
/* the background of this episode: main_bg_images_ptr->at(background_index) (BAG:990) */
static const Img *bg_image(Game *g) {
    if (g->opt.use_generated_assets) return &g->gen_bg;
    return &g->assets->img[g->assets->bg_img[g->background_index]];
}

static void jp_draw_compass(Game *g, uint32_t *dst) { /* jumper.cpp:134-169 */
    const Ent *agent = &g->pool[g->agent], *goal = &g->pool[g->goal];
    float cxf = (float)(g->view_dim - g->compass_dim - .25), cyf = (float).25;
    RectD cr_ = {cxf * g->unit, cyf * g->unit, g->compass_dim * g->unit, g->compass_dim * g->unit}; /* get_abs_rect BAG:803-805 */
    const QtCanvas cv = {dst, RES_W, RES_H, 0};
    draw_ellipse_f(&cv, cr_, 1, 0xffa8a69eu, 1, 0xffa8a69eu); /* integer aligned in hard / memory mode with center_agent, else the path route */
    float cx = (float)(cr_.x + cr_.w / 2);
    float cy = (float)(cr_.y + cr_.h / 2);
    float cr = (float)(cr_.w / 2 * .95);
    float theta = (float)atan2((double)(goal->y - agent->y), (double)(goal->x - agent->x)); /* get_theta BAG:233-238 */
    draw_line_cosmetic(dst, (int)cx, (int)cy, (int)(cx + cr * cos((double)theta)), (int)(cy - cr * sin((double)theta)), 0xfffcba03u);
    float ddx = agent->x - goal->x, ddy = agent->y - goal->y;
    float dist = (float)sqrt((double)(ddx * ddx + ddy * ddy)); /* get_distance BAG:133-143 */
    float dist_pct = (float)(dist / (g->main_width * sqrt(2.0)));
    float bar_thickness = g->compass_dim / 8;
    RectD dr = {cxf * g->unit, (float)(.25 + g->compass_dim) * g->unit, (g->compass_dim * dist_pct) * g->unit, bar_thickness * g->unit};
    fill_rect(dst, dr, 0xfffcba03u);
    if (g->jump_delta < 0 && !g->has_support) {
        RectD r1 = get_screen_rect(g, agent->x - agent->rx, agent->y + agent->ry, 2 * agent->rx, 2 * agent->ry, 0);
        draw_ellipse_i(&cv, (int)r1.x, (int)(r1.y + r1.h * (5.0 / 6)), (int)r1.w, (int)(r1.h / 3), 0, 0, 1, 0x78787878u); /* QColor(255,255,255,120) premultiplied */
    }
}

void pgo_test_draw_ellipse(double x, double y, double w, double h, int pen, int brush, uint8_t *out) {
    uint32_t px[RES_W * RES_H];
    memset(px, 0, sizeof(px));
    const QtCanvas cv = {px, RES_W, RES_H, 0};
    RectD r = {x, y, w, h};
    draw_ellipse_f(&cv, r, pen, 0xff000002u, brush, 0xff000001u);
    for (int i = 0; i < RES_W * RES_H; i++) out[i] = (uint8_t)(px[i] & 3u);
}

void pgo_test_generated_asset(int game_id, int type, uint32_t *out4096) {
    Game g;
    memset(&g, 0, sizeof(g));
    g.game_id = game_id;
    gen_assets_init(game_id);
    memcpy(out4096, generated_asset(&g, type)->px, sizeof(uint32_t) * 4096);
}
void pgo_dump_background(PgoVec *v, int env, uint32_t *out250000) {
    if (!v->games[env].gen_bg.px) fatal("no generated background");
    memcpy(out250000, v->games[env].gen_bg.px, sizeof(uint32_t) * 250000);
}
void pgo_test_generated_background(int seed, uint32_t *out250000) {
    Rng r;
    rng_seed(&r, seed);
    ag_generate_resource(&r, out250000, 500, 500, 1, 50, 1);
}

static void game_draw(Game *g, uint32_t *dst) { /* BAG:979-1012,921-970 */
    for (int i = 0; i < RES_W * RES_H; i++) dst[i] = 0xff000000u; /* fillRect black */
    if (g->game_id == GAME_STARPILOT) { /* game_draw override starpilot.cpp:108-124 */
        if (g->opt.use_backgrounds) {
            float scale = (float)(RES_H / g->main_height);
            float bg_k = 3;
            float t = (float)g->cur_time;
            float x_off = -t * scale * g->hp_slow_v * 2 / g->char_dim;
            RectD r_bg = {x_off, -RES_H * (bg_k - 1) / 2, RES_H * bg_k * 18.0f, RES_H * bg_k};
            tile_image(dst, bg_image(g), 0, r_bg, 1, 1.0f);
        }
    } else {
    prepare_for_drawing(g, (float)RES_H);
    if (g->opt.use_backgrounds) {
        RectD main_rect = get_screen_rect(g, 0, (float)g->main_height, (float)g->main_width, (float)g->main_height, 0);
        const Img *bg = bg_image(g);
        if (g->bg_tile_ratio < 0) {
            tile_image(dst, bg, 0, main_rect, g->bg_tile_ratio, 1.0f);
        } else {
            float bgw = (float)bg->w, bgh = (float)bg->h;
            float bg_ar = bgw / bgh;
            float world_ar = (float)(g->main_width * 1.0 / g->main_height);
            float extra_w = bg_ar - world_ar;
            float offset_x = g->bg_pct_x * extra_w;
            RectD a = {-offset_x, 0, bg_ar / world_ar, 1};
            draw_image_scaled(dst, bg, 0, adjust_rect(main_rect, a), 1.0f);
        }
    }
    }
    prepare_for_drawing(g, (float)RES_H);
    draw_entities(g, dst, -1);
    int low_x, high_x, low_y, high_y;
    if (g->center_agent) {
        float margin = (float)(g->visibility / 2.0 + 1);
        low_x = (int)(g->center_x - margin);
        high_x = (int)(g->center_x + margin);
        low_y = (int)(g->center_y - margin);
        high_y = (int)(g->center_y + margin);
    } else {
        low_x = 0;
        high_x = g->main_width - 1;
        low_y = 0;
        high_y = g->main_height - 1;
    }
    for (int x = low_x; x <= high_x; x++)
        for (int y = low_y; y <= high_y; y++) {
            int type = get_obj(g, x, y);
            if (type == INVALID_OBJ) continue;
            int theme = hook_theme_for_grid_obj(g, type);
            RectD r2 = get_screen_rect(g, (float)x, (float)(y + 1), 1, 1, RENDER_EPS);
            draw_image(g, dst, r2, 0, 0, type, theme, 1.0f, 0.0f);
        }
    draw_entities(g, dst, 0);
    draw_entities(g, dst, 1);
    if (g->has_useful_vel_info && g->opt.paint_vel_info) { /* BAG:960-969 */
        const Ent *agent = &g->pool[g->agent];
        float infodim = (float)(RES_H * .2);
        int s1 = (int)((float)(.5 * agent->vx / g->maxspeed + .5) * 255);
        int s2 = (int)((float)(.5 * agent->vy / g->max_jump + .5) * 255);
        if (s1 < 0) s1 = 0;
        if (s1 > 255) s1 = 255;
        if (s2 < 0) s2 = 0;
        if (s2 > 255) s2 = 255;
        RectD d2 = {0, 0, infodim, infodim}, d3 = {infodim, 0, infodim, infodim};
        fill_rect(dst, d2, 0xff000000u | ((uint32_t)s1 << 16) | ((uint32_t)s1 << 8) | (uint32_t)s1);
        fill_rect(dst, d3, 0xff000000u | ((uint32_t)s2 << 16) | ((uint32_t)s2 << 8) | (uint32_t)s2);
    }
    if (g->game_id == GAME_JUMPER && g->opt.distribution_mode != 10) jp_draw_compass(g, dst); /* jumper.cpp:171-178 */
    if (g->game_id == GAME_NINJA) { /* game_draw override ninja.cpp:170-177 */
        float bar_height = 3 * g->jump_charge;
        RectD r = {(float).25 * g->unit, (float)(g->visibility - .5 - bar_height) * g->unit, (float).5 * g->unit, bar_height * g->unit};
        fill_rect(dst, r, 0xff42f587u);
    }
    if (g->game_id == GAME_PLUNDER) { /* game_draw override plunder.cpp:65-77; get_abs_rect BAG:803-805 */
        float w1 = g->main_width * g->juice_left;
        float w2 = (float)(g->main_width * (g->targets_hit * 1.0 / g->target_quota));
        RectD r1 = {(float).25 * g->unit, (float).25 * g->unit, w1 * g->unit, (float).5 * g->unit};
        RectD r2 = {(float).25 * g->unit, (float).75 * g->unit, w2 * g->unit, (float).5 * g->unit};
        fill_rect(dst, r1, 0xff42f587u);
        fill_rect(dst, r2, 0xfff54290u);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Game::reset / step: reference src/game.cpp:93-155                                             */
static void g_reset(Game *g) {
    if (g->episodes_remaining == 0) {
        if (g->opt.use_sequential_levels && g->level_complete) {
            g->current_level_seed = (int32_t)((uint32_t)g->current_level_seed + 997u);
        } else {
            g->current_level_seed = rng_randint(&g->level_seed_rand_gen, g->level_seed_low, g->level_seed_high);
        }
        g->episodes_remaining = 1;
    } else {
        g->reward = 0;
        g->done = 0;
        g->level_complete = 0;
    }
    rng_seed(&g->rand_gen, g->current_level_seed);
    game_reset(g);
    g->cur_time = 0;
    g->total_reward = 0;
    g->episodes_remaining -= 1;
    g->action = g->default_action;
}

static void g_step(Game *g) {
    g->cur_time += 1;
    int will_force_reset = 0;
    if (g->action == -1) {
        g->action = g->default_action;
        will_force_reset = 1;
    }
    g->reward = 0;
    g->done = 0;
    g->level_complete = 0;
    game_step(g);
    g->done = g->done || will_force_reset || (g->cur_time >= g->timeout);
    g->total_reward += g->reward;
    if (g->reward != 0) {
        g->last_reward_timer = 10;
        g->last_reward = g->reward;
    }
    g->prev_level_seed = g->current_level_seed;
    if (g->done) g_reset(g);
    if (g->opt.use_sequential_levels && g->level_complete) g->done = 0;
    g->episode_done = g->done;
    game_draw(g, g->render_buf);
}

static void game_construct(Game *g, int game_id, const PgoOptions *opt) {
    memset(g, 0, sizeof(*g));
    g->game_id = game_id;
    g->assets = opt->use_generated_assets ? &g_gen_assets[game_id] : &g_assets[game_id];
    g->opt = *opt;
    g->center_agent = opt->center_agent;
    /* Game::Game src/game.cpp:25-38 */
    g->timeout = 1000;
    g->last_reward = -1;
    g->reward = 0;
    g->done = 1;
    /* BasicAbstractGame ctor BAG:22-46 */
    g->visibility = 16;
    g->min_visibility = 0;
    g->mixrate = 0.5f;
    g->maxspeed = 0.5f;
    g->max_jump = g->maxspeed;
    g->default_action = 4;
    g->last_move_action = 7;
    g->bg_tile_ratio = 0;
    g->char_dim = 5;
    g->out_of_bounds_object = INVALID_OBJ;
    g->has_useful_vel_info = 1;
    g->random_agent_start = 1; /* basic-abstract-game.h:144 */
    ents_clear(g);
    if (game_id == GAME_COINRUN) { /* coinrun.cpp:48-58 */
        g->visibility = 13;
        g->mixrate = 0.2f;
        g->main_width = 64;
        g->main_height = 64;
        g->out_of_bounds_object = CR_WALL_MID;
    } else if (game_id == GAME_CAVEFLYER) { /* caveflyer.cpp:27-30 */
        g->mixrate = 0.9f;
    } else if (game_id == GAME_CHASER) { /* chaser.cpp:38-49 */
        g->mixrate = 1;
        g->maxspeed = (float).5;
        g->eat_timeout = 75;
        g->egg_timeout = 50;
        g->has_useful_vel_info = 0;
    } else if (game_id == GAME_BOSSFIGHT) { /* bossfight.cpp:63-71 */
        g->timeout = 4000;
        g->main_width = 20;
        g->main_height = 20;
        g->mixrate = (float).5;
        g->maxspeed = 0.85f;
    } else if (game_id == GAME_DODGEBALL) { /* dodgeball.cpp:38-45 */
        g->mixrate = (float).5;
        g->enemy_fire_delay = 50;
        g->out_of_bounds_object = DB_OOB_WALL;
    } else if (game_id == GAME_NINJA) { /* ninja.cpp:34-40 */
        g->main_width = 64;
        g->main_height = 64;
        g->out_of_bounds_object = NJ_WALL_MID;
    } else if (game_id == GAME_HEIST) { /* heist.cpp:24-34 */
        g->has_useful_vel_info = 0;
        g->main_width = 20;
        g->main_height = 20;
        g->out_of_bounds_object = WALL_OBJ;
        g->visibility = 8.0;
    } else if (game_id == GAME_PLUNDER) { /* plunder.cpp:33-43 */
        g->timeout = 4000;
        g->main_width = 20;
        g->main_height = 20;
        g->mixrate = (float).5;
        g->maxspeed = 0.85f;
        g->has_useful_vel_info = 0;
    } else if (game_id == GAME_LEAPER) { /* leaper.cpp:35-38 */
        g->maxspeed = LP_MAX_SPEED;
        g->timeout = 500;
    } else if (game_id == GAME_FRUITBOT) { /* fruitbot.cpp:32-42 */
        g->mixrate = (float).5;
        g->maxspeed = 0.85f;
        g->min_dim = 5;
        g->bullet_vscale = (float).5;
        g->bg_tile_ratio = -1;
        g->out_of_bounds_object = FB_OUT_OF_BOUNDS_WALL;
    } else if (game_id == GAME_STARPILOT) { /* starpilot.cpp:50-53 */
        g->main_width = 16;
        g->main_height = 16;
    } else if (game_id == GAME_BIGFISH) { /* bigfish.cpp:25-31 */
        g->timeout = 6000;
        g->main_width = 20;
        g->main_height = 20;
    } else if (game_id == GAME_MINER) { /* miner.cpp:25-35 */
        g->main_width = 20;
        g->main_height = 20;
        g->mixrate = (float).5;
        g->maxspeed = (float).5;
        g->has_useful_vel_info = 0;
        g->out_of_bounds_object = MN_OOB_WALL;
        g->visibility = 8.0f;
    } else if (game_id == GAME_CLIMBER) { /* climber.cpp:40-42 */
        g->out_of_bounds_object = CL_WALL_MID;
    } else if (game_id == GAME_MAZE) { /* maze.cpp:16-24 */
        g->timeout = 500;
        g->random_agent_start = 0;
        g->has_useful_vel_info = 0;
        g->out_of_bounds_object = WALL_OBJ;
        g->visibility = 8.0f;
    }
}

PgoVec *pgo_make(int game_id, int num_envs, const PgoOptions *opt) { return pgo_make_strided(game_id, num_envs, opt, 0, 1); }

/* oracle env k is env (env_offset + k * env_stride) of the logical vector: env n's level-seed generator is seeded with
 * the n-th raw draw of RandGen(rand_seed) whatever num_envs is (src/vecgame.cpp:301-314), so a strided sample of a
 * large vector -- e.g. the envs of one game of a joint handle -- can be replayed without simulating the rest */
PgoVec *pgo_make_strided(int game_id, int num_envs, const PgoOptions *opt, int env_offset, int env_stride) {
    assets_build(game_id);
    if (opt->use_generated_assets) gen_assets_init(game_id);
    PgoVec *v = (PgoVec *)calloc(1, sizeof(PgoVec));
    v->n = num_envs;
    v->games = (Game *)malloc(sizeof(Game) * (size_t)num_envs);
    int lo = 0, hi = 0; /* src/vecgame.cpp:284-293 */
    if (opt->num_levels == 0) {
        lo = 0;
        hi = INT32_MAX;
    } else if (opt->num_levels > 0) {
        lo = opt->start_level;
        hi = opt->start_level + opt->num_levels;
    }
    Rng seedgen; /* src/vecgame.cpp:301-314 */
    rng_seed(&seedgen, opt->rand_seed);
    if (env_offset < 0 || env_stride < 1) fatal("pgo_make_strided: bad offset / stride");
    for (int skip = 0; skip < env_offset; skip++) (void)rng_randint_raw(&seedgen);
    for (int n = 0; n < num_envs; n++) {
        Game *g = &v->games[n];
        game_construct(g, game_id, opt);
        rng_seed(&g->level_seed_rand_gen, rng_randint_raw(&seedgen));
        for (int skip = 1; skip < env_stride; skip++) (void)rng_randint_raw(&seedgen);
        g->level_seed_low = lo;
        g->level_seed_high = hi;
    }
    return v;
}

void pgo_free(PgoVec *v) {
    if (!v) return;
    free(v->games);
    free(v);
}

void pgo_init(PgoVec *v) { /* src/vecgame.cpp:128-131 */
    for (int n = 0; n < v->n; n++) {
        g_reset(&v->games[n]);
        game_draw(&v->games[n], v->games[n].render_buf);
    }
}

void pgo_step(PgoVec *v, const int32_t *actions) { /* src/vecgame.cpp:378-401 */
    for (int n = 0; n < v->n; n++) {
        v->games[n].action = actions[n];
        g_step(&v->games[n]);
    }
}

void pgo_observe(PgoVec *v, uint8_t *rgb, float *rew, uint8_t *first, int32_t *prev_level_seed, uint8_t *prev_level_complete, int32_t *level_seed) {
    for (int n = 0; n < v->n; n++) { /* src/game.cpp:157-165, bgr32_to_rgb888 :8-23 */
        Game *g = &v->games[n];
        if (rgb) {
            uint8_t *d = rgb + (size_t)n * RES_W * RES_H * 3;
            for (int i = 0; i < RES_W * RES_H; i++) {
                uint32_t p = g->render_buf[i];
                d[3 * i + 0] = (uint8_t)(p >> 16);
                d[3 * i + 1] = (uint8_t)(p >> 8);
                d[3 * i + 2] = (uint8_t)p;
            }
        }
        if (rew) rew[n] = g->reward;
        if (first) first[n] = (uint8_t)g->done;
        if (prev_level_seed) prev_level_seed[n] = g->prev_level_seed;
        if (prev_level_complete) prev_level_complete[n] = (uint8_t)g->level_complete;
        if (level_seed) level_seed[n] = g->current_level_seed;
    }
}

int pgo_num_entities(PgoVec *v, int env) { return v->games[env].n_ents; }

void pgo_dump_entities(PgoVec *v, int env, int32_t *out) { /* order of src/entity.cpp:90-137 */
    Game *g = &v->games[env];
    for (int i = 0; i < g->n_ents; i++) {
        const Ent *e = &g->pool[g->ents[i]];
        int32_t *o = out + 31 * i;
        float f[31];
        int isf[31];
        int k = 0;
#define PF(v) f[k] = (v); isf[k] = 1; k++;
#define PI_(v) o[k] = (int32_t)(v); isf[k] = 0; k++;
        PF(e->x) PF(e->y) PF(e->vx) PF(e->vy) PF(e->rx) PF(e->ry)
        PI_(e->type) PI_(e->image_type) PI_(e->image_theme) PI_(e->render_z) PI_(e->will_erase) PI_(e->collides_with_entities)
        PF(e->collision_margin) PF(e->rotation) PF(e->vrot)
        PI_(e->is_reflected) PI_(e->fire_time) PI_(e->spawn_time) PI_(e->life_time) PI_(e->expire_time) PI_(e->use_abs_coords)
        PF(e->friction) PI_(e->smart_step) PI_(e->avoids_collisions) PI_(e->auto_erase)
        PF(e->alpha) PF(e->health) PF(e->theta) PF(e->grow_rate) PF(e->alpha_decay) PF(e->climber_spawn_x)
#undef PF
#undef PI_
        for (int q = 0; q < 31; q++)
            if (isf[q]) memcpy(&o[q], &f[q], 4);
    }
}

void pgo_dump_grid(PgoVec *v, int env, int32_t *out, int *w, int *h) {
    Game *g = &v->games[env];
    *w = g->grid_w;
    *h = g->grid_h;
    memcpy(out, g->grid, sizeof(int) * (size_t)(g->grid_w * g->grid_h));
}

void pgo_dump_scalars(PgoVec *v, int env, int32_t *out) {
    Game *g = &v->games[env];
    memset(out, 0, 16 * sizeof(int32_t));
    out[0] = g->cur_time;
    out[1] = (int32_t)(g->rand_gen.draws & 0x7fffffff);
    out[2] = g->step_rand_int;
    out[3] = g->background_index;
    memcpy(&out[4], &g->bg_pct_x, 4);
    out[5] = g->wall_theme;
    out[6] = g->has_support;
    out[7] = g->is_on_crate;
    memcpy(&out[8], &g->last_agent_y, 4);
    out[9] = g->current_level_seed;
    out[10] = g->prev_level_seed;
    out[11] = g->last_move_action;
}
