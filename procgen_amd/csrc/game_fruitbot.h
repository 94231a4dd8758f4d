// game_fruitbot.h -- FruitBot rules as a policy for Env<> / Renderer<> (reference procgen/src/games/fruitbot.cpp).
// A vertical scroller on a 20x60 world without grid objects: the agent drifts upward through gaps in walls that are
// entities drawn as rows of tiles (get_tile_aspect_ratio), collects fruit, avoids food, opens locked doors with a
// key bullet (entity-entity collisions), finishes on a row of presents.  The background is a column of tiles.
#pragma once
#include "pg_game_defaults.h"

namespace pgamd {

struct FruitBot : BagDefaults<FruitBot> {
    static constexpr int GAME_ID = GAME_FRUITBOT;
    static constexpr const char *NAME = "fruitbot";
    static constexpr int MAX_CELLS = 20 * 60;  // fruitbot.cpp:152-160
    static constexpr bool USES_ENTITY_COLLISIONS = true;
    static constexpr bool USES_ROTATION = true;  // the agent is drawn turned by -90 degrees
    static constexpr bool GRID_RARELY_ON_SCREEN = true;  // the grid holds only SPACE; out-of-bounds walls are the only cells with an image (pg_render.h build_pull_tables)
    static constexpr int RENDER_MIN_WAVES = 4;  // with the 16-record rotation pool the arena is 9.7 KB: four render waves per SIMD at <= 128 VGPRs measured +11 % over the pool alone (20.2 -> 22.5 M) on the same box (profiles/r05_rot_pool_ab.txt)
    static constexpr bool USES_TILED_ENTITIES = true;
    // 10 walls x 2 barriers + doors and locks + 20 presents + <= 19 good + <= 19 bad + agent + <= 2 bullets
    static constexpr int ENT_CAP_T0 = 96, ENT_CAP_T1 = 112, ENT_CAP_T2 = 128;
    static constexpr bool TILED_BACKGROUND = true;  // bg_tile_ratio = -1 (fruitbot.cpp:41)
    static constexpr int WIDE_ROWS = 8;  // half-band fetch batches: the renderer stays under 168 VGPRs (three waves per SIMD)
    static constexpr int RENDER_CMD_SETS = 2;  // frames with more than 64 visible entities are common
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 1 + 1; }

    static constexpr int BARRIER = 1, OUT_OF_BOUNDS_WALL = 2, PLAYER_BULLET = 3, BAD_OBJ = 4, GOOD_OBJ = 7, LOCKED_DOOR = 10, LOCK = 11, PRESENT = 12;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == BARRIER || t == LOCKED_DOOR || t == PRESENT; }  // fruitbot.cpp:137-139: generated as a rect texture (use_generated_assets)
    static constexpr int KEY_DURATION = 8;
    static constexpr float DOOR_ASPECT_RATIO = 3.25f;

#define FB_LAST_FIRE_TIME(G) (G).gsi0
#define FB_MIN_DIM(G) (G).gsf0
#define FB_BULLET_VSCALE(G) (G).gsf1

    static void construct(EnvHdr &G) {  // fruitbot.cpp:32-42
        construct_defaults(G);
        G.mixrate = (float).5;
        G.maxspeed = 0.85f;
        G.max_jump = 0.5f;
        FB_MIN_DIM(G) = 5;
        FB_BULLET_VSCALE(G) = (float).5;
        G.bg_tile_ratio = -1;
        G.out_of_bounds_object = OUT_OF_BOUNDS_WALL;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // fruitbot.cpp:152-160
        e.G.main_width = e.opt.distribution_mode == EasyMode ? 10 : 20;
        e.G.main_height = 60;
    }
    PG_DEV static bool will_reflect(int src, int target) { return src == BAD_OBJ && (target == BARRIER || target == WALL_OBJ); }  // :80-82
    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // fruitbot.cpp:84-86
        return target == WALL_OBJ || target == e.G.out_of_bounds_object || (src_type == PLAYER && target == OUT_OF_BOUNDS_WALL);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // fruitbot.cpp:96-118
        const int t = e.etype(obj);
        if (t == BARRIER) {
            e.G.done = 1;
        } else if (t == BAD_OBJ) {
            e.G.reward += -4.0f;
            e.set_flag(obj, MF_WILL_ERASE, true);
        } else if (t == LOCKED_DOOR) {
            e.G.done = 1;
        } else if (t == GOOD_OBJ) {
            e.G.reward += 1.0f;
            e.set_flag(obj, MF_WILL_ERASE, true);
        } else if (t == PRESENT) {
            e.G.reward += 10.0f;
            e.G.done = 1;
            e.G.level_complete = 1;
        }
    }
    template <class E>
    PG_DEV static void handle_collision(E &e, int src, int target) {  // fruitbot.cpp:120-138
        if (e.etype(src) != PLAYER_BULLET) return;
        const int tt = e.etype(target);
        if (tt == BARRIER) {
            e.set_flag(src, MF_WILL_ERASE, true);
        } else if (tt == LOCK) {
            e.set_flag(src, MF_WILL_ERASE, true);
            e.set_flag(target, MF_WILL_ERASE, true);
            // the first door on the lock's row
            const int n = e.G.n_ents;
            const float ty = e.ey(target);
            for (int c = 0; c < ((n + 63) >> 6); c++) {
                const uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == LOCKED_DOOR && pg_fabs((double)(e.ey((c << 6) + l) - ty)) < 1);
                if (m) {
                    e.set_flag((c << 6) + pg_ctz64(m), MF_WILL_ERASE, true);
                    break;
                }
            }
        }
    }
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {  // fruitbot.cpp:146-150
        cx = (float)(e.G.main_width / 2.0);
        cy = (float)((double)e.ey(e.G.agent) + e.G.main_width / 2.0 - (double)(2 * e.ery(e.G.agent)));
        e.G.visibility = (float)e.G.main_width;
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // fruitbot.cpp:162-166
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = 0.2f;
        e.G.action_vrot = 0;
    }

    template <class E>
    PG_DEV static void add_walls(E &e, float ry, bool use_door, float min_pct) {  // fruitbot.cpp:168-201
        EnvHdr &G = e.G;
        const float rw = (float)G.main_width;
        const float wall_ry = 0.3f;
        const float lock_rx = (float).25;
        const float lock_ry = 0.45f;
        float pct = (float)((double)min_pct + .2 * (double)e.rand01());
        if (use_door) {
            pct += 0.1f;
            const float lock_pct_w = 2 * lock_rx / G.main_width;
            const float door_pct_w = (wall_ry * 2 * DOOR_ASPECT_RATIO) / G.main_width;
            const int num_doors = (int)pg_ceil((double)((pct - 2 * lock_pct_w) / door_pct_w));
            pct = 2 * lock_pct_w + door_pct_w * num_doors;
        }
        const float gapw = pct * rw;
        const float w1 = e.rand01() * (rw - gapw);
        const float w2 = rw - w1 - gapw;
        e.add_entity_rxy(w1 / 2, ry, 0, 0, w1 / 2, wall_ry, BARRIER);
        e.add_entity_rxy(rw - w2 / 2, ry, 0, 0, w2 / 2, wall_ry, BARRIER);
        if (use_door) {
            const int is_on_right = e.randn(2);
            const float lock_x = w1 + lock_rx + is_on_right * (gapw - 2 * lock_rx);
            const float door_x = w1 + gapw / 2 - (is_on_right * 2 - 1) * lock_rx;
            e.add_entity_rxy(door_x, ry, 0, 0, gapw / 2 - lock_rx, wall_ry, LOCKED_DOOR);
            e.add_entity_rxy(lock_x, ry - lock_ry + wall_ry, 0, 0, lock_rx, lock_ry, LOCK);
        }
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // fruitbot.cpp:203-249
        e.bag_game_reset();
        EnvHdr &G = e.G;
        FB_LAST_FIRE_TIME(G) = 0;
        const int min_sep = 4, buf_h = 4;
        int num_walls = 10, object_group_size = 6;
        float door_prob = (float).125;
        float min_pct = (float).1;
        if (e.opt.distribution_mode == EasyMode) {
            num_walls = 5;
            object_group_size = 2;
            door_prob = 0;
            min_pct = (float).2;
        }
        // RandGen::partition (reference src/randgen.cpp:33-41): counts kept in the LDS word scratch
        for (int k = 0; k < num_walls; k++) e.s->tmp[k] = 0;
        PG_SYNC();
        const int px = G.main_height - min_sep * num_walls - buf_h;
        for (int i = 0; i < px; i++) {
            const int k = e.randn(num_walls);
            e.s->tmp[k] += 1;
        }
        PG_SYNC();
        int curr_h = 0;
        for (int k = 0; k < num_walls; k++) {
            const int dy = min_sep + PG_UNIFORM_I(e.s->tmp[k]);
            curr_h += dy;
            const bool use_door = (dy > 5) && e.rand01() < door_prob;
            add_walls(e, (float)curr_h, use_door, min_pct);
        }
        const int ag = G.agent;
        e.ey(ag) = e.ery(ag);
        const int num_good = e.randn(10) + 10;
        const int num_bad = e.randn(10) + 10;
        for (int i = 0; i < G.main_width; i++) {
            const int p = e.add_entity_rxy((float)(i + .5), (float)(G.main_height - .5), 0, 0, (float).5, (float).5, PRESENT);
            e.choose_random_theme(p);
        }
        PG_SYNC();
        e.spawn_entities(num_good, (float).5, GOOD_OBJ, 0, 0, (float)G.main_width, (float)G.main_height);
        e.spawn_entities(num_bad, (float).5, BAD_OBJ, 0, 0, (float)G.main_width, (float)G.main_height);
        for (int i = 0; i < G.n_ents; i++) {  // rand draws in list order
            const int t = e.etype(i);
            if (t == GOOD_OBJ || t == BAD_OBJ) {
                e.set_image_theme(i, e.randn(object_group_size));
                e.fit_aspect_ratio(i);
            }
        }
        e.ef(EF_ROTATION, ag) = -1 * PG_PI / 2;
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // fruitbot.cpp:251-262
        e.bag_game_step();
        EnvHdr &G = e.G;
        if (G.special_action == 1 && (G.cur_time - FB_LAST_FIRE_TIME(G)) >= KEY_DURATION) {
            const int ag = G.agent;
            const float vx = 0, vy = 1;
            const int b = e.add_entity(e.ex(ag), e.ey(ag), vx * FB_BULLET_VSCALE(G), vy * FB_BULLET_VSCALE(G), (float).25, PLAYER_BULLET);
            e.ei(EF_EXPIRE_TIME, b) = KEY_DURATION;
            e.set_flag(b, MF_COLLIDES, true);
            FB_LAST_FIRE_TIME(G) = G.cur_time;
        }
        PG_SYNC();
    }

    template <class E>
    PG_DEV static float tile_aspect_ratio(E &e, int i) {  // fruitbot.cpp:87-94
        const int t = e.etype(i);
        if (t == BARRIER) return 1;
        if (t == LOCKED_DOOR) return DOOR_ASPECT_RATIO;
        return 0;
    }
};

}  // namespace pgamd
