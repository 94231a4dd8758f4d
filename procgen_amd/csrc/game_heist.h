// game_heist.h -- Heist rules as a policy for Env<> / Renderer<> (reference procgen/src/games/heist.cpp).
// A maze (MazeGen::generate_maze_with_doors) on a 13x13 grid with up to three colour-coded locked doors (entities
// that block the agent until it holds the matching key), keys and the exit gem as entities, and the collected keys
// shown as small rotated sprites in absolute screen coordinates.  The agent sprite turns to face its last move
// (atan2f -> rotation).
#pragma once
#include "pg_game_defaults.h"
#include "pg_math.h"
#include "pg_mazegen.h"

namespace pgamd {

struct Heist : BagDefaults<Heist> {
    static constexpr int GAME_ID = GAME_HEIST;
    static constexpr const char *NAME = "heist";
    typedef MazeScratchT<23, true> Scratch;  // maze_dim <= world_dim <= 23 (memory mode, heist.cpp:95-110)
    static constexpr int MAX_CELLS = 23 * 23;  // heist.cpp:95-110 (memory mode)
    static constexpr bool USES_ROTATION = true;
    static constexpr int RENDER_MIN_WAVES = 4;  // with the 16-record rotation pool the arena is 9.7 KB: four render waves per SIMD at <= 128 VGPRs measured +10 % over the pool alone (37.3 -> 41.1 M) on the same box (profiles/r05_rot_pool_ab.txt)
    static constexpr int ENT_CAP_T0 = 16, ENT_CAP_T1 = 24, ENT_CAP_T2 = 32;  // agent + 3 keys + 3 doors + exit + 3 ring keys
    template <class E>
    PG_DEV static int slots_needed_next_step(E &) { return 0; }

    static constexpr int LOCKED_DOOR = 1, KEY = 2, EXIT = 9, KEY_ON_RING = 11;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == WALL_OBJ || t == LOCKED_DOOR; }  // heist.cpp:62-64: generated as a rect texture (use_generated_assets)

#define HS_NUM_KEYS(G) (G).gsi0
#define HS_WORLD_DIM(G) (G).gsi1
#define HS_HAS_KEYS(G) (G).gsi2  // bit k: key k collected

    PG_DEV static bool center_agent(const GameOptions &o) { return o.distribution_mode == MemoryMode; }  // heist.cpp:119 (after the first reset)

    static void construct(EnvHdr &G) {  // heist.cpp:24-34
        construct_defaults(G);
        G.has_useful_vel_info = 0;
        G.main_width = 20;
        G.main_height = 20;
        G.out_of_bounds_object = WALL_OBJ;
        G.visibility = 8.0f;
    }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // heist.cpp:95-110
        const int dm = e.opt.distribution_mode;
        int wd = HS_WORLD_DIM(e.G);
        if (dm == EasyMode) wd = 9;
        else if (dm == HardMode) wd = 13;
        else if (dm == MemoryMode) wd = 23;
        HS_WORLD_DIM(e.G) = wd;
        e.G.maxspeed = (float).75;
        e.G.main_width = wd;
        e.G.main_height = wd;
    }
    PG_DEV static bool should_preserve_type_themes(int type) { return type == KEY || type == LOCKED_DOOR; }  // heist.cpp:37-39
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool is_horizontal) {  // heist.cpp:63-68
        if (e.etype(target) == LOCKED_DOOR) return !((HS_HAS_KEYS(e.G) >> meta_image_theme(e.meta(target))) & 1);
        return is_blocked(e, e.etype(src), e.etype(target), is_horizontal);
    }
    template <class E>
    PG_DEV static bool may_interact(E &e, int src_type, int target_type, bool is_horizontal) {
        return target_type == LOCKED_DOOR || is_blocked(e, src_type, target_type, is_horizontal);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // heist.cpp:77-93
        const int t = e.etype(obj);
        const int theme = meta_image_theme(e.meta(obj));
        if (t == EXIT) {
            e.G.done = 1;
            e.G.reward = 10.0f;
            e.G.level_complete = 1;
        } else if (t == KEY) {
            e.set_flag(obj, MF_WILL_ERASE, true);
            HS_HAS_KEYS(e.G) |= 1 << theme;
        } else if (t == LOCKED_DOOR) {
            if ((HS_HAS_KEYS(e.G) >> theme) & 1) e.set_flag(obj, MF_WILL_ERASE, true);
        }
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // heist.cpp:112-194
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int world_dim = HS_WORLD_DIM(G);
        const int min_maze_dim = 5;
        const int max_diff = (world_dim - min_maze_dim) / 2;
        const int difficulty = e.randn(max_diff + 1);
        int num_keys;
        if (e.opt.distribution_mode == MemoryMode) num_keys = e.randn(4);
        else num_keys = difficulty + e.randn(2);
        if (num_keys > 3) num_keys = 3;
        HS_NUM_KEYS(G) = num_keys;
        HS_HAS_KEYS(G) = 0;
        const int maze_dim = difficulty * 2 + min_maze_dim;
        const float maze_scale = (float)(G.main_height / (world_dim * 1.0));
        const int ag = G.agent;
        e.erx(ag) = (float)(.375 * (double)maze_scale);
        e.ery(ag) = (float)(.375 * (double)maze_scale);
        const float r_ent = maze_scale / 2;
        PG_SYNC();
        MazeGenDev<E, Scratch> mg(e, e.s->scratch, maze_dim);
        mg.generate_maze_with_doors(num_keys);
        e.ex(ag) = -1;
        e.ey(ag) = -1;
        const int off_x = e.randn(world_dim - maze_dim + 1);
        const int off_y = e.randn(world_dim - maze_dim + 1);
        PG_SYNC();
        e.fill_elem(0, 0, world_dim, world_dim, WALL_OBJ);
        for (int j = 0; j < maze_dim; j++) {  // cells that are not walls become SPACE (objects are entities)
            PG_FOR_LANES(l) {
                if (l < maze_dim && mg.grid_at(l + MAZE_OFFSET, j + MAZE_OFFSET) != WALL_OBJ) e.s->grid[(off_y + j) * world_dim + off_x + l] = (cell_t)SPACE;
            }
        }
        PG_SYNC();
        for (int i = 0; i < maze_dim; i++) {  // entities in the reference's x-major order (spawn_entity draws from rand_gen)
            for (int j = 0; j < maze_dim; j++) {
                const int obj = PG_UNIFORM_I(mg.grid_at(i + MAZE_OFFSET, j + MAZE_OFFSET));
                if (obj == WALL_OBJ || obj == SPACE) continue;
                const int x = off_x + i, y = off_y + j;
                const float obj_x = (float)((x + .5) * (double)maze_scale);
                const float obj_y = (float)((y + .5) * (double)maze_scale);
                if (obj >= MG_KEY_OBJ) {
                    const int k = e.spawn_entity((float)(.375 * (double)maze_scale), KEY, maze_scale * x, maze_scale * y, maze_scale, maze_scale);
                    e.set_image_theme(k, obj - MG_KEY_OBJ - 1);
                    e.match_aspect_ratio(k);
                } else if (obj >= MG_DOOR_OBJ) {
                    const int dr = e.add_entity(obj_x, obj_y, 0, 0, r_ent, LOCKED_DOOR);
                    e.set_image_theme(dr, obj - MG_DOOR_OBJ - 1);
                } else if (obj == MG_EXIT_OBJ) {
                    const int x_ = e.spawn_entity((float)(.375 * (double)maze_scale), EXIT, maze_scale * x, maze_scale * y, maze_scale, maze_scale);
                    e.match_aspect_ratio(x_);
                } else if (obj == MG_AGENT_OBJ) {
                    e.ex(ag) = obj_x;
                    e.ey(ag) = obj_y;
                }
                PG_SYNC();
            }
        }
        const float ring_key_r = 0.03f;
        for (int i = 0; i < num_keys; i++) {
            const int k = e.add_entity((float)(1 - (double)ring_key_r * (2 * i + 1.25)), (float)((double)ring_key_r * .75), 0, 0, ring_key_r, KEY_ON_RING);
            e.set_image_theme(k, i);
            e.set_image_type(k, KEY);
            e.ef(EF_ROTATION, k) = PG_PI / 2;
            e.set_render_z(k, 1);
            e.set_flag(k, MF_ABS_COORDS, true);
            e.match_aspect_ratio(k);
        }
        G.grid_dirty = 1;
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // heist.cpp:196-200
        e.bag_game_step();
        const float dx = e.G.action_vx, dy = e.G.action_vy;
        if (dx != 0 || dy != 0) e.ef(EF_ROTATION, e.G.agent) = -1 * pg_atan2f(dy, dx) + 0.0f;  // Entity::face_direction
        PG_SYNC();
    }

    template <class R>
    PG_DEV static bool should_draw_entity(R &r, int i) {  // heist.cpp:70-75
        if (r.etype(i) == KEY_ON_RING) return ((HS_HAS_KEYS(r.G) >> meta_image_theme(r.meta(i))) & 1) != 0;
        return true;
    }
};

}  // namespace pgamd
