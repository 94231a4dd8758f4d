// game_ninja.h -- Ninja rules as a policy for Env<> / Renderer<> (reference procgen/src/games/ninja.cpp).
// A platformer on a 64x64 grid with a charged jump (the charge is shown as a HUD bar), bombs and fire as grid
// cells, and throwing stars: smart_step projectiles that stick to walls and detonate bombs.
#pragma once
#include "pg_game_defaults.h"
#include "pg_math.h"

namespace pgamd {

struct Ninja : BagDefaults<Ninja> {
    static constexpr int GAME_ID = GAME_NINJA;
    static constexpr int RENDER_MIN_WAVES = 5;  // the renderer fits 96 VGPRs without scratch and 8136 B of LDS: five waves per SIMD (kernels_game.hip)
    // pg_env.h GameParSmart: blocking / reflecting targets of this game are wall types only, never a smart entity's type,
    // and the hooks basic_step_object calls touch nothing but the moving object
    static constexpr bool PAR_SMART = true;
    PG_DEV static bool par_smart_type_ok(int t) { return t == PLAYER || t == THROWING_STAR; }
    static constexpr const char *NAME = "ninja";
    static constexpr int MAX_CELLS = 64 * 64;  // ninja.cpp:36-37
    static constexpr bool HAS_OVERLAY = true;
    static constexpr bool HAS_BLOCK_HOOK = true;
    static constexpr int ENT_CAP_T0 = 24, ENT_CAP_T1 = 32, ENT_CAP_T2 = 48;  // agent + goal + <= 5 stars + explosions
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return 2 * e.G.n_ents + 2; }  // every star may add an explosion, + one new star

    static constexpr int GOAL = 1, BOMB = 6, THROWING_STAR = 7, PLAYER_JUMP = 9, PLAYER_RIGHT1 = 12, PLAYER_RIGHT2 = 13, FIRE = 14, WALL_MID = 20;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == WALL_MID; }  // ninja.cpp:135-137: generated as a rect texture (use_generated_assets)
    static constexpr int NUM_WALL_THEMES = 3;

#define NJ_HAS_SUPPORT(G) (G).gsi0
#define NJ_FACING_RIGHT(G) (G).gsi1
#define NJ_LAST_FIRE_TIME(G) (G).gsi2
#define NJ_WALL_THEME(G) (G).gsi3
#define NJ_GRAVITY(G) (G).gsf0
#define NJ_AIR_CONTROL(G) (G).gsf1
#define NJ_JUMP_CHARGE(G) (G).gsf2
#define NJ_JUMP_CHARGE_INC(G) (G).gsf3

    PG_DEV static bool is_wall(int t) { return t == WALL_MID; }

    static void construct(EnvHdr &G) {  // ninja.cpp:34-40
        construct_defaults(G);
        G.main_width = 64;
        G.main_height = 64;
        G.out_of_bounds_object = WALL_MID;
    }
    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // ninja.cpp:142-156 (the stars' side effect: on_grid_block)
        if (is_wall(target) && (src_type == PLAYER || src_type == THROWING_STAR)) return true;
        return target == WALL_OBJ || target == e.G.out_of_bounds_object;
    }
    template <class E>
    PG_DEV static void on_grid_block(E &e, int obj) {  // throwing stars stick to walls
        if (e.etype(obj) == THROWING_STAR) {
            e.evx(obj) = 0;
            e.evy(obj) = 0;
        }
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // ninja.cpp:77-87
        const int t = e.etype(obj);
        if (t == EXPLOSION) {
            e.G.done = 1;
        } else if (t == GOAL) {
            e.G.reward += 10.0f;
            e.G.level_complete = 1;
            e.G.done = 1;
        }
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &e, int obj, int type, int i, int j) {  // ninja.cpp:89-107
        const int ot = e.etype(obj);
        if (ot == PLAYER) {
            if (type == FIRE || type == BOMB) e.G.done = 1;
        } else if (ot == THROWING_STAR) {
            if (type == BOMB) {
                e.set_flag(obj, MF_WILL_ERASE, true);
                e.set_obj(i, j, SPACE);
                e.add_entity((float)(i + .5), (float)(j + .5), 0, 0, (float).5, EXPLOSION);
            }
            if (is_wall(type)) e.set_flag(obj, MF_WILL_ERASE, true);
        }
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // ninja.cpp:109-124
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const float mixrate_x = NJ_HAS_SUPPORT(G) ? G.mixrate : (G.mixrate * NJ_AIR_CONTROL(G));
        e.evx(ag) = (1 - mixrate_x) * e.evx(ag) + mixrate_x * G.maxspeed * G.action_vx;
        float vy = e.evy(ag);
        if (G.action_vy < 1 && NJ_JUMP_CHARGE(G) > 0) {
            vy = NJ_JUMP_CHARGE(G) * G.max_jump;
            NJ_JUMP_CHARGE(G) = 0;
        }
        if (!NJ_HAS_SUPPORT(G)) {
            if (vy > -2) vy -= NJ_GRAVITY(G);
        }
        e.evy(ag) = vy;
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // ninja.cpp:318-347
        EnvHdr &G = e.G;
        G.action_vx = (float)(move_action / 3 - 1);
        G.action_vy = (float)((move_action % 3) - 1);
        if (G.action_vy < 0) G.action_vy = 0;
        if (G.action_vx > 0) NJ_FACING_RIGHT(G) = 1;
        if (G.action_vx < 0) NJ_FACING_RIGHT(G) = 0;
        const int ag = G.agent;
        const float ax = e.ex(ag), ay = e.ey(ag), arx = e.erx(ag), ary = e.ery(ag);
        const float by = (float)((double)ay - ((double)ary + .01));
        const int o1 = e.get_obj_from_floats((float)((double)ax - ((double)arx - .01)), by);
        const int o2 = e.get_obj_from_floats((float)((double)ax + ((double)arx - .01)), by);
        const bool s1 = is_wall(o1) || o1 == G.out_of_bounds_object;
        const bool s2 = is_wall(o2) || o2 == G.out_of_bounds_object;
        NJ_HAS_SUPPORT(G) = (s1 || s2) ? 1 : 0;
        if (NJ_HAS_SUPPORT(G) && G.action_vy == 1) {
            G.action_vy = 1;
            NJ_JUMP_CHARGE(G) += NJ_JUMP_CHARGE_INC(G);
            if (NJ_JUMP_CHARGE(G) > 1) NJ_JUMP_CHARGE(G) = 1;
        } else {
            G.action_vy = 0;
        }
        if (!NJ_HAS_SUPPORT(G)) NJ_JUMP_CHARGE(G) = 0;
    }

    template <class E>
    PG_DEV static void fill_ground_block(E &e, int x, int y, int dx, int dy) {  // fill_block_top ninja.cpp:179-188 with fill == top
        if (dy <= 0) return;
        e.fill_elem(x, y, dx, dy - 1, WALL_MID);
        e.fill_elem(x, y + dy - 1, dx, 1, WALL_MID);
    }
    template <class E>
    PG_DEV static void generate_coin_to_the_right(E &e, int difficulty) {  // ninja.cpp:197-285
        EnvHdr &G = e.G;
        int min_gap = difficulty - 1;
        int min_plat_w = 1;
        int inc_dy = 4;
        if (e.opt.distribution_mode == EasyMode) {
            min_gap -= 1;
            if (min_gap < 0) min_gap = 0;
            min_plat_w = 3;
            inc_dy = 2;
        }
        const float bomb_prob = (float)(.25 * (difficulty - 1));
        const int max_gap_inc = difficulty == 1 ? 1 : 2;
        const int num_sections = e.randn(difficulty) + difficulty;
        const int start_x = 5;
        int curr_x = start_x;
        int curr_y = G.main_height / 2;
        int min_y = curr_y;
        const int w = G.main_width;
        const float _max_dy = G.max_jump * G.max_jump / (2 * NJ_GRAVITY(G));
        const int max_dy = (int)((double)_max_dy - .5);
        int prev_x, prev_y;
        fill_ground_block(e, 0, 0, start_x, curr_y);
        e.fill_elem(0, curr_y + 8, start_x, G.main_height - curr_y - 8, WALL_MID);
        for (int i = 0; i < num_sections; i++) {
            prev_x = curr_x;
            prev_y = curr_y;
            const int num_edges = e.randn(2) + 1;
            int max_y = -1;
            int last_edge_y = -1;
            for (int j = 0; j < num_edges; j++) {
                curr_x = prev_x + j;
                if (curr_x + 15 >= w) break;
                curr_y = prev_y;
                int dy = e.randn(inc_dy) + 1 + (int)(difficulty / 3);
                if (dy > max_dy) dy = max_dy;
                if (curr_y >= G.main_height - 15) dy *= -1;
                else if (curr_y >= 5 && (double)e.rand01() < .4) dy *= -1;
                curr_y += dy;
                if (curr_y < 3) curr_y = 3;
                const int diff = curr_y - last_edge_y;
                if ((diff < 0 ? -diff : diff) <= 1) curr_y = last_edge_y + 2;
                const int dx = min_plat_w + e.randn(3);
                fill_ground_block(e, curr_x, curr_y - 1, dx, 1);
                curr_x += dx;
                curr_x += min_gap + e.randn(max_gap_inc + 1);
                if (curr_y > max_y) max_y = curr_y;
                if (curr_y < min_y) min_y = curr_y;
                last_edge_y = curr_y;
            }
            if (e.rand01() < bomb_prob) {
                const int bx = e.randn(curr_x - prev_x + 1) + prev_x;
                e.set_obj(bx, max_y + 2, BOMB);
                PG_SYNC();
            }
            const int ceiling_height = 11;
            const int ceiling_start = max_y - 1 + ceiling_height;
            fill_ground_block(e, prev_x, ceiling_start, curr_x - prev_x, G.main_height - ceiling_start);
        }
        const int goal = e.add_entity((float)(curr_x + .5), (float)(curr_y + .5), 0, 0, (float).5, GOAL);
        e.choose_random_theme(goal);
        fill_ground_block(e, curr_x, curr_y - 1, 1, 1);
        e.fill_elem(curr_x, curr_y + 6, 1, G.main_height - curr_y - 6, WALL_MID);
        int fire_y = min_y - 2;
        if (fire_y < 1) fire_y = 1;
        fill_ground_block(e, start_x, 0, G.main_width - start_x, fire_y);
        e.fill_elem(start_x, fire_y, G.main_width - start_x, 1, FIRE);
        e.fill_elem(curr_x + 1, 0, G.main_width - curr_x - 1, G.main_height, WALL_MID);
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // ninja.cpp:287-316
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        NJ_GRAVITY(G) = 0.2f;
        G.max_jump = 1.5;
        NJ_AIR_CONTROL(G) = 0.15f;
        G.maxspeed = (float).5;
        NJ_HAS_SUPPORT(G) = 0;
        NJ_FACING_RIGHT(G) = 1;
        NJ_JUMP_CHARGE(G) = 0;
        NJ_JUMP_CHARGE_INC(G) = (float).25;
        G.visibility = 16;
        e.erx(ag) = (float).5;
        e.ery(ag) = (float).5;
        e.ex(ag) = 1 + e.erx(ag);
        e.ey(ag) = G.main_height / 2 + e.ery(ag);
        if (e.opt.distribution_mode == EasyMode) {
            G.max_jump = (float)1.25;
            NJ_JUMP_CHARGE_INC(G) = 1;
            G.visibility = 10;
        }
        const int max_difficulty = 3;
        const int difficulty = e.randn(max_difficulty) + 1;
        NJ_LAST_FIRE_TIME(G) = 0;
        NJ_WALL_THEME(G) = e.randn(NUM_WALL_THEMES);
        PG_SYNC();
        e.fill_elem(0, 0, G.main_width, 1, WALL_MID);  // init_floor_and_walls ninja.cpp:190-195
        e.fill_elem(0, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(G.main_width - 1, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(0, G.main_height - 1, G.main_width, 1, WALL_MID);
        generate_coin_to_the_right(e, difficulty);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // ninja.cpp:349-383
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        if (G.special_action > 0 && (G.cur_time - NJ_LAST_FIRE_TIME(G)) >= 3) {
            float theta = 0;
            const float bullet_vel = 1;
            if (G.special_action == 1) theta = 0;
            else if (G.special_action == 2) theta = PG_PI / 4;
            else if (G.special_action == 3) theta = PG_PI / 2;
            else if (G.special_action == 4) theta = -1 * PG_PI / 4;
            if (e.eflag(ag, MF_REFLECTED)) theta = PG_PI - theta;
            const int b = e.add_entity(e.ex(ag), e.ey(ag), (float)((double)bullet_vel * pg_cos_d((double)theta)), (float)((double)bullet_vel * pg_sin_d((double)theta)), (float).25, THROWING_STAR);
            e.set_flag(b, MF_COLLIDES, true);
            e.ei(EF_EXPIRE_TIME, b) = 15;
            e.set_flag(b, MF_SMART_STEP, true);
            NJ_LAST_FIRE_TIME(G) = G.cur_time;
        }
        PG_SYNC();
    }

    template <class E>
    PG_DEV static int image_for_type(E &e, int type) {  // ninja.cpp:158-168
        if (type == PLAYER) {
            const EnvHdr &G = e.G;
            if ((double)pg_fabsf(e.evx(G.agent)) < .01 && G.action_vx == 0 && NJ_HAS_SUPPORT(G)) return PLAYER;
            return (G.cur_time / 5 % 2 == 0 || !NJ_HAS_SUPPORT(G)) ? PLAYER_RIGHT1 : PLAYER_RIGHT2;
        }
        return type < 0 ? -type : type;
    }
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &e, int type) { return is_wall(type) ? NJ_WALL_THEME(e.G) : 0; }  // ninja.cpp:130-135
    // game_draw override ninja.cpp:170-177: the jump-charge bar
    template <class R>
    PG_DEV static void draw_overlay(R &r) {
        const EnvHdr &G = r.G;
        const float bar_height = 3 * NJ_JUMP_CHARGE(G);
        r.exec_fill(r.get_abs_rect((float).25, (float)((double)G.visibility - .5 - (double)bar_height), (float).5, bar_height), 0xff42f587u);
    }
};

}  // namespace pgamd
