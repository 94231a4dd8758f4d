// game_climber.h -- Climber rules as a policy for Env<> / Renderer<> (reference procgen/src/games/climber.cpp).
// A vertical platformer on a 20x64 grid: CoinRun-like control, patrolling enemies, coins as entities, and a camera
// that follows the agent vertically (choose_center also rewrites `visibility`).
#pragma once
#include "pg_env.h"

namespace pgamd {

struct Climber {
    static constexpr int GAME_ID = GAME_CLIMBER;
    static constexpr bool DISPLAY_LIST = true;  // frames are drawn prep -> raster (pg_prep.h)
    static constexpr int RENDER_MIN_WAVES = 5;  // 102 -> 96 VGPRs with 12 B of scratch (the phase-counter address), arena 8136 B: five render waves per SIMD measured +4.5 % on the same box (55.9 -> 58.5 M, profiles/r05_try_ab.txt)
    // pg_env.h GameParSmart: blocking / reflecting targets of this game are wall types only, never a smart entity's type,
    // and the hooks basic_step_object calls touch nothing but the moving object
    static constexpr bool PAR_SMART = true;
    PG_DEV static bool par_smart_type_ok(int t) { return t == PLAYER || t == ENEMY; }
    static constexpr const char *NAME = "climber";
    typedef uint8_t cell_t;
    static constexpr int MAX_CELLS = 20 * 64;  // climber.cpp:230-233
    static constexpr bool USES_ENTITY_COLLISIONS = false;
    // <= 10 platforms, each with at most one enemy and one coin, + the agent; nothing spawns during a step
    static constexpr int ENT_CAP_T0 = 32, ENT_CAP_T1 = 64, ENT_CAP_T2 = 128;
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 3 < 26 ? 26 : e.G.n_ents + 3; }

    // object ids climber.cpp:12-24
    static constexpr int COIN = 1, ENEMY = 5, ENEMY1 = 6, ENEMY2 = 7, PLAYER_JUMP = 9, PLAYER_RIGHT1 = 12, PLAYER_RIGHT2 = 13;
    static constexpr int WALL_MID = 15, WALL_TOP = 16, ENEMY_BARRIER = 19;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == WALL_MID || t == WALL_TOP; }  // climber.cpp:128-130: generated as a rect texture (use_generated_assets)
    static constexpr float PATROL_RANGE = 4;
    static constexpr int NUM_WALL_THEMES = 4;

#define CLB_WALL_THEME(G) (G).gsi0
#define CLB_HAS_SUPPORT(G) (G).gsi1
#define CLB_FACING_RIGHT(G) (G).gsi2
#define CLB_COIN_QUOTA(G) (G).gsi3
#define CLB_COINS_COLLECTED(G) (G).gsi4
#define CLB_GRAVITY(G) (G).gsf1
#define CLB_AIR_CONTROL(G) (G).gsf2

    PG_DEV static bool center_agent(const GameOptions &o) { return o.center_agent != 0; }
    PG_DEV static bool is_wall(int t) { return t == WALL_MID || t == WALL_TOP; }

    static void construct(EnvHdr &G) {  // Game::Game, BAG ctor (BAG:22-46), Climber ctor (climber.cpp:40-42)
        G = EnvHdr{};
        G.timeout = 1000;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 16;
        G.min_visibility = 0;
        G.mixrate = 0.5f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.out_of_bounds_object = WALL_MID;
        G.has_useful_vel_info = 1;
        G.random_agent_start = 1;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // climber.cpp:230-233
        e.G.main_width = e.opt.distribution_mode == EasyMode ? 16 : 20;
        e.G.main_height = 64;
    }

    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // BAG:485-492 + climber.cpp:136-143
        if (target == WALL_OBJ) return true;
        if (target == e.G.out_of_bounds_object) return true;
        if (src_type == PLAYER && is_wall(target)) return true;
        return false;
    }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool h) { return is_blocked(e, e.etype(src), e.etype(target), h); }
    PG_DEV static bool will_reflect(int src, int target) {  // climber.cpp:110-112
        return src == ENEMY && (is_wall(target) || target == ENEMY_BARRIER);
    }
    template <class E>
    PG_DEV static bool may_interact(E &e, int s, int t, bool h) { return is_blocked(e, s, t, h) || will_reflect(s, t); }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // climber.cpp:90-100
        const int t = e.etype(obj);
        if (t == ENEMY) {
            e.G.done = 1;
        } else if (t == COIN) {
            e.G.reward += 1.0f;
            CLB_COINS_COLLECTED(e.G) += 1;
            e.set_flag(obj, MF_WILL_ERASE, true);
        }
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}

    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // climber.cpp:268-288
        EnvHdr &G = e.G;
        G.action_vx = (float)(move_action / 3 - 1);
        G.action_vy = (float)((move_action % 3) - 1);
        if (G.action_vy < 0) G.action_vy = 0;
        if (G.action_vx > 0) CLB_FACING_RIGHT(G) = 1;
        if (G.action_vx < 0) CLB_FACING_RIGHT(G) = 0;
        const int ag = G.agent;
        const float ax = e.ex(ag), ay = e.ey(ag), arx = e.erx(ag), ary = e.ery(ag);
        const float by = (float)((double)ay - ((double)ary + .01));
        const int o1 = e.get_obj_from_floats((float)((double)ax - ((double)arx - .01)), by);
        const int o2 = e.get_obj_from_floats((float)((double)ax + ((double)arx - .01)), by);
        const bool s1 = is_wall(o1) || o1 == G.out_of_bounds_object;
        const bool s2 = is_wall(o2) || o2 == G.out_of_bounds_object;
        CLB_HAS_SUPPORT(G) = (s1 || s2) ? 1 : 0;
        G.action_vy = (CLB_HAS_SUPPORT(G) && G.action_vy == 1) ? 1.0f : 0.0f;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // climber.cpp:114-126
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const float mixrate_x = CLB_HAS_SUPPORT(G) ? G.mixrate : (G.mixrate * CLB_AIR_CONTROL(G));
        e.evx(ag) = (1 - mixrate_x) * e.evx(ag) + mixrate_x * G.maxspeed * G.action_vx;
        float vy = e.evy(ag);
        if (G.action_vy > 0) vy = G.max_jump;
        if (!CLB_HAS_SUPPORT(G)) {
            if (vy > -2) vy -= CLB_GRAVITY(G);
        }
        e.evy(ag) = vy;
    }
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {  // climber.cpp:261-265
        cx = (float)(e.G.main_width / 2.0);
        cy = (float)((double)e.ey(e.G.agent) + e.G.main_width / 2.0 - (double)(5 * e.ery(e.G.agent)));
        e.G.visibility = (float)e.G.main_width;
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // climber.cpp:290-316
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        PG_SYNC();
        const int n = G.n_ents;
        const int cur_time = G.cur_time;
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                const int i = base + l;
                if (i < n && e.etype(i) == ENEMY) {
                    const float x = e.ex(i), sx = e.ef(EF_CLIMBER_SPAWN_X, i);
                    float vx = e.evx(i);
                    if (x > sx + PATROL_RANGE) vx = -1 * pg_fabsf(vx);
                    else if (x < sx - PATROL_RANGE) vx = pg_fabsf(vx);
                    e.evx(i) = vx;
                    e.set_image_type(i, cur_time / 5 % 2 == 0 ? ENEMY1 : ENEMY2);
                    e.set_flag(i, MF_REFLECTED, vx < 0);
                }
            }
        }
        PG_SYNC();
        if (CLB_COIN_QUOTA(G) == CLB_COINS_COLLECTED(G)) {
            G.done = 1;
            G.reward += 10.0f;
            G.level_complete = 1;
        }
    }

    template <class E>
    PG_DEV static void generate_platforms(E &e) {  // climber.cpp:171-228
        EnvHdr &G = e.G;
        const int difficulty = e.randn(3);
        const int min_platforms = difficulty * difficulty + 1;
        const int max_platforms = (difficulty + 1) * (difficulty + 1) + 1;
        const int num_platforms = e.randn(max_platforms - min_platforms + 1) + min_platforms;
        CLB_COIN_QUOTA(G) = 0;
        CLB_COINS_COLLECTED(G) = 0;
        int curr_x = e.randn(G.main_width - 4) + 2;
        int curr_y = 0;
        const int margin_x = 3;
        const float enemy_prob = e.opt.distribution_mode == EasyMode ? (float).2 : (float).5;
        for (int i = 0; i < num_platforms; i++) {
            const int max_dy = (int)(G.max_jump * G.max_jump / (2 * CLB_GRAVITY(G)));  // choose_delta_y :164-169
            const int min_dy = 3;
            const int delta_y = e.randn(max_dy - min_dy + 1) + min_dy;
            const bool can_spawn_enemy = (curr_x >= margin_x) && (curr_x <= G.main_width - margin_x);
            if (can_spawn_enemy && (e.rand01() < enemy_prob)) {
                // two draws in different arguments of one call: the reference build evaluates right to left
                // (velocity sign first); pinned by the oracle against the compiled reference
                const float evx = (float)(.15 * (e.randn(2) * 2 - 1));
                const float ey = (float)(curr_y + e.randn(2) + 2 + .5);
                const int ent = e.add_entity((float)(curr_x + .5), ey, evx, 0, (float).5, ENEMY);
                e.set_image_type(ent, ENEMY1);
                e.set_flag(ent, MF_SMART_STEP, true);
                e.ef(EF_CLIMBER_SPAWN_X, ent) = (float)(curr_x + .5);
                e.match_aspect_ratio(ent);
            }
            curr_y += delta_y;
            const int plat_len = 2 + e.randn(10);
            int vx = e.randn(2) * 2 - 1;
            if (curr_x < margin_x) vx = 1;
            if (curr_x > G.main_width - margin_x) vx = -1;
            int nc = 0;  // candidates are curr_x + (j + 1) * vx for j < nc
            for (int j = 0; j < plat_len; j++) {
                const int nx = curr_x + (j + 1) * vx;
                if (nx <= 0 || nx >= G.main_width - 1) break;
                nc++;
                e.set_obj(nx, curr_y, WALL_TOP);
            }
            PG_SYNC();
            if (e.rand01() < .5 || i == num_platforms - 1) {
                if (nc <= 0) {
                    e.fail(PGE_ASSERT);
                    return;
                }
                const int coin_x = curr_x + (e.randn(nc) + 1) * vx;
                e.add_entity((float)(coin_x + .5), (float)(curr_y + 1.5), 0, 0, 0.3f, COIN);
                CLB_COIN_QUOTA(G) += 1;
            }
            if (nc <= 0) {
                e.fail(PGE_ASSERT);
                return;
            }
            curr_x = curr_x + (e.randn(nc) + 1) * vx;
        }
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // climber.cpp:235-255
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        CLB_GRAVITY(G) = 0.2f;
        G.max_jump = 1.5;
        CLB_AIR_CONTROL(G) = 0.15f;
        G.maxspeed = (float).5;
        CLB_HAS_SUPPORT(G) = 0;
        CLB_FACING_RIGHT(G) = 1;
        e.erx(ag) = (float).5;
        e.ery(ag) = (float).5;
        e.ex(ag) = 1 + e.erx(ag);
        e.ey(ag) = 1 + e.ery(ag);
        e.choose_random_theme(ag);
        CLB_WALL_THEME(G) = e.randn(NUM_WALL_THEMES);
        PG_SYNC();
        e.fill_elem(0, 0, G.main_width, 1, WALL_TOP);  // init_floor_and_walls :161-166
        e.fill_elem(0, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(G.main_width - 1, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(0, G.main_height - 1, G.main_width, 1, WALL_MID);
        generate_platforms(e);
    }

    template <class E>
    PG_DEV static int image_for_type(E &e, int type) {  // climber.cpp:145-159
        if (type == PLAYER) {
            const EnvHdr &G = e.G;
            if (!CLB_HAS_SUPPORT(G)) return PLAYER_JUMP;
            if ((double)pg_fabsf(e.evx(G.agent)) < .01 && G.action_vx == 0 && CLB_HAS_SUPPORT(G)) return PLAYER;
            return (G.cur_time / 5 % 2 == 0 || !CLB_HAS_SUPPORT(G)) ? PLAYER_RIGHT1 : PLAYER_RIGHT2;
        } else if (type == ENEMY_BARRIER) {
            return -1;
        }
        return type < 0 ? -type : type;
    }
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &e, int type) { return is_wall(type) ? CLB_WALL_THEME(e.G) : 0; }  // climber.cpp:102-107
    PG_DEV static RectD adjusted_image_rect(int, RectD rect) { return rect; }
    PG_DEV static bool should_preserve_type_themes(int) { return false; }
    template <class E>
    PG_DEV static bool should_draw_entity(E &, int) { return true; }
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &, int) { return 0; }
};

}  // namespace pgamd
