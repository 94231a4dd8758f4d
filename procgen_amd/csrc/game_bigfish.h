// game_bigfish.h -- BigFish rules as a policy for Env<> / Renderer<> (reference procgen/src/games/bigfish.cpp).
// Entity-dense collision path of BASELINE configs[2]: no grid objects, fish drift across a 20x20 world, the agent
// eats smaller fish (its radius grows inside the collision pass, so the pass re-ballots after every handler).
#pragma once
#include "pg_env.h"

namespace pgamd {

struct BigFish {
    static constexpr int GAME_ID = GAME_BIGFISH;
    static constexpr bool DISPLAY_LIST = true;  // frames are drawn prep -> raster (pg_prep.h)
    static constexpr const char *NAME = "bigfish";
    typedef uint8_t cell_t;
    static constexpr int MAX_CELLS = 20 * 20;  // bigfish.cpp:29-30 (padded to a 16-byte multiple below)
    static constexpr bool USES_ENTITY_COLLISIONS = false;
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 128, ENT_CAP_T2 = 256;  // <= 1 spawn per step; ~5-15 fish alive
    static constexpr bool DRAWS_GRID = false;  // the grid holds only SPACE
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 1 + 3; }

    static constexpr int FISH = 2;
    static constexpr float FISH_MIN_R = .25f, FISH_MAX_R = 2.0f;
    static constexpr int FISH_QUOTA = 30;

#define BF_FISH_EATEN(G) (G).gsi0
#define BF_R_INC(G) (G).gsf0

    // options.center_agent = false in game_reset (bigfish.cpp:64); the first frame is drawn after the first reset
    PG_DEV static bool center_agent(const GameOptions &) { return false; }

    static void construct(EnvHdr &G) {  // Game::Game, BasicAbstractGame ctor (BAG:22-46), BigFish ctor (bigfish.cpp:25-31)
        G = EnvHdr{};
        G.timeout = 6000;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 16;
        G.min_visibility = 0;
        G.mixrate = 0.5f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.out_of_bounds_object = INVALID_OBJ;
        G.has_useful_vel_info = 1;
        G.random_agent_start = 1;
        G.main_width = 20;
        G.main_height = 20;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &) {}  // BAG:377-378

    // ---- physics hooks: all BasicAbstractGame defaults -------------------------------------------------------
    template <class E>
    PG_DEV static bool is_blocked(E &e, int, int target, bool) {  // BAG:485-492
        return target == WALL_OBJ || target == e.G.out_of_bounds_object;
    }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool is_horizontal) {  // BAG:494-496
        return is_blocked(e, e.etype(src), e.etype(target), is_horizontal);
    }
    PG_DEV static bool will_reflect(int, int) { return false; }  // BAG:498-500
    template <class E>
    PG_DEV static bool may_interact(E &e, int src_type, int target_type, bool is_horizontal) {
        return is_blocked(e, src_type, target_type, is_horizontal);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // bigfish.cpp:48-62
        if (e.etype(obj) == FISH) {
            const int ag = e.G.agent;
            if (e.erx(obj) > e.erx(ag)) {
                e.G.done = 1;
            } else {
                e.G.reward += 1.0f;
                e.set_flag(obj, MF_WILL_ERASE, true);
                e.erx(ag) += BF_R_INC(e.G);
                e.ery(ag) += BF_R_INC(e.G);
                BF_FISH_EATEN(e.G) += 1;
            }
        }
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // BAG:658-662
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = (float)(move_action % 3 - 1);
        e.G.action_vrot = 0;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) { e.bag_update_agent_velocity(1.0f); }
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {
        cx = e.ex(e.G.agent);
        cy = e.ey(e.G.agent);
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // bigfish.cpp:80-107
        e.bag_game_step();
        EnvHdr &G = e.G;
        if (e.randn(10) == 1) {
            const float ent_r = (float)((double)(FISH_MAX_R - FISH_MIN_R) * pg_pow((double)e.rand01(), 1.4) + (double)FISH_MIN_R);
            const float ent_y = e.rand01() * (G.main_height - 2 * ent_r);
            const float moves_right = e.rand01() < .5;
            const float ent_vx = (float)((.15 + (double)e.rand01() * .25) * (moves_right ? 1 : -1));
            const float ent_x = moves_right ? -1 * ent_r : G.main_width + ent_r;
            const int i = e.add_entity(ent_x, ent_y, ent_vx, 0, ent_r, FISH);
            e.choose_random_theme(i);
            e.match_aspect_ratio(i);
            e.set_flag(i, MF_REFLECTED, !moves_right);
        }
        if (BF_FISH_EATEN(G) >= FISH_QUOTA) {
            G.done = 1;
            G.reward += 10.0f;
            G.level_complete = 1;
        }
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // bigfish.cpp:64-81
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        BF_FISH_EATEN(G) = 0;
        float start_r = (float).5;
        if (e.opt.distribution_mode == EasyMode) start_r = 1;
        BF_R_INC(G) = (FISH_MAX_R - start_r) / FISH_QUOTA;
        e.erx(ag) = start_r;
        e.ery(ag) = start_r;
        e.ey(ag) = 1 + e.ery(ag);
        PG_SYNC();
    }

    // ---- drawing hooks: defaults ---------------------------------------------------------------------------------
    template <class E>
    PG_DEV static int image_for_type(E &, int type) { return type < 0 ? -type : type; }  // BAG:438-440
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &, int) { return 0; }
    PG_DEV static RectD adjusted_image_rect(int, RectD rect) { return rect; }
    PG_DEV static bool should_preserve_type_themes(int) { return false; }
    template <class E>
    PG_DEV static bool should_draw_entity(E &, int) { return true; }
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &, int) { return 0; }
};

}  // namespace pgamd
