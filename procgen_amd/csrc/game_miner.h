// game_miner.h -- Miner rules as a policy for Env<> / Renderer<> (reference procgen/src/games/miner.cpp).
// Boulder-dash gravity: every step the reference sweeps the whole grid in ascending cell order; here each 64-cell
// chunk is balloted for round objects and only those are visited (in the same order, with cells a boulder rolls
// into re-armed), since all other cells are no-ops of the sweep.
#pragma once
#include "pg_env.h"

namespace pgamd {

struct MinerScratch {
    uint8_t was_dirt[35 * 35 + 7];  // get_cells_with_type(DIRT) snapshot during game_reset
};

struct Miner {
    static constexpr int GAME_ID = GAME_MINER;
    static constexpr bool DISPLAY_LIST = true;  // frames are drawn prep -> raster (pg_prep.h)
    static constexpr int RENDER_MIN_WAVES = 5;  // the renderer fits 96 VGPRs without scratch and 8136 B of LDS: five waves per SIMD (kernels_game.hip)
    static constexpr const char *NAME = "miner";
    typedef uint8_t cell_t;
    static constexpr int MAX_CELLS = 35 * 35;  // memory mode (miner.cpp:124-126)
    static constexpr bool USES_ENTITY_COLLISIONS = false;
    static constexpr int ENT_CAP_T0 = 8, ENT_CAP_T1 = 16, ENT_CAP_T2 = 32;  // agent + exit
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 3; }
    typedef MinerScratch Scratch;

    static constexpr int BOULDER = 1, DIAMOND = 2, MOVING_BOULDER = 3, MOVING_DIAMOND = 4, ENEMY = 5, EXIT = 6, DIRT = 9, OOB_WALL = 10;
#define MN_DIAMONDS_REMAINING(G) (G).gsi0

    PG_DEV static bool center_agent(const GameOptions &o) { return o.distribution_mode == MemoryMode; }  // miner.cpp:140

    static void construct(EnvHdr &G) {  // Game::Game, BAG ctor (BAG:22-46), MinerGame ctor (miner.cpp:25-35)
        G = EnvHdr{};
        G.timeout = 1000;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 8.0f;
        G.min_visibility = 0;
        G.mixrate = 0.5f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.out_of_bounds_object = OOB_WALL;
        G.has_useful_vel_info = 0;
        G.random_agent_start = 1;
        G.main_width = 20;
        G.main_height = 20;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // miner.cpp:116-129
        const int dm = e.opt.distribution_mode;
        int dim = e.G.main_width;
        if (dm == EasyMode) dim = 10;
        else if (dm == HardMode) dim = 20;
        else if (dm == MemoryMode) dim = 35;
        e.G.main_width = dim;
        e.G.main_height = dim;
    }

    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // BAG:485-492 + miner.cpp:57-64
        if (target == WALL_OBJ) return true;
        if (target == e.G.out_of_bounds_object) return true;
        if (src_type == PLAYER && (target == BOULDER || target == MOVING_BOULDER || target == OOB_WALL)) return true;
        return false;
    }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool h) { return is_blocked(e, e.etype(src), e.etype(target), h); }
    PG_DEV static bool will_reflect(int src, int target) {  // miner.cpp:66-68 (out_of_bounds_object == OOB_WALL)
        return src == ENEMY && (target == BOULDER || target == DIAMOND || target == MOVING_BOULDER || target == MOVING_DIAMOND || target == OOB_WALL);
    }
    template <class E>
    PG_DEV static bool may_interact(E &e, int s, int t, bool h) { return is_blocked(e, s, t, h) || will_reflect(s, t); }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // miner.cpp:70-82
        const int t = e.etype(obj);
        if (t == ENEMY) {
            e.G.done = 1;
        } else if (t == EXIT) {
            if (MN_DIAMONDS_REMAINING(e.G) == 0) {
                e.G.reward += 10.0f;
                e.G.level_complete = 1;
                e.G.done = 1;
            }
        }
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // BAG:658-662 + miner.cpp:98-102
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = (float)(move_action % 3 - 1);
        e.G.action_vrot = 0;
        if (e.G.action_vx != 0) e.G.action_vy = 0;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) { e.bag_update_agent_velocity(1.0f); }  // unused: grid_step
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {
        cx = e.ex(e.G.agent);
        cy = e.ey(e.G.agent);
    }

    // index-based grid access: BAG:198-203 (get_obj(idx)), grid.h:59-62 (set_index)
    template <class E>
    PG_DEV static int get_idx(E &e, int idx) {
        if (!(0 <= idx && idx < e.G.main_width * e.G.main_height)) return e.G.out_of_bounds_object;
        return (int)e.s->grid[idx];
    }
    template <class E>
    PG_DEV static void set_idx(E &e, int idx, int v) {
        if (!(0 <= idx && idx < e.G.main_width * e.G.main_height)) {
            e.fail(PGE_GRID_OOB);
            return;
        }
        PG_FOR_LANES(l) {
            if (l == 0) e.s->grid[idx] = (cell_t)v;
        }
        e.G.grid_dirty = 1;
    }
    PG_DEV static bool is_round(int t) { return t == BOULDER || t == MOVING_BOULDER || t == DIAMOND || t == MOVING_DIAMOND; }
    PG_DEV static int stationary(int t) { return t == MOVING_DIAMOND ? DIAMOND : (t == MOVING_BOULDER ? BOULDER : t); }
    PG_DEV static int moving(int t) { return t == DIAMOND ? MOVING_DIAMOND : (t == BOULDER ? MOVING_BOULDER : t); }

    template <class E>
    PG_DEV static void game_step(E &e) {  // miner.cpp:247-307
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const int w = G.main_width;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        PG_SYNC();
        {   // handle_push miner.cpp:232-245
            const int agent_idx = (int)e.ey(ag) * w + (int)e.ex(ag);
            const int agentx = agent_idx % w;
            if (G.action_vx == 1 && e.evx(ag) == 0 && agentx < w - 2 && get_idx(e, agent_idx + 1) == BOULDER && get_idx(e, agent_idx + 2) == SPACE) {
                set_idx(e, agent_idx + 1, SPACE);
                set_idx(e, agent_idx + 2, BOULDER);
                e.ex(ag) += 1;
            } else if (G.action_vx == -1 && e.evx(ag) == 0 && agentx > 1 && get_idx(e, agent_idx - 1) == BOULDER && get_idx(e, agent_idx - 2) == SPACE) {
                set_idx(e, agent_idx - 1, SPACE);
                set_idx(e, agent_idx - 2, BOULDER);
                e.ex(ag) -= 1;
            }
            PG_SYNC();
        }
        const float ax = e.ex(ag), ay = e.ey(ag);
        const int agent_obj = e.get_obj((int)ax, (int)ay);
        if (agent_obj == DIAMOND) G.reward += 1.0f;
        if (agent_obj == DIRT || agent_obj == DIAMOND) e.set_obj((int)ax, (int)ay, SPACE);
        PG_SYNC();
        const int main_area = w * G.main_height;
        const int agent_cell = (int)ay * w + (int)ax;                                    // get_agent_index() miner.cpp:94-96
        const int agent_idx = (int)(((double)ay - .5) * w + ((double)ax - .5));          // miner.cpp:271
        int diamonds_count = 0;
        for (int base = 0; base < main_area; base += 64) {
            uint64_t m = PG_BALLOT(l, (base + l) < main_area && is_round((int)e.s->grid[base + l]));
            {
                // Settled objects (a resting boulder / diamond on dirt, rock or the floor) take the last branch below and
                // change nothing; nothing the sweep does earlier can unsettle them (it only writes to free cells and
                // to the cells of objects it moves), so they are counted here and leave the serial work list.
                PG_LANE_VAR(int, own);
                const uint64_t settled = PG_BALLOT(l, ({
                                                       const int idx = base + l;
                                                       bool st = false;
                                                       PG_LV(own, l) = 0;
                                                       if (idx < main_area) {
                                                           const int obj = (int)e.s->grid[idx];
                                                           PG_LV(own, l) = obj;
                                                           if (obj == BOULDER || obj == DIAMOND) {
                                                               const int below = get_idx(e, idx - w);
                                                               st = below != SPACE && !is_round(below);
                                                           }
                                                       }
                                                       st;
                                                   }));
                diamonds_count += pg_popc64(settled & PG_BALLOT(l, PG_LV(own, l) == DIAMOND));
                m &= ~settled;
            }
            while (m) {
                const int bit = pg_ctz64(m);
                m &= m - 1;
                const int idx = base + bit;
                const int obj = (int)e.s->grid[idx];
                if (!is_round(obj)) continue;  // re-armed bit whose cell was not filled after all
                const int obj_x = idx % w;
                const int stat_type = stationary(obj);
                if (stat_type == DIAMOND) diamonds_count++;
                const int below_idx = idx - w;
                const int obj2 = get_idx(e, below_idx);
                const bool agent_is_below = agent_idx == below_idx;
                auto is_free = [&](int i) { return get_idx(e, i) == SPACE && agent_cell != i; };
                if (obj2 == SPACE && !agent_is_below) {
                    set_idx(e, idx, SPACE);
                    set_idx(e, below_idx, moving(obj));
                } else if (agent_is_below && (obj == MOVING_BOULDER || obj == MOVING_DIAMOND)) {
                    G.done = 1;
                } else if (is_round(obj2) && obj_x > 0 && is_free(idx - 1) && is_free(idx - w - 1)) {
                    set_idx(e, idx, SPACE);
                    set_idx(e, idx - 1, stationary(obj));
                } else if (is_round(obj2) && obj_x < w - 1 && is_free(idx + 1) && is_free(idx - w + 1)) {
                    set_idx(e, idx, SPACE);
                    set_idx(e, idx + 1, stat_type);
                    if (bit < 63) m |= 1ull << (bit + 1);  // the sweep reaches idx + 1 next; in the next chunk the ballot sees it
                } else {
                    set_idx(e, idx, stat_type);
                }
                PG_SYNC();
            }
        }
        MN_DIAMONDS_REMAINING(G) = diamonds_count;
        // miner.cpp:299-305: no ENEMY entity is ever created, the random re-targeting loop never draws
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // miner.cpp:131-205
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const int w = G.main_width;
        e.erx(ag) = (float).5;
        e.ery(ag) = (float).5;
        const int main_area = G.main_height * w;
        G.grid_step = 1;
        const float diamond_pct = 12 / 400.0f, boulder_pct = 80 / 400.0f;
        const int num_diamonds = (int)(diamond_pct * main_area);
        const int num_boulders = (int)(boulder_pct * main_area);
        // all cells DIRT (miner.cpp:159-161); RandGen::simple_choose (randgen.cpp:70-88) then rejects cells that are
        // no longer DIRT -- exactly the cells it chose before -- so the std::set is the grid itself
        e.fill_elem(0, 0, w, G.main_height, DIRT);
        const int MARK = 255;
        int agent_cell = 0;
        for (int i = 0; i < num_diamonds + num_boulders + 1; i++) {
            int next = e.randn(main_area);
            while ((int)e.s->grid[next] != DIRT) next = e.randn(main_area);
            if (i == 0) agent_cell = next;
            set_idx(e, next, i == 0 ? MARK : (i <= num_diamonds ? DIAMOND : BOULDER));
            PG_SYNC();
        }
        set_idx(e, agent_cell, DIRT);
        PG_SYNC();
        const int agent_x = agent_cell % w, agent_y = agent_cell / w;
        e.ex(ag) = (float)(agent_x + .5);
        e.ey(ag) = (float)(agent_y + .5);
        for (int base = 0; base < main_area; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < main_area) e.s->scratch.was_dirt[base + l] = (uint8_t)((int)e.s->grid[base + l] == DIRT);
            }
        }
        PG_SYNC();
        e.set_obj(agent_x, agent_y, SPACE);
        for (int i = -1; i <= 1; i++)
            for (int j = -1; j <= 1; j++) {
                const int ox = agent_x + i, oy = agent_y + j;
                if (e.get_obj(ox, oy) == BOULDER) e.set_obj(ox, oy, DIRT);
            }
        PG_SYNC();
        // exit candidates in ascending cell order (miner.cpp:186-199): count, draw, then select the n-th
        auto cand_mask = [&](int base) -> uint64_t {
            return PG_BALLOT(l, ({
                                 const int cell = base + l;
                                 bool c = false;
                                 if (cell < main_area && e.s->scratch.was_dirt[cell]) {
                                     const int above = get_idx(e, cell + w);
                                     c = above == DIRT || above == e.G.out_of_bounds_object;
                                 }
                                 c;
                             }));
        };
        int ncand = 0;
        for (int base = 0; base < main_area; base += 64) ncand += pg_popc64(cand_mask(base));
        if (ncand <= 0) {
            e.fail(PGE_ASSERT);
            return;
        }
        int pick = e.randn(ncand);
        int exit_cell = -1;
        for (int base = 0; base < main_area && exit_cell < 0; base += 64) {
            uint64_t m = cand_mask(base);
            const int c = pg_popc64(m);
            if (pick < c) {
                for (int k = 0; k < pick; k++) m &= m - 1;
                exit_cell = base + pg_ctz64(m);
            } else {
                pick -= c;
            }
        }
        set_idx(e, exit_cell, SPACE);
        PG_SYNC();
        const int ex = e.add_entity((float)((exit_cell % w) + .5), (float)((exit_cell / w) + .5), 0, 0, (float).5, EXIT);
        e.set_render_z(ex, -1);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static int image_for_type(E &, int type) {  // miner.cpp:84-92
        if (type == MOVING_BOULDER) return BOULDER;
        if (type == MOVING_DIAMOND) return DIAMOND;
        return type < 0 ? -type : type;
    }
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
