// game_coinrun.h -- CoinRun rules as a policy for Env<> (pg_env.h).
// Behaviour follows reference procgen/src/games/coinrun.cpp (cited per function); the structure does not:
// hooks are static functions resolved at compile time (one kernel instantiation per game), level
// generation writes an LDS-resident u8 grid with lane-parallel fills, and the per-step trail spawn is an
// ordered lane-parallel append.
#pragma once
#include "pg_env.h"

namespace pgamd {

struct CoinRun {
    static constexpr int GAME_ID = GAME_COINRUN;
    static constexpr const char *NAME = "coinrun";
    typedef uint8_t cell_t;                        // grid values < 256 (object ids below, SPACE = 100)
    static constexpr int MAX_CELLS = 64 * 64;      // coinrun.cpp:54-55
    static constexpr int WIDE_ROWS = 8;
    static constexpr int RENDER_MIN_WAVES = 5;     // 96 VGPRs, no scratch (tests/test_no_scratch_memory.py), LDS 8068 B: five render waves per SIMD
    static constexpr bool DISPLAY_LIST = true;     // frames are drawn prep -> raster (pg_prep.h)
    static constexpr bool PULL_SINGLE_SIZE = true; // every cell image is 64 x 64 (kenney tiles): no size-class tables in the render arena
    static constexpr int PULL_CELLS = 16 * 16;     // the centred window spans visibility / 2 + 1 = 7.5 cells either side of the agent (BAG:926-933)
    static constexpr bool USES_ENTITY_COLLISIONS = false;  // no entity sets collides_with_entities
    // Worst-case entity count: 5 pit sections x 7 walking enemies x (1 + 9 live trails) + agent = 351.
    // LDS entity-arena tiers of the step kernel (measured demand n + #ENEMY + 3: > 64 in 5.6 % of env-steps,
    // > 128 in 0.5 %, max seen 154)
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 160, ENT_CAP_T2 = 384;
    // Per step the list grows by at most one trail per ENEMY (enemies are only created by a reset); a reset
    // creates <= 42 entities.  One slot is reserved (detached agent).
    // pg_env.h GameParSmart: the agent and the walking enemies are never the target of a sub_step scan (may_interact(., PLAYER)
    // and may_interact(., ENEMY) are false for every type) and their hooks touch nothing but the moving object
    static constexpr bool PAR_SMART = true;
    PG_DEV static bool par_smart_type_ok(int t) { return t == PLAYER || t == ENEMY; }
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) {
        const int n = e.G.n_ents;
        int enemies = 0;
        for (int c = 0; c < ((n + 63) >> 6); c++) enemies += pg_popc64(PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == ENEMY));
        const int need = n + enemies + 3;
        return need < 46 ? 46 : need;
    }

    // object ids coinrun.cpp:11-31
    static constexpr int GOAL = 1, SAW = 2, SAW2 = 3, ENEMY = 5, ENEMY1 = 6, ENEMY2 = 7;
    static constexpr int PLAYER_JUMP = 9, PLAYER_RIGHT1 = 12, PLAYER_RIGHT2 = 13;
    static constexpr int WALL_MID = 15, WALL_TOP = 16, LAVA_MID = 17, LAVA_TOP = 18, ENEMY_BARRIER = 19, CRATE = 20;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == WALL_MID || t == WALL_TOP; }  // coinrun.cpp:183-185: generated as a rect texture (use_generated_assets)
    static constexpr int NUM_GROUND_THEMES = 6;

    // game scalars in EnvHdr (coinrun.cpp:40-47)
#define CR_LAST_AGENT_Y(G) (G).gsf0
#define CR_GRAVITY(G) (G).gsf1
#define CR_AIR_CONTROL(G) (G).gsf2
#define CR_WALL_THEME(G) (G).gsi0
#define CR_HAS_SUPPORT(G) (G).gsi1
#define CR_FACING_RIGHT(G) (G).gsi2
#define CR_IS_ON_CRATE(G) (G).gsi3

    PG_DEV static bool center_agent(const GameOptions &o) { return o.center_agent != 0; }  // left as passed
    PG_DEV static bool is_wall(int t) { return t == WALL_MID || t == WALL_TOP; }
    PG_DEV static bool is_lava(int t) { return t == LAVA_MID || t == LAVA_TOP; }

    // constructor constants: Game::Game (src/game.cpp:25-38), BasicAbstractGame ctor (BAG:22-46), CoinRun ctor (coinrun.cpp:48-58)
    static void construct(EnvHdr &G) {
        G = EnvHdr{};
        G.timeout = 1000;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 13;
        G.min_visibility = 0;
        G.mixrate = 0.2f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.bg_tile_ratio = 0;
        G.out_of_bounds_object = WALL_MID;
        G.has_useful_vel_info = 1;
        G.random_agent_start = 1;
        G.main_width = 64;
        G.main_height = 64;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &) {}  // BAG:377-378

    // ---- physics hooks -----------------------------------------------------------------------------------
    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // BAG:485-492 + coinrun.cpp:204-211
        if (target == WALL_OBJ) return true;
        if (target == e.G.out_of_bounds_object) return true;
        if (src_type == PLAYER && is_wall(target)) return true;
        return false;
    }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool is_horizontal) {  // coinrun.cpp:187-202
        const int ttype = e.etype(target);
        if (ttype == CRATE && !is_horizontal) {
            const int ag = e.G.agent;
            if (e.evy(ag) >= 0) return false;
            if (e.G.action_vy < 0) return false;
            if (CR_LAST_AGENT_Y(e.G) < (e.ey(target) + e.ery(target) + e.ery(ag))) return false;
            CR_IS_ON_CRATE(e.G) = 1;
            return true;
        }
        return is_blocked(e, e.etype(src), ttype, is_horizontal);  // BAG:494-496
    }
    static constexpr bool BLOCKED_ENTS_HAS_EFFECT = true;  // (is_on_crate above)
    template <class E>
    PG_DEV static bool is_blocked_ents_peek(E &e, int src, int target, bool is_horizontal) {  // the predicate without the write
        const int ttype = e.etype(target);
        if (ttype == CRATE && !is_horizontal) {
            const int ag = e.G.agent;
            if (e.evy(ag) >= 0) return false;
            if (e.G.action_vy < 0) return false;
            if (CR_LAST_AGENT_Y(e.G) < (e.ey(target) + e.ery(target) + e.ery(ag))) return false;
            return true;
        }
        return is_blocked(e, e.etype(src), ttype, is_horizontal);
    }
    // superset of the (obj type, entity type) pairs for which is_blocked_ents or will_reflect can be true
    template <class E>
    PG_DEV static bool may_interact(E &e, int src_type, int target_type, bool is_horizontal) {
        return (target_type == CRATE && !is_horizontal) || is_blocked(e, src_type, target_type, is_horizontal) || will_reflect(src_type, target_type);
    }
    PG_DEV static bool will_reflect(int src, int target) {  // coinrun.cpp:140-142
        return src == ENEMY && (is_wall(target) || target == ENEMY_BARRIER);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // coinrun.cpp:123-131
        const int t = e.etype(obj);
        if (t == ENEMY) e.G.done = 1;
        else if (t == SAW) e.G.done = 1;
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &e, int obj, int type, int, int) {  // coinrun.cpp:144-154
        if (e.etype(obj) == PLAYER) {
            if (type == GOAL) {
                e.G.reward += 10.0f;
                e.G.done = 1;
                e.G.level_complete = 1;
            } else if (is_lava(type)) {
                e.G.done = 1;
            }
        }
    }
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}

    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // coinrun.cpp:451-472
        EnvHdr &G = e.G;
        G.action_vx = (float)(move_action / 3 - 1);
        G.action_vy = (float)((move_action % 3) - 1);
        if (G.action_vx > 0) CR_FACING_RIGHT(G) = 1;
        if (G.action_vx < 0) CR_FACING_RIGHT(G) = 0;
        const int ag = G.agent;
        const float ax = e.ex(ag), ay = e.ey(ag), arx = e.erx(ag), ary = e.ery(ag);
        const float by = (float)((double)ay - ((double)ary + .01));
        const int obj_below_1 = e.get_obj_from_floats((float)((double)ax - ((double)arx - .01)), by);
        const int obj_below_2 = e.get_obj_from_floats((float)((double)ax + ((double)arx - .01)), by);
        const bool s1 = is_wall(obj_below_1) || obj_below_1 == G.out_of_bounds_object;
        const bool s2 = is_wall(obj_below_2) || obj_below_2 == G.out_of_bounds_object;
        CR_HAS_SUPPORT(G) = ((CR_IS_ON_CRATE(G) || s1 || s2) && e.evy(ag) == 0) ? 1 : 0;
        CR_IS_ON_CRATE(G) = 0;
        if (G.action_vy == 1) {
            if (!CR_HAS_SUPPORT(G)) G.action_vy = 0;
        }
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // coinrun.cpp:156-173
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const float mixrate_x = CR_HAS_SUPPORT(G) ? G.mixrate : (G.mixrate * CR_AIR_CONTROL(G));
        float vx = (1 - mixrate_x) * e.evx(ag) + mixrate_x * G.maxspeed * G.action_vx;
        if (pg_fabsf(vx) < mixrate_x * G.maxspeed) vx = 0;
        float vy = e.evy(ag);
        if (G.action_vy > 0) {
            vy = G.max_jump;
        } else {
            if (CR_HAS_SUPPORT(G)) vy = (float)((double)vy + .2 * (double)G.action_vy);
        }
        if (!(CR_HAS_SUPPORT(G) && G.action_vy > 0)) {
            vy -= CR_GRAVITY(G);
            vy = clip_abs(vy, G.max_jump);
        }
        e.evx(ag) = vx;
        e.evy(ag) = vy;
    }
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {  // BAG:664-667
        cx = e.ex(e.G.agent);
        cy = e.ey(e.G.agent);
    }

    // ---- CoinRun::game_step coinrun.cpp:474-498 -----------------------------------------------------------
    template <class E>
    PG_DEV static void game_step(E &e) {
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        PG_SYNC();
        // reverse walk over the list: every ENEMY appends one TRAIL (push_back order = descending index) and
        // flips its animation frame; every SAW flips its frame.  Lane-parallel with an ordered append.
        const int n = G.n_ents;
        int added = 0;
        const int cur_time = G.cur_time;
        for (int c = (n - 1) >> 6; c >= 0; c--) {
            const uint64_t em = PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == ENEMY);
            if (n + added + pg_popc64(em) > E_CAP<E>() - 1) {
                e.fail(PGE_ENT_OVERFLOW);
                break;
            }
            PG_FOR_LANES(l) {
                const int i = (c << 6) + l;
                if (i < n) {
                    const int t = e.etype(i);
                    if (t == ENEMY) {
                        const int rank = pg_popc64(em & ~(pg_mask_lt(l) | (1ull << l)));  // enemies above me in this chunk
                        const int slot = n + added + rank;
                        e.ent_init(slot, e.ex(i), (float)((double)e.ey(i) - (double)e.ery(i) * .5), 0, 0.01f, 0.3f, 0.2f, TRAIL);
                        e.ei(EF_EXPIRE_TIME, slot) = 8;
                        e.ef(EF_ALPHA, slot) = (float).5;
                        e.set_image_type(i, cur_time / 5 % 2 == 0 ? ENEMY1 : ENEMY2);
                        e.set_flag(i, MF_REFLECTED, e.evx(i) > 0);
                    } else if (t == SAW) {
                        e.set_image_type(i, cur_time % 2 == 0 ? SAW : SAW2);
                    }
                }
            }
            added += pg_popc64(em);
        }
        G.n_ents = n + added;
        PG_SYNC();
        CR_LAST_AGENT_Y(G) = e.ey(G.agent);
    }
    template <class E>
    PG_DEV static constexpr int E_CAP() { return E::CAPACITY; }

    // ---- level generation coinrun.cpp:227-445 --------------------------------------------------------------
    template <class E>
    PG_DEV static void fill_block_top(E &e, int x, int y, int dx, int dy, int fill, int top) {  // :227-231
        if (!(dy > 0)) {
            e.fail(PGE_ASSERT);
            return;
        }
        e.fill_elem(x, y, dx, dy - 1, fill);
        e.fill_elem(x, y + dy - 1, dx, 1, top);
    }
    template <class E>
    PG_DEV static void fill_ground_block(E &e, int x, int y, int dx, int dy) { fill_block_top(e, x, y, dx, dy, WALL_MID, WALL_TOP); }
    template <class E>
    PG_DEV static void fill_lava_block(E &e, int x, int y, int dx, int dy) { fill_block_top(e, x, y, dx, dy, LAVA_MID, LAVA_TOP); }
    template <class E>
    PG_DEV static void create_saw_enemy(E &e, int x, int y) {  // :248-250
        e.add_entity((float)(x + .5), (float)(y + .5), 0, 0, (float).5, SAW);
    }
    template <class E>
    PG_DEV static void create_enemy(E &e, int x, int y) {  // :252-258
        const float vx = (float)(.15 * (e.randn(2) * 2 - 1));
        const int i = e.add_entity((float)(x + .5), (float)(y + .5), vx, 0, (float).5, ENEMY);
        e.set_flag(i, MF_SMART_STEP, true);
        e.set_image_type(i, ENEMY1);
        e.set_render_z(i, 1);
        e.choose_random_theme(i);
    }
    template <class E>
    PG_DEV static void create_crate(E &e, int x, int y) {  // :260-263
        const int i = e.add_entity((float)(x + .5), (float)(y + .5), 0, 0, (float).5, CRATE);
        e.choose_random_theme(i);
    }

    template <class E>
    PG_DEV static void generate_coin_to_the_right(E &e) {  // :265-414
        EnvHdr &G = e.G;
        const int max_difficulty = 3;
        const int dif = e.randn(max_difficulty) + 1;
        const int num_sections = e.randn(dif) + dif;
        int curr_x = 5, curr_y = 1;
        const int pit_threshold = dif;
        const int danger_type = e.randn(3);
        const bool allow_pit = (e.opt.debug_mode & (1 << 1)) == 0;
        const bool allow_crate = (e.opt.debug_mode & (1 << 2)) == 0;
        const bool allow_dy = (e.opt.debug_mode & (1 << 3)) == 0;
        const int w = G.main_width;
        const float _max_dy = G.max_jump * G.max_jump / (2 * CR_GRAVITY(G));
        const float _max_dx = G.maxspeed * 2 * G.max_jump / CR_GRAVITY(G);
        const int max_dy = (int)((double)_max_dy - .5);
        const int max_dx = (int)((double)_max_dx - .5);
        bool allow_monsters = true;
        if (e.opt.distribution_mode == EasyMode) allow_monsters = false;
        for (int section_idx = 0; section_idx < num_sections; section_idx++) {
            if (curr_x + 15 >= w) break;
            int dy = e.randn(4) + 1 + (int)(dif / 3);
            if (!allow_dy) dy = 0;
            if (dy > max_dy) dy = max_dy;
            if (curr_y >= 20) dy *= -1;
            else if (curr_y >= 5 && e.randn(2) == 1) dy *= -1;
            const int dx = e.randn(2 * dif) + 3 + (int)(dif / 3);
            curr_y += dy;
            if (curr_y < 1) curr_y = 1;
            const bool use_pit = allow_pit && (dx > 7) && (curr_y > 3) && (e.randn(20) >= pit_threshold);
            if (use_pit) {
                const int x1 = e.randn(3) + 1;
                int x2 = e.randn(3) + 1;
                int pit_width = dx - x1 - x2;
                if (pit_width > max_dx) {
                    pit_width = max_dx;
                    x2 = dx - x1 - pit_width;
                }
                fill_ground_block(e, curr_x, 0, x1, curr_y);
                fill_ground_block(e, curr_x + dx - x2, 0, x2, curr_y);
                const int lava_height = e.randn(curr_y - 3) + 1;
                if (danger_type == 0) {
                    fill_lava_block(e, curr_x + x1, 1, pit_width, lava_height);
                } else if (danger_type == 1) {
                    for (int ei = 0; ei < pit_width; ei++) create_saw_enemy(e, curr_x + x1 + ei, 1);
                } else if (danger_type == 2) {
                    for (int ei = 0; ei < pit_width; ei++) create_enemy(e, curr_x + x1 + ei, 1);
                }
                if (pit_width > 4) {
                    int x3, w1;
                    if (pit_width == 5) {
                        x3 = 1 + e.randn(2);
                        w1 = 1 + e.randn(2);
                    } else if (pit_width == 6) {
                        x3 = 2 + e.randn(2);
                        w1 = 1 + e.randn(2);
                    } else {
                        x3 = 2 + e.randn(2);
                        const int x4 = 2 + e.randn(2);
                        w1 = pit_width - x3 - x4;
                    }
                    fill_ground_block(e, curr_x + x1 + x3, curr_y - 1, w1, 1);
                }
            } else {
                fill_ground_block(e, curr_x, 0, dx, curr_y);
                int ob1_x = -1, ob2_x = -1;
                if (e.randn(10) < (2 * dif) && dx > 3) {
                    ob1_x = curr_x + e.randn(dx - 2) + 1;
                    create_saw_enemy(e, ob1_x, curr_y);
                }
                if (e.randn(10) < dif && dx > 3 && (max_dx >= 4) && allow_monsters) {
                    ob2_x = curr_x + e.randn(dx - 2) + 1;
                    create_enemy(e, ob2_x, curr_y);
                }
                if (allow_crate) {
                    for (int i = 0; i < 2; i++) {
                        const int crate_x = curr_x + e.randn(dx - 2) + 1;
                        if (e.randn(2) == 1 && ob1_x != crate_x && ob2_x != crate_x) {
                            const int pile_height = e.randn(3) + 1;
                            for (int j = 0; j < pile_height; j++) create_crate(e, crate_x, curr_y + j);
                        }
                    }
                }
            }
            if (!is_wall(e.get_obj(curr_x - 1, curr_y))) e.set_obj(curr_x - 1, curr_y, ENEMY_BARRIER);
            curr_x += dx;
            e.set_obj(curr_x, curr_y, ENEMY_BARRIER);
            PG_SYNC();
        }
        e.set_obj(curr_x, curr_y, GOAL);
        PG_SYNC();
        fill_ground_block(e, curr_x, 0, 1, curr_y);
        e.fill_elem(curr_x + 1, 0, G.main_width - curr_x - 1, G.main_height, WALL_MID);
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // coinrun.cpp:416-445
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        CR_GRAVITY(G) = 0.2f;
        G.max_jump = 1.5;
        CR_AIR_CONTROL(G) = 0.15f;
        G.maxspeed = (float).5;
        CR_HAS_SUPPORT(G) = 0;
        CR_FACING_RIGHT(G) = 1;
        if (e.opt.distribution_mode == EasyMode) {
            e.set_image_theme(ag, 0);
            CR_WALL_THEME(G) = 0;
            G.background_index = 0;
        } else {
            e.choose_random_theme(ag);
            CR_WALL_THEME(G) = e.randn(NUM_GROUND_THEMES);
        }
        e.erx(ag) = (float).5;
        e.ery(ag) = 0.5787f;
        e.ex(ag) = 1 + e.erx(ag);
        e.ey(ag) = 1 + e.ery(ag);
        CR_LAST_AGENT_Y(G) = e.ey(ag);
        CR_IS_ON_CRATE(G) = 0;
        PG_SYNC();
        // init_floor_and_walls :241-246
        e.fill_elem(0, 0, G.main_width, 1, WALL_TOP);
        e.fill_elem(0, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(G.main_width - 1, 0, 1, G.main_height, WALL_MID);
        e.fill_elem(0, G.main_height - 1, G.main_width, 1, WALL_MID);
        generate_coin_to_the_right(e);
    }

    // ---- drawing hooks --------------------------------------------------------------------------------------
    template <class E>
    PG_DEV static int image_for_type(E &e, int type) {  // coinrun.cpp:213-225
        if (type == PLAYER) {
            const EnvHdr &G = e.G;
            if ((double)pg_fabsf(e.evx(G.agent)) < .01 && G.action_vx == 0 && CR_HAS_SUPPORT(G)) return PLAYER;
            return (G.cur_time / 5 % 2 == 0 || !CR_HAS_SUPPORT(G)) ? PLAYER_RIGHT1 : PLAYER_RIGHT2;
        } else if (type == ENEMY_BARRIER) {
            return -1;
        }
        return type < 0 ? -type : type;  // BAG:438-440
    }
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &e, int type) {  // coinrun.cpp:133-138
        return is_wall(type) ? CR_WALL_THEME(e.G) : 0;
    }
    PG_DEV static RectD adjusted_image_rect(int img_type, RectD rect) {  // coinrun.cpp:64-70
        if (img_type == PLAYER || img_type == PLAYER_JUMP || img_type == PLAYER_RIGHT1 || img_type == PLAYER_RIGHT2)
            return adjust_rect(rect, 0, -.7415, 1, 1.7415);
        return rect;
    }
    PG_DEV static bool should_preserve_type_themes(int) { return false; }  // BAG:446-448
    template <class E>
    PG_DEV static bool should_draw_entity(E &, int) { return true; }  // BAG:1048-1050
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &, int) { return 0; }  // BAG:409-411
};

}  // namespace pgamd
