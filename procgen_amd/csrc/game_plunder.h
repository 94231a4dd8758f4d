// game_plunder.h -- Plunder rules as a policy for Env<> / Renderer<> (reference procgen/src/games/plunder.cpp).
// A gallery shooter on a 20x20 world without grid objects: ships cross in five lanes, the agent's cannon balls
// (entity-entity collisions) must hit the ship type shown in the legend; two HUD bars are painted over the frame.
#pragma once
#include "pg_game_defaults.h"

namespace pgamd {

struct Plunder : BagDefaults<Plunder> {
    static constexpr int GAME_ID = GAME_PLUNDER;
    static constexpr const char *NAME = "plunder";
    static constexpr int MAX_CELLS = 20 * 20;  // plunder.cpp:36-37
    static constexpr bool USES_ENTITY_COLLISIONS = true;
    static constexpr bool USES_ROTATION = true;  // agent and legend ship are drawn turned by -90 / +90 degrees
    static constexpr bool DRAWS_GRID = false;
    static constexpr bool HAS_OVERLAY = true;
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 96, ENT_CAP_T2 = 128;
    static constexpr int PLAYER_BULLET = 1, TARGET_LEGEND = 2, TARGET_BACKGROUND = 3, PANEL = 6, SHIP = 7;
    static constexpr int NUM_LANES = 5, NUM_TOTAL_SHIP_TYPES = 6, NUM_CURRENT_SHIP_TYPES = 2, TARGET_QUOTA = 20;
    static constexpr float SPAWN_PROB = 0.06f, LEGEND_R = 2;

    // one ship, one bullet, one explosion per live bullet, the reserved slot
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) {
        const int n = e.G.n_ents;
        int nb = 0;
        for (int c = 0; c < ((n + 63) >> 6); c++) nb += pg_popc64(PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == PLAYER_BULLET));
        return n + 2 + nb + 1 + 1;
    }

#define PL_LAST_FIRE_TIME(G) (G).gsi0
#define PL_TARGETS_HIT(G) (G).gsi1
#define PL_LANE_DIRS(G) (G).gsi2      // bit k: lane k moves right
#define PL_TARGET_BOOLS(G) (G).gsi3   // bit t: ship theme t is a target
#define PL_IMAGE_PERM(G) (G).gsi4     // 3 bits per entry, 6 entries
#define PL_JUICE_LEFT(G) (G).gsf5
#define PL_MIN_AGENT_X(G) (G).gsf6
    PG_DEV static float lane_vel(const EnvHdr &G, int k) { 
        const float a0 = pg_opaque_f(G.gsf0), a1 = pg_opaque_f(G.gsf1), a2 = pg_opaque_f(G.gsf2), a3 = pg_opaque_f(G.gsf3), a4 = pg_opaque_f(G.gsf4);
        return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : (k == 3 ? a3 : a4)));
    }
    PG_DEV static void set_lane_vel(EnvHdr &G, int k, float v) {  // every field is rewritten: an if / else chain of stores becomes one store at a computed offset (see pg_opaque_f)
        G.gsf0 = k == 0 ? v : G.gsf0;
        G.gsf1 = k == 1 ? v : G.gsf1;
        G.gsf2 = k == 2 ? v : G.gsf2;
        G.gsf3 = k == 3 ? v : G.gsf3;
        G.gsf4 = k >= 4 ? v : G.gsf4;
    }
    PG_DEV static int image_perm(const EnvHdr &G, int k) { return (PL_IMAGE_PERM(G) >> (3 * k)) & 7; }
    template <class O>
    PG_DEV static float r_scale(const O &opt) { return opt.distribution_mode == EasyMode ? 1.5f : 1.0f; }

    PG_DEV static bool center_agent(const GameOptions &) { return false; }  // plunder.cpp:170

    static void construct(EnvHdr &G) {  // plunder.cpp:33-43
        construct_defaults(G);
        G.timeout = 4000;
        G.main_width = 20;
        G.main_height = 20;
        G.mixrate = (float).5;
        G.maxspeed = 0.85f;
        G.has_useful_vel_info = 0;
    }
    PG_DEV static bool should_preserve_type_themes(int type) { return type == SHIP; }  // plunder.cpp:83-85

    template <class E>
    PG_DEV static void handle_collision(E &e, int src, int target) {  // plunder.cpp:87-109
        EnvHdr &G = e.G;
        if (e.etype(src) != PLAYER_BULLET) return;
        const int tt = e.etype(target);
        if (tt == SHIP) {
            e.set_flag(target, MF_WILL_ERASE, true);
            e.set_flag(src, MF_WILL_ERASE, true);
            if ((PL_TARGET_BOOLS(G) >> meta_image_theme(e.meta(target))) & 1) {
                PL_TARGETS_HIT(G) += 1;
                G.reward += 1.0f;
                PL_JUICE_LEFT(G) += 0.1f;
            } else {
                PL_JUICE_LEFT(G) -= 0.1f;
            }
        } else if (tt == PANEL) {
            e.set_flag(src, MF_WILL_ERASE, true);
        }
        if (e.eflag(target, MF_WILL_ERASE))
            e.add_entity(e.ex(target), e.ey(target), e.evx(target) / 2, e.evy(target) / 2, (float)(.5 * (double)e.erx(target)), EXPLOSION);
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // plunder.cpp:111-115
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = 0;
        e.G.action_vrot = 0;
    }

    template <class E>
    PG_DEV static bool agent_has_collision(E &e) {  // BAG:521-529
        const int n = e.G.n_ents;
        for (int c = 0; c < ((n + 63) >> 6); c++)
            if (PG_BALLOT(l, ((c << 6) + l) < n && e.has_agent_collision((c << 6) + l))) return true;
        return false;
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // plunder.cpp:117-184
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        e.set_image_type(ag, SHIP);
        PL_JUICE_LEFT(G) = 1;
        PL_TARGETS_HIT(G) = 0;
        const float rs = r_scale(e.opt);
        {   // RandGen::choose_n (reference src/randgen.cpp:49-69) of all six indices: remaining elements as 3-bit fields
            uint32_t rem = 0;
            for (int i = 0; i < NUM_TOTAL_SHIP_TYPES; i++) rem |= (uint32_t)i << (3 * i);
            int nrem = NUM_TOTAL_SHIP_TYPES;
            uint32_t perm = 0;
            for (int k = 0; k < NUM_TOTAL_SHIP_TYPES; k++) {
                const int idx = e.randn(nrem);
                perm |= ((rem >> (3 * idx)) & 7u) << (3 * k);
                const uint32_t low = rem & ((1u << (3 * idx)) - 1u);
                rem = low | ((rem >> (3 * (idx + 1))) << (3 * idx));
                nrem--;
            }
            PL_IMAGE_PERM(G) = (int)perm;
        }
        PL_TARGET_BOOLS(G) = 0;
        for (int i = 0; i < NUM_CURRENT_SHIP_TYPES / 2; i++) PL_TARGET_BOOLS(G) |= 1 << image_perm(G, i);
        PL_LANE_DIRS(G) = 0;
        for (int i = 0; i < NUM_LANES; i++) {
            if ((double)e.rand01() < .5) PL_LANE_DIRS(G) |= 1 << i;
            set_lane_vel(G, i, (float)(.15 + .1 * (double)e.rand01()));
        }
        const int num_panels = e.opt.distribution_mode == EasyMode ? 0 : e.randn(4);
        const float panel_width = 1.2f;
        PG_SYNC();
        for (int i = 0; i < num_panels; i++)
            e.spawn_entity_rxy(panel_width, (float).5, PANEL, 0, (float)(.25 * G.main_height), (float)G.main_width, (float)(.25 * G.main_height));
        const float key_scale = 1.5;
        e.add_entity(LEGEND_R, LEGEND_R, 0, 0, LEGEND_R, TARGET_BACKGROUND);
        const int lg = e.add_entity(LEGEND_R, LEGEND_R, 0, 0, rs * key_scale, TARGET_LEGEND);
        e.set_image_theme(lg, image_perm(G, 0));
        e.set_image_type(lg, SHIP);
        e.match_aspect_ratio(lg);
        e.ef(EF_ROTATION, lg) = PG_PI / 2;
        PL_LAST_FIRE_TIME(G) = 0;
        e.erx(ag) = rs;
        e.ef(EF_ROTATION, ag) = -1 * PG_PI / 2;
        e.set_image_theme(ag, image_perm(G, e.randn(NUM_CURRENT_SHIP_TYPES / 2) + NUM_CURRENT_SHIP_TYPES / 2));
        e.match_aspect_ratio(ag);
        PG_SYNC();
        {   // reposition_agent BAG:531-539
            int count = 0;
            do {
                e.ex(ag) = e.rand01() * (G.main_width - 2 * e.erx(ag)) + e.erx(ag);
                e.ey(ag) = e.rand01() * (G.main_height - 2 * e.ery(ag)) + e.ery(ag);
                PG_SYNC();
                count++;
            } while (agent_has_collision(e) && (count < 100));
        }
        e.ey(ag) = 1 + e.ery(ag);
        PL_MIN_AGENT_X(G) = 2 * LEGEND_R + e.erx(ag);
        if (e.ex(ag) < PL_MIN_AGENT_X(G)) e.ex(ag) = PL_MIN_AGENT_X(G);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // plunder.cpp:186-239
        e.bag_game_step();
        EnvHdr &G = e.G;
        PL_JUICE_LEFT(G) -= 0.0015f;
        if (e.rand01() < SPAWN_PROB) {
            const float ent_r = r_scale(e.opt);
            const int lane = e.randn(NUM_LANES);
            const float ent_y = (float)((lane * .11 + .4) * (double)(G.main_height / 2 - ent_r) + (double)(G.main_height / 2));
            const bool moves_right = ((PL_LANE_DIRS(G) >> lane) & 1) != 0;
            const float ent_vx = lane_vel(G, lane) * (moves_right ? 1 : -1);
            const int i = G.n_ents;
            if (i >= E::CAPACITY - 1) {
                e.fail(PGE_ENT_OVERFLOW);
            } else {
                e.ent_init(i, 0, ent_y, ent_vx, 0, ent_r, ent_r, SHIP);
                e.set_image_theme(i, image_perm(G, e.randn(NUM_CURRENT_SHIP_TYPES)));
                e.match_aspect_ratio(i);
                e.ex(i) = moves_right ? -1 * ent_r : (G.main_width + ent_r);
                e.set_flag(i, MF_REFLECTED, !moves_right);
                PG_SYNC();
                if (!e.has_any_collision(i, 0)) G.n_ents = i + 1;
            }
        }
        const int ag = G.agent;
        if (G.special_action == 1 && (G.cur_time - PL_LAST_FIRE_TIME(G)) >= 3) {
            const int b = e.add_entity(e.ex(ag), e.ey(ag), 0, 1, (float).25, PLAYER_BULLET);
            e.set_flag(b, MF_COLLIDES, true);
            e.ei(EF_EXPIRE_TIME, b) = 50;
            PL_LAST_FIRE_TIME(G) = G.cur_time;
            PL_JUICE_LEFT(G) -= 0.02f;
        }
        if (PL_JUICE_LEFT(G) <= 0) G.done = 1;
        else if (PL_JUICE_LEFT(G) >= 1) PL_JUICE_LEFT(G) = 1;
        if (PL_TARGETS_HIT(G) >= TARGET_QUOTA) {
            G.done = 1;
            G.reward += 10.0f;
            G.level_complete = 1;
        }
        if (e.ex(ag) < PL_MIN_AGENT_X(G)) e.ex(ag) = PL_MIN_AGENT_X(G);
        PG_SYNC();
    }

    // game_draw override plunder.cpp:65-77: two bars over the finished frame
    template <class R>
    PG_DEV static void draw_overlay(R &r) {
        const EnvHdr &G = r.G;
        const float w1 = G.main_width * PL_JUICE_LEFT(G);
        const float w2 = (float)(G.main_width * (PL_TARGETS_HIT(G) * 1.0 / TARGET_QUOTA));
        r.exec_fill(r.get_abs_rect((float).25, (float).25, w1, (float).5), 0xff42f587u);
        r.exec_fill(r.get_abs_rect((float).25, (float).75, w2, (float).5), 0xfff54290u);
    }
};

}  // namespace pgamd
