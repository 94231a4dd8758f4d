// pg_game_defaults.h -- the BasicAbstractGame defaults of the policy hooks (reference src/basic-abstract-game.cpp,
// "BAG"), for game policies to inherit: `struct Foo : BagDefaults<Foo> { ...overrides... }`.  Hooks that call other
// hooks go through the derived policy (CRTP), which is what the reference's virtual dispatch does.
#pragma once
#include "pg_env.h"

namespace pgamd {

template <class Game>
struct BagDefaults {
    typedef uint8_t cell_t;
    static constexpr bool USES_ENTITY_COLLISIONS = false;

    PG_DEV static bool center_agent(const GameOptions &o) { return o.center_agent != 0; }

    // Game::Game (reference src/game.cpp:25-38) + BasicAbstractGame ctor (BAG:22-46)
    static void construct_defaults(EnvHdr &G) {
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
        G.out_of_bounds_object = INVALID_OBJ;
        G.has_useful_vel_info = 1;
        G.random_agent_start = 1;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &) {}  // BAG:377-378
    template <class E>
    PG_DEV static bool is_blocked(E &e, int, int target, bool) {  // BAG:485-492
        return target == WALL_OBJ || target == e.G.out_of_bounds_object;
    }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool is_horizontal) {  // BAG:494-496
        return Game::is_blocked(e, e.etype(src), e.etype(target), is_horizontal);
    }
    PG_DEV static bool will_reflect(int, int) { return false; }  // BAG:498-500
    // could an entity of target_type block or reflect one of src_type?  (filter of the entity scan in sub_step)
    template <class E>
    PG_DEV static bool may_interact(E &e, int src_type, int target_type, bool is_horizontal) {
        return Game::is_blocked(e, src_type, target_type, is_horizontal) || Game::will_reflect(src_type, target_type);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &, int) {}  // BAG:383-384
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}  // BAG:398
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // BAG:658-662
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = (float)(move_action % 3 - 1);
        e.G.action_vrot = 0;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) { e.bag_update_agent_velocity(1.0f); }  // BAG:669-684
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {  // BAG:664-667
        cx = e.ex(e.G.agent);
        cy = e.ey(e.G.agent);
    }

    // drawing
    template <class E>
    PG_DEV static int image_for_type(E &, int type) { return type < 0 ? -type : type; }  // BAG:438-440
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &, int) { return 0; }
    PG_DEV static RectD adjusted_image_rect(int, RectD rect) { return rect; }
    PG_DEV static bool should_preserve_type_themes(int) { return false; }
    template <class E>
    PG_DEV static bool should_draw_entity(E &, int) { return true; }
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &, int) { return 0; }  // BAG:409-411
};

}  // namespace pgamd
