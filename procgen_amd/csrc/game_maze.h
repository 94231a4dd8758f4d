// game_maze.h -- Maze rules as a policy for Env<> / Renderer<> (reference procgen/src/games/maze.cpp).
// Grid-step movement through a Kruskal maze (pg_mazegen.h); the only entity is the agent.
#pragma once
#include "pg_mazegen.h"

namespace pgamd {

struct Maze {
    static constexpr int GAME_ID = GAME_MAZE;
    static constexpr bool DISPLAY_LIST = true;  // frames are drawn prep -> raster (pg_prep.h)
    static constexpr int RENDER_MIN_WAVES = 5;  // the renderer fits 96 VGPRs without scratch and 8136 B of LDS: five waves per SIMD (kernels_game.hip)
    static constexpr const char *NAME = "maze";
    typedef uint8_t cell_t;
    static constexpr int MAX_CELLS = 31 * 31;  // memory mode world_dim (maze.cpp:47)
    static constexpr bool USES_ENTITY_COLLISIONS = false;
    static constexpr int ENT_CAP_T0 = 8, ENT_CAP_T1 = 16, ENT_CAP_T2 = 32;  // the agent is the only entity
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 3; }
    typedef MazeScratchT<31, false> Scratch;  // maze_dim <= world_dim <= 31 (memory mode, maze.cpp:40-53)

    static constexpr int GOAL = 2;
#define MZ_MAZE_DIM(G) (G).gsi0
#define MZ_WORLD_DIM(G) (G).gsi1

    PG_DEV static bool center_agent(const GameOptions &o) { return o.distribution_mode == MemoryMode; }  // maze.cpp:66

    static void construct(EnvHdr &G) {  // Game::Game, BAG ctor (BAG:22-46), MazeGame ctor (maze.cpp:16-24)
        G = EnvHdr{};
        G.timeout = 500;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 8.0f;
        G.min_visibility = 0;
        G.mixrate = 0.5f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.out_of_bounds_object = WALL_OBJ;
        G.has_useful_vel_info = 0;
        G.random_agent_start = 0;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // maze.cpp:40-53
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        if (dm == EasyMode) MZ_WORLD_DIM(G) = 15;
        else if (dm == HardMode) MZ_WORLD_DIM(G) = 25;
        else if (dm == MemoryMode) MZ_WORLD_DIM(G) = 31;
        G.main_width = MZ_WORLD_DIM(G);
        G.main_height = MZ_WORLD_DIM(G);
    }

    template <class E>
    PG_DEV static bool is_blocked(E &e, int, int target, bool) { return target == WALL_OBJ || target == e.G.out_of_bounds_object; }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool h) { return is_blocked(e, e.etype(src), e.etype(target), h); }
    PG_DEV static bool will_reflect(int, int) { return false; }
    template <class E>
    PG_DEV static bool may_interact(E &e, int s, int t, bool h) { return is_blocked(e, s, t, h); }
    template <class E>
    PG_DEV static void handle_agent_collision(E &, int) {}
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &, int, int) {}
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // BAG:658-662 + maze.cpp:99-103
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

    template <class E>
    PG_DEV static void game_step(E &e) {  // maze.cpp:105-124
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, true);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, false);
        const int ix = (int)e.ex(ag), iy = (int)e.ey(ag);
        if (e.get_obj(ix, iy) == GOAL) {
            e.set_obj(ix, iy, SPACE);
            G.reward += 10.0f;
            G.level_complete = 1;
        }
        G.done = G.reward > 0;
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // maze.cpp:55-97
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        G.grid_step = 1;
        const int world_dim = MZ_WORLD_DIM(G);
        const int maze_dim = e.randn((world_dim - 1) / 2) * 2 + 3;
        MZ_MAZE_DIM(G) = maze_dim;
        const int margin = (world_dim - maze_dim) / 2;
        e.erx(ag) = (float).5;
        e.ery(ag) = (float).5;
        e.ex(ag) = (float)(margin + .5);
        e.ey(ag) = (float)(margin + .5);
        PG_SYNC();
        MazeGenDev<E, Scratch> mg(e, e.s->scratch, maze_dim);
        mg.generate_maze();
        mg.place_objects(GOAL, 1);
        e.fill_elem(0, 0, world_dim, world_dim, WALL_OBJ);
        for (int j = 0; j < maze_dim; j++) {  // maze cell (i, j) -> world (margin + i, margin + j)
            PG_FOR_LANES(l) {
                if (l < maze_dim) e.s->grid[(margin + j) * world_dim + margin + l] = (cell_t)mg.grid_at(l + MAZE_OFFSET, j + MAZE_OFFSET);
            }
        }
        PG_SYNC();
        if (margin > 0) {  // already WALL_OBJ from the fill above; kept for the reference's fassert semantics (in-range indices)
            e.fill_elem(margin - 1, margin - 1, 1, maze_dim + 2, WALL_OBJ);
            e.fill_elem(margin + maze_dim, margin - 1, 1, maze_dim + 2, WALL_OBJ);
            e.fill_elem(margin - 1, margin - 1, maze_dim + 2, 1, WALL_OBJ);
            e.fill_elem(margin - 1, margin + maze_dim, maze_dim + 2, 1, WALL_OBJ);
        }
        G.grid_dirty = 1;
    }

    template <class E>
    PG_DEV static int image_for_type(E &, int type) { return type < 0 ? -type : type; }
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
