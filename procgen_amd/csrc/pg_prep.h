// pg_prep.h -- the display list of a frame, built ahead of the rasterizer (round 6).
//
// Rounds 1-5 drew a frame with ONE kernel, one wave per env: the wave first worked out what to draw -- the background command, one
// draw command per entity (get_object_rect -> Qt's 16.16 stepping, fp64), the pull form's tables of the grid cells -- and then drew
// it.  The hardware counters of round 5 say that kernel is bound by vector-instruction issue, and the ablation of round 6
// (profiles/r06_valu_by_phase.txt) says where: 1532 of a frame's 5256 vector instructions are the frame set-up and 400 more the entity
// commands, all of it executed with one lane per drawable (~26 of 64 lanes busy for coinrun, 1 of 64 for the background) and with the
// frame's 69 header scalars, the options and seven command words per set held in scalar registers for the whole frame (660-830 spilled
// SGPRs).  So a display-list game splits the work:
//
//   prep<Game>    one wave per PREP_ENVS envs.  (1) Drawables of all its envs -- each env's background and entities -- are laid over the
//                 lanes densely, lane = (env, drawable): the fp64 set-up runs at ~60 of 64 lanes, and the background is just one more
//                 lane of the same code.  Visible commands are packed in draw order into the env's record.  (2) Per env: the grid type
//                 table and the pull form's tables (Renderer::build_type_table / build_pull_tables, unchanged), copied to the record.
//                 A frame the rasterizer's short path cannot draw (per-cell path, more than 192 visible commands, paint_vel_info,
//                 monochrome assets, any error) is queued for the full renderer (Renderer::render_env) and a host-mapped flag is raised:
//                 libenv_observe, having joined the step, then launches render_list<Game> over the queues -- no frame under default
//                 coinrun options in 77 000 emulated ones; a kernel behind every raster launch cost the step 4-7 % in tail latency.  A handle
//                 most of whose frames land there (center_agent = false for a wide world, monochrome assets) goes back to the one-kernel
//                 renderer (libenv_hip.cpp VecGame::read_tail).
//   raster<Game>  one wave per env, Renderer::raster_env: loads the record -- header by scalar loads, the first 64 commands one per lane, the
//                 tables into LDS -- and runs the band passes.  No fp64, no header, no options.
//
// Reference: the same calls as pg_render.h -- BasicAbstractGame::game_draw (src/basic-abstract-game.cpp:1009-1012), draw_background
// (:979-1007), draw_foreground (:921-970), draw_entities (:1052-1066), get_object_rect (:811-817); Game::render_to_buf (src/game.cpp:77-91).
#pragma once
#include "pg_render.h"

namespace pgamd {

// a policy opts in with DISPLAY_LIST = true (no turned / tiled sprites, one background image, no overlay, one command set)
template <class Game, class = void>
struct GameDisplayList {
    static constexpr bool value = false;
};
template <class Game>
struct GameDisplayList<Game, decltype((void)Game::DISPLAY_LIST)> {
    static constexpr bool value = Game::DISPLAY_LIST;
};

#ifndef PG_PREP_ENVS
#define PG_PREP_ENVS 4
#endif
constexpr int PREP_ENVS = PG_PREP_ENVS;  // envs per prep wave: 4 x (1 + ~25 entities) drawables = two dense passes of 64 lanes for coinrun

template <class Game>
struct FramePrep {
    typedef Renderer<Game, false> R;
    typedef RenderLdsT<Game> Lds;
    typedef FrameRec<Game> Rec;
    static_assert(!GameUsesRotation<Game>::value && !GameUsesTiledEntities<Game>::value && !GameTiledBackground<Game>::value &&
                      !GameCustomBackground<Game>::value && !GameHasOverlay<Game>::value && GameRenderCmdSets<Game>::value == 1,
                  "display-list games: upright sprites, one background image, no overlay, one command set");
    const DevCtx &d;
    Lds *lds;
    int *slow_count;  // [0]: envs queued for render_list by this launch
    int *slow_list;

    PG_DEV FramePrep(const DevCtx &d_, Lds *lds_, int *slow_count_, int *slow_list_) : d(d_), lds(lds_), slow_count(slow_count_), slow_list(slow_list_) {}

    // lane-local: the header of the lane's env into its renderer (the optimizer keeps the fields the drawable's set-up reads)
    PG_DEV static void bind_env(R &r, const DevCtx &d, int env) {
        const EnvHdr *h = d.hdr + env;
#define PG_X(type, name) r.G.name = h->name;
        PG_HDR_FIELDS(PG_X)
#undef PG_X
        r.opt = env_options(d.opt, r.G.opt_bits, r.G.opt_debug_mode);
        r.G.error = 0;  // (a sticky error of the env is the step kernels' to report; here: what this frame's set-up raises)
        r.row0 = 0;
        r.row1 = RES_H;
    }
    // lane-local: draw_background BAG:979-1007 as a command (what render_env's set-up computes wave-uniformly)
    PG_DEV static void background_cmd(R &r, uint32_t (&w)[7]) {
        for (int k = 0; k < 7; k++) w[k] = 0;
        if (!r.opt.use_backgrounds) return;
        const ImgDesc bgi = r.d.assets->bg_desc[r.G.background_index];
        const RectD main_rect = r.get_screen_rect(0, (float)r.G.main_height, (float)r.G.main_width, (float)r.G.main_height, 0);
        if (r.G.bg_tile_ratio < 0) {  // (only fruitbot's constructor sets it)
            r.fail(PGE_UNSUPPORTED_DRAW);
            return;
        }
        const float bgw = (float)bgi.w, bgh = (float)bgi.h;
        const float bg_ar = bgw / bgh;
        const float world_ar = (float)(r.G.main_width * 1.0 / r.G.main_height);
        const float extra_w = bg_ar - world_ar;
        const float offset_x = r.G.bg_pct_x * extra_w;
        const RectD bg_rect = adjust_rect(main_rect, (double)(-offset_x), 0, (double)(bg_ar / world_ar), 1);
        r.cmd_image(bgi, false, bg_rect, 1.0f, w[0], w[1], w[2], w[3], w[4], w[5], w[6]);  // RGB32 source: the scale path
    }
    // lane-local: draw_entity BAG:1052-1066 + draw_image BAG:877-913 for entity slot i (Renderer::setup_entities_t's lane body for a game
    // without turned or tiled sprites); z = render_z + 1
    PG_DEV static void entity_cmd(R &r, int i, uint32_t (&w)[7], uint32_t &z) {
        for (int k = 0; k < 7; k++) w[k] = 0;
        z = 0;
        if (!(i < r.G.n_ents && Game::should_draw_entity(r, i))) return;
        const uint32_t mm = r.meta(i);
        z = (uint32_t)(meta_render_z(mm) + 1);
        const float x = r.ex(i), y = r.ey(i), rx = r.erx(i), ry = r.ery(i);
        const float e_alpha = r.ef(EF_ALPHA, i);
        RectD r1 = r.object_rect(mm, x, y, rx, ry);
        const RectD r1_in = r1;
        uint32_t fc = 0;
        ImgDesc imd;
        const int im = r.resolve_image(meta_image_type(mm), meta_image_theme(mm), 0.0f, 0.0f, r1, &fc, &imd);
        const float rotation = r.ef(EF_ROTATION, i);
        const float tile_ratio = Game::tile_aspect_ratio(r, i);
        if (im == R::IMG_FILL) r.cmd_fill_rect(r1_in, fc, w[0], w[5], w[6]);
        if (im >= 0) {
            if (rotation == 0 && tile_ratio != 0) r.fail(PGE_UNSUPPORTED_DRAW);
            else if (rotation == 0) r.cmd_image_fast(imd, (mm & MF_REFLECTED) != 0, r1, e_alpha, w[0], w[1], w[2], w[3], w[4], w[5], w[6]);
            else r.cmd_image_rotated(0, imd, (mm & MF_REFLECTED) != 0, r1, rotation, e_alpha, w[0], w[1], w[2], w[3], w[4], w[5], w[6]);  // a quarter / half turn is a scale; anything else: the full renderer reports it
        }
    }

    // the records of envs [env0, env0 + ne), ne <= PREP_ENVS
    PG_DEV void run(int env0, int ne) {
        // ---- (1) drawables, dense: lane = (env, drawable); drawable 0 of an env is its background, drawable 1 + i its entity slot i
        int start[PREP_ENVS + 1];
        start[0] = 0;
        _Pragma("unroll") for (int e = 0; e < PREP_ENVS; e++) {
            const int n = e < ne ? PG_UNIFORM_I(d.hdr[env0 + (e < ne ? e : 0)].n_ents) : -1;
            start[e + 1] = start[e] + 1 + n;  // (an env past the end contributes nothing)
        }
        const int total = start[PREP_ENVS];
        int ncmd[PREP_ENVS];
        uint32_t slow = 0;  // bit e: env e's frame goes to the full renderer
        _Pragma("unroll") for (int e = 0; e < PREP_ENVS; e++) ncmd[e] = 0;
        for (int base = 0; base < total; base += 64) {
            PG_LANE_ARR(uint32_t, w, 7);
            PG_LANE_VAR(uint32_t, ez);   // env of the lane (0..PREP_ENVS-1) | z << 8 | background << 16 | error << 17 | valid << 18
            PG_FOR_LANES(l) {
                const int q = base + l;
                uint32_t cw[7] = {0, 0, 0, 0, 0, 0, 0}, z = 0, flags = 0;
                int e = 0;
                _Pragma("unroll") for (int k = 1; k < PREP_ENVS; k++) e += q >= start[k] ? 1 : 0;
                if (q < total) {
                    const int slot = q - start[e] - 1;
                    R r(d, env0 + e, lds);
                    bind_env(r, d, env0 + e);
                    if (slot < 0) {
                        background_cmd(r, cw);
                        flags |= 1u << 16;
                    } else {
                        entity_cmd(r, slot, cw, z);
                    }
                    if (r.G.error != 0) flags |= 1u << 17;
                    flags |= 1u << 18;
                }
                _Pragma("unroll") for (int k = 0; k < 7; k++) PG_LA(w, k, l) = cw[k];
                PG_LV(ez, l) = (uint32_t)e | (z << 8) | flags;
            }
            // per env: errors, the background into the header, the visible entity commands packed in draw order
            _Pragma("unroll") for (int e = 0; e < PREP_ENVS; e++) {
                if (e >= ne) continue;
                if (start[e + 1] <= base || start[e] >= base + 64) continue;  // none of this env's drawables in this pass
                uint32_t *rec = d.frame_rec + (size_t)(env0 + e) * Rec::WORDS;
                if (PG_BALLOT(l, (PG_LV(ez, l) & 0xffu) == (uint32_t)e && ((PG_LV(ez, l) >> 17) & 3u) == 3u) != 0) slow |= 1u << e;
                const uint64_t vis = PG_BALLOT(l, (PG_LV(ez, l) & 0xffu) == (uint32_t)e && ((PG_LV(ez, l) >> 16) & 5u) == 4u && PG_LA(w, 0, l) != 0);
                const int cnt = pg_popc64(vis);
                if (ncmd[e] + cnt > Rec::MAX_CMDS) slow |= 1u << e;  // more visible commands than a record holds
                PG_FOR_LANES(l) {
                    const uint32_t f = PG_LV(ez, l);
                    if ((f & 0xffu) == (uint32_t)e && ((f >> 18) & 1u)) {
                        if ((f >> 16) & 1u) {
                            _Pragma("unroll") for (int k = 0; k < 7; k++) rec[Rec::BG + k] = PG_LA(w, k, l);
                        } else if (((vis >> l) & 1ull) && ncmd[e] + cnt <= Rec::MAX_CMDS) {
                            uint32_t *c = rec + Rec::CMD + 8 * (ncmd[e] + pg_popc64(vis & pg_mask_lt(l)));
                            _Pragma("unroll") for (int k = 0; k < 7; k++) c[k] = PG_LA(w, k, l);
                            c[7] = (f >> 8) & 0xffu;
                        }
                    }
                }
                if (ncmd[e] + cnt <= Rec::MAX_CMDS) ncmd[e] += cnt;
            }
        }
        // (the counts move to lanes: the loop below is not unrolled, and a register array indexed by its counter would live in scratch)
        PG_LANE_VAR(uint32_t, ncmd_l);
        PG_FOR_LANES(l) {
            uint32_t v = 0;
            _Pragma("unroll") for (int e = 0; e < PREP_ENVS; e++) v = l == e ? (uint32_t)ncmd[e] : v;
            PG_LV(ncmd_l, l) = v;
        }
        // ---- (2) per env: the grid's tables (wave = env, as the full renderer builds them) and the record's header
        for (int e = 0; e < ne; e++) {
            const int env = env0 + e;
            uint32_t *rec = d.frame_rec + (size_t)env * Rec::WORDS;
            R r(d, env, lds);
            {
                const auto *h = PG_SCALAR_PTR(EnvHdr, d.hdr + env);  // (wave-uniform and read-only here: scalar loads)
#define PG_X(type, name) r.G.name = h->name;
                PG_HDR_FIELDS(PG_X)
#undef PG_X
            }
            r.opt = env_options(d.opt, r.G.opt_bits, r.G.opt_debug_mode);
            r.G.error = 0;
            r.row0 = 0;
            r.row1 = RES_H;
            r.keep_typeimg = false;  // (this kernel's arena ends inside the band buffer)
            int win_lx, win_hx, win_ly, win_hy;  // BAG:926-939
            if (Game::center_agent(r.opt)) {
                const float margin = (float)(r.G.visibility / 2.0 + 1);
                win_lx = (int)(r.G.center_x - margin);
                win_hx = (int)(r.G.center_x + margin);
                win_ly = (int)(r.G.center_y - margin);
                win_hy = (int)(r.G.center_y + margin);
            } else {
                win_lx = 0;
                win_hx = r.G.main_width - 1;
                win_ly = 0;
                win_hy = r.G.main_height - 1;
            }
            const int nx = win_hx - win_lx + 1;
            const int ny_full = win_hy - win_ly + 1;
            uint64_t colseam = 0, rowseam = 0, rowany = ~0ull;
            bool pull = false, multi = false;
            int nfill = 0;
            if constexpr (GameDrawsGrid<Game>::value) {
                const bool try_pull = nx > 0 && ny_full > 0 && nx <= 32 && ny_full <= 32 && nx * ny_full <= GamePullCells<Game>::value;
                PG_LANE_VAR(ImgDesc, type_desc);
                PG_LANE_ARR(int, cells0, 4);
                PG_FOR_LANES(l) {
                    PG_LV(type_desc, l) = r.request_type_desc(l);
                    for (int q = 0; q < 4; q++) PG_LA(cells0, q, l) = 0;
                }
                if (try_pull) r.request_window_cells(win_lx, nx, win_ly, ny_full, cells0);
                r.build_type_table(type_desc);
                pull = try_pull && r.build_pull_tables(win_lx, nx, win_ly, ny_full, colseam, rowseam, rowany, multi, nfill, cells0);
            }
            bool fast = ((slow >> e) & 1u) == 0 && r.G.error == 0 && !r.opt.use_monochrome_assets && !(r.G.has_useful_vel_info && r.opt.paint_vel_info) &&
                        (!GameDrawsGrid<Game>::value || pull);
#if defined(PGAMD_WAVE_EMU)
            if (!fast && getenv("PG_EMU_SLOW_WHY")) fprintf(stderr, "slow frame env %d: slowbit %d error %d mono %d vel %d pull %d nx %d ny %d n_ents %d\n", env, (int)((slow >> e) & 1u), r.G.error, (int)r.opt.use_monochrome_assets, (int)(r.G.has_useful_vel_info && r.opt.paint_vel_info), (int)pull, nx, ny_full, r.G.n_ents);
#endif
            if (PG_FDBG(d, 1048576)) fast = false;  // test aid: every frame through the full renderer
            PG_SYNC();
            // header: the scalars are wave-uniform; lane k takes word k and one store writes them (the background's words were written by its lane)
            uint32_t hw[Rec::HDR_WORDS];
            _Pragma("unroll") for (int k = 0; k < Rec::HDR_WORDS; k++) hw[k] = 0;
            // (bits 8..: why a frame left the short path -- more than 64 visible commands or a failed command, an error, monochrome assets,
            // paint_vel_info, no pull form -- for procgen_amd_display_list_frames' per-env view)
            hw[Rec::FLAGS] = (fast ? Rec::F_FAST : 0u) | (pull ? Rec::F_PULL : 0u) | (multi ? Rec::F_MULTI : 0u) | (((slow >> e) & 1u) << 8) | ((r.G.error != 0 ? 1u : 0u) << 9) |
                             ((r.opt.use_monochrome_assets ? 1u : 0u) << 10) | (((r.G.has_useful_vel_info && r.opt.paint_vel_info) ? 1u : 0u) << 11) |
                             (((GameDrawsGrid<Game>::value && !pull) ? 1u : 0u) << 12) | ((uint32_t)(r.G.error & 0xff) << 16);
            hw[Rec::DIMS] = (uint32_t)(ny_full & 0xff) | ((uint32_t)PG_READLANE(ncmd_l, e) << 8) | ((uint32_t)nfill << 16);
            hw[Rec::COLSEAM] = (uint32_t)colseam;
            hw[Rec::COLSEAM + 1] = (uint32_t)(colseam >> 32);
            hw[Rec::ROWSEAM] = (uint32_t)rowseam;
            hw[Rec::ROWSEAM + 1] = (uint32_t)(rowseam >> 32);
            hw[Rec::ROWANY] = (uint32_t)rowany;
            hw[Rec::ROWANY + 1] = (uint32_t)(rowany >> 32);
            hw[Rec::REF_W] = (uint32_t)d.assets->ref_w;
            PG_FOR_LANES(l) {
                uint32_t v = 0;
                _Pragma("unroll") for (int k = 0; k < Rec::HDR_WORDS; k++) v = l == k ? hw[k] : v;
                if (l < Rec::HDR_WORDS && !(l >= Rec::BG && l < Rec::BG + 7)) rec[l] = v;
            }
            if (fast && pull) {
                // the tables, word by word, coalesced
                const uint32_t *src = reinterpret_cast<const uint32_t *>(lds) + Rec::LDS_TAB_WORD0;
                const int words = multi ? Rec::TAB_WORDS : Rec::TAB_SINGLE_WORDS;
                for (int k0 = 0; k0 < words; k0 += 64) {
                    PG_FOR_LANES(l) {
                        if (k0 + l < words) rec[Rec::TAB + k0 + l] = src[k0 + l];
                    }
                }
            }
            if (!fast) {  // queued for the full renderer (render_list<Game>)
#if defined(PGAMD_WAVE_EMU)
                slow_list[(*slow_count)++] = env;
#else
                if (PG_LANE_ID() == 0) {
                    slow_list[atomicAdd(slow_count, 1)] = env;
                    *reinterpret_cast<volatile int *>(d.slow_flag) = 1;  // (host-mapped: the host sees it when it has joined the step)
                }
#endif
            }
            PG_SYNC();  // (the next env's tables overwrite the arena)
        }
    }
};

}  // namespace pgamd
