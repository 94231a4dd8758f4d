// libenv_hip.cpp -- the drop-in boundary: the gym3 libenv C ABI (include/libenv.h) plus get_state / set_state
// and the procgen_amd_* extension hooks (include/procgen_amd.h), implemented over the gfx950 kernels.
//
// Replaces the reference's VecGame (reference src/vecgame.cpp): the thread pool / work queue becomes one kernel
// launch per libenv_act on a HIP stream, and libenv_observe is the stream join (+ the D2H landing of the
// boundary buffers).  Option parsing follows reference src/vecoptions.cpp (consume by name and dtype; anything
// left over is fatal), errors follow reference src/cpp-utils.cpp (message + exit(EXIT_FAILURE)).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/libenv.h"
#include "../../include/procgen_amd.h"
#include "assets.h"
#include "kernels.h"
#include "shard_map.h"
#include "state_io.h"

using namespace pgamd;

namespace {

[[noreturn]] void fatal(const char *fmt, ...) {  // reference src/cpp-utils.cpp:8-19
    printf("fatal: ");
    va_list args;
    va_start(args, fmt);
    vprintf(fmt, args);
    va_end(args);
    fflush(stdout);
    if (const char *path = getenv("PROCGEN_AMD_FATAL_LOG")) {  // the message also goes to a file: a test runner that loses the dying process's stdout still has it
        if (FILE *f = fopen(path, "a")) {
            va_list again;
            va_start(again, fmt);
            fprintf(f, "[pid %d] fatal: ", (int)getpid());
            vfprintf(f, fmt, again);
            va_end(again);
            fclose(f);
        }
    }
    exit(EXIT_FAILURE);
}

#define HIP_CHECK(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) fatal("%s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- reference src/vecoptions.cpp ----------------------------------------------------------------------
class VecOptions {
  public:
    explicit VecOptions(const struct libenv_options options) : m_options(options.items, options.items + options.count) {}
    bool consume_string(const std::string &name, std::string *value) {
        libenv_option opt;
        if (!find_option(name, LIBENV_DTYPE_UINT8, &opt)) return false;
        *value = std::string((char *)opt.data, opt.count);
        return true;
    }
    bool consume_int(const std::string &name, int32_t *value) {
        libenv_option opt;
        if (!find_option(name, LIBENV_DTYPE_INT32, &opt)) return false;
        *value = *(int32_t *)opt.data;
        return true;
    }
    bool consume_bool(const std::string &name, bool *value) {
        libenv_option opt;
        if (!find_option(name, LIBENV_DTYPE_UINT8, &opt)) return false;
        uint8_t v = *(uint8_t *)opt.data;
        if (!(v == 0 || v == 1)) fatal("option %s is not a bool\n", name.c_str());
        *value = (bool)v;
        return true;
    }
    void ensure_empty() {
        if (!m_options.empty()) fatal("unused options found, first unused option: %s\n", m_options[0].name);
    }

  private:
    std::vector<libenv_option> m_options;
    bool find_option(const std::string &name, enum libenv_dtype dtype, libenv_option *out) {
        for (size_t idx = 0; idx < m_options.size(); idx++) {
            const libenv_option &opt = m_options[idx];
            if (name == std::string(opt.name, strnlen(opt.name, LIBENV_MAX_NAME_LEN))) {
                if (opt.dtype != dtype) fatal("invalid dtype for option %s\n", name.c_str());
                *out = opt;
                m_options.erase(m_options.begin() + idx);
                return true;
            }
        }
        return false;
    }
};

std::string this_library_dir() {
    Dl_info info;
    if (dladdr((void *)&this_library_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t k = p.find_last_of('/');
        if (k != std::string::npos) return p.substr(0, k);
    }
    return ".";
}

template <class T>
T *dev_alloc(size_t count) {
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, count * sizeof(T) + 16));
    HIP_CHECK(hipMemset(p, 0, count * sizeof(T) + 16));
    return (T *)p;
}

// The sprite atlas of a game is read-only and the same for every handle of the process that plays the game with the same assets: the
// decoded images are kept once per process (a 16-game x 8-device handle has 128 parts; coinrun's pack alone decodes to 230 MB) and the
// device copy once per device, shared by the parts (and handles) that sit there.  Entries live as long as a part holds them.
struct DeviceAtlas {
    std::shared_ptr<const HostAssets> host;
    int device = 0;
    GameAssetsDev *d_assets = nullptr;
    uint32_t *d_pixels = nullptr;
    ~DeviceAtlas() {
        (void)hipSetDevice(device);
        if (d_assets) (void)hipFree(d_assets);
        if (d_pixels) (void)hipFree(d_pixels);
    }
};
static std::mutex g_atlas_mutex;
static std::map<std::string, std::weak_ptr<const HostAssets>> g_host_atlas;
static std::map<std::string, std::weak_ptr<DeviceAtlas>> g_device_atlas;
// (the caller has selected `device`)
static std::shared_ptr<DeviceAtlas> shared_atlas(int device, const std::string &key, const std::function<void(HostAssets *)> &load) {
    std::lock_guard<std::mutex> lk(g_atlas_mutex);
    const std::string dkey = std::to_string(device) + "|" + key;
    if (auto have = g_device_atlas[dkey].lock()) return have;
    std::shared_ptr<const HostAssets> host = g_host_atlas[key].lock();
    if (!host) {
        auto fresh = std::make_shared<HostAssets>();
        load(fresh.get());
        host = fresh;
        g_host_atlas[key] = host;
    }
    auto da = std::make_shared<DeviceAtlas>();
    da->host = host;
    da->device = device;
    da->d_assets = dev_alloc<GameAssetsDev>(1);
    HIP_CHECK(hipMemcpy(da->d_assets, &host->table, sizeof(GameAssetsDev), hipMemcpyHostToDevice));
    da->d_pixels = dev_alloc<uint32_t>(host->pixels.size());
    HIP_CHECK(hipMemcpy(da->d_pixels, host->pixels.data(), host->pixels.size() * 4, hipMemcpyHostToDevice));
    g_device_atlas[dkey] = da;
    return da;
}

// Launch order of the render kernel by background image (VecGame::rebuild_render_order): every how many steps a game's handles re-sort it
// by default; 0 = never (launch slot = env).  Per game, from same-box measurements (profiles/r05_render_order_*): the order cuts the render
// kernel's HBM fetch traffic by 3-4x for the games that sample a large parallax background, at equal or slightly better speed; games whose
// frames are cheap enough to be bandwidth-sensitive (bigfish) lose by it -- every resident frame then reads the same image.
// Measured (profiles/r05_render_order_ab.txt, device ms per step at 65 536 envs, off -> every 16 steps): miner 1.746 -> 1.430 (+22 %),
// climber 1.130 -> 1.119, ninja 1.272 -> 1.261, coinrun 1.239 -> 1.230, jumper 1.750 -> 1.739 (+0.6-0.9 % each, and coinrun's render FETCH_SIZE
// 1.73 -> 0.46 GB per 8 steps' launches, L2 hit rate 61 -> 82 %); caveflyer +0.3 %; maze -1.1 %, heist -1.4 %, bigfish -37 %: off.
// Which stream a two-chunk handle's first chunk runs on (launch_game `order`): 0 = its own stream, behind an event of the main stream that
// uploads the actions; 4 = the main stream itself, the tier-1 list kernel on the other one.  Per game, from two same-box sweeps of all 16
// (profiles/r06_call31_order16.txt, r06_call32_order16.txt; M steps/s at 65 536 envs): plunder +3.0 / +3.4 %, bigfish +1.7 / +1.9 %, climber
// +0.6 / +2.0 %, miner +1.4 / +1.0 %; coinrun -4 % (its tier-2 list kernel ahead of chunk 1 makes the two chunks a pipeline that order 4
// shifts), ninja -1 %, starpilot -1 %, the others within +-0.8 %.
static int default_launch_order(int game_id) {
    return (game_id == GAME_PLUNDER || game_id == GAME_BIGFISH || game_id == GAME_CLIMBER || game_id == GAME_MINER) ? 4 : 0;
}
static int default_render_order_period(int game_id) {
    switch (game_id) {
        case GAME_COINRUN: case GAME_CLIMBER: case GAME_NINJA: case GAME_JUMPER: case GAME_MINER: return 16;
        default: return 0;
    }
}

// BAG:819-838 prepare_for_drawing(64): the camera scalars that depend on the frame height, for the 64-pixel observation frame (centre
// and visibility do not depend on it)
static void camera_scalars_of_the_observation_frame(EnvHdr *h) {
    const float raw_unit = 64 / h->visibility;
    h->unit = (float)((double)raw_unit * (64.0 / 64.0));
    h->view_dim = (float)(64.0 / (double)raw_unit);
    h->x_off = h->unit * (h->center_x - h->view_dim / 2);
    h->y_off = h->unit * (h->center_y - h->view_dim / 2);
}

struct VecGame {
    int num_envs = 0;
    int game_id = -1;    // assets, state wire format
    int kernel_id = -1;  // the policy instantiation the kernels run (kernel_id_for)
    int device_id = 0;
    bool host_observations = true;
    bool render_human = false;   // reference src/vecgame.cpp:190,270-282: a fourth info tensor "rgb" [512][512][3] (pg_human.h)
    bool api_observed = true;    // libenv_observe has been called since the last libenv_act (render_human: whose camera scalars get_state sees)
    bool human_stale = false;    // a set_state since the frames were last drawn: the next libenv_observe redraws them (reference src/vecgame.cpp:367-375 redraws on every observe)
    std::vector<void *> human_ptr;  // caller's info "rgb" buffers
    bool human_contig = false;
    size_t human_stride = 0;  // > 0: the caller's frames lie at this uniform distance (bytes)
    void launch_human(int env_base, int count);
    std::vector<libenv_tensortype> observation_types, action_types, info_types;
    hipStream_t stream = nullptr;
    hipEvent_t ev_fork = nullptr;
    hipStream_t lane_stream[2] = {nullptr, nullptr}, side_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_lane[2] = {nullptr, nullptr};
    hipEvent_t ev_side[3] = {};
    hipEvent_t ev_step[MAX_CHUNKS] = {};
    // rew / first / info and the list counters are final when the step kernels are done, well before the render kernels are: large handles
    // download them on a stream of their own behind the step kernels, and libenv_observe scatters them into the caller's arrays while the
    // frames are still being drawn -- ~50 us of host work per step at 65536 envs that used to sit between two steps (PROCGEN_AMD_EARLY_SMALL=0: off)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_small = nullptr, ev_out[MAX_CHUNKS] = {};
    bool early_small = false, small_in_flight = false;
    // Host-landed observations (the unmodified gym3 ABI), large handles: chunk c's slice of the caller's array is copied as soon as chunk
    // c's render kernel is done, on a stream of its own, while the other chunks still step and draw -- the 805 MB landing of a 65536-env
    // step takes ten times longer than its kernels, so everything but the first chunk's kernels hides under it (PROCGEN_AMD_OBS_CHUNK_COPY=0: one
    // copy behind the whole step, as before round 5)
    hipStream_t obs_stream = nullptr;
    hipEvent_t ev_frames[MAX_CHUNKS] = {}, ev_obs = nullptr;
    bool obs_chunk_copy = false;
    int order = 0;  // PROCGEN_AMD_ORDER

    int first_pct = 60;  // PROCGEN_AMD_FIRST_PCT: share of the first of two chunks (75 until round 6; profiles/r06_call12_ab.txt: 60 is +2..7 % for five of six games, with the round-5 kernels as well)
    int chunks = 2;  // PROCGEN_AMD_CHUNKS: env range cut in 2 so one chunk's step kernel overlaps the other's render kernel (+6 % measured)
    LaunchStreams streams() const {
        LaunchStreams ls{};
        ls.main = stream;
        ls.fork = ev_fork;
        for (int k = 0; k < 2; k++) {
            ls.lane[k] = lane_stream[k];
            ls.lane_done[k] = ev_lane[k];
        }
        for (int k = 0; k < 3; k++) {
            ls.side[k] = side_stream[k];
            ls.side_done[k] = ev_side[k];
        }
        for (int c = 0; c < MAX_CHUNKS; c++) ls.step_done[c] = ev_step[c];
        for (int c = 0; c < MAX_CHUNKS; c++) ls.outputs_done[c] = early_small ? ev_out[c] : nullptr;
        for (int c = 0; c < MAX_CHUNKS; c++) {
            ls.frames_done[c] = (obs_chunk_copy && host_observations) ? ev_frames[c] : nullptr;
            ls.render_t0[c] = (time_kernels && time_render) ? tk_r0[tk_slot][c] : nullptr;
            ls.render_t1[c] = (time_kernels && time_render) ? tk_r1[tk_slot][c] : nullptr;
        }
        ls.order = order;
        ls.first_pct = first_pct;
        ls.chunks = chunks;
        return ls;
    }
    DevCtx d{};
    std::shared_ptr<DeviceAtlas> atlas;  // sprite atlas on this part's device, shared with every other part that plays the same game there
    uint32_t *d_game_tables = nullptr;
    int32_t *d_action = nullptr;
    // [rew f32 N | prev_level_seed i32 N | level_seed i32 N | first u8 N | prev_level_complete u8 N | list counts A i32 x LIST_COUNTERS | error i32 | list counts B | error info i32 x ERROR_INFO_WORDS]
    // The tail doubles as the tier-list counters (double-buffered A / B) so that one memset clears "next counts + error" and
    // the one download per step tells the host which list kernels have work next step.
    uint8_t *d_small = nullptr;
    size_t tail_off = 0;            // offset of the tail: [list counters A | error | list counters B]
    int host_list_count[MAX_CHUNKS][NUM_TIERS] = {};  // entries of the (chunk, tier) lists the coming step reads
    size_t small_bytes = 0;
    int *d_big_list[2] = {nullptr, nullptr};
    int *d_big_count[2] = {nullptr, nullptr};
    int *d_reset_list = nullptr;   // SPLIT_RESET games: envs whose episode a step kernel ended this step (consumed by the reset kernel)
    int *d_reset_count = nullptr;  // [2][MAX_CHUNKS], double-buffered by step parity (a step kernel zeroes the next step's)
    uint8_t *d_route[2] = {nullptr, nullptr};
    // PROCGEN_AMD_RENDER_ORDER=K (experiment, off by default): every K steps the render kernel's workgroup -> env map of each launch
    // chunk is re-sorted by background image, images dealt to the XCDs (workgroup j runs on XCD j mod 8, each with its own L2)
    uint32_t *d_frame_rec = nullptr;  // display-list games: the frame records (DevCtx::frame_rec, null once the handle has gone back to the one-kernel renderer)
    int slow_streak = 0;              // steps seen to send most frames to the full renderer's list kernel
    bool step_used_display_list = false;  // the step in flight was launched with prep -> raster (read_tail may switch the handle back meanwhile)
    long long slow_passes = 0;        // draw_slow_frames calls so far (procgen_amd_display_list_frames out[2])
    int render_order_period = 0;
    int *d_render_order = nullptr, *d_render_order_scratch = nullptr;
    void rebuild_render_order();
    void draw_slow_frames();
    void bind_routing() {  // double-buffered by step parity: this step reads [cur], fills [nxt]
        const int cur = (int)(step_count & 1), nxt = cur ^ 1;
        d.big_list = d_big_list[cur];
        d.big_count = d_big_count[cur];
        d.next_big_list = d_big_list[nxt];
        d.next_big_count = d_big_count[nxt];
        d.route = d_route[cur];
        d.next_route = d_route[nxt];
        d.reset_list = d_reset_list;
        d.reset_count = d_reset_count + cur * MAX_CHUNKS;
        d.next_reset_count = d_reset_count + nxt * MAX_CHUNKS;
    }
    uint64_t step_count = 0;
    // host staging (pinned)
    int32_t *h_action = nullptr;
    uint8_t *h_small = nullptr;
    uint8_t *h_obs_stage = nullptr;  // only when the caller's ob pointers are not one contiguous array
    // caller buffers (libenv_set_buffers)
    std::vector<void *> ob_ptr, ac_ptr, info_ptr[3];
    float *rew_ptr = nullptr;
    uint8_t *first_ptr = nullptr;
    bool ob_contig = false, ac_contig = false, info_contig = false;
    bool buffers_set = false;
    bool pending = false;
    bool registered_obs = false;

    // forced_name / stride / index: one game of a joint handle owns the envs index + i * stride (index counts from the
    // handle's env 0: a device shard adds its first global index); forced_device: the shard's device (-1: from the options)
    VecGame(int nenvs, VecOptions opts, const std::string &forced_name = "", int stride = 1, int index = 0, int forced_device = -1);
    void use_device() const { HIP_CHECK(hipSetDevice(device_id)); }  // every entry point: the parts of a handle may sit on different devices
    bool external_pinned = false;  // the handle registered the caller's whole observation array once (multi-part handles)
    ~VecGame();
    void set_buffers(struct libenv_buffers *bufs);
    void launch_kernels(int mode);
    void read_tail();
    [[noreturn]] void report_device_error(int err, const int *info, const char *when);
    void check_late_error(const char *when = "a step");
    // procgen_amd_kernel_timing: HIP events around the kernels of every libenv_act of the caller's own loop (bench.py: the device time of
    // a step and the wall time of a step then come from the SAME steps)
    // (two sets of events: a step's durations are read while the NEXT step runs -- in libenv_observe ahead of its wait, or by the query --
    // so that the three hipEventElapsedTime calls are not part of the device's idle time between two steps)
    bool time_kernels = false, time_render = false, tk_pending[2] = {false, false};
    int tk_slot = 0;
    hipEvent_t tk_e0[2] = {}, tk_e1[2] = {}, tk_r0[2][MAX_CHUNKS] = {}, tk_r1[2][MAX_CHUNKS] = {};
    void collect_timing(int slot);
    double tk_sum_ms = 0, tk_render_ms = 0;
    int tk_steps = 0, tk_render_launches = 0;
    // host-mapped copy of the error record, written by the first kernel that raises a check (pg_env.h pg_report_error): [0] code (written
    // last), [1] env + 1, [2] code | line << 8, [3] kind, [4] n_ents, [5] agent.  Read after the stream join of every libenv_observe: no copy
    volatile int *h_error_rec = nullptr;
    void launch(int mode);
    void act();
    void observe(bool from_api = false);
    int get_state(int env_idx, char *data, int length, bool may_not_fit = false);
    bool load_snapshot_block(int env_idx);                              // the 256-env block of env_idx into the snapshot cache (joins the pending step)
    bool serialize_cached(int env_idx, std::string *out, std::string *err) const;  // env_idx must lie in the cached block; touches no shared state (callable from several threads)
    void set_state(int env_idx, const char *data, int length);
    void set_states(int first, int count, const char *data, const long long *offsets);  // consecutive envs of one snapshot block
    void snapshot(int env_idx, EnvSnapshot *s, bool single = false);
    static constexpr int SNAP_BLOCK = 256;
    int snap_first = -1, snap_count = 0;  // envs [snap_first, snap_first + snap_count) of the snapshot cache; -1: stale
    std::vector<EnvHdr> snap_hdr;
    std::vector<uint32_t> snap_ents, snap_rng;
    std::vector<uint8_t> snap_grid;
    void flush_routes();
    std::vector<uint8_t> h_route;  // host copy of the route table the coming step reads (valid between a set_state and the next launch)
    bool route_mirror_valid = false, route_dirty = false;
    int env_offset = 0;
    int env_stride = 1;
    std::vector<int> game_n;  // Game::game_n of every env: its global index (reference src/vecgame.cpp:317) until a set_state adopts another (src/game.cpp:253)
};

VecGame::VecGame(int nenvs, VecOptions opts, const std::string &forced_name, int stride, int index, int forced_device) {
    num_envs = nenvs;
    if (num_envs <= 0) fatal("num_envs must be positive\n");
    std::string env_name, resource_root;
    int num_levels = 0, start_level = -1, num_actions = -1, rand_seed = 0, num_threads = 4;
    bool render_human = false;
    // reference src/vecgame.cpp:183-190
    opts.consume_string("env_name", &env_name);
    opts.consume_int("num_levels", &num_levels);
    opts.consume_int("start_level", &start_level);
    opts.consume_int("num_actions", &num_actions);
    opts.consume_int("rand_seed", &rand_seed);
    opts.consume_int("num_threads", &num_threads);  // accepted, meaningless here: the GPU is the thread pool
    opts.consume_string("resource_root", &resource_root);
    opts.consume_bool("render_human", &render_human);
    // extension options of this library (include/procgen_amd.h)
    env_offset = 0;
    device_id = -1;
    opts.consume_int("device_id", &device_id);
    opts.consume_int("env_offset", &env_offset);
    opts.consume_bool("host_observations", &host_observations);

    int num_devices_opt = 1;
    opts.consume_int("num_devices", &num_devices_opt);  // handled by libenv_make (Handle); consumed here so that it is not "unused"
    if (!forced_name.empty()) env_name = forced_name;
    env_stride = stride;
    env_offset += index;
    if (forced_device >= 0) device_id = forced_device;
    if (env_name.empty()) fatal("fassert failed 'env_name != \"\"'\n");
    if (!(num_actions > 0)) fatal("fassert failed 'num_actions > 0'\n");
    if (!(num_levels >= 0)) fatal("fassert failed 'num_levels >= 0'\n");
    if (!(start_level >= 0)) fatal("fassert failed 'start_level >= 0'\n");
    this->render_human = render_human;
    game_id = game_id_from_name(env_name);
    if (game_id < 0) fatal("unknown game %s\n", env_name.c_str());
    if (!game_supported(game_id)) fatal("game %s is not implemented in the HIP stepper yet\n", env_name.c_str());

    // reference src/game.cpp:42-75 (Game::parse_options)
    GameOptions &o = d.opt;
    memset(&o, 0, sizeof(o));
    o.use_backgrounds = 1;
    bool b;
    bool use_easy_jump = false;
    opts.consume_bool("use_easy_jump", &use_easy_jump);
    if (opts.consume_bool("paint_vel_info", &b)) o.paint_vel_info = b;
    if (opts.consume_bool("use_generated_assets", &b)) o.use_generated_assets = b;
    if (opts.consume_bool("use_monochrome_assets", &b)) o.use_monochrome_assets = b;
    if (opts.consume_bool("restrict_themes", &b)) o.restrict_themes = b;
    if (opts.consume_bool("use_backgrounds", &b)) o.use_backgrounds = b;
    if (opts.consume_bool("center_agent", &b)) o.center_agent = b;
    if (opts.consume_bool("use_sequential_levels", &b)) o.use_sequential_levels = b;
    int dist_mode = EasyMode;
    opts.consume_int("distribution_mode", &dist_mode);
    o.distribution_mode = dist_mode;
    if (dist_mode == EasyMode || dist_mode == HardMode) {
    } else if (dist_mode == ExtremeMode) {
        if (!game_has_extreme_mode(game_id)) fatal("fassert failed: extreme mode unsupported for %s\n", env_name.c_str());
    } else if (dist_mode == MemoryMode) {
        if (!game_has_memory_mode(game_id)) fatal("fassert failed: memory mode unsupported for %s\n", env_name.c_str());
    } else {
        fatal("invalid distribution_mode %d\n", dist_mode);
    }
    kernel_id = kernel_id_for(game_id, dist_mode);
    if (!game_supported(kernel_id)) fatal("game %s has no kernel for distribution_mode %d in the HIP stepper\n", env_name.c_str(), dist_mode);
    int plain_assets = 0, physics_mode = 0, game_type = 0;
    opts.consume_int("plain_assets", &plain_assets);
    opts.consume_int("physics_mode", &physics_mode);
    opts.consume_int("debug_mode", &o.debug_mode);
    opts.consume_int("game_type", &game_type);
    opts.ensure_empty();
    level_seed_range(num_levels, start_level, &o.level_seed_low, &o.level_seed_high);

    // tensortypes: reference src/vecgame.cpp:212-268
    {
        libenv_tensortype s{};
        strcpy(s.name, "rgb");
        s.scalar_type = LIBENV_SCALAR_TYPE_DISCRETE;
        s.dtype = LIBENV_DTYPE_UINT8;
        s.shape[0] = RES_W;
        s.shape[1] = RES_H;
        s.shape[2] = 3;
        s.ndim = 3;
        s.low.uint8 = 0;
        s.high.uint8 = 255;
        observation_types.push_back(s);
    }
    {
        libenv_tensortype s{};
        strcpy(s.name, "action");
        s.scalar_type = LIBENV_SCALAR_TYPE_DISCRETE;
        s.dtype = LIBENV_DTYPE_INT32;
        s.ndim = 0;
        s.low.int32 = 0;
        s.high.int32 = num_actions - 1;
        action_types.push_back(s);
    }
    const char *info_names[3] = {"prev_level_seed", "prev_level_complete", "level_seed"};
    for (int i = 0; i < 3; i++) {
        libenv_tensortype s{};
        strcpy(s.name, info_names[i]);
        s.scalar_type = LIBENV_SCALAR_TYPE_DISCRETE;
        s.ndim = 0;
        if (i == 1) {
            s.dtype = LIBENV_DTYPE_UINT8;
            s.low.uint8 = 0;
            s.high.uint8 = 1;
        } else {
            s.dtype = LIBENV_DTYPE_INT32;
            s.low.int32 = 0;
            s.high.int32 = INT32_MAX;
        }
        info_types.push_back(s);
    }
    if (this->render_human) {  // reference src/vecgame.cpp:270-282
        libenv_tensortype s{};
        strcpy(s.name, "rgb");
        s.scalar_type = LIBENV_SCALAR_TYPE_DISCRETE;
        s.dtype = LIBENV_DTYPE_UINT8;
        s.shape[0] = HUMAN_RES;
        s.shape[1] = HUMAN_RES;
        s.shape[2] = 3;
        s.ndim = 3;
        s.low.uint8 = 0;
        s.high.uint8 = 255;
        info_types.push_back(s);
    }

    // device
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        fatal("no HIP device available: the MI355X stepper cannot run (there is no CPU fallback)\n");
    if (device_id < 0) {
        const char *lr = getenv("LOCAL_RANK");
        device_id = lr ? atoi(lr) % ndev : 0;
    }
    if (getenv("PROCGEN_AMD_FAKE_DEVICES") && forced_device >= 0) device_id %= ndev;  // testing aid: the shards of a multi-device handle on the GPUs there are (an explicit device_id is never wrapped)
    if (device_id >= ndev) fatal("device_id %d out of range (%d devices)\n", device_id, ndev);
    HIP_CHECK(hipSetDevice(device_id));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    // The runtime deals its streams round-robin onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default).  A handle
    // below 4096 envs launches everything on `stream` (launch_game), so it creates no other stream: the 16 parts of a
    // joint handle then sit on different queues instead of all 16 main streams sharing queue 0 (measured: the parts of
    // a 16 x 1024-env handle ran strictly one after the other, 6.9 ms per step).
    if (num_envs >= 4096) {
        HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) {
            HIP_CHECK(hipStreamCreateWithFlags(&lane_stream[k], hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&ev_lane[k], hipEventDisableTiming));
        }
        order = default_launch_order(game_id);
        if (const char *o = getenv("PROCGEN_AMD_ORDER")) order = atoi(o);
        if (order != 0 && order != 4) HIP_CHECK(hipStreamCreateWithFlags(&side_stream[0], hipStreamNonBlocking));  // four streams at most (hardware queues)
        for (int c = 0; c < MAX_CHUNKS; c++) HIP_CHECK(hipEventCreateWithFlags(&ev_step[c], hipEventDisableTiming));
        for (int k = 0; k < 3; k++) HIP_CHECK(hipEventCreateWithFlags(&ev_side[k], hipEventDisableTiming));
        const char *es = getenv("PROCGEN_AMD_EARLY_SMALL");
        // (not for the split-reset games: their outputs are final only behind the reset kernels, and the extra stream cost jumper 7 %)
        // (and only with a hardware queue to spare: main + two chunk streams take three of the runtime's default four, a fifth stream then
        // shares a queue with one of them and its copy can serialise behind a render kernel -- the A/B that found the gain ran with 16)
        const char *hq = getenv("GPU_MAX_HW_QUEUES");
        const bool spare_queue = hq && atoi(hq) >= 5;
        if (!(es && atoi(es) == 0) && (spare_queue || (es && atoi(es) != 0)) && !getenv("PROCGEN_AMD_DEBUG") && !game_split_reset(kernel_id)) {
            HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&ev_small, hipEventDisableTiming));
            for (int c = 0; c < MAX_CHUNKS; c++) HIP_CHECK(hipEventCreateWithFlags(&ev_out[c], hipEventDisableTiming));
            early_small = true;
        }
    }
    // Handles that land their observations on the host (the unmodified ABI) are bound by the 805 MB copy, not by the kernels: four launch
    // chunks -- each chunk's slice copied on a stream of its own while the next chunks step and draw -- measured 15.09 ms per step against
    // 15.42 with two chunks and one copy behind the step (65 536 envs, profiles/r05_host_landed_ab.txt; two chunks with per-chunk copies:
    // 15.63, eight: 15.30).  Device-resident handles keep two chunks (more cost the kernels 15 %) and no such stream.
    {
        const char *oc = getenv("PROCGEN_AMD_OBS_CHUNK_COPY");
        // (not for the split-reset games: their four-chunk reset lists have only ever been measured and tested with two chunks)
        const bool want = host_observations && num_envs >= 32768 && !(oc && atoi(oc) == 0) && !getenv("PROCGEN_AMD_DEBUG") && !game_split_reset(kernel_id);
        if (want) chunks = 4;
        if (const char *c = getenv("PROCGEN_AMD_CHUNKS")) chunks = atoi(c) > 0 ? (atoi(c) < MAX_CHUNKS ? atoi(c) : MAX_CHUNKS) : 1;
        if ((want || (oc && atoi(oc) != 0 && num_envs >= 4096)) && chunks >= 3) {  // (with two chunks one copy behind the whole step is the faster form)
            HIP_CHECK(hipStreamCreateWithFlags(&obs_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&ev_obs, hipEventDisableTiming));
            for (int c = 0; c < MAX_CHUNKS; c++) HIP_CHECK(hipEventCreateWithFlags(&ev_frames[c], hipEventDisableTiming));
            obs_chunk_copy = true;
        }
    }
    d.chunk_envs = chunk_envs_for(num_envs, 1);  // one list chunk (see DevCtx::big_list)

    // assets: baked pack next to the library (procgen_amd/data/<game>.atlas) or the PNG tree at resource_root
    std::string data_dir = getenv("PROCGEN_AMD_DATA_DIR") ? getenv("PROCGEN_AMD_DATA_DIR") : this_library_dir() + "/../../data";
    std::string err;
    {
        const std::string atlas_path = data_dir + "/" + env_name + ".atlas";
        const int kid = kernel_id, gid = game_id;
        const bool gen = o.use_generated_assets != 0;
        const std::string key = env_name + "|" + (gen ? "generated|" + std::to_string(kid) : atlas_path + "|" + resource_root);
        atlas = shared_atlas(device_id, key, [&](HostAssets *out) {
            std::string err;
            if (gen) generate_game_assets(env_name, game_use_block_asset(kid), out);
            else if (!load_game_assets(gid, resource_root, atlas_path, out, &err)) fatal("failed to load images %s\n", err.c_str());
        });
    }

    // per-env state in HBM
    const size_t N = (size_t)num_envs;
    game_limits(kernel_id, &d.ent_cap, &d.grid_bytes);
    d.num_envs = num_envs;
    d.reset_chunk_envs = chunk_envs_for(num_envs, chunks);  // reset lists follow the launch chunks (SPLIT_RESET games)
    if (const char *f = getenv("PROCGEN_AMD_FIRST_PCT")) first_pct = atoi(f);
    d.reset_first = chunks == 2 ? first_chunk_envs(num_envs, first_pct) : 0;
    if (d.reset_first == 0) first_pct = 0;
    d.hdr = dev_alloc<EnvHdr>(N);
    d.rng = dev_alloc<uint32_t>(N * MT_SLOTS * MT_STRIDE);
    d.ents = dev_alloc<uint32_t>(ent_table_words(num_envs, d.ent_cap));
    d.grid = dev_alloc<uint8_t>(N * d.grid_bytes);
    {
        std::vector<EnvHdr> hdr(N);
        std::vector<uint32_t> rng(N * MT_SLOTS * MT_STRIDE);
        game_init_state(kernel_id, num_envs, rand_seed, env_offset, env_stride, hdr.data(), rng.data());
        for (auto &h : hdr) {  // reference src/vecgame.cpp:316-317 (every game starts with the handle's range)
            h.level_seed_low = o.level_seed_low;
            h.level_seed_high = o.level_seed_high;
            h.opt_bits = env_option_bits(o);  // per env as well: set_state adopts the options a state was saved under (reference src/game.cpp:233-246)
            h.opt_debug_mode = o.debug_mode;
        }
        HIP_CHECK(hipMemcpy(d.hdr, hdr.data(), N * sizeof(EnvHdr), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d.rng, rng.data(), rng.size() * 4, hipMemcpyHostToDevice));
    }
    game_n.resize(N);
    for (size_t i = 0; i < N; i++) game_n[i] = env_offset + (int)i * env_stride;
    d_action = dev_alloc<int32_t>(N);
    d.action = d_action;
    d.obs = dev_alloc<uint8_t>(N * OBS_BYTES);
    constexpr size_t TAIL_WORDS = 2 * LIST_COUNTERS + 1 + ERROR_INFO_WORDS + 2 * MAX_CHUNKS;  // list counters A, error word, counters B, error record, slow-list counters
    small_bytes = N * 14 + 4 + TAIL_WORDS * sizeof(int);
    d_small = dev_alloc<uint8_t>(small_bytes);
    d.rew = (float *)d_small;
    d.prev_level_seed = (int32_t *)(d_small + 4 * N);
    d.level_seed = (int32_t *)(d_small + 8 * N);
    d.first = d_small + 12 * N;
    d.prev_level_complete = d_small + 13 * N;
    tail_off = (14 * N + 3) & ~(size_t)3;
    d.error = (int *)(d_small + tail_off) + LIST_COUNTERS;
    static_assert(ERROR_INFO_OFFSET == LIST_COUNTERS + 1, "the error record lies behind the second counter block");
    {
        int *rec = nullptr;
        void *rec_dev = nullptr;
        HIP_CHECK(hipHostMalloc((void **)&rec, 16 * sizeof(int), hipHostMallocMapped));  // words 0..7: the error record; word 8: the slow-frame flag of a display-list handle
        memset(rec, 0, 16 * sizeof(int));
        HIP_CHECK(hipHostGetDevicePointer(&rec_dev, rec, 0));
        h_error_rec = rec;
        d.slow_flag = (int *)rec_dev + 8;
        const unsigned long long a = (unsigned long long)rec_dev;
        const int words[2] = {(int)(unsigned)(a & 0xffffffffull), (int)(unsigned)(a >> 32)};
        HIP_CHECK(hipMemcpy(d.error + ERROR_INFO_OFFSET + 6, words, sizeof(words), hipMemcpyHostToDevice));
    }
    small_bytes = tail_off + TAIL_WORDS * sizeof(int);
    d.slow_count = (int *)(d_small + tail_off) + 2 * LIST_COUNTERS + 1 + ERROR_INFO_WORDS;
    for (int k = 0; k < 2; k++) {
        d_big_list[k] = dev_alloc<int>(N * NUM_TIERS);
        d_big_count[k] = (int *)(d_small + tail_off) + (LIST_COUNTERS + 1) * k;  // A, then the error word, then B
        d_route[k] = dev_alloc<uint8_t>(N);
    }
    d_reset_list = dev_alloc<int>(N);
    d_reset_count = dev_alloc<int>(2 * MAX_CHUNKS);
    render_order_period = default_render_order_period(game_id);
    if (const char *ro = getenv("PROCGEN_AMD_RENDER_ORDER")) render_order_period = atoi(ro);  // (0: off)
    if (render_order_period > 0 && !d.opt.use_generated_assets) {
        d_render_order = dev_alloc<int>(N);  // (bound to d.render_order by the first rebuild)
        d_render_order_scratch = dev_alloc<int>(MAX_CHUNKS * MAX_BACKGROUNDS);
    }
    // display-list games (pg_prep.h): a frame record per env
    if (const int rec_words = game_frame_rec_words(kernel_id); rec_words > 0 && !o.use_generated_assets && !(getenv("PROCGEN_AMD_DISPLAY_LIST") && atoi(getenv("PROCGEN_AMD_DISPLAY_LIST")) == 0)) {
        d.frame_rec = d_frame_rec = dev_alloc<uint32_t>(N * (size_t)rec_words);
        d.slow_list = dev_alloc<int>(N);
    }
    d.assets = atlas->d_assets;
    d.pixels = atlas->d_pixels;
    if (this->render_human) {
        const size_t bytes = N * HUMAN_BYTES;
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        // (a headroom check for large requests only: a small handle on a shared, nearly full GPU still gets its few MB or hipMalloc's own error)
        if (bytes > (256ull << 20) && bytes + (1ull << 30) > free_b) fatal("render_human needs %zu MB of device memory for %d info frames of 512 x 512 x 3; %zu MB free\n", bytes >> 20, num_envs, free_b >> 20);
        d.human = dev_alloc<uint8_t>(bytes);
    }
    if (o.use_generated_assets) {
        // every env owns a 500 x 500 RGB32 background canvas, repainted by each episode's reset (reference BAG:58-63,769-773)
        const size_t bytes = N * (size_t)GEN_BG_WORDS * 4;
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        if (bytes + (1ull << 30) > free_b) fatal("use_generated_assets needs %zu MB of device memory for %d background canvases (1 MB each); %zu MB free\n", bytes >> 20, num_envs, free_b >> 20);
        d.gen_bg = dev_alloc<uint32_t>(N * (size_t)GEN_BG_WORDS);
        d.bg_req = dev_alloc<int>(N * 2);
        HIP_CHECK(hipMemset(d.bg_req, 0xff, N * 2 * sizeof(int)));
    }
    {
        std::vector<uint32_t> tab(MAX_GAME_TABLE_WORDS);
        const int nw = game_host_tables(kernel_id, d.opt, tab.data(), MAX_GAME_TABLE_WORDS);
        if (nw > 0) {
            d_game_tables = dev_alloc<uint32_t>((size_t)nw);
            HIP_CHECK(hipMemcpy(d_game_tables, tab.data(), (size_t)nw * 4, hipMemcpyHostToDevice));
        }
        d.game_tables = d_game_tables;
    }
    d.debug_flags = getenv("PROCGEN_AMD_DEBUG") ? atoi(getenv("PROCGEN_AMD_DEBUG")) : 0;
    if (d.debug_flags & 2048) d.phase_cycles = dev_alloc<unsigned long long>(32 * 4096);
    if (d.debug_flags & 8192) d.wave_trace = dev_alloc<unsigned long long>(N * 32);
    HIP_CHECK(hipHostMalloc((void **)&h_action, N * 4 + 16, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h_small, small_bytes + 16, hipHostMallocDefault));
    // dev_alloc clears its arrays with hipMemset, i.e. on the NULL stream, and this handle's kernels run on non-blocking streams, which the
    // null stream does not order itself against.  Whether hipMemset returns only when the fill is done is the runtime's business (CUDA's
    // contract says it need not); recycled device memory holding a previous handle's data under a fill that lands late would explain the rare
    // failures seen in long-lived test processes under GPU sharing.  One join here makes the question moot.
    HIP_CHECK(hipDeviceSynchronize());
}

VecGame::~VecGame() {
    (void)hipSetDevice(device_id);
    if (stream) (void)hipStreamSynchronize(stream);
    if (d.wave_trace) {  // PROCGEN_AMD_DEBUG & 8192: residency trace of the last step's workgroups -> $PROCGEN_AMD_TRACE_FILE (tools/gpu/wave_trace.py reads it)
        std::vector<unsigned long long> raw((size_t)num_envs * 32);
        const char *path = getenv("PROCGEN_AMD_TRACE_FILE");
        if (hipMemcpy(raw.data(), d.wave_trace, raw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(path ? path : "wave_trace.bin", "wb")) {
                fwrite(raw.data(), 8, raw.size(), f);
                fclose(f);
            }
        }
        (void)hipFree(d.wave_trace);
    }
    if (d.phase_cycles) {  // PROCGEN_AMD_DEBUG & 2048: per-phase wave cycles of the step kernels, per env-step
        std::vector<unsigned long long> raw(32 * 4096);
        unsigned long long pc[32] = {0};
        const bool got = hipMemcpy(raw.data(), d.phase_cycles, raw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess;
        for (size_t i = 0; i < 32 * 4096; i++) pc[i & 31] += raw[i];
        if (got && pc[14] > 0) {
            static const char *names[13] = {"load_env", "action+velocity", "step_entities (rest)", "collision_pass", "erase_if_needed", "game_step tail", "reset (per reset)", "outputs+camera", "store_env",
                                            " bso: setup", " bso: sub_steps", " se: find+plain ents", " se: smart ent_step"};
            fprintf(stderr, "[procgen_amd phase cycles per env-step, %llu env-steps, %llu resets]\n", pc[14], pc[15]);
            for (int k = 0; k < 13; k++) {
                const double denom = k == 6 ? (double)(pc[15] ? pc[15] : 1) : (double)pc[14];
                fprintf(stderr, "  %-22s %10.1f\n", names[k], (double)pc[k] / denom);
            }
            if ((d.debug_flags & 16) && pc[15] > 0) {  // Env::mark: level-generator stages (game_*.h game_reset), wave cycles per reset
                fprintf(stderr, "[reset marks, wave cycles per reset]\n");
                for (int k = 0; k < 15; k++)
                    if (pc[16 + k]) fprintf(stderr, "  mark %-2d %12.1f\n", k, (double)pc[16 + k] / (double)pc[15]);
            }
            if (pc[31] > 0) {
                static const char *rn[11] = {"set-up: pull tables", "clear + background", "entities z=-1", "grid cells", "entities z=0,1 + hud", "store band",
                                             "set-up: header", "set-up: background", "set-up: entities", "set-up: window + axes", "set-up: type table"};
                fprintf(stderr, "[render kernel, wave cycles per frame, %llu frames]\n", pc[31]);
                for (int k = 0; k < 11; k++) fprintf(stderr, "  %-22s %10.1f\n", rn[k], (double)pc[16 + k] / (double)pc[31]);
            }
        }
        (void)hipFree(d.phase_cycles);
    }
    if (registered_obs && !ob_ptr.empty()) (void)hipHostUnregister(ob_ptr[0]);
    atlas.reset();
    if (d.gen_bg) (void)hipFree(d.gen_bg);
    if (d.bg_req) (void)hipFree(d.bg_req);
    if (d_game_tables) (void)hipFree(d_game_tables);
    (void)hipFree(d.hdr);
    (void)hipFree(d.rng);
    (void)hipFree(d.ents);
    (void)hipFree(d.grid);
    (void)hipFree(d_action);
    (void)hipFree(d.obs);
    if (d.human) (void)hipFree(d.human);
    (void)hipFree(d_small);
    for (int k = 0; k < 2; k++) {
        (void)hipFree(d_big_list[k]);
        (void)hipFree(d_route[k]);
    }
    (void)hipFree(d_reset_list);
    (void)hipFree(d_reset_count);
    if (d_frame_rec) (void)hipFree(d_frame_rec);
    if (d.slow_list) (void)hipFree(d.slow_list);
    if (d_render_order) (void)hipFree(d_render_order);
    if (d_render_order_scratch) (void)hipFree(d_render_order_scratch);
    if (h_action) (void)hipHostFree(h_action);
    if (h_small) (void)hipHostFree(h_small);
    if (h_error_rec) (void)hipHostFree((void *)h_error_rec);
    if (h_obs_stage) (void)hipHostFree(h_obs_stage);
    for (int k = 0; k < 2; k++) {
        if (ev_lane[k]) (void)hipEventDestroy(ev_lane[k]);
        if (lane_stream[k]) (void)hipStreamDestroy(lane_stream[k]);
    }
    for (int c = 0; c < MAX_CHUNKS; c++)
        if (ev_step[c]) (void)hipEventDestroy(ev_step[c]);
    for (int k = 0; k < 3; k++) {
        if (ev_side[k]) (void)hipEventDestroy(ev_side[k]);
        if (side_stream[k]) (void)hipStreamDestroy(side_stream[k]);
    }
    if (ev_obs) (void)hipEventDestroy(ev_obs);
    for (int c = 0; c < MAX_CHUNKS; c++)
        if (ev_frames[c]) (void)hipEventDestroy(ev_frames[c]);
    if (obs_stream) (void)hipStreamDestroy(obs_stream);
    if (ev_small) (void)hipEventDestroy(ev_small);
    for (int c = 0; c < MAX_CHUNKS; c++)
        if (ev_out[c]) (void)hipEventDestroy(ev_out[c]);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (int k = 0; k < 2; k++) {
        if (tk_e0[k]) (void)hipEventDestroy(tk_e0[k]);
        if (tk_e1[k]) (void)hipEventDestroy(tk_e1[k]);
        for (int c = 0; c < MAX_CHUNKS; c++) {
            if (tk_r0[k][c]) (void)hipEventDestroy(tk_r0[k][c]);
            if (tk_r1[k][c]) (void)hipEventDestroy(tk_r1[k][c]);
        }
    }
    if (stream) (void)hipStreamDestroy(stream);
}

void VecGame::set_buffers(struct libenv_buffers *bufs) {  // reference src/vecgame.cpp:30-40,74-83,333-361
    use_device();
    const int N = num_envs;
    ob_ptr.assign(bufs->ob, bufs->ob + N);  // one observation space
    ac_ptr.assign(bufs->ac, bufs->ac + N);  // one action space
    for (int s = 0; s < 3; s++) info_ptr[s].assign(bufs->info + (size_t)s * N, bufs->info + (size_t)(s + 1) * N);
    if (render_human) {
        human_ptr.assign(bufs->info + (size_t)3 * N, bufs->info + (size_t)4 * N);
        human_contig = true;
        for (int e = 0; e < N; e++)
            if ((uint8_t *)human_ptr[e] != (uint8_t *)human_ptr[0] + (size_t)e * HUMAN_BYTES) human_contig = false;
        if (!human_contig && N > 1 && (uint8_t *)human_ptr[1] > (uint8_t *)human_ptr[0]) {
            human_stride = (size_t)((uint8_t *)human_ptr[1] - (uint8_t *)human_ptr[0]);
            for (int e = 0; e < N; e++)
                if ((uint8_t *)human_ptr[e] != (uint8_t *)human_ptr[0] + (size_t)e * human_stride) human_stride = 0;
            if (human_stride < HUMAN_BYTES) human_stride = 0;
        }
    }
    rew_ptr = bufs->rew;
    first_ptr = bufs->first;
    ob_contig = true;
    ac_contig = true;
    info_contig = true;
    for (int e = 0; e < N; e++) {
        if ((uint8_t *)ob_ptr[e] != (uint8_t *)ob_ptr[0] + (size_t)e * OBS_BYTES) ob_contig = false;
        if ((uint8_t *)ac_ptr[e] != (uint8_t *)ac_ptr[0] + (size_t)e * 4) ac_contig = false;
        if ((uint8_t *)info_ptr[0][e] != (uint8_t *)info_ptr[0][0] + (size_t)e * 4 || (uint8_t *)info_ptr[1][e] != (uint8_t *)info_ptr[1][0] + (size_t)e ||
            (uint8_t *)info_ptr[2][e] != (uint8_t *)info_ptr[2][0] + (size_t)e * 4)
            info_contig = false;
    }
    if (host_observations) {
        if (ob_contig) {
            // pin the caller's observation array so the D2H landing is a single DMA ("one pinned host buffer")
            if (!external_pinned) {
                registered_obs = hipHostRegister(ob_ptr[0], (size_t)N * OBS_BYTES, hipHostRegisterDefault) == hipSuccess;
                if (!registered_obs) (void)hipGetLastError();
            }
        } else {
            HIP_CHECK(hipHostMalloc((void **)&h_obs_stage, (size_t)N * OBS_BYTES, hipHostMallocDefault));
        }
    }
    buffers_set = true;
    launch(0);  // initial reset + first frame (reference src/vecgame.cpp:346-357)
}

// called between steps (the handle's streams are idle): enqueued on the main stream ahead of the step -- the chunk streams fork from it
// (launch_game), so every render kernel of the step sees the new order; the sort reads the backgrounds as the last step left them (an
// episode that begins in this step is drawn from a stale slot once, which costs locality, not correctness)
// Display-list handles (pg_prep.h): the frames of the step just joined that its prep kernels queued for the full renderer -- their raster
// workgroups left them alone.  Off the hot path by construction: it runs only when a prep wave raised the host-mapped flag.
void VecGame::draw_slow_frames() {
    h_error_rec[8] = 0;
    slow_passes++;
    const int N = num_envs;
    const int nchunk = N >= 4096 ? (chunks > 1 ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1) : 1;
    const int per = chunk_envs_for(N, nchunk);
    const int first = (nchunk == 2 && first_pct > 0) ? first_chunk_envs(N, first_pct) : 0;
    int bases[MAX_CHUNKS] = {}, used = 0;
    for (int c = 0; c < nchunk; c++) {
        const int base = first > 0 ? (c == 0 ? 0 : first) : c * per;
        const int count = first > 0 ? (c == 0 ? first : N - first) : ((N - base) < per ? (N - base) : per);
        if (count <= 0) break;
        HIP_CHECK(launch_render_slow(kernel_id, d, base, count, c, stream));
        bases[used++] = base;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    check_late_error("the full renderer's pass over a display-list handle's slow frames");
    if (host_observations) {  // their frames were landed before they were drawn: once more, env by env
        int counts[MAX_CHUNKS] = {};
        HIP_CHECK(hipMemcpy(counts, d.slow_count + d.step_parity * MAX_CHUNKS, sizeof(counts), hipMemcpyDeviceToHost));
        std::vector<int> list;
        for (int c = 0; c < used; c++) {
            if (counts[c] <= 0) continue;
            list.resize((size_t)counts[c]);
            HIP_CHECK(hipMemcpy(list.data(), d.slow_list + bases[c], sizeof(int) * (size_t)counts[c], hipMemcpyDeviceToHost));
            for (int e : list) HIP_CHECK(hipMemcpy(ob_contig ? ob_ptr[e] : (void *)(h_obs_stage + (size_t)e * OBS_BYTES), d.obs + (size_t)e * OBS_BYTES, OBS_BYTES, hipMemcpyDeviceToHost));
        }
    }
}

void VecGame::rebuild_render_order() {
    const int N = num_envs;
    const int nchunk = N >= 4096 ? (chunks > 1 ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1) : 1;
    const int per = chunk_envs_for(N, nchunk);
    const int first = (nchunk == 2 && first_pct > 0) ? first_chunk_envs(N, first_pct) : 0;
    for (int c = 0; c < nchunk; c++) {  // each launch chunk's env range is permuted on its own: a chunk's render kernel is ordered behind that chunk's step kernel only
        const int base = first > 0 ? (c == 0 ? 0 : first) : c * per;
        const int count = first > 0 ? (c == 0 ? first : N - first) : ((N - base) < per ? (N - base) : per);
        if (count <= 0) break;
        HIP_CHECK(launch_render_order(d, base, count, d_render_order_scratch + c * MAX_BACKGROUNDS, d_render_order, stream));
    }
    d.render_order = d_render_order;
}

// the device work of one step: the step / reset / render kernels (what procgen_amd_time_steps brackets)
void VecGame::launch_kernels(int mode) {
    snap_first = -1;
    if (route_dirty) flush_routes();
    route_mirror_valid = false;
    bind_routing();
    // The next lists' counters are zero already: they were the lists the step before read, and its render kernel cleared them
    // (DevCtx::clear_lists).  A step never clears the error word: the first error ends the run (set_state clears it before it
    // draws the restored env, to tell that draw's errors apart).  Only with the render kernel switched off for profiling does the
    // host clear the counters.
    if (d.debug_flags & 16) HIP_CHECK(hipMemsetAsync(d.next_big_count, 0, LIST_COUNTERS * sizeof(int), stream));
    LaunchStreams ls = streams();
    for (int c = 0; c < MAX_CHUNKS; c++)
        for (int t = 0; t < NUM_TIERS; t++) ls.list_count[c][t] = mode == 0 ? 0 : host_list_count[c][t];
    d.step_parity = (int)(step_count & 1);  // (display-list games: which of the two slow-list counter sets this step fills)
    step_used_display_list = d.frame_rec != nullptr;
    HIP_CHECK(launch_step(kernel_id, d, mode, ls));
    step_count++;
}

// after the step's small download has landed: device error word, and which list kernels the next step needs
void VecGame::read_tail() {
    const int *tail = (const int *)(h_small + tail_off);
    const int err = tail[LIST_COUNTERS];
    const int *cnt = tail + (LIST_COUNTERS + 1) * (int)(step_count & 1);  // the lists the step just run filled are the ones the next step reads
    for (int c = 0; c < MAX_CHUNKS; c++)
        for (int t = 0; t < NUM_TIERS; t++) host_list_count[c][t] = cnt[c * NUM_TIERS + t];
    if (err && !d.debug_flags) report_device_error(err, tail + 2 * LIST_COUNTERS + 1, "a step");
    // display-list handles: the frames the step just run sent to the full renderer's list kernel.  When that is most of them, step after
    // step -- options the rasterizer's short path does not draw: center_agent = false over a wide world, monochrome assets, paint_vel_info --
    // the handle goes back to the one-kernel renderer, which draws such frames without the detour
    if (d.frame_rec) {
        const int *sc = tail + 2 * LIST_COUNTERS + 1 + ERROR_INFO_WORDS + MAX_CHUNKS * (int)((step_count + 1) & 1);
        int slow = 0;
        for (int c = 0; c < MAX_CHUNKS; c++) slow += sc[c];
        if (slow * 2 > num_envs) slow_streak++;  // (a reading may miss a step's count -- the small outputs can be downloaded before the prep kernels have run -- so: four sightings, not four in a row)
        if (slow_streak >= 4) d.frame_rec = nullptr;
    }
}

// The first device-side check that failed ends the run, like the reference's fassert (src/cpp-utils.h:9-11).  The message names the
// env that raised it, the source line of the check and the kernel, and the fatal log (PROCGEN_AMD_FATAL_LOG) also gets the env's whole
// header and its routing entries: what a post-mortem of a failure that does not reproduce needs.
void VecGame::report_device_error(int err, const int *info, const char *when) {
    std::string extra;
    const int env = info[0] - 1;
    if (env >= 0 && env < num_envs) {
        char buf[512];
        const int kind = info[2];
        const char *kname = kind >= ERR_KIND_BGPAINT ? "paint_backgrounds" : kind >= ERR_KIND_HUMAN ? "render_human" : kind >= ERR_KIND_RENDER ? "render" : "step / reset kernel with an entity arena of";
        snprintf(buf, sizeof buf, "  first reporter: env %d (global index %d), check at source line %d (code %d) in %s %d; n_ents %d agent %d; handle: %d envs, game %d, launch #%llu\n",
                 env, env_offset + env * env_stride, info[1] >> 8, info[1] & 0xff, kname, kind % ERR_KIND_RENDER, info[3], info[4], num_envs, game_id, (unsigned long long)step_count);
        extra = buf;
        (void)hipDeviceSynchronize();
        EnvHdr h;
        uint8_t r0 = 0, r1 = 0;
        if (hipMemcpy(&h, d.hdr + env, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&r0, d_route[0] + env, 1, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(&r1, d_route[1] + env, 1, hipMemcpyDeviceToHost) == hipSuccess) {
            extra += "  header now:";
#define PG_X(type, name) snprintf(buf, sizeof buf, std::is_same<type, float>::value ? " " #name "=%g" : " " #name "=%.0f", (double)h.name); extra += buf;
            PG_HDR_FIELDS(PG_X)
#undef PG_X
            snprintf(buf, sizeof buf, "\n  route[0]=%d route[1]=%d (this launch read route[%d]); list counts the host launched with:", r0, r1, (int)((step_count + 1) & 1));
            extra += buf;
            for (int t = 0; t < NUM_TIERS; t++) {
                snprintf(buf, sizeof buf, " t%d=%d", t, host_list_count[0][t]);
                extra += buf;
            }
            extra += "\n";
        }
    }
    fatal("device-side check failed during %s (code %d: 1 entity table overflow, 2 grid index out of range, 3 fassert, 4 asset theme, 5 unsupported draw)\n%s", when, err, extra.c_str());
}

// after the join of a step's streams: did a kernel of this step -- a render kernel included -- raise a check?  (The download of the small
// outputs carries the device word too, but large handles take it before the render kernels run.)
void VecGame::check_late_error(const char *when) {
    if (!h_error_rec || h_error_rec[0] == 0 || d.debug_flags) return;
    const int info[ERROR_INFO_WORDS] = {h_error_rec[1], h_error_rec[2], h_error_rec[3], h_error_rec[4], h_error_rec[5], 0, 0, 0};
    report_device_error(h_error_rec[0], info, when);
}

void VecGame::launch(int mode) {
    if (time_kernels) {
        tk_slot ^= 1;
        if (tk_pending[tk_slot]) collect_timing(tk_slot);  // (two steps back: long complete)
        HIP_CHECK(hipEventRecord(tk_e0[tk_slot], stream));
    }
    launch_kernels(mode);
    if (time_kernels) {
        HIP_CHECK(hipEventRecord(tk_e1[tk_slot], stream));  // (main has joined the chunk streams: behind every kernel of the step)
        tk_pending[tk_slot] = true;
    }
    if (early_small) {
        // behind every step kernel of the step (chunk grids on the lane streams, list kernels), not behind the render kernels.  The error
        // word travels with it: an error a render kernel of this step raises reaches the host one step later (the word is sticky).
        const int nchunk = chunks > 1 ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1;
        // (each chunk's stream has waited for the list kernels by the time it records its event)
        for (int c = 0; c < nchunk; c++) HIP_CHECK(hipStreamWaitEvent(copy_stream, ev_out[c], 0));
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, small_bytes, hipMemcpyDeviceToHost, copy_stream));
        HIP_CHECK(hipEventRecord(ev_small, copy_stream));
        small_in_flight = true;
    } else {
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, small_bytes, hipMemcpyDeviceToHost, stream));
    }
    if (host_observations) {
        uint8_t *dst = ob_contig ? (uint8_t *)ob_ptr[0] : h_obs_stage;
        if (obs_chunk_copy) {  // chunk by chunk on the copy's own stream, each slice behind its chunk's render kernel (the ranges of launch_game)
            const int nchunk = chunks > 1 ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1;
            const int per = chunk_envs_for(num_envs, nchunk);
            const int first = (nchunk == 2 && first_pct > 0) ? first_chunk_envs(num_envs, first_pct) : 0;
            for (int c = 0; c < nchunk; c++) {
                const int base = first > 0 ? (c == 0 ? 0 : first) : c * per;
                const int count = first > 0 ? (c == 0 ? first : num_envs - first) : ((num_envs - base) < per ? (num_envs - base) : per);
                if (count <= 0) break;
                HIP_CHECK(hipStreamWaitEvent(obs_stream, ev_frames[c], 0));
                HIP_CHECK(hipMemcpyAsync(dst + (size_t)base * OBS_BYTES, d.obs + (size_t)base * OBS_BYTES, (size_t)count * OBS_BYTES, hipMemcpyDeviceToHost, obs_stream));
            }
            HIP_CHECK(hipEventRecord(ev_obs, obs_stream));
            HIP_CHECK(hipStreamWaitEvent(stream, ev_obs, 0));  // (libenv_observe joins `stream`)
        } else {  // one copy behind the whole step
            HIP_CHECK(hipMemcpyAsync(dst, d.obs, (size_t)num_envs * OBS_BYTES, hipMemcpyDeviceToHost, stream));
        }
    }
    if (render_human) launch_human(0, num_envs);
    pending = true;
}

// the 512 x 512 info frames of envs [env_base, env_base + count): kernel + landing in the caller's buffers (reference src/vecgame.cpp:367-375)
void VecGame::launch_human(int env_base, int count) {
    snap_first = -1;  // (the kernel leaves the 512-pixel frame's camera scalars in the env headers)
    HIP_CHECK(launch_render_human(kernel_id, d, env_base, count, stream));
    if (human_contig) {
        HIP_CHECK(hipMemcpyAsync((uint8_t *)human_ptr[0] + (size_t)env_base * HUMAN_BYTES, d.human + (size_t)env_base * HUMAN_BYTES, (size_t)count * HUMAN_BYTES, hipMemcpyDeviceToHost, stream));
    } else if (human_stride > 0) {  // per-env buffers at a uniform distance (padded arrays): one strided copy
        HIP_CHECK(hipMemcpy2DAsync(human_ptr[env_base], human_stride, d.human + (size_t)env_base * HUMAN_BYTES, HUMAN_BYTES, HUMAN_BYTES, (size_t)count, hipMemcpyDeviceToHost, stream));
    } else {
        for (int e = env_base; e < env_base + count; e++) HIP_CHECK(hipMemcpyAsync(human_ptr[e], d.human + (size_t)e * HUMAN_BYTES, HUMAN_BYTES, hipMemcpyDeviceToHost, stream));
    }
    human_stale = false;
}

void VecGame::act() {  // reference src/vecgame.cpp:378-401
    if (!buffers_set) fatal("libenv_act called before libenv_set_buffers\n");
    use_device();
    observe();  // wait_for_stepping_threads()
    api_observed = false;
    if (d_render_order && step_count % (uint64_t)render_order_period == 0) rebuild_render_order();
    const int N = num_envs;
    // the action values are only valid for the duration of this call (reference src/vecgame.cpp:387-388)
    if (ac_contig) memcpy(h_action, ac_ptr[0], (size_t)N * 4);
    else
        for (int e = 0; e < N; e++) h_action[e] = *(int32_t *)ac_ptr[e];
    HIP_CHECK(hipMemcpyAsync(d_action, h_action, (size_t)N * 4, hipMemcpyHostToDevice, stream));
    launch(1);
}

void VecGame::observe(bool from_api) {  // reference src/vecgame.cpp:363-376,416-435
    if (from_api) api_observed = true;
    if (from_api && render_human && human_stale && !pending) {  // states restored since the frames were drawn: VecGame::observe redraws every env's
        use_device();
        launch_human(0, num_envs);
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (!pending) return;
    use_device();
    if (time_kernels) collect_timing(tk_slot ^ 1);  // (the step before the one in flight, while the device is busy)
    if (small_in_flight) {  // the small outputs are on the host while the render kernels still run: scatter them first, join the frames after
        HIP_CHECK(hipEventSynchronize(ev_small));
        small_in_flight = false;
    } else {
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    pending = false;
    const size_t N = (size_t)num_envs;
    read_tail();
    memcpy(rew_ptr, h_small, 4 * N);
    memcpy(first_ptr, h_small + 12 * N, N);
    const int32_t *pls = (const int32_t *)(h_small + 4 * N), *ls = (const int32_t *)(h_small + 8 * N);
    const uint8_t *plc = h_small + 13 * N;
    if (info_contig) {  // gym3's arrays are dense: three block copies instead of 3 N scattered stores (0.2 ms per step at 65536 envs)
        memcpy(info_ptr[0][0], pls, 4 * N);
        memcpy(info_ptr[1][0], plc, N);
        memcpy(info_ptr[2][0], ls, 4 * N);
    } else {
        for (size_t e = 0; e < N; e++) {
            *(int32_t *)info_ptr[0][e] = pls[e];
            *(uint8_t *)info_ptr[1][e] = plc[e];
            *(int32_t *)info_ptr[2][e] = ls[e];
        }
    }
    // (polling the stream instead of blocking on it measured 0 .. -1 %: the wake-up is not what the device waits for between two steps,
    // profiles/r06_call30_order_spin.txt)
    if (early_small) HIP_CHECK(hipStreamSynchronize(stream));  // (the render kernels, the landing of the frames)
    check_late_error();
    if (step_used_display_list && h_error_rec[8] != 0) draw_slow_frames();  // (a prep wave of this step queued a frame for the full renderer)
    if (host_observations && !ob_contig)
        for (size_t e = 0; e < N; e++) memcpy(ob_ptr[e], h_obs_stage + e * OBS_BYTES, OBS_BYTES);
}

// the durations of the step whose events are in `slot` (complete: the stream has been joined since, or a later step has been)
void VecGame::collect_timing(int slot) {
    if (!tk_pending[slot]) return;
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, tk_e0[slot], tk_e1[slot]));
    tk_sum_ms += ms;
    tk_steps++;
    const int nchunk = num_envs < 4096 ? 1 : (chunks > 1 ? (chunks < MAX_CHUNKS ? chunks : MAX_CHUNKS) : 1);  // (the render launches of launch_game)
    for (int c = 0; c < nchunk && time_render; c++) {
        if (hipEventElapsedTime(&ms, tk_r0[slot][c], tk_r1[slot][c]) == hipSuccess) {
            tk_render_ms += ms;
            tk_render_launches++;
        } else {
            (void)hipGetLastError();  // (a chunk without envs records nothing)
        }
    }
    tk_pending[slot] = false;
}

// Host copy of one env's device state.  env.get_state() walks all envs: the state of a block of SNAP_BLOCK consecutive envs is
// fetched with four copies and kept until something changes device state (a step, a restore, a redraw), instead of four small
// synchronous copies per env (262 144 of them at 65 536 envs).
// single: the caller is about to change device state (set_state), so a block fetched now would be thrown away: only env e moves
void VecGame::snapshot(int e, EnvSnapshot *s, bool single) {
    const size_t ents_w = (size_t)EF_COUNT * d.ent_cap, rng_w = 2 * MT_STRIDE, grid_b = (size_t)d.grid_bytes;
    if (single && !(snap_first >= 0 && e >= snap_first && e < snap_first + snap_count)) {
        s->ent_cap = d.ent_cap;
        s->ents.resize(ents_w);
        s->rng.resize(rng_w);
        s->grid.resize(grid_b);
        HIP_CHECK(hipMemcpy(&s->hdr, d.hdr + e, sizeof(EnvHdr), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(s->ents.data(), d.ents + ent_table_base(e, d.ent_cap), ents_w * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(s->rng.data(), d.rng + (size_t)e * MT_SLOTS * MT_STRIDE, rng_w * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(s->grid.data(), d.grid + (size_t)e * grid_b, grid_b, hipMemcpyDeviceToHost));
        return;
    }
    if (!(snap_first >= 0 && e >= snap_first && e < snap_first + snap_count)) {
        const int first = e / SNAP_BLOCK * SNAP_BLOCK, count = num_envs - first < SNAP_BLOCK ? num_envs - first : SNAP_BLOCK;  // aligned blocks: any visiting order fetches a block once
        snap_hdr.resize(count);
        snap_ents.resize(ents_w * count);
        snap_rng.resize(rng_w * count);
        snap_grid.resize(grid_b * count);
        HIP_CHECK(hipMemcpy(snap_hdr.data(), d.hdr + first, sizeof(EnvHdr) * count, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(snap_ents.data(), d.ents + ent_table_base(first, d.ent_cap), ents_w * 4 * count, hipMemcpyDeviceToHost));  // per-env tables are contiguous and dense
        // rand_gen + level_seed_rand_gen of each env: the first two of its MT_SLOTS generator states
        HIP_CHECK(hipMemcpy2D(snap_rng.data(), rng_w * 4, d.rng + (size_t)first * MT_SLOTS * MT_STRIDE, (size_t)MT_SLOTS * MT_STRIDE * 4, rng_w * 4, count, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(snap_grid.data(), d.grid + (size_t)first * grid_b, grid_b * count, hipMemcpyDeviceToHost));
        snap_first = first;
        snap_count = count;
    }
    const size_t k = (size_t)(e - snap_first);
    s->ent_cap = d.ent_cap;
    s->hdr = snap_hdr[k];
    s->ents.assign(snap_ents.begin() + k * ents_w, snap_ents.begin() + (k + 1) * ents_w);
    s->rng.assign(snap_rng.begin() + k * rng_w, snap_rng.begin() + (k + 1) * rng_w);
    s->grid.assign(snap_grid.begin() + k * grid_b, snap_grid.begin() + (k + 1) * grid_b);
}

bool VecGame::load_snapshot_block(int e) {
    if (d.opt.use_generated_assets) fatal("fassert failed '!options.use_generated_assets' (BasicAbstractGame::serialize)\n");  // BAG:1176
    if (!buffers_set) fatal("get_state called before libenv_set_buffers\n");
    if (e < 0 || e >= num_envs) fatal("get_state: env index %d out of range\n", e);
    use_device();
    observe();
    EnvSnapshot s;
    snapshot(e, &s);  // (fills the cache; the copy into s is what a lone get_state would use)
    return true;
}
bool VecGame::serialize_cached(int e, std::string *out, std::string *err) const {
    if (!(snap_first >= 0 && e >= snap_first && e < snap_first + snap_count)) {
        if (err) *err = "serialize_cached: env outside the cached block";
        return false;
    }
    const size_t ents_w = (size_t)EF_COUNT * d.ent_cap, rng_w = 2 * MT_STRIDE, grid_b = (size_t)d.grid_bytes;
    const size_t k = (size_t)(e - snap_first);
    EnvSnapshot s;
    s.ent_cap = d.ent_cap;
    s.hdr = snap_hdr[k];
    s.ents.assign(snap_ents.begin() + k * ents_w, snap_ents.begin() + (k + 1) * ents_w);
    s.rng.assign(snap_rng.begin() + k * rng_w, snap_rng.begin() + (k + 1) * rng_w);
    s.grid.assign(snap_grid.begin() + k * grid_b, snap_grid.begin() + (k + 1) * grid_b);
    if (render_human && !api_observed) camera_scalars_of_the_observation_frame(&s.hdr);
    // (a scratch buffer per thread of the reference's MAX_STATE_SIZE, procgen/env.py:20, allocated once: resizing the output to 1 MiB for every
    // state zero-filled 256 MiB per block of 256 states of ~40 KB, round-5 advisor finding)
    static thread_local std::vector<char> scratch(1 << 20);
    int written = 0;
    if (!serialize_state(game_id, d.opt, game_n[e], s, scratch.data(), (int)scratch.size(), &written, err)) return false;
    out->assign(scratch.data(), (size_t)written);
    return true;
}

// may_not_fit: a buffer too small for the state returns -1 instead of ending the process (procgen_amd_get_states packs states back to back)
int VecGame::get_state(int e, char *data, int length, bool may_not_fit) {  // reference src/vecgame.cpp:438-445
    if (d.opt.use_generated_assets) fatal("fassert failed '!options.use_generated_assets' (BasicAbstractGame::serialize)\n");  // BAG:1176
    if (!buffers_set) fatal("get_state called before libenv_set_buffers\n");
    if (e < 0 || e >= num_envs) fatal("get_state: env index %d out of range\n", e);
    use_device();
    observe();  // wait_for_stepping_threads()
    EnvSnapshot s;
    snapshot(e, &s);
    // render_human, between libenv_act and libenv_observe: the reference's stepping thread has drawn the 64-pixel frame only -- the 512-pixel
    // ones are drawn by VecGame::observe (src/vecgame.cpp:363-376) -- so its get_state carries the 64-pixel frame's camera scalars.  Here the
    // info frames were drawn behind the step already and left theirs in the header: serialize what the reference would
    if (render_human && !api_observed) camera_scalars_of_the_observation_frame(&s.hdr);
    int written = 0;
    std::string err;
    if (!serialize_state(game_id, d.opt, game_n[e], s, data, length, &written, &err)) {
        if (may_not_fit) return -1;
        fatal("%s\n", err.c_str());
    }
    return written;
}

void VecGame::set_state(int e, const char *data, int length) {  // reference src/vecgame.cpp:447-456
    if (d.opt.use_generated_assets) fatal("fassert failed '!options.use_generated_assets' (BasicAbstractGame::deserialize)\n");  // BAG:1238
    if (!buffers_set) fatal("set_state called before libenv_set_buffers\n");
    if (e < 0 || e >= num_envs) fatal("set_state: env index %d out of range\n", e);
    use_device();
    observe();
    EnvSnapshot s;
    snapshot(e, &s, true);  // fields the wire format does not carry keep their current values
    std::string err;
    int saved_game_n = game_n[e];
    if (!deserialize_state(game_id, d.opt, &s, data, length, &err, &saved_game_n)) fatal("%s\n", err.c_str());
    game_n[e] = saved_game_n;  // (Game::game_n is adopted from the state, reference src/game.cpp:253, and written back by get_state; nothing on the device reads it)
    // routing: the restored env takes a wave = env kernel for one step (conservative bound: a step at most doubles the
    // table); the lists the next step walks are rebuilt from the host's copy of the route table before the next launch,
    // so restoring the same env several times, in any tier order, leaves exactly one entry for it
    // the wire format carries the camera scalars of the LAST frame drawn, which is the 512-pixel one when the state was saved
    // under render_human; Game::observe redraws the 64-pixel frame through prepare_for_drawing(64) (BAG:819-838)
    camera_scalars_of_the_observation_frame(&s.hdr);
    human_stale = true;
    snap_first = -1;
    const int tier = game_tier_for(kernel_id, 2 * s.hdr.n_ents + 4);
    s.hdr.big = tier;
    HIP_CHECK(hipMemcpy(d.ents + ent_table_base(e, d.ent_cap), s.ents.data(), (size_t)EF_COUNT * d.ent_cap * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.rng + (size_t)e * MT_SLOTS * MT_STRIDE, s.rng.data(), s.rng.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.grid + (size_t)e * d.grid_bytes, s.grid.data(), s.grid.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.hdr + e, &s.hdr, sizeof(EnvHdr), hipMemcpyHostToDevice));
    if (!route_mirror_valid) {
        h_route.resize(num_envs);
        HIP_CHECK(hipMemcpy(h_route.data(), d_route[step_count & 1], num_envs, hipMemcpyDeviceToHost));
        route_mirror_valid = true;
    }
    h_route[e] = (uint8_t)tier;
    route_dirty = true;
    // Game::observe(): refresh this env's observation / reward / first / info (reference src/vecgame.cpp:453-455) --
    // only this env's 12 KB frame and scalars move
    const uint8_t first = (uint8_t)s.hdr.done, plc = (uint8_t)s.hdr.level_complete;
    HIP_CHECK(hipMemcpy(d.rew + e, &s.hdr.reward, 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.first + e, &first, 1, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.prev_level_seed + e, &s.hdr.prev_level_seed, 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.prev_level_complete + e, &plc, 1, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.level_seed + e, &s.hdr.current_level_seed, 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipStreamSynchronize(nullptr));  // the uploads above ran on the null stream: joined before the handle's (non-blocking) stream reads them
    // (any error a kernel raised before this call has ended the run in the observe() above: the host-mapped record is checked at every join)
    HIP_CHECK(launch_render_one(kernel_id, d, e, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    check_late_error("the drawing of a restored state");
    rew_ptr[e] = s.hdr.reward;
    first_ptr[e] = first;
    *(int32_t *)info_ptr[0][e] = s.hdr.prev_level_seed;
    *(uint8_t *)info_ptr[1][e] = plc;
    *(int32_t *)info_ptr[2][e] = s.hdr.current_level_seed;
    if (host_observations) HIP_CHECK(hipMemcpy(ob_ptr[e], d.obs + (size_t)e * OBS_BYTES, OBS_BYTES, hipMemcpyDeviceToHost));
}

// set_state for a run of consecutive envs that lie in one snapshot block (procgen_amd_set_states): the block's device state is fetched once,
// every stream is parsed into its env's slice of it (several host threads), the run's slices go back with one copy per array and ONE render
// launch redraws the run -- instead of eleven synchronous copies, a launch and two joins per env (3.4 k states/s in round 5).
void VecGame::set_states(int first, int count, const char *data, const long long *offsets) {
    if (d.opt.use_generated_assets) fatal("fassert failed '!options.use_generated_assets' (BasicAbstractGame::deserialize)\n");  // BAG:1238
    if (!buffers_set) fatal("set_state called before libenv_set_buffers\n");
    if (first < 0 || count < 1 || first + count > num_envs || first / SNAP_BLOCK != (first + count - 1) / SNAP_BLOCK) fatal("set_states: envs [%d, %d) are not a run inside one snapshot block\n", first, first + count);
    use_device();
    observe();
    {
        EnvSnapshot s0;
        snapshot(first, &s0);  // (fills the block cache: fields the wire format does not carry keep their current values)
    }
    const size_t ents_w = (size_t)EF_COUNT * d.ent_cap, rng_w = 2 * MT_STRIDE, grid_b = (size_t)d.grid_bytes;
    const size_t k0 = (size_t)(first - snap_first);
    std::vector<std::string> errs((size_t)count);
    std::vector<char> ok((size_t)count, 1);
    std::vector<int> gn((size_t)count);
    int threads = (int)std::thread::hardware_concurrency();
    threads = threads > 16 ? 16 : (threads < 1 ? 1 : threads);
    if (threads > count / 8) threads = count / 8 > 0 ? count / 8 : 1;
    auto work = [&](int t) {
        for (int j = t; j < count; j += threads) {
            const size_t k = k0 + (size_t)j;
            EnvSnapshot s;
            s.ent_cap = d.ent_cap;
            s.hdr = snap_hdr[k];
            s.ents.assign(snap_ents.begin() + k * ents_w, snap_ents.begin() + (k + 1) * ents_w);
            s.rng.assign(snap_rng.begin() + k * rng_w, snap_rng.begin() + (k + 1) * rng_w);
            s.grid.assign(snap_grid.begin() + k * grid_b, snap_grid.begin() + (k + 1) * grid_b);
            gn[(size_t)j] = game_n[first + j];
            if (!deserialize_state(game_id, d.opt, &s, data + offsets[j], (int)(offsets[j + 1] - offsets[j]), &errs[(size_t)j], &gn[(size_t)j])) {
                ok[(size_t)j] = 0;
                continue;
            }
            camera_scalars_of_the_observation_frame(&s.hdr);  // (see set_state)
            s.hdr.big = game_tier_for(kernel_id, 2 * s.hdr.n_ents + 4);
            snap_hdr[k] = s.hdr;
            std::copy(s.ents.begin(), s.ents.end(), snap_ents.begin() + k * ents_w);
            std::copy(s.rng.begin(), s.rng.end(), snap_rng.begin() + k * rng_w);
            std::copy(s.grid.begin(), s.grid.end(), snap_grid.begin() + k * grid_b);
        }
    };
    if (threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    for (int j = 0; j < count; j++)
        if (!ok[(size_t)j]) fatal("%s\n", errs[(size_t)j].c_str());
    human_stale = true;
    if (!route_mirror_valid) {
        h_route.resize(num_envs);
        HIP_CHECK(hipMemcpy(h_route.data(), d_route[step_count & 1], num_envs, hipMemcpyDeviceToHost));
        route_mirror_valid = true;
    }
    std::vector<float> rew((size_t)count);
    std::vector<uint8_t> first_v((size_t)count), plc((size_t)count);
    std::vector<int32_t> pls((size_t)count), ls((size_t)count);
    for (int j = 0; j < count; j++) {
        const EnvHdr &h = snap_hdr[k0 + (size_t)j];
        game_n[first + j] = gn[(size_t)j];
        h_route[first + j] = (uint8_t)h.big;
        rew[(size_t)j] = h.reward;
        first_v[(size_t)j] = (uint8_t)h.done;
        plc[(size_t)j] = (uint8_t)h.level_complete;
        pls[(size_t)j] = h.prev_level_seed;
        ls[(size_t)j] = h.current_level_seed;
    }
    route_dirty = true;
    HIP_CHECK(hipMemcpy(d.hdr + first, snap_hdr.data() + k0, sizeof(EnvHdr) * count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.ents + ent_table_base(first, d.ent_cap), snap_ents.data() + k0 * ents_w, ents_w * 4 * count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy2D(d.rng + (size_t)first * MT_SLOTS * MT_STRIDE, (size_t)MT_SLOTS * MT_STRIDE * 4, snap_rng.data() + k0 * rng_w, rng_w * 4, rng_w * 4, count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.grid + (size_t)first * grid_b, snap_grid.data() + k0 * grid_b, grid_b * count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.rew + first, rew.data(), 4 * (size_t)count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.first + first, first_v.data(), (size_t)count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.prev_level_seed + first, pls.data(), 4 * (size_t)count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.prev_level_complete + first, plc.data(), (size_t)count, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.level_seed + first, ls.data(), 4 * (size_t)count, hipMemcpyHostToDevice));
    HIP_CHECK(hipStreamSynchronize(nullptr));  // the uploads ran on the null stream: joined before the handle's (non-blocking) stream reads them
    snap_first = -1;
    HIP_CHECK(launch_render_one(kernel_id, d, first, stream, count));
    HIP_CHECK(hipStreamSynchronize(stream));
    check_late_error("the drawing of restored states");
    for (int j = 0; j < count; j++) {
        const int e = first + j;
        rew_ptr[e] = rew[(size_t)j];
        first_ptr[e] = first_v[(size_t)j];
        *(int32_t *)info_ptr[0][e] = pls[(size_t)j];
        *(uint8_t *)info_ptr[1][e] = plc[(size_t)j];
        *(int32_t *)info_ptr[2][e] = ls[(size_t)j];
        if (host_observations) HIP_CHECK(hipMemcpy(ob_ptr[e], d.obs + (size_t)e * OBS_BYTES, OBS_BYTES, hipMemcpyDeviceToHost));
    }
}

// the tier lists and the route table the coming step reads, rebuilt from the host's copy after set_state calls
void VecGame::flush_routes() {
    const int cur = (int)(step_count & 1);
    std::vector<int> lists((size_t)num_envs * NUM_TIERS);
    int count[MAX_CHUNKS][NUM_TIERS] = {};
    for (int e = 0; e < num_envs; e++) {
        const int t = h_route[e], c = e / d.chunk_envs;
        if (t == 1 || t == 2) lists[(size_t)t * num_envs + (size_t)c * d.chunk_envs + count[c][t]++] = e;
    }
    HIP_CHECK(hipMemcpy(d_route[cur], h_route.data(), num_envs, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_big_list[cur], lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_big_count[cur], count, sizeof(count), hipMemcpyHostToDevice));
    HIP_CHECK(hipStreamSynchronize(nullptr));  // (null-stream uploads, read next by kernels on the handle's non-blocking streams)
    memcpy(host_list_count, count, sizeof(count));
    route_dirty = false;
}

}  // namespace

// Host threads that issue the HIP calls of the parts of a multi-part handle side by side: a step of a 16-game handle is
// ~150 runtime calls (copies, memsets, launches, stream joins) which one thread issues in over a millisecond -- longer
// than the GPU needs for the kernels (reference analogue: the stepping threads of src/vecgame.cpp:103-142, there for the
// game logic itself).  run(n, f) calls f(0) .. f(n-1) across the workers and returns when all are done.
class PartPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *job = nullptr;
    int next = 0, total = 0, running = 0;
    uint64_t epoch = 0;
    bool stop = false;
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || (epoch != seen && next < total); });
            if (stop) return;
            while (next < total) {
                const int i = next++;
                running++;
                lk.unlock();
                (*job)(i);
                lk.lock();
                running--;
            }
            seen = epoch;
            if (running == 0) cv_done.notify_all();
        }
    }

public:
    explicit PartPool(int threads) {
        for (int t = 0; t < threads; t++) workers.emplace_back([this] { loop(); });
    }
    ~PartPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &w : workers) w.join();
    }
    void run(int n, const std::function<void(int)> &f) {
        std::unique_lock<std::mutex> lk(m);
        job = &f;
        next = 0;
        total = n;
        epoch++;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return next >= total && running == 0; });
        job = nullptr;
    }
};

// One libenv handle = G device shards x K games (shard_map.h): part (g, k) is a VecGame of its own -- game k of a comma
// separated env_name (reference src/vecgame.cpp:295-310: env n plays names[n % K]) over the envs base_g + k + K * i of
// device g's contiguous index range -- with its own streams, so the kernels of the parts of one libenv_act run
// concurrently, on one GPU or on several ("num_devices" option; no collective: envs never interact).
struct Handle {
    std::vector<std::unique_ptr<VecGame>> parts;
    ShardMap map;
    int num_envs = 0;
    // multi-part handles: the parts write rew / first into these and observe() scatters them to the caller's arrays
    std::vector<std::vector<float>> part_rew;
    std::vector<std::vector<uint8_t>> part_first;
    std::vector<std::vector<void *>> part_ob, part_ac, part_info;
    float *rew = nullptr;
    uint8_t *first = nullptr;
    void *pinned_ob = nullptr;  // the caller's whole observation array, registered once for all devices
    std::unique_ptr<PartPool> pool;  // multi-part handles
    std::vector<int> order;          // part indices, heaviest game first (creation and launch order)
    int P() const { return (int)parts.size(); }
    void for_parts(const std::function<void(int)> &f) {  // in `order`
        if (pool) pool->run(P(), [&](int k) { f(order[k]); });
        else
            for (int k = 0; k < P(); k++) f(order[k]);
    }
    VecGame *single() {
        if (parts.size() != 1) fatal("this extension hook is only available on single-game, single-device handles\n");
        return parts[0].get();
    }
    ~Handle() {
        pool.reset();
        parts.clear();
        if (pinned_ob) (void)hipHostUnregister(pinned_ob);
    }
};

// The ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
// queue run their kernels one after the other.  A joint handle has one stream per game: with 16 queues its 16-game step
// takes 1.7 ms instead of 2.5.  The variable is read when the runtime initialises and changes the
// stream-to-queue mapping of every HIP user of the process, so loading this library does NOT touch it by default: the Python
// package (procgen_amd/__init__.py) and bench.py set it, a C / cffi host sets it itself or asks for it with
// PROCGEN_AMD_SET_HW_QUEUES=1 (opt-in since round 4; libenv_make warns when a many-part handle finds fewer than 8 queues).
__attribute__((constructor)) static void procgen_amd_default_hw_queues() {
    const char *want = getenv("PROCGEN_AMD_SET_HW_QUEUES");
    if (want && atoi(want) != 0) setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

// relative length of one step of a ~1000-env part: the part's kernel chain (step -> reset_list / list kernels -> render) in units of
// ~50 us, from the joint handle's kernel trace (profiles/r03_kernel_trace_all16_joint.csv, the generators of round 3 included:
// leaper 68 + 565 + 111 us, jumper 55 + 345 + 150, coinrun 250 + 122 with its list kernels beside, ... chaser 56 + 86)
static int part_cost_rank(const std::string &name) {
    static const struct { const char *name; int cost; } table[] = {
        {"leaper", 15}, {"jumper", 11}, {"coinrun", 10}, {"bossfight", 10}, {"caveflyer", 9}, {"heist", 7}, {"fruitbot", 7}, {"starpilot", 6},
        {"maze", 6}, {"ninja", 5}, {"dodgeball", 4}, {"miner", 4}, {"plunder", 4}, {"bigfish", 4}, {"climber", 3}, {"chaser", 3}};
    for (const auto &t : table)
        if (name == t.name) return t.cost;
    return 1;
}

static std::vector<std::string> split_names(const std::string &s) {
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        const size_t k = s.find(',', pos);
        out.push_back(s.substr(pos, k == std::string::npos ? std::string::npos : k - pos));
        if (k == std::string::npos) break;
        pos = k + 1;
    }
    return out;
}

extern "C" {

LIBENV_API int libenv_version(void) { return LIBENV_VERSION; }

LIBENV_API libenv_env *libenv_make(int num_envs, const struct libenv_options options) {
    Handle *h = new Handle();
    h->num_envs = num_envs;
    std::string env_name;
    int num_devices = 1, device_id = -1;
    bool devices_from_env = false;
    {
        VecOptions peek(options);
        peek.consume_string("env_name", &env_name);
        const bool has_dev = peek.consume_int("device_id", &device_id);
        if (!peek.consume_int("num_devices", &num_devices) && !has_dev && getenv("PROCGEN_AMD_NUM_DEVICES")) {
            num_devices = atoi(getenv("PROCGEN_AMD_NUM_DEVICES"));
            devices_from_env = true;
        }
    }
    const std::vector<std::string> names = split_names(env_name);
    const int K = (int)names.size();
    if (num_devices == 0) {  // 0 = every visible device
        if (hipGetDeviceCount(&num_devices) != hipSuccess || num_devices <= 0) fatal("no HIP device available: the MI355X stepper cannot run (there is no CPU fallback)\n");
    }
    if (num_devices < 1) fatal("num_devices must be positive (or 0 for all visible devices)\n");
    if (devices_from_env && num_devices > 1 && num_envs % (num_devices * K) != 0) {
        // the environment variable shards every handle of the process; one that cannot be cut evenly (a 1-env evaluation
        // env next to the training vector) stays on one device instead of failing
        fprintf(stderr, "procgen_amd: PROCGEN_AMD_NUM_DEVICES=%d ignored for a handle of %d envs (not a multiple of %d x %d games): one device\n", num_devices, num_envs, num_devices, K);
        num_devices = 1;
    }
    if (K > 1 && num_envs % K != 0) fatal("fassert failed 'num_envs %% num_joint_games == 0'\n");
    h->map.num_envs = num_envs;
    h->map.num_devices = num_devices;
    h->map.num_games = K;
    if (!h->map.valid()) fatal("num_envs (%d) must be a multiple of num_devices x number of games (%d x %d)\n", num_envs, num_devices, K);
    if (K <= 1 && num_devices == 1) {
        h->parts.emplace_back(new VecGame(num_envs, VecOptions(options)));
        h->order.push_back(0);
    } else {
        const int first_device = device_id >= 0 ? device_id : 0;
        const int P = h->map.parts();
        // The runtime deals streams onto its (four) hardware queues in creation order, and the parts that share a queue run one
        // after the other.  A step of a part lasts as long as its slowest env -- with ~1000 envs per game that is a level
        // generation, 0.1 ms for most games and 0.9-1.6 ms for caveflyer / jumper / leaper -- so the parts are created (and
        // later launched) heaviest first, dealt over the queues in snake order (longest-processing-time scheduling).
        std::vector<int> by_cost(P);
        for (int p = 0; p < P; p++) by_cost[p] = p;
        auto cost = [&](int p) { return part_cost_rank(names[h->map.game_of_part(p)]); };
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return cost(a) > cost(b); });
        // (the queues the runtime really has: GPU_MAX_HW_QUEUES as the process set it, 4 when unset)
        int Q = 4;
        if (const char *q = getenv("GPU_MAX_HW_QUEUES")) Q = atoi(q) > 0 ? (atoi(q) < 64 ? atoi(q) : 64) : 4;
        for (int r = 0; r * Q < P; r++)
            for (int q = 0; q < Q; q++) {
                const int k = r * Q + ((r & 1) ? Q - 1 - q : q);
                if (k < P) h->order.push_back(by_cost[k]);
            }
        h->parts.resize(P);
        for (int p : h->order)
            h->parts[p].reset(new VecGame(h->map.envs_per_part(), VecOptions(options), names[h->map.game_of_part(p)], K, h->map.first_env(p),
                                          num_devices > 1 ? first_device + h->map.device_of_part(p) : -1));
        {
            // one stream per part: with the runtime's default of four hardware queues the parts of a 16-game handle run four deep
            const char *q = getenv("GPU_MAX_HW_QUEUES");
            static bool warned = false;
            if (!warned && h->P() > 4 && (!q || atoi(q) < 8)) {
                warned = true;
                fprintf(stderr, "procgen_amd: %d parts on %s hardware queues: set GPU_MAX_HW_QUEUES=16 before the process's first HIP call (INTEGRATION.md section 5) for ~1.5x on joint handles\n", h->P(), q ? q : "the default 4");
            }
        }
        // issuing threads: the runtime serialises calls per device, so more threads than ~8 per device only contend (the 16 parts of a
        // one-device joint handle: 16.4 M steps/s with 8 threads, 15.9 with 16, 16.2 with 4 -- profiles/r05_joint_threads_ab.txt); a handle
        // over G devices gets 2 per device more, up to the host's cores and 32
        int threads = 8;
        if (num_devices > 1) threads += 2 * num_devices;
        if (threads > h->P()) threads = h->P();
        const int cores = (int)std::thread::hardware_concurrency();
        if (cores > 0 && threads > cores) threads = cores;
        if (threads > 32) threads = 32;
        if (const char *t = getenv("PROCGEN_AMD_HOST_THREADS")) threads = atoi(t);
        if (threads > 1) h->pool.reset(new PartPool(threads));
    }
    return (libenv_env *)h;
}

LIBENV_API int libenv_get_tensortypes(libenv_env *handle, enum libenv_space_name name, struct libenv_tensortype *out_types) {
    VecGame *v = ((Handle *)handle)->parts[0].get();
    const std::vector<libenv_tensortype> *types;
    if (name == LIBENV_SPACE_OBSERVATION) types = &v->observation_types;
    else if (name == LIBENV_SPACE_ACTION) types = &v->action_types;
    else if (name == LIBENV_SPACE_INFO) types = &v->info_types;
    else return 0;
    if (out_types != nullptr)
        for (size_t i = 0; i < types->size(); i++) out_types[i] = (*types)[i];
    return (int)types->size();
}

LIBENV_API void libenv_set_buffers(libenv_env *handle, struct libenv_buffers *bufs) {
    Handle *h = (Handle *)handle;
    const int P = h->P();
    if (P == 1) {
        h->parts[0]->set_buffers(bufs);
        return;
    }
    const int N = h->num_envs, n = h->map.envs_per_part();
    h->rew = bufs->rew;
    h->first = bufs->first;
    {   // "one pinned host buffer": when the caller's observation pointers form one array, register it once for every device
        bool contig = true;
        for (int e = 1; e < N && contig; e++) contig = (uint8_t *)bufs->ob[e] == (uint8_t *)bufs->ob[0] + (size_t)e * OBS_BYTES;
        if (contig && h->parts[0]->host_observations && hipHostRegister(bufs->ob[0], (size_t)N * OBS_BYTES, hipHostRegisterPortable) == hipSuccess) h->pinned_ob = bufs->ob[0];
        else (void)hipGetLastError();
    }
    h->part_rew.assign(P, std::vector<float>(n));
    h->part_first.assign(P, std::vector<uint8_t>(n));
    h->part_ob.assign(P, std::vector<void *>(n));
    h->part_ac.assign(P, std::vector<void *>(n));
    const int S = h->parts[0]->render_human ? 4 : 3;  // info spaces (reference src/vecgame.cpp:212-282)
    h->part_info.assign(P, std::vector<void *>((size_t)S * n));
    for (int p = 0; p < P; p++) {
        for (int i = 0; i < n; i++) {
            const int e = h->map.env_of(p, i);  // global env index
            h->part_ob[p][i] = bufs->ob[e];
            h->part_ac[p][i] = bufs->ac[e];
            for (int s = 0; s < S; s++) h->part_info[p][(size_t)s * n + i] = bufs->info[(size_t)s * N + e];
        }
        struct libenv_buffers pb;
        pb.ob = h->part_ob[p].data();
        pb.rew = h->part_rew[p].data();
        pb.first = h->part_first[p].data();
        pb.info = h->part_info[p].data();
        pb.ac = h->part_ac[p].data();
        h->parts[p]->external_pinned = h->pinned_ob != nullptr;
        h->parts[p]->set_buffers(&pb);
    }
}
LIBENV_API void libenv_observe(libenv_env *handle) {
    Handle *h = (Handle *)handle;
    const int P = h->P();
    h->for_parts([&](int p) { h->parts[p]->observe(true); });  // joins the streams of every part (device)
    if (P > 1 && h->rew) {
        const int n = h->map.envs_per_part();
        for (int p = 0; p < P; p++)
            for (int i = 0; i < n; i++) {
                const int e = h->map.env_of(p, i);
                h->rew[e] = h->part_rew[p][i];
                h->first[e] = h->part_first[p][i];
            }
    }
}
LIBENV_API void libenv_act(libenv_env *handle) {
    Handle *h = (Handle *)handle;
    h->for_parts([&](int p) { h->parts[p]->act(); });  // each part launches on its own streams (and device): the parts' kernels overlap
}
LIBENV_API void libenv_close(libenv_env *handle) { delete (Handle *)handle; }

// reference src/vecgame.cpp:437-457 (wire format: state_io.cpp)
LIBENV_API int get_state(libenv_env *handle, int env_idx, char *data, int length) {
    Handle *h = (Handle *)handle;
    if (env_idx < 0 || env_idx >= h->num_envs) fatal("get_state: env index %d out of range\n", env_idx);
    return h->parts[h->map.part_of(env_idx)]->get_state(h->map.index_in_part(env_idx), data, length);
}
// get_state of envs [first, first + count), packed back to back: state k occupies data[offsets[k], offsets[k + 1]).  Returns how many
// states fit into `capacity` bytes (the caller goes on from there); the device state of 256 consecutive envs of a part moves in four
// copies (VecGame::snapshot), and the caller crosses the FFI once per block instead of once per env.
LIBENV_API int procgen_amd_get_states(libenv_env *handle, int first, int count, char *data, long long capacity, long long *offsets) {
    Handle *h = (Handle *)handle;
    if (first < 0 || count < 0 || first + count > h->num_envs) fatal("procgen_amd_get_states: envs [%d, %d) out of range\n", first, first + count);
    long long off = 0;
    offsets[0] = 0;
    // Runs of consecutive envs that share a part and a snapshot block (single-part handles: up to 256 envs) are serialized by several host
    // threads at once out of the block's cached device state -- the byte streams are built field by field (16 KB of grid ints, 5 KB of
    // generator state, 124 B per entity: ~40 KB per coinrun env), which one thread does at ~10 k states/s.
    int k = 0;
    while (k < count) {
        const int e0 = first + k;
        VecGame *v = h->parts[h->map.part_of(e0)].get();
        const int i0 = h->map.index_in_part(e0);
        v->load_snapshot_block(i0);
        int run = 1;
        while (k + run < count && h->map.part_of(first + k + run) == h->map.part_of(e0) && h->map.index_in_part(first + k + run) == i0 + run &&
               (i0 + run) / VecGame::SNAP_BLOCK == i0 / VecGame::SNAP_BLOCK)
            run++;
        std::vector<std::string> out((size_t)run), errs((size_t)run);
        std::vector<char> ok((size_t)run, 1);
        int threads = (int)std::thread::hardware_concurrency();
        threads = threads > 16 ? 16 : (threads < 1 ? 1 : threads);
        if (threads > run / 8) threads = run / 8 > 0 ? run / 8 : 1;
        auto work = [&](int t) {
            for (int j = t; j < run; j += threads) ok[(size_t)j] = v->serialize_cached(i0 + j, &out[(size_t)j], &errs[(size_t)j]) ? 1 : 0;
        };
        if (threads <= 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
            work(0);
            for (auto &th : pool) th.join();
        }
        for (int j = 0; j < run; j++) {
            if (!ok[(size_t)j]) fatal("%s\n", errs[(size_t)j].c_str());
            const long long n = (long long)out[(size_t)j].size();
            if (off + n > capacity) {
                if (k + j == 0) fatal("procgen_amd_get_states: %lld bytes do not hold one state\n", capacity);
                return k + j;
            }
            memcpy(data + off, out[(size_t)j].data(), (size_t)n);
            off += n;
            offsets[k + j + 1] = off;
        }
        k += run;
    }
    return count;
}
LIBENV_API void set_state(libenv_env *handle, int env_idx, char *data, int length) {
    Handle *h = (Handle *)handle;
    if (env_idx < 0 || env_idx >= h->num_envs) fatal("set_state: env index %d out of range\n", env_idx);
    h->parts[h->map.part_of(env_idx)]->set_state(h->map.index_in_part(env_idx), data, length);
    if (h->P() > 1) libenv_observe(handle);  // refresh the caller's rew / first entries of this env
}

// set_state for envs [first, first + count): state j is data[offsets[j], offsets[j + 1]).  Runs of consecutive envs that share a part and a
// snapshot block are restored together (VecGame::set_states); the reference restores env by env (procgen/env.py:148-153, src/vecgame.cpp:447-456)
LIBENV_API void procgen_amd_set_states(libenv_env *handle, int first, int count, const char *data, const long long *offsets) {
    Handle *h = (Handle *)handle;
    if (first < 0 || count < 0 || first + count > h->num_envs) fatal("procgen_amd_set_states: envs [%d, %d) out of range\n", first, first + count);
    int k = 0;
    while (k < count) {
        const int e0 = first + k;
        VecGame *v = h->parts[h->map.part_of(e0)].get();
        const int i0 = h->map.index_in_part(e0);
        int run = 1;
        while (k + run < count && h->map.part_of(first + k + run) == h->map.part_of(e0) && h->map.index_in_part(first + k + run) == i0 + run &&
               (i0 + run) / VecGame::SNAP_BLOCK == i0 / VecGame::SNAP_BLOCK)
            run++;
        v->set_states(i0, run, data, offsets + k);
        k += run;
    }
    if (h->P() > 1) libenv_observe(handle);  // refresh the caller's rew / first entries
}

// ---- extension hooks (include/procgen_amd.h) -------------------------------------------------------------
LIBENV_API int procgen_amd_device_buffers(libenv_env *handle, struct procgen_amd_buffers *out) {
    VecGame *v = ((Handle *)handle)->single();
    out->device_id = v->device_id;
    out->num_envs = v->num_envs;
    out->stream = (void *)v->stream;
    out->ob = v->d.obs;
    out->rew = v->d.rew;
    out->first = v->d.first;
    out->prev_level_seed = v->d.prev_level_seed;
    out->prev_level_complete = v->d.prev_level_complete;
    out->level_seed = v->d.level_seed;
    out->action = v->d_action;
    return 0;
}
LIBENV_API int procgen_amd_part_buffers(libenv_env *handle, struct procgen_amd_part *out, int max_parts) {
    Handle *h = (Handle *)handle;
    for (int p = 0; out && p < h->P() && p < max_parts; p++) {
        VecGame *v = h->parts[p].get();
        procgen_amd_buffers &b = out[p].buffers;
        b.device_id = v->device_id;
        b.num_envs = v->num_envs;
        b.stream = (void *)v->stream;
        b.ob = v->d.obs;
        b.rew = v->d.rew;
        b.first = v->d.first;
        b.prev_level_seed = v->d.prev_level_seed;
        b.prev_level_complete = v->d.prev_level_complete;
        b.level_seed = v->d.level_seed;
        b.action = v->d_action;
        out[p].first_env = h->P() == 1 ? 0 : h->map.first_env(p);
        out[p].env_stride = h->P() == 1 ? 1 : h->map.num_games;
        memset(out[p].game, 0, sizeof(out[p].game));
        strncpy(out[p].game, game_name_from_id(v->game_id), sizeof(out[p].game) - 1);
    }
    return h->P();
}
LIBENV_API double procgen_amd_kernel_timing(libenv_env *handle, int enable, int *steps_out, double *render_out) {
    VecGame *v = ((Handle *)handle)->single();
    v->use_device();
    v->observe();
    v->collect_timing(0);  // (the stream is joined: every recorded event has completed)
    v->collect_timing(1);
    const double mean = v->tk_steps > 0 ? v->tk_sum_ms / v->tk_steps : 0.0;
    if (steps_out) *steps_out = v->tk_steps;
    if (render_out) {
        render_out[0] = v->tk_render_launches > 0 ? v->tk_render_ms / v->tk_render_launches : 0.0;
        render_out[1] = v->tk_steps > 0 ? (double)v->tk_render_launches / v->tk_steps : 0.0;
    }
    if (enable) {
        if (!v->tk_e0[0]) {
            for (int k = 0; k < 2; k++) {
                HIP_CHECK(hipEventCreate(&v->tk_e0[k]));
                HIP_CHECK(hipEventCreate(&v->tk_e1[k]));
                for (int c = 0; c < MAX_CHUNKS; c++) {
                    HIP_CHECK(hipEventCreate(&v->tk_r0[k][c]));
                    HIP_CHECK(hipEventCreate(&v->tk_r1[k][c]));
                }
            }
        }
        v->tk_sum_ms = v->tk_render_ms = 0;
        v->tk_steps = v->tk_render_launches = 0;
    }
    v->time_kernels = enable != 0;
    v->time_render = enable == 2;
    return mean;
}
// the render kernel's launch order as the device holds it (single-part handles): out[slot] = env drawn by workgroup `slot`; returns the number
// of entries written (num_envs), 0 when the handle launches in env order.  chunk_out (may be NULL): [0] envs of the first launch chunk, [1] chunks.
LIBENV_API int procgen_amd_render_order(libenv_env *handle, int *out, int max_envs, int *chunk_out) {
    VecGame *v = ((Handle *)handle)->single();
    v->observe();
    HIP_CHECK(hipSetDevice(v->device_id));
    const int N = v->num_envs;
    const int nchunk = N >= 4096 ? (v->chunks > 1 ? (v->chunks < MAX_CHUNKS ? v->chunks : MAX_CHUNKS) : 1) : 1;
    const int first = (nchunk == 2 && v->first_pct > 0) ? first_chunk_envs(N, v->first_pct) : 0;
    if (chunk_out) {
        chunk_out[0] = first > 0 ? first : chunk_envs_for(N, nchunk);
        chunk_out[1] = nchunk;
    }
    if (!v->d.render_order || max_envs < N) return 0;
    HIP_CHECK(hipStreamSynchronize(v->stream));
    HIP_CHECK(hipMemcpy(out, v->d.render_order, (size_t)N * sizeof(int), hipMemcpyDeviceToHost));
    return N;
}
// display-list games (pg_prep.h; single-part handles): out[0] = envs whose current frame the rasterizer drew from its record, out[1] = envs
// whose frame went to the full renderer; returns 1, or 0 when the handle renders with one kernel (then out is untouched)
LIBENV_API int procgen_amd_display_list_frames(libenv_env *handle, int *out) {
    VecGame *v = ((Handle *)handle)->single();
    v->observe();
    const int rec_words = game_frame_rec_words(v->kernel_id);
    if (!v->d_frame_rec || rec_words <= 0) return 0;
    if (!v->d.frame_rec) {  // the handle went back to the one-kernel renderer (most frames left the short path)
        out[0] = 0;
        out[1] = v->num_envs;
        return 1;
    }
    HIP_CHECK(hipSetDevice(v->device_id));
    HIP_CHECK(hipStreamSynchronize(v->stream));
    std::vector<uint32_t> flags((size_t)v->num_envs);
    HIP_CHECK(hipMemcpy2D(flags.data(), sizeof(uint32_t), v->d.frame_rec, (size_t)rec_words * sizeof(uint32_t), sizeof(uint32_t), (size_t)v->num_envs, hipMemcpyDeviceToHost));
    int fast = 0;
    for (uint32_t f : flags) fast += (int)(f & 1u);
    out[0] = fast;
    out[1] = v->num_envs - fast;
    if (getenv("PROCGEN_AMD_DISPLAY_LIST_REPORT")) {  // measurement aid: how often the full renderer's pass ran, and why the first few frames went there
        fprintf(stderr, "[procgen_amd display list] %d of %d frames from their record; %lld slow passes in %llu steps\n", fast, v->num_envs, v->slow_passes, (unsigned long long)v->step_count);
        int shown = 0;
        for (int e = 0; e < v->num_envs && shown < 8; e++)
            if (!(flags[(size_t)e] & 1u)) {
                fprintf(stderr, "  env %d: flags 0x%08x (bit 8 too many / failed commands, 9 error, 10 monochrome, 11 vel info, 12 no pull form; error code %u)\n", e, flags[(size_t)e], flags[(size_t)e] >> 16);
                shown++;
            }
    }
    return 1;
}
// host only: out[p] = render_order_slot(p, count) (shard_map.h) for p in [0, count)
LIBENV_API void procgen_amd_selftest_render_order_slots(int count, int *out) {
    for (int p = 0; p < count; p++) out[p] = render_order_slot(p, count);
}
LIBENV_API int procgen_amd_tier_counts(libenv_env *handle, int *out) {
    VecGame *v = ((Handle *)handle)->single();
    v->observe();
    int t1 = 0, t2 = 0;
    for (int c = 0; c < MAX_CHUNKS; c++) {
        t1 += v->host_list_count[c][1];
        t2 += v->host_list_count[c][2];
    }
    out[0] = v->num_envs - t1 - t2;
    out[1] = t1;
    out[2] = t2;
    return v->num_envs;
}
LIBENV_API void procgen_amd_set_host_observations(libenv_env *handle, int enable) {
    VecGame *v = ((Handle *)handle)->single();
    v->observe();
    if (enable && !v->host_observations && v->buffers_set && !v->ob_contig && !v->h_obs_stage)
        HIP_CHECK(hipHostMalloc((void **)&v->h_obs_stage, (size_t)v->num_envs * OBS_BYTES, hipHostMallocDefault));
    v->host_observations = enable != 0;
}
// Device math self-tests: no handle; run on the current device.  Host pointers in, host pointers out.
LIBENV_API void procgen_amd_selftest_bigfish_radius(const float *r01, float *out, int n) {
    float *d_in = nullptr, *d_out = nullptr;
    HIP_CHECK(hipMalloc((void **)&d_in, (size_t)n * 4));
    HIP_CHECK(hipMalloc((void **)&d_out, (size_t)n * 4));
    HIP_CHECK(hipMemcpy(d_in, r01, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(selftest_bigfish_radius(d_in, d_out, n));
    HIP_CHECK(hipMemcpy(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
}
LIBENV_API void procgen_amd_selftest_sincos(uint32_t first_bits, int n, double *out_sin, double *out_cos) {
    double *d_s = nullptr, *d_c = nullptr;
    HIP_CHECK(hipMalloc((void **)&d_s, (size_t)n * 8));
    HIP_CHECK(hipMalloc((void **)&d_c, (size_t)n * 8));
    HIP_CHECK(selftest_sincos(first_bits, n, d_s, d_c));
    HIP_CHECK(hipMemcpy(out_sin, d_s, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_cos, d_c, (size_t)n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_s);
    (void)hipFree(d_c);
}

LIBENV_API void procgen_amd_selftest_sincos_scaled(const uint32_t *bits, int n, double scale, float *out_sin, float *out_cos) {
    uint32_t *d_b = nullptr;
    float *d_s = nullptr, *d_c = nullptr;
    HIP_CHECK(hipMalloc((void **)&d_b, (size_t)n * 4));
    HIP_CHECK(hipMalloc((void **)&d_s, (size_t)n * 4));
    HIP_CHECK(hipMalloc((void **)&d_c, (size_t)n * 4));
    HIP_CHECK(hipMemcpy(d_b, bits, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(selftest_sincos_scaled(d_b, n, scale, d_s, d_c));
    HIP_CHECK(hipMemcpy(out_sin, d_s, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_cos, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_b);
    (void)hipFree(d_s);
    (void)hipFree(d_c);
}

// average device time of one step's launch sequence -- exactly what libenv_act enqueues (VecGame::launch_kernels: list / reset /
// step / render kernels, empty lists skipped) -- over the given number of rounds, measured
// with HIP events on the library's stream (bench.py roofline leg)
LIBENV_API double procgen_amd_time_steps(libenv_env *handle, int steps, const int32_t *actions_or_null) {
    VecGame *v = ((Handle *)handle)->single();
    v->observe();
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    float total_ms = 0.f;
    for (int s = 0; s < steps; s++) {
        if (v->d_render_order && v->step_count % (uint64_t)v->render_order_period == 0) v->rebuild_render_order();
        if (actions_or_null) HIP_CHECK(hipMemcpyAsync(v->d_action, actions_or_null + (size_t)s * v->num_envs, (size_t)v->num_envs * 4, hipMemcpyHostToDevice, v->stream));
        HIP_CHECK(hipEventRecord(e0, v->stream));
        v->launch_kernels(1);
        HIP_CHECK(hipEventRecord(e1, v->stream));
        HIP_CHECK(hipMemcpyAsync(v->h_small, v->d_small, v->small_bytes, hipMemcpyDeviceToHost, v->stream));
        HIP_CHECK(hipStreamSynchronize(v->stream));
        v->read_tail();
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
    }
    v->pending = true;
    v->observe();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)total_ms / steps;
}
}
