// pz_kernels.hip -- gfx950 kernels + C ABI of libpikazoo_hip.so (see include/pikazoo_hip.h).
//
// Launch geometry: one lane per game, 64 consecutive games per workgroup.  A workgroup reads its 44 state
// columns with fully coalesced dword loads (256 B per wave instruction), runs the frame in registers, and
// writes the two row-major [n][35] observation tensors through an LDS transpose so that the global stores
// are contiguous 16-byte-per-lane streams (a wave's 64 rows are one contiguous 8 960-byte span).  The
// 65 536-game batch is 1 024 independent workgroups (4 per CU).
//   * single-frame launches below 393 216 games: TWO waves per workgroup, split by player
//     (step_pair_kernel / step_games_pair) -- a launch lasts about as long as one wave's
//     load -> frame -> store chain plus the write drain, and the split halves the frame;
//   * k-frame launches and larger batches: one wave per workgroup (step_kernel / step_games);
//   * a computer player without flight tables: frame wave + scout wave (step_kernel<..., SCOUT>).
// The state is either those 44 int32 columns (the default) or the packed format of pz_packed.hpp (36 bytes per
// game in three columns; template parameter PACKED): the frame is the same code, only the loads and stores differ,
// and the pair kernel then serves every batch size.
//
// At these batch sizes a SIMD runs one or two waves, so nothing hides a wave's own instruction latency:
// the per-wave timeline (tools/stamps.py) is load -> frame -> stores, serialized.  All global
// traffic therefore goes through buffer descriptors (SRD in SGPRs + 32-bit lane offset + scalar
// column offset): a memory instruction needs no per-lane 64-bit address arithmetic, which with
// plain pointers formed a dependent v_lshl_add_u64 chain in front of every one of the 44+44
// column accesses, and rows past the end of the batch are dropped by the hardware range check.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "pikazoo_hip.h"
#include "pz_physics.hpp"
#include "pz_packed.hpp"
#include "pz_memory.hpp"

namespace pz {

// ---- measured choices (closed experiments: the value and its one-line result; logs under profiles/) -----------------
// cache policies of the k-frame launches (aux operand: 1 = sc0, 2 = nt, 16 = sc1; pz_memory.hpp has the single-frame ones)
// The k-frame launches' observation rows (623 MB per 32-frame launch): sc0 sc1 nt.  With a computer player, rows written
// `nt` alone (the single-frame launches' policy) push the flight tables' hot lines out of the caches and the look-ups
// come from HBM: pz_rollout_random, k = 32, interleaved on one box: nt 3.21-3.25 us per frame, sc1 nt 3.13-3.15,
// sc0 sc1 nt 3.12-3.14 (plain 4.2, sc1 alone 4.3); human vs human 2.83 / 2.82 / 2.82.  A single-frame launch keeps
// `nt` (sc1 nt: 6.99 -> 7.08 us; with a computer player 8.87 -> 8.82).
constexpr int kTrajAux = 19;
// the k-frame launches' rewards / flags / actions (17 of a game-step's 297 bytes): sc1 nt where the launch gathers
// from the flight tables (3.13 -> 3.11 us per frame, k = 128: 2.99 -> 2.94), plain otherwise (nt: 2.81 -> 2.94)
constexpr int traj_small_aux(bool computer_player) { return computer_player ? 18 : 0; }
// launch_step_players: the human-vs-human rollout on one wave keeps its generic kernel -- it runs at the write ceiling of
// its two tensors either way and its leaner PLAIN form measured 0.6-3.4 % slower (profiles/r04_experiments/ab_rollout_hh_*)
constexpr bool kHhRolloutGeneric = true;
// launch_step: which human-vs-human k-frame launches run on two waves per 64 games: 1 = those on int16 rows (2.47 -> 2.23 us
// per frame), 2 = all (3.62 vs 3.63 on one wave: no gain; profiles/r03_experiments/)
constexpr int kHhPairRollout = 1;
// pair_body: what the human player's wave of a one-computer launch stores in front of the exchange barrier: 2 = its own
// player's columns and the ball's position / trail / rotation (8.62 -> 8.46 us per launch behind the LDS hand-shake,
// profiles/r05_experiments/ab_early_store_edge_*.log); 0 = nothing
constexpr int kEarlyOwnStores = diag::kNoEarlyStores ? 0 : 2;


// One game's column accessor: wave-uniform descriptor + column pitch, per-lane byte offset.
struct StateIO {
    Rsrc rsrc;
    uint32_t pitch;  // bytes between columns = stride * 4 (uniform)
    uint32_t voff;   // this lane's byte offset inside a column = lane index * 4
    __device__ __forceinline__ int ld(int col) const
    {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)col * pitch, 0);
    }
    __device__ __forceinline__ void st(int col, int v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32((unsigned int)v, rsrc, voff, (uint32_t)col * pitch, kStateAux);
    }
};

// The packed format (pz_packed.hpp): group A, group B and the tail are three columns, each behind its own descriptor
// -- the 16-byte stores then carry no SGPR offset (see flush_rows for why that matters).
struct PackedIO {
    Rsrc a, b, tail;
    uint32_t v16, v4;  // this lane's byte offset inside a 16-byte / 4-byte column
    __device__ __forceinline__ pk_u32x4 ld_a() const { return __builtin_amdgcn_raw_buffer_load_b128(a, v16, 0, 0); }
    __device__ __forceinline__ pk_u32x4 ld_b() const { return __builtin_amdgcn_raw_buffer_load_b128(b, v16, 0, 0); }
    __device__ __forceinline__ uint32_t ld_tail() const { return __builtin_amdgcn_raw_buffer_load_b32(tail, v4, 0, 0); }
    __device__ __forceinline__ void st_a(pk_u32x4 w) const { __builtin_amdgcn_raw_buffer_store_b128(w, a, v16, 0, kStateAux); }
    __device__ __forceinline__ void st_b(pk_u32x4 w) const { __builtin_amdgcn_raw_buffer_store_b128(w, b, v16, 0, kStateAux); }
    __device__ __forceinline__ void st_tail(uint32_t w) const { __builtin_amdgcn_raw_buffer_store_b32(w, tail, v4, 0, kStateAux); }
    __device__ __forceinline__ void st_ex(int ex) const
    {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ex, tail, v4, 0, kStateAux);
    }
    __device__ __forceinline__ void st_bold(int player, int bold) const  // player 0 / 1
    {
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bold, tail, v4 + 2u + (uint32_t)player, 0, kStateAux);
    }
};

__device__ __forceinline__ PackedIO make_packed_io(const void* packed, int64_t stride, int64_t lane_index)
{
    const char* base = static_cast<const char*>(packed);
    const uint32_t group = (uint32_t)stride * 16u;
    return PackedIO{make_rsrc(base, group), make_rsrc(base + stride * 16, group), make_rsrc(base + stride * 32, (uint32_t)stride * 4u),
                    (uint32_t)lane_index * 16u, (uint32_t)lane_index * 4u};
}

// What the first loads of a wave depend on.  The step kernels take these six as leading scalar kernel
// arguments (and everything, again, in StepArgs): the build preloads the first 11 argument dwords into
// SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=11), so the state loads do not have to wait
// for a scalar load from the kernarg segment first (pair kernel: 7.51 -> 7.33 us per launch).  The action
// vectors' element type is one of them: the wave branches on it in front of its very first loads.
struct HotArgs {
    int32_t* state;
    int64_t n, stride;
    const void* act_p1;  // elements of act_format
    const void* act_p2;
    int32_t act_format;  // = cfg.action_format (enum pz_action_format)
};
#define PZ_HOT_PARAMS int32_t *state, int64_t n, int64_t stride, const void *act_p1, const void *act_p2, int32_t act_format
#define PZ_HOT_ARGS(a) (a).state, (a).n, (a).stride, (a).act_p1, (a).act_p2, (a).cfg.action_format

struct StepArgs {
    int32_t* state;
    int64_t n, stride;
    const void* act_p1;  // nullptr => on-device random policy; pz_step_many: the int32 tape
    const void* act_p2;
    uint64_t action_seed, t0;
    int32_t k;  // frames per launch (random policy only)
    int32_t* act_out;  // trajectory mode only: int32[k][2][n] actions taken (may be nullptr)
    int32_t* obs_p1;
    int32_t* obs_p2;
    void* rew_p1;
    void* rew_p2;
    uint8_t* terminated;
    void* episode_stats;  // double[2][stride] returns + int32[stride] lengths (nullptr = off)
    unsigned long long* episodes_done;
    pz_flight_tables tables;  // device pointers or NULL (pz_flight_tables in the header)
    pz_config cfg;
};

__device__ __forceinline__ FlightLut make_lut(const pz_flight_tables& t)
{
    FlightLut lut;
    lut.has_landing = t.landing != nullptr;
    lut.has_power_hit = t.power_hit != nullptr;
    lut.landing = make_rsrc(t.landing, lut.has_landing ? (uint32_t)kFtLandingBytes : 0u);
    lut.power_hit = make_rsrc(t.power_hit, lut.has_power_hit ? (uint32_t)(kFtHitEntries * 16) : 0u);
    return lut;
}

// ---- state columns <-> registers -----------------------------------------------------------
__device__ __forceinline__ void load_player(Player& p, const StateIO& io, int c0)
{
    p.x = io.ld(c0 + PZ_P_X);
    p.y = io.ld(c0 + PZ_P_Y);
    p.yv = io.ld(c0 + PZ_P_Y_VELOCITY);
    p.state = io.ld(c0 + PZ_P_STATE);
    p.frame = io.ld(c0 + PZ_P_FRAME_NUMBER);
    p.arm = io.ld(c0 + PZ_P_ARM_SWING_DIRECTION);
    p.delay = io.ld(c0 + PZ_P_DELAY_BEFORE_NEXT_FRAME);
    p.dive = io.ld(c0 + PZ_P_DIVING_DIRECTION);
    p.lying = io.ld(c0 + PZ_P_LYING_DOWN_DURATION_LEFT);
    p.coll = io.ld(c0 + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED);
    p.bold = io.ld(c0 + PZ_P_COMPUTER_BOLDNESS);
    p.standby = io.ld(c0 + PZ_P_COMPUTER_WHERE_TO_STAND_BY);
    p.hitprev = io.ld(c0 + PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS);
}

__device__ __forceinline__ void load_game(Game& g, const StateIO& io)
{
    // env + ball first: the frame starts with the round bookkeeping and the ball-world step
    g.e.round_ended = io.ld(PZ_E_ROUND_ENDED);
    g.e.game_ended = io.ld(PZ_E_GAME_ENDED);
    g.e.rng = (uint32_t)io.ld(PZ_E_RNG_DRAW_COUNTER);
    g.e.s1 = io.ld(PZ_E_SCORE_P1);
    g.e.s2 = io.ld(PZ_E_SCORE_P2);
    g.e.p2serve = io.ld(PZ_E_IS_PLAYER2_SERVE);
    g.b.x = io.ld(PZ_B_X);
    g.b.y = io.ld(PZ_B_Y);
    g.b.xv = io.ld(PZ_B_X_VELOCITY);
    g.b.yv = io.ld(PZ_B_Y_VELOCITY);
    g.b.power = io.ld(PZ_B_IS_POWER_HIT);
    g.b.px = io.ld(PZ_B_PREVIOUS_X);
    g.b.py = io.ld(PZ_B_PREVIOUS_Y);
    g.b.ppx = io.ld(PZ_B_PREVIOUS_PREVIOUS_X);
    g.b.ppy = io.ld(PZ_B_PREVIOUS_PREVIOUS_Y);
    g.b.rot = io.ld(PZ_B_FINE_ROTATION);
    g.b.ex = io.ld(PZ_B_EXPECTED_LANDING_POINT_X);
    g.b.punch = io.ld(PZ_B_PUNCH_EFFECT_X);
    load_player(g.p1, io, 0);
    load_player(g.p2, io, PZ_P_WORDS);
}

__device__ __forceinline__ void store_player(const Player& p, const StateIO& io, int c0)
{
    io.st(c0 + PZ_P_X, p.x);
    io.st(c0 + PZ_P_Y, p.y);
    io.st(c0 + PZ_P_Y_VELOCITY, p.yv);
    io.st(c0 + PZ_P_STATE, p.state);
    io.st(c0 + PZ_P_FRAME_NUMBER, p.frame);
    io.st(c0 + PZ_P_ARM_SWING_DIRECTION, p.arm);
    io.st(c0 + PZ_P_DELAY_BEFORE_NEXT_FRAME, p.delay);
    io.st(c0 + PZ_P_DIVING_DIRECTION, p.dive);
    io.st(c0 + PZ_P_LYING_DOWN_DURATION_LEFT, p.lying);
    io.st(c0 + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED, p.coll);
    io.st(c0 + PZ_P_COMPUTER_BOLDNESS, p.bold);
    io.st(c0 + PZ_P_COMPUTER_WHERE_TO_STAND_BY, p.standby);
    io.st(c0 + PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS, p.hitprev);
}

// skip_ex: the lane's expected_landing_point_x is stored by the scout wave (step_games<..., SCOUT>)
__device__ __forceinline__ void store_game(const Game& g, const StateIO& io, bool skip_ex = false)
{
    store_player(g.p1, io, 0);
    store_player(g.p2, io, PZ_P_WORDS);
    io.st(PZ_B_X, g.b.x);
    io.st(PZ_B_Y, g.b.y);
    io.st(PZ_B_X_VELOCITY, g.b.xv);
    io.st(PZ_B_Y_VELOCITY, g.b.yv);
    io.st(PZ_B_IS_POWER_HIT, g.b.power);
    io.st(PZ_B_PREVIOUS_X, g.b.px);
    io.st(PZ_B_PREVIOUS_Y, g.b.py);
    io.st(PZ_B_PREVIOUS_PREVIOUS_X, g.b.ppx);
    io.st(PZ_B_PREVIOUS_PREVIOUS_Y, g.b.ppy);
    io.st(PZ_B_FINE_ROTATION, g.b.rot);
    if (!skip_ex) io.st(PZ_B_EXPECTED_LANDING_POINT_X, g.b.ex);
    io.st(PZ_B_PUNCH_EFFECT_X, g.b.punch);
    io.st(PZ_E_SCORE_P1, g.e.s1);
    io.st(PZ_E_SCORE_P2, g.e.s2);
    io.st(PZ_E_IS_PLAYER2_SERVE, g.e.p2serve);
    io.st(PZ_E_ROUND_ENDED, g.e.round_ended);
    io.st(PZ_E_GAME_ENDED, g.e.game_ended);
    io.st(PZ_E_RNG_DRAW_COUNTER, (int32_t)g.e.rng);
}

// Changed-only write-back, used by every launch that writes the state back after ONE frame.  Columns
// that change on nearly every frame are stored unconditionally; the 24 columns that change rarely
// (scores, flags, boldness, diving/lying state, collision debounce, ball x velocity ...: on average
// 80 % of their 32-byte sectors are untouched by a frame of random play) are stored only by the lanes
// whose value changed, and not at all when no lane of the wave changed.  Measured: -10 % per launch at
// 524 288 games, -5 % at 65 536 in the pair kernel (7.91 -> 7.48 us).
__device__ __forceinline__ void store_player_changed(const Player& p, const Player& o, const StateIO& io, int c0)
{
    io.st(c0 + PZ_P_X, p.x);
    io.st(c0 + PZ_P_Y, p.y);
    io.st(c0 + PZ_P_Y_VELOCITY, p.yv);
    io.st(c0 + PZ_P_FRAME_NUMBER, p.frame);
    io.st(c0 + PZ_P_DELAY_BEFORE_NEXT_FRAME, p.delay);
    io.st(c0 + PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS, p.hitprev);
    if (p.state != o.state) io.st(c0 + PZ_P_STATE, p.state);
    if (p.arm != o.arm) io.st(c0 + PZ_P_ARM_SWING_DIRECTION, p.arm);
    if (p.dive != o.dive) io.st(c0 + PZ_P_DIVING_DIRECTION, p.dive);
    if (p.lying != o.lying) io.st(c0 + PZ_P_LYING_DOWN_DURATION_LEFT, p.lying);
    if (p.coll != o.coll) io.st(c0 + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED, p.coll);
    if (p.bold != o.bold) io.st(c0 + PZ_P_COMPUTER_BOLDNESS, p.bold);
    if (p.standby != o.standby) io.st(c0 + PZ_P_COMPUTER_WHERE_TO_STAND_BY, p.standby);
}

__device__ __forceinline__ void store_game_changed(const Game& g, const Game& o, const StateIO& io, bool skip_ex = false)
{
    store_player_changed(g.p1, o.p1, io, 0);
    store_player_changed(g.p2, o.p2, io, PZ_P_WORDS);
    io.st(PZ_B_X, g.b.x);
    io.st(PZ_B_Y, g.b.y);
    io.st(PZ_B_Y_VELOCITY, g.b.yv);
    io.st(PZ_B_PREVIOUS_X, g.b.px);
    io.st(PZ_B_PREVIOUS_Y, g.b.py);
    io.st(PZ_B_PREVIOUS_PREVIOUS_X, g.b.ppx);
    io.st(PZ_B_PREVIOUS_PREVIOUS_Y, g.b.ppy);
    io.st(PZ_B_FINE_ROTATION, g.b.rot);
    if (g.b.xv != o.b.xv) io.st(PZ_B_X_VELOCITY, g.b.xv);
    if (g.b.power != o.b.power) io.st(PZ_B_IS_POWER_HIT, g.b.power);
    if (g.b.ex != o.b.ex && !skip_ex) io.st(PZ_B_EXPECTED_LANDING_POINT_X, g.b.ex);
    if (g.b.punch != o.b.punch) io.st(PZ_B_PUNCH_EFFECT_X, g.b.punch);
    if (g.e.s1 != o.e.s1) io.st(PZ_E_SCORE_P1, g.e.s1);
    if (g.e.s2 != o.e.s2) io.st(PZ_E_SCORE_P2, g.e.s2);
    if (g.e.p2serve != o.e.p2serve) io.st(PZ_E_IS_PLAYER2_SERVE, g.e.p2serve);
    if (g.e.round_ended != o.e.round_ended) io.st(PZ_E_ROUND_ENDED, g.e.round_ended);
    if (g.e.game_ended != o.e.game_ended) io.st(PZ_E_GAME_ENDED, g.e.game_ended);
    if (g.e.rng != o.e.rng) io.st(PZ_E_RNG_DRAW_COUNTER, (int32_t)g.e.rng);
}

// whole games in either format (constructor / reset / observe kernels, the single-wave step kernel)
struct PackedWords {
    pk_u32x4 a, b;
    uint32_t tail;
};

__device__ __forceinline__ PackedWords load_game_packed(Game& g, const PackedIO& pio, bool with_tail)
{
    PackedWords w{pio.ld_a(), pio.ld_b(), with_tail ? pio.ld_tail() : 0u};
    unpack_group_a(g, w.a);
    unpack_group_b(g, w.b);
    unpack_tail(g, w.tail);
    return w;
}

// `was`: what the lane loaded (its sticky overflow flags are kept; the tail is stored only when it changed)
__device__ __forceinline__ void store_game_packed(const Game& g, const PackedIO& pio, const PackedWords& was, bool with_tail)
{
    pio.st_a(pack_group_a(g, was.a.y & kPackedOverflowBit));
    pio.st_b(pack_group_b(g, was.b.y & kPackedOverflowBit));
    const uint32_t tail = pack_tail(g);
    if (with_tail && tail != was.tail) pio.st_tail(tail);
}

// Out-of-range actions.  The reference's table lookup raises IndexError on one (pikazoo_env.py:182); a launch cannot
// raise, so it counts into cfg.action_faults (NULL: unchecked, the range is the caller's contract).  `bad`: this lane
// read an action outside [0, n_actions).  One scalar test and, when some lane of the wave has one, a ballot per launch;
// the atomics only ever run on a caller's bug.  Nothing indexes memory with an action (the decode shifts bit tables, the
// tape is parked as bytes), so the damage is that game's input for the frame.
__device__ __forceinline__ uint32_t action_count(const pz_config& cfg) { return cfg.simplify_action ? 13u : 18u; }

// Both agents' actions of game i, loaded as the caller holds them (pz_config.action_format: torch's default integer dtype
// is int64 -- argmax, multinomial, Categorical.sample, randint -- and a cast kernel per agent in front of a 7 us launch
// costs more than the launch's own loads; it also WRAPS, 2**32 + 3 -> 3, before any range check sees the value).  One
// wave-uniform branch on the format, both loads inside it: nothing waits where the branch closes, the loads are in
// flight under the state loads like the int32 ones.  `high`: the OR of what the 32-bit action words do not hold -- an
// int64 element's upper dword; 0 in the other formats, whose widening (zero-extended uint8, sign-extended int16) keeps
// an out-of-range value out of range.  Valid iff high == 0 and (uint32) action < n_actions.  Rows past n read as 0.
typedef unsigned int act_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_actions(const void* p1, const void* p2, uint32_t n32, uint32_t i, int format, int& a1,
                                             int& a2, uint32_t& high)
{
    high = 0u;
    if (format == PZ_ACT_I64) {
        const act_u32x2 w1 = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(p1, n32 * 8u), i * 8u, 0, 0);
        const act_u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(p2, n32 * 8u), i * 8u, 0, 0);
        a1 = (int)w1.x;
        a2 = (int)w2.x;
        high = w1.y | w2.y;
    } else if (format == PZ_ACT_U8) {
        a1 = (int)__builtin_amdgcn_raw_buffer_load_b8(make_rsrc(p1, n32), i, 0, 0);
        a2 = (int)__builtin_amdgcn_raw_buffer_load_b8(make_rsrc(p2, n32), i, 0, 0);
    } else if (format == PZ_ACT_I16) {
        a1 = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(make_rsrc(p1, n32 * 2u), i * 2u, 0, 0);
        a2 = (int)(short)__builtin_amdgcn_raw_buffer_load_b16(make_rsrc(p2, n32 * 2u), i * 2u, 0, 0);
    } else {
        a1 = (int)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(p1, n32 * 4u), i * 4u, 0, 0);
        a2 = (int)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(p2, n32 * 4u), i * 4u, 0, 0);
    }
}
__device__ __forceinline__ bool actions_out_of_range(const pz_config& cfg, int a1, int a2, uint32_t high)
{
    return (high != 0u) | ((uint32_t)a1 >= action_count(cfg)) | ((uint32_t)a2 >= action_count(cfg));
}

__device__ __forceinline__ void count_action_faults(const pz_config& cfg, bool bad)
{
    if (cfg.action_faults == nullptr) return;                  // wave-uniform (a kernel argument)
    if (__builtin_amdgcn_ballot_w64(bad) == 0ull) return;      // wave-uniform
    if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(cfg.action_faults), 1ull);
}

__device__ __forceinline__ RngId make_rng_id(const pz_config& cfg, int64_t lane_index)
{
    const uint64_t gid = (uint64_t)(cfg.env_id_base + lane_index);
    return RngId{(uint32_t)gid, (uint32_t)(gid >> 32), make_rolling_key(cfg.seed)};
}

// ---- observation pack: _get_obs (pikazoo_env.py:576-624) ------------------------------------
// Row layout: player(13) | opponent(13) | ball(9).  Rows go to LDS at stride 35 words (odd,
// so the 64 lanes of a ds_write_b32 hit 32 distinct banks twice = conflict-free), then the
// wave copies the contiguous 64x35-word span to HBM with 16-byte lanes.
// NORM: NormalizeObservation fused (normalize_observation.py:22,30): every entry becomes
// float32 (v - low) / (high - low) with the bounds of pikazoo_env.py:485-562.  The reference
// divides in float64; for these small integers the correctly rounded float32 quotient is the
// float32 rounding of that double, so an IEEE float division reproduces it bit for bit.
template <bool NORM>
__device__ __forceinline__ int32_t obs_word(int v, int low, int range)
{
    if (!NORM) return v;
    const float f = (range == 1) ? (float)(v - low) : (float)(v - low) / (float)range;
    return (int32_t)__float_as_uint(f);
}

template <bool NORM>
__device__ __forceinline__ void player_words(const Player& p, int32_t (&w)[13])
{
    w[0] = obs_word<NORM>(p.x, 32, 368);
    w[1] = obs_word<NORM>(p.y, 108, 136);
    w[2] = obs_word<NORM>(p.yv, -15, 31);
    w[3] = obs_word<NORM>(p.dive, -1, 2);
    w[4] = obs_word<NORM>(p.lying, -2, 5);
    w[5] = obs_word<NORM>(p.frame, 0, 4);
    w[6] = obs_word<NORM>(p.delay, 0, 4);
    w[7] = obs_word<NORM>(p.state == 0, 0, 1);
    w[8] = obs_word<NORM>(p.state == 1, 0, 1);
    w[9] = obs_word<NORM>(p.state == 2, 0, 1);
    w[10] = obs_word<NORM>(p.state == 3, 0, 1);
    w[11] = obs_word<NORM>(p.state == 4, 0, 1);
    w[12] = obs_word<NORM>(p.hitprev, 0, 1);
}

template <bool NORM>
__device__ __forceinline__ void stage_obs_t(const Game& g, int32_t* __restrict__ s1, int32_t* __restrict__ s2, int lane)
{
    int32_t p1[13], p2[13], bw[9];
    player_words<NORM>(g.p1, p1);
    player_words<NORM>(g.p2, p2);
    bw[0] = obs_word<NORM>(g.b.x, 20, 412);
    bw[1] = obs_word<NORM>(g.b.y, 0, 252);
    bw[2] = obs_word<NORM>(g.b.px, 0, 432);
    bw[3] = obs_word<NORM>(g.b.py, 0, 252);
    bw[4] = obs_word<NORM>(g.b.ppx, 0, 432);
    bw[5] = obs_word<NORM>(g.b.ppy, 0, 252);
    bw[6] = obs_word<NORM>(g.b.xv, -20, 40);
    bw[7] = obs_word<NORM>(g.b.yv, -124, 248);
    bw[8] = obs_word<NORM>(g.b.power, 0, 1);
    int32_t* r1 = s1 + lane * PZ_OBS_DIM;
    int32_t* r2 = s2 + lane * PZ_OBS_DIM;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        r1[k] = p1[k];
        r1[13 + k] = p2[k];
        r2[k] = p2[k];
        r2[13 + k] = p1[k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        r1[26 + k] = bw[k];
        r2[26 + k] = bw[k];
    }
}

// one agent's row: [own player | opponent | ball] (pair kernel: each wave packs its own agent's tensor)
template <bool NORM>
__device__ __forceinline__ void stage_one_obs_t(const Player& me, const Player& opp, const Ball& b,
                                                int32_t* __restrict__ dst, int lane)
{
    int32_t pa[13], pb[13];
    player_words<NORM>(me, pa);
    player_words<NORM>(opp, pb);
    int32_t* r = dst + lane * PZ_OBS_DIM;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        r[k] = pa[k];
        r[13 + k] = pb[k];
    }
    r[26] = obs_word<NORM>(b.x, 20, 412);
    r[27] = obs_word<NORM>(b.y, 0, 252);
    r[28] = obs_word<NORM>(b.px, 0, 432);
    r[29] = obs_word<NORM>(b.py, 0, 252);
    r[30] = obs_word<NORM>(b.ppx, 0, 432);
    r[31] = obs_word<NORM>(b.ppy, 0, 252);
    r[32] = obs_word<NORM>(b.xv, -20, 40);
    r[33] = obs_word<NORM>(b.yv, -124, 248);
    r[34] = obs_word<NORM>(b.power, 0, 1);
}

__device__ __forceinline__ void stage_obs(const Game& g, int32_t* __restrict__ s1, int32_t* __restrict__ s2, int lane,
                                          bool normalize)
{
    if (normalize)  // wave-uniform (a config scalar)
        stage_obs_t<true>(g, s1, s2, lane);
    else
        stage_obs_t<false>(g, s1, s2, lane);
}


// cfg.normalize_obs == 2: int16 observations.  The rows are staged as int32 like always and narrowed on the way out:
// a lane reads eight staged values (two 16-byte LDS reads) and writes them as one 16-byte piece of the wave's
// contiguous 64 x 70-byte span -- 5 passes instead of 9, half the bytes.  `tensor_bytes` covers an EVEN number of rows
// (the caller pads an odd batch by one row): the last value of an odd row count shares its dword with the padding.
constexpr uint32_t kWaveObsBytes16 = kLanes * PZ_OBS_DIM * 2;  // 4 480
constexpr int kWaveObsVecs16 = (int)(kWaveObsBytes16 / 16);   // 280
__device__ __forceinline__ void flush_rows16(const int32_t* __restrict__ lds, const void* tensor, uint32_t tensor_bytes,
                                             int lane)
{
    const uint32_t wave_off = blockIdx.x * kWaveObsBytes16;
    const Rsrc span = make_rsrc(static_cast<const char*>(tensor) + wave_off, tensor_bytes - wave_off);
    const u32x4* src4 = reinterpret_cast<const u32x4*>(lds);
#pragma unroll
    for (int pass = 0; pass < (kWaveObsVecs16 + kLanes - 1) / kLanes; ++pass) {
        const int v = pass * kLanes + lane;
        if (v < kWaveObsVecs16) {
            const u32x4 lo = src4[2 * v], hi = src4[2 * v + 1];
            const u32x4 out = {(lo.x & 0xFFFFu) | (lo.y << 16), (lo.z & 0xFFFFu) | (lo.w << 16),
                               (hi.x & 0xFFFFu) | (hi.y << 16), (hi.z & 0xFFFFu) | (hi.w << 16)};
            __builtin_amdgcn_raw_buffer_store_b128(out, span, (uint32_t)v * 16u, 0, kObsAux);
        }
    }
}

// one agent's staged rows to frame `t` of its observation tensor, in the format cfg.normalize_obs names
// (0: int32, 1: float32 bit patterns -- both 140-byte rows --, 2: int16, 70-byte rows of an even row count)
__device__ __forceinline__ void flush_obs(const int32_t* __restrict__ lds, void* tensor, int64_t t, int64_t n, int format,
                                          int lane)
{
    if (format == 2) {
        const int64_t rows = (n + 1) & ~(int64_t)1;
        flush_rows16(lds, static_cast<char*>(tensor) + t * rows * (PZ_OBS_DIM * 2), (uint32_t)(rows * (PZ_OBS_DIM * 2)), lane);
    } else {
        flush_rows(lds, static_cast<char*>(tensor) + t * n * kRowBytes, (uint32_t)n * kRowBytes, lane);
    }
}

// PLAIN launches (k-frame kernels): no fused wrapper, no episode statistics, raw integer rows.  The configuration words
// of those features then read as compile-time constants, and their branches -- with the scalars and lane masks the
// compiler would otherwise carry around the frame loop for them (the reward table alone is eight SGPRs of a budget of
// ~100) -- leave the kernel.  The host picks the PLAIN instantiation when the configuration allows (launch_step).
template <bool PLAIN, bool OBS16>
__device__ __forceinline__ StepArgs effective_args(const StepArgs& in)
{
    StepArgs a = in;
    if (PLAIN) {
        a.cfg.simplify_action = 0;
        a.cfg.ballpos_reward = 0;
        a.cfg.normal_state_mode = 0;
        a.cfg.episode_stats_mode = 0;
        a.cfg.normalize_obs = OBS16 ? 2 : 0;
        a.episode_stats = nullptr;
    }
    return a;
}

// ---- the fused step kernel -------------------------------------------------------------------
// AI1/AI2: player 1 / 2 is the rule-based computer (compile-time so the human-vs-human build
// carries none of the predictor code or its registers).
// MODE: kActions  -- one frame, actions read from HBM (pz_step);
//       kRandom   -- k frames of the on-device random policy, outputs of the last frame (pz_step_random);
//       kRollout  -- k frames of the random policy, EVERY frame's outputs written to [k][n]...
//                    trajectory tensors (pz_rollout_random); the state stays in registers for
//                    the whole launch, so per frame only the outputs move;
//       kTape     -- the same with the actions of every frame read from an int32[k][2][n] tape
//                    (pz_step_many), parked in LDS one byte per action, 64 frames per fetch.
// SPARSE: changed-only write-back of the rarely changing columns (large batches).
enum StepMode { kActions = 0, kRandom = 1, kRollout = 2, kTape = 3 };
// pz_step_many parks its action tape in LDS, ONE BYTE per action (an action is < 18; an out-of-range one was counted
// while it was parked, include/pikazoo_hip.h): kTapeChunk frames of both players for a wave's 64 games are 8 KB.
constexpr int kTapeChunk = 64;                       // frames of the action tape fetched at once by pz_step_many (16: 3.88 vs 3.27 us per frame)
constexpr int kTapeWords = kTapeChunk * 2 * kLanes / 4;  // the parked chunk in int32 words of LDS
constexpr int kTapeAux = 18;  // sc1 nt: a cold tape streamed with the default policy pushes the flight tables' hot lines out of the caches
constexpr int kTapeBatch = 16;                        // tape rows requested together by a refill inside the frame loop (= kTapeGroup)

// Rows [s0 + f0, s0 + f0 + B) of the tape for the players asked for: ALL the loads are in flight before the first one is
// waited for -- one memory round trip per batch (a refill that waits for each row, or each handful of rows, pays the
// latency of a cold tape under the launch's write stream several times over: 3.88 vs 3.27 us per frame with a
// cache-resident tape).  kTapeGroup rows share one descriptor, which ends with the launch's last frame: a row is
// addressed through the per-lane offset (the part of a buffer address the range check covers), so rows past k read as 0
// without a memory access -- and a batch needs two descriptors instead of one per row (32 of them, 128 scalars, were
// most of what the tape kernels spilled).  A lane past the end of the batch reads some other game's action: it is never
// used.  Returns whether this lane read an action >= n_actions (count_action_faults).
constexpr int kTapeGroup = 16;  // kTapeGroup rows of 8 n bytes stay below 2^32 for every batch one launch can take
template <int B, bool P1, bool P2>
__device__ __forceinline__ bool park_tape_rows(const int32_t* tape0, int64_t n, uint32_t n32, int32_t k, int32_t s0, int f0,
                                               uint32_t voff, unsigned char* __restrict__ parked, int lane, uint32_t n_actions)
{
    static_assert(B % kTapeGroup == 0, "whole groups");
    int32_t v1[B], v2[B];
#pragma unroll
    for (int h = 0; h < B; h += kTapeGroup) {
        const int32_t first = s0 + f0 + h;
        const int32_t left = min(max(k - first, 0), kTapeGroup);  // wave-uniform
        const Rsrc rows = make_rsrc(tape0 + (int64_t)(left > 0 ? first : 0) * 2 * n, (uint32_t)left * n32 * 8u);
        uint32_t o1 = voff, o2 = voff + n32 * 4u;
#pragma unroll
        for (int j = 0; j < kTapeGroup; ++j) {
            v1[h + j] = P1 ? (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, o1, 0, kTapeAux) : 0;
            v2[h + j] = P2 ? (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rows, o2, 0, kTapeAux) : 0;
            o1 += n32 * 8u;
            o2 += n32 * 8u;
        }
    }
    uint32_t worst = 0u;
#pragma unroll
    for (int j = 0; j < B; ++j) {
        if (P1) parked[((f0 + j) * 2 + 0) * kLanes + lane] = (unsigned char)v1[j];
        if (P2) parked[((f0 + j) * 2 + 1) * kLanes + lane] = (unsigned char)v2[j];
        worst = max(worst, max((uint32_t)v1[j], (uint32_t)v2[j]));
    }
    return worst >= n_actions;
}

// One chunk (frames s0 .. s0 + kTapeChunk - 1, as far as the launch goes).  FIRST: the launch's first chunk, requested
// behind the state loads with 32 rows in flight per batch (nothing else is live yet); a refill inside the frame loop
// keeps to kTapeBatch registers per player.
template <bool FIRST, bool P1, bool P2>
__device__ __forceinline__ bool park_tape_chunk(const int32_t* tape0, int64_t n, uint32_t n32, int32_t k, int32_t s0,
                                                uint32_t voff, unsigned char* __restrict__ parked, int lane, uint32_t n_actions)
{
    constexpr int B = FIRST ? 32 : kTapeBatch;
    static_assert(kTapeChunk % B == 0, "whole batches");
    bool bad = false;
#pragma unroll 1
    for (int f0 = 0; f0 < kTapeChunk && s0 + f0 < k; f0 += B)
        bad |= park_tape_rows<B, P1, P2>(tape0, n, n32, k, s0, f0, voff, parked, lane, n_actions);
    return bad;
}

// The reward pipeline of one frame (see pz_config in the header): the reference's wrapper
// stack RewardInNormalState / RewardByBallPosition in either order, fused.
struct Rewards {
    int i1, i2;      // the env's own +1/-1/0 (pikazoo_env.py:217-228)
    float f1, f2;    // after the fused reward wrappers
};

// one entry of RewardByBallPosition's table in a per-lane register, through an opaque move.  Indexing the kernel
// argument with the (per-lane) zone is a global load from the kernarg segment -- and the compiler turns a select
// over the eight scalars back into exactly that load -- whose wait also drains every store issued before it (gfx9
// counts loads and stores in one in-order vmcnt): a dependent memory round trip in front of the reward store.
__device__ __forceinline__ float zone_reward(const pz_config& cfg, int k)
{
    float v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(cfg.additional_reward[k]));
    return v;
}

// The k-frame kernels hold the table in eight per-lane registers for the whole launch (`parked`): as scalars the
// compiler carries it around the frame loop in eight SGPRs of a budget of ~100, spills it and brings all eight back
// through v_readlane in front of every use (a lone wave per SIMD has hundreds of VGPRs to spare).
struct ZoneTable {
    float z[8];
};
__device__ __forceinline__ ZoneTable park_zone_table(const pz_config& cfg)
{
    ZoneTable t;
#pragma unroll
    for (int k = 0; k < 8; ++k) t.z[k] = zone_reward(cfg, k);
    return t;
}

__device__ __forceinline__ Rewards shape_rewards(const pz_config& cfg, const Game& g, int reward, bool frozen,
                                                 const ZoneTable* parked = nullptr)
{
    Rewards r{reward, -reward, (float)reward, (float)(-reward)};
    if (frozen) return r;
    if (cfg.normal_state_mode == 1) {  // reward_in_normal_state.py:12-14, inside RewardByBallPosition
        r.f1 = (r.f1 == 0.0f) ? cfg.normal_state_reward : r.f1;
        r.f2 = (r.f2 == 0.0f) ? cfg.normal_state_reward : r.f2;
    }
    if (cfg.ballpos_reward) {  // reward_by_ball_position.py:22-29: zone from the post-step ball position
        // additional_reward[i * 4 + zone], zone = (y > y_line) + 2 * (x >= x_line)
        const bool low = g.b.y > cfg.y_line, right = g.b.x >= cfg.x_line;
        auto z = [&](int k) { return parked != nullptr ? parked->z[k] : zone_reward(cfg, k); };
        r.f1 += right ? (low ? z(3) : z(2)) : (low ? z(1) : z(0));
        r.f2 += right ? (low ? z(7) : z(6)) : (low ? z(5) : z(4));
    }
    if (cfg.normal_state_mode == 2) {  // the wrapper outside RewardByBallPosition
        r.f1 = (r.f1 == 0.0f) ? cfg.normal_state_reward : r.f1;
        r.f2 = (r.f2 == 0.0f) ? cfg.normal_state_reward : r.f2;
    }
    return r;
}

// RecordEpisodeStatistics (record_episode_statistics.py:27-40): per game the two running returns as float64 -- the
// reference sums Python floats (:31), and so do we: float32 rewards are widened before the add, the env's own +-1/0
// sum exactly -- and the episode length.  Buffer layout (pz_step in the header): double[2][stride], int32[stride].
struct EpisodeStats {
    double r1, r2;
    int len;
};
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct StatsIO {
    Rsrc rsrc;
    uint32_t stride;  // games per row
    uint32_t lane;    // this lane's game index inside the batch
    __device__ __forceinline__ double ld_ret(int agent) const
    {
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8u, (uint32_t)agent * stride * 8u, 0);
        return __hiloint2double((int)w.y, (int)w.x);
    }
    __device__ __forceinline__ void st_ret(int agent, double v) const
    {
        const u32x2 w = {(unsigned int)__double2loint(v), (unsigned int)__double2hiint(v)};
        __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, lane * 8u, (uint32_t)agent * stride * 8u, 0);
    }
    __device__ __forceinline__ int ld_len() const
    {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, stride * 16u, 0);
    }
    __device__ __forceinline__ void st_len(int v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32((unsigned int)v, rsrc, lane * 4u, stride * 16u, 0);
    }
    __device__ __forceinline__ void load(EpisodeStats& st) const
    {
        st.r1 = ld_ret(0);
        st.r2 = ld_ret(1);
        st.len = ld_len();
    }
    __device__ __forceinline__ void store(const EpisodeStats& st) const
    {
        st_ret(0, st.r1);
        st_ret(1, st.r2);
        st_len(st.len);
    }
};

// `enabled` false: an empty descriptor (loads return 0, stores are dropped)
__device__ __forceinline__ StatsIO make_stats_io(const void* episode_stats, bool enabled, int64_t stride, int64_t lane)
{
    return StatsIO{make_rsrc(episode_stats, enabled ? (uint32_t)(stride * 20) : 0u), (uint32_t)stride, (uint32_t)lane};
}

__device__ __forceinline__ void stats_update(EpisodeStats& st, const pz_config& cfg, const Rewards& r, bool was_reset,
                                             bool counted, bool as_float)
{
    if (was_reset) st = EpisodeStats{0.0, 0.0, 0};  // reset() zeroes the sums (:23-25)
    if (!counted) return;
    const bool raw = cfg.episode_stats_mode == 1 || !as_float;
    st.r1 += raw ? (double)r.i1 : (double)r.f1;
    st.r2 += raw ? (double)r.i2 : (double)r.f2;
    st.len += 1;
}

// outputs of one frame: rewards, terminated (pikazoo_env.py:233) and the two observation
// tensors through the LDS transpose.  `t` = frame index inside a trajectory (0 otherwise).
__device__ __forceinline__ void emit_outputs(const StepArgs& a, const Game& g, const Rewards& r, bool as_float,
                                             bool live, int64_t i, int lane, int64_t t,
                                             int32_t (*lds_obs)[kLanes * PZ_OBS_DIM])
{
    const uint32_t n32 = (uint32_t)a.n;
    const uint32_t voff = (uint32_t)i * 4u;
    // descriptors are built from kernel arguments and the (uniform) frame index only
    const Rsrc rew1 = make_rsrc(static_cast<char*>(a.rew_p1) + t * a.n * 4, n32 * 4u);
    const Rsrc rew2 = make_rsrc(static_cast<char*>(a.rew_p2) + t * a.n * 4, n32 * 4u);
    const Rsrc term = make_rsrc(a.terminated + t * a.n, n32);
    if (live) {
        __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(r.f1) : (unsigned int)r.i1, rew1, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(r.f2) : (unsigned int)r.i2, rew2, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)g.e.game_ended, term, (uint32_t)i, 0, 0);
        PZ_STAMP(3);
        stage_obs(g, lds_obs[0], lds_obs[1], lane, a.cfg.normalize_obs == 1);
    }
    __syncthreads();
    PZ_STAMP(4);
    flush_obs(lds_obs[0], a.obs_p1, t, a.n, a.cfg.normalize_obs, lane);
    flush_obs(lds_obs[1], a.obs_p2, t, a.n, a.cfg.normalize_obs, lane);
}

// ---- outputs of one frame of a trajectory launch (kRollout / kTape) ------------------------------------------------
// A lone wave issues in order, so whatever it waits for it waits for alone.  The single-frame flush (flush_rows) lets
// the compiler pair every few LDS reads with their stores -- ten read -> wait -> store round trips per frame, a
// quarter of the wave's cycles in SQ_WAIT_ANY (profiles/r03a_pmc_summary.json).  Here a frame's rows leave in three
// steps: `stage` (rewards / flag / actions stored, rows written to LDS), then `flush`: all 18 16-byte pieces of both
// tensors requested from LDS at once (72 VGPRs), independent work of the caller (the NEXT frame's policy draw), the
// 18 stores back to back.  The slab pointers advance by one frame per frame; sizes are loop-invariant.
// N stores the range check drops (empty descriptor): place holders in the in-order vmcnt, see step_kernel's frame loop
template <int N>
__device__ __forceinline__ void issue_dropped_stores()
{
    const Rsrc nowhere = make_rsrc(nullptr, 0u);
#pragma unroll
    for (int k = 0; k < N; ++k)  // (distinct, non-adjacent offsets: the compiler merges identical and adjacent stores)
        __builtin_amdgcn_raw_buffer_store_b32(0u, nowhere, (uint32_t)k * 256u, 0, 0);
}

template <bool OBS16>
struct TrajOut {
    char *rew1, *rew2, *term, *act, *obs1, *obs2;  // this frame's slabs (obs: this wave's span of the slab)
    uint32_t n32, obs_frame_bytes, obs_span_bytes; // games; bytes of one frame of an observation tensor; of it from the span on
    uint32_t voff, ioff, piece_off[9];             // lane offsets: dword / byte outputs; the nine 16-byte pieces (~0u: none)

    __device__ __forceinline__ void init(const StepArgs& a, int64_t i, int lane, bool live)
    {
        n32 = (uint32_t)a.n;
        constexpr bool half = OBS16;  // int16 rows: 70 bytes, an even number of rows per frame
        const uint32_t rows = half ? (uint32_t)((a.n + 1) & ~(int64_t)1) : n32;
        const uint32_t wave_bytes = half ? kWaveObsBytes16 : kWaveObsBytes;
        obs_frame_bytes = rows * (half ? (uint32_t)(PZ_OBS_DIM * 2) : kRowBytes);
        const uint32_t wave_off = blockIdx.x * wave_bytes;
        obs_span_bytes = obs_frame_bytes - wave_off;
        rew1 = static_cast<char*>(a.rew_p1);
        rew2 = static_cast<char*>(a.rew_p2);
        term = reinterpret_cast<char*>(a.terminated);
        act = reinterpret_cast<char*>(a.act_out);
        obs1 = reinterpret_cast<char*>(a.obs_p1) + wave_off;
        obs2 = reinterpret_cast<char*>(a.obs_p2) + wave_off;
        voff = live ? (uint32_t)i * 4u : ~0u;  // (rows past the end of the batch: dropped by the range check)
        ioff = live ? (uint32_t)i : ~0u;
        const int vecs = half ? kWaveObsVecs16 : kWaveObsVecs;
#pragma unroll
        for (int pass = 0; pass < 9; ++pass) {
            const int v = pass * kLanes + lane;
            piece_off[pass] = v < vecs ? (uint32_t)v * 16u : ~0u;
        }
    }
    __device__ __forceinline__ void advance()
    {
        rew1 += n32 * 4u;
        rew2 += n32 * 4u;
        term += n32;
        act += n32 * 8u;
        obs1 += obs_frame_bytes;
        obs2 += obs_frame_bytes;
    }
    // rewards, flag, (rollout) the actions taken; both agents' rows into LDS
    template <int SMALL_AUX>
    __device__ __forceinline__ void stage(const StepArgs& a, const Game& g, const Rewards& r, bool as_float, bool live,
                                          int a1, int a2, bool with_actions, int lane,
                                          int32_t (*lds_obs)[kLanes * PZ_OBS_DIM])
    {
        if (with_actions) {
            const Rsrc ao = make_rsrc(act, n32 * 8u);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned int)a1, ao, voff, 0, SMALL_AUX);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned int)a2, ao, voff, n32 * 4u, SMALL_AUX);
        }
        __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(r.f1) : (unsigned int)r.i1,
                                              make_rsrc(rew1, n32 * 4u), voff, 0, SMALL_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(r.f2) : (unsigned int)r.i2,
                                              make_rsrc(rew2, n32 * 4u), voff, 0, SMALL_AUX);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)g.e.game_ended, make_rsrc(term, n32), ioff, 0, SMALL_AUX);
        if (live) stage_obs(g, lds_obs[0], lds_obs[1], lane, a.cfg.normalize_obs == 1);
        // the rows are read back by this wave only: its LDS instructions execute in issue order
        wave_lds_handover<false>();
    }
    // the pieces requested from LDS at once, independent work of the caller (`between`), then the stores (no SGPR
    // offset on the 16-byte stores: see flush_rows).  The row format is a compile-time parameter here.  As a
    // run-time branch it cost either way: with a store sequence per format the compiler sees a path through the frame
    // with no row store at all (the two branches are lowered through a flag it cannot correlate) and sizes the wait
    // for the computer player's loop-carried gathers for that path -- vmcnt(0), a full drain of the previous frame's
    // stores per frame; with one shared store sequence the pieces go through 72 register copies per frame.
    template <class Between>
    __device__ __forceinline__ void flush(const int32_t (*lds_obs)[kLanes * PZ_OBS_DIM], int lane, Between&& between)
    {
        const u32x4* src1 = reinterpret_cast<const u32x4*>(lds_obs[0]);
        const u32x4* src2 = reinterpret_cast<const u32x4*>(lds_obs[1]);
        if constexpr (OBS16) {
            flush_tensor(src1, obs1, lane, between);
            flush_tensor(src2, obs2, lane, [] {});
        } else {
            // both tensors' 18 pieces in flight from LDS together (72 VGPRs: the wave has the register file to itself;
            // tensor by tensor with the next frame's head under the second read batch was no faster,
            // profiles/r03_experiments/ab_rollout_p2_computer_head_orderings.log)
            u32x4 p1[9], p2[9];
#pragma unroll
            for (int pass = 0; pass < 9; ++pass) p1[pass] = src1[min(pass * kLanes + lane, kWaveObsVecs - 1)];
#pragma unroll
            for (int pass = 0; pass < 9; ++pass) p2[pass] = src2[min(pass * kLanes + lane, kWaveObsVecs - 1)];
            between();
            const Rsrc s1 = make_rsrc(obs1, obs_span_bytes), s2 = make_rsrc(obs2, obs_span_bytes);
            // (the two tensors' pieces alternating, or every other workgroup writing player 2's tensor first: the same)
#pragma unroll
            for (int pass = 0; pass < 9; ++pass)
                __builtin_amdgcn_raw_buffer_store_b128(p1[pass], s1, piece_off[pass], 0, kTrajAux);
#pragma unroll
            for (int pass = 0; pass < 9; ++pass)
                __builtin_amdgcn_raw_buffer_store_b128(p2[pass], s2, piece_off[pass], 0, kTrajAux);
        }
    }
    // one tensor's pieces requested from LDS at once (36 VGPRs), `between()`, their stores back to back
    template <class Between>
    __device__ __forceinline__ void flush_tensor(const u32x4* __restrict__ src, char* slab, int lane, Between&& between)
    {
        if constexpr (OBS16) {  // five pieces, each narrowed from two staged ones
            u32x4 lo[5], hi[5];
#pragma unroll
            for (int pass = 0; pass < 5; ++pass) {
                const int v = min(pass * kLanes + lane, kWaveObsVecs16 - 1);
                lo[pass] = src[2 * v];
                hi[pass] = src[2 * v + 1];
            }
            between();
            const Rsrc span = make_rsrc(slab, obs_span_bytes);
#pragma unroll
            for (int pass = 0; pass < 5; ++pass) {
                const u32x4 w = {(lo[pass].x & 0xFFFFu) | (lo[pass].y << 16), (lo[pass].z & 0xFFFFu) | (lo[pass].w << 16),
                                 (hi[pass].x & 0xFFFFu) | (hi[pass].y << 16), (hi[pass].z & 0xFFFFu) | (hi[pass].w << 16)};
                __builtin_amdgcn_raw_buffer_store_b128(w, span, piece_off[pass], 0, kTrajAux);
            }
        } else {
            u32x4 piece[9];
#pragma unroll
            for (int pass = 0; pass < 9; ++pass) piece[pass] = src[min(pass * kLanes + lane, kWaveObsVecs - 1)];
            between();
            const Rsrc span = make_rsrc(slab, obs_span_bytes);
#pragma unroll
            for (int pass = 0; pass < 9; ++pass)
                __builtin_amdgcn_raw_buffer_store_b128(piece[pass], span, piece_off[pass], 0, kTrajAux);
        }
    }
    static constexpr int kStores = OBS16 ? 10 : 18;  // row stores per frame
};

// The scout wave of step_kernel<..., SCOUT>: for its workgroup's 64 games it loads just what decides
// whether a computer player will scan the power-hit directions this frame (the ball, the computer
// players' x / y / state, the two round flags), advances the ball like the frame will, and evaluates the
// six candidate flights of every such game cooperatively -- while the main wave is still loading the
// 44 columns, starting rounds and predicting the landing point.  A game whose round (re)starts this
// frame never scans (its players are put back on the ground), so the scout skips it.
template <bool AI1, bool AI2>
__device__ __forceinline__ void scout_candidates(const HotArgs a, int32_t* __restrict__ cand,
                                                 int32_t* __restrict__ scratch, int lane)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const StateIO io{make_rsrc(a.state, (uint32_t)(a.stride * (PZ_STATE_WORDS * 4))), (uint32_t)a.stride * 4u,
                     (uint32_t)i * 4u};
    Ball b{};
    Player p1{}, p2{};
    bool need = false;
    if (i < a.n) {
        const int round_ended = io.ld(PZ_E_ROUND_ENDED);  // game_ended implies round_ended
        b.x = io.ld(PZ_B_X);
        b.y = io.ld(PZ_B_Y);
        b.xv = io.ld(PZ_B_X_VELOCITY);
        b.yv = io.ld(PZ_B_Y_VELOCITY);
        if (AI1) {
            p1.x = io.ld(PZ_P_X);
            p1.y = io.ld(PZ_P_Y);
            p1.state = io.ld(PZ_P_STATE);
        }
        if (AI2) {
            p2.x = io.ld(PZ_P_WORDS + PZ_P_X);
            p2.y = io.ld(PZ_P_WORDS + PZ_P_Y);
            p2.state = io.ld(PZ_P_WORDS + PZ_P_STATE);
        }
        if (!round_ended) {
            ball_world_step(b);
            need = (AI1 && power_hit_scan_needed(p1, b)) || (AI2 && power_hit_scan_needed(p2, b));
        }
    }
    int ex[6] = {0, 0, 0, 0, 0, 0};
    wave_power_hit_candidates<false>(need, b, ex, scratch, lane);
    if (need) {
#pragma unroll
        for (int c = 0; c < 6; ++c) cand[lane * kCandPitch + c] = ex[c];
    }
}

// The scout's second job: the landing point of the balls a player hit this frame (physics.py:331-332),
// posted by the main wave at the end of its frame; the scout stores the state column itself, beside the
// main wave's scoring, write-back and observation pack.
__device__ __forceinline__ void scout_landing_after_hits(const StepArgs& a, const int32_t* __restrict__ hits, int lane)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const StateIO io{make_rsrc(a.state, (uint32_t)(a.stride * (PZ_STATE_WORDS * 4))), (uint32_t)a.stride * 4u,
                     (uint32_t)i * 4u};
    const int32_t* slot = hits + lane * kHitPitch;
    if (slot[0] != 0)  // only ever set for lanes inside the batch
        io.st(PZ_B_EXPECTED_LANDING_POINT_X, predict_landing_x<true>(slot[1], slot[2], slot[3], slot[4]));
}

// The scout of the k-frame modes: one round per frame on what the main wave posted after its world step.
__device__ __forceinline__ void scout_candidates_posted(const int32_t* __restrict__ posts, int32_t* __restrict__ cand,
                                                        int32_t* __restrict__ scratch, int lane)
{
    const int32_t* post = posts + lane * kPostPitch;
    const bool need = post[0] != 0;
    Ball b{};
    if (need) {
        b.x = post[1];
        b.y = post[2];
        b.yv = post[3];
    }
    int ex[6] = {0, 0, 0, 0, 0, 0};
    wave_power_hit_candidates<false>(need, b, ex, scratch, lane);
    if (need) {
#pragma unroll
        for (int c = 0; c < 6; ++c) cand[lane * kCandPitch + c] = ex[c];
    }
}

// SCOUT (launches with a computer player, below kTwoWaveMaxLanes games): the workgroup has a
// second wave for the flight predictions that can run beside the frame -- kScoutLoads for the single
// frame of pz_step (scout_candidates, scout_landing_after_hits), kScoutPosted for the k-frame modes
// (scout_candidates_posted).  The scout executes exactly the workgroup barriers of the main wave.
// PACKED: the state buffer holds the packed format (pz_packed.hpp); the whole groups are written back (no scout wave).
// OBS16 (trajectory modes only): int16 observation rows (cfg.normalize_obs == 2), compile-time there.
// The compiler's occupancy target for the kernel (it steers its scheduling, not only its register budget; measured,
// tools/ab.py, us per frame at k = 32): the rollout of the on-device policy is at its best told that one wave per
// SIMD is all there will be (4.23 vs 4.27 with a computer player), the tape kernel -- same register count either way
// -- left alone (3.63 vs 3.96); with a scout wave two waves per SIMD must fit.
constexpr int kTrajWaves = 1, kTapeWaves = 8;
template <bool AI1, bool AI2, int MODE, bool SPARSE, int SCOUT = kNoScout, bool PACKED = false, bool OBS16 = false,
          bool PLAIN = false>
__global__ __launch_bounds__(SCOUT != kNoScout ? 2 * kLanes : kLanes)
__attribute__((amdgpu_waves_per_eu((MODE == kRollout || MODE == kTape) && SCOUT != kNoScout ? 2 : 1,
                                   MODE == kRollout ? (SCOUT != kNoScout ? 2 : kTrajWaves) : (MODE == kTape && SCOUT != kNoScout ? 2 : (MODE == kTape ? kTapeWaves : 8)))))
void step_kernel(PZ_HOT_PARAMS, const StepArgs args)
{
    const StepArgs a = effective_args<PLAIN, OBS16>(args);
    const HotArgs hot{state, n, stride, act_p1, act_p2, act_format};
    static_assert(!PACKED || (SCOUT == kNoScout && !SPARSE), "the packed format has no scout and no changed-only variant");
    static_assert(!OBS16 || MODE == kRollout || MODE == kTape, "the single-frame launches take the row format at run time");
    static_assert(SCOUT == kNoScout || ((AI1 || AI2) && (MODE == kActions) == (SCOUT == kScoutLoads)),
                  "kScoutLoads serves the single-frame AI launch, kScoutPosted the k-frame ones");
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    __shared__ int32_t tape_lds[MODE == kTape ? kTapeWords : 1];  // parked action tape (kTape only)
    __shared__ int32_t cand[SCOUT != kNoScout ? kLanes * kCandPitch : 1];
    __shared__ int32_t hits[SCOUT == kScoutLoads ? kLanes * kHitPitch : 1];
    __shared__ int32_t posts[SCOUT == kScoutPosted ? kLanes * kPostPitch : 1];
    __shared__ int32_t scout_scratch[SCOUT != kNoScout ? 576 : 1];

    const int lane = threadIdx.x & (kLanes - 1);
    if (SCOUT != kNoScout && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0) {
        if (SCOUT == kScoutLoads) {
            // the frame wave waits for this one more often than the other way round: where the two share a SIMD,
            // let the scout issue first (config 3: 14.3 -> 13.5 us; no gain in the per-frame hand-shake of the
            // k-frame modes, so only here)
            __builtin_amdgcn_s_setprio(1);
            scout_candidates<AI1, AI2>(hot, cand, scout_scratch, lane);
            __syncthreads();  // step_games: candidates handed over
            __syncthreads();  // step_games: collided balls posted
            scout_landing_after_hits(a, hits, lane);
        } else {
            for (int32_t s = 0; s < a.k; ++s) {
                __syncthreads();  // step_games: this frame's ball posted
                scout_candidates_posted(posts, cand, scout_scratch, lane);
                __syncthreads();  // step_games: candidates handed over
            }
        }
        if (MODE != kRollout && MODE != kTape) __syncthreads();  // emit_outputs (this wave stages no rows)
        return;
    }
    const int64_t base = (int64_t)blockIdx.x * kLanes;
    const int64_t i = base + lane;
    const bool live = i < hot.n;
    const uint32_t n32 = (uint32_t)hot.n;

    // descriptors are built from kernel arguments only, so they are provably wave-uniform
    const StateIO io{make_rsrc(hot.state, PACKED ? 0u : (uint32_t)(hot.stride * (PZ_STATE_WORDS * 4))),
                     (uint32_t)hot.stride * 4u, (uint32_t)i * 4u};
    const PackedIO pio = make_packed_io(hot.state, PACKED ? hot.stride : 0, i);
    const bool as_float = a.cfg.ballpos_reward != 0 || a.cfg.normal_state_mode != 0;
    const bool with_stats = a.episode_stats != nullptr && a.cfg.episode_stats_mode != 0;  // uniform
    const StatsIO sio = make_stats_io(a.episode_stats, with_stats, a.stride, i);

    Game g{};
    RngId id = make_rng_id(a.cfg, live ? i : 0);
    KeySchedule policy = make_rolling_key(a.action_seed);
    if (MODE != kActions) {  // the frame loop's two key schedules live in VGPRs (see KeySchedule)
        id.ks = make_parked_schedule(a.cfg.seed);
        if (MODE != kTape) policy = make_parked_schedule(a.action_seed);
    }
    const FlightLut lut = make_lut(a.tables);
    int reward = 0;
    bool frozen = false;
    bool ex_pending = false;  // SCOUT: the scout wave stores this lane's expected_landing_point_x
    unsigned int finished = 0;
    PZ_STAMP(0);
    int a1 = 0, a2 = 0;
    uint32_t act_high = 0u;
    if (MODE == kActions) load_actions(hot.act_p1, hot.act_p2, n32, (uint32_t)i, hot.act_format, a1, a2, act_high);
    EpisodeStats st{0.0, 0.0, 0};
    PackedWords was{};
    if (live) {
        if constexpr (PACKED)
            was = load_game_packed(g, pio, true);
        else
            load_game(g, io);
        if (with_stats) sio.load(st);
    }
    // pz_step_many: the tape is fetched kTapeChunk frames at a time and parked in LDS -- a per-frame global load would
    // put a full memory latency on every frame of a lone wave, and its wait (vmcnt is in-order) would also drain that
    // frame's stores; LDS reads only touch lgkmcnt.  The FIRST chunk is requested here, behind the state loads: its
    // latency is theirs.  (Requesting a chunk half a chunk ahead into registers was tried: the loads pending around the
    // loop's back edge make the compiler wait at every copy of those registers, every frame.)
    unsigned char* const parked = reinterpret_cast<unsigned char*>(tape_lds);
    bool bad_action = false;
    auto fetch_tape_chunk = [&](int32_t s0, auto first) {
        bad_action |= park_tape_chunk<decltype(first)::value, true, true>(static_cast<const int32_t*>(a.act_p1), a.n, n32, a.k, s0,
                                                                          io.voff, parked, lane, action_count(a.cfg));
        wave_lds_handover<SCOUT == kNoScout>();  // every lane reads back its own bytes only
    };
    if (MODE == kTape) fetch_tape_chunk(0, std::true_type{});
    const Game loaded = g;  // SPARSE: what the columns held before the frame
    PZ_DRAIN_VMEM();
    PZ_STAMP(1);
    // The frame runs in wave-uniform control flow (the computer player's power-hit candidates
    // are evaluated cooperatively by the wave); lanes past the end of the batch idle inside.
    // lds_obs[0] doubles as the cooperative scratch until the observations are staged.
    Rewards rw{0, 0, 0.0f, 0.0f};
    if (MODE != kActions) {
        const uint32_t n_actions = a.cfg.simplify_action ? 13u : 18u;
        constexpr bool kTraj = MODE == kRollout || MODE == kTape;
        // A human player's computer_boldness is drawn at every round start (physics.py:218) and read by nothing, so only
        // the LAST draw of a launch is observable (in the state written back): the frames remember that draw's
        // counter and the Philox block runs once, behind the loop, instead of on three frames out of four.
        constexpr bool kDefer1 = !AI1, kDefer2 = !AI2;
        BoldDefer bold{false, false, 0u, 0u};
        // (not beside a scout wave: two waves per SIMD leave 256 registers each, which that kernel already fills)
        const ZoneTable zones = SCOUT == kNoScout ? park_zone_table(a.cfg) : ZoneTable{};
        const ZoneTable* const parked_zones = SCOUT == kNoScout ? &zones : nullptr;
        TrajOut<OBS16> out;
        if (kTraj) out.init(a, i, lane, live);
        if (MODE != kTape) policy_actions(id.id_lo, id.id_hi, policy, a.t0, n_actions, a1, a2);
        // The frame in two halves (pz_physics.hpp, frame_head / frame_tail): the head of frame s + 1 -- round start, ball
        // step, the computer player's table gathers issued -- runs BEFORE frame s's observation rows are stored, so
        // that the gathers are not queued behind those 18 stores in the in-order vmcnt.
        const ScoutLink link{cand, hits, posts};
        bool resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
        // Every state load lands BEFORE the frame loop: left to itself the compiler waits for the 44 columns where the
        // first frame reads them -- inside the loop body, counting down to vmcnt(0) -- where on every later frame
        // those waits drain the previous frame's 18 row stores instead (and the gathers queued behind them).
        if (kTraj) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
        FrameHead head = frame_head<AI1, AI2, SCOUT, kDefer1, kDefer2>(g, a.cfg, id, live, lane, lut, link, &bold);
        // The compiler sizes the tail's wait for the head's gathers for the worst path into the loop: from here
        // nothing would follow them (vmcnt(0): every frame drains its predecessor's row stores after all), around the
        // back edge a frame's 18 row stores do.  Eighteen dropped stores make the two paths look alike: vmcnt(18).
        if (kTraj && (AI1 || AI2)) issue_dropped_stores<TrajOut<OBS16>::kStores>();
        for (int32_t s = 0; s < a.k; ++s) {
            if (MODE == kTape) {
                const int slot = s % kTapeChunk;
                if (slot == 0 && s != 0) fetch_tape_chunk(s, std::false_type{});
                a1 = parked[(slot * 2 + 0) * kLanes + lane];
                a2 = parked[(slot * 2 + 1) * kLanes + lane];
            }
            const bool last_frame = s == a.k - 1;
            frozen = head.frozen;
            reward = frame_tail<AI1, AI2, SCOUT, false>(g, a.cfg, id, a1, a2, live, head, lds_obs[0], lane, lut, link,
                                                        nullptr, last_frame);
            finished += (unsigned int)(live && g.e.game_ended && !frozen);
            rw = shape_rewards(a.cfg, g, reward, frozen, parked_zones);
            if (with_stats) stats_update(st, a.cfg, rw, resets, live && !frozen, as_float);
            // the next frame's policy draw (one block too many per launch): in a trajectory launch it is the independent
            // VALU work that runs under the LDS reads of this frame's rows
            auto next_policy = [&]() {
                if (MODE != kTape)
                    policy_actions(id.id_lo, id.id_hi, policy, a.t0 + (uint64_t)s + 1u, n_actions, a1, a2);
            };
            auto next_head = [&]() {  // this frame's outputs are staged: the game may move on
                if (s + 1 < a.k) {
                    resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
                    head = frame_head<AI1, AI2, SCOUT, kDefer1, kDefer2>(g, a.cfg, id, live, lane, lut, link, &bold);
                }
            };
            if (kTraj) {
                out.template stage<traj_small_aux(AI1 || AI2)>(a, g, rw, as_float, live, a1, a2, MODE == kRollout && a.act_out != nullptr, lane,
                                                               lds_obs);
                next_head();
                out.flush(lds_obs, lane, next_policy);
                out.advance();
            } else {
                next_policy();
                next_head();
            }
        }
        // the launch's last recorded boldness draws (physics.py:218)
        if (kDefer1 && bold.pending1) g.p1.bold = rng_integers(id, bold.counter1, 5u);
        if (kDefer2 && bold.pending2) g.p2.bold = rng_integers(id, bold.counter2, 5u);
    } else {
        const bool resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
        reward = step_games<AI1, AI2, SCOUT>(g, a.cfg, id, a1, a2, live, frozen, lds_obs[0], lane, lut,
                                             ScoutLink{cand, hits, posts}, &ex_pending);
        finished = (unsigned int)(live && g.e.game_ended && !frozen);
        rw = shape_rewards(a.cfg, g, reward, frozen);
        if (with_stats) stats_update(st, a.cfg, rw, resets, live && !frozen, as_float);
    }
    PZ_STAMP(2);
    if (live) {
        if constexpr (PACKED) {
            store_game_packed(g, pio, was, true);
        } else if (SPARSE) {
            store_game_changed(g, loaded, io, ex_pending);
        } else {
            // behind a frame loop the column offsets are computed afresh: the compiler otherwise keeps the 44 products
            // column x pitch of the loads alive across the loop for these stores -- 44 scalars of a budget of ~100,
            // spilled and brought back through v_readlane (the opaque copy of the pitch is a new value to it)
            StateIO back = io;
            if (MODE != kActions) asm volatile("" : "+s"(back.pitch));
            store_game(g, back, ex_pending);
        }
        if (with_stats) sio.store(st);
    }
    if (MODE != kRollout && MODE != kTape) emit_outputs(a, g, rw, as_float, live, i, lane, 0, lds_obs);
    PZ_STAMP(5);
    PZ_DRAIN_VMEM();
    PZ_STAMP(6);

    // the launches that read actions: pz_step (a1 / a2 as loaded) and pz_step_many (noted while the tape was parked)
    if (MODE == kActions) bad_action = actions_out_of_range(a.cfg, a1, a2, act_high);
    if (MODE == kActions || MODE == kTape) count_action_faults(a.cfg, live && bad_action);
    if (a.episodes_done != nullptr) {
        // one atomic per wave: reduce the per-lane counts across the wavefront first
        unsigned int total = finished;
        for (int off = kLanes / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kLanes);
        if (lane == 0 && total != 0) atomicAdd(a.episodes_done, (unsigned long long)total);
    }
}

// ---- the pair kernel: two waves per 64 games, split by player (see step_games_pair) ---------------
// Used for single-frame launches below kTwoWaveMaxLanes games: human-vs-human (the bench headline)
// and, when flight tables are passed, every computer-player configuration.  Wave ROLE loads/stores its own
// player's 13 columns, its half of the ball columns (both waves load all 12), player 1's wave also the 6
// env columns, the episode statistics and `terminated`; each wave writes its own agent's reward and
// observation tensor; the wave of the (last) computer player keeps ball.expected_landing_point_x.
// PACKED (pz_packed.hpp): both waves load groups A and B (16 bytes per lane each), a computer player's wave also the
// tail; player 1's wave writes group A, player 2's group B, and the tail's three fields are stored byte-wise by their
// owners behind everything else: a new round's boldness, and the landing point by the wave that keeps it.
// RANDOM: pz_step_random with k = 1 -- the uniform random policy drawn inside the launch (both waves draw both
// actions: one Philox block, under the state loads) instead of two action words fetched from HBM.
template <int ROLE, bool AI1, bool AI2, bool PACKED, bool RANDOM>
__device__ __forceinline__ void pair_body(const StepArgs& a, const HotArgs hot, int32_t (*lds_obs)[kLanes * PZ_OBS_DIM],
                                          int32_t* __restrict__ xchg, int lane)
{
    constexpr bool kOwnAI = ROLE == 0 ? AI1 : AI2;
    constexpr bool kKeepsEx = (AI1 || AI2) && (ROLE == 1 ? AI2 : !AI2);
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const bool live = i < hot.n;
    const uint32_t n32 = (uint32_t)hot.n;
    const StateIO io{make_rsrc(hot.state, PACKED ? 0u : (uint32_t)(hot.stride * (PZ_STATE_WORDS * 4))),
                     (uint32_t)hot.stride * 4u, (uint32_t)i * 4u};
    const PackedIO pio = make_packed_io(hot.state, PACKED ? hot.stride : 0, i);
    const bool as_float = a.cfg.ballpos_reward != 0 || a.cfg.normal_state_mode != 0;
    const bool with_stats = ROLE == 0 && a.episode_stats != nullptr && a.cfg.episode_stats_mode != 0;
    const StatsIO sio = make_stats_io(a.episode_stats, with_stats, a.stride, i);
    constexpr int kOwn = ROLE * PZ_P_WORDS, kOther = (1 - ROLE) * PZ_P_WORDS;

    // Packed state, human vs human: every other workgroup issues at priority 1.  The two waves of a SIMD belong to two
    // workgroups and compete for its VALU through the whole frame (packing and unpacking make this the format with the
    // most VALU work per byte moved); with one of them preferred, half of the launch's workgroups reach their stores
    // earlier and the write drain starts earlier: 6.36 -> 6.11 us per launch, 524 288 games 31.7 -> 31.2, with int16
    // rows 5.61 -> 5.46 / 23.7 -> 22.8.  The int32 columns do not care (6.996 vs 7.010; by role: 7.25-7.52).
    if (PACKED && !AI1 && !AI2 && (blockIdx.x & 1u) != 0) __builtin_amdgcn_s_setprio(1);
    // One computer player: its wave is the one the launch waits for, and it shares its SIMD with the human player's
    // wave of another workgroup, which has time to spare at its exchange barrier -- let the computer's wave issue first
    // (interleaved A/B: config 3 8.87 -> 8.67 us per launch, packed 7.63 -> 7.39; priority 3 the same)
    if (kOwnAI && (AI1 != AI2)) __builtin_amdgcn_s_setprio(1);
    Game g{};
    const RngId id = make_rng_id(a.cfg, live ? i : 0);
    const FlightLut lut = make_lut(a.tables);
    PZ_PAIR_STAMP(ROLE, 0);
    PZ_PAIR_WHERE(ROLE);
    // ---- one computer player: who may store what, and when ("the ordering argument") ----------------------
    // The human player's wave reaches the exchange barrier ~1 000 cycles before its partner (tools/stamps.py) and stores,
    // while it waits, columns that are final by then (kEarlyOwn / kEarlyBall below).  The computer's wave LOADS some of
    // those very columns at the top of the launch -- the ball, and the human player's x / state / diving direction, from
    // which it re-derives that player's move.  A store in front of the barrier is ordered behind those loads by nothing
    // but this hand-shake: the human player's wave clears one LDS word before anything else; the computer's wave waits
    // until every one of its state loads has RETURNED (s_waitcnt vmcnt(0): the values are in its registers) and then sets
    // the word; the human player's wave reads it in front of the barrier and stores early only if it is set.  If it is
    // not (the partner was held back: preemption, a debugger, an instruction-cache miss) -- or the set was overtaken by
    // the clear -- the same stores are issued behind the barrier like every other store of the frame: slower by the
    // ~2 % the early stores gain, never different.  Interleaved A/B, config 3, us per launch (cold | hot tape): no early
    // stores 8.62 | 8.32, round 4's unordered early stores 8.43 | 8.15, with the hand-shake 8.46 | 8.19
    // (profiles/r05_experiments/ab_early_store_edge_*.log).  The edge itself, deterministically:
    // a diagnostic build (pz_diagnostic.hpp, bits 8-15 = 2) holds the computer's wave back for ~16 000 cycles in front of its first load -- the
    // trajectory stays bit-exact with the hand-shake and breaks without it (early_store_edge_partner_held_back.log);
    // tests/test_cabi_and_host.py scans the shipped code object for the hand-shake's shape.
    constexpr bool kOneComputer = kEarlyOwnStores != 0 && !PACKED && (AI1 != AI2);
    constexpr int kHumanRole = AI1 ? 1 : 0;
    constexpr int kLoadsDoneAt = 2048;  // word of the human wave's exchange region (rows 0..1151 carry the exchange)
    constexpr int32_t kLoadsDone = 0x10ADD0E5;
    static_assert(kLoadsDoneAt >= kEarlyPostAt + 2 * kLanes && kLoadsDoneAt < kLanes * PZ_OBS_DIM, "inside the region, past the posts");
    // (relaxed workgroup-scope atomics: plain ds_write_b32 / ds_read_b32 that the compiler neither drops nor merges --
    // a `volatile` pointer would keep the generic address space and turn them into flat_ accesses)
    int32_t* const partner_loaded = xchg + kHumanRole * (kLanes * PZ_OBS_DIM) + kLoadsDoneAt;
    if (kOneComputer && !kOwnAI && lane == 0)
        __hip_atomic_store(partner_loaded, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if constexpr (diag::kDelayPartnerLoads != 0) {  // (diagnostic builds only: see above)
        if (kOneComputer && kOwnAI)
            for (int nap = 0; nap < diag::kDelayPartnerLoads; ++nap) __builtin_amdgcn_s_sleep(127);
    }
    int a1 = 0, a2 = 0;
    uint32_t act_high = 0u;
    if (!RANDOM) load_actions(hot.act_p1, hot.act_p2, n32, (uint32_t)i, hot.act_format, a1, a2, act_high);
    EpisodeStats st{0.0, 0.0, 0};
    uint32_t sticky = 0;  // PACKED: the own group's overflow flag, kept
    if (PACKED && live) {
        const pk_u32x4 ga = pio.ld_a(), gb = pio.ld_b();
        const uint32_t tail = kOwnAI ? pio.ld_tail() : 0u;  // computer_boldness is read by the decision only
        unpack_group_a(g, ga);
        unpack_group_b(g, gb);
        unpack_tail(g, tail);
        sticky = (ROLE == 0 ? ga.y : gb.y) & kPackedOverflowBit;
        if (with_stats) sio.load(st);
    }
    if (!PACKED && live) {
        g.e.round_ended = io.ld(PZ_E_ROUND_ENDED);
        g.e.game_ended = io.ld(PZ_E_GAME_ENDED);
        g.e.rng = (uint32_t)io.ld(PZ_E_RNG_DRAW_COUNTER);
        g.e.s1 = io.ld(PZ_E_SCORE_P1);
        g.e.s2 = io.ld(PZ_E_SCORE_P2);
        g.e.p2serve = io.ld(PZ_E_IS_PLAYER2_SERVE);
        g.b.x = io.ld(PZ_B_X);
        g.b.y = io.ld(PZ_B_Y);
        g.b.xv = io.ld(PZ_B_X_VELOCITY);
        g.b.yv = io.ld(PZ_B_Y_VELOCITY);
        g.b.power = io.ld(PZ_B_IS_POWER_HIT);
        g.b.px = io.ld(PZ_B_PREVIOUS_X);
        g.b.py = io.ld(PZ_B_PREVIOUS_Y);
        g.b.ppx = io.ld(PZ_B_PREVIOUS_PREVIOUS_X);
        g.b.ppy = io.ld(PZ_B_PREVIOUS_PREVIOUS_Y);
        g.b.rot = io.ld(PZ_B_FINE_ROTATION);
        if (kKeepsEx) g.b.ex = io.ld(PZ_B_EXPECTED_LANDING_POINT_X);
        g.b.punch = io.ld(PZ_B_PUNCH_EFFECT_X);
        load_player(ROLE == 0 ? g.p1 : g.p2, io, kOwn);
        Player& other = ROLE == 0 ? g.p2 : g.p1;
        other.coll = io.ld(kOther + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED);
        if (kOwnAI && ROLE == 0) other.x = io.ld(kOther + PZ_P_X);  // the decision reads the other player's x
        if (kOwnAI && ROLE == 1 && !AI1) {  // ... player 2's after player 1's move: recomputed from these
            other.x = io.ld(kOther + PZ_P_X);
            other.state = io.ld(kOther + PZ_P_STATE);
            other.dive = io.ld(kOther + PZ_P_DIVING_DIRECTION);
        }
        if (with_stats) sio.load(st);
    }
    if (kOneComputer && kOwnAI) {
        // every load above has returned -- the gathers' addresses need the ball columns next anyway -- then tell the partner
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(partner_loaded, kLoadsDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (RANDOM)  // (issued behind the loads: the block runs while they are in flight)
        policy_actions(id.id_lo, id.id_hi, make_rolling_key(a.action_seed), a.t0, a.cfg.simplify_action ? 13u : 18u, a1, a2);
    const Game loaded = g;  // what the columns held before the frame
    const bool resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
    bool frozen = false;
    PZ_DRAIN_VMEM();
    PZ_PAIR_STAMP(ROLE, 1);
    LandingProbe after_hit{false, false, 0u, 0u};
    bool bold_pending = false;  // a human player's new-round boldness draw, made behind the stores
    // In a launch with ONE computer player the human player's wave puts on their way, while it waits at the barrier,
    // (1) its own player's columns -- final once it has moved; the collision flag follows behind the barrier --
    // (interleaved A/B, config 3, cold tape: 8.58 -> 8.44 us per launch; human vs human, where nobody waits, the same
    // stores gained nothing in round 2)
    // (the computer's wave doing the same with its player costs: 8.36 -> 8.54, profiles/r04_experiments/)
    constexpr bool kEarlyOwn = kOneComputer && !kOwnAI;
    // ... and (2) the ball's position, trail and rotation: final once the world step has run (a ball-player collision
    // changes velocities, power-hit flag and punch_effect_x only) -- the human player's wave stores all seven, the
    // computer's wave none of them (8.44 -> 8.40; hot tape 8.30 -> 8.17 -> 8.15: profiles/r04_experiments/).  (Letting the
    // human player's wave SLEEP 256 - 1 536 cycles before its frame, so that the computer's wave has the SIMD to itself
    // up to its gathers, changes nothing: 8.38 -> 8.40 - 8.42, and 8.57 when it sleeps past its slack.)
    constexpr bool kEarlyBall = kEarlyOwn && kEarlyOwnStores >= 2;
    constexpr bool kPartnerStoresBall = kEarlyOwnStores >= 2 && kOneComputer && kOwnAI;
    bool stored_early = false;  // wave-uniform: the partner's loads were known to be back in front of the barrier
    auto store_ball_trail = [&]() {  // the human wave's seven ball columns (kEarlyBall): early, or behind the barrier
        io.st(PZ_B_X, g.b.x);
        io.st(PZ_B_Y, g.b.y);
        io.st(PZ_B_PREVIOUS_X, g.b.px);
        io.st(PZ_B_PREVIOUS_Y, g.b.py);
        io.st(PZ_B_PREVIOUS_PREVIOUS_X, g.b.ppx);
        io.st(PZ_B_PREVIOUS_PREVIOUS_Y, g.b.ppy);
        io.st(PZ_B_FINE_ROTATION, g.b.rot);
    };
    auto before_barrier = [&]() {
        if constexpr (kEarlyOwn) {
            if constexpr (diag::kUnorderedEarlyStores) {  // (diagnostic builds only: round 4's form, no hand-shake)
                stored_early = true;
            } else {
                asm volatile("" ::: "memory");  // (read here, in front of the barrier: as late as the stores allow)
                stored_early = __builtin_amdgcn_readfirstlane(__hip_atomic_load(partner_loaded, __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_WORKGROUP)) == kLoadsDone;
            }
            if (stored_early && live) {
                Player& mine = ROLE == 0 ? g.p1 : g.p2;
                const Player& was = ROLE == 0 ? loaded.p1 : loaded.p2;
                const int coll_now = mine.coll;
                mine.coll = was.coll;  // (not final yet: stored behind the barrier when it changed)
                store_player_changed(mine, was, io, kOwn);
                mine.coll = coll_now;
                if constexpr (kEarlyBall) store_ball_trail();
            }
        }
    };
    const int reward = step_games_pair<ROLE, AI1, AI2>(g, a.cfg, id, a1, a2, live, frozen, xchg, kLanes * PZ_OBS_DIM,
                                                        lane, lut, after_hit, bold_pending, before_barrier);
    const Rewards rw = shape_rewards(a.cfg, g, reward, frozen);
    if (with_stats) stats_update(st, a.cfg, rw, resets, live && !frozen, as_float);
    PZ_PAIR_STAMP(ROLE, 2);

    // the two halves of the write-back; their order is a compile-time choice (below)
    auto store_state = [&]() {
        if (!live) return;
        if constexpr (PACKED) {
            if (ROLE == 0) {
                pio.st_a(pack_group_a(g, sticky));
                if (with_stats) sio.store(st);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)g.e.game_ended, make_rsrc(a.terminated, n32),
                                                     (uint32_t)i, 0, 0);
            } else {
                pio.st_b(pack_group_b(g, sticky));
            }
            const Rsrc rew = make_rsrc(ROLE == 0 ? a.rew_p1 : a.rew_p2, n32 * 4u);
            const unsigned int bits = as_float ? __float_as_uint(ROLE == 0 ? rw.f1 : rw.f2)
                                               : (unsigned int)(ROLE == 0 ? rw.i1 : rw.i2);
            __builtin_amdgcn_raw_buffer_store_b32(bits, rew, io.voff, 0, 0);
            PZ_PAIR_STAMP(ROLE, 3);
            return;
        }
        // changed-only write-back of the rarely changing columns, as in store_game_changed
        if (kEarlyOwn && stored_early) {
            const Player& mine = ROLE == 0 ? g.p1 : g.p2;
            if (mine.coll != (ROLE == 0 ? loaded.p1 : loaded.p2).coll) io.st(kOwn + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED, mine.coll);
        } else {
            store_player_changed(ROLE == 0 ? g.p1 : g.p2, ROLE == 0 ? loaded.p1 : loaded.p2, io, kOwn);
            if (kEarlyBall) store_ball_trail();  // (the partner was late: what the early stores would have carried)
        }
        if (ROLE == 0) {
            if (!kEarlyBall && !kPartnerStoresBall) {
                io.st(PZ_B_X, g.b.x);
                io.st(PZ_B_Y, g.b.y);
            }
            io.st(PZ_B_Y_VELOCITY, g.b.yv);
            if (g.b.xv != loaded.b.xv) io.st(PZ_B_X_VELOCITY, g.b.xv);
            if (g.b.power != loaded.b.power) io.st(PZ_B_IS_POWER_HIT, g.b.power);
            if (g.b.punch != loaded.b.punch) io.st(PZ_B_PUNCH_EFFECT_X, g.b.punch);
            if (g.e.s1 != loaded.e.s1) io.st(PZ_E_SCORE_P1, g.e.s1);
            if (g.e.s2 != loaded.e.s2) io.st(PZ_E_SCORE_P2, g.e.s2);
            if (g.e.p2serve != loaded.e.p2serve) io.st(PZ_E_IS_PLAYER2_SERVE, g.e.p2serve);
            if (g.e.round_ended != loaded.e.round_ended) io.st(PZ_E_ROUND_ENDED, g.e.round_ended);
            if (g.e.game_ended != loaded.e.game_ended) io.st(PZ_E_GAME_ENDED, g.e.game_ended);
            if (g.e.rng != loaded.e.rng) io.st(PZ_E_RNG_DRAW_COUNTER, (int32_t)g.e.rng);
            if (with_stats) sio.store(st);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)g.e.game_ended, make_rsrc(a.terminated, n32),
                                                 (uint32_t)i, 0, 0);
        } else if (!kEarlyBall && !kPartnerStoresBall) {
            io.st(PZ_B_PREVIOUS_X, g.b.px);
            io.st(PZ_B_PREVIOUS_Y, g.b.py);
            io.st(PZ_B_PREVIOUS_PREVIOUS_X, g.b.ppx);
            io.st(PZ_B_PREVIOUS_PREVIOUS_Y, g.b.ppy);
            io.st(PZ_B_FINE_ROTATION, g.b.rot);
        }
        const Rsrc rew = make_rsrc(ROLE == 0 ? a.rew_p1 : a.rew_p2, n32 * 4u);
        const unsigned int bits = as_float ? __float_as_uint(ROLE == 0 ? rw.f1 : rw.f2)
                                           : (unsigned int)(ROLE == 0 ? rw.i1 : rw.i2);
        __builtin_amdgcn_raw_buffer_store_b32(bits, rew, io.voff, 0, 0);
        PZ_PAIR_STAMP(ROLE, 3);
    };
    auto store_observations = [&]() {
        if (live) {
            const Player& me = ROLE == 0 ? g.p1 : g.p2;
            const Player& opp = ROLE == 0 ? g.p2 : g.p1;
            if (a.cfg.normalize_obs == 1)
                stage_one_obs_t<true>(me, opp, g.b, lds_obs[ROLE], lane);
            else
                stage_one_obs_t<false>(me, opp, g.b, lds_obs[ROLE], lane);
        }
        // A wave stages and flushes ITS OWN rows, and after the exchange barrier nobody else touches them (the partner
        // wrote its player into these rows before that barrier): the wave's LDS instructions execute in issue order,
        // so only the compiler has to be kept from reordering them -- no second workgroup barrier (7.24 -> 7.12 us).
        wave_lds_handover<false>();
        PZ_PAIR_STAMP(ROLE, 4);
        flush_obs(lds_obs[ROLE], ROLE == 0 ? a.obs_p1 : a.obs_p2, 0, a.n, a.cfg.normalize_obs, lane);
        PZ_PAIR_STAMP(ROLE, 5);
    };
    // Interleaved A/B (tools/ab.py, us per launch, state first | observations first): human vs human 7.14 | 7.27,
    // player 2 = computer 8.58 | 8.41 -- the observation tensors are three quarters of the written bytes, and in the
    // computer-player launch the waves reach their stores less evenly (again with round 4's early stores: 8.36 | 8.43).
    if (AI1 || AI2) {
        store_observations();
        store_state();
    } else {
        store_state();
        store_observations();
    }
    PZ_DRAIN_VMEM();
    PZ_PAIR_STAMP(ROLE, 6);
    if (bold_pending) {  // draw number loaded.rng + ROLE of the env stream: player 1's, then player 2's (physics.py:218)
        uint32_t counter = loaded.e.rng + (uint32_t)ROLE;
        const int bold = rng_integers(id, counter, 5u);
        if constexpr (PACKED)
            pio.st_bold(ROLE, bold);
        else
            io.st(kOwn + PZ_P_COMPUTER_BOLDNESS, bold);
    }
    if (PACKED && (AI1 || AI2)) {
        // the boldness drawn inside the frame (player_new_round): a round started iff the game was live and between rounds
        const bool started = live && loaded.e.round_ended != 0 && !(loaded.e.game_ended != 0 && a.cfg.auto_reset == 0);
        if (started) pio.st_bold(ROLE, (ROLE == 0 ? g.p1 : g.p2).bold);
    }
    // last: after a ball-player collision the value comes from a table gather issued at the end of the frame
    // (taken over by the human player's wave -- disjoint lanes, the same store -- it costs: 8.58 -> 8.65 us per launch
    // alone, nothing on top of the early stores, packed 7.22 -> 7.60: profiles/r04_experiments/)
    if (kKeepsEx) {
        const int ex = lut.landing_finish(after_hit, g.b.x, g.b.y, g.b.xv, g.b.yv, g.b.ex);
        if (live && ex != loaded.b.ex) {
            if constexpr (PACKED)
                pio.st_ex(ex);
            else
                io.st(PZ_B_EXPECTED_LANDING_POINT_X, ex);
        }
    }
    // (player 1's wave checks both action words: both waves load both)
    if (!RANDOM && ROLE == 0) count_action_faults(a.cfg, live && actions_out_of_range(a.cfg, a1, a2, act_high));
    if (RANDOM && ROLE == 0 && a.episodes_done != nullptr) {  // pz_step_random's counter: one atomic per workgroup
        unsigned int total = (unsigned int)(live && g.e.game_ended && !frozen);
        for (int off = kLanes / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kLanes);
        if (lane == 0 && total != 0) atomicAdd(a.episodes_done, (unsigned long long)total);
    }
}

// (32 games per workgroup -- half-filled waves, four per SIMD at 65 536 games instead of two, more of them to hide each
// other's memory latency -- was built for this kernel in round 4 and lost: human vs human 6.97 -> 7.49 us per launch,
// player 2 = computer 8.57 -> 9.35, profiles/r04_experiments/ab_g32_and_early_stores.log)
template <bool AI1, bool AI2, bool PACKED = false, bool RANDOM = false>
__global__ __launch_bounds__(2 * kLanes) void step_pair_kernel(PZ_HOT_PARAMS, const StepArgs a)
{
    const HotArgs hot{state, n, stride, act_p1, act_p2, act_format};
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    // the player exchange of step_games_pair lives in the staging rows (each wave's incoming data in its own
    // rows, overwritten by nobody else): 17.5 KB of LDS per workgroup, 8 workgroups per CU
    int32_t* xchg = &lds_obs[0][0];
    // the wave index is uniform by construction; readfirstlane makes that visible to the compiler
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kLanes - 1);
    if (role == 0)
        pair_body<0, AI1, AI2, PACKED, RANDOM>(a, hot, lds_obs, xchg, lane);
    else
        pair_body<1, AI1, AI2, PACKED, RANDOM>(a, hot, lds_obs, xchg, lane);
}

// ---- the k-frame pair kernel: pz_rollout_random on two waves per 64 games ------------------------------------------
// With a computer player the single-wave rollout issues ~1 050 VALU instructions plus a third as many scalar ones per
// frame from ONE wave per SIMD: 9 800 cycles, 4.3 us -- a third more than the memory system needs for the frame's
// bytes (section 4.8 of DESIGN.md).  Split by player like the single-frame pair kernel, each wave carries half of
// the frame (its player's decision and move; the cheap shared parts twice) -- and with ONE computer player the human
// player's wave, which would otherwise wait for the computer's at every frame's barrier, writes ALL of the frame's
// outputs, so that the computer's wave has nothing but its decision on the frame's critical path and no store in
// its in-order vmcnt.  The launch is left with the write rate as its bound: 3.5 us per frame.  One LDS exchange + one workgroup barrier per frame (two with two computer players);
// the frame in two halves as in the single-wave loop (pair_frame_head / pair_frame_tail), for the gathers' sake.
// MODE: kRollout (pz_rollout_random: both waves draw the policy's block) or kTape (pz_step_many: the tape is parked in
// LDS kTapeChunk frames at a time, player 1's wave fetching player 1's row of every frame, player 2's wave the other).
template <int ROLE, bool AI1, bool AI2, int MODE, bool PACKED, bool OBS16>
__device__ __forceinline__ void rollout_pair_body(const StepArgs& a, const HotArgs hot,
                                                  int32_t (*lds_obs)[kLanes * PZ_OBS_DIM], int32_t* __restrict__ xchg,
                                                  int32_t* __restrict__ tape_lds, int lane)
{
    static_assert(MODE == kRollout || MODE == kTape, "the trajectory launches");
    constexpr bool kOwnAI = ROLE == 0 ? AI1 : AI2;
    constexpr bool kKeepsEx = (AI1 || AI2) && (ROLE == 1 ? AI2 : !AI2);
    // (re-measured with the observation tensors in two HBM ranks: symmetric outputs 3.70 vs 3.42 us per frame, tape 4.45 vs 3.90)
    constexpr bool kWritesAll = AI1 != AI2 && !kOwnAI;   // one computer player: the human player's wave writes the outputs
    constexpr bool kWritesNone = AI1 != AI2 && kOwnAI;   // ... and the computer's wave none of them
    constexpr int kOwn = ROLE * PZ_P_WORDS, kOther = (1 - ROLE) * PZ_P_WORDS;
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const bool live = i < hot.n;
    const uint32_t n32 = (uint32_t)hot.n;
    const StateIO io{make_rsrc(hot.state, PACKED ? 0u : (uint32_t)(hot.stride * (PZ_STATE_WORDS * 4))),
                     (uint32_t)hot.stride * 4u, (uint32_t)i * 4u};
    const PackedIO pio = make_packed_io(hot.state, PACKED ? hot.stride : 0, i);
    const bool as_float = a.cfg.ballpos_reward != 0 || a.cfg.normal_state_mode != 0;
    const bool with_stats = ROLE == 0 && a.episode_stats != nullptr && a.cfg.episode_stats_mode != 0;
    const StatsIO sio = make_stats_io(a.episode_stats, with_stats, a.stride, i);

    // (as in the single-frame pair kernel: the computer's wave is every frame's critical path; pz_rollout_random at
    // k = 32 3.12-3.25 -> 2.99-3.01 us per frame in the same position of the interleaved rounds, k = 128 3.09 -> 2.79)
    if (kOwnAI && (AI1 != AI2)) __builtin_amdgcn_s_setprio(1);
    Game g{};
    RngId id = make_rng_id(a.cfg, live ? i : 0);
    id.ks = make_parked_schedule(a.cfg.seed);  // the frame loop's key schedules live in VGPRs (KeySchedule)
    const KeySchedule policy = make_parked_schedule(a.action_seed);
    const FlightLut lut = make_lut(a.tables);
    Player& own = ROLE == 0 ? g.p1 : g.p2;
    Player& other = ROLE == 0 ? g.p2 : g.p1;
    EpisodeStats st{0.0, 0.0, 0};
    uint32_t sticky = 0;  // PACKED: the own group's overflow flag, kept
    if (PACKED && live) {
        const pk_u32x4 ga = pio.ld_a(), gb = pio.ld_b();
        const uint32_t tail = kOwnAI ? pio.ld_tail() : 0u;  // computer_boldness / the landing point: the computer's wave only
        unpack_group_a(g, ga);
        unpack_group_b(g, gb);
        unpack_tail(g, tail);
        sticky = (ROLE == 0 ? ga.y : gb.y) & kPackedOverflowBit;
    }
    if (!PACKED && live) {
        g.e.round_ended = io.ld(PZ_E_ROUND_ENDED);
        g.e.game_ended = io.ld(PZ_E_GAME_ENDED);
        g.e.rng = (uint32_t)io.ld(PZ_E_RNG_DRAW_COUNTER);
        g.e.s1 = io.ld(PZ_E_SCORE_P1);
        g.e.s2 = io.ld(PZ_E_SCORE_P2);
        g.e.p2serve = io.ld(PZ_E_IS_PLAYER2_SERVE);
        g.b.x = io.ld(PZ_B_X);
        g.b.y = io.ld(PZ_B_Y);
        g.b.xv = io.ld(PZ_B_X_VELOCITY);
        g.b.yv = io.ld(PZ_B_Y_VELOCITY);
        g.b.power = io.ld(PZ_B_IS_POWER_HIT);
        g.b.px = io.ld(PZ_B_PREVIOUS_X);
        g.b.py = io.ld(PZ_B_PREVIOUS_Y);
        g.b.ppx = io.ld(PZ_B_PREVIOUS_PREVIOUS_X);
        g.b.ppy = io.ld(PZ_B_PREVIOUS_PREVIOUS_Y);
        g.b.rot = io.ld(PZ_B_FINE_ROTATION);
        if (kKeepsEx) g.b.ex = io.ld(PZ_B_EXPECTED_LANDING_POINT_X);
        g.b.punch = io.ld(PZ_B_PUNCH_EFFECT_X);
        load_player(own, io, kOwn);
        // of the partner: what an exchange hands over (its state before its next move) and the collision flag
        other.x = io.ld(kOther + PZ_P_X);
        other.y = io.ld(kOther + PZ_P_Y);
        other.yv = io.ld(kOther + PZ_P_Y_VELOCITY);
        other.state = io.ld(kOther + PZ_P_STATE);
        other.frame = io.ld(kOther + PZ_P_FRAME_NUMBER);
        other.delay = io.ld(kOther + PZ_P_DELAY_BEFORE_NEXT_FRAME);
        other.dive = io.ld(kOther + PZ_P_DIVING_DIRECTION);
        other.lying = io.ld(kOther + PZ_P_LYING_DOWN_DURATION_LEFT);
        other.hitprev = io.ld(kOther + PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS);
        other.coll = io.ld(kOther + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED);
    }
    if (with_stats && live) sio.load(st);
    const uint32_t n_actions = a.cfg.simplify_action ? 13u : 18u;
    BoldDefer bold{false, false, 0u, 0u};
    const ZoneTable zones = park_zone_table(a.cfg);
    TrajOut<OBS16> out;
    out.init(a, i, lane, live);
    char*& own_rows = ROLE == 0 ? out.obs1 : out.obs2;
    int a1 = 0, a2 = 0;
    unsigned int finished = 0;
    bool any_round_started = false;
    if (MODE == kRollout) policy_actions(id.id_lo, id.id_hi, policy, a.t0, n_actions, a1, a2);
    // pz_step_many: every wave parks its own player's row of kTapeChunk frames in LDS (the computer's wave fetching both
    // -- it has no stores the wait would drain -- was slower: 4.31 vs 4.12 us per frame); the first chunk behind the
    // state loads, its latency is theirs
    unsigned char* const parked = reinterpret_cast<unsigned char*>(tape_lds);
    bool bad_action = false;  // (every wave checks the row it parks: its own player's)
    auto fetch_tape_chunk = [&](int32_t s0, auto first) {
        bad_action |= park_tape_chunk<decltype(first)::value, ROLE == 0, ROLE == 1>(static_cast<const int32_t*>(a.act_p1), a.n, n32,
                                                                                     a.k, s0, io.voff, parked, lane, action_count(a.cfg));
        __syncthreads();
    };
    if (MODE == kTape) fetch_tape_chunk(0, std::true_type{});
    __builtin_amdgcn_s_waitcnt(0x0F70);  // every state load lands before the frame loop (see step_kernel)
    bool resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
    any_round_started |= live && g.e.round_ended != 0 && !(g.e.game_ended != 0 && a.cfg.auto_reset == 0);
    bool ex_fresh = false;  // a computer's wave: g.b.ex is the landing point of the ball as it stands (pz_physics.hpp)
    PairHead head = pair_frame_head<ROLE, AI1, AI2, !kOwnAI>(g, a.cfg, id, live, lut, &bold, ex_fresh);
    // (the gathers' wait, see step_kernel's frame loop -- for a computer's wave that stores rows: with two computer players)
    if (kOwnAI && !kWritesNone) issue_dropped_stores<TrajOut<OBS16>::kStores / 2>();
    for (int32_t s = 0; s < a.k; ++s) {
        if (MODE == kTape) {
            // (both waves read frame s - 1's actions before that frame's exchange barrier: nobody needs the old chunk here)
            const int slot = s % kTapeChunk;
            if (slot == 0 && s != 0) fetch_tape_chunk(s, std::false_type{});
            a1 = parked[(slot * 2 + 0) * kLanes + lane];
            a2 = parked[(slot * 2 + 1) * kLanes + lane];
        }
        const bool last_frame = s == a.k - 1;
        const bool frozen = head.frozen;
        const int reward = pair_frame_tail<ROLE, AI1, AI2>(g, a.cfg, id, a1, a2, live, head,
                                                           xchg + (s & 1) * (2 * kLoopXchgRegion), lane, lut, last_frame,
                                                           &ex_fresh);
        finished += (unsigned int)(live && g.e.game_ended && !frozen);
        const Rewards rw = shape_rewards(a.cfg, g, reward, frozen, &zones);
        if (with_stats) stats_update(st, a.cfg, rw, resets, live && !frozen, as_float);
        // The frame's outputs.  With ONE computer player its wave is the frame's critical path (its decision on top of
        // everything the partner does), so the HUMAN player's wave writes everything -- both agents' rewards and rows,
        // the flag, the actions -- and the computer's wave nothing (kWritesAll / kWritesNone; it then has no store in
        // its in-order vmcnt either); with two computer players each wave writes its agent's share.
        if (!kWritesNone) {
            if (kWritesAll || ROLE == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(rw.f1) : (unsigned int)rw.i1,
                                                      make_rsrc(out.rew1, n32 * 4u), out.voff, 0, traj_small_aux(AI1 || AI2));
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)g.e.game_ended, make_rsrc(out.term, n32), out.ioff, 0, traj_small_aux(AI1 || AI2));
                if (MODE == kRollout && a.act_out != nullptr) {
                    const Rsrc ao = make_rsrc(out.act, n32 * 8u);
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned int)a1, ao, out.voff, 0, traj_small_aux(AI1 || AI2));
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned int)a2, ao, out.voff, n32 * 4u, traj_small_aux(AI1 || AI2));
                }
            }
            if (kWritesAll || ROLE == 1)
                __builtin_amdgcn_raw_buffer_store_b32(as_float ? __float_as_uint(rw.f2) : (unsigned int)rw.i2,
                                                      make_rsrc(out.rew2, n32 * 4u), out.voff, 0, traj_small_aux(AI1 || AI2));
            if (live) {
                if (kWritesAll) {
                    stage_obs(g, lds_obs[0], lds_obs[1], lane, a.cfg.normalize_obs == 1);
                } else if (a.cfg.normalize_obs == 1) {
                    stage_one_obs_t<true>(own, other, g.b, lds_obs[ROLE], lane);
                } else {
                    stage_one_obs_t<false>(own, other, g.b, lds_obs[ROLE], lane);
                }
            }
            wave_lds_handover<false>();  // the rows are read back by this wave only
        }
        if (s + 1 < a.k) {  // this frame's outputs are staged: the game may move on
            resets = live && g.e.game_ended != 0 && a.cfg.auto_reset != 0;
            any_round_started |= live && g.e.round_ended != 0 && !(g.e.game_ended != 0 && a.cfg.auto_reset == 0);
            head = pair_frame_head<ROLE, AI1, AI2, !kOwnAI>(g, a.cfg, id, live, lut, &bold, ex_fresh);
        }
        auto next_policy = [&]() {
            if (MODE == kRollout) policy_actions(id.id_lo, id.id_hi, policy, a.t0 + (uint64_t)s + 1u, n_actions, a1, a2);
        };
        if (kWritesNone)
            next_policy();
        else if (kWritesAll)
            out.flush(lds_obs, lane, next_policy);
        else
            out.flush_tensor(reinterpret_cast<const u32x4*>(lds_obs[ROLE]), own_rows, lane, next_policy);
        out.advance();
    }
    if (!kOwnAI && bold.pending1) own.bold = rng_integers(id, bold.counter1, 5u);  // the launch's last recorded draw

    // ---- the state back: every wave its player, its half of the ball; player 1's wave the env words
    if (live) {
        if constexpr (PACKED) {
            // (the three descriptors built afresh behind the frame loop -- the opaque copy of the stride is a new value to
            // the compiler: twelve scalars less to carry around the loop, as with the column offsets below)
            int64_t stride_again = hot.stride;
            asm volatile("" : "+s"(stride_again));
            const PackedIO back = make_packed_io(hot.state, stride_again, i);
            if (ROLE == 0)
                back.st_a(pack_group_a(g, sticky));
            else
                back.st_b(pack_group_b(g, sticky));
            if (kOwnAI ? any_round_started : bold.pending1) back.st_bold(ROLE, own.bold);
            if (kKeepsEx) back.st_ex(g.b.ex);
        } else {
            // (column offsets computed afresh behind the frame loop: see step_kernel)
            StateIO back = io;
            asm volatile("" : "+s"(back.pitch));
            store_player(own, back, kOwn);
            if (ROLE == 0) {
                back.st(PZ_B_X, g.b.x);
                back.st(PZ_B_Y, g.b.y);
                back.st(PZ_B_X_VELOCITY, g.b.xv);
                back.st(PZ_B_Y_VELOCITY, g.b.yv);
                back.st(PZ_B_IS_POWER_HIT, g.b.power);
                back.st(PZ_B_PUNCH_EFFECT_X, g.b.punch);
                back.st(PZ_E_SCORE_P1, g.e.s1);
                back.st(PZ_E_SCORE_P2, g.e.s2);
                back.st(PZ_E_IS_PLAYER2_SERVE, g.e.p2serve);
                back.st(PZ_E_ROUND_ENDED, g.e.round_ended);
                back.st(PZ_E_GAME_ENDED, g.e.game_ended);
                back.st(PZ_E_RNG_DRAW_COUNTER, (int32_t)g.e.rng);
            } else {
                back.st(PZ_B_PREVIOUS_X, g.b.px);
                back.st(PZ_B_PREVIOUS_Y, g.b.py);
                back.st(PZ_B_PREVIOUS_PREVIOUS_X, g.b.ppx);
                back.st(PZ_B_PREVIOUS_PREVIOUS_Y, g.b.ppy);
                back.st(PZ_B_FINE_ROTATION, g.b.rot);
            }
            if (kKeepsEx) back.st(PZ_B_EXPECTED_LANDING_POINT_X, g.b.ex);
        }
        if (with_stats) sio.store(st);
    }
    if (MODE == kTape) count_action_faults(a.cfg, live && bad_action);
    if (ROLE == 0 && a.episodes_done != nullptr) {
        unsigned int total = finished;
        for (int off = kLanes / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kLanes);
        if (lane == 0 && total != 0) atomicAdd(a.episodes_done, (unsigned long long)total);
    }
}

template <bool AI1, bool AI2, int MODE, bool PACKED = false, bool OBS16 = false, bool PLAIN = false>
__global__ __launch_bounds__(2 * kLanes) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rollout_pair_kernel(PZ_HOT_PARAMS, const StepArgs args)
{
    const StepArgs a = effective_args<PLAIN, OBS16>(args);
    const HotArgs hot{state, n, stride, act_p1, act_p2, act_format};
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    __shared__ int32_t xchg[kLoopXchgWords];  // the players' exchange: LDS of its own, double-buffered by frame parity
    __shared__ int32_t tape_lds[MODE == kTape ? kTapeWords : 1];  // parked action tape (kTape only)
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kLanes - 1);
    if (role == 0)
        rollout_pair_body<0, AI1, AI2, MODE, PACKED, OBS16>(a, hot, lds_obs, xchg, tape_lds, lane);
    else
        rollout_pair_body<1, AI1, AI2, MODE, PACKED, OBS16>(a, hot, lds_obs, xchg, tape_lds, lane);
}

// ---- constructor / reset / observe / policy kernels --------------------------------------------
// One game in either format, for the kernels outside the step path.
template <bool PACKED>
struct AnyIO {
    StateIO io;
    PackedIO pio;
    __device__ __forceinline__ AnyIO(const void* state, int64_t stride, int64_t i)
        : io{make_rsrc(state, PACKED ? 0u : (uint32_t)(stride * (PZ_STATE_WORDS * 4))), (uint32_t)stride * 4u, (uint32_t)i * 4u},
          pio(make_packed_io(state, PACKED ? stride : 0, i))
    {
    }
    __device__ __forceinline__ PackedWords load(Game& g) const
    {
        if constexpr (PACKED) return load_game_packed(g, pio, true);
        load_game(g, io);
        return PackedWords{};
    }
    __device__ __forceinline__ void store(const Game& g, const PackedWords& was) const
    {
        if constexpr (PACKED) {
            pio.st_a(pack_group_a(g, was.a.y & kPackedOverflowBit));
            pio.st_b(pack_group_b(g, was.b.y & kPackedOverflowBit));
            pio.st_tail(pack_tail(g));
        } else {
            store_game(g, io);
        }
    }
};

template <bool PACKED>
__global__ __launch_bounds__(kLanes) void init_kernel(int32_t* state, int64_t n, int64_t stride, const pz_config cfg)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    const AnyIO<PACKED> io(state, stride, i);
    Game g;
    const RngId id = make_rng_id(cfg, i);
    construct_game(g, id);
    io.store(g, PackedWords{});
}

template <bool PACKED>
__global__ __launch_bounds__(kLanes) void reset_kernel(int32_t* state, int64_t n, int64_t stride, const pz_config cfg,
                                                       const uint8_t* mask, int32_t* obs_p1, int32_t* obs_p2,
                                                       void* episode_stats)
{
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const AnyIO<PACKED> io(state, stride, i);
    if (i < n) {
        Game g;
        const PackedWords was = io.load(g);
        if (mask == nullptr || mask[i] != 0) {
            const RngId id = make_rng_id(cfg, i);
            reset_game(g, cfg, id);
            io.store(g, was);
            if (episode_stats != nullptr)  // RecordEpisodeStatistics.reset (:23-25)
                make_stats_io(episode_stats, true, stride, i).store(EpisodeStats{0.0, 0.0, 0});
        }
        stage_obs(g, lds_obs[0], lds_obs[1], lane, cfg.normalize_obs == 1);
    }
    __syncthreads();
    if (obs_p1 != nullptr) flush_obs(lds_obs[0], obs_p1, 0, n, cfg.normalize_obs, lane);
    if (obs_p2 != nullptr) flush_obs(lds_obs[1], obs_p2, 0, n, cfg.normalize_obs, lane);
}

template <bool PACKED>
__global__ __launch_bounds__(kLanes) void observe_kernel(const int32_t* state, int64_t n, int64_t stride,
                                                         int normalize, int32_t* obs_p1, int32_t* obs_p2)
{
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kLanes + lane;
    const AnyIO<PACKED> io(state, stride, i);
    if (i < n) {
        Game g;
        io.load(g);
        stage_obs(g, lds_obs[0], lds_obs[1], lane, normalize == 1);
    }
    __syncthreads();
    if (obs_p1 != nullptr) flush_obs(lds_obs[0], obs_p1, 0, n, normalize, lane);
    if (obs_p2 != nullptr) flush_obs(lds_obs[1], obs_p2, 0, n, normalize, lane);
}

__global__ __launch_bounds__(256) void random_actions_kernel(int32_t* act_p1, int32_t* act_p2, int64_t n,
                                                             int64_t env_id_base, uint64_t action_seed, uint64_t t,
                                                             uint32_t n_actions)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t gid = (uint64_t)(env_id_base + i);
    int a1, a2;
    policy_actions((uint32_t)gid, (uint32_t)(gid >> 32), make_rolling_key(action_seed), t, n_actions, a1, a2);
    act_p1[i] = a1;
    act_p2[i] = a2;
}

// ---- memory-placement probe (pz_probe_write): the trajectory kernels' store pattern into one or two buffers -------
// Each buffer is taken as [frames][kProbeSpans spans of kProbeSpanBytes]: workgroup w writes span w of every frame,
// 16 bytes per lane, into both buffers in turn -- two observation tensors of a k-frame launch, nothing in front of
// the stores.  Timing it on (a), (b) and (a, b) tells whether the two allocations share an HBM rank.
constexpr uint32_t kProbeSpanBytes = 64u * 35u * 4u;  // one wave's observation rows of a frame
constexpr uint32_t kProbeSpans = 1024;                // = waves of a 65 536-game launch
constexpr int64_t kProbeFrameBytes = (int64_t)kProbeSpanBytes * kProbeSpans;

__global__ __launch_bounds__(64) void probe_write_kernel(char* a, char* b, int32_t frames)
{
    const int lane = threadIdx.x;
    const uint32_t span_off = blockIdx.x * kProbeSpanBytes;
    for (int32_t f = 0; f < frames; ++f) {
        const int64_t at = (int64_t)f * kProbeFrameBytes + span_off;
        for (int side = 0; side < 2; ++side) {
            char* base = side ? b : a;
            if (base == nullptr) continue;  // (uniform)
            const Rsrc span = make_rsrc(base + at, kProbeSpanBytes);
#pragma unroll
            for (int pass = 0; pass < 9; ++pass) {
                const uint32_t v = (uint32_t)(pass * 64 + lane);
                const u32x4 w = {(uint32_t)f, v, (uint32_t)side, 0u};
                // (nt, the policy placement.py's thresholds were measured with -- not the k-frame launches' kTrajAux)
                __builtin_amdgcn_raw_buffer_store_b128(w, span, v * 16u, 0, kObsAux);  // beyond the span: dropped
            }
        }
    }
}



// ---- int32 columns <-> packed format (pz_pack_state / pz_unpack_state) ----------------------------------------
__global__ __launch_bounds__(kLanes) void pack_state_kernel(const int32_t* state, int64_t n, int64_t stride, void* packed,
                                                            int64_t packed_stride, unsigned long long* misfits)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    Game g;
    AnyIO<false>(state, stride, i).load(g);
    const bool fits = game_fits(g);
    if (!fits && misfits != nullptr) atomicAdd(misfits, 1ull);
    PackedWords flagged{};
    flagged.a.y = fits ? 0u : kPackedOverflowBit;  // a misfit stays visible in the packed state
    AnyIO<true>(packed, packed_stride, i).store(g, flagged);
}

__global__ __launch_bounds__(256) void count_misfits_kernel(const void* packed, int64_t n, int64_t packed_stride,
                                                            unsigned long long* flagged)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // word 1 of group A and of group B carry the sticky flags (pz_packed.hpp)
    const uint32_t* a = static_cast<const uint32_t*>(packed);
    const uint32_t* b = a + packed_stride * 4;
    const bool bad = i < n && ((a[i * 4 + 1] | b[i * 4 + 1]) & kPackedOverflowBit) != 0;
    const unsigned long long votes = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && votes != 0ull) atomicAdd(flagged, (unsigned long long)__popcll(votes));
}

__global__ __launch_bounds__(kLanes) void unpack_state_kernel(const void* packed, int64_t n, int64_t packed_stride,
                                                              int32_t* state, int64_t stride, unsigned long long* flagged)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    Game g;
    const PackedWords w = AnyIO<true>(packed, packed_stride, i).load(g);
    if (((w.a.y | w.b.y) & kPackedOverflowBit) != 0 && flagged != nullptr) atomicAdd(flagged, 1ull);
    AnyIO<false>(state, stride, i).store(g, PackedWords{});
}

// ---- flight tables (pz_build_flight_tables) -----------------------------------------------------
// One thread per entry, filled with the frame-by-frame iteration of the reference.
__global__ __launch_bounds__(256) void build_landing_table_kernel(uint16_t* table)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= kFtLandingEntries) return;
    const int x = (int)(e % kFtXCount) + kBallRadius;
    int64_t r = e / kFtXCount;
    const int y = (int)(r % kFtYCount);
    r /= kFtYCount;
    const int xv = ft_xv_value((int)(r % kFtXvCount));
    const int yv = (int)(r / kFtXvCount) - PZ_FT_YV_MAX;
    table[e] = (uint16_t)predict_landing_x_iterative<true>(x, y, xv, yv);
}

__global__ __launch_bounds__(256) void build_power_hit_table_kernel(uint16_t* table)
{
    // one thread per (entry, candidate): 8 slots per entry, slots 6 and 7 are padding
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= kFtHitEntries * 8) return;
    const int c = (int)(t & 7);
    const int64_t e = t >> 3;
    const int x = (int)(e % kFtXCount) + kBallRadius;
    const int64_t r = e / kFtXCount;
    const int y = (int)(r % kFtHitYCount) + kFtHitYMin;
    const int ayv = (int)(r / kFtHitYCount);
    uint16_t v = 0;
    if (c < 6) {
        const int xdir = candidate_xdir(c), ydir = candidate_ydir(c);
        const int sxv = (x < kGroundHalfWidth) ? (xdir + 1) * 10 : -(xdir + 1) * 10;  // physics.py:841-844
        v = (uint16_t)predict_landing_x_iterative<false>(x, y, sxv, ayv * ydir * 2);  // :845
    }
    table[t] = v;
}

// ---- rgb_array frames from the state (pz_render; raw_env.render pikazoo_env.py:250-384) -------------------------
// One thread per four horizontally adjacent pixels (12 output bytes = three dwords; a wave writes 768 contiguous
// bytes), grid = (pixel groups, frames).  The draw list of a frame is a handful of blits derived from wave-uniform
// state words, so every thread walks the same short list and tests its pixels against each rectangle.
struct Blit {
    int sprite, x0, y0, flip;
};

__device__ __forceinline__ uint32_t blend_over(uint32_t dst, uint32_t src)
{
    // pygame's per-pixel-alpha blit onto an opaque surface: dC = (((sC - dC) * sA + sC) >> 8) + dC, skipped for sA == 0
    const int a = (int)(src >> 24);
    if (a == 0) return dst;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = (int)((src >> (8 * c)) & 0xFFu), dc = (int)((dst >> (8 * c)) & 0xFFu);
        const int v = (((sc - dc) * a + sc) >> 8) + dc;
        out |= ((uint32_t)v & 0xFFu) << (8 * c);
    }
    return out;
}

// ---- clouds and waves (cloud_and_wave.py): renderer-owned state outside the 44 words, driven by the env stream ----
// get_all_image's `Cloud(self.np_random)` x 10 (pikazoo_env.py:475-477 -> cloud_and_wave.py:15-19) and Wave() (:42-50)
__global__ __launch_bounds__(kLanes) void scenery_init_kernel(int32_t* scenery, int32_t* state, int64_t n, int64_t stride,
                                                              const pz_config cfg)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    const RngId id = make_rng_id(cfg, i);
    uint32_t rng = (uint32_t)state[(int64_t)PZ_E_RNG_DRAW_COUNTER * stride + i];
    auto word = [&](int w) -> int32_t& { return scenery[(int64_t)w * stride + i]; };
    for (int c = 0; c < 10; ++c) {
        word(4 * c + 0) = -68 + rng_integers(id, rng, 432u + 68u);
        word(4 * c + 1) = rng_integers(id, rng, 152u);
        word(4 * c + 2) = 1 + rng_integers(id, rng, 2u);
        word(4 * c + 3) = rng_integers(id, rng, 11u);
    }
    word(40) = 0;
    word(41) = 2;
    for (int k = 0; k < 27; ++k) word(42 + k) = 314;
    for (int k = 69; k < PZ_SCENERY_WORDS; ++k) word(k) = 0;  // no punch effect (physics.py:275), nothing remembered
    state[(int64_t)PZ_E_RNG_DRAW_COUNTER * stride + i] = (int32_t)rng;
}

// The punch effect's two ball attributes (physics.py:240-242,275), re-derived after a frame from the state it left:
// the ball-world step sets them on a ground touch (:427-430 -- exactly the frames that end with round_ended set), the
// ball-player collision on a power hit (:628-632 -- a player's collision flag rising while its state is 2; player 1
// first, player 2's overrides it like in physics_engine :319-335).
__global__ __launch_bounds__(kLanes) void scenery_track_kernel(int32_t* scenery, const int32_t* state, int64_t n,
                                                               int64_t stride, const pz_config cfg, int resync)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    auto word = [&](int w) -> int32_t& { return scenery[(int64_t)w * stride + i]; };
    auto st = [&](int f) { return state[(int64_t)f * stride + i]; };
    const int coll1 = st(PZ_P_IS_COLLISION_WITH_BALL_HAPPENED), coll2 = st(PZ_P_WORDS + PZ_P_IS_COLLISION_WITH_BALL_HAPPENED);
    const int ended = st(PZ_E_GAME_ENDED), round_ended = st(PZ_E_ROUND_ENDED);
    if (resync) {
        word(69) = 0;
    } else if (!(word(73) != 0 && cfg.auto_reset == 0)) {  // (a finished game without auto-reset was not stepped)
        int radius = word(69), y = word(70);
        if (word(74) != 0) radius = 0;  // the frame started a new round: Ball.initialize_for_new_round (:274-275)
        if (round_ended != 0) {
            radius = kBallRadius;
            y = kBallGroundY + kBallRadius;
        }
        if ((coll1 != 0 && word(71) == 0 && st(PZ_P_STATE) == 2) ||
            (coll2 != 0 && word(72) == 0 && st(PZ_P_WORDS + PZ_P_STATE) == 2)) {
            radius = kBallRadius;
            y = st(PZ_B_Y);
        }
        word(69) = radius;
        word(70) = y;
    }
    word(71) = coll1;
    word(72) = coll2;
    word(73) = ended;
    word(74) = round_ended;
}

// cloud_and_wave_engine (cloud_and_wave.py:53-78) for the games about to be drawn: one thread per frame
__global__ __launch_bounds__(kLanes) void scenery_tick_kernel(int32_t* scenery, int32_t* state, int64_t n, int64_t stride,
                                                              const pz_config cfg, const int32_t* lanes, int64_t m)
{
    const int64_t j = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (j >= m) return;
    const int64_t i = lanes != nullptr ? (int64_t)lanes[j] : j;
    if (i < 0 || i >= n) return;
    const RngId id = make_rng_id(cfg, i);
    uint32_t rng = (uint32_t)state[(int64_t)PZ_E_RNG_DRAW_COUNTER * stride + i];
    auto word = [&](int w) -> int32_t& { return scenery[(int64_t)w * stride + i]; };
    for (int c = 0; c < 10; ++c) {
        int x = word(4 * c) + word(4 * c + 2);
        if (x > 432) {
            x = -68;
            word(4 * c + 1) = rng_integers(id, rng, 152u);
            word(4 * c + 2) = 1 + rng_integers(id, rng, 2u);
        }
        word(4 * c) = x;
        const int turn = word(4 * c + 3) + 1;
        word(4 * c + 3) = turn >= 11 ? turn - 11 : turn;
    }
    int vc = word(40) + word(41);
    if (vc > 32) {
        vc = 32;
        word(41) = -1;
    } else if (vc < 0 && word(41) < 0) {
        word(41) = 2;
        vc = -rng_integers(id, rng, 40u);
    }
    word(40) = vc;
    for (int k = 0; k < 27; ++k) word(42 + k) = 314 - vc + rng_integers(id, rng, 3u);
    if (word(69) > 0) word(69) -= 2;  // draw_ball counts the punch effect down itself (pikazoo_env.py:292-293)
    state[(int64_t)PZ_E_RNG_DRAW_COUNTER * stride + i] = (int32_t)rng;
}

__global__ __launch_bounds__(256) void render_kernel(const int32_t* __restrict__ state, int64_t n, int64_t stride,
                                                     const int32_t* __restrict__ lanes, const uint32_t* __restrict__ atlas,
                                                     const pz_sprite* __restrict__ sprites,
                                                     const uint32_t* __restrict__ background,
                                                     const int32_t* __restrict__ scenery, uint8_t* __restrict__ frames)
{
    constexpr int kGroupsPerRow = PZ_FRAME_WIDTH / 4;
    const int64_t game = lanes != nullptr ? (int64_t)lanes[blockIdx.y] : (int64_t)blockIdx.y;  // wave-uniform
    const int g = (int)(blockIdx.x * 256 + threadIdx.x);
    if (g >= kGroupsPerRow * PZ_FRAME_HEIGHT || game < 0 || game >= n) return;
    const int row = g / kGroupsPerRow, col = (g - row * kGroupsPerRow) * 4;
    auto word = [&](int f) { return state[(int64_t)f * stride + game]; };

    // the draw list, in the order of raw_env.draw (:250-255): twelve fixed slots (sprite < 0: not drawn), so that the
    // list stays in registers
    Blit list[12];
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // draw_player :257-275
        const int c0 = p * PZ_P_WORDS;
        const int st = word(c0 + PZ_P_STATE), fr = word(c0 + PZ_P_FRAME_NUMBER), dive = word(c0 + PZ_P_DIVING_DIRECTION);
        const int idx = st < 4 ? 5 * st + fr : (st == 4 ? 17 + fr : 18 + 5 * (st - 5) + fr);  // :63-68
        const bool diving = st == 3 || st == 4;
        const bool flip = p == 0 ? (diving && dive == -1) : !(diving && dive == 1);  // :263-264
        list[p] = Blit{PZ_SPRITE_PIKACHU + min(max(idx, 0), 27), word(c0 + PZ_P_X), word(c0 + PZ_P_Y), flip ? 3 : 1};
    }
    list[2] = Blit{PZ_SPRITE_SHADOW, word(PZ_P_X), 273, 1};                     // :277-278
    list[3] = Blit{PZ_SPRITE_SHADOW, word(PZ_P_WORDS + PZ_P_X), 273, 1};
    const int rotation = min(max(word(PZ_B_FINE_ROTATION) / 10, 0), 5);         // physics.py:388
    list[4] = Blit{PZ_SPRITE_BALL + rotation, word(PZ_B_X), word(PZ_B_Y), 1};   // draw_ball :282-290
    list[5] = Blit{PZ_SPRITE_SHADOW, word(PZ_B_X), 273, 1};
    const bool power = word(PZ_B_IS_POWER_HIT) != 0;
    list[6] = Blit{power ? PZ_SPRITE_BALL_HYPER : -1, word(PZ_B_PREVIOUS_X), word(PZ_B_PREVIOUS_Y), 1};
    list[7] = Blit{power ? PZ_SPRITE_BALL_TRAIL : -1, word(PZ_B_PREVIOUS_PREVIOUS_X), word(PZ_B_PREVIOUS_PREVIOUS_Y), 1};
    const int s1 = word(PZ_E_SCORE_P1), s2 = word(PZ_E_SCORE_P2);                // :327-336 (top-left blits)
    list[8] = Blit{s1 >= 10 ? PZ_SPRITE_NUMBER + 1 : -1, 14, 10, 0};
    list[9] = Blit{PZ_SPRITE_NUMBER + s1 % 10, 14 + 32, 10, 0};
    list[10] = Blit{s2 >= 10 ? PZ_SPRITE_NUMBER + 1 : -1, PZ_FRAME_WIDTH - 32 - 32 - 14, 10, 0};
    list[11] = Blit{PZ_SPRITE_NUMBER + s2 % 10, PZ_FRAME_WIDTH - 32 - 32 - 14 + 32, 10, 0};

    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) px[k] = background[row * PZ_FRAME_WIDTH + col + k];
    if (scenery != nullptr) {  // draw_clouds_and_wave :351-364, behind the players
        auto sc = [&](int w) { return scenery[(int64_t)w * stride + game]; };
        const pz_sprite cloud = sprites[PZ_SPRITE_CLOUD], wave = sprites[PZ_SPRITE_WAVE];
        for (int c = 0; c < 10; ++c) {
            const int turn = sc(4 * c + 3);
            const int d = 5 - abs(turn - 5);                                    // Cloud.size_diff :22-23
            const int x0 = sc(4 * c) - d, y0 = sc(4 * c + 1) - d;               // sprite_top_left_point :26-31
            const int w = cloud.width + 2 * d, h = cloud.height + 2 * d;         // sprite_width / height :34-39
            const int dy = row - y0;
            if ((unsigned)dy >= (unsigned)h || col + 3 < x0 || col >= x0 + w) continue;
            const int sy = dy * cloud.height / h;  // pygame.transform.scale: source = floor(k * source size / scaled size)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int dx = col + k - x0;
                if ((unsigned)dx < (unsigned)w)
                    px[k] = blend_over(px[k], atlas[cloud.offset + sy * cloud.width + dx * cloud.width / w]);
            }
        }
        const int tile = col / wave.width;  // the four pixels of a thread share one 16-pixel wave tile
        if (tile < 27) {
            const int sy = row - sc(42 + tile);
            if ((unsigned)sy < (unsigned)wave.height) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    px[k] = blend_over(px[k], atlas[wave.offset + sy * wave.width + (col + k - tile * wave.width)]);
            }
        }
    }
    auto blit = [&](const Blit& bl) {
        if (bl.sprite < 0) return;
        const pz_sprite sp = sprites[bl.sprite];
        const bool centred = (bl.flip & 1) != 0, mirrored = (bl.flip & 2) != 0;
        const int x0 = centred ? bl.x0 - sp.width / 2 : bl.x0;    // blit_center :40-43
        const int y0 = centred ? bl.y0 - sp.height / 2 : bl.y0;
        const int sy = row - y0;
        if ((unsigned)sy >= (unsigned)sp.height || col + 3 < x0 || col >= x0 + sp.width) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sx = col + k - x0;
            if ((unsigned)sx < (unsigned)sp.width)
                px[k] = blend_over(px[k], atlas[sp.offset + sy * sp.width + (mirrored ? sp.width - 1 - sx : sx)]);
        }
    };
#pragma unroll
    for (int b = 0; b < 8; ++b) blit(list[b]);  // players, shadows, ball, hyper ball, trail
    if (scenery != nullptr) {  // the punch effect, last blit of draw_ball (:292-294): ball_punch scaled to 2r x 2r
        const int r = scenery[(int64_t)69 * stride + game];
        if (r > 0) {
            const pz_sprite sp = sprites[PZ_SPRITE_BALL_PUNCH];
            const int size = 2 * r;
            const int x0 = word(PZ_B_PUNCH_EFFECT_X) - r, y0 = scenery[(int64_t)70 * stride + game] - r;  // blit_center
            const int dy = row - y0;
            if ((unsigned)dy < (unsigned)size && col + 3 >= x0 && col < x0 + size) {
                const int sy = dy * sp.height / size;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int dx = col + k - x0;
                    if ((unsigned)dx < (unsigned)size)
                        px[k] = blend_over(px[k], atlas[sp.offset + sy * sp.width + dx * sp.width / size]);
                }
            }
        }
    }
#pragma unroll
    for (int b = 8; b < 12; ++b) blit(list[b]);  // score boards
    // 4 x RGB = 12 bytes = 3 dwords
    const uint32_t r0 = (px[0] & 0xFFFFFFu) | (px[1] << 24);
    const uint32_t r1 = ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16);
    const uint32_t r2 = ((px[2] >> 16) & 0xFFu) | (px[3] << 8);
    uint32_t* out = reinterpret_cast<uint32_t*>(frames + ((int64_t)blockIdx.y * PZ_FRAME_HEIGHT + row) * (PZ_FRAME_WIDTH * 3) +
                                                col * 3);
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
}


// ---- host side ---------------------------------------------------------------------------------
// Buffer descriptors address with 32-bit byte offsets: one launch handles at most this many games
// (4 GiB / 176 B of state per game); larger jobs are sharded by the caller (env_id_base).
constexpr int64_t kMaxLanesPerLaunch = (int64_t)0xFFFFFFFFu / (PZ_STATE_WORDS * 4);
// batches from this size on are HBM-bound and use the changed-only write-back
// Kernel selection by batch size (interleaved A/B on MI355X, tools/ab.py; us per pz_step launch):
//   human-vs-human, pair kernel | single-wave kernel, both with the changed-only write-back:
//       65 536: 7.58 | 8.01    131 072: 11.1 | 11.5    262 144: 22.4 | 23.8    294 912: 25.3 | 26.5
//      524 288: 47.2 | 46.5    1 048 576: 93.9 | 90.6      (single-wave without changed-only at 65 536: 8.52)
//   player 2 = computer, scout kernel | single-wave changed-only:   262 144: 36.5 | 38.2    524 288: 70.3 | 66.4
constexpr int64_t kTwoWaveMaxLanes = 393216;  // below: two waves per workgroup (pair kernel / scout)
//   the changed-only write-back also pays in the scout kernel (65 536: 14.3 | 14.65 without, 262 144: 35.6 | 36.6)
//   and is used by every launch that writes the state back after ONE frame; a trajectory launch writes it once
//   per k frames, where the plain write-back is always right.

static inline bool is_packed(const pz_config& cfg) { return (cfg.packed_state & 1) != 0; }
// no fused wrapper, no episode statistics, raw integer rows: what the PLAIN k-frame kernels are compiled for
static inline bool is_plain(const StepArgs& a)
{
    return a.cfg.simplify_action == 0 && a.cfg.ballpos_reward == 0 && a.cfg.normal_state_mode == 0 &&
           a.cfg.normalize_obs != 1 && (a.cfg.episode_stats_mode == 0 || a.episode_stats == nullptr);
}
static inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

static int check_common(const void* state, int64_t n, int64_t stride, const pz_config* cfg)
{
    if (state == nullptr || cfg == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n || stride > kMaxLanesPerLaunch) return PZ_E_SIZE;
    if (cfg->winning_score < 1 || cfg->serve_mode < 0 || cfg->serve_mode > 2 || cfg->normal_state_mode < 0 ||
        cfg->normal_state_mode > 2 || cfg->episode_stats_mode < 0 || cfg->episode_stats_mode > 2 ||
        cfg->normalize_obs < 0 || cfg->normalize_obs > 2)
        return PZ_E_CONFIG;
    if (cfg->packed_state != 0 && cfg->packed_state != 1) return PZ_E_CONFIG;
    if (cfg->action_format < PZ_ACT_I32 || cfg->action_format > PZ_ACT_I16) return PZ_E_CONFIG;
    if (is_packed(*cfg)) {
        if (cfg->winning_score > 65535) return PZ_E_CONFIG;  // the scores are 16-bit fields
        if (misaligned16(state)) return PZ_E_ALIGN;
    }
    return PZ_OK;
}

static inline unsigned int blocks_for(int64_t n, int per) { return (unsigned int)((n + per - 1) / per); }

// Diagnostic builds only (pz_diagnostic.hpp: diag::kSubset, bits 16-29 of PZ_DIAGNOSTIC_BUILD; 0 in the product = keep
// everything): a subset of the step kernels' instantiations -- a variant that is only ever timed on one configuration
// builds in a fraction of the 100 s the full library takes.  bits 0-3: launch modes kActions / kRandom / kRollout / kTape;
// 4: the packed state format; 5: int16 rows; 6: human vs human; 7: player 2 = computer; 8: the other computer-player
// combinations; 9: step_pair_kernel; 10: rollout_pair_kernel; 11: step_kernel; 12: the PLAIN forms of the k-frame
// kernels; 13: their generic forms.  A launch that was left out returns PZ_E_CONFIG.
enum DevFamily { kDevPair = 9, kDevRolloutPair = 10, kDevSingle = 11 };
constexpr bool dev_keep(int family, int mode, bool ai1, bool ai2, bool packed, bool obs16, bool plain = false)
{
    constexpr unsigned m = diag::kSubset;
    if (m == 0u) return true;
    const bool players = (!ai1 && !ai2) ? (m >> 6) & 1u : ((!ai1 && ai2) ? (m >> 7) & 1u : (m >> 8) & 1u);
    const bool form = (mode != kRollout && mode != kTape) || ((m >> (plain ? 12 : 13)) & 1u);
    return ((m >> family) & 1u) && ((m >> mode) & 1u) && players && (!packed || ((m >> 4) & 1u)) &&
           (!obs16 || ((m >> 5) & 1u)) && form;
}
#define PZ_KEEP(...) pz::dev_keep(__VA_ARGS__)

// one step_kernel instantiation per player configuration; the trajectory modes also per observation row format
// (OBS16, compile-time there: see TrajOut::flush)
template <int MODE, bool SPARSE, int SCOUT, bool PACKED, bool OBS16>
static int launch_step_players(const StepArgs& a, hipStream_t stream)
{
    const dim3 grid(blocks_for(a.n, kLanes)), block(SCOUT != kNoScout ? 2 * kLanes : kLanes);
    const bool ai1 = a.cfg.p1_computer != 0, ai2 = a.cfg.p2_computer != 0;
    // the k-frame launches of a configuration without fused wrappers / statistics: the PLAIN instantiation (effective_args)
    // -- but for the human-vs-human rollout on one wave: that launch runs at the write ceiling of the two observation
    // tensors either way, and its leaner PLAIN form (6 SGPR spills instead of 53) measured 0.6 - 3.4 % SLOWER there,
    // on every box and in every position of the interleaved rounds (us per frame, k = 32: 2.91 vs 2.81-2.84; k = 128:
    // 2.73 vs 2.65; store order within the frame: no effect; profiles/r04_experiments/ab_rollout_hh_*), while the tape
    // kernel and every computer-player launch gain 1-2 % from theirs
    constexpr bool kHasPlain = (MODE == kRollout || MODE == kTape) && SCOUT == kNoScout && !PACKED;
    const bool plain = kHasPlain && is_plain(a);
#define PZ_PLAIN_HERE(A1, A2) (kHasPlain && !(kHhRolloutGeneric && MODE == kRollout && !(A1) && !(A2)))
#define PZ_LAUNCH_SINGLE_AS(A1, A2, PLAIN)                                                                                \
    do {                                                                                                                  \
        if constexpr (PZ_KEEP(kDevSingle, MODE, A1, A2, PACKED, OBS16, PLAIN))                                             \
            hipLaunchKernelGGL((step_kernel<A1, A2, MODE, SPARSE, SCOUT, PACKED, OBS16, PLAIN>), grid, block, 0, stream,   \
                               PZ_HOT_ARGS(a), a);                                                                        \
        else                                                                                                              \
            return PZ_E_CONFIG;                                                                                           \
    } while (0)
#define PZ_LAUNCH_SINGLE(A1, A2)                                                                                          \
    do {                                                                                                                  \
        if constexpr (PZ_PLAIN_HERE(A1, A2)) {                                                                            \
            if (plain) {                                                                                                  \
                PZ_LAUNCH_SINGLE_AS(A1, A2, true);                                                                        \
                break;                                                                                                    \
            }                                                                                                             \
        }                                                                                                                 \
        PZ_LAUNCH_SINGLE_AS(A1, A2, false);                                                                               \
    } while (0)
    if (ai1 && ai2)
        PZ_LAUNCH_SINGLE(true, true);
    else if (ai1)
        PZ_LAUNCH_SINGLE(true, false);
    else if (ai2)
        PZ_LAUNCH_SINGLE(false, true);
    else if constexpr (SCOUT == kNoScout)  // (a scout wave only ever serves a computer player)
        PZ_LAUNCH_SINGLE(false, false);
#undef PZ_LAUNCH_SINGLE
#undef PZ_LAUNCH_SINGLE_AS
#undef PZ_PLAIN_HERE
    return (int)hipGetLastError();
}

template <int MODE, bool SPARSE, bool PACKED = false, int SCOUT = kNoScout>
static int launch_step_ai(const StepArgs& a, hipStream_t stream)
{
    if constexpr (MODE == kRollout || MODE == kTape) {
        if (a.cfg.normalize_obs == 2) return launch_step_players<MODE, SPARSE, SCOUT, PACKED, true>(a, stream);
    }
    return launch_step_players<MODE, SPARSE, SCOUT, PACKED, false>(a, stream);
}

template <bool AI1, bool AI2, bool RANDOM = false>
static int launch_pair(const StepArgs& a, hipStream_t stream)
{
    const dim3 grid(blocks_for(a.n, kLanes)), block(2 * kLanes);
    constexpr int kMode = RANDOM ? kRandom : kActions;
    if (is_packed(a.cfg)) {
        if constexpr (PZ_KEEP(kDevPair, kMode, AI1, AI2, true, false))
            hipLaunchKernelGGL((step_pair_kernel<AI1, AI2, true, RANDOM>), grid, block, 0, stream, PZ_HOT_ARGS(a), a);
        else
            return PZ_E_CONFIG;
    } else {
        if constexpr (PZ_KEEP(kDevPair, kMode, AI1, AI2, false, false))
            hipLaunchKernelGGL((step_pair_kernel<AI1, AI2, false, RANDOM>), grid, block, 0, stream, PZ_HOT_ARGS(a), a);
        else
            return PZ_E_CONFIG;
    }
    return (int)hipGetLastError();
}

template <int MODE>
static int launch_step(const StepArgs& a, hipStream_t stream)
{
    const bool ai1 = a.cfg.p1_computer != 0, ai2 = a.cfg.p2_computer != 0;
    // with the power-hit table the six candidate flights of a deciding player are one gather: no scout wave is needed, and
    // the single frame splits by player.  The landing table is optional on top of it (pz_flight_tables in the header):
    // without it the landing point is predicted in the kernel (predict_landing_x<true>: the closed-form fast-forward)
    const bool tables = a.tables.power_hit != nullptr;
    if constexpr (!diag::kNoPairKernel) {
        // (the packed format: at every size -- 524 288 games, us per launch, pair | single wave: human 30.25 | 30.13,
        // player 2 = computer 38.6 | 40.6; 1 048 576 human 56.2 | 56.4)
        if (MODE == kActions && (a.n < kTwoWaveMaxLanes || is_packed(a.cfg)) && (tables || !(ai1 || ai2))) {
            if (ai1 && ai2) return launch_pair<true, true>(a, stream);
            if (ai1) return launch_pair<true, false>(a, stream);
            if (ai2) return launch_pair<false, true>(a, stream);
            return launch_pair<false, false>(a, stream);
        }
        // one frame of the on-device random policy is the same launch with the policy's Philox block in place of the two
        // action loads (65 536 games: 8.2 -> 7.0 us against the single-wave kernel)
        if (MODE == kRandom && a.k == 1 && (a.n < kTwoWaveMaxLanes || is_packed(a.cfg)) && (tables || !(ai1 || ai2))) {
            if (ai1 && ai2) return launch_pair<true, true, true>(a, stream);
            if (ai1) return launch_pair<true, false, true>(a, stream);
            if (ai2) return launch_pair<false, true, true>(a, stream);
            return launch_pair<false, false, true>(a, stream);
        }
    }
    // pz_rollout_random / pz_step_many with a computer player on the flight tables: two waves per 64 games below the size switch
    // (interleaved A/B, us per frame at k = 32: 3.49 vs 4.34 on one wave; human vs human the single wave is at the
    // write ceiling already: 3.62 on two waves -- player 1's writing all outputs -- vs 3.63 on one, 3.76 with the
    // outputs split between the waves)
    const bool hh_pair = !(ai1 || ai2) && (kHhPairRollout == 2 || (kHhPairRollout == 1 && a.cfg.normalize_obs == 2));
    if constexpr ((MODE == kRollout || MODE == kTape) && !diag::kNoRolloutPair) if (a.n < kTwoWaveMaxLanes && ((tables && (ai1 || ai2)) || hh_pair)) {
        const dim3 grid(blocks_for(a.n, kLanes)), block(2 * kLanes);
        const bool packed = is_packed(a.cfg), obs16 = a.cfg.normalize_obs == 2, plain = is_plain(a);
#define PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, PK, O16, PLAIN)                                                                 \
    do {                                                                                                                  \
        if constexpr (PZ_KEEP(kDevRolloutPair, MODE, A1, A2, PK, O16, PLAIN))                                              \
            hipLaunchKernelGGL((rollout_pair_kernel<A1, A2, MODE, PK, O16, PLAIN>), grid, block, 0, stream,                \
                               PZ_HOT_ARGS(a), a);                                                                        \
        else                                                                                                              \
            return PZ_E_CONFIG;                                                                                           \
    } while (0)
#define PZ_LAUNCH_ROLLOUT_PAIR(A1, A2)                                                                                    \
    do {                                                                                                                  \
        if (packed && obs16)                                                                                              \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, true, true, false);                                                         \
        else if (packed)                                                                                                  \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, true, false, false);                                                        \
        else if (obs16 && plain)                                                                                          \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, false, true, true);                                                         \
        else if (obs16)                                                                                                   \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, false, true, false);                                                        \
        else if (plain)                                                                                                   \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, false, false, true);                                                        \
        else                                                                                                              \
            PZ_LAUNCH_ROLLOUT_PAIR_AS(A1, A2, false, false, false);                                                       \
    } while (0)
        if (ai1 && ai2)
            PZ_LAUNCH_ROLLOUT_PAIR(true, true);
        else if (ai1)
            PZ_LAUNCH_ROLLOUT_PAIR(true, false);
        else if (ai2)
            PZ_LAUNCH_ROLLOUT_PAIR(false, true);
        else
            PZ_LAUNCH_ROLLOUT_PAIR(false, false);
#undef PZ_LAUNCH_ROLLOUT_PAIR
#undef PZ_LAUNCH_ROLLOUT_PAIR_AS
        return (int)hipGetLastError();
    }
    // the packed format: pair kernel above, else one wave per workgroup (a computer player without tables computes its
    // flights in that wave: no scout)
    if (is_packed(a.cfg)) return launch_step_ai<MODE, false, true>(a, stream);
    if constexpr (!diag::kNoScoutWave) {
        if (a.n < kTwoWaveMaxLanes && !tables) {  // a computer player is present (else: pair kernel above)
            constexpr int kScout = MODE == kActions ? kScoutLoads : kScoutPosted;
            constexpr bool kSparse = MODE == kActions || MODE == kRandom;
            if (ai1 || ai2) return launch_step_ai<MODE, kSparse, false, kScout>(a, stream);
        }
    }
    return launch_step_ai<MODE, MODE == kActions || MODE == kRandom>(a, stream);
}

// the look-ups load whole dwords / 16-byte rows
static inline bool tables_misaligned(const pz_flight_tables* t)
{
    return t != nullptr && (misaligned16(t->power_hit) || (reinterpret_cast<uintptr_t>(t->landing) & 3u) != 0);
}

static pz_flight_tables tables_of(const pz_flight_tables* t)
{
    return t != nullptr ? *t : pz_flight_tables{nullptr, nullptr};
}

}  // namespace pz

using namespace pz;

extern "C" {

int pz_abi_version(void) { return PZ_ABI_VERSION; }
int pz_state_words(void) { return PZ_STATE_WORDS; }
int pz_obs_dim(void) { return PZ_OBS_DIM; }
int pz_config_bytes(void) { return (int)sizeof(pz_config); }

#ifndef PZ_BUILD_ID
#define PZ_BUILD_ID "unstamped"
#endif
// stored behind a marker so that build.py can read the id from the file without mapping the library; a diagnostic
// build (pz_diagnostic.hpp) never carries a product id, whatever it was compiled with: _native.load() refuses it
static const char kBuildIdRecord[] = "pz_build_id:" PZ_BUILD_ID;
const char* pz_build_id(void) { return diag::kDiagnosticBuild ? "diagnostic" : kBuildIdRecord + 12; }

int64_t pz_flight_table_bytes(int32_t which)
{
    return which == 0 ? kFtLandingBytes : (which == 1 ? kFtHitEntries * 16 : 0);
}

int pz_build_flight_tables(uint16_t* landing, uint16_t* power_hit, void* stream)
{
    if (landing == nullptr && power_hit == nullptr) return PZ_E_NULL;
    if (misaligned16(power_hit) || (reinterpret_cast<uintptr_t>(landing) & 3u) != 0) return PZ_E_ALIGN;
    if (landing != nullptr)
        hipLaunchKernelGGL(build_landing_table_kernel, dim3(blocks_for(kFtLandingEntries, 256)), dim3(256), 0,
                           (hipStream_t)stream, landing);
    if (power_hit != nullptr)
        hipLaunchKernelGGL(build_power_hit_table_kernel, dim3(blocks_for(kFtHitEntries * 8, 256)), dim3(256), 0,
                           (hipStream_t)stream, power_hit);
    return (int)hipGetLastError();
}

const char* pz_error_string(int code)
{
    switch (code) {
        case PZ_OK: return "ok";
        case PZ_E_NULL: return "required pointer is NULL";
        case PZ_E_SIZE: return "bad size (n < 0, stride < n, k < 1, or more than 24 403 223 games in one launch)";
        case PZ_E_CONFIG: return "pz_config field out of range";
        case PZ_E_ALIGN: return "buffer is not 16-byte aligned (observations, packed state, power_hit table; landing table: 4)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown pikazoo error";
    }
}

int pz_init(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (n == 0) return PZ_OK;
    if (is_packed(*cfg))
        hipLaunchKernelGGL(init_kernel<true>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                           stride, *cfg);
    else
        hipLaunchKernelGGL(init_kernel<false>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                           stride, *cfg);
    return (int)hipGetLastError();
}

int pz_reset(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const uint8_t* mask, int32_t* obs_p1,
             int32_t* obs_p2, void* episode_stats, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    if (is_packed(*cfg))
        hipLaunchKernelGGL(reset_kernel<true>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                           stride, *cfg, mask, obs_p1, obs_p2, episode_stats);
    else
        hipLaunchKernelGGL(reset_kernel<false>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                           stride, *cfg, mask, obs_p1, obs_p2, episode_stats);
    return (int)hipGetLastError();
}

int pz_observe(const int32_t* state, int64_t n, int64_t stride, int32_t normalize, int32_t packed, int32_t* obs_p1,
               int32_t* obs_p2, void* stream)
{
    if (state == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n || stride > kMaxLanesPerLaunch) return PZ_E_SIZE;
    if (normalize < 0 || normalize > 2) return PZ_E_CONFIG;
    if (misaligned16(obs_p1) || misaligned16(obs_p2) || (packed != 0 && misaligned16(state))) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    if (packed != 0)
        hipLaunchKernelGGL(observe_kernel<true>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state,
                           n, stride, (int)normalize, obs_p1, obs_p2);
    else
        hipLaunchKernelGGL(observe_kernel<false>, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state,
                           n, stride, (int)normalize, obs_p1, obs_p2);
    return (int)hipGetLastError();
}

int64_t pz_packed_state_bytes(int64_t stride) { return stride < 0 ? 0 : stride * PZ_PACKED_BYTES_PER_GAME; }

int pz_pack_state(const int32_t* state, int64_t n, int64_t stride, void* packed, int64_t packed_stride, int64_t* misfits,
                  void* stream)
{
    if (state == nullptr || packed == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n || packed_stride < n || stride > kMaxLanesPerLaunch || packed_stride > kMaxLanesPerLaunch)
        return PZ_E_SIZE;
    if (misaligned16(packed)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(pack_state_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                       stride, packed, packed_stride, reinterpret_cast<unsigned long long*>(misfits));
    return (int)hipGetLastError();
}

int pz_probe_write(void* a, void* b, int64_t bytes, void* stream)
{
    if (a == nullptr && b == nullptr) return PZ_E_NULL;
    if (misaligned16(a) || misaligned16(b)) return PZ_E_ALIGN;
    const int64_t frames = bytes / kProbeFrameBytes;
    if (frames < 1 || frames > 65536) return PZ_E_SIZE;
    hipLaunchKernelGGL(probe_write_kernel, dim3(kProbeSpans), dim3(64), 0, (hipStream_t)stream, (char*)a, (char*)b,
                       (int32_t)frames);
    return (int)hipGetLastError();
}

int64_t pz_probe_frame_bytes(void) { return kProbeFrameBytes; }


int pz_count_packed_misfits(const void* packed, int64_t n, int64_t packed_stride, int64_t* flagged, void* stream)
{
    if (packed == nullptr || flagged == nullptr) return PZ_E_NULL;
    if (n < 0 || packed_stride < n || packed_stride > kMaxLanesPerLaunch) return PZ_E_SIZE;
    if (misaligned16(packed)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(count_misfits_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, packed, n,
                       packed_stride, reinterpret_cast<unsigned long long*>(flagged));
    return (int)hipGetLastError();
}

int pz_unpack_state(const void* packed, int64_t n, int64_t packed_stride, int32_t* state, int64_t stride, int64_t* flagged,
                    void* stream)
{
    if (state == nullptr || packed == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n || packed_stride < n || stride > kMaxLanesPerLaunch || packed_stride > kMaxLanesPerLaunch)
        return PZ_E_SIZE;
    if (misaligned16(packed)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(unpack_state_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, packed, n,
                       packed_stride, state, stride, reinterpret_cast<unsigned long long*>(flagged));
    return (int)hipGetLastError();
}

int pz_step(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const void* act_p1,
            const void* act_p2, int32_t* obs_p1, int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated,
            void* episode_stats, const pz_flight_tables* tables, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!act_p1 || !act_p2 || !obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    if (tables_misaligned(tables)) return PZ_E_ALIGN;
    StepArgs a{state,  n,          stride,        act_p1,  act_p2, 0, 0, 1, nullptr, obs_p1, obs_p2, rew_p1,
               rew_p2, terminated, episode_stats, nullptr, tables_of(tables), *cfg};
    return launch_step<kActions>(a, (hipStream_t)stream);
}

// pz_step with its arguments prepared once: the block holds the StepArgs pz_step would build
struct BoundStep {
    uint64_t magic;
    StepArgs args;
};
constexpr uint64_t kBoundMagic = 0x70696b617a6f6f36ull;  // "pikazoo6"

int64_t pz_step_bound_bytes(void) { return (int64_t)sizeof(BoundStep); }

int pz_step_bind(void* bound, int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, int32_t* obs_p1,
                 int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated, void* episode_stats,
                 const pz_flight_tables* tables)
{
    if (bound == nullptr) return PZ_E_NULL;
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (misaligned16(obs_p1) || misaligned16(obs_p2) || tables_misaligned(tables)) return PZ_E_ALIGN;
    BoundStep* b = static_cast<BoundStep*>(bound);
    b->args = StepArgs{state,  n,          stride,        nullptr, nullptr, 0, 0, 1, nullptr, obs_p1, obs_p2, rew_p1,
                       rew_p2, terminated, episode_stats, nullptr, tables_of(tables), *cfg};
    b->magic = kBoundMagic;
    return PZ_OK;
}

int pz_step_bound(const void* bound, const void* act_p1, const void* act_p2, void* stream)
{
    const BoundStep* b = static_cast<const BoundStep*>(bound);
    if (b == nullptr || !act_p1 || !act_p2) return PZ_E_NULL;
    if (b->magic != kBoundMagic) return PZ_E_CONFIG;  // not a block pz_step_bind has filled
    if (b->args.n == 0) return PZ_OK;
    StepArgs a = b->args;
    a.act_p1 = act_p1;
    a.act_p2 = act_p2;
    return launch_step<kActions>(a, (hipStream_t)stream);
}

int pz_step_random(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, uint64_t action_seed, uint64_t t0,
                   int32_t k, int32_t* obs_p1, int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated,
                   void* episode_stats, int64_t* episodes_done, const pz_flight_tables* tables, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (k < 1) return PZ_E_SIZE;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (tables_misaligned(tables)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    StepArgs a{state,  n,          stride,        nullptr, nullptr, action_seed, t0, k, nullptr, obs_p1, obs_p2, rew_p1,
               rew_p2, terminated, episode_stats, reinterpret_cast<unsigned long long*>(episodes_done), tables_of(tables),
               *cfg};
    return launch_step<kRandom>(a, (hipStream_t)stream);
}

int pz_rollout_random(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, uint64_t action_seed,
                      uint64_t t0, int32_t k, int32_t* actions, int32_t* obs_p1, int32_t* obs_p2, void* rew_p1,
                      void* rew_p2, uint8_t* terminated, void* episode_stats, int64_t* episodes_done,
                      const pz_flight_tables* tables, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (k < 1) return PZ_E_SIZE;
    // every frame's [n][35] slab must keep the 16-byte alignment of the vector stores: n * 140 % 16 == 0
    if (misaligned16(obs_p1) || misaligned16(obs_p2) || (k > 1 && (n & (cfg->normalize_obs == 2 ? 7 : 3)) != 0))
        return PZ_E_ALIGN;
    if (tables_misaligned(tables)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    StepArgs a{state,  n,          stride,        nullptr, nullptr, action_seed, t0, k, actions, obs_p1, obs_p2, rew_p1,
               rew_p2, terminated, episode_stats, reinterpret_cast<unsigned long long*>(episodes_done), tables_of(tables),
               *cfg};
    return launch_step<kRollout>(a, (hipStream_t)stream);
}

int pz_step_many(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const void* actions, int32_t k,
                 int32_t* obs_p1, int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated,
                 void* episode_stats, int64_t* episodes_done, const pz_flight_tables* tables, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!actions || !obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (k < 1) return PZ_E_SIZE;
    // the tape is parked from int32 rows (the header says how a caller brings another element type)
    if (cfg->action_format != PZ_ACT_I32) return PZ_E_CONFIG;
    if (misaligned16(obs_p1) || misaligned16(obs_p2) || (k > 1 && (n & (cfg->normalize_obs == 2 ? 7 : 3)) != 0))
        return PZ_E_ALIGN;
    if (tables_misaligned(tables)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    StepArgs a{state,  n,          stride,        actions, nullptr, 0, 0, k, nullptr, obs_p1, obs_p2, rew_p1,
               rew_p2, terminated, episode_stats, reinterpret_cast<unsigned long long*>(episodes_done), tables_of(tables),
               *cfg};
    return launch_step<kTape>(a, (hipStream_t)stream);
}

int pz_random_actions(int32_t* act_p1, int32_t* act_p2, int64_t n, int64_t env_id_base, uint64_t action_seed,
                      uint64_t t, int32_t n_actions, void* stream)
{
    if (!act_p1 || !act_p2) return PZ_E_NULL;
    if (n < 0 || n_actions < 1) return PZ_E_SIZE;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(random_actions_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, act_p1,
                       act_p2, n, env_id_base, action_seed, t, (uint32_t)n_actions);
    return (int)hipGetLastError();
}

int pz_scenery_init(int32_t* scenery, int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, void* stream)
{
    if (!scenery || !state || !cfg) return PZ_E_NULL;
    if (n < 0 || stride < n || stride > kMaxLanesPerLaunch) return PZ_E_SIZE;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(scenery_init_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, scenery, state,
                       n, stride, *cfg);
    return (int)hipGetLastError();
}

int pz_scenery_track(int32_t* scenery, const int32_t* state, int64_t n, int64_t stride, const pz_config* cfg,
                     int32_t resync, void* stream)
{
    if (!scenery || !state || !cfg) return PZ_E_NULL;
    if (n < 0 || stride < n || stride > kMaxLanesPerLaunch) return PZ_E_SIZE;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(scenery_track_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, scenery, state,
                       n, stride, *cfg, (int)resync);
    return (int)hipGetLastError();
}

int pz_render(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const int32_t* lanes, int64_t m,
              const uint32_t* atlas, const pz_sprite* sprites, const uint32_t* background, int32_t* scenery,
              uint8_t* frames, void* stream)
{
    if (!state || !atlas || !sprites || !background || !frames || (scenery != nullptr && cfg == nullptr)) return PZ_E_NULL;
    if (n < 0 || stride < n || stride > kMaxLanesPerLaunch || m < 0 || m > 65535) return PZ_E_SIZE;
    if ((lanes == nullptr && m > n) || (reinterpret_cast<uintptr_t>(frames) & 3u) != 0) return lanes == nullptr && m > n ? PZ_E_SIZE : PZ_E_ALIGN;
    if (m == 0) return PZ_OK;
    const dim3 grid(blocks_for((int64_t)(PZ_FRAME_WIDTH / 4) * PZ_FRAME_HEIGHT, 256), (unsigned int)m);
    if (scenery != nullptr)
        hipLaunchKernelGGL(scenery_tick_kernel, dim3(blocks_for(m, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, scenery,
                           state, n, stride, *cfg, lanes, m);
    hipLaunchKernelGGL(render_kernel, grid, dim3(256), 0, (hipStream_t)stream, state, n, stride, lanes, atlas, sprites,
                       background, scenery, frames);
    return (int)hipGetLastError();
}


#ifdef PZ_DIAGNOSTIC_BUILD  // (tools/stamps.py; the product exports exactly the header's functions)
int pz_debug_read_stamps(unsigned long long* dst_host, int64_t count)
{
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(pz::diag::g_pz_stamps), count * sizeof(unsigned long long));
}
int pz_debug_read_frame_stamps(unsigned long long* dst_host, int64_t count)
{
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(pz::diag::g_pz_frame_stamps), count * sizeof(unsigned long long));
}
#endif

}  // extern "C"
