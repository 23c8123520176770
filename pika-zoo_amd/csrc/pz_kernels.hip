// pz_kernels.hip -- gfx950 kernels + C ABI of libpikazoo_hip.so (see include/pikazoo_hip.h).
//
// Launch geometry: one lane per game, one wave64 per workgroup (64 consecutive games).  A
// workgroup reads its 44 state columns with fully coalesced dword loads (256 B per wave
// instruction), runs the frame in registers, and writes the two row-major [n][35]
// observation tensors through an LDS transpose so that the global stores are contiguous
// 16-byte-per-lane streams (a wave's 64 rows are one contiguous 8 960-byte span).
// 64-lane workgroups keep the transpose barrier-free across waves and give the dispatcher
// 1 024 independent workgroups at the 65 536-game batch (4 per CU, one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pikazoo_hip.h"
#include "pz_physics.hpp"

namespace pz {

constexpr int kLanes = 64;  // lanes (games) per workgroup = one wavefront

struct StepArgs {
    int32_t* state;
    int64_t n, stride;
    const int32_t* act_p1;  // nullptr => on-device random policy
    const int32_t* act_p2;
    uint64_t action_seed, t0;
    int32_t k;  // frames per launch (random policy only)
    int32_t* obs_p1;
    int32_t* obs_p2;
    void* rew_p1;
    void* rew_p2;
    uint8_t* terminated;
    unsigned long long* episodes_done;
    pz_config cfg;
};

// ---- state columns <-> registers -----------------------------------------------------------
__device__ __forceinline__ void load_player(Player& p, const int32_t* __restrict__ s, int64_t stride)
{
    p.x = s[PZ_P_X * stride];
    p.y = s[PZ_P_Y * stride];
    p.yv = s[PZ_P_Y_VELOCITY * stride];
    p.state = s[PZ_P_STATE * stride];
    p.frame = s[PZ_P_FRAME_NUMBER * stride];
    p.arm = s[PZ_P_ARM_SWING_DIRECTION * stride];
    p.delay = s[PZ_P_DELAY_BEFORE_NEXT_FRAME * stride];
    p.dive = s[PZ_P_DIVING_DIRECTION * stride];
    p.lying = s[PZ_P_LYING_DOWN_DURATION_LEFT * stride];
    p.coll = s[PZ_P_IS_COLLISION_WITH_BALL_HAPPENED * stride];
    p.bold = s[PZ_P_COMPUTER_BOLDNESS * stride];
    p.standby = s[PZ_P_COMPUTER_WHERE_TO_STAND_BY * stride];
    p.hitprev = s[PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS * stride];
}

__device__ __forceinline__ void store_player(const Player& p, int32_t* __restrict__ s, int64_t stride)
{
    s[PZ_P_X * stride] = p.x;
    s[PZ_P_Y * stride] = p.y;
    s[PZ_P_Y_VELOCITY * stride] = p.yv;
    s[PZ_P_STATE * stride] = p.state;
    s[PZ_P_FRAME_NUMBER * stride] = p.frame;
    s[PZ_P_ARM_SWING_DIRECTION * stride] = p.arm;
    s[PZ_P_DELAY_BEFORE_NEXT_FRAME * stride] = p.delay;
    s[PZ_P_DIVING_DIRECTION * stride] = p.dive;
    s[PZ_P_LYING_DOWN_DURATION_LEFT * stride] = p.lying;
    s[PZ_P_IS_COLLISION_WITH_BALL_HAPPENED * stride] = p.coll;
    s[PZ_P_COMPUTER_BOLDNESS * stride] = p.bold;
    s[PZ_P_COMPUTER_WHERE_TO_STAND_BY * stride] = p.standby;
    s[PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS * stride] = p.hitprev;
}

__device__ __forceinline__ void load_game(Game& g, const int32_t* __restrict__ s, int64_t stride)
{
    load_player(g.p1, s, stride);
    load_player(g.p2, s + PZ_P_WORDS * stride, stride);
    g.b.x = s[PZ_B_X * stride];
    g.b.y = s[PZ_B_Y * stride];
    g.b.xv = s[PZ_B_X_VELOCITY * stride];
    g.b.yv = s[PZ_B_Y_VELOCITY * stride];
    g.b.power = s[PZ_B_IS_POWER_HIT * stride];
    g.b.px = s[PZ_B_PREVIOUS_X * stride];
    g.b.py = s[PZ_B_PREVIOUS_Y * stride];
    g.b.ppx = s[PZ_B_PREVIOUS_PREVIOUS_X * stride];
    g.b.ppy = s[PZ_B_PREVIOUS_PREVIOUS_Y * stride];
    g.b.rot = s[PZ_B_FINE_ROTATION * stride];
    g.b.ex = s[PZ_B_EXPECTED_LANDING_POINT_X * stride];
    g.b.punch = s[PZ_B_PUNCH_EFFECT_X * stride];
    g.e.s1 = s[PZ_E_SCORE_P1 * stride];
    g.e.s2 = s[PZ_E_SCORE_P2 * stride];
    g.e.p2serve = s[PZ_E_IS_PLAYER2_SERVE * stride];
    g.e.round_ended = s[PZ_E_ROUND_ENDED * stride];
    g.e.game_ended = s[PZ_E_GAME_ENDED * stride];
    g.e.rng = (uint32_t)s[PZ_E_RNG_DRAW_COUNTER * stride];
}

__device__ __forceinline__ void store_game(const Game& g, int32_t* __restrict__ s, int64_t stride)
{
    store_player(g.p1, s, stride);
    store_player(g.p2, s + PZ_P_WORDS * stride, stride);
    s[PZ_B_X * stride] = g.b.x;
    s[PZ_B_Y * stride] = g.b.y;
    s[PZ_B_X_VELOCITY * stride] = g.b.xv;
    s[PZ_B_Y_VELOCITY * stride] = g.b.yv;
    s[PZ_B_IS_POWER_HIT * stride] = g.b.power;
    s[PZ_B_PREVIOUS_X * stride] = g.b.px;
    s[PZ_B_PREVIOUS_Y * stride] = g.b.py;
    s[PZ_B_PREVIOUS_PREVIOUS_X * stride] = g.b.ppx;
    s[PZ_B_PREVIOUS_PREVIOUS_Y * stride] = g.b.ppy;
    s[PZ_B_FINE_ROTATION * stride] = g.b.rot;
    s[PZ_B_EXPECTED_LANDING_POINT_X * stride] = g.b.ex;
    s[PZ_B_PUNCH_EFFECT_X * stride] = g.b.punch;
    s[PZ_E_SCORE_P1 * stride] = g.e.s1;
    s[PZ_E_SCORE_P2 * stride] = g.e.s2;
    s[PZ_E_IS_PLAYER2_SERVE * stride] = g.e.p2serve;
    s[PZ_E_ROUND_ENDED * stride] = g.e.round_ended;
    s[PZ_E_GAME_ENDED * stride] = g.e.game_ended;
    s[PZ_E_RNG_DRAW_COUNTER * stride] = (int32_t)g.e.rng;
}

__device__ __forceinline__ RngId make_rng_id(const pz_config& cfg, int64_t lane_index)
{
    const uint64_t gid = (uint64_t)(cfg.env_id_base + lane_index);
    return RngId{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32)};
}

// ---- observation pack: _get_obs (pikazoo_env.py:576-624) ------------------------------------
// Row layout: player(13) | opponent(13) | ball(9).  Rows go to LDS at stride 35 words (odd,
// so the 64 lanes of a ds_write_b32 hit 32 distinct banks twice = conflict-free), then the
// wave copies the contiguous 64x35-word span to HBM with 16-byte lanes.
__device__ __forceinline__ void player_row(const Player& p, int32_t* __restrict__ o)
{
    o[0] = p.x;
    o[1] = p.y;
    o[2] = p.yv;
    o[3] = p.dive;
    o[4] = p.lying;
    o[5] = p.frame;
    o[6] = p.delay;
    o[7] = p.state == 0;
    o[8] = p.state == 1;
    o[9] = p.state == 2;
    o[10] = p.state == 3;
    o[11] = p.state == 4;
    o[12] = p.hitprev;
}

__device__ __forceinline__ void ball_row(const Ball& b, int32_t* __restrict__ o)
{
    o[0] = b.x;
    o[1] = b.y;
    o[2] = b.px;
    o[3] = b.py;
    o[4] = b.ppx;
    o[5] = b.ppy;
    o[6] = b.xv;
    o[7] = b.yv;
    o[8] = b.power;
}

__device__ __forceinline__ void stage_obs(const Game& g, int32_t* __restrict__ s1, int32_t* __restrict__ s2, int lane)
{
    int32_t* r1 = s1 + lane * PZ_OBS_DIM;
    int32_t* r2 = s2 + lane * PZ_OBS_DIM;
    player_row(g.p1, r1);
    player_row(g.p2, r1 + 13);
    ball_row(g.b, r1 + 26);
    player_row(g.p2, r2);
    player_row(g.p1, r2 + 13);
    ball_row(g.b, r2 + 26);
}

// copy `words` int32 from LDS to a 16-byte aligned global span, 16 B per lane per pass
__device__ __forceinline__ void flush_rows(const int32_t* __restrict__ lds, int32_t* __restrict__ dst, int words,
                                           int lane)
{
    const int vecs = words >> 2;
    const int4* src4 = reinterpret_cast<const int4*>(lds);
    int4* dst4 = reinterpret_cast<int4*>(dst);
    for (int v = lane; v < vecs; v += kLanes) dst4[v] = src4[v];
    const int tail = vecs << 2;
    if (tail + lane < words) dst[tail + lane] = lds[tail + lane];
}

// ---- the fused step kernel -------------------------------------------------------------------
// AI1/AI2: player 1 / 2 is the rule-based computer (compile-time so the human-vs-human build
// carries none of the predictor code or its registers).  RANDOM: actions are drawn on device
// and k frames may run per launch; otherwise actions are read from HBM and k == 1.
template <bool AI1, bool AI2, bool RANDOM>
__global__ __launch_bounds__(kLanes) void step_kernel(const StepArgs a)
{
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];

    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kLanes;
    const int64_t i = base + lane;
    const bool live = i < a.n;
    const int valid = (int)min((int64_t)kLanes, a.n - base);

    Game g{};
    RngId id = make_rng_id(a.cfg, live ? i : 0);
    int reward = 0;
    bool frozen = false;
    unsigned int finished = 0;
#ifdef PZ_ABLATE
    // timing-only build (tools/ablate.py): cfg.reserved bits redirect traffic to one workgroup's
    // span (so it stays in cache) or skip the frame; results are wrong by construction.
    const int ab = a.cfg.reserved;
    g_pz_ablate_bits = ab;  // every lane stores the same value; read back by the predictor hooks
    const int64_t i_ld = (ab & 1) ? lane : i;          // bit0: state loads hit workgroup 0's columns
    const int64_t i_st = (ab & 2) ? lane : i;          // bit1: state stores go to workgroup 0's columns
    const int64_t obs_base = (ab & 4) ? 0 : base;      // bit2: observation rows go to workgroup 0's span
#define PZ_LD_INDEX i_ld
#define PZ_ST_INDEX i_st
#define PZ_OBS_BASE obs_base
#define PZ_SKIP_FRAME (ab & 8)                         // bit3: no game logic
#define PZ_SKIP_OBS (ab & 16)                          // bit4: no observation staging / flush at all
#else
#define PZ_LD_INDEX i
#define PZ_ST_INDEX i
#define PZ_OBS_BASE base
#define PZ_SKIP_FRAME 0
#define PZ_SKIP_OBS 0
#endif
    if (live) load_game(g, a.state + PZ_LD_INDEX, a.stride);
    // The frame runs in wave-uniform control flow (the computer player's power-hit candidates
    // are evaluated cooperatively by the wave); lanes past the end of the batch idle inside.
    // lds_obs[0] doubles as the cooperative scratch until the observations are staged.
    if (RANDOM) {
        const uint32_t n_actions = a.cfg.simplify_action ? 13u : 18u;
        for (int32_t s = 0; s < a.k; ++s) {
            int a1, a2;
            policy_actions(id.id_lo, id.id_hi, a.action_seed, a.t0 + (uint64_t)s, n_actions, a1, a2);
            reward = step_games<AI1, AI2>(g, a.cfg, id, a1, a2, live, frozen, lds_obs[0], lane);
            finished += (unsigned int)(live && g.e.game_ended && !frozen);
        }
    } else if (!PZ_SKIP_FRAME) {
        const int a1 = live ? a.act_p1[i] : 0, a2 = live ? a.act_p2[i] : 0;
        reward = step_games<AI1, AI2>(g, a.cfg, id, a1, a2, live, frozen, lds_obs[0], lane);
        finished = (unsigned int)(live && g.e.game_ended && !frozen);
    }
    if (live) {
        store_game(g, a.state + PZ_ST_INDEX, a.stride);

        // rewards (pikazoo_env.py:217-228), optionally with RewardByBallPosition fused
        // (reward_by_ball_position.py:22-29: zone from the post-step ball position)
        if (a.cfg.ballpos_reward) {
            const int zone = (g.b.y > a.cfg.y_line ? 1 : 0) + (g.b.x >= a.cfg.x_line ? 2 : 0);
            float r1 = (float)reward, r2 = (float)(-reward);
            if (!frozen) {
                r1 += a.cfg.additional_reward[zone];
                r2 += a.cfg.additional_reward[4 + zone];
            }
            static_cast<float*>(a.rew_p1)[i] = r1;
            static_cast<float*>(a.rew_p2)[i] = r2;
        } else {
            static_cast<int32_t*>(a.rew_p1)[i] = reward;
            static_cast<int32_t*>(a.rew_p2)[i] = -reward;
        }
        a.terminated[i] = (uint8_t)g.e.game_ended;  // :233
        if (!PZ_SKIP_OBS) stage_obs(g, lds_obs[0], lds_obs[1], lane);
    }
    __syncthreads();
    if (!PZ_SKIP_OBS) {
        flush_rows(lds_obs[0], a.obs_p1 + PZ_OBS_BASE * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
        flush_rows(lds_obs[1], a.obs_p2 + PZ_OBS_BASE * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
    }

    if (a.episodes_done != nullptr) {
        // one atomic per wave: reduce the per-lane counts across the wavefront first
        unsigned int total = finished;
        for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, kLanes);
        if (lane == 0 && total != 0) atomicAdd(a.episodes_done, (unsigned long long)total);
    }
}

// ---- constructor / reset / observe / policy kernels --------------------------------------------
__global__ __launch_bounds__(kLanes) void init_kernel(int32_t* state, int64_t n, int64_t stride, const pz_config cfg)
{
    const int64_t i = (int64_t)blockIdx.x * kLanes + threadIdx.x;
    if (i >= n) return;
    Game g;
    const RngId id = make_rng_id(cfg, i);
    construct_game(g, id);
    store_game(g, state + i, stride);
}

__global__ __launch_bounds__(kLanes) void reset_kernel(int32_t* state, int64_t n, int64_t stride, const pz_config cfg,
                                                       const uint8_t* mask, int32_t* obs_p1, int32_t* obs_p2)
{
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kLanes;
    const int64_t i = base + lane;
    const int valid = (int)min((int64_t)kLanes, n - base);
    if (i < n) {
        Game g;
        load_game(g, state + i, stride);
        if (mask == nullptr || mask[i] != 0) {
            const RngId id = make_rng_id(cfg, i);
            reset_game(g, cfg, id);
            store_game(g, state + i, stride);
        }
        stage_obs(g, lds_obs[0], lds_obs[1], lane);
    }
    __syncthreads();
    if (obs_p1 != nullptr) flush_rows(lds_obs[0], obs_p1 + base * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
    if (obs_p2 != nullptr) flush_rows(lds_obs[1], obs_p2 + base * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
}

__global__ __launch_bounds__(kLanes) void observe_kernel(const int32_t* state, int64_t n, int64_t stride,
                                                         int32_t* obs_p1, int32_t* obs_p2)
{
    __shared__ __attribute__((aligned(16))) int32_t lds_obs[2][kLanes * PZ_OBS_DIM];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kLanes;
    const int64_t i = base + lane;
    const int valid = (int)min((int64_t)kLanes, n - base);
    if (i < n) {
        Game g;
        load_game(g, state + i, stride);
        stage_obs(g, lds_obs[0], lds_obs[1], lane);
    }
    __syncthreads();
    if (obs_p1 != nullptr) flush_rows(lds_obs[0], obs_p1 + base * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
    if (obs_p2 != nullptr) flush_rows(lds_obs[1], obs_p2 + base * PZ_OBS_DIM, valid * PZ_OBS_DIM, lane);
}

__global__ __launch_bounds__(256) void random_actions_kernel(int32_t* act_p1, int32_t* act_p2, int64_t n,
                                                             int64_t env_id_base, uint64_t action_seed, uint64_t t,
                                                             uint32_t n_actions)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t gid = (uint64_t)(env_id_base + i);
    int a1, a2;
    policy_actions((uint32_t)gid, (uint32_t)(gid >> 32), action_seed, t, n_actions, a1, a2);
    act_p1[i] = a1;
    act_p2[i] = a2;
}

// Self-test hook: both forms of the flight predictor on caller-supplied ball states.
__global__ __launch_bounds__(256) void predictor_selftest_kernel(const int32_t* x, const int32_t* y, const int32_t* xv,
                                                                 const int32_t* yv, int64_t n, int full_net,
                                                                 int32_t* out_fast, int32_t* out_iter)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (full_net) {
        out_fast[i] = predict_landing_x<true>(x[i], y[i], xv[i], yv[i]);
        out_iter[i] = predict_landing_x_iterative<true>(x[i], y[i], xv[i], yv[i]);
    } else {
        out_fast[i] = predict_landing_x<false>(x[i], y[i], xv[i], yv[i]);
        out_iter[i] = predict_landing_x_iterative<false>(x[i], y[i], xv[i], yv[i]);
    }
}

// ---- host side ---------------------------------------------------------------------------------
static int check_common(const void* state, int64_t n, int64_t stride, const pz_config* cfg)
{
    if (state == nullptr || cfg == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n) return PZ_E_SIZE;
    if (cfg->winning_score < 1 || cfg->serve_mode < 0 || cfg->serve_mode > 2) return PZ_E_CONFIG;
    return PZ_OK;
}

static inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

static inline unsigned int blocks_for(int64_t n, int per) { return (unsigned int)((n + per - 1) / per); }

template <bool RANDOM>
static int launch_step(const StepArgs& a, hipStream_t stream)
{
    const dim3 grid(blocks_for(a.n, kLanes)), block(kLanes);
    const bool ai1 = a.cfg.p1_computer != 0, ai2 = a.cfg.p2_computer != 0;
    if (ai1 && ai2)
        hipLaunchKernelGGL((step_kernel<true, true, RANDOM>), grid, block, 0, stream, a);
    else if (ai1)
        hipLaunchKernelGGL((step_kernel<true, false, RANDOM>), grid, block, 0, stream, a);
    else if (ai2)
        hipLaunchKernelGGL((step_kernel<false, true, RANDOM>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((step_kernel<false, false, RANDOM>), grid, block, 0, stream, a);
    return (int)hipGetLastError();
}

}  // namespace pz

using namespace pz;

extern "C" {

int pz_abi_version(void) { return PZ_ABI_VERSION; }
int pz_state_words(void) { return PZ_STATE_WORDS; }
int pz_obs_dim(void) { return PZ_OBS_DIM; }
int pz_config_bytes(void) { return (int)sizeof(pz_config); }

const char* pz_error_string(int code)
{
    switch (code) {
        case PZ_OK: return "ok";
        case PZ_E_NULL: return "required pointer is NULL";
        case PZ_E_SIZE: return "bad size (n < 0, stride < n or k < 1)";
        case PZ_E_CONFIG: return "pz_config field out of range";
        case PZ_E_ALIGN: return "observation buffer is not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown pikazoo error";
    }
}

int pz_init(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(init_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n, stride,
                       *cfg);
    return (int)hipGetLastError();
}

int pz_reset(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const uint8_t* mask, int32_t* obs_p1,
             int32_t* obs_p2, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(reset_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                       stride, *cfg, mask, obs_p1, obs_p2);
    return (int)hipGetLastError();
}

int pz_observe(const int32_t* state, int64_t n, int64_t stride, int32_t* obs_p1, int32_t* obs_p2, void* stream)
{
    if (state == nullptr) return PZ_E_NULL;
    if (n < 0 || stride < n) return PZ_E_SIZE;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(observe_kernel, dim3(blocks_for(n, kLanes)), dim3(kLanes), 0, (hipStream_t)stream, state, n,
                       stride, obs_p1, obs_p2);
    return (int)hipGetLastError();
}

int pz_step(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, const int32_t* act_p1,
            const int32_t* act_p2, int32_t* obs_p1, int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated,
            void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!act_p1 || !act_p2 || !obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    StepArgs a{state, n, stride, act_p1, act_p2, 0, 0, 1, obs_p1, obs_p2, rew_p1, rew_p2, terminated, nullptr, *cfg};
    return launch_step<false>(a, (hipStream_t)stream);
}

int pz_step_random(int32_t* state, int64_t n, int64_t stride, const pz_config* cfg, uint64_t action_seed, uint64_t t0,
                   int32_t k, int32_t* obs_p1, int32_t* obs_p2, void* rew_p1, void* rew_p2, uint8_t* terminated,
                   int64_t* episodes_done, void* stream)
{
    if (int e = check_common(state, n, stride, cfg)) return e;
    if (!obs_p1 || !obs_p2 || !rew_p1 || !rew_p2 || !terminated) return PZ_E_NULL;
    if (k < 1) return PZ_E_SIZE;
    if (misaligned16(obs_p1) || misaligned16(obs_p2)) return PZ_E_ALIGN;
    if (n == 0) return PZ_OK;
    StepArgs a{state,  n,      stride, nullptr, nullptr,    action_seed, t0, k, obs_p1, obs_p2, rew_p1,
               rew_p2, terminated, reinterpret_cast<unsigned long long*>(episodes_done), *cfg};
    return launch_step<true>(a, (hipStream_t)stream);
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

int pz_selftest_predictor(const int32_t* x, const int32_t* y, const int32_t* xv, const int32_t* yv, int64_t n,
                          int32_t full_net, int32_t* out_fast, int32_t* out_iter, void* stream)
{
    if (!x || !y || !xv || !yv || !out_fast || !out_iter) return PZ_E_NULL;
    if (n < 0) return PZ_E_SIZE;
    if (n == 0) return PZ_OK;
    hipLaunchKernelGGL(predictor_selftest_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, xv,
                       yv, n, (int)full_net, out_fast, out_iter);
    return (int)hipGetLastError();
}

}  // extern "C"
