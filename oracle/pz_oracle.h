/*
 * pz_oracle.h -- CPU restatement of the pika-zoo per-timestep path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the HIP kernels in pika-zoo_amd/csrc. It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Plain C, scalar, one game per loop iteration, written to follow the
 * reference function by function (citations are `file:line` under /root/reference/).
 *
 * Parity status: PINNED -- checked against golden trajectories captured from the
 * unmodified reference (oracle/ref_capture.py writes the fixtures under tests/golden/) and live against
 * the reference in the build container (tests/test_oracle_vs_reference.py).
 *
 * State layout: int32 state[PZO_W][stride], field-major (structure of arrays); lane i
 * owns column i.  Field indices below are shared by convention with
 * include/pikazoo_hip.h (the product declares its own copy; nothing is included
 * across the oracle/product boundary).
 */
#ifndef PZ_ORACLE_H
#define PZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PZO_W 44
#define PZO_OBS 35

/* player block: p1 at 0, p2 at 13 (physics.py:159-218, :51) */
enum {
    PZO_P_X = 0, PZO_P_Y, PZO_P_YVEL, PZO_P_STATE, PZO_P_FRAME, PZO_P_ARM_SWING,
    PZO_P_DELAY, PZO_P_DIVING_DIR, PZO_P_LYING_DOWN, PZO_P_COLLISION,
    PZO_P_BOLDNESS, PZO_P_STAND_BY, PZO_P_HIT_KEY_PREV, PZO_P_WORDS /* 13 */
};
/* ball block at 26 (physics.py:232-277) */
enum {
    PZO_B_X = 26, PZO_B_Y, PZO_B_XVEL, PZO_B_YVEL, PZO_B_POWER_HIT, PZO_B_PREV_X,
    PZO_B_PREV_Y, PZO_B_PPREV_X, PZO_B_PPREV_Y, PZO_B_FINE_ROT, PZO_B_EXPECTED_X,
    PZO_B_PUNCH_X
};
/* env block at 38 (pikazoo_env.py:100-111) */
enum {
    PZO_E_SCORE1 = 38, PZO_E_SCORE2, PZO_E_P2_SERVE, PZO_E_ROUND_ENDED,
    PZO_E_GAME_ENDED, PZO_E_RNG_COUNTER
};

enum { PZO_SERVE_WINNER = 0, PZO_SERVE_ALTERNATE = 1, PZO_SERVE_RANDOM = 2 };

typedef struct pzo_config {
    int32_t winning_score;       /* pikazoo_env.py:102 */
    int32_t serve_mode;          /* pikazoo_env.py:104-105 */
    int32_t p1_computer;         /* pikazoo_env.py:83 */
    int32_t p2_computer;         /* pikazoo_env.py:84 */
    int32_t simplify_action;     /* wrappers/simplify_action.py:16-25 fused */
    int32_t ballpos_reward;      /* wrappers/reward_by_ball_position.py:20-31 fused */
    int32_t x_line;              /* reward_by_ball_position.py:11 */
    int32_t y_line;              /* reward_by_ball_position.py:12 */
    float   additional_reward[8];/* reward_by_ball_position.py:10 */
    int32_t auto_reset;          /* batched-env addition: reset() in place before the next step */
    int32_t packed_state;         /* product-side storage format flag; the oracle always holds int32 columns */
    int32_t normal_state_mode;   /* wrappers/reward_in_normal_state.py:10-15 fused: 0 off, 1 applied
                                    before additional_reward (wrapper inside RewardByBallPosition),
                                    2 applied after it (wrapper outside) */
    float   normal_state_reward; /* reward_in_normal_state.py:8 */
    int32_t normalize_obs;       /* wrappers/normalize_observation.py:18-35 fused: float32 obs */
    int32_t episode_stats_mode;  /* wrappers/record_episode_statistics.py:27-40 fused: 0 off, 1 sums
                                    the env's own reward, 2 sums the fully wrapped reward */
    uint64_t seed;               /* Philox key of the env RNG stream */
    int64_t env_id_base;         /* global id of lane 0 (multi-GPU sharding) */
    uint64_t *action_faults;     /* product-side diagnostic pointer (pz_config, ABI 8): kept for the identical byte
                                    layout, never read here -- the reference raises IndexError on such an action
                                    (pikazoo_env.py:182), the oracle's behaviour for one is undefined */
    int32_t action_format;       /* product-side (ABI 10): element type of the action vectors; layout only -- the oracle
                                    takes int32 */
    int32_t reserved0;
} pzo_config;

/* Philox4x32-10 block (Salmon et al., SC'11); out[4] */
void pzo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t k0, uint32_t k1, uint32_t out[4]);
/* draw #idx of the env stream of game env_id, bounded to [0,n) */
int32_t pzo_env_draw(uint64_t seed, int64_t env_id, uint32_t idx, uint32_t n);
/* uniform random policy: actions of both players of game env_id at step t, in [0,n_actions) */
void pzo_random_actions(int32_t *act_p1, int32_t *act_p2, int64_t n, int64_t env_id_base,
                        uint64_t action_seed, uint64_t t, int32_t n_actions);

/* raw_env.__init__ semantics (pikazoo_env.py:79-141 -> physics.py:107-123): fresh state,
 * two boldness draws. */
void pzo_init(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg);
/* raw_env.reset (pikazoo_env.py:149-173) on lanes with mask[i]!=0 (mask NULL = all);
 * observations written for every lane (obs may be NULL). */
void pzo_reset(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
               const uint8_t *mask, int32_t *obs_p1, int32_t *obs_p2, void *episode_stats);
/* raw_env.step (pikazoo_env.py:175-240). rew_* are int32[n], or float32[n] when a reward
 * wrapper is fused (ballpos_reward or normal_state_mode); obs_* are int32[n][35], or float32
 * bit patterns when normalize_obs.  episode_stats (may be NULL): double[2][stride] + int32[stride] =
 * episode return of player 1, of player 2 (float64 sums, record_episode_statistics.py:31), episode length.
 * nthreads<=1: scalar loop; >1: static lane partition (OpenMP). */
void pzo_step(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
              const int32_t *act_p1, const int32_t *act_p2,
              int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
              uint8_t *terminated, void *episode_stats, int nthreads);
/* k steps of the random policy (actions from pzo_random_actions at t0..t0+k-1); outputs of
 * the last step are kept. Returns nothing; used for long digest runs and the CPU baseline. */
void pzo_rollout_random(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
                        uint64_t action_seed, uint64_t t0, int32_t k,
                        int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                        uint8_t *terminated, void *episode_stats, int64_t *episodes_finished,
                        int nthreads);
/* _get_obs (pikazoo_env.py:576-624) from the state, no mutation */
void pzo_observe(const int32_t *state, int64_t n, int64_t stride, int32_t normalize,
                 int32_t *obs_p1, int32_t *obs_p2);
/* order-sensitive 64-bit digest of a [W][n] state (FNV-1a over lanes, field-major) */
uint64_t pzo_digest(const int32_t *state, int64_t n, int64_t stride);
/* flight-simulation probes for unit tests (physics.py:643-686, :820-884) */
int32_t pzo_expected_landing_x(int32_t x, int32_t y, int32_t xv, int32_t yv);
int32_t pzo_expected_landing_x_power_hit(int32_t xdir, int32_t ydir,
                                         int32_t x, int32_t y, int32_t xv, int32_t yv);

#ifdef __cplusplus
}
#endif
#endif
