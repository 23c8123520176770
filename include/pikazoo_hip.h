/*
 * pikazoo_hip.h -- C ABI of libpikazoo_hip.so, the MI355X (gfx950) batched Pikachu-Volleyball
 * step path.  Plain pointers and sizes only: no torch / HIP types in the signatures
 * (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *
 * The reference (helpingstar/pika-zoo) has no FFI layer: the path sits behind the
 * PettingZoo ParallelEnv Python API of `pikazoo_v0.env()` (pikazoo/env/pikazoo_env.py:72-240).
 * Each entry point below names the reference interface it replaces; INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - every buffer is device memory owned by the caller (torch tensors in the Python host);
 *    the library allocates nothing persistent and keeps no global state, so calls on
 *    distinct state buffers are thread-safe;
 *  - all calls are asynchronous on `stream`;
 *  - return value: 0 = ok, negative = PZ_E_* argument error, positive = hipError_t;
 *  - state is int32[PZ_STATE_WORDS][stride], field-major (structure of arrays): lane i
 *    (one independent game) owns column i; `n` lanes are live, stride >= n -- or, with
 *    cfg->packed_state, the bit-packed format below (36 bytes per game instead of 176);
 *  - observations are int32[n][35] row-major per agent (pikazoo_env.py:576-624), or float32 / int16 rows as
 *    cfg->normalize_obs names;
 *  - actions are int32 / int64 / uint8 / int16 vectors as cfg->action_format names (what the caller's policy
 *    produced: nothing is cast on the way in); an action outside the range is counted into cfg->action_faults on its
 *    FULL value (the reference raises IndexError at pikazoo_env.py:182), see pz_config.
 */
#ifndef PIKAZOO_HIP_H
#define PIKAZOO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 10: pz_config grows to 120 bytes: action_format (int64 / uint8 / int16 action vectors straight into the launch,
 *     range-checked on the full value; pz_step / pz_step_bound take `const void *` action vectors); the landing table
 *     of pz_flight_tables is optional on its own (power_hit alone is a supported mode).
 *  9: the diagnostics pz_probe_launch / pz_selftest_predictor left this library (include/pikazoo_diag.h,
 *     libpikazoo_diag.so) */
#define PZ_ABI_VERSION 10
#define PZ_STATE_WORDS 44
#define PZ_OBS_DIM 35

/* ---- state columns ------------------------------------------------------------------- */
/* player block (13 words): player 1 at 0, player 2 at 13.
 * Player attributes physics.py:159-218; power_hit_key_is_down_previous physics.py:51 */
enum pz_player_field {
    PZ_P_X = 0, PZ_P_Y, PZ_P_Y_VELOCITY, PZ_P_STATE, PZ_P_FRAME_NUMBER,
    PZ_P_ARM_SWING_DIRECTION, PZ_P_DELAY_BEFORE_NEXT_FRAME, PZ_P_DIVING_DIRECTION,
    PZ_P_LYING_DOWN_DURATION_LEFT, PZ_P_IS_COLLISION_WITH_BALL_HAPPENED,
    PZ_P_COMPUTER_BOLDNESS, PZ_P_COMPUTER_WHERE_TO_STAND_BY,
    PZ_P_POWER_HIT_KEY_IS_DOWN_PREVIOUS, PZ_P_WORDS
};
/* ball block (12 words) at 26: physics.py:232-277 */
enum pz_ball_field {
    PZ_B_X = 26, PZ_B_Y, PZ_B_X_VELOCITY, PZ_B_Y_VELOCITY, PZ_B_IS_POWER_HIT,
    PZ_B_PREVIOUS_X, PZ_B_PREVIOUS_Y, PZ_B_PREVIOUS_PREVIOUS_X, PZ_B_PREVIOUS_PREVIOUS_Y,
    PZ_B_FINE_ROTATION, PZ_B_EXPECTED_LANDING_POINT_X, PZ_B_PUNCH_EFFECT_X
};
/* env block (6 words) at 38: pikazoo_env.py:100-111 + the RNG draw counter */
enum pz_env_field {
    PZ_E_SCORE_P1 = 38, PZ_E_SCORE_P2, PZ_E_IS_PLAYER2_SERVE, PZ_E_ROUND_ENDED,
    PZ_E_GAME_ENDED, PZ_E_RNG_DRAW_COUNTER
};

enum pz_serve_mode { PZ_SERVE_WINNER = 0, PZ_SERVE_ALTERNATE = 1, PZ_SERVE_RANDOM = 2 };

/* element type of the action vectors (pz_config.action_format).  The reference indexes a Python list with whatever
 * integer it is handed (pikazoo_env.py:182); torch's default integer dtype is int64 (argmax, multinomial,
 * Categorical.sample, randint), small policies emit uint8 / int16.  The launch loads the element as it is and checks
 * the FULL value against [0, n_actions): an int64 of 2**32 + 3 or -1 is a fault, never action 3. */
enum pz_action_format { PZ_ACT_I32 = 0, PZ_ACT_I64 = 1, PZ_ACT_U8 = 2, PZ_ACT_I16 = 3 };

enum pz_error {
    PZ_OK = 0,
    PZ_E_NULL = -1,       /* a required pointer is NULL */
    PZ_E_SIZE = -2,       /* n < 0, stride < n, k < 1 ... */
    PZ_E_CONFIG = -3,     /* config field out of range */
    PZ_E_ALIGN = -4       /* buffer not 16-byte aligned (observations, packed state, power_hit table) */
};

/* Constructor kwargs of pikazoo_v0.env (pikazoo_env.py:79-86) + the fused wrappers
 * (wrappers/simplify_action.py, reward_by_ball_position.py, reward_in_normal_state.py,
 * normalize_observation.py, record_episode_statistics.py) + the batched-env additions
 * (auto_reset, seed, env_id_base, action_faults, action_format). POD, 120 bytes, passed by pointer from the host and
 * by value to the kernels.
 *
 * Reward pipeline of one frame, in the order the reference's wrapper stack would apply it:
 *   r = +1/-1/0 (pikazoo_env.py:217-228)
 *   [episode_stats_mode 1: statistics sum r]
 *   [normal_state_mode 1: r == 0 -> normal_state_reward]      RewardInNormalState inside ...
 *   [ballpos_reward:      r += additional_reward[4*i + zone]]  ... RewardByBallPosition
 *   [normal_state_mode 2: r == 0 -> normal_state_reward]      RewardInNormalState outside it
 *   [episode_stats_mode 2: statistics sum r]
 * Rewards are float32 when ballpos_reward or normal_state_mode is set, else int32. */
typedef struct pz_config {
    int32_t winning_score;        /* >= 1 */
    int32_t serve_mode;           /* enum pz_serve_mode */
    int32_t p1_computer;          /* is_player1_computer */
    int32_t p2_computer;          /* is_player2_computer */
    int32_t simplify_action;      /* 1: actions are Discrete(13), remapped per side */
    int32_t ballpos_reward;       /* 1: rewards are float32 and include additional_reward */
    int32_t x_line;               /* RewardByBallPosition x_line (default 216) */
    int32_t y_line;               /* RewardByBallPosition y_line (default 176) */
    float   additional_reward[8]; /* [0..3] player_1 zones, [4..7] player_2 zones */
    int32_t auto_reset;           /* 1: a finished game is reset() in place before its next frame */
    int32_t packed_state;         /* 0: `state` is int32[PZ_STATE_WORDS][stride]; 1: the packed format (below) */
    int32_t normal_state_mode;    /* RewardInNormalState (reward_in_normal_state.py:10-15): 0 off,
                                     1 applied before additional_reward, 2 after it */
    float   normal_state_reward;  /* its constant (reward_in_normal_state.py:8) */
    int32_t normalize_obs;        /* observation format: 0 int32 (the reference's Box dtype); 1 NormalizeObservation
                                     (normalize_observation.py:18-35): float32 (obs - low) / (high - low); 2 int16: the
                                     values of format 0 in 70-byte rows -- the observation tensors are the largest
                                     stream a step writes, this halves them.  A format-2 tensor holds an EVEN number
                                     of rows (n rounded up); k-frame launches then need n % 8 == 0 */
    int32_t episode_stats_mode;   /* RecordEpisodeStatistics (record_episode_statistics.py:27-40): 0 off,
                                     1 sums the env's own reward, 2 the fully wrapped reward */
    uint64_t seed;                /* Philox4x32-10 key of the env RNG stream */
    int64_t env_id_base;          /* global id of lane 0 (shards of one job use disjoint ranges) */
    uint64_t *action_faults;      /* NULL, or a device counter (caller-owned, 8-byte aligned, zeroed by the caller): the
                                     launches that READ actions (pz_step, pz_step_bound, pz_step_many) add to it when an
                                     action lies outside [0, 18) -- [0, 13) with simplify_action.  The reference's table
                                     lookup raises IndexError there (pikazoo_env.py:182); a launch cannot raise, so it
                                     counts: non-zero afterwards = some action was out of range (at least one count per
                                     offending game and launch).  Such an action is never used as an index (the decode
                                     shifts bit tables): that game's input for the frame is undefined, no memory is
                                     touched out of bounds, every other game is unaffected.  With NULL nothing is
                                     checked (the range is then the caller's contract, as in ABI 7). */
    int32_t action_format;        /* enum pz_action_format: element type of act_p1 / act_p2 (pz_step, pz_step_bound) and of
                                     the tape of pz_step_many (PZ_ACT_I32 or PZ_ACT_I64 there) */
    int32_t reserved0;            /* 0 */
} pz_config;

/* ---- the packed state format (cfg->packed_state = 1) -----------------------------------------------
 * SURVEY section 8(f)-3 "int16 / bit-packed state".  The same 44 values in 36 bytes per game, three columns
 * (structure of arrays, game i at element i of each):
 *   group A  uint32[4][..] at byte 0            player 1, env block, ball.punch_effect_x, ball.previous_previous_y
 *   group B  uint32[4][..] at byte 16 * stride  player 2, the rest of the ball
 *   tail     uint32[..]    at byte 32 * stride  expected_landing_point_x (uint16) | computer_boldness of player 1,
 *                                               player 2 (uint8 each)
 * i.e. a buffer of pz_packed_state_bytes(stride) = 36 * stride bytes, 16-byte aligned.  Field layout:
 *   A0 / B0  x 9 | y 8 <<9 | y_velocity+32 6 <<17 | state 3 <<23 | frame_number 3 <<26 | delay_before_next_frame 3 <<29
 *   A1 / B1  bits 0-8: arm_swing_direction==1 | diving_direction+1 2 <<1 | lying_down_duration_left+2 3 <<3 |
 *            is_collision_with_ball_happened <<6 | computer_where_to_stand_by <<7 | power_hit_key_is_down_previous <<8
 *   A1       | punch_effect_x 9 <<9 | is_player2_serve <<18 | round_ended <<19 | game_ended <<20 |
 *            previous_previous_y 10 signed <<21 | misfit flag <<31
 *   A2       score_p1 16 | score_p2 16 <<16          A3   rng draw counter
 *   B1       | is_power_hit <<9 | x_velocity+32 6 <<10 | y_velocity 13 signed <<16 | misfit flag <<31
 *   B2       x 9 | previous_x 9 <<9 | previous_previous_x 9 <<18
 *   B3       y 10 signed | previous_y 10 signed <<10 | fine_rotation 6 <<20
 *            (the ball's y is signed: a ball falling onto the net top faster than its height is bounced to
 *            y - y_velocity < 0, physics.py:406-419 -- the ceiling is tested before the net)
 * Every entry point that takes a pz_config reads / writes `state` in the format the config names (so winning_score
 * must be <= 65535 then); pz_observe takes the format as an argument; pz_render reads int32 columns only.  The
 * results are the same bit for bit: pz_unpack_state of a packed run equals the int32 run.  The field widths hold
 * every value play can produce; pz_pack_state counts the games of a caller-supplied state that
 * do not fit, and a step kernel that ever met a ball y velocity outside +-4095 or a ball y outside -512..511 would
 * raise the game's misfit flag (sticky; counted by pz_unpack_state). */
#define PZ_PACKED_BYTES_PER_GAME 36
int64_t pz_packed_state_bytes(int64_t stride);
/* int32 columns -> packed; *misfits (int64, device, may be NULL) += games with a value outside its field */
int pz_pack_state(const int32_t *state, int64_t n, int64_t stride, void *packed, int64_t packed_stride,
                  int64_t *misfits, void *stream);
/* *flagged (int64, device) += games whose sticky misfit flag is set: reads 8 bytes per game -- what a training loop
 * on the packed format polls now and then (the step kernels raise the flag, nothing else reports it until the state
 * is unpacked) */
int pz_count_packed_misfits(const void *packed, int64_t n, int64_t packed_stride, int64_t *flagged, void *stream);
/* packed -> int32 columns; *flagged (int64, device, may be NULL) += games whose misfit flag is set */
int pz_unpack_state(const void *packed, int64_t n, int64_t packed_stride, int32_t *state, int64_t stride,
                    int64_t *flagged, void *stream);

/* ---- memory-placement probe ------------------------------------------------------------------
 * MI355X's HBM3E answers two concurrent write streams a quarter faster when they go to different thirds (ranks) of
 * the device memory than when both go to the same one: the k-frame launches write two large
 * observation tensors at once, so where the caller allocated them decides 2.8 vs 3.6 us per frame.  pz_probe_write
 * issues exactly those stores -- [frames][1024 spans of 8 960 bytes], 16 bytes per lane -- into `a`, into `b`, or
 * (both non-NULL) into both in turn, with nothing in front of them; a caller that times the three cases learns
 * whether its two allocations share a rank: t(a, b) ~ t(a) + t(b) when they do, ~ 0.8 of that when they do not.
 * `bytes` (per buffer; the first floor(bytes / pz_probe_frame_bytes()) frames are written, at least one) are
 * OVERWRITTEN.  Nothing in the library calls it; pikazoo_amd/placement.py uses it when it allocates trajectory
 * tensors. */
int pz_probe_write(void *a, void *b, int64_t bytes, void *stream);
int64_t pz_probe_frame_bytes(void);   /* = 1024 * 8960 */

/* ---- flight look-up tables of the computer player (optional; caller-owned device memory) -----
 * The two flight predictors of the rule-based computer player are pure functions of a few small
 * integers, so they can be tabulated once per device and looked up by the step kernels instead of
 * being iterated every frame (the slowest flight of a launch otherwise sets the launch's duration):
 *   landing    calculate_expected_landing_point_x_for (physics.py:643-686):
 *              uint16 [2*PZ_FT_YV_MAX+1 y velocities][23 x velocities][253 y][413 x]
 *              (x 20..432, y 0..252, x velocity -20, -10..10, 20 -- the only values play produces:
 *              physics.py:606-613,626-629 -- |y velocity| <= PZ_FT_YV_MAX);
 *   power_hit  expected_landing_point_x_when_power_hit (physics.py:820-884) for the six
 *              (x_direction, y_direction) candidates of decide_whether_input_power_hit (:796-816):
 *              uint16 [PZ_FT_HIT_YV_MAX+1 |y velocity|][192 y][413 x][8] (y 61..252; entries 6,7 unused).
 * pz_build_flight_tables fills them with the frame-by-frame iteration of the reference (the form
 * pz_selftest_predictor of the diagnostics library, pikazoo_diag.h, exposes as out_iter).  A ball state outside a table's domain is computed in
 * the kernel as before, so results never depend on whether tables are passed. */
#define PZ_FT_YV_MAX 96
#define PZ_FT_HIT_YV_MAX 64
typedef struct pz_flight_tables {
    const uint16_t *landing;     /* pz_flight_table_bytes(0) bytes (the entries + 2 bytes of padding: 927 MB), 4-byte
                                    aligned, or NULL: the landing point is then predicted in the kernel (closed-form
                                    fast-forward) */
    const uint16_t *power_hit;   /* pz_flight_table_bytes(1) bytes (82 MB), 16-byte aligned, or NULL */
} pz_flight_tables;
int64_t pz_flight_table_bytes(int32_t which);   /* 0: landing, 1: power_hit */
int pz_build_flight_tables(uint16_t *landing, uint16_t *power_hit, void *stream);

/* ---- introspection -------------------------------------------------------------------- */
int pz_abi_version(void);
int pz_state_words(void);           /* = PZ_STATE_WORDS */
int pz_obs_dim(void);               /* = PZ_OBS_DIM */
int pz_config_bytes(void);          /* = sizeof(pz_config), for binding self-checks */
const char *pz_error_string(int code);
/* hex digest of the sources this library was compiled from (pz_kernels.hip, pz_physics.hpp, this header),
 * baked in by pika-zoo_amd/build.py; the Python binding refuses a library whose id differs from the tree's. */
const char *pz_build_id(void);

/* ---- raw_env.__init__ : pikazoo_env.py:79-141 -> PikaPhysics physics.py:107-123 ---------
 * Fresh state for n games; consumes env-RNG draws 0,1 (the two boldness draws). */
int pz_init(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg, void *stream);

/* ---- raw_env.reset : pikazoo_env.py:149-173 ---------------------------------------------
 * reset() semantics in place on lanes with mask[i] != 0 (mask == NULL: every lane); scores,
 * flags and per-round fields are re-initialised, carry-over fields are kept, draws continue
 * from the lane's counter.  Observations of ALL lanes are written (obs_* may be NULL). */
int pz_reset(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
             const uint8_t *mask, int32_t *obs_p1, int32_t *obs_p2, void *episode_stats,
             void *stream);

/* ---- raw_env._get_obs : pikazoo_env.py:576-624 (normalize = the observation format, as cfg->normalize_obs) */
int pz_observe(const int32_t *state, int64_t n, int64_t stride, int32_t normalize, int32_t packed,
               int32_t *obs_p1, int32_t *obs_p2, void *stream);

/* ---- raw_env.step : pikazoo_env.py:175-240 (one frame of every game, one launch) ---------
 * act_p1/act_p2: [n] elements of cfg->action_format (int32 by default) in [0,18) (or [0,13) with simplify_action).
 * rew_p1/rew_p2: int32[n] (+1/-1/0), or float32[n] when cfg->ballpos_reward or cfg->normal_state_mode.
 * terminated:    uint8[n] = game_ended after this frame (terminations of both agents);
 *                truncations are always False in the reference (:234) and are not written.
 * episode_stats: NULL, or 20 * stride bytes (8-byte aligned): double[2][stride] = running episode return of
 *                player 1 / player 2, then int32[stride] = episode length -- what RecordEpisodeStatistics
 *                reports as infos[agent]["episode"] = {"r", "l"} on a terminal frame; zeroed by reset (pz_reset
 *                or the in-place auto reset). Used when cfg->episode_stats_mode != 0.  The returns are summed
 *                in float64 like the reference's Python floats (record_episode_statistics.py:31); float32
 *                rewards are widened before the add.
 * With cfg->normalize_obs == 1 the observation buffers receive float32 bit patterns, with 2 int16 rows.
 * tables: NULL, or the flight look-up tables above (used when a player is the computer). */
int pz_step(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
            const void *act_p1, const void *act_p2,
            int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
            uint8_t *terminated, void *episode_stats, const pz_flight_tables *tables,
            void *stream);

/* ---- pz_step with its arguments prepared once ---------------------------------------------
 * A per-step caller (raw_env.step, pikazoo_env.py:175-240, is called once per frame) hands over the same twelve
 * buffers and the same configuration every time; only the two action vectors and the stream change.  pz_step_bind
 * validates and records everything else in a caller-provided HOST block of pz_step_bound_bytes() bytes (plain data:
 * it may be copied or freed at will, the library keeps no reference and allocates nothing), pz_step_bound is then
 * exactly pz_step on the recorded arguments.  What it saves is host time per step (an FFI call marshals 4 instead
 * of 14 arguments and nothing is re-validated): where the launch lasts 5-7 us that is what decides whether the
 * host or the GPU sets the step rate.  `cfg` and `tables` are copied at bind time: re-bind after changing them. */
int64_t pz_step_bound_bytes(void);
int pz_step_bind(void *bound, int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                 int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                 uint8_t *terminated, void *episode_stats, const pz_flight_tables *tables);
int pz_step_bound(const void *bound, const void *act_p1, const void *act_p2, void *stream);

/* ---- the same frame with the uniform random policy drawn on device ----------------------
 * actions of game g at step t come from Philox4x32-10(key=action_seed,
 * ctr=(g.lo, g.hi, t.lo, 1 + 2*t.hi)): words 0/1 -> player 1/2, bounded by (u32*n_act)>>32.
 * k >= 1 frames are run back to back inside ONE launch with the state held in registers
 * (steps t0 .. t0+k-1); outputs hold the LAST frame, exactly as k calls of pz_step would
 * leave them.  episodes_done (int64[1], device, may be NULL) is incremented by the number
 * of games that terminated (atomicAdd; used for the aggregate counters). */
int pz_step_random(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                   uint64_t action_seed, uint64_t t0, int32_t k,
                   int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                   uint8_t *terminated, void *episode_stats, int64_t *episodes_done,
                   const pz_flight_tables *tables, void *stream);

/* ---- a k-frame rollout of the random policy with EVERY frame's outputs kept ---------------
 * Same trajectories as k calls of pz_step_random(k=1), in ONE launch: the state is read once,
 * held in registers for k frames and written once; frame t (0 <= t < k) writes
 *   actions[t][2][n] (int32, the policy's draws; may be NULL), obs_p1/obs_p2[t][n][35],
 *   rew_p1/rew_p2[t][n], terminated[t][n].
 * n must be a multiple of 4 when k > 1 (16-byte alignment of every frame's observation slab).
 * This is what a `for t in range(k): env.step(sample())` collection loop around the reference
 * produces (pikazoo_env.py:175-240 with action_space.sample()). */
int pz_rollout_random(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                      uint64_t action_seed, uint64_t t0, int32_t k, int32_t *actions,
                      int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                      uint8_t *terminated, void *episode_stats, int64_t *episodes_done,
                      const pz_flight_tables *tables, void *stream);

/* ---- k frames of GIVEN actions in one launch, every frame's outputs kept -------------------
 * actions: int32[k][2][n] (frame, agent, game), or int64[k][2][n] with cfg->action_format = PZ_ACT_I64 (the other
 * formats: PZ_E_CONFIG -- widen a uint8 / int16 tape to int32 first, which cannot wrap) -- e.g. a recorded action tape
 * or an open-loop plan; outputs as in pz_rollout_random.  Identical to k calls of pz_step on the k slices.
 * n must be a multiple of 4 when k > 1. */
int pz_step_many(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                 const void *actions, int32_t k,
                 int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                 uint8_t *terminated, void *episode_stats, int64_t *episodes_done,
                 const pz_flight_tables *tables, void *stream);

/* ---- the policy stream alone (for hosts that want the actions in HBM) -------------------- */
int pz_random_actions(int32_t *act_p1, int32_t *act_p2, int64_t n, int64_t env_id_base,
                      uint64_t action_seed, uint64_t t, int32_t n_actions, void *stream);

/* ---- raw_env.render, render_mode="rgb_array" : pikazoo_env.py:250-384 ------------------------
 * The frame of game lanes[j] (lanes == NULL: game j) for j < m, in the order of raw_env.draw (:250-255): background
 * (draw_background :296-325, composed once by the host into `background`, RGBA8 [304][432]) -> clouds and waves
 * (draw_clouds_and_wave :338-353, only when `scenery` is given, see below) -> players and their shadows (draw_player :257-278: sprite
 * get_frame_number_for_player_animated_sprite(state, frame_number) :46-68, mirrored by the diving rules :263-264,
 * centred on (x, y); shadows centred on (x, 273)) -> ball (draw_ball :280-290: ball[rotation] with
 * rotation = fine_rotation // 10 physics.py:388, shadow, and on a power hit the hyper ball / trail at the two
 * previous positions) -> score boards (:327-336).  Sprites are RGBA8 (R | G<<8 | B<<16 | A<<24) in `atlas`,
 * described by `sprites[PZ_SPRITE_COUNT]` (device memory); blits use pygame's per-pixel-alpha rule
 * dC = (((sC - dC) * sA + sC) >> 8) + dC.  The punch effect (:292-294) is drawn with `scenery` only (below).
 * frames: uint8 [m][304][432][3].
 *
 * Clouds and waves (cloud_and_wave.py) are state OUTSIDE the 44 words, owned by the reference's renderer and driven by
 * the env RNG: the constructor of an env with a render_mode draws the ten clouds (get_all_image :475-477: 40 draws right
 * behind the two boldness draws), and EVERY render() call runs cloud_and_wave_engine (:53-78) first, which draws 27 wave
 * heights and a few cloud respawns from the same stream -- so in the reference rendering changes the game's later
 * random draws.  `scenery` reproduces exactly that: int32[PZ_SCENERY_WORDS][stride], field-major like the state
 * (cloud i at words 4i..4i+3: top_left_point_x, top_left_point_y, top_left_point_x_velocity, size_diff_turn_number;
 * 40 wave.vertical_coord, 41 its velocity, 42..68 wave.y_coords; 69 ball.punch_effect_radius, 70 ball.punch_effect_y;
 * 71..74 what pz_scenery_track remembers of the previous frame: both is_collision_with_ball_happened flags, game_ended,
 * round_ended).
 * pz_scenery_init = the constructor's part (call it between pz_init and the first pz_reset); pz_render with
 * scenery != NULL runs the engine for the rendered games (advancing their rng draw counter in `state`; `lanes` must then
 * hold distinct games and cfg must be given) and draws the clouds -- scaled like pygame.transform.scale: source pixel =
 * floor(k * source size / scaled size) -- the waves and the punch effect.  With scenery == NULL nothing is written to
 * `state` (cfg may be NULL) and none of the three is drawn.
 * The punch effect's radius and y are two ball attributes outside the 44 state words: the physics sets them when the
 * ball touches the ground (physics.py:427-430) or is power-hit (:628-632), render() counts the radius down by 2 per call
 * (:292-294).  pz_scenery_track, called after EVERY pz_step of a batch whose scenery is kept, re-derives those two
 * events from the state the step left (ground touch = round_ended; power hit = a player's collision flag rising while
 * its state is 2; a frame that started a new round clears the radius first, :274-275) and sets radius / y like the
 * physics does; after pz_reset the caller clears words 69 and 71..74 of the reset games; `resync` != 0 (after a k-frame launch, whose inner frames
 * it cannot see) only clears the effect and re-reads the flags.
 * All three calls take the int32 columns (a packed state is converted by the caller). */
#define PZ_FRAME_WIDTH 432
#define PZ_FRAME_HEIGHT 304
enum pz_sprite_id {
    PZ_SPRITE_PIKACHU = 0,        /* 28: pikachu_<state>_<frame>.png in the order of get_all_image :445-474 */
    PZ_SPRITE_BALL = 28,          /* 6: ball_0..4, ball_hyper (the tuple `self.ball`, :395-402) */
    PZ_SPRITE_BALL_HYPER = 33,
    PZ_SPRITE_BALL_TRAIL = 34,
    PZ_SPRITE_SHADOW = 35,
    PZ_SPRITE_NUMBER = 36,        /* 10: number_0..9 */
    PZ_SPRITE_CLOUD = 46,
    PZ_SPRITE_WAVE = 47,
    PZ_SPRITE_BALL_PUNCH = 48,
    PZ_SPRITE_COUNT = 49
};
#define PZ_SCENERY_WORDS 75
typedef struct pz_sprite {
    int32_t offset;   /* first pixel in the atlas */
    int32_t width, height;
} pz_sprite;
int pz_scenery_init(int32_t *scenery, int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                    void *stream);
int pz_scenery_track(int32_t *scenery, const int32_t *state, int64_t n, int64_t stride, const pz_config *cfg,
                     int32_t resync, void *stream);
int pz_render(int32_t *state, int64_t n, int64_t stride, const pz_config *cfg, const int32_t *lanes, int64_t m,
              const uint32_t *atlas, const pz_sprite *sprites, const uint32_t *background,
              int32_t *scenery, uint8_t *frames, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKAZOO_HIP_H */
