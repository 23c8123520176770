/* pikazoo_diag.h -- C ABI of libpikazoo_diag.so: DIAGNOSTICS, not product.
 *
 * Two entry points that measure / check the product's device code from outside (they are compiled from the product's
 * own headers, pika-zoo_amd/csrc/pz_physics.hpp and pz_memory.hpp) and that a maintainer of the reference would never
 * bind: include/pikazoo_hip.h is the drop-in boundary, this file is for bench.py and tests/ only.  The library is
 * built beside libpikazoo_hip.so by pika-zoo_amd/build.py; nothing in libpikazoo_hip.so or in the pikazoo_amd package
 * loads it.  Return codes are pikazoo_hip.h's (PZ_OK, PZ_E_*).
 */
#ifndef PIKAZOO_DIAG_H
#define PIKAZOO_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* source digest this library was compiled from: equals pz_build_id() of the product library built beside it */
const char *pz_diag_build_id(void);

/* ---- launch-floor probe -------------------------------------------------------------------------
 * One launch with pz_step's GEOMETRY for n games on int32 columns (ceil(n / 64) workgroups of two waves, the same LDS,
 * the same seven buffers) and NONE of its game logic -- what a caller (bench.py, tools/launch_floor.hip is the
 * standalone form) replays as a chain of dependent launches beside the real one to see what that launch is made of
 * (DESIGN.md section 4.4).  `what`:
 *   0  nothing at all: what one launch of a dependent chain costs on this runtime;
 *   1  each wave loads what the pair kernel's waves load (the own player's 13 columns, the ball's 12, the env's 6, both
 *      action words) and keeps them alive;
 *   2  ... and stores what a frame always stores: 10 state columns per wave, its agent's reward, its agent's observation
 *      rows staged in LDS and flushed as 16-byte pieces (`nt`), nothing computed in between;
 *   3  ... with `frame_steps` steps of the frame's own idiom per wave between the loads and the stores (two compares
 *      into SGPR masks, an s_and_b64, a v_cndmask_b32 on it, an add: 4 VALU + 1 SALU per step, every step depending on
 *      the one before), one LDS exchange and one workgroup barrier half way: 102 steps = the 408 VALU instructions a
 *      wave of the shipped human-vs-human frame issues (that count runs as straight-line code like the frame, any
 *      other in a loop of ten steps per trip).
 * The buffers are the shapes pz_step takes (state int32[44][stride >= n], act int32[n], obs int32[n][35], rew int32[n]);
 * from `what` = 2 on state, rewards and observations are OVERWRITTEN with meaningless values: hand it scratch buffers.
 * No reference counterpart; nothing in the product calls it. */
int pz_probe_launch(int32_t *state, int64_t n, int64_t stride, const int32_t *act_p1, const int32_t *act_p2,
                    int32_t *obs_p1, int32_t *obs_p2, int32_t *rew_p1, int32_t *rew_p2, int32_t what,
                    int32_t frame_steps, void *stream);

/* ---- self-test hook ------------------------------------------------------------------------
 * The computer player's flight predictors (calculate_expected_landing_point_x_for
 * physics.py:643-686 when full_net != 0, expected_landing_point_x_when_power_hit
 * physics.py:848-884 otherwise) evaluated two ways on n caller-supplied ball states
 * (x, y, x_velocity, y_velocity; for the power-hit form the velocities are the already
 * substituted ones): out_fast = the closed-form fast-forward the step kernel uses, out_iter =
 * the frame-by-frame iteration of the reference.  They must be identical. */
int pz_selftest_predictor(const int32_t *x, const int32_t *y, const int32_t *xv, const int32_t *yv,
                          int64_t n, int32_t full_net, int32_t *out_fast, int32_t *out_iter,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIKAZOO_DIAG_H */
