// pz_physics.hpp -- per-lane game logic of the fused step kernel (device code, gfx950).
//
// One lane = one independent Pikachu-Volleyball game held entirely in registers.  All
// arithmetic is int32; there is no floating point on the path except the optional
// fused reward wrappers' adds.  Written for SIMT: short bodies are predicated selects, the
// only real loops are the two ball-flight predictors of the rule-based computer player -- and
// those are normally replaced by look-ups in the flight tables (FlightLut).
//
// Behavioural spec: helpingstar/pika-zoo @ 2024_10_08 (citations `file:line` are relative
// to the reference checkout and name the rule each block implements).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pikazoo_hip.h"
#include "pz_diagnostic.hpp"

namespace pz {

// games (active lanes) per workgroup: a full wavefront (32 -- half-filled waves, twice as many of them -- lost on every
// kernel it was tried on: human vs human 6.97 -> 7.49 us per launch, profiles/r04_experiments/ab_g32_and_early_stores.log)
constexpr int kWaveGames = 64;

// pikazoo/env/physics.py:9-33
constexpr int kGroundWidth = 432;
constexpr int kGroundHalfWidth = 216;
constexpr int kPlayerLength = 64;
constexpr int kPlayerHalfLength = 32;
constexpr int kPlayerGroundY = 244;
constexpr int kBallRadius = 20;
constexpr int kBallGroundY = 252;
constexpr int kNetPillarHalfWidth = 25;
constexpr int kNetTopTopY = 176;
constexpr int kNetTopBottomY = 192;
constexpr int kLoopLimit = 1000;

struct Player {
    int x, y, yv, state, frame, arm, delay, dive, lying, coll, bold, standby, hitprev;
};
struct Ball {
    int x, y, xv, yv, power, px, py, ppx, ppy, rot, ex, punch;
};
struct Env {
    int s1, s2, p2serve, round_ended, game_ended;
    uint32_t rng;  // env-stream draw counter
};
struct Game {
    Player p1, p2;
    Ball b;
    Env e;
};
struct Input {
    int xd, yd, hit;
};
// A Philox key and how its ten round keys (k + r * Weyl constant) are held.  They are wave-uniform values:
//   rolling  (straight-line kernels) only the key itself is kept, every block steps through its round keys with
//            scalar adds -- what the compiler schedules best there (materialising the twenty of them once costs the
//            pair kernel 26 more SGPRs and 1.5-4 % per launch);
//   parked   (the k-frame kernels) in a frame loop the compiler hoists all twenty and then spills them -- two keys
//            are 40 SGPRs of the 102, and every spilled round key comes back through a v_readlane in front of its
//            xor -- so the loop kernels hold their schedules in VGPRs (park_in_vgprs), of which a lone wave per SIMD
//            has hundreds to spare.
// `rolling` is a compile-time constant wherever it is read (everything here is inlined).
struct KeySchedule {
    bool rolling;
    uint32_t k0[10], k1[10];  // rolling: only [0] is set
};

__device__ __forceinline__ KeySchedule make_rolling_key(uint64_t key)
{
    KeySchedule ks{};
    ks.rolling = true;
    ks.k0[0] = (uint32_t)key;
    ks.k1[0] = (uint32_t)(key >> 32);
    return ks;
}

__device__ __forceinline__ KeySchedule make_parked_schedule(uint64_t key)
{
    KeySchedule ks{};
    ks.rolling = false;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        asm("v_mov_b32 %0, %1" : "=v"(ks.k0[r]) : "s"(k0));
        asm("v_mov_b32 %0, %1" : "=v"(ks.k1[r]) : "s"(k1));
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ks;
}

// per-lane RNG identity (Philox counter words 0,1) and the env stream's key
struct RngId {
    uint32_t id_lo, id_hi;
    KeySchedule ks;
};

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Only words 0 and 1 of the output block are used.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // one v_bitop3_b32 (gfx950) instead of two v_xor_b32
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              const KeySchedule& ks, uint32_t& o0, uint32_t& o1)
{
    uint32_t r0 = ks.k0[0], r1 = ks.k1[0];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per lane pair (v_mad_u64_u32) instead of separate hi/lo products
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = xor3((uint32_t)(p1 >> 32), c1, ks.rolling ? r0 : ks.k0[r]);
        c1 = (uint32_t)p1;
        c2 = xor3((uint32_t)(p0 >> 32), c3, ks.rolling ? r1 : ks.k1[r]);
        c3 = (uint32_t)p0;
        r0 += 0x9E3779B9u;
        r1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

// np_random.integers(0, n): draw number `counter` of this game's env stream
// (ctr = (id.lo, id.hi, counter, 0)), bounded by (u32 * n) >> 32; one draw = one tick.
__device__ __forceinline__ int rng_integers(const RngId& id, uint32_t& counter, uint32_t n)
{
    uint32_t o0, o1;
    philox4x32_10(id.id_lo, id.id_hi, counter, 0u, id.ks, o0, o1);
    counter += 1;
    return (int)__umulhi(o0, n);
}

// uniform random policy: both players' actions of one game at step t (`policy`: the schedule of the action seed)
__device__ __forceinline__ void policy_actions(uint32_t id_lo, uint32_t id_hi, const KeySchedule& policy,
                                               uint64_t t, uint32_t n_actions, int& a1, int& a2)
{
    uint32_t o0, o1;
    philox4x32_10(id_lo, id_hi, (uint32_t)t, 1u + 2u * (uint32_t)(t >> 32), policy, o0, o1);
    a1 = (int)__umulhi(o0, n_actions);
    a2 = (int)__umulhi(o1, n_actions);
}

// ---------------------------------------------------------------------------------------
// Action decode: action id -> (x_direction, y_direction, fire) as packed bit tables, built
// at compile time from the key rows [left,right,up,down,power_hit] of
// pikazoo_env.py:119-141 and the precedence rules of PikaUserInput.get_input
// (physics.py:80-92: left beats right, up beats down; rows have 5 entries so there is no
// down_right key, :69-70).  Two bits per action for each direction (value+1), one for fire.
// ---------------------------------------------------------------------------------------
struct ActionTables {
    uint64_t xd, yd;
    uint32_t fire;
};

constexpr ActionTables make_tables(const int* ids, int count)
{
    // keys per action id 0..17, bit0=left bit1=right bit2=up bit3=down bit4=power_hit
    constexpr int kKeys[18] = {0x00, 0x10, 0x04, 0x02, 0x01, 0x08, 0x06, 0x05, 0x0A,
                               0x09, 0x14, 0x12, 0x11, 0x18, 0x16, 0x15, 0x1A, 0x19};
    ActionTables t{0, 0, 0};
    for (int i = 0; i < count; ++i) {
        const int k = kKeys[ids[i]];
        const int xd = (k & 1) ? -1 : ((k & 2) ? 1 : 0);
        const int yd = (k & 4) ? -1 : ((k & 8) ? 1 : 0);
        t.xd |= (uint64_t)(xd + 1) << (2 * i);
        t.yd |= (uint64_t)(yd + 1) << (2 * i);
        t.fire |= (uint32_t)((k >> 4) & 1) << i;
    }
    return t;
}

constexpr int kAllActions[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};
// wrappers/simplify_action.py:17-18
constexpr int kSimplifyP1[13] = {0, 1, 2, 3, 4, 6, 7, 10, 11, 12, 13, 14, 16};
constexpr int kSimplifyP2[13] = {0, 1, 2, 4, 3, 7, 6, 10, 12, 11, 13, 15, 17};
constexpr ActionTables kFullTables = make_tables(kAllActions, 18);
constexpr ActionTables kSimpleTablesP1 = make_tables(kSimplifyP1, 13);
constexpr ActionTables kSimpleTablesP2 = make_tables(kSimplifyP2, 13);

// get_input (physics.py:59-99): direction decode + power-hit edge detection
__device__ __forceinline__ Input decode_action(const ActionTables& t, int action, int& hitprev)
{
    Input in;
    const uint32_t sh = (uint32_t)action;
    in.xd = (int)((t.xd >> (2u * sh)) & 3u) - 1;
    in.yd = (int)((t.yd >> (2u * sh)) & 3u) - 1;
    const int fire = (int)((t.fire >> sh) & 1u);
    in.hit = fire & (hitprev ^ 1);
    hitprev = fire;
    return in;
}

// ---------------------------------------------------------------------------------------
// Round / game initialisation
// ---------------------------------------------------------------------------------------
// Player.initialize_for_new_round (physics.py:181-218); boldness is drawn for humans too.
__device__ __forceinline__ void player_new_round_undrawn(Player& p, int x0)
{
    p.x = x0;
    p.y = kPlayerGroundY;
    p.yv = 0;
    p.coll = 0;
    p.state = 0;
    p.frame = 0;
    p.arm = 1;
    p.delay = 0;
}

__device__ __forceinline__ void player_new_round(Player& p, int x0, const RngId& id, uint32_t& rng)
{
    player_new_round_undrawn(p, x0);
    p.bold = rng_integers(id, rng, 5u);
}

// Ball.initialize_for_new_round (physics.py:258-277): only these five fields are touched.
__device__ __forceinline__ void ball_new_round(Ball& b, int p2_serves)
{
    b.x = p2_serves ? kGroundWidth - 56 : 56;
    b.y = 0;
    b.xv = 0;
    b.yv = 1;
    b.power = 0;
}

// raw_env.get_server (pikazoo_env.py:242-248); "random": integers(0,2)==0 => player 2 serves
__device__ __forceinline__ int get_server(const pz_config& cfg, Env& e, const RngId& id)
{
    if (cfg.serve_mode == PZ_SERVE_WINNER) return e.p2serve;
    if (cfg.serve_mode == PZ_SERVE_RANDOM) return rng_integers(id, e.rng, 2u) == 0;
    return ((e.s1 + e.s2) & 1) == 1;
}

// players + ball for a new round; shared by reset() (pikazoo_env.py:162-164) and the
// start-of-step new-round branch (:176-180)
__device__ __forceinline__ void start_round(Game& g, const pz_config& cfg, const RngId& id)
{
    player_new_round(g.p1, 36, id, g.e.rng);
    player_new_round(g.p2, kGroundWidth - 36, id, g.e.rng);
    ball_new_round(g.b, get_server(cfg, g.e, id));
}

// raw_env.__init__ (pikazoo_env.py:96-111 -> physics.py:120-123,159-171,232-249)
__device__ __forceinline__ void construct_game(Game& g, const RngId& id)
{
    g.e.rng = 0;
    player_new_round(g.p1, 36, id, g.e.rng);
    player_new_round(g.p2, kGroundWidth - 36, id, g.e.rng);
    g.p1.dive = g.p2.dive = 0;
    g.p1.lying = g.p2.lying = -1;
    g.p1.standby = g.p2.standby = 0;
    g.p1.hitprev = g.p2.hitprev = 0;
    ball_new_round(g.b, 0);
    g.b.px = g.b.py = g.b.ppx = g.b.ppy = 0;
    g.b.rot = 0;
    g.b.ex = 0;
    g.b.punch = 0;
    g.e.s1 = g.e.s2 = 0;
    g.e.p2serve = g.e.round_ended = g.e.game_ended = 0;
}

// raw_env.reset (pikazoo_env.py:149-173): flags + scores cleared, then a new round.  The
// `seed` argument is ignored by the reference; carry-over fields stay untouched.
__device__ __forceinline__ void reset_game(Game& g, const pz_config& cfg, const RngId& id)
{
    g.e.game_ended = 0;
    g.e.round_ended = 0;
    g.e.p2serve = 0;
    g.e.s1 = 0;
    g.e.s2 = 0;
    start_round(g, cfg, id);
}

// ---------------------------------------------------------------------------------------
// Ball vs world (physics.py:359-436).  Returns true when the ball touches the ground.
// ---------------------------------------------------------------------------------------
// Written as selects, not branches: in a wave of 64 unsynchronised games every branch is taken
// by some lane anyway, and a lone wave per SIMD pays an issue slot for every exec-mask
// instruction around it (tools/stamps.py: the branchy form of this function cost 550 cycles).
__device__ __forceinline__ bool ball_world_step(Ball& b)
{
    b.ppx = b.px;
    b.ppy = b.py;
    b.px = b.x;
    b.py = b.y;

    int rot = b.rot + (b.xv >> 1);  // Python floor division (:373): arithmetic shift
    rot = rot < 0 ? rot + 50 : (rot > 50 ? rot - 50 : rot);
    b.rot = rot;

    const int x = b.x, y = b.y;
    int xv = b.xv, yv = b.yv;
    const int fx = x + xv;
    xv = ((unsigned)(fx - kBallRadius) > (unsigned)(kGroundWidth - kBallRadius)) ? -xv : xv;  // asymmetric walls (:403)
    yv = (y + yv < 0) ? 1 : yv;

    // net pillar (:411-419): above its top edge bounce the fall back, below it push sideways
    const bool at_net = ((unsigned)(x - (kGroundHalfWidth - kNetPillarHalfWidth + 1)) <
                         (unsigned)(2 * kNetPillarHalfWidth - 1)) & (y > kNetTopTopY);
    const bool on_top = y <= kNetTopBottomY;
    const int axv = abs(xv);
    yv = (at_net & on_top & (yv > 0)) ? -yv : yv;
    xv = (at_net & !on_top) ? ((x < kGroundHalfWidth) ? -axv : axv) : xv;

    const int fy = y + yv;
    const bool ground = fy > kBallGroundY;
    b.punch = ground ? x : b.punch;
    b.y = ground ? kBallGroundY : fy;  // x is not advanced on the touching frame (:428-431)
    b.x = ground ? x : x + xv;
    b.yv = ground ? -yv : yv + 1;
    b.xv = xv;
    return ground;
}

// ---------------------------------------------------------------------------------------
// Flight predictors of the computer player.
// FULL_NET=true : calculate_expected_landing_point_x_for (physics.py:643-686), net top split
//                 at y < 192 (strict) with side bounce below it.
// FULL_NET=false: expected_landing_point_x_when_power_hit (physics.py:848-884), simplified
//                 net rule (no side bounce).
// Both keep the reference's iteration cap (physics.py:33): the x at the cap is the result.
// ---------------------------------------------------------------------------------------
// The frame-by-frame form, exactly as the reference iterates it.  Kept as the in-library
// cross-check of the fast-forward form below (pz_selftest_predictor).
template <bool FULL_NET>
__device__ __forceinline__ int predict_landing_x_iterative(int x, int y, int xv, int yv)
{
    int count = 0;
    for (;;) {
        ++count;
        const int fx = x + xv;
        if (fx < kBallRadius || fx > kGroundWidth) xv = -xv;
        if (y + yv < 0) yv = 1;
        if (abs(x - kGroundHalfWidth) < kNetPillarHalfWidth && y > kNetTopTopY) {
            if (!FULL_NET || y < kNetTopBottomY) {
                if (yv > 0) yv = -yv;
            } else {
                xv = (x < kGroundHalfWidth) ? -abs(xv) : abs(xv);
            }
        }
        y += yv;
        if (y > kBallGroundY || count >= kLoopLimit) break;
        x += xv;
        yv += 1;
    }
    return x;
}

// Fast-forward form.  On a GPU the predictor's cost is its longest lane (a wave -- and at one
// wave per SIMD the whole launch -- waits for the slowest flight), so the loop runs once per
// *event* of the flight (wall flip, ceiling clamp, hit on the net box, landing) instead of once
// per frame: per wave and launch the longest lane needs 5.7 trips instead of 56 iterations for
// the landing predictor and 4.3 instead of 54 for the power-hit candidates (tools/flight_trips.c,
// states sampled from play).
//
// With Y(m) = y + m*yv + m(m-1)/2 and X(m) = x + m*xv the state after m "plain" iterations (no
// wall flip, no ceiling clamp, no box hit, no landing, cap not reached), iterations 1..K are all
// plain exactly when
//     20 <= X(m) <= 432   for m = 1..K    (linear: check m = K; X(0) is in range)
//     0  <= Y(m)          for m = 1..K    (convex: check its lowest point m = clamp(-yv, 1, K);
//                                           y itself can be negative -- the net-top bounce runs
//                                           after the ceiling clamp and can throw a very fast
//                                           ball above 0)
//     Y(m) <= 252         for m = 0..K    (convex: check both ends)
//     Y(m) <= 176 for every m in 0..K-1 with 192 <= X(m) <= 240   (iteration m+1 tests the box at
//                                           position m.)  X is linear, so these m form one interval
//                                           [m1, m2]; Y is convex, so its ends suffice.  The
//                                           power-hit form's box only acts on a falling ball, so
//                                           there the interval starts no earlier than m = 1 - yv.
//     count + K <= 998                    (the cap test of iterations 0..K-1 stays false)
// Then x = X(K), y = Y(K), yv += K, count += K, and the single reference iteration that follows is
// the event itself.  K, m1 and m2 are *proposed* in float (distances times 1/|xv|; roots of
// Y(K) = 252 or, for a ball whose apex would cross the ceiling, of Y(K) = 0; if the ball would be
// inside the box on its way, the first offending position instead) and *verified* with the exact
// integer conditions above (for m1 and m2: X(m1 - 1) has not reached the columns, X(m2 + 1) has
// left them); a failed verification falls back to the single iteration, so the result is
// identical to the iterative form by construction.  pz_selftest_predictor compares the two over
// the whole input domain on the GPU.
// 24-bit multiplies (full-rate v_mul_i32_i24 / v_mad_i32_i24; every operand here is far below 2^23)
__device__ __forceinline__ int mul24(int a, int b)
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int flight_height(int y, int yv, int m)
{
    return y + mul24(m, yv) + (mul24(m, m - 1) >> 1);
}

// far root K of Y(K) = target (near root if `near`), rounded down with a small safety margin
__device__ __forceinline__ int flight_root(float hb, float hb2, int dy, bool near)
{
    const float root = __builtin_amdgcn_sqrtf(fmaxf(fmaf(8.0f, (float)dy, hb2), 0.0f));
    return (int)(((near ? -root : root) - hb) * 0.5f - 0.001f);
}

template <bool FULL_NET>
__device__ __forceinline__ int predict_landing_x(int x, int y, int xv, int yv)
{
    constexpr int kColsFirst = kGroundHalfWidth - kNetPillarHalfWidth + 1;  // 192: |x - 216| < 25
    constexpr int kColsLast = kGroundHalfWidth + kNetPillarHalfWidth - 1;   // 240
    // Inputs outside what the 24-bit products are sized for, or below the ground line (neither is
    // produced by play), take the plain loop.  Inside the range |yv| cannot grow past it: a flight
    // only gains speed while falling 252 rows at most.
    if (abs(yv) >= 2048 || y > kBallGroundY) return predict_landing_x_iterative<FULL_NET>(x, y, xv, yv);
    const int axv = abs(xv);  // invariant: the rules only ever flip the sign of xv
    const float r = __builtin_amdgcn_rcpf((float)max(axv, 1));
    int count = 0;
    for (;;) {
        // ---- proposal + verification, written branch-free: in a divergent wave every trip pays
        // for every path anyway, and straight-line code lets the lone wave overlap the
        // transcendental chains (rcp, sqrt) with the integer work.
        const bool rightward = xv > 0;
        // plain moves available along x up to the wall (a still ball: `room`, which is harmless)
        const int room = rightward ? kGroundWidth - x : x - kBallRadius;
        const int kw = (int)fmaf((float)room, r, 0.001f);
        // plain moves available along y: far root of Y(K) = 252, or, if the apex Y(-yv) would be
        // above the ceiling, the near root of Y(K) = 0 (which comes first)
        const float hb = (float)(2 * yv - 1), hb2 = hb * hb;
        const int apex = y - (mul24(yv, yv - 1) >> 1);
        const bool to_ceiling = (yv < 0) & (apex < 0);
        const int ky = flight_root(hb, hb2, to_ceiling ? -y : kBallGroundY - y, to_ceiling);
        int K = max(min(min(kw, ky), kLoopLimit - 2 - count), 0);
        // positions m1..m2 inside the columns of the net box, in travel coordinates: the ball is d1
        // short of the first column and d2 short of the last one and advances |xv| per iteration
        const int d1 = rightward ? kColsFirst - x : x - kColsLast;
        const int d2 = d1 + (kColsLast - kColsFirst);
        const int m1 = max((int)fmaf((float)(d1 + axv - 1), r, 0.001f), 0);
        const int m2 = (int)floorf(fmaf((float)d2, r, 0.001f));  // < 0: already past
        const int lo = FULL_NET ? m1 : max(m1, 1 - yv);
        const int ylo = flight_height(y, yv, lo);
        const bool lo_hits = ylo > kNetTopTopY;
        // first offending position: lo itself, or the first one past the far root of Y = 176 if the
        // ball sinks below the net top while over the box; it limits K if it is inside the columns
        const int first_hit = lo_hits ? lo : max(lo, flight_root(hb, hb2, kNetTopTopY - y, false) + 1);
        K = first_hit <= m2 ? min(K, first_hit) : K;
        // exact verification
        const int ye = flight_height(y, yv, K);
        const int xe = x + mul24(K, xv);
        const int lowest = yv < 0 ? (K >= -yv ? apex : ye) : y + yv;  // Y(clamp(-yv, 1, K))
        const int hi = min(m2, K - 1);
        const bool box_ok = (lo > hi) | (!lo_hits & (flight_height(y, yv, hi) <= kNetTopTopY));
        const bool cols_ok = ((m1 == 0) | (mul24(m1 - 1, axv) < d1)) & ((m2 >= K - 1) | (mul24(m2 + 1, axv) > d2));
        const bool ok = (K >= 1) & ((unsigned)(xe - kBallRadius) <= (unsigned)(kGroundWidth - kBallRadius)) &
                        ((unsigned)ye <= (unsigned)kBallGroundY) & (lowest >= 0) & box_ok & cols_ok;
        x = ok ? xe : x;
        y = ok ? ye : y;
        yv += ok ? K : 0;
        count += ok ? K : 0;

        // ---- one iteration, exactly as the reference
        ++count;
        const int fx = x + xv;
        if (fx < kBallRadius || fx > kGroundWidth) xv = -xv;
        if (y + yv < 0) yv = 1;
        if (abs(x - kGroundHalfWidth) < kNetPillarHalfWidth && y > kNetTopTopY) {
            if (!FULL_NET || y < kNetTopBottomY) {
                if (yv > 0) yv = -yv;
            } else {
                xv = (x < kGroundHalfWidth) ? -abs(xv) : abs(xv);
            }
        }
        y += yv;
        if (y > kBallGroundY || count >= kLoopLimit) break;
        x += xv;
        yv += 1;
    }
    return x;
}

// ---------------------------------------------------------------------------------------
// Flight look-up tables (pz_flight_tables in the header): both predictors are pure functions of a few
// small integers, tabulated once per device by pz_build_flight_tables with the iterative forms above.
// A launch lasts as long as its slowest lane, and with the predictors iterated in the kernel that lane
// is the one 20-event flight among the launch's ~80 000; a table turns every flight into one gather.
// States outside a table's domain (or a NULL table) take the computed path, so results are identical
// with and without tables by construction.
// ---------------------------------------------------------------------------------------
using TableRsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned int lut_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kFtXCount = kGroundWidth - kBallRadius + 1;  // x 20..432
constexpr int kFtYCount = kBallGroundY + 1;                // y 0..252
constexpr int kFtXvCount = 23;                             // -20, -10..10, 20
constexpr int kFtYvCount = 2 * PZ_FT_YV_MAX + 1;
constexpr int kFtHitYMin = 61;                             // the scan needs |ball.y - player.y| < 48, player.y >= 108
constexpr int kFtHitYCount = kBallGroundY - kFtHitYMin + 1;
constexpr int64_t kFtLandingEntries = (int64_t)kFtYvCount * kFtXvCount * kFtYCount * kFtXCount;
// the look-up loads the DWORD that holds an entry (an odd number of 2-byte entries: the buffer ends with two bytes of padding)
constexpr int64_t kFtLandingBytes = (kFtLandingEntries * 2 + 3) & ~(int64_t)3;
constexpr int64_t kFtHitEntries = (int64_t)(PZ_FT_HIT_YV_MAX + 1) * kFtHitYCount * kFtXCount;  // of 8 x uint16

// x velocity -> table row, or -1
__device__ __forceinline__ int ft_xv_index(int xv)
{
    const int a = abs(xv);
    return a <= 10 ? xv + 11 : (a == 20 ? (xv > 0 ? 22 : 0) : -1);
}
__device__ __forceinline__ int ft_xv_value(int index) { return index == 0 ? -20 : (index == 22 ? 20 : index - 11); }

// A look-up in two halves, so that a caller can put independent work between them: `*_issue` computes the entry and
// issues the load unconditionally (a lane outside the table's domain -- or without a table -- reads entry 0; the
// range check of an empty descriptor returns 0), `*_finish` takes the loaded value or, for the rare lane outside
// the domain, runs the computed form.  (A load inside the `inside the domain` branch would be waited for before the
// branch closes: the two gathers of a decision and the Philox blocks between them would run one after the other.)
struct LandingProbe {
    bool in, wanted;
    uint32_t shift;  // 0 / 16: the half of `value` that is the entry
    uint32_t value;  // the table dword holding the entry, exactly as loaded: nothing touches the register before
                     // landing_finish (a 16-bit load's widening would be an instruction waiting for it at the issue site)
};
struct CandidateProbe {
    bool in, wanted;
    lut_u32x4 value;
};

struct FlightLut {
    TableRsrc landing, power_hit;
    bool has_landing, has_power_hit;  // wave-uniform (kernel arguments)

    // calculate_expected_landing_point_x_for (physics.py:643-686)
    // (`wanted` false: the lane needs no prediction; finish returns `keep`)
    // locate: everything but the load (a k-frame launch issues the load in one frame half and re-derives the rest,
    // a pure function of the unmoved ball, in the other: only the loaded register crosses its loop's back edge)
    __device__ __forceinline__ LandingProbe landing_locate(bool wanted, int x, int y, int xv, int yv, uint32_t& offset) const
    {
        const int xi = ft_xv_index(xv);
        const bool in = wanted & has_landing & (xi >= 0) & ((unsigned)(x - kBallRadius) < (unsigned)kFtXCount) &
                        ((unsigned)y < (unsigned)kFtYCount) & (abs(yv) <= PZ_FT_YV_MAX);
        // 24-bit multiplies (every factor is far below 2^24, the products below 2^32): three cheap instructions, so
        // the entry is computed for every lane and selected -- not put behind a branch that the loads would wait at
        uint32_t e = __umul24((uint32_t)(yv + PZ_FT_YV_MAX), (uint32_t)kFtXvCount) + (uint32_t)xi;
        e = __umul24(e, (uint32_t)kFtYCount) + (uint32_t)y;
        e = __umul24(e, (uint32_t)kFtXCount) + (uint32_t)(x - kBallRadius);
        offset = in ? (e * 2u) & ~3u : 0u;
        return LandingProbe{in, wanted, (e & 1u) * 16u, 0u};
    }
    __device__ __forceinline__ LandingProbe landing_issue(bool wanted, int x, int y, int xv, int yv) const
    {
        uint32_t offset;
        LandingProbe p = landing_locate(wanted, x, y, xv, yv, offset);
        p.value = __builtin_amdgcn_raw_buffer_load_b32(landing, offset, 0, 0);
        return p;
    }
    __device__ __forceinline__ int landing_finish(const LandingProbe& p, int x, int y, int xv, int yv, int keep) const
    {
        if (p.wanted & !p.in) return predict_landing_x<true>(x, y, xv, yv);
        return p.in ? (int)((p.value >> p.shift) & 0xFFFFu) : keep;
    }
    __device__ __forceinline__ int landing_x(int x, int y, int xv, int yv) const
    {
        return landing_finish(landing_issue(true, x, y, xv, yv), x, y, xv, yv, 0);
    }

    // the six candidates of decide_whether_input_power_hit (physics.py:796-816) for the ball (x, y, |yv|);
    // `wanted`: this lane's computer player scans this frame
    __device__ __forceinline__ CandidateProbe candidates_locate(bool wanted, int x, int y, int ayv, uint32_t& offset) const
    {
        const bool in = wanted & has_power_hit & ((unsigned)(x - kBallRadius) < (unsigned)kFtXCount) &
                        ((unsigned)(y - kFtHitYMin) < (unsigned)kFtHitYCount) & ((unsigned)ayv <= (unsigned)PZ_FT_HIT_YV_MAX);
        uint32_t e = __umul24((uint32_t)ayv, (uint32_t)kFtHitYCount) + (uint32_t)(y - kFtHitYMin);
        e = __umul24(e, (uint32_t)kFtXCount) + (uint32_t)(x - kBallRadius);
        offset = in ? e * 16u : 0u;
        return CandidateProbe{in, wanted, lut_u32x4{0u, 0u, 0u, 0u}};
    }
    __device__ __forceinline__ CandidateProbe candidates_issue(bool wanted, int x, int y, int ayv) const
    {
        uint32_t offset;
        CandidateProbe p = candidates_locate(wanted, x, y, ayv, offset);
        p.value = __builtin_amdgcn_raw_buffer_load_b128(power_hit, offset, 0, 0);
        return p;
    }
    __device__ __forceinline__ void candidates_finish(const CandidateProbe& p, int x, int y, int ayv, int (&ex)[6]) const
    {
        // the row's fourth dword is padding: keep its register reserved until here all the same -- handed out as a
        // temporary while the load is in flight, the first write to it would have to wait for the whole gather
        asm volatile("" ::"v"(p.value.w));
        ex[0] = (int)(p.value.x & 0xFFFFu);
        ex[1] = (int)(p.value.x >> 16);
        ex[2] = (int)(p.value.y & 0xFFFFu);
        ex[3] = (int)(p.value.y >> 16);
        ex[4] = (int)(p.value.z & 0xFFFFu);
        ex[5] = (int)(p.value.z >> 16);
        if (p.wanted & !p.in) {
#pragma unroll 1
            for (int c = 0; c < 6; ++c) {
                const int xdir = c < 3 ? 1 : 0, ydir = (c < 3 ? c : c - 3) - 1;
                const int sxv = (x < kGroundHalfWidth) ? (xdir + 1) * 10 : -(xdir + 1) * 10;  // :841-844
                const int e = predict_landing_x<false>(x, y, sxv, ayv * ydir * 2);          // :845
#pragma unroll
                for (int k = 0; k < 6; ++k) ex[k] = (k == c) ? e : ex[k];  // (no dynamic register indexing)
            }
        }
    }
    __device__ __forceinline__ void power_hit_candidates(int x, int y, int ayv, int (&ex)[6]) const
    {
        candidates_finish(candidates_issue(true, x, y, ayv), x, y, ayv, ex);
    }
};

// ---------------------------------------------------------------------------------------
// Rule-based computer player: let_computer_decide_user_input (physics.py:689-771) with
// decide_whether_input_power_hit (:774-817).  IS_P2 selects the court side.
//
// The decision is split in three so that the expensive part -- up to six power-hit flight
// predictions per deciding player -- is shared by the whole wavefront instead of being run
// one after the other by the few lanes that need it:
//   begin  (per lane)   everything up to the point where the six candidates are needed,
//                       including the RNG draws in the reference's order (:728,:729,:795);
//   candidates (wave)   every (deciding lane, candidate) pair becomes one work item; items are
//                       dealt to the 64 lanes, so a wave's cost is the longest single flight,
//                       not the sum of six;
//   finish (per lane)   first candidate in the drawn scan order that lands on the opponent's
//                       side and > 64 away from the opponent wins (:796-816), then :768-771.
// The candidates are pure functions of the ball, so evaluating all six is result-identical
// to the reference's early-exit scan.
// ---------------------------------------------------------------------------------------
struct HitScan {
    bool need;       // this lane's computer player is choosing a power-hit direction
    bool ascending;  // y_direction scanned -1,0,1 (draw == 0) or 1,0,-1
};

template <bool IS_P2>
__device__ __forceinline__ HitScan computer_decide_begin(Player& p, const Ball& b, Input& in, const RngId& id,
                                                         uint32_t& rng)
{
    constexpr int kLeft = IS_P2 ? kGroundHalfWidth : 0;                      // left boundary of own side
    constexpr int kRight = kLeft + kGroundHalfWidth;                         // right boundary
    constexpr int kOppHigh = (IS_P2 ? kGroundWidth : 0) + kGroundHalfWidth;  // :718

    HitScan hs{false, false};
    in.xd = 0;
    in.yd = 0;
    in.hit = 0;

    const int dxb = abs(b.x - p.x);
    int target = b.ex;
    if (dxb > 100 && abs(b.xv) < p.bold + 5) {
        if ((b.ex <= kLeft || b.ex >= kOppHigh) && p.standby == 0) target = kLeft + kGroundHalfWidth / 2;
    }

    if (abs(target - p.x) > p.bold + 8) {
        in.xd = (p.x < target) ? 1 : -1;
    } else if (rng_integers(id, rng, 20u) == 0) {  // :728
        p.standby = rng_integers(id, rng, 2u);     // :729
    }

    if (p.state == 0) {
        if (abs(b.xv) < p.bold + 3 && dxb < kPlayerHalfLength && b.y > -36 && b.y < 10 * p.bold + 84 && b.yv > 0)
            in.yd = -1;
        if (b.ex > kLeft && b.ex < kRight && dxb > p.bold * 5 + kPlayerLength && b.x > kLeft && b.x < kRight &&
            b.y > 174) {
            in.hit = 1;  // dive
            in.xd = (p.x < b.x) ? 1 : -1;
        }
    } else if (p.state == 1 || p.state == 2) {
        if (dxb > 8) in.xd = (p.x < b.x) ? 1 : -1;
        if (dxb < 48 && abs(b.y - p.y) < 48) {
            hs.need = true;
            hs.ascending = rng_integers(id, rng, 2u) == 0;  // :795
        }
    }
    return hs;
}

// candidate c = 0..5 in canonical order: x_direction 1 for c<3 else 0; y_direction (c mod 3) - 1
__device__ __forceinline__ int candidate_xdir(int c) { return c < 3 ? 1 : 0; }
__device__ __forceinline__ int candidate_ydir(int c) { return (c < 3 ? c : c - 3) - 1; }

// LDS hand-over between the lanes of ONE wave.  LONE_WAVE: the workgroup is this wave, so the
// workgroup barrier is the cheapest correct statement of it.  Otherwise (the scout wave of
// step_kernel<..., SCOUT>) the other wave must not be involved: a wave's LDS instructions execute
// in issue order, so only the compiler has to be kept from reordering them.
template <bool LONE_WAVE>
__device__ __forceinline__ void wave_lds_handover()
{
    if (LONE_WAVE) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The lanes whose computer player will scan the power-hit directions this frame (:757-758 via
// :796-816): a pure function of the player before its move and of the ball after the world step.
__device__ __forceinline__ bool power_hit_scan_needed(const Player& p, const Ball& b)
{
    return (p.state == 1 || p.state == 2) && abs(b.x - p.x) < 48 && abs(b.y - p.y) < 48;
}

// Wave-cooperative evaluation of expected_landing_point_x_when_power_hit (:820-884) for the six
// candidates of every lane with `need`.  Must be called by all 64 lanes of a wave in uniform
// control flow.  scratch: >= 576 words of LDS owned by this wave.
template <bool LONE_WAVE = true>
__device__ __forceinline__ void wave_power_hit_candidates(bool need, const Ball& b, int (&ex)[6],
                                                          int32_t* __restrict__ scratch, int lane)
{
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return;  // wave-uniform
    const int deciders = __popcll(mask);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (need) {
        scratch[rank] = b.x;
        scratch[64 + rank] = b.y;
        scratch[128 + rank] = abs(b.yv);
    }
    wave_lds_handover<LONE_WAVE>();
    const int items = deciders * 6;
    for (int first = 0; first < items; first += kWaveGames) {
        const int item = first + lane;
        if (item < items) {
            const int r = item / 6, c = item - 6 * r;
            const int sx = scratch[r], sy = scratch[64 + r], sayv = scratch[128 + r];
            const int xdir = candidate_xdir(c), ydir = candidate_ydir(c);
            const int sxv = (sx < kGroundHalfWidth) ? (xdir + 1) * 10 : -(xdir + 1) * 10;  // :841-844
            const int syv = sayv * ydir * 2;                                              // :845
            scratch[192 + item] = predict_landing_x<false>(sx, sy, sxv, syv);
        }
    }
    wave_lds_handover<LONE_WAVE>();
    if (need) {
#pragma unroll
        for (int c = 0; c < 6; ++c) ex[c] = scratch[192 + rank * 6 + c];
    }
    wave_lds_handover<LONE_WAVE>();  // scratch is reused (observation staging)
}

template <bool IS_P2>
__device__ __forceinline__ void computer_decide_finish(const HitScan& hs, const int (&ex)[6], const Player& p,
                                                       const Player& other, Input& in)
{
    constexpr int kLeft = IS_P2 ? kGroundHalfWidth : 0;
    constexpr int kOppHigh = (IS_P2 ? kGroundWidth : 0) + kGroundHalfWidth;  // :801
    if (!hs.need) return;
    bool found = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        // scan order: x_direction 1 then 0; y_direction ascending (-1,0,1) or descending (1,0,-1)
        const int c = hs.ascending ? k : (k < 3 ? 2 - k : 8 - k);
        const int e = ex[c];
        if (!found && (e <= kLeft || e >= kOppHigh) && abs(e - other.x) > kPlayerLength) {
            in.xd = candidate_xdir(c);
            in.yd = candidate_ydir(c);
            found = true;
        }
    }
    if (found) {
        in.hit = 1;
        if (abs(other.x - p.x) < 80 && in.yd != -1) in.yd = -1;
    }
}

// The same decision for the pair kernel, in one straight-line piece: let_computer_decide_user_input
// (physics.py:689-771) incl. decide_whether_input_power_hit (:774-817), with
//   * the env stream's next three draws handed in (`pre`: Philox word 0 at counters rng, rng+1, rng+2, which the
//     caller computes unconditionally while its table gathers are in flight): the decision makes at most three
//     draws -- (0,20) :728, (0,2) :729, (0,2) :795 -- always from consecutive counters, so every draw becomes a
//     select instead of a divergent 10-round Philox on the wave's critical path;
//   * the six candidate landing points handed in (`ex`, meaningful where `scan`: gathered from the power-hit
//     table as soon as the ball has moved -- `scan` = power_hit_scan_needed is a pure function of the player before
//     its move and of the ball).
// Returns the number of draws made (the caller advances the stream by it).
struct PreDrawn {
    uint32_t w0, w1, w2;
};

__device__ __forceinline__ PreDrawn predraw3(const RngId& id, uint32_t counter)
{
    PreDrawn pre;
    uint32_t unused;
    philox4x32_10(id.id_lo, id.id_hi, counter, 0u, id.ks, pre.w0, unused);
    philox4x32_10(id.id_lo, id.id_hi, counter + 1u, 0u, id.ks, pre.w1, unused);
    philox4x32_10(id.id_lo, id.id_hi, counter + 2u, 0u, id.ks, pre.w2, unused);
    return pre;
}

template <bool IS_P2>
__device__ __forceinline__ uint32_t computer_decide_predrawn(Player& p, const Ball& b, int other_x, const PreDrawn& pre,
                                                            bool scan, const int (&ex)[6], Input& in)
{
    constexpr int kLeft = IS_P2 ? kGroundHalfWidth : 0;                      // left boundary of own side
    constexpr int kRight = kLeft + kGroundHalfWidth;                         // right boundary
    constexpr int kOppHigh = (IS_P2 ? kGroundWidth : 0) + kGroundHalfWidth;  // :718,:801

    const int dxb = abs(b.x - p.x);
    const bool far_slow = (dxb > 100) & (abs(b.xv) < p.bold + 5);
    const bool ex_outside = (b.ex <= kLeft) | (b.ex >= kOppHigh);
    const int target = (far_slow & ex_outside & (p.standby == 0)) ? kLeft + kGroundHalfWidth / 2 : b.ex;  // :713-722

    const bool moving = abs(target - p.x) > p.bold + 8;
    int xd = moving ? ((p.x < target) ? 1 : -1) : 0;  // :724-728
    int yd = 0, hit = 0;
    // :728-729 -- integers(0, 20) == 0, then integers(0, 2); (u32 * n) >> 32 as in rng_integers
    const bool redraw = !moving & (__umulhi(pre.w0, 20u) == 0u);
    p.standby = redraw ? (int)__umulhi(pre.w1, 2u) : p.standby;
    uint32_t draws = moving ? 0u : (redraw ? 2u : 1u);

    const bool toward = p.x < b.x;
    if (p.state == 0) {
        const bool jump = (abs(b.xv) < p.bold + 3) & (dxb < kPlayerHalfLength) & (b.y > -36) & (b.y < 10 * p.bold + 84) &
                          (b.yv > 0);
        yd = jump ? -1 : 0;
        const bool dive = (b.ex > kLeft) & (b.ex < kRight) & (dxb > p.bold * 5 + kPlayerLength) & (b.x > kLeft) &
                          (b.x < kRight) & (b.y > 174);
        hit = dive ? 1 : 0;
        xd = dive ? (toward ? 1 : -1) : xd;
    } else if (p.state == 1 || p.state == 2) {
        xd = (dxb > 8) ? (toward ? 1 : -1) : xd;
        if (scan) {  // == dxb < 48 && |b.y - p.y| < 48 in these states (:757)
            const uint32_t w = draws == 0u ? pre.w0 : (draws == 1u ? pre.w1 : pre.w2);
            const bool ascending = __umulhi(w, 2u) == 0u;  // :795
            draws += 1u;
            bool found = false;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                // scan order: x_direction 1 then 0; y_direction ascending (-1,0,1) or descending (1,0,-1)
                const int c = ascending ? k : (k < 3 ? 2 - k : 8 - k);
                const int e = ascending ? ex[k] : ex[k < 3 ? 2 - k : 8 - k];
                const bool good = !found & ((e <= kLeft) | (e >= kOppHigh)) & (abs(e - other_x) > kPlayerLength);
                xd = good ? candidate_xdir(c) : xd;
                yd = good ? candidate_ydir(c) : yd;
                found |= good;
            }
            hit = found ? 1 : hit;
            yd = (found & (abs(other_x - p.x) < 80)) ? -1 : yd;  // :768-771
        }
    }
    in.xd = xd;
    in.yd = yd;
    in.hit = hit;
    return draws;
}

// ---------------------------------------------------------------------------------------
// Player movement (physics.py:457-552; the game-end tail :554-564 is unreachable under the
// env because termination is immediate, pikazoo_env.py:230-233).
// ---------------------------------------------------------------------------------------
template <bool IS_P2>
__device__ __forceinline__ void player_move(Player& p, const Input& in)
{
    constexpr int kMinX = IS_P2 ? kGroundHalfWidth + kPlayerHalfLength : kPlayerHalfLength;
    constexpr int kMaxX = IS_P2 ? kGroundWidth - kPlayerHalfLength : kGroundHalfWidth - kPlayerHalfLength;

    // lying down after a dive: only the countdown runs (:458-462); computed for every lane and
    // selected at the end (predication instead of an early return)
    const bool lying = p.state == 4;
    const int lie_count = p.lying - 1;
    const int lie_state = (lie_count < -1) ? 0 : 4;

    // x movement and clamp to the own half (:465-488)
    const int vx = (p.state < 3) ? in.xd * 6 : p.dive * 8;
    const int x = min(max(p.x + vx, kMinX), kMaxX);

    // jump (:491-499)
    const bool jump = (p.state < 3) & (in.yd == -1) & (p.y == kPlayerGroundY);
    int yv = jump ? -16 : p.yv;
    int st = jump ? 1 : p.state;
    int fr = jump ? 0 : p.frame;

    // gravity / landing (:502-517)
    const int fy = p.y + yv;
    const bool air = fy < kPlayerGroundY, land = fy > kPlayerGroundY;
    yv = air ? yv + 1 : (land ? 0 : yv);
    const int y = land ? kPlayerGroundY : fy;
    fr = land ? 0 : fr;
    const bool dive_landed = land & (st == 3);
    const int lying_left = dive_landed ? 3 : p.lying;
    st = land ? (dive_landed ? 4 : 0) : st;

    // power hit / dive trigger (:519-533)
    const bool hit = in.hit == 1;
    const bool power = hit & (st == 1);
    const bool dive = hit & (st == 0) & (in.xd != 0);
    int delay = power ? 5 : p.delay;
    fr = (power | dive) ? 0 : fr;
    st = power ? 2 : (dive ? 3 : st);
    const int dive_dir = dive ? in.xd : p.dive;
    yv = dive ? -5 : yv;

    // animation frame machine (:535-552), one of three by state
    int arm = p.arm;
    {
        // state 1: frame cycles 0,1,2
        const int f1 = (fr + 1 >= 3) ? fr + 1 - 3 : fr + 1;  // frame is 0..2 here ((frame + 1) % 3)
        // state 2: hold `delay` frames, then run frames 1..4 and fall back to state 1
        const bool s2_advance = delay < 1;
        const int f2n = fr + 1;
        const bool s2_done = s2_advance & (f2n > 4);
        const int f2 = s2_advance ? (s2_done ? 0 : f2n) : fr;
        const int d2 = s2_advance ? delay : delay - 1;
        // state 0: every 4th frame swing the arm 0..4..0
        const int d0n = delay + 1;
        const bool s0_tick = d0n > 3;
        const int probe = fr + arm;
        const int arm0 = (s0_tick & ((unsigned)probe > 4u)) ? -arm : arm;
        const int f0 = s0_tick ? fr + arm0 : fr;
        const int d0 = s0_tick ? 0 : d0n;

        const bool is0 = st == 0, is1 = st == 1, is2 = st == 2;
        fr = is1 ? f1 : (is2 ? f2 : (is0 ? f0 : fr));
        delay = is2 ? d2 : (is0 ? d0 : delay);
        arm = is0 ? arm0 : arm;
        st = (is2 & s2_done) ? 1 : st;
    }

    p.x = lying ? p.x : x;
    p.y = lying ? p.y : y;
    p.yv = lying ? p.yv : yv;
    p.state = lying ? lie_state : st;
    p.frame = lying ? p.frame : fr;
    p.arm = lying ? p.arm : arm;
    p.delay = lying ? p.delay : delay;
    p.dive = lying ? p.dive : dive_dir;
    p.lying = lying ? lie_count : lying_left;
}

// is_collision_between_ball_and_player_happened (physics.py:340-356)
__device__ __forceinline__ bool ball_touches_player(const Ball& b, const Player& p)
{
    return abs(b.x - p.x) <= kPlayerHalfLength && abs(b.y - p.y) <= kPlayerHalfLength;
}

// process_collision_between_ball_and_player (physics.py:580-640), predicated on `hit` (this
// player touches the ball and did not on the previous frame, physics.py:325-335).  Only the
// tie-break draw (:613) is a real branch: it is rare and ticks the RNG.
__device__ __forceinline__ void ball_player_collision(Ball& b, bool hit, int player_x, const Input& in,
                                                      int player_state, const RngId& id, uint32_t& rng)
{
    const int d = b.x - player_x;
    const int third = (int)((unsigned)abs(d) / 3u);
    int xv = (d < 0) ? -third : ((d > 0) ? third : b.xv);
    if (hit & (xv == 0)) xv = rng_integers(id, rng, 3u) - 1;  // :613

    const int ayv = abs(b.yv);
    int yv = ayv < 15 ? -15 : -ayv;

    const bool power = player_state == 2;  // jumping and power hitting
    const int pxv = (abs(in.xd) + 1) * 10;
    xv = power ? ((b.x < kGroundHalfWidth) ? pxv : -pxv) : xv;
    yv = power ? abs(yv) * in.yd * 2 : yv;

    b.xv = hit ? xv : b.xv;
    b.yv = hit ? yv : b.yv;
    b.punch = (hit & power) ? b.x : b.punch;
    b.power = hit ? (int)power : b.power;
}

// ---------------------------------------------------------------------------------------
// One frame of the wave's 64 games: raw_env.step (pikazoo_env.py:175-240) around
// physics_engine (physics.py:280-337).  Called by ALL lanes of the wave in uniform control
// flow (the power-hit candidates are evaluated cooperatively); `live` is false for lanes past
// the end of the batch.  Returns player_1's reward (+1/-1/0); player_2's is its negation.
// `frozen` (auto_reset off and the game already over) leaves the game untouched.
// ---------------------------------------------------------------------------------------
// SCOUT: a second wave of the workgroup takes the flight predictions that can run beside this one
// (pz_kernels.hip: scout_candidates, scout_candidates_posted, scout_landing_after_hits):
//   * the power-hit candidates, evaluated while this wave predicts the landing point (kScoutLoads: and
//     while it loads and starts rounds -- the scout fetches its inputs itself; kScoutPosted, the
//     k-frame modes: this wave posts ball and `need` in link.posts behind a barrier after the world
//     step); picked up from link.cand behind one workgroup barrier;
//   * kScoutLoads only: the landing point after a ball-player collision (:331-332), which nothing in the
//     frame reads any more: the collided ball is posted in link.hits behind a barrier at the end of the
//     frame, the scout predicts and stores the state column itself, and `ex_pending` tells the caller
//     not to.
constexpr int kCandPitch = 7;  // six landing points per game, odd pitch
constexpr int kHitPitch = 5;   // flag, x, y, x velocity, y velocity
constexpr int kPostPitch = 5;  // need, x, y, |y velocity| (+1: odd pitch)
enum ScoutMode { kNoScout = 0, kScoutLoads = 1, kScoutPosted = 2 };
struct ScoutLink {
    const int32_t* cand;  // LDS [64][kCandPitch]
    int32_t* hits;        // LDS [64][kHitPitch]   (kScoutLoads)
    int32_t* posts;       // LDS [64][kPostPitch]  (kScoutPosted)
};

// ---- a ball keeps its landing point along a free flight --------------------------------------------------------------
// calculate_expected_landing_point_x_for (physics.py:643-686) iterates a copy of the ball with the statements of
// process_collision_between_ball_and_world_and_set_ball_position (:359-431) until it would touch the ground.  So for a
// ball B that the world step moves to W(B) without touching the ground, P(W(B)) = P(B) -- the predictor's first iteration
// on B IS that world step -- with two exceptions: (1) over the net at y == 192 exactly the world step bounces the ball off
// the net's top (`y <= NET_PILLAR_TOP_BOTTOM_Y_COORD`, :408) where the predictor pushes it sideways (`<`, :667); (2) P(B)
// counts its iterations against INFINITE_LOOP_LIMIT (:33), which only a ball without x velocity bouncing on the net top
// for ever reaches.  tests/flight_rule.c checks the statement on every one of the landing table's 4.6e8 balls against the
// predictor's CPU restatement (416 234 714 balls covered, no violation; 2 387 182 excluded, 155 343 of them rightly).
// The k-frame pair kernel, which holds the ball and its last prediction in registers, looks a landing point up only for
// the games whose flight was interrupted: a collision, a ball outside the table's domain, the launch's first frame --
// a round that starts serves a ball without x velocity at x = 56 / 376, which comes down where it is.
// (Carrying the same note ACROSS launches -- one byte per game beside the state, every single-frame launch skipping the
// prediction of a free-flying ball -- was built in full in round 6, bit-identical in every table mode, and measured:
// config 3 8.71 -> 8.81 us per launch with both tables, 13.23 -> 12.86 with the power-hit table alone, 14.49 -> 14.30
// with none: at 65 536 games every wave is in the same phase at once and a launch lasts as long as its slowest wave,
// which still has a lane whose flight a player just interrupted.  profiles/r06_experiments/landing_fresh.patch.)
__device__ __forceinline__ bool ball_in_landing_domain(int x, int y, int xv, int yv)
{
    return (ft_xv_index(xv) >= 0) & ((unsigned)(x - kBallRadius) < (unsigned)kFtXCount) & ((unsigned)y < (unsigned)kFtYCount) &
           (abs(yv) <= PZ_FT_YV_MAX);
}
// B = (x, y, xv) BEFORE the world step
__device__ __forceinline__ bool flight_keeps_landing_point(int x, int y, int xv)
{
    const bool over_net = abs(x - kGroundHalfWidth) < kNetPillarHalfWidth;
    return !(over_net & ((y == kNetTopBottomY) | (xv == 0)));
}

// `known`, part 1 (B = the ball BEFORE the world step): the stored landing point is B's and the step will not change it
__device__ __forceinline__ bool landing_known_before_step(bool round_began, bool fresh, const Ball& b)
{
    return round_began | (fresh & ball_in_landing_domain(b.x, b.y, b.xv, b.yv) & flight_keeps_landing_point(b.x, b.y, b.xv));
}
// part 2 (the ball AFTER the world step): it did not touch the ground and is still inside the domain the rule was checked on
__device__ __forceinline__ bool landing_known_after_step(bool known, bool round_began, bool ground, const Ball& b)
{
    return known & !ground & (round_began | ball_in_landing_domain(b.x, b.y, b.xv, b.yv));
}

// Deferred boldness draws (k-frame launches): computer_boldness is drawn for every player at every round start
// (physics.py:218) but read by the computer player's decision only.  For a HUMAN player a round start therefore just
// records the draw's counter; the caller makes the launch's last recorded draw once, behind its frame loop, instead of
// a Philox block on three frames out of four (a wave of 64 games almost always has a round starting somewhere).
struct BoldDefer {
    bool pending1, pending2;
    uint32_t counter1, counter2;
};

// The frame comes in two halves, so that a k-frame launch can run the head of frame t+1 BEFORE it stores frame t's
// observation rows (pz_kernels.hip, step_kernel): the head ends by issuing the computer player's table gathers, and
// gfx9 counts loads and stores in one in-order vmcnt -- gathers issued behind a frame's 18 row stores are not back
// before those have drained, gathers issued in front of them are.
//   frame_head  [auto-reset / new round (its draws)] -> ball-world step -> [scout post] -> gathers + pre-drawn decision
//               words issued;
//   frame_tail  action decode -> decisions -> player moves -> ball-player collisions -> scoring.
// Nothing between the two may touch the game: the caller stages frame t's outputs before head(t+1).
// Only what cannot be re-derived crosses from head to tail (in a k-frame launch: around the loop's back edge) -- the
// ball-world step's result and the registers the gathers land in, untouched; everything else the tail recomputes from
// the game, which nothing moves in between.  (Issuing the gathers earlier still -- the next ball computed on a copy
// before the outputs are staged, the round start committed afterwards -- was built and lost: twelve more live
// registers and the duplicated restart logic cost more than the extra latency hiding bought,
// profiles/r03_experiments/ab_rollout_p2_computer_head_orderings.log.)
struct FrameHead {
    bool frozen, ground;
    uint32_t landing_word;    // LandingProbe::value
    lut_u32x4 candidate_row;  // CandidateProbe::value
    PreDrawn pre;
};

// DEFER1 / DEFER2: player 1's / 2's boldness draw is recorded in *bold instead of made (human players, k-frame launches)
template <bool AI1, bool AI2, int SCOUT = kNoScout, bool DEFER1 = false, bool DEFER2 = false>
__device__ __forceinline__ FrameHead frame_head(Game& g, const pz_config& cfg, const RngId& id, bool live, int lane,
                                                const FlightLut& lut, const ScoutLink link, BoldDefer* bold)
{
    static_assert(!(DEFER1 && AI1) && !(DEFER2 && AI2), "a computer player reads its boldness");
    // The reference empties `agents` on termination (:237-238) and expects reset() before the
    // next step; auto_reset applies reset() (:149-164) in place.  Both that and the new-round
    // branch (:176-180) end in the same per-round initialisation, kept at one call site so the
    // wave runs the (divergent, Philox-drawing) body once.
    FrameHead h{};
    h.frozen = live && g.e.game_ended && !cfg.auto_reset;
    const bool active = live && !h.frozen;
    PZ_FRAME_STAMP(0);
    if (active) {
        if (g.e.round_ended) {  // game_ended implies round_ended
            if (g.e.game_ended) {
                g.e.game_ended = 0;
                g.e.p2serve = 0;
                g.e.s1 = 0;
                g.e.s2 = 0;
            }
            g.e.round_ended = 0;
            // draw order of the reference: player 1's boldness, player 2's [, the serve]
            if (DEFER1) {
                player_new_round_undrawn(g.p1, 36);
                bold->pending1 = true;
                bold->counter1 = g.e.rng;
                g.e.rng += 1u;
            } else {
                player_new_round(g.p1, 36, id, g.e.rng);
            }
            if (DEFER2) {
                player_new_round_undrawn(g.p2, kGroundWidth - 36);
                bold->pending2 = true;
                bold->counter2 = g.e.rng;
                g.e.rng += 1u;
            } else {
                player_new_round(g.p2, kGroundWidth - 36, id, g.e.rng);
            }
            ball_new_round(g.b, get_server(cfg, g.e, id));
        }
        PZ_FRAME_STAMP(1);
        // physics_engine
        h.ground = ball_world_step(g.b);
        PZ_FRAME_STAMP(3);
    }
    if (SCOUT == kScoutPosted) {
        const bool need = active && ((AI1 && power_hit_scan_needed(g.p1, g.b)) || (AI2 && power_hit_scan_needed(g.p2, g.b)));
        int32_t* post = link.posts + lane * kPostPitch;
        post[0] = need;
        if (need) {
            post[1] = g.b.x;
            post[2] = g.b.y;
            post[3] = abs(g.b.yv);
        }
        __syncthreads();  // the scout starts on this frame's candidates
    }
    // With the power-hit table the six candidates are one gather and nothing needs the wave's cooperation: the decision
    // runs as in the pair kernel (step_games_pair) -- both gathers issued (without the landing table that one reads
    // nothing and the lanes that need a landing point predict it), the first decision's three possible draws computed
    // under them, branch-free decisions.  One candidate gather serves both players (same ball).
    const bool by_tables = (AI1 || AI2) && SCOUT == kNoScout && lut.has_power_hit;  // wave-uniform
    if (by_tables && active) {
        const bool scan = (AI1 && power_hit_scan_needed(g.p1, g.b)) || (AI2 && power_hit_scan_needed(g.p2, g.b));
        h.landing_word = lut.landing_issue(true, g.b.x, g.b.y, g.b.xv, g.b.yv).value;
        h.candidate_row = lut.candidates_issue(scan, g.b.x, g.b.y, abs(g.b.yv)).value;
        h.pre = predraw3(id, g.e.rng);
    }
    return h;
}

// PIN: the caller runs head and tail back to back (single frame): keep the head's Philox blocks under its gathers
template <bool AI1, bool AI2, int SCOUT = kNoScout, bool PIN = true>
__device__ __forceinline__ int frame_tail(Game& g, const pz_config& cfg, const RngId& id, int a1, int a2, bool live,
                                          FrameHead& h, int32_t* __restrict__ scratch, int lane, const FlightLut& lut,
                                          const ScoutLink link, bool* ex_pending, const bool last_frame)
{
    const bool active = live && !h.frozen, ground = h.ground;
    Input in1{0, 0, 0}, in2{0, 0, 0};
    if (active) {
        // :182-184 -- every player's key state is sampled, computer-controlled or not
        if (cfg.simplify_action) {
            in1 = decode_action(kSimpleTablesP1, a1, g.p1.hitprev);
            in2 = decode_action(kSimpleTablesP2, a2, g.p2.hitprev);
        } else {
            in1 = decode_action(kFullTables, a1, g.p1.hitprev);
            in2 = decode_action(kFullTables, a2, g.p2.hitprev);
        }
        PZ_FRAME_STAMP(2);
    }
    const bool by_tables = (AI1 || AI2) && SCOUT == kNoScout && lut.has_power_hit;  // wave-uniform
    if (by_tables) {
        if (active) {
            int ex[6] = {0, 0, 0, 0, 0, 0};
            if (PIN)
                asm volatile("" : "+v"(h.pre.w0), "+v"(h.pre.w1), "+v"(h.pre.w2), "+v"(h.landing_word), "+v"(h.candidate_row.x),
                             "+v"(h.candidate_row.y), "+v"(h.candidate_row.z));  // (see step_games_pair)
            // what the head's gathers were issued for, re-derived (nobody has moved since): player 2 has not moved yet
            const bool scan1 = AI1 && power_hit_scan_needed(g.p1, g.b);
            const bool scan2 = AI2 && power_hit_scan_needed(g.p2, g.b);
            const int ayv = abs(g.b.yv);
            uint32_t unused;
            LandingProbe lp = lut.landing_locate(true, g.b.x, g.b.y, g.b.xv, g.b.yv, unused);
            lp.value = h.landing_word;
            CandidateProbe cp = lut.candidates_locate(scan1 | scan2, g.b.x, g.b.y, ayv, unused);
            cp.value = h.candidate_row;
            g.b.ex = lut.landing_finish(lp, g.b.x, g.b.y, g.b.xv, g.b.yv, 0);  // :314-315, one evaluation serves both
            lut.candidates_finish(cp, g.b.x, g.b.y, ayv, ex);
            PreDrawn pre = h.pre;
            if (AI1) g.e.rng += computer_decide_predrawn<false>(g.p1, g.b, g.p2.x, pre, scan1, ex, in1);
            player_move<false>(g.p1, in1);
            if (AI2) {
                if (AI1) pre = predraw3(id, g.e.rng);  // player 2's draws continue where player 1's ended
                g.e.rng += computer_decide_predrawn<true>(g.p2, g.b, g.p1.x, pre, scan2, ex, in2);
            }
        }
    } else {
        if ((AI1 || AI2) && active) {
            // :314-315 recomputes the landing point before each player; the ball does not move
            // between the two calls, so one evaluation serves both.
            g.b.ex = lut.landing_x(g.b.x, g.b.y, g.b.xv, g.b.yv);
        }

        if (AI1) {
            HitScan hs{false, false};
            int ex[6] = {0, 0, 0, 0, 0, 0};
            if (active) hs = computer_decide_begin<false>(g.p1, g.b, in1, id, g.e.rng);
            if (SCOUT) {
                __syncthreads();  // the scout's candidates are in place
                if (hs.need) {
    #pragma unroll
                    for (int c = 0; c < 6; ++c) ex[c] = link.cand[lane * kCandPitch + c];
                }
            } else if (lut.has_power_hit) {  // wave-uniform
                if (hs.need) lut.power_hit_candidates(g.b.x, g.b.y, abs(g.b.yv), ex);
            } else {
                wave_power_hit_candidates(hs.need, g.b, ex, scratch, lane);
            }
            computer_decide_finish<false>(hs, ex, g.p1, g.p2, in1);
        }
        if (active) player_move<false>(g.p1, in1);
        PZ_FRAME_STAMP(4);
        if (AI2) {
            HitScan hs{false, false};
            int ex[6] = {0, 0, 0, 0, 0, 0};
            if (active) hs = computer_decide_begin<true>(g.p2, g.b, in2, id, g.e.rng);
            if (SCOUT) {
                if (!AI1) __syncthreads();  // (with two computer players the barrier above already passed)
                if (hs.need) {
    #pragma unroll
                    for (int c = 0; c < 6; ++c) ex[c] = link.cand[lane * kCandPitch + c];
                }
            } else if (lut.has_power_hit) {
                if (hs.need) lut.power_hit_candidates(g.b.x, g.b.y, abs(g.b.yv), ex);
            } else {
                wave_power_hit_candidates(hs.need, g.b, ex, scratch, lane);
            }
            computer_decide_finish<true>(hs, ex, g.p2, g.p1, in2);
        }
    }

    int reward = 0;
    bool hit_for_scout = false;
    if (active) {
        player_move<true>(g.p2, in2);
        PZ_FRAME_STAMP(5);

        // physics.py:319-335: player 1 first, then player 2 against the possibly changed velocities
        const bool touch1 = ball_touches_player(g.b, g.p1), hit1 = touch1 & (g.p1.coll == 0);
        ball_player_collision(g.b, hit1, g.p1.x, in1, g.p1.state, id, g.e.rng);
        g.p1.coll = touch1;
        const bool touch2 = ball_touches_player(g.b, g.p2), hit2 = touch2 & (g.p2.coll == 0);
        ball_player_collision(g.b, hit2, g.p2.x, in2, g.p2.state, id, g.e.rng);
        g.p2.coll = touch2;
        const bool hit_processed = hit1 | hit2;

        PZ_FRAME_STAMP(6);
        // scoring / round end / game end (:190-210); round_ended and game_ended are both 0 here
        const bool p2_scores = ground & (g.b.punch < kGroundHalfWidth), p1_scores = ground & !p2_scores;
        g.e.s1 += p1_scores;
        g.e.s2 += p2_scores;
        g.e.p2serve = ground ? (int)p2_scores : g.e.p2serve;
        g.e.game_ended = ground & ((p2_scores ? g.e.s2 : g.e.s1) >= cfg.winning_score);
        g.e.round_ended = ground;
        reward = ground ? (p2_scores ? -1 : 1) : 0;

        // :331-332 -- the landing point is predicted again after a processed collision (when both
        // players hit in one frame the second evaluation overwrites the first, so one after both
        // collisions leaves the same value).  Nothing in this frame reads it any more, and the next
        // frame of an active game starts by predicting it afresh (:314-315): between the frames of one
        // launch (`last_frame` false) it only has to be evaluated for a game that freezes here.
        const bool ex_observable = last_frame | (g.e.game_ended != 0 && cfg.auto_reset == 0);
        if ((AI1 || AI2) && hit_processed && ex_observable) {
            if (SCOUT == kScoutLoads)
                hit_for_scout = true;
            else
                g.b.ex = lut.landing_x(g.b.x, g.b.y, g.b.xv, g.b.yv);
        }
    }
    if (SCOUT == kScoutLoads) {
        int32_t* slot = link.hits + lane * kHitPitch;
        slot[0] = hit_for_scout;
        if (hit_for_scout) {
            slot[1] = g.b.x;
            slot[2] = g.b.y;
            slot[3] = g.b.xv;
            slot[4] = g.b.yv;
        }
        __syncthreads();  // collided balls posted
        *ex_pending = hit_for_scout;
    }
    PZ_FRAME_STAMP(7);
    return reward;
}

// head and tail back to back: the single-frame launches
template <bool AI1, bool AI2, int SCOUT = kNoScout>
__device__ __forceinline__ int step_games(Game& g, const pz_config& cfg, const RngId& id, int a1, int a2, bool live,
                                          bool& frozen, int32_t* __restrict__ scratch, int lane,
                                          const FlightLut& lut, const ScoutLink link = ScoutLink{nullptr, nullptr, nullptr},
                                          bool* ex_pending = nullptr, const bool last_frame = true)
{
    FrameHead h = frame_head<AI1, AI2, SCOUT>(g, cfg, id, live, lane, lut, link, nullptr);
    frozen = h.frozen;
    return frame_tail<AI1, AI2, SCOUT, true>(g, cfg, id, a1, a2, live, h, scratch, lane, lut, link, ex_pending, last_frame);
}

// ---------------------------------------------------------------------------------------
// The same frame for a PAIR of waves per 64 games (single frame, every player configuration).
// A lone wave per SIMD only fills every second issue slot, and its ~1 250 instructions run one
// after the other.  Here the work of a workgroup's 64 games is split between two waves:
// wave ROLE owns player ROLE+1 -- its new-round draw, its computer player's decision, its movement,
// its state columns, its agent's reward and observation tensor -- and both redo the cheap shared
// parts (round bookkeeping, action decode, ball-world step, ball-player collisions, scoring),
// which are deterministic, so the two copies stay identical.  One LDS exchange per frame hands
// the moved player to the partner.  `g` holds: the own player complete, of the partner `coll` plus
// what the caller loaded for the own computer player's decision (see pair_body) and whatever the
// exchange fills in, ball and env complete.
//
// Computer players (AI1 / AI2).  The reference decides and moves player 1, then player 2
// (physics.py:304-316); a decision reads the other player's x only (:801,:768 via
// decide_whether_input_power_hit), player 1's sees player 2 before its move, player 2's sees player 1
// after.  So
//   * player 1's wave needs player 2's old x (loaded by the caller);
//   * player 2's wave needs player 1's NEW x: with a human player 1 it recomputes that from the decoded
//     action (x, state and diving direction loaded by the caller: the first lines of player_move); with a
//     computer player 1 it waits for wave 0's early post {new x, draws made} behind one extra barrier;
//   * the env stream's draws stay in the reference's order (p1 decision, p2 decision, p1 collision, p2
//     collision): each wave counts its own decision's draws and posts the count with its player, both
//     then continue from base + d1 + d2;
//   * the decided (x_direction, y_direction) travel with the player too: the ball-player collision takes
//     the power hit's direction from them (physics.py:329 passes the mutated user_input).
// The flight predictions come from `lut` (tables or, outside their domain, the computed form); the wave
// that owns the (last) computer player keeps ball.expected_landing_point_x (kKeepsEx) and hands the look-up of its
// value after a ball-player collision back to the caller (`after_hit`, finished behind the caller's stores).
// xchg: two regions of LDS, `xchg_region` words apart; a wave writes into the PARTNER's region and reads
// from its own, so a wave may reuse its own region afterwards without asking (the pair kernel aliases them
// with the observation staging rows).  One __syncthreads(), two when both players are computers.
// ---------------------------------------------------------------------------------------
constexpr int kXchgPitch = 64;      // exchange word k of lane l lives at [k * 64 + l] (10 words: 9 player + decision)
constexpr int kEarlyPostAt = 1024;  // word offset of the early post {x, draws} inside a region (2 x 64 words)

// before_barrier(): called once the own player has moved and been posted, in front of the exchange barrier -- what the
// caller can do with its finished player while the partner wave is still deciding
template <int ROLE, bool AI1, bool AI2, class BeforeBarrier>
__device__ __forceinline__ int step_games_pair(Game& g, const pz_config& cfg, const RngId& id, int a1, int a2,
                                               bool live, bool& frozen, int32_t* __restrict__ xchg, int xchg_region,
                                               int lane, const FlightLut& lut, LandingProbe& after_hit, bool& bold_pending,
                                               BeforeBarrier&& before_barrier)
{
    constexpr bool kOwnAI = ROLE == 0 ? AI1 : AI2;
    constexpr bool kOtherAI = ROLE == 0 ? AI2 : AI1;
    constexpr bool kKeepsEx = (AI1 || AI2) && (ROLE == 1 ? AI2 : !AI2);
    Player& own = ROLE == 0 ? g.p1 : g.p2;
    Player& other = ROLE == 0 ? g.p2 : g.p1;
    frozen = live && g.e.game_ended && !cfg.auto_reset;
    const bool active = live && !frozen;
    Input in1{0, 0, 0}, in2{0, 0, 0};
    bool ground = false;
    bool bold_late = false;       // the computer's round-start boldness, drawn behind the gathers' issue
    uint32_t bold_counter = 0u;
    PZ_FRAME_STAMP(0);
    // (Moving the ball first and issuing the computer's two gathers in front of the players' round start and the action
    // decode -- the gathers depend on the ball columns, the round flags and the serve only -- was built twice in round 4
    // and lost twice: issued in a block of their own 8.36 -> 8.49 us per launch (the compiler waits for a load where the
    // branch it was issued in closes), issue and consumption in one block 8.33 -> 8.36, packed 7.25 -> 7.71:
    // profiles/r04_experiments/ab_early_gather_*.log.  A hundred instructions earlier buys nothing.)
    if (active) {
        if (g.e.round_ended) {  // reset (:149-164) or new round (:176-180); game_ended implies round_ended
            if (g.e.game_ended) {
                g.e.game_ended = 0;
                g.e.p2serve = 0;
                g.e.s1 = 0;
                g.e.s2 = 0;
            }
            g.e.round_ended = 0;
            // draw order of the reference: player 1 boldness, player 2 boldness [, serve]; each wave
            // evaluates its own player's draw (index rng + ROLE)
            if (!AI1 && !AI2) {
                // human vs human: computer_boldness (drawn for humans too, physics.py:218) is read by nothing, so the
                // wave resets its player here and leaves the draw -- a whole Philox block in front of the frame's
                // first store -- to its caller, behind the stores (7.25 -> 7.15 us per launch; with a computer player
                // in the game the human player's wave is not the one the launch waits for, and deferring cost 1 %)
                const int keep = own.bold;
                own.x = ROLE == 0 ? 36 : kGroundWidth - 36;
                own.y = kPlayerGroundY;
                own.yv = 0;
                own.coll = 0;
                own.state = 0;
                own.frame = 0;
                own.arm = 1;
                own.delay = 0;
                own.bold = keep;
                bold_pending = true;
            } else {
                if (kOwnAI) {
                    // the computer's own boldness is read by its decision only: drawn behind the issue of the frame's
                    // two gathers, whose latency has room for a fourth Philox block (8.62 -> 8.53 us per launch, packed
                    // 7.46 -> 7.32, with the computer's wave at priority 1; without it, round 2: 8.39 -> 8.48)
                    const int keep = own.bold;
                    player_new_round_undrawn(own, ROLE == 0 ? 36 : kGroundWidth - 36);
                    own.bold = keep;
                    bold_late = true;
                    bold_counter = g.e.rng + (uint32_t)ROLE;
                } else {
                    uint32_t own_draw = g.e.rng + (uint32_t)ROLE;
                    player_new_round(own, ROLE == 0 ? 36 : kGroundWidth - 36, id, own_draw);
                }
            }
            other.x = ROLE == 0 ? kGroundWidth - 36 : 36;
            other.y = kPlayerGroundY;
            other.yv = 0;
            other.coll = 0;
            other.state = 0;
            other.frame = 0;
            other.delay = 0;
            g.e.rng += 2u;
            ball_new_round(g.b, get_server(cfg, g.e, id));  // a random serve is drawn by both waves
        }
        PZ_FRAME_STAMP(1);
        int other_prev = 0;  // the partner's key edge is only needed by the partner
        if (cfg.simplify_action) {
            in1 = decode_action(kSimpleTablesP1, a1, ROLE == 0 ? g.p1.hitprev : other_prev);
            in2 = decode_action(kSimpleTablesP2, a2, ROLE == 1 ? g.p2.hitprev : other_prev);
        } else {
            in1 = decode_action(kFullTables, a1, ROLE == 0 ? g.p1.hitprev : other_prev);
            in2 = decode_action(kFullTables, a2, ROLE == 1 ? g.p2.hitprev : other_prev);
        }
        PZ_FRAME_STAMP(2);

        ground = ball_world_step(g.b);
        PZ_FRAME_STAMP(3);
    }
    const uint32_t rng_base = g.e.rng;  // the env stream before this frame's decisions
    uint32_t draws_own = 0, draws_other = 0;
    Input& in_own = ROLE == 0 ? in1 : in2;
    Input& in_other = ROLE == 0 ? in2 : in1;

    if (ROLE == 1 && AI1 && AI2) {
        __syncthreads();  // player 1's early post is in place
        if (active) {
            const int32_t* early = xchg + ROLE * xchg_region + kEarlyPostAt + lane * 2;
            other.x = early[0];
            draws_other = (uint32_t)early[1];
        }
    }
    if (kOwnAI) {
        if (active) {
            // :314-315 recomputes the landing point before each player; the ball does not move in between.
            // Both gathers go out first; the three Philox blocks of the decision's draws run under their latency.
            const bool scan = power_hit_scan_needed(own, g.b);
            const int ayv = abs(g.b.yv);
            int ex[6] = {0, 0, 0, 0, 0, 0};
            LandingProbe lp = lut.landing_issue(true, g.b.x, g.b.y, g.b.xv, g.b.yv);
            CandidateProbe cp = lut.candidates_issue(scan, g.b.x, g.b.y, ayv);
            PreDrawn pre = predraw3(id, rng_base + draws_other);
            if (__builtin_amdgcn_ballot_w64(bold_late) != 0ull) {  // (wave-uniform: a lane of the wave starts a round)
                const int drawn = rng_integers(id, bold_counter, 5u);
                own.bold = bold_late ? drawn : own.bold;
            }
            // Keep the Philox blocks where they are written -- under the two gathers.  Left alone the compiler sinks
            // them below the (rare) out-of-domain branches of the look-ups and waits for the gathers first.  The empty
            // statement reads the draws together with the gathered registers: the draws must be complete before it,
            // and the first wait for the gathers lands immediately in front of it.
            asm volatile("" : "+v"(pre.w0), "+v"(pre.w1), "+v"(pre.w2), "+v"(lp.value), "+v"(cp.value.x), "+v"(cp.value.y),
                         "+v"(cp.value.z));
            if (ROLE == 1 && !AI1) {
                // player 1 (human) has moved when player 2 decides: its new x as player_move computes it
                const int vx = (other.state < 3) ? in1.xd * 6 : other.dive * 8;
                const int nx = min(max(other.x + vx, kPlayerHalfLength), kGroundHalfWidth - kPlayerHalfLength);
                other.x = (other.state == 4) ? other.x : nx;
            }
            g.b.ex = lut.landing_finish(lp, g.b.x, g.b.y, g.b.xv, g.b.yv, 0);
            lut.candidates_finish(cp, g.b.x, g.b.y, ayv, ex);
            draws_own = computer_decide_predrawn<ROLE == 1>(own, g.b, other.x, pre, scan, ex, in_own);
        }
    }
    if (active) player_move<ROLE == 1>(own, in_own);
    PZ_FRAME_STAMP(4);
    if (ROLE == 0 && AI1 && AI2) {
        if (active) {
            int32_t* early = xchg + (1 - ROLE) * xchg_region + kEarlyPostAt + lane * 2;
            early[0] = own.x;
            early[1] = (int32_t)draws_own;
        }
        __syncthreads();
    }
    // hand the own player to the partner wave: everything the collisions and the observations read
    // (frozen games exchange their unchanged players, so their observations stay complete)
    if (live) {
        // word k of lane l at [k * 64 + l]: conflict-free, and pairs of words become one ds_write2st64_b32
        int32_t* mine = xchg + (1 - ROLE) * xchg_region + lane;  // into the partner's region
        mine[0 * kXchgPitch] = own.x;
        mine[1 * kXchgPitch] = own.y;
        mine[2 * kXchgPitch] = own.yv;
        mine[3 * kXchgPitch] = own.state;
        mine[4 * kXchgPitch] = own.frame;
        mine[5 * kXchgPitch] = own.delay;
        mine[6 * kXchgPitch] = own.dive;
        mine[7 * kXchgPitch] = own.lying;
        mine[8 * kXchgPitch] = own.hitprev;
        if (kOwnAI) mine[9 * kXchgPitch] = (in_own.xd + 1) | ((in_own.yd + 1) << 2) | (int32_t)(draws_own << 4);
    }
    before_barrier();
    __syncthreads();
    if (live) {
        const int32_t* theirs = xchg + ROLE * xchg_region + lane;
        other.x = theirs[0 * kXchgPitch];
        other.y = theirs[1 * kXchgPitch];
        other.yv = theirs[2 * kXchgPitch];
        other.state = theirs[3 * kXchgPitch];
        other.frame = theirs[4 * kXchgPitch];
        other.delay = theirs[5 * kXchgPitch];
        other.dive = theirs[6 * kXchgPitch];
        other.lying = theirs[7 * kXchgPitch];
        other.hitprev = theirs[8 * kXchgPitch];
        if (kOtherAI) {
            const int32_t w = theirs[9 * kXchgPitch];
            if (active) {
                in_other.xd = (w & 3) - 1;
                in_other.yd = ((w >> 2) & 3) - 1;
                draws_other = (uint32_t)w >> 4;
            }
        }
    }
    PZ_FRAME_STAMP(5);
    int reward = 0;
    bool hit_processed = false;
    if (active) {
        g.e.rng = rng_base + draws_own + draws_other;

        // physics.py:319-335: player 1 first, then player 2 against the possibly changed velocities
        const bool touch1 = ball_touches_player(g.b, g.p1), hit1 = touch1 & (g.p1.coll == 0);
        ball_player_collision(g.b, hit1, g.p1.x, in1, g.p1.state, id, g.e.rng);
        g.p1.coll = touch1;
        const bool touch2 = ball_touches_player(g.b, g.p2), hit2 = touch2 & (g.p2.coll == 0);
        ball_player_collision(g.b, hit2, g.p2.x, in2, g.p2.state, id, g.e.rng);
        g.p2.coll = touch2;

        // scoring / round end / game end (:190-210)
        const bool p2_scores = ground & (g.b.punch < kGroundHalfWidth), p1_scores = ground & !p2_scores;
        g.e.s1 += p1_scores;
        g.e.s2 += p2_scores;
        g.e.p2serve = ground ? (int)p2_scores : g.e.p2serve;
        g.e.game_ended = ground & ((p2_scores ? g.e.s2 : g.e.s1) >= cfg.winning_score);
        g.e.round_ended = ground;
        reward = ground ? (p2_scores ? -1 : 1) : 0;

        hit_processed = hit1 | hit2;
    }
    PZ_FRAME_STAMP(6);
    // :331-332 -- predicted again after a processed collision (one evaluation after both collisions leaves what
    // the second of two would).  Nothing in the frame reads it any more: the gather is issued here and taken
    // by the caller behind its other stores (`after_hit`).
    if (kKeepsEx) after_hit = lut.landing_issue(hit_processed, g.b.x, g.b.y, g.b.xv, g.b.yv);
    PZ_FRAME_STAMP(7);
    return reward;
}

// ---------------------------------------------------------------------------------------
// The pair frame in two halves, for the k-frame pair kernel (pz_kernels.hip, rollout_pair_kernel): the same split by
// player as step_games_pair, looped -- the game stays in the two waves' registers for k frames -- and cut where
// frame_head / frame_tail cut the single-wave frame, for the same reason: the head of frame t+1 (round start, ball
// step, the own computer player's gathers issued) runs before frame t's row stores, so that the gathers are not
// queued behind them in the in-order vmcnt.  Between the frames every wave holds: its own player complete, the
// partner's nine exchanged words + collision flag (what the last exchange left: the partner's state before its next
// move), ball and env complete.
// The exchange lives in LDS of its own, DOUBLE-BUFFERED by frame parity: a wave posts into the partner's region
// before the frame's barrier and reads its own behind it; with one barrier per frame the partner may still be reading
// frame t's words when this wave posts frame t+1's only if they shared a buffer.
// ---------------------------------------------------------------------------------------
constexpr int kLoopXchgEarlyAt = 10 * kXchgPitch;            // the early post {x, draws} behind the ten exchange words
constexpr int kLoopXchgRegion = kLoopXchgEarlyAt + 2 * 64;   // words per region (one per wave)
constexpr int kLoopXchgWords = 2 * 2 * kLoopXchgRegion;      // two regions, two frame parities

struct PairHead {
    bool frozen, ground;
    bool known;               // own computer player: g.b.ex already is this frame's landing point (no look-up)
    uint32_t rng_base;        // the env stream's counter before this frame's decisions
    uint32_t landing_word;    // own computer player: LandingProbe::value
    lut_u32x4 candidate_row;  //                      CandidateProbe::value
    PreDrawn pre;
};

// DEFER_OWN: the own (human) player's boldness draw is recorded in *bold (counter1 / pending1) instead of made
template <int ROLE, bool AI1, bool AI2, bool DEFER_OWN>
__device__ __forceinline__ PairHead pair_frame_head(Game& g, const pz_config& cfg, const RngId& id, bool live,
                                                    const FlightLut& lut, BoldDefer* bold, const bool ex_fresh = false)
{
    constexpr bool kOwnAI = ROLE == 0 ? AI1 : AI2;
    static_assert(!(DEFER_OWN && kOwnAI), "a computer player reads its boldness");
    Player& own = ROLE == 0 ? g.p1 : g.p2;
    Player& other = ROLE == 0 ? g.p2 : g.p1;
    PairHead h{};
    h.frozen = live && g.e.game_ended && !cfg.auto_reset;
    const bool active = live && !h.frozen;
    bool round_began = false;
    if (active) {
        if (g.e.round_ended) {  // reset (:149-164) or new round (:176-180); game_ended implies round_ended
            round_began = true;
            if (g.e.game_ended) {
                g.e.game_ended = 0;
                g.e.p2serve = 0;
                g.e.s1 = 0;
                g.e.s2 = 0;
            }
            g.e.round_ended = 0;
            // draw order of the reference: player 1 boldness, player 2 boldness [, serve]; each wave handles its own
            // player's draw (index rng + ROLE)
            if (DEFER_OWN) {
                player_new_round_undrawn(own, ROLE == 0 ? 36 : kGroundWidth - 36);
                bold->pending1 = true;
                bold->counter1 = g.e.rng + (uint32_t)ROLE;
            } else {
                uint32_t own_draw = g.e.rng + (uint32_t)ROLE;
                player_new_round(own, ROLE == 0 ? 36 : kGroundWidth - 36, id, own_draw);
            }
            other.x = ROLE == 0 ? kGroundWidth - 36 : 36;
            other.y = kPlayerGroundY;
            other.yv = 0;
            other.coll = 0;
            other.state = 0;
            other.frame = 0;
            other.delay = 0;
            g.e.rng += 2u;
            ball_new_round(g.b, get_server(cfg, g.e, id));  // a random serve is drawn by both waves
        }
        if (kOwnAI) {
            // ex_fresh: g.b.ex is the landing point of the ball as it stands (the last frame predicted it for its moved
            // ball and no collision has changed the flight since)
            h.known = landing_known_before_step(round_began, ex_fresh, g.b);
            g.b.ex = round_began ? g.b.x : g.b.ex;  // (a serve has no x velocity and is not over the net)
        }
        h.ground = ball_world_step(g.b);
        if (kOwnAI) h.known = landing_known_after_step(h.known, round_began, h.ground, g.b);
    }
    h.rng_base = g.e.rng;
    if (kOwnAI && active) {
        const bool scan = power_hit_scan_needed(own, g.b);
        // (a lane that knows its landing point reads entry 0 like a lane outside the domain: a line the whole wave shares)
        h.landing_word = lut.landing_issue(!h.known, g.b.x, g.b.y, g.b.xv, g.b.yv).value;
        h.candidate_row = lut.candidates_issue(scan, g.b.x, g.b.y, abs(g.b.yv)).value;
        // (player 2 behind a computer player 1 learns the counter of its first draw from player 1's early post: tail)
        if (!(ROLE == 1 && AI1)) h.pre = predraw3(id, h.rng_base);
    }
    return h;
}

// xchg: this frame's exchange buffer (kLoopXchgRegion words per wave's region); returns player 1's reward
template <int ROLE, bool AI1, bool AI2>
__device__ __forceinline__ int pair_frame_tail(Game& g, const pz_config& cfg, const RngId& id, int a1, int a2, bool live,
                                               PairHead& h, int32_t* __restrict__ xchg, int lane, const FlightLut& lut,
                                               const bool last_frame, bool* ex_fresh = nullptr)
{
    constexpr bool kOwnAI = ROLE == 0 ? AI1 : AI2;
    constexpr bool kOtherAI = ROLE == 0 ? AI2 : AI1;
    constexpr bool kKeepsEx = (AI1 || AI2) && (ROLE == 1 ? AI2 : !AI2);
    Player& own = ROLE == 0 ? g.p1 : g.p2;
    Player& other = ROLE == 0 ? g.p2 : g.p1;
    const bool active = live && !h.frozen, ground = h.ground;
    Input in1{0, 0, 0}, in2{0, 0, 0};
    if (active) {
        int other_prev = 0;  // the partner's key edge is only needed by the partner
        if (cfg.simplify_action) {
            in1 = decode_action(kSimpleTablesP1, a1, ROLE == 0 ? g.p1.hitprev : other_prev);
            in2 = decode_action(kSimpleTablesP2, a2, ROLE == 1 ? g.p2.hitprev : other_prev);
        } else {
            in1 = decode_action(kFullTables, a1, ROLE == 0 ? g.p1.hitprev : other_prev);
            in2 = decode_action(kFullTables, a2, ROLE == 1 ? g.p2.hitprev : other_prev);
        }
    }
    const uint32_t rng_base = h.rng_base;
    uint32_t draws_own = 0, draws_other = 0;
    Input& in_own = ROLE == 0 ? in1 : in2;
    Input& in_other = ROLE == 0 ? in2 : in1;
    int32_t* mine = xchg + (1 - ROLE) * kLoopXchgRegion;         // what this wave posts: into the partner's region
    const int32_t* theirs = xchg + ROLE * kLoopXchgRegion;       // what the partner posted for this wave

    if (ROLE == 1 && AI1 && AI2) {
        __syncthreads();  // player 1's early post is in place
        if (active) {
            other.x = theirs[kLoopXchgEarlyAt + lane * 2];
            draws_other = (uint32_t)theirs[kLoopXchgEarlyAt + lane * 2 + 1];
        }
    }
    if (kOwnAI && active) {
        const bool scan = power_hit_scan_needed(own, g.b);  // (as in the head: nobody has moved since)
        const int ayv = abs(g.b.yv);
        int ex[6] = {0, 0, 0, 0, 0, 0};
        const PreDrawn pre = (ROLE == 1 && AI1) ? predraw3(id, rng_base + draws_other) : h.pre;
        if (ROLE == 1 && !AI1) {
            // player 1 (human) has moved when player 2 decides: its new x as player_move computes it
            const int vx = (other.state < 3) ? in1.xd * 6 : other.dive * 8;
            const int nx = min(max(other.x + vx, kPlayerHalfLength), kGroundHalfWidth - kPlayerHalfLength);
            other.x = (other.state == 4) ? other.x : nx;
        }
        uint32_t unused;
        LandingProbe lp = lut.landing_locate(!h.known, g.b.x, g.b.y, g.b.xv, g.b.yv, unused);
        lp.value = h.landing_word;
        CandidateProbe cp = lut.candidates_locate(scan, g.b.x, g.b.y, ayv, unused);
        cp.value = h.candidate_row;
        g.b.ex = lut.landing_finish(lp, g.b.x, g.b.y, g.b.xv, g.b.yv, g.b.ex);  // :314-315 (known: it stands, see the head)
        lut.candidates_finish(cp, g.b.x, g.b.y, ayv, ex);
        draws_own = computer_decide_predrawn<ROLE == 1>(own, g.b, other.x, pre, scan, ex, in_own);
    }
    if (active) player_move<ROLE == 1>(own, in_own);
    if (ROLE == 0 && AI1 && AI2) {
        if (active) {
            mine[kLoopXchgEarlyAt + lane * 2] = own.x;
            mine[kLoopXchgEarlyAt + lane * 2 + 1] = (int32_t)draws_own;
        }
        __syncthreads();
    }
    // hand the own player to the partner wave: everything the collisions, the observations and the partner's next
    // decision read (frozen games exchange their unchanged players)
    if (live) {
        int32_t* w = mine + lane;
        w[0 * kXchgPitch] = own.x;
        w[1 * kXchgPitch] = own.y;
        w[2 * kXchgPitch] = own.yv;
        w[3 * kXchgPitch] = own.state;
        w[4 * kXchgPitch] = own.frame;
        w[5 * kXchgPitch] = own.delay;
        w[6 * kXchgPitch] = own.dive;
        w[7 * kXchgPitch] = own.lying;
        w[8 * kXchgPitch] = own.hitprev;
        if (kOwnAI) w[9 * kXchgPitch] = (in_own.xd + 1) | ((in_own.yd + 1) << 2) | (int32_t)(draws_own << 4);
    }
    __syncthreads();
    if (live) {
        const int32_t* r = theirs + lane;
        other.x = r[0 * kXchgPitch];
        other.y = r[1 * kXchgPitch];
        other.yv = r[2 * kXchgPitch];
        other.state = r[3 * kXchgPitch];
        other.frame = r[4 * kXchgPitch];
        other.delay = r[5 * kXchgPitch];
        other.dive = r[6 * kXchgPitch];
        other.lying = r[7 * kXchgPitch];
        other.hitprev = r[8 * kXchgPitch];
        if (kOtherAI) {
            const int32_t w = r[9 * kXchgPitch];
            if (active) {
                in_other.xd = (w & 3) - 1;
                in_other.yd = ((w >> 2) & 3) - 1;
                draws_other = (uint32_t)w >> 4;
            }
        }
    }
    int reward = 0;
    if (active) {
        g.e.rng = rng_base + draws_own + draws_other;
        // physics.py:319-335: player 1 first, then player 2 against the possibly changed velocities
        const bool touch1 = ball_touches_player(g.b, g.p1), hit1 = touch1 & (g.p1.coll == 0);
        ball_player_collision(g.b, hit1, g.p1.x, in1, g.p1.state, id, g.e.rng);
        g.p1.coll = touch1;
        const bool touch2 = ball_touches_player(g.b, g.p2), hit2 = touch2 & (g.p2.coll == 0);
        ball_player_collision(g.b, hit2, g.p2.x, in2, g.p2.state, id, g.e.rng);
        g.p2.coll = touch2;
        // scoring / round end / game end (:190-210)
        const bool p2_scores = ground & (g.b.punch < kGroundHalfWidth), p1_scores = ground & !p2_scores;
        g.e.s1 += p1_scores;
        g.e.s2 += p2_scores;
        g.e.p2serve = ground ? (int)p2_scores : g.e.p2serve;
        g.e.game_ended = ground & ((p2_scores ? g.e.s2 : g.e.s1) >= cfg.winning_score);
        g.e.round_ended = ground;
        reward = ground ? (p2_scores ? -1 : 1) : 0;
        // :331-332 -- between the frames of one launch the value after a processed collision is overwritten by the
        // next frame's prediction before anything reads it: evaluated on the last frame and for a game that freezes
        const bool ex_observable = last_frame | (g.e.game_ended != 0 && cfg.auto_reset == 0);
        if (kKeepsEx && (hit1 | hit2) && ex_observable) g.b.ex = lut.landing_x(g.b.x, g.b.y, g.b.xv, g.b.yv);
        // a processed collision has changed the flight: g.b.ex no longer belongs to the ball (pair_frame_head)
        if (kOwnAI && ex_fresh != nullptr) *ex_fresh = !(hit1 | hit2);
    } else if (kOwnAI && ex_fresh != nullptr) {
        *ex_fresh = false;
    }
    return reward;
}

}  // namespace pz
