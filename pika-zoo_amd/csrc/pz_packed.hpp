// pz_packed.hpp -- the bit-packed state format (cfg->packed_state; PZ_PACKED_* in include/pikazoo_hip.h).
//
// The 44 int32 state words of a game hold small integers (positions < 512, velocities of a few tens, flags).  Packed,
// a game is 36 bytes instead of 176: two 16-byte groups and one 4-byte tail, each a column of its own
// (structure of arrays, lane i at element i):
//   group A  uint32[4]  written by player 1's wave: player 1, the env block, ball.punch_effect_x, ball.previous_previous_y
//   group B  uint32[4]  written by player 2's wave: player 2, the rest of the ball
//   tail     uint32     the three values that are written late and rarely, each by a byte-granular store of the wave
//                       that owns it: ball.expected_landing_point_x (uint16), the two computer_boldness (uint8 each)
// so a step launch reads 32-36 and writes 32-36 bytes of state per game with one 16-byte access per lane and group.
// Field ranges follow from the rules of play (see each field); pz_pack_state validates states that come from outside,
// and the step kernels flag (sticky bit 31 of A1 / B1) a ball y velocity or ball y that would not fit -- never seen in
// play, where |y velocity| stays below 300 -- and likewise a player's y / y velocity, which only a
// planted, unreachable state can drive out of the court.  The ball's y is SIGNED: a ball that falls onto the
// net top faster than its height is bounced to y - y_velocity < 0 (physics.py:406-419: the ceiling is tested before
// the net), so y and its two trail copies are 10-bit signed fields.
#pragma once

#include <stdint.h>

#include "pz_physics.hpp"

namespace pz {

typedef unsigned int pk_u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kPackedOverflowBit = 1u << 31;
constexpr int kPackedBallYvMax = 4095;  // 13-bit signed field
constexpr int kPackedBallYMax = 511;    // 10-bit signed fields (y, previous_y, previous_previous_y)

// word 0 of a group: x 9 | y 8 | y_velocity+32 6 | state 3 | frame_number 3 | delay_before_next_frame 3
__device__ __forceinline__ uint32_t pack_player_core(const Player& p)
{
    return (uint32_t)p.x | ((uint32_t)p.y << 9) | ((uint32_t)(p.yv + 32) << 17) | ((uint32_t)p.state << 23) |
           ((uint32_t)p.frame << 26) | ((uint32_t)p.delay << 29);
}

__device__ __forceinline__ void unpack_player_core(Player& p, uint32_t w)
{
    p.x = (int)(w & 0x1FFu);
    p.y = (int)((w >> 9) & 0xFFu);
    p.yv = (int)((w >> 17) & 0x3Fu) - 32;
    p.state = (int)((w >> 23) & 7u);
    p.frame = (int)((w >> 26) & 7u);
    p.delay = (int)(w >> 29);
}

// bits 0..8 of word 1: arm_swing_direction == 1 | diving_direction+1 2 | lying_down_duration_left+2 3 |
// is_collision_with_ball_happened | computer_where_to_stand_by | power_hit_key_is_down_previous
__device__ __forceinline__ uint32_t pack_player_misc(const Player& p)
{
    return (uint32_t)(p.arm == 1) | ((uint32_t)(p.dive + 1) << 1) | ((uint32_t)(p.lying + 2) << 3) |
           ((uint32_t)(p.coll != 0) << 6) | ((uint32_t)(p.standby != 0) << 7) | ((uint32_t)(p.hitprev != 0) << 8);
}

__device__ __forceinline__ void unpack_player_misc(Player& p, uint32_t w)
{
    p.arm = (w & 1u) ? 1 : -1;
    p.dive = (int)((w >> 1) & 3u) - 1;
    p.lying = (int)((w >> 3) & 7u) - 2;
    p.coll = (int)((w >> 6) & 1u);
    p.standby = (int)((w >> 7) & 1u);
    p.hitprev = (int)((w >> 8) & 1u);
}

// The two player fields that a frame moves without clamping them (a planted, unreachable (y, y_velocity) pair can leave
// the court's y range): checked with every pack, like the ball's y and y velocity.
__device__ __forceinline__ uint32_t player_overflow(const Player& p)
{
    return ((unsigned)p.y < 256u && (unsigned)(p.yv + 32) < 64u) ? 0u : kPackedOverflowBit;
}

// group A: {player 1 core, player 1 misc | punch_effect_x 9 <<9 | is_player2_serve <<18 | round_ended <<19 |
//           game_ended <<20 | previous_previous_y 10 (signed) <<21 | overflow <<31, score 1 16 | score 2 16 <<16,
//           rng draw counter}
__device__ __forceinline__ pk_u32x4 pack_group_a(const Game& g, uint32_t sticky)
{
    pk_u32x4 w;
    w.x = pack_player_core(g.p1);
    w.y = pack_player_misc(g.p1) | ((uint32_t)g.b.punch << 9) | ((uint32_t)(g.e.p2serve != 0) << 18) |
          ((uint32_t)(g.e.round_ended != 0) << 19) | ((uint32_t)(g.e.game_ended != 0) << 20) |
          (((uint32_t)g.b.ppy & 0x3FFu) << 21) | sticky | player_overflow(g.p1);
    w.z = (uint32_t)g.e.s1 | ((uint32_t)g.e.s2 << 16);
    w.w = g.e.rng;
    return w;
}

__device__ __forceinline__ void unpack_group_a(Game& g, const pk_u32x4 w)
{
    unpack_player_core(g.p1, w.x);
    unpack_player_misc(g.p1, w.y);
    g.b.punch = (int)((w.y >> 9) & 0x1FFu);
    g.e.p2serve = (int)((w.y >> 18) & 1u);
    g.e.round_ended = (int)((w.y >> 19) & 1u);
    g.e.game_ended = (int)((w.y >> 20) & 1u);
    g.b.ppy = (int)(w.y << 1) >> 22;
    g.e.s1 = (int)(w.z & 0xFFFFu);
    g.e.s2 = (int)(w.z >> 16);
    g.e.rng = w.w;
}

// group B: {player 2 core, player 2 misc | is_power_hit <<9 | x_velocity+32 6 <<10 | y_velocity 13 (signed) <<16 |
//           overflow <<31, x 9 | previous_x 9 <<9 | previous_previous_x 9 <<18,
//           y 10 (signed) | previous_y 10 (signed) <<10 | fine_rotation 6 <<20}
__device__ __forceinline__ pk_u32x4 pack_group_b(const Game& g, uint32_t sticky)
{
    pk_u32x4 w;
    w.x = pack_player_core(g.p2);
    const bool fits = (unsigned)(g.b.yv + kPackedBallYvMax + 1) <= (unsigned)(2 * kPackedBallYvMax + 1) &&
                      (unsigned)(g.b.y + kPackedBallYMax + 1) <= (unsigned)(2 * kPackedBallYMax + 1);
    const uint32_t over = (fits ? 0u : kPackedOverflowBit) | player_overflow(g.p2);
    w.y = pack_player_misc(g.p2) | ((uint32_t)(g.b.power != 0) << 9) | ((uint32_t)(g.b.xv + 32) << 10) |
          (((uint32_t)g.b.yv & 0x1FFFu) << 16) | sticky | over;
    w.z = (uint32_t)g.b.x | ((uint32_t)g.b.px << 9) | ((uint32_t)g.b.ppx << 18);
    w.w = ((uint32_t)g.b.y & 0x3FFu) | (((uint32_t)g.b.py & 0x3FFu) << 10) | ((uint32_t)g.b.rot << 20);
    return w;
}

__device__ __forceinline__ void unpack_group_b(Game& g, const pk_u32x4 w)
{
    unpack_player_core(g.p2, w.x);
    unpack_player_misc(g.p2, w.y);
    g.b.power = (int)((w.y >> 9) & 1u);
    g.b.xv = (int)((w.y >> 10) & 0x3Fu) - 32;
    g.b.yv = (int)(w.y << 3) >> 19;
    g.b.x = (int)(w.z & 0x1FFu);
    g.b.px = (int)((w.z >> 9) & 0x1FFu);
    g.b.ppx = (int)((w.z >> 18) & 0x1FFu);
    g.b.y = (int)(w.w << 22) >> 22;
    g.b.py = (int)(w.w << 12) >> 22;
    g.b.rot = (int)((w.w >> 20) & 0x3Fu);
}

// tail: expected_landing_point_x uint16 | player 1 computer_boldness uint8 <<16 | player 2's <<24
__device__ __forceinline__ uint32_t pack_tail(const Game& g)
{
    return (uint32_t)g.b.ex | ((uint32_t)g.p1.bold << 16) | ((uint32_t)g.p2.bold << 24);
}

__device__ __forceinline__ void unpack_tail(Game& g, uint32_t w)
{
    g.b.ex = (int)(w & 0xFFFFu);
    g.p1.bold = (int)((w >> 16) & 0xFFu);
    g.p2.bold = (int)(w >> 24);
}

// does every field of the game fit its packed field?  (pz_pack_state; the step kernels only check the ball's y velocity
// and y, whose copies the two trail fields are)
__device__ __forceinline__ bool player_fits(const Player& p)
{
    return (unsigned)p.x < 512u && (unsigned)p.y < 256u && (unsigned)(p.yv + 32) < 64u && (unsigned)p.state < 8u &&
           (unsigned)p.frame < 8u && (unsigned)p.delay < 8u && (p.arm == 1 || p.arm == -1) && (unsigned)(p.dive + 1) < 3u &&
           (unsigned)(p.lying + 2) < 8u && (unsigned)p.coll < 2u && (unsigned)p.standby < 2u && (unsigned)p.hitprev < 2u &&
           (unsigned)p.bold < 256u;
}

__device__ __forceinline__ bool game_fits(const Game& g)
{
    const Ball& b = g.b;
    return player_fits(g.p1) && player_fits(g.p2) && (unsigned)b.x < 512u && (unsigned)b.px < 512u && (unsigned)b.ppx < 512u &&
           (unsigned)(b.y + kPackedBallYMax + 1) <= (unsigned)(2 * kPackedBallYMax + 1) &&
           (unsigned)(b.py + kPackedBallYMax + 1) <= (unsigned)(2 * kPackedBallYMax + 1) &&
           (unsigned)(b.ppy + kPackedBallYMax + 1) <= (unsigned)(2 * kPackedBallYMax + 1) && (unsigned)(b.xv + 32) < 64u &&
           (unsigned)(b.yv + kPackedBallYvMax + 1) <= (unsigned)(2 * kPackedBallYvMax + 1) && (unsigned)b.power < 2u &&
           (unsigned)b.rot < 64u && (unsigned)b.ex < 65536u && (unsigned)b.punch < 512u && (unsigned)g.e.s1 < 65536u &&
           (unsigned)g.e.s2 < 65536u && (unsigned)g.e.p2serve < 2u && (unsigned)g.e.round_ended < 2u &&
           (unsigned)g.e.game_ended < 2u;
}

}  // namespace pz
