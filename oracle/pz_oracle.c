/*
 * pz_oracle.c -- CPU restatement of the pika-zoo per-timestep path (TEST INFRASTRUCTURE).
 *
 * See pz_oracle.h for the role of this file.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/).  Nothing here is used by
 * the product path; the product (pika-zoo_amd/) fails loudly without its HIP library.
 *
 * Parity: pinned by the .npz fixtures under tests/golden/ (captured from the unmodified reference with
 * oracle/ref_capture.py) and by live comparison in the build container.
 */
#include "pz_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- constants: pikazoo/env/physics.py:9-33 ------------------------------------------- */
#define GROUND_WIDTH 432
#define GROUND_HALF_WIDTH 216
#define PLAYER_LENGTH 64
#define PLAYER_HALF_LENGTH 32
#define PLAYER_TOUCHING_GROUND_Y_COORD 244
#define BALL_RADIUS 20
#define BALL_TOUCHING_GROUND_Y_COORD 252
#define NET_PILLAR_HALF_WIDTH 25
#define NET_PILLAR_TOP_TOP_Y_COORD 176
#define NET_PILLAR_TOP_BOTTOM_Y_COORD 192
#define INFINITE_LOOP_LIMIT 1000

typedef struct {
    int x, y, y_velocity, state, frame_number, normal_status_arm_swing_direction,
        delay_before_next_frame, diving_direction, lying_down_duration_left,
        is_collision_with_ball_happened, computer_boldness, computer_where_to_stand_by;
    int power_hit_key_is_down_previous; /* PikaUserInput field, physics.py:51 */
    int is_player2, is_computer;        /* construction constants, physics.py:152-154 */
} Player;

typedef struct {
    int x, y, x_velocity, y_velocity, is_power_hit, previous_x, previous_y,
        previous_previous_x, previous_previous_y, fine_rotation,
        expected_landing_point_x, punch_effect_x;
} Ball;

typedef struct {
    int x_direction, y_direction, power_hit;
} UserInput;

typedef struct {
    Player p[2];
    Ball ball;
    int scores[2], is_player2_serve, round_ended, game_ended;
    uint32_t rng_counter;
    uint64_t seed;
    int64_t env_id;
} Game;

/* ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
 * 1, 2, 3", SC'11).  This is the build's RNG contract: the
 * reference's only RNG call is np_random.integers(lo, hi) (physics.py:218,613,728,729,795;
 * pikazoo_env.py:246) and the stream is injected, so PCG64 is not a parity dependency. --- */
void pzo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream tags in counter word 3 */
#define STREAM_ENV 0u
#define STREAM_ACTION 1u

int32_t pzo_env_draw(uint64_t seed, int64_t env_id, uint32_t idx, uint32_t n)
{
    uint32_t o[4];
    pzo_philox4x32_10((uint32_t)env_id, (uint32_t)((uint64_t)env_id >> 32), idx, STREAM_ENV,
                      (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return (int32_t)(((uint64_t)o[0] * n) >> 32);
}

/* np_random.integers(0, n): one draw = one counter tick, in the reference's call order */
static int rng_integers(Game *g, int n)
{
    return pzo_env_draw(g->seed, g->env_id, g->rng_counter++, (uint32_t)n);
}

void pzo_random_actions(int32_t *act_p1, int32_t *act_p2, int64_t n, int64_t env_id_base,
                        uint64_t action_seed, uint64_t t, int32_t n_actions)
{
    for (int64_t i = 0; i < n; ++i) {
        uint64_t id = (uint64_t)(env_id_base + i);
        uint32_t o[4];
        pzo_philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), (uint32_t)t,
                          STREAM_ACTION + 2u * (uint32_t)(t >> 32),
                          (uint32_t)action_seed, (uint32_t)(action_seed >> 32), o);
        act_p1[i] = (int32_t)(((uint64_t)o[0] * (uint32_t)n_actions) >> 32);
        act_p2[i] = (int32_t)(((uint64_t)o[1] * (uint32_t)n_actions) >> 32);
    }
}

static int iabs(int v) { return v < 0 ? -v : v; }
/* Python floor division by 2 (physics.py:373: -15 // 2 == -8) */
static int floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

/* ---- Player.initialize_for_new_round: physics.py:181-218 ----------------------------- */
static void player_initialize_for_new_round(Game *g, Player *pl)
{
    pl->x = pl->is_player2 ? GROUND_WIDTH - 36 : 36;
    pl->y = PLAYER_TOUCHING_GROUND_Y_COORD;
    pl->y_velocity = 0;
    pl->is_collision_with_ball_happened = 0;
    pl->state = 0;
    pl->frame_number = 0;
    pl->normal_status_arm_swing_direction = 1;
    pl->delay_before_next_frame = 0;
    pl->computer_boldness = rng_integers(g, 5); /* drawn for human players too (:218) */
}

/* ---- Ball.initialize_for_new_round: physics.py:258-277 ------------------------------- */
static void ball_initialize_for_new_round(Ball *b, int is_player2_serve)
{
    b->x = is_player2_serve ? GROUND_WIDTH - 56 : 56;
    b->y = 0;
    b->x_velocity = 0;
    b->y_velocity = 1;
    b->is_power_hit = 0;
}

/* ---- raw_env.get_server: pikazoo_env.py:242-248 -------------------------------------- */
static int get_server(Game *g, const pzo_config *cfg)
{
    if (cfg->serve_mode == PZO_SERVE_WINNER)
        return g->is_player2_serve;
    if (cfg->serve_mode == PZO_SERVE_RANDOM)
        return rng_integers(g, 2) == 0; /* True => player 2 serves (:246) */
    return (g->scores[0] + g->scores[1]) % 2 == 1;
}

/* ---- constructor: pikazoo_env.py:79-111 -> physics.py:107-123,143-171,224-249 ---------- */
static void game_construct(Game *g, const pzo_config *cfg, int64_t env_id)
{
    memset(g, 0, sizeof(*g));
    g->seed = cfg->seed;
    g->env_id = env_id;
    g->rng_counter = 0;
    for (int i = 0; i < 2; ++i) {
        Player *pl = &g->p[i];
        pl->is_player2 = i;
        pl->is_computer = i ? cfg->p2_computer : cfg->p1_computer;
        player_initialize_for_new_round(g, pl); /* physics.py:156 */
        pl->diving_direction = 0;               /* :159 */
        pl->lying_down_duration_left = -1;      /* :160 */
        pl->computer_where_to_stand_by = 0;     /* :171 */
        pl->power_hit_key_is_down_previous = 0; /* :51 */
    }
    ball_initialize_for_new_round(&g->ball, 0); /* physics.py:122 Ball(False) */
    g->ball.expected_landing_point_x = 0;       /* :232 */
    g->ball.fine_rotation = 0;                  /* :238 */
    g->ball.punch_effect_x = 0;                 /* :240 */
    g->ball.previous_x = g->ball.previous_previous_x = 0; /* :246-247 */
    g->ball.previous_y = g->ball.previous_previous_y = 0; /* :248-249 */
    g->scores[0] = g->scores[1] = 0;            /* pikazoo_env.py:100 */
    g->game_ended = g->round_ended = g->is_player2_serve = 0; /* :107-111 */
}

/* ---- raw_env.reset: pikazoo_env.py:149-173 (seed/options ignored there) ---------------- */
static void game_reset(Game *g, const pzo_config *cfg)
{
    g->game_ended = 0;
    g->round_ended = 0;
    g->is_player2_serve = 0;
    g->scores[0] = 0;
    g->scores[1] = 0;
    player_initialize_for_new_round(g, &g->p[0]);
    player_initialize_for_new_round(g, &g->p[1]);
    ball_initialize_for_new_round(&g->ball, get_server(g, cfg));
}

/* ---- action table pikazoo_env.py:119-141 + PikaUserInput.get_input physics.py:59-99 ---- */
static const uint8_t ACTION_KEY_MAP[18][5] = {
    /* left right up down power_hit */
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 1}, {0, 0, 1, 0, 0}, {0, 1, 0, 0, 0}, {1, 0, 0, 0, 0},
    {0, 0, 0, 1, 0}, {0, 1, 1, 0, 0}, {1, 0, 1, 0, 0}, {0, 1, 0, 1, 0}, {1, 0, 0, 1, 0},
    {0, 0, 1, 0, 1}, {0, 1, 0, 0, 1}, {1, 0, 0, 0, 1}, {0, 0, 0, 1, 1}, {0, 1, 1, 0, 1},
    {1, 0, 1, 0, 1}, {0, 1, 0, 1, 1}, {1, 0, 0, 1, 1},
};

static void get_input(Player *pl, UserInput *in, int action)
{
    const uint8_t *k = ACTION_KEY_MAP[action];
    /* 5-entry rows => down_right_key is None for both players (physics.py:69-70) */
    if (k[0]) in->x_direction = -1;
    else if (k[1]) in->x_direction = 1;
    else in->x_direction = 0;
    if (k[2]) in->y_direction = -1;
    else if (k[3]) in->y_direction = 1;
    else in->y_direction = 0;
    int is_down = k[4];
    in->power_hit = (!pl->power_hit_key_is_down_previous && is_down) ? 1 : 0;
    pl->power_hit_key_is_down_previous = is_down;
}

/* ---- wrappers/simplify_action.py:16-19 ------------------------------------------------ */
static const int8_t SIMPLIFY_MAP[2][13] = {
    {0, 1, 2, 3, 4, 6, 7, 10, 11, 12, 13, 14, 16},
    {0, 1, 2, 4, 3, 7, 6, 10, 12, 11, 13, 15, 17},
};

/* ---- is_collision_between_ball_and_player_happened: physics.py:340-356 ---------------- */
static int is_collision_between_ball_and_player_happened(const Ball *b, int px, int py)
{
    if (iabs(b->x - px) <= PLAYER_HALF_LENGTH)
        if (iabs(b->y - py) <= PLAYER_HALF_LENGTH)
            return 1;
    return 0;
}

/* ---- process_collision_between_ball_and_world_and_set_ball_position: physics.py:359-436 */
static int process_collision_between_ball_and_world_and_set_ball_position(Ball *b)
{
    b->previous_previous_x = b->previous_x;
    b->previous_previous_y = b->previous_y;
    b->previous_x = b->x;
    b->previous_y = b->y;

    int future_fine_rotation = b->fine_rotation + floordiv2(b->x_velocity);
    if (future_fine_rotation < 0)
        future_fine_rotation += 50;
    else if (future_fine_rotation > 50)
        future_fine_rotation += -50;
    b->fine_rotation = future_fine_rotation;

    int future_ball_x = b->x + b->x_velocity;
    if (future_ball_x < BALL_RADIUS || future_ball_x > GROUND_WIDTH)
        b->x_velocity = -b->x_velocity;

    int future_ball_y = b->y + b->y_velocity;
    if (future_ball_y < 0)
        b->y_velocity = 1;

    if (iabs(b->x - GROUND_HALF_WIDTH) < NET_PILLAR_HALF_WIDTH && b->y > NET_PILLAR_TOP_TOP_Y_COORD) {
        if (b->y <= NET_PILLAR_TOP_BOTTOM_Y_COORD) {
            if (b->y_velocity > 0)
                b->y_velocity = -b->y_velocity;
        } else {
            if (b->x < GROUND_HALF_WIDTH)
                b->x_velocity = -iabs(b->x_velocity);
            else
                b->x_velocity = iabs(b->x_velocity);
        }
    }

    future_ball_y = b->y + b->y_velocity;
    if (future_ball_y > BALL_TOUCHING_GROUND_Y_COORD) {
        b->y_velocity = -b->y_velocity;
        b->punch_effect_x = b->x;
        b->y = BALL_TOUCHING_GROUND_Y_COORD;
        return 1;
    }
    b->y = future_ball_y;
    b->x = b->x + b->x_velocity;
    b->y_velocity += 1;
    return 0;
}

/* Diagnostic builds only (tools/flight_trips.c defines PZO_FLIGHT_TRACE and the hook): every call of
 * the two flight predictors is reported, so trip counts of alternative formulations can be modelled
 * on states sampled from play. */
#ifdef PZO_FLIGHT_TRACE
void pzo_flight_trace(int kind, int x, int y, int xv, int yv);
#else
#define pzo_flight_trace(kind, x, y, xv, yv) ((void)0)
#endif

/* ---- calculate_expected_landing_point_x_for: physics.py:643-686 ----------------------- */
static int landing_x(int x, int y, int xv, int yv)
{
    pzo_flight_trace(0, x, y, xv, yv);
    int loop_counter = 0;
    for (;;) {
        loop_counter += 1;
        int future_x = xv + x;
        if (future_x < BALL_RADIUS || future_x > GROUND_WIDTH)
            xv = -xv;
        if (y + yv < 0)
            yv = 1;
        if (iabs(x - GROUND_HALF_WIDTH) < NET_PILLAR_HALF_WIDTH && y > NET_PILLAR_TOP_TOP_Y_COORD) {
            if (y < NET_PILLAR_TOP_BOTTOM_Y_COORD) { /* strict here (:670), <= in the real ball (:412) */
                if (yv > 0)
                    yv = -yv;
            } else {
                if (x < GROUND_HALF_WIDTH)
                    xv = -iabs(xv);
                else
                    xv = iabs(xv);
            }
        }
        y = y + yv;
        if (y > BALL_TOUCHING_GROUND_Y_COORD || loop_counter >= INFINITE_LOOP_LIMIT)
            break;
        x = x + xv;
        yv += 1;
    }
    return x;
}

static void calculate_expected_landing_point_x_for(Ball *b)
{
    b->expected_landing_point_x = landing_x(b->x, b->y, b->x_velocity, b->y_velocity);
}

int32_t pzo_expected_landing_x(int32_t x, int32_t y, int32_t xv, int32_t yv)
{
    return landing_x(x, y, xv, yv);
}

/* ---- expected_landing_point_x_when_power_hit: physics.py:820-884 ---------------------- */
static int expected_landing_point_x_when_power_hit(int xdir, int ydir, const Ball *b)
{
    int x = b->x, y = b->y, xv, yv = b->y_velocity;
    if (x < GROUND_HALF_WIDTH)
        xv = (iabs(xdir) + 1) * 10;
    else
        xv = -(iabs(xdir) + 1) * 10;
    yv = iabs(yv) * ydir * 2;

    int loop_counter = 0;
    for (;;) {
        loop_counter += 1;
        int future_x = x + xv;
        if (future_x < BALL_RADIUS || future_x > GROUND_WIDTH)
            xv = -xv;
        if (y + yv < 0)
            yv = 1;
        if (iabs(x - GROUND_HALF_WIDTH) < NET_PILLAR_HALF_WIDTH && y > NET_PILLAR_TOP_TOP_Y_COORD) {
            /* simplified net rule, no side bounce (:860-866) */
            if (yv > 0)
                yv = -yv;
        }
        y = y + yv;
        if (y > BALL_TOUCHING_GROUND_Y_COORD || loop_counter >= INFINITE_LOOP_LIMIT)
            return x;
        x = x + xv;
        yv += 1;
    }
}

int32_t pzo_expected_landing_x_power_hit(int32_t xdir, int32_t ydir,
                                         int32_t x, int32_t y, int32_t xv, int32_t yv)
{
    Ball b;
    memset(&b, 0, sizeof(b));
    b.x = x; b.y = y; b.x_velocity = xv; b.y_velocity = yv;
    return expected_landing_point_x_when_power_hit(xdir, ydir, &b);
}

/* ---- decide_whether_input_power_hit: physics.py:774-817 ------------------------------- */
static int decide_whether_input_power_hit(Game *g, const Player *pl, const Ball *b,
                                          const Player *other, UserInput *in)
{
    int ascending = rng_integers(g, 2) == 0; /* :795 */
    pzo_flight_trace(1, b->x, b->y, 0, iabs(b->y_velocity));
    for (int x_direction = 1; x_direction > -1; --x_direction) {
        for (int j = 0; j < 3; ++j) {
            int y_direction = ascending ? (-1 + j) : (1 - j); /* :797 vs :808 */
            int ex = expected_landing_point_x_when_power_hit(x_direction, y_direction, b);
            if ((ex <= pl->is_player2 * GROUND_HALF_WIDTH ||
                 ex >= pl->is_player2 * GROUND_WIDTH + GROUND_HALF_WIDTH) &&
                iabs(ex - other->x) > PLAYER_LENGTH) {
                in->x_direction = x_direction;
                in->y_direction = y_direction;
                return 1;
            }
        }
    }
    return 0;
}

/* ---- let_computer_decide_user_input: physics.py:689-771 -------------------------------- */
static void let_computer_decide_user_input(Game *g, Player *pl, const Ball *b,
                                           const Player *other, UserInput *in)
{
    in->x_direction = 0;
    in->y_direction = 0;
    in->power_hit = 0;

    int virtual_expected_landing_point_x = b->expected_landing_point_x;
    if (iabs(b->x - pl->x) > 100 && iabs(b->x_velocity) < pl->computer_boldness + 5) {
        int left_boundary = pl->is_player2 * GROUND_HALF_WIDTH;
        if ((b->expected_landing_point_x <= left_boundary ||
             b->expected_landing_point_x >= pl->is_player2 * GROUND_WIDTH + GROUND_HALF_WIDTH) &&
            pl->computer_where_to_stand_by == 0) {
            virtual_expected_landing_point_x = left_boundary + GROUND_HALF_WIDTH / 2;
        }
    }

    if (iabs(virtual_expected_landing_point_x - pl->x) > pl->computer_boldness + 8) {
        if (pl->x < virtual_expected_landing_point_x)
            in->x_direction = 1;
        else
            in->x_direction = -1;
    } else if (rng_integers(g, 20) == 0) {                     /* :728 */
        pl->computer_where_to_stand_by = rng_integers(g, 2);   /* :729 */
    }

    if (pl->state == 0) {
        if (iabs(b->x_velocity) < pl->computer_boldness + 3 &&
            iabs(b->x - pl->x) < PLAYER_HALF_LENGTH &&
            b->y > -36 && b->y < 10 * pl->computer_boldness + 84 && b->y_velocity > 0) {
            in->y_direction = -1;
        }
        int left_boundary = pl->is_player2 * GROUND_HALF_WIDTH;
        int right_boundary = (pl->is_player2 + 1) * GROUND_HALF_WIDTH;
        if (b->expected_landing_point_x > left_boundary &&
            b->expected_landing_point_x < right_boundary &&
            iabs(b->x - pl->x) > pl->computer_boldness * 5 + PLAYER_LENGTH &&
            b->x > left_boundary && b->x < right_boundary && b->y > 174) {
            in->power_hit = 1; /* dive */
            if (pl->x < b->x)
                in->x_direction = 1;
            else
                in->x_direction = -1;
        }
    } else if (pl->state == 1 || pl->state == 2) {
        if (iabs(b->x - pl->x) > 8) {
            if (pl->x < b->x)
                in->x_direction = 1;
            else
                in->x_direction = -1;
        }
        if (iabs(b->x - pl->x) < 48 && iabs(b->y - pl->y) < 48) {
            int will_input_power_hit = decide_whether_input_power_hit(g, pl, b, other, in);
            if (will_input_power_hit) {
                in->power_hit = 1;
                if (iabs(other->x - pl->x) < 80 && in->y_direction != -1)
                    in->y_direction = -1;
            }
        }
    }
}

/* ---- process_player_movement_and_set_player_position: physics.py:439-564 --------------- */
static void process_player_movement_and_set_player_position(Game *g, Player *pl, UserInput *in,
                                                            const Player *other, const Ball *b)
{
    if (pl->is_computer)
        let_computer_decide_user_input(g, pl, b, other, in);

    if (pl->state == 4) {
        pl->lying_down_duration_left += -1;
        if (pl->lying_down_duration_left < -1)
            pl->state = 0;
        return;
    }

    int player_velocity_x = 0;
    if (pl->state < 5) {
        if (pl->state < 3)
            player_velocity_x = in->x_direction * 6;
        else
            player_velocity_x = pl->diving_direction * 8;
    }
    int future_player_x = pl->x + player_velocity_x;
    pl->x = future_player_x;

    if (!pl->is_player2) {
        if (future_player_x < PLAYER_HALF_LENGTH)
            pl->x = PLAYER_HALF_LENGTH;
        else if (future_player_x > GROUND_HALF_WIDTH - PLAYER_HALF_LENGTH)
            pl->x = GROUND_HALF_WIDTH - PLAYER_HALF_LENGTH;
    } else {
        if (future_player_x < GROUND_HALF_WIDTH + PLAYER_HALF_LENGTH)
            pl->x = GROUND_HALF_WIDTH + PLAYER_HALF_LENGTH;
        else if (future_player_x > GROUND_WIDTH - PLAYER_HALF_LENGTH)
            pl->x = GROUND_WIDTH - PLAYER_HALF_LENGTH;
    }

    if (pl->state < 3 && in->y_direction == -1 && pl->y == PLAYER_TOUCHING_GROUND_Y_COORD) {
        pl->y_velocity = -16;
        pl->state = 1;
        pl->frame_number = 0;
    }

    int future_player_y = pl->y + pl->y_velocity;
    pl->y = future_player_y;
    if (future_player_y < PLAYER_TOUCHING_GROUND_Y_COORD) {
        pl->y_velocity += 1;
    } else if (future_player_y > PLAYER_TOUCHING_GROUND_Y_COORD) {
        pl->y_velocity = 0;
        pl->y = PLAYER_TOUCHING_GROUND_Y_COORD;
        pl->frame_number = 0;
        if (pl->state == 3) {
            pl->state = 4;
            pl->frame_number = 0;
            pl->lying_down_duration_left = 3;
        } else {
            pl->state = 0;
        }
    }

    if (in->power_hit == 1) {
        if (pl->state == 1) {
            pl->delay_before_next_frame = 5;
            pl->frame_number = 0;
            pl->state = 2;
        } else if (pl->state == 0 && in->x_direction != 0) {
            pl->state = 3;
            pl->frame_number = 0;
            pl->diving_direction = in->x_direction;
            pl->y_velocity = -5;
        }
    }

    if (pl->state == 1) {
        pl->frame_number = (pl->frame_number + 1) % 3;
    } else if (pl->state == 2) {
        if (pl->delay_before_next_frame < 1) {
            pl->frame_number += 1;
            if (pl->frame_number > 4) {
                pl->frame_number = 0;
                pl->state = 1;
            }
        } else {
            pl->delay_before_next_frame -= 1;
        }
    } else if (pl->state == 0) {
        pl->delay_before_next_frame += 1;
        if (pl->delay_before_next_frame > 3) {
            pl->delay_before_next_frame = 0;
            int future_frame_number = pl->frame_number + pl->normal_status_arm_swing_direction;
            if (future_frame_number < 0 || future_frame_number > 4)
                pl->normal_status_arm_swing_direction = -pl->normal_status_arm_swing_direction;
            pl->frame_number = pl->frame_number + pl->normal_status_arm_swing_direction;
        }
    }
    /* physics.py:554-564 (game-end animation) is unreachable under the env: the flags are
     * set after the physics call on the terminal step (pikazoo_env.py:194-208) and reset()
     * clears them before the next frame. */
}

/* ---- process_collision_between_ball_and_player: physics.py:580-640 --------------------- */
static void process_collision_between_ball_and_player(Game *g, Ball *b, int player_x,
                                                      const UserInput *in, int player_state)
{
    if (b->x < player_x)
        b->x_velocity = -(iabs(b->x - player_x) / 3);
    else if (b->x > player_x)
        b->x_velocity = iabs(b->x - player_x) / 3;

    if (b->x_velocity == 0)
        b->x_velocity = rng_integers(g, 3) - 1; /* :613 */

    int ball_abs_y_velocity = iabs(b->y_velocity);
    b->y_velocity = -ball_abs_y_velocity;
    if (ball_abs_y_velocity < 15)
        b->y_velocity = -15;

    if (player_state == 2) {
        if (b->x < GROUND_HALF_WIDTH)
            b->x_velocity = (iabs(in->x_direction) + 1) * 10;
        else
            b->x_velocity = -(iabs(in->x_direction) + 1) * 10;
        b->punch_effect_x = b->x;
        b->y_velocity = iabs(b->y_velocity) * in->y_direction * 2;
        b->is_power_hit = 1;
    } else {
        b->is_power_hit = 0;
    }
}

/* ---- physics_engine: physics.py:280-337 ------------------------------------------------ */
static int physics_engine(Game *g, UserInput in[2])
{
    Player *player1 = &g->p[0], *player2 = &g->p[1];
    Ball *ball = &g->ball;
    int is_ball_touching_ground = process_collision_between_ball_and_world_and_set_ball_position(ball);

    for (int i = 0; i < 2; ++i) {
        Player *player = i == 0 ? player1 : player2;
        Player *the_other_player = i == 0 ? player2 : player1;
        if (player1->is_computer || player2->is_computer)
            calculate_expected_landing_point_x_for(ball);
        process_player_movement_and_set_player_position(g, player, &in[i], the_other_player, ball);
    }

    for (int i = 0; i < 2; ++i) {
        Player *player = i == 0 ? player1 : player2;
        int is_happened = is_collision_between_ball_and_player_happened(ball, player->x, player->y);
        if (is_happened) {
            if (!player->is_collision_with_ball_happened) {
                process_collision_between_ball_and_player(g, ball, player->x, &in[i], player->state);
                if (player1->is_computer || player2->is_computer)
                    calculate_expected_landing_point_x_for(ball);
                player->is_collision_with_ball_happened = 1;
            }
        } else {
            player->is_collision_with_ball_happened = 0;
        }
    }
    return is_ball_touching_ground;
}

/* ---- _get_obs: pikazoo_env.py:576-624 --------------------------------------------------- */
static void player_info(const Player *pl, int32_t *o)
{
    o[0] = pl->x;
    o[1] = pl->y;
    o[2] = pl->y_velocity;
    o[3] = pl->diving_direction;
    o[4] = pl->lying_down_duration_left;
    o[5] = pl->frame_number;
    o[6] = pl->delay_before_next_frame;
    for (int s = 0; s < 5; ++s)
        o[7 + s] = pl->state == s;
    o[12] = pl->power_hit_key_is_down_previous;
}

/* observation_space bounds, pikazoo_env.py:485-562 (player, opponent, ball) */
static const int OBS_LOW[35] = {32, 108, -15, -1, -2, 0, 0, 0, 0, 0, 0, 0, 0,
                                32, 108, -15, -1, -2, 0, 0, 0, 0, 0, 0, 0, 0,
                                20, 0, 0, 0, 0, 0, -20, -124, 0};
static const int OBS_HIGH[35] = {400, 244, 16, 1, 3, 4, 4, 1, 1, 1, 1, 1, 1,
                                 400, 244, 16, 1, 3, 4, 4, 1, 1, 1, 1, 1, 1,
                                 432, 252, 432, 252, 432, 252, 20, 124, 1};

/* NormalizeObservation (wrappers/normalize_observation.py:22,30): (obs - low) / (high - low),
 * computed there in float64; the build emits float32 = the correctly rounded quotient, which
 * equals float32(float64 quotient) for these small integers. */
static void normalize_row(int32_t *row)
{
    for (int j = 0; j < 35; ++j) {
        float f = (float)(row[j] - OBS_LOW[j]) / (float)(OBS_HIGH[j] - OBS_LOW[j]);
        memcpy(&row[j], &f, 4);
    }
}

static void get_obs_n(const Game *g, int32_t *obs1, int32_t *obs2, int normalize)
{
    int32_t p1[13], p2[13], bo[9];
    player_info(&g->p[0], p1);
    player_info(&g->p[1], p2);
    const Ball *b = &g->ball;
    bo[0] = b->x; bo[1] = b->y; bo[2] = b->previous_x; bo[3] = b->previous_y;
    bo[4] = b->previous_previous_x; bo[5] = b->previous_previous_y;
    bo[6] = b->x_velocity; bo[7] = b->y_velocity; bo[8] = b->is_power_hit;
    if (obs1) {
        memcpy(obs1, p1, sizeof p1); memcpy(obs1 + 13, p2, sizeof p2); memcpy(obs1 + 26, bo, sizeof bo);
    }
    if (obs2) {
        memcpy(obs2, p2, sizeof p2); memcpy(obs2 + 13, p1, sizeof p1); memcpy(obs2 + 26, bo, sizeof bo);
    }
    if (normalize) {
        if (obs1) normalize_row(obs1);
        if (obs2) normalize_row(obs2);
    }
}

/* ---- SoA <-> Game ------------------------------------------------------------------------ */
static void load_player(Player *pl, const int32_t *s, int64_t stride, int base, int is_p2, int is_comp)
{
    pl->x = s[(base + PZO_P_X) * stride];
    pl->y = s[(base + PZO_P_Y) * stride];
    pl->y_velocity = s[(base + PZO_P_YVEL) * stride];
    pl->state = s[(base + PZO_P_STATE) * stride];
    pl->frame_number = s[(base + PZO_P_FRAME) * stride];
    pl->normal_status_arm_swing_direction = s[(base + PZO_P_ARM_SWING) * stride];
    pl->delay_before_next_frame = s[(base + PZO_P_DELAY) * stride];
    pl->diving_direction = s[(base + PZO_P_DIVING_DIR) * stride];
    pl->lying_down_duration_left = s[(base + PZO_P_LYING_DOWN) * stride];
    pl->is_collision_with_ball_happened = s[(base + PZO_P_COLLISION) * stride];
    pl->computer_boldness = s[(base + PZO_P_BOLDNESS) * stride];
    pl->computer_where_to_stand_by = s[(base + PZO_P_STAND_BY) * stride];
    pl->power_hit_key_is_down_previous = s[(base + PZO_P_HIT_KEY_PREV) * stride];
    pl->is_player2 = is_p2;
    pl->is_computer = is_comp;
}

static void store_player(const Player *pl, int32_t *s, int64_t stride, int base)
{
    s[(base + PZO_P_X) * stride] = pl->x;
    s[(base + PZO_P_Y) * stride] = pl->y;
    s[(base + PZO_P_YVEL) * stride] = pl->y_velocity;
    s[(base + PZO_P_STATE) * stride] = pl->state;
    s[(base + PZO_P_FRAME) * stride] = pl->frame_number;
    s[(base + PZO_P_ARM_SWING) * stride] = pl->normal_status_arm_swing_direction;
    s[(base + PZO_P_DELAY) * stride] = pl->delay_before_next_frame;
    s[(base + PZO_P_DIVING_DIR) * stride] = pl->diving_direction;
    s[(base + PZO_P_LYING_DOWN) * stride] = pl->lying_down_duration_left;
    s[(base + PZO_P_COLLISION) * stride] = pl->is_collision_with_ball_happened;
    s[(base + PZO_P_BOLDNESS) * stride] = pl->computer_boldness;
    s[(base + PZO_P_STAND_BY) * stride] = pl->computer_where_to_stand_by;
    s[(base + PZO_P_HIT_KEY_PREV) * stride] = pl->power_hit_key_is_down_previous;
}

static void load_game(Game *g, const int32_t *s, int64_t stride, const pzo_config *cfg, int64_t env_id)
{
    load_player(&g->p[0], s, stride, 0, 0, cfg->p1_computer);
    load_player(&g->p[1], s, stride, PZO_P_WORDS, 1, cfg->p2_computer);
    Ball *b = &g->ball;
    b->x = s[PZO_B_X * stride];
    b->y = s[PZO_B_Y * stride];
    b->x_velocity = s[PZO_B_XVEL * stride];
    b->y_velocity = s[PZO_B_YVEL * stride];
    b->is_power_hit = s[PZO_B_POWER_HIT * stride];
    b->previous_x = s[PZO_B_PREV_X * stride];
    b->previous_y = s[PZO_B_PREV_Y * stride];
    b->previous_previous_x = s[PZO_B_PPREV_X * stride];
    b->previous_previous_y = s[PZO_B_PPREV_Y * stride];
    b->fine_rotation = s[PZO_B_FINE_ROT * stride];
    b->expected_landing_point_x = s[PZO_B_EXPECTED_X * stride];
    b->punch_effect_x = s[PZO_B_PUNCH_X * stride];
    g->scores[0] = s[PZO_E_SCORE1 * stride];
    g->scores[1] = s[PZO_E_SCORE2 * stride];
    g->is_player2_serve = s[PZO_E_P2_SERVE * stride];
    g->round_ended = s[PZO_E_ROUND_ENDED * stride];
    g->game_ended = s[PZO_E_GAME_ENDED * stride];
    g->rng_counter = (uint32_t)s[PZO_E_RNG_COUNTER * stride];
    g->seed = cfg->seed;
    g->env_id = env_id;
}

static void store_game(const Game *g, int32_t *s, int64_t stride)
{
    store_player(&g->p[0], s, stride, 0);
    store_player(&g->p[1], s, stride, PZO_P_WORDS);
    const Ball *b = &g->ball;
    s[PZO_B_X * stride] = b->x;
    s[PZO_B_Y * stride] = b->y;
    s[PZO_B_XVEL * stride] = b->x_velocity;
    s[PZO_B_YVEL * stride] = b->y_velocity;
    s[PZO_B_POWER_HIT * stride] = b->is_power_hit;
    s[PZO_B_PREV_X * stride] = b->previous_x;
    s[PZO_B_PREV_Y * stride] = b->previous_y;
    s[PZO_B_PPREV_X * stride] = b->previous_previous_x;
    s[PZO_B_PPREV_Y * stride] = b->previous_previous_y;
    s[PZO_B_FINE_ROT * stride] = b->fine_rotation;
    s[PZO_B_EXPECTED_X * stride] = b->expected_landing_point_x;
    s[PZO_B_PUNCH_X * stride] = b->punch_effect_x;
    s[PZO_E_SCORE1 * stride] = g->scores[0];
    s[PZO_E_SCORE2 * stride] = g->scores[1];
    s[PZO_E_P2_SERVE * stride] = g->is_player2_serve;
    s[PZO_E_ROUND_ENDED * stride] = g->round_ended;
    s[PZO_E_GAME_ENDED * stride] = g->game_ended;
    s[PZO_E_RNG_COUNTER * stride] = (int32_t)g->rng_counter;
}

/* ---- raw_env.step: pikazoo_env.py:175-240, one game -------------------------------------- */
typedef struct {
    double *ret1, *ret2; /* this lane's running episode returns (NULL = off) ... */
    int32_t *len;        /* ... and episode length */
} Stats;

static int rewards_are_float(const pzo_config *cfg) { return cfg->ballpos_reward || cfg->normal_state_mode; }

static void stats_zero(const Stats *st)
{
    if (st && st->len) { *st->ret1 = 0.0; *st->ret2 = 0.0; *st->len = 0; }
}

static void stats_add(const Stats *st, const pzo_config *cfg, float f1, float f2, int i1, int i2, int as_float)
{
    if (!st || !st->len) return;
    (void)cfg;
    /* record_episode_statistics.py:31 sums Python floats: float64.  The fused wrappers' rewards are float32
     * (f1, f2), widened here; the env's own rewards are the integers +-1 / 0. */
    if (as_float) {
        *st->ret1 += (double)f1; *st->ret2 += (double)f2;
    } else {
        *st->ret1 += (double)i1; *st->ret2 += (double)i2;
    }
    *st->len += 1;
}

static void game_step(Game *g, const pzo_config *cfg, int a1, int a2,
                      int32_t *obs1, int32_t *obs2, void *rew1, void *rew2, uint8_t *term, const Stats *st)
{
    int frozen = 0;
    if (g->game_ended) {
        /* The reference empties `agents` on termination (:237-238) and the caller must
         * reset() before stepping again.  auto_reset: do exactly that, in place
         * (RecordEpisodeStatistics.reset zeroes its sums, record_episode_statistics.py:23-25). */
        if (cfg->auto_reset) {
            game_reset(g, cfg);
            stats_zero(st);
        } else {
            frozen = 1;
        }
    }

    int player1_reward = 0;
    if (!frozen) {
        if (g->round_ended && !g->game_ended) { /* :176-180 */
            player_initialize_for_new_round(g, &g->p[0]);
            player_initialize_for_new_round(g, &g->p[1]);
            ball_initialize_for_new_round(&g->ball, get_server(g, cfg));
            g->round_ended = 0;
        }

        if (cfg->simplify_action) { /* simplify_action.py:24 */
            a1 = SIMPLIFY_MAP[0][a1];
            a2 = SIMPLIFY_MAP[1][a2];
        }
        UserInput in[2];
        get_input(&g->p[0], &in[0], a1); /* :182-184 */
        get_input(&g->p[1], &in[1], a2);

        int is_ball_touching_ground = physics_engine(g, in); /* :186 */

        if (is_ball_touching_ground && !g->round_ended && !g->game_ended) { /* :190-210 */
            if (g->ball.punch_effect_x < GROUND_HALF_WIDTH) {
                g->is_player2_serve = 1;
                g->scores[1] += 1;
                if (g->scores[1] >= cfg->winning_score)
                    g->game_ended = 1;
            } else {
                g->is_player2_serve = 0;
                g->scores[0] += 1;
                if (g->scores[0] >= cfg->winning_score)
                    g->game_ended = 1;
            }
            g->round_ended = 1;
        }
        if (g->round_ended) /* :217-223 */
            player1_reward = g->is_player2_serve ? -1 : 1;
    }

    get_obs_n(g, obs1, obs2, cfg->normalize_obs); /* :215 */

    const int as_float = rewards_are_float(cfg);
    const int i1 = player1_reward, i2 = -player1_reward;
    float r1 = (float)i1, r2 = (float)i2;
    if (!frozen) {
        if (cfg->episode_stats_mode == 1)
            stats_add(st, cfg, (float)i1, (float)i2, i1, i2, as_float);
        if (cfg->normal_state_mode == 1) { /* reward_in_normal_state.py:12-14, inside the other wrapper */
            if (r1 == 0.0f) r1 = cfg->normal_state_reward;
            if (r2 == 0.0f) r2 = cfg->normal_state_reward;
        }
        if (cfg->ballpos_reward) { /* reward_by_ball_position.py:22-29 */
            int x_sign = g->ball.x >= cfg->x_line;
            int y_sign = g->ball.y > cfg->y_line;
            int ball_pos = 1 * y_sign + 2 * x_sign;
            r1 = r1 + cfg->additional_reward[0 * 4 + ball_pos];
            r2 = r2 + cfg->additional_reward[1 * 4 + ball_pos];
        }
        if (cfg->normal_state_mode == 2) { /* the wrapper outside RewardByBallPosition */
            if (r1 == 0.0f) r1 = cfg->normal_state_reward;
            if (r2 == 0.0f) r2 = cfg->normal_state_reward;
        }
        if (cfg->episode_stats_mode == 2)
            stats_add(st, cfg, r1, r2, i1, i2, as_float);
    }
    if (as_float) {
        *(float *)rew1 = r1;
        *(float *)rew2 = r2;
    } else {
        *(int32_t *)rew1 = i1;
        *(int32_t *)rew2 = i2;
    }
    *term = (uint8_t)g->game_ended; /* :233 */
}

/* ---- batched entry points ---------------------------------------------------------------- */
void pzo_init(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg)
{
    for (int64_t i = 0; i < n; ++i) {
        Game g;
        game_construct(&g, cfg, cfg->env_id_base + i);
        store_game(&g, state + i, stride);
    }
}

/* episode_stats buffer: double[2][stride] returns, then int32[stride] lengths (20 * stride bytes) */
static Stats lane_stats(void *episode_stats, int64_t stride, int64_t i)
{
    Stats st = {0, 0, 0};
    if (episode_stats) {
        st.ret1 = (double *)episode_stats + i;
        st.ret2 = (double *)episode_stats + stride + i;
        st.len = (int32_t *)((double *)episode_stats + 2 * stride) + i;
    }
    return st;
}

void pzo_reset(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
               const uint8_t *mask, int32_t *obs_p1, int32_t *obs_p2, void *episode_stats)
{
    for (int64_t i = 0; i < n; ++i) {
        Game g;
        load_game(&g, state + i, stride, cfg, cfg->env_id_base + i);
        if (!mask || mask[i]) {
            Stats st = lane_stats(episode_stats, stride, i);
            game_reset(&g, cfg);
            stats_zero(&st);
            store_game(&g, state + i, stride);
        }
        get_obs_n(&g, obs_p1 ? obs_p1 + i * PZO_OBS : 0, obs_p2 ? obs_p2 + i * PZO_OBS : 0, cfg->normalize_obs);
    }
}

void pzo_observe(const int32_t *state, int64_t n, int64_t stride, int32_t normalize, int32_t *obs_p1,
                 int32_t *obs_p2)
{
    pzo_config cfg;
    memset(&cfg, 0, sizeof cfg);
    for (int64_t i = 0; i < n; ++i) {
        Game g;
        load_game(&g, state + i, stride, &cfg, i);
        get_obs_n(&g, obs_p1 ? obs_p1 + i * PZO_OBS : 0, obs_p2 ? obs_p2 + i * PZO_OBS : 0, normalize);
    }
}

static void step_range(int32_t *state, int64_t lo, int64_t hi, int64_t stride, const pzo_config *cfg,
                       const int32_t *act_p1, const int32_t *act_p2, int32_t *obs_p1, int32_t *obs_p2,
                       void *rew_p1, void *rew_p2, uint8_t *terminated, void *episode_stats)
{
    for (int64_t i = lo; i < hi; ++i) {
        Game g;
        Stats st = lane_stats(episode_stats, stride, i);
        load_game(&g, state + i, stride, cfg, cfg->env_id_base + i);
        game_step(&g, cfg, act_p1[i], act_p2[i], obs_p1 + i * PZO_OBS, obs_p2 + i * PZO_OBS,
                  (char *)rew_p1 + 4 * i, (char *)rew_p2 + 4 * i, terminated + i, &st);
        store_game(&g, state + i, stride);
    }
}

void pzo_step(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
              const int32_t *act_p1, const int32_t *act_p2, int32_t *obs_p1, int32_t *obs_p2,
              void *rew_p1, void *rew_p2, uint8_t *terminated, void *episode_stats, int nthreads)
{
    if (nthreads <= 1) {
        step_range(state, 0, n, stride, cfg, act_p1, act_p2, obs_p1, obs_p2, rew_p1, rew_p2, terminated,
                   episode_stats);
        return;
    }
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int t = 0; t < nthreads; ++t) {
        int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        step_range(state, lo, hi, stride, cfg, act_p1, act_p2, obs_p1, obs_p2, rew_p1, rew_p2, terminated,
                   episode_stats);
    }
}

void pzo_rollout_random(int32_t *state, int64_t n, int64_t stride, const pzo_config *cfg,
                        uint64_t action_seed, uint64_t t0, int32_t k,
                        int32_t *obs_p1, int32_t *obs_p2, void *rew_p1, void *rew_p2,
                        uint8_t *terminated, void *episode_stats, int64_t *episodes_finished, int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    int32_t n_actions = cfg->simplify_action ? 13 : 18;
    int64_t finished = 0;
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(+ : finished)
    for (int t = 0; t < nthreads; ++t) {
        int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        for (int64_t i = lo; i < hi; ++i) {
            Game g;
            Stats st = lane_stats(episode_stats, stride, i);
            load_game(&g, state + i, stride, cfg, cfg->env_id_base + i);
            for (int32_t s = 0; s < k; ++s) {
                int32_t a1, a2;
                pzo_random_actions(&a1, &a2, 1, cfg->env_id_base + i, action_seed, t0 + (uint64_t)s, n_actions);
                game_step(&g, cfg, a1, a2, obs_p1 + i * PZO_OBS, obs_p2 + i * PZO_OBS,
                          (char *)rew_p1 + 4 * i, (char *)rew_p2 + 4 * i, terminated + i, &st);
                finished += terminated[i];
            }
            store_game(&g, state + i, stride);
        }
    }
    if (episodes_finished)
        *episodes_finished += finished;
}

uint64_t pzo_digest(const int32_t *state, int64_t n, int64_t stride)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int f = 0; f < PZO_W; ++f)
        for (int64_t i = 0; i < n; ++i) {
            uint32_t v = (uint32_t)state[f * stride + i];
            for (int b = 0; b < 4; ++b) {
                h ^= (v >> (8 * b)) & 0xffu;
                h *= 0x100000001b3ull;
            }
        }
    return h;
}
