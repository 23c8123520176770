"""CPU restatement of the reference's ``rgb_array`` frame (TEST INFRASTRUCTURE, not product).

``frame(state_column, sprites, background[, scenery])`` draws one game the way ``raw_env.draw`` does
(pikazoo/env/pikazoo_env.py:250-353) minus the punch effect (:292-294: its radius and y are set by the physics in two
ball attributes outside the 44 state words), with numpy slices instead of pygame blits; ``scenery_init`` /
``scenery_tick`` restate the clouds and waves (cloud_and_wave.py), which live outside the 44 words and advance the env
RNG on every render; ``draw_list`` is the list of blits a frame consists of.

Parity: **the draw list, the clouds / waves and the RNG consumption are pinned** against the unmodified reference:
``oracle/ref_capture.py`` runs the reference's own ``render()`` on a recording stand-in for pygame and commits what it
drew (which file, mirrored, position, size of every blit), the cloud / wave state and the env state after every
frame (tests/golden/render_*.npz; tests/test_render_cpu.py).  **The pixel arithmetic is not pinned** -- pygame is not
installed in the build container: the per-pixel-alpha blit rule is restated from pygame ``surface.h``
(``ALPHA_BLEND_COMP(sC, dC, sA) = (((sC - dC) * sA + sC) >> 8) + dC``, pixels with ``sA == 0`` skipped) and the
nearest-neighbour walk of ``pygame.transform.scale`` (scaled clouds) from its ``transform.c``.  The PNG reader of the product is cross-checked against Pillow on the reference's
own assets when both are present (tests/test_render_cpu.py).
"""
import numpy as np

W, H = 432, 304
# state columns (pz_oracle.h)
P_X, P_Y, P_STATE, P_FRAME, P_DIVE = 0, 1, 3, 4, 7
P_WORDS = 13
B_X, B_Y, B_POWER, B_PX, B_PY, B_PPX, B_PPY, B_ROT = 26, 27, 30, 31, 32, 33, 34, 35
E_S1, E_S2 = 38, 39
SPRITE_PIKACHU, SPRITE_BALL, SPRITE_HYPER, SPRITE_TRAIL, SPRITE_SHADOW, SPRITE_NUMBER = 0, 28, 33, 34, 35, 36


def _blend(dst, src):
    s = src[..., :3].astype(np.int32)
    a = src[..., 3:4].astype(np.int32)
    d = dst.astype(np.int32)
    return np.where(a == 0, d, (((s - d) * a + s) >> 8) + d).astype(np.uint8)


def _blit(screen, sprite, x, y):
    """screen.blit(sprite, (x, y)) with clipping."""
    h, w = sprite.shape[:2]
    x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + w, W), min(y + h, H)
    if x0 >= x1 or y0 >= y1:
        return
    screen[y0:y1, x0:x1] = _blend(screen[y0:y1, x0:x1], sprite[y0 - y:y1 - y, x0 - x:x1 - x])


def _blit_center(screen, sprite, x, y):  # pikazoo_env.py:40-43
    _blit(screen, sprite, x - sprite.shape[1] // 2, y - sprite.shape[0] // 2)


def sprite_index(state, frame):  # get_frame_number_for_player_animated_sprite :46-68
    if state < 4:
        return 5 * state + frame
    if state == 4:
        return 17 + frame
    return 18 + 5 * (state - 5) + frame


# sprite ids of the product (include/pikazoo_hip.h: pz_sprite_id) and the reference's file behind each
SPRITE_CLOUD, SPRITE_WAVE, SPRITE_PUNCH, SPRITE_COUNT = 46, 47, 48, 49
_PIKACHU_FRAMES = [(0, 5), (1, 5), (2, 5), (3, 2), (4, 1), (5, 5), (6, 5)]  # get_all_image :445-474


def sprite_files():
    files = [f"pikachu_{st}_{fr}.png" for st, count in _PIKACHU_FRAMES for fr in range(count)]
    files += [f"ball_{k}.png" for k in range(5)] + ["ball_hyper.png", "ball_trail.png", "shadow.png"]
    files += [f"number_{k}.png" for k in range(10)] + ["cloud.png", "wave.png", "ball_punch.png"]
    assert len(files) == SPRITE_COUNT
    return files


# ---- clouds and waves (cloud_and_wave.py): state outside the 44 words, driven by the env RNG ----------------------
# scenery words: cloud i at 4i: top_left_point_x, top_left_point_y, top_left_point_x_velocity, size_diff_turn_number;
# 40: wave.vertical_coord, 41: wave.vertical_coord_velocity, 42..68: wave.y_coords; 69: ball.punch_effect_radius,
# 70: ball.punch_effect_y (ball attributes outside the 44 state words); 71..74: what scenery_track remembers of the
# previous frame (both collision flags, game_ended, round_ended)
SCENERY_WORDS = 75
P_COLL = 9
B_PUNCH_X = 37
E_ROUND_ENDED, E_GAME_ENDED = 41, 42


def scenery_init(draw):
    """get_all_image's ten Cloud(np_random) (:475-477, cloud_and_wave.py:15-19) + Wave() (:42-50); `draw(n)` is the
    env stream's integers(0, n)."""
    sc = np.zeros(SCENERY_WORDS, np.int32)
    for i in range(10):
        sc[4 * i] = -68 + draw(432 + 68)
        sc[4 * i + 1] = draw(152)
        sc[4 * i + 2] = 1 + draw(2)
        sc[4 * i + 3] = draw(11)
    sc[40], sc[41] = 0, 2
    sc[42:69] = 314
    return sc


def scenery_track(sc, col, auto_reset=True):
    """What one frame of the physics does to the punch effect's radius / y, re-derived from the state `col` it left:
    a frame that started a new round clears the radius (Ball.initialize_for_new_round physics.py:274-275), the
    ball-world step sets them on a ground touch (:427-430 -- the frames that end with round_ended set), a ball-player
    collision on a power hit (:628-632 -- the player's collision flag rose and its state is 2; player 2's after
    player 1's, physics_engine :319-335)."""
    col = [int(v) for v in col]
    if not (sc[73] and not auto_reset):  # a finished game without auto-reset is not stepped
        if sc[74]:
            sc[69] = 0
        if col[E_ROUND_ENDED]:
            sc[69], sc[70] = 20, 252 + 20
        for p, prev in ((0, 71), (1, 72)):
            if col[p * P_WORDS + P_COLL] and not sc[prev] and col[p * P_WORDS + P_STATE] == 2:
                sc[69], sc[70] = 20, col[B_Y]
    sc[71], sc[72] = col[P_COLL], col[P_WORDS + P_COLL]
    sc[73], sc[74] = col[E_GAME_ENDED], col[E_ROUND_ENDED]


def scenery_tick(sc, draw):
    """cloud_and_wave_engine (cloud_and_wave.py:53-78), in place."""
    for i in range(10):
        sc[4 * i] += sc[4 * i + 2]
        if sc[4 * i] > 432:
            sc[4 * i] = -68
            sc[4 * i + 1] = draw(152)
            sc[4 * i + 2] = 1 + draw(2)
        sc[4 * i + 3] = (sc[4 * i + 3] + 1) % 11
    sc[40] += sc[41]
    if sc[40] > 32:
        sc[40], sc[41] = 32, -1
    elif sc[40] < 0 and sc[41] < 0:
        sc[41] = 2
        sc[40] = -draw(40)
    for i in range(27):
        sc[42 + i] = 314 - sc[40] + draw(3)
    if sc[69] > 0:  # draw_ball counts the punch effect down itself (pikazoo_env.py:292-293)
        sc[69] -= 2


def draw_list(col, sizes, scenery=None, punch_drawn=None):
    """The blits of raw_env.draw behind the static background, in the reference's order (:250-255), as
    (sprite id, mirrored, x, y, width, height) with (x, y) the top-left corner; `sizes[id]` = (width, height) of the
    sprite files.  With `scenery` (after its tick): clouds, waves and the punch effect (:292-294: drawn whenever its
    radius was positive before the tick's decrement -- a 0 x 0 blit at the end)."""
    col = [int(v) for v in col]
    out = []
    if punch_drawn is None:  # (the caller knows whether the radius was positive before the tick; default: is it now)
        punch_drawn = scenery is not None and int(scenery[69]) > 0

    def centred(sid, x, y, flip=0):
        w, h = sizes[sid]
        out.append((sid, flip, x - w // 2, y - h // 2, w, h))

    if scenery is not None:  # draw_clouds_and_wave :338-353 (after the engine ran)
        for i in range(10):
            x, y, _, turn = (int(v) for v in scenery[4 * i:4 * i + 4])
            d = 5 - abs(turn - 5)
            out.append((SPRITE_CLOUD, 0, x - d, y - d, 48 + 2 * d, 24 + 2 * d))
        for i in range(27):
            out.append((SPRITE_WAVE, 0, 16 * i, int(scenery[42 + i]), *sizes[SPRITE_WAVE]))
    players = []
    for p in range(2):  # draw_player :257-275
        c0 = p * P_WORDS
        st, fr, dive = col[c0 + P_STATE], col[c0 + P_FRAME], col[c0 + P_DIVE]
        diving = st in (3, 4)
        flip = (diving and dive == -1) if p == 0 else not (diving and dive == 1)
        players.append((SPRITE_PIKACHU + sprite_index(st, fr), col[c0 + P_X], col[c0 + P_Y], int(flip)))
    for sid, x, y, flip in players:
        centred(sid, x, y, flip)
    centred(SPRITE_SHADOW, col[P_X], 273)
    centred(SPRITE_SHADOW, col[P_WORDS + P_X], 273)
    centred(SPRITE_BALL + col[B_ROT] // 10, col[B_X], col[B_Y])  # draw_ball :280-290
    centred(SPRITE_SHADOW, col[B_X], 273)
    if col[B_POWER]:
        centred(SPRITE_HYPER, col[B_PX], col[B_PY])
        centred(SPRITE_TRAIL, col[B_PPX], col[B_PPY])
    if scenery is not None and punch_drawn:
        r = int(scenery[69])
        out.append((SPRITE_PUNCH, 0, col[B_PUNCH_X] - r, int(scenery[70]) - r, 2 * r, 2 * r))
    s1, s2 = col[E_S1], col[E_S2]  # draw_scores_to_score_boards :327-336
    if s1 >= 10:
        out.append((SPRITE_NUMBER + 1, 0, 14, 10, *sizes[SPRITE_NUMBER + 1]))
    out.append((SPRITE_NUMBER + s1 % 10, 0, 14 + 32, 10, *sizes[SPRITE_NUMBER + s1 % 10]))
    if s2 >= 10:
        out.append((SPRITE_NUMBER + 1, 0, 432 - 32 - 32 - 14, 10, *sizes[SPRITE_NUMBER + 1]))
    out.append((SPRITE_NUMBER + s2 % 10, 0, 432 - 32 - 32 - 14 + 32, 10, *sizes[SPRITE_NUMBER + s2 % 10]))
    return out


def stretch_map(src, dst):
    """Source index of every destination index under pygame.transform.scale, as pygame's transform.c `stretch` walks
    it (an error-accumulating nearest-neighbour walk, restated from the library source; not checked against the
    library, which is not installed here): for each destination pixel copy the current source pixel, then advance
    the source while the error term is non-negative."""
    out = np.zeros(dst, np.int32)
    src2, dst2 = 2 * src, 2 * dst
    err, s = src2 - dst2, 0
    for d in range(dst):
        out[d] = min(s, src - 1)
        while err >= 0:
            s += 1
            err -= dst2
        err += src2
    return out


def scaled(sprite, w, h):
    hh, ww = sprite.shape[:2]
    if (ww, hh) == (w, h):
        return sprite
    if w <= 0 or h <= 0:
        return sprite[:0, :0]
    return sprite[stretch_map(hh, h)][:, stretch_map(ww, w)]


def frame(col, sprites, background, scenery=None):
    """uint8 [304, 432, 3] for one game: `col` = its 44 state words, `sprites` = list of the RGBA arrays (46, or 48 with
    cloud and wave when `scenery` = the game's clouds / waves is given)."""
    screen = background.copy()
    sizes = [(s.shape[1], s.shape[0]) for s in sprites]
    for sid, flip, x, y, w, h in draw_list(col, sizes, scenery):
        spr = scaled(sprites[sid], w, h)
        _blit(screen, spr[:, ::-1] if flip else spr, x, y)
    return screen
