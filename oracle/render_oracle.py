"""CPU restatement of the reference's ``rgb_array`` frame (TEST INFRASTRUCTURE, not product).

``frame(state_column, sprites, background)`` draws one game the way ``raw_env.draw`` does
(pikazoo/env/pikazoo_env.py:250-336) minus what is not a function of the game state (clouds and waves :338-353,
punch effect :292-294), with numpy slices instead of pygame blits.

Parity note: pygame is not installed in the build container, so this oracle is **not pinned** against the
reference's renderer; it restates the reference's draw order / coordinates from the source and pygame's published
per-pixel-alpha blit rule (pygame ``surface.h``: ``ALPHA_BLEND_COMP(sC, dC, sA) = (((sC - dC) * sA + sC) >> 8) + dC``,
pixels with ``sA == 0`` skipped).  The PNG reader of the product is cross-checked against Pillow on the reference's
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


def frame(col, sprites, background):
    """uint8 [304, 432, 3] for one game: `col` = its 44 state words, `sprites` = list of 46 RGBA arrays."""
    screen = background.copy()
    col = [int(v) for v in col]
    for p in range(2):  # draw_player :257-275
        c0 = p * P_WORDS
        st, fr, dive = col[c0 + P_STATE], col[c0 + P_FRAME], col[c0 + P_DIVE]
        spr = sprites[SPRITE_PIKACHU + sprite_index(st, fr)]
        diving = st in (3, 4)
        flip = (diving and dive == -1) if p == 0 else not (diving and dive == 1)
        _blit_center(screen, spr[:, ::-1] if flip else spr, col[c0 + P_X], col[c0 + P_Y])
    _blit_center(screen, sprites[SPRITE_SHADOW], col[P_X], 273)
    _blit_center(screen, sprites[SPRITE_SHADOW], col[P_WORDS + P_X], 273)
    _blit_center(screen, sprites[SPRITE_BALL + col[B_ROT] // 10], col[B_X], col[B_Y])  # draw_ball :280-290
    _blit_center(screen, sprites[SPRITE_SHADOW], col[B_X], 273)
    if col[B_POWER]:
        _blit_center(screen, sprites[SPRITE_HYPER], col[B_PX], col[B_PY])
        _blit_center(screen, sprites[SPRITE_TRAIL], col[B_PPX], col[B_PPY])
    s1, s2 = col[E_S1], col[E_S2]  # draw_scores_to_score_boards :327-336
    if s1 >= 10:
        _blit(screen, sprites[SPRITE_NUMBER + 1], 14, 10)
    _blit(screen, sprites[SPRITE_NUMBER + s1 % 10], 14 + 32, 10)
    if s2 >= 10:
        _blit(screen, sprites[SPRITE_NUMBER + 1], 432 - 32 - 32 - 14, 10)
    _blit(screen, sprites[SPRITE_NUMBER + s2 % 10], 432 - 32 - 32 - 14 + 32, 10)
    return screen
