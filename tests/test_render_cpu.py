"""CPU tests of the renderer's host side: the PNG reader, the background composition and the numpy frame oracle
(no GPU needed).  The frame itself (HIP vs oracle) is tested in tests/test_gpu_render.py."""
import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

REFERENCE_IMG = Path("/root/reference/pikazoo/env/img")  # only in the build container; never on the GPU box


def _write_png(path, img, filter_type):
    """A minimal 8-bit RGBA PNG writer using ONE filter type for every scanline (test fixture generator)."""
    h, w, _ = img.shape
    bpp, stride = 4, w * 4
    raw = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if filter_type == 0:
            pred = np.zeros(stride, np.int32)
        elif filter_type == 1:
            pred = left
        elif filter_type == 2:
            pred = prev
        elif filter_type == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(filter_type)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    data = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    Path(path).write_bytes(data)


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
def test_png_reader_round_trip_every_filter(filter_type, tmp_path):
    from pikazoo_amd.render import read_png_rgba

    rng = np.random.default_rng(filter_type)
    img = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    _write_png(tmp_path / "t.png", img, filter_type)
    assert np.array_equal(read_png_rgba(tmp_path / "t.png"), img)


def test_png_reader_matches_pillow_on_the_reference_assets():
    """The product reads the reference's own PNG files at run time; in the build container (reference checkout and
    Pillow present) every asset it uses must decode to what Pillow decodes."""
    PIL = pytest.importorskip("PIL.Image")
    if not REFERENCE_IMG.is_dir():
        pytest.skip("reference assets not present (GPU box)")
    from pikazoo_amd.render import BACKGROUND_FILES, BACKGROUND_SHAPES, SPRITE_FILES, SPRITE_SHAPES, read_png_rgba

    for name, shape in list(zip(SPRITE_FILES, SPRITE_SHAPES)) + [(f, BACKGROUND_SHAPES[f]) for f in BACKGROUND_FILES]:
        mine = read_png_rgba(REFERENCE_IMG / name)
        ref = np.asarray(PIL.open(REFERENCE_IMG / name).convert("RGBA"))
        assert mine.shape == (shape[1], shape[0], 4), name
        assert np.array_equal(mine, ref), name


def test_blend_rule_and_background_composition():
    from pikazoo_amd import render as R

    # pygame's rule at its ends: alpha 255 copies the source, alpha 0 keeps the destination
    dst = np.array([[[10, 200, 77]]], np.uint8)
    for a, want in ((255, [250, 3, 128]), (0, [10, 200, 77])):
        src = np.array([[[250, 3, 128, a]]], np.uint8)
        assert R.blend_over(dst, src).tolist() == [[want]]
    mid = R.blend_over(dst, np.array([[[250, 3, 128, 128]]], np.uint8))[0, 0].tolist()
    assert mid == [(((250 - 10) * 128 + 250) >> 8) + 10, (((3 - 200) * 128 + 3) >> 8) + 200, (((128 - 77) * 128 + 128) >> 8) + 77]
    # draw_background tiles the whole screen with opaque tiles: no black pixel of the initial surface is left above y = 312
    rng = np.random.default_rng(1)
    tiles = {f: np.concatenate([rng.integers(1, 256, (h, w, 3), dtype=np.uint8), np.full((h, w, 1), 255, np.uint8)], 2)
             for f, (w, h) in R.BACKGROUND_SHAPES.items()}
    bg = R.compose_background(tiles)
    assert bg.shape == (304, 432, 3)
    assert np.array_equal(bg[0:16, 0:16], tiles["sky_blue.png"][..., :3])
    assert np.array_equal(bg[188:192, 0:213], tiles["mountain.png"][0:4, 0:213, :3])
    assert np.array_equal(bg[176:184, 213:221], tiles["net_pillar_top.png"][..., :3])
    assert np.array_equal(bg[296:304, 416:432], tiles["ground_yellow.png"][0:8, :, :3])
    assert (bg.reshape(-1, 3).max(axis=1) > 0).all()


def test_frame_oracle_on_a_reference_state():
    """The numpy frame oracle on states of a captured reference trajectory: sprites land where raw_env.draw puts
    them (centre-anchored players / ball, shadows at y = 273, score digits at their fixed boxes)."""
    from conftest import golden_state, load_golden
    from oracle import render_oracle as ro

    d = load_golden("both_computer")
    sprites = []
    for i, (w, h) in enumerate([(64, 64)] * 28 + [(40, 40)] * 7 + [(32, 8)] + [(32, 32)] * 10):
        s = np.zeros((h, w, 4), np.uint8)
        s[..., 0], s[..., 1], s[..., 3] = 10 + i, 200, 255  # opaque, colour-coded by sprite id
        sprites.append(s)
    bg = np.zeros((304, 432, 3), np.uint8)
    st = golden_state(d, 300)
    for lane in range(st.shape[1]):
        col = st[:, lane]
        f = ro.frame(col, sprites, bg)
        assert f.shape == (304, 432, 3) and f.dtype == np.uint8
        # ball sprite: colour-coded id at the ball's centre (the ball is drawn after players and shadows)
        bx, by = int(col[26]), int(col[27])
        under_digits = by < 42 and (14 <= bx < 78 or 354 <= bx < 418)  # the score boards are drawn last
        under_trail = col[30] and ((abs(int(col[31]) - bx) <= 20 and abs(int(col[32]) - by) <= 20)
                                   or (abs(int(col[33]) - bx) <= 20 and abs(int(col[34]) - by) <= 20))
        if 0 <= by < 304 and 0 <= bx < 432 and not under_digits and not under_trail:
            assert f[by, bx, 0] == 10 + 28 + int(col[35]) // 10
        # score digits (drawn last)
        assert f[10 + 5, 46 + 5, 0] == 10 + 36 + int(col[38]) % 10
        assert f[10 + 5, 386 + 5, 0] == 10 + 36 + int(col[39]) % 10
        # player 1's shadow row is untouched at the far right end of the court
        assert f[273, 431].tolist() in ([0, 0, 0], [10 + 35, 200, 0])
