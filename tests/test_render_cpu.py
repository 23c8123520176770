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


# ------------------------------------------------------------------------------------------------
# the reference's own render(), recorded (oracle/ref_capture.capture_render): draw lists, clouds / waves, env RNG
# ------------------------------------------------------------------------------------------------
RENDER_FIXTURES = ["render_human_human", "render_p2_computer"]


def replay_render_fixture(d, on_frame):
    """Step the C oracle through a render fixture -- construction, clouds, reset, the random policy, a scenery tick
    wherever the reference rendered -- checking states / clouds / waves against the reference at every point, and call
    `on_frame(k, lane, state_column, scenery, punch_drawn)` for every recorded frame k."""
    from oracle import pz_oracle as po
    from oracle import render_oracle as ro

    po.build()
    meta = d["meta"]
    lanes, steps, periods = meta["lanes"], meta["steps"], meta["periods"]
    seed, base = meta["seed"], meta["env_id_base"]
    kw = meta["env_kwargs"]
    ref = po.OracleEnv(lanes, po.make_config(winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
                                             is_player2_computer=kw.get("is_player2_computer", False), seed=seed,
                                             env_id_base=base, auto_reset=True))

    def stream(i):  # the env stream of lane i, continuing from the lane's draw counter
        def draw(n):
            v = po.env_draw(seed, base + i, int(ref.state[43, i]) & 0xFFFFFFFF, n)
            ref.state[43, i] += 1
            return v
        return draw

    # constructor: physics (draws 0, 1), then the ten clouds (get_all_image, pikazoo_env.py:475-477)
    scenery = [ro.scenery_init(stream(i)) for i in range(lanes)]
    assert np.array_equal(ref.state, d["state_ctor"])
    assert np.array_equal(np.stack(scenery, axis=1)[:71], d["scenery_ctor"])
    ref.reset()
    assert np.array_equal(ref.state, d["state0"])
    k = 0

    def frame(i, t):
        nonlocal k
        assert (int(d["frame_lane"][k]), int(d["frame_step"][k])) == (i, t)
        punch_drawn = scenery[i][69] > 0     # draw_ball tests the radius before it counts it down
        ro.scenery_tick(scenery[i], stream(i))
        assert np.array_equal(scenery[i][:71], d["frame_scenery"][k]), (k, i, t)   # clouds, waves, punch radius / y
        assert np.array_equal(ref.state[:, i], d["frame_state"][k]), (k, i, t)
        on_frame(k, i, ref.state[:, i].copy(), scenery[i].copy(), punch_drawn)
        k += 1

    for i in range(lanes):
        frame(i, -1)
    for t in range(steps):
        a1, a2 = po.random_actions(lanes, base, meta["action_seed"], t, 18)
        ref.step(a1, a2)
        assert np.array_equal(ref.state, d["states"][t]), t
        for i in range(lanes):
            ro.scenery_track(scenery[i], ref.state[:, i])   # what the frame did to the punch effect
            if (t + 1) % periods[i] == 0:
                frame(i, t)
    assert k == len(d["frame_lane"])


@pytest.mark.parametrize("name", RENDER_FIXTURES)
def test_draw_list_clouds_waves_and_rng_match_the_reference_render(name):
    """The reference's render() run unmodified on a recording stand-in for pygame: every blit it issued (file,
    mirrored, position, size), its clouds and waves, and what rendering does to the env RNG -- against the oracle."""
    from conftest import load_golden
    from oracle import render_oracle as ro

    d = load_golden(name)
    files = d["meta"]["files"]
    ours = ro.sprite_files()
    fid = {f: ours.index(f) for f in files if f in ours}
    # sprite sizes as the reference's surfaces report them
    sizes = {}
    for row in d["frame_draw"].reshape(-1, 6):
        if row[0] >= 0 and files[row[0]] in fid and files[row[0]] not in ("cloud.png", "ball_punch.png"):
            sizes.setdefault(fid[files[row[0]]], (int(row[4]), int(row[5])))
    sizes[ro.SPRITE_CLOUD], sizes[ro.SPRITE_PUNCH] = (48, 24), (40, 40)
    size_list = [sizes.get(i, (0, 0)) for i in range(ro.SPRITE_COUNT)]
    seen = set()

    def on_frame(k, lane, col, scenery, punch_drawn):
        want = [tuple(int(v) for v in r) for r in d["frame_draw"][k][:d["frame_count"][k]]]
        want = [(fid[files[r[0]]],) + r[1:] for r in want]
        got = ro.draw_list(col, size_list, scenery, punch_drawn)
        assert got == want, (k, lane)
        seen.update(r[0] for r in got)

    replay_render_fixture(d, on_frame)
    assert len(seen) >= 30  # players in most poses, all balls, hyper ball / trail, all digits, cloud, wave
    # the static part of every frame: the product composes the same blits (file, position, size) in the same order
    from pikazoo_amd import render as R

    want = [(files[r[0]], int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in d["background"]]
    got = [(f, x, y, *R.BACKGROUND_SHAPES[f]) for f, x, y in R.background_blits()]
    assert got == want


def test_stretch_map_is_a_monotone_cover():
    from oracle import render_oracle as ro

    for src, dst in [(48, 48), (48, 50), (48, 58), (24, 34), (40, 36), (40, 2), (7, 19)]:
        m = ro.stretch_map(src, dst)
        assert m[0] == 0 and (np.diff(m) >= 0).all() and m.max() <= src - 1
        if dst >= src:
            assert set(m) == set(range(src))  # enlarging drops no source pixel
