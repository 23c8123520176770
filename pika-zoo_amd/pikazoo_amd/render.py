"""``render_mode="rgb_array"`` for the batched env: frames drawn on the GPU from the state tensor (``pz_render``).

The reference draws with pygame from its own PNG assets (``pikazoo/env/pikazoo_env.py:250-479``, ``pikazoo/env/img/``).
Those assets are not part of this repository: :func:`load_sprites` reads them at run time from a directory the user
points to (by default the ``img`` directory of an installed ``pikazoo`` package) with the small PNG reader below
(8-bit RGBA, the only format the assets use; zlib is in the standard library), packs them into one device atlas and
composes the static background once on the host, exactly in the order of ``draw_background`` (:296-325).
:func:`synthetic_sprites` builds a sprite set of the same geometry from a seed (tests, demos without the assets).

Drawn: background, both players with their mirroring rules, shadows, ball / hyper ball / trail, score boards; and,
for an env created with ``scenery=True``, the clouds and waves -- renderer-owned state outside the 44 words that, as in
the reference, is drawn from the env RNG (ten clouds at construction, a tick of ``cloud_and_wave_engine`` with every
``render()``), so rendering then changes the game's later random draws exactly like the reference's does -- and the punch
effect, whose two ball attributes (set by the physics on a ground touch / power hit, counted down by ``render()``) the env
tracks after every ``step()`` (``pz_scenery_track``).
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from . import _native

WIDTH, HEIGHT = 432, 304

PIKACHU_FRAMES = [(0, 5), (1, 5), (2, 5), (3, 2), (4, 1), (5, 5), (6, 5)]  # get_all_image :445-474
SPRITE_FILES = ([f"pikachu_{s}_{f}.png" for s, k in PIKACHU_FRAMES for f in range(k)]          # 0..27
                + [f"ball_{i}.png" for i in range(5)] + ["ball_hyper.png"]                    # 28..33
                + ["ball_trail.png", "shadow.png"]                                            # 34, 35
                + [f"number_{i}.png" for i in range(10)]                                      # 36..45
                + ["cloud.png", "wave.png", "ball_punch.png"])                                # 46, 47, 48
BACKGROUND_FILES = ["sky_blue.png", "mountain.png", "ground_red.png", "ground_line.png", "ground_line_leftmost.png",
                    "ground_line_rightmost.png", "ground_yellow.png", "net_pillar_top.png", "net_pillar.png"]
SPRITE_SHAPES = ([(64, 64)] * 28 + [(40, 40)] * 7 + [(32, 8)] + [(32, 32)] * 10 + [(48, 24), (16, 32), (40, 40)])  # (width, height)
SCENERY_WORDS = _native.SCENERY_WORDS
BACKGROUND_SHAPES = {"sky_blue.png": (16, 16), "mountain.png": (432, 64), "ground_red.png": (16, 16),
                     "ground_line.png": (16, 16), "ground_line_leftmost.png": (16, 16),
                     "ground_line_rightmost.png": (16, 16), "ground_yellow.png": (16, 16),
                     "net_pillar_top.png": (8, 8), "net_pillar.png": (8, 8)}
assert len(SPRITE_FILES) == len(SPRITE_SHAPES) == 49


class PzSprite(C.Structure):
    """`pz_sprite` (include/pikazoo_hip.h)."""

    _fields_ = [("offset", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


def read_png_rgba(path) -> np.ndarray:
    """uint8 [height, width, 4] of an 8-bit, non-interlaced RGB / RGBA PNG (PNG spec: filters 0-4)."""
    data = Path(path).read_bytes()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, header = 8, [], None
    while pos < len(data):
        length, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    width, height, depth, colour, _, _, interlace = header
    if depth != 8 or colour not in (2, 6) or interlace != 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced RGB/RGBA PNGs are supported")
    bpp = 4 if colour == 6 else 3
    raw = zlib.decompress(b"".join(idat))
    stride = width * bpp
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(height):
        ftype = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        else:  # 1 (sub), 3 (average), 4 (Paeth) depend on the pixel to the left: sequential
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                elif ftype == 4:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f"{path}: bad filter type {ftype}")
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(height, width, bpp)
    if bpp == 3:
        img = np.concatenate([img, np.full((height, width, 1), 255, np.uint8)], axis=2)
    return img


def blend_over(dst_rgb: np.ndarray, src_rgba: np.ndarray) -> np.ndarray:
    """pygame's per-pixel-alpha blit onto an opaque surface: dC = (((sC - dC) * sA + sC) >> 8) + dC (sA == 0: skip)."""
    s = src_rgba[..., :3].astype(np.int32)
    a = src_rgba[..., 3:4].astype(np.int32)
    d = dst_rgb.astype(np.int32)
    out = (((s - d) * a + s) >> 8) + d
    return np.where(a == 0, d, out).astype(np.uint8)


def background_blits():
    """draw_background (pikazoo_env.py:296-325) as the list of its blits (file, x, y), in the reference's order."""
    out = []
    for j in range(12):                       # sky :298-300
        for i in range(432 // 16):
            out.append(("sky_blue.png", 16 * i, 16 * j))
    out.append(("mountain.png", 0, 188))      # :303
    for i in range(432 // 16):                # ground_red :306-307
        out.append(("ground_red.png", 16 * i, 248))
    for i in range(1, 432 // 16 - 1):         # ground_line :310-313
        out.append(("ground_line.png", 16 * i, 264))
    out.append(("ground_line_leftmost.png", 0, 264))
    out.append(("ground_line_rightmost.png", 432 - 16, 264))
    for j in range(2):                        # ground_yellow :316-318
        for i in range(432 // 16):
            out.append(("ground_yellow.png", 16 * i, 280 + 16 * j))
    out.append(("net_pillar_top.png", 213, 176))  # :321-324
    for j in range(12):
        out.append(("net_pillar.png", 213, 184 + 8 * j))
    return out


def compose_background(tiles: Dict[str, np.ndarray]) -> np.ndarray:
    """uint8 [304, 432, 3]: draw_background onto a black screen."""
    screen = np.zeros((HEIGHT, WIDTH, 3), np.uint8)
    for name, x, y in background_blits():
        t = tiles[name]
        h, w = t.shape[:2]
        y1, x1 = min(y + h, HEIGHT), min(x + w, WIDTH)
        screen[y:y1, x:x1] = blend_over(screen[y:y1, x:x1], t[:y1 - y, :x1 - x])
    return screen


class SpriteSet:
    """The 49 dynamic sprites (RGBA8 atlas + descriptor table) and the composed background, on one device."""

    def __init__(self, sprites, tiles, device):
        if len(sprites) != len(SPRITE_FILES):
            raise ValueError(f"expected {len(SPRITE_FILES)} sprites")
        self.sprites_host = [np.ascontiguousarray(s, np.uint8) for s in sprites]
        self.background_host = compose_background(tiles)
        table = (PzSprite * len(sprites))()
        flat, off = [], 0
        for i, s in enumerate(self.sprites_host):
            h, w = s.shape[:2]
            table[i] = PzSprite(off, w, h)
            flat.append(s.reshape(-1, 4))
            off += h * w
        atlas = np.concatenate(flat).astype(np.uint32)
        packed = atlas[:, 0] | (atlas[:, 1] << 8) | (atlas[:, 2] << 16) | (atlas[:, 3] << 24)
        bg = self.background_host.astype(np.uint32)
        bg_packed = bg[..., 0] | (bg[..., 1] << 8) | (bg[..., 2] << 16) | np.uint32(0xFF000000)
        self.device = torch.device(device)
        self.atlas = torch.from_numpy(packed.view(np.int32).copy()).to(self.device)
        self.background = torch.from_numpy(bg_packed.view(np.int32).copy()).to(self.device)
        self.table = torch.from_numpy(np.frombuffer(bytes(table), np.int32).copy()).to(self.device)


def default_image_dir() -> Optional[Path]:
    """``img`` directory of an installed reference package (``pikazoo/env/img``), or None."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("pikazoo")
        if spec is not None and spec.origin:
            cand = Path(spec.origin).resolve().parent / "env" / "img"
            if cand.is_dir():
                return cand
    except Exception:  # noqa: BLE001
        pass
    return None


def load_sprites(image_dir, device) -> SpriteSet:
    """Read the reference's PNG assets from `image_dir` (they are never copied into this package)."""
    d = Path(image_dir)
    sprites = [read_png_rgba(d / f) for f in SPRITE_FILES]
    tiles = {f: read_png_rgba(d / f) for f in BACKGROUND_FILES}
    return SpriteSet(sprites, tiles, device)


def synthetic_sprites(seed: int, device) -> SpriteSet:
    """A sprite set of the reference's geometry with seeded random pixels: opaque cores, soft (partially
    transparent) rims and fully transparent corners, so every branch of the blit is exercised."""
    rng = np.random.default_rng(seed)

    def make(w, h):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        r = np.hypot((xx - (w - 1) / 2) / (w / 2), (yy - (h - 1) / 2) / (h / 2))
        img[..., 3] = np.where(r < 0.7, 255, np.where(r < 1.0, img[..., 3], 0))
        return img

    sprites = [make(w, h) for (w, h) in SPRITE_SHAPES]
    tiles = {f: make(*BACKGROUND_SHAPES[f]) for f in BACKGROUND_FILES}
    for f in ("sky_blue.png", "ground_red.png", "ground_yellow.png", "mountain.png"):
        tiles[f][..., 3] = 255
    return SpriteSet(sprites, tiles, device)


def render(lib, state_ptr: int, device, n: int, stride: int, sprite_set: SpriteSet, lanes: Optional[torch.Tensor],
           stream: int, out: Optional[torch.Tensor] = None, scenery: Optional[torch.Tensor] = None, cfg_ref=None) -> torch.Tensor:
    """uint8 ``[m, 304, 432, 3]`` frames of the games `lanes` (None: all n) through ``pz_render``; `state_ptr` = the
    ``int32[44, stride]`` columns on `device`.  With `scenery` (``int32[69, stride]``) the clouds / waves of the drawn
    games advance first (and with them the games' env RNG counters in the state) and are drawn."""
    m = n if lanes is None else int(lanes.numel())
    if out is None:
        out = torch.empty((m, HEIGHT, WIDTH, 3), dtype=torch.uint8, device=device)
    else:
        # the kernel writes m * 304 * 432 * 3 bytes as dwords straight through this pointer: anything but exactly that
        # buffer on this device would be overrun silently
        want = (m, HEIGHT, WIDTH, 3)
        if not isinstance(out, torch.Tensor) or tuple(out.shape) != want or out.dtype != torch.uint8 \
                or out.device != torch.device(device) or not out.is_contiguous():
            raise ValueError(f"render(out=): need a contiguous uint8 tensor of shape {want} on {device}, got "
                             f"{getattr(out, 'dtype', type(out))} {tuple(getattr(out, 'shape', ()))} on "
                             f"{getattr(out, 'device', None)}")
    _native.check(lib.pz_render(state_ptr, n, stride, cfg_ref, None if lanes is None else lanes.data_ptr(), m,
                                sprite_set.atlas.data_ptr(), sprite_set.table.data_ptr(),
                                sprite_set.background.data_ptr(), None if scenery is None else scenery.data_ptr(),
                                out.data_ptr(), stream), "pz_render")
    return out
