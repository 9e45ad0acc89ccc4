"""Seeded synthetic frames for the 4K / 8K configurations (BASELINE.json configs 4 and 5).

Integer arithmetic only, so the bytes are identical on every host and numpy build.
Content is *blocky* on the 8x8 grid -- a smooth low-frequency base plus one DC offset per 8x8
block plus +-1 noise -- because uniform noise never passes the filter's local-activity test
 and would measure a memcpy instead of the filter.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix(h):
    """32-bit avalanche on a uint64 array that holds 32-bit values."""
    h = (h ^ (h >> np.uint64(16))) & _M32
    h = (h * np.uint64(0x7FEB352D)) & _M32
    h = (h ^ (h >> np.uint64(15))) & _M32
    h = (h * np.uint64(0x846CA68B)) & _M32
    h = (h ^ (h >> np.uint64(16))) & _M32
    return h


def _tri(t, period):
    """Integer triangle wave in [-256, 256] with the given period."""
    half = period // 2
    p = t % period
    up = (p * 512) // half - 256
    down = 256 - ((p - half) * 512) // half
    return np.where(p < half, up, down)


def blocky_plane(width, height, seed=1, frame=0, bit_depth=8, dc_range=6):
    """One plane (height x width), uint8 for 8-bit, uint16 (values < 2**bit_depth) otherwise."""
    x = np.arange(width, dtype=np.int64)[None, :]
    y = np.arange(height, dtype=np.int64)[:, None]
    base = 128 + (60 * _tri(x + 37 * frame, 611) * _tri(y + 11 * frame, 383)) // 65536 + ((x - y) * 5) // 256
    bxi = (x // 8).astype(np.uint64)
    byi = (y // 8).astype(np.uint64)
    key = np.uint64((seed * 0x9E3779B1 + frame * 0x85EBCA77) & 0xFFFFFFFF)
    hb = _mix((((bxi * np.uint64(73856093)) & _M32) ^ ((byi * np.uint64(19349663)) & _M32)) ^ key)
    dc = (hb % np.uint64(2 * dc_range + 1)).astype(np.int64) - dc_range
    hp = _mix((((x.astype(np.uint64) * np.uint64(0x27D4EB2F)) & _M32) ^ ((y.astype(np.uint64) * np.uint64(0x165667B1)) & _M32))
               ^ key ^ np.uint64(0x5BD1E995))
    # per-block texture class: 8/16 flat, 7/16 +-1 noise, 1/16 +-6 texture.  Gives a decision mix
    # close to the bundled images' (7-34 % strong, 62-87 % normal, 4-7 % off).
    cls = (hb >> np.uint64(8)) % np.uint64(16)
    amp = np.where(cls < np.uint64(8), 0, np.where(cls < np.uint64(15), 1, 6)).astype(np.int64)
    noise = ((hp % np.uint64(13)).astype(np.int64) * (2 * amp + 1)) // 13 - amp
    v = np.clip(base + dc + noise, 0, 255)
    if bit_depth == 8:
        return v.astype(np.uint8)
    # 10-bit and up: same picture scaled to the sample range, plus sub-LSB noise of the 8-bit scale
    sh = bit_depth - 8
    fine = ((hp >> np.uint64(8)) % np.uint64(1 << sh)).astype(np.int64)
    return np.clip((v << sh) + fine, 0, (1 << bit_depth) - 1).astype(np.uint16)


def blocky_yuv420(width, height, seed=1, frame=0, bit_depth=8):
    """(Y, U, V) planes of a 4:2:0 frame."""
    yv = blocky_plane(width, height, seed, frame, bit_depth)
    u = blocky_plane(width // 2, height // 2, seed + 101, frame, bit_depth, dc_range=4)
    v = blocky_plane(width // 2, height // 2, seed + 202, frame, bit_depth, dc_range=4)
    return yv, u, v


def ctu_qp_map(width, height, seed=1, lo=22, hi=42, ctu_log2=6):
    """Seeded per-CTU QP map in [lo, hi] (config 3b)."""
    n = 1 << ctu_log2
    cw, ch = (width + n - 1) // n, (height + n - 1) // n
    idx = np.arange(cw * ch, dtype=np.uint64).reshape(ch, cw)
    h = _mix((idx * np.uint64(0x9E3779B1)) & _M32 ^ np.uint64(seed & 0xFFFFFFFF))
    return (lo + (h % np.uint64(hi - lo + 1))).astype(np.uint8)
