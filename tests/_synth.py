"""Small deterministic test images (numpy only)."""
import numpy as np


def value_noise(h, w, seed, octaves=4, base=8):
    """Band-limited texture: sum of bilinearly upsampled random grids."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    amp = 1.0
    tot = 0.0
    cell = float(max(h, w)) / base
    for _ in range(octaves):
        gh, gw = int(np.ceil(h / cell)) + 2, int(np.ceil(w / cell)) + 2
        g = rng.random((gh, gw))
        ys = np.arange(h) / cell
        xs = np.arange(w) / cell
        y0 = np.floor(ys).astype(int)
        x0 = np.floor(xs).astype(int)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        tot += amp
        amp *= 0.6
        cell /= 2.2
    img /= tot
    return img


def texture_u8(h, w, seed):
    t = value_noise(h, w, seed, octaves=5, base=6)
    t = (t - t.min()) / (t.max() - t.min())
    return np.clip(np.round(t * 255), 0, 255).astype(np.uint8)


def shifted_pair(h, w, seed, dx, dy, margin=64):
    """Two crops of one big texture related by a pure (sub-pixel) translation: I1(x) = I0(x - d)."""
    big = value_noise(h + 2 * margin, w + 2 * margin, seed, octaves=5, base=6)
    big = (big - big.min()) / (big.max() - big.min()) * 255.0

    def sample(ox, oy):
        ys = np.arange(h) + margin + oy
        xs = np.arange(w) + margin + ox
        y0 = np.floor(ys).astype(int)
        x0 = np.floor(xs).astype(int)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = big[y0][:, x0]
        b = big[y0][:, x0 + 1]
        c = big[y0 + 1][:, x0]
        d = big[y0 + 1][:, x0 + 1]
        return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy

    i0 = np.ascontiguousarray(np.clip(np.round(sample(0.0, 0.0)), 0, 255).astype(np.uint8))
    i1 = np.ascontiguousarray(np.clip(np.round(sample(-dx, -dy)), 0, 255).astype(np.uint8))
    return i0, i1


def corner_img(h, w, seed):
    """texture + random bright/dark rectangles: plenty of FAST corners at every pyramid level."""
    rng = np.random.default_rng(seed)
    img = texture_u8(h, w, seed).astype(np.int32)
    for _ in range(60):
        x, y = rng.integers(0, w - 8), rng.integers(0, h - 8)
        ww, hh = rng.integers(6, 60), rng.integers(6, 60)
        img[y:y + hh, x:x + ww] = np.clip(img[y:y + hh, x:x + ww] + rng.integers(-120, 120), 0, 255)
    return img.astype(np.uint8)
