"""The vertical scheme of the walking pyramid kernels (flvis_amd/csrc/pyr_walk.hip), restated in numpy and checked against the checker's
cv::pyrDown restatement on CPU: two running sums per output column, a source row 2m adding 6x to output row m and 1x to rows m-1 / m+1, a
row 2m+1 adding 4x to rows m and m+1, and BORDER_REFLECT_101 at the top / bottom expressed as changed weights only (row 1 counts 8x for
output row 0, row 2 twice; the last rows likewise, differently for even and odd heights).  The kernel's band logic (which rows a band
loads, which it stores) is modelled too: every band must reproduce exactly its own rows of every level."""
import numpy as np
import pytest

import _oracle as O
import _synth as S


def _hsum(row):
    """horizontal [1 4 6 4 1] at the even columns with REFLECT_101 (the kernel takes the two / one outside pixels from the lane's own bytes)"""
    w = row.shape[0]
    p = np.pad(row.astype(np.int64), (2, 2), mode="reflect")
    c = np.arange(0, w, 2) + 2
    return p[c - 2] + 4 * p[c - 1] + 6 * p[c] + 4 * p[c + 1] + p[c + 2]


class _Level:
    """rows of one level arrive in increasing order (the first one even); emits the rows of the next level inside [lo, hi]"""

    def __init__(self, H, lo, hi, sink):
        self.H, self.lo, self.hi, self.sink = H, lo, hi, sink
        self.cur = self.nxt = 0

    def push(self, r, row):
        h, H = _hsum(row), self.H
        if r % 2 == 0:
            we, wc, wn = (2 if r == 2 else 1), (7 if r == H - 2 else 6), (2 if r == H - 3 else 1)
            ev = self.cur + we * h
            self.cur = self.nxt + wc * h
            self.nxt = wn * h
            ya, yb = r // 2 - 1, r // 2
        else:
            ev = None
            self.cur = self.cur + (8 if r == 1 else 4) * h
            self.nxt = self.nxt + (8 if r == H - 2 else 4) * h
            ya, yb = -1, (r - 1) // 2
        if ev is not None and self.lo <= ya <= self.hi:
            self.sink(ya, ((ev + 128) >> 8).astype(np.uint8))
        if r == H - 1 and self.lo <= yb <= self.hi:
            self.sink(yb, ((self.cur + 128) >> 8).astype(np.uint8))


def _walk(img, nout, rows_per_band):
    """levels 1 .. nout of img as the bands of one launch produce them"""
    Hs = [img.shape[0]]
    Ws = [img.shape[1]]
    for _ in range(nout):
        Hs.append((Hs[-1] + 1) // 2)
        Ws.append((Ws[-1] + 1) // 2)
    out = [None] + [np.full((Hs[j], Ws[j]), -1, np.int32) for j in range(1, nout + 1)]
    for y0 in range(0, Hs[nout], rows_per_band):
        y1 = min(y0 + rows_per_band, Hs[nout])
        need = {nout: (y0, y1 - 1)}
        own = {nout: (y0, y1)}
        for j in range(nout - 1, -1, -1):
            need[j] = (max(0, 2 * need[j + 1][0] - 2), min(Hs[j] - 1, 2 * need[j + 1][1] + 2))
            own[j] = (2 * own[j + 1][0], min(Hs[j], 2 * own[j + 1][1]))
        levels = {}

        def make_sink(j):
            def sink(y, row):
                if own[j][0] <= y < own[j][1]:
                    assert out[j][y, 0] == -1, "a row stored twice"
                    out[j][y] = row
                if j < nout:
                    levels[j].push(y, row)
            return sink

        for j in range(nout - 1, -1, -1):
            levels[j] = _Level(Hs[j], need[j + 1][0], need[j + 1][1], make_sink(j + 1))
        for r in range(need[0][0], need[0][1] + 1):
            levels[0].push(r, img[r])
    return out[1:]


_CASES = [(h, w, nout, band) for (h, w) in [(480, 640), (97, 144), (64, 64), (33, 80), (35, 96), (16, 64), (61, 128)]
          for (nout, band) in [(1, 4), (1, 8), (2, 2), (2, 1), (3, 2)] if h >= (8 << nout)]  # (pyr_walk_ok: >= 8 rows in every filtered level)


@pytest.mark.parametrize("h,w,nout,band", _CASES)
def test_running_sum_scheme_equals_pyr_down(h, w, nout, band):
    img = S.texture_u8(h, w, 5)
    got = _walk(img, nout, band)
    want = img
    for j in range(nout):
        want = O.pyr_down(want)
        assert np.array_equal(got[j], want.astype(np.int32)), (j, np.argwhere(got[j] != want)[:4])
