"""TEST INFRASTRUCTURE ONLY — CPU restatement of cv2.inpaint(img, mask, r, cv2.INPAINT_TELEA) for
8-bit 3-channel images, the tail of the contour remover (1_lama_contour_remover/predict.py:61-64).

The algorithm lives in a third-party dependency that is absent from /root/reference and from this
image (opencv-python, unpinned in the reference's requirements): Telea, "An image inpainting
technique based on the fast marching method" (J. Graphics Tools 9(1), 2004) as OpenCV implements it
(modules/photo/src/inpaint.cpp: cvInpaint -> icvCalcFMM -> icvTeleaInpaintFMM).  It is restated
here from the published source as remembered — **parity unpinned**: there is no OpenCV here to
generate golden vectors with, and the reference's tests hold none for this call.  What the tests
can and do check: this pure-Python restatement (heapq with a push counter = OpenCV's FIFO-stable
ordered queue; numpy float32 scalars wherever OpenCV computes in `float`) against the native
implementation in libdsu_hip.so bit for bit, plus the properties any correct inpainting has
(known pixels untouched, constant images stay constant, results inside the convex hull of the
known neighbourhood).

Pure-Python loops: use on small images only (tests: <= 40 x 40)."""
import heapq
import math

import numpy as np

KNOWN, BAND, INSIDE, CHANGE = 0, 1, 2, 3
F32 = np.float32
_NB = ((-1, 0), (0, -1), (1, 0), (0, 1))           # up, left, down, right


class _Queue:
    def __init__(self):
        self.h, self.n = [], 0

    def push(self, i, j, T):
        heapq.heappush(self.h, (float(T), self.n, i, j))
        self.n += 1

    def pop(self):
        if not self.h:
            return None
        _, _, i, j = heapq.heappop(self.h)
        return i, j


def _solve(i1, j1, i2, j2, f, t):
    a11, a22 = float(t[i1, j1]), float(t[i2, j2])
    m12 = min(a11, a22)
    if f[i1, j1] != INSIDE:
        if f[i2, j2] != INSIDE:
            if abs(a11 - a22) >= 1.0:
                sol = 1 + m12
            else:
                sol = (a11 + a22 + math.sqrt(2 - (a11 - a22) * (a11 - a22))) * 0.5
        else:
            sol = 1 + a11
    elif f[i2, j2] != INSIDE:
        sol = 1 + a22
    else:
        sol = 1 + m12
    return F32(sol)


def _dist(i, j, f, t):
    return min(_solve(i - 1, j, i, j - 1, f, t), _solve(i + 1, j, i, j - 1, f, t),
               _solve(i - 1, j, i, j + 1, f, t), _solve(i + 1, j, i, j + 1, f, t))


def _calc_fmm_negated(f, t, heap):
    er, ec = f.shape
    while True:
        p = heap.pop()
        if p is None:
            break
        ii, jj = p
        f[ii, jj] = CHANGE
        for di, dj in _NB:
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= er - 1 or j >= ec - 1:
                continue
            if f[i, j] == INSIDE:
                d = _dist(i, j, f, t)
                t[i, j] = d
                f[i, j] = BAND
                heap.push(i, j, d)
    ch = f == CHANGE
    f[ch] = KNOWN
    t[ch] = -t[ch]


def _sat_u8(v):
    r = np.rint(F32(v))                  # cvRound: nearest, ties to even
    return np.uint8(min(max(float(r), 0.0), 255.0))


def inpaint_telea(img, mask, radius):
    """img (H,W,3) uint8, mask (H,W) uint8 (non-zero = fill), radius int -> (H,W,3) uint8."""
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = mask.shape
    assert img.shape == (rows, cols, 3) and rows >= 3 and cols >= 3
    rng = min(max(int(radius), 1), 100)
    er, ec = rows + 2, cols + 2
    out = img.copy()
    m = np.zeros((er, ec), np.uint8)
    m[1:-1, 1:-1][mask != 0] = INSIDE
    t = np.full((er, ec), 1.0e6, F32)
    hole = m == INSIDE
    nb = np.zeros_like(hole)
    nb[1:-1, 1:-1] = hole[:-2, 1:-1] | hole[2:, 1:-1] | hole[1:-1, :-2] | hole[1:-1, 2:]
    band = nb & ~hole
    t[band] = 0.0
    heap = _Queue()
    for i, j in zip(*np.nonzero(band)):                 # row-major
        heap.push(int(i), int(j), 0.0)
    # outside distances over the square-dilated neighbourhood of the hole
    dil = np.zeros_like(hole)
    ys, xs = np.nonzero(hole)
    for y, x in zip(ys, xs):
        dil[max(y - rng, 0):y + rng + 1, max(x - rng, 0):x + rng + 1] = True
    o = np.where(dil & ~hole & ~band, INSIDE, KNOWN).astype(np.uint8)
    o[0, :] = o[-1, :] = KNOWN
    o[:, 0] = o[:, -1] = KNOWN
    outq = _Queue()
    for i, j in zip(*np.nonzero(band)):
        outq.push(int(i), int(j), 0.0)
    _calc_fmm_negated(o, t, outq)
    # Telea march into the hole; flags = the hole mask (band pixels are KNOWN in it)
    f = m
    while True:
        p = heap.pop()
        if p is None:
            break
        ii, jj = p
        f[ii, jj] = KNOWN
        for di, dj in _NB:
            i, j = ii + di, jj + dj
            if i <= 1 or j <= 1 or i >= er - 1 or j >= ec - 1:
                continue
            if f[i, j] != INSIDE:
                continue
            dist = _dist(i, j, f, t)
            t[i, j] = dist
            if f[i, j + 1] != INSIDE:
                gtx = F32(t[i, j + 1] - t[i, j - 1]) * F32(0.5) if f[i, j - 1] != INSIDE \
                    else F32(t[i, j + 1] - t[i, j])
            else:
                gtx = F32(t[i, j] - t[i, j - 1]) if f[i, j - 1] != INSIDE else F32(0)
            if f[i + 1, j] != INSIDE:
                gty = F32(t[i + 1, j] - t[i - 1, j]) * F32(0.5) if f[i - 1, j] != INSIDE \
                    else F32(t[i + 1, j] - t[i, j])
            else:
                gty = F32(t[i, j] - t[i - 1, j]) if f[i - 1, j] != INSIDE else F32(0)
            for c in range(3):
                Ia, Jx, Jy, s = F32(0), F32(0), F32(0), F32(1.0e-20)
                for k in range(i - rng, i + rng + 1):
                    km = k - 1 + (1 if k == 1 else 0)
                    kp = k - 1 - (1 if k == er - 2 else 0)
                    for l in range(j - rng, j + rng + 1):
                        lm = l - 1 + (1 if l == 1 else 0)
                        lp = l - 1 - (1 if l == ec - 2 else 0)
                        if not (0 < k < er - 1 and 0 < l < ec - 1):
                            continue
                        if f[k, l] == INSIDE or (l - j) * (l - j) + (k - i) * (k - i) > rng * rng:
                            continue
                        ry, rx = F32(i - k), F32(j - l)
                        len2 = F32(F32(rx * rx) + F32(ry * ry))
                        dst = F32(1.0 / (float(len2) * math.sqrt(float(len2))))
                        lev = F32(1.0 / (1 + abs(float(F32(t[k, l] - t[i, j])))))
                        dirv = F32(F32(rx * gtx) + F32(ry * gty))
                        if abs(float(dirv)) <= 0.01:
                            dirv = F32(0.000001)
                        w = F32(abs(F32(F32(dst * lev) * dirv)))
                        o_ = lambda a, b: int(out[a, b, c])
                        if f[k, l + 1] != INSIDE:
                            gix = F32(o_(km, lp + 1) - o_(km, lm - 1)) * F32(2.0) if f[k, l - 1] != INSIDE \
                                else F32(o_(km, lp + 1) - o_(km, lm))
                        else:
                            gix = F32(o_(km, lp) - o_(km, lm - 1)) if f[k, l - 1] != INSIDE else F32(0)
                        if f[k + 1, l] != INSIDE:
                            giy = F32(o_(kp + 1, lm) - o_(km - 1, lm)) * F32(2.0) if f[k - 1, l] != INSIDE \
                                else F32(o_(kp + 1, lm) - o_(km, lm))
                        else:
                            giy = F32(o_(kp, lm) - o_(km - 1, lm)) if f[k - 1, l] != INSIDE else F32(0)
                        Ia = F32(Ia + F32(w * F32(o_(km, lm))))
                        Jx = F32(Jx - F32(w * F32(gix * rx)))
                        Jy = F32(Jy - F32(w * F32(giy * ry)))
                        s = F32(s + w)
                nrm = F32(np.sqrt(F32(F32(Jx * Jx) + F32(Jy * Jy))))       # std::sqrt(float)
                sat = F32(F32(F32(Ia / s) + F32(F32(Jx + Jy) / F32(nrm + F32(1.0e-20)))) + F32(0.5))
                out[i - 1, j - 1, c] = _sat_u8(sat)
            f[i, j] = BAND
            heap.push(i, j, dist)
    return out
