"""CPU: the native TELEA inpainting (libdsu_hip.so, host code) against the pure-Python restatement
of OpenCV's algorithm (oracle/telea_ref.py) bit for bit, and properties of any correct result.
OpenCV itself is absent here: parity with cv2.inpaint is unpinned (see the oracle's header)."""
import numpy as np
import pytest

from drawingspinup_amd.contour.predict import inpaint
from oracle import telea_ref as R


def _case(seed, h, w, kind):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + 40 + 30 * np.sin(yy / 3.0)) % 256,
                    rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    mask = np.zeros((h, w), np.uint8)
    if kind == "lines":                       # contour-like strokes
        mask[h // 3, 2:w - 2] = 255
        mask[3:h - 3, w // 2] = 255
        mask[h // 2:h // 2 + 2, w // 4:w // 2] = 255
    elif kind == "blob":
        mask[(yy - h / 2) ** 2 + (xx - w / 2) ** 2 < (min(h, w) / 4) ** 2] = 1
    elif kind == "border":                    # background around a character + a stroke, touches the frame
        mask[:] = 255
        mask[4:h - 5, 5:w - 4] = 0
        mask[h // 2, 6:w - 6] = 200
    elif kind == "random":
        mask[rng.random((h, w)) < 0.3] = 255
    return img, mask


@pytest.mark.parametrize("kind", ["lines", "blob", "border", "random"])
@pytest.mark.parametrize("radius", [3, 1])
def test_native_telea_equals_the_restatement(kind, radius):
    img, mask = _case(7, 21, 26, kind)
    got = inpaint(img, mask, radius)
    ref = R.inpaint_telea(img, mask, radius)
    assert got.dtype == np.uint8 and got.shape == img.shape
    assert np.array_equal(got, ref)
    # known pixels are never touched
    assert np.array_equal(got[mask == 0], img[mask == 0])


def test_properties_at_contour_stage_size():
    """512 x 512, predict.py's mask shape (strokes + the whole background): constant colour regions
    stay constant, every filled value lies within the range of the known pixels, runs in well under
    a second."""
    import time
    h = w = 512
    yy, xx = np.mgrid[0:h, 0:w]
    inside = ((yy - 256) / 200.0) ** 2 + ((xx - 256) / 140.0) ** 2 < 1.0
    img = np.full((h, w, 3), 255, np.uint8)
    img[inside] = (200, 120, 60)
    stroke = inside & ((np.abs(yy - 256 - 40 * np.sin(xx / 30.0)) < 2) | (np.abs(xx - 230) < 2))
    img[stroke] = (10, 10, 10)                          # dark contour lines on a flat fill
    mask = np.where(stroke | ~inside, 255, 0).astype(np.uint8)
    t = time.time()
    out = inpaint(img, mask, 3)
    dt = time.time() - t
    assert dt < 2.0, dt
    assert np.array_equal(out[mask == 0], img[mask == 0])
    # the strokes are repainted with the fill colour (all their known neighbours have it); the
    # normalised gradient term (Jx + Jy) / |J| of OpenCV's formula is +-1.41 as soon as two already
    # filled neighbours differ by one grey level, plus the +0.5 before rounding: a few levels of drift
    filled = out[stroke & (yy > 1) & (xx > 1)]
    assert np.abs(filled.astype(int) - np.array([200, 120, 60])).max() <= 3
    # (far inside the background hole the same term drifts further: no range property there)


def test_argument_checks():
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(ValueError):
        inpaint(img, np.zeros((8, 7), np.uint8))
    from drawingspinup_amd._lib import DsuError
    with pytest.raises(DsuError):
        inpaint(np.zeros((2, 8, 3), np.uint8), np.zeros((2, 8), np.uint8))
    # nothing to fill: identity
    assert np.array_equal(inpaint(img + 5, np.zeros((8, 8), np.uint8)), img + 5)
