"""CPU: known answers for the stand-ins behind the colour back-projection fixture
(oracle/mesh_post_ref.py) and the fixture's own consistency."""
import os

import numpy as np

from oracle import mesh_post_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_color_reference.npz"))


def test_elliptic_element_known_values():
    """OpenCV's documented 3x3 and 5x5 elliptic elements."""
    assert R.getStructuringElement(R.MORPH_ELLIPSE, (3, 3)).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert R.getStructuringElement(R.MORPH_ELLIPSE, (5, 5)).tolist() == \
        [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    el = R.getStructuringElement(R.MORPH_ELLIPSE, (19, 19))
    assert el.shape == (19, 19) and el[9].sum() == 19 and el[0].sum() == 1 and np.array_equal(el, el.T[::-1, ::-1].T)
    img = np.full((40, 40), 255, np.uint8)
    img[20, 20] = 0
    e = R.erode(img, el)
    assert (e == 0).sum() == el.sum() and e[0, 0] == 255          # a hole grows into the element; border stays


def test_raycast_known_answers():
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 0, 1], [0, 1, 1]]], np.float32)
    rs = R.raycast((0.25, 0.25, 2.0), (0, 0, -1), tri)
    assert sorted(r["distance"] for r in rs) == [1.0, 2.0] and {r["face"] for r in rs} == {0, 1}
    assert R.raycast((0.25, 0.25, 0.5), (0, 0, 1), tri)[0]["face"] == 1            # only what is ahead
    assert R.raycast((0.9, 0.9, 2.0), (0, 0, -1), tri) == []                        # outside both
    assert len(R.raycast((0.5, 0.5, 2.0), (0, 0, -1), tri)) == 2                    # on the hypotenuse: inclusive
    rs = R.raycast((1, 0, 0), (0, 0, 1), tri)                                       # from a vertex
    assert [(r["face"], r["distance"]) for r in rs] == [(0, 0.0), (1, 1.0)]


def test_fixture_is_self_consistent():
    v, f, c = GOLD["verts"], GOLD["faces"], GOLD["vert_colors"]
    assert c.shape == (len(v), 3) and np.all((c >= 0) & (c <= 1))
    assert f.min() == 0 and f.max() == len(v) - 1
    for ty in ("double", "front", "back"):
        m, o = GOLD["offset_mask_" + ty], GOLD["offset_values_" + ty]
        assert np.all(o[~m] == 0) and np.all(o[:, :2] == 0) and np.abs(o).max() < 0.06
    assert np.all(GOLD["offset_values_double"][:, 2][GOLD["verts"][:, 2] > 0.02] <= 0)   # front side moves back
