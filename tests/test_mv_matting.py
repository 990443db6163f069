"""Side-view matting seam (drawingspinup_amd/mv/matting.py, mv.py:113-150): the function around the
session against the reference's own `remove_background` / `add_gray` (tests/golden/
matting_reference.npz, make_matting_golden.py), and the IS-Net restatement's structure."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from drawingspinup_amd.entry import data as D
from drawingspinup_amd.mv import matting

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_matting_golden as G  # noqa: E402

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "matting_reference.npz"))


def test_remove_background_equals_the_reference_function():
    sess = G.StubSession()
    matte = matting.remove_background(sess, G.synthetic_rgb())
    assert sess.fed.dtype == np.float32 and np.array_equal(sess.fed, FIX["fed"])    # what the session sees
    assert matte.mode == str(FIX["matte_mode"]) and np.array_equal(np.array(matte), FIX["matte"])
    # the clip is exercised: the stub leaves [0, 1] on both sides
    raw = G.stub_network(FIX["fed"])[0]
    assert raw.min() < 0 and raw.max() > 1 and FIX["matte"].min() == 0 and FIX["matte"].max() == 255


def test_add_gray_equals_the_reference_function():
    g = D.add_gray(G.synthetic_rgba())
    assert g.mode == str(FIX["add_gray_mode"]) and np.array_equal(np.array(g), FIX["add_gray"])


def test_isnet_parameter_names_and_shapes_follow_the_dis_checkpoint():
    net = matting.ISNetDIS()
    sd = net.state_dict()
    # a sample of isnet-general-use.pth's keys / shapes (xuebinqin/DIS, models/isnet.py)
    for key, shape in {
        "conv_in.weight": (64, 3, 3, 3),
        "stage1.rebnconvin.conv_s1.weight": (64, 64, 3, 3),
        "stage1.rebnconv1.conv_s1.weight": (32, 64, 3, 3),
        "stage1.rebnconv7.bn_s1.running_var": (32,),
        "stage1.rebnconv6d.conv_s1.weight": (32, 64, 3, 3),
        "stage1.rebnconv1d.conv_s1.weight": (64, 64, 3, 3),
        "stage2.rebnconv6.conv_s1.weight": (32, 32, 3, 3),
        "stage4.rebnconv4.conv_s1.weight": (128, 128, 3, 3),
        "stage5.rebnconv4.conv_s1.weight": (256, 256, 3, 3),
        "stage6.rebnconv3d.conv_s1.weight": (256, 512, 3, 3),
        "stage5d.rebnconvin.conv_s1.weight": (512, 1024, 3, 3),
        "stage4d.rebnconv1d.conv_s1.weight": (256, 256, 3, 3),
        "stage1d.rebnconv1.conv_s1.weight": (16, 64, 3, 3),
        "side1.weight": (1, 64, 3, 3),
        "side6.bias": (1,),
    }.items():
        assert tuple(sd[key].shape) == shape, key
    assert not any(k.startswith("stage2.rebnconv7") for k in sd)            # RSU6 has six levels
    assert net.stage1.rebnconv7.conv_s1.dilation == (2, 2) and net.stage6.rebnconv4.conv_s1.dilation == (8, 8)
    n = sum(p.numel() for p in net.parameters())
    assert 43e6 < n < 45e6                                                  # IS-Net: ~44 M parameters


def test_isnet_full_key_and_shape_list():
    """The whole state_dict against the committed list (tests/golden/isnet_dis_state_dict_keys.json:
    generated from this restatement — the DIS checkpoint itself is not available here; the list is
    what tools/isnet_keys_check.py diffs against a real isnet-general-use.pth)."""
    import json
    import os
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                       "isnet_dis_state_dict_keys.json")))["entries"]
    sd = matting.ISNetDIS().state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == want
    convs = [k for k in want if k.endswith("conv_s1.weight")]
    assert len(convs) == 112 and len(want) > 700     # 112 REBNCONV blocks (tools/matting_time.py)


def test_isnet_session_contract_on_cpu():
    torch.manual_seed(0)
    net = matting.load_isnet(None, "cpu")
    sess = matting.IsnetSession(net)
    img = G.synthetic_rgb(3, 64, 48)
    m = matting.remove_background(sess, img)
    assert m.mode == "L" and m.size == img.size
    out = sess.run(None, {sess.get_inputs()[0].name: np.zeros((1, 3, 48, 64), np.float32)})
    assert out[0].shape == (1, 1, 48, 64) and 0.0 <= out[0].min() and out[0].max() <= 1.0


def test_dilated_convolution_as_sublattice_convolutions(monkeypatch):
    """The decomposition _rebnconv_hip uses (d x d ordinary convolutions on x[i::d, j::d]) against
    torch's dilated convolution, with the library call replaced by its torch meaning."""
    def conv(x, conv, bias, stride, padding, scale, shift, act):
        y = F.conv2d(x, conv.weight, bias, stride, padding) * scale[None, :, None, None] \
            + shift[None, :, None, None]
        return F.relu(y) if act == "relu" else y
    monkeypatch.setattr(matting, "_conv", conv)
    torch.manual_seed(1)
    for d, hw in ((2, (16, 16)), (4, (32, 32)), (8, (32, 32)), (2, (13, 18)), (4, (9, 7))):
        m = matting.REBNCONV(5, 7, dirate=d).eval()
        m.bn_s1.running_mean.normal_(); m.bn_s1.running_var.uniform_(0.5, 2.0)
        m.bn_s1.weight.data.normal_(); m.bn_s1.bias.data.normal_()
        x = torch.randn(2, 5, *hw)
        ref = m.relu_s1(m.bn_s1(m.conv_s1(x)))
        torch.testing.assert_close(matting._rebnconv_hip(m, x), ref, rtol=1e-5, atol=1e-5)
