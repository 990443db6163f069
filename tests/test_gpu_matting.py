"""IS-Net forward of the side-view matting seam on the library's convolution kernel (mv/matting.py)
against the same module evaluated by torch on the CPU, and the session contract on the device."""
import numpy as np
import pytest
import torch

from drawingspinup_amd.mv import matting

pytestmark = pytest.mark.gpu


def _randomise_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.mark.parametrize("x3", [True, False])
def test_isnet_forward_on_the_hip_convolution_matches_torch_cpu(dev, x3, monkeypatch):
    monkeypatch.setattr(matting, "EVAL_X3", x3)
    net = matting.load_isnet(None, "cpu", seed=3)
    _randomise_bn(net, 4)
    x = torch.rand(1, 3, 208, 176, generator=torch.Generator().manual_seed(5)) - 0.5   # odd pooled sizes below
    with torch.no_grad():
        ref = net(x)
        got = net.to(dev)(x.to(dev)).cpu()
    assert got.shape == ref.shape == (1, 1, 208, 176)
    assert float(ref.std()) > 1e-4                                   # not a constant map
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-4)


def test_isnet_forward_batch_of_four_at_1024_matches_torch_cpu(dev, monkeypatch):
    """The shape the drawing pipeline runs (drawing.py: the four 1024^2 side views as one batch),
    both arithmetics, against torch on the host CPU."""
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))          # torch's CPU convolutions crawl at 256 threads
    try:
        net = matting.load_isnet(None, "cpu", seed=8)
        _randomise_bn(net, 9)
        x = torch.rand(4, 3, 1024, 1024, generator=torch.Generator().manual_seed(10)) - 0.5
        with torch.no_grad():
            ref = net(x)
    finally:
        torch.set_num_threads(threads)
    assert ref.shape == (4, 1, 1024, 1024) and float(ref.std()) > 1e-4
    net = net.to(dev)
    for x3 in (False, True):
        monkeypatch.setattr(matting, "EVAL_X3", x3)
        with torch.no_grad():
            got = net(x.to(dev)).cpu()
            one = net(x[2:3].to(dev)).cpu()          # the per-image call of the reference's session
        torch.testing.assert_close(got, ref, rtol=0, atol=2e-4)
        torch.testing.assert_close(got[2:3], one, rtol=0, atol=2e-6)     # batch-independent


def test_remove_background_through_the_device_session(dev):
    net = matting.load_isnet(None, dev, seed=6)
    sess = matting.IsnetSession(net, dev)
    rng = np.random.default_rng(7)
    from PIL import Image
    img = Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8), "RGB")
    m = matting.remove_background(sess, img)
    assert m.mode == "L" and m.size == (256, 256)
    again = matting.remove_background(sess, img)
    assert np.array_equal(np.array(m), np.array(again))
