"""Contour-remover generator against the REFERENCE's own FFC-ResNet: the fixture
tests/golden/contour_reference.npz was produced by importing
1_lama_contour_remover/saicinpainting/training/modules (make_contour_golden.py), so these tests
pin state_dict layout and values to the reference end to end."""
import os

import numpy as np
import pytest
import torch

from drawingspinup_amd.contour import FFCResNetGenerator, LAMA_FOURIER_GENERATOR, make_generator
from drawingspinup_amd.contour.ffc import irfft2_ortho, rfft2_ortho

GOLD = os.path.join(os.path.dirname(__file__), "golden", "contour_reference.npz")
SMALL = dict(LAMA_FOURIER_GENERATOR, ngf=8, n_blocks=2)


def _small_from_fixture(device="cpu"):
    z = np.load(GOLD)
    net = make_generator(**SMALL).eval()
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    net.load_state_dict(sd, strict=True)            # same keys, same shapes as the reference
    return z, net.to(device)


def test_full_size_state_dict_layout_matches_reference():
    z = np.load(GOLD)
    net = FFCResNetGenerator(**LAMA_FOURIER_GENERATOR)
    mine = [(k, ",".join(map(str, v.shape))) for k, v in net.state_dict().items()]
    ref = list(zip(z["full_keys"].tolist(), z["full_shapes"].tolist()))
    assert mine == ref
    assert sum(p.numel() for p in net.parameters()) == int(z["full_params"]) == 27042561


def test_reduced_generator_matches_reference_output():
    z, net = _small_from_fixture()
    with torch.no_grad():
        y = net(torch.from_numpy(z["x"]))
    # f32 on both sides; the FourierUnit's FFT is evaluated as DFT products here
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=0, atol=2e-6)
    logit = lambda p: np.log(p / (1 - p))
    np.testing.assert_allclose(logit(y.numpy()), logit(z["y"]), rtol=0, atol=1e-5)


@pytest.mark.parametrize("h,w", [(8, 8), (12, 10), (9, 7), (64, 64)])
def test_dft_products_match_fft_library(h, w):
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(2, 3, h, w, generator=g, dtype=torch.float64)
    ref = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")
    re, im = rfft2_ortho(x)
    torch.testing.assert_close(re, ref.real, rtol=0, atol=1e-12)
    torch.testing.assert_close(im, ref.imag, rtol=0, atol=1e-12)
    zc = torch.randn(2, 3, h, w // 2 + 1, generator=g, dtype=torch.float64) \
        + 1j * torch.randn(2, 3, h, w // 2 + 1, generator=g, dtype=torch.float64)
    back = torch.fft.irfftn(zc, s=(h, w), dim=(-2, -1), norm="ortho")
    torch.testing.assert_close(irfft2_ortho(zc.real, zc.imag, h, w), back, rtol=0, atol=1e-12)
    torch.testing.assert_close(irfft2_ortho(re, im, h, w), x, rtol=0, atol=1e-12)   # round trip


def test_unsupported_options_raise():
    with pytest.raises(NotImplementedError):
        FFCResNetGenerator(**dict(LAMA_FOURIER_GENERATOR, out_ffc=True))
    with pytest.raises(NotImplementedError):
        FFCResNetGenerator(**dict(LAMA_FOURIER_GENERATOR,
                                  resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout=0.75,
                                                          enable_lfu=True)))
    with pytest.raises(ValueError):
        make_generator("pix2pixhd_global")


@pytest.mark.gpu
def test_generator_on_device_matches_fixture(dev):
    z, net = _small_from_fixture(dev)
    with torch.no_grad():
        y = net(torch.from_numpy(z["x"]).to(dev))
    np.testing.assert_allclose(y.cpu().numpy(), z["y"], rtol=0, atol=5e-6)


def test_prepare_input_and_masks_follow_predict_py():
    from PIL import Image
    from drawingspinup_amd.contour.predict import contour_masks, prepare_input
    rng = np.random.default_rng(0)
    rgba = rng.integers(0, 256, (64, 48, 4), dtype=np.uint8)
    rgba[:16, :, 3] = 0                                  # transparent band -> white, mask 0
    rgba[48:, :, 3] = 255
    x = prepare_input(rgba, size=32)
    assert x.shape == (1, 4, 32, 32) and x.dtype == torch.float32
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    # restated with PIL primitives directly: paste-on-white through alpha, bicubic resize
    img = Image.fromarray(rgba)
    ref = Image.new("RGB", img.size, (255, 255, 255))
    ref.paste(img, (0, 0), img)
    ref = np.asarray(ref.resize((32, 32), Image.BICUBIC), np.float32) / 255
    np.testing.assert_array_equal(x[0, :3].permute(1, 2, 0).numpy(), ref)
    assert float(x[0, :3, :6].min()) == 1.0              # fully transparent rows are white
    assert float(x[0, 3, :6].max()) == 0.0

    class Fixed(torch.nn.Module):                        # a "model" with a known probability map
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, inp):
            prob = torch.full((1, 1, 32, 32), 0.1)
            prob[0, 0, 10:12] = 0.9
            prob[0, 0, 20] = 0.2                         # not strictly above the threshold
            return prob

    img8, alpha, contour, mask = contour_masks(Fixed(), x)
    assert img8.shape == (32, 32, 3) and img8.dtype == np.uint8
    assert contour[10:12].min() == 255 and contour[20].max() == 0 and contour[:10].max() == 0
    np.testing.assert_array_equal(mask, np.maximum(contour, 255 - alpha))


def test_remove_contour_end_to_end_on_the_host():
    """predict.py:47-66 for one drawing (prepare -> generator -> masks -> TELEA -> RGBA) with a
    model whose contour map is known: contour strokes are repainted from the character's own
    pixels, known character pixels and the alpha channel are untouched."""
    from drawingspinup_amd.contour.predict import prepare_input, remove_contour
    size = 48
    yy, xx = np.mgrid[0:size, 0:size]
    inside = ((yy - 24) / 18.0) ** 2 + ((xx - 24) / 14.0) ** 2 < 1.0
    stroke = inside & (np.abs(yy - 24) < 1)
    rgba = np.zeros((size, size, 4), np.uint8)
    rgba[inside] = (180, 90, 40, 255)
    rgba[stroke, :3] = 0                                  # a dark contour line across the body

    class Known(torch.nn.Module):                         # probability 0.9 on the stroke, 0.05 elsewhere
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            m = torch.from_numpy(stroke).float()
            return (0.05 + 0.85 * m)[None, None] + 0 * self.p

    out = remove_contour(Known(), rgba, size=size)
    assert out.shape == (size, size, 4) and out.dtype == np.uint8
    x = prepare_input(rgba, size)[0].permute(1, 2, 0).numpy()
    alpha = (x[:, :, 3] * 255).astype(np.uint8)
    assert np.array_equal(out[:, :, 3], alpha)
    keep = inside & ~stroke
    assert np.array_equal(out[keep][:, :3], (x[:, :, :3] * 255).astype(np.uint8)[keep])
    body = out[stroke & (xx > 14) & (xx < 34)][:, :3].astype(int)
    assert np.abs(body - np.array([180, 90, 40])).max() <= 3          # the line is gone


def test_kernel_form_of_the_generator_is_the_same_function(monkeypatch):
    """contour/ffc_hip.py re-expresses every layer through ONE operator (ops.conv2d: zero-padded
    convolution + folded BatchNorm / activation / residual epilogue): concatenated FFC branches,
    reflect padding as a copy, transposed convolutions as convolutions of the zero-dilated input,
    the FourierUnit's transforms as 1x1 convolutions over the transformed axis.  With the operator
    served by torch on the CPU the re-expression must reproduce the module tree's output."""
    import torch.nn.functional as F
    from drawingspinup_amd.contour import ffc_hip

    def conv2d(x, w, bias=None, stride=1, padding=0, ep_scale=None, ep_shift=None, act=None,
               residual=None, in_relu=False):
        y = F.conv2d(x, w, bias, stride, padding)
        if ep_scale is not None:
            y = y * ep_scale[None, :, None, None] + ep_shift[None, :, None, None]
        if act == "relu":
            y = torch.relu(y)
        return y if residual is None else y + residual
    monkeypatch.setattr(ffc_hip.ops, "conv2d", conv2d)
    torch.manual_seed(0)
    cfg = dict(LAMA_FOURIER_GENERATOR, ngf=8, n_blocks=2)
    net = make_generator(**cfg).eval()
    for m in net.modules():                                   # non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    for hw in ((32, 32), (48, 40)):
        x = torch.rand(2, 4, *hw)
        with torch.no_grad():
            want = net.model(x)
            got = ffc_hip.generator_forward(net, x)
        torch.testing.assert_close(got, want, rtol=0, atol=2e-5)
