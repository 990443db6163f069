"""drawingspinup_amd/mv/preprocess.py against the real libraries that are installed here: Pillow
(the resampler `CLIPImageProcessor` calls) and transformers' own CLIPImageProcessor."""
import numpy as np
import pytest
import torch
from PIL import Image

from drawingspinup_amd.mv import preprocess as P


def _img(h, w, seed):
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3)).astype(np.uint8)
    img = np.kron(base, np.ones((4, 4, 1), np.uint8))[:h, :w]
    img[::7, ::5] = g.integers(0, 256, img[::7, ::5].shape)           # some single-pixel detail
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("hw,out", [((256, 256), (224, 224)), ((128, 128), (224, 224)),
                                    ((300, 200), (224, 224)), ((256, 256), (256, 256)),
                                    ((64, 80), (17, 31))])
def test_resize_is_pillows_bicubic_bit_for_bit(hw, out):
    img = _img(*hw, seed=hw[0] + out[0])
    want = np.asarray(Image.fromarray(img).resize((out[1], out[0]), Image.BICUBIC))
    got = P.pil_resize_bicubic_u8(torch.from_numpy(img), out).numpy()
    assert np.array_equal(got, want)


def test_to_pil_truncates_in_the_tensor_dtype():
    # exact k/255 values survive the f16 round trip (the f16 product snaps back to k) ...
    k = torch.arange(256, dtype=torch.float32)
    x = (k / 255).half().view(1, 1, 16, 16).expand(1, 3, 16, 16)
    q = P.to_pil_u8(x)[0, :, :, 0].reshape(-1).int()
    assert torch.equal(q, k.int())
    # ... composited values (single_image_dataset.py:118-121: rgb * alpha + bg * (1 - alpha)) do
    # not: the product is truncated, not rounded (torchvision to_pil_image: pic.mul(255).byte())
    v = torch.tensor([0.5, 0.9999, 0.2509, 0.7]).half().view(1, 1, 2, 2).expand(1, 3, 2, 2)
    got = P.to_pil_u8(v)[0, :, :, 0].reshape(-1).tolist()
    assert got == v[0, 0].reshape(-1).mul(255).byte().tolist() == [127, 255, 64, 178]


def test_clip_pixel_values_match_transformers_processor():
    from transformers import CLIPImageProcessor
    img = _img(256, 256, 3)
    ref = CLIPImageProcessor()(images=[Image.fromarray(img)], return_tensors="pt").pixel_values
    got = P.clip_pixel_values(torch.from_numpy(img)[None])
    assert got.shape == ref.shape == (1, 3, 224, 224)
    assert float((got - ref).abs().max()) < 2e-6


# ------------------------------------------------------------------------------------------------
# The stage hand-offs of DrawingPipeline (what bench.py times) use the resamplers below on the
# device; pinned here to the installed Pillow, the reference's own resampler for those hand-offs:
# mv.py:105-106 (256 -> 1024 LANCZOS), coloring_utils.py:62,100 (1024 -> 2048 LANCZOS),
# single_image_dataset.py:112 (RGBA default = BICUBIC resize to 256, premultiplied alpha).
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("filt,pil_filter", [("lanczos", Image.LANCZOS), ("bicubic", Image.BICUBIC)])
@pytest.mark.parametrize("hw,out,ch", [((256, 256), (1024, 1024), 3), ((64, 48), (200, 130), 3),
                                       ((100, 100), (37, 41), 1), ((512, 512), (1024, 1024), 1)])
def test_resize_is_pillows_filter_bit_for_bit(filt, pil_filter, hw, out, ch):
    from drawingspinup_amd.mv.preprocess import pil_resize_u8
    rng = np.random.default_rng(sum(hw) + ch)
    a = rng.integers(0, 256, hw + (ch,), dtype=np.uint8)
    pil = Image.fromarray(a[..., 0], "L") if ch == 1 else Image.fromarray(a, "RGB")
    want = np.asarray(pil.resize((out[1], out[0]), pil_filter)).reshape(out + (ch,))
    got = pil_resize_u8(torch.from_numpy(a), out, filt).numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("hw,out", [((512, 512), (256, 256)), ((60, 90), (33, 47)), ((40, 40), (100, 100))])
def test_rgba_resize_is_pillows_premultiplied_route_bit_for_bit(hw, out):
    from drawingspinup_amd.mv.preprocess import pil_resize_rgba_u8
    rng = np.random.default_rng(hw[0])
    a = rng.integers(0, 256, hw + (4,), dtype=np.uint8)
    a[: hw[0] // 3, :, 3] = 0                      # transparent band: colour must come back untouched
    a[hw[0] // 3: hw[0] // 2, :, 3] = 255
    want = np.asarray(Image.fromarray(a, "RGBA").resize((out[1], out[0])))
    got = pil_resize_rgba_u8(torch.from_numpy(a), out).numpy()
    assert np.array_equal(got, want)


def test_drawing_pipeline_input_preparation_equals_load_image_rgba():
    """DrawingPipeline.multiview's 256^2 network input == entry/data.py load_image_rgba (the
    restatement of SingleImageDataset.load_image the entry script uses) on the same RGBA drawing."""
    from drawingspinup_amd.drawing import _u8_hwc, synthetic_drawing
    from drawingspinup_amd.entry.data import load_image_rgba
    from drawingspinup_amd.mv.preprocess import pil_resize_rgba_u8
    d = synthetic_drawing(5, device="cpu")
    d = (d * 255 + 0.5).to(torch.uint8).float() / 255.0             # an 8-bit image, as after remove_contour
    u8 = _u8_hwc(d)
    want, _ = load_image_rgba(Image.fromarray(u8.numpy(), "RGBA"))
    small = pil_resize_rgba_u8(u8, (256, 256), "bicubic").float() / 255.0
    got = small[..., :3] * small[..., 3:4] + (1.0 - small[..., 3:4])
    assert torch.equal(got, want)
