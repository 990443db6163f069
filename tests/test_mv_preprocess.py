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
