"""Generate tests/golden/style_train_reference.npz by running the REFERENCE's own training code
(3_style_translator/training/{trainers,models,data}.py) on the CPU in this container.

    python tests/golden/make_style_train_golden.py     # needs /root/reference (not on the GPU box)

What is pinned:
  * the loop body of Trainer.train (trainers.py:148-172) — compute_discriminator_loss,
    compute_generator_loss, both Adam steps — for GeneratorJ_RIC (stage 1) and GeneratorJ
    (stage 2) with DiscriminatorN_IN and PerceptualVGG19(feature_layers=[0,3,5]): losses of two
    iterations, all parameter gradients of the first, parameters and BatchNorm buffers after
    the second;
  * DatasetPatches_M (data.py:56-180): patch cutting and the midpoint sampling order for a
    fixed numpy seed, on synthetic RGBA renders written to a temporary directory.

torchvision / cv2 are not installable here.  The shims below supply only third-party pieces:
`torchvision.ops.deform_conv2d` (oracle/style_ref.py restatement, differentiable through
autograd — "parity unpinned" for that op alone), `torchvision.models.vgg19` (the published
layer list, seeded random weights: there is no network for the ImageNet file),
`torchvision.transforms.{Compose,ToTensor,Normalize}` and the two cv2 calls of
custom_transforms.py.  Reduced widths keep the fixture small; the layer structure is the
shipped configs'.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/3_style_translator"

from oracle import style_ref  # noqa: E402

# ---------------------------------------------------------------- third-party shims
tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
tv.models = types.ModuleType("torchvision.models")
tv.transforms = types.ModuleType("torchvision.transforms")
tv.ops.deform_conv2d = lambda input, offset, weight, padding=(1, 1): \
    style_ref.deform_conv2d(input, offset, weight, padding).to(input.dtype)

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
        512, 512, 512, 512, "M"]


class _VGG19(nn.Module):
    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in _CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential()


def _vgg19(pretrained=False):
    g = torch.get_rng_state()
    torch.manual_seed(1234)
    m = _VGG19()
    torch.set_rng_state(g)
    return m


tv.models.vgg19 = _vgg19


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, pic):
        a = np.array(pic, np.float32) / 255.0
        if a.ndim == 2:
            a = a[..., None]
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()


class _Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


tv.transforms.Compose, tv.transforms.ToTensor, tv.transforms.Normalize = _Compose, _ToTensor, _Normalize
cv2 = types.ModuleType("cv2")
cv2.ROTATE_90_CLOCKWISE = 0
cv2.rotate = lambda img, code: np.ascontiguousarray(np.rot90(img, k=-1))
cv2.merge = lambda parts: np.dstack(parts)
for name, mod in (("torchvision", tv), ("torchvision.ops", tv.ops), ("torchvision.models", tv.models),
                  ("torchvision.transforms", tv.transforms), ("cv2", cv2)):
    sys.modules[name] = mod
torch.Tensor.cuda = lambda self, *a, **k: self      # generate_coordinates hard-codes .cuda()
sys.path.insert(0, REF)
from training import models as ref_models  # noqa: E402
from training import trainers as ref_trainers  # noqa: E402
from training import data as ref_data  # noqa: E402

G_ARGS = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
              filters=[8, 16, 24, 24, 24, 16], input_channels=6)
D_ARGS = dict(num_filters=4, n_layers=2)
OPT = dict(lr=0.0004, betas=[0.9, 0.999], weight_decay=0.00001)
TRAINER = dict(reconstruction_weight=4.0, adversarial_weight=0.5, perception_weight=6.0)
B, P = 4, 32


def make_trainer(perc):
    t = object.__new__(ref_trainers.Trainer)         # skip the file-reading constructor
    t.reconstruction_criterion = nn.L1Loss()
    t.adversarial_criterion = nn.MSELoss()
    t.reconstruction_weight = TRAINER["reconstruction_weight"]
    t.adversarial_weight = TRAINER["adversarial_weight"]
    t.perception_loss_weight = TRAINER["perception_weight"]
    t.perception_loss_model = perc
    t.use_adversarial_loss = True
    t.use_image_loss = True
    t.device = "cpu"
    return t


def train_fixture(gen_type, seed):
    torch.manual_seed(seed)
    gen = getattr(ref_models, gen_type)(**G_ARGS)
    disc = ref_models.DiscriminatorN_IN(**D_ARGS)
    perc = ref_models.PerceptualVGG19(feature_layers=[0, 3, 5], use_normalization=False)
    opt_g = ref_trainers.build_optimizer("Adam", gen, dict(OPT))
    opt_d = ref_trainers.build_optimizer("Adam", disc, dict(OPT))
    t = make_trainer(perc)
    out = {}
    pre = f"{gen_type}."
    for k, v in gen.state_dict().items():
        out[pre + "g0." + k] = v.clone().numpy()
    for k, v in disc.state_dict().items():
        out[pre + "d0." + k] = v.clone().numpy()
    for f in (0, 2, 5):                    # same seeded stack for both fixtures: stored once
        out[f"vgg.features.{f}.weight"] = perc.model.features[f].weight.detach().numpy()
        out[f"vgg.features.{f}.bias"] = perc.model.features[f].bias.detach().numpy()
    g = torch.Generator().manual_seed(seed + 7)
    for it in range(2):
        mask = (torch.rand(B, 1, P, P, generator=g) > 0.3).float()
        batch = {"pre": torch.rand(B, 6, P, P, generator=g) * 2 - 1, "pre_mask": mask,
                 "post": torch.rand(B, 3, P, P, generator=g) * 2 - 1,
                 "already": torch.rand(B, 3, P, P, generator=g) * 2 - 1,
                 "already_mask": (torch.rand(B, 1, P, P, generator=g) > 0.3).float()}
        for k, v in batch.items():
            out[pre + f"it{it}.batch.{k}"] = v.clone().numpy()
        # ---- Trainer.train loop body, trainers.py:148-172
        gen.train(); disc.train()
        opt_d.zero_grad()
        d_loss = t.compute_discriminator_loss(gen, disc, batch)
        d_loss.backward()
        if it == 0:
            for k, p in disc.named_parameters():
                out[pre + "it0.dgrad." + k] = p.grad.clone().numpy()
        opt_d.step()
        opt_g.zero_grad()
        li, lp, la, generated = t.compute_generator_loss(gen, disc, batch, use_gan=True, use_mask=False)
        g_loss = t.reconstruction_weight * li + t.perception_loss_weight * lp + t.adversarial_weight * la
        g_loss.backward()
        if it == 0:
            out[pre + "it0.generated"] = generated.detach().numpy()
            for k, p in gen.named_parameters():
                if p.grad is not None:
                    out[pre + "it0.ggrad." + k] = p.grad.clone().numpy()
        opt_g.step()
        out[pre + f"it{it}.losses"] = np.array([d_loss.item(), li.item(), lp.item(), la.item(),
                                                g_loss.item()], np.float64)
    for k, v in gen.state_dict().items():
        out[pre + "g2." + k] = v.clone().numpy()
    for k, v in disc.state_dict().items():
        out[pre + "d2." + k] = v.clone().numpy()
    return out


def dataset_fixture(use_edge, seed):
    """Reference DatasetPatches_M on synthetic renders; returns the images and the first items."""
    rng = np.random.RandomState(seed)
    H = W = 96
    yy, xx = np.mgrid[0:H, 0:W]
    alpha = (((yy - 50) ** 2 + (xx - 44) ** 2) < 30 ** 2).astype(np.uint8) * 255
    alpha[60:96, 70:96] = 255                       # touches the bottom/right borders
    color = np.dstack([rng.randint(0, 256, (H, W, 3)).astype(np.uint8), alpha])
    pos = np.dstack([rng.randint(0, 256, (H, W, 3)).astype(np.uint8), alpha])
    post = np.dstack([rng.randint(0, 256, (H, W, 3)).astype(np.uint8),
                      np.full((H, W), 255, np.uint8)])
    edge = np.full((H, W), 255, np.uint8)
    edge[rng.rand(H, W) < 0.05] = 0
    out = {"color": color, "pos": pos, "post": post, "edge": edge}
    with tempfile.TemporaryDirectory() as d:
        root = os.path.join(d, "rest_pose")
        for sub in ("color", "pos", "edge"):
            os.makedirs(os.path.join(root, sub))
        os.makedirs(os.path.join(d, "char"))
        Image.fromarray(color).save(os.path.join(root, "color", "0001.png"))
        Image.fromarray(pos).save(os.path.join(root, "pos", "0001.png"))
        Image.fromarray(edge).save(os.path.join(root, "edge", "0001.png"))
        Image.fromarray(post).save(os.path.join(d, "char", "tex.png"))
        ds = ref_data.DatasetPatches_M(root, "color", os.path.join(d, "char"), "tex", 32,
                                       use_mask=True, use_pos=True, use_edge=use_edge)
        out["len"] = np.array(len(ds))
        out["images_pre"] = ds.images_pre.numpy()
        out["images_post"] = ds.images_post.numpy()
        out["images_mask"] = ds.images_mask.numpy()
        np.random.seed(seed)
        items = [ds[0] for _ in range(6)]
    for k in ("pre", "pre_mask", "post", "already", "already_mask"):
        out["items." + k] = torch.stack([it[k] for it in items]).numpy()
    return {f"dataset.edge{int(use_edge)}.{k}": v for k, v in out.items()}


if __name__ == "__main__":
    out = {}
    out.update(train_fixture("GeneratorJ_RIC", 31))
    out.update(train_fixture("GeneratorJ", 32))
    out.update(dataset_fixture(False, 5))
    out.update(dataset_fixture(True, 6))
    path = os.path.join(ROOT, "tests", "golden", "style_train_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;",
          {k: out[k].tolist() for k in out if k.endswith(".losses")})
