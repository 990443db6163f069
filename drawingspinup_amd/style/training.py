"""Per-character training of the style translator on gfx950.

Mirrors 3_style_translator/training/{trainers.py, models.py:426-549, data.py:56-180} and the
train_stage{1,2}.py drivers: same class names, constructor arguments, state_dict keys, loss
composition and optimiser settings, so a checkpoint or config of the reference drops in.
What differs is where the work runs:

  * every layer forward/backward is a libdsu_hip kernel (style/functions.py);
  * the rest-pose images stay resident on the GPU and the 32x32 patches of a batch are cut
    there (the reference cuts them on the host in a DataLoader worker and uploads 5 tensors per
    iteration);
  * the generator is evaluated ONCE per iteration: the reference runs it for the discriminator
    step and again for the generator step on the same batch with unchanged generator weights
    (trainers.py:88,102), which gives the same activations; the BatchNorm running statistics
    take the batch twice (`stat_updates = 2`) so the buffers match the reference's.

There is no CPU path: the modules raise on host tensors.
"""
import os
import shutil
import time

import numpy as np
import torch
import torch.nn as nn
from PIL import Image, ImageFilter

from ..entry.data import (DatasetFullImages, _rgb_normalised, _to_tensor, overlap_edge_on_img,
                          to_image_space)
from . import functions as Fn
from .generators import GeneratorJ, GeneratorJ_RIC


def _require_device(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"gfx950 {what} needs a device tensor (no CPU fallback)")


# ------------------------------------------------------------------ losses (torch.nn names)
class L1Loss(nn.Module):
    def forward(self, x, target):
        return Fn.l1_loss(x, target)


class MSELoss(nn.Module):
    def forward(self, x, target):
        return Fn.mse_loss(x, target)


_CRITERIA = {"L1Loss": L1Loss, "MSELoss": MSELoss}


# ------------------------------------------------------------------ DiscriminatorN_IN
class DiscriminatorN_IN(nn.Module):
    """models.py:426-477: k=4 convolutions, InstanceNorm2d on the inner blocks,
    LeakyReLU(0.2); returns (patch logits, None)."""

    def __init__(self, num_filters=64, input_channels=3, n_layers=3, use_noise=False,
                 noise_sigma=0.2, norm_layer="instance_norm", use_bias=True):
        super().__init__()
        assert norm_layer == "instance_norm", "gfx950 discriminator: instance_norm only"
        self.num_filters, self.input_channels = num_filters, input_channels
        self.use_noise, self.noise_sigma, self.use_bias = use_noise, noise_sigma, use_bias
        self.norm_layer = nn.InstanceNorm2d
        self.net = self.make_net(n_layers, input_channels, 1, 4, 2, use_bias)

    def make_net(self, n, flt_in, flt_out=1, k=4, stride=2, bias=True):
        padding = 1
        model = nn.Sequential()
        model.add_module("conv0", self.make_block(flt_in, self.num_filters, k, stride, padding,
                                                  bias, None, nn.LeakyReLU))
        mult = 1
        for layer in range(1, n):
            prev, mult = mult, min(2 ** layer, 8)
            model.add_module("conv_%d" % layer,
                             self.make_block(self.num_filters * prev, self.num_filters * mult, k,
                                             stride, padding, bias, self.norm_layer, nn.LeakyReLU))
        prev, mult = mult, min(2 ** n, 8)
        model.add_module("conv_%d" % n,
                         self.make_block(self.num_filters * prev, self.num_filters * mult, k, 1,
                                         padding, bias, self.norm_layer, nn.LeakyReLU))
        model.add_module("conv_out", self.make_block(self.num_filters * mult, flt_out, k, 1,
                                                     padding, bias, None, None))
        return model

    @staticmethod
    def make_block(flt_in, flt_out, k, stride, padding, bias, norm, relu):
        m = nn.Sequential()
        m.add_module("conv", nn.Conv2d(flt_in, flt_out, k, stride=stride, padding=padding,
                                       bias=bias))
        if norm is not None:
            m.add_module("norm", norm(flt_out))
        if relu is not None:
            m.add_module("relu", relu(0.2, True))
        return m

    def forward(self, x):
        _require_device(x, "discriminator")
        h = x.float()
        for block in self.net:
            conv = block.conv
            has_norm, has_act = hasattr(block, "norm"), hasattr(block, "relu")
            fused = "leaky_relu" if (has_act and not has_norm) else None
            h = Fn.conv(h, conv.weight, conv.bias, conv.stride[0], conv.padding[0], fused)
            if has_norm:
                h = Fn.instance_norm(h, "leaky_relu" if has_act else None, block.norm.eps)
        return h, None


# ------------------------------------------------------------------ PerceptualVGG19
_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
              512, 512, 512, 512, "M"]


def _vgg19_features():
    layers, c = [], 3
    for v in _VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


def find_vgg19_weights():
    """DSU_VGG19_WEIGHTS, else ~/.cache/torch/hub/checkpoints/vgg19-*.pth (where
    torchvision's models.vgg19(pretrained=True) leaves the file)."""
    import glob
    cand = [os.environ.get("DSU_VGG19_WEIGHTS")]
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub",
                       "checkpoints")
    cand += sorted(glob.glob(os.path.join(hub, "vgg19-*.pth")))
    for c in cand:
        if c and os.path.isfile(c):
            return c
    return None


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _vgg19_features()


class PerceptualVGG19(nn.Module):
    """models.py:480-549.  Holds torchvision's vgg19 `features` stack under the same keys
    (`model.features.N.weight`); only layers up to max(feature_layers) are ever evaluated.
    `path`: a vgg19 state_dict file (torchvision layout; classifier entries are ignored).
    Without a path the reference downloads the ImageNet weights (models.vgg19(pretrained=True),
    models.py:497).  There is no network here: the file is looked up in DSU_VGG19_WEIGHTS and the
    torch hub cache, and a missing file is an ERROR unless `random_init=True` is passed — the
    perceptual term carries the largest weight of the three (6.0) and random features would train
    a different model without any sign of it."""

    def __init__(self, feature_layers, use_normalization=True, path=None, random_init=False):
        super().__init__()
        self.model = _VGG()
        if path is None and not random_init:
            path = find_vgg19_weights()
            if path is None:
                raise FileNotFoundError(
                    "PerceptualVGG19: no ImageNet vgg19 weights (pass path=..., set "
                    "DSU_VGG19_WEIGHTS, put vgg19-*.pth in the torch hub cache, or pass "
                    "random_init=True to train against random features knowingly)")
        if path is not None:
            sd = torch.load(path, map_location="cpu")
            self.model.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.")})
        self.model.float().eval()
        self.feature_layers = list(feature_layers)
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]), persistent=False)
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]), persistent=False)
        self.use_normalization = use_normalization
        for p in self.parameters():
            p.requires_grad = False

    def normalize(self, x):
        if not self.use_normalization:
            return x
        x = (x + 1) / 2
        return (x - self.mean.view(1, 3, 1, 1)) / self.std.view(1, 3, 1, 1)

    def run(self, x):
        feats, h = [], x
        last = max(self.feature_layers)
        f = 0
        while f <= last:
            layer = self.model.features[f]
            if isinstance(layer, nn.Conv2d):
                # the reference's ReLU is in place, applied after the pre-activation feature was
                # cloned: feature f is the raw convolution, feature f+1 its ReLU
                nxt_relu = f + 1 <= last and (f not in self.feature_layers)
                if nxt_relu:
                    h = Fn.conv(h, layer.weight, layer.bias, 1, 1, "relu")
                    f += 1
                else:
                    h = Fn.conv(h, layer.weight, layer.bias, 1, 1)
            elif isinstance(layer, nn.ReLU):
                h = Fn.activation(h, "relu")
            else:
                h = Fn.maxpool2(h)
            if f in self.feature_layers:
                feats.append(h.reshape(h.size(0), -1))
            f += 1
        return None, torch.cat(feats, dim=1)

    def forward(self, x):
        _require_device(x, "VGG feature stack")
        return self.run(self.normalize(x.float()))


# ------------------------------------------------------------------ model / optimiser factories
_MODELS = {"GeneratorJ": GeneratorJ, "GeneratorJ_RIC": GeneratorJ_RIC,
           "DiscriminatorN_IN": DiscriminatorN_IN, "PerceptualVGG19": PerceptualVGG19}


def build_model(model_type, args, device):
    """trainers.py:33-35."""
    return _MODELS[model_type](**args).to(device)


def build_optimizer(opt_type, model, args):
    """trainers.py:38-41 (torch.optim by name).  Adam runs as one fused multi-tensor launch."""
    args = dict(args)
    args["params"] = [p for p in model.parameters()]
    if opt_type == "Adam" and args["params"] and args["params"][0].is_cuda:
        args.setdefault("fused", True)
    return getattr(torch.optim, opt_type)(**args)


class ModelLogger:
    """trainers.py:18-30."""

    def __init__(self, log_dir, save_func):
        self.log_dir, self.save_func = log_dir, save_func

    def save(self, model, epoch, isGenerator):
        name = ("model_%05d.pth" if isGenerator else "disc_%05d.pth") % epoch
        self.save_func(model.state_dict(), os.path.join(self.log_dir, name))

    def copy_file(self, source):
        shutil.copy(source, self.log_dir)


# ------------------------------------------------------------------ patch dataset
def _rot90cw(a):
    return np.rot90(a, k=-1)


def overlap_img(img):
    """custom_transforms.py:38-52 without cv2: the image alpha-composited over its own 90-degree
    clockwise rotation (uint8 arithmetic as in the reference)."""
    img1 = np.array(img)
    img2 = np.ascontiguousarray(_rot90cw(img1))
    a1, a2 = img1[:, :, 3] / 255.0, img2[:, :, 3] / 255.0
    rgb = np.zeros_like(img1[:, :, 0:3])
    for c in range(3):
        rgb[:, :, c] = a1 * img1[:, :, c] + a2 * img2[:, :, c] * (1 - a1)
    alpha = a1 + a2 * (1 - a1)
    return Image.fromarray(np.dstack((rgb, (alpha * 255).astype("uint8"))))


def cat_img(img):
    w, h = img.size
    out = Image.new("RGBA", (w * 2, h))
    out.paste(img, (0, 0))
    out.paste(overlap_img(img), (w, 0))
    return out


def cat_mask(mask):
    m1 = np.array(mask)
    m = Image.fromarray(np.maximum(m1, np.ascontiguousarray(_rot90cw(m1))))
    w, h = mask.size
    out = Image.new("L", (w * 2, h))
    out.paste(mask, (0, 0))
    out.paste(m, (w, 0))
    return out


def white_bg(img):
    a = np.array(img).astype(np.float32)
    alpha = a[:, :, 3:4] / 255.0
    return Image.fromarray((a[:, :, 0:3] * alpha + 255 * (1 - alpha)).astype(np.uint8))


def replace_alpha(img, mask):
    a = np.array(img)
    a[:, :, 3] = np.array(mask)
    return Image.fromarray(a)


def patch_rows(mid, size, extent):
    """data.py:120-133 cut_patch along one axis: source index of each of the `size` patch
    positions, and which positions stay zero.  The window is [max(0, m - s/2),
    min(m + s/2, extent - 1)) — the last row/column is never read — and a clipped window is
    written at the START of the zero patch, not centred."""
    hs = size // 2
    lo = np.maximum(0, mid - hs)
    hi = np.minimum(mid + hs, extent - 1)
    src = lo[:, None] + np.arange(size)[None, :]
    return src, src < hi[:, None]


class DatasetPatches_M:
    """data.py:56-180.  The pre/post/mask images live on `device`; `batch(n)` draws n patch
    centres with the reference's sampling rule (centres without replacement from the dilated
    mask, an independent random centre for the discriminator's real patch) from numpy's global
    RNG, and cuts all patches of the batch on the GPU."""

    def __init__(self, data_root, pre_dir, post_dir, post_name, patch_size, use_mask=False,
                 use_pos=False, use_edge=False, device="cuda"):
        self.data_root, self.pre_dir = data_root, pre_dir
        self.post_dir, self.post_name = post_dir, post_name
        self.patch_size = patch_size
        self.use_mask, self.use_pos, self.use_edge = use_mask, use_pos, use_edge
        self.device = torch.device(device)
        self.load_image()

    @classmethod
    def from_images(cls, pre_color, post_color, mask, pre_pos=None, patch_size=32, use_mask=False,
                    use_pos=False, device="cuda"):
        """The same dataset from in-memory PIL images (no PNG round trip through
        blender_render/rest_pose): pre_color RGBA render, post_color RGB(A) target already on
        white, mask 'L'."""
        self = object.__new__(cls)
        self.data_root = self.pre_dir = self.post_dir = self.post_name = None
        self.patch_size = patch_size
        self.use_mask, self.use_pos, self.use_edge = use_mask, use_pos, False
        self.device = torch.device(device)
        self.set_images(pre_color, post_color, mask, pre_pos)
        return self

    def load_image(self, fileName="0001.png"):
        pre_color = Image.open(os.path.join(self.data_root, self.pre_dir, fileName))
        mask = pre_color.split()[-1]
        # (data.py:80-81 compares instead of assigning: the fallback name is never applied)
        post_color = Image.open(os.path.join(self.post_dir, self.post_name + ".png"))
        post_color = replace_alpha(post_color, mask)
        pre_pos = None
        if self.use_pos:
            pre_pos = Image.open(os.path.join(self.data_root, "pos", fileName))
        if self.use_edge:
            pre_edge = Image.open(os.path.join(self.data_root, "edge", fileName))
            pre_color = cat_img(overlap_edge_on_img(pre_edge, pre_color))
            mask = cat_mask(mask)
            pre_pos = cat_img(pre_pos)
            post_color = cat_img(post_color)
        post_color = white_bg(post_color)
        self.set_images(pre_color, post_color, mask, pre_pos)

    def set_images(self, pre_color, post_color, mask, pre_pos=None):
        """data.py:100-118 preprocessing_image."""
        mask_tensor = _to_tensor(mask)
        feats = [_rgb_normalised(pre_color)]
        if self.use_mask:
            feats.append(mask_tensor)
        if self.use_pos:
            feats.append(_rgb_normalised(pre_pos)[0:2])
        self.images_pre = torch.cat(feats, 0).to(self.device)
        self.images_post = _rgb_normalised(post_color).to(self.device)
        self.images_mask = mask_tensor.to(self.device)
        valid = _to_tensor(mask.filter(ImageFilter.MaxFilter(7)))
        self.valid_indices = valid.squeeze().nonzero(as_tuple=False).numpy()
        self.valid_indices_left = list(range(len(self.valid_indices)))

    def __len__(self):
        return len(self.valid_indices)      # data.py:179-180: one item per valid pixel

    def draw_midpoints(self, n):
        """The index arithmetic of data.py:147-157 for n consecutive items."""
        mids, mids_r = [], []
        for _ in range(n):
            i = np.random.randint(0, len(self.valid_indices_left))
            j = np.random.randint(0, len(self.valid_indices))
            mids.append(self.valid_indices[self.valid_indices_left[i]])
            mids_r.append(self.valid_indices[j])
            del self.valid_indices_left[i]
            if len(self.valid_indices_left) < 1:
                self.valid_indices_left = list(range(len(self.valid_indices)))
        return np.asarray(mids), np.asarray(mids_r)

    def patch_index(self, mids, H, W):
        """patch_rows for both axes, evaluated on the device: row / column gather indices
        (n, size) and the (n, size, size) mask of positions that receive image data."""
        size, hs = self.patch_size, self.patch_size // 2
        ar = torch.arange(size, device=mids.device)

        def axis(m, extent):
            lo = (m - hs).clamp_min(0)
            hi = (m + hs).clamp_max(extent - 1)
            src = lo[:, None] + ar[None, :]
            return src.clamp_max(extent - 1), src < hi[:, None]
        ry, vy = axis(mids[:, 0], H)
        rx, vx = axis(mids[:, 1], W)
        return ry, rx, vy[:, :, None] & vx[:, None, :]

    @staticmethod
    def cut(image, index):
        """cut_patch for a batch of centres: (n, C, size, size) gathered on the device."""
        ry, rx, ok = index
        patches = image[:, ry[:, :, None], rx[:, None, :]]          # (C, n, size, size)
        zero = torch.zeros((), dtype=image.dtype, device=image.device)
        return torch.where(ok.unsqueeze(0), patches, zero).permute(1, 0, 2, 3).contiguous()

    def batch(self, n):
        mids, mids_r = self.draw_midpoints(n)
        # the only host -> device traffic of an iteration: 2n centre coordinates, from pinned
        # memory and without blocking the host (a pageable copy would wait for the device to
        # drain and stop the host from queueing the next iteration)
        both = torch.from_numpy(np.concatenate([mids, mids_r]).astype(np.int64))
        if self.device.type == "cuda":
            both = both.pin_memory().to(self.device, non_blocking=True)
        _, H, W = self.images_pre.shape
        assert self.images_post.shape[1:] == (H, W) and self.images_mask.shape[1:] == (H, W)
        idx, idx_r = self.patch_index(both[:n], H, W), self.patch_index(both[n:], H, W)
        return {"pre": self.cut(self.images_pre, idx),
                "pre_mask": self.cut(self.images_mask, idx),
                "post": self.cut(self.images_post, idx),
                "already": self.cut(self.images_post, idx_r),
                "already_mask": self.cut(self.images_mask, idx_r)}

    def batches(self, batch_size):
        """One epoch: len(self) // batch_size batches (DataLoader(drop_last=True))."""
        for _ in range(len(self) // batch_size):
            yield self.batch(batch_size)


# ------------------------------------------------------------------ Trainer
class Trainer:
    """trainers.py:44-244."""

    def __init__(self, data_root, trainer_config, opt_discriminator, opt_generator, model_logger,
                 perception_loss_model, perception_loss_weight, use_mask, use_pos, use_edge,
                 device, dataset=None):
        self.device = device
        self.dataset = dataset if dataset is not None else DatasetPatches_M(
            os.path.join(data_root, "rest_pose"), trainer_config["pre_dir"],
            trainer_config["post_dir"], trainer_config["post_name"],
            trainer_config["patch_size"], use_mask, use_pos, use_edge, device)
        self.batch_size = trainer_config["batch_size"]
        self.opt_discriminator, self.opt_generator = opt_discriminator, opt_generator
        self.reconstruction_criterion = _CRITERIA[trainer_config["reconstruction_criterion"]]()
        self.adversarial_criterion = _CRITERIA[trainer_config["adversarial_criterion"]]()
        self.reconstruction_weight = trainer_config["reconstruction_weight"]
        self.adversarial_weight = trainer_config["adversarial_weight"]
        self.model_logger = model_logger
        self.training_log = {}
        self.log_interval = trainer_config["log_interval"]
        self.perception_loss_weight = perception_loss_weight
        self.perception_loss_model = perception_loss_model
        self.use_adversarial_loss = False
        self.use_image_loss = trainer_config["use_image_loss"]
        self.data_root = data_root
        self.testing_name_list = trainer_config.get("testing_name_list", [])
        self.use_mask, self.use_pos, self.use_edge = use_mask, use_pos, use_edge
        self.pre_dir = trainer_config["pre_dir"]

    # -- the reference's helper surface
    def run_discriminator(self, discriminator, images):
        return discriminator(images)

    def apply_mask(self, x, batch, mask_key):
        if mask_key in batch:
            return x * batch[mask_key].expand(x.size())
        return x

    def ones_like(self, x):
        return torch.ones_like(x)

    def zeros_like(self, x):
        return torch.zeros_like(x)

    def compute_discriminator_loss(self, generator, discriminator, batch, generated=None):
        """trainers.py:87-101.  `generated`: an already evaluated generator(batch['pre'])."""
        if generated is None:
            generated = generator(batch["pre"])
        fake = self.apply_mask(generated, batch, "pre_mask")
        fake_labels, _ = self.run_discriminator(discriminator, fake.detach())
        true = self.apply_mask(batch["already"], batch, "already_mask")
        true_labels, _ = self.run_discriminator(discriminator, true)
        return self.adversarial_criterion(fake_labels, 0.0) + \
            self.adversarial_criterion(true_labels, 1.0)

    def compute_generator_loss(self, generator, discriminator, batch, use_gan, use_mask,
                               generated=None):
        """trainers.py:103-137."""
        image_loss = perception_loss = adversarial_loss = 0
        if generated is None:
            generated = generator(batch["pre"])
        if use_mask:
            generated = generated * batch["mask"]
            batch["post"] = batch["post"] * batch["mask"]
        if self.use_image_loss:
            post = batch["post"]
            if generated.shape[2:] != post.shape[2:]:
                if (post.shape[2] - generated.shape[2]) % 2 != 0:
                    raise RuntimeError("post and generated heights must differ by an even number")
                if generated.shape[2] != generated.shape[3] or post.shape[2] != post.shape[3]:
                    raise RuntimeError("square patches are expected")
                bnd = (post.shape[2] - generated.shape[2]) // 2
                post = post[:, :, bnd:-bnd, bnd:-bnd]
            image_loss = self.reconstruction_criterion(generated, post)
        if self.perception_loss_model is not None:
            _, fake_features = self.perception_loss_model(generated)
            with torch.no_grad():
                _, target_features = self.perception_loss_model(batch["post"])
            perception_loss = Fn.mse_loss(fake_features, target_features)
        if self.use_adversarial_loss and use_gan:
            fake = self.apply_mask(generated, batch, "pre_mask")
            # the reference lets this backward also deposit gradients on the discriminator's
            # parameters; they are cleared before they are ever used (trainers.py:154), so the
            # discriminator is frozen for this pass instead
            frozen = [p for p in discriminator.parameters() if p.requires_grad]
            for p in frozen:
                p.requires_grad_(False)
            try:
                fake_labels, _ = self.run_discriminator(discriminator, fake)
            finally:
                for p in frozen:
                    p.requires_grad_(True)
            adversarial_loss = self.adversarial_criterion(fake_labels, 1.0)
        return image_loss, perception_loss, adversarial_loss, generated

    def train_step(self, generator, discriminator, batch):
        """One iteration of trainers.py:148-172; returns the scalar log entries (0-dim device
        tensors, no host sync)."""
        generator.train()
        generator.stat_updates = 1
        log = {}
        generated = None
        if self.use_adversarial_loss:
            discriminator.train()
            generator.stat_updates = 2          # one forward stands for the reference's two
            generated = generator(batch["pre"])
            self.opt_discriminator.zero_grad()
            discriminator_loss = self.compute_discriminator_loss(generator, discriminator, batch,
                                                                 generated)
            discriminator_loss.backward()
            self.opt_discriminator.step()
            log["discriminator_loss"] = discriminator_loss.detach()
        self.opt_generator.zero_grad()
        g_image_loss, g_perc_loss, g_adv_loss, _ = self.compute_generator_loss(
            generator, discriminator, batch, use_gan=True, use_mask=False, generated=generated)
        generator_loss = self.reconstruction_weight * g_image_loss + \
            self.perception_loss_weight * g_perc_loss + self.adversarial_weight * g_adv_loss
        generator_loss.backward()
        self.opt_generator.step()
        for key, value in (("g_image_loss", g_image_loss), ("g_perc_loss", g_perc_loss),
                           ("g_adv_loss", g_adv_loss), ("generator_loss", generator_loss)):
            if torch.is_tensor(value):
                log[key] = value.detach()
        return log

    def train(self, generator, discriminator, epochs, result_folder, starting_batch_num):
        self.use_adversarial_loss = discriminator is not None
        batch_num, save_num = starting_batch_num, 0
        start = time.time()
        for _ in range(epochs):
            np.random.seed()
            for batch in self.dataset.batches(self.batch_size):
                self.add_log(self.train_step(generator, discriminator, batch))
                batch_num += 1
                if batch_num % self.log_interval == 0 or batch_num == 1:
                    eval_start = time.time()
                    generator.eval()
                    self.test_on_full_image(generator, result_folder)
                    self.flush_scalar_log(batch_num, time.time() - start)
                    self.model_logger.save(generator, save_num, True)
                    save_num += 1
                    print(f"Eval of batch: {batch_num} took {time.time() - eval_start}", flush=True)
        self.model_logger.save(generator, 99999, True)

    def add_log(self, log):
        for k, v in log.items():
            self.training_log[k] = self.training_log[k] + v if k in self.training_log else v

    def flush_scalar_log(self, batch_num, took):
        line = "[%d]" % batch_num
        for key in sorted(self.training_log.keys()):
            line += " [%s] % 7.4f" % (key, float(self.training_log[key]) / self.log_interval)
        print(line + ". Took {}".format(took), flush=True)
        self.training_log = {}

    def test_on_full_image(self, generator, result_folder, save_alpha=True):
        """trainers.py:214-232."""
        for test_name in self.testing_name_list:
            data_root = os.path.join(self.data_root, test_name)
            dataset = DatasetFullImages(data_root, self.pre_dir, self.use_mask, self.use_pos,
                                        self.use_edge)
            out_dir = os.path.join(data_root, result_folder)
            os.makedirs(out_dir, exist_ok=True)
            with torch.no_grad():
                for i in range(len(dataset)):
                    item = dataset[i]
                    out = generator(item["pre"][None].to(self.device))[0]
                    img = to_image_space(out.cpu().numpy()).transpose(1, 2, 0)
                    if save_alpha:
                        alpha = (item["pre_mask"].numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
                        img = np.concatenate((img, alpha), 2)
                    Image.fromarray(img).save(os.path.join(out_dir, item["file_name"]))
