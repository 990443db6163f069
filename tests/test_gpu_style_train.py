"""Training kernels of the style translator against torch autograd on the CPU
(float64), the deformable-convolution oracle, and the fixture produced by the reference's own
Trainer code (tests/golden/make_style_train_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from drawingspinup_amd import ops
from drawingspinup_amd.style import functions as Fn
from drawingspinup_amd.style import training as T
from oracle import style_ref as sr

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "style_train_reference.npz"))


def _rand(shape, seed, s=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * s


def _close(got, ref, rtol=2e-4, atol=None):
    ref = ref.double()
    got = got.detach().cpu().double()
    if atol is None:
        atol = rtol * float(ref.abs().max()) + 1e-12
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


# (B, C, H, W, O, k, stride, pad)
CONV_CASES = [(3, 5, 9, 11, 7, 3, 1, 1),       # ragged everything, O < 32
              (40, 64, 8, 8, 128, 3, 1, 1),    # images smaller than a 128-pixel chunk
              (4, 24, 32, 32, 64, 3, 2, 1),    # stride 2 (GeneratorJ conv1/conv2)
              (2, 13, 32, 32, 16, 7, 1, 3),    # 7x7 (GeneratorJ conv0 / conv_11)
              (4, 64, 16, 16, 3, 1, 1, 0),     # conv_12
              (5, 3, 32, 32, 12, 4, 2, 1),     # discriminator conv0
              (5, 24, 8, 8, 48, 4, 1, 1),      # discriminator conv_2
              (2, 70, 20, 12, 96, 3, 1, 1)]


@pytest.mark.parametrize("B,C,H,W,O,k,s,p", CONV_CASES)
def test_conv_backward_vs_autograd(dev, B, C, H, W, O, k, s, p):
    x = _rand((B, C, H, W), 1).double().requires_grad_()
    w = _rand((O, C, k, k), 2, 0.2).double().requires_grad_()
    b = _rand((O,), 3).double().requires_grad_()
    y = F.conv2d(x, w, b, s, p)
    dy = _rand(tuple(y.shape), 4)
    y.backward(dy.double())
    xd, wd, dyd = x.detach().float().to(dev), w.detach().float().to(dev), dy.to(dev)
    _close(ops.conv2d_wgrad(xd, dyd, k, s, p), w.grad)
    _close(ops.conv2d_dgrad(dyd, wd, (H, W), s, p), x.grad)
    _close(ops.channel_sum(dyd), b.grad)
    # accumulate flag
    base = _rand((O, C, k, k), 5).to(dev)
    out = base.clone()
    ops.conv2d_wgrad(xd, dyd, k, s, p, out=out, accumulate=True)
    _close(out, w.grad + base.cpu().double())


@pytest.mark.parametrize("B,C,H,W,O", [(3, 6, 32, 32, 32), (40, 128, 8, 8, 128), (2, 19, 12, 20, 70),
                                       (4, 166, 16, 16, 64)])
def test_deform_backward_vs_oracle(dev, B, C, H, W, O):
    off = sr.generate_coordinates(H, W)
    x = _rand((B, C, H, W), 1).double().requires_grad_()
    w = _rand((O, C, 3, 3), 2, 0.2).double().requires_grad_()
    y = sr.deform_conv2d(x, off[None].expand(B, -1, -1, -1), w)
    dy = _rand(tuple(y.shape), 4)
    y.backward(dy.double())
    plan = ops.deform_plan(off.to(dev))
    xd, wd, dyd = x.detach().float().to(dev), w.detach().float().to(dev), dy.to(dev)
    _close(ops.conv2d_wgrad(xd, dyd, 3, 1, 1, plan), w.grad)
    _close(ops.deform_conv3x3_dgrad(dyd, wd, plan), x.grad)
    # random (non-RIC) offsets that leave the image exercise the zeroed corners
    off2 = _rand((18, H, W), 9, 1.7)
    x.grad = w.grad = None
    sr.deform_conv2d(x, off2[None].expand(B, -1, -1, -1), w).backward(dy.double())
    plan2 = ops.DeformPlan(off2.to(dev))
    _close(ops.conv2d_wgrad(xd, dyd, 3, 1, 1, plan2), w.grad)
    _close(ops.deform_conv3x3_dgrad(dyd, wd, plan2), x.grad)


def test_wgrad_is_deterministic(dev):
    x, dy = _rand((40, 64, 16, 16), 1).to(dev), _rand((40, 96, 16, 16), 2).to(dev)
    a = ops.conv2d_wgrad(x, dy, 3, 1, 1)
    b = ops.conv2d_wgrad(x, dy, 3, 1, 1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("act", [None, "relu", "leaky_relu"])
@pytest.mark.parametrize("B,C,H,W", [(40, 24, 8, 8), (3, 5, 7, 9)])
def test_batchnorm_train(dev, act, B, C, H, W):
    x = _rand((B, C, H, W), 1, 2.0).double().requires_grad_()
    g = (_rand((C,), 2) * 0.5 + 1).double().requires_grad_()
    b = _rand((C,), 3, 0.3).double().requires_grad_()
    rm, rv = _rand((C,), 4, 0.1).double(), (_rand((C,), 5).abs() + 0.5).double()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    for _ in range(2):                                    # stat_updates = 2
        y = F.batch_norm(x, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)
    y = {None: lambda t: t, "relu": F.relu, "leaky_relu": lambda t: F.leaky_relu(t, 0.2)}[act](y)
    dy = _rand(tuple(y.shape), 6)
    y.backward(dy.double())
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(g.float()); bn.bias.copy_(b.float())
        bn.running_mean.copy_(rm.float()); bn.running_var.copy_(rv.float())
    xd = x.detach().float().to(dev).requires_grad_()
    yd = Fn.batch_norm_train(xd, bn, act, stat_updates=2)
    yd.backward(dy.to(dev))
    _close(yd, y.detach(), atol=2e-5)
    _close(xd.grad, x.grad, rtol=1e-3)
    _close(bn.weight.grad, g.grad, rtol=1e-3)
    _close(bn.bias.grad, b.grad, rtol=1e-3)
    _close(bn.running_mean, rm_ref, atol=1e-5)
    _close(bn.running_var, rv_ref, atol=1e-5)
    assert int(bn.num_batches_tracked) == 2


@pytest.mark.parametrize("act", [None, "leaky_relu"])
def test_instance_norm(dev, act):
    x = _rand((5, 24, 8, 8), 1, 2.0).double().requires_grad_()
    y = F.instance_norm(x, eps=1e-5)
    if act:
        y = F.leaky_relu(y, 0.2)
    dy = _rand(tuple(y.shape), 2)
    y.backward(dy.double())
    xd = x.detach().float().to(dev).requires_grad_()
    yd = Fn.instance_norm(xd, act)
    yd.backward(dy.to(dev))
    _close(yd, y.detach(), atol=2e-5)
    _close(xd.grad, x.grad, rtol=1e-3)


def test_pool_resample_activation_loss(dev):
    x = _rand((3, 5, 8, 12), 1).double().requires_grad_()
    dy = _rand((3, 5, 4, 6), 2)
    y = F.max_pool2d(x, 2, 2); y.backward(dy.double())
    xd = x.detach().float().to(dev).requires_grad_()
    yd = Fn.maxpool2(xd); yd.backward(dy.to(dev))
    _close(yd, y.detach(), atol=0); _close(xd.grad, x.grad, atol=0)
    x.grad = None
    dy = _rand((3, 5, 16, 24), 3)
    y = F.interpolate(x, scale_factor=2); y.backward(dy.double())
    xd = x.detach().float().to(dev).requires_grad_()
    yd = Fn.upsample2(xd); yd.backward(dy.to(dev))
    _close(yd, y.detach(), atol=0); _close(xd.grad, x.grad, atol=1e-6)
    for act, f in (("relu", F.relu), ("leaky_relu", lambda t: F.leaky_relu(t, 0.2)),
                   ("tanh", torch.tanh)):
        x.grad = None
        dy = _rand(tuple(x.shape), 4)
        y = f(x); y.backward(dy.double())
        xd = x.detach().float().to(dev).requires_grad_()
        yd = Fn.activation(xd, act); yd.backward(dy.to(dev))
        _close(yd, y.detach(), atol=1e-6); _close(xd.grad, x.grad, atol=1e-5)
    t = _rand(tuple(x.shape), 5).double()
    for kind, f in (("l1", F.l1_loss), ("mse", F.mse_loss)):
        x.grad = None
        loss = f(x, t); (loss * 3.0).backward()
        xd = x.detach().float().to(dev).requires_grad_()
        ld = (Fn.l1_loss if kind == "l1" else Fn.mse_loss)(xd, t.float().to(dev))
        (ld * 3.0).backward()
        _close(ld, loss.detach(), atol=1e-6); _close(xd.grad, x.grad, atol=1e-7)
    x.grad = None
    loss = F.mse_loss(x, torch.ones_like(x)); loss.backward()
    xd = x.detach().float().to(dev).requires_grad_()
    ld = Fn.mse_loss(xd, 1.0); ld.backward()
    _close(ld, loss.detach(), atol=1e-6); _close(xd.grad, x.grad, atol=1e-7)


# ------------------------------------------------------------------ the reference's loop body
G_ARGS = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
              filters=[8, 16, 24, 24, 24, 16], input_channels=6)
OPT = dict(lr=0.0004, betas=[0.9, 0.999], weight_decay=0.00001)


def _setup(name, dev):
    pre = name + "."
    gen = T.build_model(name, dict(G_ARGS), dev)
    gen.load_state_dict({k[len(pre) + 3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files
                         if k.startswith(pre + "g0.")})
    disc = T.build_model("DiscriminatorN_IN", dict(num_filters=4, n_layers=2), dev)
    disc.load_state_dict({k[len(pre) + 3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files
                          if k.startswith(pre + "d0.")})
    perc = T.build_model("PerceptualVGG19", dict(feature_layers=[0, 3, 5],
                                                 use_normalization=False, random_init=True), dev)
    sd = perc.state_dict()
    for f in (0, 2, 5):
        sd[f"model.features.{f}.weight"] = torch.from_numpy(GOLD[f"vgg.features.{f}.weight"])
        sd[f"model.features.{f}.bias"] = torch.from_numpy(GOLD[f"vgg.features.{f}.bias"])
    perc.load_state_dict(sd)
    cfg = dict(batch_size=4, reconstruction_criterion="L1Loss", adversarial_criterion="MSELoss",
               reconstruction_weight=4.0, adversarial_weight=0.5, log_interval=1000,
               use_image_loss=True, pre_dir="color", patch_size=32)
    tr = T.Trainer(None, cfg, T.build_optimizer("Adam", disc, OPT),
                   T.build_optimizer("Adam", gen, OPT), None, perc, 6.0, True, True, False, dev,
                   dataset=object())
    tr.use_adversarial_loss = True
    return gen, disc, tr


@pytest.mark.parametrize("name", ["GeneratorJ_RIC", "GeneratorJ"])
def test_training_iterations_match_reference(dev, name):
    pre = name + "."
    gen, disc, tr = _setup(name, dev)
    for it in range(2):
        batch = {k: torch.from_numpy(GOLD[pre + f"it{it}.batch.{k}"]).to(dev)
                 for k in ("pre", "pre_mask", "post", "already", "already_mask")}
        if it == 0:
            # generator output in train mode + every gradient of the first iteration
            gen.train(); gen.stat_updates = 0
            saved = {k: v.clone() for k, v in gen.state_dict().items()}
            out = gen(batch["pre"])
            _close(out, torch.from_numpy(GOLD[pre + "it0.generated"]), atol=2e-5)
            gen.load_state_dict(saved)          # undo num_batches_tracked side effects
            tr.opt_generator.zero_grad()
        log = tr.train_step(gen, disc, batch)
        want = GOLD[pre + f"it{it}.losses"]
        got = [float(log[k]) for k in ("discriminator_loss", "g_image_loss", "g_perc_loss",
                                       "g_adv_loss", "generator_loss")]
        np.testing.assert_allclose(got, want, rtol=2e-4, err_msg=f"losses of iteration {it}")
        if it == 0:
            for k, p in gen.named_parameters():
                key = pre + "it0.ggrad." + k
                if key in GOLD.files:
                    _close(p.grad, torch.from_numpy(GOLD[key]), rtol=2e-3)
                else:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    # after two Adam steps: BatchNorm buffers tightly, parameters within a fraction of the
    # 2 * lr a parameter can move (Adam's first steps are +-lr whatever the gradient's size, so
    # elements whose true gradient is ~0 follow rounding noise)
    sd = gen.state_dict()
    for k in GOLD.files:
        if not k.startswith(pre + "g2."):
            continue
        name_k = k[len(pre) + 3:]
        got, want = sd[name_k].cpu().double(), torch.from_numpy(GOLD[k]).double()
        if "running_" in name_k:
            _close(got, want, rtol=1e-3)
        elif "num_batches" in name_k:
            assert int(got) == int(want), name_k
        else:
            d = (got - want).abs()
            assert float(d.mean()) < 4e-5 and float(d.max()) < 8.5e-4, (name_k, float(d.mean()),
                                                                      float(d.max()))
    dsd = disc.state_dict()
    for k in GOLD.files:
        if k.startswith(pre + "d2."):
            d = (dsd[k[len(pre) + 3:]].cpu().double() - torch.from_numpy(GOLD[k]).double()).abs()
            assert float(d.max()) < 8.5e-4, (k, float(d.max()))


def test_discriminator_gradients_match_reference(dev):
    pre = "GeneratorJ."
    gen, disc, tr = _setup("GeneratorJ", dev)
    batch = {k: torch.from_numpy(GOLD[pre + f"it0.batch.{k}"]).to(dev)
             for k in ("pre", "pre_mask", "post", "already", "already_mask")}
    gen.train(); disc.train()
    loss = tr.compute_discriminator_loss(gen, disc, batch)
    loss.backward()
    assert abs(float(loss) - GOLD[pre + "it0.losses"][0]) < 2e-4
    for k, p in disc.named_parameters():
        want = torch.from_numpy(GOLD[pre + "it0.dgrad." + k])
        # biases in front of an InstanceNorm have a zero true gradient: compare absolutely
        _close(p.grad, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()) + 1e-6)
