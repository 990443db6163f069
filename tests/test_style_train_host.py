"""CPU tests of the style-training row: the patch dataset against the
reference's DatasetPatches_M (fixture made by tests/golden/make_style_train_golden.py), the
transposed sampling table, module layouts, and the no-CPU-fallback guards."""
import os
import tempfile

import numpy as np
import pytest
import torch
from PIL import Image

from drawingspinup_amd import ops
from drawingspinup_amd.style import training as T
from oracle import style_ref as sr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "style_train_reference.npz"))


def _dataset(edge):
    pre = f"dataset.edge{edge}."
    d = tempfile.mkdtemp()
    root = os.path.join(d, "rest_pose")
    for sub in ("color", "pos", "edge"):
        os.makedirs(os.path.join(root, sub))
    os.makedirs(os.path.join(d, "char"))
    Image.fromarray(GOLD[pre + "color"]).save(os.path.join(root, "color", "0001.png"))
    Image.fromarray(GOLD[pre + "pos"]).save(os.path.join(root, "pos", "0001.png"))
    Image.fromarray(GOLD[pre + "edge"]).save(os.path.join(root, "edge", "0001.png"))
    Image.fromarray(GOLD[pre + "post"]).save(os.path.join(d, "char", "tex.png"))
    return T.DatasetPatches_M(root, "color", os.path.join(d, "char"), "tex", 32, use_mask=True,
                              use_pos=True, use_edge=bool(edge), device="cpu")


@pytest.mark.parametrize("edge,seed", [(0, 5), (1, 6)])
def test_patch_dataset_matches_reference(edge, seed):
    pre = f"dataset.edge{edge}."
    ds = _dataset(edge)
    assert len(ds) == int(GOLD[pre + "len"])
    assert np.array_equal(ds.images_pre.numpy(), GOLD[pre + "images_pre"])
    assert np.array_equal(ds.images_post.numpy(), GOLD[pre + "images_post"])
    assert np.array_equal(ds.images_mask.numpy(), GOLD[pre + "images_mask"])
    np.random.seed(seed)
    b = ds.batch(6)
    for k in ("pre", "pre_mask", "post", "already", "already_mask"):
        assert np.array_equal(b[k].numpy(), GOLD[pre + "items." + k]), k


def test_patch_rows_clipping_rule():
    # centre near the top: window starts at 0; near the bottom: the last row is never read and
    # the clipped window is written at the start of the patch
    src, ok = T.patch_rows(np.array([3, 50, 94]), 32, 96)
    assert src[0, 0] == 0 and ok[0].sum() == 19            # rows 0..18
    assert src[1, 0] == 34 and ok[1].all()
    assert src[2, 0] == 78 and ok[2].sum() == 95 - 78      # rows 78..94


def test_sampling_order_without_replacement():
    ds = _dataset(0)
    np.random.seed(0)
    n = len(ds)
    mids, _ = ds.draw_midpoints(n)                          # one full pass: every pixel once
    assert len({tuple(m) for m in mids}) == n
    assert len(ds.valid_indices_left) == n                  # refilled


def _tap_table_numpy(offset):
    """numpy restatement of deform_tap_table_kernel / make_tap (f32 arithmetic)."""
    off = offset.numpy().astype(np.float32)
    _, H, W = off.shape
    rec = np.zeros((H * W * 9, 8), np.int32)
    f = np.float32
    for pix in range(H * W):
        oy, ox = divmod(pix, W)
        for t in range(9):
            h = f(oy - 1 + t // 3) + off[2 * t, oy, ox]
            w = f(ox - 1 + t % 3) + off[2 * t + 1, oy, ox]
            inside = h > -1 and w > -1 and h < H and w < W
            hl, wl = np.floor(h), np.floor(w)
            h0, w0 = int(hl), int(wl)
            h1, w1 = h0 + 1, w0 + 1
            lh, lw = f(h - hl), f(w - wl)
            hh, hw = f(1) - lh, f(1) - lw
            vh0, vh1 = inside and h0 >= 0, inside and h1 <= H - 1
            vw0, vw1 = w0 >= 0, w1 <= W - 1
            ws = [hh * hw if vh0 and vw0 else 0, hh * lw if vh0 and vw1 else 0,
                  lh * hw if vh1 and vw0 else 0, lh * lw if vh1 and vw1 else 0]
            r0, r1 = min(max(h0, 0), H - 1) * W, min(max(h1, 0), H - 1) * W
            c0, c1 = min(max(w0, 0), W - 1), min(max(w1, 0), W - 1)
            rec[pix * 9 + t, :4] = [r0 + c0, r0 + c1, r1 + c0, r1 + c1]
            rec[pix * 9 + t, 4:] = np.array(ws, np.float32).view(np.int32)
    return rec


def test_transposed_sampling_table_is_the_transpose():
    H, W = 8, 12
    off = sr.generate_coordinates(H, W)
    rec = _tap_table_numpy(off)
    npix = H * W
    rowptr, src, wgt = ops.transpose_tap_table(rec.tobytes(), npix)
    # dense S: rows (tap, out pixel), columns input pixel
    S = np.zeros((9 * npix, npix), np.float64)
    for r in range(npix * 9):
        pix, t = divmod(r, 9)
        for k in range(4):
            S[t * npix + pix, rec[r, k]] += rec[r, 4 + k:5 + k].view(np.float32)[0]
    St = np.zeros((npix, 9 * npix), np.float64)
    for q in range(npix):
        for e in range(rowptr[q], rowptr[q + 1]):
            St[q, src[e]] += wgt[e]
    assert np.array_equal(St, S.T)
    assert rowptr[-1] == len(src) == len(wgt) == int((rec[:, 4:].view(np.float32) != 0).sum())
    # the table reproduces the oracle's deformable convolution (as a linear operator)
    x = torch.randn(1, 2, H, W, generator=torch.Generator().manual_seed(0))
    w = torch.randn(3, 2, 3, 3, generator=torch.Generator().manual_seed(1))
    col = (S @ x.double().reshape(2, npix).T.numpy()).reshape(9, npix, 2)       # (tap, pix, c)
    out = np.einsum("ock,kpc->op", w.double().reshape(3, 2, 9).numpy(), col).reshape(1, 3, H, W)
    ref = sr.deform_conv2d(x, off[None], w).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)


def test_module_layouts_match_reference():
    d = T.DiscriminatorN_IN(num_filters=4, n_layers=2)
    want = {k[len("GeneratorJ.d0."):]: GOLD[k].shape for k in GOLD.files
            if k.startswith("GeneratorJ.d0.")}
    got = {k: tuple(v.shape) for k, v in d.state_dict().items()}
    assert got == want
    v = T.PerceptualVGG19(feature_layers=[0, 3, 5], use_normalization=False, random_init=True)
    keys = set(v.state_dict().keys())
    for f in (0, 2, 5):
        assert f"model.features.{f}.weight" in keys
        assert tuple(v.state_dict()[f"model.features.{f}.weight"].shape) == \
            GOLD[f"vgg.features.{f}.weight"].shape
    assert not any(p.requires_grad for p in v.parameters())
    for name in ("GeneratorJ", "GeneratorJ_RIC"):
        g = T.build_model(name, dict(use_bias=False, tanh=True, append_smoothers=True,
                                     resnet_blocks=2, filters=[8, 16, 24, 24, 24, 16],
                                     input_channels=6), "cpu")
        want = {k[len(name) + 4:]: GOLD[k].shape for k in GOLD.files if k.startswith(name + ".g0.")}
        assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == want


def test_no_cpu_fallback():
    d = T.DiscriminatorN_IN(num_filters=4, n_layers=2)
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 3, 32, 32))
    v = T.PerceptualVGG19(feature_layers=[0, 3, 5], use_normalization=False, random_init=True)
    with pytest.raises(RuntimeError):
        v(torch.zeros(1, 3, 32, 32))
    g = T.build_model("GeneratorJ", dict(resnet_blocks=1, input_channels=6), "cpu").train()
    with pytest.raises(RuntimeError):
        g(torch.zeros(1, 6, 32, 32))


def test_default_job_is_the_shipped_config():
    from drawingspinup_amd.entry._train_stage import default_job
    j1, j2 = default_job(1), default_job(2)
    assert j1["generator"]["type"] == "GeneratorJ_RIC" and j2["generator"]["type"] == "GeneratorJ"
    assert j1["trainer"]["epochs"] == 3 and j2["trainer"]["epochs"] == 2
    assert j1["trainer"]["batch_size"] == 40 and j1["trainer"]["patch_size"] == 32
    assert j1["opt_generator"]["args"] == {"lr": 0.0004, "betas": [0.9, 0.999],
                                           "weight_decay": 0.00001}
    assert j1["perception_loss"]["weight"] == 6.0
    assert j2["trainer"]["pre_dir"] == "res_stage1_mask_pos"


# ------------------------------------------------------------------ graph wiring on the CPU
# The modules and the Trainer are evaluated with torch stand-ins for the HIP kernels (TEST
# INFRASTRUCTURE: the product has no such path) so that the wiring — layer order, skip
# connections, the dead smoother branch, the single shared generator forward with doubled
# BatchNorm updates, loss composition, frozen discriminator, both Adam steps — is pinned
# against the reference's own trainer fixture on every CPU run, independently of the kernels.
from drawingspinup_amd.style import functions as Fn
from drawingspinup_amd.style import generators as G
from oracle import style_train_ref as tref


@pytest.fixture
def cpu_kernels(monkeypatch):
    for name in ("conv", "batch_norm_train", "instance_norm", "activation", "maxpool2",
                 "upsample2", "l1_loss", "mse_loss"):
        monkeypatch.setattr(Fn, name, getattr(tref, name))
    monkeypatch.setattr(G.ops, "deform_plan", tref.deform_plan)
    monkeypatch.setattr(G._GeneratorBase, "_check", lambda self, x: None)
    monkeypatch.setattr(T, "_require_device", lambda x, what: None)


G_ARGS = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
              filters=[8, 16, 24, 24, 24, 16], input_channels=6)
OPT = dict(lr=0.0004, betas=[0.9, 0.999], weight_decay=0.00001)


def _setup(name):
    pre = name + "."
    gen = T.build_model(name, dict(G_ARGS), "cpu")
    gen.load_state_dict({k[len(pre) + 3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files
                         if k.startswith(pre + "g0.")})
    disc = T.build_model("DiscriminatorN_IN", dict(num_filters=4, n_layers=2), "cpu")
    disc.load_state_dict({k[len(pre) + 3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files
                          if k.startswith(pre + "d0.")})
    perc = T.build_model("PerceptualVGG19", dict(feature_layers=[0, 3, 5],
                                                 use_normalization=False, random_init=True), "cpu")
    sd = perc.state_dict()
    for f in (0, 2, 5):
        sd[f"model.features.{f}.weight"] = torch.from_numpy(GOLD[f"vgg.features.{f}.weight"])
        sd[f"model.features.{f}.bias"] = torch.from_numpy(GOLD[f"vgg.features.{f}.bias"])
    perc.load_state_dict(sd)
    cfg = dict(batch_size=4, reconstruction_criterion="L1Loss", adversarial_criterion="MSELoss",
               reconstruction_weight=4.0, adversarial_weight=0.5, log_interval=2,
               use_image_loss=True, pre_dir="color", patch_size=32)
    tr = T.Trainer(None, cfg, T.build_optimizer("Adam", disc, OPT),
                   T.build_optimizer("Adam", gen, OPT), None, perc, 6.0, True, True, False, "cpu",
                   dataset=object())
    tr.use_adversarial_loss = True
    return gen, disc, tr


@pytest.mark.parametrize("name", ["GeneratorJ_RIC", "GeneratorJ"])
def test_trainer_wiring_matches_reference_on_cpu(cpu_kernels, name):
    pre = name + "."
    gen, disc, tr = _setup(name)
    for it in range(2):
        batch = {k: torch.from_numpy(GOLD[pre + f"it{it}.batch.{k}"])
                 for k in ("pre", "pre_mask", "post", "already", "already_mask")}
        log = tr.train_step(gen, disc, batch)
        got = [float(log[k]) for k in ("discriminator_loss", "g_image_loss", "g_perc_loss",
                                       "g_adv_loss", "generator_loss")]
        np.testing.assert_allclose(got, GOLD[pre + f"it{it}.losses"], rtol=2e-5)
        if it == 0:
            for k, p in gen.named_parameters():
                key = pre + "it0.ggrad." + k
                if key in GOLD.files:
                    want = torch.from_numpy(GOLD[key])
                    torch.testing.assert_close(p.grad, want, rtol=1e-3,
                                               atol=1e-4 * float(want.abs().max()) + 1e-9)
                else:                                   # the dead smoother branch (models.py:347)
                    assert p.grad is None, k
            # the discriminator was frozen only for the generator's backward
            assert all(p.requires_grad for p in disc.parameters())
    sd = gen.state_dict()
    for k in GOLD.files:
        if k.startswith(pre + "g2.") and ("running_" in k or "num_batches" in k):
            torch.testing.assert_close(sd[k[len(pre) + 3:]], torch.from_numpy(GOLD[k]),
                                       rtol=1e-4, atol=1e-6)
    assert gen.stat_updates == 2


def test_trainer_loop_bookkeeping(cpu_kernels, tmp_path, capsys):
    """Trainer.train: epochs x (len // batch) iterations, evaluation + checkpoint at iteration 1
    and every log_interval, final checkpoint 99999 (trainers.py:140-192)."""
    gen, disc, tr = _setup("GeneratorJ")
    ds = _dataset(0)
    ds.valid_indices = ds.valid_indices[:13]                 # 13 pixels -> 3 batches of 4
    ds.valid_indices_left = list(range(13))
    tr.dataset = ds
    tr.model_logger = T.ModelLogger(str(tmp_path), torch.save)
    tr.testing_name_list = []
    steps = []
    orig = tr.train_step
    tr.train_step = lambda g, d, b: (steps.append(b["pre"].shape), orig(g, d, b))[1]
    tr.train(gen, disc, 2, "res_x", 0)
    assert len(steps) == 2 * 3 and steps[0] == (4, 6, 32, 32)
    saved = sorted(os.listdir(tmp_path))
    # saves at batch 1, 2, 4, 6 (log_interval = 2) + the final one
    assert saved == ["model_00000.pth", "model_00001.pth", "model_00002.pth", "model_00003.pth",
                     "model_99999.pth"]
    out = capsys.readouterr().out
    assert "[1] " in out and "[generator_loss]" in out and "[6] " in out
    sd = torch.load(os.path.join(tmp_path, "model_99999.pth"))
    assert set(sd.keys()) == set(gen.state_dict().keys())


def test_eval_after_training_refolds_batchnorm():
    """The folded eval-BatchNorm constants must not survive a training phase: running statistics
    are written through raw pointers (no tensor version bump)."""
    g = T.build_model("GeneratorJ", dict(resnet_blocks=1, input_channels=6), "cpu").eval()
    bn = g.conv0.normalization
    scale0, shift0 = G._bn_fold(bn)
    assert hasattr(bn, "_dsu_fold")
    g.train()
    bn.running_var.detach().numpy()[:] = 4.0          # in place, version counter untouched
    bn.running_mean.detach().numpy()[:] = 1.0
    g.eval()
    scale1, shift1 = G._bn_fold(bn)
    want = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    assert torch.allclose(scale1, want) and not torch.allclose(scale1, scale0)
    assert torch.allclose(shift1, bn.bias - bn.running_mean * want)


def test_perceptual_vgg_refuses_to_run_on_random_features_silently(tmp_path, monkeypatch):
    """models.py:497 loads the ImageNet weights; without a weights file the product must fail
    loudly unless random features are asked for by name."""
    import pytest
    monkeypatch.delenv("DSU_VGG19_WEIGHTS", raising=False)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        T.PerceptualVGG19(feature_layers=[0, 3, 5])
    v = T.PerceptualVGG19(feature_layers=[0, 3, 5], random_init=True)
    # a weights file in the torchvision layout is picked up from DSU_VGG19_WEIGHTS
    sd = {"features." + k: t for k, t in v.model.features.state_dict().items()}
    sd["classifier.0.weight"] = torch.zeros(1)
    path = tmp_path / "vgg19-test.pth"
    torch.save(sd, path)
    monkeypatch.setenv("DSU_VGG19_WEIGHTS", str(path))
    w = T.PerceptualVGG19(feature_layers=[0, 3, 5])
    assert torch.equal(w.model.features[0].weight, v.model.features[0].weight)
