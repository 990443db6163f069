"""The product's loss section (drawingspinup_amd/nsr/system.py: ray_losses / sample_losses, torch
path) against the REFERENCE's own OrthoNeuSSystem.training_step run on the same `out` and batch
(tests/golden/nsr_step_reference.npz, tests/golden/make_nsr_step_golden.py).  CPU only."""
import os

import numpy as np
import torch

from drawingspinup_amd.nsr.system import DEFAULT_SYSTEM_CONFIG, Cfg, OrthoNeuSSystem

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "nsr_step_reference.npz"))
L = DEFAULT_SYSTEM_CONFIG.loss


class LossOnly(OrthoNeuSSystem):
    def __init__(self, loss_cfg=None, has_mask=True):
        self.config = Cfg({"loss": dict(loss_cfg or L)})

        class _D:
            pass
        self.dataset = _D()
        self.dataset.has_mask = has_mask


def fixture_out_and_batch(device="cpu", requires_grad=False):
    t = lambda k: torch.from_numpy(GOLD[k]).to(device)
    out = {k: t("fwd." + k) for k in ("comp_rgb", "comp_normal", "opacity", "depth",
                                      "sdf_grad_samples", "random_sdf", "random_sdf_grad",
                                      "normal_perturb")}
    if requires_grad:
        out = {k: v.clone().requires_grad_(True) for k, v in out.items()}
    batch = {k: t("batch." + k) for k in ("rgb", "normal", "mask", "cosines", "view_weights")}
    return out, batch


def reference_terms():
    """name -> (reference value, lambda) in the product's term names."""
    g = lambda k: float(GOLD["loss." + k])
    return {"rgb_mse": (g("loss_rgb_mse"), L.lambda_rgb_mse), "normal": (g("loss_normal"), L.lambda_normal),
            "mask": (g("loss_mask"), L.lambda_mask), "eikonal": (g("loss_eikonal"), L.lambda_eikonal),
            "sparsity": (g("loss_sparsity"), L.lambda_sparsity),
            "normal_smooth": (g("loss_3d_normal_smooth"), L.lambda_3d_normal_smooth)}


def test_loss_terms_and_total_match_reference_training_step():
    out, batch = fixture_out_and_batch(requires_grad=True)
    terms = LossOnly().losses(out, batch)
    ref = reference_terms()
    assert set(terms) == set(ref)                      # lambda_rgb_l1 = 0 in the shipped config
    for k, (val, lam) in ref.items():
        np.testing.assert_allclose(float(terms[k]), val * lam, rtol=2e-6, atol=1e-8, err_msg=k)
    total = sum(terms.values())
    np.testing.assert_allclose(float(total), float(GOLD["loss.total"]), rtol=2e-6)
    total.backward()
    for k in ("comp_rgb", "opacity", "sdf_grad_samples", "random_sdf", "random_sdf_grad",
              "normal_perturb"):
        want = GOLD["dloss." + k]
        got = out[k].grad.numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7 * max(np.abs(want).max(), 1e-12) + 1e-12,
                                   err_msg=k)


def test_l1_term_matches_reference_when_enabled():
    """lambda_rgb_l1 is 0 in the shipped config (the reference still logs the term)."""
    out, batch = fixture_out_and_batch()
    cfg = dict(L)
    cfg["lambda_rgb_l1"] = 1.0
    terms = LossOnly(cfg).ray_losses(
        torch.cat([out["opacity"], out["depth"], out["comp_rgb"], out["comp_normal"]], 1), batch)
    np.testing.assert_allclose(float(terms["rgb_l1"]), float(GOLD["loss.loss_rgb"]), rtol=2e-6)


def test_dynamic_ray_count_update_matches_reference():
    """neus_ortho.py:90-92 on the fixture's sample count."""
    n = int(GOLD["fwd.num_samples"][0])
    tr = int(256 * (256 * 1024 / n))
    assert min(int(256 * 0.9 + tr * 0.1), 8192) == int(GOLD["train_num_rays_after"])
