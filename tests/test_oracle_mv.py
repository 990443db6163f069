"""Hand-derived known-answer tests of oracle/mv_ref.py (CPU).  The oracle restates diffusers
0.19.3 / xformers 0.0.17 pieces that cannot be executed here (PARITY UNPINNED); these KATs pin its
arithmetic to properties that follow from the published definitions and from the reference's own
processor code (mvdiffusion/models/transformer_mv2d.py:722-906)."""
import math

import torch

from oracle import mv_ref as mr


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)


def test_attention_on_one_key_is_the_value():
    q, k, v = _rand(3, 5, 8, seed=1), _rand(3, 1, 8, seed=2), _rand(3, 1, 8, seed=3)
    out = mr.memory_efficient_attention(q, k, v)
    torch.testing.assert_close(out, v.expand(3, 5, 8))


def test_attention_is_invariant_to_key_order_and_scaled_by_sqrt_d():
    q, k, v = _rand(2, 4, 16, seed=4), _rand(2, 7, 16, seed=5), _rand(2, 7, 16, seed=6)
    perm = torch.tensor([3, 0, 6, 1, 5, 2, 4])
    torch.testing.assert_close(mr.memory_efficient_attention(q, k, v),
                               mr.memory_efficient_attention(q, k[:, perm], v[:, perm]))
    # explicit softmax(q k^T / sqrt(d)) v for one (batch, query)
    s = (q[0, 2] @ k[0].T) / math.sqrt(16)
    want = torch.softmax(s, 0) @ v[0]
    torch.testing.assert_close(mr.memory_efficient_attention(q, k, v)[0, 2], want)


def test_head_split_round_trip_and_layout():
    x = _rand(2, 3, 16, seed=7)
    hb = mr.head_to_batch_dim(x, 4)
    assert hb.shape == (8, 3, 4)
    # (batch b, head h) lands at row b*heads + h and takes channels [h*d, (h+1)*d)
    torch.testing.assert_close(hb[1 * 4 + 2], x[1, :, 8:12])
    torch.testing.assert_close(mr.batch_to_head_dim(hb, 4), x)


def test_mv_core_every_view_attends_to_the_tokens_of_all_views():
    """transformer_mv2d.py:783-796: K/V of the 6 views of one object are concatenated on the
    token axis and repeated for the 6 queries, so the result for view i equals plain attention of
    view i's queries over the 6*N keys, per head."""
    views, n, heads, d = 6, 5, 2, 4
    q, k, v = (_rand(2 * views, n, heads * d, seed=s) for s in (8, 9, 10))
    out = mr.mv_attention_core(q, k, v, heads, views)
    for obj in range(2):
        kk = k[obj * views:(obj + 1) * views].reshape(views * n, heads * d)
        vv = v[obj * views:(obj + 1) * views].reshape(views * n, heads * d)
        for view in range(views):
            for h in range(heads):
                sl = slice(h * d, (h + 1) * d)
                s = q[obj * views + view][:, sl] @ kk[:, sl].T / math.sqrt(d)
                want = torch.softmax(s, -1) @ vv[:, sl]
                torch.testing.assert_close(out[obj * views + view][:, sl], want)


def test_joint_core_pairs_sample_i_of_one_domain_with_sample_i_of_the_other():
    """transformer_mv2d.py:876-883: chunk(2) on the batch -> cat on tokens -> both halves see the
    same 2N keys: [domain-0 tokens of sample i | domain-1 tokens of sample i]."""
    b, n, heads, d = 6, 3, 2, 4
    q, k, v = (_rand(2 * b, n, heads * d, seed=s) for s in (11, 12, 13))
    out = mr.joint_attention_core(q, k, v, heads)
    for half in range(2):
        for i in range(b):
            kk = torch.cat([k[i], k[b + i]], 0)
            vv = torch.cat([v[i], v[b + i]], 0)
            for h in range(heads):
                sl = slice(h * d, (h + 1) * d)
                s = q[half * b + i][:, sl] @ kk[:, sl].T / math.sqrt(d)
                torch.testing.assert_close(out[half * b + i][:, sl], torch.softmax(s, -1) @ vv[:, sl])


def test_timestep_embedding_kat():
    e = mr.timestep_embedding(torch.tensor([0.0, 1.0]), 8)
    # flip_sin_to_cos: [cos | sin]; t = 0 -> cos 1, sin 0
    torch.testing.assert_close(e[0], torch.tensor([1.0, 1, 1, 1, 0, 0, 0, 0], dtype=torch.float64))
    freqs = torch.exp(-math.log(10000) * torch.arange(4, dtype=torch.float64) / 4)
    torch.testing.assert_close(e[1], torch.cat([torch.cos(freqs), torch.sin(freqs)]))


def test_unet_ref_building_blocks_geglu_and_single_token_cross_attention():
    """GEGLU = first half * gelu(second half) (diffusers GEGLU.forward); cross-attention over the
    single CLIP token equals to_out(to_v(ctx)) for every query (softmax over one key is 1)."""
    c, heads = 16, 2
    g = torch.Generator().manual_seed(14)
    sd = {}
    for name, shape in [("b.attn2.to_q.weight", (c, c)), ("b.attn2.to_k.weight", (c, 8)),
                        ("b.attn2.to_v.weight", (c, 8)), ("b.attn2.to_out.0.weight", (c, c)),
                        ("b.attn2.to_out.0.bias", (c,))]:
        sd[name] = torch.randn(*shape, generator=g, dtype=torch.float64)
    ref = mr.UNetRef(sd, (c,), (), (), heads=heads)
    h, ctx = _rand(3, 5, c, seed=15), _rand(3, 1, 8, seed=16)
    q, k, v = ref.attn_proj("b.attn2", h, ctx)
    o = mr.memory_efficient_attention(mr.head_to_batch_dim(q, heads), mr.head_to_batch_dim(k, heads),
                                      mr.head_to_batch_dim(v, heads))
    got = ref.lin("b.attn2.to_out.0", mr.batch_to_head_dim(o, heads))
    want = ref.lin("b.attn2.to_out.0", ref.lin("b.attn2.to_v", ctx, False)).expand(3, 5, c)
    torch.testing.assert_close(got, want)
    x = _rand(4, 2 * c, seed=17)
    a, gate = x.chunk(2, -1)
    want = a * 0.5 * gate * (1 + torch.erf(gate / math.sqrt(2)))
    torch.testing.assert_close(a * torch.nn.functional.gelu(gate), want)


def test_ddim_schedule_and_step_kats():
    acp = mr.ddim_alphas_cumprod()
    assert abs(float(acp[0]) - (1 - 0.00085)) < 1e-7          # scaled_linear: beta_0 = beta_start
    betas_last = 0.012
    assert abs(float(acp[999] / acp[998]) - (1 - betas_last)) < 1e-6
    ts = mr.ddim_timesteps(75)
    assert ts[0] == 74 * 13 + 1 and ts[-1] == 1 and len(ts) == 75 and ts[0] - ts[1] == 13
    # eta = 0 and the TRUE noise as the model output: x_prev is the same clean sample re-noised
    # to the previous level with the same noise
    x0, eps = _rand(2, 4, 3, 3, seed=18), _rand(2, 4, 3, 3, seed=19)
    t = ts[10]
    x_t = acp[t] ** 0.5 * x0 + (1 - acp[t]) ** 0.5 * eps
    prev = mr.ddim_step(eps, t, x_t, 75, 0.0, None, acp)
    want = acp[t - 13] ** 0.5 * x0 + (1 - acp[t - 13]) ** 0.5 * eps
    torch.testing.assert_close(prev, want)
    # eta = 1 adds sigma_t * noise with sigma_t^2 = (1-a_prev)/(1-a_t) (1 - a_t/a_prev)
    z = _rand(2, 4, 3, 3, seed=20)
    prev1 = mr.ddim_step(eps, t, x_t, 75, 1.0, z, acp)
    var = (1 - acp[t - 13]) / (1 - acp[t]) * (1 - acp[t] / acp[t - 13])
    want1 = acp[t - 13] ** 0.5 * x0 + (1 - acp[t - 13] - var) ** 0.5 * eps + var ** 0.5 * z
    torch.testing.assert_close(prev1, want1)
    # last step (t = 1): previous alpha is alphas_cumprod[0] (set_alpha_to_one False)
    last = mr.ddim_step(eps, 1, x_t, 75, 0.0, None, acp)
    x0_hat = (x_t - (1 - acp[1]) ** 0.5 * eps) / acp[1] ** 0.5
    torch.testing.assert_close(last, acp[0] ** 0.5 * x0_hat + (1 - acp[0]) ** 0.5 * eps)


def test_product_scheduler_matches_the_oracle_step():
    """The product's DDIMScheduler (host arithmetic, drawingspinup_amd/mv/pipeline.py) against the
    float64 restatement, on CPU tensors."""
    from drawingspinup_amd.mv.pipeline import DDIMScheduler
    sch = DDIMScheduler()
    sch.set_timesteps(75)
    assert sch.timesteps.tolist() == mr.ddim_timesteps(75)
    x, eps, z = (_rand(2, 4, 3, 3, seed=s).float() for s in (21, 22, 23))
    for t in (963, 482, 1):
        got = sch.step(eps, torch.tensor(t), x, eta=1.0, variance_noise=z)
        want = mr.ddim_step(eps, t, x, 75, 1.0, z)
        torch.testing.assert_close(got.double(), want, rtol=2e-6, atol=2e-6)
