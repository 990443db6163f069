"""UNetMV2DConditionModel on the gfx950 kernels.

Mirrors 2_charactor_reconstructor/mvdiffusion/models/{unet_mv2d_condition,unet_mv2d_blocks,
transformer_mv2d}.py with the diffusers==0.19.3 building blocks they import: same constructor
arguments (the subset the Wonder3D joint config uses), same module tree and parameter names,
so a `flamehaze1115/wonder3d-v1.0` UNet state_dict loads with strict=True.

Data layout: activations are NHWC f16 for the whole network, so the (B,C,H,W)->(B,HW,C)
permutes around every transformer (transformer_mv2d.py:313,330) are free views.  Hand-written
HIP: 3x3 / strided / upsampling convolutions (implicit GEMM on f16 MFMA), multi-view and
cross-domain attention (shared K/V read in place), GroupNorm(+SiLU), LayerNorm, GEGLU.
Plain dense projections (to_q/k/v/out, FeedForward, 1x1 proj_in/out, embeddings) are library
GEMMs (torch.nn.functional.linear -> hipBLASLt).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _okc(conv):
    """cached (O, k*k, C) f16 image of an nn.Conv2d weight."""
    w = conv.weight
    c = getattr(conv, "_dsu_okc", None)
    if c is None or c[0] != w._version or c[1].device != w.device:
        conv._dsu_okc = (w._version, ops.conv_weight_okc(w))
        c = conv._dsu_okc
    return c[1]


def conv_nhwc(conv, x, upsample2x=False, addvec=None, residual=None):
    k = conv.kernel_size[0]
    return ops.conv2d_nhwc_f16(x, _okc(conv), conv.bias, k, conv.stride[0], conv.padding[0],
                               upsample2x, addvec, residual)


def group_norm(gn, x, silu=False):
    return ops.groupnorm_nhwc_f16(x, gn.weight, gn.bias, gn.num_groups, gn.eps, silu)


def layer_norm(ln, x):
    return ops.layernorm_f16(x, ln.weight, ln.bias, ln.eps)


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32,
                                                   device=timesteps.device) / (half - self.shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        """Both layers on the MFMA GEMM (inputs narrower than a 16-byte K group — the 10-wide
        class embedding — are zero-padded to 16 once, weight columns alike)."""
        w1 = self.linear_1.weight
        if w1.shape[1] % 8:
            pad = (-w1.shape[1]) % 16
            c = getattr(self, "_dsu_w1p", None)
            if c is None or c[0] != w1._version or c[1].device != w1.device:
                self._dsu_w1p = (w1._version, F.pad(w1.detach(), (0, pad)).contiguous())
                c = self._dsu_w1p
            sample, w1 = F.pad(sample, (0, pad)), c[1]
        h = ops.linear_f16(sample.contiguous(), w1, self.linear_1.bias)
        return ops.linear_f16(self.act(h), self.linear_2.weight, self.linear_2.bias)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0) \
            if in_channels != out_channels else None

    def forward(self, x, temb_act, tproj=None):
        """x NHWC f16; temb_act = silu(emb) (B, temb_channels) computed once per UNet call;
        tproj: this block's time_emb_proj(temb_act) when the UNet has projected all blocks at once."""
        h = group_norm(self.norm1, x, silu=True)
        t = tproj if tproj is not None else \
            ops.linear_f16(temb_act, self.time_emb_proj.weight, self.time_emb_proj.bias)
        h = conv_nhwc(self.conv1, h, addvec=t)
        h = group_norm(self.norm2, h, silu=True)
        sc = x if self.conv_shortcut is None else conv_nhwc(self.conv_shortcut, x)
        return conv_nhwc(self.conv2, h, residual=sc)       # output_scale_factor == 1


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return conv_nhwc(self.conv, x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return conv_nhwc(self.conv, x, upsample2x=True)


class Attention(nn.Module):
    """Parameter holder with diffusers' Attention names (to_q, to_k, to_v, to_out.0)."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0),
                                  nn.Linear(dim * mult, dim)])

    def forward(self, x, residual):
        # proj + GEGLU in one launch (the (M, 8C) projection is never written), then the output
        # layer with bias and residual in its epilogue
        h = ops.linear_geglu_f16(x, self.net[0].proj.weight, self.net[0].proj.bias)
        return ops.linear_f16(h, self.net[2].weight, self.net[2].bias, residual=residual)


# lazily built, process-wide; an entry is published only once the stream that filled it has finished
# (several drawings may be in flight on one GPU, each on its own thread + stream)
_SEG_CACHE = {}


def _seg_table(kind, B, num_views, device):
    key = (kind, B, num_views, str(device))
    if key not in _SEG_CACHE:
        if kind == "mv":      # transformer_mv2d.py:785: "(b t) d c -> b (t d) c" then repeat t
            rows = [[(b // num_views) * num_views + s for s in range(num_views)] for b in range(B)]
        else:                 # joint, transformer_mv2d.py:878-883: chunk(2) / cat seq / cat batch
            half = B // 2
            rows = [[b % half, b % half + half] for b in range(B)]
        t = torch.tensor(rows, dtype=torch.int32, device=device)
        from .._lib import publish_sync
        publish_sync(device)
        _SEG_CACHE[key] = t
    return _SEG_CACHE[key]


def _qk_weight(attn):
    """to_q and to_k stacked (2C, C): ONE GEMM for both projections of a self-attention (the 768 ..
    3072-row projections fill 60-120 of 256 CUs each; Q and K are strided views of the result,
    the attention kernel takes row strides).  Kept ON the module (it dies with it; a module-level
    dict keyed by id() leaked one 2C x C tensor per attention of every UNet ever built and could
    hand a recycled id the previous model's weights); rebuilt when either parameter is written."""
    wq, wk = attn.to_q.weight, attn.to_k.weight
    tag = (wq._version, wk._version, wq.data_ptr(), wk.data_ptr())
    hit = attn.__dict__.get("_dsu_qk")
    if hit is None or hit[0] != tag:
        hit = (tag, torch.cat([wq.detach(), wk.detach()], 0).contiguous())
        attn.__dict__["_dsu_qk"] = hit            # plain attribute: not a buffer, not in state_dict
    return hit[1]


def _self_attention(attn, x, residual, table):
    """x (B,N,C) normalised input; returns to_out(attention) + residual."""
    c = attn.to_q.weight.shape[0]
    qk = ops.linear_f16(x, _qk_weight(attn))
    q, k = qk[..., :c], qk[..., c:]
    vt = ops.linear_f16(x, attn.to_v.weight, transposed_tokens=x.shape[1])   # (B, C, N): V^T per head
    o = ops.mv_attention(q, k, vt, table, attn.heads, x.shape[1])
    return ops.linear_f16(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=residual)


class BasicMVTransformerBlock(nn.Module):
    """transformer_mv2d.py:376-625 for the configuration the pipeline runs
    (layer_norm, xformers MV processor, cd_attention_mid)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim, num_views=1,
                 cd_attention_last=False, cd_attention_mid=False, multiview_attention=True):
        super().__init__()
        self.num_views, self.multiview_attention = num_views, multiview_attention
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.cd_attention_last, self.cd_attention_mid = cd_attention_last, cd_attention_mid
        if cd_attention_last:
            self.attn_joint_last = Attention(dim, heads, dim_head)
            nn.init.zeros_(self.attn_joint_last.to_out[0].weight.data)
            self.norm_joint_last = nn.LayerNorm(dim)
        if cd_attention_mid:
            self.attn_joint_mid = Attention(dim, heads, dim_head)
            nn.init.zeros_(self.attn_joint_mid.to_out[0].weight.data)
            self.norm_joint_mid = nn.LayerNorm(dim)

    def _cross_attention_term(self, ctx):
        """to_out(to_v(ctx)) — a function of the image embedding and two weights only, i.e. the
        same tensor at every one of the 75 denoising steps of a drawing: computed at the first
        step, kept while the caller keeps handing over the SAME context tensor (identity through a
        weak reference and the version counters: a new tensor, an in-place write or a weight
        update recomputes)."""
        import weakref
        a = self.attn2
        tag = (ctx._version, a.to_v.weight._version, a.to_out[0].weight._version,
               a.to_out[0].bias._version, a.to_v.weight.data_ptr())
        hit = self.__dict__.get("_dsu_ctx_term")
        if hit is not None and hit[0]() is ctx and hit[1] == tag:
            return hit[2]
        v = ops.linear_f16(ctx[:, 0].contiguous(), a.to_v.weight)
        o = ops.linear_f16(v, a.to_out[0].weight, a.to_out[0].bias)
        self.__dict__["_dsu_ctx_term"] = (weakref.ref(ctx), tag, o)
        return o

    def forward(self, h, encoder_hidden_states):
        B, N, C = h.shape
        dev = h.device
        views = self.num_views if self.multiview_attention else 1
        h = _self_attention(self.attn1, layer_norm(self.norm1, h), h,
                            _seg_table("mv", B, views, dev))
        if self.cd_attention_mid:
            h = _self_attention(self.attn_joint_mid, layer_norm(self.norm_joint_mid, h), h,
                                _seg_table("joint", B, 0, dev))
        # cross attention on the CLIP image embedding (transformer_mv2d.py:579-590).  The
        # pipeline passes ONE context token (pipeline_mvdiffusion_image.py:155-156), for which
        # softmax over a single key is exactly 1: attn2 == to_out(to_v(ctx)) for every query.
        ctx = encoder_hidden_states
        if ctx.shape[1] != 1:
            raise NotImplementedError("cross-attention context with more than one token")
        h = h + self._cross_attention_term(ctx)[:, None, :]
        h = self.ff(layer_norm(self.norm3, h), h)
        if self.cd_attention_last:
            h = _self_attention(self.attn_joint_last, layer_norm(self.norm_joint_last, h), h,
                                _seg_table("joint", B, 0, dev))
        return h


class TransformerMV2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32, **blk):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1, 1, 0)
        self.transformer_blocks = nn.ModuleList(
            [BasicMVTransformerBlock(inner, heads, dim_head, cross_attention_dim, **blk)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1, 1, 0)

    def forward(self, x, encoder_hidden_states):
        B, H, W, C = x.shape
        h = group_norm(self.norm, x)
        h = ops.linear_f16(h.view(B, H * W, C), self.proj_in.weight.view(-1, C), self.proj_in.bias)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states)
        inner = h.shape[-1]
        return ops.linear_f16(h, self.proj_out.weight.view(C, inner), self.proj_out.bias,
                              residual=x.view(B, H * W, C)).view(B, H, W, C)


class _Block(nn.Module):
    pass


class UNetMV2DConditionModel(nn.Module):
    def __init__(self, sample_size=32, in_channels=8, out_channels=4,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 projection_class_embeddings_input_dim=10, num_views=6, cd_attention_last=False,
                 cd_attention_mid=True, multiview_attention=True,
                 down_block_types=("CrossAttnDownBlockMV2D", "CrossAttnDownBlockMV2D",
                                   "CrossAttnDownBlockMV2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlockMV2D", "CrossAttnUpBlockMV2D",
                                 "CrossAttnUpBlockMV2D")):
        super().__init__()
        self.config = dict(sample_size=sample_size, in_channels=in_channels,
                           out_channels=out_channels, class_embed_type="projection",
                           projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                           num_views=num_views, cross_attention_dim=cross_attention_dim)
        ted = block_out_channels[0] * 4
        heads = attention_head_dim            # SD-1.x naming quirk: this IS the head count
        blk = dict(num_views=num_views, cd_attention_last=cd_attention_last,
                   cd_attention_mid=cd_attention_mid, multiview_attention=multiview_attention)
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.time_proj = Timesteps(block_out_channels[0], True, 0)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], ted)
        self.class_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, ted)

        def tf(ch):
            return TransformerMV2DModel(heads, ch // heads, ch, cross_attention_dim,
                                        norm_num_groups, **blk)

        self.down_blocks = nn.ModuleList()
        out_ch = block_out_channels[0]
        for i, typ in enumerate(down_block_types):
            in_ch, out_ch = out_ch, block_out_channels[i]
            final = i == len(block_out_channels) - 1
            b = _Block()
            b.resnets = nn.ModuleList([ResnetBlock2D(in_ch if j == 0 else out_ch, out_ch, ted,
                                                     norm_num_groups, norm_eps)
                                       for j in range(layers_per_block)])
            if typ.startswith("CrossAttn"):
                b.attentions = nn.ModuleList([tf(out_ch) for _ in range(layers_per_block)])
            if not final:
                b.downsamplers = nn.ModuleList([Downsample2D(out_ch)])
            self.down_blocks.append(b)

        mid = _Block()
        mc = block_out_channels[-1]
        mid.attentions = nn.ModuleList([tf(mc)])
        mid.resnets = nn.ModuleList([ResnetBlock2D(mc, mc, ted, norm_num_groups, norm_eps)
                                     for _ in range(2)])
        self.mid_block = mid

        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        out_ch = rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(rev) - 1)]
            final = i == len(rev) - 1
            b = _Block()
            n = layers_per_block + 1
            b.resnets = nn.ModuleList()
            for j in range(n):
                skip = in_ch if j == n - 1 else out_ch
                rin = prev if j == 0 else out_ch
                b.resnets.append(ResnetBlock2D(rin + skip, out_ch, ted, norm_num_groups, norm_eps))
            if typ.startswith("CrossAttn"):
                b.attentions = nn.ModuleList([tf(out_ch) for _ in range(n)])
            if not final:
                b.upsamplers = nn.ModuleList([Upsample2D(out_ch)])
            self.up_blocks.append(b)

        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels):
        """sample (B,in,H,W) f16 NCHW; timestep scalar / (B,); encoder_hidden_states (B,1,768);
        class_labels (B,10).  Returns (B,out,H,W) f16 (the `.sample` of the reference)."""
        if not sample.is_cuda:
            raise RuntimeError("the gfx950 UNet needs device tensors (no CPU fallback)")
        dt = torch.float16
        B = sample.shape[0]
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], device=sample.device)
        t = t.to(sample.device).reshape(-1).expand(B)
        emb = self.time_embedding(self.time_proj(t).to(dt))
        emb = emb + self.class_embedding(class_labels.to(dt))
        temb_act = F.silu(emb)
        ctx = encoder_hidden_states.to(dt)
        tp = self._time_projections(temb_act)            # every ResnetBlock2D.time_emb_proj in ONE GEMM

        x = sample.to(dt).permute(0, 2, 3, 1).contiguous()          # NHWC
        x = conv_nhwc(self.conv_in, _pad_c8(x))
        skips = [x]
        for b in self.down_blocks:
            for j, res in enumerate(b.resnets):
                x = res(x, temb_act, tp[id(res)])
                if hasattr(b, "attentions"):
                    x = b.attentions[j](x, ctx)
                skips.append(x)
            if hasattr(b, "downsamplers"):
                x = b.downsamplers[0](x)
                skips.append(x)
        m = self.mid_block
        x = m.resnets[0](x, temb_act, tp[id(m.resnets[0])])
        x = m.attentions[0](x, ctx)
        x = m.resnets[1](x, temb_act, tp[id(m.resnets[1])])
        for b in self.up_blocks:
            for j, res in enumerate(b.resnets):
                x = res(torch.cat([x, skips.pop()], dim=-1), temb_act, tp[id(res)])
                if hasattr(b, "attentions"):
                    x = b.attentions[j](x, ctx)
            if hasattr(b, "upsamplers"):
                x = b.upsamplers[0](x)
        x = group_norm(self.conv_norm_out, x, silu=True)
        x = conv_nhwc(self.conv_out, x)
        return x.permute(0, 3, 1, 2).contiguous()


def _time_projections(self, temb_act):
    """time_emb_proj(silu(emb)) of all ResnetBlock2D at once: their weights stacked (sum of
    out_channels, temb_channels) -> one GEMM with M = batch rows instead of 22 launches of
    12-row GEMMs (13-16 us each, latency only); one gather re-packs the column blocks as the
    contiguous (B, out_channels) vectors the convolution epilogue reads.  Same products, same
    f32 accumulation, per block."""
    blocks = self.__dict__.get("_dsu_resblocks")
    if blocks is None:                     # the module tree is fixed after construction
        blocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        self.__dict__["_dsu_resblocks"] = blocks
    tag = tuple((m.time_emb_proj.weight._version, m.time_emb_proj.bias._version,
                 m.time_emb_proj.weight.data_ptr()) for m in blocks)
    hit = self.__dict__.get("_dsu_tproj")
    if hit is None or hit[0] != tag:
        w = torch.cat([m.time_emb_proj.weight.detach() for m in blocks], 0).contiguous()
        b = torch.cat([m.time_emb_proj.bias.detach() for m in blocks], 0).contiguous()
        hit = (tag, w, b, [m.time_emb_proj.out_features for m in blocks])
        self.__dict__["_dsu_tproj"] = hit
    _, w, b, sizes = hit
    out = ops.linear_f16(temb_act.contiguous(), w, b)                   # (B, sum O)
    B, total = out.shape
    idx = self.__dict__.get("_dsu_tproj_idx")
    if idx is None or idx[0] != (B, total, str(out.device)):
        cols, rows = torch.arange(total), torch.arange(B)
        parts, c0 = [], 0
        for o in sizes:                      # block i, packed: rows of (B, o) one after the other
            parts.append((rows[:, None] * total + cols[None, c0:c0 + o]).reshape(-1))
            c0 += o
        idx = ((B, total, str(out.device)), torch.cat(parts).to(out.device))
        self.__dict__["_dsu_tproj_idx"] = idx
    flat = out.reshape(-1).index_select(0, idx[1])                       # ONE gather for all blocks
    res, off = {}, 0
    for m, o in zip(blocks, sizes):
        res[id(m)] = flat[off:off + B * o].view(B, o)
        off += B * o
    return res


UNetMV2DConditionModel._time_projections = _time_projections


def _pad_c8(x):
    """the conv kernel reads 8-channel (16-byte) groups; in_channels=8 already satisfies it."""
    if x.shape[-1] % 8 == 0:
        return x
    raise NotImplementedError("input channels must be a multiple of 8")
