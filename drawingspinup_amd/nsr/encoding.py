"""Hash-grid encoding modules.

`Encoding` has the interface of `tinycudann.Encoding` as the reference uses it
(2_charactor_reconstructor/instant_nsr/models/network_utils.py:46,55): constructor
(n_input_dims, config dict), `.n_output_dims`, `.params` (flat f32 nn.Parameter, level-major /
entry-major / feature-minor — the tcnn checkpoint layout), `__call__((N,3) in [0,1]) -> (N,20)
half`.  `active_levels` fuses the ProgressiveBandHashGrid mask (network_utils.py:54-64) into
the kernel: masked levels are neither fetched nor written non-zero.
"""
import torch
import torch.nn as nn

from .. import ops


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, enc, active):
        ctx.save_for_backward(x)
        ctx.enc, ctx.active = enc, active
        return ops.hashgrid_encode_fwd(enc.cfg, enc.table_f16(), x, active)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        g = ops.hashgrid_encode_bwd(ctx.enc.cfg, x, dout.float(), ctx.active)
        return None, g, None, None


_INIT_CACHE = {}


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        assert n_input_dims == 3, "the gfx950 hash grid is 3-D"
        otype = encoding_config.get("otype", "HashGrid")
        assert otype in ("HashGrid", "Grid"), f"unsupported encoding {otype}"
        self.cfg = ops.HashGridConfig(
            n_levels=int(encoding_config.get("n_levels", 16)),
            n_features_per_level=int(encoding_config.get("n_features_per_level", 2)),
            log2_hashmap_size=int(encoding_config.get("log2_hashmap_size", 19)),
            base_resolution=int(encoding_config.get("base_resolution", 16)),
            per_level_scale=float(encoding_config.get("per_level_scale", 2.0)))
        self.n_input_dims = 3
        self.n_output_dims = self.cfg.n_output_dims
        # tcnn initialises grid parameters U(-1e-4, 1e-4) from its own fixed seed (1337): every
        # encoding of a given size starts from the same values, so the draw (7.7 M values: ~20 ms of
        # a drawing's reconstruction on the host) is made once per process and copied
        key = (self.cfg.n_params, int(seed))
        init = _INIT_CACHE.get(key)
        if init is None:
            g = torch.Generator().manual_seed(seed)
            init = (torch.rand(self.cfg.n_params, generator=g) * 2 - 1) * 1e-4
            _INIT_CACHE.clear()                       # one size at a time (30 MB of host memory)
            _INIT_CACHE[key] = init
        self.params = nn.Parameter(init.clone())
        self._shadow = None
        self._shadow_version = -1
        self._shadow_locked = False
        self.active_levels = self.cfg.n_levels

    def table_f16(self):
        """f16 image of the f32 master parameters (what tcnn's kernels read).  Refreshed when the
        parameter tensor has been written since the last call.  In-place optimizers are detected
        through the tensor version counter EXCEPT the fused multi-tensor ones (torch's
        `_fused_adamw_` does not bump it), so callers that step such an optimizer call
        `invalidate()`; while the module is in training mode the image is simply rebuilt on every
        call (one 15 us pass over 12.6 M entries)."""
        p = self.params
        if self._shadow_locked and self._shadow is not None and self._shadow.device == p.device:
            if self._shadow_version == p._version:
                return self._shadow      # maintained by the fused table optimizer (TableAdamW)
            # somebody wrote the master parameters through torch (load_state_dict, a checkpoint
            # resume, a test fixture): the fused kernels write through raw pointers and never bump
            # the version, so a bump means the image is stale.  Rebuild; the optimizer re-locks.
            self._shadow_locked = False
        if (self._shadow is None or self.training or self._shadow_version != p._version
                or self._shadow.device != p.device):
            self._shadow = p.detach().to(torch.float16).contiguous()
            self._shadow_version = p._version
        return self._shadow

    def invalidate(self):
        """Force the next table_f16() to re-read the master parameters."""
        self._shadow_version = -1
        self._shadow_locked = False

    def lock_shadow(self):
        """The caller keeps the f16 image in step with the master parameters itself (the fused
        table optimizer writes both in one pass); until invalidate() the image is not rebuilt."""
        self._shadow_locked = False
        img = self.table_f16() if not self.training else \
            self.params.detach().to(torch.float16).contiguous()
        self._shadow, self._shadow_version = img, self.params._version
        self._shadow_locked = True
        return img

    def set_shadow(self, table_f16):
        """Used by the fused optimizer step, which writes master and f16 image in one pass."""
        self._shadow = table_f16
        self._shadow_version = self.params._version

    def forward(self, x):
        return _EncodeFn.apply(x.float().contiguous(), self.params, self, self.active_levels)


class ProgressiveBandHashGrid(nn.Module):
    """network_utils.py:39-64 with the level mask applied inside the kernel."""

    def __init__(self, in_channels, config):
        super().__init__()
        self.n_input_dims = in_channels
        encoding_config = dict(config)
        encoding_config["otype"] = "HashGrid"
        self.encoding = Encoding(in_channels, encoding_config)
        self.n_output_dims = self.encoding.n_output_dims
        self.n_level = config["n_levels"]
        self.n_features_per_level = config["n_features_per_level"]
        self.start_level, self.start_step, self.update_steps = \
            config["start_level"], config["start_step"], config["update_steps"]
        self.current_level = self.start_level
        self.encoding.active_levels = self.current_level

    def forward(self, x):
        return self.encoding(x).float()

    def update_step(self, epoch, global_step):
        self.current_level = min(
            self.start_level + max(global_step - self.start_step, 0) // self.update_steps,
            self.n_level)
        self.encoding.active_levels = self.current_level
