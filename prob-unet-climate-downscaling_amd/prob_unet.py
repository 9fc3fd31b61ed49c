"""Drop-in host side of the MI355X Probabilistic U-Net engine.

Mirrors the interface of the reference class `ProbabilisticUNet` (src/prob_unet.py:140-267 of
MaryamAlipourH/prob-unet-climate-downscaling) so that train_prob_unet_model.py / latent_exploration*.py
call patterns run unchanged:

    model = ProbabilisticUNet(input_channels, num_classes, latent_dim, num_filters, model_channels,
                              channel_mult, beta_0, beta_1, beta_2).to("cuda")
    loss, recon_list, kl_div = model.elbo(inputs, targets, timestamps, M=5)     # train_prob_unet_model.py:133
    optimizer.zero_grad(); loss.backward(); optimizer.step()                    # :139-141
    out = model(inputs, t=timestamps, training=False)                           # :245
    feat = model.unet(x); p = model.prior(x); out = model.fcomb(feat.expand(K, -1, -1, -1), z)   # latent_exploration.py:119-129

All arithmetic runs in libprobunet.so (hand-written HIP for gfx950) through the C ABI of include/probunet.h;
this file only owns the nn.Parameters (reference state_dict keys, shapes and initialisation order), the autograd
plumbing and the torch.distributed gradient all-reduce.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Independent, Normal

from . import _lib as L


# ----------------------------------------------------------------------------------------------- param tree
class _Holder(nn.Module):
    """A parameter container standing in for one reference sub-module (Conv2d / GroupNorm / Linear / nn.Conv2d)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("this module only holds parameters; compute runs in libprobunet.so")


class _IndexedHolders(nn.Module):
    """nn.Sequential look-alike: integer-indexable children (fcomb.layers[0], prior.encoder[7] ...)."""

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __len__(self):
        return len(self._modules)


def _ensure_child(parent: nn.Module, name: str, cls):
    if name not in parent._modules:
        parent.add_module(name, cls())
    return parent._modules[name]


# ----------------------------------------------------------------------------------------------- init replay
def _weight_init(shape, fan_in):
    """kaiming_uniform of the reference U-Net (networks.py:21-26): sqrt(3/fan_in) * U(-1, 1)."""
    return np.sqrt(3 / fan_in) * (torch.rand(*shape) * 2 - 1)


def _truncated_normal_(t: torch.Tensor, std: float):
    """prob_unet_utils.py:10-16: 4 normal draws per element, first one inside (-2, 2), times std."""
    tmp = t.new_empty(tuple(t.shape) + (4,)).normal_()
    valid = (tmp < 2) & (tmp > -2)
    ind = valid.max(-1, keepdim=True)[1]
    t.copy_(tmp.gather(-1, ind).squeeze(-1))
    t.mul_(std)


def _init_reference_order(model: "ProbabilisticUNet"):
    """Fill every parameter consuming torch's global CPU generator in exactly the order the reference constructor
    does (UNet: networks.py:243-297; prior/posterior/fcomb: nn.Conv2d default init, then init_weights,
    prob_unet_utils.py:18-23), so that `torch.manual_seed(s); ProbabilisticUNet(...)` reproduces the reference's
    initial state_dict (pinned by tests/golden/*.json `init_seed42`)."""
    P = dict(model.named_parameters())
    s13 = np.sqrt(1 / 3)
    with torch.no_grad():
        emb = model.model_channels * 4
        P["unet.map_label.weight"].copy_(np.sqrt(1 / 1) * torch.randn(emb, 1) * np.sqrt(1))
        names = [n for n in P if n.startswith("unet.")]
        # group by module prefix in registration order
        seen = []
        for n in names:
            pre = n.rsplit(".", 1)[0]
            if pre not in seen:
                seen.append(pre)
        for pre in seen:
            if pre == "unet.map_label":
                continue
            leaf = pre.rsplit(".", 1)[1]
            w = P.get(pre + ".weight"); b = P.get(pre + ".bias")
            if leaf.startswith("norm") or leaf == "out_norm":
                w.fill_(1.0); b.fill_(0.0)
                continue
            zero = leaf in ("conv1", "out_conv")
            fan_in = w[0].numel() if w.dim() == 4 else w.shape[1]
            wv = _weight_init(list(w.shape), fan_in) * (0 if zero else s13)
            bv = _weight_init([b.shape[0]], fan_in) * (0 if zero else s13)
            w.copy_(wv); b.copy_(bv)

        def conv_default(cout, cin, k):
            return nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2)   # consumes the generator like the reference

        def init_weights_(m):
            nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
            _truncated_normal_(m.bias.data, 0.001)

        cin0 = model.input_channels
        for net, ref_cin0 in (("prior", cin0), ("posterior", 2 * cin0)):
            mods = []
            keys = [n.rsplit(".", 1)[0] for n in P if n.startswith(net + ".") and n.endswith(".weight")]
            for k in keys:
                shp = list(P[k + ".weight"].shape)
                if k == "posterior.encoder.0":
                    shp[1] = ref_cin0               # reference stem has 2*Cin planes (prob_unet.py:27-28)
                mods.append((k, conv_default(shp[0], shp[1], shp[2])))
            for k, m in mods:
                init_weights_(m)
            for k, m in mods:
                wv = m.weight.data
                if k == "posterior.encoder.0":
                    wv = wv[:, : P[k + ".weight"].shape[1]]
                P[k + ".weight"].copy_(wv); P[k + ".bias"].copy_(m.bias.data)
        mods = []
        for k in ("fcomb.layers.0", "fcomb.layers.2", "fcomb.layers.4"):
            shp = P[k + ".weight"].shape
            mods.append((k, conv_default(shp[0], shp[1], 1)))
        for k, m in mods:
            init_weights_(m)
        for k, m in mods:
            P[k + ".weight"].copy_(m.weight.data); P[k + ".bias"].copy_(m.bias.data)


# ----------------------------------------------------------------------------------------------- autograd glue
class _DeliverGrads(torch.autograd.Function):
    """Identity on `value`; in backward, scales the engine's parameter gradients by grad_output and delivers them
    to the nn.Parameters of `owner` in [lo, hi) (flat offsets).  `anchor` is a dummy leaf that makes autograd call us."""

    @staticmethod
    def forward(ctx, value, anchor, owner, lo, hi):
        ctx.owner, ctx.lo, ctx.hi = owner, lo, hi
        ctx.gen = owner._gen["grads"]
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        ctx.owner._check_fresh("grads", ctx.gen, "elbo()")
        ctx.owner._deliver(g, ctx.lo, ctx.hi)
        return None, None, None, None, None


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, owner):
        ctx.owner = owner
        out = owner._unet_fwd(x)
        ctx.gen, ctx.xin = owner._gen["unet"], owner._xin_token
        return out

    @staticmethod
    def backward(ctx, dfeat):
        o = ctx.owner
        o._check_fresh("unet", ctx.gen, "model.unet(x)")
        if o._xin_token != ctx.xin:
            raise L.ProbUNetLibraryError("stale engine state: the input planes saved by model.unet(x) were overwritten by a later "
                                         "model.prior(x') call with a different x before backward() (one forward in flight per engine)")
        o._gen["grads"] += 1
        lo, hi = o._ranges["unet"]
        o._unalias_grads(lo, hi)
        o._engine_grads[lo:hi].zero_()
        L.check(L.lib().pu_unet_bwd(o._ctx, L.ptr(dfeat.contiguous().float()), o._stream()), o._ctx, "pu_unet_bwd")
        o._deliver(None, lo, hi)
        return None, None, None


class _GaussFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, anchor, owner, which):
        ctx.owner, ctx.which = owner, which
        mu, ls = owner._gauss_fwd(which, x, target)
        ctx.name = "posterior" if which == L.PU_POSTERIOR else "prior"
        ctx.gen, ctx.xin = owner._gen[ctx.name], owner._xin_token
        return mu, ls

    @staticmethod
    def backward(ctx, dmu, dls):
        o = ctx.owner
        o._check_fresh(ctx.name, ctx.gen, f"model.{ctx.name}(...)")
        if ctx.which == L.PU_PRIOR and o._xin_token != ctx.xin:
            raise L.ProbUNetLibraryError("stale engine state: the input planes saved by model.prior(x) were overwritten by a later "
                                         "model.unet(x') call with a different x before backward() (one forward in flight per engine)")
        o._gen["grads"] += 1
        lo, hi = o._ranges[ctx.name]
        o._unalias_grads(lo, hi)
        o._engine_grads[lo:hi].zero_()
        dmu = torch.zeros_like(o._last_mu[ctx.which]) if dmu is None else dmu.contiguous().float()
        dls = torch.zeros_like(dmu) if dls is None else dls.contiguous().float()
        L.check(L.lib().pu_gauss_bwd(o._ctx, ctx.which, L.ptr(dmu), L.ptr(dls), o._stream()), o._ctx, "pu_gauss_bwd")
        o._deliver(None, lo, hi)
        return None, None, None, None, None


class _FcombFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, z, anchor, owner):
        ctx.owner = owner
        ctx.shape = tuple(feat.shape)
        out = owner._fcomb_fwd(feat, z)
        ctx.gen = owner._gen["fcomb"]
        return out

    @staticmethod
    def backward(ctx, dout):
        o = ctx.owner
        o._check_fresh("fcomb", ctx.gen, "model.fcomb(...)")
        o._gen["grads"] += 1
        lo, hi = o._ranges["fcomb"]
        o._unalias_grads(lo, hi)
        o._engine_grads[lo:hi].zero_()
        B = dout.shape[0]
        dfeat = torch.empty(ctx.shape, device=dout.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dz = torch.empty(B, o.latent_dim, device=dout.device, dtype=torch.float32)
        L.check(L.lib().pu_fcomb_bwd(o._ctx, L.ptr(dout.contiguous().float()), L.ptr(dfeat), L.ptr(dz), o._stream()),
                o._ctx, "pu_fcomb_bwd")
        o._deliver(None, lo, hi)
        return dfeat, dz, None, None


# ----------------------------------------------------------------------------------------------- sub-modules
class _UNetModule(nn.Module):
    """model.unet(x) -> [B, F0, H, W]  (networks.py:299-333)."""

    def forward(self, x):
        o = self._owner()
        return _UNetFn.apply(x, o._anchor_t(), o)


class _GaussianModule(nn.Module):
    """model.prior(x) / model.posterior(x, target) -> Independent(Normal(mu, exp(log_sigma) + 1e-7), 1)
    (prob_unet.py:56-85)."""

    def forward(self, x, target=None):
        o = self._owner()
        which = L.PU_POSTERIOR if self._posterior else L.PU_PRIOR
        if which == L.PU_POSTERIOR and target is None:
            raise ValueError("posterior needs a target")
        mu, ls = _GaussFn.apply(x, target if which == L.PU_POSTERIOR else None, o._anchor_t(), o, which)
        return Independent(Normal(loc=mu, scale=torch.exp(ls) + 1e-7), 1)


class _FcombModule(nn.Module):
    """model.fcomb(feature_map, z) -> [B, Cout, H, W]  (prob_unet.py:120-138)."""

    def tile(self, a, dim, n_tile):
        """TensorFlow-style tile (prob_unet.py:109-118): element i of `dim` repeated n_tile times in place."""
        return a.repeat_interleave(n_tile, dim=dim)

    def forward(self, feature_map, z):
        o = self._owner()
        return _FcombFn.apply(feature_map, z, o._anchor_t(), o)


# ----------------------------------------------------------------------------------------------- the model
class ProbabilisticUNet(nn.Module):
    """MI355X engine behind the reference constructor signature (prob_unet.py:146).

    Extra keyword-only arguments (all optional): dtype ("f32" parity path | "f16" | "bf16" MFMA paths),
    max_batch / max_members (engine planning; grown on demand), recon ("afcrps" | "l1" | "wmse_msssim"), dropout.
    """

    def __init__(self, input_channels, num_classes, latent_dim, num_filters, model_channels, channel_mult,
                 beta_0, beta_1, beta_2, *, dtype: str = "f32", max_batch: int = 0, max_members: int = 0,
                 recon: str = "afcrps", dropout: float = 0.10, init: bool = True, grad_scale: float = 0.0):
        super().__init__()
        self.input_channels = int(input_channels)
        self.num_classes = int(num_classes)
        self.latent_dim = int(latent_dim)
        self.num_filters = [int(v) for v in num_filters]
        self.model_channels = int(model_channels)
        self.channel_mult = [int(v) for v in channel_mult]
        self.beta_0, self.beta_1, self.beta_2 = beta_0, beta_1, beta_2
        if len(self.num_filters) != len(self.channel_mult):
            raise ValueError("num_filters and channel_mult must have the same length")
        if dtype not in L.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(L.DTYPES)}")
        self.compute_dtype = dtype
        if recon not in ("afcrps", "l1", "wmse_msssim"):
            raise ValueError('recon must be "afcrps", "l1" or "wmse_msssim"')
        self.recon = recon
        self.dropout = float(dropout)
        self.grad_scale = float(grad_scale)   # f16 static loss scale (0 = sized automatically from B, M, C, H, W)
        self.sync_scalars = True          # reference returns python floats (.item()); set False to keep device scalars
        self.assume_static_parameters = False   # True: forward calls do not re-pack the weights (inference loops; see _params_dirty)
        self._packed_once = False
        self._pls = None                  # prior_latent_space / posterior_latent_space: a distribution, None, or a pending
        self._qls = None                  # fetch (int B) filled from the engine on first access after a fused elbo()
        self._want_batch, self._want_members = int(max_batch), int(max_members)
        self._ctx = None
        self._ctx_key = None
        self._flat = None                 # engine-bound flat fp32 parameters
        self._eg = [None, None]           # engine-written flat fp32 gradient buffers (two: p.grad may keep referencing the one written by
        self._eg_cur = 0                  # the previous backward while the next fused call writes the other), index of the bound one,
        self._eg_flag = [False, False]    # and whether parameters' .grad may still be views of each
        self._flat_grad = None            # what p.grad aliases (stable across calls)
        self._anchor = None
        self._step = 0
        self._last_mu = {}
        self._dp_group = None
        self._dp_world = 1
        self._dp_active = False           # gradients go through the process group (world > 1, or a single-rank group when asked for)
        self.dp_bucket_elems = 0
        self.dp_overlap_buckets = 4       # U-Net gradient buckets all-reduced under the rest of the backward (0: one all-reduce in backward())
        self._dp_works = None             # in-flight bucket collectives of the last fused elbo()
        self.dp_global_data_range = True  # WMSE-MS-SSIM with data_range=None under data parallelism: range of the GLOBAL batch (min / max all-reduced)
        self._range_dev = None
        self.dp_wire_dtype = None         # None: fp32 all-reduce (default); "bf16": gradients cross the wire as bfloat16 (optional compression)
        self._dp_wire = None              # persistent bf16 staging buffer of that mode
        self._comm_stream = None
        self.use_sample_graph = True      # cfg5: pu_sample / pu_sample_hr launch sequences are captured in a hipGraph and replayed
        # one forward in flight per engine: every entry that overwrites saved engine state bumps its generation; backward checks it
        self._gen = {"unet": 0, "prior": 0, "posterior": 0, "fcomb": 0, "grads": 0}
        self._xin_token = None            # identity of the tensor last converted into the engine's shared input planes
        self._drop_masks = None

        # ---- parameter tree from the engine's own table (names/shapes/order of the reference state_dict)
        table = self._query_table()
        self._table = table
        self.unet = _UNetModule(); self.prior = _GaussianModule(); self.posterior = _GaussianModule(); self.fcomb = _FcombModule()
        self.prior._posterior = False; self.posterior._posterior = True
        import weakref
        ref = weakref.ref(self)
        for m in (self.unet, self.prior, self.posterior, self.fcomb):
            object.__setattr__(m, "_owner", ref)
        for name, shape, off, is_buf in table:
            parts = name.split(".")
            node = getattr(self, parts[0])
            for i, p in enumerate(parts[1:-1]):
                if p in ("enc", "dec"):
                    cls = nn.ModuleDict
                elif p in ("encoder", "layers"):
                    cls = _IndexedHolders
                else:
                    cls = _Holder
                if isinstance(node, nn.ModuleDict):
                    if p not in node:
                        node[p] = cls()
                    node = node[p]
                else:
                    node = _ensure_child(node, p, cls)
            if is_buf:
                node.register_buffer(parts[-1], torch.full(shape, 0.25))
            else:
                node.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
        self._ranges = {}
        for pre in ("unet", "prior", "posterior", "fcomb"):
            offs = [(off, int(np.prod(shape))) for name, shape, off, is_buf in table if not is_buf and name.startswith(pre + ".")]
            self._ranges[pre] = (min(o for o, _ in offs), max(o + n for o, n in offs))
        self._nparams = max(hi for _, hi in self._ranges.values())
        # the metadata scripts read off the leaf modules (latent_exploration.py:296: `model.fcomb.layers[0].in_channels`)
        for mod in self.modules():
            w = mod._parameters.get("weight") if hasattr(mod, "_parameters") else None
            if w is not None and w.dim() == 4:
                mod.out_channels, mod.in_channels = int(w.shape[0]), int(w.shape[1])
                mod.kernel_size = (int(w.shape[2]), int(w.shape[3])); mod.stride = (1, 1); mod.padding = (int(w.shape[2]) // 2, int(w.shape[3]) // 2)
            elif w is not None and w.dim() == 2:
                mod.out_features, mod.in_features = int(w.shape[0]), int(w.shape[1])
            elif w is not None and w.dim() == 1:
                mod.num_channels = int(w.shape[0])
        if init:
            _init_reference_order(self)

    # ------------------------------------------------------------------ engine management
    # side-effect attributes of the reference (prob_unet.py:214,220,241-242).  After the fused elbo() they are materialised
    # lazily: the engine still holds (mu, log_sigma) of its last forward, nothing is copied unless somebody looks.
    def _latent_dist(self, which, B):
        dev = self._owner_device()
        mu = torch.empty(B, self.latent_dim, device=dev, dtype=torch.float32); sg = torch.empty_like(mu)
        L.check(L.lib().pu_last_latent(self._ctx, which, L.ptr(mu), L.ptr(sg), B, self._stream()), self._ctx, "pu_last_latent")
        return Independent(Normal(loc=mu, scale=sg), 1)

    @property
    def prior_latent_space(self):
        if isinstance(self._pls, int):
            self._pls = self._latent_dist(L.PU_PRIOR, self._pls)
        return self._pls

    @prior_latent_space.setter
    def prior_latent_space(self, v):
        self._pls = v

    @property
    def posterior_latent_space(self):
        if isinstance(self._qls, int):
            self._qls = self._latent_dist(L.PU_POSTERIOR, self._qls)
        return self._qls

    @posterior_latent_space.setter
    def posterior_latent_space(self, v):
        self._qls = v

    @property
    def _engine_grads(self):
        return self._eg[self._eg_cur]

    @_engine_grads.setter
    def _engine_grads(self, v):
        self._eg = [v, None]; self._eg_cur = 0; self._eg_flag = [False, False]

    def _any_grad_alias(self, k):
        buf = self._eg[k]
        if buf is None:
            return False
        base = buf.data_ptr()
        for p, off, n in self._params_in(0, self._nparams):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * off:
                return True
        return False

    def _prepare_grad_buffer(self):
        """Before a fused backward overwrites the engine's gradient buffer: if parameters' .grad still are views of it (the reference's
        step order is elbo -> zero_grad -> backward, so at elbo() time the previous step's gradients are still attached), switch the
        engine to the other buffer instead of copying them out; only when both buffers are referenced (gradients kept across two
        calls) are the current one's views moved to the stable flat buffer."""
        cur = self._eg_cur
        if not self._eg_flag[cur]:
            return
        if not self._any_grad_alias(cur):
            self._eg_flag[cur] = False
            return
        other = 1 - cur
        if self._eg[other] is None:
            self._eg[other] = torch.zeros_like(self._eg[cur])
        if self._eg_flag[other] and self._any_grad_alias(other):
            self._unalias_grads()
            return
        self._eg_flag[other] = False
        self._eg_cur = other
        L.check(L.lib().pu_bind_grads(self._ctx, L.ptr(self._eg[other])), self._ctx, "pu_bind_grads")

    def _cfg_struct(self, H, W, max_batch, max_members):
        cfg = L.PuConfig()
        cfg.input_channels, cfg.num_classes, cfg.latent_dim = self.input_channels, self.num_classes, self.latent_dim
        cfg.depth = len(self.num_filters)
        for i, v in enumerate(self.num_filters): cfg.num_filters[i] = v
        for i, v in enumerate(self.channel_mult): cfg.channel_mult[i] = v
        cfg.model_channels = self.model_channels
        cfg.H, cfg.W, cfg.max_batch, cfg.max_members = H, W, max_batch, max_members
        cfg.dtype = L.DTYPES[self.compute_dtype]
        cfg.dropout_p = self.dropout
        cfg.grad_scale = float(getattr(self, "grad_scale", 0.0))
        return cfg

    def _query_table(self):
        """Ask the library for the reference-ordered parameter table (needs no GPU: planning only... but pu_create
        allocates, so a tiny throw-away context is avoided by a pure-host table query when no device is present)."""
        return _param_table_host(self)

    def _owner_device(self):
        return next(self.parameters()).device

    def _stream(self):
        return L.current_stream(self._owner_device())

    def _check_fresh(self, kind, gen, what):
        if self._gen[kind] != gen:
            raise L.ProbUNetLibraryError(
                f"stale engine state: {what} was followed by another call that overwrote the engine's saved "
                f"{'gradients' if kind == 'grads' else 'activations'} before backward().  The engine keeps ONE forward in flight "
                "(static activation plan): call backward() before the next forward of the same sub-module / elbo(), or run the "
                "second forward under torch.no_grad().")

    def _touch_xin(self, x):
        self._xin_token = (x.data_ptr(), x._version, tuple(x.shape))

    def set_drop_masks(self, masks):
        """Inject dropout keep-masks (parity runs against a reference that draws them from torch's RNG, networks.py:177).
        masks: {block prefix (e.g. "unet.enc.128x128_block0"): 0/1 tensor [B, C, H, W]} for EVERY UNetBlock, or None to return
        to the engine's counter-hash stream.  They apply to train-mode calls with that batch size until cleared."""
        self._drop_masks = masks
        if self._ctx is not None:
            self._push_drop_masks()

    def _push_drop_masks(self):
        masks = self._drop_masks
        lib = L.lib()
        if masks is None:
            L.check(lib.pu_set_drop_masks(self._ctx, None, 0, self._stream()), self._ctx, "pu_set_drop_masks")
            return
        n = lib.pu_drop_site_count(self._ctx)
        parts, B = [], None
        for i in range(n):
            name = C.create_string_buffer(96); c_, h_, w_ = C.c_int(), C.c_int(), C.c_int()
            L.check(lib.pu_drop_site(self._ctx, i, name, C.byref(c_), C.byref(h_), C.byref(w_)), self._ctx, "pu_drop_site")
            key = name.value.decode()
            if key not in masks:
                raise ValueError(f"set_drop_masks: no mask for dropout site {key}")
            m = masks[key]
            if B is None:
                B = int(m.shape[0])
            if tuple(m.shape) != (B, c_.value, h_.value, w_.value):
                raise ValueError(f"set_drop_masks: mask of {key} must be [{B}, {c_.value}, {h_.value}, {w_.value}], got {tuple(m.shape)}")
            parts.append(m.to(self._owner_device(), torch.float32).reshape(-1))
        flat = torch.cat(parts).contiguous()
        L.check(lib.pu_set_drop_masks(self._ctx, L.ptr(flat), B, self._stream()), self._ctx, "pu_set_drop_masks")
        self._mask_keepalive = flat

    def _ensure(self, H, W, B, M):
        dev = self._owner_device()
        if dev.type != "cuda":
            raise L.ProbUNetLibraryError("ProbabilisticUNet parameters must live on a ROCm device (model.to('cuda')); "
                                         "this engine has no CPU path")
        mb = max(B, self._want_batch, self._ctx_key[2] if self._ctx_key else 0)
        mm = max(M, self._want_members, self._ctx_key[3] if self._ctx_key else 0, 1)
        key = (H, W, mb, mm, self.compute_dtype, dev.index or 0, self.dropout, float(getattr(self, "grad_scale", 0.0)))
        if self._ctx is not None and self._ctx_key == key:
            self._check_views()
            return
        self._release()
        cfg = self._cfg_struct(H, W, mb, mm)
        ctx = C.c_void_p()
        L.check(L.lib().pu_create(C.byref(cfg), dev.index or 0, C.byref(ctx)), None, "pu_create")
        self._ctx, self._ctx_key = ctx, key
        self._packed_once = False
        for k in self._gen:                      # a new plan: every saved activation / gradient of the old one is gone
            self._gen[k] += 1
        self._xin_token = None
        n = L.lib().pu_param_count(ctx)
        if n != self._nparams:
            raise L.ProbUNetLibraryError(f"parameter count mismatch: engine {n} vs host {self._nparams}")
        self._flatten(dev)
        L.check(L.lib().pu_bind_params(ctx, L.ptr(self._flat), L.ptr(self._engine_grads)), ctx, "pu_bind_params")
        L.lib().pu_set_sample_graph(ctx, 1 if self.use_sample_graph else 0)
        if self._dp_active and self.dp_overlap_buckets > 0:
            L.lib().pu_set_grad_buckets(ctx, int(self.dp_overlap_buckets))
        if self._drop_masks is not None:
            self._push_drop_masks()

    def _release(self):
        if self._ctx is not None:
            torch.cuda.synchronize()
            L.lib().pu_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _flatten(self, dev):
        """Make every nn.Parameter a view of one flat fp32 device buffer in the engine's order."""
        if self._flat is None or self._flat.device != dev:
            flat = torch.empty(self._nparams, device=dev, dtype=torch.float32)
            self._engine_grads = torch.zeros(self._nparams, device=dev, dtype=torch.float32)
            self._flat_grad = torch.zeros(self._nparams, device=dev, dtype=torch.float32)
        else:
            flat = self._flat
        P = dict(self.named_parameters())
        with torch.no_grad():
            for name, shape, off, is_buf in self._table:
                if is_buf:
                    continue
                p = P[name]
                n = p.numel()
                if p.data_ptr() != flat.data_ptr() + 4 * off:
                    flat[off:off + n].copy_(p.data.reshape(-1).to(dev, torch.float32))
                    p.data = flat[off:off + n].view(shape)
        self._flat = flat
        self._probe = [(P[name], off) for name, shape, off, is_buf in (self._table[1], self._table[len(self._table) // 2], self._table[-1]) if not is_buf]

    def _check_views(self):
        base = self._flat.data_ptr()
        for p, off in self._probe:
            if p.data_ptr() != base + 4 * off:
                self._flatten(self._owner_device())
                L.check(L.lib().pu_bind_params(self._ctx, L.ptr(self._flat), L.ptr(self._engine_grads)), self._ctx, "pu_bind_params")
                return

    def _anchor_t(self):
        if self._anchor is None or self._anchor.device != self._owner_device():
            self._anchor = torch.zeros(1, device=self._owner_device(), requires_grad=True)
        return self._anchor

    def _params_dirty(self, force: bool = True):
        """Tell the engine that the fp32 parameters may have changed (it re-packs its compute-dtype copies, 0.4 ms at cfg3).
        The forward entry points call this with force=False: skipped when `assume_static_parameters` is set (an explicit opt-in
        for sampling loops - torch cannot tell us about writes through `.data` - that optimizers, load_state_dict and
        enable_data_parallel override by forcing)."""
        if force or not self.assume_static_parameters or not self._packed_once:
            L.lib().pu_params_changed(self._ctx)
            self._packed_once = True

    def _deliver(self, g, lo, hi):
        """Engine gradients [lo, hi) (x grad_output g) -> p.grad (accumulating like autograd does).
        Under data parallelism the engine gradients are first averaged over the process group (RCCL all-reduce).
        When no parameter of the range has a gradient yet, p.grad becomes a VIEW of the engine's gradient buffer: no copy - the
        grad_output (a device scalar, 1 for a plain loss.backward()) and the 1 / world of the data-parallel mean are applied in place
        by one conditional pass that does nothing when the factor is exactly 1 (pu_scale_grads).  The next call that makes the engine
        rewrite its buffer first moves gradients that are still referenced out of it (`_unalias_grads`)."""
        eg = self._engine_grads[lo:hi]
        P = self._params_in(lo, hi)
        fresh = all(p.grad is None for p, _, _ in P)
        host_factor = 1.0
        if self._dp_active:
            from .dp import allreduce_mean_
            if self._dp_works is not None and lo == 0 and hi == self._nparams:
                self._finish_dp_works()                   # bucketed collectives issued by elbo(): only wait for them here
                host_factor = 1.0 / self._dp_world
            else:
                allreduce_mean_(eg, self._dp_group, self.dp_bucket_elems, average=False, wire_dtype=self._wire_dtype())
                host_factor = 1.0 / self._dp_world
            # an overflow on ANY rank makes the SUM non-finite on EVERY rank (inf + x = inf, inf - inf = NaN): the optimizer derives
            # its skip flag from the averaged buffer (FlatAdamW.step), so all ranks skip together without another collective
        if fresh:
            gs = g.reshape(()).float().contiguous() if g is not None else None
            if gs is not None or host_factor != 1.0:
                if eg.data_ptr() % 16 == 0:
                    L.check(L.lib().pu_scale_grads(L.ptr(eg), hi - lo, L.ptr(gs), float(host_factor), self._stream()), self._ctx, "pu_scale_grads")
                else:                                     # a sub-module range that does not start on a 16-byte boundary
                    eg.mul_(gs * host_factor if gs is not None else host_factor)
            for (p, off, n), v in zip(P, self._grad_views(lo, hi, self._engine_grads)):
                p.grad = v
            self._eg_flag[self._eg_cur] = True
        else:
            if host_factor != 1.0:
                eg = eg * host_factor
            if g is not None:
                eg = eg * g.reshape(())
            for p, off, n in P:
                ge = eg[off - lo:off - lo + n].view(p.shape)
                if p.grad is None:
                    p.grad = ge.clone()
                else:
                    p.grad.add_(ge)

    def _unalias_grads(self, lo=0, hi=None):
        """Called before the engine rewrites its gradient buffer in [lo, hi): parameters whose .grad still is a view of that buffer
        (no zero_grad() since the last backward - gradient accumulation) get their gradient moved to the stable flat buffer."""
        if self._engine_grads is None or not self._eg_flag[self._eg_cur]:
            return
        hi = self._nparams if hi is None else hi
        base = self._engine_grads.data_ptr()
        P = self._params_in(lo, hi)
        views = None
        moved = False
        for i, (p, off, n) in enumerate(P):
            gr = p.grad
            if gr is not None and gr.data_ptr() == base + 4 * off:
                if views is None:
                    views = self._grad_views(lo, hi, self._flat_grad)
                if not moved:
                    self._flat_grad[lo:hi].copy_(self._engine_grads[lo:hi])      # one pass; ranges without a live view are harmless
                    moved = True
                p.grad = views[i]
        if lo == 0 and hi == self._nparams:
            self._eg_flag[self._eg_cur] = False

    def _start_bucket_allreduce(self):
        """Issue one all-reduce (SUM) per gradient bucket of the fused backward on a side stream that waits for the bucket's
        completion events (pu_grad_bucket_wait): the collectives run while the GPU is still working through the rest of the
        backward.  Called right after pu_elbo_fwd_bwd returned (everything is enqueued by then); `_deliver` waits for them."""
        import torch.distributed as dist
        lib = L.lib()
        n = C.c_int(0)
        lo = (C.c_int64 * 32)(); hi = (C.c_int64 * 32)()
        L.check(lib.pu_grad_buckets(self._ctx, lo, hi, 32, C.byref(n)), self._ctx, "pu_grad_buckets")
        if n.value == 0:
            return
        dev = self._owner_device()
        if self._comm_stream is None or self._comm_stream.device != dev:
            self._comm_stream = torch.cuda.Stream(device=dev)
        cs = self._comm_stream
        works = []
        wd = self._wire_dtype()
        if wd is not None and (self._dp_wire is None or self._dp_wire.numel() != self._engine_grads.numel() or self._dp_wire.device != dev):
            self._dp_wire = torch.empty(self._engine_grads.numel(), dtype=wd, device=dev)      # persistent: used on the comm streams
        with torch.cuda.stream(cs):
            for k in range(n.value):
                L.check(lib.pu_grad_bucket_wait(self._ctx, k, C.c_void_p(cs.cuda_stream)), self._ctx, "pu_grad_bucket_wait")
                seg = self._engine_grads[lo[k]:hi[k]]
                if wd is not None:                      # compress on the comm stream, behind the bucket's events
                    wire = self._dp_wire[lo[k]:hi[k]]
                    wire.copy_(seg)
                    works.append((dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self._dp_group, async_op=True), seg, wire))
                else:
                    works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self._dp_group, async_op=True), None, None))
        self._dp_works = works

    def _dp_bucket_ranges(self):
        n = C.c_int(0); lo = (C.c_int64 * 32)(); hi = (C.c_int64 * 32)()
        if self._ctx is None:
            return []
        L.check(L.lib().pu_grad_buckets(self._ctx, lo, hi, 32, C.byref(n)), self._ctx, "pu_grad_buckets")
        return [(int(lo[k]), int(hi[k])) for k in range(n.value)]

    def _finish_dp_works(self):
        if self._dp_works is not None:
            for w, seg, wire in self._dp_works:
                w.wait()                                  # the current stream waits for the collective's stream (no host sync with RCCL)
                if wire is not None:
                    seg.copy_(wire)                       # reduced bf16 values back into the fp32 gradient buffer
            self._dp_works = None

    def _wire_dtype(self):
        if self.dp_wire_dtype in (None, "", "f32", "fp32", torch.float32):
            return None
        if self.dp_wire_dtype in ("bf16", "bfloat16", torch.bfloat16):
            return torch.bfloat16
        raise ValueError(f"dp_wire_dtype must be None or 'bf16', got {self.dp_wire_dtype!r}")

    def _grad_views(self, lo, hi, buf=None):
        buf = self._flat_grad if buf is None else buf
        key = ("gv", lo, hi, buf.data_ptr())
        cache = self.__dict__.setdefault("_pin_cache", {})
        if key not in cache:
            cache[key] = [buf[off:off + n].view(p.shape) for p, off, n in self._params_in(lo, hi)]
        return cache[key]

    def _params_in(self, lo, hi):
        key = (lo, hi)
        cache = self.__dict__.setdefault("_pin_cache", {})
        if key not in cache:
            P = dict(self.named_parameters())
            cache[key] = [(P[name], off, int(np.prod(shape))) for name, shape, off, is_buf in self._table
                          if not is_buf and lo <= off < hi]
        return cache[key]

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__.pop("_pin_cache", None)
        self._anchor = None
        return r

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        k = "posterior.encoder.0.weight"
        if k in sd:
            want = dict(self.named_parameters())[k].shape[1]
            if sd[k].shape[1] != want:      # reference checkpoints carry 2*Cin planes; extra planes only ever saw zeros
                sd[k] = sd[k][:, :want]
        r = super().load_state_dict(sd, strict=strict, **kw)
        if self._ctx is not None:
            self._params_dirty()
        return r

    # ------------------------------------------------------------------ data parallel
    def enable_data_parallel(self, process_group=None, single_rank_ok=False):
        """One process per GPU; gradients are averaged with a torch.distributed all-reduce (backend 'nccl' == RCCL
        over xGMI) on the flat gradient buffer before they reach p.grad.  Parameters are broadcast from rank 0.
        single_rank_ok: keep the collective path on even for a group of one rank (the RCCL smoke test of a one-GPU box:
        communicator creation, the flat broadcast and the bucketed all-reduce behind the engine's events all run for real)."""
        import torch.distributed as dist
        self._dp_group = process_group
        self._dp_world = dist.get_world_size(process_group)
        self._dp_active = self._dp_world > 1 or bool(single_rank_ok)
        if self._dp_active:
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            dev = self._owner_device()
            if dev.type == "cuda":
                if self._flat is None:
                    self._flatten(dev)                    # every nn.Parameter becomes a view of the flat buffer: ONE broadcast of 303 MB
                dist.broadcast(self._flat, src=src, group=process_group)
            else:                                         # parameters still on the host (model.to(device) comes later): per tensor
                for p in self.parameters():
                    dist.broadcast(p.data, src=src, group=process_group)
            if self._ctx is not None:
                self._params_dirty()
                if self.dp_overlap_buckets > 0:
                    L.lib().pu_set_grad_buckets(self._ctx, int(self.dp_overlap_buckets))
        return self

    # ------------------------------------------------------------------ raw engine calls
    def _prep(self, x):
        if x.dim() != 4:
            raise ValueError(f"expected [B, C, H, W], got {tuple(x.shape)}")
        return x.contiguous().float()

    def _unet_fwd(self, x):
        x = self._prep(x)
        B, Cc, H, W = x.shape
        if Cc != self.input_channels:
            raise ValueError(f"expected {self.input_channels} input planes, got {Cc}")
        self._ensure(H, W, B, 1)
        self._params_dirty(force=False)
        feat = torch.empty(B, self.num_filters[0], H, W, device=x.device, dtype=torch.float32)
        train = 1 if (self.training and self.dropout > 0) else 0
        self._gen["unet"] += 1; self._touch_xin(x)
        L.check(L.lib().pu_unet_fwd(self._ctx, L.ptr(x), L.ptr(feat), B, train, self._next_seed(), self._stream()), self._ctx, "pu_unet_fwd")
        return feat

    def _gauss_fwd(self, which, x, target):
        x = self._prep(x)
        B, Cc, H, W = x.shape
        self._ensure(H, W, B, 1)
        self._params_dirty(force=False)
        if which == L.PU_POSTERIOR:
            target = self._prep(target)
            if target.shape[1] != self.num_classes:
                if target.shape[1] == self.input_channels and self.num_classes < self.input_channels:
                    target = target[:, : self.num_classes].contiguous()   # reference-style zero-padded target
                else:
                    raise ValueError(f"expected {self.num_classes} target planes, got {target.shape[1]}")
        mu = torch.empty(B, self.latent_dim, device=x.device, dtype=torch.float32)
        ls = torch.empty_like(mu)
        self._gen["posterior" if which == L.PU_POSTERIOR else "prior"] += 1
        if which == L.PU_PRIOR:
            self._touch_xin(x)
        L.check(L.lib().pu_gauss_fwd(self._ctx, which, L.ptr(x), L.ptr(target) if which == L.PU_POSTERIOR else None, L.ptr(mu), L.ptr(ls), B,
                                     self._stream()), self._ctx, "pu_gauss_fwd")
        self._last_mu[which] = mu
        return mu, ls

    def _fcomb_fwd(self, feat, z):
        if feat.dim() != 4 or z.dim() != 2:
            raise ValueError("fcomb(feature_map [B,F0,H,W], z [B,L])")
        B, F0, H, W = feat.shape
        if z.shape[0] != B or z.shape[1] != self.latent_dim or F0 != self.num_filters[0]:
            raise ValueError(f"fcomb shape mismatch: feature_map {tuple(feat.shape)}, z {tuple(z.shape)}")
        if feat.stride(0) == 0 and feat[0].is_contiguous():
            src, bstride = feat[0], 0                      # expand()-ed view: one feature map broadcast to the batch
        else:
            src = feat.contiguous(); bstride = F0 * H * W
        src = src.float()
        self._ensure(H, W, B, 1)
        self._params_dirty(force=False)
        out = torch.empty(B, self.num_classes, H, W, device=feat.device, dtype=torch.float32)
        self._gen["fcomb"] += 1
        L.check(L.lib().pu_fcomb_fwd(self._ctx, L.ptr(src), bstride, L.ptr(z.contiguous().float()), L.ptr(out), B, self._stream()),
                self._ctx, "pu_fcomb_fwd")
        return out

    def _next_seed(self):
        self._step += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._step * 0xD1B54A32D192ED03 + (self._dp_rank() << 48)) & 0xFFFFFFFFFFFFFFFF

    def _dp_rank(self):
        if self._dp_world > 1:
            import torch.distributed as dist
            return dist.get_rank(self._dp_group)
        return 0

    # ------------------------------------------------------------------ reference API
    def forward(self, x, target=None, t=None, training=True):
        """prob_unet.py:194-224.  `training` is an argument (independent of self.training); `t` is ignored."""
        if not torch.is_grad_enabled():
            use_post = bool(training) and target is not None
            r = self.sample(x, 1, target=target if use_post else None, _return_dist=True)
            out, dist_ = r
            if use_post: self.posterior_latent_space = dist_
            else: self.prior_latent_space = dist_
            return out[:, 0]
        unet_features = self.unet(x)
        if training and target is not None:
            self.posterior_latent_space = self.posterior(x, target)
            z = self.posterior_latent_space.rsample()
        else:
            self.prior_latent_space = self.prior(x)
            z = self.prior_latent_space.rsample()
        return self.fcomb(unet_features, z)

    def elbo(self, x, target, t=None, M: Optional[int] = None, alpha: float = 0.95, eps: Optional[torch.Tensor] = None,
             alpha_w: float = 0.007, beta_w: float = 0.048, lam_w: float = 0.0, data_range: Optional[float] = None):
        """Fused ELBO forward(+backward when grad is enabled).

        recon == "afcrps" (prob_unet.py:273-317; what train_prob_unet_model.py:133 unpacks), M defaults to 5:
            returns (total_loss, [crps], kl_div[B])
        recon == "l1" (prob_unet.py:325-381): returns (total_loss, [l1], kl_div[B], kl_div2[B])
        recon == "wmse_msssim" (the live elbo, prob_unet.py:229-267), M defaults to 1 as there; alpha_w / beta_w / lam_w are its
            keyword arguments; returns (total_loss, [recon], kl_div[B], wmse, msssim) with the last member's wmse and
            (1 - MS-SSIM).  data_range=None infers max(target) - min(target) on the device (prob_unet_utils.py:288-289).
        eps: optional explicit reparameterisation noise [M, B, L] (default: torch.randn on the device generator).
        """
        x = self._prep(x); target = self._prep(target)
        B, Cc, H, W = x.shape
        if Cc != self.input_channels:
            raise ValueError(f"expected {self.input_channels} input planes, got {Cc}")
        if target.shape[1] != self.num_classes:
            if target.shape[1] == self.input_channels and self.num_classes < self.input_channels:
                target = target[:, : self.num_classes].contiguous()
            else:
                raise ValueError(f"expected {self.num_classes} target planes, got {target.shape[1]}")
        afcrps = self.recon == "afcrps"
        msssim = self.recon == "wmse_msssim"
        if M is None:
            M = 5 if afcrps else 1
        if afcrps and M < 2:
            raise ValueError(f"M must be at least 2 to compute afCRPS but got M={M}")
        if msssim and min(H, W) <= 96:
            raise AssertionError("Image size should be larger than 96 due to the 4 downsamplings in ms-ssim")
        Mx = M if (afcrps or msssim) else 1
        self._ensure(H, W, B, Mx)
        self._params_dirty(force=False)
        if eps is None:
            eps = torch.randn(Mx, B, self.latent_dim, device=x.device, dtype=torch.float32)
        eps = eps.contiguous().float()
        if tuple(eps.shape) != (Mx, B, self.latent_dim):
            raise ValueError(f"eps must be [{Mx}, {B}, {self.latent_dim}]")
        with_bwd = 1 if torch.is_grad_enabled() else 0
        scal = torch.empty(L.PU_NUM_SCALARS, device=x.device, dtype=torch.float32)
        klv = torch.empty(B, device=x.device, dtype=torch.float32)
        kl2v = torch.empty(B, device=x.device, dtype=torch.float32) if self.recon == "l1" else None
        train = 1 if (self.training and self.dropout > 0) else 0
        kind = L.PU_RECON_AFCRPS if afcrps else (L.PU_RECON_WMSE_MSSSIM if msssim else L.PU_RECON_L1)
        if msssim:
            L.check(L.lib().pu_set_recon_wmse_msssim(self._ctx, float(alpha_w), float(beta_w), float(lam_w),
                                                     -1.0 if data_range is None else float(data_range)), self._ctx, "pu_set_recon_wmse_msssim")
            rng_dev = None
            if data_range is None and self._dp_active and self.dp_global_data_range:
                # the reference infers max(target) - min(target) over the batch it sees (prob_unet_utils.py:288-289); under data
                # parallelism that batch is the GLOBAL one: exchange the shard minima / maxima (two scalar all-reduces, on the device,
                # no host round trip) and hand the range to the loss kernels through a device float
                import torch.distributed as dist
                mn, mx = torch.aminmax(target)
                dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self._dp_group)
                dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self._dp_group)
                rng_dev = (mx - mn).reshape(1).float().contiguous()
            self._range_dev = rng_dev                                  # kept alive until the next call replaces it
            L.check(L.lib().pu_set_recon_range_dev(self._ctx, L.ptr(rng_dev) if rng_dev is not None else None), self._ctx, "pu_set_recon_range_dev")
        for k in ("unet", "prior", "posterior"):          # the fused call overwrites every saved activation (and, with backward, the gradients)
            self._gen[k] += 1
        self._touch_xin(x)
        if with_bwd:
            self._gen["grads"] += 1
            self._finish_dp_works()                       # a previous elbo()'s collectives still own the gradient buffer
            self._prepare_grad_buffer()                   # gradients still referenced keep their buffer: the engine writes the other one
        L.check(L.lib().pu_elbo_fwd_bwd(self._ctx, L.ptr(x), L.ptr(target), L.ptr(eps), B, Mx, kind,
                                        float(self.beta_0), float(self.beta_1), float(self.beta_2), float(alpha), train,
                                        self._next_seed(), with_bwd, L.ptr(scal), L.ptr(klv), L.ptr(kl2v), self._stream()),
                self._ctx, "pu_elbo_fwd_bwd")
        total = scal[L.PU_S_TOTAL]
        if with_bwd:
            if self._dp_active and self.dp_overlap_buckets > 0:
                self._start_bucket_allreduce()            # enqueued now, runs under the rest of the backward on the GPU
            total = _DeliverGrads.apply(total, self._anchor_t(), self, 0, self._nparams)
        recon = scal[L.PU_S_RECON]
        self._last_scalars = scal
        self._pls, self._qls = int(B), int(B)          # prob_unet.py:241-242: both latent spaces are left behind by elbo()
        if self.sync_scalars:
            host = scal.tolist()                                  # one device->host sync for every logged scalar
            recon_list = [host[L.PU_S_RECON]]
        else:
            host = None
            recon_list = [recon]
        if afcrps:
            return total, recon_list, klv
        if msssim:
            if host is not None:
                return total, recon_list, klv, host[L.PU_S_WMSE], host[L.PU_S_MSSSIM]
            return total, recon_list, klv, scal[L.PU_S_WMSE], scal[L.PU_S_MSSSIM]
        return total, recon_list, klv, kl2v

    @torch.no_grad()
    def sample(self, x, n: int, target=None, eps: Optional[torch.Tensor] = None, _return_dist: bool = False, out: Optional[torch.Tensor] = None):
        """n samples per input with the U-Net and the latent encoder evaluated ONCE (the pattern of
        latent_exploration.py:119-129; replaces the n x model(x, training=False) loop of
        train_prob_unet_model.py:244-247).  Returns [B, n, Cout, H, W].
        out: optional preallocated fp32 result buffer of that shape.  With `use_sample_graph` the captured hipGraph is keyed on
        every pointer of the call, so a sampling loop that wants replays passes the same x / eps / out buffers each time
        (`sample_graph_stats()` reports what actually happened)."""
        x = self._prep(x)
        B, Cc, H, W = x.shape
        out_given = out is not None
        self._ensure(H, W, B, n)
        self._params_dirty(force=False)
        if target is not None:
            target = self._prep(target)
            if target.shape[1] != self.num_classes:
                target = target[:, : self.num_classes].contiguous()
        if eps is None:
            eps = torch.randn(n, B, self.latent_dim, device=x.device, dtype=torch.float32)
        eps = eps.contiguous().float()
        out = self._out_buffer(out, (B, n, self.num_classes, H, W), x.device)
        mu, sg = self._latent_out_buffers(B, x.device, stable=out_given)
        self._gen["unet"] += 1; self._gen["posterior" if target is not None else "prior"] += 1; self._touch_xin(x)
        L.check(L.lib().pu_sample(self._ctx, L.ptr(x), L.ptr(target), L.ptr(eps), B, n, L.ptr(out), L.ptr(mu), L.ptr(sg), self._stream()),
                self._ctx, "pu_sample")
        if _return_dist:
            return out, Independent(Normal(loc=mu, scale=sg), 1)
        return out

    @torch.no_grad()
    def sample_hr(self, x, n: int, lrinterp, residual_std, epsilon: float = 1e-10, softplus: bool = False, softplus_c: float = 1e-7,
                  target=None, eps: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """`sample()` with ClimExDataset.residual_to_hr (climex_utils.py:277-285) fused into the Fcomb store: returns
        physical-unit fields hr[b, s] = lrinterp[b] + residual[b, s] * (residual_std + epsilon), [B, n, Cout, H, W], without the
        per-sample `.cpu()` round trip of train_prob_unet_model.py:246.  softplus=True additionally applies
        climex_utils.softplus (:41-45), the inverse of the precipitation pre-transform."""
        x = self._prep(x)
        B, Cc, H, W = x.shape
        self._ensure(H, W, B, n)
        self._params_dirty(force=False)
        if target is not None:
            target = self._prep(target)
            if target.shape[1] != self.num_classes:
                target = target[:, : self.num_classes].contiguous()
        lrinterp = self._prep(lrinterp)
        if lrinterp.shape[1] != self.num_classes:
            lrinterp = lrinterp[:, : self.num_classes].contiguous()
        residual_std = residual_std.to(device=x.device, dtype=torch.float32)
        if residual_std.dim() == 4:
            residual_std = residual_std[0]
        residual_std = residual_std[: self.num_classes].contiguous()
        if tuple(lrinterp.shape) != (B, self.num_classes, H, W) or tuple(residual_std.shape) != (self.num_classes, H, W):
            raise ValueError("lrinterp must be [B, Cout, H, W] and residual_std [Cout, H, W]")
        if eps is None:
            eps = torch.randn(n, B, self.latent_dim, device=x.device, dtype=torch.float32)
        eps = eps.contiguous().float()
        out = self._out_buffer(out, (B, n, self.num_classes, H, W), x.device)
        self._gen["unet"] += 1; self._gen["posterior" if target is not None else "prior"] += 1; self._touch_xin(x)
        L.check(L.lib().pu_sample_hr(self._ctx, L.ptr(x), L.ptr(target), L.ptr(eps), B, n, L.ptr(lrinterp), L.ptr(residual_std),
                                     float(epsilon), 1 if softplus else 0, float(softplus_c), L.ptr(out), None, None, self._stream()),
                self._ctx, "pu_sample_hr")
        return out

    @staticmethod
    def _out_buffer(out, shape, device):
        if out is None:
            return torch.empty(*shape, device=device, dtype=torch.float32)
        if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or out.device != device or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous fp32 tensor of shape {tuple(shape)} on {device}")
        return out

    def _latent_out_buffers(self, B, device, stable):
        """(mu, sigma) result buffers of sample(); with a caller-provided `out` they are kept per batch size so that every pointer of
        the call repeats (the hipGraph cache key)."""
        if not stable:
            mu = torch.empty(B, self.latent_dim, device=device, dtype=torch.float32)
            return mu, torch.empty_like(mu)
        cache = self.__dict__.setdefault("_mu_sg_cache", {})
        k = (B, str(device))
        if k not in cache:
            mu = torch.empty(B, self.latent_dim, device=device, dtype=torch.float32)
            cache[k] = (mu, torch.empty_like(mu))
        return cache[k]

    def sample_graph_stats(self):
        """(captures, replays, eager_fallbacks) of the hipGraph sampling path since the engine context was created."""
        if self._ctx is None:
            return (0, 0, 0)
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(L.lib().pu_sample_graph_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)), self._ctx, "pu_sample_graph_stats")
        return (int(a.value), int(b.value), int(c.value))

    @staticmethod
    def reconstruct(residual, lrinterp, residual_std, epsilon: float = 1e-10):
        """climex_utils.py:277-285 (`lrinterp_to_residuals` datasets): hr = lrinterp + residual * (std + eps) on tensors that
        already exist (torch arithmetic; `sample_hr` is the fused on-device path)."""
        return lrinterp + residual * (residual_std + epsilon)

    def elbo_fwd_flops(self, B, M):
        return L.lib().pu_elbo_fwd_flops(self._ctx, B, M) if self._ctx is not None else float("nan")


# ----------------------------------------------------------------------------------------------- host-side table
def _param_table_host(m: ProbabilisticUNet):
    """The reference state_dict layout (names, shapes, flat offsets) — the host twin of build_plan() in
    csrc/engine.hip; pu_param_count() is cross-checked against it when the engine is created and
    tests/test_host_cpu.py compares it with the keys captured from the reference."""
    out = []
    off = 0

    def add(name, shape, buf=False):
        nonlocal off
        shape = tuple(int(s) for s in shape)
        if buf:
            out.append((name, shape, -1, True))
        else:
            out.append((name, shape, off, False))
            off += int(np.prod(shape))

    mc, mult, D = m.model_channels, m.channel_mult, len(m.channel_mult)
    emb = mc * 4
    add("unet.map_label.weight", (emb, 1))

    def block(p, cin, cout, up=False, down=False):
        add(p + ".norm0.weight", (cin,)); add(p + ".norm0.bias", (cin,))
        add(p + ".conv0.weight", (cout, cin, 3, 3)); add(p + ".conv0.bias", (cout,))
        if up or down: add(p + ".conv0.resample_filter", (1, 1, 2, 2), True)
        add(p + ".affine.weight", (2 * cout, emb)); add(p + ".affine.bias", (2 * cout,))
        add(p + ".norm1.weight", (cout,)); add(p + ".norm1.bias", (cout,))
        add(p + ".conv1.weight", (cout, cout, 3, 3)); add(p + ".conv1.bias", (cout,))
        if cin != cout:
            add(p + ".skip.weight", (cout, cin, 1, 1)); add(p + ".skip.bias", (cout,))
            if up or down: add(p + ".skip.resample_filter", (1, 1, 2, 2), True)
        elif up or down:
            add(p + ".skip.resample_filter", (1, 1, 2, 2), True)

    cout = m.input_channels
    skips = []
    for lv in range(D):
        r = 128 >> lv
        p = f"unet.enc.{r}x{r}"
        if lv == 0:
            cin, cout = cout, mc * mult[0]
            add(p + "_conv.weight", (cout, cin, 3, 3)); add(p + "_conv.bias", (cout,))
        else:
            block(p + "_down", cout, cout, down=True)
        skips.append(cout)
        for i in range(2):
            cin, cout = cout, mc * mult[lv]
            block(p + f"_block{i}", cin, cout)
            skips.append(cout)
    for lv in reversed(range(D)):
        r = 128 >> lv
        p = f"unet.dec.{r}x{r}"
        if lv == D - 1:
            block(p + "_in0", cout, cout); block(p + "_in1", cout, cout)
        else:
            block(p + "_up", cout, cout, up=True)
        for i in range(3):
            cin = cout + skips.pop(); cout = mc * mult[lv]
            block(p + f"_block{i}", cin, cout)
    add("unet.out_norm.weight", (cout,)); add("unet.out_norm.bias", (cout,))
    add("unet.out_conv.weight", (m.num_filters[0], cout, 3, 3)); add("unet.out_conv.bias", (m.num_filters[0],))
    for net, cin0 in (("prior", m.input_channels), ("posterior", m.input_channels + m.num_classes)):
        cin, idx = cin0, 0
        for lv in range(D):
            if lv: idx += 1
            for _ in range(3):
                add(f"{net}.encoder.{idx}.weight", (m.num_filters[lv], cin, 3, 3)); add(f"{net}.encoder.{idx}.bias", (m.num_filters[lv],))
                cin = m.num_filters[lv]; idx += 2
        add(f"{net}.conv_mu.weight", (m.latent_dim, m.num_filters[-1], 1, 1)); add(f"{net}.conv_mu.bias", (m.latent_dim,))
        add(f"{net}.conv_log_sigma.weight", (m.latent_dim, m.num_filters[-1], 1, 1)); add(f"{net}.conv_log_sigma.bias", (m.latent_dim,))
    f0 = m.num_filters[0]
    add("fcomb.layers.0.weight", (f0, f0 + m.latent_dim, 1, 1)); add("fcomb.layers.0.bias", (f0,))
    add("fcomb.layers.2.weight", (f0, f0, 1, 1)); add("fcomb.layers.2.bias", (f0,))
    add("fcomb.layers.4.weight", (m.num_classes, f0, 1, 1)); add("fcomb.layers.4.bias", (m.num_classes,))
    return out


class FlatAdamW:
    """torch.optim.AdamW(model.parameters(), lr=1e-4) of the reference trainer (main.py:103) as ONE fused HIP pass over the
    engine's flat parameter / gradient buffers (pu_adamw_step_dev).  Same update rule and defaults as torch (betas 0.9/0.999,
    eps 1e-8, weight_decay 0.01; parameters with a zero gradient still decay).  Use: opt = FlatAdamW(model, lr=1e-4);
    loss.backward(); opt.step(); opt.zero_grad().

    Semantics kept from torch: parameters whose .grad is None are skipped (no decay, no moment update; the update then runs per
    contiguous range that has gradients); step() is a no-op when no parameter has a gradient.  One difference: the step counter
    behind the bias corrections is global, not per parameter.
    f16 engine: the step is skipped ON THE DEVICE when the gradients the optimizer is about to read contain inf / NaN (what
    torch.cuda.amp.GradScaler does with a host sync); the step counter lives on the device and only advances on applied
    updates, so the bias corrections match GradScaler's behaviour."""

    def __init__(self, model: ProbabilisticUNet, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model, self.lr, self.betas, self.eps, self.weight_decay = model, lr, betas, eps, weight_decay
        self.skip_nonfinite = True        # f16 engine only: leave parameters untouched when the gradients overflowed (see step())
        self.exp_avg = None
        self.exp_avg_sq = None
        self._state = None                # device [4]: applied updates, lr / bc1, 1 / sqrt(bc2), skip marker
        self._flag = None                 # device [1]: non-finite flag of the gradients read by the current step

    @property
    def step_count(self) -> int:
        """Number of APPLIED updates (device counter; reading it synchronises)."""
        return 0 if self._state is None else int(self._state[0].item())

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.parameters():
            p.grad = None
        self.model._eg_flag = [False, False]              # nothing references the engine's gradient buffers any more

    @torch.no_grad()
    def step(self):
        m = self.model
        if m._flat is None:
            raise L.ProbUNetLibraryError("FlatAdamW.step() before the first forward: the flat buffers do not exist yet")
        m._check_views()
        dev = m._flat.device
        if self.exp_avg is None or self.exp_avg.device != dev:
            self.exp_avg = torch.zeros_like(m._flat); self.exp_avg_sq = torch.zeros_like(m._flat)
            self._state = torch.zeros(4, device=dev, dtype=torch.float32); self._flag = torch.zeros(1, device=dev, dtype=torch.float32)
        # which buffer do the gradients live in?  Views of the engine's gradient buffer (the usual case: elbo() -> backward() with no
        # accumulation, see ProbabilisticUNet._deliver) are read in place; otherwise the stable flat buffer is brought in line with what
        # p.grad says: views of it stay, foreign tensors are copied in.  Parameters whose .grad is None are left out of the update
        # altogether (torch.optim skips them: no decay, no moment update) - their range may hold stale values and is never read
        P = m._params_in(0, m._nparams)
        ebase = m._engine_grads.data_ptr()
        have = [(p, off, n) for p, off, n in P if p.grad is not None]
        if have and all(p.grad.data_ptr() == ebase + 4 * off and p.grad.dtype == torch.float32 for p, off, n in have):
            gbuf = m._engine_grads
        else:
            gbuf = m._flat_grad
        base = gbuf.data_ptr()
        runs = []                         # contiguous [lo, hi) runs of parameters that have a gradient
        for p, off, n in have:
            g = p.grad
            if g.data_ptr() != base + 4 * off or g.dtype != torch.float32:
                gbuf[off:off + n].copy_(g.reshape(-1))
            if runs and runs[-1][1] == off:
                runs[-1][1] = off + n
            else:
                runs.append([off, off + n])
        if not runs:
            return
        lib = L.lib()
        flag = None
        if self.skip_nonfinite and m.compute_dtype in ("f16", "fp16", "float16"):
            # derived from the buffer this step reads (covers accumulation over several backward() calls and, under data
            # parallelism, the averaged gradients: identical on every rank, so all ranks skip together)
            self._flag.zero_()
            for lo, hi in runs:
                L.check(lib.pu_nonfinite_flag(C.c_void_p(base + 4 * lo), hi - lo, L.ptr(self._flag), m._stream()), m._ctx, "pu_nonfinite_flag")
            flag = L.ptr(self._flag)
        hyper = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay))
        if len(runs) == 1 and runs[0] == [0, m._nparams]:
            L.check(lib.pu_adamw_step_dev(L.ptr(m._flat), L.ptr(gbuf), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), m._nparams,
                                          *hyper, L.ptr(self._state), flag, m._stream()), m._ctx, "pu_adamw_step_dev")
        else:
            L.check(lib.pu_adamw_prepare(L.ptr(self._state), flag, hyper[0], hyper[1], hyper[2], m._stream()), m._ctx, "pu_adamw_prepare")
            for lo, hi in runs:
                at = lambda t: C.c_void_p(t.data_ptr() + 4 * lo)
                L.check(lib.pu_adamw_apply(at(m._flat), at(gbuf), at(self.exp_avg), at(self.exp_avg_sq), hi - lo, *hyper,
                                           L.ptr(self._state), m._stream()), m._ctx, "pu_adamw_apply")
        m._params_dirty()
