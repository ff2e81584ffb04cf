"""XUNet + TimeConditioningPlugin on the hand-written gfx950 kernels (host orchestration only).

Replaces the a_unet module tree that /root/reference/audio_diffusion_pytorch/components.py:64-105
builds (XUNet / Block / ResnetItem / ModulationItem / AttentionItem / CrossAttentionItem /
InjectChannelsItem / SkipModulate, wrapped by TimeConditioningPlugin at :74-76).  Semantics of
each block: SURVEY.md section 8a rows a10-a17 and oracle/a_unet_restatement.py.

MI355X-first structure (not a module-per-op translation):
  * the whole U-Net is ONE autograd node: forward walks the block recursion issuing fused
    kernels and records a tape; backward replays the tape in reverse with hand-written
    gradient kernels.  No ATen arithmetic on the path, nothing returns to the host, so a
    step is capturable in a hipGraph (static shapes).
  * GroupNorm+SiLU is never materialised: it is applied inside the conv loaders (forward,
    weight-gradient) from per-(b,group) statistics.
  * nearest-upsample, channel concat (AppendChannels / InjectChannels), residual adds and the
    SkipModulate merge are loader / epilogue modes of the same MFMA conv kernel.
  * every `Linear(SiLU(features))` of the 42+9 Modulation / SkipModulate items lives in one
    contiguous weight bank: one HBM-bound GEMV-family launch produces all scale/shift vectors,
    one launch back-propagates them.
  * parameter gradients are written straight into one flat HBM buffer (contiguous, in
    parameter order) so data-parallel training all-reduces it in place over RCCL/xGMI.
"""
from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .attention import PendingLN

ITEM_RESNET = "resnet"
ITEM_MODULATION = "modulation"
ITEM_INJECT = "inject"
ITEM_ATTENTION = "attention"
ITEM_CROSS_ATTENTION = "cross_attention"

TIME_EMBED_DIM = 256
TIME_NUM_LAYERS = 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

# The recalled a_unet semantics this file and the kernels HARD-WIRE (a_unet itself is not available offline: SURVEY.md
# section 8c, Appendix A).  oracle/a_unet_restatement.py carries the same choices as named [switch] constants;
# tests/test_oracle.py::test_product_semantics_match_oracle_switches asserts the two tables agree, so flipping a switch
# in the oracle (e.g. once tools/pin_a_unet.py can compare it with a real a_unet) fails loudly and names the product
# site that has to change with it.
A_UNET_SEMANTICS = {
    "DOWNSAMPLE_WIDTH": 1,                 # _Run.block: Conv1d(kernel = stride = factor), blk.down
    "UPSAMPLE_KERNEL_SIZE": 3,             # _Run.block: nearest upsample (conv loader, up=f) + Conv1d(k=3, pad=1), blk.up
    "RESNET_KERNEL_SIZE": 3,               # _Run.resnet: conv1 / conv2
    "MODULATION_ONE_PLUS_SCALE": True,     # csrc/norm.hip chan_ln_fwd_kernel (modulation mode): xn * (1 + scale) + shift
    "TIME_EMBED_DIM": TIME_EMBED_DIM,      # UNetV0Net.__init__: NumberEmbedder(dim=256)
    "TIME_NUM_LAYERS": TIME_NUM_LAYERS,    # UNetV0Net.__init__: time_mlp
    "TIME_GELU_AFTER_EMBEDDER": True,      # _Run.conditioning: act_fwd(GELU) on time_linear's output before the MLP
    "ATTN_SEPARATE_CONTEXT_NORM": True,    # attention.attention_item: norm for q, norm_context for k / v (also self-attn)
    "SKIP_SCALES_BRANCH": True,            # _Run.block: y = skip + scale * (up conv), conv epilogue e_scale / res
    "SKIP_CAT_SCALE": 2 ** -0.5,           # _Run.block (use_modulation=False): cat[skip * 2^-1/2, x]
    "GN_EPS": ops.GN_EPS,
    "MODULATION_LN_EPS": ops.MODULATION_LN_EPS,  # ops.modulation_fwd -> adp_modulation_fwd(eps); stats carry rstd to the backward
    "ATTENTION_LN_EPS": ops.ATTENTION_LN_EPS,    # ops.ln_affine_fwd / ln_stats -> adp_ln_affine_fwd(eps) (norm, norm_context, CtxBank)
}


def item_list(items: int, use_modulation: bool, ctx_channels: int, att: int, cross: int) -> List[str]:
    """Item types of one depth, composed exactly as components.py:88-95 does."""
    return (
        [ITEM_RESNET]
        + [ITEM_MODULATION] * int(use_modulation)
        + [ITEM_INJECT] * int(ctx_channels > 0)
        + [ITEM_ATTENTION] * att
        + [ITEM_CROSS_ATTENTION] * cross
    ) * items


class _P(nn.Module):
    """Parameter holder (no forward): keeps state_dict names hierarchical."""


def _conv_params(cin: int, cout: int, k: int) -> _P:
    ref = nn.Conv1d(cin, cout, k)  # PyTorch default init == what a_unet's Conv gets
    p = _P()
    p.weight = nn.Parameter(ref.weight.detach().clone())
    p.bias = nn.Parameter(ref.bias.detach().clone())
    return p


def _linear_params(cin: int, cout: int, bias: bool = True) -> _P:
    ref = nn.Linear(cin, cout, bias=bias)
    p = _P()
    p.weight = nn.Parameter(ref.weight.detach().clone())
    if bias:
        p.bias = nn.Parameter(ref.bias.detach().clone())
    return p


def _norm_params(c: int) -> _P:
    p = _P()
    p.weight = nn.Parameter(torch.ones(c))
    p.bias = nn.Parameter(torch.zeros(c))
    return p


class UNetV0Net(nn.Module):
    """`UNetV0(...)` instance: forward(x [B,C,L], time [B], *, features=None, embedding=None, channels=None)."""

    def __init__(self, dim: int, in_channels: int, channels: Sequence[int], factors: Sequence[int],
                 items: Sequence[int], attentions: Sequence[int], cross_attentions: Sequence[int],
                 context_channels: Sequence[int], attention_features: Optional[int], attention_heads: Optional[int],
                 embedding_features: Optional[int], resnet_groups: int, modulation_features: int,
                 out_channels: Optional[int], use_modulation: bool = True, use_time_conditioning: bool = True,
                 blocks=None):
        super().__init__()
        if blocks is not None:  # the package also exports this class under a_unet's name XUNet (__init__.py:1 there)
            raise NotImplementedError("a_unet's XUNet(in_channels, blocks=[XBlock(...), ...]) block-list constructor is "
                                      "not rebuilt here (a_unet API, outside the hot-path scope); build the net through "
                                      "UNetV0(...), which composes the same blocks (components.py:79-105)")
        assert dim == 1, "audio U-Net is 1-D"
        assert use_modulation or not use_time_conditioning, "use_time_conditioning requires use_modulation=True"
        n = len(channels)
        self.use_modulation, self.use_time = bool(use_modulation), bool(use_time_conditioning)
        self.in_channels = in_channels
        self.out_channels = out_channels if out_channels is not None else in_channels
        self.channels, self.factors = list(channels), list(factors)
        self.context_channels = list(context_channels)
        self.groups = resnet_groups
        self.mf = modulation_features
        self.heads, self.head_features = attention_heads, attention_features
        self.embedding_features = embedding_features
        for c in channels:
            assert c % resnet_groups == 0, "channels must be divisible by resnet_groups"
        for f in factors:
            assert int(f) == f and f >= 1, "down/upsample factors are positive integers"
        # factors 1 / 2 / 4 run on the strided-conv and upsample-loader kernel variants; any other integer factor
        # goes through adp_unshuffle + a 1x1 conv (down) and the generic upsample loader + adp_pool_sum (up)

        # ---- TimeConditioningPlugin: NumberEmbedder(features=MF, dim=256) + 2 x (Linear + GELU)
        if self.use_time:
            self.time_weights = nn.Parameter(torch.randn(TIME_EMBED_DIM // 2))
            self.time_linear = _linear_params(TIME_EMBED_DIM + 1, self.mf)
            self.time_mlp = nn.ModuleList([_linear_params(self.mf, self.mf) for _ in range(TIME_NUM_LAYERS)])

        # ---- blocks; every Modulation / SkipModulate Linear goes into the bank
        bank_w, bank_b = [], []
        self.bank_slices: Dict[tuple, tuple] = {}
        off = 0

        def bank_add(key, nout):
            nonlocal off
            ref = nn.Linear(self.mf, nout)
            bank_w.append(ref.weight.detach())
            bank_b.append(ref.bias.detach())
            self.bank_slices[key] = (off, nout)
            off += nout

        self.item_types: List[List[str]] = []
        blocks = []
        for d in range(n):
            in_ch = in_channels if d == 0 else channels[d - 1]
            out_ch = self.out_channels if d == 0 else in_ch
            C, f = channels[d], factors[d]
            its = item_list(items[d], self.use_modulation, context_channels[d], attentions[d], cross_attentions[d])
            self.item_types.append(its)
            blk = _P()
            blk.down = _conv_params(in_ch, C, f)
            blk.items_down = nn.ModuleList([self._make_item(t, d, C, ("down", i), bank_add)
                                            for i, t in enumerate(its)])
            blk.items_up = nn.ModuleList([self._make_item(t, d, C, ("up", i), bank_add) for i, t in enumerate(its)])
            blk.up = _conv_params(C, out_ch, 3)
            if in_ch != out_ch:
                blk.skip_adapter = _conv_params(in_ch, out_ch, 1)
            if self.use_modulation:
                bank_add((d, "skip"), out_ch)                        # SkipModulate (components.py:99)
            else:
                blk.skip_cat = _conv_params(2 * out_ch, out_ch, 1)   # SkipCat: Conv1x1(cat[skip * 2^-1/2, x])
            blk.in_ch, blk.out_ch = in_ch, out_ch
            blocks.append(blk)
        self.blocks = nn.ModuleList(blocks)
        self.bank_total = off
        # rows of the bank owned by each depth (contiguous: the bank is filled depth by depth)
        self.bank_depth_rows = []
        for d in range(n):
            offs = [(o, o + c) for k, (o, c) in self.bank_slices.items() if k[0] == d]
            self.bank_depth_rows.append((min(a for a, _ in offs), max(b for _, b in offs)) if offs else (0, 0))
        if off > 0:
            self.bank_weight = nn.Parameter(torch.cat(bank_w, 0).contiguous())
            self.bank_bias = nn.Parameter(torch.cat(bank_b, 0).contiguous())

    def _make_item(self, t: str, d: int, C: int, key, bank_add) -> nn.Module:
        p = _P()
        if t == ITEM_RESNET:
            p.gn1, p.conv1 = _norm_params(C), _conv_params(C, C, 3)
            p.gn2, p.conv2 = _norm_params(C), _conv_params(C, C, 3)
        elif t == ITEM_MODULATION:
            bank_add((d,) + key, 2 * C)
        elif t == ITEM_INJECT:
            p.conv = _conv_params(C + self.context_channels[d], C, 1)
        elif t in (ITEM_ATTENTION, ITEM_CROSS_ATTENTION):
            assert self.heads and self.head_features, "attention_heads and attention_features are required"
            cf = C
            if t == ITEM_CROSS_ATTENTION:
                assert self.embedding_features, "embedding_features is required for cross attention"
                cf = self.embedding_features
            mid = self.heads * self.head_features
            p.norm, p.norm_context = _norm_params(C), _norm_params(cf)
            p.to_q = _linear_params(C, mid, bias=False)
            p.to_kv = _linear_params(cf, 2 * mid, bias=False)
            p.to_out = _linear_params(mid, C, bias=False)
        else:
            raise ValueError(t)
        return p

    # ------------------------------------------------------------------ state-dict interchange with the oracle
    def load_oracle_state_dict(self, sd: Dict[str, Tensor]) -> None:
        """Loads a state dict in oracle/a_unet_restatement.py naming (per-module Modulation / skip Linears)."""
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                if k == "time_weights" or k.startswith("time_"):
                    own[k].copy_(v)
                    continue
                parts = k.split(".")
                d = int(parts[1])
                if parts[2] in ("items_down", "items_up") and parts[4] == "to_scale_shift":
                    off, nout = self.bank_slices[(d, parts[2][6:], int(parts[3]))]
                    (self.bank_weight if parts[5] == "weight" else self.bank_bias)[off:off + nout].copy_(v)
                elif parts[2] == "skip" and parts[3] == "conv":      # SkipCat's 1x1 conv
                    own[f"blocks.{d}.skip_cat.{parts[4]}"].copy_(v)
                elif parts[2] == "skip":
                    off, nout = self.bank_slices[(d, "skip")]
                    (self.bank_weight if parts[4] == "weight" else self.bank_bias)[off:off + nout].copy_(v)
                else:
                    own[k].copy_(v)

    # ------------------------------------------------------------------ checkpoints of the reference (a_unet naming)
    def a_unet_key_order(self) -> List[str]:
        """Parameter names (in oracle/a_unet_restatement.py naming) in the order a_unet's module tree REGISTERS them, as
        recalled (/root/reference/audio_diffusion_pytorch/components.py:157, :178 show the
        `Module([...], fn)` -> ModuleList `blocks` convention): TimeConditioningPlugin = Module([embedder, mlp, net]) and
        every Block = Module([skip_adapter, Sequential(downsample, *items, inner_block, *items_up, upsample), skip]) -- i.e.
        depth d + 1 sits NESTED between depth d's down items and up items, where this package (and the restatement) keep one
        flat list per depth.  UNVERIFIED offline like the rest of the a_unet half;
        tools/pin_a_unet.py checks it against the real package where that exists."""
        names: List[str] = []
        if self.use_time:
            names += ["time_weights", "time_linear.weight", "time_linear.bias"]
            for i in range(TIME_NUM_LAYERS):
                names += [f"time_mlp.{i}.weight", f"time_mlp.{i}.bias"]
        item_keys = {
            ITEM_RESNET: ["gn1.weight", "gn1.bias", "conv1.weight", "conv1.bias", "gn2.weight", "gn2.bias", "conv2.weight",
                          "conv2.bias"],
            ITEM_MODULATION: ["to_scale_shift.weight", "to_scale_shift.bias"],
            ITEM_INJECT: ["conv.weight", "conv.bias"],
            ITEM_ATTENTION: ["norm.weight", "norm.bias", "norm_context.weight", "norm_context.bias", "to_q.weight",
                             "to_kv.weight", "to_out.weight"],
        }
        item_keys[ITEM_CROSS_ATTENTION] = item_keys[ITEM_ATTENTION]

        def block(d: int):
            if d == len(self.blocks):
                return
            blk = self.blocks[d]
            if hasattr(blk, "skip_adapter"):
                names.extend([f"blocks.{d}.skip_adapter.weight", f"blocks.{d}.skip_adapter.bias"])
            names.extend([f"blocks.{d}.down.weight", f"blocks.{d}.down.bias"])
            for i, t in enumerate(self.item_types[d]):
                names.extend(f"blocks.{d}.items_down.{i}.{k}" for k in item_keys[t])
            block(d + 1)
            for i, t in enumerate(self.item_types[d]):
                names.extend(f"blocks.{d}.items_up.{i}.{k}" for k in item_keys[t])
            names.extend([f"blocks.{d}.up.weight", f"blocks.{d}.up.bias"])
            if self.use_modulation:
                names.extend([f"blocks.{d}.skip.to_scale.weight", f"blocks.{d}.skip.to_scale.bias"])
            else:
                names.extend([f"blocks.{d}.skip.conv.weight", f"blocks.{d}.skip.conv.bias"])
        block(0)
        return names

    def load_a_unet_state_dict(self, sd: Dict[str, Tensor]) -> Dict[str, str]:
        """EXPERIMENTAL (stays so until tools/pin_a_unet.py has run against the real a_unet).  Loads a checkpoint of the
        REFERENCE's UNetV0 (a_unet's own key names, e.g. `blocks.2.blocks.1.blocks.3...`): the floating-point tensors of `sd`,
        in its (registration) order, are assigned to `a_unet_key_order()` position by position.  Guards against a silently
        wrong order: every shape is checked, and the checkpoint key's LAST component (`weight` / `bias` / `weights`) has to
        be the target's -- a GroupNorm's (weight, bias) pair or a conv's swapped, or a bias-free Linear where a biased layer
        is expected, is refused (equal-shaped layers in another order -- conv1 / conv2 of a ResnetItem, depths with equal
        channel counts -- cannot be told apart without a_unet; that is what the pin recipe is for).  The nesting names
        themselves are never interpreted.  Returns {checkpoint key: name it was loaded as}.  Checkpoints of the plugin
        wrappers (use_embedding_cfg / use_text_conditioning / AppendChannels): components.load_reference_state_dict."""
        theirs = [(k, v) for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point]
        order = self.a_unet_key_order()
        if len(theirs) != len(order):
            raise ValueError(f"checkpoint holds {len(theirs)} floating-point tensors, this UNetV0 has {len(order)} parameters "
                             f"(Modulation / SkipModulate layers counted per item)")
        shapes = self._oracle_shapes()
        for (k, v), name in zip(theirs, order):
            # (exact shape, or the same shape up to unit axes: a Linear [M, R] stored where this net keeps a 1x1 conv [M, R, 1])
            if tuple(d for d in v.shape if d != 1) != tuple(d for d in shapes[name] if d != 1):
                raise ValueError(f"checkpoint entry {k!r} has shape {tuple(v.shape)}, expected {shapes[name]} for {name!r}: the "
                                 f"checkpoint was not produced by UNetV0 with this configuration (or a_unet registers its "
                                 f"parameters in another order than recalled: see a_unet_key_order)")
            last, want = k.rsplit(".", 1)[-1], name.rsplit(".", 1)[-1].replace("time_weights", "weights")
            if last in ("weight", "bias", "weights") and last != want:
                raise ValueError(f"checkpoint entry {k!r} is a `{last}` where {name!r} (a `{want}`) is expected at this position: "
                                 f"a_unet registers its parameters in another order than recalled (see a_unet_key_order)")
        self.load_oracle_state_dict({name: v.reshape(shapes[name]) for (k, v), name in zip(theirs, order)})
        return {k: name for (k, _), name in zip(theirs, order)}

    def _oracle_shapes(self) -> Dict[str, tuple]:
        """Shape of every parameter under its oracle-naming key (the conditioning bank split per item)."""
        out = {}
        for k, p in self.named_parameters():
            if k not in ("bank_weight", "bank_bias"):
                out[k.replace(".skip_cat.", ".skip.conv.")] = tuple(p.shape)
        for key, (off, nout) in self.bank_slices.items():
            base = f"blocks.{key[0]}.skip.to_scale" if key[1] == "skip" else f"blocks.{key[0]}.items_{key[1]}.{key[2]}.to_scale_shift"
            out[base + ".weight"] = (nout, self.mf)
            out[base + ".bias"] = (nout,)
        return out

    def oracle_named_grads(self, grads: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Maps {own parameter name: tensor} to oracle naming (splitting the bank)."""
        out = {}
        for k, v in grads.items():
            if k in ("bank_weight", "bank_bias"):
                continue
            out[k.replace(".skip_cat.", ".skip.conv.")] = v
        for key, (off, nout) in self.bank_slices.items():
            if key[1] == "skip":
                base = f"blocks.{key[0]}.skip.to_scale"
            else:
                base = f"blocks.{key[0]}.items_{key[1]}.{key[2]}.to_scale_shift"
            out[base + ".weight"] = grads["bank_weight"][off:off + nout]
            out[base + ".bias"] = grads["bank_bias"][off:off + nout]
        return out

    # ------------------------------------------------------------------ flat gradient layout
    def _is_ctx_side(self, name: str) -> bool:
        """Context-side parameters (norm_context, to_kv) of a CROSS-attention item: their gradients come out of the context
        bank's backward (attention.CtxBank), one launch for all items after the last of them has run -- not when their block
        is done.  They live in ONE trailing region of the flat gradient buffer so that every block's region stays contiguous
        and the data-parallel hook gets the context side as a single bucket (`ctx_param_range`)."""
        parts = name.split(".")
        if len(parts) < 6 or parts[0] != "blocks" or parts[4] not in ("norm_context", "to_kv"):
            return False
        return self.item_types[int(parts[1])][int(parts[3])] == ITEM_CROSS_ATTENTION

    def _named_params(self):
        """[(name, parameter)] in registration order, walked ONCE: the module tree holds ~600 parameters and a traversal costs
        1-2 ms of host time -- four of them per step were a tenth of an eager step (the data-parallel path at N > 1 is eager).
        The cache is VALIDATED on every call (one dict lookup + identity test per parameter, ~30 us): `.to()` swaps storages and
        keeps the Parameter objects, but load_state_dict(assign=True), to_empty(), torch.__future__'s overwrite-on-conversion
        and copy.deepcopy put NEW objects into the modules' `_parameters` dicts -- the kernels read those through the module
        attributes, so gradients keyed on the old objects would land on orphans."""
        cached = getattr(self, "_named_cache", None)
        holders = getattr(self, "_named_holders", None)
        # (a copy.deepcopy of the net -- an EMA copy -- carries a structurally valid copy of the cache, but the id() keys of
        #  _pname_cache name the ORIGINAL's objects)
        if cached is not None and holders is not None and len(cached) == len(holders) \
                and getattr(self, "_named_cache_of", None) == id(self):
            for (_, p), (params, leaf) in zip(cached, holders):
                if params.get(leaf) is not p:
                    cached = None
                    break
        else:
            cached = None
        if cached is None:
            cached, holders = [], []
            for prefix, mod in self.named_modules():
                for leaf, p in mod._parameters.items():
                    if p is not None:
                        cached.append((prefix + ("." if prefix else "") + leaf, p))
                        holders.append((mod._parameters, leaf))
            assert [n for n, _ in cached] == [n for n, _ in self.named_parameters()], \
                "UNetV0Net: shared / re-registered parameters are not supported by the flat gradient layout"
            self._named_cache, self._named_holders = cached, holders
            self._pname_cache = {id(p): n for n, p in cached}
            self._named_cache_of = id(self)
            self._ctx_tables = None  # (pointer tables of the cross-attention context bank: rebuilt from the new objects)
        return cached

    def _param_offsets(self):
        """{parameter name: (start, end)} in the flat gradient buffer: parameter order, except that the context side of the
        cross-attention items is moved behind everything else (per item still [norm_context.weight | norm_context.bias |
        to_kv.weight], so the (gamma, beta) pair stays adjacent)."""
        if getattr(self, "_offsets", None) is None:
            off, table, late = 0, {}, []
            for name, p in self._named_params():
                if self._is_ctx_side(name):
                    late.append((name, p.numel()))
                    continue
                table[name] = (off, off + p.numel())
                off += p.numel()
            self._ctx_range = (off, off)
            for name, n in late:
                table[name] = (off, off + n)
                off += n
            self._ctx_range = (self._ctx_range[0], off)
            self._offsets = table
        return self._offsets

    def ctx_param_range(self):
        """(start, end) of the cross-attention context-side gradients in the flat buffer (empty without cross attention)."""
        self._param_offsets()
        return self._ctx_range

    def ctx_final_depth(self) -> int:
        """The shallowest depth holding a CrossAttentionItem: once that block's backward is done no item adds to the context
        side any more (-1 without cross attention)."""
        for d, its in enumerate(self.item_types):
            if ITEM_CROSS_ATTENTION in its:
                return d
        return -1

    def block_param_range(self, d: int):
        cache = self.__dict__.setdefault("_block_ranges", {})
        if d not in cache:
            t = self._param_offsets()
            spans = [v for k, v in t.items() if k.startswith(f"blocks.{d}.") and not self._is_ctx_side(k)]
            cache[d] = (min(a for a, _ in spans), max(b for _, b in spans))
        return cache[d]

    def nonblock_param_ranges(self):
        """Maximal contiguous ranges of the parameters outside the blocks and the conditioning bank's weight (time
        MLP + bank bias); their gradients are final only after conditioning_backward.  (The bank weight is handed
        out per depth.)"""
        if getattr(self, "_nonblock_ranges", None) is None:
            t = self._param_offsets()
            spans = sorted(v for k, v in t.items() if not k.startswith("blocks.") and k != "bank_weight")
            out = []
            for a, b in spans:
                if out and out[-1][1] == a:
                    out[-1] = (out[-1][0], b)
                else:
                    out.append((a, b))
            self._nonblock_ranges = out
        return self._nonblock_ranges

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, time: Optional[Tensor] = None, *, features: Optional[Tensor] = None,
                embedding: Optional[Tensor] = None, channels: Optional[Sequence[Optional[Tensor]]] = None,
                x_append: Optional[Tensor] = None, conditioning: Optional[Tensor] = None) -> Tensor:
        """`conditioning` (inference only): the rows `conditioning_table` returned for this call's `time` values -- the time
        MLP and the conditioning bank's projection are then skipped (VSampler hoists them out of its loop)."""
        if conditioning is not None:
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for _, p in self._named_params())):
                raise RuntimeError("UNetV0.forward(conditioning=...) is an inference-only path (torch.no_grad())")
            assert features is None and conditioning.shape == (x.shape[0], self.bank_total), \
                "conditioning: [B, bank_total] rows of conditioning_table, without `features`"
        if self.use_time:
            assert time is not None, "TimeConditioningPlugin requires time in forward"
        elif time is not None:  # the reference's bare XUNet.forward takes x only (keyword conditioning)
            raise TypeError("UNetV0(use_time_conditioning=False).forward() takes 1 positional argument (x); "
                            "`time` was given")
        if self.bank_total > 0 and not self.use_time:
            assert features is not None, "ModulationItem requires `features` when use_time_conditioning=False"
        params = [p for _, p in self._named_params()]
        ctx_list = [c for c in (channels or []) if c is not None]
        # (inside an autograd Function's forward grad mode is always off and needs_input_grad only mirrors requires_grad: whether
        #  anybody can ask for a backward pass has to be read HERE -- under torch.no_grad() nothing is recorded)
        grad_on = torch.is_grad_enabled()
        if not x.is_cuda:  # CPU tensors only reach the kernels through the test-suite's SIMT emulator build
            return _UNetFn.apply(self, x, time, features, embedding, x_append, channels, len(ctx_list), conditioning,
                                 grad_on, *ctx_list, *params)
        for t in (time, features, embedding, x_append, conditioning, *ctx_list, params[0]):
            if t is not None and t.device != x.device:
                raise RuntimeError(f"UNetV0: every tensor must live on the input's device {x.device}; got {t.device}")
        with torch.cuda.device(x.device):  # launches go to the current device's stream
            return _UNetFn.apply(self, x, time, features, embedding, x_append, channels, len(ctx_list), conditioning,
                                 grad_on, *ctx_list, *params)

    @torch.no_grad()
    def conditioning_table(self, times: Tensor) -> Optional[Tensor]:
        """[T, bank_total]: what every Modulation / SkipModulate item reads (SiLU(time MLP) through the conditioning bank) for
        each of the T time values, in one batched pass -- the bank's weights (the largest tensor of the net: 184 MB in the
        README configuration) are streamed once per 16 rows instead of once per U-Net call.  None when this net has no time
        conditioning (nothing to hoist).  Row b of a later call: forward(x, time, conditioning=table[rows of that call])."""
        if not (self.use_time and self.bank_total > 0):
            return None
        run = _Run(self, False)
        return run.conditioning(times.reshape(-1), None)


NATIVE_FACTORS = (1, 2, 4)      # down/upsample factors with dedicated conv kernel variants
SKIP_CAT_SCALE = 2 ** -0.5      # a_unet SkipCat / MergeCat: cat[skip * scale, x]
POISON_GRADS = os.environ.get("ADP_DEBUG_POISON", "0") == "1"
# GroupNorm statistics from the producing conv's epilogue (1) or from a pass over the tensor (0); A/B switch
GN_EPILOGUE = int(os.environ.get("ADP_GN_EPILOGUE", "1"))


def _grad_buffer(shape, device) -> Tensor:
    """Gradient destination the kernels OVERWRITE slice by slice: left uninitialised (no 0.7 GB memset per backward).
    ADP_DEBUG_POISON=1 (set by the test-suite) fills it with NaN so that a slice nobody wrote cannot pass a test."""
    buf = torch.empty(shape, dtype=torch.float32, device=device)
    if POISON_GRADS:
        buf.fill_(float("nan"))
    return buf


# ModulationItem + the LayerNorm of the attention item behind it as one launch (adp_modulation_ln_fwd); "0": two launches (A/B)
MOD_LN_FUSE = os.environ.get("ADP_MOD_LN_FUSE", "1") != "0"

# ... and the backward of that pair as one pass (adp_modulation_ln_bwd_partial; cross-attention items); "0": A/B
MOD_LN_BWD_FUSE = os.environ.get("ADP_MOD_LN_BWD_FUSE", "1") != "0"

# channel count from which SiLU(GroupNorm(x)) is materialised instead of recomputed in the conv loaders
# (round 3, with the Winograd variants: 128 -> 14.14, 256 -> 14.15, 512 -> 14.20, 64 -> 14.21, 1024 -> 14.37 ms per step;
#  round 6: 64, together with the F(4,3) block from 64 input channels -- ADP_WINO4_MIN_R, conv_mm4.hip -- 11.11 -> 11.05 ms, 64
#  alone 11.17)
ACT_MATERIALIZE_MIN_C = int(os.environ.get("ADP_ACT_MATERIALIZE_MIN_C", "64"))


class _Run:
    """One forward execution: issues kernels, and (when grad is needed) records the tape."""

    def __init__(self, net: UNetV0Net, need_grad: bool):
        self.net = net
        self.need_grad = need_grad
        self.tape: List = []           # list of callables g -> g_prev
        self.grads: Dict[str, Tensor] = {}
        net._named_params()
        self.pnames = net._pname_cache
        self.gn: Optional[ops.GnPart] = None  # GroupNorm partial statistics of the tensor produced last (if any)
        self.ln_ready = None  # (y, xn, cn, stats): LayerNorm(s) of y already formed by the ModulationItem that produced y
        self.mod_sums = ops.ModulationSums()  # parked second stages of the Modulation backwards
        # parked second stages of the split ConvBlock weight gradients: summed per side of a block in one launch per shape
        # (ADP_WGRAD_PARK=0: every weight gradient finishes itself, A/B)
        self.wpark = ops.WgradPark() if os.environ.get("ADP_WGRAD_PARK", "1") != "0" else None

    # -- gradient destination views -------------------------------------------------------
    def g(self, p: nn.Parameter) -> Tensor:
        try:
            return self.grads[self.pnames[id(p)]]
        except KeyError:
            raise RuntimeError("UNetV0: a parameter the kernels read is not one of the net's registered Parameters (replaced "
                               "between forward and backward, or wrapped by nn.utils.parametrize -- not supported by the "
                               "one-node U-Net: its gradient would have no slot in the flat gradient buffer)") from None

    def gspan(self, p_first: nn.Parameter, numel: int) -> Tensor:
        """Flat gradient slice starting at `p_first` and covering `numel` floats (adjacent parameters)."""
        a, _ = self.net._param_offsets()[self.pnames[id(p_first)]]
        return self.flat[a:a + numel]

    # -- conditioning ---------------------------------------------------------------------
    def conditioning(self, time: Tensor, features: Optional[Tensor], ss_pre: Optional[Tensor] = None):
        n = self.net
        self.ss_all = self.dss_all = None
        if ss_pre is not None:  # rows of UNetV0Net.conditioning_table (inference)
            assert not self.need_grad
            self.feats = None
            self.ss_all = ss_pre.contiguous()
            return self.ss_all
        if n.use_time:
            t = time.reshape(-1).to(torch.float32).contiguous()
            four = ops.time_fourier_fwd(t, n.time_weights)
            pre = [ops.linear_fwd(four, n.time_linear.weight, n.time_linear.bias)]
            acts = [ops.act_fwd(pre[0], ACT_GELU)]
            for lin in n.time_mlp:
                pre.append(ops.linear_fwd(acts[-1], lin.weight, lin.bias))
                acts.append(ops.act_fwd(pre[-1], ACT_GELU))
            feats = acts[-1] if features is None else ops.add(features.contiguous(), acts[-1])
            self.t, self.four, self.pre, self.acts = t, four, pre, acts
        else:
            feats = features.contiguous() if features is not None else None
        self.feats = feats
        if n.bank_total > 0:
            self.ss_all = ops.linear_fwd(feats, n.bank_weight, n.bank_bias, act=ACT_SILU)
            if self.need_grad:  # every Modulation / SkipModulate backward OVERWRITES its slice: no zero fill
                self.dss_all = _grad_buffer(self.ss_all.shape, self.ss_all.device)
        return self.ss_all

    def conditioning_backward(self) -> Tensor:
        """Back-propagates dss_all through the bank and the time MLP; returns d(features)."""
        n = self.net
        if n.bank_total == 0:
            return None
        assert not self.mod_sums.items, "dss_all is read while second stages of Modulation backwards are still parked"
        dfa = ops.linear_bwd_data(self.dss_all, n.bank_weight)  # the bank's own gradient: bank_grad_for_depth
        dfeat = ops.act_bwd(self.feats, dfa, ACT_SILU)
        if not n.use_time:
            return dfeat
        dcur = dfeat
        lins = [n.time_linear] + list(n.time_mlp)
        for i in range(len(lins) - 1, -1, -1):
            dp = ops.act_bwd(self.pre[i], dcur, ACT_GELU)
            xin = self.four if i == 0 else self.acts[i - 1]
            ops.linear_bwd_weight(dp, xin, act=ACT_NONE, dw=self.g(lins[i].weight), dbias=self.g(lins[i].bias))
            dcur = ops.linear_bwd_data(dp, lins[i].weight)
        ops.time_fourier_bwd(self.t, n.time_weights, dcur, dw=self.g(n.time_weights))
        return dfeat

    def bank_grad_all(self) -> None:
        """The whole conditioning bank's weight / bias gradient in ONE launch (no data-parallel hook: nobody reads a depth's
        rows before the end of the backward pass; nine launches of 12-27 us become one streaming write of the bank)."""
        n = self.net
        if n.bank_total > 0:
            assert not self.mod_sums.items, "parked Modulation sums while the bank's gradient is formed"
            ops.linear_bwd_weight(self.dss_all.view(-1), self.feats, act=ACT_SILU, dw=self.g(n.bank_weight),
                                  dbias=self.g(n.bank_bias), rows=n.bank_total, dy_bstride=n.bank_total)

    def bank_grad_for_depth(self, d: int, defer: bool = False):
        """Weight / bias gradient of depth d's rows of the conditioning bank.  Every Modulation / SkipModulate of
        block d has run its backward when block d's tape entries are done, so these rows are final long before the
        end of the backward pass -- their all-reduce overlaps the shallower blocks instead of trailing the step.
        `defer`: only the Modulation sums of these rows are flushed; bank_grad_all forms the gradient at the end.
        Returns the (start, end) row range."""
        n = self.net
        a, b = n.bank_depth_rows[d] if n.bank_total > 0 else (0, 0)
        self.mod_sums.flush(a, b)
        assert not any(a <= it[0] < b for it in self.mod_sums.items), "parked Modulation sums inside the rows being read"
        if b > a and not defer:
            K = n.mf
            ops.linear_bwd_weight(self.dss_all.view(-1)[a:], self.feats, act=ACT_SILU,
                                  dw=self.g(n.bank_weight)[a:b], dbias=self.g(n.bank_bias)[a:b], rows=b - a,
                                  dy_bstride=n.bank_total)
        return a, b

    def ss(self, key):
        off, nout = self.net.bank_slices[key]
        return self.ss_all.view(-1)[off:], (self.dss_all.view(-1)[off:] if self.need_grad else None)

    # -- GroupNorm statistics: from the producer's epilogue partials when the producing kernel wrote them --------
    def gn_part_for(self, C: int) -> Optional[ops.GnPart]:
        """A GnPart for the kernel about to produce a C-channel tensor that a GroupNorm reads next -- or None when the
        statistics have to come from a pass over the tensor: the producers write one partial per 4-channel row quad,
        which only nests into the groups when a group holds a multiple of 4 channels."""
        if GN_EPILOGUE and (C // self.net.groups) % 4 == 0:
            return ops.GnPart()
        return None

    def gn_stats_of(self, x: Tensor) -> Tensor:
        """stats [B, G, 2] of x: one tiny launch over the partials its producer left (conv / Modulation epilogue),
        else the two-launch statistics pass over the tensor."""
        G = self.net.groups
        if self.gn is not None and self.gn.covers(x):
            return ops.gn_finalize(self.gn.part, G)
        return ops.gn_stats(x, G)

    def gn_stats_act_of(self, x: Tensor, gnp):
        """(stats, SiLU(GroupNorm(x)) materialised) for the wide layers."""
        G = self.net.groups
        if self.gn is not None and self.gn.covers(x):
            part = self.gn.part
            if (x.shape[1] // G // 4) * part.shape[2] <= 1024:  # few entries per group: one launch does both
                return ops.gn_finalize_act(x, part, G, gnp.weight, gnp.bias)
            st = ops.gn_finalize(part, G)
            return st, ops.gn_act(x, st, G, gnp.weight, gnp.bias)
        return ops.gn_stats_act(x, G, gnp.weight, gnp.bias)

    # -- items ----------------------------------------------------------------------------
    def resnet(self, p, x: Tensor) -> Tensor:
        G = self.net.groups
        if x.shape[1] >= ACT_MATERIALIZE_MIN_C and x.shape[0] * x.shape[1] <= 65535:  # (grid limit of gn_apply)
            return self.resnet_wide(p, x)
        st1 = self.gn_stats_of(x)
        self.gn = self.gn_part_for(x.shape[1])
        h1 = ops.conv1d(x, p.conv1.weight, p.conv1.bias, pad=1, prologue=1, pro_stats=st1, pro_gamma=p.gn1.weight,
                        pro_beta=p.gn1.bias, groups=G, gn=self.gn)
        st2 = self.gn_stats_of(h1)
        self.gn = None
        y = ops.conv1d(h1, p.conv2.weight, p.conv2.bias, pad=1, prologue=1, pro_stats=st2, pro_gamma=p.gn2.weight,
                       pro_beta=p.gn2.bias, groups=G, res=x)
        if self.need_grad:
            def bwd(gy):
                ops.conv1d_wgrad(h1, gy, 3, pad=1, park=self.wpark, prologue=1, pro_stats=st2, pro_gamma=p.gn2.weight,
                                 pro_beta=p.gn2.bias, groups=G, dw=self.g(p.conv2.weight), dbias=self.g(p.conv2.bias))
                gb2 = ops.GnBwdPart(h1, st2, p.gn2.weight, p.gn2.bias, G)
                dact2 = ops.conv1d(gy, p.conv2.weight, None, pad=1, transposed=True, gnb=gb2)
                dh1, _, _ = ops.gn_silu_bwd(h1, dact2, st2, p.gn2.weight, p.gn2.bias, G, dgamma=self.g(p.gn2.weight),
                                            dbeta=self.g(p.gn2.bias), ab=gb2.ab)
                ops.conv1d_wgrad(x, dh1, 3, pad=1, park=self.wpark, prologue=1, pro_stats=st1, pro_gamma=p.gn1.weight,
                                 pro_beta=p.gn1.bias, groups=G, dw=self.g(p.conv1.weight), dbias=self.g(p.conv1.bias))
                gb1 = ops.GnBwdPart(x, st1, p.gn1.weight, p.gn1.bias, G)
                dact1 = ops.conv1d(dh1, p.conv1.weight, None, pad=1, transposed=True, gnb=gb1)
                dx, _, _ = ops.gn_silu_bwd(x, dact1, st1, p.gn1.weight, p.gn1.bias, G, dres=gy,
                                           dgamma=self.g(p.gn1.weight), dbeta=self.g(p.gn1.bias), ab=gb1.ab)
                return dx
            self.tape.append((bwd, None))
        return y

    def resnet_wide(self, p, x: Tensor) -> Tensor:
        """ResnetBlock of the wide layers (C >= ACT_MATERIALIZE_MIN_C): same arithmetic, but SiLU(GroupNorm(.)) is materialised once by
        the statistics' second stage instead of being recomputed by each of the 8-16 conv / weight-gradient
        workgroups that stage a tile of it (the tensors are 2-8 MB here; see gn_apply_kernel in csrc/norm.hip)."""
        G = self.net.groups
        st1, a1 = self.gn_stats_act_of(x, p.gn1)
        self.gn = self.gn_part_for(x.shape[1])
        h1 = ops.conv1d(a1, p.conv1.weight, p.conv1.bias, pad=1, gn=self.gn)
        st2, a2 = self.gn_stats_act_of(h1, p.gn2)
        self.gn = None
        y = ops.conv1d(a2, p.conv2.weight, p.conv2.bias, pad=1, res=x)
        if self.need_grad:
            def bwd(gy):
                ops.conv1d_wgrad(a2, gy, 3, pad=1, dw=self.g(p.conv2.weight), dbias=self.g(p.conv2.bias), park=self.wpark)
                # (the data-gradient convs leave the first stage of the GroupNorm backward behind them: ops.GnBwdPart)
                gb2 = ops.GnBwdPart(h1, st2, p.gn2.weight, p.gn2.bias, G)
                dact2 = ops.conv1d(gy, p.conv2.weight, None, pad=1, transposed=True, gnb=gb2)
                dh1, _, _ = ops.gn_silu_bwd(h1, dact2, st2, p.gn2.weight, p.gn2.bias, G, dgamma=self.g(p.gn2.weight),
                                            dbeta=self.g(p.gn2.bias), ab=gb2.ab)
                ops.conv1d_wgrad(a1, dh1, 3, pad=1, dw=self.g(p.conv1.weight), dbias=self.g(p.conv1.bias), park=self.wpark)
                gb1 = ops.GnBwdPart(x, st1, p.gn1.weight, p.gn1.bias, G)
                dact1 = ops.conv1d(dh1, p.conv1.weight, None, pad=1, transposed=True, gnb=gb1)
                dx, _, _ = ops.gn_silu_bwd(x, dact1, st1, p.gn1.weight, p.gn1.bias, G, dres=gy,
                                           dgamma=self.g(p.gn1.weight), dbeta=self.g(p.gn1.bias), ab=gb1.ab)
                return dx
            self.tape.append((bwd, None))
        return y

    def modulation(self, key, x: Tensor, attn_next=None) -> Tensor:
        """`attn_next`: the parameters of an AttentionItem / CrossAttentionItem that follows immediately (and whether it is a
        self-attention item): its LayerNorm(s) of this item's output come out of the same launch (adp_modulation_ln_fwd) and
        wait in self.ln_ready for attention.attention_item."""
        ss, dss = self.ss(key)
        NT = self.net.bank_total
        # (Modulation-side GroupNorm partials were measured and rejected: the per-channel lane reductions cost the
        # kernel more than the statistics launch they save -- tools/rejected/README.md)
        self.gn = None
        if attn_next is not None and MOD_LN_FUSE:
            p, dual = attn_next
            y, stats, xn, cn, st = ops.modulation_ln_fwd(x, ss, NT, p.norm.weight, p.norm.bias,
                                                         p.norm_context.weight if dual else None,
                                                         p.norm_context.bias if dual else None)
            self.ln_ready = (y, xn, cn, st)
        else:
            y, stats = ops.modulation_fwd(x, ss, NT)
        if self.need_grad:
            off = self.net.bank_slices[key][0]

            # (second stage parked: the depth's Modulation items are summed together when its bank rows are formed)
            def bwd(gy):
                if isinstance(gy, PendingLN):  # the attention item behind this one left its LayerNorm's backward to us
                    return self.mod_sums.partial_ln(off, x, ss, NT, stats, dss, NT, gy.y, gy.dxn, gy.gamma, gy.ln_stats,
                                                    gy.dres, gy.dgb)
                return self.mod_sums.partial(off, x, gy, ss, NT, stats, dss, NT)
            self.tape.append((bwd, None))
        return y

    def inject(self, p, x: Tensor, ctx: Tensor, ctx_index: int) -> Tensor:
        assert ctx.shape[0] == x.shape[0] and ctx.shape[2] == x.shape[2], "context `channels` shape mismatch"
        ctx = ctx.contiguous()
        y = ops.conv1d(x, p.conv.weight, p.conv.bias, x2=ctx, res=x)
        if self.need_grad:
            def bwd(gy):
                ops.conv1d_wgrad(x, gy, 1, x2=ctx, dw=self.g(p.conv.weight), dbias=self.g(p.conv.bias))
                C = x.shape[1]
                w = p.conv.weight
                # data gradient w.r.t. x only uses the first C input channels of the 1x1 weight
                dx = ops.conv1d(gy, w[:, :C, :].contiguous(), None, transposed=True, res=gy)
                if self.ctx_grads is not None and self.ctx_needs[ctx_index]:
                    # the down- and the up-branch InjectChannelsItem of a depth read the same context tensor:
                    # the second one to run accumulates through the conv's residual epilogue
                    prev = self.ctx_grads[ctx_index]
                    self.ctx_grads[ctx_index] = ops.conv1d(gy, w[:, C:, :].contiguous(), None, transposed=True, res=prev)
                return dx
            self.tape.append((bwd, None))
        return y

    def attention(self, p, x: Tensor, context: Optional[Tensor]) -> Tensor:
        from . import attention as attn_host
        return attn_host.attention_item(self, p, x, context)

    def run_items(self, d: int, which: str, mods, x: Tensor, embedding, channels) -> Tensor:
        if self.need_grad and self.wpark is not None:  # (first on the tape = last in the backward of this side of the block)
            self.tape.append((self._flush_wpark, None))
        for i, (t, p) in enumerate(zip(self.net.item_types[d], mods)):
            if t == ITEM_RESNET:
                x = self.resnet(p, x)
            elif t == ITEM_MODULATION:
                nxt = self.net.item_types[d][i + 1] if i + 1 < len(mods) else None
                # (a cross-attention item without the context bank normalises its context itself: only `norm` is taken along)
                attn_next = (mods[i + 1], nxt == ITEM_ATTENTION) if nxt in (ITEM_ATTENTION, ITEM_CROSS_ATTENTION) else None
                x = self.modulation((d, which, i), x, attn_next)
            elif t == ITEM_INJECT:
                assert channels is not None and channels[d] is not None, f"Missing context `channels` at depth {d}"
                x = self.inject(p, x, channels[d], self.ctx_index[d])
            elif t == ITEM_ATTENTION:
                x = self.attention(p, x, None)
            elif t == ITEM_CROSS_ATTENTION:
                assert embedding is not None, "You must provide a context when using context_features"
                x = self.attention(p, x, embedding)
        return x

    def _flush_wpark(self, gy):
        self.wpark.flush()
        return gy

    # -- block recursion ------------------------------------------------------------------
    def block(self, d: int, x: Tensor, x2: Optional[Tensor], embedding, channels, need_dx: bool) -> Tensor:
        n = self.net
        if d == len(n.blocks):
            return x
        blk = n.blocks[d]
        f, NT = n.factors[d], n.bank_total
        native = f in NATIVE_FACTORS          # strided-conv / upsample-loader kernel variants exist for 1, 2, 4
        has_adapter = hasattr(blk, "skip_adapter")
        if has_adapter:
            skip = ops.conv1d(x, blk.skip_adapter.weight, blk.skip_adapter.bias, x2=x2)
        else:
            assert x2 is None
            skip = x
        wd = blk.down.weight
        self.gn = self.gn_part_for(n.channels[d])  # the first item of every depth is a ResnetItem (GroupNorm of h0)
        if native:
            xs = x2s = None
            h0 = ops.conv1d(x, wd, blk.down.bias, stride=f, x2=x2, gn=self.gn)
        else:  # Conv1d(kernel = stride = f) == 1x1 conv over the space-to-depth view of its input
            xs = ops.unshuffle(x, f)
            x2s = ops.unshuffle(x2, f) if x2 is not None else None
            h0 = ops.conv1d(xs, wd.view(wd.shape[0], -1, 1), blk.down.bias, x2=x2s, gn=self.gn)
        tape_mark_down = len(self.tape)
        h = self.run_items(d, "down", blk.items_down, h0, embedding, channels)
        h = self.block(d + 1, h, None, embedding, channels, True)
        h = self.run_items(d, "up", blk.items_up, h, embedding, channels)
        modulate = n.use_modulation
        if modulate:
            # y = skip + scale[b,c] * (conv_k3(nearest_up_f(h)) + bias); u keeps the pre-merge value for the backward
            sc, dsc = self.ss((d, "skip"))
            u = torch.empty((x.shape[0], blk.out_ch, h.shape[2] * f), dtype=torch.float32, device=x.device) \
                if self.need_grad else None
            self.gn = self.gn_part_for(blk.out_ch) if d > 0 else None  # the outer depth's first up ResnetItem reads y
            y = ops.conv1d(h, blk.up.weight, blk.up.bias, pad=1, up=f, e_scale=sc, e_bstride=NT, res=skip, out_pre=u,
                           gn=self.gn)
        else:
            self.gn = None
            # SkipCat: y = Conv1x1(cat[skip * 2^-1/2, u]) -- the concat is the conv's two input pointers
            u = ops.conv1d(h, blk.up.weight, blk.up.bias, pad=1, up=f)
            skip_s = ops.axpby(SKIP_CAT_SCALE, skip)
            y = ops.conv1d(skip_s, blk.skip_cat.weight, blk.skip_cat.bias, x2=u)
        if self.need_grad:
            h_up = h

            def bwd_up(gy):
                if modulate:
                    du = ops.skipmod_bwd(gy, u, sc, NT, dsc, NT)
                    gskip = gy
                else:
                    wc, C = blk.skip_cat.weight, blk.out_ch
                    ops.conv1d_wgrad(skip_s, gy, 1, x2=u, dw=self.g(wc), dbias=self.g(blk.skip_cat.bias))
                    gskip = ops.conv1d(gy, wc[:, :C, :].contiguous(), None, transposed=True)
                    gskip = ops.axpby(SKIP_CAT_SCALE, gskip, out=gskip)
                    du = ops.conv1d(gy, wc[:, C:, :].contiguous(), None, transposed=True)
                ops.conv1d_wgrad(h_up, du, 3, pad=1, up=f, dw=self.g(blk.up.weight), dbias=self.g(blk.up.bias), park=self.wpark)
                if native:
                    gh = ops.conv1d(du, blk.up.weight, None, pad=1, transposed=True, store=2 if f > 1 else 0, sp=f)
                else:  # gradient of the nearest upsample = sum over the f replicas of each source position
                    gh = ops.pool_sum(ops.conv1d(du, blk.up.weight, None, pad=1, transposed=True), f)
                self.skip_grads.append(gskip)
                return gh

            def bwd_down(gh):
                gskip = self.skip_grads.pop()
                if native:
                    ops.conv1d_wgrad(x, gh, f, stride=f, x2=x2, dw=self.g(wd), dbias=self.g(blk.down.bias), park=self.wpark)
                else:
                    ops.conv1d_wgrad(xs, gh, 1, x2=x2s, dw=self.g(wd).view(wd.shape[0], -1, 1),
                                     dbias=self.g(blk.down.bias))
                if has_adapter:
                    ops.conv1d_wgrad(x, gskip, 1, x2=x2, dw=self.g(blk.skip_adapter.weight),
                                     dbias=self.g(blk.skip_adapter.bias))
                if not need_dx:
                    return None
                assert x2 is None, "input gradient through an appended-channel input is not needed on the hot path"
                if has_adapter:
                    gx = ops.conv1d(gskip, blk.skip_adapter.weight, None, transposed=True)
                else:
                    gx = gskip
                if f == 1:
                    return ops.conv1d(gh, wd, None, transposed=True, res=gx)
                M, R, KT = wd.shape
                return ops.conv1d(gh, wd.view(M, R * KT, 1), None, transposed=True, store=1, sp=f, res=gx)

            # tape order: [..., bwd_down, items_down..., inner..., items_up..., bwd_up]
            self.tape.insert(tape_mark_down, (bwd_down, d))   # tag d: block d's parameter gradients are complete
            self.tape.append((bwd_up, None))
        return y


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: UNetV0Net, x, time, features, embedding, x_append, channels, n_ctx, cond_pre, grad_on, *rest):
        ctx_list, params = rest[:n_ctx], rest[n_ctx:]
        need_grad = bool(grad_on) and any(ctx.needs_input_grad)
        run = _Run(net, need_grad)
        run.skip_grads = []
        run.ctx_index = {}
        j = 0
        for d, c in enumerate(channels or []):
            if c is not None:
                run.ctx_index[d] = j
                j += 1
        run.ctx_grads = [None] * n_ctx if need_grad else None
        run.ctx_needs = [bool(c.requires_grad) for c in ctx_list]
        x = x.contiguous()
        x2 = x_append.contiguous() if x_append is not None else None
        run.conditioning(time, features, cond_pre if not need_grad else None)
        run.emb_grad = None
        run.want_emb_grad = bool(embedding is not None and ctx.needs_input_grad[4])
        run.ctx_bank = None
        if embedding is not None and os.environ.get("ADP_CTX_BANK", "1") != "0":
            from .attention import CtxBank
            run.ctx_bank = CtxBank.prepare(run, embedding.contiguous())
            if run.ctx_bank is not None:  # (visible to tests / bench: which path a step took)
                net._ctx_bank_runs = getattr(net, "_ctx_bank_runs", 0) + 1
        y = run.block(0, x, x2, embedding.contiguous() if embedding is not None else None, channels,
                      need_dx=bool(ctx.needs_input_grad[1]))
        ctx.run = run
        ctx.params = params
        ctx.n_ctx = n_ctx
        ctx.has_features = features is not None
        ctx.has_embedding = embedding is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        run: _Run = ctx.run
        if run is None:
            raise RuntimeError("the U-Net tape was released by a previous backward through this forward; "
                               "run the forward again (retain_graph is not supported by the one-node U-Net)")
        net = run.net
        named = net._named_params()
        offs = net._param_offsets()
        total = max(b for _, b in offs.values())
        flat = _grad_buffer((total,), gy.device)  # every parameter's slice is overwritten by its gradient kernel
        run.flat = flat
        views = []
        for name, p in named:  # (returned to autograd in parameter order; the views point where the layout says)
            a, b = offs[name]
            v = flat[a:b].view(p.shape)
            run.grads[name] = v
            views.append(v)
        # data-parallel hook: called with (flat, start, end) as soon as a contiguous region of the flat
        # gradient buffer is final (deepest blocks first), and with (flat, None, None) at the very end
        hook = getattr(net, "_grad_ready_hook", None)
        # the conditioning bank's gradient: one launch at the end when nobody wants a depth's rows early; under the hook it
        # stays per depth ON PURPOSE (184 MB in one trailing bucket would be all-reduced after the backward instead of under it)
        defer_bank = hook is None and os.environ.get("ADP_BANK_DEFER", "1") != "0"  # (A/B switch)
        ctx_depth = net.ctx_final_depth()
        g = gy.contiguous()
        for fn, tag in reversed(run.tape):
            g = fn(g)
            if tag is not None:
                if run.wpark is not None:
                    run.wpark.flush()  # (a block's gradients are final when its tag is reached: the data-parallel hook)
                ra, rb = run.bank_grad_for_depth(tag, defer=defer_bank)
                if hook is not None:
                    if run.wpark is not None:
                        run.wpark.join()
                    hook(flat, *net.block_param_range(tag))
                    if rb > ra:  # this depth's weight rows of the conditioning bank (the small bias goes out at the end)
                        w0 = offs["bank_weight"][0]
                        hook(flat, w0 + ra * net.mf, w0 + rb * net.mf)
                if tag == ctx_depth:
                    # the last CrossAttentionItem has run its backward: the context bank's one weight gradient / un-fold /
                    # data gradient, at the earliest legal point -- and the context side leaves as ONE bucket (it is its own
                    # region of the flat buffer; on the per-item path the same region is simply final by now)
                    if run.ctx_bank is not None:
                        run.ctx_bank.backward(run)
                        if hook is not None:
                            net._ctx_bank_hooked_backwards = getattr(net, "_ctx_bank_hooked_backwards", 0) + 1
                    if hook is not None:
                        ca, cb = net.ctx_param_range()
                        if cb > ca:
                            hook(flat, ca, cb)
        run.mod_sums.flush()  # (nothing is left when every Modulation belongs to a tagged block)
        if run.wpark is not None:
            run.wpark.flush()
        if defer_bank:
            run.bank_grad_all()
        dfeat = run.conditioning_backward()
        if run.wpark is not None:
            run.wpark.join()  # (side-stream weight gradients: final before anybody can read the flat buffer)
        if hook is not None:
            for a, b in net.nonblock_param_ranges():
                hook(flat, a, b)
            hook(flat, None, None)
        gx = g if ctx.needs_input_grad[1] else None
        gfeat = dfeat if (ctx.has_features and ctx.needs_input_grad[3]) else None
        gemb = None
        if ctx.has_embedding and ctx.needs_input_grad[4] and run.emb_grad is not None:
            gemb = run.emb_grad.transpose(1, 2)  # [B, E, m] -> the embedding's [B, m, E] (a view)
        gctx = tuple(run.ctx_grads) if run.ctx_grads is not None else (None,) * ctx.n_ctx
        # AccumulateGrad adopts an incoming gradient without a copy only when nobody else holds it: drop every
        # reference of ours to the per-parameter views (they stay views of the one flat buffer RCCL reduced)
        run.grads.clear()
        run.flat = None
        ctx.run = None
        out = (None, gx, None, gfeat, gemb, None, None, None, None, None) + gctx + tuple(views)
        del views, flat, v
        return out
