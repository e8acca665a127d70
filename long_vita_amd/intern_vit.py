"""InternViT encoder forward (the frozen vision tower of the stage-3/4 and inference scripts) on MI355X.

Mirrors ``InternViTModel`` / ``InternViTTransformerLayer`` of
lcvlm_modellink/core/models/vision/intern_vit_model.py (:18-89, :93-261), built there from Megatron modules
(vit_layer_specs.py:80-103) with the InternViT-300M configuration of pretrain_lcvlm.py:177-217.  Parameter
names are Megatron's (``conv1``, ``class_token``, ``position_embeddings``,
``decoder.layers.N.{input_layernorm, self_attention.linear_qkv, self_attention.linear_proj,
pre_mlp_layernorm, mlp.linear_fc1, mlp.linear_fc2, ls1, ls2}``), so a checkpoint written by
ckpt_converter_intern_vit.py loads with ``load_state_dict``; ``linear_qkv`` keeps Megatron's per-head
``[q | k | v]`` interleaving (:28-40 there).

Work split: the GEMMs go to the BLAS library through ``torch.nn.functional.linear`` (plain library GEMMs);
everything between them is a HIP kernel of liblvattn (patch extraction, token assembly, LayerNorm,
attention straight on the strided qkv buffer, bias + LayerScale + residual, bias + GELU).  With ``freeze=True``
(``--vision-model-freeze``, how the long-context stages and the servers run the tower) the forward runs without
a graph; otherwise every piece is differentiable (HIP backward kernels for attention, LayerNorm and GELU, torch
for the GEMMs and the small per-column reductions), which is what stage 2 needs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops


@dataclass
class InternViTConfig:
    """InternViT-300M-448px as the reference configures it (pretrain_lcvlm.py:177-217; stage-4 script)."""
    hidden_size: int = 1024
    ffn_hidden_size: int = 4096
    num_attention_heads: int = 16
    num_layers: int = 24
    patch_dim: int = 14
    img_h: int = 448
    img_w: int = 448
    class_token_len: int = 1
    add_class_token: bool = True
    layernorm_epsilon: float = 1e-6


class _Norm(torch.nn.Module):
    def __init__(self, h):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(h))
        self.bias = torch.nn.Parameter(torch.zeros(h))


class _SelfAttention(torch.nn.Module):
    def __init__(self, h):
        super().__init__()
        self.linear_qkv = torch.nn.Linear(h, 3 * h, bias=True)
        self.linear_proj = torch.nn.Linear(h, h, bias=True)


class _MLP(torch.nn.Module):
    def __init__(self, h, f):
        super().__init__()
        self.linear_fc1 = torch.nn.Linear(h, f, bias=True)
        self.linear_fc2 = torch.nn.Linear(f, h, bias=True)


class InternViTTransformerLayer(torch.nn.Module):
    def __init__(self, cfg: InternViTConfig):
        super().__init__()
        h = cfg.hidden_size
        self.input_layernorm = _Norm(h)
        self.self_attention = _SelfAttention(h)
        self.pre_mlp_layernorm = _Norm(h)
        self.mlp = _MLP(h, cfg.ffn_hidden_size)
        self.ls1 = torch.nn.Parameter(0.01 * torch.ones(h))
        self.ls2 = torch.nn.Parameter(0.01 * torch.ones(h))
        self.heads = cfg.num_attention_heads
        self.eps = cfg.layernorm_epsilon

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """hidden_states [s, b, h] bf16 -> [s, b, h] (intern_vit_model.py:32-89)."""
        s, b, h = hidden_states.shape
        hn = h // self.heads
        sa, mlp = self.self_attention, self.mlp
        x = ops.layernorm(hidden_states, self.input_layernorm.weight, self.input_layernorm.bias, self.eps)
        qkv = F.linear(x.view(s * b, h), sa.linear_qkv.weight, sa.linear_qkv.bias)
        # Megatron layout: per head [q | k | v]; as rows = s, heads = b * np with a head stride of 3 * hn the
        # three operands are strided views of the GEMM output (no copies, no transposes)
        hv = qkv.view(s, b * self.heads, 3 * hn)
        if torch.is_grad_enabled() and qkv.requires_grad:
            ctx, _ = ops.varlen_attention(hv[:, :, :hn], hv[:, :, hn:2 * hn], hv[:, :, 2 * hn:], [0, s], 1.0 / math.sqrt(hn), False)
        else:
            ctx, _ = ops.attn_fwd_dense_launch(hv[:, :, :hn], hv[:, :, hn:2 * hn], hv[:, :, 2 * hn:], 1.0 / math.sqrt(hn), False)
        proj = F.linear(ctx.reshape(s * b, h), sa.linear_proj.weight)
        x1 = ops.bias_scale_residual(proj, sa.linear_proj.bias, self.ls1, hidden_states.view(s * b, h))
        y = ops.layernorm(x1, self.pre_mlp_layernorm.weight, self.pre_mlp_layernorm.bias, self.eps)
        act = ops.bias_gelu(F.linear(y, mlp.linear_fc1.weight), mlp.linear_fc1.bias)
        out = ops.bias_scale_residual(F.linear(act, mlp.linear_fc2.weight), mlp.linear_fc2.bias, self.ls2, x1)
        return out.view(s, b, h)

    def forward_fused(self, hidden_states: torch.Tensor, normed: torch.Tensor, next_layer=None):
        """The same layer without a graph (frozen tower), every LayerNorm folded into the pass that produces its input:
        takes the layer input AND its input_layernorm output, returns (layer output, the NEXT layer's input_layernorm
        of it, or None after the last layer).  Bit-identical to ``forward``."""
        s, b, h = hidden_states.shape
        hn = h // self.heads
        sa, mlp = self.self_attention, self.mlp
        qkv = F.linear(normed.view(s * b, h), sa.linear_qkv.weight, sa.linear_qkv.bias)
        hv = qkv.view(s, b * self.heads, 3 * hn)
        ctx, _ = ops.attn_fwd_dense_launch(hv[:, :, :hn], hv[:, :, hn:2 * hn], hv[:, :, 2 * hn:], 1.0 / math.sqrt(hn), False)
        proj = F.linear(ctx.reshape(s * b, h), sa.linear_proj.weight)
        x1, y = ops.bias_scale_residual_layernorm_launch(proj, sa.linear_proj.bias, self.ls1, hidden_states.view(s * b, h),
                                                         self.pre_mlp_layernorm.weight, self.pre_mlp_layernorm.bias, self.eps)
        act = ops.bias_gelu_launch(F.linear(y, mlp.linear_fc1.weight), mlp.linear_fc1.bias)
        fc2 = F.linear(act, mlp.linear_fc2.weight)
        if next_layer is None:
            return ops.bias_scale_residual_launch(fc2, mlp.linear_fc2.bias, self.ls2, x1).view(s, b, h), None
        out, nxt = ops.bias_scale_residual_layernorm_launch(fc2, mlp.linear_fc2.bias, self.ls2, x1, next_layer.input_layernorm.weight,
                                                            next_layer.input_layernorm.bias, next_layer.eps)
        return out.view(s, b, h), nxt.view(s, b, h)


class _Decoder(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = torch.nn.ModuleList([InternViTTransformerLayer(cfg) for _ in range(cfg.num_layers)])


class InternViTModel(torch.nn.Module):
    def __init__(self, cfg: InternViTConfig | None = None, vision_context_parallel: bool = False, freeze: bool = True):
        super().__init__()
        cfg = cfg or InternViTConfig()
        self.freeze = freeze
        if vision_context_parallel:
            raise NotImplementedError("vision context parallelism (intern_vit_model.py:218-233) is not built")
        if not cfg.add_class_token:
            raise NotImplementedError("the shipped InternViT configuration uses a class token")
        assert cfg.img_h % cfg.patch_dim == 0 and cfg.img_w % cfg.patch_dim == 0
        self.cfg = cfg
        self.num_patches = (cfg.img_h // cfg.patch_dim) * (cfg.img_w // cfg.patch_dim)
        self.seq_length = self.num_patches + cfg.class_token_len
        h = cfg.hidden_size
        self.conv1 = torch.nn.Conv2d(3, h, kernel_size=cfg.patch_dim, stride=cfg.patch_dim, bias=True)
        self.position_embeddings = torch.nn.Embedding(self.seq_length, h)
        self.class_token = torch.nn.Parameter(torch.randn(1, cfg.class_token_len, h))
        self.decoder = _Decoder(cfg)
        self._wpad = None

    def _patch_weight(self) -> torch.Tensor:
        """conv1.weight as a GEMM operand [h, kpad], the reduction dimension zero-padded to a multiple of 16."""
        w = self.conv1.weight
        k = w[0].numel()
        kpad = (k + 15) // 16 * 16
        key = (w.data_ptr(), w._version, w.dtype, w.device)
        if torch.is_grad_enabled() and w.requires_grad:
            return F.pad(w.reshape(w.shape[0], k), (0, kpad - k))      # differentiable, rebuilt per step
        if self._wpad is None or self._wpad[0] != key:
            wp = torch.zeros((w.shape[0], kpad), dtype=w.dtype, device=w.device)
            wp[:, :k] = w.detach().reshape(w.shape[0], k)
            self._wpad = (key, wp)
        return self._wpad[1]

    def forward(self, x: torch.Tensor, attention_mask=None) -> torch.Tensor:
        """x [batch, 3, img_h, img_w] bf16 -> [batch, seq, hidden] (intern_vit_model.py:190-261)."""
        if self.freeze:
            with torch.no_grad():
                return self._forward(x, attention_mask)
        return self._forward(x, attention_mask)

    def _forward(self, x: torch.Tensor, attention_mask=None) -> torch.Tensor:
        if attention_mask is not None:
            raise NotImplementedError("InternViT attends without a mask (vit_layer_specs.py:84)")
        cfg = self.cfg
        n = x.shape[0]
        wp = self._patch_weight()
        patches = ops.patchify(x, cfg.patch_dim, wp.shape[1])
        proj = torch.matmul(patches, wp.t()).view(n, self.num_patches, cfg.hidden_size)
        tokens = ops.embed_tokens(proj, self.conv1.bias, self.class_token, self.position_embeddings.weight)
        layers = list(self.decoder.layers)
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())) and layers:
            # frozen tower: no graph, each LayerNorm rides on the pass that writes its input
            first = layers[0]
            normed = ops.layernorm_launch(tokens, first.input_layernorm.weight, first.input_layernorm.bias, first.eps)
            for i, layer in enumerate(layers):
                tokens, normed = layer.forward_fused(tokens, normed, layers[i + 1] if i + 1 < len(layers) else None)
            return tokens.permute(1, 0, 2).contiguous()
        for layer in layers:
            tokens = layer(tokens)
        return tokens.permute(1, 0, 2).contiguous()
