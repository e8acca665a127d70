"""CPU restatement of the InternViT encoder forward.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path.

What is pinned and what is not.  The wiring of one layer (LayerNorm -> attention (+bias) -> residual +
x * ls1 -> LayerNorm -> MLP (+bias) -> residual + x * ls2) and of the model (conv patch projection, class
token, position embedding, [b,s,h] <-> [s,b,h]) follows the reference's own code
(lcvlm_modellink/core/models/vision/intern_vit_model.py:32-89 and :190-261) and is pinned by
tests/golden/vit.npz, produced by running those two ``forward`` functions themselves over the
submodules below.  The submodules are Megatron-LM core_r0.6.0 classes (SelfAttention, MLP, LayerNorm as
vit_layer_specs.py:80-103 and pretrain_lcvlm.py:177-217 configure them), a dependency that is absent from
the reference tree: their published algorithm is restated here and is "parity unpinned" beyond the eager
attention core, which tests/golden/eager_attention.npz pins.

All arithmetic in fp32 on bf16-representable inputs; the tests compare the HIP path to this within
2e-2 relative (bf16 intermediates on the device).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """Megatron 'LayerNorm' normalization (pretrain_lcvlm.py:210): biased variance over the last dim."""
    x = x.float()
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * weight.float() + bias.float()


def gelu(x: torch.Tensor) -> torch.Tensor:
    """torch.nn.functional.gelu, erf form (pretrain_lcvlm.py:200; bias_activation_fusion False at :207)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def split_qkv(mixed: torch.Tensor, heads: int):
    """Megatron SelfAttention.get_query_key_value_tensors with num_query_groups == heads: the linear_qkv
    output [s, b, heads * 3 * hn] holds, per head, [q | k | v] (the layout ckpt_converter_intern_vit.py:28-40
    writes).  Returns q, k, v as [s, b, heads, hn]."""
    s, b, three_h = mixed.shape
    hn = three_h // (3 * heads)
    m = mixed.view(s, b, heads, 3 * hn)
    return m[..., :hn], m[..., hn:2 * hn], m[..., 2 * hn:]


def self_attention(x: torch.Tensor, p: dict, heads: int):
    """Megatron SelfAttention with AttnMaskType.no_mask (vit_layer_specs.py:84): returns (output, bias) with
    the projection bias handed back separately (skip_bias_add), which intern_vit_model.py:59-60 then adds."""
    s, b, h = x.shape
    mixed = F.linear(x.float(), p["linear_qkv.weight"].float(), p["linear_qkv.bias"].float())
    q, k, v = split_qkv(mixed, heads)
    hn = q.shape[-1]
    scores = torch.einsum("sbnd,tbnd->bnst", q, k) / math.sqrt(hn)
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bnst,tbnd->sbnd", probs, v).reshape(s, b, h)
    return F.linear(ctx, p["linear_proj.weight"].float()), p["linear_proj.bias"].float()


def mlp(x: torch.Tensor, p: dict):
    """Megatron MLP without bias-activation fusion: fc2(gelu(fc1(x) + b1)), fc2's bias handed back."""
    inter = F.linear(x.float(), p["linear_fc1.weight"].float()) + p["linear_fc1.bias"].float()
    return F.linear(gelu(inter), p["linear_fc2.weight"].float()), p["linear_fc2.bias"].float()


def layer_forward(x: torch.Tensor, p: dict, heads: int, eps: float) -> torch.Tensor:
    """intern_vit_model.py:32-89 on x [s, b, h]; ``p`` holds the layer's parameters by Megatron name."""
    x = x.float()
    a, ab = self_attention(layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps),
                           {k[len("self_attention."):]: v for k, v in p.items() if k.startswith("self_attention.")}, heads)
    x = x + (a + ab) * p["ls1"].float()
    m, mb = mlp(layer_norm(x, p["pre_mlp_layernorm.weight"], p["pre_mlp_layernorm.bias"], eps),
                {k[len("mlp."):]: v for k, v in p.items() if k.startswith("mlp.")})
    return x + (m + mb) * p["ls2"].float()


def model_forward(img: torch.Tensor, sd: dict, heads: int, patch: int, eps: float) -> torch.Tensor:
    """intern_vit_model.py:190-261 (class token on, no vision context parallelism): img [n, 3, H, W] ->
    [n, seq, h]; ``sd`` is a Megatron-named state dict."""
    x = F.conv2d(img.float(), sd["conv1.weight"].float(), sd["conv1.bias"].float(), stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["class_token"].float().expand(x.shape[0], -1, -1), x], dim=1)
    x = x + sd["position_embeddings.weight"].float()[: x.shape[1]][None]
    x = x.permute(1, 0, 2)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.layers."))
    for i in range(n_layers):
        pre = f"decoder.layers.{i}."
        x = layer_forward(x, {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, heads, eps)
    return x.permute(1, 0, 2).contiguous()


def pixel_shuffle(x: torch.Tensor, scale_factor: float = 0.5) -> torch.Tensor:
    """pretrain_lcvlm.py:546-556 (pinned bit-exact by tests/golden/indexing.npz)."""
    n, w, h, c = x.shape
    x = x.reshape(n, w, int(h * scale_factor), int(c / scale_factor)).permute(0, 2, 1, 3)
    x = x.reshape(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
    return x.permute(0, 2, 1, 3).contiguous()


def projector_mlp(x: torch.Tensor, sd: dict):
    """Megatron MLP as pretrain_lcvlm.py:391-400 configures it for the projector: no gating, no biases, erf GELU."""
    inter = gelu(F.linear(x.float(), sd["vision_projection.encoder.linear_fc1.weight"].float()))
    return F.linear(inter, sd["vision_projection.encoder.linear_fc2.weight"].float()), None


def forward_projection(x: torch.Tensor, sd: dict, eps: float = 1e-5) -> torch.Tensor:
    """pretrain_lcvlm.py:439-451 + multimodal_projector.py:53-80: torch.nn.LayerNorm (default eps) then the MLP."""
    x = layer_norm(x, sd["pre_proj_layernorm.weight"], sd["pre_proj_layernorm.bias"], eps)
    out, bias = projector_mlp(x, sd)
    return out if bias is None else out + bias


def vision_model_forward(images: torch.Tensor, sd: dict, heads: int, patch: int, vit_eps: float) -> torch.Tensor:
    """pretrain_lcvlm.py:453-507 (class token dropped, downsample ratio 0.5, pre-norm projector): images
    [n, 3, H, W] -> [n, tokens / 4, llm_hidden]."""
    vit_sd = {k[4:]: v for k, v in sd.items() if k.startswith("vit.")}
    outs = []
    for chunk in torch.split(images, 256, dim=0):
        y = model_forward(chunk, vit_sd, heads, patch, vit_eps)[:, 1:, :]
        g = int(y.shape[1] ** 0.5)
        y = pixel_shuffle(y.reshape(y.shape[0], g, g, -1), 0.5)
        outs.append(forward_projection(y.reshape(y.shape[0], -1, y.shape[-1]), sd))
    return torch.cat(outs, dim=0)
