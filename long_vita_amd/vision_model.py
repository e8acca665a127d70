"""The vision side of the model up to the LLM's embedding: InternViT tower -> class-token drop + pixel
shuffle -> pre-projection LayerNorm -> two-layer GELU projector.

Mirrors ``MegatronVisionModel`` (lcvlm_modellink/pretrain_lcvlm.py:304-556: ``forward_projection`` :439-451,
``forward_once`` :453-494, ``forward_chunk`` :496-507, ``forward`` :509-544, ``pixel_shuffle`` :546-556) and
``MultimodalProjector`` (core/models/vision/multimodal_projector.py:9-80, type "mlp") in the configuration every
shipped script uses: ``--vision-model-type intern_300m --vision-downsample-ratio 0.5 --vision-projector-type mlp
--vision-projector-pre-norm``.  Parameter names follow the reference (``vit.*``, ``pre_proj_layernorm.*``,
``vision_projection.encoder.linear_fc1/linear_fc2.weight``).

The tower runs without a graph under ``--vision-model-freeze`` (the long-context stages and the servers) and is
differentiable otherwise (stage 2); the projector trains in every stage.  LayerNorm, GELU and attention have HIP
forward and backward kernels, the GEMMs go through the BLAS library under torch autograd.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops
from .intern_vit import InternViTConfig, InternViTModel
from .vision import pixel_shuffle as _pixel_shuffle


class _ProjectorMLP(torch.nn.Module):
    def __init__(self, input_size: int, ffn_hidden_size: int, hidden_size: int):
        super().__init__()
        self.linear_fc1 = torch.nn.Linear(input_size, ffn_hidden_size, bias=False)
        self.linear_fc2 = torch.nn.Linear(ffn_hidden_size, hidden_size, bias=False)

    def forward(self, hidden_states):
        inter = ops.bias_gelu(F.linear(hidden_states, self.linear_fc1.weight), None)
        return F.linear(inter, self.linear_fc2.weight), None


class _ProjectorAffine(torch.nn.Module):
    """The one linear layer of the "affine" projector (multimodal_projector.py:37-49; no bias: add_bias_linear is False,
    pretrain_lcvlm.py:391-400).  Returns (output, bias) as Megatron's linear layers do."""

    def __init__(self, input_size: int, hidden_size: int):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, input_size))
        torch.nn.init.normal_(self.weight, std=0.02)

    def forward(self, hidden_states):
        return F.linear(hidden_states, self.weight), None


class MultimodalProjector(torch.nn.Module):
    """multimodal_projector.py:9-80: ``projector_type == "mlp"`` (every shipped script) is fc2(gelu(fc1(x))), no biases
    (pretrain_lcvlm.py:391-400: gated_linear_unit False, add_bias_linear False, ffn = the ViT's hidden size); "affine"
    (the argument's default, :962) is one linear layer ``encoder.weight``; anything else raises as the reference does."""

    def __init__(self, input_size: int, ffn_hidden_size: int, hidden_size: int, projector_type: str = "mlp"):
        super().__init__()
        self.projector_type = projector_type
        if projector_type == "mlp":
            self.encoder = _ProjectorMLP(input_size, ffn_hidden_size, hidden_size)
        elif projector_type == "affine":
            self.encoder = _ProjectorAffine(input_size, hidden_size)
        else:
            raise Exception(f"Unsupported multimodal projection type {projector_type}")

    def forward(self, hidden_states):
        encoder_output, encoder_output_bias = self.encoder(hidden_states)
        if encoder_output_bias is not None:
            encoder_output = encoder_output + encoder_output_bias
        return encoder_output


class MegatronVisionModel(torch.nn.Module):
    def __init__(self, hidden_size: int = 5120, vit_config: InternViTConfig | None = None, vision_downsample_ratio: float = 0.5,
                 vision_projector_pre_norm: bool = True, vision_model_freeze: bool = True, vision_projector_freeze: bool = False,
                 vision_context_parallel: bool = False, vision_downsample_stride: int = 1, vision_projector_type: str = "mlp"):
        super().__init__()
        vit_config = vit_config or InternViTConfig()
        # pretrain_lcvlm.py:415-420, 466-476: either the ratio form (0.5, every shipped script) or the stride form
        vision_downsample_stride = int(vision_downsample_stride)
        if vision_downsample_stride != 1:
            if vision_downsample_ratio not in (1, 1.0):
                raise NotImplementedError("vision_downsample_ratio and vision_downsample_stride together are not built: set the ratio to 1")
        elif vision_downsample_ratio != 0.5:
            raise NotImplementedError("only vision_downsample_ratio=0.5 (or a vision_downsample_stride) is built")
        self.vision_context_parallel = vision_context_parallel
        self.vision_downsample_ratio = vision_downsample_ratio
        self.vision_downsample_stride = vision_downsample_stride
        self.add_class_token = vit_config.add_class_token
        self.vision_model_freeze = vision_model_freeze
        self.vision_projector_freeze = vision_projector_freeze
        self.vit = InternViTModel(vit_config, vision_context_parallel=vision_context_parallel, freeze=vision_model_freeze)
        self.vision_seq_length = self.vit.seq_length
        if vision_downsample_stride != 1:
            side = int(round(self.vit.num_patches ** 0.5))
            self.image_token_length = (-(-side // vision_downsample_stride)) ** 2
            proj_input_size = vit_config.hidden_size * vision_downsample_stride ** 2
        else:
            self.image_token_length = int(self.vit.num_patches * vision_downsample_ratio ** 2)
            proj_input_size = vit_config.hidden_size * int(1 / vision_downsample_ratio) ** 2
        self.vision_projection = MultimodalProjector(proj_input_size, vit_config.hidden_size, hidden_size, vision_projector_type)
        self.pre_proj_layernorm = torch.nn.LayerNorm(proj_input_size) if vision_projector_pre_norm else torch.nn.Identity()

    def forward_projection(self, vit_output):
        """pretrain_lcvlm.py:439-451."""
        ln = self.pre_proj_layernorm
        if isinstance(ln, torch.nn.LayerNorm):
            vit_output = ops.layernorm(vit_output, ln.weight, ln.bias, ln.eps)
        return self.vision_projection(vit_output)

    def forward_once(self, images, attention_mask=None):
        """pretrain_lcvlm.py:453-494: [n, 3, H, W] -> [n, image_token_length, hidden]."""
        vit_output = self.vit(images, attention_mask)           # without a graph when the tower is frozen
        # class-token drop, [n, 32, 32, c] view and pixel_shuffle(0.5), flattened: one pass over the data
        skip = 1 if self.add_class_token else 0
        if self.vision_downsample_stride != 1:      # :472-476
            vit_output = ops.pixel_shuffle_v2_tokens(vit_output, skip=skip, scale_stride=self.vision_downsample_stride)
        else:
            vit_output = ops.pixel_shuffle_tokens(vit_output, skip=skip)
        if self.vision_projector_freeze:
            with torch.no_grad():
                return self.forward_projection(vit_output)
        return self.forward_projection(vit_output)

    def forward_chunk(self, images, attention_mask=None):
        """pretrain_lcvlm.py:496-507: at most 256 images per pass through the tower."""
        outs = [self.forward_once(chunk, attention_mask) for chunk in torch.split(images, 256, dim=0)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, **kw_args):
        """pretrain_lcvlm.py:509-544 without vision context parallelism."""
        return self.forward_chunk(kw_args["images"], None)

    def pixel_shuffle_v2(self, x, scale_stride=2):
        """pretrain_lcvlm.py:558-570."""
        from .vision import pixel_shuffle_v2 as _v2
        return _v2(x, scale_stride)

    def pixel_shuffle(self, x, scale_factor=0.5):
        """pretrain_lcvlm.py:546-556."""
        return _pixel_shuffle(x, scale_factor)
