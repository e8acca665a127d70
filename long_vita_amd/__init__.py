"""MI355X-native attention hot path for Long-VITA (gfx950 HIP kernels behind a C ABI).

Sub-modules
  _lib                    ctypes binding of liblvattn.so (include/lvattn.h)
  ops                     tensor-level wrappers + autograd functions
  fusion_attention        npu_fusion_attention-shaped adapter
  dot_product_attention   flash_attention_forward / do_ring_context_parallel mirrors
  ring_context_parallel   ringattn_context_parallel (zig-zag causal ring over RCCL)
  cp_partition            zig-zag batch slicing, image index remap, un-zig-zag
  packing                 sample packing -> actual_seq_len / image indices
  vision                  pixel-shuffle token compression, feature scatter
  rotary_pos_embedding    RotaryEmbedding angle tables, apply_rotary_pos_emb on the HIP kernel
  state                   the ambient state the reference reads (args, actual_seq_len, CP group)
"""
__version__ = "0.1.0"
