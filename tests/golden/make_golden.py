"""Generate the golden fixtures in this directory from the REFERENCE'S OWN code.

Runs only in the build container (needs /root/reference); the fixtures it
writes (small .npz files) are committed, this script is committed, no
reference source is copied.  Usage:  python tests/golden/make_golden.py

How: the reference imports Megatron / MindSpeed / ModelLink / torch_npu, none
of which is installed.  A ``sys.meta_path`` finder fabricates empty stand-in
modules for those roots (every attribute is an inert stub), the reference
modules are then imported unmodified, and the few ambient hooks the exercised
functions read (``get_args``, ``mpu`` ranks, the global memory buffer, the RNG
tracker, the scale-mask-softmax of Megatron) are set on the imported module
objects.  Functions that hard-code ``device='cuda'`` are run with
``torch.arange``/``torch.tensor`` wrapped to drop the device argument.

Fixtures written:
  eager_attention.npz  reference eager attention wrapper
                       (dot_product_attention.py:148-288) on seeded inputs
  indexing.npz         index_of_a_in_b, zig-zag CP slicing, src/tgt image
                       index remap (training/utils.py), collate_fn
                       (legacy/data/data_samplers.py), pixel_shuffle
                       (pretrain_lcvlm.py), get_pos_emb_on_this_cp_rank
                       (rotary_pos_embedding.py)
  packing.npz          Qwen2Dataset.add_ret / process_ret (cognitron_vl/data/dataset_qwen2.py)
  rope.npz             RotaryEmbedding.forward angle tables (incl. packed position ids),
                       apply_rotary_pos_emb_bshd / _thd on bf16 tensors (rotary_pos_embedding.py)
  vit.npz              InternViTTransformerLayer.forward / InternViTModel.forward (intern_vit_model.py)
                       run over the oracle's restatement of the Megatron submodules
  eager_attention_grad.npz  the same eager wrapper with torch autograd through it (dq / dk / dv), causal and
                       block-diagonal-causal (packed segments) masks, more than one 256-row query block
  eager_attention_band.npz  the eager wrapper + autograd under band masks (causal, and keys more than `band` positions
                       before the query masked too: pre_tockens / --sliding-window), plain and packed
  embedding_scatter.npz  LanguageModelEmbedding.forward (language_model_embedding.py:91-174) in bf16 with autograd:
                       the visual-feature scatter in its indices / pre_len / src+tgt / no-indices forms
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
STUB_ROOTS = ("megatron", "mindspeed", "torch_npu", "modellink", "apex", "amp_C", "cognitron_vl", "cv2", "decord", "natsort", "datasets")


class _StubMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub


class _Stub(metaclass=_StubMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Stub


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_shim():
    sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)


@contextlib.contextmanager
def no_cuda_device():
    """Drop device='cuda' from the tensor factories the reference calls."""
    saved = {}

    def wrap(fn):
        def inner(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k = dict(k)
                k.pop("device")
            return fn(*a, **k)
        return inner

    for name in ("arange", "tensor", "ones", "zeros", "empty"):
        saved[name] = getattr(torch, name)
        setattr(torch, name, wrap(saved[name]))
    try:
        yield
    finally:
        for name, fn in saved.items():
            setattr(torch, name, fn)


def ns(**k):
    return types.SimpleNamespace(**k)


# --------------------------------------------------------------------------
def gen_eager_attention():
    dpa = importlib.import_module("lcvlm_modellink.core.transformer.dot_product_attention")

    def smsoftmax(scores, mask):
        # Megatron FusedScaleMaskSoftmax (core_r0.6.0), unfused path with
        # attention_softmax_in_fp32 and scale=None: mask fill -10000, fp32
        # softmax, cast back.  Not part of the reference tree; restated.
        x = scores.float()
        if mask is not None:
            x = x.masked_fill(mask, -10000.0)
        return torch.softmax(x, dim=-1).to(scores.dtype)

    dpa.get_args = lambda: ns(use_flash_attn=False)
    dpa.parallel_state = ns(get_global_memory_buffer=lambda: ns(
        get_tensor=lambda shape, dtype, name: torch.empty(shape, dtype=dtype)))
    dpa.tensor_parallel = ns(get_cuda_rng_tracker=lambda: ns(fork=contextlib.nullcontext))
    wrapper = dpa.dot_product_attention_forward_wrapper(lambda *a, **k: None)

    out = {}
    cases = [
        # name, sq, b, np, ng, hn
        ("gqa_small", 48, 1, 4, 2, 16),
        ("gqa_b2", 33, 2, 10, 2, 32),
        ("mha_d64", 65, 1, 4, 4, 64),
        ("qwen_ratio", 96, 1, 10, 2, 128),
    ]
    g = torch.Generator().manual_seed(20241223)
    for name, sq, b, np_, ng, hn in cases:
        q = torch.randn(sq, b, np_, hn, generator=g).to(torch.bfloat16).float()
        k = torch.randn(sq, b, ng, hn, generator=g).to(torch.bfloat16).float()
        v = torch.randn(sq, b, ng, hn, generator=g).to(torch.bfloat16).float()
        mask = torch.triu(torch.ones(sq, sq, dtype=torch.bool), 1)[None, None]
        self = ns(num_attention_heads_per_partition=np_, num_query_groups_per_partition=ng,
                  alibi=None, norm_factor=float(hn) ** 0.5, attn_logit_softcapping=None,
                  square_alibi_mask=False, scale_mask_softmax=smsoftmax,
                  config=ns(sequence_parallel=True), attention_dropout=lambda x: x,
                  hidden_size_per_partition=np_ * hn)
        o = wrapper(self, q, k, v, mask, None, None)
        out[f"{name}.q"] = q.numpy()
        out[f"{name}.k"] = k.numpy()
        out[f"{name}.v"] = v.numpy()
        out[f"{name}.out"] = o.numpy()
        # non-causal (ViT branch semantics: no mask at all)
        o2 = wrapper(self, q, k, v, torch.zeros(1, 1, sq, sq, dtype=torch.bool), None, None)
        out[f"{name}.out_nomask"] = o2.numpy()
    np.savez_compressed(os.path.join(HERE, "eager_attention.npz"), **out)
    print("eager_attention.npz:", len(out), "arrays")


def _bf16_bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> uint16 bit patterns (half the bytes of the fp32 form, lossless)."""
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def gen_eager_attention_grad():
    """Outputs AND gradients of the reference's eager wrapper (dot_product_attention.py:148-288; torch autograd
    through its baddbmm / softmax / bmm), for plain causal masks and for the block-diagonal-causal mask that packed
    segments mean.  Inputs are bf16-valued fp32 tensors (stored as bf16 bit patterns); results are fp32."""
    dpa = importlib.import_module("lcvlm_modellink.core.transformer.dot_product_attention")

    def smsoftmax(scores, mask):   # Megatron FusedScaleMaskSoftmax, unfused fp32 path (not in the reference tree; restated)
        x = scores.float()
        if mask is not None:
            x = x.masked_fill(mask, -10000.0)
        return torch.softmax(x, dim=-1).to(scores.dtype)

    dpa.get_args = lambda: ns(use_flash_attn=False)
    dpa.parallel_state = ns(get_global_memory_buffer=lambda: ns(
        get_tensor=lambda shape, dtype, name: torch.empty(shape, dtype=dtype)))
    dpa.tensor_parallel = ns(get_cuda_rng_tracker=lambda: ns(fork=contextlib.nullcontext))
    wrapper = dpa.dot_product_attention_forward_wrapper(lambda *a, **k: None)
    cases = [
        # name, sq, np, ng, hn, cu_seqlens (None = one causal segment)
        ("qwen5to1_causal", 320, 5, 1, 128, None),
        ("qwen5to1_packed", 300, 5, 1, 128, [0, 37, 38, 170, 261, 300]),   # 1-token segment; 170..261 straddles row 256
        ("gqa_packed_d64", 200, 4, 2, 64, [0, 64, 65, 200]),
        ("mha_packed_small", 72, 2, 2, 16, [0, 1, 30, 31, 72]),            # oracle only (the kernels take d = 64 / 128)
    ]
    g = torch.Generator().manual_seed(20250101)
    out = {}
    for name, sq, np_, ng, hn, cu in cases:
        q = torch.randn(sq, 1, np_, hn, generator=g).to(torch.bfloat16)
        k = torch.randn(sq, 1, ng, hn, generator=g).to(torch.bfloat16)
        v = torch.randn(sq, 1, ng, hn, generator=g).to(torch.bfloat16)
        do = torch.randn(sq, 1, np_ * hn, generator=g).to(torch.bfloat16)
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
        mask = torch.triu(torch.ones(sq, sq, dtype=torch.bool), 1)
        if cu is not None:
            seg = torch.zeros(sq, dtype=torch.long)
            for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
                seg[a:b] = i
            mask = mask | (seg[:, None] != seg[None, :])
        self = ns(num_attention_heads_per_partition=np_, num_query_groups_per_partition=ng,
                  alibi=None, norm_factor=float(hn) ** 0.5, attn_logit_softcapping=None,
                  square_alibi_mask=False, scale_mask_softmax=smsoftmax,
                  config=ns(sequence_parallel=True), attention_dropout=lambda x: x,
                  hidden_size_per_partition=np_ * hn)
        o = wrapper(self, qf, kf, vf, mask[None, None], None, None)
        o.backward(do.float())
        out[f"{name}.meta"] = np.array([sq, np_, ng, hn], dtype=np.int64)
        out[f"{name}.cu"] = np.array(cu if cu is not None else [0, sq], dtype=np.int64)
        out[f"{name}.q"], out[f"{name}.k"], out[f"{name}.v"], out[f"{name}.dout"] = map(_bf16_bits, (q, k, v, do))
        out[f"{name}.out"] = o.detach().numpy()
        out[f"{name}.dq"], out[f"{name}.dk"], out[f"{name}.dv"] = qf.grad.numpy(), kf.grad.numpy(), vf.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "eager_attention_grad.npz"), **out)
    print("eager_attention_grad.npz:", len(out), "arrays")


def gen_eager_attention_band():
    """Outputs and autograd gradients of the reference's eager wrapper under BAND masks: causal, and a key more than
    ``band`` positions before the query masked as well (what pre_tockens / --sliding-window ask of the fused op,
    dot_product_attention.py:403-406, 454-467; the eager wrapper takes any mask), plain and packed."""
    dpa = importlib.import_module("lcvlm_modellink.core.transformer.dot_product_attention")

    def smsoftmax(scores, mask):   # Megatron FusedScaleMaskSoftmax, unfused fp32 path (not in the reference tree; restated)
        x = scores.float()
        if mask is not None:
            x = x.masked_fill(mask, -10000.0)
        return torch.softmax(x, dim=-1).to(scores.dtype)

    dpa.get_args = lambda: ns(use_flash_attn=False)
    dpa.parallel_state = ns(get_global_memory_buffer=lambda: ns(
        get_tensor=lambda shape, dtype, name: torch.empty(shape, dtype=dtype)))
    dpa.tensor_parallel = ns(get_cuda_rng_tracker=lambda: ns(fork=contextlib.nullcontext))
    wrapper = dpa.dot_product_attention_forward_wrapper(lambda *a, **k: None)
    cases = [
        # name, sq, np, ng, hn, cu_seqlens (None = one segment), band
        ("band_causal", 420, 5, 1, 128, None, 100),
        ("band_packed", 520, 2, 1, 128, [0, 200, 201, 520], 64),
        ("band_d64", 300, 4, 2, 64, None, 37),
        ("band_self_only", 70, 2, 1, 64, None, 0),
        ("band_wider_than_rows", 140, 2, 1, 128, None, 500),
    ]
    g = torch.Generator().manual_seed(20250303)
    out = {}
    for name, sq, np_, ng, hn, cu, band in cases:
        q = torch.randn(sq, 1, np_, hn, generator=g).to(torch.bfloat16)
        k = torch.randn(sq, 1, ng, hn, generator=g).to(torch.bfloat16)
        v = torch.randn(sq, 1, ng, hn, generator=g).to(torch.bfloat16)
        do = torch.randn(sq, 1, np_ * hn, generator=g).to(torch.bfloat16)
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
        ones = torch.ones(sq, sq, dtype=torch.bool)
        mask = torch.triu(ones, 1) | torch.tril(ones, -band - 1)     # True = masked: key after the row, or more than `band` before it
        if cu is not None:
            seg = torch.zeros(sq, dtype=torch.long)
            for i, (a, b) in enumerate(zip(cu[:-1], cu[1:])):
                seg[a:b] = i
            mask = mask | (seg[:, None] != seg[None, :])
        self = ns(num_attention_heads_per_partition=np_, num_query_groups_per_partition=ng,
                  alibi=None, norm_factor=float(hn) ** 0.5, attn_logit_softcapping=None,
                  square_alibi_mask=False, scale_mask_softmax=smsoftmax,
                  config=ns(sequence_parallel=True), attention_dropout=lambda x: x,
                  hidden_size_per_partition=np_ * hn)
        o = wrapper(self, qf, kf, vf, mask[None, None], None, None)
        o.backward(do.float())
        out[f"{name}.meta"] = np.array([sq, np_, ng, hn, band], dtype=np.int64)
        out[f"{name}.cu"] = np.array(cu if cu is not None else [0, sq], dtype=np.int64)
        out[f"{name}.q"], out[f"{name}.k"], out[f"{name}.v"], out[f"{name}.dout"] = map(_bf16_bits, (q, k, v, do))
        out[f"{name}.out"] = _bf16_bits(o.detach())          # (stored as bf16: the comparison's tolerance is 2e-2)
        out[f"{name}.dq"], out[f"{name}.dk"], out[f"{name}.dv"] = (x.grad.numpy().astype(np.float16) for x in (qf, kf, vf))
    np.savez_compressed(os.path.join(HERE, "eager_attention_band.npz"), **out)
    print("eager_attention_band.npz:", len(out), "arrays")


def gen_embedding_scatter():
    """LanguageModelEmbedding.forward (language_model_embedding.py:91-174) run unbound in bf16 with autograd, for each
    form of external_feature_dict.  The word-embedding lookup is a stand-in that returns a fresh non-leaf copy of a
    leaf tensor, so that the leaf's .grad is exactly the gradient that reaches the embedding output."""
    lme = importlib.import_module("lcvlm_modellink.core.models.common.embeddings.language_model_embedding")
    lme.mpu = ns(get_context_parallel_world_size=lambda: 1)
    fwd = lme.LanguageModelEmbedding.forward
    g = torch.Generator().manual_seed(606)
    b, s, h, n_img, n_tok = 2, 48, 32, 3, 8
    out = {"meta": np.array([b, s, h, n_img, n_tok], dtype=np.int64)}

    def run(tag, make_dict):
        we = torch.randn(b, s, h, generator=g).to(torch.bfloat16).requires_grad_(True)
        feats = torch.randn(n_img, n_tok, h, generator=g).to(torch.bfloat16).requires_grad_(True)
        do = torch.randn(s, b, h, generator=g).to(torch.bfloat16)
        self = ns(word_embeddings=lambda ids: we.clone(), add_position_embedding=False, tokentype_embeddings=None,
                  config=ns(fp32_residual_connection=False, sequence_parallel=False), embedding_dropout=lambda x: x)
        d = make_dict(feats)
        emb = fwd(self, torch.zeros(b, s, dtype=torch.long), None, None, d)
        assert emb.shape == (s, b, h) and emb.dtype == torch.bfloat16
        emb.backward(do)
        out[f"{tag}.we"], out[f"{tag}.features"], out[f"{tag}.dout"] = map(_bf16_bits, (we.detach(), feats.detach(), do))
        out[f"{tag}.out"] = _bf16_bits(emb.detach())
        out[f"{tag}.dwe"] = _bf16_bits(we.grad)
        out[f"{tag}.dfeatures"] = _bf16_bits(feats.grad)
        for key, val in d.items():
            if key == "features":
                continue
            if key == "pre_len":
                out[f"{tag}.pre_len"] = np.array(val, dtype=np.int64)
            elif isinstance(val, (tuple, list)):
                out[f"{tag}.{key}"] = np.stack([x.numpy() for x in val])
            else:
                out[f"{tag}.{key}"] = val.numpy()

    rng = np.random.default_rng(11)
    starts = [3, 20, 33]
    ind_s = torch.from_numpy(np.stack([np.arange(st, st + n_tok) for st in starts]).astype(np.int64))
    ind_b = torch.tensor([0, 1, 1])[:, None].expand(n_img, n_tok).contiguous()
    run("indices", lambda f: {"features": f, "indices": torch.stack([ind_b, ind_s])})
    run("pre_len", lambda f: {"features": f[:b], "pre_len": 5})
    # src/tgt: a CP rank that owns part of image 0, none of image 1, all of image 2 (rows in scrambled order)
    src_b = torch.tensor([0] * 5 + [2] * n_tok)
    src_s = torch.tensor([3, 4, 5, 6, 7] + list(range(n_tok)))
    tgt_b = torch.tensor([0] * 5 + [1] * n_tok)
    tgt_s = torch.tensor([0, 1, 2, 3, 4] + list(range(30, 30 + n_tok)))
    perm = torch.from_numpy(rng.permutation(src_b.numel()))
    run("src_tgt", lambda f: {"features": f, "src_indices": (src_b[perm], src_s[perm]),
                              "tgt_indices": (tgt_b[perm], tgt_s[perm])})
    run("none", lambda f: {"features": f})
    np.savez_compressed(os.path.join(HERE, "embedding_scatter.npz"), **out)
    print("embedding_scatter.npz:", len(out), "arrays")


def gen_indexing():
    out = {}
    tu = importlib.import_module("lcvlm_modellink.training.utils")

    # index_of_a_in_b ------------------------------------------------------
    rng = np.random.default_rng(7)
    b = rng.permutation(200)[:64].astype(np.int64)
    a = rng.permutation(b)[:40]
    out["ioab.a"], out["ioab.b"] = a, b
    out["ioab.out"] = tu.index_of_a_in_b(torch.from_numpy(a), torch.from_numpy(b)).numpy()

    # zig-zag slicing + image remap for several (S, CP) --------------------
    for S, cp, n_img, n_tok in [(64, 2, 3, 4), (256, 4, 5, 16), (1024, 8, 3, 256)]:
        tokens = torch.arange(2 * S, dtype=torch.int64).view(2, S) * 3 + 1
        pos = torch.arange(S, dtype=torch.int64).repeat(2, 1)
        # images: disjoint runs of n_tok consecutive positions, batch row alternating
        starts = np.sort(rng.choice(np.arange(0, S - n_tok, n_tok + 1), size=n_img, replace=False))
        ind_s = torch.from_numpy(np.stack([np.arange(s, s + n_tok) for s in starts]).astype(np.int64))
        ind_b = torch.from_numpy((np.arange(n_img) % 2)[:, None].repeat(n_tok, 1).astype(np.int64))
        ext = torch.stack([ind_b, ind_s])
        images = torch.arange(n_img * 6, dtype=torch.float32).view(n_img, 6)
        tag = f"zz.S{S}.cp{cp}"
        out[f"{tag}.tokens"] = tokens.numpy()
        out[f"{tag}.external_indices"] = ext.numpy()
        out[f"{tag}.images"] = images.numpy()
        for r in range(cp):
            tu.get_args = lambda: ns(seq_length=S, vision_context_parallel=False)
            tu.mpu = ns(get_context_parallel_rank=lambda r=r: r, get_context_parallel_world_size=lambda: cp)
            batch = {"tokens": tokens.clone(), "position_ids": pos.clone(), "attention_mask": None,
                     "external_images": images.clone(), "external_indices": ext.clone()}
            with no_cuda_device():
                res = tu._get_batch_on_this_cp_rank_in_megatron_cp(batch)
            out[f"{tag}.r{r}.tokens"] = res["tokens"].numpy()
            out[f"{tag}.r{r}.position_ids"] = res["position_ids"].numpy()
            out[f"{tag}.r{r}.images"] = res["external_images"].numpy()
            if "external_src_indices" in res:
                out[f"{tag}.r{r}.src"] = res["external_src_indices"].numpy()
                out[f"{tag}.r{r}.tgt"] = res["external_tgt_indices"].numpy()

    # collate_fn -----------------------------------------------------------
    ds = importlib.import_module("lcvlm_modellink.legacy.data.data_samplers")
    S = 32
    rows = []
    asl = [[5, 20, 32], [11, 12, 32], [7, 30, 32]]   # default_collate stacks them: equal counts
    imgs = [np.stack([np.zeros((2, 4), np.int64), np.array([[1, 2, 3, 4], [10, 11, 12, 13]])]), None,
            np.stack([np.zeros((1, 4), np.int64), np.array([[20, 21, 22, 23]])])]
    for i in range(3):
        d = {"tokens": torch.arange(S) + 100 * i, "actual_seq_len": torch.tensor(asl[i], dtype=torch.int64)}
        if imgs[i] is not None:
            d["image_indices"] = torch.from_numpy(imgs[i].copy())
            d["images"] = torch.full((imgs[i].shape[1], 3), float(i))
        rows.append(d)
    col = ds.collate_fn(rows)
    out["collate.in_actual0"] = np.asarray(asl[0])
    out["collate.in_actual1"] = np.asarray(asl[1])
    out["collate.in_actual2"] = np.asarray(asl[2])
    out["collate.in_img0"] = imgs[0]
    out["collate.in_img2"] = imgs[2]
    out["collate.actual_seq_len"] = col["actual_seq_len"].numpy()
    out["collate.image_indices"] = col["image_indices"].numpy()
    out["collate.images"] = col["images"].numpy()

    # pixel_shuffle: take the method from the class body without building the model
    pl = importlib.import_module("lcvlm_modellink.pretrain_lcvlm")
    ps = pl.MegatronVisionModel.pixel_shuffle
    x = torch.arange(2 * 8 * 8 * 16, dtype=torch.float32).view(2, 8, 8, 16)
    out["ps.small.in"] = x.numpy()
    out["ps.small.out"] = ps(None, x, scale_factor=0.5).numpy()
    # full-size permutation as an index map: value = flat source index (int32 exact in fp32? no -> use int64)
    xi = torch.arange(32 * 32 * 1024, dtype=torch.int64).view(1, 32, 32, 1024)
    out["ps.full.index_map"] = ps(None, xi, scale_factor=0.5).numpy().astype(np.int32)

    # RoPE zig-zag slice ------------------------------------------------------
    rope = importlib.import_module("lcvlm_modellink.core.models.common.embeddings.rotary_pos_embedding")
    emb = torch.arange(64 * 8, dtype=torch.float32).view(64, 1, 1, 8)
    real_tensor = torch.tensor

    class _T(torch.Tensor):
        pass

    def tensor_nopin(*a, **k):
        k.pop("pin_memory", None)
        k.pop("device", None)
        t = real_tensor(*a, **k)
        return t
    for cp in (2, 4):
        for r in range(cp):
            rope.parallel_state = ns(get_context_parallel_world_size=lambda cp=cp: cp,
                                     get_context_parallel_rank=lambda r=r: r)
            torch.tensor = tensor_nopin
            torch.Tensor.cuda_saved = getattr(torch.Tensor, "cuda")
            torch.Tensor.cuda = lambda self, *a, **k: self
            try:
                o = rope.get_pos_emb_on_this_cp_rank(emb, 0)
            finally:
                torch.tensor = real_tensor
                torch.Tensor.cuda = torch.Tensor.cuda_saved
            out[f"rope.cp{cp}.r{r}"] = o.numpy()
    out["rope.in"] = emb.numpy()

    np.savez_compressed(os.path.join(HERE, "indexing.npz"), **out)
    print("indexing.npz:", len(out), "arrays")


def gen_pixel_shuffle_v2():
    """MegatronVisionModel.pixel_shuffle_v2 (lcvlm_modellink/pretrain_lcvlm.py:558-570), taken from the class body without
    building the model: even and odd grids (the odd ones are zero-padded), strides 2 / 3 / 4, and the full-size grid as an
    index map (value = flat source element + 1, so that the padding's zeros stay distinguishable)."""
    pl = importlib.import_module("lcvlm_modellink.pretrain_lcvlm")
    ps2 = pl.MegatronVisionModel.pixel_shuffle_v2
    out = {}
    for tag, shape, st in [("even", (2, 8, 8, 16), 2), ("odd", (2, 5, 5, 8), 2), ("s4", (1, 8, 8, 8), 4), ("s3", (1, 7, 7, 8), 3)]:
        x = torch.arange(1, 1 + int(np.prod(shape)), dtype=torch.float32).view(*shape)
        out[f"{tag}.in"] = x.numpy()
        out[f"{tag}.stride"] = np.asarray(st)
        out[f"{tag}.out"] = ps2(None, x, scale_stride=st).numpy()
    xi = torch.arange(1, 1 + 32 * 32 * 1024, dtype=torch.int64).view(1, 32, 32, 1024)
    out["full.index_map"] = ps2(None, xi, scale_stride=2).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "pixel_shuffle_v2.npz"), **out)
    print("pixel_shuffle_v2.npz:", len(out), "arrays")


def gen_packing():
    """Qwen2Dataset.add_ret / process_ret (cognitron_vl/data/dataset_qwen2.py:132-254) run
    unbound on a namespace standing in for the dataset object.  The module is loaded by
    file path under its own dotted name so that its relative imports hit the shim."""
    import importlib.util
    path = os.path.join(REF, "cognitron_vl", "data", "dataset_qwen2.py")
    spec = importlib.util.spec_from_file_location("cognitron_vl.data.dataset_qwen2", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DS = mod.Qwen2Dataset
    out = {}

    def run(case, samples, max_len, reset, shift_token=False):
        self = ns(ret={"src": {"tokens": [], "labels": [], "actual_seq_len": [], "position_ids": []}},
                  create_attention_mask=False, create_attention_mask_2d=False, create_position_ids=True,
                  create_loss_mask=True, reset_position_ids=reset, reset_attention_mask=reset, shift_token=shift_token,
                  variable_length=False, max_len=max_len, tokenizer=ns(pad_token_id=0))
        for i, (L, img) in enumerate(samples):
            ret = {"input_ids": list(range(1000 * i, 1000 * i + L)), "labels": list(range(L)), "images": []}
            if img is not None:
                pos = torch.tensor(img, dtype=torch.int64)
                ret["image_indices"] = torch.stack([torch.zeros_like(pos), pos], dim=0)
                ret["images"] = torch.zeros(pos.shape[0], 3)
            DS.add_ret(self, ret, "src")
        to_ret = DS.process_ret(self, self.ret["src"])
        out[f"{case}.lengths"] = np.asarray([L for L, _ in samples])
        for k in ("tokens", "labels", "position_ids", "actual_seq_len", "loss_mask", "image_indices"):
            if k in to_ret:
                out[f"{case}.{k}"] = to_ret[k].numpy() if hasattr(to_ret[k], "numpy") else np.asarray(to_ret[k])
    img_a = [[2, 3, 4, 5], [8, 9, 10, 11]]
    img_b = [[1, 2, 3, 4]]
    run("pad", [(15, img_a), (9, None), (6, img_b)], 40, True)
    run("trunc", [(15, None), (9, img_b), (20, None)], 32, True)
    run("late_image", [(10, None), (12, img_b), (8, img_a)], 32, True)
    run("shift", [(15, img_a), (9, None)], 32, True, shift_token=True)
    run("noreset", [(15, img_a), (9, None)], 32, False)
    np.savez_compressed(os.path.join(HERE, "packing.npz"), **out)
    print("packing.npz:", len(out), "arrays")


# --------------------------------------------------------------------------
def gen_rope():
    """RotaryEmbedding.forward (angle table, packed position-id gather) and the eager appliers
    apply_rotary_pos_emb_bshd / _thd of rotary_pos_embedding.py, on bf16 inputs as the model runs them."""
    rope = importlib.import_module("lcvlm_modellink.core.models.common.embeddings.rotary_pos_embedding")
    out = {}
    saved_dev = torch.cuda.current_device
    torch.cuda.current_device = lambda: "cpu"
    try:
        g = torch.Generator().manual_seed(77)
        # ---- angle tables: Qwen2.5 shape (d=128, base 1e6) and a small one
        for tag, d, base, n in (("qwen", 128, 1000000, 4096), ("small", 16, 10000, 40)):
            re = rope.RotaryEmbedding(d, 1.0, rotary_interleaved=False, rotary_base=base)
            rope.parallel_state = ns(get_context_parallel_world_size=lambda: 1, get_context_parallel_rank=lambda: 0)
            rope.get_position_ids = lambda: None
            emb = re.forward(n)
            out[f"table.{tag}.inv_freq"] = re.inv_freq.numpy()
            out[f"table.{tag}.emb"] = emb.numpy() if n <= 64 else emb[:: n // 32].numpy()  # a row sample of the big one
            out[f"table.{tag}.meta"] = np.array([d, base, n], dtype=np.int64)
        # packed sequences: position ids restart per sample ([s, b] as mindspeed hands them over)
        re = rope.RotaryEmbedding(16, 1.0, rotary_interleaved=False, rotary_base=10000)
        pos = torch.tensor([[0, 1, 2, 3, 4, 0, 1, 2, 0, 1, 2, 3, 4, 5, 6, 7]], dtype=torch.int64).t().contiguous()  # [16, 1]
        rope.get_position_ids = lambda: pos
        out["table.packed.pos"] = pos.numpy()
        out["table.packed.emb"] = re.forward(16).numpy()
        rope.get_position_ids = lambda: None

        # ---- appliers on bf16 tensors
        for tag, s, b, h, d, base in (("a", 24, 2, 3, 16, 10000), ("b", 96, 1, 5, 128, 1000000), ("c", 33, 1, 2, 64, 10000)):
            re = rope.RotaryEmbedding(d, 1.0, rotary_interleaved=False, rotary_base=base)
            freqs = re.forward(s)  # [s, 1, 1, d] fp32
            t = (torch.randn(s, b, h, d, generator=g) * 1.5).to(torch.bfloat16).requires_grad_(True)
            o = rope.apply_rotary_pos_emb_bshd(t, freqs, rotary_interleaved=False)
            do = torch.randn(s, b, h, d, generator=g).to(torch.bfloat16)
            o.backward(do)
            out[f"bshd.{tag}.t"] = t.detach().float().numpy()
            out[f"bshd.{tag}.freqs"] = freqs.numpy()
            out[f"bshd.{tag}.out"] = o.detach().float().numpy()
            out[f"bshd.{tag}.dout"] = do.float().numpy()
            out[f"bshd.{tag}.dt"] = t.grad.float().numpy()  # torch autograd through the reference's eager ops
            assert o.dtype == torch.bfloat16 and t.grad.dtype == torch.bfloat16
        # partial rotary dimension (rot_dim < d passes the tail through)
        re = rope.RotaryEmbedding(32, 0.5, rotary_interleaved=False, rotary_base=10000)
        freqs = re.forward(12)
        t = torch.randn(12, 1, 2, 32, generator=g).to(torch.bfloat16)
        out["bshd.partial.t"] = t.float().numpy()
        out["bshd.partial.freqs"] = freqs.numpy()
        out["bshd.partial.out"] = rope.apply_rotary_pos_emb_bshd(t, freqs).float().numpy()
        # thd (packed): positions restart at every cu_seqlens boundary
        re = rope.RotaryEmbedding(64, 1.0, rotary_interleaved=False, rotary_base=1000000)
        freqs = re.forward(40)
        cu = torch.tensor([0, 7, 7 + 19, 7 + 19 + 1, 40], dtype=torch.int32)
        t = torch.randn(40, 4, 64, generator=g).to(torch.bfloat16)
        out["thd.t"] = t.float().numpy()
        out["thd.cu"] = cu.numpy()
        out["thd.freqs"] = freqs.numpy()
        out["thd.out"] = rope.apply_rotary_pos_emb_thd(t, cu, freqs).float().numpy()
    finally:
        torch.cuda.current_device = saved_dev
    np.savez_compressed(os.path.join(HERE, "rope.npz"), **out)
    print("rope.npz:", len(out), "arrays")


def gen_vit():
    """The reference's own layer and model ``forward`` (intern_vit_model.py:32-89, :190-261), called unbound on
    namespaces whose submodules (LayerNorm, SelfAttention, MLP: Megatron classes that are not installed) are the
    oracle's restatements.  What this pins is the wiring those two functions do: the order of the norms, where
    the two biases are added, LayerScale and residuals, the patch projection / class token / position embedding
    and the [b,s,h] <-> [s,b,h] permutes."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import vit_ref
    ivm = importlib.import_module("lcvlm_modellink.core.models.vision.intern_vit_model")
    ivm.make_viewless_tensor = lambda inp, requires_grad, keep_graph: inp
    ivm.mpu = ns(get_context_parallel_world_size=lambda: 1, get_context_parallel_rank=lambda: 0)
    g = torch.Generator().manual_seed(4242)
    h, heads, ffn, n_layers, patch, img_hw, n, eps = 128, 2, 160, 2, 14, 42, 2, 1e-6  # head dim 64: a size the device kernels take
    seq = (img_hw // patch) ** 2 + 1

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()

    def make_sd(seq):
        sd = {"conv1.weight": rnd(h, 3, patch, patch, scale=0.05), "conv1.bias": rnd(h, scale=0.1),
              "class_token": rnd(1, 1, h), "position_embeddings.weight": rnd(seq, h, scale=0.2)}
        for i in range(n_layers):
            pre = f"decoder.layers.{i}."
            sd[pre + "input_layernorm.weight"] = 1 + rnd(h, scale=0.1)
            sd[pre + "input_layernorm.bias"] = rnd(h, scale=0.1)
            sd[pre + "self_attention.linear_qkv.weight"] = rnd(3 * h, h, scale=0.15)
            sd[pre + "self_attention.linear_qkv.bias"] = rnd(3 * h, scale=0.1)
            sd[pre + "self_attention.linear_proj.weight"] = rnd(h, h, scale=0.15)
            sd[pre + "self_attention.linear_proj.bias"] = rnd(h, scale=0.1)
            sd[pre + "pre_mlp_layernorm.weight"] = 1 + rnd(h, scale=0.1)
            sd[pre + "pre_mlp_layernorm.bias"] = rnd(h, scale=0.1)
            sd[pre + "mlp.linear_fc1.weight"] = rnd(ffn, h, scale=0.15)
            sd[pre + "mlp.linear_fc1.bias"] = rnd(ffn, scale=0.1)
            sd[pre + "mlp.linear_fc2.weight"] = rnd(h, ffn, scale=0.15)
            sd[pre + "mlp.linear_fc2.bias"] = rnd(h, scale=0.1)
            sd[pre + "ls1"] = rnd(h, scale=0.5)   # larger than the 0.01 initial value so the branches matter
            sd[pre + "ls2"] = rnd(h, scale=0.5)
        return sd
    sd = make_sd(seq)

    def layer_ns(i):
        pre = f"decoder.layers.{i}."
        p = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        sub = lambda name: {k[len(name) + 1:]: v for k, v in p.items() if k.startswith(name + ".")}
        return ns(input_layernorm=lambda x: vit_ref.layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps),
                  pre_mlp_layernorm=lambda x: vit_ref.layer_norm(x, p["pre_mlp_layernorm.weight"], p["pre_mlp_layernorm.bias"], eps),
                  self_attention=lambda x, **kw: vit_ref.self_attention(x, sub("self_attention"), heads),
                  mlp=lambda x: vit_ref.mlp(x, sub("mlp")), ls1=p["ls1"], ls2=p["ls2"])

    layer_fwd = ivm.InternViTTransformerLayer.forward
    out = {"meta": np.array([h, heads, ffn, n_layers, patch, img_hw, n], dtype=np.int64), "eps": np.array(eps)}
    for k, v in sd.items():
        out["sd." + k] = v.numpy()
    x = rnd(seq, n, h)
    out["layer0.in"] = x.numpy()
    y, _ = layer_fwd(layer_ns(0), x, None)
    out["layer0.out"] = y.numpy()

    def vit_self(sd, seq):
        conv = torch.nn.Conv2d(3, h, kernel_size=patch, stride=patch, bias=True)
        conv.weight.data.copy_(sd["conv1.weight"]); conv.bias.data.copy_(sd["conv1.bias"])
        pos = torch.nn.Embedding(seq, h); pos.weight.data.copy_(sd["position_embeddings.weight"])

        def layer_of(i):
            pre = f"decoder.layers.{i}."
            p = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
            sub = lambda name: {k[len(name) + 1:]: v for k, v in p.items() if k.startswith(name + ".")}
            return ns(input_layernorm=lambda x: vit_ref.layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps),
                      pre_mlp_layernorm=lambda x: vit_ref.layer_norm(x, p["pre_mlp_layernorm.weight"], p["pre_mlp_layernorm.bias"], eps),
                      self_attention=lambda x, **kw: vit_ref.self_attention(x, sub("self_attention"), heads),
                      mlp=lambda x: vit_ref.mlp(x, sub("mlp")), ls1=p["ls1"], ls2=p["ls2"])

        def decoder(x, mask):
            for i in range(n_layers):
                x, _ = layer_fwd(layer_of(i), x, mask)
            return x
        return ns(conv1=conv, add_class_token=True, class_token=sd["class_token"], seq_length=seq, position_embeddings=pos,
                  position_ids=torch.arange(seq).expand(1, -1), vision_context_parallel=False, decoder=decoder)

    img = rnd(n, 3, img_hw, img_hw)
    out["model.img"] = img.numpy()
    with torch.no_grad():
        out["model.out"] = ivm.InternViTModel.forward(vit_self(sd, seq), img).numpy()

    # ---- the vision model around the tower: MegatronVisionModel.forward -> forward_chunk -> forward_once
    # (class-token drop, pixel_shuffle 0.5) -> forward_projection -> MultimodalProjector.forward
    # (pretrain_lcvlm.py:439-556, multimodal_projector.py:53-80), all run from the reference's class bodies.
    pl = importlib.import_module("lcvlm_modellink.pretrain_lcvlm")
    mp = importlib.import_module("lcvlm_modellink.core.models.vision.multimodal_projector")
    pl.mpu = ns(get_context_parallel_world_size=lambda: 1, get_context_parallel_rank=lambda: 0)
    VM = pl.MegatronVisionModel
    vm_img_hw, vm_n, llm_h = 56, 3, 192
    vm_seq = (vm_img_hw // patch) ** 2 + 1
    vsd = make_sd(vm_seq)
    vm_sd = {"vit." + k: v for k, v in vsd.items()}
    vm_sd["pre_proj_layernorm.weight"] = 1 + rnd(4 * h, scale=0.1)
    vm_sd["pre_proj_layernorm.bias"] = rnd(4 * h, scale=0.1)
    vm_sd["vision_projection.encoder.linear_fc1.weight"] = rnd(h, 4 * h, scale=0.08)
    vm_sd["vision_projection.encoder.linear_fc2.weight"] = rnd(llm_h, h, scale=0.15)
    ln = torch.nn.LayerNorm(4 * h)
    ln.weight.data.copy_(vm_sd["pre_proj_layernorm.weight"]); ln.bias.data.copy_(vm_sd["pre_proj_layernorm.bias"])
    vit_ns = vit_self(vsd, vm_seq)
    proj_ns = ns(encoder=lambda x: vit_ref.projector_mlp(x, vm_sd))
    vm = ns(vit=lambda images, mask: ivm.InternViTModel.forward(vit_ns, images, mask), add_class_token=True,
            vision_model_freeze=True, vision_projector_freeze=False, vision_projector_recompute=False,
            vision_downsample_ratio=0.5, vision_downsample_stride=1, vision_context_parallel=False,
            pre_proj_layernorm=ln, vision_projection=lambda x: mp.MultimodalProjector.forward(proj_ns, x))
    vm.pixel_shuffle = lambda x, scale_factor=0.5: VM.pixel_shuffle(vm, x, scale_factor=scale_factor)
    vm.forward_projection = lambda x: VM.forward_projection(vm, x)
    vm.forward_once = lambda images, mask: VM.forward_once(vm, images, mask)
    vm.forward_chunk = lambda images, mask: VM.forward_chunk(vm, images, mask)
    vimg = rnd(vm_n, 3, vm_img_hw, vm_img_hw)
    with torch.no_grad():
        vout = VM.forward(vm, images=vimg)
    assert vout.shape == (vm_n, (vm_seq - 1) // 4, llm_h), vout.shape
    out["vm.meta"] = np.array([vm_img_hw, vm_n, llm_h], dtype=np.int64)
    out["vm.img"] = vimg.numpy()
    out["vm.out"] = vout.numpy()
    for k, v in vm_sd.items():
        out["vmsd." + k] = v.numpy()
    # gradients of the projector part alone (it trains in every stage): autograd through the reference's
    # forward_projection on a given tower output
    px = rnd(2, 4, 4 * h).requires_grad_(True)
    for prm in ln.parameters():
        prm.grad = None
    w1 = vm_sd["vision_projection.encoder.linear_fc1.weight"].clone().requires_grad_(True)
    w2 = vm_sd["vision_projection.encoder.linear_fc2.weight"].clone().requires_grad_(True)
    gsd = dict(vm_sd); gsd["vision_projection.encoder.linear_fc1.weight"] = w1; gsd["vision_projection.encoder.linear_fc2.weight"] = w2
    proj_ns_g = ns(encoder=lambda x: vit_ref.projector_mlp(x, gsd))
    vm_g = ns(pre_proj_layernorm=ln, vision_projection=lambda x: mp.MultimodalProjector.forward(proj_ns_g, x))
    po = VM.forward_projection(vm_g, px)
    pdo = rnd(*po.shape)
    po.backward(pdo)
    out["proj.x"], out["proj.dout"], out["proj.out"] = px.detach().numpy(), pdo.numpy(), po.detach().numpy()
    out["proj.dx"], out["proj.dw1"], out["proj.dw2"] = px.grad.numpy(), w1.grad.numpy(), w2.grad.numpy()
    out["proj.dln_w"], out["proj.dln_b"] = ln.weight.grad.numpy(), ln.bias.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "vit.npz"), **out)
    print("vit.npz:", len(out), "arrays")


if __name__ == "__main__":
    install_shim()
    which = sys.argv[1:] or ["eager", "eager_grad", "eager_band", "embedding", "indexing", "pixel_shuffle_v2", "packing", "rope", "vit"]
    if "eager" in which:
        gen_eager_attention()
    if "eager_grad" in which:
        gen_eager_attention_grad()
    if "eager_band" in which:
        gen_eager_attention_band()
    if "embedding" in which:
        gen_embedding_scatter()
    if "indexing" in which:
        gen_indexing()
    if "pixel_shuffle_v2" in which:
        gen_pixel_shuffle_v2()
    if "packing" in which:
        gen_packing()
    if "rope" in which:
        gen_rope()
    if "vit" in which:
        gen_vit()
