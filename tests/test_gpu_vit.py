"""InternViT forward on the device (HIP kernels through the C ABI + library GEMMs) against the oracle and the
fixture made by the reference's own layer / model forward.  Tolerance: max error <= 2e-2 of the output's
largest magnitude (bf16 activations on the device, fp32 in the oracle)."""
import os

import numpy as np
import pytest
import torch

from oracle import vit_ref

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _close(got: torch.Tensor, want: torch.Tensor, tol=TOL):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= tol * want.abs().max().item(), f"max err {err} vs scale {want.abs().max().item()}"


def _bf(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("rows,h", [(1, 8), (77, 1024), (513, 4096), (5, 200), (3, 8192)])
def test_layernorm(rows, h):
    from long_vita_amd import ops
    x, w, b = _bf(rows, h, seed=1, scale=3.0) + 0.5, 1 + _bf(h, seed=2, scale=0.2), _bf(h, seed=3, scale=0.2)
    y = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    want = vit_ref.layer_norm(x, w, b, 1e-6)
    # one bf16 rounding of the result: half an ulp relative (2^-9) on each element
    assert torch.all((y.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6)


def test_layernorm_strided_rows_and_constant_row():
    from long_vita_amd import ops
    buf = _bf(9, 3 * 256, seed=4).cuda()
    buf[4, 256:512] = 2.5     # zero variance: output is exactly beta
    w, b = 1 + _bf(256, seed=5, scale=0.2), _bf(256, seed=6, scale=0.2)
    y = ops.layernorm(buf[:, 256:512], w.cuda(), b.cuda(), 1e-5)
    want = vit_ref.layer_norm(buf[:, 256:512].cpu(), w, b, 1e-5)
    _close(y, want, 2.0 ** -7)
    assert torch.equal(y[4].cpu(), b)


def test_bias_gelu_and_bias_scale_residual():
    from long_vita_amd import ops
    rows, h = 131, 4096
    x, bias = _bf(rows, h, seed=7, scale=2.0), _bf(h, seed=8)
    y = ops.bias_gelu(x.cuda(), bias.cuda())
    want = vit_ref.gelu(x.float() + bias.float())
    assert torch.all((y.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6)
    ls, res = _bf(h, seed=9, scale=0.1), _bf(rows, h, seed=10)
    out = ops.bias_scale_residual(x.cuda(), bias.cuda(), ls.cuda(), res.cuda())
    want = res.float() + (x.float() + bias.float()) * ls.float()
    assert torch.all((out.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6)
    out = ops.bias_scale_residual(x.cuda(), None, ls.cuda(), res.cuda())
    want = res.float() + x.float() * ls.float()
    assert torch.all((out.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6)


def test_patchify_is_the_conv_unfold_bit_exact():
    from long_vita_amd import ops
    n, c, hw, p = 3, 3, 56, 14
    img = _bf(n, c, hw, hw, seed=11)
    kpad = 592
    got = ops.patchify(img.cuda(), p, kpad).cpu()
    want = torch.nn.functional.unfold(img.float(), kernel_size=p, stride=p).permute(0, 2, 1).reshape(-1, c * p * p)
    assert torch.equal(got[:, : c * p * p].float(), want)
    assert torch.count_nonzero(got[:, c * p * p:]) == 0


def test_embed_tokens():
    from long_vita_amd import ops
    n, npatch, h = 3, 16, 256
    proj, cb, cls, pos = _bf(n, npatch, h, seed=12), _bf(h, seed=13), _bf(1, 1, h, seed=14), _bf(npatch + 1, h, seed=15)
    got = ops.embed_tokens(proj.cuda(), cb.cuda(), cls.cuda(), pos.cuda())
    want = torch.cat([cls.float().expand(n, -1, -1), proj.float() + cb.float()], dim=1) + pos.float()[None]
    want = want.permute(1, 0, 2)
    assert got.shape == (npatch + 1, n, h)
    assert torch.all((got.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6)


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "vit.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    return g, sd


def _model(cfg_kwargs, sd):
    from long_vita_amd import intern_vit as iv
    m = iv.InternViTModel(iv.InternViTConfig(**cfg_kwargs))
    m.load_state_dict(sd, strict=True)
    return m.to(torch.bfloat16).cuda()


def test_layer_and_model_against_reference_fixture(golden_dir):
    from long_vita_amd import ops
    g, sd = _golden(golden_dir)
    h, heads, ffn, n_layers, patch, img_hw, n = (int(v) for v in g["meta"])
    m = _model(dict(hidden_size=h, ffn_hidden_size=ffn, num_attention_heads=heads, num_layers=n_layers, patch_dim=patch,
                    img_h=img_hw, img_w=img_hw, layernorm_epsilon=float(g["eps"])), sd)
    x = torch.from_numpy(g["layer0.in"]).to(torch.bfloat16).cuda()
    with torch.no_grad():
        y = m.decoder.layers[0](x)
    _close(y, torch.from_numpy(g["layer0.out"]))
    out = m(torch.from_numpy(g["model.img"]).to(torch.bfloat16).cuda())
    _close(out, torch.from_numpy(g["model.out"]))


def test_model_at_internvit_width_against_oracle():
    """hidden 1024 / 16 heads / ffn 4096 as shipped, 3 layers and 112-pixel tiles to keep the oracle quick."""
    from long_vita_amd import intern_vit as iv
    cfg = iv.InternViTConfig(num_layers=3, img_h=112, img_w=112)
    torch.manual_seed(5)
    m = iv.InternViTModel(cfg)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("ls1") or name.endswith("ls2"):
                p.copy_(torch.randn_like(p) * 0.3)
            elif "layernorm" in name and name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn_like(p))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
        m = m.to(torch.bfloat16)
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    img = _bf(5, 3, 112, 112, seed=21)
    want = vit_ref.model_forward(img.float(), sd, cfg.num_attention_heads, cfg.patch_dim, cfg.layernorm_epsilon)
    got = m.cuda()(img.cuda())
    assert got.shape == (5, 65, 1024)
    _close(got, want)


def test_rejects_what_it_does_not_build():
    from long_vita_amd import intern_vit as iv
    with pytest.raises(NotImplementedError):
        iv.InternViTModel(iv.InternViTConfig(num_layers=1), vision_context_parallel=True)


@pytest.mark.parametrize("rows,h", [(1, 8), (77, 1024), (2051, 4096), (5, 200)])
def test_layernorm_backward(rows, h):
    from long_vita_amd import ops
    x, w, b = _bf(rows, h, seed=31, scale=2.0) + 0.3, 1 + _bf(h, seed=32, scale=0.2), _bf(h, seed=33, scale=0.2)
    dy = _bf(rows, h, seed=34)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    vit_ref.layer_norm(xr, wr, br, 1e-5).backward(dy.float())
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    ops.layernorm(xg, wg, bg, 1e-5).backward(dy.cuda())
    _close(xg.grad, xr.grad, 2.0 ** -7)
    _close(wg.grad, wr.grad, 2.0 ** -7)
    _close(bg.grad, br.grad, 2.0 ** -7)
    # parameter gradients are summed in a fixed order: two runs agree bit for bit
    dx2, dg2, db2 = ops.layernorm_bwd_launch(x.cuda(), dy.cuda(), w.cuda(), 1e-5)
    dx3, dg3, db3 = ops.layernorm_bwd_launch(x.cuda(), dy.cuda(), w.cuda(), 1e-5)
    assert torch.equal(dg2, dg3) and torch.equal(db2, db3) and torch.equal(dx2, dx3)


def test_layernorm_backward_rejects_wide_rows():
    from long_vita_amd import ops
    x = torch.zeros(4, 8192, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ops.LvAttnError):
        ops.layernorm_bwd_launch(x, x, torch.ones(8192, dtype=torch.bfloat16, device="cuda"), 1e-5)


def test_gelu_backward():
    from long_vita_amd import ops
    rows, h = 67, 1024
    x, bias, dy = _bf(rows, h, seed=41, scale=2.0), _bf(h, seed=42), _bf(rows, h, seed=43)
    for use_bias in (True, False):
        xr = x.float().requires_grad_(True)
        br = bias.float().requires_grad_(True)
        torch.nn.functional.gelu(xr + (br if use_bias else 0.0)).backward(dy.float())
        xg, bg = x.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
        ops.bias_gelu(xg, bg if use_bias else None).backward(dy.cuda())
        _close(xg.grad, xr.grad, 2.0 ** -7)
        if use_bias:
            _close(bg.grad, br.grad, 2.0 ** -6)


def test_vision_model_against_reference_fixture(golden_dir):
    from long_vita_amd import intern_vit as iv, vision_model as vmod
    g, _ = _golden(golden_dir)
    h, heads, ffn, n_layers, patch, _, _ = (int(v) for v in g["meta"])
    img_hw, n, llm_h = (int(v) for v in g["vm.meta"])
    cfg = iv.InternViTConfig(hidden_size=h, ffn_hidden_size=ffn, num_attention_heads=heads, num_layers=n_layers, patch_dim=patch,
                             img_h=img_hw, img_w=img_hw, layernorm_epsilon=float(g["eps"]))
    m = vmod.MegatronVisionModel(hidden_size=llm_h, vit_config=cfg)
    m.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("vmsd.")}, strict=True)
    m = m.to(torch.bfloat16).cuda()
    out = m(images=torch.from_numpy(g["vm.img"]).to(torch.bfloat16).cuda())
    assert out.requires_grad          # the projector trains; the tower ran without a graph
    _close(out, torch.from_numpy(g["vm.out"]))

    # projector gradients against the reference's forward_projection under autograd
    x = torch.from_numpy(g["proj.x"]).to(torch.bfloat16).cuda().requires_grad_(True)
    po = m.forward_projection(x)
    _close(po, torch.from_numpy(g["proj.out"]))
    po.backward(torch.from_numpy(g["proj.dout"]).to(torch.bfloat16).cuda())
    enc = m.vision_projection.encoder
    for got, key in ((x.grad, "proj.dx"), (enc.linear_fc1.weight.grad, "proj.dw1"), (enc.linear_fc2.weight.grad, "proj.dw2"),
                     (m.pre_proj_layernorm.weight.grad, "proj.dln_w"), (m.pre_proj_layernorm.bias.grad, "proj.dln_b")):
        _close(got, torch.from_numpy(g[key]))


def test_vision_model_chunks_of_256_images():
    """forward_chunk splits at 256 images; a 258-image batch equals the two passes run separately."""
    from long_vita_amd import intern_vit as iv, vision_model as vmod
    cfg = iv.InternViTConfig(hidden_size=128, ffn_hidden_size=256, num_attention_heads=2, num_layers=1, img_h=28, img_w=28)
    torch.manual_seed(3)
    m = vmod.MegatronVisionModel(hidden_size=64, vit_config=cfg, vision_projector_freeze=True).to(torch.bfloat16).cuda()
    img = _bf(258, 3, 28, 28, seed=51).cuda()
    whole = m(images=img)
    assert whole.shape == (258, 1, 64) and not whole.requires_grad
    assert torch.equal(whole[:256], m.forward_once(img[:256])) and torch.equal(whole[256:], m.forward_once(img[256:]))


def test_vision_model_with_downsample_stride():
    """--vision-downsample-stride 2 instead of the ratio form (pretrain_lcvlm.py:418-420, 472-476): an odd 5 x 5 patch grid
    (70-pixel images) is zero-padded to 6 x 6, nine tokens of 4 x hidden channels go to the projector; the model's output
    equals tower -> oracle's pixel_shuffle_v2 on the bit patterns -> projector."""
    from long_vita_amd import intern_vit as iv, vision_model as vmod
    from oracle import indexing_ref as I
    cfg = iv.InternViTConfig(hidden_size=128, ffn_hidden_size=256, num_attention_heads=2, num_layers=1, img_h=70, img_w=70)
    torch.manual_seed(4)
    m = vmod.MegatronVisionModel(hidden_size=64, vit_config=cfg, vision_projector_freeze=True, vision_downsample_ratio=1,
                                 vision_downsample_stride=2).to(torch.bfloat16).cuda()
    assert m.image_token_length == 9 and m.vision_projection.encoder.linear_fc1.weight.shape[1] == 4 * 128
    img = _bf(3, 3, 70, 70, seed=52).cuda()
    out = m(images=img)
    assert out.shape == (3, 9, 64)
    tower = m.vit(img, None)                                  # [3, 26, 128]
    grid = tower[:, 1:].reshape(3, 5, 5, 128).contiguous().view(torch.int16).cpu().numpy()
    tokens = torch.from_numpy(I.pixel_shuffle_v2(grid, 2)).view(torch.bfloat16).cuda()
    assert torch.equal(out, m.forward_projection(tokens))
    with pytest.raises(NotImplementedError):
        vmod.MegatronVisionModel(hidden_size=64, vit_config=cfg, vision_downsample_ratio=0.5, vision_downsample_stride=2)
    # the "affine" projector (multimodal_projector.py:37-49): one linear layer after the pre-projection LayerNorm
    ma = vmod.MegatronVisionModel(hidden_size=64, vit_config=cfg, vision_projector_type="affine").to(torch.bfloat16).cuda()
    assert [n for n, _ in ma.vision_projection.named_parameters()] == ["encoder.weight"]
    x = _bf(2, 4, 4 * 128, seed=53).cuda().requires_grad_(True)
    y = ma.forward_projection(x)
    ln = ma.pre_proj_layernorm
    want = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.float(), (512,), ln.weight.float(), ln.bias.float(), ln.eps).to(torch.bfloat16),
                                      ma.vision_projection.encoder.weight)
    _close(y, want.float().cpu())
    y.sum().backward()
    assert x.grad is not None and ma.vision_projection.encoder.weight.grad is not None
    with pytest.raises(Exception, match="Unsupported multimodal projection type"):
        vmod.MultimodalProjector(8, 8, 8, "conv")


def test_image_to_llm_embedding_end_to_end():
    """One 448-pixel image through the vision model at InternViT-300M width (2 layers to keep the oracle quick):
    patch projection -> 1025 tokens -> tower -> class-token drop + pixel shuffle -> 256 tokens -> projector -> 5120 wide,
    then scattered into the LLM's embedding rows at the positions the data pipeline assigns to the image
    ([START] + 256 x CONTEXT + [END], dataset_qwen2.py:676-697; language_model_embedding.py:114-134)."""
    from long_vita_amd import intern_vit as iv, vision_model as vmod, vision, packing
    cfg = iv.InternViTConfig(num_layers=2)
    torch.manual_seed(11)
    m = vmod.MegatronVisionModel(hidden_size=5120, vit_config=cfg)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("ls1") or name.endswith("ls2"):
                prm.copy_(torch.randn_like(prm) * 0.3)
        m = m.to(torch.bfloat16)
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    img = _bf(1, 3, 448, 448, seed=61)
    want = vit_ref.vision_model_forward(img.float(), sd, cfg.num_attention_heads, cfg.patch_dim, cfg.layernorm_epsilon)
    feats = m.cuda()(images=img.cuda())
    assert feats.shape == (1, 256, 5120)
    _close(feats, want)

    seq, cur_len = 8192, 37
    image_indices, new_len = packing.frame_token_span(cur_len, 256)
    assert new_len == cur_len + 258
    emb = _bf(1, seq, 5120, seed=62).cuda()
    before = emb.clone()
    assert feats.requires_grad                                              # the projector trains in every stage
    out = vision.scatter_external_features(emb, {"features": feats, "indices": image_indices.cuda()})
    assert out.requires_grad                                                # the scatter keeps the projector in the graph
    (out.float() * 0.5).sum().backward()
    w2 = m.vision_projection.encoder.linear_fc2.weight
    assert w2.grad is not None and float(w2.grad.float().abs().max()) > 0
    pos = image_indices[1, 0]
    assert torch.equal(out[0, pos], feats.detach()[0])                      # the 256 context rows, bit for bit
    keep = torch.ones(seq, dtype=torch.bool)
    keep[pos] = False
    assert torch.equal(out[0, keep], before[0, keep])                       # every other row untouched


def test_tower_gradients_against_oracle_autograd(golden_dir):
    """The unfrozen tower (stage 2): gradients of every parameter through the HIP backward kernels (attention,
    LayerNorm, GELU) and the torch parts, against torch autograd through the fp32 oracle on the fixture's weights."""
    from long_vita_amd import intern_vit as iv
    g, sd = _golden(golden_dir)
    h, heads, ffn, n_layers, patch, img_hw, n = (int(v) for v in g["meta"])
    cfg = iv.InternViTConfig(hidden_size=h, ffn_hidden_size=ffn, num_attention_heads=heads, num_layers=n_layers, patch_dim=patch,
                             img_h=img_hw, img_w=img_hw, layernorm_epsilon=float(g["eps"]))
    m = iv.InternViTModel(cfg, freeze=False)
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.bfloat16).cuda()
    img = torch.from_numpy(g["model.img"])
    dout = _bf(*g["model.out"].shape, seed=71)
    out = m(img.to(torch.bfloat16).cuda())
    assert out.requires_grad
    _close(out, torch.from_numpy(g["model.out"]))
    out.backward(dout.cuda())
    rsd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    ref = vit_ref.model_forward(img, rsd, heads, patch, float(g["eps"]))
    ref.backward(dout.float())
    worst = 0.0
    for name, prm in m.named_parameters():
        want = rsd[name].grad
        assert prm.grad is not None, name
        err = (prm.grad.float().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        worst = max(worst, err)
        assert err <= 3e-2, (name, err)      # bf16 activations and gradients on the device against fp32


def test_frozen_tower_builds_no_graph():
    from long_vita_amd import intern_vit as iv
    cfg = iv.InternViTConfig(hidden_size=128, ffn_hidden_size=256, num_attention_heads=2, num_layers=1, img_h=28, img_w=28)
    m = iv.InternViTModel(cfg).to(torch.bfloat16).cuda()
    out = m(_bf(2, 3, 28, 28, seed=81).cuda())
    assert not out.requires_grad


def _vision_cp_worker(rank, world, port):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from long_vita_amd import intern_vit as iv, vision_model as vmod
    from long_vita_amd.vision_parallel import encode_images_context_parallel, images_needed_by_rank
    cfg = iv.InternViTConfig(hidden_size=128, ffn_hidden_size=256, num_attention_heads=2, num_layers=2, img_h=56, img_w=56)
    torch.manual_seed(5)                                     # the same weights on both ranks
    m = vmod.MegatronVisionModel(hidden_size=64, vit_config=cfg, vision_projector_freeze=True).to(torch.bfloat16).cuda()
    T = m.image_token_length
    assert T == 4
    S, n_img = 256, 7
    images = _bf(n_img, 3, 56, 56, seed=91).cuda()
    starts = torch.tensor([1, 11, 21, 62, 100, 61 + 64, 200])        # image 3 straddles the chunk edge at 64
    ind_s = starts[:, None] + torch.arange(T)[None]
    ext = torch.stack([torch.zeros_like(ind_s), ind_s]).cuda()
    calls = []

    def encode(x):
        calls.append(x.shape[0])
        return m(images=x)
    got = encode_images_context_parallel(images, ext, S, world, rank, encode, dist.group.WORLD)
    need = images_needed_by_rank(ext, S, world, rank)
    want = m(images=images[need])                            # what the reference's per-rank encode holds
    torch.cuda.synchronize()
    assert got.shape == want.shape == (need.numel(), T, 64)
    assert calls == [4 if rank == 0 else 3], calls           # 7 images dealt 4 + 3, whatever each rank needs
    err = (got.float() - want.float()).abs().max().item()
    assert err <= 2e-2 * want.float().abs().max().item(), err    # batch composition differs, so GEMM tiling may too
    dist.barrier()
    dist.destroy_process_group()


def test_image_encode_dealt_over_cp_ranks_two_processes_real_model():
    """vision_parallel on the GPU: two processes (gloo group; both ranks share the box's one GPU), the real
    MegatronVisionModel as the encoder, features routed point to point to the ranks that own tokens of each image."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_vision_cp_worker, args=(2, 37500 + (os.getpid() % 2000)), nprocs=2, join=True)


def test_full_size_tower_64_frames_24_layers():
    """BASELINE configs[1] at full size: InternViT-300M (24 layers, 16 heads x 64, 1025 tokens) on 64 frames of
    448 x 448 -> pixel shuffle -> projector.  Frames are independent, so every frame of the batched run must equal the
    same frame run alone (GEMM tilings may differ with the batch: a small tolerance), and the outputs are finite."""
    from long_vita_amd import intern_vit as iv, vision_model as vmod
    torch.manual_seed(2)
    cfg = iv.InternViTConfig()
    assert cfg.num_layers == 24 and cfg.hidden_size == 1024 and cfg.num_attention_heads == 16
    m = vmod.MegatronVisionModel(hidden_size=5120, vit_config=cfg, vision_projector_freeze=True).to(torch.bfloat16).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    img = (torch.randn(64, 3, cfg.img_h, cfg.img_w, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    with torch.no_grad():
        feats = m(images=img)
        assert feats.shape == (64, 256, 5120) and torch.isfinite(feats.float()).all()
        for i in (0, 31, 63):
            alone = m(images=img[i:i + 1])
            scale = alone.float().abs().max().item()
            err = (feats[i:i + 1].float() - alone.float()).abs().max().item()
            assert err <= 3e-2 * scale, (i, err, scale)
        # the tower alone: [64, 1025, 1024]
        tower = m.vit(img)
        assert tower.shape == (64, 1025, 1024) and torch.isfinite(tower.float()).all()


@pytest.mark.parametrize("rows,h", [(37, 1024), (9, 128), (5, 4096)])
def test_bias_scale_residual_layernorm_is_the_two_passes_in_one(rows, h):
    """out = residual + (x + bias) * ls and LayerNorm(out) from ONE pass equal the two separate kernels bit for bit
    (the fused pass rounds `out` to bf16 before the statistics, as the second pass would read it)."""
    from long_vita_amd import ops
    x, res = _bf(rows, h, seed=1).cuda(), _bf(rows, h, seed=2, scale=2.0).cuda()
    bias, ls = _bf(h, seed=3, scale=0.1).cuda(), _bf(h, seed=4, scale=0.3).cuda()
    gamma, beta = _bf(h, seed=5).cuda(), _bf(h, seed=6, scale=0.2).cuda()
    want_out = ops.bias_scale_residual_launch(x, bias, ls, res)
    want_y = ops.layernorm_launch(want_out, gamma, beta, 1e-6)
    out, y = ops.bias_scale_residual_layernorm_launch(x, bias, ls, res, gamma, beta, 1e-6)
    assert torch.equal(out, want_out) and torch.equal(y, want_y)
    out, y = ops.bias_scale_residual_layernorm_launch(x, None, ls, res, gamma, beta, 1e-6)
    want_out = ops.bias_scale_residual_launch(x, None, ls, res)
    assert torch.equal(out, want_out) and torch.equal(y, ops.layernorm_launch(want_out, gamma, beta, 1e-6))


def test_frozen_tower_fused_loop_equals_layer_by_layer():
    """The no-graph layer loop (LayerNorms folded into the passes that write their inputs) against the plain
    layer-by-layer forward of the same module."""
    from long_vita_amd import intern_vit as iv, ops
    torch.manual_seed(5)
    cfg = iv.InternViTConfig(num_layers=3, img_h=112, img_w=112)
    m = iv.InternViTModel(cfg).to(torch.bfloat16).cuda()
    img = _bf(2, 3, 112, 112, seed=9).cuda()
    with torch.no_grad():
        fused = m(img)
        patches = ops.patchify(img, cfg.patch_dim, m._patch_weight().shape[1])
        proj = torch.matmul(patches, m._patch_weight().t()).view(2, m.num_patches, cfg.hidden_size)
        tokens = ops.embed_tokens(proj, m.conv1.bias, m.class_token, m.position_embeddings.weight)
        for layer in m.decoder.layers:
            tokens = layer(tokens)
    assert torch.equal(fused, tokens.permute(1, 0, 2).contiguous())
