"""TransformerAM on libpk2hip.so against the reference PyTorch CPU computation: the parameter containers are
the torch modules the reference instantiates, so their own CPU forward (looped layer by layer as SURVEY.md 8c
prescribes for torch >= 2) is the reference."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pykaldi2_amd import transformer

pytestmark = pytest.mark.gpu


def _reference_forward(m, x, src_mask, kpm):
    h = m.input_layer(x)
    for lp in m.transformer.layers:
        h = lp.encoder_layer(h, src_mask, kpm)
        h = F.relu(lp.conv1d(h.permute(1, 2, 0))).permute(2, 0, 1)
    return m.output_layer(m.transformer.norm(h))


@pytest.mark.parametrize("cfg", [dict(D=20, C=64, H=4, FF=128, L=2, P=37, T=11, B=3, look=2),
                                 dict(D=80, C=512, H=8, FF=2048, L=2, P=301, T=48, B=2, look=-1)])
def test_transformer_matches_torch_cpu(cfg):
    torch.manual_seed(0)
    m = transformer.TransformerAM(cfg["D"], cfg["C"], cfg["H"], cfg["FF"], cfg["L"], 0.0, cfg["P"])
    for lp in m.transformer.layers:          # break the deep-copy symmetry of the default init
        for p in lp.parameters():
            p.data.add_(0.02 * torch.randn_like(p))
    ref = copy.deepcopy(m).eval()
    T, B = cfg["T"], cfg["B"]
    x = torch.randn(T, B, cfg["D"])
    lens = [T] + [max(1, T - 4 - 2 * i) for i in range(B - 1)]
    kpm = torch.ones(B, T)
    for i, n in enumerate(lens):
        kpm[i, :n] = 0
    kpm = kpm.bool()
    src_mask = None
    if cfg["look"] > -1:
        tri = torch.tril(torch.ones(T, T), diagonal=cfg["look"])
        src_mask = tri.float().masked_fill(tri == 0, float("-inf")).masked_fill(tri == 1, 0.0)
    w = torch.randn(T, B, cfg["P"])
    for i, n in enumerate(lens):
        w[n:, i] = 0        # padded query rows carry no gradient (their loss is ignored in the reference)
    want = _reference_forward(ref, x, src_mask, kpm)
    (want * w).sum().backward()
    m = m.cuda().train()
    got = m(x.cuda(), src_mask.cuda() if src_mask is not None else None, kpm.cuda())
    valid = torch.zeros(T, B, dtype=torch.bool)
    for i, n in enumerate(lens):
        valid[:n, i] = True
    err = (got.cpu() - want.detach())[valid].abs().max().item()
    assert err < 2e-4 * max(1.0, want.detach()[valid].abs().max().item()), err
    (got * w.cuda()).sum().backward()
    refg = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        g, rg = p.grad.cpu(), refg[name].grad
        e = (g - rg).abs().max().item()
        assert e < 5e-4 * max(1e-2, rg.abs().max().item()), (name, e, rg.abs().max().item())


def test_transformer_state_dict_keys_and_init_follow_torch():
    torch.manual_seed(3)
    m = transformer.TransformerAM(80, 512, 8, 2048, 2, 0.1, 5768)
    keys = list(m.state_dict().keys())
    assert keys[0] == "pos_encoder.pe" and "transformer.layers.1.encoder_layer.self_attn.in_proj_weight" in keys
    assert "transformer.layers.0.conv1d.weight" in keys and keys[-2:] == ["transformer.norm.weight", "transformer.norm.bias"]
    n = sum(p.numel() for p in m.parameters())
    assert n == 3001480 + 3939328 * 2
    # all layers start as copies of one initialised layer (nn.TransformerEncoder deep copies)
    a, b = m.transformer.layers[0], m.transformer.layers[1]
    assert torch.equal(a.conv1d.weight, b.conv1d.weight)
