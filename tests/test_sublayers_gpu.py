"""The fused sub-layer nodes (ops.attn_sublayer / ops.ffn_sublayer: whole attention / feed-forward blocks with
hand-written backward) checked by directional finite differences in fp32, with and without dropout: the
counter-based masks are a pure function of (seed, step, call site), so f(x + eps d) sees the same masks."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _adopt(module, train):
    from shg_vqa_amd import modeling as M
    from shg_vqa_amd.engine import engine, reset_engine
    reset_engine(compute_dtype=torch.float32)
    E = engine()
    groups = []
    for name, sub in module.named_modules():
        if isinstance(sub, M.BertAttention):
            groups += sub.fusion_groups(name + ".")
    E.adopt(module, {n for n, _ in module.named_parameters()}, groups)
    gen = torch.Generator().manual_seed(1)
    E.param_arena.add_((0.05 * torch.randn(E.param_arena.numel(), generator=gen)).to(DEV))
    E.refresh_shadows()
    E.training = train
    return E


def _check(E, fn, inputs, rel=3e-2):
    """fn(*inputs) -> tensor.  Compares <grad, d> with central differences for every input and for the parameters."""
    gen = torch.Generator().manual_seed(2)
    inputs = [t.clone().requires_grad_(True) for t in inputs]

    def run(ts):
        E.begin_step()
        return fn(*ts)

    y = run(inputs)
    w = torch.randn(y.shape, generator=gen).to(DEV)
    E.zero_grad()
    (y.float() * w).sum().backward()
    E.join_side_streams()
    torch.cuda.synchronize()
    eps = 2e-3
    with torch.no_grad():
        for i, t in enumerate(inputs):
            d = torch.randn(t.shape, generator=gen).to(DEV)
            plus = [u.detach() + (eps * d if j == i else 0) for j, u in enumerate(inputs)]
            minus = [u.detach() - (eps * d if j == i else 0) for j, u in enumerate(inputs)]
            fd = ((run(plus).double() * w).sum() - (run(minus).double() * w).sum()) / (2 * eps)
            an = (t.grad.double() * d).sum()
            typ = (t.grad.double().norm() * d.double().norm() / d.numel() ** 0.5).item()     # size of a random projection
            assert abs(fd - an) <= rel * max(abs(fd), abs(an)) + 0.05 * typ, ("input %d" % i, fd.item(), an.item(), typ)
        d = torch.randn(E.param_arena.numel(), generator=gen).to(DEV)
        base = [u.detach() for u in inputs]
        # all weights move at once: ReLU units that flip inside the step add an error proportional to eps
        # (measured ~1 % at 5e-4 on the decoder's FFN), so take a small step; the slack term is 5 % of a typical projection
        eps = 1e-4
        E.param_arena.add_(eps * d)
        fp = (run(base).double() * w).sum()
        E.param_arena.add_(-2 * eps * d)
        fm = (run(base).double() * w).sum()
        E.param_arena.add_(eps * d)
        fd = (fp - fm) / (2 * eps)
        an = (E.grad_arena.double() * d[:E.n_active]).sum()
        typ = (E.grad_arena.double().norm() * d[:E.n_active].double().norm() / E.n_active ** 0.5).item()
        assert abs(fd - an) <= rel * max(abs(fd), abs(an)) + 0.05 * typ, ("params", fd.item(), an.item(), typ)
        assert E.grad_arena.abs().max() > 0


def _cfg(p_drop):
    from shg_vqa_amd.modeling import BertConfig
    return BertConfig(30522, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                      hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)


@pytest.mark.parametrize("train", [False, True])
def test_bert_layer_fused_sublayers(train):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.modeling import BertLayer
    torch.manual_seed(7)                      # module initialisers draw from the global generator
    layer = BertLayer(_cfg(0.1))
    E = _adopt(layer, train)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(3, 40, 128, generator=gen).to(DEV)
    mask = torch.zeros(3, 1, 1, 40)
    mask[1, :, :, 30:] = -10000.0
    mask = mask.to(DEV)
    _check(E, lambda h: layer(h, mask)[0], [x])


@pytest.mark.parametrize("train", [False, True])
def test_cross_layer_fused_sublayers(train):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.modeling import CrossLayer
    torch.manual_seed(7)                      # module initialisers draw from the global generator
    layer = CrossLayer(_cfg(0.1))
    E = _adopt(layer, train)
    gen = torch.Generator().manual_seed(4)
    lang = torch.randn(2, 40, 128, generator=gen).to(DEV)
    visn = torch.randn(2, 72, 128, generator=gen).to(DEV)
    lmask = torch.zeros(2, 1, 1, 40)
    lmask[0, :, :, 25:] = -10000.0
    lmask = lmask.to(DEV)

    def fn(l, v):
        lo, vo, _ = layer(l, lmask, v, None)
        return torch.cat([lo, vo], dim=1)

    _check(E, fn, [lang, visn])


@pytest.mark.parametrize("train", [False, True])
def test_decoder_layer_fused_sublayers(train):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.entry import rel_target_mask_device
    from shg_vqa_amd.transformer import TransformerDecoderLayer
    torch.manual_seed(7)                      # module initialisers draw from the global generator
    layer = TransformerDecoderLayer(128, 2, dim_feedforward=256, dropout=0.15)
    E = _adopt(layer, train)
    gen = torch.Generator().manual_seed(5)
    tgt = torch.randn(2, 32, 128, generator=gen).to(DEV)
    pos = torch.randn(2, 32, 128, generator=gen).to(DEV)
    mem = torch.randn(2, 56, 128, generator=gen).to(DEV)
    tmask = rel_target_mask_device(8, 4, DEV).float().contiguous()          # block-causal [32, 32]
    _check(E, lambda t, qp, m: layer.forward_bf(t, m, qp, tmask), [tgt, pos, mem])


def test_decoder_kv_ahead_gives_the_same_bits():
    """shg_run_t.kv_ahead moves the key / value projections of the encoder memory to the weight-gradient stream, ahead of the
    chain (one event per layer): same kernels, same operands - outputs and gradients must not change by a bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.entry import rel_target_mask_device
    from shg_vqa_amd.transformer import TransformerDecoder, TransformerDecoderLayer
    res = {}
    for mode in (0, 2):
        torch.manual_seed(7)
        dec = TransformerDecoder(TransformerDecoderLayer(128, 2, dim_feedforward=256, dropout=0.15), 3)
        E = _adopt(dec, True)
        E.kv_ahead = mode
        gen = torch.Generator().manual_seed(5)
        pos = torch.randn(2, 32, 128, generator=gen).to(DEV).requires_grad_(True)
        mem = torch.randn(2, 56, 128, generator=gen).to(DEV).requires_grad_(True)
        tmask = rel_target_mask_device(8, 4, DEV).float().contiguous()
        E.begin_step()
        y = dec.forward_bf(None, mem, pos, tmask)
        E.zero_grad()
        y.float().square().sum().backward()
        E.join_side_streams()
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), mem.grad.clone(), pos.grad.clone(), E.grad_arena.clone())
    for a, b in zip(res[0], res[2]):
        assert torch.equal(a, b)
    assert res[0][3].abs().max() > 0
