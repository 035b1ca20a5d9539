"""End-to-end parity of the HIP path (through the C ABI) with golden vectors produced by the real
reference (tests/golden, oracle/gen_golden.py) and with the CPU oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _load_det_weights(model):
    from oracle import detweights
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.data.copy_(torch.from_numpy(detweights.tensor_for(name, tuple(prm.shape))))
    from shg_vqa_amd.engine import engine
    engine().refresh_shadows()


def _rel_err(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


def _build(compute_dtype, star=False):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.agqa_hgqa import AGQA, SyntheticAGQA, DataTuple
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd.engine import reset_engine
    from shg_vqa_amd.param import hgqa_args
    reset_engine(compute_dtype=compute_dtype)
    args = hgqa_args(compute_dtype="fp32" if compute_dtype == torch.float32 else "bf16", use_hg_mask=star)
    n_ans, n_rel, n_act = (4, 563, 111) if star else (171, 456, 157)
    model = AGQAModel(n_ans, num_queries=128, num_classes=n_rel, num_actions=n_act, args=args)
    model.to_engine(compute_dtype)
    _load_det_weights(model)
    dset = SyntheticAGQA(n=4)
    dset.num_answers, dset.rel_classes, dset.action_classes = n_ans, n_rel, list(range(n_act))
    tup = DataTuple(dset, [None] * 10, None)
    trainer = AGQA(args, train_tuple=tup, model=model, t_total=100)
    return trainer


def _oracle_batch(tag, g):
    from oracle import shg_ref
    cfg = shg_ref.Cfg() if tag == "hgqa" else shg_ref.Cfg(num_answers=4, rel_classes=564, act_classes=112, use_hg_mask=True)
    return cfg, shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]))


def _device_batch(batch):
    out = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items() if not k.endswith("_targets")}
    out["pos"] = out["pos"].float()
    out["lengths"] = out["lengths"].to(torch.int32)
    out["act_lengths"] = out["act_lengths"].to(torch.int32)
    return out


@pytest.mark.parametrize("tag", ["hgqa", "star"])
def test_fp32_forward_losses_matching_and_gradients_vs_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"agqa_{tag}_b2.npz"))
    tr = _build(torch.float32, star=(tag == "star"))
    cfg, batch = _oracle_batch(tag, g)
    b = _device_batch(batch)
    tr.model.eval()
    from shg_vqa_amd.engine import engine
    engine().begin_step()
    engine().zero_grad()
    out = tr.forward_losses(b)
    # north_star: answer-logit parity <= 1e-3 relative, fp32
    for key, ref in (("logit", "logit"), ("hg_logit", "hg_logit"), ("rel_logit", "rel_preds"), ("act_logit", "act_preds")):
        e = _rel_err(out[key], g[ref])
        assert e < 1e-3, (key, e)
    # Hungarian indices: bit-exact
    for key in ("rel", "act"):
        q, t = out[f"{key}_idx"]
        assert np.array_equal(q.cpu().numpy(), g[f"{key}_q"]), key
        assert np.array_equal(t.cpu().numpy(), g[f"{key}_t"]), key
        assert np.array_equal(out[f"{key}_grid"].cpu().numpy(), g[f"{key}_grid"]), key
    for key in ("bce", "rel_ce", "act_ce", "total", "rel_err", "act_err"):
        ref = float(g[key])
        assert abs(float(out[key]) - ref) <= 1e-3 * max(abs(ref), 1.0), (key, float(out[key]), ref)
    out["total"].backward()
    engine().join_side_streams()           # weight gradients are produced on a side stream
    names = [str(x) for x in g["grad_names"]]
    params = dict(tr.model.named_parameters())
    worst = 0.0
    for i, n in enumerate(names):
        gr = params[n].grad
        assert gr is not None, n
        ref = g["grad_norms"][i]
        got = gr.double().norm().item()
        worst = max(worst, abs(got - ref) / max(ref, 1e-9))
        # (the key-bias gradients are mathematically zero - softmax shift invariance - so only rounding noise)
        assert abs(got - ref) <= 5e-3 * ref + 1e-6, (n, got, ref)
        head = gr.reshape(-1)[:4].float().cpu().numpy() if gr.is_contiguous() else gr.permute(0, 2, 3, 4, 1).reshape(-1)[:0].cpu().numpy()
        if gr.is_contiguous():
            assert np.allclose(head, g["grad_heads"][i], rtol=2e-2, atol=1e-6 + 2e-4 * ref), n
    from shg_vqa_amd.optimization import clip_grad_norm_
    tot = clip_grad_norm_(tr.model.parameters(), 5.0).item()
    assert abs(tot - float(g["grad_total_norm"])) <= 2e-3 * float(g["grad_total_norm"])
    # parameters outside the active set received nothing
    act = tr.model.active_parameter_names()
    for n, p in tr.model.named_parameters():
        if n not in act:
            assert p.grad is None


def test_bf16_forward_is_close_to_reference_and_matching_is_self_consistent(golden_dir):
    """bf16 storage/operands: tolerance documented in DESIGN.md (2^-8 relative rounding per op); the
    Hungarian indices must equal the oracle's solution for the bf16 logits the model produced."""
    from oracle import shg_ref
    g = np.load(os.path.join(golden_dir, "agqa_hgqa_b2.npz"))
    tr = _build(torch.bfloat16)
    cfg, batch = _oracle_batch("hgqa", g)
    b = _device_batch(batch)
    tr.model.eval()
    from shg_vqa_amd.engine import engine
    engine().begin_step()
    engine().zero_grad()
    out = tr.forward_losses(b)
    measured = {}
    for key, ref in (("logit", "logit"), ("hg_logit", "hg_logit"), ("rel_logit", "rel_preds"), ("act_logit", "act_preds")):
        measured[key] = _rel_err(out[key], g[ref])
    exp = shg_ref.hungarian_per_frame(out["rel_logit"].float().cpu(), batch["rel_targets"], 16)
    q, t = out["rel_idx"]
    for n, (qi, ti) in enumerate(exp):
        k = len(qi)
        assert torch.equal(q[n, :k].cpu(), qi) and torch.equal(t[n, :k].cpu(), ti), n
    measured["total"] = abs(float(out["total"].detach()) - float(g["total"])) / float(g["total"])
    out["total"].backward()
    engine().join_side_streams()
    names = [str(x) for x in g["grad_names"]]
    params = dict(tr.model.named_parameters())
    rel = []
    for i, n in enumerate(names):
        ref = g["grad_norms"][i]
        got = params[n].grad.double().norm().item()
        rel.append((abs(got - ref) / (ref + 1e-6), n))
    rel.sort()
    measured["grad_norm_p50"], measured["grad_norm_p95"], measured["grad_norm_max"] = rel[len(rel) // 2][0], rel[int(len(rel) * 0.95)][0], rel[-1][0]
    tot_ref = float(g["grad_total_norm"])
    from shg_vqa_amd.optimization import clip_grad_norm_
    measured["grad_total_norm"] = abs(clip_grad_norm_(tr.model.parameters(), 5.0).item() - tot_ref) / tot_ref
    print("bf16 end-to-end errors vs the reference golden (B=2):", {k: float("%.3g" % v) for k, v in measured.items()},
          "worst gradient:", rel[-1][1])
    # bars = 2x what this prints on MI355X (round 2, gpurun_out/r2_t3.log: logit 1.20e-2, hg_logit 1.18e-2, rel_logit 1.40e-2,
    # act_logit 1.53e-2 of the logit scale; total loss 1.6e-4; gradient norms p50 3.9e-4 / p95 1.7e-3; total gradient norm 2.4e-4;
    # the worst single tensor is a key bias whose gradient is mathematically zero - softmax shift invariance - i.e. noise / 0)
    for key, bar in BF16_BARS.items():
        assert measured[key] <= bar, (key, measured[key], bar)


BF16_BARS = dict(logit=2.5e-2, hg_logit=2.5e-2, rel_logit=3e-2, act_logit=3e-2, total=4e-4, grad_norm_p95=4e-3, grad_total_norm=6e-4)


def test_train_steps_match_oracle_fp32():
    """Three full optimiser steps (dropout off) against the CPU oracle's train_step: losses, the
    clipped gradient norm and a sample of updated weights."""
    from oracle import shg_ref
    tr = _build(torch.float32)
    cfg = shg_ref.Cfg()
    p = shg_ref.det_params(cfg, requires_grad=True)
    state = {}
    sample = ["logit_fc.3.weight", "rel_decoder.layers.4.linear2.weight", "lxrt_encoder.model.bert.encoder.r_layers.0.output.dense.weight",
              "lxrt_encoder.model.bert.encoder.visn_fc.conv.4.bias", "hgq_encoder.rel_token"]
    params = dict(tr.model.named_parameters())
    from shg_vqa_amd.engine import engine
    for step in range(3):
        batch = shg_ref.synthetic_batch(2, cfg, seed=500 + step)
        out_o, losses_o, grads_o, norm_o = shg_ref.train_step(p, cfg, batch, state, lr=1e-4, step=step, t_total=100)
        b = _device_batch(batch)
        tr.optim.param_groups[0]["lr"] = 1e-4
        # dropout off for a deterministic comparison: run the step with the engine in eval mode
        engine().begin_step()
        tr.optim.zero_grad()
        engine().training = False
        out = tr.forward_losses(b)
        out["total"].backward()
        from shg_vqa_amd.optimization import clip_grad_norm_
        norm = clip_grad_norm_(tr.model.parameters(), 5.0)
        tr.optim.step()
        assert abs(float(out["total"]) - float(losses_o["total"])) < 2e-3 * abs(float(losses_o["total"])), step
        assert abs(norm.item() - float(norm_o)) < 5e-3 * float(norm_o), (step, norm.item(), float(norm_o))
        for n in sample:
            got, ref = params[n].detach().float().cpu(), p[n].detach()
            assert torch.allclose(got, ref, rtol=1e-3, atol=2e-6), (step, n, (got - ref).abs().max().item())
    assert int(engine().step_state.item()) == 3


def test_state_dict_roundtrip_keeps_reference_keys(tmp_path, golden_dir):
    import json
    tr = _build(torch.bfloat16)
    spec = json.load(open(os.path.join(golden_dir, "agqa_state_dict_spec.json")))
    sd = tr.model.state_dict()
    assert [k for k, _, _ in spec["state_dict"]] == list(sd.keys())
    for k, shape, _ in spec["state_dict"]:
        assert list(sd[k].shape) == shape, k
    tr.output = str(tmp_path)
    tr.save("CKPT")
    w = sd["lxrt_encoder.model.bert.encoder.visn_fc.conv.4.weight"].detach().clone()
    with torch.no_grad():
        for p in tr.model.parameters():
            p.data.zero_()
    tr.load(os.path.join(str(tmp_path), "CKPT"))
    assert torch.equal(tr.model.state_dict()["lxrt_encoder.model.bert.encoder.visn_fc.conv.4.weight"], w)
    # aliases of the shared x-layer share storage, as in the reference
    sd2 = tr.model.state_dict()
    a = sd2["lxrt_encoder.model.bert.encoder.x_layers.0.visual_attention.att.query.weight"]
    c = sd2["lxrt_encoder.model.bert.encoder.cross_attn_layer.cross.visual_attention.att.query.weight"]
    assert a.data_ptr() == c.data_ptr()


def _build_task(task_flag, extra=(), n_ans=171, t_total=100):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.agqa_hgqa import AGQA, SyntheticAGQA, DataTuple
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd.engine import reset_engine
    from shg_vqa_amd.param import parse_args
    reset_engine(compute_dtype=torch.float32)
    args = parse_args(["--noCaps", task_flag, "--fromScratch", "--computeDtype", "fp32", "--lr", "1e-4"] + list(extra))
    model = AGQAModel(n_ans, args=args)
    model.to_engine(torch.float32)
    _load_det_weights(model)
    return AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=4), [None] * 10, None), model=model, t_total=t_total)


def test_question_only_model_matches_reference_golden(golden_dir):
    """BASELINE.json configs[0]: agqaQ.py --taskQ, llayers=2 (forward logits + BCE loss vs the reference)."""
    from oracle import shg_ref
    g = np.load(os.path.join(golden_dir, "agqa_q_b4.npz"))
    tr = _build_task("--taskQ", ["--llayers", "2"])
    cfg = shg_ref.Cfg(llayers=2, task="q")
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]), with_feat=False)
    b = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    tr.model.eval()
    from shg_vqa_amd.engine import engine
    engine().begin_step()
    out = tr.forward_losses(b)
    assert _rel_err(out["logit"], g["logit"]) < 1e-3
    assert abs(float(out["total"]) - float(g["loss"])) < 1e-3 * float(g["loss"])
    assert set(n for n, _ in tr.model.named_parameters()) == set(str(x) for x in g["param_names"])


def test_question_only_backward_and_two_optimiser_steps_match_reference_golden(golden_dir):
    """BASELINE.json configs[0] through the loop of agqaQ.py:186-300 on the REAL reference (oracle/gen_golden.py q): the gradient
    of each of the 43 tensors that train, the clipped global norm, and two BertAdam steps (the first at learning rate 0 under the
    warm-up): first values of every updated tensor and the loss afterwards."""
    from oracle import shg_ref
    g = np.load(os.path.join(golden_dir, "agqa_q_b4.npz"))
    tr = _build_task("--taskQ", ["--llayers", "2", "--lr", repr(float(g["lr"]))], t_total=int(g["t_total"]))
    cfg = shg_ref.Cfg(llayers=2, task="q")
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]), with_feat=False)
    b = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.optimization import clip_grad_norm_
    E = engine()
    tr.model.eval()
    names = [str(x) for x in g["grad_names"]]
    params = dict(tr.model.named_parameters())
    assert set(names) == tr.model.active_parameter_names()
    for step in range(2):
        E.begin_step()
        tr.optim.zero_grad()
        E.training = False
        out = tr.forward_losses(b)
        out["total"].backward()
        E.join_side_streams()
        if step == 0:
            for i, n in enumerate(names):
                ref = g["grad_norms"][i]
                got = params[n].grad.double().norm().item()
                assert abs(got - ref) <= 5e-3 * ref + 1e-6, (n, got, ref)
                head = params[n].grad.reshape(-1)[:4].float().cpu().numpy()
                assert np.allclose(head, g["grad_heads"][i], rtol=5e-3, atol=5e-3 * ref / max(params[n].numel(), 1) ** 0.5 + 1e-7), (n, head)
        tot = clip_grad_norm_(tr.model.parameters(), 5.0).item()
        if step == 0:
            assert abs(tot - float(g["grad_total_norm"])) <= 2e-3 * float(g["grad_total_norm"])
        tr.optim.step()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        head = params[n].detach().reshape(-1)[:4].float().cpu().numpy()
        assert np.allclose(head, g["after_heads"][i], rtol=1e-3, atol=2e-5), (n, head, g["after_heads"][i])
    E.begin_step()
    out = tr.forward_losses(b)
    assert abs(float(out["total"]) - float(g["loss_after"])) < 2e-3 * float(g["loss_after"])


def test_vqa_task_matches_reference_golden(golden_dir):
    """BASELINE.json configs[1] against the REAL reference (agqaVQA.py:237-258; oracle/gen_golden.py vqa): answer logits <= 1e-3,
    BCE * n_answers, and the norm of every gradient (201 tensors: here the x-layers and pooler_dict.cross train)."""
    from oracle import shg_ref
    g = np.load(os.path.join(golden_dir, "agqa_vqa_b2.npz"))
    tr = _build_task("--taskVQA")
    cfg = shg_ref.Cfg(task="vqa")
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]))
    b = _device_batch(batch)
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.optimization import clip_grad_norm_
    engine().begin_step()
    tr.optim.zero_grad()
    tr.model.eval()
    out = tr.forward_losses(b)
    assert _rel_err(out["logit"], g["logit"]) < 1e-3
    assert abs(float(out["total"]) - float(g["loss"])) < 1e-3 * float(g["loss"])
    out["total"].backward()
    engine().join_side_streams()
    names = [str(x) for x in g["grad_names"]]
    params = dict(tr.model.named_parameters())
    assert set(names) == tr.model.active_parameter_names()
    for i, n in enumerate(names):
        ref = g["grad_norms"][i]
        got = params[n].grad.double().norm().item()
        assert abs(got - ref) <= 5e-3 * ref + 1e-6, (n, got, ref)
    tot = clip_grad_norm_(tr.model.parameters(), 5.0).item()
    assert abs(tot - float(g["grad_total_norm"])) <= 2e-3 * float(g["grad_total_norm"])


def test_vqa_task_train_step_matches_oracle():
    """BASELINE.json configs[1]: agqaVQA.py --taskVQA 5/2/5 (video + question, BCE on the LXRT answer logit):
    here the x-layers and their pooler DO receive gradients."""
    from oracle import shg_ref
    tr = _build_task("--taskVQA")
    cfg = shg_ref.Cfg(task="vqa")
    p = shg_ref.det_params(cfg, requires_grad=True)
    batch = shg_ref.synthetic_batch(2, cfg, seed=31)
    out_o, losses_o, grads_o, norm_o = shg_ref.train_step(p, cfg, batch, {}, lr=1e-4, step=3, t_total=100)
    b = _device_batch(batch)
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.optimization import clip_grad_norm_
    engine().step_state.fill_(3)
    engine().begin_step()
    tr.optim.zero_grad()
    engine().training = False
    out = tr.forward_losses(b)
    out["total"].backward()
    norm = clip_grad_norm_(tr.model.parameters(), 5.0)
    tr.optim.step()
    assert _rel_err(out["logit"], out_o["logit"].detach()) < 1e-3
    assert abs(float(out["total"]) - float(losses_o["total"])) < 1e-3 * float(losses_o["total"])
    assert abs(norm.item() - float(norm_o)) < 5e-3 * float(norm_o)
    params = dict(tr.model.named_parameters())
    assert set(grads_o) == tr.model.active_parameter_names()
    for n in ("lxrt_encoder.model.bert.encoder.cross_attn_layer.cross.visn_output.dense.weight",
              "lxrt_encoder.model.bert.pooler_dict.cross.dense2.weight", "logit_fc.0.bias"):
        assert torch.allclose(params[n].detach().float().cpu(), p[n].detach(), rtol=1e-3, atol=2e-6), n


def test_training_loop_entry_points_run_and_learn(tmp_path):
    """get_tuple / AGQA.train / predict / test / evaluate / save / load on a tiny synthetic split: the loss of a
    repeated batch must go down and a reloaded checkpoint must reproduce the predictions."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd.agqa_hgqa import AGQA, batch_to_device, get_tuple
    from shg_vqa_amd.engine import engine, reset_engine
    from shg_vqa_amd.param import hgqa_args
    reset_engine(compute_dtype=torch.bfloat16)
    args = hgqa_args(batch_size=4, epochs=1, lr=2e-4, output=str(tmp_path), log_freq=1000)
    torch.manual_seed(1)
    train = get_tuple("train", 4, shuffle=False, drop_last=True, n=8)
    valid = get_tuple("valid", 4, n=8)
    agqa = AGQA(args, train_tuple=train, valid_tuple=valid, t_total=40)
    engine().step_state.fill_(10)                      # past the lr = 0 first step of the schedule
    b = batch_to_device(next(iter(train.loader)), agqa.device)
    losses = [float(agqa.train_step(b)["total"]) for _ in range(6)]
    assert losses[-1] < losses[0], losses
    agqa.train(train, valid)                           # one epoch through the loop (saves CURRENT/BEST/LAST)
    assert os.path.exists(os.path.join(str(tmp_path), "LAST.pth"))
    p1 = agqa.predict(valid)
    t1 = agqa.test(valid, dump=os.path.join(str(tmp_path), "pred.json"))
    assert p1 == t1 and len(p1) == 8
    with torch.no_grad():
        for prm in agqa.model.parameters():
            prm.data.mul_(0.5)
    engine().refresh_shadows()
    agqa.load(os.path.join(str(tmp_path), "LAST"))
    assert agqa.predict(valid) == p1
    assert 0.0 <= agqa.evaluate(valid) <= 1.0


def test_cached_channels_last_features_feed_conv1_like_the_ncdhw_tensor(tmp_path):
    """feature_cache.py: bf16 channels-last clips through the pinned double-buffered loader give the first
    convolution exactly the input the NCDHW fp32 path builds."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from shg_vqa_amd import ops
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.feature_cache import FeatureCache, PrefetchLoader, write_feature_cache
    tr = _build(torch.bfloat16)
    gen = torch.Generator().manual_seed(5)
    clips = [torch.randn(2048, 16, 7, 7, generator=gen) for _ in range(5)]
    prefix = str(tmp_path / "feats")
    write_feature_cache(prefix, clips)
    cache = FeatureCache(prefix)
    w1, b1 = tr._conv1_params()
    batches = [[0, 1], [2, 3], [4, 0]]
    junk = torch.randn(4096, 4096, device=DEV)
    for idx, dev_batch in zip(batches, PrefetchLoader(cache, batches, device=DEV)):
        x_cl, y1p, pre1 = ops.conv1_forward(dev_batch, w1, b1)
        ref = torch.stack([clips[i] for i in idx]).to(DEV)
        x_ref, y_ref, pre_ref = ops.conv1_forward(ref, w1, b1)
        junk = junk @ junk * 1e-4                     # keep the consumer stream busy while the next copy runs
        assert torch.equal(x_cl, x_ref) and torch.equal(y1p, y_ref) and torch.equal(pre1, pre_ref)


def test_per_clip_matching_losses_vs_oracle(golden_dir):
    """Without --LossHGPerFrame the matcher solves one problem per clip (matcher.py:82-104) and the weighted CE runs
    over (B, Q) slots (agqaHGQA.py:203-229): matcher interface, device-side target packing and both set losses."""
    from oracle import shg_ref
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.matcher import HungarianMatcher
    g = np.load(os.path.join(golden_dir, "agqa_hgqa_b2.npz"))
    tr = _build(torch.float32)
    tr.args.loss_hg_per_frame = False
    tr.matcher = HungarianMatcher(cost_class=1, loss_hg_per_frame=False, clip_len=tr.clip_len)
    cfg, batch = _oracle_batch("hgqa", g)
    b = _device_batch(batch)
    tr.model.eval()
    engine().begin_step()
    engine().zero_grad()
    out = tr.forward_losses(b)
    for key, logit, trip, lens, w in (("rel", out["rel_logit"], batch["rel_triplets"], batch["lengths"], tr.empty_weight),
                                      ("act", out["act_logit"], batch["act_tokens"], batch["act_lengths"], tr.empty_weight_acts)):
        B, T, per = trip.shape
        labels = [torch.cat([trip[i, j, :int(lens[i, j])] for j in range(T)]) for i in range(B)]
        lg = logit.detach().float().cpu()
        idx = shg_ref.hungarian_per_frame(lg, labels, clip_len=1)
        loss, err, grid = shg_ref.set_loss(lg, labels, idx, w.detach().float().cpu(), clip_len=1)
        q, t = out[key + "_idx"]
        for i, (qi, ti) in enumerate(idx):
            n = len(qi)
            assert torch.equal(q[i, :n].cpu(), qi) and torch.equal(t[i, :n].cpu(), ti), (key, i)
        assert torch.equal(out[key + "_grid"].cpu(), grid)
        assert abs(float(out[key + "_ce"]) - float(loss)) <= 1e-3 * max(abs(float(loss)), 1.0), (key, float(out[key + "_ce"]), float(loss))
        # reference-style interface: list of (index_i, index_j) per sample
        pairs = tr.matcher({"pred_logits": logit.detach()}, [{"labels": l} for l in labels])
        for (qi, ti), (rq, rt) in zip(pairs, idx):
            assert torch.equal(qi, rq) and torch.equal(ti, rt)
    out["total"].backward()
    engine().join_side_streams()
    assert torch.isfinite(engine().grad_arena).all() and engine().grad_arena.abs().max() > 0


def test_overlapped_optimizer_update_gives_the_same_parameters():
    """train_step(overlap_update=True) leaves BertAdam's sweep over everything but conv1's weight / bias on a side stream
    (it overlaps the next step's conv1) and zeroes the gradients in the same pass: three steps must end bit-identical to
    the plain schedule (dropout off; the split changes where kernels run, not what they compute)."""
    from shg_vqa_amd.engine import engine
    res = []
    for overlap in (False, True):
        tr = _build(torch.bfloat16)
        from oracle import shg_ref
        cfg = shg_ref.Cfg()
        batches = [_device_batch(shg_ref.synthetic_batch(2, cfg, seed=50 + i)) for i in range(3)]
        tr.model.eval()                              # dropout off (train_step switches the engine flag back on: force p = 0)
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        for b in batches:
            tr.train_step(b, overlap_update=overlap)
        engine().wait_params_ready()
        torch.cuda.synchronize()
        res.append((engine().param_arena.clone(), engine().grad_arena.clone(), int(engine().step_state.item())))
    (p0, g0, s0), (p1, g1, s1) = res
    assert s0 == s1 == 3
    # the optimiser pass leaves the gradient arena zeroed - but for the gradients their single writer SETS in every step (the conv
    # weights, Engine.claim_overwrite)
    for off, n in engine().unzeroed.items():
        g0[off:off + n] = 0
        g1[off:off + n] = 0
    assert (g0 == 0).all() and (g1 == 0).all()
    # fp32 atomics (split-K weight gradients, the bias-gradient column sums of the attention backward kernels, shared weights)
    # make two runs differ in the last bits of a gradient; BertAdam's m / (sqrt(v) + eps) turns that into an O(1) change of the
    # update direction where a gradient element is itself ~0 (without bias correction |m / sqrt(v)| reaches 0.1 / sqrt(0.001) = 3.2 in
    # the first steps), so single elements may differ by up to 2 x 3.2 x the sum of the steps' learning rates (1e-5 x (0 + 0.1 +
    # 0.2) = 3e-6 here) - but only a vanishing share of the 365 M parameters does
    d = (p0 - p1).abs()
    assert d.max().item() <= 2e-5, d.max()
    assert (d > 2e-7).float().mean().item() < 1e-4, (d > 2e-7).float().mean()


def test_single_writer_gradients_give_the_same_norm_and_parameters():
    """The conv weights' gradients are SET by their one writer, which adds their share of the gradient norm on the way; the norm's
    pass skips them and BertAdam does not zero them (Engine.claim_overwrite, shg_conv3d_k533_wgrad_sumsq).  Against the plain
    scheme (accumulate into a zeroed arena, one pass over all of it): the norm equals a full pass over the very same gradients
    in every step; after the first step the conv weights' moments are bit-identical; three steps end at the same parameters."""
    from shg_vqa_amd.engine import engine
    from oracle import shg_ref
    import shg_vqa_amd.agqa_hgqa as trainer_mod
    from shg_vqa_amd import kernels as KK
    cfg = shg_ref.Cfg()
    real_clip = trainer_mod.clip_grad_norm_
    same_grads = []

    def checked_clip(params, max_norm):
        e = engine()
        e.join_side_streams()
        full = float(KK.grad_norm(e.grad_arena))           # one pass over the whole arena, on the very same gradients
        out = real_clip(params, max_norm)
        same_grads.append((float(out), full))
        return out

    res = []
    for fused in (False, True):
        tr = _build(torch.bfloat16)
        e = engine()
        e.fused_conv_norm = fused
        batches = [_device_batch(shg_ref.synthetic_batch(2, cfg, seed=70 + i)) for i in range(3)]
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        trainer_mod.clip_grad_norm_ = checked_clip
        try:
            norms, first = [], None
            for b in batches:
                norms.append(float(tr.train_step(b)["grad_norm"]))
                if first is None:
                    torch.cuda.synchronize()
                    convs = [p for n, p in tr.model.named_parameters() if p.dim() == 5]
                    assert len(convs) == 2
                    first = [(e.m_arena[p._shg_off:p._shg_off + p._shg_numel].clone(), e.v_arena[p._shg_off:p._shg_off + p._shg_numel].clone())
                             for p in convs]
                    assert sorted(e.unzeroed) == (sorted(p._shg_off for p in convs) if fused else [])
        finally:
            trainer_mod.clip_grad_norm_ = real_clip
        torch.cuda.synchronize()
        if fused:
            assert float(e.norm_scalar()) == 0.0              # consumed and reset by the norm's last kernel
        res.append((e.param_arena.clone(), norms, first))
    (p0, n0, f0), (p1, n1, f1) = res
    assert len(same_grads) == 6
    for got, full in same_grads:
        assert abs(got - full) <= 2e-6 * full, same_grads
    for (m0, v0), (m1, v1) in zip(f0, f1):                    # same gradients, same clip factor, same update - bit for bit
        assert torch.equal(m0, m1) and torch.equal(v0, v1)
    # (later steps: two RUNS drift apart through the fp32 atomics of other kernels' bias gradients, as in the test above)
    for a, b in zip(n0, n1):
        assert abs(a - b) <= 2e-3 * abs(a), (n0, n1)
    assert (p0 - p1).abs().max().item() <= 2e-5
    # and a step that does NOT write the conv gradients afterwards must not see last step's values
    e = engine()
    assert e.unzeroed
    e.settle_stale_grads()
    assert not e.unzeroed and float(e.grad_arena.abs().max()) == 0.0


def test_whole_step_hipgraph_replay_matches_eager_steps():
    """AGQA.capture / train_step_graphed (bench.py --exec graph): the optimiser step captured into one hipGraph (the action
    decoder / heads / set losses on their branch stream, weight gradients on theirs, the language branch inline) gives the
    losses of eager multi-stream steps."""
    from oracle import shg_ref
    from shg_vqa_amd.transformer import MultiheadAttention
    cfg = shg_ref.Cfg()
    losses = []
    for graphed in (False, True):
        tr = _build(torch.bfloat16)
        batches = [_device_batch(shg_ref.synthetic_batch(2, cfg, seed=90 + i)) for i in range(3)]
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, MultiheadAttention):
                m.dropout = 0.0
        tr.train_step(batches[0])              # (creates the side streams the capture then has to leave alone)
        if graphed:
            tr.capture(batches[0])             # two warm-up steps on the example batch; the capture pass itself runs nothing
            out = [float(tr.train_step_graphed(b)["total"]) for b in batches]
            out.append(float(tr.train_step(batches[0])["total"]))       # and eager steps still work afterwards
        else:
            for _ in range(2):
                tr.train_step(batches[0])
            out = [float(tr.train_step(b)["total"]) for b in batches + batches[:1]]
        torch.cuda.synchronize()
        losses.append(out)
    for a, b in zip(*losses):
        assert abs(a - b) <= 2e-3 * abs(a), losses


def test_command_line_trains_saves_and_tests(tmp_path):
    """`python -m shg_vqa_amd.agqa_hgqa` with the reference's flags (agqaHGQA.py:877-1075, the headline configuration of
    BASELINE.json on the synthetic split): one epoch of training with validation and checkpoints, then --test valid,test from
    the saved weights with the per-category report."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--noCaps", "--crossAttnType", "cross", "--taskHGQA", "--fromScratch", "--LossHGPerFrame", "--batchSize", "8",
             "--output", str(tmp_path)]
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "shg_vqa_amd.agqa_hgqa", "--epochs", "1", "--valid", "valid"] + flags,
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Epoch 0: Valid" in r.stdout and "Rel class error" in r.stdout, r.stdout[-2000:]
    for f in ("BEST.pth", "LAST.pth"):
        assert (tmp_path / f).exists()
    r = subprocess.run([sys.executable, "-m", "shg_vqa_amd.agqa_hgqa", "--test", "valid,test", "--load", str(tmp_path / "BEST")] + flags,
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Valid HQ results:" in r.stdout and "Test:" in r.stdout and "overall:" in r.stdout, r.stdout[-2000:]
    assert (tmp_path / "test_predictions.json").exists()
