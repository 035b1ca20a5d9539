"""Pins the CPU oracle (oracle/shg_ref.py, oracle/lsap.c) against golden vectors produced by the
real reference in the build container (oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import shg_ref


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err}"


def test_lsap_c_and_py_match_scipy_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lsap_scipy.npz"))
    n = g["cost"].shape[0]
    assert n > 3000
    for i in range(n):
        nr, nc = g["shape"][i]
        c = g["cost"][i, :nr, :nc]
        k = min(nr, nc)
        r, cc = shg_ref.lsap_c(c)
        assert np.array_equal(r, g["rows"][i, :k]) and np.array_equal(cc, g["cols"][i, :k]), i
        if i % 7 == 0:
            r2, c2 = shg_ref.lsap_py(c)
            assert np.array_equal(r2, r) and np.array_equal(c2, cc), i


def test_lsap_known_answers():
    r, c = shg_ref.lsap_c(np.zeros((8, 0)))
    assert len(r) == 0 and len(c) == 0
    r, c = shg_ref.lsap_c(np.full((8, 3), -0.25))
    assert r.tolist() == [0, 1, 2] and c.tolist() == [0, 1, 2]
    r, c = shg_ref.lsap_c(np.array([[-.5, -.5], [-.5, -.5], [-.1, -.9]]))
    assert r.tolist() == [0, 2] and c.tolist() == [0, 1]


def test_param_spec_is_subset_of_reference_parameters(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "agqa_state_dict_spec.json")))
    ref = {k: tuple(s) for k, s in spec["parameters"]}
    mine = shg_ref.param_spec(shg_ref.Cfg())
    for k, s in mine:
        assert ref[k] == tuple(s), k
    assert len(spec["state_dict"]) == 669 and len(ref) == 576


def test_masks_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "agqa_hgqa_b2.npz"))
    assert np.array_equal(shg_ref.frame_causal_mask(16, 8).numpy(), g["rel_mask"])
    assert np.array_equal(shg_ref.frame_causal_mask(16, 3).numpy(), g["act_mask"])


@pytest.mark.parametrize("tag", ["hgqa", "star"])
def test_full_forward_loss_backward_match_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"agqa_{tag}_b2.npz"))
    cfg = shg_ref.Cfg() if tag == "hgqa" else shg_ref.Cfg(num_answers=4, rel_classes=564, act_classes=112,
                                                           use_hg_mask=True)
    torch.manual_seed(0)
    p = shg_ref.det_params(cfg, requires_grad=True)
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]))
    out = shg_ref.agqa_forward(p, cfg, batch)
    _close(out["logit"].detach(), g["logit"], 1e-4, 2e-5)
    _close(out["hg_logit"].detach(), g["hg_logit"], 1e-4, 2e-5)
    _close(out["rel_preds"].detach(), g["rel_preds"], 1e-4, 2e-5)
    _close(out["act_preds"].detach(), g["act_preds"], 1e-4, 2e-5)
    _close(out["memory"].detach()[:, ::8, ::16], g["memory_sl"], 1e-4, 2e-5)
    _close(out["lang_pre_x"].detach()[:, :, ::8], g["lang_pre_x_sl"], 1e-4, 2e-5)
    losses = shg_ref.hgqa_losses(out, batch, cfg)
    for key, per in (("rel", cfg.num_rel), ("act", cfg.num_act)):
        q, t = g[f"{key}_q"], g[f"{key}_t"]
        for n, (qi, ti) in enumerate(losses[f"{key}_idx"]):
            k = len(qi)
            assert np.array_equal(qi.numpy(), q[n, :k]) and np.array_equal(ti.numpy(), t[n, :k]), (key, n)
            assert (q[n, k:] == -1).all()
        assert np.array_equal(losses[f"{key}_grid"].numpy(), g[f"{key}_grid"])
    for k in ("bce", "rel_ce", "act_ce", "total", "rel_err", "act_err"):
        _close(losses[k].detach(), g[k], 1e-4, 1e-4)
    names = list(p)
    grads = torch.autograd.grad(losses["total"], [p[k] for k in names], allow_unused=True)
    got = {k: gr for k, gr in zip(names, grads) if gr is not None}
    ref_names = [str(x) for x in g["grad_names"]]
    assert set(got) == set(ref_names)
    for i, k in enumerate(ref_names):
        n_ref = g["grad_norms"][i]
        n_got = float(got[k].double().norm())
        assert abs(n_got - n_ref) <= 2e-3 * max(n_ref, 1e-6) + 1e-7, (k, n_got, n_ref)
        _close(got[k].reshape(-1)[:4], g["grad_heads"][i], 5e-3, 1e-6 + 1e-4 * n_ref)
    tot = float(torch.sqrt(sum((x.double() ** 2).sum() for x in got.values())))
    assert abs(tot - float(g["grad_total_norm"])) <= 1e-3 * float(g["grad_total_norm"])


def test_q_only_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "agqa_q_b4.npz"))
    cfg = shg_ref.Cfg(llayers=2, task="q")
    p = shg_ref.det_params(cfg)
    assert set(p) <= set(str(x) for x in g["param_names"])
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]), with_feat=False)
    out = shg_ref.agqa_forward(p, cfg, batch)
    _close(out["logit"], g["logit"], 1e-4, 2e-5)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out["logit"], batch["target"]) * cfg.num_answers
    _close(loss, g["loss"], 1e-5, 1e-5)


def test_q_only_backward_and_two_optimiser_steps_match_reference(golden_dir):
    """agqaQ.py:186-300 through the REAL reference (golden) vs the oracle's train_step: every gradient, the clipped norm, two
    BertAdam steps (first values of every updated tensor) and the loss afterwards."""
    g = np.load(os.path.join(golden_dir, "agqa_q_b4.npz"))
    cfg = shg_ref.Cfg(llayers=2, task="q")
    p = shg_ref.det_params(cfg, requires_grad=True)
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]), with_feat=False)
    names = [str(x) for x in g["grad_names"]]
    state = {}
    for step in range(2):
        out, losses, grads, norm = shg_ref.train_step(p, cfg, batch, state, lr=float(g["lr"]), step=step, t_total=int(g["t_total"]))
        if step == 0:
            assert set(grads) == set(names)
            assert abs(float(norm) - float(g["grad_total_norm"])) <= 1e-3 * float(g["grad_total_norm"])
            coef = min(5.0 / (float(norm) + 1e-6), 1.0)               # train_step returns the CLIPPED gradients
            for i, k in enumerate(names):
                n_ref, n_got = g["grad_norms"][i], float(grads[k].double().norm()) / coef
                assert abs(n_got - n_ref) <= 2e-3 * max(n_ref, 1e-6) + 1e-7, (k, n_got, n_ref)
    for i, k in enumerate(names):
        _close(p[k].detach().reshape(-1)[:4], g["after_heads"][i], 2e-4, 2e-6)
    out = shg_ref.agqa_forward(p, cfg, batch)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out["logit"], batch["target"]) * cfg.num_answers
    _close(loss.detach(), g["loss_after"], 1e-3, 1e-3)


def _wide_frames(g, tag):
    """matcher_frames_wide.npz keeps the generator seed instead of the logits (oracle/gen_golden.py matcher_frames_wide)."""
    import zlib
    B, T, per, C = (int(x) for x in g[tag + "_shape"])
    rs = np.random.RandomState(int(g[tag + "_seed"]))
    k = rs.randint(-256, 257, size=(B, T * per, C)).astype(np.int16)
    k[B // 2:] = (k[B // 2:] // 32) * 32
    assert zlib.crc32(k.tobytes()) == int(g[tag + "_crc"]), "numpy's legacy generator no longer reproduces the golden logits"
    return k, B, T, per, C


def test_oracle_per_frame_matching_equals_reference_matcher_at_model_class_widths(golden_dir):
    """matcher.py:62-80 as the model calls it (8 queries x 457 relation classes, 3 x 158 action classes per frame): the REAL
    reference matcher's indices on 2 560 frames per head (golden) vs the oracle, bit-exact."""
    g = np.load(os.path.join(golden_dir, "matcher_frames_wide.npz"))
    for tag in ("rel", "act"):
        k, B, T, per, C = _wide_frames(g, tag)
        n = B * T
        assert n >= 2048
        logits = torch.from_numpy(k.astype(np.float32) / 64.0)
        lens = g[tag + "_len"].astype(np.int64)
        labels = [torch.from_numpy(g[tag + "_tgt"][f, :lens[f]].astype(np.int64)) for f in range(n)]
        idx = shg_ref.hungarian_per_frame(logits, labels, clip_len=16)
        for f, (qi, ti) in enumerate(idx):
            m = int(lens[f])
            assert np.array_equal(qi.numpy(), g[tag + "_q"][f, :m]) and np.array_equal(ti.numpy(), g[tag + "_t"][f, :m]), (tag, f)


def test_bertadam_and_clip_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "bertadam_steps.npz"))
    params = {str(i): torch.from_numpy(g[f"init{i}"].copy()) for i in range(3)}
    state = {}
    for s in range(4):
        grads = {str(i): torch.from_numpy(g[f"grad{s}_{i}"].copy()) for i in range(3)}
        norm = shg_ref.clip_grad_norm(list(grads.values()), 5.0)
        assert abs(float(norm) - g["norms"][s]) < 1e-4 * g["norms"][s]
        shg_ref.bertadam_step(params, grads, state, lr=1e-3, step=s, t_total=20)
        for i in range(3):
            _close(params[str(i)], g[f"after{s}_{i}"], 1e-6, 1e-7)


def test_oracle_per_clip_matching_equals_reference_matcher(golden_dir):
    """matcher.py:82-104 through the REAL reference (golden) vs the oracle's per-clip call (clip_len = 1)."""
    import numpy as np
    import torch
    from oracle import shg_ref
    g = np.load(os.path.join(golden_dir, "matcher_clip.npz"))
    for tag in ("rel", "act", "ties"):
        logits = torch.from_numpy(g[tag + "_logits"])
        lens = g[tag + "_len"]
        labels = [torch.from_numpy(g[tag + "_tgt"][b, :int(lens[b])]) for b in range(logits.shape[0])]
        for solver in (shg_ref.lsap_c, shg_ref.lsap_py):
            if solver is shg_ref.lsap_py and tag != "act":
                continue                                  # the pure-Python solver only on the small problems
            idx = shg_ref.hungarian_per_frame(logits, labels, clip_len=1, solver=solver)
            for b, (qi, ti) in enumerate(idx):
                n = int(lens[b])
                assert np.array_equal(qi.numpy(), g[tag + "_q"][b, :n]) and np.array_equal(ti.numpy(), g[tag + "_t"][b, :n]), (tag, b)


def test_vqa_task_matches_reference(golden_dir):
    """--taskVQA (agqaVQA.py:237-258) through the REAL reference (golden) vs the oracle: answer logits, BCE * n_answers and the
    gradient of every parameter that trains (here the x-layers and pooler_dict.cross do)."""
    g = np.load(os.path.join(golden_dir, "agqa_vqa_b2.npz"))
    cfg = shg_ref.Cfg(task="vqa")
    p = shg_ref.det_params(cfg, requires_grad=True)
    assert set(p) <= set(str(x) for x in g["param_names"])
    batch = shg_ref.synthetic_batch(int(g["batch_size"]), cfg, seed=int(g["batch_seed"]))
    out = shg_ref.agqa_forward(p, cfg, batch)
    _close(out["logit"].detach(), g["logit"], 1e-4, 2e-5)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out["logit"], batch["target"]) * cfg.num_answers
    _close(loss.detach(), g["loss"], 1e-5, 1e-5)
    names = list(p)
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    got = {k: gr for k, gr in zip(names, grads) if gr is not None}
    ref_names = [str(x) for x in g["grad_names"]]
    assert set(got) == set(ref_names)
    for i, k in enumerate(ref_names):
        n_ref = g["grad_norms"][i]
        n_got = float(got[k].double().norm())
        assert abs(n_got - n_ref) <= 2e-3 * max(n_ref, 1e-6) + 1e-7, (k, n_got, n_ref)
    tot = float(torch.sqrt(sum((x.double() ** 2).sum() for x in got.values())))
    assert abs(tot - float(g["grad_total_norm"])) <= 1e-3 * float(g["grad_total_norm"])


def test_oracle_per_frame_matching_equals_reference_matcher_on_10k_frames(golden_dir):
    """matcher.py:66-80 (--LossHGPerFrame) through the REAL reference on raw logits, 10 240 frames per head (golden), vs
    the oracle's per-frame call: indices bit-exact."""
    g = np.load(os.path.join(golden_dir, "matcher_frames.npz"))
    for tag in ("rel", "act"):
        k = g[tag + "_logits_x64"]
        n, per, c = k.shape
        assert n >= 10000
        logits = torch.from_numpy(k.astype(np.float32) / 64.0).view(n // 16, 16 * per, c)
        lens = g[tag + "_len"].astype(np.int64)
        labels = [torch.from_numpy(g[tag + "_tgt"][f, :lens[f]].astype(np.int64)) for f in range(n)]
        idx = shg_ref.hungarian_per_frame(logits, labels, clip_len=16)
        for f, (qi, ti) in enumerate(idx):
            m = int(lens[f])
            assert np.array_equal(qi.numpy(), g[tag + "_q"][f, :m]) and np.array_equal(ti.numpy(), g[tag + "_t"][f, :m]), (tag, f)
            assert (g[tag + "_q"][f, m:] == -1).all()
