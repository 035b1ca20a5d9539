"""Golden-vector generator (TEST INFRASTRUCTURE; runs only in the build container).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden all

Imports the REAL reference (oracle/ref_harness.py), gives it the name-derived weights of
oracle/detweights.py, runs it on the seeded synthetic batches of oracle/shg_ref.synthetic_batch
and writes inputs/outputs as small fixtures under tests/golden/.  The reference has no tests
or golden files of its own, so these files are what pins the oracle.
The LSAP fixtures come from the SciPy installed here (the reference's third-party solver).

Each variant needs a fresh interpreter because the reference parses its flags at import
(param.py:201); `all` therefore re-invokes this module once per variant.
"""
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _load_det_weights(torch, model, base_seed=2024):
    from . import detweights
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.copy_(torch.from_numpy(detweights.tensor_for(name, prm.shape, base_seed)))


def _np(t):
    return t.detach().cpu().numpy()


def gen_hgqa(star=False):
    from . import ref_harness, shg_ref
    argv = list(ref_harness.HGQA_ARGV)
    if star:
        argv += ["--useHGMask"]
    R = ref_harness.load(argv)
    torch = R.torch
    torch.manual_seed(0)
    if star:   # STAR head widths, star.py:84-90
        cfg = shg_ref.Cfg(num_answers=4, rel_classes=564, act_classes=112, use_hg_mask=True)
    else:
        cfg = shg_ref.Cfg()
    model = R.agqa_model.AGQAModel(cfg.num_answers, num_queries=cfg.rel_queries, num_classes=cfg.rel_classes - 1,
                                   num_actions=cfg.act_classes - 1)
    model.eval()
    _load_det_weights(torch, model)
    tag = "star" if star else "hgqa"

    if not star:
        sd = model.state_dict()
        ptr2name = {}
        alias = {}
        for k, v in sd.items():
            key = (v.data_ptr(), tuple(v.shape))
            if key in ptr2name:
                alias[k] = ptr2name[key]
            else:
                ptr2name[key] = k
        spec = dict(state_dict=[[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
                    parameters=[[k, list(v.shape)] for k, v in model.named_parameters()],
                    aliases=alias)
        with open(os.path.join(GOLD, "agqa_state_dict_spec.json"), "w") as f:
            json.dump(spec, f)

    bsz = 2
    batch = shg_ref.synthetic_batch(bsz, cfg, seed=1234)
    # host prep exactly as agqaHGQA.py:270-297 does it
    rel_f = R.entry.convert_relations_to_features(batch["rel_triplets"], num_rel=cfg.num_rel,
                                                  num_situations=cfg.num_situations, lengths=batch["lengths"],
                                                  loss_hg_per_frame=True)
    act_f = R.entry.convert_relations_to_features(batch["act_tokens"], num_rel=cfg.num_act,
                                                  num_situations=cfg.num_situations, lengths=batch["act_lengths"],
                                                  loss_hg_per_frame=True)
    rel_seg = torch.as_tensor(np.array([f.segment_ids for f in rel_f]), dtype=torch.long)
    act_seg = torch.as_tensor(np.array([f.segment_ids for f in act_f]), dtype=torch.long)
    assert torch.equal(rel_seg, batch["rel_segment_ids"]) and torch.equal(act_seg, batch["act_segment_ids"])
    tgts = [{"labels": f.targets} for f in rel_f]
    act_tgts = [{"labels": f.targets} for f in act_f]
    rel_mask = torch.as_tensor(R.entry.generate_rel_target_mask(cfg.num_situations, cfg.num_rel))
    act_mask = torch.as_tensor(R.entry.generate_rel_target_mask(cfg.num_situations, cfg.num_act))

    grabbed = {}
    hook = model.lxrt_encoder.register_forward_hook(lambda m, i, o: grabbed.update(pre_x=o[2][-1]))
    logit, rel_logit, act_logit, hg_logit, _ = model(
        batch["feat"], batch["pos"], input_ids=batch["input_ids"], input_masks=batch["input_mask"],
        segment_ids=batch["segment_ids"], rel_segment_ids=rel_seg, rel_tgt_mask=rel_mask,
        act_segment_ids=act_seg, act_tgt_mask=act_mask, hg_mask=batch["hg_mask"])
    hook.remove()
    lang_pre_x, _, memory, _ = grabbed["pre_x"]      # agqa_model.py:218

    H = R.agqa_hgqa
    matcher = R.matcher.HungarianMatcher(cost_class=1, loss_hg_per_frame=True, clip_len=16)
    S = types.SimpleNamespace(background_idx=0, clip_len=16)
    S._get_src_permutation_idx = lambda ind: H.AGQA._get_src_permutation_idx(S, ind)
    w_rel = torch.ones(cfg.rel_classes)
    w_rel[0] = 0.1
    w_act = torch.ones(cfg.act_classes)
    w_act[0] = 0.1
    bce = torch.nn.BCEWithLogitsLoss()(hg_logit, batch["target"]) * hg_logit.size(1)
    idx_r = matcher({"pred_logits": rel_logit}, tgts)
    idx_a = matcher({"pred_logits": act_logit}, act_tgts)
    # vis_utils.accuracy is the stock top-k accuracy helper; the stub cv2 import chain provides it
    lr_ = H.AGQA.loss_labels(S, {"pred_logits": rel_logit}, tgts, idx_r, empty_weight=w_rel, loss_hg_per_frame=True)
    la_ = H.AGQA.loss_labels(S, {"pred_logits": act_logit}, act_tgts, idx_a, empty_weight=w_act, loss_hg_per_frame=True)
    grid_r, _ = H.AGQA.get_target_classes(S, {"pred_logits": rel_logit}, tgts, idx_r, w_rel, log=False, loss_hg_per_frame=True)
    grid_a, _ = H.AGQA.get_target_classes(S, {"pred_logits": act_logit}, act_tgts, idx_a, w_act, log=False, loss_hg_per_frame=True)
    total = bce + lr_["loss_ce"] + la_["loss_ce"]
    model.zero_grad()
    total.backward()
    gnames, gnorms, gheads = [], [], []
    for n, prm in model.named_parameters():
        if prm.grad is not None:
            gnames.append(n)
            gnorms.append(float(prm.grad.double().norm()))
            gheads.append(_np(prm.grad.reshape(-1)[:4]).astype(np.float32))
    total_norm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0))

    def pad_idx(ind, per):
        q = -np.ones((len(ind), per), np.int64)
        t = -np.ones((len(ind), per), np.int64)
        for n, (a, b) in enumerate(ind):
            q[n, : len(a)] = _np(a)
            t[n, : len(b)] = _np(b)
        return q, t

    rq, rt = pad_idx(idx_r, cfg.num_rel)
    aq, at = pad_idx(idx_a, cfg.num_act)
    np.savez_compressed(
        os.path.join(GOLD, f"agqa_{tag}_b2.npz"),
        batch_seed=np.int64(1234), batch_size=np.int64(bsz),
        logit=_np(logit), rel_preds=_np(rel_logit), act_preds=_np(act_logit), hg_logit=_np(hg_logit),
        memory_sl=_np(memory[:, ::8, ::16]), lang_pre_x_sl=_np(lang_pre_x[:, :, ::8]),
        rel_mask=_np(rel_mask), act_mask=_np(act_mask),
        rel_q=rq, rel_t=rt, act_q=aq, act_t=at, rel_grid=_np(grid_r), act_grid=_np(grid_a),
        bce=_np(bce), rel_ce=_np(lr_["loss_ce"]), act_ce=_np(la_["loss_ce"]),
        rel_err=_np(lr_["class_error"]), act_err=_np(la_["class_error"]), total=_np(total),
        grad_names=np.array(gnames), grad_norms=np.array(gnorms, np.float64), grad_heads=np.stack(gheads),
        grad_total_norm=np.float64(total_norm))
    print(tag, "done: total loss", float(total), "grad norm", total_norm, "n grads", len(gnames))


def gen_q():
    """--taskQ (tasks/agqaQ.py:186-300): question-only model; forward logits + BCE * n_answers, the gradient of every parameter
    that trains (norm + first values), the clipped global norm and TWO BertAdam steps as the loop runs them (the first has
    learning rate 0 under the linear warm-up, optimization.py:142-173), with the first values of every updated tensor."""
    from . import ref_harness, shg_ref
    argv = ["ref", "--llayers", "2", "--noCaps", "--batchSize", "4", "--taskQ", "--fromScratch",
            "--optim", "bert", "--lr", "1e-5"]
    R = ref_harness.load(argv)
    torch = R.torch
    cfg = shg_ref.Cfg(llayers=2, task="q")
    model = R.agqa_model.AGQAModel(cfg.num_answers)
    model.eval()
    _load_det_weights(torch, model)
    batch = shg_ref.synthetic_batch(4, cfg, seed=77, with_feat=False)

    def fwd():
        logit, _ = model(None, None, input_ids=batch["input_ids"], input_masks=batch["input_mask"],
                         segment_ids=batch["segment_ids"], rel_segment_ids=None, rel_tgt_mask=None,
                         act_segment_ids=None, act_tgt_mask=None, hg_mask=None)
        return logit, torch.nn.BCEWithLogitsLoss()(logit, batch["target"]) * logit.size(1)

    logit, loss = fwd()
    lr, t_total = 1e-3, 10
    opt = R.optimization.BertAdam(list(model.parameters()), lr=lr, warmup=0.1, t_total=t_total)
    opt.zero_grad()
    loss.backward()
    gnames, gnorms, gheads = [], [], []
    for n, prm in model.named_parameters():
        if prm.grad is not None:
            gnames.append(n)
            gnorms.append(float(prm.grad.double().norm()))
            gheads.append(_np(prm.grad.reshape(-1)[:4]).astype(np.float32))
    total_norm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0))
    opt.step()                                   # step 0: lr = 0
    opt.zero_grad()
    _, loss1 = fwd()
    loss1.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()                                   # step 1: lr = 0.9 * lr
    named = dict(model.named_parameters())
    after = np.stack([_np(named[n].reshape(-1)[:4]).astype(np.float32) for n in gnames])
    _, loss2 = fwd()
    np.savez_compressed(os.path.join(GOLD, "agqa_q_b4.npz"), batch_seed=np.int64(77), batch_size=np.int64(4),
                        logit=_np(logit), loss=_np(loss), grad_names=np.array(gnames), grad_norms=np.array(gnorms, np.float64),
                        grad_heads=np.stack(gheads), grad_total_norm=np.float64(total_norm), lr=np.float64(lr),
                        t_total=np.int64(t_total), after_heads=after, loss_after=_np(loss2),
                        param_names=np.array([n for n, _ in model.named_parameters()]))
    print("q done", float(loss), "grad norm", total_norm, "n grads", len(gnames), "loss after 2 steps", float(loss2))


def gen_vqa():
    """--taskVQA (tasks/agqaVQA.py:237-258): video + question through the 5/2/5 stack, x-layers WITH gradients, pooler_dict.cross
    and the answer head; BCE * n_answers; gradient norms of every parameter that trains."""
    from . import ref_harness, shg_ref
    argv = ["ref", "--llayers", "5", "--xlayers", "2", "--rlayers", "5", "--noCaps", "--crossAttnType", "cross",
            "--batchSize", "2", "--taskVQA", "--fromScratch", "--backbone", "slow_r50", "--optim", "bert", "--lr", "1e-5"]
    R = ref_harness.load(argv)
    torch = R.torch
    cfg = shg_ref.Cfg(task="vqa")
    model = R.agqa_model.AGQAModel(cfg.num_answers)
    model.eval()
    _load_det_weights(torch, model)
    batch = shg_ref.synthetic_batch(2, cfg, seed=4321)
    logit, _ = model(batch["feat"], batch["pos"], input_ids=batch["input_ids"], input_masks=batch["input_mask"],
                     segment_ids=batch["segment_ids"], rel_segment_ids=None, rel_tgt_mask=None, act_segment_ids=None,
                     act_tgt_mask=None, hg_mask=None, rel_tgt_ids=None, act_tgt_ids=None)
    loss = torch.nn.BCEWithLogitsLoss()(logit, batch["target"]) * logit.size(1)
    model.zero_grad()
    loss.backward()
    gnames, gnorms, gheads = [], [], []
    for n, prm in model.named_parameters():
        if prm.grad is not None:
            gnames.append(n)
            gnorms.append(float(prm.grad.double().norm()))
            gheads.append(_np(prm.grad.reshape(-1)[:4]).astype(np.float32))
    total_norm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0))
    np.savez_compressed(os.path.join(GOLD, "agqa_vqa_b2.npz"), batch_seed=np.int64(4321), batch_size=np.int64(2),
                        logit=_np(logit), loss=_np(loss), grad_names=np.array(gnames), grad_norms=np.array(gnorms, np.float64),
                        grad_heads=np.stack(gheads), grad_total_norm=np.float64(total_norm),
                        param_names=np.array([n for n, _ in model.named_parameters()]))
    print("vqa done: loss", float(loss), "grad norm", total_norm, "n grads", len(gnames))


def gen_matcher_frames():
    """Per-frame branch of the REAL reference matcher (lxrt/matcher.py:66-80, --LossHGPerFrame): raw logits -> indices for
    >= 10 000 frames per head.  Logits are multiples of 1/64 stored as int16 (exact); every frame has 0..per labels, with
    duplicated classes in a share of the frames (ties between identical cost columns)."""
    from . import ref_harness
    R = ref_harness.load()
    torch = R.torch
    m = R.matcher.HungarianMatcher(cost_class=1, loss_hg_per_frame=True, clip_len=16)
    g = torch.Generator().manual_seed(20260101)
    out = {}
    for tag, (per, C) in {"rel": (8, 20), "act": (3, 12)}.items():
        n_batches, B, T = 16, 40, 16                       # 16 x 40 x 16 = 10 240 frames
        lg_all, tgt_all, len_all, q_all, t_all = [], [], [], [], []
        for it in range(n_batches):
            k = torch.randint(-256, 257, (B, T * per, C), generator=g)
            if it % 4 == 3:                                  # coarse logits: exact ties between queries of a frame
                k = (k // 64) * 64
            logits = k.float() / 64.0
            lens = torch.randint(0, per + 1, (B * T,), generator=g)
            tgt = torch.zeros(B * T, per, dtype=torch.int64)
            targets = []
            for b in range(B):
                labs = []
                for f in range(T):
                    n = int(lens[b * T + f])
                    lab = torch.randint(1, C, (n,), generator=g)          # duplicates allowed
                    tgt[b * T + f, :n] = lab
                    labs.append(lab)
                targets.append({"labels": labs})
            idx = m({"pred_logits": logits}, targets)
            oq = -torch.ones(B * T, per, dtype=torch.int64)
            ot = -torch.ones(B * T, per, dtype=torch.int64)
            for f, (i, j) in enumerate(idx):
                oq[f, :len(i)], ot[f, :len(j)] = i, j
            lg_all.append(k.to(torch.int16).view(B * T, per, C).numpy())
            tgt_all.append(tgt.numpy().astype(np.int16))
            len_all.append(lens.numpy().astype(np.int8))
            q_all.append(oq.numpy().astype(np.int8))
            t_all.append(ot.numpy().astype(np.int8))
        out.update({tag + "_logits_x64": np.concatenate(lg_all), tag + "_tgt": np.concatenate(tgt_all),
                    tag + "_len": np.concatenate(len_all), tag + "_q": np.concatenate(q_all), tag + "_t": np.concatenate(t_all)})
    np.savez_compressed(os.path.join(GOLD, "matcher_frames.npz"), **out)
    print("matcher_frames done:", {k: v.shape for k, v in out.items()})


def gen_lsap():
    """Known answers + random problems solved by scipy.optimize.linear_sum_assignment."""
    from scipy.optimize import linear_sum_assignment
    import scipy
    rng = np.random.default_rng(20240607)
    costs, shapes, rows, cols = [], [], [], []

    def add(c):
        c = np.asarray(c, dtype=np.float32)
        r, k = linear_sum_assignment(c.astype(np.float64))
        full = np.zeros((8, 8), np.float32)
        full[: c.shape[0], : c.shape[1]] = c
        rr = -np.ones(8, np.int64)
        kk = -np.ones(8, np.int64)
        rr[: len(r)] = r
        kk[: len(k)] = k
        costs.append(full), shapes.append(c.shape), rows.append(rr), cols.append(kk)

    add(np.zeros((8, 0)))
    add(np.full((8, 3), -0.25))
    add([[-.5, -.5], [-.5, -.5], [-.1, -.9]])
    for nr in (8, 3):
        for _ in range(700):                                   # generic fp32 costs like -softmax
            n = int(rng.integers(0, nr + 1))
            add(-rng.random((nr, n), dtype=np.float32))
        for _ in range(500):                                   # heavy ties: few distinct values
            n = int(rng.integers(0, nr + 1))
            add(-rng.integers(0, 3, (nr, n)).astype(np.float32) / 4)
        for _ in range(300):                                   # duplicated target class => identical columns
            n = int(rng.integers(2, nr + 1))
            base = -rng.random((nr, n), dtype=np.float32)
            base[:, int(rng.integers(1, n))] = base[:, 0]
            add(base)
    for _ in range(600):                                       # general 1..8 x 1..8 (both orientations)
        add(rng.normal(size=(int(rng.integers(1, 9)), int(rng.integers(1, 9)))).astype(np.float32))
    np.savez_compressed(os.path.join(GOLD, "lsap_scipy.npz"), cost=np.stack(costs), shape=np.array(shapes, np.int64),
                        rows=np.stack(rows), cols=np.stack(cols), scipy_version=np.array(scipy.__version__))
    print("lsap done:", len(costs), "problems, scipy", scipy.__version__)


def gen_matcher_clip():
    """Per-clip branch of the REAL reference matcher (lxrt/matcher.py:82-104, loss_hg_per_frame=False): one
    assignment problem per sample, num_queries x (number of labels in the clip)."""
    from . import ref_harness
    R = ref_harness.load()
    torch = R.torch
    m = R.matcher.HungarianMatcher(cost_class=1, loss_hg_per_frame=False)
    g = torch.Generator().manual_seed(77)
    out = {}
    for tag, (B, Q, C) in {"rel": (6, 128, 96), "act": (6, 48, 40), "ties": (4, 128, 24)}.items():
        logits = torch.randn(B, Q, C, generator=g)
        if tag == "ties":                                   # few distinct logit values: many exact ties in the cost
            logits = torch.round(logits * 2) / 2
        lens = [0, Q, 1] + [int(torch.randint(2, Q, (1,), generator=g)) for _ in range(B - 3)]
        tgt = -torch.ones(B, Q, dtype=torch.int64)
        targets = []
        for b in range(B):
            lab = torch.randint(1, C, (lens[b],), generator=g)      # duplicates are the norm
            tgt[b, :lens[b]] = lab
            targets.append({"labels": lab})
        idx = m({"pred_logits": logits}, targets)
        oq, ot = -torch.ones(B, Q, dtype=torch.int64), -torch.ones(B, Q, dtype=torch.int64)
        for b, (i, j) in enumerate(idx):
            oq[b, :len(i)], ot[b, :len(j)] = i, j
        out.update({tag + "_logits": logits.numpy(), tag + "_tgt": tgt.numpy(), tag + "_len": np.array(lens, np.int32),
                    tag + "_q": oq.numpy(), tag + "_t": ot.numpy()})
    np.savez_compressed(os.path.join(GOLD, "matcher_clip.npz"), **out)
    print("matcher_clip done")


def gen_bertadam():
    """Four BertAdam steps (the first has lr 0) on three small tensors, with clipping."""
    from . import ref_harness
    R = ref_harness.load()
    torch = R.torch
    g = torch.Generator().manual_seed(5)
    params = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.05) for s in ((7, 5), (11,), (3, 4, 2))]
    init = [_np(p).copy() for p in params]
    opt = R.optimization.BertAdam(params, lr=1e-3, warmup=0.1, t_total=20)
    grads, after, norms = [], [], []
    for step in range(4):
        gs = [torch.randn(p.shape, generator=g) * (3.0 if step == 2 else 0.1) for p in params]
        for p, gg in zip(params, gs):
            p.grad = gg.clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 5.0)))
        opt.step()
        grads.append([_np(x) for x in gs])
        after.append([_np(p).copy() for p in params])
    out = dict(norms=np.array(norms))
    for i in range(3):
        out[f"init{i}"] = init[i]
        for s in range(4):
            out[f"grad{s}_{i}"] = grads[s][i]
            out[f"after{s}_{i}"] = after[s][i]
    np.savez_compressed(os.path.join(GOLD, "bertadam_steps.npz"), **out)
    print("bertadam done", norms)


def gen_matcher_frames_wide():
    """The per-frame branch of the REAL reference matcher (lxrt/matcher.py:62-80) at the class widths the model calls it with
    (C = 457 relation classes x 8 queries per frame, C = 158 action classes x 3): 2 560 frames per head.  The logits are NOT
    stored (2 560 x 8 x 457 values): they are multiples of 1/64 drawn from numpy's frozen legacy generator
    (RandomState(seed).randint), regenerated by the tests and pinned here by a CRC of their bytes."""
    import zlib
    from . import ref_harness
    R = ref_harness.load()
    torch = R.torch
    m = R.matcher.HungarianMatcher(cost_class=1, loss_hg_per_frame=True, clip_len=16)
    out = {}
    for tag, (per, C, seed) in {"rel": (8, 457, 457001), "act": (3, 158, 158001)}.items():
        B, T = 160, 16                                        # 2 560 frames
        rs = np.random.RandomState(seed)
        k = rs.randint(-256, 257, size=(B, T * per, C)).astype(np.int16)
        k[B // 2:] = (k[B // 2:] // 32) * 32                  # second half: coarse logits (17 levels) -> exact cost ties
        lens = rs.randint(0, per + 1, size=(B * T,)).astype(np.int64)
        tgt = np.zeros((B * T, per), np.int64)
        for f in range(B * T):
            n = int(lens[f])
            lab = rs.randint(1, C, size=(n,))
            if n >= 2 and f % 3 == 0:
                lab[1] = lab[0]                               # duplicated class: identical cost columns
            tgt[f, :n] = lab
        logits = torch.from_numpy(k.astype(np.float32) / 64.0)
        targets = [{"labels": [torch.from_numpy(tgt[b * T + f, :int(lens[b * T + f])]) for f in range(T)]} for b in range(B)]
        idx = m({"pred_logits": logits}, targets)
        oq = -np.ones((B * T, per), np.int8)
        ot = -np.ones((B * T, per), np.int8)
        for f, (i, j) in enumerate(idx):
            oq[f, :len(i)], ot[f, :len(j)] = i.numpy(), j.numpy()
        out.update({tag + "_seed": np.int64(seed), tag + "_shape": np.array([B, T, per, C], np.int64),
                    tag + "_crc": np.int64(zlib.crc32(k.tobytes())), tag + "_tgt": tgt.astype(np.int16),
                    tag + "_len": lens.astype(np.int8), tag + "_q": oq, tag + "_t": ot})
    np.savez_compressed(os.path.join(GOLD, "matcher_frames_wide.npz"), **out)
    print("matcher_frames_wide done:", {k: getattr(v, "shape", v) for k, v in out.items()})


def synthetic_annotations(n=2000, seed=31):
    """A synthetic AGQA annotation set with every category of agqa_data.py:341-1101 populated: -> (id2datum, answerVocab)."""
    rs = np.random.RandomState(seed)
    vocab = {("ans%d" % i): i for i in range(40)}
    vocab["yes"], vocab["no"] = 40, 41
    reasoning = ["obj-rel", "rel-act", "obj-act", "superlative", "sequencing", "exists", "duration-comparison", "action-recognition"]
    id2datum = {}
    ids = ["q%05d" % i for i in range(n)]
    for i, qid in enumerate(ids):
        binary = bool(rs.randint(0, 2))
        k = int(rs.randint(1, 4))
        glob = [reasoning[int(j)] for j in rs.randint(0, len(reasoning), size=k)]       # repeats happen (counted per occurrence)
        d = {"question_id": qid, "question": "synthetic %d" % i, "ans_type": "binary" if binary else "open",
             "answer": ("yes" if rs.randint(0, 2) else "no") if binary else "ans%d" % int(rs.randint(0, 40)),
             "global": glob, "semantic": ["object", "relation", "action"][int(rs.randint(0, 3))],
             "structural": ["query", "compare", "choose", "logic", "verify"][int(rs.randint(0, 5))],
             "nc_seq": int(rs.randint(0, 2)), "nc_sup": int(rs.randint(0, 2)), "nc_dur": int(rs.randint(0, 2)),
             "nc_objrel": int(rs.randint(0, 2)), "i_obj": int(rs.randint(0, 2)), "i_act": int(rs.randint(0, 2)),
             "i_temp": int(rs.randint(0, 2)), "indirect": int(rs.randint(0, 2)), "direct_equiv": None}
        if d["indirect"] and i > 0 and rs.randint(0, 4) != 0:
            d["direct_equiv"] = ids[int(rs.randint(0, i))]
        elif d["indirect"] and rs.randint(0, 2):
            d["direct_equiv"] = "absent%05d" % i              # an equivalent that is not part of the split
        id2datum[qid] = d
    return id2datum, vocab


def gen_evaluator():
    """All result lists of the REAL AGQAEvaluator (tasks/agqa_data.py:341-1101) on a 2 000-question synthetic annotation set with
    every category populated and predictions that are right about half of the time."""
    from . import ref_harness
    ref_harness.load()
    import src.tasks.agqa_data as ad
    id2datum, vocab = synthetic_annotations()
    ds = types.SimpleNamespace(id2datum=id2datum, answerVocab=vocab)
    ev = ad.AGQAEvaluator(ds)
    rs = np.random.RandomState(7)
    q2a = {}
    for qid, d in id2datum.items():
        right = rs.randint(0, 2)
        q2a[qid] = int(vocab[d["answer"]]) if right else int(rs.randint(0, len(vocab)))
    res = {"overall": ev.evaluateOverall(q2a), "all_qtypes": ev.evaluateAllQtypes(q2a), "comp_steps": ev.evaluateCompSteps(q2a),
           "novel_comp": ev.evaluateNovelComp(q2a)}
    recall, precision_qs = ev.evaluateIndirectRef(q2a)
    res["indirect_recall"] = recall
    res["precision_ids"] = [q["question_id"] for q in precision_qs]
    res["precision"] = ev.evaluatePrecision(precision_qs)
    for d in id2datum.values():
        d.pop("prediction", None)                             # (the reference writes its prediction into the annotation)
    with open(os.path.join(GOLD, "evaluator_2k.json"), "w") as f:
        json.dump({"n": len(id2datum), "annotation_seed": 31, "answer_vocab": vocab, "id2datum": id2datum,
                   "quesid2ans": q2a, "expected": res}, f, separators=(",", ":"))
    print("evaluator done:", {k: (len(v) if isinstance(v, list) else v) for k, v in res.items()})


def main():
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "all":
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for v in ("hgqa", "star", "q", "vqa", "lsap", "bertadam", "matcher_clip", "matcher_frames", "matcher_frames_wide", "evaluator"):
            subprocess.check_call([sys.executable, "-m", "oracle.gen_golden", v], env=env,
                                  cwd=os.path.dirname(HERE))
    elif what == "hgqa":
        gen_hgqa(False)
    elif what == "star":
        gen_hgqa(True)
    elif what == "q":
        gen_q()
    elif what == "lsap":
        gen_lsap()
    elif what == "matcher_clip":
        gen_matcher_clip()
    elif what == "bertadam":
        gen_bertadam()
    elif what == "vqa":
        gen_vqa()
    elif what == "matcher_frames":
        gen_matcher_frames()
    elif what == "matcher_frames_wide":
        gen_matcher_frames_wide()
    elif what == "evaluator":
        gen_evaluator()
    else:
        raise SystemExit(f"unknown variant {what}")


if __name__ == "__main__":
    main()
