"""CPU-only tests of the host logic: flags, feature converters, masks, state_dict compatibility with
the reference, the active-parameter rule, and that the product refuses to run without the GPU path."""
import json
import os

import numpy as np
import pytest
import torch


def test_flags_keep_reference_names_and_defaults():
    from shg_vqa_amd.param import hgqa_args, parse_args
    a = parse_args([])
    assert (a.llayers, a.xlayers, a.rlayers, a.dlayers) == (5, 2, 5, 5)
    assert a.num_rel == 8 and a.num_act == 3 and a.num_situations == 16 and a.CLIP_LEN == 16
    assert a.emb_drop_rate == 0.15 and a.decoder_drop_rate == 0.15
    h = hgqa_args()
    assert h.task_hgqa and h.loss_hg_per_frame and h.no_caps and h.cross_attn_type == "cross" and h.from_scratch


def test_block_causal_mask_matches_reference(golden_dir):
    from shg_vqa_amd.entry import generate_rel_target_mask
    g = np.load(os.path.join(golden_dir, "agqa_hgqa_b2.npz"))
    assert np.array_equal(generate_rel_target_mask(16, 8), g["rel_mask"])
    assert np.array_equal(generate_rel_target_mask(16, 3), g["act_mask"])


def test_relation_feature_converter_layout():
    from oracle import shg_ref
    from shg_vqa_amd.entry import convert_relations_to_features, frame_segment_ids
    cfg = shg_ref.Cfg()
    b = shg_ref.synthetic_batch(3, cfg, seed=5, with_feat=False)
    feats = convert_relations_to_features(b["rel_triplets"], num_rel=8, num_situations=16, lengths=b["lengths"],
                                          loss_hg_per_frame=True)
    seg = torch.as_tensor(np.array([f.segment_ids for f in feats]))
    assert torch.equal(seg, b["rel_segment_ids"])
    assert torch.equal(frame_segment_ids(3, 16, 8, "cpu"), b["rel_segment_ids"])
    flat = [t for f in feats for t in f.targets]
    assert len(flat) == 48 and all(torch.equal(x, y) for x, y in zip(flat, b["rel_targets"]))
    from shg_vqa_amd.matcher import pad_frame_targets
    tgt, lens = pad_frame_targets([{"labels": f.targets} for f in feats], 8, "cpu")
    assert torch.equal(tgt, b["rel_triplets"].view(-1, 8)) and torch.equal(lens.long(), b["lengths"].view(-1))


def test_sentence_converter_pads_like_the_reference():
    from shg_vqa_amd.entry import HashTokenizer, convert_sents_to_features
    f = convert_sents_to_features(["What did they do before opening the door?", "x " * 100], 40, HashTokenizer())
    assert len(f[0].input_ids) == 40 and f[0].input_ids[0] == 101
    n = sum(f[0].input_mask)
    assert f[0].input_ids[n - 1] == 102 and all(v == 0 for v in f[0].input_ids[n:])
    assert sum(f[1].input_mask) == 40 and f[1].input_ids[39] == 102       # truncated to 38 tokens + CLS/SEP


@pytest.fixture(scope="module")
def cpu_model():
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd.param import hgqa_args
    return AGQAModel(171, num_queries=128, num_actions=157, args=hgqa_args())


def test_state_dict_matches_reference_spec(cpu_model, golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "agqa_state_dict_spec.json")))
    sd = cpu_model.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec["state_dict"]]
    for k, shape, dt in spec["state_dict"]:
        assert list(sd[k].shape) == shape and str(sd[k].dtype).replace("torch.", "") == dt, k
    assert [n for n, _ in cpu_model.named_parameters()] == [k for k, _ in spec["parameters"]]
    # storage aliases (shared x-layer, pooler) are the reference's
    ptr = {}
    mine = {}
    for k, v in sd.items():
        key = (v.data_ptr(), tuple(v.shape))
        if key in ptr:
            mine[k] = ptr[key]
        else:
            ptr[key] = k
    assert mine == spec["aliases"]


def test_active_parameters_are_exactly_those_the_reference_gives_gradients(cpu_model, golden_dir):
    g = np.load(os.path.join(golden_dir, "agqa_hgqa_b2.npz"))
    assert cpu_model.active_parameter_names() == set(str(x) for x in g["grad_names"])


def test_hgdecoder_signature_and_keys(cpu_model):
    import inspect
    from shg_vqa_amd.agqa_model import HGDecoder
    assert list(inspect.signature(HGDecoder.forward).parameters)[1:] == ["memory", "rel_segment_ids", "act_segment_ids"]
    keys = set(cpu_model.hg_decoder.state_dict().keys())
    for k in ("rel_decoder.layers.0.self_attn.in_proj_weight", "action_decoder.layers.4.norm3.bias",
              "relation_query_embed.word_embeddings.weight", "class_embed.3.bias", "action_embed.0.weight"):
        assert k in keys


def test_forward_signatures_match_reference():
    import inspect
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd.entry import LXRTEncoder
    from shg_vqa_amd.matcher import HungarianMatcher
    from shg_vqa_amd.modeling import CrossEncoder
    assert list(inspect.signature(LXRTEncoder.forward).parameters) == ["self", "sents", "feats", "visual_attention_mask"]
    assert list(inspect.signature(AGQAModel.forward).parameters) == [
        "self", "feat", "pos", "input_ids", "input_masks", "segment_ids", "rel_segment_ids", "rel_tgt_mask",
        "act_segment_ids", "act_tgt_mask", "hg_mask", "rel_tgt_ids", "act_tgt_ids"]
    assert list(inspect.signature(CrossEncoder.forward).parameters) == [
        "self", "lang_feats", "lang_attention_mask", "hg_feats", "hg_attention_mask", "output_all_attention_masks"]
    assert list(inspect.signature(HungarianMatcher.__init__).parameters) == ["self", "cost_class", "loss_hg_per_frame", "clip_len"]


def test_no_cpu_fallback():
    """The product path must fail loudly when it is not on the GPU."""
    from shg_vqa_amd import kernels as K
    with pytest.raises(ValueError):
        K.bias_act_fwd(torch.zeros(4, 8), None, 0)
    with pytest.raises(ValueError):
        K.hungarian_per_frame(torch.zeros(2, 8, 10), torch.zeros(2, 8, dtype=torch.int64), torch.zeros(2, dtype=torch.int32))


def test_feature_cache_round_trip_and_prefetch_order(tmp_path):
    """cached slow_r50-shaped features (bf16, channels-last) and the double-buffered loader."""
    import torch
    from shg_vqa_amd.feature_cache import FeatureCache, PrefetchLoader, write_feature_cache
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(64, 6, 3, 3, generator=g) for _ in range(9)]
    prefix = str(tmp_path / "clips")
    assert write_feature_cache(prefix, feats, ids=["vid%d" % i for i in range(9)]) == 9
    cache = FeatureCache(prefix)
    assert len(cache) == 9 and cache.shape == (6, 3, 3, 64) and cache.index["vid7"] == 7
    for i in (0, 4, 8):
        assert torch.equal(cache[i], feats[i].permute(1, 2, 3, 0).to(torch.bfloat16))
    batches = [[0, 1, 2, 3], [8, 7], [4, 4, 5, 6], [2]]
    got = [b.clone() for b in PrefetchLoader(cache, batches, device="cpu")]
    assert [x.shape[0] for x in got] == [4, 2, 4, 1]
    for b, x in zip(batches, got):
        for j, i in enumerate(b):
            assert torch.equal(x[j], cache[i])
    # a truncated file is rejected
    with open(prefix + ".feat", "ab") as f:
        f.write(b"\0\0")
    import pytest
    with pytest.raises(ValueError):
        FeatureCache(prefix)


def test_clip_targets_packing_matches_reference_converter():
    """Per-clip targets: the reference flattens the unpadded per-frame targets (entry.py:87-89)."""
    import torch
    from shg_vqa_amd.entry import clip_targets_device, convert_relations_to_features
    g = torch.Generator().manual_seed(3)
    B, T, per = 5, 16, 8
    lens = torch.randint(0, per + 1, (B, T), generator=g)
    trip = torch.randint(1, 456, (B, T, per), generator=g) * (torch.arange(per).view(1, 1, per) < lens.view(B, T, 1))
    tgt, n = clip_targets_device(trip, lens)
    feats = convert_relations_to_features(trip, per, T, lens, loss_hg_per_frame=False)
    for i, f in enumerate(feats):
        assert int(n[i]) == len(f.targets)
        assert tgt[i, :len(f.targets)].tolist() == list(f.targets)
        assert (tgt[i, len(f.targets):] == 0).all()


def test_streamk_plan_covers_every_k_tile_exactly_once():
    """The stream-K split of the conv forward (gemm.hip: streamk_plan, evaluated on the host through shg_streamk_plan): for the
    tile counts and K lengths of the step's convolutions and a sweep around them, heads and tails together cover every K-tile of
    every output tile exactly once; every tile has exactly one owner, and the owner expects exactly the parts that tail
    workgroups publish, each to its own slot; tails have lower block indices than every head of their XCD (the no-deadlock
    argument); all 256 workgroups have work and none more than 7 % above the even share."""
    import ctypes
    from shg_vqa_amd import _lib
    lib = _lib.lib()
    out = (ctypes.c_int * 6)()
    # heads get sigma % of an even share (tail workgroups read their operand panels alone and are slower; gemm.hip: streamk_sigma)
    sigma = min(200, max(100, int(os.environ.get("SHG_STREAMK_SIGMA", "112"))))

    def plan(n_tiles, nk, block, seg):
        rc = lib.shg_streamk_plan(n_tiles, nk, block, seg, out)
        assert rc in (0, 1)
        return tuple(out) if rc else None

    # (the last two sit at the launcher's bound tiles * nk < 2^23, where the 32-bit plan arithmetic is largest: gemm.hip launch8)
    for n_tiles, nk in [(222, 1440), (147, 540), (132, 180), (128, 64), (255, 97), (200, 333), (129, 1000), (248, 75),
                        (128, 65535), (255, 32896)]:
        assert n_tiles * nk < 2 ** 23
        r_min, r_max = n_tiles // 8, (n_tiles + 7) // 8
        den = 100 * (32 - r_max) + sigma * r_max
        per_wg = (sigma * r_max * nk + den - 1) // den         # head length on the fullest XCD
        if not (r_min >= 16 and r_max < 32 and per_wg >= 64 and nk - per_wg >= 8):
            continue                                            # (launch8 keeps the one-tile-per-workgroup launch there)
        cover = [[] for _ in range(n_tiles)]                     # per tile: the [kb, kb + n) ranges handed out
        owners, expected_parts, published, work = {}, {}, {}, []
        first_head = {}
        for block in range(256):
            xcd, total = block & 7, 0
            for seg in range(64):
                d = plan(n_tiles, nk, block, seg)
                if d is None:
                    break
                tile, kb, n, owner, slot, parts = d
                assert 0 <= tile < n_tiles and n > 0 and 0 <= kb and kb + n <= nk
                cover[tile].append((kb, kb + n))
                total += n
                if owner:
                    assert tile not in owners and kb == 0 and seg == 0
                    owners[tile] = block
                    expected_parts[tile] = parts
                    first_head.setdefault(xcd, block)
                else:
                    assert slot in (2 * tile, 2 * tile + 1) and slot not in published and slot < 512
                    published[slot] = block
                    assert xcd not in first_head, "a tail workgroup after a head of its XCD"
            work.append(total)
        for tile, ranges in enumerate(cover):                   # exact cover of [0, nk): sorted ranges abut
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == nk, (n_tiles, nk, tile, ranges)
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), (n_tiles, nk, tile, ranges)
        assert sorted(owners) == list(range(n_tiles))
        for tile, parts in expected_parts.items():
            got = [s for s in (2 * tile, 2 * tile + 1) if s in published]
            assert got == [2 * tile + p for p in range(parts)], (tile, parts, got)
            for s in got:                                      # same XCD, lower block index than the owner
                assert published[s] & 7 == owners[tile] & 7 and published[s] < owners[tile]
        busy = [w for w in work if w]
        assert len(busy) == 256 and max(busy) <= 1.07 * sigma / 100 * n_tiles * nk / 256, (n_tiles, nk, min(busy), max(busy))


def test_weighted_streamk_plan_covers_tiles_of_different_length():
    """The weighted stream-K split (gemm.hip: streamk_plan_w / streamk_w_build through shg_streamk_plan_weighted): the conv forwards in
    position-major row order, whose tiles keep 20 / 30 / 40 / 45 of the 45 taps.  Same checks as for the uniform plan: exact cover of
    every tile's own K range, one owner per tile expecting exactly the published parts (at most three, slots 3 tile + part), tails
    before heads in every XCD (the no-deadlock argument), and no workgroup far above the even share."""
    import ctypes
    import math
    from shg_vqa_amd import _lib
    lib = _lib.lib()
    out = (ctypes.c_int * 6)()
    sigma = min(200, max(100, int(os.environ.get("SHG_STREAMK_SIGMA", "112"))))

    def valid9(p, H=7, W=7):
        h, w = divmod(p, W)
        return {(kh, kw) for kh in range(3) for kw in range(3) if 0 <= h + kh - 1 < H and 0 <= w + kw - 1 < W}

    def tile_lengths(M, rpp, cin, gn, m_fastest):
        gm = math.ceil(M / 256)
        per_m = []
        for bm in range(gm):
            taps = set()
            for p in range(bm * 256 // rpp, min(M - 1, bm * 256 + 255) // rpp + 1):
                taps |= valid9(p)
            per_m.append(5 * len(taps) * (cin // 64))
        return [per_m[t % gm] if m_fastest else per_m[t // gn] for t in range(gm * gn)]

    cases = [(tile_lengths(18816, 384, 2048, 3, False), 1.13), (tile_lengths(18816, 384, 2048, 3, True), 1.20),   # conv1: 222 tiles, 640 .. 1 440
             (tile_lengths(12544, 256, 768, 3, False), 1.13), (tile_lengths(12544, 256, 768, 3, True), 1.20),    # conv2: 147 tiles, 240 .. 540
             ([100 + (7 * t) % 50 for t in range(200)], 1.10), ([300] * 180, 1.08)]
    for nk, slack in cases:
        n_tiles = len(nk)
        arr = (ctypes.c_uint16 * n_tiles)(*nk)

        def plan(block, seg):
            rc = lib.shg_streamk_plan_weighted(n_tiles, arr, block, seg, out)
            assert rc in (0, 1), rc
            return tuple(out) if rc else None

        cover = [[] for _ in range(n_tiles)]
        owners, expected_parts, published, work, first_head = {}, {}, {}, [], {}
        for block in range(256):
            xcd, total = block & 7, 0
            for seg in range(64):
                d = plan(block, seg)
                if d is None:
                    break
                tile, kb, n, owner, slot, parts = d
                assert 0 <= tile < n_tiles and n > 0 and 0 <= kb and kb + n <= nk[tile], (d, nk[tile])
                cover[tile].append((kb, kb + n))
                total += n
                if owner:
                    assert tile not in owners and kb == 0 and seg == 0 and slot == 3 * tile and 0 <= parts <= 3
                    owners[tile] = block
                    expected_parts[tile] = parts
                    first_head.setdefault(xcd, block)
                else:
                    assert 3 * tile <= slot < 3 * tile + 3 and slot not in published and slot < 768
                    published[slot] = block
                    assert xcd not in first_head, "a tail workgroup after a head of its XCD"
            work.append(total)
        for tile, ranges in enumerate(cover):
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == nk[tile], (tile, ranges, nk[tile])
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), (tile, ranges)
        assert sorted(owners) == list(range(n_tiles))
        for tile, parts in expected_parts.items():
            got = [s for s in range(3 * tile, 3 * tile + 3) if s in published]
            assert got == [3 * tile + p for p in range(parts)], (tile, parts, got)
            for s_ in got:
                assert published[s_] & 7 == owners[tile] & 7 and published[s_] < owners[tile]
        busy = [w for w in work if w]
        print(n_tiles, "workgroups busy", len(busy), "min / max / mean K-tiles", min(busy), max(busy), sum(nk) / 256)
        assert len(busy) >= 250 and max(busy) <= slack * sigma / 100 * sum(nk) / 256, (n_tiles, min(busy), max(busy), sum(nk) / 256)


def test_single_writer_gradient_bookkeeping():
    """Engine.claim_overwrite / settle_stale_grads (host logic of the single-writer gradients): a gradient the optimiser left
    unzeroed is overwritten again, or zeroed before anything accumulates into it or reads it; a second writer in one step poisons the
    fused norm; nothing is claimed on a device without the kernels."""
    import types
    import torch
    from shg_vqa_amd.engine import Engine
    e = Engine(compute_dtype=torch.float32, device="cpu")
    e.grad_arena = torch.ones(64)
    p = types.SimpleNamespace(_shg_off=8, _shg_numel=16, _shg_grad=e.grad_arena[8:24])
    assert e.claim_overwrite(p) is False and not e.overwritten          # (cpu: the accumulate path, always)
    # what the cuda path records, replayed by hand: set in step 1, left unzeroed by the optimiser ...
    e.overwritten = {8: 16}
    e.unzeroed, e.overwritten = dict(e.overwritten), {}
    # ... step 2 does not touch it: stale values must not survive to the norm / the update
    e.settle_stale_grads()
    assert not e.unzeroed and float(e.grad_arena[8:24].abs().sum()) == 0.0 and float(e.grad_arena[:8].sum()) == 8.0
    # ... or step 2 accumulates into it (fallback path): zeroed first
    e.grad_arena.fill_(1.0)
    e.unzeroed = {8: 16}
    assert e.claim_overwrite(p) is False and not e.unzeroed and float(p._shg_grad.abs().sum()) == 0.0
    # a second writer of an already overwritten gradient in the same step: the fused sum no longer describes it
    e.overwritten = {8: 16}
    assert e.claim_overwrite(p) is False and e.overwrite_poisoned
    e.grad_dirty = True
    e.zero_grad()
    assert not e.overwritten and not e.unzeroed and not e.overwrite_poisoned and float(e.grad_arena.abs().sum()) == 0.0


def test_bench_constants_follow_the_survey_flop_table():
    """bench.py prices its fractions with SURVEY section 8(d): 428.21 / 178.79 / 45.50 GFLOP per QA pair (training, forward, the
    attention stack's forward) and conv1 = 83.236 GFLOP per QA pair = 2 x (12 x 49) x 768 x (45 x 2048) flop."""
    import bench
    assert bench.PEAK_BF16_TFLOPS == 2500.0
    assert (bench.TRAIN_GFLOP_PER_QA, bench.FWD_GFLOP_PER_QA, bench.STACK_FWD_GFLOP_PER_QA) == (428.21, 178.79, 45.50)
    conv1 = 2.0 * (12 * 49) * 768 * (45 * 2048)
    assert abs(conv1 / 1e9 - 83.236) < 1e-3
    # per-layer formulas of the same table: BertLayer(S) = S (4 H^2 + 2 H F) + 2 S^2 H MAC with H = 768, F = 3072
    layer = lambda S: 2.0 * (S * (4 * 768 ** 2 + 2 * 768 * 3072) + 2 * S * S * 768) / 1e9
    assert abs(5 * layer(40) - 2.856) < 2e-3 and abs(5 * layer(393) - 30.188) < 2e-3


def test_agqa_evaluator_categories_follow_the_reference_order_and_arithmetic():
    """agqa_data.py:363-700 / :702-883 / :886-1098: per-category accuracy = correct answers of the category / its questions, in
    the order the reference's __main__ prints by position; checked on a hand-countable annotation set."""
    import math
    import types
    from shg_vqa_amd.agqa_eval import ALL_QTYPES, INDIRECT, NOVEL_COMP, AGQAEvaluator
    vocab = {"yes": 0, "no": 1, "cup": 2, "sit": 3}
    D = {
        1: dict(answer="yes", ans_type="binary", **{"global": ["obj-rel", "exists"]}, semantic="object", structural="verify",
                nc_seq=1, nc_sup=0, nc_dur=0, nc_objrel=0, i_obj=1, i_act=0, i_temp=0, indirect=0, direct_equiv=None),
        2: dict(answer="cup", ans_type="open", **{"global": ["obj-rel"]}, semantic="object", structural="query",
                nc_seq=0, nc_sup=0, nc_dur=0, nc_objrel=1, i_obj=1, i_act=1, i_temp=0, indirect=1, direct_equiv=1),
        3: dict(answer="sit", ans_type="open", **{"global": ["action-recognition", "sequencing"]}, semantic="action",
                structural="query", nc_seq=1, nc_sup=0, nc_dur=0, nc_objrel=0, i_obj=0, i_act=1, i_temp=1, indirect=1, direct_equiv=9),
        4: dict(answer="no", ans_type="binary", **{"global": ["superlative"]}, semantic="relation", structural="compare",
                nc_seq=0, nc_sup=1, nc_dur=0, nc_objrel=0, i_obj=0, i_act=0, i_temp=0, indirect=1, direct_equiv=3),
    }
    ev = AGQAEvaluator(types.SimpleNamespace(id2datum=D, answerVocab=vocab))
    pred = {1: 0, 2: 3, 3: 3, 4: 0}                    # correct: 1, 3; wrong: 2, 4
    assert ev.evaluateOverall(pred) == 0.5
    r = dict(zip([n for n, _ in ALL_QTYPES], ev.evaluateAllQtypes(pred)))
    assert len(ALL_QTYPES) == 31 and [n for n, _ in ALL_QTYPES][:4] == ["overall", "binary", "open", "object-relationship"]
    assert r["overall"] == 0.5 and r["binary"] == 0.5 and r["open"] == 0.5
    assert r["object-relationship"] == 0.5 and r["object-relationship binary"] == 1.0 and r["object-relationship open"] == 0.0
    assert r["exists"] == 1.0 and r["sequencing"] == 1.0 and r["sequencing open"] == 1.0 and math.isnan(r["sequencing binary"])
    assert r["superlative"] == 0.0 and r["action-recognition"] == 1.0 and math.isnan(r["relationship-action"])
    assert r["object"] == 0.5 and r["relationship"] == 0.0 and r["action"] == 1.0 and r["query"] == 0.5 and r["verify"] == 1.0
    nc = dict(zip([n for n, _ in NOVEL_COMP], ev.evaluateNovelComp(pred)))
    assert len(NOVEL_COMP) == 15 and nc["sequencing"] == 1.0 and nc["superlative"] == 0.0 and nc["object relationship open"] == 0.0
    assert ev.evaluateCompSteps(pred) == [0.5, 0.5, 0.5]
    recall, pq = ev.evaluateIndirectRef(pred)
    rc = dict(zip([n for n, _ in INDIRECT], recall))
    assert rc["object"] == 0.5 and rc["action"] == 0.5 and rc["localization"] == 1.0
    # precision set: indirect questions whose DIRECT equivalent (in the split) was answered correctly: 2 (<- 1 correct), 4 (<- 3 correct)
    assert sorted(q["answer"] for q in pq) == ["cup", "no"]
    pr = dict(zip([n for n, _ in INDIRECT], ev.evaluatePrecision(pq)))
    assert pr["object"] == 0.0 and pr["action"] == 0.0 and math.isnan(pr["localization"])
    # the ground truth scores 1.0 everywhere a category is populated
    truth = {q: vocab[d["answer"]] for q, d in D.items()}
    assert all(v == 1.0 or math.isnan(v) for v in ev.evaluateAllQtypes(truth))


def test_synthetic_split_carries_evaluator_annotations_and_the_cli_parses_the_reference_flags():
    from shg_vqa_amd.agqa_hgqa import AGQA, get_tuple
    from shg_vqa_amd.param import parse_args
    t = get_tuple("valid", 4, n=12)
    assert AGQA.oracle_score(t) == 1.0
    q2a = {i: t.dataset.answer_index(i) for i in range(12)}
    assert t.evaluator.evaluateAllQtypes(q2a)[0] == 1.0
    it = t.dataset[5]
    assert int(it["target"].argmax()) == t.dataset.answer_index(5)
    a = parse_args(["--taskHGQA", "--noCaps", "--LossHGPerFrame", "--test", "valid,test", "--indirectRef", "--multiGPU", "--load", "x"])
    assert a.indirect_ref and a.multiGPU and a.test == "valid,test" and not a.novel_comp


class _TinyBackbone(torch.nn.Module):
    """Stand-in for VideoBackbone (video_encoder.py): frozen, `encode` maps (B, 3, T, H, W) to (B, C, T, H', W')."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.conv = torch.nn.Conv3d(3, 16, (1, 2, 2), stride=(1, 2, 2))

    def encode(self, x):
        return self.conv(x)


def tiny_backbone():
    return _TinyBackbone()


def test_precompute_features_fills_the_cache_the_loader_reads(tmp_path):
    """precompute_features.py: the frozen backbone once per clip (agqa_model.py:197 runs it in every step) -> the bf16
    channels-last cache of feature_cache.py; also the conversion of already extracted features, and the command line."""
    from shg_vqa_amd import precompute_features as P
    from shg_vqa_amd.feature_cache import FeatureCache
    gen = torch.Generator().manual_seed(4)
    clips = tmp_path / "clips"
    clips.mkdir()
    frames = {}
    for i in range(5):
        frames["v%02d" % i] = torch.randn(3, 4, 8, 8, generator=gen)
        torch.save(frames["v%02d" % i], clips / ("v%02d.pt" % i))
    net = _TinyBackbone()
    assert P.precompute(net, str(clips), str(tmp_path / "cache"), batch_size=2) == 5
    cache = FeatureCache(str(tmp_path / "cache"))
    assert cache.ids == sorted(frames) and cache.shape == (4, 4, 4, 16)
    with torch.no_grad():
        for k, vid in enumerate(cache.ids):
            want = net.encode(frames[vid][None])[0].permute(1, 2, 3, 0).to(torch.bfloat16)
            assert torch.equal(cache[k], want), vid
    feats = tmp_path / "feats"
    feats.mkdir()
    with torch.no_grad():
        np.save(feats / "a.npy", net.encode(frames["v00"][None])[0].numpy())
    assert P.main(["--features", str(feats), "--out", str(tmp_path / "c2")]) == 0
    assert torch.equal(FeatureCache(str(tmp_path / "c2"))[0], cache[0])
    assert P.main(["--backbone", "tests.test_host_cpu:tiny_backbone", "--clips", str(clips), "--out", str(tmp_path / "c3"),
                   "--batch", "3", "--device", "cpu"]) == 0
    assert torch.equal(FeatureCache(str(tmp_path / "c3"))[4], cache[4])


def test_agqa_evaluator_reproduces_the_reference_evaluator_on_2000_questions(golden_dir):
    """tests/golden/evaluator_2k.json (oracle/gen_golden.py evaluator): every result list of the REAL AGQAEvaluator
    (agqa_data.py:350-362, :364-700, :702-733, :737-883, :886-976, :978-1101) on a synthetic annotation set with all categories
    populated and repeated reasoning types - reproduced value for value, by position."""
    import json
    import types
    from shg_vqa_amd.agqa_eval import ALL_QTYPES, COMP_STEPS, INDIRECT, NOVEL_COMP, AGQAEvaluator
    g = json.load(open(os.path.join(golden_dir, "evaluator_2k.json")))
    assert g["n"] == 2000 and len(g["quesid2ans"]) == 2000
    assert any(len(set(d["global"])) < len(d["global"]) for d in g["id2datum"].values())      # per-occurrence counting is exercised
    ev = AGQAEvaluator(types.SimpleNamespace(id2datum=g["id2datum"], answerVocab=g["answer_vocab"]))
    q2a, exp = g["quesid2ans"], g["expected"]
    assert (len(ALL_QTYPES), len(COMP_STEPS), len(NOVEL_COMP), len(INDIRECT)) == (31, 3, 15, 9)
    assert ev.evaluateOverall(q2a) == exp["overall"]
    for got, key in ((ev.evaluateAllQtypes(q2a), "all_qtypes"), (ev.evaluateCompSteps(q2a), "comp_steps"),
                     (ev.evaluateNovelComp(q2a), "novel_comp")):
        assert len(got) == len(exp[key]), key
        assert all(abs(a - b) < 1e-12 for a, b in zip(got, exp[key])), (key, got, exp[key])
    recall, pq = ev.evaluateIndirectRef(q2a)
    assert all(abs(a - b) < 1e-12 for a, b in zip(recall, exp["indirect_recall"]))
    assert [q["question_id"] for q in pq] == exp["precision_ids"]
    prec = ev.evaluatePrecision(pq)
    assert all(abs(a - b) < 1e-12 for a, b in zip(prec, exp["precision"]))
