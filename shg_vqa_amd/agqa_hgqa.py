"""Training-loop entry points of AGQA/src/tasks/agqaHGQA.py on the HIP path: get_tuple(), class AGQA
with train() / predict() / evaluate() / save() / load(), and the single-step function the benchmark
times.  One optimiser step = host batch hand-over -> forward -> BCE + Hungarian set losses ->
backward -> global-norm clip -> BertAdam, with no device->host synchronisation inside the step.

The AGQA dataset, tokenizer vocabulary and frozen video backbone are not part of the hot path (and
not available offline); get_tuple() therefore serves synthetic AGQA-shaped batches with the batch
tuple layout of agqa_data.py:266.
"""
import collections
import os

import torch
import torch.nn as nn

from . import ops
from .agqa_eval import AGQAEvaluator as Evaluator          # per-category tables, agqa_data.py:341-1146
from .agqa_model import AGQAModel
from .engine import engine
from .entry import clip_targets_device, frame_segment_ids
from .matcher import HungarianMatcher
from .optimization import BertAdam, clip_grad_norm_

DataTuple = collections.namedtuple("DataTuple", "dataset loader evaluator")


class SyntheticAGQA(torch.utils.data.Dataset):
    """AGQA-shaped synthetic samples: slow_r50-shaped features, a tokenised question,
    per-frame relation / action targets with distinct classes, a one-hot answer."""

    num_answers = 171
    action_classes = list(range(157))
    num_situations, num_rel, num_act = 16, 8, 3
    rel_classes = 456

    QTYPES = ("obj-rel", "rel-act", "obj-act", "superlative", "sequencing", "exists", "duration-comparison", "action-recognition")
    SEMANTIC = ("object", "relation", "action")
    STRUCTURAL = ("query", "compare", "choose", "logic", "verify")

    def __init__(self, n=256, seed=1234, feat_pool=8):
        self.n, self.seed = n, seed
        g = torch.Generator().manual_seed(seed)
        self.feat_pool = [torch.randn(2048, 16, 7, 7, generator=g) for _ in range(feat_pool)]
        self.answerVocab = {"answer_%03d" % k: k for k in range(self.num_answers)}       # agqa_data.py:343: answer string -> index
        self._id2datum = None

    def __len__(self):
        return self.n

    def answer_index(self, i):
        return (self.seed * 31 + i * 7919) % self.num_answers

    @property
    def id2datum(self):
        """Annotation records with the fields AGQAEvaluator reads (agqa_data.py:341-1146), a fixed function of the index."""
        if self._id2datum is None:
            d = {}
            for i in range(self.n):
                a = self.answer_index(i)
                d[i] = dict(question_id=i, question="synthetic question %d" % i, answer="answer_%03d" % a,
                            ans_type="binary" if a % 3 == 0 else "open",
                            **{"global": [self.QTYPES[i % 8]] + ([self.QTYPES[(i // 8) % 8]] if i % 5 == 0 else [])},
                            semantic=self.SEMANTIC[i % 3], structural=self.STRUCTURAL[i % 5],
                            nc_seq=int(i % 4 == 0), nc_sup=int(i % 4 == 1), nc_dur=int(i % 4 == 2), nc_objrel=int(i % 4 == 3),
                            i_obj=int(i % 2 == 0), i_act=int(i % 3 == 0), i_temp=int(i % 5 == 0),
                            indirect=int(i % 2 == 1), direct_equiv=(i - 1 if i % 2 == 1 else None))
            self._id2datum = d
        return self._id2datum

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        t = self.num_situations
        n = int(torch.randint(8, 31, (1,), generator=g))
        ids = torch.zeros(40, dtype=torch.int64)
        ids[:n] = torch.randint(1000, 30522, (n,), generator=g)
        ids[0], ids[n - 1] = 101, 102
        mask = torch.zeros(40, dtype=torch.int64)
        mask[:n] = 1

        def ragged(per, ncls):
            tri = torch.zeros(t, per, dtype=torch.int64)
            lens = torch.randint(0, per + 1, (t,), generator=g)
            for f in range(t):
                k = int(lens[f])
                tri[f, :k] = torch.randperm(ncls, generator=g)[:k] + 1
            return tri, lens

        rel, rel_len = ragged(self.num_rel, self.rel_classes)
        act, act_len = ragged(self.num_act, len(self.action_classes))
        target = torch.zeros(self.num_answers)
        target[self.answer_index(i)] = 1.0
        return dict(ques_id=i, feat=self.feat_pool[i % len(self.feat_pool)], pos=torch.ones(393),
                    input_ids=ids, input_mask=mask, segment_ids=torch.zeros(40, dtype=torch.int64),
                    rel_triplets=rel, lengths=rel_len, act_tokens=act, act_lengths=act_len,
                    hg_mask=torch.cat([(act > 0), (rel > 0)], dim=1).float(), target=target)


def get_tuple(splits, bs, shuffle=False, drop_last=False, n=256, seed=1234):
    """agqaHGQA.py:50-63."""
    dset = SyntheticAGQA(n=n, seed=seed + (0 if "train" in splits else 1))
    loader = torch.utils.data.DataLoader(dset, batch_size=bs, shuffle=shuffle, drop_last=drop_last, num_workers=0)
    return DataTuple(dataset=dset, loader=loader, evaluator=Evaluator(dset))


def batch_to_device(batch, device):
    """ONE host->device hand-over per tensor; the ragged per-frame targets travel as the padded
    (B,16,per) + lengths arrays the dataset already holds (the reference makes 2*B*16 tiny copies,
    agqaHGQA.py:304-318)."""
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
    out["lengths"] = out["lengths"].to(torch.int32)
    out["act_lengths"] = out["act_lengths"].to(torch.int32)
    return out


class AGQA:
    def __init__(self, args=None, train_tuple=None, valid_tuple=None, model=None, t_total=None, world=None):
        if args is None:
            from .param import hgqa_args
            args = hgqa_args()
        self.args = args
        self.device = torch.device("cuda")
        self.train_tuple = train_tuple if train_tuple is not None else get_tuple(args.train, args.batch_size, True, True)
        self.valid_tuple = valid_tuple
        dset = self.train_tuple.dataset
        self.num_situations, self.num_rel, self.num_act = dset.num_situations, dset.num_rel, dset.num_act
        self.num_actions = len(dset.action_classes)
        self.rel_classes = getattr(dset, "rel_classes", 456)
        self.clip_len = args.CLIP_LEN
        self.background_idx = 0
        E = engine()
        E.compute_dtype = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32
        self.model = model if model is not None else AGQAModel(dset.num_answers, num_queries=self.num_rel * self.num_situations,
                                                                num_classes=self.rel_classes, num_actions=self.num_actions,
                                                                args=args)
        if E.model is not self.model:
            self.model.to_engine()
        self.empty_weight = torch.ones(self.rel_classes + 1, device=self.device)
        self.empty_weight[self.background_idx] = 0.1
        self.empty_weight_acts = torch.ones(self.num_actions + 1, device=self.device)
        self.empty_weight_acts[self.background_idx] = 0.1
        self.matcher = HungarianMatcher(cost_class=1, loss_hg_per_frame=args.loss_hg_per_frame, clip_len=self.clip_len)
        if t_total is None:
            t_total = int(len(self.train_tuple.loader) * args.epochs)
        self.optim = BertAdam(list(self.model.parameters()), lr=args.lr, warmup=0.1, t_total=t_total)
        self.output = args.output
        self.world = world            # shg_vqa_amd.ddp.GradReducer or None
        if world is not None:
            E.grad_ready_hook = world.on_grad

    # ------------------------------------------------------------------ one step
    def forward_losses(self, b):
        """Forward + losses of agqaHGQA.py:326-378 on a device batch.  Returns a dict of device scalars."""
        a = self.args
        B = b["input_ids"].shape[0]
        dev = b["input_ids"].device
        if a.task_hgqa:
            rel_seg = frame_segment_ids(B, self.num_situations, self.num_rel, dev)
            act_seg = frame_segment_ids(B, self.num_situations, self.num_act, dev)
            logit, rel_logit, act_logit, hg_logit, _ = self.model(
                b["feat"], b["pos"], input_ids=b["input_ids"], input_masks=b["input_mask"], segment_ids=b["segment_ids"],
                rel_segment_ids=rel_seg, act_segment_ids=act_seg, hg_mask=b.get("hg_mask"))
            # the three losses are independent chains of small, latency-bound kernels (the Hungarian solver runs
            # 180 us on a fraction of the chip): the two set losses go to side streams, the BCE stays here
            # (the prediction heads already ran on side stream 1, agqa_model.HGDecoder.forward: no wait for the main
            # stream, which is busy with the hyper-graph cross encoder by now)
            br_r = br_a = ops.Branch(1, rel_logit, act_logit, wait=not a.loss_hg_per_frame)   # (per-clip targets are built here)
            if a.loss_hg_per_frame:                  # one assignment problem per frame (matcher.py:62-80)
                r_tgt, r_len, r_per = b["rel_triplets"].view(-1, self.num_rel), b["lengths"].view(-1), self.num_rel
                a_tgt, a_len, a_per = b["act_tokens"].view(-1, self.num_act), b["act_lengths"].view(-1), self.num_act
            else:                                    # one per clip over all its labels (matcher.py:82-104)
                r_tgt, r_len = clip_targets_device(b["rel_triplets"].view(B, self.num_situations, self.num_rel), b["lengths"])
                a_tgt, a_len = clip_targets_device(b["act_tokens"].view(B, self.num_situations, self.num_act), b["act_lengths"])
                r_per, a_per = rel_logit.shape[1], act_logit.shape[1]
            with br_r:
                rs, rgrid, rq, rt = ops.set_loss(rel_logit, r_tgt, r_len, self.empty_weight, r_per)
            with br_a:
                as_, agrid, aq, at = ops.set_loss(act_logit, a_tgt, a_len, self.empty_weight_acts, a_per)
            bce = ops.bce_with_logits_times_c(hg_logit, b["target"])
            br_r.join(rs, rgrid, rq, rt, as_, agrid, aq, at, rel_logit, act_logit)
            if self.world is not None:
                rs, as_ = self.world.global_loss_sums2(rs, as_)
            # data parallel: CE terms are already global (their gradient sums to the global gradient);
            # the BCE mean is local, so it enters with 1/world (ddp.py).  total = bce * scale + rel CE + act CE
            scale = self.world.bce_scale() if self.world is not None else 1.0
            if os.environ.get("SHG_LOSS_COMBINE", "1") == "0":        # the same arithmetic as ~40 one-thread torch kernels (A/B)
                rel_ce, act_ce = rs[0] / rs[1], as_[0] / as_[1]
                total = bce.sum() * scale + rel_ce + act_ce
                diag = (bce.sum().detach(), rel_ce.detach(), act_ce.detach(), 100.0 - 100.0 * rs[2] / rs[3].clamp(min=1),
                        100.0 - 100.0 * as_[2] / as_[3].clamp(min=1))
            else:
                total, diag = ops.combine_losses(rs, as_, bce, scale)
            return dict(total=total, bce=diag[0], rel_ce=diag[1], act_ce=diag[2], rel_err=diag[3], act_err=diag[4],
                        logit=logit, hg_logit=hg_logit, rel_logit=rel_logit, act_logit=act_logit,
                        rel_idx=(rq, rt), act_idx=(aq, at), rel_grid=rgrid, act_grid=agrid)
        if a.task_vqa:
            logit, _ = self.model(b["feat"], b["pos"], input_ids=b["input_ids"], input_masks=b["input_mask"],
                                  segment_ids=b["segment_ids"])
        else:
            logit, _ = self.model(None, None, input_ids=b["input_ids"], input_masks=b["input_mask"],
                                  segment_ids=b["segment_ids"])
        bce = ops.bce_with_logits_times_c(logit, b["target"])
        return dict(total=bce.sum() * (self.world.bce_scale() if self.world is not None else 1.0), bce=bce.sum().detach(),
                    logit=logit, hg_logit=logit)

    def _conv1_params(self):
        vf = self.model.lxrt_encoder.model.bert.encoder.visn_fc
        return vf.conv[1].weight, vf.conv[1].bias

    def _step_body(self, b):
        """zero_grad -> forward -> losses -> backward -> (gradient exchange) -> clip -> BertAdam.
        Everything here only enqueues device work (no host synchronisation), so it can be captured."""
        E = engine()
        E.begin_step()
        self.optim.zero_grad(set_to_none=True)
        if self.world is not None:
            self.world.begin_step()
        out = self.forward_losses(b)
        out["total"].backward()
        ops.flush_wgrads()                          # weight gradients still queued (Engine.wgrad_batch)
        if self.world is not None:
            self.world.finish()
        out["grad_norm"] = clip_grad_norm_(self.model.parameters(), 5.0)
        self.optim.step()
        return out

    def train_step(self, b, overlap_update=False):
        """agqaHGQA.py:262-392 for one device batch (eager launches).
        overlap_update: BertAdam's sweep over everything but conv1's weight / bias is left running on a side stream and
        overlaps the NEXT step's first convolution (HBM-bound next to a matrix-core-bound kernel).  Parameters are then
        only safe to read after engine().wait_params_ready(), a device synchronisation or the next train_step; the
        training loop (train()) uses it between the steps of an epoch, bench.py inside its synchronised timed region."""
        if not self.model.training:                 # nn.Module.train() walks ~1 900 modules: only on a mode change
            self.model.train()
        E = engine()
        E.training = True
        E.conv1_cache = None
        E.lazy_adam = bool(overlap_update) and os.environ.get("SHG_LAZY_ADAM", "1") != "0"
        try:
            return self._step_body(b)
        finally:
            E.lazy_adam = False

    # ------------------------------------------------------------------ hipGraph execution
    def capture(self, example):
        """Captures one optimiser step into a hipGraph (torch.cuda.CUDAGraph).  The ~1 600 launches of a
        step are then replayed by the GPU front-end without any Python / launch overhead; per-step
        state that changes (dropout step counter, schedule step, batch) lives in device memory.
        The conv1 forward (the dominant kernel) stays OUTSIDE the graph: it is launched eagerly into
        persistent buffers right before the replay, which keeps it individually timeable."""
        from . import ops as _ops
        E = engine()
        self.model.train()
        self._static = {k: v.clone() for k, v in example.items() if torch.is_tensor(v)}
        w1, b1 = self._conv1_params() if not self.args.task_q else (None, None)
        self._conv1_bufs = None
        if w1 is not None:
            self._conv1_bufs = _ops.conv1_forward(self._static["feat"], w1, b1, None)

        def body():
            if w1 is not None:
                E.conv1_cache = self._conv1_bufs
            return self._step_body(self._static)

        # Streams that join the capture, and where each re-joins the capturing (origin) stream O - DESIGN.md section 7 has the
        # full sequence: S2 (language layers, deferred x-layers, pooler, answer head) -> O at ops.join_deferred_branch; S2 again
        # (language <- hyper-graph half of the cross encoder) -> O at CrossLayer's side.join; S1 (action decoder) -> O at
        # HGDecoder's branch.join; S1 again (heads, set losses) -> O at forward_losses' br_r.join; the backward nodes autograd
        # replays on S1 / S2 and every weight gradient on W -> O at Engine.join_side_streams (clip_grad_norm_); nothing is
        # issued on a side stream after that.  The ledger (Engine.unjoined) proves it for THIS body before the capture ends.
        # ... with ONE branch stream: a capture that forks both branch streams ends cleanly, but its replay computes the
        # relation head from a stale decoder output (deterministically; either stream alone replays bit-for-bit like eager
        # steps, tools/graph_capture_ab.py) - so the language branch runs inline in the captured step.
        cap_mask = int(os.environ.get("SHG_CAPTURE_BRANCH_MASK", "2"))
        if (E.branch_mask & cap_mask & 6) == 6:
            raise RuntimeError("AGQA.capture: SHG_CAPTURE_BRANCH_MASK=%d keeps BOTH branch streams in the captured step; such a "
                               "graph replays a stale relation-decoder output (DESIGN.md section 7 (5)) - keep one of bits 1 / 2"
                               % cap_mask)
        mask, E.branch_mask = E.branch_mask, E.branch_mask & cap_mask
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                  # warm-up: caches (masks, gather tables, workspaces), DDP write counts
                    if w1 is not None:
                        _ops.conv1_forward(self._static["feat"], w1, b1, self._conv1_bufs)
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._graph_out = body()
                if E.unjoined:                      # would end in hipErrorStreamCaptureUnjoined (or worse) inside the runtime
                    left = sorted(E.unjoined)
                    E.join_side_streams()           # leave the capture in a state that can be ended, then report
                    raise RuntimeError("AGQA.capture: side streams %s still hold work that the capturing stream never waited "
                                       "for (a Branch / weight-gradient fork without its join)" % left)
        finally:
            E.branch_mask = mask
            E.conv1_cache = None
        return self

    def train_step_graphed(self, b):
        """Copies the batch into the captured buffers, runs conv1 eagerly, replays the rest of the step."""
        from . import ops as _ops
        st = self._static
        for k, v in b.items():
            if torch.is_tensor(v) and k in st:
                st[k].copy_(v, non_blocking=True)
        if self._conv1_bufs is not None:
            w1, b1 = self._conv1_params()
            _ops.conv1_forward(st["feat"], w1, b1, self._conv1_bufs)
        self._graph.replay()
        return self._graph_out

    # ------------------------------------------------------------------ loops (agqaHGQA.py:233-455, :459-630)
    def train(self, train_tuple=None, eval_tuple=None):
        dset, loader, evaluator = train_tuple or self.train_tuple
        best = 0.0
        rank0 = _rank() == 0        # one process per GPU: rank 0 alone prints, validates and writes checkpoints
        for epoch in range(self.args.epochs):
            quesid2ans = {}
            for i, batch in enumerate(loader):
                b = batch_to_device(batch, self.device)
                out = self.train_step(b, overlap_update=True)
                if rank0 and i % self.args.log_freq == 0:
                    msg = "\nEpoch %d: Total loss= %0.4f \tHGQA loss= %0.4f" % (epoch, out["total"].item(), out["bce"].item())
                    if "rel_ce" in out:
                        msg += "\tRel loss= %0.4f \tAct loss= %0.4f\nRel class error= %0.4f \t Act class error= %0.4f" % (
                            out["rel_ce"].item(), out["act_ce"].item(), out["rel_err"].item(), out["act_err"].item())
                    print(msg, flush=True)
                for qid, l in zip(batch["ques_id"].tolist(), out["hg_logit"].argmax(1).cpu().tolist()):
                    quesid2ans[qid] = l
            engine().wait_params_ready()                 # the last step's update may still run on its side stream
            if rank0:                                    # (under data parallelism: accuracy over rank 0's shard of the epoch)
                print("Epoch %d: Train %0.2f" % (epoch, evaluator.evaluateOverall(quesid2ans) * 100.0), flush=True)
            self.save("CURRENT")
            if eval_tuple is not None:
                # every rank holds the same weights after the all-reduced step: rank 0 validates, the others wait in save()'s barrier
                if rank0:
                    score = self.evaluate(eval_tuple)
                    improved = score > best
                    best = max(best, score)
                    print("Epoch %d: Valid %0.2f  Best %0.2f" % (epoch, score * 100.0, best * 100.0), flush=True)
                else:
                    improved = False
                self.save("BEST", only_if=improved)
        self.save("LAST")

    @torch.no_grad()
    def predict(self, eval_tuple, dump=None):
        dset, loader, evaluator = eval_tuple
        self.model.eval()
        engine().wait_params_ready()
        quesid2ans = {}
        for batch in loader:
            b = batch_to_device(batch, self.device)
            engine().begin_step()
            out = self.forward_losses(b)
            for qid, l in zip(batch["ques_id"].tolist(), out["hg_logit"].argmax(1).cpu().tolist()):
                quesid2ans[qid] = l
        return quesid2ans

    @torch.no_grad()
    def test(self, eval_tuple, dump=None):
        """agqaHGQA.py:632-798: target-free inference (segment ids only, no matcher / losses); returns and
        optionally dumps {question id: answer index}."""
        import json
        dset, loader, evaluator = eval_tuple
        self.model.eval()
        quesid2ans = {}
        for batch in loader:
            b = batch_to_device(batch, self.device)
            engine().begin_step()
            B = b["input_ids"].shape[0]
            if self.args.task_hgqa:
                rel_seg = frame_segment_ids(B, self.num_situations, self.num_rel, self.device)
                act_seg = frame_segment_ids(B, self.num_situations, self.num_act, self.device)
                _, _, _, hg_logit, _ = self.model(b["feat"], b["pos"], input_ids=b["input_ids"], input_masks=b["input_mask"],
                                                  segment_ids=b["segment_ids"], rel_segment_ids=rel_seg,
                                                  act_segment_ids=act_seg, hg_mask=b.get("hg_mask"))
            else:
                hg_logit = self.forward_losses(b)["logit"]
            for qid, l in zip(batch["ques_id"].tolist(), hg_logit.argmax(1).cpu().tolist()):
                quesid2ans[qid] = l
        if dump is not None:
            with open(dump, "w") as f:
                json.dump({str(k): int(v) for k, v in quesid2ans.items()}, f)
        return quesid2ans

    def evaluate(self, eval_tuple, dump=None):
        return eval_tuple.evaluator.evaluateOverall(self.predict(eval_tuple, dump))

    def evaluateAllQtypes(self, eval_tuple, dump=None):
        """agqaHGQA.py:808-811: target-free inference, then the 31 per-category accuracies."""
        return eval_tuple.evaluator.evaluateAllQtypes(self.test(eval_tuple, dump))

    def evaluateTestSplits(self, eval_tuple, dump=None):
        """agqaHGQA.py:816-838: the AGQA test splits selected by --indirectRef / --novelComp / --compSteps."""
        ev = eval_tuple.evaluator
        q2a = self.test(eval_tuple, dump)
        if getattr(self.args, "indirect_ref", False):
            recall, precision_qs = ev.evaluateIndirectRef(q2a)
            return ev.evaluateAllQtypes(q2a), recall, ev.evaluatePrecision(precision_qs)
        if getattr(self.args, "novel_comp", False):
            return ev.evaluateNovelComp(q2a)
        if getattr(self.args, "comp_steps", False):
            return ev.evaluateCompSteps(q2a)
        return ev.evaluateAllQtypes(q2a)

    @staticmethod
    def oracle_score(data_tuple):
        """agqaHGQA.py:843-856: the score of the ground-truth labels themselves (1.0 unless the answer vocabulary drops labels)."""
        dset, loader, evaluator = data_tuple
        quesid2ans = {}
        for batch in loader:
            for qid, l in zip(batch["ques_id"].tolist(), batch["target"].argmax(1).tolist()):
                quesid2ans[qid] = int(l)
        return evaluator.evaluate(quesid2ans)

    def save(self, name, only_if=True):
        """agqaHGQA.py:859-862.  One process per GPU: rank 0 alone writes - to a temporary file that is renamed into place, so a
        reader or a crash never sees a torn checkpoint - and every rank leaves through a barrier (all ranks must call save)."""
        engine().wait_params_ready()
        if _rank() == 0 and only_if:
            os.makedirs(self.output, exist_ok=True)
            path = os.path.join(self.output, "%s.pth" % name)
            tmp = "%s.tmp.%d" % (path, os.getpid())
            torch.save({k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}, tmp)
            os.replace(tmp, path)
        _barrier()

    def load(self, path):
        """agqaHGQA.py:864-874: strips DataParallel's `module.` prefix, strict load, refreshes bf16 shadows."""
        engine().wait_params_ready()
        sd = torch.load("%s.pth" % path if not path.endswith(".pth") else path, map_location="cpu")
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.model.load_state_dict(sd, strict=True)
        engine().refresh_shadows()


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


# ------------------------------------------------------------------ command line (agqaHGQA.py:877-1075)
def _report_test(agqa, args, split):
    from .agqa_eval import ALL_QTYPES, COMP_STEPS, INDIRECT, NOVEL_COMP, format_report
    data = get_tuple(split, bs=args.batch_size, shuffle=False, drop_last=False)
    out = os.path.join(args.output, "%s_predictions.json" % split)
    os.makedirs(args.output, exist_ok=True)
    if split == "valid":
        print(format_report("Valid HQ results:", ALL_QTYPES, agqa.evaluateAllQtypes(data, dump=None)), flush=True)
    elif getattr(args, "indirect_ref", False):
        allq, recall, prec = agqa.evaluateTestSplits(data, dump=out)
        print(format_report("\nTest Results:", ALL_QTYPES, allq), flush=True)
        print(format_report("\nTest Indirect References (recall):", INDIRECT, recall), flush=True)
        print(format_report("\nTest Precision:", INDIRECT, prec), flush=True)
    elif getattr(args, "novel_comp", False):
        print(format_report("\nTest Novel Compositions:", NOVEL_COMP, agqa.evaluateTestSplits(data, dump=out)), flush=True)
    elif getattr(args, "comp_steps", False):
        print(format_report("\nTest Compositional Steps:", COMP_STEPS, agqa.evaluateTestSplits(data, dump=out)), flush=True)
    else:
        print(format_report("\nTest:", ALL_QTYPES, agqa.evaluateAllQtypes(data, dump=out)), flush=True)


def main(argv=None):
    """`python -m shg_vqa_amd.agqa_hgqa <flags of param.py>`: train / --test valid,test like the reference's __main__.
    --multiGPU: one process per GPU (the reference wraps the model in nn.DataParallel, agqaHGQA.py:124-129): this process
    re-launches itself under torch.distributed.run BEFORE anything touches the GPU and exits with the children's code."""
    import sys
    from .param import parse_args
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.multiGPU and "LOCAL_RANK" not in os.environ:
        import subprocess
        n = torch.cuda.device_count()              # (counting devices does not initialise the runtime)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(max(n, 1)), "--master-addr",
               "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), "-m", "shg_vqa_amd.agqa_hgqa"] + argv
        return subprocess.call(cmd)
    from .engine import reset_engine
    rank = int(os.environ.get("RANK", "0"))
    multi = "LOCAL_RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dev = torch.device("cuda", torch.cuda.current_device())
    reset_engine(compute_dtype=torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32, device=dev, seed=args.seed + rank)
    if multi:
        from . import ddp
        ddp.init_process_group(dev)                # (binds the engine's streams to the hardware queues before RCCL's)
    torch.manual_seed(args.seed)                   # identical --fromScratch initialisation on every rank
    train = get_tuple(args.train, args.batch_size, shuffle=True, drop_last=True, seed=1234 + rank)
    valid = get_tuple(args.valid, args.batch_size, shuffle=False, drop_last=False) if args.valid else None
    agqa = AGQA(args, train_tuple=train, valid_tuple=valid)
    if multi:
        from .ddp import GradReducer, default_wire_dtype
        agqa.world = GradReducer(engine().grad_arena, grad_dtype=default_wire_dtype())
        engine().grad_ready_hook = agqa.world.on_grad
    if args.load is not None:
        agqa.load(args.load)
    if args.test is not None:
        if rank == 0:                              # inference needs no exchange: one rank reports and dumps the predictions
            for split in ("valid", "test"):
                if split in args.test:
                    _report_test(agqa, args, split)
        _barrier()
        return 0
    if rank == 0:
        print("Splits in Train data:", args.train, "| oracle score of the labels: %0.2f" % (100.0 * AGQA.oracle_score(train)), flush=True)
    agqa.train(train, valid)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
