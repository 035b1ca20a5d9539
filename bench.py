#!/usr/bin/env python3
"""Benchmark of the SHG-VQA hot path on MI355X: full --taskHGQA training steps (5/2/5 layers, 5 decoder
layers, B=32 x 16 frames x 2048-d slow_r50-shaped features, --LossHGPerFrame, bf16 operands / fp32
accumulation and fp32 master weights) on synthetic AGQA-shaped batches that are resident in HBM when
the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL)

One step = forward -> BCE + Hungarian set losses -> backward -> global-norm clip -> BertAdam (nothing
skipped: every result of the step is computed, bit for bit what the unskipped kernels give; the only arithmetic
not executed are the first convolution's weight-gradient products with its zero padding, DESIGN.md section 9 (10),
reported in `roofline_rows`).  Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel (the implicit-GEMM
(5,3,3) Conv3d 2048->768: 46 % of the step's algorithmic FLOPs) timed with events on its own stream
inside the timed steps; `cpu_baseline` is the CPU oracle (oracle/shg_ref.py) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
TRAIN_GFLOP_PER_QA = 428.21        # fwd + required bwd GEMM-like work per QA pair (HGQA)
TRAIN_GFLOP_PER_QA_VQA = 365.40    # same table, --taskVQA (BASELINE.json configs[1]: no HG decoder / no Hungarian; the x-layers train)
FWD_GFLOP_PER_QA = 178.79          # same table: forward only
FWD_GFLOP_PER_QA_VQA = 149.55
STACK_FWD_GFLOP_PER_QA = 45.50     # same table: the attention stack alone (5 l-layers S=40, 5 r-layers S=393, 2 x-layers 40<->393)


def synthetic_device_batches(n_batches, bsz, seed, device):
    from shg_vqa_amd.agqa_hgqa import SyntheticAGQA, batch_to_device
    ds = SyntheticAGQA(n=n_batches * bsz, seed=seed, feat_pool=min(16, n_batches * bsz))
    out = []
    for i in range(n_batches):
        items = [ds[i * bsz + j] for j in range(bsz)]
        batch = {k: (torch.stack([it[k] for it in items]) if torch.is_tensor(items[0][k]) else torch.tensor([it[k] for it in items]))
                 for k in items[0]}
        out.append(batch_to_device(batch, device))
    return out


def cpu_baseline(bsz=4, timed=4, task="hgqa"):
    """The CPU oracle's full train step (same arithmetic, fp32, dropout on) on this box's host cores."""
    from oracle import shg_ref
    # the box's CPU share, not the host's core count (oversubscribing the cgroup makes MKL crawl)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))
    cfg = shg_ref.Cfg() if task == "hgqa" else shg_ref.Cfg(task=task)
    p = shg_ref.det_params(cfg, requires_grad=True)
    state = {}
    times = []
    for step in range(1 + timed):
        batch = shg_ref.synthetic_batch(bsz, cfg, seed=1234 + step)
        t0 = time.perf_counter()
        shg_ref.train_step(p, cfg, batch, state, lr=1e-5, step=step, t_total=1000, train=True)
        times.append(time.perf_counter() - t0)
        log("  oracle step %d: %.1f s" % (step, times[-1]))
    per = sum(times[1:]) / timed
    return {"value": round(bsz / per, 4), "unit": "QA-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/shg_ref.train_step, B=%d slice of the B=32 workload, fp32, dropout on, 1 warm-up + %d timed steps "
                      "(%.1f s/step)" % (bsz, timed, per)}


def forward_only(trainer, batches, iters=10, fwd_gflop=FWD_GFLOP_PER_QA):
    """SURVEY 8(d) "forward-only QA-pairs/s": the predict() pass (eval mode, no autograd graph, same kernels)."""
    from shg_vqa_amd.engine import engine
    E = engine()
    trainer.model.eval()
    E.wait_params_ready()
    times = []
    with torch.no_grad():
        for i in range(2 + iters):
            if i == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            E.begin_step()
            trainer.forward_losses(batches[i % len(batches)])
        torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / iters
    trainer.model.train()
    bsz = batches[0]["input_ids"].shape[0]
    qa = bsz / per
    return {"value": round(qa, 1), "unit": "QA-pairs/s", "ms_per_batch": round(1e3 * per, 3),
            "mfma_frac": round(qa * fwd_gflop * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4)}


def fp32_parity_mode(batch, steps=4):
    """The arithmetic the <= 1e-3 logit parity is proven in (tests/test_model_gpu.py: fp32 operands, exact-fp32 MFMA chains)
    on the SAME workload: full training steps at the benchmark's batch size in a child process (the engine's arenas are per
    process), so that the parity-grade path has a throughput next to the bf16 headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "fp32", "--steps", str(steps), "--warmup", "2", "--batch", str(batch),
           "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:
        return {"error": (res.stderr or "")[-300:]}
    d = json.loads(res.stdout.strip().splitlines()[-1])
    return {"value": d["value"], "unit": "QA-pairs/s", "ms_per_step": d["ms_per_step"], "dtype": "fp32", "steps": steps,
            "note": "fp32 operands / fp32 accumulation (v_mfma_f32_16x16x4_f32 chains): the mode of the <= 1e-3 parity tests"}


def attention_stack(trainer, bsz, iters=8, schedule="serial"):
    """SURVEY 8(d) sub-roofline of the attention stack alone: the encoder's 5 language layers (S = 40), 5 relation layers
    (S = 393) and 2 cross layers (40 <-> 393), forward + backward (input, weight and bias gradients, dropout on) on
    hidden states of the training shape, without the convolutions / decoders / losses around them.
    schedule "model": the layers are issued as NoCapsEncoder.forward issues them in the step - the language layers on the
    engine's branch stream beside the relation layers (their backward is replayed there too), the cross layers after the join;
    "serial": every layer on one stream, one after the other (the headline figure, as in rounds 1-2; measured on MI355X the two
    schedules are within 2 % of each other - 8.60 ms serial, 8.73 ms "model": the relation layers' kernels fill the chip, the
    language layers' small kernels beside them take from them what they save)."""
    from shg_vqa_amd import ops
    from shg_vqa_amd.engine import engine
    from shg_vqa_amd.modeling import additive_mask
    E = engine()
    enc = trainer.model.lxrt_encoder.model.bert.encoder
    dev, cdt = E.device, E.compute_dtype
    gen = torch.Generator(device="cpu").manual_seed(4321)
    lang0 = torch.randn(bsz, 40, 768, generator=gen).to(dev, cdt)
    visn0 = torch.randn(bsz, 393, 768, generator=gen).to(dev, cdt)
    lens = torch.randint(8, 31, (bsz,), generator=gen)
    mask01 = (torch.arange(40)[None, :] < lens[:, None]).long().to(dev)
    lmask = additive_mask(mask01, mask01)
    trainer.model.train()
    E.training = True
    ms = []
    for i in range(2 + iters):
        lang, visn = lang0.clone().requires_grad_(True), visn0.clone().requires_grad_(True)
        E.begin_step()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        l, v = lang, visn
        if schedule == "model":
            branch = ops.Branch(2, lang, lmask)
            with branch:
                for layer in enc.layer:
                    l, _ = layer(l, lmask)
            for layer in enc.r_layers:
                v, _ = layer(v, None)
            branch.join(l)
        else:
            for layer in enc.layer:
                l, _ = layer(l, lmask)
            for layer in enc.r_layers:
                v, _ = layer(v, None)
        for layer in enc.x_layers:
            l, v, _ = layer(l, lmask, v, None)
        torch.autograd.backward([l, v], [torch.ones_like(l), torch.ones_like(v)])
        ops.flush_wgrads()
        E.join_side_streams()
        s1.record()
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(s0.elapsed_time(s1))
    E.grad_dirty = True
    E.zero_grad()
    per = sum(ms) / len(ms)
    tflops = 3.0 * STACK_FWD_GFLOP_PER_QA * 1e9 * bsz / (per * 1e-3) / 1e12
    return {"ms_fwd_bwd": round(per, 3), "achieved": round(tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_BF16_TFLOPS, 4), "schedule": schedule,
            "flop": "3 x %.2f GFLOP per QA pair (forward, SURVEY 8(d)) x %d" % (STACK_FWD_GFLOP_PER_QA, bsz)}


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--task", default="hgqa", choices=["hgqa", "vqa"],
                    help="hgqa: BASELINE.json configs[2] (the headline: full model); vqa: configs[1] (agqaVQA.py --taskVQA 5/2/5, "
                         "video + question path, BCE on the answer logit, no HG decoder / no Hungarian)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (the measured configuration); gloo = host "
                         "collectives, for rehearsing the N > 1 code path of this script on a box with fewer GPUs than ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only and attention-stack measurements (N=1)")
    ap.add_argument("--overlap-update", action="store_true", help="overlap BertAdam's sweep with the next step's conv1")
    ap.add_argument("--force-ddp", action="store_true", help="use the gradient reducer / RCCL path even with one rank (testing)")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="wire format of the gradient all-reduce (N > 1); auto = fp32, bf16 with exactly two ranks (one xGMI link)")
    ap.add_argument("--reducer-only", action="store_true", help="attach the gradient reducer's hooks without any collective (measures their host cost)")
    ap.add_argument("--exec", dest="exec_mode", default="auto", choices=["auto", "graph", "eager"],
                    help="eager: launch every kernel from Python (weight gradients overlap the input-gradient chain on a "
                         "side stream); graph: replay the step from a captured hipGraph (no launch overhead, but the "
                         "runtime serialises the side-stream branch: measured slower); auto = eager")
    a = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything native libraries print there (RCCL writes a version banner to stdout
    # when the communicator is created) goes to stderr instead, until the line itself is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs a GPU"
    n_dev = torch.cuda.device_count()
    if local >= n_dev:                                  # more ranks than GPUs: only meaningful for the gloo rehearsal
        assert a.backend == "gloo", "one rank per GPU: %d ranks but %d GPUs" % (world, n_dev)
        local = local % n_dev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from shg_vqa_amd.agqa_hgqa import AGQA, DataTuple, SyntheticAGQA
    from shg_vqa_amd.agqa_model import AGQAModel
    from shg_vqa_amd import ddp
    from shg_vqa_amd.ddp import GradReducer
    from shg_vqa_amd.engine import engine, reset_engine
    from shg_vqa_amd.param import hgqa_args, parse_args

    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    reset_engine(compute_dtype=cdt, device=dev, seed=9595 + rank)
    if world > 1 or a.force_ddp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        ddp.init_process_group(dev, backend=a.backend)   # the engine's streams take their hardware queues before RCCL creates its own

    log("imports done; building model")
    torch.manual_seed(9595)                                   # identical --fromScratch initialisation on every rank
    if a.task == "vqa":
        args = parse_args(["--noCaps", "--crossAttnType", "cross", "--taskVQA", "--fromScratch", "--computeDtype", a.dtype,
                           "--batchSize", str(a.batch), "--lr", "1e-5"])
        model = AGQAModel(171, args=args)
    else:
        args = hgqa_args(compute_dtype=a.dtype, batch_size=a.batch, lr=1e-5)
        model = AGQAModel(171, num_queries=128, num_classes=456, num_actions=157, args=args)
    train_gflop = TRAIN_GFLOP_PER_QA_VQA if a.task == "vqa" else TRAIN_GFLOP_PER_QA
    model.to_engine(cdt)
    E = engine()
    # --force-ddp with one rank: the collectives run anyway (a 1-rank all-reduce is the identity)
    wire = {"bf16": torch.bfloat16, "fp32": None, "auto": ddp.default_wire_dtype(world)}[a.grad_wire]
    reducer = GradReducer(E.grad_arena, force_collectives=a.force_ddp, grad_dtype=wire) if (world > 1 or a.force_ddp or a.reducer_only) else None
    trainer = AGQA(args, train_tuple=DataTuple(SyntheticAGQA(n=1), [None] * 1000, None), model=model, t_total=10000,
                   world=reducer)
    log("model in HBM arenas (%d params, %d with gradients); building batches" % (E.n_total, E.n_active))
    batches = synthetic_device_batches(4, a.batch, 1234 + rank, dev)
    log("batches resident; warm-up")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    mode = a.exec_mode if a.exec_mode != "auto" else "eager"
    if mode == "graph":
        trainer.capture(batches[0])
        log("step captured into a hipGraph")
        step_fn = trainer.train_step_graphed
    else:
        # --overlap-update: the optimiser's HBM-bound sweep (all but conv1's weight) overlaps the next step's conv1 (the
        # timed region ends with a device synchronisation, so every update of its K steps is complete when the clock
        # stops).  Off by default: it buys ~0.25 ms per step but slows the dominant kernel it shares the fabric with by 4 %.
        def step_fn(bt):
            return trainer.train_step(bt, overlap_update=a.overlap_update)
    for i in range(a.warmup):
        step_fn(batches[i % len(batches)])
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    sync()
    E.kernel_events = []
    E.kernel_events_wgrad = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        step_fn(batches[i % len(batches)])
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ranks_seen = 1
    if dist.is_initialized():                              # proof that RCCL joined N ranks: every rank contributes its id
        ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
        ranks_seen = len({int(x.item()) for x in ids})
    log("timed region: %.3f s for %d steps" % (elapsed, a.steps))

    # dominant kernel: conv1 implicit GEMM, events recorded around its launch on its own stream
    evs = E.kernel_events or []
    evs_w = E.kernel_events_wgrad or []
    E.kernel_events = E.kernel_events_wgrad = None
    k_ms = sum(s.elapsed_time(e) for s, e in evs) / max(len(evs), 1)
    kw_ms = sum(s.elapsed_time(e) for s, e in evs_w) / max(len(evs_w), 1)
    B = a.batch
    conv1_flop = 2.0 * (B * 12 * 49) * 768 * (45 * 2048)      # forward and weight gradient contract the same three extents
    achieved = conv1_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    achieved_w = conv1_flop / (kw_ms * 1e-3) / 1e12 if kw_ms > 0 else 0.0

    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_conv1_latest.json")
    if os.path.exists(pmc) and B == 32 and a.dtype == "bf16":
        with open(pmc) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")
        traffic_src = "profiles/pmc_conv1_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 x2 fetch correction)"

    if rank == 0:
        qa = world * B * a.steps / elapsed
        line = {
            "metric": "training QA-pairs/sec (node) for 5/2/5-layer SHG-VQA", "value": round(qa, 2), "unit": "QA-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("agqaVQA.py --taskVQA 5/2/5 layers (video + question path, no HG decoder / no Hungarian), "
                                    "slow_r50-shaped feats (B,2048,16,7,7), per-GPU batch %d, random --fromScratch init "
                                    "(BASELINE.json configs[1])" % B) if a.task == "vqa" else
                                   ("agqaHGQA.py --taskHGQA --LossHGPerFrame full SHG-VQA model, llayers/xlayers/rlayers 5/2/5, "
                                    "dlayers 5, slow_r50-shaped feats (B,2048,16,7,7), per-GPU batch %d, random --fromScratch init "
                                    "(BASELINE.json configs[2]; configs[4] at 8 GPUs)" % B),
                       "backend": (a.backend if dist.is_initialized() else None),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "execution": mode, "ranks_seen": ranks_seen,
                       "grad_wire": ("bf16" if wire is not None else "fp32") if reducer is not None else None},
            "step_mfma_frac": round(qa / world * train_gflop * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4),
        }
        # the two longest kernels of the step, both conv1 (2048 -> 768): its forward (stream-K implicit GEMM) and its weight
        # gradient (main launch + split remainder launch, timed as one); `roofline` is whichever is LONGER, `roofline_rows` both
        row_f = {"kernel": "gemm8_sk_kernel<bf16, ConvRowSrc, PlainSrc> (stream-K; shg_conv3d_k533_fwd, 2048->768)", "bound": "mfma",
                 "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_live": False,
                 "launch_ms": round(k_ms, 4), "flop_per_launch": conv1_flop, "launches_timed": len(evs)}
        row_w = {"kernel": "gemm8_kernel<float, PlainSrc<kstrided>, ConvColSrc> x 2 launches (whole rounds + split remainder; "
                           "shg_conv3d_k533_wgrad, 2048->768)", "bound": "mfma",
                 "achieved": round(achieved_w, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(achieved_w / PEAK_BF16_TFLOPS, 4), "traffic": None,
                 "launch_ms": round(kw_ms, 4), "flop_per_launch": conv1_flop, "launches_timed": len(evs_w)}
        # position-major rows with whole K-tiles per position: the kernel leaves out the products with the zero border (361 of the
        # 441 (position, kh, kw) pairs of a 3 x 3 window on 7 x 7 remain); `achieved` stays on the ALGORITHMIC count above
        if getattr(E, "conv1_row_order", 0) == 1 and (B * 12) % 64 == 0:
            row_w["executed_flop_per_launch"] = conv1_flop * 361.0 / 441.0
            row_w["executed_tflops"] = round(achieved_w * 361.0 / 441.0, 2)
            row_w["note"] = "zero-border products skipped (DESIGN.md section 9 (10)): executed flop = 361/441 of the algorithmic count"
        if getattr(E, "conv1_row_order", 0) == 1 and (getattr(E, "conv_fwd_pm", 0) & 1) and B == 32:
            # forward in position-major rows: a 256-row tile leaves out the (kh, kw) taps that read only the zero border for all of its
            # rows - 562 of the 666 (tile, kh, kw) pairs of the 74 row blocks remain (B = 32: 384 rows per position)
            row_f["executed_flop_per_launch"] = conv1_flop * 562.0 / 666.0
            row_f["executed_tflops"] = round(achieved * 562.0 / 666.0, 2)
            row_f["note"] = "zero-border taps skipped per tile, stream-K with the weighted plan (DESIGN.md section 9 (10)): executed flop = 562/666 of the algorithmic count"
        line["roofline"] = row_w if kw_ms > k_ms else row_f
        line["roofline_rows"] = [row_f, row_w]
        if world == 1 and mode == "eager" and a.dtype == "bf16" and not a.no_extras:
            log("forward-only pass and attention-stack sub-roofline ...")
            line["forward_only"] = forward_only(trainer, batches, fwd_gflop=FWD_GFLOP_PER_QA_VQA if a.task == "vqa" else FWD_GFLOP_PER_QA)
            line["attention_stack"] = attention_stack(trainer, B)
            line["attention_stack"]["model_schedule_ms_fwd_bwd"] = attention_stack(trainer, B, schedule="model")["ms_fwd_bwd"]
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle) ...")
            line["cpu_baseline"] = cpu_baseline(task=a.task)
            log("cpu baseline done")
        fp32_wanted = world == 1 and mode == "eager" and a.dtype == "bf16" and not a.no_extras and a.task == "hgqa"
    else:
        fp32_wanted, line = False, None
    if fp32_wanted:
        # last: the child process needs the GPU memory this process still holds only partly (5.7 GB of arenas each)
        log("fp32 parity-mode steps (child process) ...")
        del trainer, model, batches
        torch.cuda.empty_cache()
        line["fp32_parity_mode"] = fp32_parity_mode(B)
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    if rank == 0:
        print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    if dist.is_initialized():
        dist.destroy_process_group()
    return


if __name__ == "__main__":
    main()
