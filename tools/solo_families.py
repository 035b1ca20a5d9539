"""Solo kernel time per kernel family of ONE training step, from a rocprofv3 --kernel-trace run of bench.py with everything
on one stream (SHG_BRANCH_MASK=0 SHG_OVERLAP_WGRAD=0 SHG_KV_AHEAD=0): no kernel shares the chip, so a duration is a cost.

    python tools/solo_families.py <rocprof dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

FAMILIES = ["gemm8_sk_kernel", "gemm8_group_kernel", "gemm8_kernel", "gemm4_kernel", "gemm_kernel", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv",
            "ln_fwd", "ln_bwd", "colsum_atomic", "colsum_finish_multi", "colsum_finish", "colsum_partial", "bertadam", "bias_act_bwd",
            "bias_act_fwd", "hungarian", "add2", "sumsq", "ncdhw", "tokens_assemble", "wce", "bce"]


def family(n):
    base = "torch / runtime glue"
    for k in FAMILIES:
        if k in n:
            base = k
            break
    if base.startswith("gemm"):
        if "ConvRowSrc" in n and "ConvWeightColSrc" in n:
            base += " (conv input gradient)"
        elif "ConvRowSrc" in n:
            base += " (conv forward)"
        elif "ConvColSrc" in n:
            base += " (conv weight gradient)"
        elif re.search(r"kernelI(DF16b)?f", n):
            base += " (fp32 out: weight gradients)"
    return base


def main():
    d = sys.argv[1]
    f = (glob.glob(os.path.join(d, "*", "*kernel_trace.csv")) + glob.glob(os.path.join(d, "*kernel_trace.csv")))[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    adam = [e for s, e, n in rows if "bertadam" in n.lower()]
    t0, t1 = adam[-3], adam[-2]
    win = [r for r in rows if r[1] > t0 and r[0] < t1]
    busy = sum(e - s for s, e, _ in win) / 1e6
    print("step window %.2f ms, %d kernels, kernel time %.2f ms" % ((t1 - t0) / 1e6, len(win), busy))
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n in win:
        a = agg[family(n)]
        a[0] += 1
        a[1] += (e - s) / 1e6
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-48s %4d launches  %7.3f ms  %5.1f %%  avg %7.1f us" % (k, c, t, 100 * t / busy, 1e3 * t / c))


if __name__ == "__main__":
    main()
