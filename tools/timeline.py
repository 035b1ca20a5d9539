"""Per-queue timeline of ONE training step out of a rocprofv3 --kernel-trace run of bench.py.

    python tools/timeline.py <rocprof dir> [bucket_ms]

A step = the window between the ends of two consecutive BertAdam launches (the last complete one is used).  For every
hardware queue: busy time and kernel count; then the window in buckets: per queue the kernel that held most of the bucket and
the fraction of the bucket the queue was busy; then the longest idle gaps of the busiest queue (the main-stream chain)."""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"void |shg::|\(anonymous namespace\)::|at::native::", "", name)
    m = re.match(r"([A-Za-z0-9_]+)<(.*)", name)
    if m:
        tag = ""
        for k in ("ConvRowSrc", "ConvColSrc", "ConvWeightColSrc"):
            if k in m.group(2):
                tag += "," + k
        if "float" in m.group(2).split(",")[0:2] or m.group(2).startswith("float"):
            tag += ",f32"
        return m.group(1) + tag
    return name[:40]


def main():
    d = sys.argv[1]
    bucket = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    f = (glob.glob(os.path.join(d, "*", "*kernel_trace.csv")) + glob.glob(os.path.join(d, "*kernel_trace.csv")))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    adam = [e for s, e, q, n in rows if "bert_adam" in n or "bertadam" in n.lower()]
    if len(adam) < 3:
        raise SystemExit("fewer than 3 BertAdam launches in the trace")
    t0, t1 = adam[-3], adam[-2]
    win = [(max(s, t0), min(e, t1), q, n) for s, e, q, n in rows if e > t0 and s < t1]
    print("step window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
    per_q = defaultdict(list)
    for s, e, q, n in win:
        per_q[q].append((s, e, n))
    qs = sorted(per_q, key=lambda q: -sum(e - s for s, e, _ in per_q[q]))
    for q in qs:
        print("queue %s: %d kernels, busy %.2f ms" % (q, len(per_q[q]), sum(e - s for s, e, _ in per_q[q]) / 1e6))
    # chip idle: no kernel on any queue
    ev = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
    depth, last, idle, conc = 0, t0, 0, defaultdict(int)
    for t, dlt in ev:
        if depth == 0:
            idle += t - last
        conc[depth] += t - last
        last, depth = t, depth + dlt
    print("no kernel on any queue: %.2f ms; time by number of kernels in flight: %s" %
          (idle / 1e6, {k: round(v / 1e6, 2) for k, v in sorted(conc.items())}))
    nb = int((t1 - t0) / 1e6 / bucket) + 1
    print("\nbucket(ms)  " + "  ".join("queue %-28s" % q for q in qs))
    for b in range(nb):
        lo, hi = t0 + b * bucket * 1e6, t0 + (b + 1) * bucket * 1e6
        cells = []
        for q in qs:
            by = defaultdict(float)
            for s, e, n in per_q[q]:
                o = min(e, hi) - max(s, lo)
                if o > 0:
                    by[short(n)] += o
            busy = sum(by.values()) / (bucket * 1e6)
            top = max(by, key=by.get) if by else "-"
            cells.append("%3d%% %-29s" % (round(100 * busy), top[:29]))
        print("%5.1f       " % (b * bucket) + "  ".join(cells))
    q0 = qs[0]
    ks = sorted(per_q[q0])
    gaps = sorted(((ks[i + 1][0] - ks[i][1], ks[i][1], short(ks[i][2]), short(ks[i + 1][2])) for i in range(len(ks) - 1)), reverse=True)
    print("\nlongest idle gaps on queue %s (total idle %.2f ms of %.2f):" % (q0, ((t1 - t0) - sum(e - s for s, e, _ in ks)) / 1e6, (t1 - t0) / 1e6))
    for g, at, a, b in gaps[:25]:
        print("  %.1f us at %.2f ms: after %s, before %s" % (g / 1e3, (at - t0) / 1e6, a, b))
    # aggregate kernel time by short name in the window
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, q, n in win:
        agg[short(n)][0] += 1
        agg[short(n)][1] += (e - s) / 1e6
    print("\nkernel time in the window:")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("  %-44s %4d  %.3f ms" % (n, c, t))


if __name__ == "__main__":
    main()
