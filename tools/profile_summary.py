"""Turns rocprofv3 output directories into the small files tracked under profiles/:

    python tools/profile_summary.py stats <dir> <out_prefix>       kernel-trace run  -> <prefix>_kernel_stats.csv (copy),
                                                                   <prefix>_conv1_row.json (the conv1 forward launches ALONE)
    python tools/profile_summary.py pmc <fetch_dir> <write_dir> <mfma_dir|-> <out.json>   PMC passes -> bytes / busy cycles per conv1 launch

conv1 forward = gemm8_sk_kernel<bf16, ConvRowSrc, PlainSrc<true>> (gemm8_kernel<..., SK = true> before r02_d) launches longer than 1.5 ms (conv2 forward shares the
instantiation but runs ~0.55 ms)."""
import csv, glob, json, os, shutil, sys


def _one(d, pat):
    f = glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, pat))
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return f[0]


def is_conv_fwd(name):
    return "gemm8" in name and "ConvRowSrc" in name and "ConvWeightColSrc" not in name and "ConvColSrc" not in name


def stats(d, prefix):
    shutil.copy(_one(d, "*kernel_stats.csv"), prefix + "_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(_one(d, "*kernel_trace.csv"))) if is_conv_fwd(r["Kernel_Name"])]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    c1 = [(x, r) for x, r in zip(dur, rows) if x > 1.5]
    c2 = [(x, r) for x, r in zip(dur, rows) if x <= 1.5]
    out = {"kernel": rows[0]["Kernel_Name"] if rows else None,
           "conv1_forward": {"launches": len(c1), "avg_ms": sum(x for x, _ in c1) / max(len(c1), 1), "min_ms": min((x for x, _ in c1), default=None),
                             "max_ms": max((x for x, _ in c1), default=None), "grid": sorted({r["Grid_Size_X"] for _, r in c1}),
                             "scratch_bytes_per_lane": sorted({r["Scratch_Size"] for _, r in c1}), "vgpr": sorted({r["VGPR_Count"] for _, r in c1}),
                             "tflops_at_avg": 2.0 * 18816 * 768 * 92160 / (sum(x for x, _ in c1) / max(len(c1), 1) * 1e-3) / 1e12 if c1 else None},
           "conv2_forward": {"launches": len(c2), "avg_ms": sum(x for x, _ in c2) / max(len(c2), 1),
                             "scratch_bytes_per_lane": sorted({r["Scratch_Size"] for _, r in c2})}}
    json.dump(out, open(prefix + "_conv1_row.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


def _counter(d, want):
    f = _one(d, "*counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != want:
            continue
        per.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), 0.0)
        per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
    return per


def pmc(fetch_dir, write_dir, mfma_dir, out):
    res = {}
    for label, d, names in (("fetch", fetch_dir, ["FETCH_SIZE"]), ("write", write_dir, ["WRITE_SIZE"]),
                            ("mfma", mfma_dir, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])):
        if d == "-":
            continue
        # conv1 = the conv-forward dispatches with the largest counter values of their kind (conv2 is ~4x smaller)
        trace = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                 for r in csv.DictReader(open(_one(d, "*kernel_trace.csv")))}
        for n in names:
            per = _counter(d, n)
            conv = [(v, k) for k, v in per.items() if is_conv_fwd(k[1]) and trace.get(k[0], 0) > 1.5]
            adam = [v for k, v in per.items() if "bertadam" in k[1]]
            if conv:
                res[n] = {"conv1_per_launch": sum(v for v, _ in conv) / len(conv), "launches": len(conv)}
            if adam:
                res[n + "_bertadam"] = sum(adam) / len(adam)
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        res["traffic_bytes_per_launch"] = res["FETCH_SIZE"]["conv1_per_launch"] * 1024 * 2 + res["WRITE_SIZE"]["conv1_per_launch"] * 1024
        res["correction"] = "FETCH_SIZE (KB) x2 on gfx950 (128-B requests tallied at 64 B; MI355X_MICROARCH.md), WRITE_SIZE exact"
        res["algorithmic_bytes"] = 369300000.0
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:6])
