"""Per-kernel counter rows of the three attention kernels from rocprofv3 --pmc passes over `tools/op_bench.py attn393`
(B 32, H 12, 393 x 393, key mask, p = 0.1): one directory per pass, counters summed per dispatch, averaged over the launches.

    python tools/pmc_attention.py <dir> [<dir> ...] > profiles/rNN_pmc_attention.txt

Derived columns: VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES), VALU-busy share (SQ_ACTIVE_INST_VALU over SQ_BUSY_CYCLES:
both are per-SIMD cycle sums on gfx950), MFMA-busy share (SQ_VALU_MFMA_BUSY_CYCLES / 1 024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, the
recipe of profiles/README.md)."""
import csv, glob, os, sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sys.argv[1:]:
    one = lambda pat: (glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, pat)))[0]
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(one("*kernel_trace.csv")))}
    ctr = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(one("*counter_collection.csv"))):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for did, r in trace.items():
        n = r["Kernel_Name"]
        if "attn_" not in n:
            continue
        k = "attn_fwd" if "attn_fwd" in n else ("attn_bwd_dq" if "bwd_dq" in n else ("attn_bwd_dkv" if "bwd_dkv" in n else "attn_bwd_fused"))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for c, v in ctr.get(did, {}).items():
            agg[k][c].append(v)
        agg[k]["VGPR_Count"].append(float(r["VGPR_Count"]))
        agg[k]["LDS_Block_Size"].append(float(r["LDS_Block_Size"]))
        agg[k]["Scratch_Size"].append(float(r["Scratch_Size"]))
for k in sorted(agg):
    a = {c: sum(v) / len(v) for c, v in agg[k].items()}
    print("%s: %d profiled launches, avg %.1f us under the counters" % (k, len(dur[k]), sum(dur[k]) / len(dur[k])))
    for c in sorted(a):
        print("    %-28s %16.0f" % (c, a[c]))
    if a.get("SQ_WAVES"):
        print("    -> VALU instructions per wave          %10.0f" % (a.get("SQ_INSTS_VALU", 0) / a["SQ_WAVES"]))
        print("    -> LDS instructions per wave           %10.0f" % (a.get("SQ_INSTS_LDS", 0) / a["SQ_WAVES"]))
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_ACTIVE_INST_VALU"):
        print("    -> VALU busy / SQ busy                 %10.1f %%" % (100.0 * a["SQ_ACTIVE_INST_VALU"] / a["SQ_BUSY_CYCLES"]))
    if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("    -> MFMA busy                           %10.1f %%" % (100.0 * (a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (a["GRBM_GUI_ACTIVE"] / 8.0)))
