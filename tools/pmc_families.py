"""Matrix-pipe utilisation per kernel family from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
--kernel-trace run of bench.py with everything on one stream (SHG_BRANCH_MASK=0 SHG_OVERLAP_WGRAD=0): per family and grid
size, launches of the last step, average duration, and MFMA-busy cycles per SIMD as a fraction of the launch's shader cycles
(SQ_VALU_MFMA_BUSY_CYCLES / 1 024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs - the recipe of profiles/README.md).

    python tools/pmc_families.py <rocprof dir>"""
import csv, glob, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from solo_families import family


def main():
    d = sys.argv[1]
    one = lambda pat: (glob.glob(os.path.join(d, "*", pat)) + glob.glob(os.path.join(d, pat)))[0]
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(one("*kernel_trace.csv")))}
    ctr = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(one("*counter_collection.csv"))):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = sorted(trace.values(), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "bertadam" in r["Kernel_Name"].lower()]
    lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for r in rows[lo:hi]:
        c = ctr.get(r["Dispatch_Id"])
        if not c:
            continue
        wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * max(int(r["Grid_Size_Y"]), 1)
        key = (family(r["Kernel_Name"]), wgs)
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[2] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0
        a[3] += c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    print("%-44s %7s %5s %9s %9s" % ("family", "wgs", "n", "avg us", "MFMA busy"))
    for (fam, wgs), (n, us, mf, act) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if us / n < 8 and mf == 0:
            continue
        print("%-44s %7d %5d %9.1f %8.1f %%" % (fam[:44], wgs, n, us / n, 100.0 * mf / act if act else 0.0))


if __name__ == "__main__":
    main()
