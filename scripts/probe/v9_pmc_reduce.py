"""reduces the counter / kernel-trace CSVs of scripts/probe/v9_pmc_ablate.sh: per (run, kernel, grid) the median of
GRBM_GUI_ACTIVE / 8 (cycles per launch: the counter sums the 8 XCDs), the MFMA-busy share, the kernel duration and the
clock they imply; then the K-slope in cycles per K-tile between the two K of the shape file."""
import csv
import glob
import os
import statistics
import sys

out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "pmc_*.csv"))):
    tag = os.path.basename(f)[4:-4]
    per = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "gemm" not in name.lower() and "Cijk" not in name:
            continue
        key = (name[:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        per.setdefault((key, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg = {}
    for (key, _), c in per.items():
        agg.setdefault(key, []).append(c)
    dur = {}
    kt = os.path.join(out, f"kt_{tag}.csv")
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            name = r["Kernel_Name"]
            if "gemm" not in name.lower() and "Cijk" not in name:
                continue
            dur.setdefault((name[:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", "")), []).append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for key, cs in agg.items():
        med = lambda k: statistics.median(c.get(k, 0.0) for c in cs)
        cyc = med("GRBM_GUI_ACTIVE") / 8
        us = statistics.median(dur[key]) if key in dur else float("nan")
        rows.append((tag, key[0], key[1], len(cs), cyc, med("SQ_VALU_MFMA_BUSY_CYCLES") / max(med("SQ_BUSY_CYCLES"), 1) ,
                     med("SQ_INSTS_MFMA"), med("SQ_WAIT_ANY") / max(med("SQ_WAVE_CYCLES"), 1),
                     med("SQ_WAIT_INST_ANY") / max(med("SQ_WAVE_CYCLES"), 1), us, cyc / us * 1e-3 if us == us else float("nan")))
print("run,kernel,grid,launches,cycles_per_launch,mfma_busy_share,insts_mfma,wait_any_share,wait_inst_share,us,ghz")
for r in rows:
    print(",".join(str(round(x, 4)) if isinstance(x, float) else str(x) for x in r))
