import csv, sys, statistics, glob, os
out = sys.argv[1]
agg = {}
for f in sorted(glob.glob(os.path.join(out, "attn_pmc*.csv"))):
    per = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "flash" not in n:
            continue
        key = n[n.find("flash"):][:40]
        d = per.setdefault((key, r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for (key, _), c in per.items():
        for k, v in c.items():
            agg.setdefault(key, {}).setdefault(k, []).append(v)
for key, c in agg.items():
    m = {k: statistics.median(v) for k, v in c.items()}
    print(key)
    print("   " + "  ".join(f"{k}={m[k]:.4g}" for k in sorted(m)))
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print(f"   shares of wave cycles: wait_any {m.get('SQ_WAIT_ANY',0)/wc:.3f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.3f} "
              f"active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f} valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f}; "
              f"mfma busy / (4 SIMD x GRBM cycles/8 x 256 CU): {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*256*m.get('GRBM_GUI_ACTIVE',1)/8):.3f}; "
              f"clock {m.get('GRBM_GUI_ACTIVE',0)/8/m['us']*1e-3:.2f} GHz")
