"""HBM-side read traffic of ONE decode step from a rocprofv3 --pmc FETCH_SIZE counter_collection CSV of
scripts/decode_once.py (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide
coalesced read stream, unit KB).  usage: pmc_decode_traffic.py <counter_collection.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
rows.sort()
em = [i for i, r in enumerate(rows) if "decode_emit" in r[1]]
step = rows[em[-3] + 1:em[-2] + 1]
agg = {}
for _, name, v in step:
    key = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][-48:]
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += v
print("kernel,launches,read_GB_corrected(2x FETCH_SIZE)")
tot = 0.0
for key, (n, kb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    gb = 2 * kb * 1024 / 1e9
    tot += gb
    print(f"{key},{n},{gb:.3f}")
print(f"TOTAL,{len(step)},{tot:.2f}   (weights of LLaMA-7B: 13.48 GB)")
