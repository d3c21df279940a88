"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs of a bench.py run into
per-kernel HBM-side traffic of ONE training step (the last complete one), with the gfx950
correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE reports 1/2 of a wide coalesced read stream;
unit KB).  usage: pmc_step_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import csv
import sys


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def last_step(rows):
    idx = [i for i, r in enumerate(rows) if "adamw" in r[1]]
    end = idx[-1]
    j = end
    while j > 0 and "adamw" in rows[j][1]:
        j -= 1
    k = j
    while k > 0 and "adamw" not in rows[k][1]:
        k -= 1
    return rows[k + 1:end + 1]


fetch = last_step(load(sys.argv[1], "FETCH_SIZE"))
write = last_step(load(sys.argv[2], "WRITE_SIZE"))
agg = {}
for rows, slot in ((fetch, 0), (write, 1)):
    for _, name, v in rows:
        key = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][-64:]
        a = agg.setdefault(key, [0, 0.0, 0.0])
        if slot == 0:
            a[0] += 1
        a[1 + slot] += v
print("kernel,launches,read_GB_corrected(2x FETCH_SIZE),write_GB,per_launch_MB")
tot_r = tot_w = 0.0
for key, (n, r_kb, w_kb) in sorted(agg.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2])):
    r_gb, w_gb = 2 * r_kb * 1024 / 1e9, w_kb * 1024 / 1e9
    tot_r += r_gb
    tot_w += w_gb
    if r_gb + w_gb > 0.05:
        print(f"{key},{n},{r_gb:.2f},{w_gb:.2f},{(r_gb + w_gb) * 1e3 / max(n, 1):.1f}")
print(f"TOTAL,,{tot_r:.1f},{tot_w:.1f},")
