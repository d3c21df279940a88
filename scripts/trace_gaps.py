"""Reads a rocprofv3 kernel-trace CSV (…_kernel_trace.csv) and reports, for the LAST training
step, wall time vs the sum of kernel durations (idle gaps = launch-bound time) and the top
kernels.  A step boundary is the first embedding/patchify kernel after an adamw_kernel."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps end with a run of adamw kernels
last_adam = max(i for i, r in enumerate(rows) if "adamw" in r[2])
j = last_adam
while j > 0 and "adamw" in rows[j][2]:
    j -= 1
# go back to the previous adamw run = start of the last step
k = j
while k > 0 and "adamw" not in rows[k][2]:
    k -= 1
step = rows[k + 1:last_adam + 1]
t0, t1 = step[0][0], max(r[1] for r in step)
busy = sum(r[1] - r[0] for r in step)
gaps = [max(0, step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)]
print(f"last step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.2f} ms, sum of kernel durations {1e-6 * busy:.2f} ms, "
      f"idle (gaps) {1e-6 * sum(gaps):.2f} ms, max gap {1e-3 * max(gaps):.1f} us, gaps > 20 us: {sum(g > 20000 for g in gaps)}")
agg = {}
for s, e, n in step:
    key = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    key = key.split("(")[0][-70:]
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += e - s
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{1e-6 * t:9.3f} ms {c:5d}  {key}")
