"""Reads a rocprofv3 kernel-trace CSV of scripts/bench_generate.py-style decoding and reports ONE
decode step (between two consecutive token-selection kernels near the end): kernels, wall, busy, gaps."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
am = [i for i, r in enumerate(rows) if "argmax" in r[2] or "decode_emit" in r[2]]
a, b = am[-3], am[-2]
step = rows[a + 1:b + 1]
t0, t1 = rows[a][1], step[-1][1]
busy = sum(r[1] - r[0] for r in step)
gaps = [max(0, step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)]
print(f"one decode step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.3f} ms, kernel time {1e-6 * busy:.3f} ms, "
      f"gaps {1e-6 * sum(gaps):.3f} ms (mean {1e-3 * sum(gaps) / max(1, len(gaps)):.1f} us)")
agg = {}
for s, e, n in step:
    key = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]
    x = agg.setdefault(key, [0, 0])
    x[0] += 1
    x[1] += e - s
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{1e-3 * t:9.1f} us {c:4d}  ({1e-3 * t / c:6.1f} us each)  {key}")
