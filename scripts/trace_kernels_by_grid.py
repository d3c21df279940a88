"""rocprofv3 kernel-trace CSV -> one line per (kernel, grid, workgroup, LDS): launches, mean duration (us).
Used to read which vendor kernel hipBLASLt's heuristic picks per shape (profiles/r06_vendor_isa.txt)."""
import collections
import csv
import sys

c = collections.OrderedDict()
pref = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not n.startswith(pref):
        continue
    k = (n, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"),
         r.get("LDS_Block_Size"))
    c.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in c.items():
    print(len(v), round(sum(v) / len(v), 1), k[1], k[2], k[3], k[0])
