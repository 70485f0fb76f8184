"""Per-decode-step kernel breakdown from a rocprofv3 kernel trace CSV:
isolates the last step (between the last two sampling kernels) and aggregates
kernel time by name.  usage: step_breakdown.py <kernel_trace.csv> [marker]"""
import collections
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "argmax"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"]
    if not n.startswith(("Cijk", "Custom_Cijk", "_Z")):
        n = n.replace("void ", "").replace("(anonymous namespace)::", "")
        for stop in ("<", "("):
            if stop in n:
                n = n[:n.index(stop)]
    n = n[:64]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, kernels {len(step)}, kernel time {tot / 1e6:.3f} ms")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / 1e6:9.3f} ms {100 * v[1] / tot:5.1f}%  x{v[0]:4d}  avg {v[1] / v[0] / 1e3:9.1f} us  {n}")
