"""Timeline of one steady-state training step from a rocprofv3 kernel trace (tools/prof.sh <tag> leaves it in
gpurun_out/<tag>/run_kernel_trace.csv): start offset, duration, queue and gap to the previous kernel on the same queue.
usage: python tools/step_trace.py gpurun_out/<tag>/run_kernel_trace.csv [step index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "render_bwd" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
busy = 0
for r in rows[i0:i1]:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1000 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f}  q{q:>2s}  gap {gap:6.1f}  {name}")
print("step", (int(rows[i1]["Start_Timestamp"]) - t0) / 1000, "us")
