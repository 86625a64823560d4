"""Idle time between kernels in the training step, from a rocprofv3 --kernel-trace csv of bench.py:
    tools/gap_stats.py <kernel_trace.csv> [steps to analyse, default 100]
Takes the last N occurrences of the step's first kernel as step boundaries; prints the step's wall time, the time at least one
kernel was running (union over the streams), the idle remainder, and the idle time by (previous kernel -> next kernel) pair."""
import csv, sys, collections, re
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"<.*", "", n.split("(")[0])
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
first = "deform_field_fwd_b3_kernel"
starts = [i for i, r in enumerate(rows) if r[2] == first]
starts = starts[-(n_steps + 1):]
wall = busy = 0
gaps = collections.Counter(); gapn = collections.Counter()
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall += rows[b][0] - seg[0][0]
    cur_end = seg[0][1]; busy += seg[0][1] - seg[0][0]; last_name = seg[0][2]
    for s, e, name in seg[1:] + [rows[b]]:
        if s > cur_end:
            gaps[(last_name, name)] += s - cur_end; gapn[(last_name, name)] += 1
        if e > cur_end:
            busy += e - max(s, cur_end) if name is not rows[b][2] or (s, e, name) != rows[b] else 0
            if (s, e, name) != rows[b]:
                cur_end = e; last_name = name
n = len(starts) - 1
print(f"steps {n}: wall {wall / n / 1e3:.1f} us, some kernel running {busy / n / 1e3:.1f} us, idle {(wall - busy) / n / 1e3:.1f} us")
for (a, b), t in gaps.most_common(30):
    print(f"  {t / n / 1e3:6.2f} us/step  ({gapn[(a, b)] / n:.2f} x {t / gapn[(a, b)] / 1e3:5.2f} us)  {a} -> {b}")
# per-kernel average duration inside the analysed steps
dur = collections.Counter(); cnt = collections.Counter()
for s_, e_, name in rows[starts[0]:starts[-1]]:
    dur[name] += e_ - s_; cnt[name] += 1
print("kernel: launches per step x average us")
for name, t in dur.most_common(24):
    print(f"  {name:34s} {cnt[name] / n:5.2f} x {t / cnt[name] / 1e3:7.1f}")
