#!/bin/bash
# Timeline of ONE fused training step: every kernel between two adam launches with its duration and the gap before it,
# from a rocprofv3 kernel trace of tools/kbench.py.   usage: tools/step_timeline.sh <tag>
tag=${1:-timeline}
out=$PWD/gpurun_out/$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 $repo/${STEP_SCRIPT:-tools/kbench.py} render_bwd > $out/kbench.log 2> $out/err.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/timeline.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:70]
idx = [i for i, r in enumerate(rows) if nm(r).startswith("adam_kernel")]
a, b = idx[-3], idx[-2]
prev_end = int(rows[a]["End_Timestamp"])
t0 = prev_end
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm(r)}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us, launches {b - a}")
PY
