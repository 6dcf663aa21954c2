set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $R/tools/prof_train.py 30 > $O/train.txt 2>&1
cd $R
python3 tools/window_trace.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 10 > $O/window_trace.txt
python3 - <<'PY' $(find gpurun_out/r3t/tr -name "*kernel_trace.csv" | head -1) > gpurun_out/r3t/streams.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
packs = [e for e in ev if "pack_batch_kernel" in e[2]]
t0, t1 = packs[-2][1], packs[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("window span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
byq = collections.defaultdict(list)
for e in win: byq[e[3]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e[1] - e[0] for e in es)
    print("queue %s: %4d kernels, busy %.2f ms, first +%.2f ms, last +%.2f ms" % (q, len(es), busy / 1e6, (es[0][0] - t0) / 1e6, (es[-1][1] - t0) / 1e6))
# timeline in 0.5 ms bins: sum of kernel-time per bin (concurrency) and idle fraction
import math
nb = int(math.ceil((t1 - t0) / 5e5))
load = [0.0] * nb
for s, e, k, q in win:
    b0, b1 = int((s - t0) // 5e5), int((e - t0) // 5e5)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = t0 + b * 5e5, t0 + (b + 1) * 5e5
        load[b] += max(0, min(e, hi) - max(s, lo))
print("kernel-time per 0.5 ms bin (1.0 = one kernel resident all the time):")
print(" ".join("%.1f" % (x / 5e5) for x in load))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
head -12 $O/window_trace.txt; cat $O/streams.txt
