"""Post-process a rocprofv3 kernel trace of tools/host_bound.py: GPU busy / idle time and per-kernel totals of the last
10 training windows (windows are delimited by the one pack_batch_kernel launch that ends each of them).
usage: python tools/window_trace.py <kernel_trace.csv> [n_windows]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
packs = [e for e in ev if e[2].startswith("void pack_batch_kernel") or e[2].startswith("pack_batch_kernel")]
t0, t1 = packs[-n - 1][1], packs[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    tot[k][0] += e - s
    tot[k][1] += 1
print("windows %d: span %.2f ms/window, GPU busy (union) %.2f, idle %.2f, kernel sum %.2f, launches %d/window"
      % (n, (t1 - t0) / n / 1e6, busy / n / 1e6, (t1 - t0 - busy) / n / 1e6, sum(v[0] for v in tot.values()) / n / 1e6, len(win) // n))
# idle gaps histogram
gaps = []
cur_e = None
for s, e, _ in win:
    if cur_e is not None and s > cur_e:
        gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 20e3), (20e3, 50e3), (50e3, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps %5.0f-%5.0f us: %5d/window, %.2f ms/window" % (lo / 1e3, min(hi, 1e9) / 1e3, len(g) // n, sum(g) / n / 1e6))
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%-100s %6.1f/win %8.3f ms/win %7.1f us" % (k[:100], c / n, t / n / 1e6, t / c / 1e3))
