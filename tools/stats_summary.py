"""Summary of a rocprofv3 *_kernel_stats.csv by kernel family: python tools/stats_summary.py file.csv [windows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nw = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
def cat(n):
    if 'wgrad' in n: return 'weight gradients'
    if 'conv' in n and any(k in n for k in ('igemm', 'patch', 'dot', 'thin')): return 'convolutions (fwd + data gradient)'
    if 'bn_bwd' in n: return 'BatchNorm backward'
    if n.startswith(('bn_', 'void bn_')): return 'BatchNorm forward'
    if 'adam' in n or 'pack_' in n: return 'Adam + repack'
    if 'at::native' in n or 'rocclr' in n or 'Cijk' in n: return 'torch operators + copies'
    if 'loss' in n: return 'loss kernels'
    if any(k in n for k in ('corr', 'resample', 'channelnorm', 'warp_diff', 'flow_up')): return 'FlowNet2 operators'
    return 'other HIP kernels'
agg = {}
for r in rows:
    a = agg.setdefault(cat(r['Name']), [0, 0]); a[0] += int(r['TotalDurationNs']); a[1] += int(r['Calls'])
print("total kernel time %.2f ms, %d launches (per window: %.2f ms, %.0f launches)" % (tot / 1e6, sum(int(r['Calls']) for r in rows), tot / 1e6 / nw, sum(int(r['Calls']) for r in rows) / nw))
for c, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-36s %8.2f ms/window %5.1f %%  %6.0f launches/window" % (c, t / 1e6 / nw, 100 * t / tot, n / nw))
print()
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-74s %7.1f /win %8.1f us %5.1f%%" % (re.sub(r'\(.*', '', r['Name'])[:74], int(r['Calls']) / nw, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
