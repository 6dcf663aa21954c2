# Round-end evidence on a GPU box: the GPU test suite, smoke(), two bench.py runs and rocprofv3 kernel stats of the bench, the
# training windows, the north-star graph forward and config 5 -> gpurun_out/evidence/ (copy what is to be judged into profiles/).
# usage (from the repository root):  gpurun --timeout 1200 -- bash tools/collect_evidence.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1 || { tail -40 $O/gpu_tests.txt; exit 1; }
tail -2 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python bench.py > $O/bench_b.json 2> $O/bench_b.err
python -c "
import json
for f in ('bench.json','bench_b.json'):
    d=json.load(open('$O/'+f)); e=d['extra']
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'], e['north_star_generator_forward_512x1024_hip_graph'], e['config5_two_scale_forward_1024x2048_f16']['hip_graph'])
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/benchprof -o b -- python3 $R/bench.py > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainprof -o t -- python3 $R/tools/prof_train.py 30 > $O/train_prof.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwdprof -o f -- python3 $R/tools/prof_forward.py 2 --graph > $O/fwd_prof.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -o c -- python3 $R/tools/prof_config5.py > $O/c5_prof.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
grep "last 10" $O/train_prof.txt; grep "graphed forward" $O/fwd_prof.txt; grep algorithmic $O/c5_prof.txt | cut -c1-300
