set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1 || { tail -30 $O/gpu_tests.txt; exit 1; }
tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/benchprof -o b -- python3 $R/bench.py > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainprof -o t -- python3 $R/tools/prof_train.py 30 > $O/train_prof.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwdprof -o f -- python3 $R/tools/prof_forward.py 2 --graph > $O/fwd_prof.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -o c -- python3 $R/tools/prof_config5.py > $O/c5_prof.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/train_prof.txt | tail -2; tail -2 $O/fwd_prof.txt; tail -1 $O/c5_prof.txt
