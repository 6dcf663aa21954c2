set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainprof -o t -- python3 $R/tools/prof_train.py 30 > $O/train_prof.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
python3 tools/stats_summary.py $O/trainprof/t_kernel_stats.csv 30 30
