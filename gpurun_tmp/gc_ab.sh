for i in 1 2 3; do
python tools/prof_train.py 40 2>&1 | tail -1
GC_TUNE=1 python tools/prof_train.py 40 2>&1 | tail -1
done
python tools/prof_forward.py 2 --graph 2>&1 | tail -1
