for i in 1 2; do
python tools/prof_train.py 40 2>&1 | tail -1
GPU_MAX_HW_QUEUES=8 python tools/prof_train.py 40 2>&1 | tail -1
GPU_MAX_HW_QUEUES=6 python tools/prof_train.py 40 2>&1 | tail -1
done
GPU_MAX_HW_QUEUES=8 python tools/prof_forward.py 2 --graph 2>&1 | tail -1
python tools/prof_forward.py 2 --graph 2>&1 | tail -1
