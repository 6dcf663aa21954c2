set -e
python -m pytest tests/test_streams_gpu.py tests/test_rccl_gpu.py tests/test_harness_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_ADAM_STREAM=0 python tools/prof_train.py 40 2>&1 | tail -1
done
