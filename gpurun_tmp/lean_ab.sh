set -e
python -m pytest tests/test_stage_backward_gpu.py tests/test_networks_gpu.py tests/test_harness_gpu.py tests/test_losses_gpu.py tests/test_streams_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -4
for i in 1 2 3; do
python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_LEAN_STAGE=0 python tools/prof_train.py 40 2>&1 | tail -1
done
python tools/host_backward.py 2>&1 | tail -27 | head -6
