set -e
mkdir -p gpurun_out/r3f
python -m pytest tests/test_streams_gpu.py -x -q -k "discriminator_streams or sixteen" 2>&1 | tail -15
python tools/prof_train.py 30 2>&1 | tail -1
IR2RGB_D_STREAMS=0 python tools/prof_train.py 30 2>&1 | tail -1
python tools/prof_train.py 30 2>&1 | tail -1
