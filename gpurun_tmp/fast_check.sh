set -e
mkdir -p gpurun_out/r3f
python -m pytest tests -m gpu -x -q > gpurun_out/r3f/gpu_tests2.txt 2>&1 || { tail -40 gpurun_out/r3f/gpu_tests2.txt; exit 1; }
tail -2 gpurun_out/r3f/gpu_tests2.txt
python tools/prof_train.py 30 2>&1 | tail -1
IR2RGB_FASTBIND=0 python tools/prof_train.py 30 2>&1 | tail -1
IR2RGB_D_STREAMS=0 python tools/prof_train.py 30 2>&1 | tail -1
python tools/prof_train.py 30 2>&1 | tail -1
python tools/host_backward.py 2>&1 | tail -27 | head -14
