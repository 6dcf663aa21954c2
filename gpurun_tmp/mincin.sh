ONLY=local3x3 python tools/bench_conv.py 2>&1 | tail -2
IR2RGB_CONV3X3P_MIN_CIN=128 ONLY=local3x3 python tools/bench_conv.py 2>&1 | tail -2
IR2RGB_CONV3X3P_MIN_CIN=128 ONLY=local3x3 python tools/bench_conv.py --f16 2>&1 | tail -2
python tools/prof_config5.py 2>&1 | tail -1 | cut -c1-260
IR2RGB_CONV3X3P_MIN_CIN=128 python tools/prof_config5.py 2>&1 | tail -1 | cut -c1-260
python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_CONV3X3P_MIN_CIN=128 python tools/prof_train.py 40 2>&1 | tail -1
python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_CONV3X3P_MIN_CIN=128 python tools/prof_train.py 40 2>&1 | tail -1
