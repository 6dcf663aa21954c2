for i in 1 2; do
IR2RGB_D_SCALE_SLOT=9 IR2RGB_D_T_SLOTS=own python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_D_SCALE_SLOT=0 IR2RGB_D_T_SLOTS=own python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_D_SCALE_SLOT=0 IR2RGB_D_T_SLOTS=shared python tools/prof_train.py 40 2>&1 | tail -1
IR2RGB_D_SCALE_SLOT=9 IR2RGB_D_T_SLOTS=shared python tools/prof_train.py 40 2>&1 | tail -1
done
