import torch, ctypes
lib = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", lib.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p)
        print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "error", e)
