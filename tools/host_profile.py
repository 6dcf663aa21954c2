"""cProfile of the host side of the training window (steady state): where the issue time goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ir2rgb_amd import vid2vid as V
dev = torch.device("cuda:0")
tr = V.Vid2VidTrainer(dev, n_scales_spatial=2, resident_inputs=True)
A, B = V.synthetic_sequence(40, 512, 1024, 1234, dev)
for i in range(16):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(16, 26):
    tr.train_window(A[:, i:i + 3], B[:, i:i + 3])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
