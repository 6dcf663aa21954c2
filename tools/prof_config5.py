"""BASELINE config 5 forward only (two-scale generator at 1024x2048, f16; bench.config5_forward): for rocprofv3
--kernel-trace --stats.  The HIP-graph replays dominate the trace (5 eager + 12 graphed forwards)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.config5_forward(torch.device("cuda:0"))))
