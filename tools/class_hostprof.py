"""Where the host time of one CLaSS round goes (cProfile over sample_pipeline.run_rounds, 1 M proposals, beam-5 of everything)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import bench  # noqa: E402
import sample_pipeline as sp  # noqa: E402

dev = torch.device("cuda")
m, Q, ds = bench.class_setup(dev)
sp.run_rounds(m, ds, Q, 65536, 10 ** 9, max_rounds=1, sample_mode='beam')
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
df, st = sp.run_rounds(m, ds, Q, 1000000, 10 ** 9, max_rounds=1, return_stats=True, sample_mode='beam')
torch.cuda.synchronize()
pr.disable()
print("wall %.3f s" % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
