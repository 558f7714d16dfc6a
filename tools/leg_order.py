"""Does an LSTM training leg depend on what ran before it in the process?  (bench.py's lstm leg: 138 us per paired BPTT launch after
the GRU legs against 80 us alone.)  python tools/leg_order.py"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "controlled-peptide-generation_amd")]
import torch
import bench

args = types.SimpleNamespace(cell="gru")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def leg(tag, dtype, cell):
    r = bench.train_leg(args, dev, 0, 1, dtype, 512, 1, 2048, 25, 10, 3, cell=cell)
    fams = [r["roofline"]] + r["kernel_families"]
    print("%-22s %7.3f ms/step  %s" % (tag, r["ms_per_step"], {f["family"]: f["avg_launch_us"] for f in fams}), flush=True)


for order in sys.argv[1:] or ["L", "GL", "BL", "GBL", "LL"]:
    print("order", order)
    for ch in order:
        leg({"L": "lstm f32", "G": "gru f32", "B": "gru bf16", "M": "lstm bf16"}[ch], "bf16" if ch in "BM" else "f32", "lstm" if ch in "LM" else "gru")
