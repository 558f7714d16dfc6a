#!/usr/bin/env python3
"""rocprofv3 target: one CLaSS round at config-B width (bench.class_wide) - which launches the per-step decode chain spends its time in."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "controlled-peptide-generation_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
print(bench.class_wide(torch.device("cuda")))
