"""Where the HOST spends a training step at the reference's default sizes (batch 32: the eager step is bound by Python + launch
overhead, not by the device): cProfile over 300 eager steps of bench.train_leg's model.   python tools/host_profile.py [batch]"""
import cProfile, pstats, sys, types, os, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench  # noqa: E402  (sets up the package path)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a = types.SimpleNamespace(cell='gru', steps=300, warmup=20)
pr = cProfile.Profile()
orig = bench.time.perf_counter
state = {"on": False}


def hooked():
    # the timed region of train_leg starts at its first perf_counter() call and ends at the third
    state["n"] = state.get("n", 0) + 1
    if state["n"] == 1:
        pr.enable()
    elif state["n"] == 2:
        pr.disable()
    return orig()


bench.time = types.SimpleNamespace(perf_counter=hooked, time=bench.time.time, sleep=bench.time.sleep)
r = bench.train_leg(a, torch.device('cuda:0'), 0, 1, 'f32', 80, 1, B, 25, 300, 20, z_dim=100)
print({k: r[k] for k in ('ms_per_step', 'host_enqueue_ms_per_step')})
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
