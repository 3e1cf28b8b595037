"""Which ranges does the clip's sum-of-squares pass still read in the bench's train leg?  (monkeypatches ops.sumsq_partials)"""
import os, sys, runpy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from a3vlm_amd import ops, dp
calls = []
orig = ops.sumsq_partials
def spy(x, out):
    calls.append(x.numel())
    return orig(x, out)
ops.sumsq_partials = spy
orig_slots = dp.GradSquareSums.wgrad_slots
cov = collections.Counter()
def spy_slots(self, grad, M, N):
    r = orig_slots(self, grad, M, N)
    cov["slots" if r is not None else "none"] += 1
    return r
dp.GradSquareSums.wgrad_slots = spy_slots
sys.argv = ["bench.py", "--legs", "train", "--steps", "1", "--warmup", "1"]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
print("wgrad_slots:", dict(cov), file=sys.stderr)
c = collections.Counter(calls)
print("sumsq_partials calls by numel:", sorted(c.items())[-12:], "total calls", len(calls), file=sys.stderr)
