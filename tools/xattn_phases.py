"""Phase timing of csrc/xattn.hip at full size: the operator with debug_stop = 1 (first GEMM), 2 (+ q projection), 3 (+ attention), 0 (all),
GILL_OP_REPEAT launches each, HIP events around the repeats (the op's set-up kernels are timed separately with repeat = 1 and subtracted).
  python tools/xattn_phases.py C heads HW B"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import os, sys, torch
sys.path.insert(0, %r)
from gill_amd import ops, synth
C, heads, HW, B, stop = %d, %d, %d, %d, %d
M = B * HW
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
r = lambda s, std=1.0: (torch.randn(s, device=dev, generator=g) * std).bfloat16()
args = (r((M, C)), r((M, C)), r((C, C), C ** -0.5), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev),
        r((C, C), C ** -0.5), r((B, 77, C)), r((B, 77, C)), r((C, C), C ** -0.5), torch.zeros(C, device=dev))
def run():
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); ops.xattn_block(*args, heads, B, debug_stop=stop); e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3
run()
print('T', min(run() for _ in range(3)))
"""
a = [int(x) for x in sys.argv[1:5]]
for stop in (1, 2, 3, 0):
  t = {}
  for rep in (1, 21):
    r = subprocess.run([sys.executable, "-c", code % ((ROOT,) + tuple(a) + (stop,))], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GILL_OP_REPEAT=str(rep)))
    t[rep] = float([l for l in r.stdout.splitlines() if l.startswith("T")][0].split()[1]) if r.returncode == 0 else float("nan")
  print(f"geometry {a} debug_stop {stop}: {(t[21] - t[1]) / 20:.1f} us per launch", flush=True)
