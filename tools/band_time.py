import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, time
from boslam_amd import Context
from test_band_solve_gpu import band_system, gpu_solve, scipy_solve
ctx = Context(0)
low, rhs = band_system(30000, 245, 1)
ref = scipy_solve(low, rhs, 245)
for mode in (0, 1, 3):
    gpu_solve(ctx, low, rhs, 245, mode)
    t = time.perf_counter(); x, clk = gpu_solve(ctx, low, rhs, 245, mode); dt = time.perf_counter() - t
    print("mode", mode, "wall ms (incl. 59 MB H2D)", round(dt * 1e3, 1), "err", np.abs(x - ref).max(), "clocks us A,B,C,back | factor,dump,fwd", (clk[:7] / 100.0).round(0))
