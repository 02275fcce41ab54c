import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from boslam_amd import Context
from test_band_solve_gpu import band_system, gpu_solve
ctx = Context(0)
for n in (16, 32, 80, 144, 288):
    low, rhs = band_system(n, n - 1, 1)
    for mode in (4, 0):
        gpu_solve(ctx, low, rhs, n - 1, mode)
        t = time.perf_counter()
        for _ in range(50): gpu_solve(ctx, low, rhs, n - 1, mode)
        dt = (time.perf_counter() - t) / 50
        print(f"n {n} mode {mode}: {dt * 1e6:.0f} us per call (incl. H2D/D2H + sync)")
