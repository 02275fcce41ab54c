"""Phase clocks of k_ba_chol_dense (the panel wave): build boslam_amd/csrc/ba.hip with -DCD_CLOCKS into a side library and point
BH_LIB_PATH at it.  r02: n = 256: prologue 4.6 us, columns 105.8 us (factor + inverse of the diagonal tiles 69.0, waiting for their
last update 9.0), backward substitution + output 16.8 us - the panel wave's readlane-driven 16 x 16 factorisation is the chain."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from boslam_amd import Context
from test_band_solve_gpu import band_system, gpu_solve
ctx = Context(0)
for n in (256, 288):
    low, rhs = band_system(n, n - 1, 1)
    for _ in range(3): x, clk = gpu_solve(ctx, low, rhs, n - 1, 4)
    print(n, "prologue %.1f us, columns %.1f us (wait %.1f, factor + inverse + y %.1f = %.1f + %.1f + %.1f), backward+out %.1f us" %
          tuple(c / 100.0 for c in (clk[0], clk[1], clk[3], clk[4], clk[5], clk[6], clk[4] - clk[5] - clk[6], clk[2])))
