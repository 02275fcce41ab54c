"""The two reduced-camera-system solvers (panel kernel, on-chip sliding-window kernel) against scipy on random banded SPD
systems, including sizes that are not multiples of the 16x16 tile and the widest band the window kernel takes."""
import ctypes as C
import numpy as np
import pytest
from scipy.linalg import solveh_banded

from boslam_amd import _lib

pytestmark = pytest.mark.gpu


def band_system(n, bw, seed):
    rng = np.random.default_rng(seed)
    low = rng.normal(size=(n, bw + 1))                    # low[r, bw - k] = element (r, r - k)
    for k in range(1, bw + 1):
        low[:k, bw - k] = 0.0
    low[:, bw] = np.abs(low[:, :bw]).sum(axis=1) + np.abs(low[:, bw]) + 1.0 + 2.0 * bw ** 0.5   # diagonally dominant enough
    # make strictly SPD: add column sums too
    colsum = np.zeros(n)
    for k in range(1, bw + 1):
        colsum[:n - k] += np.abs(low[k:, bw - k])
    low[:, bw] += colsum
    rhs = rng.normal(size=n)
    return np.ascontiguousarray(low), rhs


def scipy_solve(low, rhs, bw):
    n = len(rhs)
    ab = np.zeros((bw + 1, n))                            # scipy lower form: ab[k, j] = a[j + k, j]
    for k in range(bw + 1):
        ab[k, :n - k] = low[k:, bw - k]
    return solveh_banded(ab, rhs, lower=True)


def gpu_solve(ctx, low, rhs, bw, mode):
    n = len(rhs)
    x = np.zeros(n); clocks = np.zeros(8, np.int64)
    ctx.check(_lib.lib().bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), n, bw, mode, _lib.ptr(x), _lib.ptr(clocks)))
    return x, clocks


@pytest.mark.parametrize("n,bw", [(300, 15), (777, 77), (1000, 245), (2048, 251), (2500, 256), (530, 130), (4099, 245)])
def test_window_and_panel_kernels_match_scipy(ctx, n, bw):
    low, rhs = band_system(n, bw, seed=n + bw)
    ref = scipy_solve(low, rhs, bw)
    modes = [0, 1] + ([3] if (bw + 15) // 16 == 16 and (n + 15) // 16 >= 40 else [])
    for mode in modes:                                   # panel, window, two-way window
        x, _ = gpu_solve(ctx, low, rhs, bw, mode)
        assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), (mode, np.abs(x - ref).max())


def test_not_positive_definite_is_reported(ctx):
    low, rhs = band_system(600, 40, seed=1)
    low[300, 40] = -5.0
    for mode in (0, 1):
        x = np.zeros(600)
        rc = _lib.lib().bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), 600, 40, mode, _lib.ptr(x), None)
        assert rc == -5
    # two-way mode: a bad pivot in the top part, in the bottom part and inside the separator - every case must return (no
    # workgroup left waiting for the other) and report the failure
    n, bw = 1600, 245
    for bad_row in (100, 1500, 800):
        low, rhs = band_system(n, bw, seed=2)
        low[bad_row, bw] = -5.0
        x = np.zeros(n)
        rc = _lib.lib().bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), n, bw, 3, _lib.ptr(x), None)
        assert rc == -5, bad_row


@pytest.mark.parametrize("n,bw", [(288, 287), (256, 255), (16, 15), (17, 16), (100, 99), (271, 270), (288, 40), (5, 4), (1, 0)])
def test_on_chip_dense_kernel_matches_scipy(ctx, n, bw):
    """k_ba_chol_dense (mode 4): the whole factor in registers / LDS - the local-BA window (n = 288 at configs[2]) and the 256 x 256
    separator system of the two-way solve; sizes that are not multiples of the tile, a banded matrix, the 1 x 1 case."""
    low, rhs = band_system(n, bw, seed=3 * n + bw)
    ref = scipy_solve(low, rhs, bw) if bw > 0 else rhs / low[:, 0]
    x, _ = gpu_solve(ctx, low, rhs, bw, 4)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), np.abs(x - ref).max()
    x2, _ = gpu_solve(ctx, low, rhs, bw, 4)
    assert np.array_equal(x, x2)                                        # fixed summation order: bitwise reproducible


def test_on_chip_dense_kernel_reports_a_bad_pivot(ctx):
    for bad in (0, 100, 287):
        low, rhs = band_system(288, 287, seed=9)
        low[bad, 287] = -5.0
        x = np.ones(288)
        rc = _lib.lib().bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), 288, 287, 4, _lib.ptr(x), None)
        assert rc == -5 and not x.any()


@pytest.mark.parametrize("n,bw", [(100, 20), (256, 255), (257, 30), (600, 245), (1000, 256), (2500, 245), (4099, 130), (30000, 245)])
def test_block_cyclic_reduction_matches_scipy(ctx, n, bw):
    """mode 5: the 256-block cyclic reduction (k_ba_chol_dense<true> / k_bcr_trsm / k_bcr_update / k_bcr_back) - one block, two
    blocks, chains of odd and even length at several levels, sizes that are not multiples of 256, configs[4]'s 30 000 x 245."""
    low, rhs = band_system(n, bw, seed=7 * n + bw)
    ref = scipy_solve(low, rhs, bw)
    x, _ = gpu_solve(ctx, low, rhs, bw, 5)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), np.abs(x - ref).max()
    x2, _ = gpu_solve(ctx, low, rhs, bw, 5)
    assert np.array_equal(x, x2)                                        # fixed summation order: bitwise reproducible


def test_block_cyclic_reduction_reports_a_bad_pivot(ctx):
    n, bw = 3000, 245
    for bad_row in (100, 700, 2900):                                    # blocks eliminated at different levels
        low, rhs = band_system(n, bw, seed=11)
        low[bad_row, bw] = -5.0
        x = np.ones(n)
        rc = _lib.lib().bh_ba_debug_band_solve(ctx.h, _lib.ptr(low), _lib.ptr(rhs), n, bw, 5, _lib.ptr(x), None)
        assert rc == -5, bad_row
        assert not x.any()
