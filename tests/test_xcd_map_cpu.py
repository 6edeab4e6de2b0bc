"""The XCD-aware work map of the GEMM launches (option xcd_map; kernels.hpp xcd_map_choose, k_common.hpp gemm_work_of_block).

The planner is host code and is called here through the C ABI (sdmi_plan_xcd_map, no device needed); the device-side decode
"block -> (M tile, N tile, split-K slice)" is ten lines of integer arithmetic and is restated below line by line.  Checked:
every work item of the launch is produced by exactly one block of the grid the launcher uses, whatever the shape; the blocks
of one XCD (block % 8) stay inside that XCD's sub-box; the plan is never worse balanced than the legacy map and never moves
more bytes than it; and on the shapes of the batch-1 UNet it moves what the docstring of Engine::choose_xcd_map says.
"""
import ctypes as C
import itertools
import random

import pytest

from stable_diffusion_burn_amd import _capi


CU_FLOPS = 2.3e14 / 256      # the split kernel's k loop on one CU
FABRIC = 4.0e12               # kernels.hpp kXcdFabricBytesPerSec


def plan(mt, nt, s, a, w, flops=None):
    lib = _capi.load_library()
    out = (C.c_int32 * 5)()
    if flops is None:
        flops = mt * nt * s * 2.0 * 128 * 160 * 32 * 10      # ten k tiles of a 128x160 tile per work item
    rc = lib.sdmi_plan_xcd_map(mt, nt, s, float(a), float(w), float(flops), CU_FLOPS, out)
    assert rc == 0
    return tuple(out)


def work_of_block(b, grid_x, mt, nt, s, pl):
    """k_common.hpp gemm_work_of_block, xcd_m > 0 branch (blockIdx.x = b, 1-D grid)."""
    xm, xn, ml, nl, zl = pl
    x, j = b & 7, b >> 3
    im = x % xm
    t = x // xm
    i_n = t % xn
    iz = t // xn
    mn = ml * nl
    z_l = j // mn
    r = j - z_l * mn
    tml = r // nl
    tm = im * ml + tml
    tn = i_n * nl + (r - tml * nl)
    z = iz * zl + z_l
    live = tm < mt and tn < nt and z < s and z_l < zl
    return tm, tn, z, live


def legacy_work_of_block(bx, bz, grid_x, mt, nt):
    tpx = grid_x >> 3
    lid = (bx & 7) * tpx + (bx >> 3)
    return lid // nt, lid % nt, bz, lid < mt * nt


def shapes():
    rng = random.Random(5)
    fixed = [(64, 1, 4), (16, 2, 4), (16, 4, 8), (4, 8, 8), (1, 10, 16), (1, 8, 24), (4, 10, 9), (64, 2, 1), (256, 1, 1), (3, 5, 7), (1, 1, 1), (1, 1, 40), (7, 1, 3)]
    rand = [(rng.randint(1, 70), rng.randint(1, 12), rng.randint(1, 40)) for _ in range(150)]
    return fixed + rand


@pytest.mark.parametrize("a,w", [(1.0, 100.0), (100.0, 1.0), (1.0, 1.0)])
def test_every_work_item_exactly_once(a, w):
    for mt, nt, s in shapes():
        pl = plan(mt, nt, s, a, w)
        xm, xn, ml, nl, zl = pl
        assert xm * xn in (1, 2, 4, 8) and 8 % (xm * xn) == 0
        xz = 8 // (xm * xn)
        assert ml * xm >= mt and nl * xn >= nt and zl * xz >= s
        grid_x = 8 * ml * nl * zl            # kernels.hpp gemm_grid
        seen = {}
        for b in range(grid_x):
            tm, tn, z, live = work_of_block(b, grid_x, mt, nt, s, pl)
            if not live:
                continue
            assert (tm, tn, z) not in seen, f"{(mt, nt, s)} plan {pl}: item {(tm, tn, z)} by blocks {seen[(tm, tn, z)]} and {b}"
            seen[(tm, tn, z)] = b
            # the XCD of a block (b % 8) owns one sub-box
            x = b & 7
            assert tm // ml == x % xm and tn // nl == (x // xm) % xn and z // zl == x // (xm * xn)
        assert len(seen) == mt * nt * s, f"{(mt, nt, s)} plan {pl}: {len(seen)} of {mt * nt * s} items covered"


def test_legacy_map_restated_covers_too():
    for mt, nt, s in shapes()[:40]:
        grid_x = ((mt * nt + 7) // 8) * 8
        seen = set()
        for bz in range(s):
            for bx in range(grid_x):
                tm, tn, z, live = legacy_work_of_block(bx, bz, grid_x, mt, nt)
                if live:
                    assert (tm, tn, z) not in seen
                    seen.add((tm, tn, z))
        assert len(seen) == mt * nt * s


def _model_time(mt, nt, s, a, w, flops, per, bytes_):
    return max(1.0, per / 32.0) * flops / (mt * nt * s) / CU_FLOPS + bytes_ / FABRIC


def _legacy_cost(mt, nt, s, a, w):
    """work items on the busiest XCD and operand bytes into the L2s under the legacy map: XCD x owns tiles [x tpx, (x + 1) tpx) of
    the n-fastest tile order, all slices."""
    tpx = (mt * nt + 7) // 8
    per = tpx * s
    bytes_ = 0.0
    for x in range(8):
        lids = range(x * tpx, min((x + 1) * tpx, mt * nt))
        if not len(lids):
            continue
        bytes_ += len({l // nt for l in lids}) / mt * a + len({l % nt for l in lids}) / nt * w
    return per, bytes_


def test_modelled_time_not_worse_than_the_legacy_map():
    """by the planner's own model (rounds x item time + bytes / fabric bandwidth) the chosen cut is at least as good as the legacy
    bands up to the rounding of the sub-box edges (the bands are not one of the candidate cuts, but the 8-way cut along M -- or
    along N when there is one M tile -- is their aligned twin), and on weight-heavy split-K launches it is much better."""
    worse = 0
    for (mt, nt, s), (a, w) in itertools.product(shapes(), [(1.0e6, 30.0e6), (30.0e6, 1.0e6)]):
        flops = mt * nt * s * 2.0 * 128 * 160 * 32 * 10
        xm, xn, ml, nl, zl = plan(mt, nt, s, a, w, flops)
        per, bytes_ = ml * nl * zl, -(-nt // nl) * a + -(-mt // ml) * w
        lper, lbytes = _legacy_cost(mt, nt, s, a, w)
        t, lt = _model_time(mt, nt, s, a, w, flops, per, bytes_), _model_time(mt, nt, s, a, w, flops, lper, lbytes)
        if t > 1.15 * lt:
            worse += 1
    assert worse == 0
    # (61, 4, 25), activations heavy: a 3 % better balance must not buy three times the bytes
    xm, xn, ml, nl, zl = plan(61, 4, 25, 30e6, 1e6)
    assert -(-4 // nl) * 30e6 + -(-61 // ml) * 1e6 <= 1.3 * _legacy_cost(61, 4, 25, 30e6, 1e6)[1]


def test_batch1_unet_shapes_move_each_operand_once():
    # 64x64 level, 320 -> 320 3x3, 128x320 tiles, 4 slices: legacy = activations + 8 x weights; plan = activations + 2 x weights
    assert plan(64, 1, 4, 10.5e6, 5.5e6, 2.0 * 8192 * 320 * 2880)[:2] == (2, 1)
    # 16x16 level, 1280 -> 1280 3x3 (88 MB of weight planes, 2.6 MB of activations), 128x160 tiles, 8 slices: every XCD one slice
    assert plan(4, 8, 8, 2.6e6, 88e6, 2.0 * 512 * 1280 * 11520)[:2] == (1, 1)
    # 8x8 level, one M tile, 16 slices
    assert plan(1, 10, 16, 0.7e6, 88e6, 2.0 * 128 * 1280 * 11520)[:2] == (1, 1)
    # batch 16 bf16, no split-K, activations are the big operand: bands of M tiles as before
    assert plan(256, 1, 1, 42e6, 1.8e6, 2.0 * 65536 * 320 * 2880)[:2] == (8, 1)
