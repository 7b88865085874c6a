"""The run layout of conv2d_widep_kernel (shadernn_amd/csrc/conv2d_widep_f16.hip: `seg_extras`, `run_of_tile`, the block's first tile), restated in Python
and checked exhaustively on the host: the kernel's hand-off of statistics records (one record per (block, image), the block that writes an image's last
record folds it) relies on every block being able to compute, from the launch's scalars alone, which runs touch an image.  Properties: every tile belongs
to exactly one run; `run_of_tile` inverts the assignment; the tiles / grid remainder goes to the first-dispatched blocks (physical index < remainder), i.e.
to every XCD alike; the runs that touch an image are consecutive logical indices and never more than the plan's `recsMax`.  The GPU tests
(tests/test_conv_widep_gpu.py, grids 1 / 3 / 7 / 8 / 16) run the device code itself."""
import random

import pytest


def _layout(tiles, grid):
    base, rem = tiles // grid, tiles % grid
    nseg = 8 if grid % 8 == 0 else 1
    seg_runs = grid // nseg

    def extras(x):  # runs of segment x that are one tile longer
        if nseg == 1:
            return rem
        return min(seg_runs, (rem - x + 7) >> 3) if rem > x else 0

    def run_of_tile(t):
        x, start = 0, 0
        while x + 1 < nseg:
            nxt = start + seg_runs * base + extras(x)
            if t < nxt:
                break
            start, x = nxt, x + 1
        ex, tt = extras(x), t - start
        cut = ex * (base + 1)
        return x * seg_runs + (tt // (base + 1) if tt < cut else ex + (tt - cut) // base)

    runs = {}
    for b in range(grid):  # physical block index: XCD b % 8
        x, j = (b & 7, b >> 3) if nseg == 8 else (0, b)
        first = x * seg_runs * base + j * base + min(j, extras(x)) + sum(extras(i) for i in range(x))
        runs[b] = (x * seg_runs + j, first, base + (1 if j < extras(x) else 0))
    return base, rem, runs, run_of_tile


CASES = [(t, g) for g in (1, 3, 7, 8, 16, 24, 40, 64, 511, 512) for t in (g, g + 1, g + 7, 2 * g + 3, 8 * g + 5)] + [(4048, 512), (4224, 512), (4400, 512), (4576, 512)]


@pytest.mark.parametrize("tiles,grid", CASES)
def test_every_tile_belongs_to_one_run_and_run_of_tile_inverts_it(tiles, grid):
    base, rem, runs, run_of_tile = _layout(tiles, grid)
    owner = [None] * tiles
    for b, (logical, first, n) in runs.items():
        assert (n == base + 1) == (b < rem), "the remainder goes to the first-dispatched blocks"
        for t in range(first, first + n):
            assert owner[t] is None
            owner[t] = logical
            assert run_of_tile(t) == logical
    assert all(o is not None for o in owner)
    assert sorted(r[0] for r in runs.values()) == list(range(grid)), "logical run indices are a permutation"
    assert owner == sorted(owner), "logical order is the tile order"


def test_runs_touching_an_image_are_consecutive_and_fit_the_record_slots():
    rng = random.Random(5)
    for _ in range(300):
        per_image, images = rng.randint(1, 300), rng.randint(1, 17)
        tiles = per_image * images
        grid = min(tiles, rng.choice([1, 3, 7, 8, 16, 64, 512]))
        base, _, runs, run_of_tile = _layout(tiles, grid)
        recs_max = per_image // base + 2  # make_conv2d_widep_plan
        for n in range(images):
            first, last = run_of_tile(n * per_image), run_of_tile((n + 1) * per_image - 1)
            touching = sorted({run_of_tile(t) for t in range(n * per_image, (n + 1) * per_image)})
            assert touching == list(range(first, last + 1))
            assert last - first + 1 <= recs_max, (tiles, grid, per_image)
