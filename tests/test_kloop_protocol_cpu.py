"""The LDS-stage protocol of the pipelined k loops (k_gemm3x.hip HOIST = 3, k_gemm_bf16x.hip PIPE = 2), checked on the CPU.

These loops put the "next k tile has landed" barrier INSIDE a tile instead of in front of it, read the next tile's first fragments
behind that barrier, and let the DMA of a later tile overwrite a stage one barrier after its last read.  What can go wrong is a
race, which a GPU parity test only catches when the timing happens to expose it.  This file models the protocol -- every wave's
sequence of {issue DMA pieces, wait for own pieces, barrier, read stage} events, DMA pieces landing at ANY moment between their
issue and the issuing wave's covering wait, waves interleaved at random -- and asserts that (1) every read of tile t finds every
wave's piece of tile t in the stage, and (2) no piece of a later tile lands in a stage while some wave still has a read of the
older tile ahead of it.  Forms: "plain" = barrier in front of the tile (the default loops, two or three stages); "hoist" = barrier
in front of the last fragment row(s), the next tile's DMA issued in the first rows (k_gemm_bf16x.hip PIPE = 2 with two stages; the
round-2 HOIST = 1 / 2 loops of k_gemm3x.hip, since removed, were this form with two and three stages); "hoist3" = k_gemm3x.hip
HOIST = 3: barrier two rows early, the DMA of tile t + 2 issued behind it.  The event sequences are transcribed from the kernels
(S3Wave::tile3 / rows3 / mfmas3, BxWave::slots); tools/dev/isa_summary.py shows the same order in the compiled loops.
"""
import random

import pytest


def wave_program(form, nstg, n_t, mi):
    """One wave's event list.  Events: ("dma", tile, stage) one event = ALL this wave's pieces of that tile;
    ("wait", k) = s_waitcnt vmcnt leaving the k most recent dma events outstanding; ("bar",); ("read", tile, stage, what)."""
    ev = []
    stage = lambda t: t % nstg   # noqa: E731
    last = n_t - 1

    def dma(t):   # the tile after the last one is the last one again, fetched into the stage nobody reads any more
        ev.append(("dma", min(t, last), stage(t), t > last))

    if form == "plain":            # barrier in front of the tile (the loops these variants are measured against)
        dma(0)
        if nstg == 3:
            dma(1)
        for t in range(n_t):
            ev.append(("wait", nstg - 2))
            ev.append(("bar",))
            ev.append(("read", t, stage(t), "head"))
            dma(t + nstg - 1)
            ev.append(("read", t, stage(t), "rows"))
        return ev
    if form == "hoist3":           # k_gemm3x.hip HOIST = 3: two stages, tiles 0 and 1 before the loop, tile t + 2 issued behind tile t's barrier
        assert nstg == 2
        dma(0)
        dma(1)
        ev.append(("wait", 1))
        ev.append(("bar",))
        ev.append(("read", 0, stage(0), "head: fragments 0, 1, l and m planes"))
        for t in range(n_t):
            ev.append(("read", t, stage(t), "row 0: h planes, fragments 2, 3"))
            ev.append(("wait", 0))                                # row BR: every read of this stage has landed (lgkmcnt(0)), the next tile's pieces too
            ev.append(("bar",))
            ev.append(("read", t + 1, stage(t + 1), "row BR: next tile's fragments 0, 1"))
            dma(t + 2)                                            # rows BR .. MI - 1, into the stage this tile has just given up
            ev.append(("read", t + 1, stage(t + 1), "last row: next tile's l and m planes"))
        return ev
    # hoisted forms
    dma(0)
    if nstg == 3:
        dma(1)
    ev.append(("wait", nstg - 2))
    ev.append(("bar",))
    ev.append(("read", 0, stage(0), "head: fragments 0, 1 (+ l, m planes)"))
    for t in range(n_t):
        ev.append(("read", t, stage(t), "top: weight planes"))
        dma(t + nstg - 1)                                     # rows 0 .. DMA_ROWS - 1
        if mi > 2:
            ev.append(("read", t, stage(t), "fragments 2 .. MI-1"))
        ev.append(("wait", nstg - 2))                         # in front of the last row
        ev.append(("bar",))
        ev.append(("read", t + 1, stage(t + 1), "last row: next tile's fragments 0, 1 (+ l, m planes)"))
    return ev


def simulate(form, nstg, n_t, mi, n_waves, rng, program=None):
    progs = [(program or wave_program)(form, nstg, n_t, mi) for _ in range(n_waves)]
    pc = [0] * n_waves
    at_bar = [False] * n_waves
    content = [[None] * n_waves for _ in range(nstg)]         # content[stage][wave] = tile id of that wave's pieces
    inflight = [[] for _ in range(n_waves)]                   # per wave, in issue order: [tile, stage, dead]
    last = n_t - 1

    def future_reads(w, stage_id, tile):
        """does wave w still have a checked read of `tile` from `stage_id` ahead of it?"""
        return any(e[0] == "read" and e[2] == stage_id and e[1] == tile and e[1] <= last for e in progs[w][pc[w]:])

    def land(w, idx):
        tile, st, dead = inflight[w].pop(idx)
        old = content[st][w]
        if old is not None and old != tile:
            for v in range(n_waves):
                assert not future_reads(v, st, old), f"{form}/{nstg}: tile {tile} lands in stage {st} while wave {v} still reads tile {old}"
        content[st][w] = tile

    steps = 0
    while any(pc[w] < len(progs[w]) for w in range(n_waves)):
        steps += 1
        assert steps < 100000
        # DMA pieces land whenever they like (in order per wave: the hardware completes a wave's LDS-DMA in issue order)
        for w in range(n_waves):
            while inflight[w] and rng.random() < 0.3:
                land(w, 0)
        if all(at_bar[w] or pc[w] >= len(progs[w]) for w in range(n_waves)):
            for w in range(n_waves):
                if at_bar[w]:
                    at_bar[w] = False
                    pc[w] += 1
            continue
        w = rng.choice([w for w in range(n_waves) if not at_bar[w] and pc[w] < len(progs[w])])
        e = progs[w][pc[w]]
        if e[0] == "dma":
            inflight[w].append([e[1], e[2], e[3]])
            pc[w] += 1
        elif e[0] == "wait":
            while len(inflight[w]) > e[1]:
                land(w, 0)
            pc[w] += 1
        elif e[0] == "bar":
            at_bar[w] = True
        else:
            _, tile, st, what = e
            if tile <= last:      # reads past the last tile fetch garbage that is never used
                for v in range(n_waves):
                    assert content[st][v] == tile, f"{form}/{nstg}: wave {w} reads tile {tile} ({what}) from stage {st}, wave {v}'s piece holds {content[st][v]}"
            pc[w] += 1
    return True


@pytest.mark.parametrize("form,nstg,mi", [("plain", 2, 4), ("plain", 3, 2), ("hoist", 2, 4), ("hoist", 2, 2), ("hoist", 3, 2), ("hoist", 3, 4), ("hoist3", 2, 4), ("hoist3", 2, 2)])
def test_stage_protocol_has_no_race(form, nstg, mi):
    rng = random.Random(1234 + nstg * 10 + mi)
    for n_t in range(1, 8):
        for n_waves in (1, 2, 4):
            for _ in range(60):
                simulate(form, nstg, n_t, mi, n_waves, rng)


def test_the_model_catches_a_broken_protocol():
    """sanity of the checker itself: drop the wait in front of the in-tile barrier and the simulation must find the stale read."""
    def broken(form, nstg, n_t, mi):
        return [e for e in wave_program(form, nstg, n_t, mi) if e[0] != "wait"]
    rng = random.Random(7)
    with pytest.raises(AssertionError):
        for _ in range(200):
            simulate("hoist", 2, 5, 4, 4, rng, program=broken)


def test_a_barrier_in_front_of_the_tile_would_not_cover_the_hoisted_reads():
    """the other way to break it: keep the barrier at the top of the tile (plain placement) but read the next tile in the last row."""
    def broken(form, nstg, n_t, mi):
        ev = wave_program("plain", nstg, n_t, mi)
        out = []
        for e in ev:
            out.append(e)
            if e[0] == "read" and e[3] == "rows":
                out.append(("read", e[1] + 1, (e[1] + 1) % nstg, "hoisted without its barrier"))
        return out
    rng = random.Random(11)
    with pytest.raises(AssertionError):
        for _ in range(200):
            simulate("plain", 2, 5, 4, 4, rng, program=broken)
