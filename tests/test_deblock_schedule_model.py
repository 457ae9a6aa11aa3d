"""CPU model of K5's inter-wavefront protocol (h264decode_amd/csrc/k_deblock.hip): wavefront w of a picture's workgroup
runs the 8-row groups w, w + nwaves, ...; group g hands rows 12..15 of its last macroblock row to group g + 1 through
an LDS ring, ordered by two counters (prog: columns in the ring, cons: columns consumed) with back-pressure.  A column
enters the ring ONCE, final (one step after its horizontal pass, when the vertical pass of its right neighbour -- or the
end of the row -- has completed its last four columns).  The model replays the kernel's waits and ring accesses for the
launch plan the library computes (mi_deblock8_plan), under adversarial wavefront scheduling, and checks: no deadlock,
every slot read holds exactly the column the reader expects, and no slot is overwritten before it was read.  A wrong plan
or protocol hangs the GPU, so it is checked here.  (The banded kernel k_deblock_x keeps round 2's 4-row groups and its
provisional-then-final slots: second half of this file.)"""
import ctypes
import random

import pytest

ROWS = 8  # macroblock rows per group (k_deblock.hip: sub-rows of a wavefront)


def _plan(H, wmb, hmb):
    f = H.load_hooks().h264mi_internal_deblock_plan
    I32 = ctypes.c_int32
    f.restype = I32
    f.argtypes = [I32, I32] + [ctypes.POINTER(I32)] * 4 + [ctypes.POINTER(ctypes.c_int64)]
    nw, ring, rl, nb, lds = I32(), I32(), I32(), I32(), ctypes.c_int64()
    assert f(wmb, hmb, ctypes.byref(nw), ctypes.byref(ring), ctypes.byref(rl), ctypes.byref(nb), ctypes.byref(lds)) == 0
    assert lds.value <= 160 * 1024 and 1 <= nw.value <= 10
    return nw.value, ring.value, rl.value, nb.value


def _wave_program(w, nw, ring, ring_last, last_bufs, wmb, hmb, prog, cons, slots):
    """Generator: yields a predicate whenever the kernel would spin; everything else happens between yields."""
    ngroups = (hmb + ROWS - 1) // ROWS
    for g in range(w, ngroups, nw):
        last_sub = min(ROWS - 1, hmb - 1 - ROWS * g)
        feeds = g + 1 < ngroups
        out_last = w == nw - 1
        out_depth = ring_last if out_last else ring
        out_buf = (w, (g // nw) % last_bufs if out_last else 0)
        in_wave = (g + nw - 1) % nw
        in_last = in_wave == nw - 1
        in_depth = ring_last if in_last else ring
        in_buf = (in_wave, ((g - 1) // nw) % last_bufs if (in_last and g > 0) else 0)
        reuse = nw * last_bufs if out_last else nw
        if g >= reuse and feeds:
            yield lambda g=g, reuse=reuse: cons[g - reuse + 1] >= wmb
        for t in range(wmb + ROWS):
            xl = t - last_sub
            # top of the step: the group's first row takes the rows above column t from the group above (the copy), ...
            if g > 0 and t < wmb:
                yield lambda g=g, t=t: prog[g - 1] >= t + 1
                key = (in_buf, t % in_depth)
                assert slots.get(key, (None,))[:2] == (g - 1, t), ("reader finds the wrong column", g, t, slots.get(key))
                slots[key] = slots[key][:2] + (True,)  # consumed
                # ... says so behind the vertical pass
                cons[g] = t + 1
            # then rows 12..15 of column xl - 1 of the group's last row go into the ring (back-pressure first) and the column is published
            if feeds and 1 <= xl <= wmb:
                c = xl - 1
                if c >= out_depth:
                    yield lambda g=g, c=c, d=out_depth: cons[g + 1] >= c - d + 1
                key = (out_buf, c % out_depth)
                old = slots.get(key)
                assert old is None or old[2], ("slot overwritten before it was read", g, c, old)
                slots[key] = (g, c, False)
                prog[g] = xl


def _simulate(nw, ring, ring_last, last_bufs, wmb, hmb, rng):
    ngroups = (hmb + ROWS - 1) // ROWS
    prog, cons, slots = [0] * (ngroups + 1), [0] * (ngroups + 1), {}
    waves = [_wave_program(w, nw, ring, ring_last, last_bufs, wmb, hmb, prog, cons, slots) for w in range(nw)]
    blocked = [None] * nw
    done = [False] * nw
    while not all(done):
        runnable = [w for w in range(nw) if not done[w] and (blocked[w] is None or blocked[w]())]
        assert runnable, ("deadlock", nw, ring, ring_last, last_bufs, wmb, hmb, prog, cons)
        w = rng.choice(runnable) if rng.random() < 0.7 else runnable[-1 if rng.random() < 0.5 else 0]
        try:
            blocked[w] = next(waves[w])
        except StopIteration:
            done[w] = True
    for g in range(1, ngroups):
        assert cons[g] == wmb and prog[g - 1] == wmb


@pytest.mark.parametrize("wmb,hmb", [(1, 1), (2, 9), (11, 9), (20, 15), (45, 45), (80, 45), (120, 68), (240, 135), (17, 200), (300, 320), (512, 100), (512, 320), (3, 17), (120, 34)])
def test_deblock_protocol_is_deadlock_free_and_race_free(H, wmb, hmb):
    nw, ring, ring_last, last_bufs = _plan(H, wmb, hmb)
    for seed in range(3):
        _simulate(nw, ring, ring_last, last_bufs, wmb, hmb, random.Random(seed))


def test_model_detects_the_single_buffer_deadlock():
    """Three rounds with ONE whole-row buffer for the last wavefront and rows longer than the short rings can bridge: the
    last wavefront's group of round 2 waits for wavefront 0 to finish reading round 1's buffer, wavefront 0's group cannot
    finish because the groups below it are held back by that very group -- the model must report it."""
    with pytest.raises(AssertionError, match="deadlock"):
        _simulate(6, 16, 240, 1, 240, 135, random.Random(0))
    _simulate(6, 16, 240, 2, 240, 135, random.Random(0))


# ---------------------------------------------------------------------------------------------------------------------
# The banded kernels (k_deblock_x / k_deblock_b_x): a picture is spread over `nbands` workgroups.  A workgroup draws a ticket
# when it STARTS (whatever order the hardware starts workgroups in) and the ticket decides (picture, band); inside a band
# the LDS protocol above applies with one round; between bands the slots travel as {epoch, data} granules through a
# whole-row ring in global memory, so there is no back-pressure there.  The model runs the protocol with FEWER resident
# workgroups than the grid has (a workgroup only becomes resident when another one has exited) and adversarial scheduling:
# it must still finish, because a band only waits for the band above it, which drew an earlier ticket and is running.
def _band_plan(H, n_pics, wmb, hmb, max_wgs=256):
    f = H.load_hooks().h264mi_internal_band_plan
    I32 = ctypes.c_int32
    f.restype = I32
    f.argtypes = [I32] * 4 + [ctypes.POINTER(I32)] * 3 + [ctypes.POINTER(ctypes.c_int64)] + [ctypes.POINTER(I32)] * 2
    v = [I32() for _ in range(3)] + [ctypes.c_int64()] + [I32(), I32()]
    assert f(n_pics, wmb, hmb, max_wgs, *[ctypes.byref(x) for x in v]) == 0
    return tuple(x.value for x in v)  # k5_bands, k5_waves, k5_ring, k5_lds, k3_bands, k3_waves


def _band_wave_program(g, g0, g1, ring, wmb, hmb, prog, cons, slots, xring, pic, band, pband, epoch):
    ngroups = (hmb + 3) // 4
    last_sub = min(3, hmb - 1 - 4 * g)
    feeds = g + 1 < ngroups
    to_global = feeds and g == g1 - 1
    band_first = g == g0 and g > 0
    out_buf, in_buf = (pic, band, g - g0), (pic, band, g - g0 - 1)
    for t in range(wmb + 3):
        xl = t - last_sub
        if feeds and 1 <= xl < wmb:
            key = (out_buf, (xl - 1) % ring)
            assert slots[key][:2] == (g, xl - 1), ("fix-up hits a foreign slot", g, xl, slots[key])
            slots[key] = (g, xl - 1, True, slots[key][3])
            if to_global:
                xring[(pic, band, xl - 1)] = (epoch, g, xl - 1)
                slots[key] = slots[key][:3] + (True,)  # copied out: the slot may be reused
            else:
                prog[(pic, g)] = xl
        if g > 0 and t < wmb:
            if band_first:
                yield lambda t=t: xring.get((pic, pband, t), (None,))[0] == epoch
                assert xring[(pic, pband, t)] == (epoch, g - 1, t), ("granule of a foreign column", g, t, xring[(pic, pband, t)])
            else:
                yield lambda t=t: prog.get((pic, g - 1), 0) >= t + 1
                key = (in_buf, t % ring)
                assert slots.get(key, (None,))[:3] == (g - 1, t, True), ("reader finds the wrong column", g, t, slots.get(key))
                slots[key] = slots[key][:3] + (True,)
            cons[(pic, g)] = t + 1
        if feeds and not to_global and ring <= xl < wmb:
            yield lambda xl=xl: cons.get((pic, g + 1), 0) >= xl - ring + 1
        if feeds and 0 <= xl < wmb:
            key = (out_buf, xl % ring)
            old = slots.get(key)
            assert old is None or old[3], ("slot overwritten before it was read", g, xl, old)
            slots[key] = (g, xl, xl == wmb - 1, False)
            if xl == wmb - 1:
                if to_global:
                    xring[(pic, band, xl)] = (epoch, g, xl)
                else:
                    prog[(pic, g)] = wmb


def _simulate_banded(n_pics, nbands, nwaves, ring, wmb, hmb, resident, rng, stale_epoch=None):
    ngroups = (hmb + 3) // 4
    prog, cons, slots, xring = {}, {}, {}, {}
    epoch = 7
    if stale_epoch is not None:  # what an earlier launch left in the ring must never satisfy a reader
        for p in range(n_pics):
            for b in range(nbands):
                for c in range(wmb):
                    xring[(p, b, c)] = (stale_epoch, -1, c)
    next_ticket, total = 0, n_pics * nbands
    running = []  # [(waves, blocked)] of the resident workgroups
    finished_groups = 0
    while next_ticket < total or running:
        while next_ticket < total and len(running) < resident:  # a free slot: some workgroup starts and draws the next ticket
            pic, band = divmod(next_ticket, nbands)
            next_ticket += 1
            g0, g1 = band * ngroups // nbands, (band + 1) * ngroups // nbands
            assert g1 - g0 <= nwaves, ("band larger than the workgroup", g0, g1, nwaves)
            pband = band - 1
            while pband > 0 and pband * ngroups // nbands == (pband + 1) * ngroups // nbands:
                pband -= 1
            waves = [_band_wave_program(g, g0, g1, ring, wmb, hmb, prog, cons, slots, xring, pic, band, max(pband, 0), epoch) for g in range(g0, g1)]
            running.append([waves, [None] * len(waves), [False] * len(waves)])
        runnable = [(i, w) for i, (waves, blocked, done) in enumerate(running) for w in range(len(waves)) if not done[w] and (blocked[w] is None or blocked[w]())]
        if not runnable:
            assert all(all(d) for _, _, d in running), ("deadlock", n_pics, nbands, nwaves, ring, wmb, hmb, resident, next_ticket)
        else:
            i, w = rng.choice(runnable) if rng.random() < 0.7 else runnable[-1 if rng.random() < 0.5 else 0]
            try:
                running[i][1][w] = next(running[i][0][w])
            except StopIteration:
                running[i][2][w] = True
                finished_groups += 1
        running = [r for r in running if not all(r[2])]
    assert finished_groups == n_pics * ngroups
    for p in range(n_pics):
        for b in range(nbands - 1):
            if (b + 1) * ngroups // nbands > b * ngroups // nbands and (b + 1) * ngroups // nbands < ngroups:
                assert all(xring[(p, b, c)][0] == epoch for c in range(wmb))


@pytest.mark.parametrize("n_pics,wmb,hmb", [(1, 120, 68), (32, 120, 68), (3, 11, 9), (1, 1, 8), (2, 240, 135), (7, 20, 15), (100, 120, 68), (1, 300, 320), (5, 17, 200)])
def test_banded_deblock_protocol_finishes_with_few_resident_workgroups(H, n_pics, wmb, hmb):
    nbands, nwaves, ring, lds, _, _ = _band_plan(H, n_pics, wmb, hmb)
    ngroups = (hmb + 3) // 4
    if nbands == 1:
        assert n_pics * 2 > 256 or ngroups < 2 or ngroups > 12  # nothing to spread, or one band would need more than one round
        return
    assert n_pics * nbands <= 256 and nwaves * nbands >= ngroups and nwaves <= 12 and lds <= 160 * 1024
    small = min(n_pics, 2)  # the model is O(groups * columns): a few pictures are enough
    for resident in (1, 2, n_pics * nbands):
        for seed in range(2):
            _simulate_banded(small, nbands, nwaves, ring, wmb, hmb, resident, random.Random(seed), stale_epoch=6)


def test_banded_model_sees_a_band_that_waits_for_a_later_ticket():
    """If bands drew tickets in the opposite order (band b + 1 before band b) a single resident workgroup would wait for one
    that cannot start: the model must report that as a deadlock."""
    import unittest.mock as um
    real = divmod

    def flipped(t, nb):
        p, b = real(t, nb)
        return p, nb - 1 - b
    with um.patch("builtins.divmod", flipped):
        with pytest.raises(AssertionError, match="deadlock"):
            _simulate_banded(1, 4, 5, 16, 20, 68, 1, random.Random(0))
    _simulate_banded(1, 4, 5, 16, 20, 68, 1, random.Random(0))
