"""CPU model of K5's inter-wavefront protocol (h264decode_amd/csrc/k_deblock.hip): wavefront w of a picture's workgroup
runs the 4-row groups w, w + nwaves, ...; group g hands the bottom rows of its last macroblock row to group g + 1 through
an LDS ring, ordered by two counters (prog: columns final, cons: columns consumed) with back-pressure.  The model replays
the kernel's waits and ring accesses for the launch plan the library computes (mi_deblock_plan), under adversarial
wavefront scheduling, and checks: no deadlock, every slot read holds exactly the column the reader expects in its final
state, and no slot is overwritten before it was read.  A wrong plan or protocol hangs the GPU, so it is checked here."""
import ctypes
import random

import pytest


def _plan(H, wmb, hmb):
    f = H.load().h264mi_internal_deblock_plan
    I32 = ctypes.c_int32
    f.restype = I32
    f.argtypes = [I32, I32] + [ctypes.POINTER(I32)] * 4 + [ctypes.POINTER(ctypes.c_int64)]
    nw, ring, rl, nb, lds = I32(), I32(), I32(), I32(), ctypes.c_int64()
    assert f(wmb, hmb, ctypes.byref(nw), ctypes.byref(ring), ctypes.byref(rl), ctypes.byref(nb), ctypes.byref(lds)) == 0
    return nw.value, ring.value, rl.value, nb.value


def _wave_program(w, nw, ring, ring_last, last_bufs, wmb, hmb, prog, cons, slots):
    """Generator: yields ('wait', predicate) whenever the kernel would spin; everything else happens between yields."""
    ngroups = (hmb + 3) // 4
    for g in range(w, ngroups, nw):
        last_sub = min(3, hmb - 1 - 4 * g)
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
        for t in range(wmb + 3):
            xl = t - last_sub
            # 3a: columns 12..15 of column xl - 1 complete its slot, then the column is published
            if feeds and 1 <= xl < wmb:
                key = (out_buf, (xl - 1) % out_depth)
                assert slots[key][:2] == (g, xl - 1), ("fix-up hits a foreign slot", g, xl, slots[key])
                slots[key] = (g, xl - 1, True, slots[key][3])
                prog[g] = xl
            # 3b: sub-row 0 takes the rows above column t from the group above
            if g > 0 and t < wmb:
                yield lambda g=g, t=t: prog[g - 1] >= t + 1
                key = (in_buf, t % in_depth)
                assert slots.get(key, (None,))[:3] == (g - 1, t, True), ("reader finds the wrong column", g, t, slots.get(key))
                slots[key] = slots[key][:3] + (True,)  # consumed
                cons[g] = t + 1
            # 5: back-pressure, then the group's last row writes its bottom rows into the slot of its column
            if feeds and out_depth <= xl < wmb:
                yield lambda g=g, xl=xl, d=out_depth: cons[g + 1] >= xl - d + 1
            if feeds and 0 <= xl < wmb:
                key = (out_buf, xl % out_depth)
                old = slots.get(key)
                assert old is None or old[3], ("slot overwritten before it was read", g, xl, old)
                slots[key] = (g, xl, xl == wmb - 1, False)  # final at once only at the end of the row
                if xl == wmb - 1:
                    prog[g] = wmb


def _simulate(nw, ring, ring_last, last_bufs, wmb, hmb, rng):
    ngroups = (hmb + 3) // 4
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


@pytest.mark.parametrize("wmb,hmb", [(1, 1), (2, 9), (11, 9), (20, 15), (45, 45), (80, 45), (120, 68), (240, 135), (17, 200), (300, 320), (512, 100), (512, 320)])
def test_deblock_protocol_is_deadlock_free_and_race_free(H, wmb, hmb):
    nw, ring, ring_last, last_bufs = _plan(H, wmb, hmb)
    for seed in range(3):
        _simulate(nw, ring, ring_last, last_bufs, wmb, hmb, random.Random(seed))


def test_model_detects_the_single_buffer_deadlock():
    """Three rounds with ONE whole-row buffer for the last wavefront and rows longer than the short rings can bridge: the
    last wavefront's group of round 2 waits for wavefront 0 to finish reading round 1's buffer, wavefront 0's group cannot
    finish because the groups below it are held back by that very group -- the model must report it."""
    with pytest.raises(AssertionError, match="deadlock"):
        _simulate(12, 16, 240, 1, 240, 135, random.Random(0))
    _simulate(12, 16, 240, 2, 240, 135, random.Random(0))
