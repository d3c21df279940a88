"""The LDS ring / barrier protocol of flash_fwd8_kernel (csrc/attention_impl.inc), checked on CPU by simulation.

The kernel cannot be thread-traced on this pool, so its hazard argument (the sigma timeline in its header comment) is
restated here as a small model and CHECKED for every tile count: eight waves run their programs between workgroup
barriers; inside one barrier interval everything may interleave arbitrarily, so

  * a (ring, slot) written by any wave in an interval must not be read or written -- with another tile -- by any OTHER
    wave in the same interval (and not read by the same wave AFTER its write);
  * a read of tile T from (ring, slot) must find that every wave's part of T was written there in EARLIER intervals and
    nothing else since;
  * every wave passes the same number of barriers (a mismatch is a hang on the GPU).

Two forms are modelled, exactly as the source orders them: the shipped MIXED segment (all waves in phase, one barrier
per tile) and the two-group experiment build (MK_F8_MIXED = 0: waves 4-7 one segment behind waves 0-3, two barriers per
tile).  The model is deliberately written from the kernel's loop structure, not from its
comments: K ring slot = t & 1, V ring slot = t & 1 at offset KBYTES, validity ring slot = t & 3; K(t + 1) / V(t) are
written behind the products of tile t; the first tile multiplies P = 0 with the zero-filled V slot 1; the last P V runs
behind the loop.  A deliberately broken variant (one-deep K / V ring) must be caught."""
import pytest

WAVES = 8


def _program(w, ntiles, mixed, mask_depth=4, kv_depth=2):
    """list of intervals for wave w; an interval = list of ('r' | 'w', ring, slot, tile) in program order, a barrier
    after each interval"""
    late = (w >= 4) and not mixed
    prog = []
    # prologue: K(0) + validity(0) into slot 0, zero tile 'Z' into V slot 1
    pro = []
    if ntiles > 0:
        pro.append(("w", "K", 0, 0))
        if w == 0:
            pro.append(("w", "M", 0, 0))
    pro.append(("w", "V", (-1) % kv_depth, "Z"))
    prog.append(pro)
    if late:
        prog.append([])                                   # the extra barrier of the late group

    def products(t, with_qk=True):
        ops = []
        if with_qk:
            ops.append(("r", "M", t % mask_depth, t))
            ops.append(("r", "K", t % kv_depth, t))
        ops.append(("r", "V", (t - 1) % kv_depth, t - 1 if t > 0 else "Z"))
        return ops

    def staging(t):
        ops = []
        if t + 1 < ntiles:
            ops.append(("w", "K", (t + 1) % kv_depth, t + 1))
            if w == 0:
                ops.append(("w", "M", (t + 1) % mask_depth, t + 1))
        ops.append(("w", "V", t % kv_depth, t))
        return ops

    for t in range(ntiles):
        prog.append(products(t) + staging(t))             # matrix (or mixed) segment, barrier
        if not mixed:
            prog.append([])                               # softmax segment: no LDS traffic, barrier
    tail = []
    if ntiles > 0:
        tail.append(("r", "V", (ntiles - 1) % kv_depth, ntiles - 1))   # the last P V
    prog.append(tail)
    if not mixed and not late:
        prog.append([])                                   # pairs with the last barrier of the late group
    return prog


def _simulate(ntiles, mixed, mask_depth=4, kv_depth=2):
    progs = [_program(w, ntiles, mixed, mask_depth, kv_depth) for w in range(WAVES)]
    n = {len(p) for p in progs}
    assert len(n) == 1, f"barrier counts differ between waves: {sorted(n)} (a hang)"
    state = {}                                            # (ring, slot) -> (tile, set of waves that wrote their part)
    writers_needed = {"K": set(range(WAVES)), "V": set(range(WAVES)), "M": {0}}
    for k in range(n.pop()):
        writes = {}                                       # (ring, slot) -> {wave: tile}
        reads = {}                                        # (ring, slot) -> [(wave, tile, position, wrote_before)]
        for w, p in enumerate(progs):
            wrote = set()
            for kind, ring, slot, tile in p[k]:
                key = (ring, slot)
                if kind == "w":
                    writes.setdefault(key, {})[w] = tile
                    wrote.add(key)
                else:
                    reads.setdefault(key, []).append((w, tile, key in wrote))
        for key, rs in reads.items():
            for w, tile, after_own_write in rs:
                assert not after_own_write, f"interval {k}: wave {w} reads {key} after writing it"
                others = {x for x in writes.get(key, {}) if x != w}
                assert not others, f"interval {k}: wave {w} reads {key} (tile {tile}) while waves {sorted(others)} write it"
                assert key in state, f"interval {k}: wave {w} reads {key} before anything was written"
                have, who = state[key]
                assert have == tile, f"interval {k}: wave {w} expects tile {tile} in {key}, finds {have}"
                assert who >= writers_needed[key[0]], f"interval {k}: tile {tile} in {key} incomplete: {sorted(who)}"
            if key in writes:                             # (a wave may read, then overwrite, a slot inside one interval)
                pass
        for key, ws in writes.items():
            tiles = set(ws.values())
            assert len(tiles) == 1, f"interval {k}: {key} written with different tiles {tiles}"
            tile = tiles.pop()
            old = state.get(key)
            if old is not None and old[0] == tile:
                state[key] = (tile, old[1] | set(ws))
            else:
                state[key] = (tile, set(ws))
    return True


@pytest.mark.parametrize("mixed", [True, False])
def test_ring_protocol_is_hazard_free_for_every_tile_count(mixed):
    for ntiles in range(0, 41):
        assert _simulate(ntiles, mixed)


@pytest.mark.parametrize("mixed", [True, False])
def test_the_model_catches_a_one_deep_ring(mixed):
    """negative control: with ONE K / V slot a wave stages tile t + 1 into the slot its neighbours still read tile t
    from -- the simulation must flag it (otherwise a green run above would mean nothing)"""
    with pytest.raises(AssertionError, match="while waves|expects tile"):
        for ntiles in range(2, 12):
            _simulate(ntiles, mixed, kv_depth=1)


def test_a_two_deep_validity_ring_would_do_since_the_bytes_are_read_in_the_matrix_segment():
    """the first version of the two-group form read the validity bytes in the SOFTMAX segment, where a 2-deep ring is
    overwritten by wave 0 of the early group while the late group still reads it (hence 4 slots); since the bytes are read
    under the MFMAs of the matrix segment two slots are hazard-free in both forms.  Four ship (256 bytes of LDS)."""
    for mixed in (True, False):
        for ntiles in range(0, 20):
            assert _simulate(ntiles, mixed, mask_depth=2)
