"""Properties of the interleaved macro-tile partition (mnv_partition: rounds of `world`, every root_period-th round without rank 0) over
random frame sizes, tile sizes, world sizes and periods -- the index arithmetic the multi-GPU path is "correct by construction" on:
the Python mirror (TilePartition, used by the harness and the gloo tests) against the C ABI, and the structural invariants."""
import numpy as np
import torch
from hypothesis import given, settings
from hypothesis import strategies as st


@settings(max_examples=150, deadline=None)
@given(w=st.integers(8, 4200), h=st.integers(8, 2400), world=st.integers(1, 16), tw8=st.integers(1, 32), th8=st.integers(1, 16),
       period=st.sampled_from([0, 2, 3, 4, 8, 11, 64]))
def test_partition_is_a_partition_and_matches_the_c_abi(mnv, w, h, world, tw8, th8, period):
    from mega_nerf_viewer_amd.multigpu import TilePartition

    tw, th = 8 * tw8, 8 * th8
    part = TilePartition(w, h, world, tw, th, period)
    owned = [part.tiles_of(r) for r in range(world)]
    # every macro tile has exactly one owner, local indices are 0 .. n-1 in deal order
    flat = sorted(m for tiles in owned for m in tiles)
    assert flat == list(range(part.n_macro))
    for r in range(world):
        assert [part.owner(m) for m in owned[r]] == [(r, j) for j in range(len(owned[r]))]
        assert len(owned[r]) == part.local_tiles(r) == mnv.partition_local_tiles((0, 0, w, h), r, world, tw, th, part.root_period)
    assert part.j_max == max(len(t) for t in owned)
    if world > 1 and part.root_period >= 2 and part.n_macro >= world * part.root_period:
        assert len(owned[0]) <= len(owned[1])         # the root is relieved: one tile fewer per full period (a ragged last period may give one back)
    # balance: no rank owns more than one tile per round above the smallest share of the non-root ranks
    rest = [len(t) for t in owned[1:]] or [len(owned[0])]
    assert max(rest) - min(rest) <= 1


@settings(max_examples=40, deadline=None)
@given(w=st.integers(8, 700), h=st.integers(8, 400), world=st.integers(1, 9), period=st.sampled_from([0, 2, 5, 8]), frames=st.integers(0, 3))
def test_unpermute_inverts_the_deal(mnv, w, h, world, period, frames):
    """gathered[rank][frame][local tile] -> frame: every pixel lands where the rank that owns its macro tile put it."""
    from mega_nerf_viewer_amd.multigpu import TilePartition

    part = TilePartition(w, h, world, 64, 24, period)
    lead = (frames,) if frames else ()
    f = max(frames, 1)
    frame = torch.arange(f * h * w, dtype=torch.float32).reshape(f, h, w, 1).repeat(1, 1, 1, 4)
    gathered = torch.full((world,) + lead + (part.j_max, 24, 64, 4), -1.0)
    for m in range(part.n_macro):
        r, j = part.owner(m)
        x0, y0, tw_, th_ = part.tile_rect(m)
        for k in range(f):
            dst = gathered[r][k][j] if frames else gathered[r][j]
            dst[:th_, :tw_] = frame[k, y0:y0 + th_, x0:x0 + tw_]
    out = part.unpermute(gathered)
    want = frame if frames else frame[0]
    assert torch.equal(out, want)
