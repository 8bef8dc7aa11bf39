"""Host logic of the weight-gradient tile FIFO (engine.WgradQueue, engine._Plan.hook_alias): the schedule that decides which output
tiles of which weight gradients (the reverse-mode products of attention.py:29-37,60-63 and ff.py:26-31, train.py:94-95) go into
which grouped launch.  No GPU: the queue and the plan are pure Python."""
import savit_amd  # noqa: F401
from savit_amd.engine import WgradQueue, _Plan


def _drain(tiles_per_layer, layers, cap, max_lag=None):
    """The engines' loop: push a layer's weights (processed last layer first), launch while a launch's worth is waiting."""
    q, launches = WgradQueue(cap, max_lag), []
    for l in range(layers - 1, -1, -1):
        for w, t in enumerate(tiles_per_layer):
            q.push((l, w), l, t)
        while q.pending() > 0 and (q.due(l) or l == 0):
            entries, done, oldest = q.take(q.cap)
            launches.append((l, entries, done, oldest))
    assert q.pending() == 0
    return launches


def test_launches_are_full_rounds_and_cover_every_tile_once():
    sizes = [36, 36, 9, 27]  # DeiT-B: W2, W1, Wo, Wqkv in 256 x 256 tiles
    launches = _drain(sizes, 12, 256)
    assert [sum(c for _, _, c in e) for _, e, _, _ in launches] == [256] * 5 + [16]  # every launch but the last is one tile per CU
    seen = {}
    for _, entries, _, _ in launches:
        for (l, w), t0, c in entries:
            rng = seen.setdefault((l, w), [])
            assert not rng or rng[-1][0] + rng[-1][1] == t0  # a weight cut between launches continues where it stopped
            rng.append((t0, c))
    assert len(seen) == 12 * 4
    for (l, w), rng in seen.items():
        assert rng[0][0] == 0 and sum(c for _, c in rng) == sizes[w]


def test_done_layers_and_lag():
    launches = _drain([36, 36, 9, 27], 12, 256)
    finished = [l for _, _, done, _ in launches for l in done]
    assert sorted(finished) == list(range(12)) and len(set(finished)) == 12  # each layer's DDP trigger fires behind exactly one launch
    for at_layer, entries, done, oldest in launches:
        assert oldest == max(l for (l, _), _, _ in entries)
        assert all(at_layer <= l <= oldest for l in done)
    # how far a launch reaches back sizes the cotangent rings (engine: wgrad_lag + 1 layers; the same dry run computes it).  With all
    # four weights of every layer queued it is 3 here; the engine diverts two d x d gradients (no 16-tile last launch) and gets 2.
    assert max(oldest - at for at, _, _, oldest in launches) == 3


def test_entry_cap_for_narrow_models():
    """CaiT-XXS-like: 10 tiles per layer - a launch is due once 64 (weight, tile range) entries wait, whatever the tile count."""
    launches = _drain([3, 3, 1, 3], 24, 256)
    assert all(len(e) <= WgradQueue.MAX_ENTRIES for _, e, _, _ in launches)
    assert sum(sum(c for _, _, c in e) for _, e, _, _ in launches) == 24 * 10


def test_max_lag_bounds_the_reach_back_with_partial_rounds():
    """Narrow models fill a round of 256 tiles only every 6-15 layers: with max_lag a launch goes out, as a partial round,
    once its oldest gradient has waited that many layers - the bound on deferred data-parallel triggers and on the ring depth."""
    for sizes, layers in (([3, 3, 1, 3], 24), ([12, 12, 3, 9], 12), ([6, 6, 2, 6], 12)):  # CaiT-XXS, DeiT-S (128 x 384 tiles), ViT-Ti
        free = _drain(sizes, layers, 256)
        capped = _drain(sizes, layers, 256, max_lag=3)
        assert max(o - at for at, _, _, o in free) > 3
        assert max(o - at for at, _, _, o in capped) <= 3
        assert sum(sum(c for _, _, c in e) for _, e, _, _ in capped) == layers * sum(sizes)
        finished = [l for _, _, done, _ in capped for l in done]
        assert sorted(finished) == list(range(layers))
    # where full rounds come often enough nothing changes (DeiT-B: lag 3 either way)
    assert _drain([36, 36, 9, 27], 12, 256, max_lag=3) == _drain([36, 36, 9, 27], 12, 256)


def test_reserved_cus_shrink_the_round():
    """A data-parallel rank leaves CUs to the resident all-reduce: a launch is then one tile per REMAINING CU."""
    launches = _drain([36, 36, 9, 27], 12, 224)
    counts = [sum(c for _, _, c in e) for _, e, _, _ in launches]
    assert counts[:-1] == [224] * (len(counts) - 1) and 0 < counts[-1] <= 224 and sum(counts) == 12 * 108


def test_hook_alias_defers_triggers_behind_the_group_launch():
    P = _Plan()
    fired = []
    hooks = {f"l{j}.ln1.bwd": (lambda j=j: fired.append(j)) for j in range(4)}
    P.hook_alias["wgrad.group.0.l3-l2"] = ["l3.ln1.bwd"]
    P.hook_alias["wgrad.group.1.l1-l0"] = ["l2.ln1.bwd", "l1.ln1.bwd"]
    assert P.hook_for(hooks, "l3.ln1.bwd") is None and P.hook_for(hooks, "l1.ln1.bwd") is None  # deferred
    P.hook_for(hooks, "wgrad.group.0.l3-l2")()
    P.hook_for(hooks, "wgrad.group.1.l1-l0")()
    P.hook_for(hooks, "l0.ln1.bwd")()  # not deferred: fires in place
    assert fired == [3, 2, 1, 0]
    assert P.hook_for({}, "l0.ln1.bwd") is None and P.hook_for(hooks, "qkv") is None
