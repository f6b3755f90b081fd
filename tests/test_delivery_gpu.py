"""Output delivery of diee_self_play / diee_self_play_multi (alpha_parallel.rs:215-230: the call returns all_memories): every
move-step's flushed games are listed, relabelled and gathered on the device and copied into page-locked host arrays on a
second stream while the next move-steps search.  The records themselves are held to the oracle bit for bit elsewhere
(tests/test_search_gpu.py, tests/test_parity_baseline_sizes_gpu.py: every one of those calls fetches); here: the mechanics --
staging chunks, growth of the host arrays, the block pool, views without a copy -- change nothing."""
import numpy as np
import pytest

import diee_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    yield e
    e.close()


def same(a, b):
    return all(a[k].tobytes() == b[k].tobytes() for k in ("outcome", "ps", "state", "game"))


def test_chunked_staging_and_grown_host_arrays_deliver_the_same_records(eng, monkeypatch):
    cfg = diee_amd.MctsConfig(iterations=6, c=2.0, round_limit=60, dir_alpha=0.3, dir_eps=0.25)
    ref = eng.self_play_parallel(48, cfg, 1.25, seed=5)
    n = len(ref["outcome"])
    assert n == ref["stats"]["fragments"] > 48 * 20
    assert ref["stats"]["deliver_bytes"] == n * (1352 * 4 + 144 * 4 + 1 + 4)
    # 7-row staging chunks (every delivery takes many gather + copy rounds) and host arrays sized for 1 record per game
    # (they grow several times in mid-flight: the copy stream is drained, the rows so far move to a bigger block)
    eng.set_options(deliver_stage_rows=7, deliver_rows_per_game=1)
    try:
        out = eng.self_play_parallel(48, cfg, 1.25, seed=5)
    finally:
        eng.set_options(deliver_stage_rows=16384, deliver_rows_per_game=128)
    assert same(out, ref)
    # the round limit flushes (Q18: a game flushed twice in one step) through the same path
    cfg2 = diee_amd.MctsConfig(iterations=4, c=2.0, round_limit=9, dir_alpha=0.3, dir_eps=0.25)
    a = eng.self_play_parallel(40, cfg2, 1.25, seed=8)
    eng.set_option("deliver_stage_rows", 5)
    try:
        b = eng.self_play_parallel(40, cfg2, 1.25, seed=8)
    finally:
        eng.set_option("deliver_stage_rows", 16384)
    assert same(a, b) and (a["outcome"] == 0).all() and 40 * 6 <= len(a["outcome"]) <= 40 * 9      # (a skipped turn leaves no record)


def test_views_of_the_engine_owned_arrays_and_the_block_pool(eng):
    cfg = diee_amd.MctsConfig(iterations=5, c=2.0, round_limit=50, dir_alpha=0.3, dir_eps=0.25)
    ref = eng.self_play_parallel(16, cfg, 1.25, seed=21)
    v1 = eng.self_play_parallel(16, cfg, 1.25, seed=21, copy=False)        # views: valid until free()
    v2 = eng.self_play_parallel(16, cfg, 1.25, seed=22, copy=False)        # a second call while the first result is still held
    assert same(v1, ref) and not same(v2, ref)
    keep = {k: v2[k].copy() for k in ("outcome", "ps", "state", "game")}
    v1["free"](); v1["free"]()                                              # idempotent; the blocks go back to the pool
    assert "ps" not in v1
    v3 = eng.self_play_parallel(16, cfg, 1.25, seed=21, copy=False)        # takes pooled blocks again
    assert same(v3, ref) and same(v2, keep)                                 # ... and never the ones v2 still holds
    v2["free"](); v3["free"]()
    # fetch=False: counted, not delivered
    st = eng.self_play_parallel(16, cfg, 1.25, seed=21, fetch=False)["stats"]
    assert st["fragments"] == len(ref["outcome"]) and st["deliver_bytes"] == 0


def test_pipelined_batches_deliver_per_batch_what_single_calls_deliver(eng):
    """diee_self_play_multi: every batch's records land in its own arrays, in its own order (the batches' flushes of one
    move-step share the staging buffer one after the other)"""
    cfg = diee_amd.MctsConfig(iterations=5, c=2.0, round_limit=50, dir_alpha=0.3, dir_eps=0.25)
    eng.set_invariant_nn(True)                                              # (network output independent of what shares the launch)
    try:
        batches = [(12, 0, 31), (20, 100, 32), (7, 200, 33)]
        multi = eng.self_play_multi(batches, cfg, 1.25)
        for (n, first, seed), m in zip(batches, multi):
            one = eng.self_play_parallel(n, cfg, 1.25, seed=seed, first_game_id=first)
            assert same(m, one) and m["stats"]["fragments"] == len(one["outcome"])
            assert m["game"].min() >= first and m["game"].max() < first + n
    finally:
        eng.set_invariant_nn(False)


def test_growth_at_batch_scale(eng, monkeypatch):
    """512 games played to completion (~55 k records, 330 MB): the host arrays sized for 16 records per game grow several times while
    copies are in flight, the staging buffer of 1 000 rows takes tens of rounds per busy move-step -- the same bytes as the default"""
    cfg = diee_amd.MctsConfig(iterations=3, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    ref = eng.self_play_parallel(512, cfg, 1.25, seed=77, copy=False)
    assert len(ref["outcome"]) == ref["stats"]["fragments"] > 512 * 60
    eng.set_options(deliver_rows_per_game=16, deliver_stage_rows=1000)
    try:
        out = eng.self_play_parallel(512, cfg, 1.25, seed=77, copy=False)
    finally:
        eng.set_options(deliver_stage_rows=16384, deliver_rows_per_game=128)
    assert same(out, ref)
    # order of the records: by the move-step that removed the game, then by game (a game's records are contiguous and in play order)
    g = ref["game"]
    starts = np.flatnonzero(np.r_[True, g[1:] != g[:-1]])
    assert len(starts) == len(np.unique(g)) == 512                          # every game exactly one contiguous run
    out["free"](); ref["free"]()
