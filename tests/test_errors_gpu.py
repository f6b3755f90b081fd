"""GPU tests of the boundary's error behaviour and edge sizes: the reference panics (assert!/unwrap); the C ABI
returns status codes and never truncates silently."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_inputs_are_no_ops(oracle):
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    s0 = np.zeros(0, dtype=diee_amd.BG_STATE)
    p, c = e.get_valid_moves(s0)
    assert p.shape == (0, 256, 4) and c.shape == (0,)
    assert e.encode(s0, np.zeros((0, 4), np.int8)).shape == (0,)
    assert e.decode(s0, np.zeros(0, np.uint32)).shape == (0, 4)
    assert e.as_tensor(s0).shape == (0, 144)
    pol, val = e.forward_t(s0)
    assert pol.shape == (0, 1352) and val.shape == (0,)
    r = e.alpha_mcts_parallel(s0, diee_amd.MctsConfig.default(4))
    assert r["probs"].shape == (0, 1352)
    e.close()


def test_status_codes_instead_of_panics(oracle, monkeypatch):
    import diee_amd
    e = diee_amd.Engine(0)
    roots = oracle.random_walk_states(1, 1)[:4]
    cfg = diee_amd.MctsConfig.default(4)
    with pytest.raises(diee_amd.DieeError) as ei:               # no weights yet
        e.alpha_mcts_parallel(roots, cfg)
    assert ei.value.status == diee_amd.ERR_NO_WEIGHTS
    with pytest.raises(diee_amd.DieeError) as ei:
        e.forward_t(roots)
    assert ei.value.status == diee_amd.ERR_NO_WEIGHTS
    with pytest.raises(diee_amd.DieeError) as ei:               # wrong blob size
        e.load_weights(np.zeros(10, np.float32))
    assert ei.value.status == diee_amd.ERR_ARG
    e.load_weights(diee_amd.random_weights(0))
    bad = roots.copy(); bad["roll"] = 0
    with pytest.raises(diee_amd.DieeError) as ei:               # "die has not been rolled!" (backgammon_logic.rs:404)
        e.alpha_mcts_parallel(bad, cfg)
    assert ei.value.status == diee_amd.ERR_ARG
    with pytest.raises(diee_amd.DieeError) as ei:               # iterations = 0 would divide 0/0 in the reference
        e.self_play_parallel(2, diee_amd.MctsConfig.default(0))
    assert ei.value.status == diee_amd.ERR_ARG
    e.close()
    # a tree arena that is too small is reported, never silently truncated
    e2 = diee_amd.Engine(0); e2.load_weights(diee_amd.random_weights(0)); e2.set_option("nodes_per_expansion", 1)
    with pytest.raises(diee_amd.DieeError) as ei:
        e2.alpha_mcts_parallel(roots, diee_amd.MctsConfig.default(64))
    assert ei.value.status == diee_amd.ERR_CAPACITY
    e2.set_option("nodes_per_expansion", 128)
    r = e2.alpha_mcts_parallel(roots, diee_amd.MctsConfig.default(8))      # the engine stays usable afterwards
    assert np.allclose(r["probs"].sum(1), 1.0, atol=1e-5)
    e2.close()


def test_8192_games_on_one_gpu(oracle):
    """BASELINE configs[2] puts 8192 games on 8 GPUs; one MI355X's 288 GB holds all of them too"""
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    cfg = diee_amd.MctsConfig(iterations=3, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    out = e.self_play_parallel(8192, cfg, 1.25, 5, max_steps=2, fetch=False)
    st = out["stats"]
    assert st["move_steps"] == 2 and st["plies"] == 2 * 8192 and st["nn_evals"] == 2 * 4 * 8192
    assert st["illegal_decodes"] == 0
    e.close()


def test_starved_cluster_launch_falls_back_and_repeats_the_search(oracle):
    """a starved in-launch hand-over of the cluster tower (another process on the GPU) must not fail the call: the
    engine drops the cluster tower for the rest of the process and repeats that move-step's search on the per-layer
    kernels, which compute the same bits.  Forced here through DIEE_TEST_STARVE_AT in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
cfg = diee_amd.MctsConfig(iterations=6, c=2.0, round_limit=30, dir_alpha=0.3, dir_eps=0.25)
out = e.self_play_parallel(9, cfg, 1.25, 77, ref_quirks=True)
print("RESULT", hashlib.sha256(out["ps"].tobytes() + out["state"].tobytes() + out["outcome"].tobytes()).hexdigest(),
      out["stats"]["nn_evals"], out["stats"]["expansions"], out["stats"]["move_steps"])
""" % root
    res = {}
    for name, env in (("normal", {}), ("starved", {"DIEE_TEST_STARVE_AT": "3"})):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        res[name] = ([l for l in p.stdout.decode().splitlines() if l.startswith("RESULT")][0], p.stderr.decode())
    assert "falling back to the per-layer kernels" in res["starved"][1] and "falling back" not in res["normal"][1]
    assert res["starved"][0] == res["normal"][0]          # same records, same counters: the repeated search left no trace


def test_a_tail_meeting_that_times_out_on_the_device_is_repeated_launch_by_launch(oracle):
    """k_tail's own time-out path (round-5 review, weak #8): one game's workgroup never arrives at a meeting of the games' workgroups
    (option test_tail_skip), so every workgroup's bounded spin runs out with the games in mid-iteration and their LDS state half written
    back.  The starved bit is raised on the device, the state word keeps its 2 whatever slot 0 writes afterwards, tail_run takes the starved
    path, and the move-step's search is repeated launch by launch: the same records as the undisturbed run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
cfg = diee_amd.MctsConfig(iterations=8, c=2.0, round_limit=30, dir_alpha=0.3, dir_eps=0.25)
out = e.self_play_parallel(12, cfg, 1.25, 78, ref_quirks=True)
print("RESULT", hashlib.sha256(out["ps"].tobytes() + out["state"].tobytes() + out["outcome"].tobytes()).hexdigest(),
      out["stats"]["nn_evals"], out["stats"]["expansions"], out["stats"]["move_steps"])
""" % root
    res = {}
    for name, env in (("normal", {}), ("timed out", {"DIEE_TEST_TAIL_SKIP": "3"})):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        res[name] = ([l for l in p.stdout.decode().splitlines() if l.startswith("RESULT")][0], p.stderr.decode())
    assert "falling back to the per-layer kernels" in res["timed out"][1] and "falling back" not in res["normal"][1]
    assert res["timed out"][0] == res["normal"][0]


def test_options_travel_through_the_abi():
    """diee_set_option / diee_get_option (include/diee.h): what used to be environment switches.  Values round-trip, unknown keys and
    malformed values are DIEE_ERR_ARG and change nothing, the dispatch follows (bands of the development probe), `shared_gpu` takes
    every kernel that waits for a co-resident workgroup out of the dispatch -- and the environment is not consulted after diee_create"""
    import os
    import diee_amd
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    assert e.get_option("spec_eval") == "1" and e.get_option("tower_cl") == "default" and e.get_option("shared_gpu") == "0"
    e.set_options(compact=0, deliver_stage_rows=77, tower_cl="32:1,64:2")
    assert (e.get_option("compact"), e.get_option("deliver_stage_rows"), e.get_option("tower_cl")) == ("0", "77", "32:1,64:2")
    for key, value in (("no_such_option", "1"), ("compact", "yes"), ("compact", "-1"), ("tower_table", "5"), ("tower_cl", "32:1,"), ("tower_table", "0:8")):
        with pytest.raises(diee_amd.DieeError) as ei:
            e.set_option(key, value)
        assert ei.value.status == diee_amd.ERR_ARG, (key, value)
    assert e.get_option("compact") == "0" and e.get_option("tower_table") == "default"
    e.set_options(compact=1, tower_cl="default")
    full = e.dispatch_bands(1024)
    assert [k for _, _, k in full][:4] == ["k_tower_cl<1, 8>", "k_tower_cl<2, 8>", "k_tower16p<2, 6>", "k_tower16p<4, 6>"]
    os.environ["DIEE_SHARED_GPU"] = "1"                          # too late: the environment was read when the ctx was created
    try:
        assert e.dispatch_bands(1024) == full
    finally:
        del os.environ["DIEE_SHARED_GPU"]
    e.set_option("shared_gpu", 1)
    shared = [k for _, _, k in e.dispatch_bands(1024)]
    assert not any(k.startswith(("k_tower_cl", "k_tower16p")) for k in shared) and shared[-1] == "k_tower16<4, 4, 3, 0>"
    states = np.zeros(3, dtype=diee_amd.BG_STATE); states["pts"][:, 0] = 2; states["pts"][:, 23] = -2; states["roll"] = (3, 1); states["player"] = -1
    r = e.alpha_mcts_parallel(states, diee_amd.MctsConfig.default(8), 1, 0, np.arange(3, dtype=np.uint32), np.zeros(3, dtype=np.uint32))
    assert r["stats"]["tail_iterations"] == 0                   # the tail's looping kernel waits for co-resident workgroups too
    e.set_option("shared_gpu", 0)
    assert e.dispatch_bands(1024) == full
    e.close()
    t = diee_amd.Engine(0, diee_amd.GAME_TTT)
    with pytest.raises(diee_amd.DieeError) as ei:
        t.set_option("compact", 0)
    assert ei.value.status == diee_amd.ERR_UNSUPPORTED
    t.close()
