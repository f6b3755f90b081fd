"""GPU tests of the host-side callers: the arena on the engine equals the arena on
the oracle backends bit for bit, and one learn iteration (self-play on the engine -> PyTorch-ROCm training ->
weights folded back into the engine) leaves the engine consistent with the trained network."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

az = importlib.import_module("die-e_amd.alphazero")
versus = importlib.import_module("die-e_amd.versus")


def test_arena_engine_equals_oracle(oracle):
    import diee_amd
    from oracle.arena_backend import OracleRules, OracleSearch
    e1 = diee_amd.Engine(0); e1.load_weights(diee_amd.random_weights(0))
    e2 = diee_amd.Engine(0); e2.load_weights(diee_amd.random_weights(1))
    cfg = diee_amd.MctsConfig(iterations=6, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    P = versus.Player
    res = versus.play(P(versus.Agent.MODEL, e1), P(versus.Agent.MODEL, e2), cfg, 1.25, seed=21, num_games=8, round_limit=60)

    def ev(e):
        return oracle.make_eval(lambda st: e.forward_t(st.view(oracle.BG_STATE).reshape(-1)), 1352)
    f1, f2 = ev(e1), ev(e2)
    ref = versus.play(P(versus.Agent.MODEL), P(versus.Agent.MODEL), cfg, 1.25, seed=21, num_games=8, round_limit=60,
                      rules=OracleRules(), search1=OracleSearch(f1), search2=OracleSearch(f2))
    assert res.final_states.tobytes() == ref.final_states.tobytes()
    assert (res.wins_p1, res.wins_p2, res.draws, res.rounds) == (ref.wins_p1, ref.wins_p2, ref.draws, ref.rounds)
    assert sorted((g.initial_state["id"], g.winner) for g in res.games) == sorted((g.initial_state["id"], g.winner) for g in ref.games)
    # Model vs Random on the engine
    r2 = versus.play(P(versus.Agent.MODEL, e1), P(versus.Agent.RANDOM), cfg, 1.25, seed=4, num_games=6, round_limit=400)
    assert r2.wins_p1 + r2.wins_p2 == 6
    e1.close(); e2.close()


def test_learn_iteration_smoke(oracle, tmp_path):
    import torch
    import diee_amd
    from oracle import nn_ref
    eng = diee_amd.Engine(0)
    conf = az.AlphaZeroConfig(temperature=1.25, learn_iterations=1, self_play_iterations=2, num_epochs=1,
                              training_batch_size=64, num_self_play_batches=8)
    a = az.AlphaZero(eng, conf, diee_amd.MctsConfig(iterations=6, c=2.0, round_limit=80, dir_alpha=0.3, dir_eps=0.25),
                     az.OptimizerParams(1e-4, 1e-3), blob=diee_amd.random_weights(0), root=str(tmp_path), quiet=True)
    assert a.device == "cuda"
    rep = a.learn_parallel(arena=True, arena_games=4)
    assert len(rep) == 1 and rep[0]["fragments"] > 0 and np.isfinite(rep[0]["loss_last"])
    assert rep[0]["arena"] == "saved-as-best"                                    # alpha_versus.rs:19-27
    run = next((tmp_path / "data" / "backgammon").iterdir())
    sp0 = az.AlphaZero.load_training_data(str(run / "lrn-0" / "sp-0")); sp1 = az.AlphaZero.load_training_data(str(run / "lrn-0" / "sp-1"))
    assert len(sp1["outcome"]) > len(sp0["outcome"])                             # cumulative memory per sp dir (Q20)
    assert (tmp_path / "models" / "backgammon" / "model_0.npy").exists() and (tmp_path / "models" / "backgammon" / "best_model.npy").exists()
    # the engine now runs the trained weights: compare with the fp32 restatement on the same blob
    states = oracle.random_walk_states(8, 2)[:24]
    pol, val = eng.forward_t(states)
    rp, rv, _ = nn_ref.forward_t(nn_ref.parse(a.blob), oracle.planes_batch(states))
    assert np.abs(pol - rp).max() <= 2e-3 and np.abs(val - rv).max() <= 1e-2
    assert (a.blob != diee_amd.random_weights(0)).any()
    # second learn iteration plays the arena against the saved best model
    verdict = a.play_vs_best_model(n_games=4)
    assert verdict in ("new model was better!", "current best model is still better!",
                       "new model vs current best was inconclusive, keeping current best!")
    eng.close()


def _config5_loop(tmp_path, games, label):
    """the learn loop at BASELINE configs[4]'s parameter values with `games` games per self-play batch: runs every phase of both
    learn iterations, prints the wall clock per phase, returns the report"""
    import time
    import diee_amd
    eng = diee_amd.Engine(0)
    conf = az.AlphaZeroConfig(temperature=1.25, learn_iterations=2, self_play_iterations=4, num_epochs=4,
                              training_batch_size=256, num_self_play_batches=games)
    a = az.AlphaZero(eng, conf, diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3), blob=diee_amd.random_weights(0),
                     root=str(tmp_path), quiet=True)
    assert a.train_backend == "fp32"                                             # the reference's arithmetic is the default
    t = time.time()
    rep = a.learn_parallel(arena=True, arena_games=400)
    total = time.time() - t
    assert len(rep) == 2
    for r in rep:
        means = r["epoch_loss_means"]
        # The signal: the MEAN loss of an epoch.  (The last step of an epoch is the reference's partial batch, alphazero.rs:205-206:
        # n mod 256 samples -- 4 in the full-size run, whose single-step loss says nothing.)  Training descends over the four epochs
        # of a learn iteration, and every epoch mean sits below the iteration's very first step (a random-init net: ~12).
        assert len(means) == 4 and all(np.isfinite(m) for m in means)
        assert means[-1] < means[0], means
        assert r["train_steps"] == 4 * -(-r["fragments"] // 256)
        print(f"[config5, {label}] learn iteration {r['learn_iteration']}: {r['fragments']} fragments, self-play {r['self_play_s']:.1f} s, "
              f"train {r['train_s']:.1f} s ({r['train_steps']} steps), arena {r['arena_s']:.1f} s, epoch mean loss "
              + " -> ".join(f"{m:.3f}" for m in means) + f", arena: {r['arena']}")
    assert rep[0]["epoch_loss_means"][-1] < rep[0]["loss_first"]
    print(f"[config5, {label}] whole loop {total:.1f} s")
    assert rep[0]["arena"] == "saved-as-best"
    assert rep[1]["arena"] in ("new model was better!", "current best model is still better!",
                               "new model vs current best was inconclusive, keeping current best!")
    run = next((tmp_path / "data" / "backgammon").iterdir())
    for li in range(2):
        dirs = [run / f"lrn-{li}" / f"sp-{j}" for j in range(4)]
        sizes = [len(np.load(d / "outcomes.npy", mmap_mode="r")) for d in dirs]
        assert sizes == sorted(sizes) and sizes[0] > 0 and sizes[3] == rep[li]["fragments"]      # cumulative memory per sp dir (Q20)
        assert all(np.load(d / "ps.npy", mmap_mode="r").shape == (n, 1352) and np.load(d / "states.npy", mmap_mode="r").shape == (n, 6, 4, 6)
                   for d, n in zip(dirs, sizes))                                                  # alphazero.rs:149-200
        assert (tmp_path / "models" / "backgammon" / f"model_{li}.npy").exists()
    eng.close()
    return rep, total


def test_learn_loop_config5_full_size(tmp_path):
    """BASELINE configs[4] AT ITS OWN SIZE on one GPU: learn_iterations=2, self_play_iterations=4, num_epochs=4,
    training_batch_size=256, num_self_play_batches=1024, iterations=100, temperature 1.25, arena of 400 games, the default fp32
    training step (alpha_parallel.rs:17-99, alphazero.rs:202-261).  ~5 minutes: 2 x (4 batches of 1024 games side by side,
    ~430 k / ~640 k cumulative fragments written per sp-j dir, 4 epochs of ~1 700 / ~2 500 steps, fold-back, arena)."""
    rep, total = _config5_loop(tmp_path, 1024, "1024 games per batch = configs[4]")
    assert all(r["fragments"] > 4 * 1024 * 60 for r in rep)                      # ~105 records per game
    # the second learn iteration trains the first one's network on fresh games: it starts far below a random-init network's loss
    assert rep[1]["epoch_loss_means"][0] < rep[0]["epoch_loss_means"][0]
    import shutil
    shutil.rmtree(tmp_path / "data", ignore_errors=True)                         # ~16 GB of cumulative sp-j dirs


def test_cli_end_to_end_in_a_child_process(tmp_path):
    """F4: `diee.py -c tiny.toml -g backgammon learn | train | play` driven like die-e's binary (main.rs:15-216), a fresh
    child process per command; the model handed to `learn` is a libtorch-style .ot archive (F3)"""
    import os
    import subprocess
    import sys
    import diee_amd
    ot = importlib.import_module("die-e_amd.ot")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tmp_path / "tiny.toml"
    cfg.write_text("temperature = 1.25\nlearn_iterations = 1\nnum_epochs = 1\ntraining_batch_size = 16\nself_play_iterations = 2\n"
                   "num_self_play_batches = 6\niterations = 4\nexploration_const = 2.0\nsimulate_round_limit = 30\n"
                   "dirichlet_alpha = 0.3\ndirichlet_epsilon = 0.25\nwd = 0.0001\nlr = 0.001\n")
    model = tmp_path / "start.ot"
    ot.save_model_ot(diee_amd.random_weights(2), str(model))

    def run(*args):
        p = subprocess.run([sys.executable, os.path.join(root, "diee.py"), "-c", str(cfg), "-g", "backgammon", *args],
                           cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        return p.returncode, p.stdout.decode()
    rc, out = run("learn", "-m", str(model))
    assert rc == 0, out
    assert "Staring up run with run_id" in out and "Iteration 0 saved successfully" in out
    mdir = tmp_path / "models" / "backgammon"
    assert (mdir / "model_0.npy").exists() and (mdir / "best_model.npy").exists()
    run_dir = next((tmp_path / "data" / "backgammon").iterdir())
    assert sorted(p.name for p in (run_dir / "lrn-0").iterdir()) == ["sp-0", "sp-1"]
    # train on everything that run wrote (main.rs:172-207), saving a .ot archive
    rc, out = run("train", "-r", run_dir.name[len("run-"):], "-o", str(tmp_path / "trained.ot"))
    assert rc == 0 and "Trained model saved successfully" in out, out
    assert ot.load_model_ot(str(tmp_path / "trained.ot")).size == diee_amd.weights_count()
    # play: Model (the .ot archive) against Random, 400 games of the arena are too many here -> the agents' surface only
    games = tmp_path / "games"; games.mkdir()
    rc, out = run("play", "-a", "Random", "--agent-two", "Random", "-o", str(games))
    assert rc == 0 and "Saving games" in out, out
    assert len(list(games.iterdir())) == 400                                      # versus.rs:160: 400 games
    rc, out = run("replay", "-g", str(next(games.iterdir())))
    assert rc == 0 and "Player 1: Random" in out


def test_cpp_host_drives_the_engine_through_the_c_abi(tmp_path):
    """examples/self_play.cpp (g++, include/diee.hpp over include/diee.h): legal plays, one search, a self-play batch and
    two batches side by side from a compiled host; batch 0 of the pipelined call equals the single call"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "self_play")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "self_play.cpp"),
                           "-L", os.path.join(root, "die-e_amd"), "-ldiee", "-Wl,-rpath," + os.path.join(root, "die-e_amd"), "-o", exe])
    p = subprocess.run([exe, "6", "6", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert "15 legal plays, first code 785" in out                     # the 15 opening plays, code of the first
    assert "self_play_parallel: 6 games" in out and "equals the single call: yes" in out
