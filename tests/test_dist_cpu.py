"""world_size-2 gloo test of the N>1 path of bench.py: block-partitioned game ids, no data-path
collective, barrier + MAX(time) / SUM(counters) reduction.  Each rank plays its shard with the CPU
oracle (the stand-in for the GPU engine on this box); the union must equal the unsharded batch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import importlib
    import time
    import torch.distributed as dist
    ddist = importlib.import_module("die-e_amd.dist")
    from oracle import oracle as orc
    r, lr, w = ddist.rank_world()
    assert (r, w) == (rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    games = 3
    first = ddist.shard_first_game_id(r, games)
    cfg = orc.MctsCfg(iterations=6, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    dist.barrier()
    t0 = time.perf_counter()
    out = orc.self_play_parallel(1, games, cfg, 1.25, 11, orc.hash_eval_fn(), orc.game(1), ref_quirks=0, first_game_id=first)
    dist.barrier()
    dt = time.perf_counter() - t0
    tot = {"games": games, "expansions": out["stats"]["expansions"], "plies": int(out["plies"].sum())}
    tmax, red = ddist.reduce_stats(dist, dt, tot, ["games", "expansions", "plies"], "cpu")
    q.put((rank, dt, tmax, red, out["game"].tolist(), out["ps"].tobytes(), out["stats"]["expansions"]))
    dist.destroy_process_group()


def test_two_rank_sharded_self_play_gloo(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, tmax0, red0, games0, ps0, e0), (r1, dt1, tmax1, red1, games1, ps1, e1) = res
    assert tmax0 == tmax1 == max(dt0, dt1)                        # MAX over ranks
    assert red0 == red1 and red0["games"] == 6 and red0["expansions"] == e0 + e1
    assert set(games0) == {0, 1, 2} and set(games1) == {3, 4, 5}  # block partition of the ids
    cfg = oracle.MctsCfg(iterations=6, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    full = oracle.self_play_parallel(1, 6, cfg, 1.25, 11, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=0)
    a = np.frombuffer(ps0, dtype=np.float32).reshape(-1, 1352); b = np.frombuffer(ps1, dtype=np.float32).reshape(-1, 1352)
    for g in range(6):
        part, gl = (a, np.array(games0)) if g < 3 else (b, np.array(games1))
        assert part[gl == g].tobytes() == full["ps"][full["game"] == g].tobytes()


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import importlib
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    import diee_amd
    az = importlib.import_module("die-e_amd.alphazero")
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = orc.MctsCfg(iterations=4, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    r = orc.self_play_parallel(1, 2, cfg, 1.25, 3, orc.hash_eval_fn(), orc.game(1), first_game_id=2 * rank)
    # each rank trains on its own games -- and on a DIFFERENT number of fragments (rank 0: 8 = 2 batches of 4, rank 1: 14 =
    # 4 batches): ranks must take the same number of all-reduced steps (the minimum) or the longer one hangs in backward
    take = 8 if rank == 0 else 14
    assert len(r["outcome"]) >= 14
    mem = {k: r[k][:take] for k in ("outcome", "ps", "state")}
    a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 1, 4, 2), diee_amd.MctsConfig.default(4),
                     az.OptimizerParams(1e-4, 1e-3), blob=diee_amd.random_weights(0), train_device="cpu",
                     rank=rank, world=world, quiet=True)
    losses = a.train(mem)
    a.sync_engine()
    w = torch.from_numpy(a.blob.copy())
    # sync_engine broadcasts rank 0's blob (BatchNorm running statistics are per-rank buffers): whole blobs must agree
    import hashlib
    q.put((rank, losses, float(w[:256 * 6 * 9].double().sum()), hashlib.sha256(a.blob.tobytes()).hexdigest()))
    dist.destroy_process_group()


def test_two_rank_ddp_training_step_gloo():
    """config 5's training leg: DistributedDataParallel averages the gradients, so both ranks hold the same weights"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, a0, b0), (_, l1, a1, b1) = res
    assert np.isfinite(l0 + l1).all()
    assert len(l0) == len(l1) == 2     # min(ceil(8/4), ceil(14/4)) all-reduced steps on both ranks: nobody waits forever
    assert l0 != l1                    # different shards, different losses
    assert a0 == a1 and b0 == b1       # identical blobs (parameters AND BatchNorm statistics) after sync_engine


# ---- bench.py's own main() on two gloo ranks ---------------------------------------------------------------------------
class _StubEngine:
    """stand-in for diee_amd.Engine in the CPU test of bench.main(): plays the rank's shard with the CPU oracle (hash
    evaluator) and returns the engine's stats dictionary; the sampled tower timings stay zero (no network kernels ran)"""

    def __init__(self, device):
        self.device = device

    def load_weights(self, blob):
        assert len(blob) > 1000

    @staticmethod
    def _stats(out, n):
        import importlib
        pkg = importlib.import_module("die-e_amd")
        st = {name: 0 for name, _ in pkg.Stats._fields_}
        for k in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "illegal_decodes", "max_children"):
            st[k] = out["stats"][k]
        st.update(games=n, plies=int(out["plies"].sum()), move_steps=out["steps"], fragments=len(out["outcome"]), nn_rows=out["stats"]["nn_evals"])
        return st

    def self_play_parallel(self, n_games, cfg, temperature=1.25, seed=0, ref_quirks=True, first_game_id=0, max_steps=0, fetch=True, invariant_nn=False,
                           copy=True):
        from oracle import oracle as orc
        ocfg = orc.MctsCfg(iterations=cfg.iterations, c=cfg.c, round_limit=cfg.round_limit, dir_alpha=cfg.dir_alpha, dir_eps=cfg.dir_eps)
        out = orc.self_play_parallel(1, n_games, ocfg, temperature, seed, orc.hash_eval_fn(), orc.game(1), ref_quirks=1 if ref_quirks else 0,
                                     first_game_id=first_game_id, max_steps=max_steps)
        res = {"stats": self._stats(out, n_games if not max_steps else int((out["winners"] != 0).sum()))}
        if fetch:                      # (the records, as the engine hands them over; nothing to free on this side)
            res.update({k: out[k] for k in ("outcome", "ps", "state", "game")})
        return res

    def self_play_multi(self, batches, cfg, temperature=1.25, ref_quirks=True, max_steps=0, fetch=True, invariant_nn=False, copy=True):
        return [self.self_play_parallel(n, cfg, temperature, seed, ref_quirks, first, max_steps, fetch) for n, first, seed in batches]


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), DIEE_BENCH_BACKEND="gloo")
    import contextlib
    import io
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--games", "3", "--iterations", "5", "--pipeline", "2",
                    "--no-cpu-baseline"], engine_factory=_StubEngine)
    q.put((rank, buf.getvalue()))


def test_bench_main_on_two_gloo_ranks(oracle):
    """the N > 1 branches of bench.py -- init_process_group, the barriers around the timed region, MAX(time) / SUM(counters),
    the all_gather of per-rank fragment counts, rank 0's line -- executed by bench.main() itself at world size 2"""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 1000
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1].strip() == ""                                    # only rank 0 prints
    lines = [l for l in res[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1                                         # ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["metric"] == "self-play games/sec" and d["unit"] == "games/s" and d["vs_baseline"] is None
    assert d["stats"]["games"] == 2 * 2 * 3                        # ranks x steps x games per rank: the whole job
    assert abs(d["value"] - d["stats"]["games"] / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert len(d["fragments_per_rank"]) == 2 and sum(d["fragments_per_rank"]) == d["stats"]["fragments"]
    assert min(d["fragments_per_rank"]) > 0 and d["fragments_per_rank"][0] != d["fragments_per_rank"][1]    # different shards
    assert d["pipelined"]["batches"] == 2 and d["pipelined"]["games"] == 2 * 2 * 3
    assert d["roofline"] is None and "scale_note" in d             # the stand-in ran no network kernel: nothing sampled
    assert "cpu_baseline" not in d                                 # rank 0 at N = 1 only
    # the shards are the block partition of the unsharded batch: rank r's games are ids [3r, 3r + 3)
    cfg = oracle.MctsCfg(iterations=5, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    per_rank = []
    for r in range(2):
        tot = 0
        for i in range(2):
            o = oracle.self_play_parallel(1, 3, cfg, 1.25, 0xD1EE0001 + 0x9E37 * i, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=1, first_game_id=3 * r)
            tot += len(o["outcome"])
        per_rank.append(tot)
    assert per_rank == d["fragments_per_rank"]


def _bench_worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), DIEE_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    import contextlib
    import io
    import torch
    torch.set_num_threads(1)
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--games", "2", "--iterations", "4", "--pipeline", "2",
                    "--hbm-only-steps", "1", "--no-cpu-baseline"], engine_factory=_StubEngine)
    q.put((rank, buf.getvalue()))


def test_bench_main_on_eight_gloo_ranks(oracle):
    """the driver's 8-GPU launch (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`) rehearsed once before hardware:
    bench.main() on EIGHT gloo ranks with the stand-in engine -- the 8-way barriers, MAX(time) / SUM(counters), the all_gather
    of 8 fragment counts, rank 0's single line with `fragments_per_rank` of length 8, the HBM-only and pipelined legs"""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + os.getpid() % 1000
    world = 8
    procs = [ctx.Process(target=_bench_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(res[r].strip() == "" for r in range(1, world))      # only rank 0 prints
    lines = [l for l in res[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["stats"]["games"] == 8 * 2
    assert len(d["fragments_per_rank"]) == 8 and sum(d["fragments_per_rank"]) == d["stats"]["fragments"] and min(d["fragments_per_rank"]) > 0
    assert d["pipelined"]["games"] == 8 * 2 * 2 and d["value_hbm_only"] > 0 and "output_delivery_ms" in d
    assert d["config"]["parallelism"].startswith("dp8")
    # shards = block partition of the global ids: rank r plays games [2r, 2r + 2)
    cfg = oracle.MctsCfg(iterations=4, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    want = [len(oracle.self_play_parallel(1, 2, cfg, 1.25, 0xD1EE0001, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=1, first_game_id=2 * r)["outcome"])
            for r in range(world)]
    assert want == d["fragments_per_rank"]


def test_plain_bench_command_starts_its_own_ranks(oracle):
    """the driver's command shape for N > 1 is the plain `python bench.py --gpus N ...`: bench.py itself must start the N ranks
    (a child torch.distributed.run on 127.0.0.1 and a free port, before anything could touch a GPU), pass rank 0's single line
    through and return the child's exit code.  Here: N = 2 over gloo with the stand-in engine named through DIEE_BENCH_ENGINE."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DIEE_BENCH_BACKEND="gloo", DIEE_BENCH_ENGINE="tests.test_dist_cpu:_StubEngine", OMP_NUM_THREADS="1",
               PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    args = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--games", "3", "--iterations", "5", "--pipeline", "0", "--hbm-only-steps", "0",
            "--no-cpu-baseline", "--launch-timeout", "600"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stats"]["games"] == 2 * 3 and len(d["fragments_per_rank"]) == 2 and d["config"]["parallelism"].startswith("dp2")
    assert d["engine"].startswith("STAND-IN")
    # a rank that fails makes the plain command fail too (the launcher returns its child's code): an engine that cannot be imported
    env["DIEE_BENCH_ENGINE"] = "tests.test_dist_cpu:_NoSuchEngine"
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert q.returncode != 0 and not [l for l in q.stdout.decode().splitlines() if l.startswith("{")]


def test_rank_refuses_a_gpus_flag_that_is_not_its_world_size():
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", DIEE_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE is 1" in p.stderr
