#!/usr/bin/env python3
"""bench.py -- self-play throughput of the MI355X-native die-e engine (BASELINE.json metric).

One "step" = one complete self_play_parallel batch (alpha_parallel.rs:101-231): 1024 backgammon
games per GPU played to completion with iterations=100 MCTS and a random-init 19-block ResNet
(BASELINE.json configs[1]).  `value` = games retired by all ranks / wall time of the K timed
steps, inputs (weights, game states) resident in HBM when the timed region starts and the call's
outputs -- all_memories, alpha_parallel.rs:215-230: relabelled, ordered MemoryFragments -- DELIVERED
in host memory when it ends (`value_hbm_only`: the same batches with the records left in HBM).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3}] [--games G] [--iterations I]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` typed alone (N > 1, no RANK in the environment) starts its own N ranks -- a child torch.distributed.run on
127.0.0.1 and a free port, before anything that could touch a GPU is imported -- and passes rank 0's line through (launch_ranks).

Multi-GPU: games are independent, so ranks are independent data-parallel workers (weak scaling,
no data-path collective); torch.distributed (RCCL) only provides the barrier and the max/sum of
the per-rank timings and counters.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FLOPS_PER_EVAL = 1_082_450_064
PROFILE_SET = "r06z"          # ONE prefix under profiles/ for every figure this line reads from committed counter passes (scripts/profile_bench.sh + scripts/pmc_mfma.sh TAG)
MFMA_BUSY_FILE, MFMA_BUSY_FILE_32 = f"{PROFILE_SET}_mfma_busy.json", f"{PROFILE_SET}_mfma_busy_32boards.json"     # (the 32-board pass with DIEE_SPEC_EVAL=0)
TRAFFIC_FILE, TRAFFIC_FILE_32 = f"{PROFILE_SET}_pmc_traffic.json", f"{PROFILE_SET}_pmc_traffic_32boards.json"


def host_cores():
    """CPUs this process may really use: the affinity mask and the cgroup quota, not the machine's core count (a GPU box
    gives one GPU's share of the host, e.g. 16 of 128 threads; an OpenMP pool sized to the machine then spins 8-fold
    oversubscribed and a conv takes 100x longer)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    if "DIEE_CPU_THREADS" in os.environ:
        n = max(1, int(os.environ["DIEE_CPU_THREADS"]))
    return n


# ---- B-omp workers: one PROCESS per host core (forked before anything starts a thread pool), each with its own
# single-threaded fp32 PyTorch ResNet and its own oracle trees: no GIL between the searches (round 2 ran them as threads of
# one interpreter and measured the interpreter lock: 222 expansions/s on 16 threads, slower than the serial B-ref)
_W = {}


def _omp_init(seed):
    import numpy as np
    import torch
    torch.set_num_threads(1)
    import diee_amd
    from oracle import oracle as orc
    from oracle import nn_ref
    orc.build()
    net = nn_ref.parse(diee_amd.random_weights(0))

    def fn(states_u8):
        st = states_u8.view(orc.BG_STATE).reshape(-1)
        pol, val, _ = nn_ref.forward_t(net, orc.planes_batch(st))
        return pol, val
    _W.update(orc=orc, fn=fn, np=np, seed=seed)


def _omp_calibrate(root_bytes):
    np = _W["np"]
    st = np.frombuffer(root_bytes, dtype=np.uint8).reshape(-1, 32)
    _W["fn"](st)
    t = time.time(); _W["fn"](st); _W["fn"](st)
    return (time.time() - t) / 2


def _omp_search(job):
    root_bytes, first_id, iters = job
    orc, np = _W["orc"], _W["np"]
    states = np.frombuffer(root_bytes, dtype=orc.BG_STATE)
    out = []
    for i in range(len(states)):                       # one root after the other: batch-1 evaluations, as rayon over games would
        cfg = orc.MctsCfg(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
        ev = orc.make_eval(_W["fn"], 1352)
        _, _, st, _ = orc.alpha_mcts_parallel(1, states[i:i + 1], cfg, ev, None, _W["seed"], 0, np.arange(first_id + i, first_id + i + 1, dtype=np.uint32),
                                              np.zeros(1, dtype=np.uint32), 1)
        out.append(st.as_dict())
    return out


def cpu_baseline(iterations, seed, exp_per_game, n_roots=64, budget_s=12.0, games=16, games_budget_s=75.0, big_roots=1024, big_budget_s=30.0, plies_per_game=0.0):
    """The reference's CPU path restated (oracle = C restatement of its serial tree / game loops, PyTorch fp32 CPU
    ResNet = what tch/libtorch gives it on a CPU-only host), timed on a bounded sample: one move-step of search on
    `n_roots` positions drawn from random self-play walks (opening, middle game and bear-off alike); the number of MCTS
    iterations of the sample is cut so that each variant takes about `budget_s` seconds on this host.

      B-ref  as die-e runs self-play today: ONE batched search over all roots, tree loops on one thread
             (alpha_mcts.rs:153-168,192-200), the network batch on every core (libtorch intra-op pool);
      B-ref-1024  the same at the metric's own batch: one search over `big_roots` = num_self_play_batches = 1024 roots (the CPU network
             batches 16 x better than at 64 roots), iterations cut to fit ~`big_budget_s` seconds;
      B-omp  the generous variant BASELINE.md promises ("rayon over games", versus.rs:304,308, pool sized at main.rs:107-110):
             one search per root, one PROCESS per host core (the reference's rayon threads share no interpreter lock; Python
             threads would), batch-1 single-threaded network evaluations -- no cross-game batching, no serial section.
      B-games  `games` (16) games of the real self-play loop from the opening to a ply cap that fits `games_budget_s` (B-ref machinery; network
             batch = the live games): MEASURED plies/s -> `games_played_value`, beside the extrapolation.
    Returns the fastest of B-ref / B-ref-1024 / B-omp as `value`: games/s EXTRAPOLATED from that sample's expansions/s with the GPU run's
    expansions per game (`kind` says so)."""
    import multiprocessing as mp
    cores = host_cores()
    workers = max(1, min(cores, n_roots))
    # fork the B-omp workers FIRST: this process has not imported torch yet, so no OpenMP / intra-op pool exists to be
    # broken by the fork; every worker builds its own network
    pool = mp.get_context("fork").Pool(workers, initializer=_omp_init, initargs=(seed,))
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    walk = orc.random_walk_states(seed & 0xFFFF, 40)
    roots = walk[np.linspace(3, len(walk) - 1, n_roots).astype(int)]
    out = {"unit": "games/s", "kind": "port, extrapolated from expansions/s", "cores": cores, "host_threads_visible": os.cpu_count(), "variants": {}}
    # ---- B-omp ----
    try:
        t_eval = max(pool.map(_omp_calibrate, [roots[i:i + 1].tobytes() for i in range(workers)]))     # per-evaluation time with every worker busy
        shares = [roots[i::workers] for i in range(workers)]
        per_worker = max(len(sh) for sh in shares)
        it_omp = int(max(2, min(iterations, budget_s / (t_eval * per_worker) - 1)))
        t = time.time()
        res = pool.map(_omp_search, [(sh.tobytes(), 1000 * i, it_omp) for i, sh in enumerate(shares)])
        dt = time.time() - t
        sts = [s for r in res for s in r]
        exps = sum(s["expansions"] for s in sts)
        out["variants"]["B-omp"] = {
            "expansions_per_s": exps / dt, "seconds": dt, "threads": workers, "iterations_of_the_sample": it_omp,
            "batch1_eval_ms": t_eval * 1e3,
            "mean_children": sum(s["children"] for s in sts) / max(exps, 1),
            "mean_leaf_depth": sum(s["depth_sum"] for s in sts) / max(sum(s["selections"] for s in sts), 1),
            "what": f"{n_roots} independent searches over {workers} worker PROCESSES (one per host core, forked before any thread pool), "
                    "oracle tree in C + batch-1 single-threaded fp32 PyTorch evaluations in each"}
    finally:
        pool.close(); pool.join()
    # ---- B-ref ----
    import torch
    import diee_amd
    from oracle import nn_ref
    net = nn_ref.parse(diee_amd.random_weights(0))

    def fn(states_u8):
        st = states_u8.view(orc.BG_STATE).reshape(-1)
        pol, val, _ = nn_ref.forward_t(net, orc.planes_batch(st))
        return pol, val

    def search(states, first_id, iters):
        cfg = orc.MctsCfg(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
        ev = orc.make_eval(fn, 1352)
        n = len(states)
        _, _, st, _ = orc.alpha_mcts_parallel(1, states, cfg, ev, None, seed, 0, np.arange(first_id, first_id + n, dtype=np.uint32),
                                              np.zeros(n, dtype=np.uint32), 1)
        return st.as_dict()

    def calibrate(states):
        fn(states.view(np.uint8).reshape(-1, 32))                           # warm-up (primitive creation)
        t = time.time(); fn(states.view(np.uint8).reshape(-1, 32)); return max(time.time() - t, 1e-4)

    torch.set_num_threads(cores)
    it_ref = int(max(2, min(iterations, budget_s / calibrate(roots) - 1)))
    t = time.time(); st = search(roots, 0, it_ref); dt = time.time() - t
    out["variants"]["B-ref"] = {
        "expansions_per_s": st["expansions"] / dt, "seconds": dt, "threads": cores, "iterations_of_the_sample": it_ref,
        "mean_children": st["children"] / max(st["expansions"], 1), "mean_leaf_depth": st["depth_sum"] / max(st["selections"], 1),
        "what": f"one batched search over {n_roots} roots (serial C tree loops + fp32 PyTorch CPU ResNet on {cores} intra-op threads)"}
    # ---- B-ref at the metric's own batch size (BASELINE configs[1]: num_self_play_batches = 1024): one move-step's search over 1024 roots ----
    if big_roots and big_roots > n_roots:
        roots_big = walk[np.linspace(3, len(walk) - 1, big_roots).astype(int)] if len(walk) >= big_roots else np.resize(walk[3:], big_roots)
        t_big = calibrate(roots_big)
        it_big = int(max(1, min(iterations, big_budget_s / t_big - 1)))
        t = time.time(); st = search(roots_big, 0, it_big); dt = time.time() - t
        out["variants"][f"B-ref-{big_roots}"] = {
            "expansions_per_s": st["expansions"] / dt, "seconds": dt, "threads": cores, "iterations_of_the_sample": it_big,
            "eval_s_per_batch": t_big,
            "mean_children": st["children"] / max(st["expansions"], 1), "mean_leaf_depth": st["depth_sum"] / max(st["selections"], 1),
            "what": f"one batched search over {big_roots} roots = the metric's num_self_play_batches (serial C tree loops + fp32 PyTorch CPU ResNet on "
                    f"{cores} intra-op threads, {t_big:.2f} s per evaluation of the batch)"}
    # ---- B-games: a MEASURED run of the metric's own loop, not a search sample: `games` games of self_play_parallel played by the same B-ref
    # machinery (serial C driver + tree, the network batch = the live games on every core) at the full iteration count, from the opening
    # to a ply cap that fits the budget (a whole game is ~110 plies x 101 evaluations: hours at this rate); games/s = plies played per
    # second / the plies of a game (the GPU run's mean)
    if games > 0:
        cfg = orc.MctsCfg(iterations=iterations, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
        # the first move-step alone tells what one costs (a single timed evaluation proved unreliable: 100x off on a busy host); then the
        # measured run plays as many move-steps as fit the budget, from the opening again
        t = time.time()
        orc.self_play_parallel(1, games, cfg, 1.25, seed, orc.make_eval(fn, 1352), None, ref_quirks=1, max_steps=1)
        t_step = max(time.time() - t, 1e-3)
        t_g = t_step / (iterations + 1)
        cap = int(max(1, min(400, games_budget_s / t_step)))
        t = time.time()
        sp = orc.self_play_parallel(1, games, cfg, 1.25, seed, orc.make_eval(fn, 1352), None, ref_quirks=1, max_steps=cap)
        dt = time.time() - t
        plies = int(sp["plies"].sum())
        ppg = plies_per_game or 110.0
        out["variants"]["B-games"] = {
            "games_per_s": plies / dt / ppg, "plies_per_s": plies / dt, "games": games, "move_steps": int(sp["steps"]), "plies_played": plies,
            "plies_per_game_assumed": ppg, "seconds": dt, "threads": cores, "iterations": iterations,
            "expansions_per_s": sp["stats"]["expansions"] / dt, "eval_s_per_batch": t_g,
            "what": f"{games} games of self_play_parallel, iterations = {iterations}, played from the opening for {int(sp['steps'])} move-steps "
                    f"({plies} plies; oracle driver + tree in C on one thread, fp32 PyTorch CPU ResNet on {cores} intra-op threads, network batch = "
                    f"the {games} live games): MEASURED plies/s, / {ppg:.1f} plies per game (the GPU run's mean) = games/s"}
        out["games_played_value"] = plies / dt / ppg
    best = max((k for k in out["variants"] if k != "B-games"), key=lambda k: out["variants"][k]["expansions_per_s"])
    eps = out["variants"][best]["expansions_per_s"]
    out["expansions_per_s"] = eps
    out["value"] = eps / exp_per_game if exp_per_game else None
    out["cores"] = out["variants"][best]["threads"]
    n_best = big_roots if best == f"B-ref-{big_roots}" else n_roots
    out["sample"] = (f"{best} (the fastest of B-ref at {n_roots} roots, B-ref at {big_roots} roots = the metric's batch, and B-omp, all in `variants`): oracle (C restatement of the reference's tree / game "
                     f"loops) + PyTorch fp32 CPU ResNet, one move-step of search (iterations cut to "
                     f"{out['variants'][best]['iterations_of_the_sample']} of {iterations} to fit the time budget) on {n_best} positions "
                     f"drawn evenly from random self-play walks (opening to bear-off); games/s extrapolated with the GPU run's "
                     f"{exp_per_game:.0f} expansions per game")
    return out


def cpu_baseline_guarded(iterations, seed, exp_per_game, timeout_s=480, games=16, plies_per_game=0.0):
    """cpu_baseline in a child process (its own thread-pool settings; killed by PID after timeout_s): the baseline is a
    report and must never hang the bench line"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--iterations", str(iterations),
           "--seed", str(seed), "--exp-per-game", repr(float(exp_per_game)), "--cpu-games", str(games), "--plies-per-game", repr(float(plies_per_game))]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        why = f"rc {p.returncode}: {p.stderr.decode()[-300:]}"
    except subprocess.TimeoutExpired:
        why = f"no result within {timeout_s} s"
    return {"value": None, "unit": "games/s", "cores": host_cores(), "kind": "port, extrapolated from expansions/s", "sample": f"failed: {why}"}


def learn_loop_leg(eng, games, iterations, timeout_s=3000):
    """BASELINE configs[4] (learn_iterations=2, self_play_iterations=4, num_epochs=4, training_batch_size=256; SURVEY 8(d) item 5:
    wall-clock per phase) in a CHILD process: the learn loop trains with PyTorch-ROCm, whose bundled HIP runtime has to
    initialise before libdiee.so's -- this process loaded the engine first.  The parent's engine is idle meanwhile."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "learn_config5.py"), str(games), "2", str(iterations)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"failed": f"rc {p.returncode}: {p.stderr.decode()[-400:]}"}
    except subprocess.TimeoutExpired:
        return {"failed": f"no result within {timeout_s} s"}


PRESETS = {   # BASELINE.json configs[] that are self-play runs (configs[0] = tic-tac-toe plumbing, configs[4] = the learn loop: --learn-loop)
    1: dict(games=1024, iterations=100, name="configs[1]: backgammon, num_self_play_batches=1024, iterations=100, 1 GPU (the metric's own configuration)"),
    2: dict(games=1024, iterations=400, name="configs[2]: backgammon, num_self_play_batches=8192 over 8 GPUs = 1024 per GPU, iterations=400"),
    3: dict(games=1024, iterations=1600, name="configs[3]: backgammon deep tree, iterations=1600, simulate_round_limit=400"),
}
BAND_NAMES = ["1-16", "17-32", "33-64", "65-128", "129-256", "257-512", "513-928", "929-1024", ">1024"]     # DIEE_BANDS (include/diee.h)


def pin_to_gpu_numa(torch, dev):
    """N > 1: keep this rank's host thread (it draws the Dirichlet samples and enqueues ~300 launches per move-step) on the
    NUMA node of its GPU.  Best effort: returns a description for the line, never fails the run."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return f"gpu {bdf}: no NUMA node reported"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return f"gpu {bdf} on NUMA node {node}: none of its CPUs is in this process's affinity mask"
        os.sched_setaffinity(0, allowed)
        return f"gpu {bdf} on NUMA node {node}: pinned to {len(allowed)} of its CPUs"
    except Exception as e:      # (containers hide sysfs entries; a box without NUMA info runs unpinned)
        return f"not pinned ({type(e).__name__}: {e})"


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_group(cmd, env, timeout_s, hold_json):
    """one child process group (start_new_session): stdout passed through line by line -- the last JSON line held back when
    `hold_json` --, killed as a group (by its own pgid, never by pattern) when `timeout_s` expires or this process is told to stop"""
    import signal
    import subprocess
    import threading
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, start_new_session=True)
    held = []

    def pump():
        for raw in p.stdout:
            line = raw.decode(errors="replace")
            if hold_json and line.startswith("{"):
                held.append(line)
            else:
                sys.stdout.write(line); sys.stdout.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()

    def stop(signum, _frame):
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
    old = {sg: signal.signal(sg, stop) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        try:
            rc = p.wait(timeout=timeout_s or None)
        except subprocess.TimeoutExpired:
            sys.stderr.write(f"[bench] the ranks did not finish within {timeout_s} s: stopping them\n")
            for sg in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(p.pid, sg)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=20)
                    break
                except subprocess.TimeoutExpired:
                    continue
            rc = 124
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    th.join(timeout=30)
    return rc, held


def launch_ranks(args, argv):
    """`python bench.py --gpus N` typed alone (N > 1, no RANK in the environment -- the driver's command shape): THIS process
    becomes the launcher.  It has imported nothing that can initialise the GPU and never will; it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same argv>`
    as a CHILD process (never an exec), passes rank 0's JSON line through and exits with the child's return code -- non-zero
    when any rank failed (torch.distributed.run stops the others) or the ranks outlived --launch-timeout.
    --learn-loop (BASELINE configs[4]) at N > 1: a second group of N ranks runs the learn loop behind the self-play legs
    (scripts/learn_config5.py: DistributedDataParallel over RCCL) and its report joins the line as `learn_loop`."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between the ranks' processes needs it on this driver
    child_argv = [a for a in argv if a != "--learn-loop"]
    run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1"]
    rc, held = _run_group(run + ["--master-port", str(_free_port()), os.path.abspath(__file__)] + child_argv, env, args.launch_timeout, args.learn_loop)
    if args.learn_loop and held:
        line = json.loads(held[-1])
        if rc == 0:
            games = args.learn_games or args.games
            rc2, rep = _run_group(run + ["--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "learn_config5.py"),
                                         str(games), "2", str(args.iterations)], env, args.launch_timeout, True)
            line["learn_loop"] = json.loads(rep[-1]) if rc2 == 0 and rep else {"failed": f"rc {rc2}"}
            rc = rc or rc2
        print(json.dumps(line)); sys.stdout.flush()
    return rc


def main(argv=None, engine_factory=None):
    """argv / engine_factory: tests/test_dist_cpu.py drives this very function on gloo ranks with a stand-in engine
    (DIEE_BENCH_BACKEND=gloo: CPU tensors for the reductions, no torch.cuda call); the driver and users run it as a script."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="the job's size = WORLD_SIZE of the launcher (checked); N > 1: python -m torch.distributed.run ...")
    ap.add_argument("--share-lock", default=None, help="tests: ranks that share one GPU keep the full kernel set and take turns through this lock file "
                    "(default for shared GPUs: diee_set_option shared_gpu = 1)")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--config", type=int, choices=sorted(PRESETS), default=1, help="BASELINE.json configs[] preset: 1 = 1024 games x iterations 100 "
                    "(the metric's configuration, default), 2 = 1024 per GPU x 400, 3 = 1024 x 1600 (deep tree)")
    ap.add_argument("--games", type=int, default=None, help="num_self_play_batches per GPU (overrides the preset)")
    ap.add_argument("--iterations", type=int, default=None, help="MCTS iterations (overrides the preset)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xD1EE0001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-games", type=int, default=16, help="games the CPU baseline also plays through the real self-play loop to a ply cap (measured plies/s; 0 = extrapolation only)")
    ap.add_argument("--plies-per-game", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--max-steps", type=int, default=0, help="profiling aid: stop each batch after this many move-steps (0 = play to completion)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--exp-per-game", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--pipeline", type=int, default=4, help="also time K batches played side by side through diee_self_play_multi "
                    "(self_play_iterations of the learn loop; 0 = skip); reported as value_pipelined, never as value")
    ap.add_argument("--hbm-only-steps", type=int, default=2, help="also repeat this many of the timed batches (same seeds) with the records left in HBM: "
                    "value_hbm_only / output_delivery_ms (0 = skip)")
    ap.add_argument("--learn-loop", action="store_true", help="also run BASELINE configs[4] (learn_iterations=2, self_play_iterations=4, num_epochs=4, "
                    "training_batch_size=256) after the self-play legs and report its wall-clock per phase as `learn_loop` (adds minutes)")
    ap.add_argument("--learn-games", type=int, default=None, help="num_self_play_batches of the --learn-loop run (default: --games)")
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="N > 1 typed alone: stop the ranks this process started after so many seconds (0 = never)")
    args = ap.parse_args(argv)
    preset = PRESETS[args.config]
    custom = args.games is not None or args.iterations is not None
    if args.games is None:
        args.games = preset["games"]
    if args.iterations is None:
        args.iterations = preset["iterations"]
    if args.cpu_baseline_only:                   # child of cpu_baseline_guarded: CPU only, never touches the GPU
        print(json.dumps(cpu_baseline(args.iterations, args.seed, args.exp_per_game, games=args.cpu_games, plies_per_game=args.plies_per_game)))
        return

    if args.gpus > 1 and "RANK" not in os.environ and engine_factory is None:
        # the driver's own command shape, `python bench.py --gpus N ...`: start the N ranks as a child process group (nothing that can
        # touch the GPU has been imported by this process, and nothing will be)
        raise SystemExit(launch_ranks(args, list(sys.argv[1:] if argv is None else argv)))
    if engine_factory is None and os.environ.get("DIEE_BENCH_ENGINE"):
        # tests only (tests/test_dist_cpu.py): "module:attr" of a stand-in engine for boxes without a GPU; the line says so (`engine`)
        import importlib
        mod, _, attr = os.environ["DIEE_BENCH_ENGINE"].partition(":")
        engine_factory = getattr(importlib.import_module(mod), attr)

    import importlib
    ddist = importlib.import_module("die-e_amd.dist")
    rank, local_rank, world = ddist.rank_world()
    if args.gpus != world:
        # inside a rank `--gpus N` names the job size and must equal the launcher's WORLD_SIZE (a rank that reported n_gpus = 1 under
        # an 8-GPU flag would be a wrong line, not a slow one)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: `python bench.py --gpus {args.gpus} ...` starts its own ranks; under "
                         f"a launcher use `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {args.gpus} ...`")
    backend = os.environ.get("DIEE_BENCH_BACKEND", "nccl")      # "gloo": the CPU test of this function
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    dev = local_rank
    affinity = None
    shares_gpu = False                           # ranks really share a GPU (tests; never on the 8-GPU node)
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch
        import torch.distributed as dist
        # device_count() does not initialise the GPU; a launcher may have narrowed HIP_VISIBLE_DEVICES to one GPU per rank
        ndev = torch.cuda.device_count()
        narrowed = "HIP_VISIBLE_DEVICES" in os.environ or "ROCR_VISIBLE_DEVICES" in os.environ
        if backend == "nccl" and world > 1 and ndev < world and not (narrowed and ndev >= 1):
            # RCCL cannot put two ranks on one device, and a rank that silently shares a GPU would halve the figure it reports
            raise SystemExit(f"bench.py: {world} ranks but {ndev} visible GPU(s): launch with --nproc-per-node <= the GPUs of the node "
                             "(or one HIP_VISIBLE_DEVICES entry per rank)")
        if backend == "nccl" or ndev > 0:          # (gloo on a GPU box: ranks that share a device, tests/test_dist_gpu.py)
            ndev = max(ndev, 1)
            dev = local_rank % ndev
            if world > ndev and not narrowed:
                # ranks really share a GPU: the small-batch cluster tower needs its workgroups resident together,
                # which two processes on one GPU cannot promise each other: told to the engine
                # below through its ABI (diee_set_option "shared_gpu"), not through the environment
                shares_gpu = True
            torch.cuda.set_device(dev)             # torch's HIP runtime initialises before libdiee.so's
            if world > 1:
                affinity = pin_to_gpu_numa(torch, dev)
        dist.init_process_group(backend)           # "nccl" is RCCL on ROCm

    import diee_amd
    if dist is not None and world > 1 and not shares_gpu and engine_factory is None:
        # a narrowed HIP_VISIBLE_DEVICES may be one GPU per rank (a launcher's doing) or the same GPU for every rank (a one-GPU box): the
        # PCI bus ids tell -- two ranks on one id share it, whatever the ordinals say
        try:
            mine = diee_amd.device_pci_bus_id(dev)
            ids = [None] * world
            dist.all_gather_object(ids, mine)
            shares_gpu = ids.count(mine) > 1
        except Exception as ex:                      # (no GPU: the engine below refuses anyway)
            sys.stderr.write(f"[bench] rank {rank}: PCI bus ids not compared ({ex})\n")
    pci = None
    if dist is not None and backend == "nccl" and engine_factory is None:
        # two HIP runtimes live in this process (PyTorch's bundled one and the system one behind libdiee.so): before the engine is
        # created, make sure both mean the same GPU by ordinal `dev` -- a mismatch would time the engine on another rank's GPU
        ok, ours, theirs = diee_amd.same_gpu_as_torch(dev)
        pci = ours
        if not ok:
            raise SystemExit(f"bench.py rank {rank}: device {dev} is {ours} for libdiee.so but {theirs} for torch: the two HIP runtimes "
                             "enumerate the GPUs differently (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set for one of them only?)")
    eng = (engine_factory or diee_amd.Engine)(dev)   # raises without a GPU: there is no CPU path
    share_lock = None
    if shares_gpu and hasattr(eng, "set_option"):
        if args.share_lock:
            # test mode (tests/test_dist_gpu.py): the ranks keep the kernel set a GPU of their own would run -- in-launch hand-overs
            # included -- and take turns on the shared GPU through an exclusive lock around every engine call
            import fcntl
            share_lock = open(args.share_lock, "a+")
        else:
            eng.set_option("shared_gpu", 1)
            diee_amd.load_library().diee_train_set_bn_coop(0)
    eng.load_weights(diee_amd.random_weights(0))

    class _Turn:                                     # the shared-GPU lock of --share-lock; a no-op otherwise
        def __enter__(self):
            if share_lock is not None:
                fcntl.flock(share_lock, fcntl.LOCK_EX)
        def __exit__(self, *a):
            if share_lock is not None:
                fcntl.flock(share_lock, fcntl.LOCK_UN)
    turn = _Turn()
    cfg = diee_amd.MctsConfig(iterations=args.iterations, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    first_id = ddist.shard_first_game_id(rank, args.games)
    # primer (not a step): pages in the code objects, sizes the HBM arenas and pins the host blocks the records land in
    with turn:
        o = eng.self_play_parallel(args.games, cfg, 1.25, args.seed, first_game_id=first_id, max_steps=1, fetch=True, copy=False)
        o.get("free", lambda: None)()

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()

    def run_step(i, fetch=True):
        """one self_play_parallel call as a host binding the C ABI makes it: the call returns with the MemoryFragments in
        engine-owned host arrays (fetch) -- looked at, handed back -- or with the records left in HBM (the HBM-only leg)"""
        t = time.perf_counter()
        with turn:
            o = eng.self_play_parallel(args.games, cfg, 1.25, args.seed + 0x9E37 * i, ref_quirks=True,
                                       first_game_id=first_id, fetch=fetch, copy=False, max_steps=args.max_steps)
        st = o["stats"]
        if fetch:
            assert len(o["outcome"]) == st["fragments"] and o["ps"].shape == (st["fragments"], 1352), "delivered records != counted records"
            o.get("free", lambda: None)()
        return st, time.perf_counter() - t

    def add(tot, st):
        for k, v in st.items():
            if isinstance(v, list):
                for b, x in enumerate(v):
                    tot[f"{k}.{b}"] = tot.get(f"{k}.{b}", 0) + x
            else:
                tot[k] = tot.get(k, 0) + v

    for i in range(args.warmup):
        run_step(1000 + i)
    barrier()
    t0 = time.perf_counter()
    tot, step_s = {}, []
    for i in range(args.steps):
        st, ds = run_step(i)                        # the call returns after its last record has landed in host memory
        step_s.append(ds)
        add(tot, st)
    barrier()
    dt = time.perf_counter() - t0

    # ---- the same batches with the records left in HBM (what earlier rounds timed): what delivery costs ----
    hbm = None
    H = min(args.hbm_only_steps, args.steps)
    if H > 0:
        barrier()
        th0 = time.perf_counter()
        hb = [run_step(i, fetch=False) for i in range(H)]
        barrier()
        dth = time.perf_counter() - th0
        hbm = {"steps": H, "seconds": dth, "games": sum(st["games"] for st, _ in hb),
               "same_seed_delta_ms": [1e3 * (step_s[i] - hb[i][1]) for i in range(H)]}

    # ---- second figure: the same batches, K at a time, sharing every network launch (diee_self_play_multi) ----
    pipe = None
    if args.pipeline > 1:
        K = args.pipeline
        batches = [(args.games, first_id, args.seed + 0x9E37 * i) for i in range(K)]
        with turn:
            for o in eng.self_play_multi(batches, cfg, 1.25, ref_quirks=True, max_steps=1, fetch=True, copy=False):      # sizes the arenas
                o.get("free", lambda: None)()
        barrier()
        tp0 = time.perf_counter()
        with turn:
            outs = eng.self_play_multi(batches, cfg, 1.25, ref_quirks=True, fetch=True, copy=False, max_steps=args.max_steps)
        sts = [o["stats"] for o in outs]
        for o in outs:
            o.get("free", lambda: None)()
        barrier()
        dtp = time.perf_counter() - tp0
        pipe = {k: sum(st[k] for st in sts) for k in ("games", "expansions", "nn_evals", "nn_rows", "plies", "fragments")}
        pipe["move_steps"] = max(st["move_steps"] for st in sts)
        pipe["tower_seconds"], pipe["tower_launches"], pipe["tower_flops"] = (sts[0][k] for k in ("tower_seconds", "tower_launches", "tower_flops"))
        pipe["deliver_seconds"], pipe["deliver_bytes"] = sts[0].get("deliver_seconds", 0.0), sts[0].get("deliver_bytes", 0)

    keys = ["games", "expansions", "nn_evals", "nn_rows", "plies", "move_steps", "children", "selections", "depth_sum",
            "conv_seconds", "conv_launches", "conv_flops", "tower_seconds", "tower_launches", "tower_flops",
            "full_seconds", "full_launches", "full_flops",
            "cluster_seconds", "cluster_launches", "cluster_flops",
            "fragments", "illegal_decodes", "deliver_seconds", "deliver_bytes", "tail_iterations", "tail_launches", "tail_spec_rows"]
    keys += [f"{n}.{b}" for n in ("band_seconds", "band_launches", "band_flops", "band_flops_demanded") for b in range(len(BAND_NAMES))]
    frags_per_rank = [int(tot.get("fragments", 0))]
    # every rank's own rate over the shared clock window (games it retired / its own time inside the K calls): a straggler shows here
    own_s = sum(step_s)
    rank_values = [round(tot.get("games", 0) / own_s, 3) if own_s > 0 else 0.0]
    rank_pci = [pci]
    if dist is not None:
        rank_values = [v / 1000.0 for v in ddist.gather_counts(dist, int(rank_values[0] * 1000), red_dev)]
        if pci is not None:
            gathered = [None] * world
            dist.all_gather_object(gathered, pci)
            rank_pci = gathered
        # after a self-play batch the ranks all-gather their fragment counts (8 x u64 on a node) -- what a
        # consumer of the sharded records (the learn loop's trainer) needs to size its receive buffers; the records stay put
        frags_per_rank = ddist.gather_counts(dist, frags_per_rank[0], red_dev)
        dt, red = ddist.reduce_stats(dist, dt, tot, keys, red_dev)
        tot.update(red)
        if hbm is not None:
            hbm["seconds"], red = ddist.reduce_stats(dist, hbm["seconds"], hbm, ["games"], red_dev)
            hbm.update(red)
        if pipe is not None:
            pkeys = ["games", "expansions", "nn_evals", "nn_rows", "plies", "fragments", "tower_seconds", "tower_launches", "tower_flops"]
            dtp, red = ddist.reduce_stats(dist, dtp, pipe, pkeys, red_dev)
            pipe.update(red)

    learn = None
    if args.learn_loop and world == 1 and engine_factory is None:
        learn = learn_loop_leg(eng, args.learn_games or args.games, args.iterations)

    if rank == 0:
        games = tot["games"]
        exp_per_game = tot["expansions"] / max(games, 1)
        def pmc_traffic(name, fname=TRAFFIC_FILE):
            # HBM-side bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            # (profiles/*_pmc_traffic*.json, scripts/profile_bench.sh; FETCH_SIZE doubled per the gfx950 correction), at 1024 / 32 boards
            try:
                doc = json.load(open(os.path.join(ROOT, "profiles", fname)))
                for k, v in doc["kernels"].items():
                    if k.startswith(name) and "hbm_side_bytes_per_launch_corrected" in v:
                        return v["hbm_side_bytes_per_launch_corrected"]
                return None
            except Exception:
                return None

        def pmc_mfma(name, fname):
            # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of `bench.py --max-steps 1 --pipeline 0 [--games 32]`
            # (scripts/pmc_mfma.sh): share of the chip's SIMD-cycles with a matrix instruction executing, and the clock held
            try:
                doc = json.load(open(os.path.join(ROOT, "profiles", fname)))
                for k, v in doc["kernels"].items():
                    if k.startswith(name) and "mfma_busy_frac" in v:
                        us = v["duration_us"]
                        us = next(iter(us.values())) if isinstance(us, dict) else us      # (the pass that held SQ_VALU_MFMA_BUSY_CYCLES comes first)
                        # MFMA FLOPs really ISSUED per second against the peak: a v_mfma_f32_16x16x32_bf16 keeps its SIMD busy for 16 cycles
                        # and performs 16 384 FLOP (all-padding fragments are skipped and do not count: this is below `frac`, whose
                        # algorithmic FLOPs include them)
                        issued = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 16.0 * 16384.0
                        return {"mfma_busy_frac": v["mfma_busy_frac"], "clock_mhz": v["clock_mhz"], "us": us,
                                "mfma_issue_frac": issued / (us * 1e-6) / (PEAK_BF16_TFLOPS * 1e12) if us else None}
            except Exception:
                pass
            return None

        sampled_total = tot["conv_seconds"] + tot["tower_seconds"] + tot["cluster_seconds"]

        def roof(kernel, sec, launches, flops, traffic=None, busy=None, busy_file=None):
            if not sec:
                return None
            a = flops / sec / 1e12
            return {"bound": "mfma", "kernel": kernel, "achieved": a, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": a / PEAK_BF16_TFLOPS, "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/ (separate rocprofv3 --pmc passes of this command, committed; not re-measured in this run)",
                    "mfma_busy_frac": None if busy is None else busy["mfma_busy_frac"],
                    "mfma_busy_clock_mhz": None if busy is None else busy["clock_mhz"],
                    "mfma_issue_frac": None if busy is None else busy["mfma_issue_frac"],
                    "mfma_busy_source": None if busy is None else f"profiles/{busy_file} (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                                                                   "of this command's first move-step, committed; busy cycles / (256 CUs x 4 SIMDs x GRBM_GUI_ACTIVE / 8); not re-measured in this run)",
                    "launches_sampled": launches / world,
                    "avg_launch_us": sec / max(launches, 1) * 1e6,
                    "algorithmic_flops_per_launch": flops / max(launches, 1),
                    "share_of_sampled_tower_time": sec / sampled_total}
        # the tower of 38 3x3 convs is ~90 % of the GPU time; it runs as ONE fused launch (k_tower16: activations in LDS)
        # (k_tower16p: a board group over two workgroups, 129 ... 512 boards) while more than 128 games are alive and as ONE
        # cluster launch (k_tower_cl: 8-workgroup clusters per board group) at 128 games or fewer; the two `roofline_other` rows
        # go by band (> 256 / <= 256 boards) as in earlier rounds; the 38 per-layer launches (k_conv3x3_sk) remain as the fallback and the test reference
        # The dominant KERNEL is k_tower16<4,4,3> (the instantiation with BAND = 0): 929 ... 1024 live games, the whole batch in one launch.  Its sampled launches are
        # timed one to one, so `avg_launch_us` is that kernel's AverageNs in a rocprofv3 --kernel-trace --stats summary of the timed leg
        # (`bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0`, profiles/).  `roofline_other` keeps
        # the average over every fused-tower evaluation (257 ... 1024 boards; a compacted evaluation is up to three launches).
        r_full = roof("k_tower16<4,4,3> (38 fused 3x3 conv layers + init block + head convs in one launch, v_mfma_f32_16x16x32_bf16, one wave per SIMD x 4 column fragments; 929 ... 1024 boards = one pass of the chip)",
                      tot["full_seconds"], tot["full_launches"], tot["full_flops"], pmc_traffic("diee::k_tower16<4"),
                      pmc_mfma("diee::k_tower16<4", MFMA_BUSY_FILE), MFMA_BUSY_FILE)
        r_fused = roof("k_tower16 / k_tower16p<4>, every geometry (<4,4,3> at 513 ... 1024 boards, pair tower <= 512 boards): all evaluations of batches > 256 boards, timed per evaluation (a compacted evaluation is up to three launches)",
                       tot["tower_seconds"], tot["tower_launches"], tot["tower_flops"], pmc_traffic("diee::k_tower16<4"))
        r_cluster = roof("k_tower_cl (38 tower layers + head convs + policy FC in one launch, 8-workgroup clusters exchanging activations through tagged device-coherent loads, latency-bound) at <= 128 boards and k_tower16p<2> (two-board pair tower) at 129 ... 256: all evaluations of batches <= 256 boards, the same population as in earlier rounds; FLOPs counted: the 38 tower layers)",
                         tot["cluster_seconds"], tot["cluster_launches"], tot["cluster_flops"],
                         pmc_traffic("diee::k_tower_cl<1", TRAFFIC_FILE_32),
                         pmc_mfma("diee::k_tower_cl<1", MFMA_BUSY_FILE_32), MFMA_BUSY_FILE_32)
        r_layer = roof("k_conv3x3_sk (per-layer 3x3 tower conv, split-K; only when the other two are disabled)",
                       tot["conv_seconds"], tot["conv_launches"], tot["conv_flops"])
        ranked = sorted([r for r in (r_fused, r_cluster, r_layer) if r], key=lambda r: -r["share_of_sampled_tower_time"])
        if r_full:                                   # a single kernel, comparable with its row in a rocprofv3 summary
            dominant, other = r_full, ranked
        elif ranked:                                 # batches that never fill the chip (--games <= 928)
            dominant, other = ranked[0], ranked[1:]
        else:                                        # nothing was sampled (a run too short for a sample)
            dominant, other = None, []
        # end to end = node expansions/s x 1.0825 GFLOP (one evaluated state) / the dense bf16 peak -- the work the SEARCH asked
        # for.  Rows the engine evaluated on speculation (the tail's free rows) and the reference's stale rows are NOT achieved work: they
        # are counted beside it (`end_to_end_frac_rows_evaluated`, `speculative_row_share`), never in it.
        e2e = tot["expansions"] * FLOPS_PER_EVAL / dt / 1e12 / (PEAK_BF16_TFLOPS * world)
        e2e_rows = tot["nn_rows"] * FLOPS_PER_EVAL / dt / 1e12 / (PEAK_BF16_TFLOPS * world)
        spec_share = tot.get("tail_spec_rows", 0) / max(tot["nn_rows"], 1)
        if dominant is not None:
            # FIRST thing to read: the whole job against the roof -- every kernel, every launch gap, the host, the delivery of the
            # records.  `frac` below describes the dominant kernel alone.
            dominant = dict({"end_to_end_frac": e2e, "end_to_end_frac_rows_evaluated": e2e_rows, "speculative_row_share": spec_share,
                             "profile_set": f"profiles/{PROFILE_SET}_*"}, **dominant)
            # where a batch's network time goes as it shrinks: every sampled evaluation of this run (each 17th, HIP events on the
            # engine's stream) binned by the live games of its move-step.  `frac` counts the rows the SEARCH asked for (the demanded
            # leaves of a tail / free-running launch, every row of a plain evaluation), `frac_rows_evaluated` also the speculative ones
            bands = []
            for b, name in enumerate(BAND_NAMES):
                sec, n, fl = tot.get(f"band_seconds.{b}", 0), tot.get(f"band_launches.{b}", 0), tot.get(f"band_flops.{b}", 0)
                fld = tot.get(f"band_flops_demanded.{b}", fl)
                if n:
                    bands.append({"boards": name, "sampled_evaluations": n / world, "share_of_network_time": sec / sampled_total,
                                  "avg_us": sec / n * 1e6, "frac": fld / sec / 1e12 / PEAK_BF16_TFLOPS,
                                  "frac_rows_evaluated": fl / sec / 1e12 / PEAK_BF16_TFLOPS})
            dominant["bands"] = bands
        workload = (f"{preset['name']}: " if not custom else "") + (
            f"backgammon self_play_parallel, num_self_play_batches={args.games} per GPU, "
            f"iterations={args.iterations}, exploration_const=2, temperature=1.25, "
            "simulate_round_limit=400, dirichlet 0.3/0.25, random-init 19x256 ResNet (seed 0), "
            "ref_quirks on, records delivered to host memory inside the timed region"
            + (f", TRUNCATED to {args.max_steps} move-steps per batch (profiling run)" if args.max_steps else ""))
        out = {
            "metric": "self-play games/sec", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "parallelism": f"dp{world} (independent games, no collective)"},
            "node_expansions_per_s": tot["expansions"] / dt,
            # nn_evals = batch rows as the reference counts them (all N slots per iteration, stale rows included);
            # nn_rows = rows the engine really evaluated (stale rows are skipped above 256 live games)
            "nn_evals_per_s": tot["nn_evals"] / dt, "nn_rows_per_s": tot["nn_rows"] / dt,
            "mfma_fraction_end_to_end": e2e, "mfma_fraction_end_to_end_rows_evaluated": e2e_rows,
            "stats": {"games": games, "plies_per_game": tot["plies"] / max(games, 1), "move_steps": tot["move_steps"],
                      "expansions_per_game": exp_per_game, "mean_children": tot["children"] / max(tot["expansions"], 1),
                      "mean_leaf_depth": tot["depth_sum"] / max(tot["selections"], 1),
                      "fragments": tot["fragments"], "illegal_decodes": tot["illegal_decodes"],
                      # SURVEY 8(d) item 4 (configs[3]): the tree one search builds for one game -- nodes created (56 B each: SoA statistics,
                      # links, the 32-byte state) against the arena the engine reserves per live game ((iterations + 1) * 128 + 64 nodes)
                      "tree_nodes_per_search": tot["children"] / max(tot["plies"], 1) + 1,
                      "tree_bytes_per_search": 56 * (tot["children"] / max(tot["plies"], 1) + 1),
                      "tree_arena_bytes_per_game": 56 * ((args.iterations + 1) * 128 + 64),
                      # the free-running search and the tail of a batch (die-e_amd/csrc/search_types.h Free, Tail): search iterations they ran,
                      # the network launches they needed (one per iteration without them) and the rows evaluated ahead of the search, per batch
                      "tail": {"what": "search iterations that ran outside the launch-per-iteration search -- the free-running search (17 ... 800 live games, round 6) and the looping "
                                       "kernel k_tail (<= 16) --, the network launches they needed, the rows of those launches that no game was waiting for when they went out",
                               "iterations_per_step": tot.get("tail_iterations", 0) / max(args.steps, 1) / world,
                               "launches_per_step": tot.get("tail_launches", 0) / max(args.steps, 1) / world,
                               "speculative_rows_per_step": tot.get("tail_spec_rows", 0) / max(args.steps, 1) / world,
                               "launches_per_iteration": tot.get("tail_launches", 0) / max(tot.get("tail_iterations", 0), 1)}},
            "roofline": dominant, "roofline_other": other,
            "value_per_rank": rank_values,           # games/s of each rank by its own clock (value = all games / the slowest rank's window)
            "pci_bus_id_per_rank": rank_pci,         # libdiee.so's GPU per rank, checked against torch's before the engine was created (N > 1)
            "fragments_per_rank": frags_per_rank,    # all_gather of the per-rank record counts; the records stay on their rank
        }
        # what the delivery of the records costs: the same batches (same seeds) with the records left in HBM
        out["output_delivery"] = {
            "bytes_per_step": tot.get("deliver_bytes", 0) / max(args.steps, 1) / world,
            "host_ms_per_step_in_delivery_calls": 1e3 * tot.get("deliver_seconds", 0) / max(args.steps, 1) / world,
            "what": "every move-step's flushed games are relabelled / ordered / gathered on the device (k_deliver_scan, k_deliver_copy) and copied "
                    "into page-locked host arrays on a second stream while the next move-steps search; host_ms = enqueueing those + the wait for "
                    "the last copy after the last move-step"}
        if hbm is not None:
            # value_hbm_only covers the FIRST hbm_only_steps batches only (batches differ by +-2 % in length): compare it with
            # value_delivered_same_seeds, the delivered rate of those very batches -- not with `value`, which covers all K
            out["value_hbm_only"] = hbm["games"] / hbm["seconds"]
            out["value_delivered_same_seeds"] = hbm["games"] / sum(step_s[:hbm["steps"]])      # (N > 1: on rank 0's clock)
            out["output_delivery_ms"] = sum(hbm["same_seed_delta_ms"]) / len(hbm["same_seed_delta_ms"])
            # (a same-seed difference of two ~9 s runs: inside +-15 ms it is run-to-run noise, and it can come out negative)
            out["output_delivery_ms_per_batch"] = hbm["same_seed_delta_ms"]
            out["output_delivery_note"] = ("same-seed delivered - HBM-only time, one value per repeated batch; |x| < ~15 ms is noise.  `value` times the call as a host "
                                           "binding of the C ABI makes it (records in engine-owned page-locked arrays, looked at, handed back); the Python binding's "
                                           "default copy=True adds one host memcpy of the records (~645 MB per batch) on the consumer's side, outside `value`")
            out["output_delivery"].update(hbm_only_steps=hbm["steps"], same_seed_delta_ms=hbm["same_seed_delta_ms"],
                                          delivered_over_hbm_only=(hbm["seconds"] / hbm["steps"]) / (sum(step_s[:hbm["steps"]]) / hbm["steps"]))
        if affinity is not None:
            out["host_affinity_rank0"] = affinity
        out["scale_note"] = ("N>1 never run on hardware by the build (gpurun boxes have one GPU); ranks are independent workers (no data-path collective): expected weak scaling = N x the 1-GPU value; "
                                 "`python bench.py --gpus N` starts its own N ranks (child torch.distributed.run); this code path has run at world size 1 on RCCL and at world size 2 on one shared GPU "
                                 "(tests/test_dist_gpu.py) and at world sizes 2 and 8 on gloo (tests/test_dist_cpu.py)")
        if os.environ.get("DIEE_BENCH_ENGINE"):
            out["engine"] = f"STAND-IN {os.environ['DIEE_BENCH_ENGINE']} (tests only: not the HIP engine, no figure of this line is a measurement)"
        if pipe is not None:
            # NOT the headline: `value` above stays one self_play_parallel call per step, as the reference issues them
            out["value_pipelined"] = pipe["games"] / dtp
            out["pipelined"] = {
                "what": f"{args.pipeline} self_play_parallel batches of {args.games} games per GPU played side by side in one "
                        "diee_self_play_multi call (the learn loop's self_play_iterations, alpha_parallel.rs:49-62): all "
                        "batches share every network launch, each keeps its own seed / Dirichlet stream / Q14 bookkeeping; records delivered",
                "batches": args.pipeline, "seconds": dtp, "games": pipe["games"], "move_steps": pipe["move_steps"],
                "node_expansions_per_s": pipe["expansions"] / dtp, "nn_evals_per_s": pipe["nn_evals"] / dtp,
                "nn_rows_per_s": pipe["nn_rows"] / dtp,
                "mfma_fraction_end_to_end": pipe["expansions"] * FLOPS_PER_EVAL / dtp / 1e12 / (PEAK_BF16_TFLOPS * world),
                "mfma_fraction_end_to_end_rows_evaluated": pipe["nn_rows"] * FLOPS_PER_EVAL / dtp / 1e12 / (PEAK_BF16_TFLOPS * world),
                "fused_tower_tflops": (pipe["tower_flops"] / pipe["tower_seconds"] / 1e12) if pipe["tower_seconds"] else None,
                "fused_tower_avg_launch_us": (pipe["tower_seconds"] / max(pipe["tower_launches"], 1) * 1e6) if pipe["tower_seconds"] else None,
                "deliver_bytes": pipe["deliver_bytes"], "host_ms_in_delivery_calls": 1e3 * pipe["deliver_seconds"],
            }
        if learn is not None:
            out["learn_loop"] = learn
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_guarded(args.iterations, args.seed, exp_per_game, games=args.cpu_games,
                                                       plies_per_game=tot["plies"] / max(games, 1))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
