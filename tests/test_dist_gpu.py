"""The multi-rank code path of bench.py on hardware: launched exactly as the driver launches it for N > 1
(python -m torch.distributed.run ... bench.py --gpus N), here with ONE rank on the one GPU of the box.  RCCL
initialises, torch's HIP runtime and libdiee.so's share the process, barrier + MAX / SUM reductions run on the GPU.
The launcher is a FRESH child process (it starts before anything touches the GPU; nothing is re-exec'ed).
Round 6: the plain `python bench.py --gpus 2` starts its own ranks (second test)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()[-3000:]


def test_bench_under_torch_distributed_run_world_size_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--gpus", "1", "--steps", "1", "--max-steps", "2", "--no-cpu-baseline", "--games", "256", "--pipeline", "0"]
    port = str(29400 + os.getpid() % 500)
    rc, dist_line, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                               "--master-addr", "127.0.0.1", "--master-port", port, "bench.py"] + args, env)
    assert rc == 0, err
    assert dist_line is not None, err
    assert dist_line["n_gpus"] == 1 and dist_line["scaling"] == "weak" and dist_line["node_expansions_per_s"] > 0
    rc2, plain, err2 = _run([sys.executable, "bench.py"] + args, env)
    assert rc2 == 0 and plain is not None, err2
    # same seeds, same shard (rank 0): the counters of the distributed run equal the plain run's
    for key in ("games", "move_steps", "fragments", "illegal_decodes"):
        assert dist_line["stats"][key] == plain["stats"][key], key
    assert abs(dist_line["stats"]["expansions_per_game"] - plain["stats"]["expansions_per_game"]) < 1e-9
    assert dist_line["config"]["parallelism"].startswith("dp1")


def test_bench_two_ranks_on_the_one_gpu_over_gloo():
    """world size 2 on hardware: two ranks share the box's one GPU (RCCL refuses two ranks per device, so the reductions go
    over gloo: DIEE_BENCH_BACKEND); bench.py drops the co-residency kernels when ranks share a device.  Each rank plays its own
    block of game ids on a real engine; the line carries both ranks' fragment counts and the sum of their games."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DIEE_BENCH_BACKEND="gloo")
    args = ["--gpus", "2", "--steps", "1", "--iterations", "6", "--no-cpu-baseline", "--games", "16", "--pipeline", "0"]   # whole games: records exist
    # the PLAIN command, as the driver types it for N > 1: bench.py starts its own two ranks (child torch.distributed.run); the
    # explicit launcher form is the next test's
    rc, line, err = _run([sys.executable, "bench.py"] + args + ["--launch-timeout", "800"], env)
    assert rc == 0, err
    assert line is not None, err
    assert line["n_gpus"] == 2 and line["config"]["parallelism"].startswith("dp2")
    assert line["stats"]["games"] == 2 * 16
    assert line["config"]["workload"].count("num_self_play_batches=16 per GPU") == 1
    assert len(line["fragments_per_rank"]) == 2 and all(f > 0 for f in line["fragments_per_rank"])
    assert sum(line["fragments_per_rank"]) == line["stats"]["fragments"]
    # rank 0 alone plays the same first block of games: the two-rank line holds its records plus rank 1's
    rc1, one, err1 = _run([sys.executable, "bench.py", "--gpus", "1"] + args[2:], dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert rc1 == 0 and one is not None, err1
    assert one["stats"]["fragments"] == line["fragments_per_rank"][0]


def test_two_ranks_take_turns_on_the_one_gpu_with_the_full_kernel_set(tmp_path):
    """what 8 GPUs will run, as far as one GPU can show it: TWO processes under torch.distributed.run, each after torch.cuda.set_device +
    process-group set-up, each with the kernel set a GPU of its own gets -- the in-launch hand-overs of the cluster tower (<= 128 boards)
    and the pair tower (129 ... 512) LEFT ON -- taking turns on the shared GPU through a lock file around every engine call
    (bench.py --share-lock; without it bench.py tells the engines `shared_gpu` through diee_set_option, previous test).  300 games per
    rank played to completion cross every dispatch band below 513 boards; no hand-over starves, and rank 0's records are the ones it
    produces alone."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("DIEE_")}          # a clean environment: options travel through the ABI
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", DIEE_BENCH_BACKEND="gloo")
    args = ["--gpus", "2", "--steps", "1", "--iterations", "4", "--no-cpu-baseline", "--games", "300", "--pipeline", "0", "--hbm-only-steps", "0",
            "--share-lock", str(tmp_path / "gpu.lock")]
    port = str(29700 + os.getpid() % 90)
    rc, line, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, "bench.py"] + args, env)
    assert rc == 0, err
    assert line is not None, err
    assert "starved" not in err and "falling back" not in err, err
    assert line["n_gpus"] == 2 and line["stats"]["games"] == 2 * 300 and line["stats"]["illegal_decodes"] == 0
    assert len(line["value_per_rank"]) == 2 and all(v > 0 for v in line["value_per_rank"])
    one_env = {k: v for k, v in os.environ.items() if not k.startswith("DIEE_")}
    one_env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc1, one, err1 = _run([sys.executable, "bench.py", "--gpus", "1"] + args[2:-2], one_env)
    assert rc1 == 0 and one is not None, err1
    assert one["stats"]["fragments"] == line["fragments_per_rank"][0]                   # same games, same kernels, same records
