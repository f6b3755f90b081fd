"""BASELINE.json configs[4]: the full learn loop (learn_iterations=2, self_play_iterations=4, num_epochs=4,
training_batch_size=256) -- wall clock per phase.  One rank per GPU under torch.distributed.run; alone it
runs the single-GPU leg."""
import importlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")     # imports torch first (HIP runtime order)
import diee_amd

games = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
li = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lr = int(os.environ.get("LOCAL_RANK", "0"))
if world > 1:
    import torch, torch.distributed as dist
    torch.cuda.set_device(lr); dist.init_process_group("nccl")
eng = diee_amd.Engine(lr)
conf = az.AlphaZeroConfig(temperature=1.25, learn_iterations=li, self_play_iterations=4, num_epochs=4,
                          training_batch_size=256, num_self_play_batches=games)
root = tempfile.mkdtemp(prefix="diee_learn_")
a = az.AlphaZero(eng, conf, diee_amd.MctsConfig.default(iters), az.OptimizerParams(1e-4, 1e-3), root=root, rank=rank, world=world)
t = time.time()
rep = a.learn_parallel(arena=True, arena_games=400)
if rank == 0:
    print(json.dumps({"config": "learn_iterations=%d self_play_iterations=4 num_epochs=4 training_batch_size=256 "
                                "num_self_play_batches=%d iterations=%d, %d GPU(s)" % (li, games, iters, world),
                      "total_s": time.time() - t, "per_learn_iteration": rep}))
