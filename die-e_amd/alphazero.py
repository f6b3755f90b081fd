"""Host-side callers of the hot path, mirroring the reference's driver layer (SURVEY section 8(f) rows
F1, F3, F4): AlphaZero::{from_config, self_play_parallel, train, learn_parallel, save/load_training_data,
play_vs_best_model} (src/alphazero/alphazero.rs, alpha_parallel.rs, alpha_versus.rs).

Self-play and model-vs-model search run on the HIP engine through the C ABI.  The training step is
PyTorch-ROCm (the reference's is tch/libtorch autograd; SURVEY: "do it in PyTorch-ROCm, not
hand-written backward"), data-parallel over ranks with DistributedDataParallel on RCCL.  Tensors are
stored as .npy; libtorch .ot archives (models and training data) are read wherever a path is given and
written on request (die-e_amd/ot.py, scripts/ot_convert.py: F3).
"""
import os
import secrets
import time

import numpy as np

# the init block and the two head convolutions stay in PyTorch (MIOpen): its default exhaustive solver search costs 5.3 s on
# the first training step of a process (naive reference kernels included) against 0.7 s with the heuristic pick, for three
# convolutions that are 2 % of the step.  The user's own setting wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch

from . import BG_ACTIONS, BG_PLANES, ERR_ARG, ERR_HIP, GAME_BACKGAMMON, GAME_TTT, DieeError, Engine, MctsConfig, random_weights
from . import ot as _ot

# PyTorch bundles its own HIP runtime; it has to initialise BEFORE libdiee.so's (system ROCm) runtime does,
# otherwise torch.cuda reports no device.  Importing this module first (the CLI and the learn loop do) is enough.
TORCH_CUDA = torch.cuda.is_available()

try:                       # Python 3.11+
    import tomllib as _toml
except ImportError:        # this image: Python 3.10 + tomli
    import tomli as _toml


# --------------------------------------------------------------------------- config (alphazero.rs:25-59, lib.rs:43-51)
CONFIG_KEYS = ("temperature", "learn_iterations", "num_epochs", "training_batch_size", "self_play_iterations",
               "num_self_play_batches", "iterations", "exploration_const", "simulate_round_limit",
               "dirichlet_alpha", "dirichlet_epsilon", "wd", "lr")


def load_config(path):
    """the reference loads `-c FILE` as TOML through the `config` crate (main.rs:88-98); 13 flat keys"""
    with open(path, "rb") as f:
        conf = _toml.load(f)
    missing = [k for k in CONFIG_KEYS if k not in conf]
    if missing:
        raise KeyError(f"Unable to load config, missing keys: {missing}")      # from_config panics, alphazero.rs:113-127
    return conf


class AlphaZeroConfig:
    def __init__(self, temperature, learn_iterations, self_play_iterations, num_epochs, training_batch_size,
                 num_self_play_batches):
        self.temperature = float(temperature)
        self.learn_iterations = int(learn_iterations)
        self.self_play_iterations = int(self_play_iterations)
        self.num_epochs = int(num_epochs)
        self.training_batch_size = int(training_batch_size)
        self.num_self_play_batches = int(num_self_play_batches)

    @classmethod
    def from_config(cls, conf):                                                 # alphazero.rs:35-44
        return cls(conf["temperature"], conf["learn_iterations"], conf["self_play_iterations"], conf["num_epochs"],
                   conf["training_batch_size"], conf["num_self_play_batches"])


def mcts_config_from(conf):                                                     # lib.rs:43-51
    return MctsConfig(iterations=int(conf["iterations"]), c=float(conf["exploration_const"]),
                      round_limit=int(conf["simulate_round_limit"]), dir_alpha=float(conf["dirichlet_alpha"]),
                      dir_eps=float(conf["dirichlet_epsilon"]))


class OptimizerParams:                                                          # alphazero.rs:48-59
    def __init__(self, wd, lr):
        self.wd, self.lr = float(wd), float(lr)

    @classmethod
    def from_config(cls, conf):
        return cls(conf["wd"], conf["lr"])


# --------------------------------------------------------------------------- the two games (base.rs:8-51 consts)
class GameSpec:
    """LearnableGame's associated constants + name(): backgammon_logic.rs:74-78,96-98 / tictactoe/mod.rs:18-34"""
    def __init__(self, name, game_id, filters, blocks, actions, cin, h, w):
        self.name, self.game_id, self.filters, self.blocks, self.actions, self.cin, self.h, self.w = name, game_id, filters, blocks, actions, cin, h, w
        self.planes = cin * h * w


BACKGAMMON = GameSpec("backgammon", GAME_BACKGAMMON, 256, 19, BG_ACTIONS, 6, 4, 6)
TICTACTOE = GameSpec("tictactoe", GAME_TTT, 64, 4, 9, 3, 3, 3)


# --------------------------------------------------------------------------- trainable ResNet (nnet.rs:24-34,57-155)
def make_resnet(spec=BACKGAMMON):
    import torch
    from torch import nn

    F, BLOCKS, A, HW = spec.filters, spec.blocks, spec.actions, spec.h * spec.w

    class ResBlock(nn.Module):                                                  # nnet.rs:16-46
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(F, F, 3, padding=1); self.conv2 = nn.Conv2d(F, F, 3, padding=1)
            self.bn1 = nn.BatchNorm2d(F); self.bn2 = nn.BatchNorm2d(F)

        def forward(self, x):
            h = torch.relu(self.bn1(self.conv1(x)))
            return torch.relu(self.bn2(self.conv2(h)) + x)

    class ResNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.init_conv = nn.Conv2d(spec.cin, F, 3, padding=1); self.init_bn = nn.BatchNorm2d(F)
            self.blocks = nn.ModuleList([ResBlock() for _ in range(BLOCKS)])
            self.p_conv = nn.Conv2d(F, 32, 3, padding=1); self.p_bn = nn.BatchNorm2d(32); self.p_fc = nn.Linear(32 * HW, A)
            self.v_conv = nn.Conv2d(F, 3, 3, padding=1); self.v_bn = nn.BatchNorm2d(3); self.v_fc = nn.Linear(3 * HW, 1)

        engine_tower = False        # True: the 38 tower convolutions and their BatchNorm + ReLU run on the engine's own
                                    # kernels in the bf16 token layout (die-e_amd/train_ops.py); needs CUDA tensors

        def forward_train(self, x):
            """raw policy logits + tanh value (nnet.rs:137-148; the value head's tanh is inside the head)"""
            if self.engine_tower and x.is_cuda and self.training:
                from . import train_ops
                return train_ops.forward_train_tokens(self, x)
            x = torch.relu(self.init_bn(self.init_conv(x)))
            for b in self.blocks:
                x = b(x)
            logits = self.p_fc(torch.relu(self.p_bn(self.p_conv(x))).flatten(1))
            value = torch.tanh(self.v_fc(torch.relu(self.v_bn(self.v_conv(x))).flatten(1)))
            return logits, value

        forward = forward_train

        def _tensors(self):
            """the blob order of include/diee.h (creation order of nnet.rs:62-97)"""
            def conv(c): return [c.weight, c.bias]
            def bn(b): return [b.weight, b.bias, b.running_mean, b.running_var]
            out = conv(self.init_conv) + bn(self.init_bn)
            for b in self.blocks:
                out += conv(b.conv1) + conv(b.conv2) + bn(b.bn1) + bn(b.bn2)
            out += conv(self.p_conv) + bn(self.p_bn) + [self.p_fc.weight, self.p_fc.bias]
            out += conv(self.v_conv) + bn(self.v_bn) + [self.v_fc.weight, self.v_fc.bias]
            return out

        @torch.no_grad()
        def load_blob(self, blob):
            t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32))
            o = 0
            for p in self._tensors():
                n = p.numel()
                p.copy_(t[o:o + n].reshape(p.shape)); o += n
            assert o == t.numel(), (o, t.numel())
            return self

        @torch.no_grad()
        def to_blob(self):
            return torch.cat([p.detach().float().reshape(-1).cpu() for p in self._tensors()]).numpy()

    return ResNet()


# --------------------------------------------------------------------------- AlphaZero (alphazero.rs:61-67)
class AlphaZero:
    def __init__(self, engine, config, mcts_config, op, blob=None, train_device=None, seed=0xD1EE0001,
                 rank=0, world=1, root=".", quiet=False, game=BACKGAMMON, train_backend=None):
        import torch
        self.engine, self.config, self.mcts_config, self.op = engine, config, mcts_config, op
        self.game = game                                            # handle_command::<T>, main.rs:119: the same driver for either game
        self._in_shape = (game.cin, game.h, game.w)
        self.rank, self.world, self.root, self.quiet = rank, world, root, quiet
        self.seed = seed
        self.calls = 0
        self.shuffle_rng = np.random.default_rng(seed ^ 0x5EED)     # memory.shuffle(&mut thread_rng()) once per train() call, alphazero.rs:203-204
        self.blob = np.ascontiguousarray(blob if blob is not None else random_weights(0, game.game_id), dtype=np.float32)
        if engine is not None:
            engine.load_weights(self.blob)
        self.device = train_device or ("cuda" if TORCH_CUDA and game is BACKGAMMON else "cpu")   # tic-tac-toe: BASELINE configs[0], CPU
        self.model = make_resnet(game).load_blob(self.blob).to(self.device)
        self.ddp = None
        if world > 1:
            from torch.nn.parallel import DistributedDataParallel as DDP
            dev = torch.device(self.device)
            self.ddp = DDP(self.model, device_ids=[dev.index if dev.index is not None else torch.cuda.current_device()]
                           if dev.type == "cuda" else None)
            if dev.type == "cuda":
                # a data-parallel step overlaps RCCL's all-reduce kernels with the backward pass: the one-launch BatchNorm
                # passes (every workgroup resident at once, meeting on a device counter) would share the CUs with them
                self._set_bn_coop(False)
        on_gpu = torch.device(self.device).type == "cuda"
        # The reference trains in fp32 (tch autograd, alphazero.rs:202-261), and so does the DEFAULT here: the all-PyTorch fp32 step
        # ("fp32"; "torch" is the old name).  train_backend / DIEE_TRAIN = "bf16" (old name "engine") opts into the 3 x faster step
        # whose tower runs on the engine's bf16 MFMA kernels (die-e_amd/train_ops.py; fp32 master weights and accumulation, bf16
        # activations): its gradients sit within mixed-precision noise of fp32 (tests/test_train_gpu.py: 300-step loss curves,
        # held-out policy KL / value MSE and an arena between the two resulting networks), not within fp32 rounding.
        backend = (train_backend or os.environ.get("DIEE_TRAIN", "fp32")).lower()
        if backend not in ("fp32", "torch", "bf16", "engine"):
            raise ValueError(f"train backend {backend!r}: fp32 (default) or bf16")
        self.train_backend = "bf16" if backend in ("bf16", "engine") and on_gpu and game is BACKGAMMON else "fp32"
        self.model.engine_tower = self.train_backend == "bf16"
        if self.train_backend == "bf16":
            self.log("[train] opt-in bf16 training step (the tower on the engine's MFMA kernels); the default, like the reference, is fp32")
        # the whole step (forward, backward, Adam) replayed as one HIP graph for full batches (single-rank training only)
        self.use_graph = on_gpu and world == 1 and os.environ.get("DIEE_TRAIN_GRAPH", "1") != "0"
        self._graph = None
        # Adam::default().wd(op.wd).build(&vs, op.lr), alphazero.rs:102 (L2 added to the gradient, not AdamW)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=op.lr, betas=(0.9, 0.999), eps=1e-8,
                                          weight_decay=op.wd, fused=on_gpu, capturable=self.use_graph)

    @classmethod
    def from_config(cls, engine, conf, model_path=None, **kw):                   # alphazero.rs:113-127, :81-100
        blob = None
        mdir = os.path.join(kw.get("root", "."), "models", kw.get("game", BACKGAMMON).name)
        best = next((p for p in (os.path.join(mdir, "best_model.npy"), os.path.join(mdir, "best_model.ot")) if os.path.exists(p)), None)
        if model_path:
            blob = _ot.load_model(model_path)                                    # .npy blob or die-e's own .ot archive
        elif best:
            blob = _ot.load_model(best)
        return cls(engine, AlphaZeroConfig.from_config(conf), mcts_config_from(conf), OptimizerParams.from_config(conf),
                   blob=blob, **kw)

    def log(self, *a):
        if not self.quiet and self.rank == 0:
            print(*a, flush=True)

    # ---- self_play_parallel, alpha_parallel.rs:101-231 (the hot path, on the HIP engine) ----
    def self_play_parallel(self):
        n = self.config.num_self_play_batches
        self.calls += 1
        out = self.engine.self_play_parallel(n, self.mcts_config, self.config.temperature,
                                             seed=self.seed + 0x9E3779B1 * self.calls, ref_quirks=True,
                                             first_game_id=self.rank * n)
        self.last_stats = out["stats"]
        return {"outcome": out["outcome"], "ps": out["ps"], "state": out["state"]}

    def self_play_iterations_pipelined(self):
        """the `for sp_i in 0..self_play_iterations` loop of learn_parallel (alpha_parallel.rs:49-62) as ONE
        diee_self_play_multi call: the calls share the network, so they are played side by side (same seeds, game ids
        and per-call outputs as the sequential calls; the GPU does not idle through each call's tail)"""
        n, K = self.config.num_self_play_batches, self.config.self_play_iterations
        batches = []
        for _ in range(K):
            self.calls += 1
            batches.append((n, self.rank * n, self.seed + 0x9E3779B1 * self.calls))
        # a call takes at most 64 batches (kMaxSegments) and its fragment arena grows with the batches in flight
        # (games x (round_limit + 2) x 6 KB: 2.5 GB per 1024-game batch): groups within both limits, one call per group
        per_batch = n * (self.mcts_config.round_limit + 2) * (self.game.actions + self.game.planes) * 4 + n * (self.mcts_config.iterations + 1) * 128 * 56
        budget = int(float(os.environ.get("DIEE_PIPELINE_GB", "96")) * 2 ** 30)
        group = max(1, min(64, budget // max(per_batch, 1)))
        outs = []
        for g0 in range(0, K, group):
            part = batches[g0:g0 + group]
            try:
                outs += self.engine.self_play_multi(part, self.mcts_config, self.config.temperature, ref_quirks=True)
            except DieeError as e:
                if e.status not in (ERR_ARG, ERR_HIP) or len(part) == 1:
                    raise
                self.log(f"[self-play] {len(part)} batches side by side failed ({e}); playing them one after the other")
                for nb, first, seed in part:
                    outs.append(self.engine.self_play_parallel(nb, self.mcts_config, self.config.temperature, seed=seed,
                                                               ref_quirks=True, first_game_id=first))
        self.last_stats = outs[-1]["stats"]
        return [{"outcome": o["outcome"], "ps": o["ps"], "state": o["state"]} for o in outs]

    # ---- save/load_training_data, alphazero.rs:149-200 (ps [M,1352], states [M,6,4,6], outcomes [M] i8) ----
    @staticmethod
    def save_training_data(memory, path, fmt=None):
        """fmt: "npy" (default), "ot" (the reference's ps.ot / states.ot / outcomes.ot) or "both"; DIEE_DATA_FORMAT overrides"""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"path: {path} does not exist!")
        fmt = fmt or os.environ.get("DIEE_DATA_FORMAT", "npy")
        shape = (3, 3, 3) if memory["state"].shape[-1] == 27 else (6, 4, 6)      # states [M,C,H,W] like Tensor::concat of as_tensor, alphazero.rs:160-170
        arrays = {"ps": memory["ps"], "states": memory["state"].reshape(-1, *shape), "outcomes": memory["outcome"].astype(np.int8)}
        for stem, a in arrays.items():
            if fmt in ("npy", "both"):
                np.save(os.path.join(path, stem + ".npy"), a)
            if fmt in ("ot", "both"):
                _ot.save_tensor_ot(a, os.path.join(path, stem + ".ot"))

    @staticmethod
    def load_training_data(path):
        if not os.path.isdir(path):
            raise FileNotFoundError(f"path: {path} does not exist!")

        def one(stem):
            npy = os.path.join(path, stem + ".npy")
            return np.load(npy) if os.path.exists(npy) else _ot.load_tensor_ot(os.path.join(path, stem + ".ot"))
        ps = one("ps")
        ps = ps.reshape(-1, ps.shape[-1]).astype(np.float32)                     # [M, ACTION_SPACE_SIZE]
        st = one("states")
        return {"ps": ps, "state": st.reshape(len(ps), -1).astype(np.float32), "outcome": one("outcomes").reshape(-1).astype(np.int8)}

    @staticmethod
    def concat(mems):
        mems = [m for m in mems if len(m["outcome"])]
        if not mems:
            return {"outcome": np.zeros(0, np.int8), "ps": np.zeros((0, BG_ACTIONS), np.float32),
                    "state": np.zeros((0, BG_PLANES), np.float32)}            # (empty: the shapes carry no game)
        return {k: np.concatenate([m[k] for m in mems]) for k in ("outcome", "ps", "state")}

    # ---- train, alphazero.rs:202-261 ----
    def _loss(self, net, st, ps, oc):
        import torch.nn.functional as Fn
        logits, value = net(st)
        policy_loss = Fn.cross_entropy(logits.float(), ps)                      # soft targets = un-renormalised ps (Q17), :239-245
        outcome_loss = Fn.mse_loss(value.float(), oc)                           # :246
        return policy_loss + outcome_loss

    def _snapshot(self):
        import torch
        return ({k: v.detach().clone() for k, v in self.model.state_dict().items()},
                {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.optimizer.state.items()})

    def _restore(self, snap):
        """put parameters, BatchNorm statistics and Adam's state back IN PLACE (same tensors, same addresses)"""
        import torch
        model_snap, opt_snap = snap
        self.model.load_state_dict(model_snap)                                  # copies into the existing tensors
        for p, st in self.optimizer.state.items():
            for k, v in st.items():
                if torch.is_tensor(v):
                    if p in opt_snap and k in opt_snap[p]:
                        v.copy_(opt_snap[p][k])
                    else:
                        v.zero_()                                               # state born during a warm-up: Adam starts from zero moments, step 0

    def _graph_selfcheck(self, g, st, ps, oc):
        """One replay on a real batch, checked against the eager loss of the same batch, then undone.  On this ROCm stack
        the first replays after a fresh large hipMalloc have been seen to compute a zero policy loss (all-PyTorch steps
        too) until the device was synchronised behind one of them; the training buffers are therefore allocated before the
        capture, and this check makes sure of the graph at the start of every train() call.  False = fall back to eager."""
        import torch
        for _ in range(3):
            snap = self._snapshot()
            g["st"].copy_(st); g["ps"].copy_(ps); g["oc"].copy_(oc)
            torch.cuda.synchronize()
            g["graph"].replay()
            torch.cuda.synchronize()
            got = float(g["loss"].detach())
            self._restore(snap)
            with torch.no_grad():
                want = float(self._loss(self.model, g["st"], g["ps"], g["oc"]))
            self._restore(snap)                                                 # the eager forward moved the BatchNorm statistics
            torch.cuda.synchronize()
            if np.isfinite(got) and abs(got - want) <= 2e-2 * max(1.0, abs(want)):
                return True
        return False

    def _graphed_step(self, bs):
        """forward + backward + Adam of one full batch captured once as a HIP graph (static input tensors; every kernel of
        the step, the engine's included, goes out on the capturing stream): a replay costs one launch instead of ~1500"""
        import torch
        if self._graph is not None and self._graph["bs"] == bs:
            return self._graph
        g = {"bs": bs, "st": torch.zeros(bs, *self._in_shape, device=self.device), "ps": torch.zeros(bs, self.game.actions, device=self.device),
             "oc": torch.zeros(bs, 1, device=self.device)}
        g["ps"][:, 0] = 1.0
        # warm-up and capture run real steps: parameters, BatchNorm statistics and Adam's moments are put back afterwards,
        # IN PLACE (the captured graph holds the addresses of these very tensors)
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                                                  # allocator / autotune warm-up outside the capture
                self.optimizer.zero_grad(set_to_none=True)
                self._loss(self.model, g["st"], g["ps"], g["oc"]).backward()
                self.optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # no .grad tensors going in: the captured backward then WRITES its gradients (into graph-pool memory the captured Adam
        # reads) instead of accumulating into zero-filled ones -- 250 fills and 250 adds per step less
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss = self._loss(self.model, g["st"], g["ps"], g["oc"])
            loss.backward()
            self.optimizer.step()
        g["graph"], g["loss"] = graph, loss
        self._restore(snap)
        self._graph = g
        return g

    def train(self, memory, rng=None, resident=False):
        """resident=True: `memory` is the one the previous call uploaded (the learn loop's epochs 2..n over one memory):
        only the new permutation travels to the device"""
        import torch
        n = len(memory["outcome"])
        timing = os.environ.get("DIEE_TRAIN_TIMING") == "1"                     # wall clock of the call's phases on the log
        t_ph = [time.time()]

        def phase(name):
            if timing:
                torch.cuda.synchronize() if torch.cuda.is_available() else None
                t_ph.append(time.time()); self.log(f"[train timing] {name}: {t_ph[-1] - t_ph[-2]:.2f} s")
        rng = rng or self.shuffle_rng                                           # a fresh permutation every call (every epoch)
        perm = rng.permutation(n)                                               # memory.shuffle(&mut rng), :203-204
        net = self.ddp or self.model
        net.train()                                                             # forward_train(.., true): BN batch statistics
        bs = self.config.training_batch_size
        n_steps = -(-n // bs)
        if self.ddp is not None:
            # ranks play different games, so their fragment counts differ: every rank must take the SAME number of
            # all-reduced steps or the longest one waits in backward() forever.  All take the minimum; the ranks with
            # more data leave their last permuted fragments out of this epoch (a fresh permutation comes next epoch).
            import torch.distributed as dist
            t = torch.tensor([n_steps], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_steps = int(t.item())
        on_gpu = torch.device(self.device).type == "cuda"
        if on_gpu:
            # The whole memory goes to HBM once (0.44 M fragments = 2.6 GB), into buffers that persist across calls and only
            # grow.  Growing them drops the captured graph first: on this ROCm stack the replays that follow a fresh
            # hipMalloc of a large block compute garbage until the device has been synchronised after one of them
            # (observed: the policy loss of the first replays reads 0; an all-PyTorch step under torch.cuda.graph shows the
            # same), so the step is captured only after every large allocation of the call exists.
            cap = getattr(self, "_mem_cap", 0)
            if n > cap:
                self._graph = None
                cap = max(1 << 14, 1 << int(np.ceil(np.log2(n))))
                self._mem = None
                torch.cuda.empty_cache()
                self._mem = {"st": torch.empty(cap, self.game.planes, device=self.device), "ps": torch.empty(cap, self.game.actions, device=self.device),
                             "oc": torch.empty(cap, device=self.device), "perm": torch.empty(cap, dtype=torch.int64, device=self.device),
                             "loss": torch.zeros(-(-cap // max(bs, 1)) + 1, device=self.device)}
                self._mem_cap = cap
                self._mem_n = -1
                torch.cuda.synchronize()
            M = self._mem
            if not (resident and getattr(self, "_mem_n", -1) == n):             # 2.6 GB of pageable host memory per upload: once per memory
                M["st"][:n].copy_(torch.from_numpy(memory["state"])); M["ps"][:n].copy_(torch.from_numpy(memory["ps"]))
                M["oc"][:n].copy_(torch.from_numpy(memory["outcome"].astype(np.float32)))
                self._mem_n = n
            M["perm"][:n].copy_(torch.from_numpy(perm))
            mem_st, mem_ps, mem_oc, perm_t = M["st"], M["ps"], M["oc"], M["perm"]
            loss_buf = M["loss"] if len(M["loss"]) >= n_steps else torch.zeros(max(n_steps, 1), device=self.device)
        else:
            loss_buf = torch.zeros(max(n_steps, 1))                            # read back once: no host sync per step
        phase("buffers + upload")
        use_graph = self.use_graph and n >= bs
        if use_graph:
            g = self._graphed_step(bs)
            i0 = perm_t[:bs]
            if not self._graph_selfcheck(g, mem_st[i0].reshape(-1, *self._in_shape), mem_ps[i0], mem_oc[i0].unsqueeze(1)):
                self.log("[train] the captured step does not reproduce the eager loss: training eagerly")
                use_graph = False
        # The reference asserts on the loss BEFORE backward / step of every batch (alphazero.rs:248-255); here the losses are
        # read back once per epoch (no host sync per step), so the epoch runs on a snapshot: a non-finite loss anywhere in it
        # puts parameters, BatchNorm statistics and Adam's moments back to where the epoch started before anything is raised --
        # a caller that catches the exception holds the model it had, not one a NaN went through Adam on.
        phase("capture + self-check")
        snap = self._snapshot()

        def run_epoch(use_graph):
            for i, b0 in enumerate(range(0, n_steps * bs, bs)):                 # :205-206
                if on_gpu:
                    idx = perm_t[b0:min(b0 + bs, n)]
                    st = mem_st[idx].reshape(-1, *self._in_shape); ps = mem_ps[idx]; oc = mem_oc[idx].unsqueeze(1)
                else:
                    idx = perm[b0:b0 + bs]
                    st = torch.from_numpy(memory["state"][idx]).reshape(-1, *self._in_shape)
                    ps = torch.from_numpy(memory["ps"][idx]); oc = torch.from_numpy(memory["outcome"][idx].astype(np.float32)).unsqueeze(1)
                if use_graph and len(idx) == bs:
                    g["st"].copy_(st); g["ps"].copy_(ps); g["oc"].copy_(oc)
                    g["graph"].replay()
                    loss_buf[i] = g["loss"].detach()
                else:
                    loss = self._loss(net, st, ps, oc)
                    self.optimizer.zero_grad(set_to_none=True)                  # (the captured step keeps its own gradient tensors)
                    loss.backward()
                    self.optimizer.step()
                    loss_buf[i] = loss.detach()
            return loss_buf[:n_steps].tolist()

        losses = run_epoch(use_graph)
        phase(f"{n_steps} steps")
        if not np.isfinite(losses).all():
            self._restore(snap)
            # a starved one-launch BatchNorm pass (something else held CUs: another process, a collective) writes NaN
            # statistics and raises its timeout flag: re-run the epoch once on the three-launch passes before giving up
            timed_out = self._bn_coop_timeouts(clear=True) if on_gpu else 0
            if timed_out > 0:
                self.log(f"[train] a one-launch BatchNorm pass timed out (flags {timed_out}): epoch restored and repeated on the three-launch passes")
                self._set_bn_coop(False)
                self._graph = None                                              # the captured step holds the one-launch kernels
                losses = run_epoch(False)
                if not np.isfinite(losses).all():
                    self._restore(snap)
                    raise FloatingPointError("Total loss is nan or inf!")
            else:
                raise FloatingPointError("Total loss is nan or inf!")          # :248-255
        return losses

    @staticmethod
    def _bn_coop_timeouts(clear=True):
        from . import load_library
        return int(load_library().diee_train_bn_coop_timeouts(1 if clear else 0))

    @staticmethod
    def _set_bn_coop(on):
        from . import load_library
        load_library().diee_train_set_bn_coop(1 if on else 0)

    def sync_engine(self):
        """fold the trained weights back into the HIP engine (BN running statistics included)"""
        self.model.eval()
        self.blob = self.model.to_blob()
        if self.ddp is not None:
            # parameters are identical on every rank (all-reduced gradients), BatchNorm running statistics are not
            # (each rank saw its own batches): every engine gets rank 0's blob
            import torch.distributed as dist
            t = torch.from_numpy(self.blob).to(self.device)
            dist.broadcast(t, src=0)
            self.blob = t.cpu().numpy()
            self.model.load_blob(self.blob)
        if not np.isfinite(self.blob).all():
            raise FloatingPointError("nan variables detected!")                 # alpha_parallel.rs:83
        if self.engine is not None:
            self.engine.load_weights(self.blob)

    # ---- learn_parallel, alpha_parallel.rs:17-99 ----
    def learn_parallel(self, arena=True, arena_games=400, pipelined=True):
        run_id = secrets.token_urlsafe(16)[:21]                                 # nanoid!()
        base = os.path.join(self.root, "data", self.game.name, f"run-{run_id}")
        if self.rank == 0:
            os.makedirs(base, exist_ok=True)
        self.log(f"Staring up run with run_id: {run_id}")
        report = []
        for l_i in range(self.config.learn_iterations):                         # :41
            lrn = os.path.join(base, f"lrn-{l_i}")
            memory = []
            t_sp = time.time()
            played = (self.self_play_iterations_pipelined() if pipelined and self.game is BACKGAMMON and hasattr(self.engine, "self_play_multi")
                      else None)                                        # (the tic-tac-toe host path plays its batches one after the other)
            for sp_i in range(self.config.self_play_iterations):                # :49
                memory.append(played[sp_i] if played is not None else self.self_play_parallel())
                if self.rank == 0:
                    sp_dir = os.path.join(lrn, f"sp-{sp_i}")
                    os.makedirs(sp_dir, exist_ok=True)
                    self.save_training_data(self.concat(memory), sp_dir)        # cumulative memory (Q20), :53,62
            t_sp = time.time() - t_sp
            mem = self.concat(memory)
            t_tr = time.time()
            losses, epoch_means = [], []
            for ep in range(self.config.num_epochs):                            # :78-81
                ep_losses = self.train(mem, resident=ep > 0)
                losses += ep_losses
                # the signal of an epoch is its MEAN loss: the last step of an epoch is the reference's partial batch
                # (n mod training_batch_size samples, alphazero.rs:205-206 -- 4 samples in the full-size configs[4] run)
                epoch_means.append(float(np.mean(ep_losses)) if ep_losses else None)
            self.sync_engine()
            t_tr = time.time() - t_tr
            if self.rank == 0:
                mdir = os.path.join(self.root, "models", self.game.name)
                os.makedirs(mdir, exist_ok=True)
                np.save(os.path.join(mdir, f"model_{l_i}.npy"), self.blob)      # :85-95
                self.log(f"Iteration {l_i} saved successfully; {len(mem['outcome'])} fragments, self-play {t_sp:.1f} s, "
                         f"train {t_tr:.1f} s, loss {losses[0]:.4f} -> {losses[-1]:.4f}, epoch means "
                         + " ".join(f"{m:.4f}" for m in epoch_means))
            t_ar = time.time()
            verdict = self.play_vs_best_model(n_games=arena_games) if arena and self.rank == 0 else None   # :96
            t_ar = time.time() - t_ar
            report.append({"learn_iteration": l_i, "fragments": len(mem["outcome"]), "self_play_s": t_sp, "train_s": t_tr, "arena_s": t_ar,
                           "loss_first": losses[0] if losses else None, "loss_last": losses[-1] if losses else None,
                           "epoch_loss_means": epoch_means, "train_steps": len(losses), "arena": verdict})
        return report

    # ---- play_vs_best_model / play_vs_model, alpha_versus.rs:16-81 ----
    def play_vs_best_model(self, n_games=400):
        from .versus import Agent, Player, play
        if self.game is TICTACTOE:
            from .versus import play_tictactoe as play
        mdir = os.path.join(self.root, "models", self.game.name)
        best = os.path.join(mdir, "best_model.npy")
        if not os.path.exists(best) and os.path.exists(os.path.join(mdir, "best_model.ot")):
            np.save(best, _ot.load_model_ot(os.path.join(mdir, "best_model.ot")))   # a die-e checkpoint brought along
        if not os.path.exists(best):
            self.log("No best model was found, saving current model as best...")
            os.makedirs(mdir, exist_ok=True)
            np.save(best, self.blob)
            return "saved-as-best"
        other = Engine(self.engine.device, self.game.game_id)
        other.load_weights(np.load(best))
        res = play(Player(Agent.MODEL, self.engine), Player(Agent.MODEL, other), self.mcts_config,
                   self.config.temperature, seed=self.seed + 77 * self.calls, num_games=n_games)
        other.close()
        self.log(f"Match result: {res}")
        if res.winrate >= 0.55:                                                 # alpha_versus.rs:74-80
            np.save(best, self.blob)
            return "new model was better!"
        if res.winrate <= 0.45:
            return "current best model is still better!"
        return "new model vs current best was inconclusive, keeping current best!"
