"""die-e's command line (src/main.rs:15-216) over the HIP engine:

    diee.py [-c FILE] -g {tic-tac-toe,backgammon} [-n N] learn  [-m MODEL]
    diee.py ...                                          play   -a AGENT [-m MODEL] --agent-two AGENT [--model-path-two MODEL] -o DIR
    diee.py ...                                          train  [-m MODEL] [-o OUT] [-r RUN [-l LRN [-s SP]]]
    diee.py ...                                          replay -g GAME.json

Same subcommands, short flags and 13 TOML keys as the reference; clap's kebab-case long flags and the
README's snake_case spellings are both accepted.  Models and training data are .npy files (the
reference's .ot libtorch archives need tch).
"""
import argparse
import os
import sys

import numpy as np


def build_parser():
    ap = argparse.ArgumentParser(prog="die-e")
    ap.add_argument("-c", "--config", metavar="FILE", default="./config")                     # main.rs:17-19,89-91
    ap.add_argument("-g", "--game", required=True, choices=["tic-tac-toe", "backgammon"])     # :21-22,80-83
    ap.add_argument("-n", "--n-cpus", "--n_cpus", type=int, default=None)                     # :24-26
    sub = ap.add_subparsers(dest="command", required=True)
    le = sub.add_parser("learn")                                                              # :35-39
    le.add_argument("-m", "--model-path", "--model_path", default=None)
    pl = sub.add_parser("play")                                                               # :40-56
    pl.add_argument("-a", "--agent-one", "--agent_one", default=None)
    pl.add_argument("-m", "--model-path-one", "--model_path_one", default=None)
    pl.add_argument("--agent-two", "--agent_two", default=None)
    pl.add_argument("--model-path-two", "--model_path_two", default=None)
    pl.add_argument("-o", "--output-path", "--output_path", default=None)
    tr = sub.add_parser("train")                                                              # :57-73
    tr.add_argument("-m", "--model-path", "--model_path", default=None)
    tr.add_argument("-o", "--out-path", "--out_path", default=None)
    tr.add_argument("-r", "--run-id", "--run_id", default=None)
    tr.add_argument("-l", "--learn", default=None)
    tr.add_argument("-s", "--self-play", "--self_play", default=None)
    rp = sub.add_parser("replay")                                                             # :74-78
    rp.add_argument("-g", "--game-path", "--game_path", required=True)
    return ap


def training_data_path(game_name, run_id, learn, self_play, root="."):
    """main.rs:175-182"""
    base = os.path.join(root, "data", game_name)
    if run_id is None and learn is None and self_play is None:
        return base
    if run_id is not None and learn is None and self_play is None:
        return os.path.join(base, f"run-{run_id}")
    if run_id is not None and learn is not None and self_play is None:
        return os.path.join(base, f"run-{run_id}", f"lrn-{learn}")
    if run_id is not None and learn is not None and self_play is not None:
        return os.path.join(base, f"run-{run_id}", f"lrn-{learn}", f"sp-{self_play}")
    raise ValueError("the request for the training data is incorrect, run die-e learn --help for more info")


def get_all_paths_rec(d, res):
    """main.rs:218-231: every directory whose name contains "sp" is a data directory"""
    if os.path.isdir(d):
        for name in sorted(os.listdir(d)):
            p = os.path.join(d, name)
            if "sp" in name:
                res.append(p)
            else:
                get_all_paths_rec(p, res)
    return res


def main_tictactoe(args):
    """handle_command::<TicTacToe> (main.rs:112-114,119-216): the same driver on the tic-tac-toe host engine (BASELINE
    configs[0]: CPU path, no GPU): learn, train and play; nothing here touches the HIP engine"""
    from . import GAME_TTT, Engine
    from .alphazero import TICTACTOE, AlphaZero, load_config, mcts_config_from
    from .versus import Agent, Player, play_tictactoe
    from .ot import load_model
    if args.command == "replay":
        sys.exit("replay prints backgammon boards (versus.rs:75-105 over Game<Backgammon> files)")
    conf = load_config(args.config)
    eng = Engine(0, GAME_TTT)
    if args.command == "learn":
        az = AlphaZero.from_config(eng, conf, model_path=args.model_path, game=TICTACTOE, train_device="cpu")
        for row in az.learn_parallel():
            print(row)
    elif args.command == "train":
        data_path = training_data_path("tictactoe", args.run_id, args.learn, args.self_play)
        if not os.path.exists(data_path):
            sys.exit(f"[TRAIN] the specified path {data_path} does not exist!")
        mem = AlphaZero.concat([AlphaZero.load_training_data(p) for p in get_all_paths_rec(data_path, [])])
        print(f"Total memory fragments: {len(mem['outcome'])}")
        az = AlphaZero.from_config(eng, conf, model_path=args.model_path, game=TICTACTOE, train_device="cpu")
        az.train(mem); az.sync_engine()
        out = args.out_path or os.path.join(".", "models", "tictactoe", "trained_model.npy")
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        np.save(out, az.blob)
        print(f"Trained model saved successfully, saved to {out}")
    elif args.command == "play":
        def model(path):
            if path is None:
                return None
            e = Engine(0, GAME_TTT); e.load_weights(load_model(path)); return e
        a1, a2 = Agent.parse(args.agent_one), Agent.parse(args.agent_two)
        print(play_tictactoe(Player(a1, model(args.model_path_one)), Player(a2, model(args.model_path_two)), mcts_config_from(conf),
                             float(conf["temperature"])))
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.game != "backgammon":
        return main_tictactoe(args)
    from . import Engine
    from .alphazero import AlphaZero, load_config, mcts_config_from
    from .versus import Agent, EngineRules, Player, play, print_game, save_game
    from .ot import load_model, save_model_ot
    if args.command == "replay":                                                              # main.rs:208-213
        print_game(args.game_path, wait_user_input=sys.stdin.isatty())
        return 0
    conf = load_config(args.config)
    n_cpus = os.cpu_count() or 1
    if args.n_cpus is not None and args.n_cpus > n_cpus:                                      # main.rs:100-106
        sys.exit(f"Value provided in n_cpus flag ({args.n_cpus}) is larger than total cpus in the device ({n_cpus})!")
    print(f"Number of CPU's to use {args.n_cpus or n_cpus // 2}")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = local_rank
    shares_gpu = False
    if world > 1:
        # under torch.distributed.run: one rank per GPU.  torch picks its device and RCCL comes up BEFORE the engine
        # touches the GPU (torch's HIP runtime has to initialise first; nothing is re-exec'ed afterwards)
        import torch
        import torch.distributed as dist
        ndev = max(torch.cuda.device_count(), 1)
        device = local_rank % ndev
        if torch.cuda.is_available():
            torch.cuda.set_device(device)
        if not dist.is_initialized():
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
        # ranks that share a GPU are recognised by the PCI bus id of their device, not by counting ordinals: a box-wide
        # HIP_VISIBLE_DEVICES=0 gives every rank "its own" device 0 (bench.py does the same)
        from . import device_pci_bus_id
        try:
            mine = device_pci_bus_id(device)
            ids = [None] * world
            dist.all_gather_object(ids, mine)
            shares_gpu = ids.count(mine) > 1           # told to the engine below (diee_set_option "shared_gpu")
        except Exception as ex:                        # (no GPU: the engine below refuses anyway)
            sys.stderr.write(f"[diee] rank {rank}: PCI bus ids not compared ({ex})\n")
            shares_gpu = world > ndev
    eng = Engine(device)
    if shares_gpu:
        from . import load_library
        eng.set_option("shared_gpu", 1)                # no in-launch hand-overs between co-resident grids (cluster / pair tower) ...
        load_library().diee_train_set_bn_coop(0)       # ... nor in the training step's BatchNorm passes
    if args.command == "learn":                                                               # main.rs:121-124
        az = AlphaZero.from_config(eng, conf, model_path=args.model_path, rank=rank, world=world,
                                   train_device=(f"cuda:{device}" if world > 1 else None))
        for row in az.learn_parallel():
            print(row)
    elif args.command == "play":                                                              # main.rs:125-171
        a1, a2 = Agent.parse(args.agent_one), Agent.parse(args.agent_two)
        if args.output_path is None:
            sys.exit("No output path given.")
        if not os.path.isdir(args.output_path):
            sys.exit("Output path is not a directory or does not exist!")

        def model(path):
            if path is None:
                return None
            e = Engine(device)
            e.load_weights(load_model(path))                                                  # .npy blob or die-e's .ot archive
            return e
        m1, m2 = model(args.model_path_one), model(args.model_path_two)
        res = play(Player(a1, m1), Player(a2, m2), mcts_config_from(conf), float(conf["temperature"]),
                   rules=EngineRules(m1 or m2 or eng))
        print(f"{res}\n Saving games...")
        for g in res.games:
            save_game(g, args.output_path)
    elif args.command == "train":                                                             # main.rs:172-207
        print("Starting training process")
        data_path = training_data_path("backgammon", args.run_id, args.learn, args.self_play)
        if not os.path.exists(data_path):
            sys.exit(f"[TRAIN] the specified path {data_path} does not exist!")
        print(f"Loading all data under {data_path}")
        paths = get_all_paths_rec(data_path, [])
        mem = AlphaZero.concat([AlphaZero.load_training_data(p) for p in paths])
        print(f"Total memory fragments: {len(mem['outcome'])}")
        az = AlphaZero.from_config(eng, conf, model_path=args.model_path)
        az.train(mem)
        az.sync_engine()
        out = args.out_path or os.path.join(".", "models", "backgammon", "trained_model.npy")
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        if out.endswith(".ot"):
            save_model_ot(az.blob, out)
        else:
            np.save(out, az.blob)
        print(f"Trained model saved successfully, saved to {out}")
    return 0
