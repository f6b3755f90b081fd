"""die-e_amd -- host-side mirror (Python/ctypes) of the reference interface for the self-play hot
path, over the C ABI in include/diee.h (libdiee.so: hand-written HIP for gfx950).

The method names follow the reference's own API for this path so parity tests read like the
reference's tests:

    LearnableGame trait (src/base.rs:8-51)        -> Engine.get_valid_moves / encode / decode /
                                                     apply_move / as_tensor        (batched)
    ResNet::forward_t (src/alphazero/nnet.rs:120) -> Engine.forward_t
    alpha_mcts_parallel + get_prob_tensor_parallel (src/mcts/alpha_mcts.rs:91, utils.rs:42)
                                                  -> Engine.alpha_mcts_parallel
    AlphaZero::self_play_parallel (src/alphazero/alpha_parallel.rs:101) -> Engine.self_play_parallel

There is no CPU fallback: the library must load and a GPU must be present, otherwise the
constructor raises.  Import with importlib.import_module("die-e_amd") (the directory name is
not a Python identifier) or through the `diee_amd` shim at the repo root.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIEE_LIB") or os.path.join(_HERE, "libdiee.so")      # DIEE_LIB: an A/B or diagnostic build (scripts/)

BG_ACTIONS = 1352
BG_PLANES = 144
NO_MOVE = -2
GAME_TTT, GAME_BACKGAMMON = 0, 1
FLAG_REF_QUIRKS = 1
FLAG_INVARIANT_NN = 2
BAND_BOUNDS = (16, 32, 64, 128, 256, 512, 928, 1024)     # DIEE_BANDS: upper bounds of Stats.band_* (the ninth band is everything above)

OK, ERR_ARG, ERR_HIP, ERR_NO_WEIGHTS, ERR_CAPACITY, ERR_UNSUPPORTED = range(6)

BG_STATE = np.dtype([("pts", "i1", 24), ("bar", "u1", 2), ("off", "u1", 2), ("roll", "u1", 2),
                     ("player", "i1"), ("second", "u1")])
assert BG_STATE.itemsize == 32
# tic-tac-toe (BASELINE configs[0]): diee_ttt_state; an Engine(game_id=GAME_TTT) runs on the host (csrc/ttt_host.cpp)
TTT_ACTIONS, TTT_PLANES = 9, 27
TTT_STATE = np.dtype([("board", "i1", 9), ("player", "i1"), ("pad", "u1", 22)])
assert TTT_STATE.itemsize == 32

# every symbol include/diee.h declares (checked by the CPU test-suite against the built library); the development probes
# of include/diee_dev.h are listed in DEV_EXPORTS
EXPORTS = [
    "diee_create", "diee_destroy", "diee_last_error", "diee_version", "diee_set_option", "diee_get_option", "diee_device_pci_bus_id", "diee_weights_count",
    "diee_random_weights", "diee_load_weights", "diee_nn_forward", "diee_mcts_batch", "diee_self_play",
    "diee_self_play_multi", "diee_set_invariant_nn", "diee_train_pack_conv3x3", "diee_train_pack_conv3x3_multi",
    "diee_train_conv3x3", "diee_train_im2col3x3",
    "diee_train_scratch_floats", "diee_train_bn_relu_fwd", "diee_train_bn_relu_bwd", "diee_train_colsum",
    "diee_train_set_bn_coop", "diee_train_bn_coop_timeouts",
    "diee_train_wgrad_scratch_floats", "diee_train_wgrad3x3",
    "diee_free_fragments", "diee_bg_legal_moves", "diee_bg_encode", "diee_bg_decode", "diee_bg_apply",
    "diee_bg_planes", "diee_det_pow",
    "diee_ttt_valid_moves", "diee_ttt_apply_move", "diee_ttt_check_winner", "diee_ttt_planes",
]
DEV_EXPORTS = ["diee_probe_f32", "diee_probe_dice", "diee_dev_conv_bench", "diee_dev_rules_bench", "diee_dev_wave_selftest",
               "diee_dev_last_dispatch", "diee_dev_dispatch_bands"]


class DieeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"diee status {status}: {msg}")
        self.status = status


class MctsConfig(C.Structure):
    """MctsConfig, src/lib.rs:33-40"""
    _fields_ = [("iterations", C.c_uint32), ("c", C.c_float), ("round_limit", C.c_uint32),
                ("dir_alpha", C.c_float), ("dir_eps", C.c_float)]

    @classmethod
    def default(cls, iterations=100):
        # config-example.toml:11-15
        return cls(iterations=iterations, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)


class Stats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("games", "plies", "move_steps", "nn_evals", "expansions", "children",
                                           "terminal_hits", "depth_sum", "selections", "illegal_decodes",
                                           "max_children", "fragments")]
                + [("seconds", C.c_double), ("nn_seconds", C.c_double), ("conv_seconds", C.c_double),
                   ("conv_launches", C.c_uint64), ("conv_flops", C.c_double),
                   ("tower_seconds", C.c_double), ("tower_launches", C.c_uint64), ("tower_flops", C.c_double),
                   ("cluster_seconds", C.c_double), ("cluster_launches", C.c_uint64), ("cluster_flops", C.c_double),
                   ("nn_rows", C.c_uint64),
                   ("full_seconds", C.c_double), ("full_launches", C.c_uint64), ("full_flops", C.c_double),
                   ("deliver_seconds", C.c_double), ("deliver_bytes", C.c_uint64),
                   ("band_seconds", C.c_double * 9), ("band_launches", C.c_uint64 * 9), ("band_flops", C.c_double * 9),
                   ("tail_iterations", C.c_uint64), ("tail_launches", C.c_uint64), ("tail_spec_rows", C.c_uint64),
                   ("band_flops_demanded", C.c_double * 9)])

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.startswith("band_") else getattr(self, n)) for n, _ in self._fields_}


class Fragments(C.Structure):
    _fields_ = [("n", C.c_uint32), ("outcome", C.POINTER(C.c_int8)), ("ps", C.POINTER(C.c_float)),
                ("state", C.POINTER(C.c_float)), ("game", C.POINTER(C.c_uint32))]


class DevLaunch(C.Structure):
    """diee_dev_launch (include/diee_dev.h)"""
    _fields_ = [("family", C.c_int), ("geometry", C.c_int), ("boards", C.c_int), ("kernel", C.c_char * 120)]


class DevBand(C.Structure):
    """diee_dev_band (include/diee_dev.h)"""
    _fields_ = [("boards_min", C.c_int), ("boards_max", C.c_int), ("family", C.c_int), ("geometry", C.c_int), ("kernel", C.c_char * 120)]


class Batch(C.Structure):
    """diee_batch: one self_play_parallel call of a pipelined diee_self_play_multi"""
    _fields_ = [("n_games", C.c_uint32), ("first_game_id", C.c_uint32), ("seed", C.c_uint64)]


_lib = None
_TORCH_LOADED_FIRST = False     # PyTorch bundles its own ROCm runtime libraries.  Whichever side is loaded first
                                # provides them to the whole process (same sonames): if torch came first both torch.cuda
                                # and the engine work; if libdiee.so came first the engine works and torch.cuda must
                                # not be touched.  Processes that need both import torch first (alphazero.py, bench.py).


def load_library(path=None):
    """dlopen libdiee.so and declare the prototypes.  Does not touch the GPU."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    global _TORCH_LOADED_FIRST
    import sys
    _TORCH_LOADED_FIRST = "torch" in sys.modules
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} is missing: build it with `python die-e_amd/build.py` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(p)
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.diee_version.restype = C.c_char_p
    L.diee_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]; L.diee_create.restype = C.c_int
    L.diee_destroy.argtypes = [vp]; L.diee_destroy.restype = None
    L.diee_last_error.argtypes = [vp]; L.diee_last_error.restype = C.c_char_p
    L.diee_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]; L.diee_device_pci_bus_id.restype = C.c_int
    L.diee_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]; L.diee_set_option.restype = C.c_int
    L.diee_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]; L.diee_get_option.restype = C.c_int
    L.diee_weights_count.argtypes = [C.c_int]; L.diee_weights_count.restype = C.c_size_t
    L.diee_random_weights.argtypes = [C.c_int, u64, vp, C.c_size_t]; L.diee_random_weights.restype = C.c_int
    L.diee_load_weights.argtypes = [vp, vp, C.c_size_t]; L.diee_load_weights.restype = C.c_int
    L.diee_nn_forward.argtypes = [vp, vp, u32, vp, vp]; L.diee_nn_forward.restype = C.c_int
    L.diee_mcts_batch.argtypes = [vp, vp, u32, vp, u64, u32, vp, vp, u32, vp, vp, vp, vp]
    L.diee_mcts_batch.restype = C.c_int
    L.diee_self_play.argtypes = [vp, u32, u32, vp, f32, u64, u32, u32, vp, vp]; L.diee_self_play.restype = C.c_int
    L.diee_self_play_multi.argtypes = [vp, vp, u32, vp, f32, u32, u32, vp, vp]; L.diee_self_play_multi.restype = C.c_int
    L.diee_set_invariant_nn.argtypes = [vp, C.c_int]; L.diee_set_invariant_nn.restype = C.c_int
    L.diee_train_pack_conv3x3.argtypes = [vp, vp, C.c_int, vp]; L.diee_train_pack_conv3x3.restype = C.c_int
    L.diee_train_pack_conv3x3_multi.argtypes = [C.POINTER(vp), C.c_int, vp, vp]; L.diee_train_pack_conv3x3_multi.restype = C.c_int
    L.diee_train_conv3x3.argtypes = [vp, vp, vp, vp, C.c_int, vp]; L.diee_train_conv3x3.restype = C.c_int
    L.diee_train_im2col3x3.argtypes = [vp, vp, C.c_int, vp]; L.diee_train_im2col3x3.restype = C.c_int
    L.diee_train_scratch_floats.argtypes = [C.c_int]; L.diee_train_scratch_floats.restype = C.c_size_t
    L.diee_train_bn_relu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, C.c_int, vp, vp]; L.diee_train_bn_relu_fwd.restype = C.c_int
    L.diee_train_bn_relu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]; L.diee_train_bn_relu_bwd.restype = C.c_int
    L.diee_train_colsum.argtypes = [vp, vp, C.c_int, vp, vp]; L.diee_train_colsum.restype = C.c_int
    L.diee_train_set_bn_coop.argtypes = [C.c_int]; L.diee_train_set_bn_coop.restype = C.c_int
    L.diee_train_bn_coop_timeouts.argtypes = [C.c_int]; L.diee_train_bn_coop_timeouts.restype = C.c_int
    L.diee_train_wgrad_scratch_floats.argtypes = []; L.diee_train_wgrad_scratch_floats.restype = C.c_size_t
    L.diee_train_wgrad3x3.argtypes = [vp, vp, vp, C.c_int, vp, vp]; L.diee_train_wgrad3x3.restype = C.c_int
    L.diee_free_fragments.argtypes = [vp]; L.diee_free_fragments.restype = None
    L.diee_bg_legal_moves.argtypes = [vp, vp, u32, vp, u32, vp]; L.diee_bg_legal_moves.restype = C.c_int
    L.diee_bg_encode.argtypes = [vp, vp, vp, u32, vp]; L.diee_bg_encode.restype = C.c_int
    L.diee_bg_decode.argtypes = [vp, vp, vp, u32, vp]; L.diee_bg_decode.restype = C.c_int
    L.diee_bg_apply.argtypes = [vp, vp, vp, vp, u32]; L.diee_bg_apply.restype = C.c_int
    L.diee_bg_planes.argtypes = [vp, vp, u32, vp]; L.diee_bg_planes.restype = C.c_int
    L.diee_det_pow.argtypes = [vp, vp, vp, u32, vp]; L.diee_det_pow.restype = C.c_int
    L.diee_ttt_valid_moves.argtypes = [vp, vp]; L.diee_ttt_valid_moves.restype = u32
    L.diee_ttt_apply_move.argtypes = [vp, C.c_uint8]; L.diee_ttt_apply_move.restype = None
    L.diee_ttt_check_winner.argtypes = [vp, C.POINTER(C.c_int)]; L.diee_ttt_check_winner.restype = C.c_int
    L.diee_ttt_planes.argtypes = [vp, vp]; L.diee_ttt_planes.restype = None
    L.diee_probe_f32.argtypes = [vp, vp, vp, u32, vp, vp, vp]; L.diee_probe_f32.restype = C.c_int
    L.diee_probe_dice.argtypes = [vp, u64, vp, u32, vp, vp]; L.diee_probe_dice.restype = C.c_int
    L.diee_dev_conv_bench.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]; L.diee_dev_conv_bench.restype = C.c_int
    L.diee_dev_rules_bench.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, vp]; L.diee_dev_rules_bench.restype = C.c_int
    L.diee_dev_wave_selftest.argtypes = [vp, C.c_uint32, vp]; L.diee_dev_wave_selftest.restype = C.c_int
    L.diee_dev_last_dispatch.argtypes = [vp, vp, u32, vp]; L.diee_dev_last_dispatch.restype = C.c_int
    L.diee_dev_dispatch_bands.argtypes = [vp, C.c_int, vp, u32, vp]; L.diee_dev_dispatch_bands.restype = C.c_int
    if path is None:
        _lib = L
    return L


def device_pci_bus_id(device=0):
    """PCI address of the GPU libdiee.so's HIP runtime calls `device` (diee_device_pci_bus_id)"""
    buf = C.create_string_buffer(64)
    st = load_library().diee_device_pci_bus_id(device, buf, 64)
    if st != OK:
        raise DieeError(st, f"diee_device_pci_bus_id({device})")
    return buf.value.decode()


def same_gpu_as_torch(device):
    """(ok, engine's PCI address, torch's) for ordinal `device`: the two HIP runtimes of a PyTorch process must mean the same GPU by it"""
    import torch
    pr = torch.cuda.get_device_properties(device)
    ours = device_pci_bus_id(device).lower()
    if not hasattr(pr, "pci_bus_id"):
        return True, ours, None
    theirs = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
    # compared as numbers (domain, bus, device): a formatting difference between the two runtimes must not read as another GPU, and an
    # address this code cannot parse is no evidence of a mismatch
    import re
    m = re.fullmatch(r"([0-9a-f]+):([0-9a-f]+):([0-9a-f]+)\.([0-9a-f]+)", ours)
    if not m:
        return True, ours, theirs
    mine = tuple(int(x, 16) for x in m.groups()[:3])
    return mine == (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0)), ours, theirs


def weights_count(game_id=GAME_BACKGAMMON):
    return int(load_library().diee_weights_count(game_id))


def random_weights(seed=0, game_id=GAME_BACKGAMMON):
    """tch-default random init of the ResNet (nnet.rs:57-107) as the flat fp32 blob of include/diee.h"""
    L = load_library()
    n = int(L.diee_weights_count(game_id))
    blob = np.zeros(n, dtype=np.float32)
    st = L.diee_random_weights(game_id, seed, blob.ctypes.data, n)
    if st != OK:
        raise DieeError(st, "diee_random_weights")
    return blob


def _states(a):
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 32 or (a.dtype == np.uint8 and a.shape[-1] == 32), a.dtype
    return a


class Engine:
    """one engine per GPU (diee_ctx)"""

    def __init__(self, device=0, game_id=GAME_BACKGAMMON):
        import sys
        self._L = load_library()
        self.game_id = game_id
        self.n_actions, self.n_planes = (TTT_ACTIONS, TTT_PLANES) if game_id == GAME_TTT else (BG_ACTIONS, BG_PLANES)
        if _TORCH_LOADED_FIRST and game_id != GAME_TTT:   # torch's runtime serves the process: let it initialise before the engine does
            sys.modules["torch"].cuda.is_available()
        h = C.c_void_p()
        st = self._L.diee_create(device, game_id, C.byref(h))
        if st != OK:
            raise DieeError(st, "diee_create failed (no HIP device / unsupported game): the HIP path is the only path")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.diee_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st != OK:
            raise DieeError(st, self._L.diee_last_error(self._h).decode())

    # ---- development probes: which tower kernel ran (include/diee_dev.h) ----
    def last_dispatch(self):
        """[(kernel name, boards)] of the tower launches of the last network evaluation"""
        buf = (DevLaunch * 16)(); n = C.c_uint32(0)
        self._chk(self._L.diee_dev_last_dispatch(self._h, buf, 16, C.byref(n)))
        return [(buf[i].kernel.decode(), buf[i].boards) for i in range(n.value)]

    def dispatch_bands(self, upto=1024):
        """[(boards_min, boards_max, kernel name)]: the tower kernel a plain evaluation of that many boards runs on this ctx"""
        buf = (DevBand * 64)(); n = C.c_uint32(0)
        self._chk(self._L.diee_dev_dispatch_bands(self._h, upto, buf, 64, C.byref(n)))
        return [(buf[i].boards_min, buf[i].boards_max, buf[i].kernel.decode()) for i in range(n.value)]

    # ---- options of the ctx (diee_set_option: what used to be DIEE_* environment switches) ----
    def set_option(self, key, value):
        """e.g. set_option("shared_gpu", 1) when several processes compute on this GPU; include/diee.h lists the keys"""
        self._chk(self._L.diee_set_option(self._h, str(key).encode(), str(value).encode()))
        return self

    def set_options(self, **kw):
        for k, v in kw.items():
            self.set_option(k, v)
        return self

    def get_option(self, key):
        buf = C.create_string_buffer(256)
        self._chk(self._L.diee_get_option(self._h, str(key).encode(), buf, 256))
        return buf.value.decode()

    # ---- LearnableGame trait, batched -------------------------------------------------------
    def get_valid_moves(self, states, cap=256):
        """get_valid_moves (backgammon_logic.rs:403-414) -> (plays int8 [n,cap,4], counts uint32 [n])"""
        s = _states(states); n = len(s)
        plays = np.full((n, cap, 4), NO_MOVE, dtype=np.int8)
        counts = np.zeros(n, dtype=np.uint32)
        self._chk(self._L.diee_bg_legal_moves(self._h, s.ctypes.data, n, plays.ctypes.data, cap, counts.ctypes.data))
        # slots past the count are unspecified on the device side: blank them
        idx = np.arange(cap)[None, :] >= counts[:, None]
        plays[idx] = NO_MOVE
        return plays, counts

    def encode(self, states, plays):
        s = _states(states); p = np.ascontiguousarray(plays, dtype=np.int8); n = len(s)
        assert p.shape == (n, 4)
        codes = np.zeros(n, dtype=np.uint32)
        self._chk(self._L.diee_bg_encode(self._h, s.ctypes.data, p.ctypes.data, n, codes.ctypes.data))
        return codes

    def decode(self, states, codes):
        s = _states(states); c = np.ascontiguousarray(codes, dtype=np.uint32); n = len(s)
        plays = np.zeros((n, 4), dtype=np.int8)
        self._chk(self._L.diee_bg_decode(self._h, s.ctypes.data, c.ctypes.data, n, plays.ctypes.data))
        return plays

    def apply_move(self, states, plays, dice):
        s = _states(states).copy(); p = np.ascontiguousarray(plays, dtype=np.int8)
        d = np.ascontiguousarray(dice, dtype=np.uint8); n = len(s)
        assert p.shape == (n, 4) and d.shape == (n, 2)
        self._chk(self._L.diee_bg_apply(self._h, s.ctypes.data, p.ctypes.data, d.ctypes.data, n))
        return s

    def as_tensor(self, states):
        s = _states(states); n = len(s)
        out = np.zeros((n, BG_PLANES), dtype=np.float32)
        self._chk(self._L.diee_bg_planes(self._h, s.ctypes.data, n, out.ctypes.data))
        return out

    def det_pow(self, x, y):
        """Tensor::pow_ of the drivers (alpha_parallel.rs:165, versus.rs:283) with the engine's deterministic powf"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(np.broadcast_to(np.asarray(y, dtype=np.float32), x.shape))
        out = np.zeros_like(x)
        self._chk(self._L.diee_det_pow(self._h, x.ctypes.data, y.ctypes.data, x.size, out.ctypes.data))
        return out

    def probe_f32(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
        sq, dv, pw = (np.zeros_like(a) for _ in range(3))
        self._chk(self._L.diee_probe_f32(self._h, a.ctypes.data, b.ctypes.data, len(a), sq.ctypes.data,
                                         dv.ctypes.data, pw.ctypes.data))
        return sq, dv, pw

    def probe_dice(self, seed, ctr):
        ctr = np.ascontiguousarray(ctr, dtype=np.uint32); n = len(ctr)
        dice = np.zeros((n, 2), dtype=np.uint8); uni = np.zeros(n, dtype=np.float64)
        self._chk(self._L.diee_probe_dice(self._h, seed, ctr.ctypes.data, n, dice.ctypes.data, uni.ctypes.data))
        return dice, uni

    # ---- ResNet ------------------------------------------------------------------------------
    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._chk(self._L.diee_load_weights(self._h, blob.ctypes.data, blob.size))

    def forward_t(self, states):
        """ResNet::forward_t (nnet.rs:120-133), eval mode -> (softmax policy [n,1352], tanh value [n])"""
        s = _states(states); n = len(s)
        pol = np.zeros((n, self.n_actions), dtype=np.float32); val = np.zeros(n, dtype=np.float32)
        self._chk(self._L.diee_nn_forward(self._h, s.ctypes.data, n, pol.ctypes.data, val.ctypes.data))
        return pol, val

    def conv_bench(self, G, variant=0, reps=50):
        a, b, f = C.c_float(0), C.c_float(0), C.c_float(0)
        self._chk(self._L.diee_dev_conv_bench(self._h, G, variant, reps, C.byref(a), C.byref(b), C.byref(f)))
        return a.value, b.value, f.value

    def rules_bench(self, states, reps=20):
        """device time (us) of one get_valid_moves launch over resident states, mean plays per state"""
        s = _states(states)
        us, k = C.c_float(0), C.c_float(0)
        self._chk(self._L.diee_dev_rules_bench(self._h, s.ctypes.data, len(s), reps, C.byref(us), C.byref(k)))
        return us.value, k.value

    def wave_selftest(self, salt=0):
        """csrc/wave_ops.h against the shuffles it replaces: number of disagreeing lanes x cases (0 = correct)"""
        bad = C.c_uint32(0)
        self._chk(self._L.diee_dev_wave_selftest(self._h, salt, C.byref(bad)))
        return int(bad.value)

    # ---- search ------------------------------------------------------------------------------
    def set_invariant_nn(self, on=True):
        """batch-size independent network arithmetic (DIEE_FLAG_INVARIANT_NN) for every later call"""
        self._chk(self._L.diee_set_invariant_nn(self._h, 1 if on else 0))

    def alpha_mcts_parallel(self, states, cfg, seed=0, step=0, game_ids=None, rounds=None, ref_quirks=True):
        """alpha_mcts_parallel + get_prob_tensor_parallel -> dict(probs [n,1352], n_children, root_visits, stats)"""
        s = _states(states); n = len(s)
        probs = np.zeros((n, self.n_actions), dtype=np.float32)
        nch = np.zeros(n, dtype=np.uint32); rv = np.zeros(n, dtype=np.float32)
        gi = None if game_ids is None else np.ascontiguousarray(game_ids, dtype=np.uint32)
        rd = None if rounds is None else np.ascontiguousarray(rounds, dtype=np.uint32)
        st = Stats()
        self._chk(self._L.diee_mcts_batch(self._h, s.ctypes.data, n, C.byref(cfg), seed, step,
                                          None if gi is None else gi.ctypes.data,
                                          None if rd is None else rd.ctypes.data,
                                          FLAG_REF_QUIRKS if ref_quirks else 0,
                                          probs.ctypes.data, nch.ctypes.data, rv.ctypes.data, C.byref(st)))
        return {"probs": probs, "n_children": nch, "root_visits": rv, "stats": st.as_dict()}

    def _take_fragments(self, fr, out, copy=True):
        """diee_fragments -> numpy.  copy=True: arrays of their own, the engine's blocks go back at once.  copy=False: views of
        the engine-owned (page-locked) arrays, valid until out["free"]() -- what a host that binds the C ABI works with."""
        n = fr.n
        A, P = self.n_actions, self.n_planes
        view = {"outcome": np.ctypeslib.as_array(fr.outcome, shape=(n,)) if n else np.zeros(0, np.int8),
                "ps": np.ctypeslib.as_array(fr.ps, shape=(n, A)) if n else np.zeros((0, A), np.float32),
                "state": np.ctypeslib.as_array(fr.state, shape=(n, P)) if n else np.zeros((0, P), np.float32),
                "game": np.ctypeslib.as_array(fr.game, shape=(n,)) if n else np.zeros(0, np.uint32)}
        if copy:
            out.update({k: v.copy() for k, v in view.items()})
            self._L.diee_free_fragments(C.byref(fr))
            return
        out.update(view)
        L = self._L

        def free(fr=fr, done=[False]):
            if not done[0]:
                done[0] = True
                for k in view:
                    out.pop(k, None)
                L.diee_free_fragments(C.byref(fr))
        out["free"] = free

    def self_play_parallel(self, n_games, cfg, temperature=1.25, seed=0xD1EE0001, ref_quirks=True,
                           first_game_id=0, max_steps=0, fetch=True, invariant_nn=False, copy=True):
        """AlphaZero::self_play_parallel -> dict(outcome, ps, state, game, stats).  fetch=False: results stay in HBM (timing
        aid); copy=False: see _take_fragments"""
        fr = Fragments(); st = Stats()
        flags = (FLAG_REF_QUIRKS if ref_quirks else 0) | (FLAG_INVARIANT_NN if invariant_nn else 0)
        self._chk(self._L.diee_self_play(self._h, n_games, first_game_id, C.byref(cfg), temperature, seed,
                                         flags, max_steps, C.byref(fr) if fetch else None, C.byref(st)))
        out = {"stats": st.as_dict()}
        if fetch:
            self._take_fragments(fr, out, copy)
        return out

    def self_play_multi(self, batches, cfg, temperature=1.25, ref_quirks=True, max_steps=0, fetch=True,
                        invariant_nn=False, copy=True):
        """K self_play_parallel calls played side by side (diee_self_play_multi): batches = [(n_games,
        first_game_id, seed), ...] -> list of per-batch dicts like self_play_parallel's"""
        K = len(batches)
        bt = (Batch * K)(*[Batch(int(n), int(f), int(s)) for n, f, s in batches])
        frs = (Fragments * K)(); sts = (Stats * K)()
        flags = (FLAG_REF_QUIRKS if ref_quirks else 0) | (FLAG_INVARIANT_NN if invariant_nn else 0)
        self._chk(self._L.diee_self_play_multi(self._h, C.byref(bt), K, C.byref(cfg), temperature, flags, max_steps,
                                               C.byref(frs) if fetch else None, C.byref(sts)))
        outs = []
        for k in range(K):
            out = {"stats": sts[k].as_dict()}
            if fetch:
                self._take_fragments(frs[k], out, copy)
                if not copy:
                    out["_keep"] = frs            # (the array of diee_fragments the free closures point into)
            outs.append(out)
        return outs


# ---- LearnableGame for TicTacToe (src/tictactoe/mod.rs), host functions of the C ABI -------------------------------------
def ttt_new(n=None):
    """TicTacToe::new (mod.rs:28-30): empty board, player -1 to move"""
    s = np.zeros(() if n is None else n, dtype=TTT_STATE)
    s["player"] = -1
    return s


def ttt_valid_moves(state):
    mv = np.zeros(9, dtype=np.uint8)
    k = load_library().diee_ttt_valid_moves(np.ascontiguousarray(state).ctypes.data, mv.ctypes.data)
    return [int(x) for x in mv[:k]]


def ttt_apply_move(state, move):
    s = np.array(state, dtype=TTT_STATE, copy=True)
    load_library().diee_ttt_apply_move(s.ctypes.data, int(move))
    return s


def ttt_check_winner(state):
    """Some(winner) -> -1 / 0 (draw) / 1, None -> None"""
    w = C.c_int(0)
    over = load_library().diee_ttt_check_winner(np.ascontiguousarray(state).ctypes.data, C.byref(w))
    return int(w.value) if over else None


def ttt_planes(state):
    out = np.zeros(TTT_PLANES, dtype=np.float32)
    load_library().diee_ttt_planes(np.ascontiguousarray(state).ctypes.data, out.ctypes.data)
    return out
