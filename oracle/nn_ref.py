"""PyTorch fp32 restatement of the reference's ResNet (src/alphazero/nnet.rs:24-34,57-133) built from
the flat weight blob of include/diee.h.  TEST INFRASTRUCTURE (part of the oracle): the fp32 reference the bf16 MFMA
kernels are compared against (the reference pins no network outputs: SURVEY section 4)."""
import numpy as np
import torch
import torch.nn.functional as Fn

F, BLOCKS, A = 256, 19, 1352


class BlobReader:
    def __init__(self, blob):
        self.b = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)); self.o = 0

    def take(self, *shape):
        n = int(np.prod(shape)); t = self.b[self.o:self.o + n].reshape(*shape); self.o += n
        return t

    def conv(self, cout, cin):
        return self.take(cout, cin, 3, 3), self.take(cout)

    def bn(self, c):
        return self.take(c), self.take(c), self.take(c), self.take(c)   # gamma, beta, mean, var


def parse(blob, filters=F, blocks=BLOCKS, actions=A, cin=6, hw=24):
    """defaults: backgammon (backgammon_logic.rs:74-78); tic-tac-toe: filters=64, blocks=4, actions=9, cin=3, hw=9 (tictactoe/mod.rs:20-24)"""
    r = BlobReader(blob)
    net = {"init": (r.conv(filters, cin), r.bn(filters)), "blocks": []}
    for _ in range(blocks):
        c1 = r.conv(filters, filters); c2 = r.conv(filters, filters); b1 = r.bn(filters); b2 = r.bn(filters)
        net["blocks"].append((c1, b1, c2, b2))
    net["p"] = (r.conv(32, filters), r.bn(32), r.take(actions, 32 * hw), r.take(actions))
    net["v"] = (r.conv(3, filters), r.bn(3), r.take(1, 3 * hw), r.take(1))
    assert r.o == len(r.b), (r.o, len(r.b))
    return net


def conv_bn(x, conv, bn):
    (w, b), (g, be, m, v) = conv, bn
    return Fn.batch_norm(Fn.conv2d(x, w, b, padding=1), m, v, g, be, training=False, eps=1e-5)


@torch.no_grad()
def forward_t(net, planes, dtype=torch.float32, shape=(6, 4, 6), device=None):
    """planes [n,144] (c*24+p) -> (softmax policy [n,1352], tanh value [n]); nnet.rs:120-133, train=false
    (shape = the game's input planes: (6, 4, 6) backgammon, (3, 3, 3) tic-tac-toe; device: where PyTorch evaluates it --
    the GPU tests run the fp32 restatement at full batch sizes on "cuda", still fp32 and still not the engine's kernels)"""
    x = torch.as_tensor(planes, dtype=torch.float32).reshape(-1, *shape).to(device=device, dtype=dtype)
    cast = lambda t: tuple(cast(u) for u in t) if isinstance(t, tuple) else t.to(device=device, dtype=dtype)
    net = {k: cast(v) if k != "blocks" else [cast(b) for b in v] for k, v in net.items()}
    x = torch.relu(conv_bn(x, *net["init"]))
    for c1, b1, c2, b2 in net["blocks"]:                       # ResBlock::forward_t, nnet.rs:24-34
        h = torch.relu(conv_bn(x, c1, b1))
        x = torch.relu(conv_bn(h, c2, b2) + x)
    pc, pbn, pw, pb = net["p"]
    logits = torch.relu(conv_bn(x, pc, pbn)).flatten(1) @ pw.T + pb
    vc, vbn, vw, vb = net["v"]
    value = torch.tanh(torch.relu(conv_bn(x, vc, vbn)).flatten(1) @ vw.T + vb)
    return torch.softmax(logits, 1).float().cpu().numpy(), value[:, 0].float().cpu().numpy(), logits.float().cpu().numpy()
