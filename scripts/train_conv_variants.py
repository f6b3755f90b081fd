"""Training-step variants for the 4x6-board ResNet: MIOpen convs vs 3x3 conv as unfold/gather + GEMM (hipBLASLt)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
import numpy as np, torch, torch.nn.functional as Fn
from torch import nn
import diee_amd
blob = diee_amd.random_weights(0)
B = 256
x = torch.randn(B, 6, 4, 6, device="cuda"); ps = torch.softmax(torch.randn(B, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(B, 1, device="cuda"))

def conv_unfold(x, conv):
    Bn = x.shape[0]
    cols = Fn.unfold(x, 3, padding=1)                                   # [B, C*9, 24]
    w = conv.weight.view(conv.weight.shape[0], -1)                       # [N, C*9]
    out = torch.matmul(cols.transpose(1, 2).reshape(Bn * 24, -1), w.t())   # [B*24, N]
    return (out + conv.bias).view(Bn, 24, -1).transpose(1, 2).reshape(Bn, -1, 4, 6)

# token layout [B*24, C]: 9 shifted row gathers + one GEMM
def make_idx(Bn, device):
    idx = torch.full((Bn * 24, 9), Bn * 24, dtype=torch.long)
    for p in range(24):
        y, xx = divmod(p, 6)
        for t in range(9):
            dy, dx = t // 3 - 1, t % 3 - 1
            if 0 <= y + dy < 4 and 0 <= xx + dx < 6:
                idx[torch.arange(Bn) * 24 + p, t] = torch.arange(Bn) * 24 + p + 6 * dy + dx
    return idx.to(device)
IDX = make_idx(B, "cuda")
def conv_tok(xt, conv):                                                  # xt [B*24, C]
    C = xt.shape[1]
    xp = torch.cat([xt, xt.new_zeros(1, C)], 0)
    cols = xp[IDX.view(-1)].view(-1, 9 * C)                              # [B*24, 9*C], k = t*C + c
    w = conv.weight.permute(2, 3, 1, 0).reshape(9 * C, -1)               # [(ky,kx,c), N]
    return cols @ w + conv.bias

def run(name, mode, amp, steps=20):
    torch.backends.cudnn.benchmark = True
    net = az.make_resnet().load_blob(blob).cuda(); net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    def bn2(bn, t):  # BatchNorm2d parameters applied to token layout
        return Fn.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    def fwd(xx):
        if mode == "conv":
            return net(xx)
        if mode == "unfold":
            h = torch.relu(net.init_bn(conv_unfold(xx, net.init_conv)))
            for b in net.blocks:
                g = torch.relu(b.bn1(conv_unfold(h, b.conv1)))
                h = torch.relu(b.bn2(conv_unfold(g, b.conv2)) + h)
            lg = net.p_fc(torch.relu(net.p_bn(conv_unfold(h, net.p_conv))).flatten(1))
            v = torch.tanh(net.v_fc(torch.relu(net.v_bn(conv_unfold(h, net.v_conv))).flatten(1)))
            return lg, v
        xt = xx.permute(0, 2, 3, 1).reshape(-1, 6)
        h = torch.relu(bn2(net.init_bn, conv_tok(xt, net.init_conv)))
        for b in net.blocks:
            g = torch.relu(bn2(b.bn1, conv_tok(h, b.conv1)))
            h = torch.relu(bn2(b.bn2, conv_tok(g, b.conv2)) + h)
        hp = torch.relu(bn2(net.p_bn, conv_tok(h, net.p_conv))).view(-1, 24, 32).transpose(1, 2).reshape(-1, 768)
        hv = torch.relu(bn2(net.v_bn, conv_tok(h, net.v_conv))).view(-1, 24, 3).transpose(1, 2).reshape(-1, 72)
        return net.p_fc(hp), torch.tanh(net.v_fc(hv))
    for i in range(steps + 4):
        if i == 4:
            torch.cuda.synchronize(); t = time.time()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            lg, v = fwd(x)
            loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"{name:44s} {(time.time() - t) / steps * 1e3:7.2f} ms/step  loss {float(loss):.4f}", flush=True)

run("nn.Conv2d fp32", "conv", False)
run("nn.Conv2d bf16 autocast", "conv", True)
run("unfold + GEMM fp32", "unfold", False)
run("unfold + GEMM bf16 autocast", "unfold", True)
run("token gather + GEMM fp32", "tok", False)
run("token gather + GEMM bf16 autocast", "tok", True)
