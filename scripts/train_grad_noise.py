"""How far are mixed-precision gradients of this 40-layer random-init ResNet from fp32 autograd?  (a) PyTorch bf16
autocast (MIOpen), (b) the engine's token path -- same metric as tests/test_train_gpu.py."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
ops = importlib.import_module("die-e_amd.train_ops")
import torch, torch.nn.functional as Fn
import diee_amd
torch.manual_seed(1)
blob = diee_amd.random_weights(0)
B = 64
x = torch.randn(B, 6, 4, 6, device="cuda").round().clamp(-3, 3); ps = torch.softmax(torch.randn(B, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(B, 1, device="cuda"))
def grads(mode):
    net = az.make_resnet().load_blob(blob).cuda().train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "amp")):
        lg, v = ops.forward_train_tokens(net, x) if mode == "engine" else net(x)
        loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters()}
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
l0, g0 = grads("fp32"); l0b, g0b = grads("fp32")
names = [n for n in g0 if not (n.endswith("conv.bias") or n.endswith("conv1.bias") or n.endswith("conv2.bias"))]
for mode in ("fp32-again", "amp", "engine"):
    l, g = (l0b, g0b) if mode == "fp32-again" else grads(mode)
    e = sorted(rel(g[n], g0[n]) for n in names)
    cos = min(float(Fn.cosine_similarity(g[n].flatten().double(), g0[n].flatten().double(), dim=0)) for n in names)
    print(f"{mode:10s} loss {l:.5f} (fp32 {l0:.5f})  grad rel L2: median {e[len(e)//2]:.3e}  p90 {e[int(len(e)*0.9)]:.3e}  max {e[-1]:.3e}  min cosine {cos:.5f}", flush=True)
