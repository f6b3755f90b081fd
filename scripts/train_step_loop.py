"""The learn loop's training step (alphazero.py: fp32, NCHW, Adam) in a loop, for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats -d /tmp/tp --output-format csv -- python3 scripts/train_step_loop.py [variant]
variant: base | graph (CUDA-graph replay of the whole step, fused Adam, capturable) | amp (bf16 autocast, channels_last, fused Adam)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
import torch, torch.nn.functional as Fn
import diee_amd
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.backends.cudnn.benchmark = True
blob = diee_amd.random_weights(0)
B = 256
x = torch.randn(B, 6, 4, 6, device="cuda"); ps = torch.softmax(torch.randn(B, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(B, 1, device="cuda"))
net = az.make_resnet().load_blob(blob).cuda().train()
amp = variant == "amp"
if amp:
    net = net.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=(variant != "base"), capturable=(variant == "graph"))


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        lg, v = net(x)
        loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
    opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    return loss


if variant == "graph":
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g = step()
    run = lambda: g.replay()
else:
    run = step
for _ in range(5):
    run()
torch.cuda.synchronize(); t = time.time()
for _ in range(steps):
    run()
torch.cuda.synchronize()
print(f"{variant}: {(time.time() - t) / steps * 1e3:.2f} ms/step", flush=True)
