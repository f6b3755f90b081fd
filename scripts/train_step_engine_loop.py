"""the training step with the tower on the engine's kernels, in a loop (for rocprofv3 --kernel-trace --stats)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
ops = importlib.import_module("die-e_amd.train_ops")
import torch, torch.nn.functional as Fn
import diee_amd
blob = diee_amd.random_weights(0)
B = 256
x = torch.randn(B, 6, 4, 6, device="cuda"); ps = torch.softmax(torch.randn(B, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(B, 1, device="cuda"))
net = az.make_resnet().load_blob(blob).cuda().train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
for i in range(25):
    if i == 5:
        torch.cuda.synchronize(); t = time.time()
    lg, v = ops.forward_train_tokens(net, x)
    loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print(f"{(time.time() - t) / 20 * 1e3:.2f} ms/step", flush=True)
