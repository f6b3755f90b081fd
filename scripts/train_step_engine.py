"""ms per training step: all-PyTorch (MIOpen) vs the tower on the engine's kernels (die-e_amd/train_ops.py), batch 256."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
ops = importlib.import_module("die-e_amd.train_ops")
import torch, torch.nn.functional as Fn
import diee_amd
torch.backends.cudnn.benchmark = True
blob = diee_amd.random_weights(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 6, 4, 6, device="cuda"); ps = torch.softmax(torch.randn(B, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(B, 1, device="cuda"))
for name, eng, fused in (("pytorch fp32 (MIOpen)", False, False), ("engine tower kernels", True, False), ("engine tower kernels + fused Adam", True, True)):
    net = az.make_resnet().load_blob(blob).cuda().train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=fused)
    for i in range(25):
        if i == 5:
            torch.cuda.synchronize(); t = time.time()
        lg, v = ops.forward_train_tokens(net, x) if eng else net(x)
        loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"{name:36s} {(time.time() - t) / 20 * 1e3:7.2f} ms/step  loss {float(loss):.4f}", flush=True)
