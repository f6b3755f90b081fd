import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
az = importlib.import_module("die-e_amd.alphazero")
import numpy as np, torch, torch.nn.functional as Fn
import diee_amd
blob = diee_amd.random_weights(0)
x = torch.randn(256, 6, 4, 6, device="cuda"); ps = torch.softmax(torch.randn(256, 1352, device="cuda"), 1); oc = torch.sign(torch.randn(256, 1, device="cuda"))
def run(name, bench=False, cl=False, amp=False, steps=30):
    torch.backends.cudnn.benchmark = bench
    net = az.make_resnet().load_blob(blob).cuda()
    xx = x
    if cl:
        net = net.to(memory_format=torch.channels_last); xx = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    net.train()
    for i in range(steps + 5):
        if i == 5:
            torch.cuda.synchronize(); t = time.time()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            lg, v = net(xx)
            loss = Fn.cross_entropy(lg.float(), ps) + Fn.mse_loss(v.float(), oc)
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.time() - t) / steps * 1e3:7.2f} ms/step  loss {float(loss):.4f}", flush=True)
run("fp32 NCHW")
run("fp32 NCHW + miopen benchmark", bench=True)
run("fp32 channels_last + benchmark", bench=True, cl=True)
run("bf16 autocast NCHW + benchmark", bench=True, amp=True)
run("bf16 autocast channels_last + benchmark", bench=True, cl=True, amp=True)
