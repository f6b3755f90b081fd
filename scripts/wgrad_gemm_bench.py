"""wgrad GEMM formulations for one tower conv at batch 256: col^T @ dy vs dy^T @ col (bf16, fp32 accumulate), us per call"""
import time, torch
M, K9, N = 6144, 2304, 256
col = torch.randn(M, K9, device="cuda").to(torch.bfloat16); dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
colT = col.t().contiguous(); dyT = dy.t().contiguous()
def bench(name, f, reps=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); print(f"{name:44s} {(time.time() - t) / reps * 1e6:8.1f} us", flush=True)
bench("col.t() @ dy            [2304,6144]x[6144,256]", lambda: torch.matmul(col.t(), dy))
bench("dy.t() @ col            [256,6144]x[6144,2304]", lambda: torch.matmul(dy.t(), col))
bench("colT(contig) @ dy", lambda: torch.matmul(colT, dy))
bench("dyT(contig) @ col", lambda: torch.matmul(dyT, col))
out = torch.empty(K9, N, device="cuda", dtype=torch.float32)
bench("fp32 out: col.t().float() path n/a -> baddbmm skip", lambda: None, 1)
# split-K by hand: 4 chunks of rows, bmm then sum
c4 = col.view(4, M // 4, K9); d4 = dy.view(4, M // 4, N)
bench("bmm 4 row-chunks + sum", lambda: torch.bmm(c4.transpose(1, 2), d4).float().sum(0))
c8 = col.view(8, M // 8, K9); d8 = dy.view(8, M // 8, N)
bench("bmm 8 row-chunks + sum", lambda: torch.bmm(c8.transpose(1, 2), d8).float().sum(0))
