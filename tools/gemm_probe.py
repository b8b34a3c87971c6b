import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
N = 65536
def bench(f, flops, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    return us, flops / us / 1e6
for (I, O) in ((256, 256), (320, 256), (64, 256)):
    x = torch.randn(N, I, device=dev); w = torch.randn(O, I, device=dev); dy = torch.randn(N, O, device=dev)
    out = torch.empty(N, O, device=dev); dx = torch.empty(N, I, device=dev); dw = torch.empty(O, I, device=dev)
    print(f"I={I} O={O}")
    print("  fwd   x @ w.T      %7.1f us %6.1f TF" % bench(lambda: torch.matmul(x, w.t(), out=out), 2.0 * N * I * O))
    print("  dgrad dy @ w       %7.1f us %6.1f TF" % bench(lambda: torch.matmul(dy, w, out=dx), 2.0 * N * I * O))
    print("  wgrad dy.T @ x     %7.1f us %6.1f TF" % bench(lambda: torch.matmul(dy.t(), x, out=dw), 2.0 * N * I * O))
