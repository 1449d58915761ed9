import sys, time
sys.path.insert(0, "kb-ner_amd")
import torch
from kbner import ops, lib as L
dev = "cuda"
M, H = 256, 1024
h = torch.randn(M, H, device=dev).to(torch.bfloat16); y = torch.empty_like(h)
g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
x = torch.randn(256, 1024, device=dev).to(torch.bfloat16); w = torch.randn(1024, 1024, device=dev).to(torch.bfloat16); c = torch.empty(256, 1024, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(1024, device=dev)
for name, fn in (("ln_fwd", lambda: ops.ln_fwd(h, g, b, 1e-5, y, mean, rstd)),
                 ("gemm(256 path)", lambda: ops.gemm(0, x, w, 256, 1024, 1024, C=c, bias=bias, epi=1)),
                 ("torch.empty", lambda: torch.empty(256, 1024, device=dev)),
                 ("stream_ptr", lambda: L.stream_ptr())):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s host %.1f us/call   (incl. drain %.1f us/call)" % (name, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
