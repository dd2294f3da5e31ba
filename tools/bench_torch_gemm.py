"""Developer tool (GPU): the library GEMM (torch.matmul / F.linear -> hipBLASLt / rocBLAS) on the decode's 1x1 / linear shapes at model
batch 32, as a yardstick for the hand-written GEMM family (tools/bench_gemm.py prints the same shapes through dc_conv_igemm_bf16)."""
import torch
import torch.nn.functional as F
SHAPES = [(131072, 320, 320), (131072, 320, 960), (131072, 320, 2560), (131072, 1280, 320), (32768, 640, 640), (32768, 640, 1920),
          (32768, 640, 5120), (32768, 2560, 640), (8192, 1280, 1280), (8192, 1280, 3840), (8192, 1280, 10240), (8192, 5120, 1280),
          (2048, 1280, 1280), (2048, 1280, 10240), (2048, 5120, 1280)]
print("us  TFLOP/s  shape (M K N), F.linear with bias, bf16")
for (m, k, n) in SHAPES:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    f = lambda: F.linear(x, w, b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{us:9.1f} {2.0 * m * n * k / us / 1e6:8.1f}  M={m} K={k} N={n}", flush=True)
