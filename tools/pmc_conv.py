"""Launch the dominant implicit-GEMM shapes a few times (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffcodec_amd import ops
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (h, c, cout, k) in [(64, 320, 320, 3), (32, 640, 640, 3), (64, 320, 320, 1), (16, 1280, 1280, 3), (8, 1280, 1280, 3)]:
    x = torch.randn(n, h, h, c, generator=g).to("cuda", torch.bfloat16)
    pc = ops.PackedConv(torch.randn(cout, c, k, k, generator=g) / math.sqrt(c * k * k), torch.zeros(cout), "cuda")
    for _ in range(5):
        y = ops.conv(x, pc)
    torch.cuda.synchronize()
    print(f"shape n={n} {h}x{h} {c}->{cout} k{k}: algorithmic bytes = {x.numel()*2 + pc.w.numel()*2 + y.numel()*2} M={n*h*h} grid_threads={'?'}")
# the K = 320 transformer linears on the row-panel kernel: folded-LayerNorm QKV (N = 960) and GEGLU (N = 2560)
xr = torch.randn(1, n * 4096, 320, generator=g).to("cuda", torch.bfloat16)
mr = ops.ln_finalize(ops.row_stats(xr), 320, 1e-5)
for (cout, geglu) in [(960, False), (2560, True)]:
    lnp = (1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g), 1e-5)
    pcl = ops.PackedConv(torch.randn(cout, 320, generator=g) / math.sqrt(320), torch.zeros(cout), "cuda", geglu=geglu, ln=lnp)
    for _ in range(5):
        yl = ops.linear(xr, pcl, ln_stats=mr)
    torch.cuda.synchronize()
    print(f"shape rowpanel M={n*4096} K=320 N={cout} geglu={int(geglu)}: algorithmic bytes = {xr.numel()*2 + pcl.w.numel()*2 + yl.numel()*2}")
# the wide GEGLU projections of the 32x32 and 16x16 transformers on the 256 x 256 four-phase kernel (gemm_p8.hip)
for (hw, cin, cout) in [(1024, 640, 5120), (256, 1280, 10240)]:
    xp = torch.randn(1, n * hw, cin, generator=g).to("cuda", torch.bfloat16)
    mrp = ops.ln_finalize(ops.row_stats(xp), cin, 1e-5)
    lnp = (1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g), 1e-5)
    pcp = ops.PackedConv(torch.randn(cout, cin, generator=g) / math.sqrt(cin), torch.zeros(cout), "cuda", geglu=True, ln=lnp)
    for _ in range(5):
        yp = ops.linear(xp, pcp, ln_stats=mrp)
    torch.cuda.synchronize()
    print(f"shape p8 M={n*hw} K={cin} N={cout} geglu=1: algorithmic bytes = {xp.numel()*2 + pcp.w.numel()*2 + yp.numel()*2}")
# flash attention, d = 40, 64x64 tokens (the third family by time)
q = torch.randn(n, 4096, 960, generator=g).to("cuda", torch.bfloat16)
for _ in range(3):
    o = ops.attention(q[..., :320], q[..., 320:640], q[..., 640:], 8)
torch.cuda.synchronize()
print(f"shape attention B={n} H=8 N=4096 d=40: algorithmic bytes = {q.numel()*2 + o.numel()*2}")
# the 32x32 self-attention (d = 80) and the 64x64 text cross-attention (77 keys, d = 40)
q8 = torch.randn(n, 1024, 1920, generator=g).to("cuda", torch.bfloat16)
for _ in range(3):
    o8 = ops.attention(q8[..., :640], q8[..., 640:1280], q8[..., 1280:], 8)
kv = torch.randn(n, 77, 640, generator=g).to("cuda", torch.bfloat16)
qs = torch.randn(n, 4096, 320, generator=g).to("cuda", torch.bfloat16)
for _ in range(3):
    os_ = ops.attention(qs, kv[..., :320], kv[..., 320:], 8)
torch.cuda.synchronize()
print(f"shape attention B={n} H=8 N=1024 d=80: algorithmic bytes = {q8.numel()*2 + o8.numel()*2}")
print(f"shape attention B={n} H=8 Nq=4096 Nk=77 d=40: algorithmic bytes = {qs.numel()*2 + kv.numel()*2 + os_.numel()*2}")
# GroupNorm apply (+SiLU) and statistics, 64x64x640: the HBM-bound passes
xg = torch.randn(n, 64, 64, 640, generator=g).to("cuda", torch.bfloat16)
ab = torch.randn(n, 640, 2, generator=g).to("cuda")
for _ in range(3):
    yg = ops.gn_apply(xg, ab, silu=True)
    st = ops.gn_stats(xg)
torch.cuda.synchronize()
print(f"shape gn_apply N={n} 64x64x640: algorithmic bytes = {xg.numel()*4}; gn_stats: {xg.numel()*2}")
