"""Developer tool: where does a long-context attention wave spend its life, and at what clock?  Builds csrc/attention.hip with -DDC_STAMP
into a scratch .so, runs the d = 40 64x64 self-attention back to back for ~2 s (so that the chip is at the clock it holds under this
load), then reads the stamps of the LAST launch: per-wave phase sums (shader cycles, s_memtime), the in-kernel clock of the tile loop
(delta s_memtime / delta s_memrealtime x 100 MHz), the cycles a wave spends before / after the loop, and — from HW_ID / XCC_ID — the
timeline of every CU: how long it sits between the exit of one workgroup and the entry of the next.
usage: python tools/attn_stamp.py [-DDC_...] [B=32]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diffcodec_amd import lib
PKG = os.path.dirname(lib.LIB_PATH)
so = "/tmp/libdc_astamp.so"
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("-D"))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-DDC_STAMP", "-o", so] + extra +
                      [os.path.join(PKG, "csrc", "attention.hip")])
L = ctypes.CDLL(so)
vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
L.dc_attention_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ctypes.c_float, vp]
L.dc_attn_stamp_read.argtypes = [vp, ci]
B, H, N, D = int(kv.get("B", 32)), 8, 4096, 40
SLOTS = 16
q = torch.randn(B, N, 3 * H * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B, N, H * D, device="cuda", dtype=torch.bfloat16)
st = 3 * H * D
def launch():
    rc = L.dc_attention_bf16(q.data_ptr(), q.data_ptr() + H * D * 2, q.data_ptr() + 2 * H * D * 2, o.data_ptr(), B, H, N, N, D, st, st, st, H * D,
                             D ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n_l = 1500
for _ in range(n_l):
    launch()
e1.record()
torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) / n_l * 1e3
nw = 1 << 14
buf = torch.zeros(nw * SLOTS, dtype=torch.int64)
assert L.dc_attn_stamp_read(buf.data_ptr(), nw * SLOTS) == 0
t = buf.view(nw, SLOTS).double()
t = t[t[:, 4] > 0]
tiles = N // 64
m = lambda c: t[:, c].median().item()
print("launch wall time (stamped build, mean of %d back-to-back launches): %.1f us" % (n_l, wall_us))
print("waves sampled", len(t), "key tiles per wave", tiles)
print("median cycles per key tile and wave: QK^T + row max %.0f | exp + convert (+ rescale) %.0f | PV %.0f | stage next tile + barrier %.0f | sum %.0f" %
      (m(0) / tiles, m(1) / tiles, m(2) / tiles, m(3) / tiles, (m(0) + m(1) + m(2) + m(3)) / tiles))
pp = t[:, 5].max() > 0
if pp:
    print("  (ping-pong form: the four columns are MFMA block {PV(t-1), QK^T(t)} | barrier wait | softmax block | barrier wait)")
print("MFMA floor per tile and wave: QK^T 12 x 32 = 384, PV 16 x 32 = 512;  tile loop %.0f cycles per wave" % m(4))
if pp:
    wv = t.view(-1, 8, SLOTS)
    for g, name in ((slice(0, 4), "group 0 (waves 0-3, stage K)"), (slice(4, 8), "group 1 (waves 4-7, stage V)")):
        x = wv[:, g].reshape(-1, SLOTS)
        mm = lambda c: x[:, c].mean().item() / (tiles - 1)
        print("  %s: mean cycles per tile: MFMA block %.0f (of it V wait + write %.0f) | wait at mid barrier %.0f | softmax block %.0f | wait at end barrier %.0f | sum %.0f" %
              (name, mm(0), mm(14), mm(1), mm(2), mm(3), mm(0) + mm(1) + mm(2) + mm(3)))
    clk = (t[:, 4] / t[:, 7]).median().item() * 0.1          # cycles per 10 ns tick -> GHz
    print("in-kernel clock of the tile loop (delta s_memtime / delta s_memrealtime): %.3f GHz (median over waves; min %.3f max %.3f)" %
          (clk, (t[:, 4] / t[:, 7]).min().item() * 0.1, (t[:, 4] / t[:, 7]).max().item() * 0.1))
    print("cycles per wave: entry -> loop %.0f | loop %.0f | loop end -> stores retired %.0f | whole wave %.0f" % (m(12), m(4), m(13), m(6)))
    # per-CU timeline: workgroup = 8 consecutive waves of the buffer
    wg = t.view(-1, 8, SLOTS)
    t_in, t_out = wg[:, :, 8].min(dim=1).values, wg[:, :, 9].max(dim=1).values
    hw, xcc = wg[:, 0, 10].long(), wg[:, 0, 11].long()
    cu = ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff)              # XCC, SE / SH / CU bits of HW_ID
    span = (t_out.max() - t_in.min()).item() / 100.0          # us
    busy, gaps, n_cu = 0.0, [], 0
    for c in cu.unique().tolist():
        idx = (cu == c).nonzero().flatten()
        order = idx[t_in[idx].argsort()]
        n_cu += 1
        busy += (t_out[order] - t_in[order]).sum().item() / 100.0
        if len(order) > 1:
            gaps += ((t_in[order][1:] - t_out[order][:-1]) / 100.0).tolist()
    g = torch.tensor(gaps)
    print("launch span first entry -> last exit %.1f us; %d CUs seen, %.2f workgroups per CU; per CU: occupied %.1f us (mean), between workgroups %.2f us median / %.2f mean (%d gaps, %d negative = two workgroups resident)" %
          (span, n_cu, len(wg) / n_cu, busy / n_cu, g.median().item(), g.mean().item(), len(g), int((g < 0).sum())))
    wl = (t_out - t_in) / 100.0
    print("workgroup lifetime %.1f us median (min %.1f, max %.1f); x %.2f rounds = %.1f us" % (wl.median().item(), wl.min().item(), wl.max().item(), len(wg) / n_cu, wl.median().item() * len(wg) / n_cu))
