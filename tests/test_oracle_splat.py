"""CPU: hand-computed known-answer cases for the plain-C forward-splat restatement (softsplat.py:285-335).
The reference kernel is CUDA-only and ships no fixture, so these KATs are the pin ("parity unpinned"
against an actual CUDA run)."""
import torch

from oracle import splat as S


def test_integer_shift_moves_pixels():
    x = torch.arange(12.0).reshape(1, 1, 3, 4)
    flow = torch.zeros(1, 2, 3, 4)
    flow[:, 0] = 1.0                       # +1 in x
    out = S.splat_sum(x, flow)
    exp = torch.zeros_like(x)
    exp[..., 1:] = x[..., :-1]             # last column leaves the frame
    assert torch.equal(out, exp)


def test_bilinear_weights_quarter():
    x = torch.zeros(1, 1, 4, 4)
    x[0, 0, 1, 1] = 8.0
    flow = torch.zeros(1, 2, 4, 4)
    flow[0, 0, 1, 1] = 0.25                # target (1.25, 1.5)
    flow[0, 1, 1, 1] = 0.5
    out = S.splat_sum(x, flow)
    assert out[0, 0, 1, 1].item() == 8.0 * 0.75 * 0.5
    assert out[0, 0, 1, 2].item() == 8.0 * 0.25 * 0.5
    assert out[0, 0, 2, 1].item() == 8.0 * 0.75 * 0.5
    assert out[0, 0, 2, 2].item() == 8.0 * 0.25 * 0.5
    assert out.sum().item() == 8.0


def test_out_of_bounds_and_nonfinite_are_dropped():
    x = torch.ones(1, 2, 2, 2)
    flow = torch.zeros(1, 2, 2, 2)
    flow[0, 0, 0, 0] = -5.0                # fully outside
    flow[0, 1, 0, 1] = float("nan")        # skipped (softsplat.py:301-302)
    flow[0, 0, 1, 0] = float("inf")
    flow[0, 0, 1, 1] = 0.5                 # half leaves the frame on the right
    out = S.splat_sum(x, flow)
    exp = torch.zeros(1, 2, 2, 2)
    exp[0, :, 1, 1] = 0.5
    assert torch.equal(out, exp)


def test_soft_mode_normalises_collisions():
    # two sources land on the same target; 'soft' = exp(metric)-weighted mean (softsplat.py:246-270)
    x = torch.tensor([[[[2.0, 6.0]]]])
    flow = torch.zeros(1, 2, 1, 2)
    flow[0, 0, 0, 0] = 1.0                 # pixel 0 -> pixel 1, pixel 1 stays
    metric = torch.tensor([[[[0.0, 1.0]]]])
    out = S.softsplat(x, flow, metric, "soft")
    e0, e1 = torch.tensor(0.0).exp(), torch.tensor(1.0).exp()
    exp1 = (2.0 * e0 + 6.0 * e1) / (e0 + e1 + 1e-7)
    assert out[0, 0, 0, 0].item() == 0.0
    assert abs(out[0, 0, 0, 1].item() - exp1.item()) < 1e-6


def test_c_loop_matches_vectorised_restatement():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 7, 20, 24, generator=g)
    flow = torch.randn(2, 2, 20, 24, generator=g) * 4
    m = torch.randn(2, 1, 20, 24, generator=g)
    torch.testing.assert_close(S.softsplat(x, flow, m, "soft"), S.softsplat_torch(x, flow, m, "soft"), rtol=1e-5, atol=1e-5)
