"""CPU: the GOP / tile decode driver (clip -> units -> shard -> batched decode -> gather -> blend) with a stand-in pipe
whose output is a pure function of its inputs, so that sharded, batched and unsharded decodes must agree exactly.
Clip layout and intra/inter split: validation.py:85-93, uvc_codec_eval.py:19-26."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffcodec_amd import clip_decode as CD
from diffcodec_amd import sharding


class FakePipe:
    """Same call surface as the pipeline for the keywords the driver uses; image = f(cond window, flow window, latents)."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def __call__(self, prompt_embeds=None, negative_prompt_embeds=None, controlnet_cond=None, flow_cond=None, latents=None,
                 output_type="pt", warp_cond=None, **kw):
        self.calls.append((controlnet_cond.shape[0], kw))
        assert prompt_embeds.shape[0] == controlnet_cond.shape[0] == flow_cond.shape[0] == latents.shape[0]
        up = torch.nn.functional.interpolate(latents[:, :3], scale_factor=8, mode="nearest")
        img = (controlnet_cond[:, :3] * 0.5 + controlnet_cond[:, 3:] * 0.25 + 0.01 * torch.tanh(flow_cond[:, :3]) + 0.05 * torch.tanh(up))
        if warp_cond is not None:
            img = img + 0.1 * warp_cond
        return SimpleNamespace(images=img.clamp(0, 1))


def test_units_follow_the_reference_gop_split():
    units = CD.plan_units(97, 12, 512, 512)
    assert len(units) == 88 and units[0].frame == 1 and (units[0].prev, units[0].next) == (0, 12) and units[-1].frame == 95
    assert all(u.window == (0, 512, 0, 512) and u.tile == 0 for u in units)
    assert CD.trailing_frames(97, 12) == [] and CD.trailing_frames(100, 12) == [97, 98, 99]
    assert len(CD.plan_units(100, 12, 512, 512)) == 88          # trailing frames have no closing anchor: dropped
    c4 = CD.plan_units(5, 4, 512, 960)                           # config 4: 960x512 -> two windows, x = 0 and 448
    assert [u.window for u in c4[:2]] == [(0, 512, 0, 512), (0, 512, 448, 960)] and len(c4) == 6
    c5 = CD.plan_units(13, 12, 1080, 1920)                       # config 5: 3 x 5 full-size windows per 1080p frame
    assert len(c5) == 11 * 15 and all(u.window[1] - u.window[0] == 512 and u.window[3] - u.window[2] == 512 for u in c5)
    for world in (1, 2, 8):
        for mode in ("unit", "frame"):
            got = sorted(u.uid for r in range(world) for u in CD.shard(c5, r, world, mode))
            assert got == list(range(len(c5)))
    assert {u.frame for u in CD.shard(c5, 1, 8, "frame")} == {2, 10}        # whole frames stay on one rank


def test_directory_source_paths_match_validation_py():
    src = CD.DirectorySource("data", "Bosphorus", 4, (512, 512), device="cpu")
    assert src.paths(3, 0, 4) == ("data/Bosphorus/images/frame_0000.png", "data/Bosphorus/images/frame_0004.png",
                                  "data/Bosphorus/optical_flow/optical_flow_gop_4_raft/flow_0000_0003.flo",
                                  "data/Bosphorus/optical_flow_bwd/optical_flow_gop_4_raft/flow_0004_0003.flo")


def test_directory_source_reads_the_layout(tmp_path):
    from PIL import Image
    from diffcodec_amd.io_utils import write_flo
    g = np.random.default_rng(0)
    root = tmp_path / "clip"
    for sub in ("images", "optical_flow/optical_flow_gop_4_raft", "optical_flow_bwd/optical_flow_gop_4_raft"):
        os.makedirs(root / "v" / sub)
    for i in (0, 4):
        Image.fromarray(g.integers(0, 255, (40, 48, 3), dtype=np.uint8)).save(root / "v" / "images" / f"frame_{i:04d}.png")
    write_flo(str(root / "v" / "optical_flow/optical_flow_gop_4_raft/flow_0000_0001.flo"), g.normal(size=(40, 48, 2)))
    write_flo(str(root / "v" / "optical_flow_bwd/optical_flow_gop_4_raft/flow_0004_0001.flo"), g.normal(size=(40, 48, 2)))
    cond, flow = CD.DirectorySource(str(root), "v", 4, (32, 32), device="cpu").controls(1, 0, 4)
    assert cond.shape == (1, 6, 32, 32) and flow.shape == (1, 4, 32, 32) and 0 <= float(cond.min()) and float(cond.max()) <= 1


def test_batching_and_sharding_do_not_change_the_frames():
    h, w = 512, 960
    src = CD.SyntheticSource(h, w, device="cpu", with_warp=True)
    pe = torch.zeros(1, 77, 8)
    ref = CD.decode_clip(FakePipe(), src, 9, 4, h, w, pe, pe, batch=64, seed=3, rank=0, world=1, num_inference_steps=50)
    assert sorted(ref["frames"]) == [1, 2, 3, 5, 6, 7] and ref["frames"][1].shape == (h, w, 3) and ref["frames"][1].dtype == np.uint8
    p = FakePipe()
    small = CD.decode_clip(p, src, 9, 4, h, w, pe, pe, batch=5, seed=3, rank=0, world=1, num_inference_steps=50)
    assert [c[0] for c in p.calls] == [5, 5, 2] and p.calls[0][1] == dict(num_inference_steps=50)   # 12 units in batches of 5
    assert all(np.array_equal(small["frames"][f], ref["frames"][f]) for f in ref["frames"])
    parts = [CD.decode_clip(FakePipe(), src, 9, 4, h, w, pe, pe, batch=4, seed=3, rank=r, world=2, shard_mode="frame", gather=False)
             for r in range(2)]
    merged = {**parts[0]["frames"], **parts[1]["frames"]}
    assert sorted(merged) == sorted(ref["frames"]) and all(np.array_equal(merged[f], ref["frames"][f]) for f in merged)
    per_unit = [CD.decode_clip(FakePipe(), src, 9, 4, h, w, pe, pe, batch=4, seed=3, rank=r, world=2, gather=False) for r in range(2)]
    assert per_unit[0]["frames"] is None and [u.uid for u in per_unit[0]["mine"]] == [0, 2, 4, 6, 8, 10]   # tiles split: nothing complete


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = sharding.init_from_env(backend="gloo")
    h, wd = 512, 960
    src = CD.SyntheticSource(h, wd, device="cpu")
    pe = torch.zeros(1, 77, 8)
    out = CD.decode_clip(FakePipe(), src, 5, 4, h, wd, pe, pe, batch=2, seed=11)        # rank / world from the process group
    ok = None
    if r == 0:
        ref = CD.decode_clip(FakePipe(), src, 5, 4, h, wd, pe, pe, batch=8, seed=11, rank=0, world=1)
        ok = sorted(out["frames"]) == [1, 2, 3] and all(np.array_equal(out["frames"][f], ref["frames"][f]) for f in ref["frames"])
    q.put((r, [u.uid for u in out["mine"]], ok, out["frames"] is None))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_tile_sharded_clip_equals_single_rank():
    """config-5 style per-patch shard at world size 2 (gloo): tiles of one frame are decoded on different ranks, gathered on
    rank 0 and blended; identical to the single-rank decode."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3, 5]
    assert res[0][2] is True and res[1][3] is True          # frames exist on the gathering rank only


def _worker_more_ranks_than_units(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = sharding.init_from_env(backend="gloo")
    h = wd = 256
    src = CD.SyntheticSource(h, wd, device="cpu")
    pe = torch.zeros(1, 77, 8)
    out = CD.decode_clip(FakePipe(), src, 3, 2, h, wd, pe, pe, tile=256, batch=2, seed=5)       # ONE inter frame = one unit for two ranks
    ok = None
    if r == 0:
        ref = CD.decode_clip(FakePipe(), src, 3, 2, h, wd, pe, pe, tile=256, batch=2, seed=5, rank=0, world=1)
        ok = sorted(out["frames"]) == [1] and np.array_equal(out["frames"][1], ref["frames"][1]) and out["frames"][1].shape == (h, wd, 3)
    q.put((r, len(out["mine"]), tuple(out["images"].shape), ok))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_more_ranks_than_units():
    """a rank with an empty share pads its gather buffer to the SAME unit size as the others (frame- / tile-sized units,
    here a 256x256 clip decoded as 256x256 units): the tensor gather neither hangs nor mis-shapes, rank 0 gets the frame."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_more_ranks_than_units, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == (1, (1, 3, 256, 256), True) and res[1][1:] == (0, (0, 3, 256, 256), None)


def test_driver_edge_cases_empty_share_and_prompt_batch():
    """a rank with no units keeps the [n,3,th,tw] shape (world > #units), the uint8 gather handles it, and a prompt batch
    that is neither 1 nor >= the chunk is refused instead of silently truncated."""
    h = w = 512
    src = CD.SyntheticSource(h, w, device="cpu")
    pe = torch.zeros(1, 77, 8)
    units = CD.plan_units(5, 4, h, w)                                   # 3 units
    shares = [CD.shard(units, r, 4) for r in range(4)]
    assert [len(s) for s in shares] == [1, 1, 1, 0]
    empty = CD.decode_units(FakePipe(), shares[3], src, pe, pe, batch=2)
    assert empty.shape == (0, 3, 512, 512) and CD.units_to_u8(empty).shape == (0, 512, 512, 3)
    assert CD.decode_units(FakePipe(), [], src, pe, pe, output="latent").shape == (0, 4, 64, 64)
    full = torch.cat([CD.decode_units(FakePipe(), s, src, pe, pe, batch=2) for s in shares], 0)      # concatenates without a special case
    assert full.shape == (3, 3, 512, 512)
    out = CD.decode_clip(FakePipe(), src, 5, 4, h, w, pe, pe, batch=2, rank=3, world=4, gather=False)
    assert out["images"].shape == (0, 3, 512, 512) and out["frames"] is None
    with pytest.raises(ValueError, match="prompt_embeds has batch 2"):
        CD.decode_units(FakePipe(), units, src, torch.zeros(2, 77, 8), None, batch=3)
    with pytest.raises(ValueError, match="uint8"):
        CD.gather_units(full, units, 0, 1)
    one = CD.gather_units(CD.units_to_u8(full), units, 0, 1)
    assert one.shape == (3, 512, 512, 3) and one.dtype == torch.uint8
