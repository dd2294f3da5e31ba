"""CPU, world_size 2, gloo: the N>1 path of bench.py — weight broadcast from rank 0, frame sharding, frame gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Holder:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.convs = [dict(w=torch.randn(37, 5, generator=g).to(torch.bfloat16), bias=torch.randn(37, generator=g)) for _ in range(5)]
        self.norm = (torch.randn(16, generator=g), torch.randn(16, generator=g))
        self.kv_ctx = torch.zeros(3)          # cache: must NOT be broadcast


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from diffcodec_amd import sharding
    r, _, w = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    m = _Holder(seed=100 + rank)              # ranks start with DIFFERENT weights
    params = sharding.module_param_tensors(m)
    assert len(params) == 12
    nbytes = sharding.broadcast_params(params, src=0, bucket_bytes=300)     # tiny buckets -> several collectives
    ref = _Holder(seed=100)
    same = all(torch.equal(a["w"], b["w"]) and torch.equal(a["bias"], b["bias"]) for a, b in zip(m.convs, ref.convs))
    same = same and torch.equal(m.norm[0], ref.norm[0]) and torch.equal(m.norm[1], ref.norm[1])
    # frames: 7 units round-robin, each rank "decodes" its units (value = unit id), rank 0 gathers in unit order
    units = sharding.shard_units(7, rank, world)
    frames = torch.stack([torch.full((2, 2, 3), u, dtype=torch.uint8) for u in units])
    out = sharding.gather_frames(frames, units, 7, dst=0)
    ok_gather = True if rank != 0 else bool((out[:, 0, 0, 0] == torch.arange(7, dtype=torch.uint8)).all())
    q.put((rank, same, nbytes > 0, ok_gather, units))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "weights differ after broadcast"
    assert res[0][2] and res[0][3]
    assert res[0][4] == [0, 2, 4, 6] and res[1][4] == [1, 3, 5]
