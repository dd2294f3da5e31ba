"""Multi-GPU decode: frames (or (frame, tile) units) of a GOP are independent, so the units
are sharded across ranks with NO data-path collective.  The only collective is one bucketed RCCL broadcast of the
packed weight tensors from rank 0 at start-up (xGMI, `torch.distributed` backend "nccl" == RCCL on ROCm; "gloo" in
the CPU tests), plus an optional gather of decoded frames."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """One process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_units(num_units, rank, world):
    """Round-robin assignment of decode units (inter frames / tiles) to ranks: unit u -> rank u % world.
    Returns this rank's unit indices (ascending).  Every unit is decoded exactly once."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, num_units, world))


def gop_inter_frames(num_frames, gop_size):
    """Frame indices decoded by diffusion (inter frames): every gop_size-th frame is intra
    (uvc_codec_eval.py:19-26); returns (inter frame idx, previous intra idx, next intra idx) triples."""
    units = []
    for f in range(num_frames):
        if f % gop_size == 0:
            continue
        prev_i = (f // gop_size) * gop_size
        next_i = prev_i + gop_size
        if next_i >= num_frames:
            continue                     # trailing frames without a closing anchor are not decodable bidirectionally
        units.append((f, prev_i, next_i))
    return units


def module_param_tensors(*modules):
    """All device-resident parameter tensors of the given operator objects, in deterministic construction order."""
    seen, out = set(), []

    def walk(o):
        if isinstance(o, torch.Tensor):
            if o.data_ptr() not in seen and o.numel() > 0:
                seen.add(o.data_ptr())
                out.append(o)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        elif isinstance(o, dict):
            for k in sorted(o, key=str):
                walk(o[k])
        elif hasattr(o, "__dict__") and type(o).__module__.split(".")[0] not in ("torch", "builtins", "types"):
            for k in sorted(vars(o)):
                if k.startswith("_") or k in ("kv_ctx", "gamma_beta", "all"):
                    continue                 # caches / activations, not parameters
                walk(vars(o)[k])

    for m in modules:
        walk(m)
    return out


def broadcast_params(tensors, src=0, bucket_bytes=256 << 20):
    """Bucketed broadcast (few, large collectives — ring/tree cost over xGMI is per-link bound, so bucket >= 64 MB)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    rank = dist.get_rank()
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    total = 0
    for dtype, ts in by_dtype.items():
        bucket, size = [], 0
        esz = ts[0].element_size()

        def flush():
            nonlocal bucket, size, total
            if not bucket:
                return
            flat = torch.empty(size, dtype=dtype, device=bucket[0].device)
            off = 0
            if rank == src:
                for t in bucket:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src=src)
            if rank != src:
                off = 0
                for t in bucket:
                    t.copy_(flat[off:off + t.numel()].reshape(t.shape))
                    off += t.numel()
            total += size * esz
            bucket, size = [], 0

        for t in ts:
            if not t.is_contiguous():
                raise ValueError("parameter tensors must be contiguous")
            if size and (size + t.numel()) * esz > bucket_bytes:
                flush()
            bucket.append(t)
            size += t.numel()
        flush()
    return total


def gather_frames(local_frames_u8, unit_ids, num_units, dst=0):
    """Optional: collect decoded uint8 frames [n_local,H,W,3] on `dst` in unit order (0.79 MB per 512x512 frame)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_frames_u8
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (num_units + world - 1) // world
    pad = torch.zeros((per,) + tuple(local_frames_u8.shape[1:]), dtype=local_frames_u8.dtype, device=local_frames_u8.device)
    pad[:local_frames_u8.shape[0]] = local_frames_u8
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((num_units,) + tuple(pad.shape[1:]), dtype=pad.dtype, device=pad.device)
    for r in range(world):
        ids = shard_units(num_units, r, world)
        out[ids] = bufs[r][:len(ids)]
    return out
