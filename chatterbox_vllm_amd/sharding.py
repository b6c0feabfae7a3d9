"""Request sharding over the GPUs of one node: one process per GPU, ``torch.distributed``.

The reference serves every request from a single ``vllm.LLM`` (src/chatterbox_vllm/tts.py:130-140;
no tensor/data parallel setting anywhere), so this layer has no counterpart there.  T3 requests
are independent, which makes the multi-GPU form embarrassingly parallel: every rank holds the
whole checkpoint (1.02 GB) and its own KV pool and scheduler, requests are dealt round-robin,
and the data path has no collective.  Collectives (RCCL over xGMI with the "nccl" backend, gloo
on CPU for tests) are used only for

* one broadcast of the checkpoint from rank 0 at start-up, in 256 MiB buckets,
* one all-gather of the generated ids (int32, padded) + lengths at the end of a call.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

BUCKET_BYTES = 256 << 20


def shard_indices(n_items: int, world: int, rank: int) -> List[int]:
    """Round-robin deal: item i goes to rank i % world (keeps long and short prompts mixed)."""
    return list(range(rank, n_items, world))


def broadcast_state_dict(sd: Dict[str, torch.Tensor], src: int = 0, device: Optional[torch.device] = None,
                         group=None) -> Dict[str, torch.Tensor]:
    """Every rank passes a state dict with identical names/shapes/dtypes (e.g. empty tensors on
    ranks != src); returns CPU tensors holding rank `src`'s values."""
    device = device or torch.device("cpu")
    names = sorted(sd)
    out: Dict[str, torch.Tensor] = {}
    bucket: List[str] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([sd[n].reshape(-1).view(torch.uint8) for n in bucket]).to(device)
        dist.broadcast(flat, src=src, group=group)
        flat = flat.cpu()
        off = 0
        for n in bucket:
            nb = sd[n].numel() * sd[n].element_size()
            out[n] = flat[off:off + nb].view(sd[n].dtype).reshape(sd[n].shape).clone()
            off += nb
        bucket, size = [], 0

    for n in names:
        nb = sd[n].numel() * sd[n].element_size()
        if size and size + nb > BUCKET_BYTES:
            flush()
        bucket.append(n)
        size += nb
    flush()
    return out


def gather_token_ids(local: Sequence[Sequence[int]], device: Optional[torch.device] = None, group=None
                     ) -> List[List[int]]:
    """All-gather ragged id lists.  Returns, on every rank, rank 0's lists, then rank 1's, ..."""
    device = device or torch.device("cpu")
    world = dist.get_world_size(group)
    n_local = torch.tensor([len(local), max((len(x) for x in local), default=0)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    n_max = max(int(s[0]) for s in sizes)
    l_max = max(int(s[1]) for s in sizes)
    buf = torch.full((n_max, l_max + 1), -1, dtype=torch.int32, device=device)
    for i, x in enumerate(local):
        buf[i, 0] = len(x)
        if len(x):
            buf[i, 1:1 + len(x)] = torch.tensor(list(x), dtype=torch.int32)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out: List[List[int]] = []
    for r in range(world):
        b = bufs[r].cpu()
        for i in range(int(sizes[r][0])):
            out.append(b[i, 1:1 + int(b[i, 0])].tolist())
    return out


def generate_sharded(engine, request_items: Sequence[dict], sampling_params, device: Optional[torch.device] = None,
                     group=None) -> List[List[int]]:
    """Each rank generates its round-robin share on its own engine; every rank returns the token
    ids of ALL requests in input order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = shard_indices(len(request_items), world, rank)
    sps = sampling_params if isinstance(sampling_params, (list, tuple)) else [sampling_params] * len(request_items)
    outs = engine.generate([request_items[i] for i in mine], [sps[i] for i in mine]) if mine else []
    gathered = gather_token_ids([o.outputs[0].token_ids for o in outs], device=device, group=group)
    result: List[Optional[List[int]]] = [None] * len(request_items)
    k = 0
    for r in range(world):
        for i in shard_indices(len(request_items), world, r):
            result[i] = gathered[k]
            k += 1
    return result  # type: ignore[return-value]
