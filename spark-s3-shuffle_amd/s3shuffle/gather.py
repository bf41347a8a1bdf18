"""Optional reduce-side gather over xGMI (the path's only exchange step).

Map outputs are sharded `mapId % nGPU`, so after the verify+decompress step rank g holds the decoded
partitions of ITS map outputs only, while reducer r (placed on rank r % nGPU) needs partition r of
EVERY map output.  That is an all-to-all-v of byte ranges: with `torch.distributed` on the
`nccl` backend it runs as RCCL send/recv pairs over the point-to-point xGMI links (all 7 links of a
GPU busy at once, unlike a ring); with `gloo` the same code runs on CPU tensors (used by the tests).
Nothing on the compress path calls this.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def reducer_rank(reduce_id: int, world: int) -> int:
    return reduce_id % world


def gather_reduce_partitions(decoded: Dict[Tuple[int, int], torch.Tensor], num_reduce: int,
                             group=None) -> Dict[Tuple[int, int], torch.Tensor]:
    """decoded: {(mapId, reduceId): uint8 tensor} for the map outputs this rank decoded.
    Returns {(mapId, reduceId): tensor} for every reduceId this rank reduces (reduceId % world == rank),
    across ALL map outputs of all ranks.  Two collectives: one all_to_all of (mapId, reduceId, length)
    headers, one all_to_all of the concatenated payloads."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = next(iter(decoded.values())).device if decoded else torch.device("cpu")
    send_keys: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    for (m, r) in sorted(decoded):
        if not 0 <= r < num_reduce:
            raise ValueError(f"reduce id {r} out of range")
        send_keys[reducer_rank(r, world)].append((m, r))
    # headers: [count, (mapId, reduceId, nbytes)*]  -- first exchange the header sizes, then the headers
    hdr = [torch.tensor([v for (m, r) in ks for v in (m, r, decoded[(m, r)].numel())], dtype=torch.int64, device=device)
           for ks in send_keys]
    n_send = torch.tensor([h.numel() for h in hdr], dtype=torch.int64, device=device)
    n_recv = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(n_recv, n_send, group=group)
    hdr_in = torch.empty(int(n_recv.sum()), dtype=torch.int64, device=device)
    dist.all_to_all_single(hdr_in, torch.cat(hdr) if hdr else torch.empty(0, dtype=torch.int64, device=device),
                           output_split_sizes=n_recv.tolist(), input_split_sizes=n_send.tolist(), group=group)
    triples = hdr_in.view(-1, 3).tolist()
    # payloads
    send_bytes = [sum(decoded[k].numel() for k in ks) for ks in send_keys]
    recv_bytes = [0] * world
    off = 0
    for src in range(world):
        cnt = int(n_recv[src]) // 3
        recv_bytes[src] = sum(t[2] for t in triples[off:off + cnt])
        off += cnt
    payload_out = torch.cat([decoded[k].reshape(-1) for ks in send_keys for k in ks]) if decoded else \
        torch.empty(0, dtype=torch.uint8, device=device)
    payload_in = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=device)
    dist.all_to_all_single(payload_in, payload_out, output_split_sizes=recv_bytes, input_split_sizes=send_bytes, group=group)
    out: Dict[Tuple[int, int], torch.Tensor] = {}
    pos = 0
    for m, r, n in triples:
        assert reducer_rank(r, world) == rank
        out[(m, r)] = payload_in[pos:pos + n]
        pos += n
    return out
