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
                             group=None, local_through_collective: bool = False) -> Dict[Tuple[int, int], torch.Tensor]:
    """decoded: {(mapId, reduceId): uint8 tensor} for the map outputs this rank decoded (views into the decode
    buffers are fine: nothing is staged).  Returns {(mapId, reduceId): tensor} for every reduceId this rank reduces
    (reduceId % world == rank), across ALL map outputs of all ranks.

    Exchange: one all_to_all of header sizes + one of (mapId, reduceId, length) headers (a few KiB), then ONE grouped
    batch of point-to-point sends / receives (`batch_isend_irecv` = ncclGroupStart .. ncclGroupEnd on RCCL): every
    partition goes from the decode buffer it was written to straight into its slice of the receive buffer — no
    concatenation on either side.  Partitions that stay on this rank are returned as the caller's own tensors
    (no copy) unless `local_through_collective` asks for the self send / recv (used by the single-GPU RCCL test)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = next(iter(decoded.values())).device if decoded else torch.device("cpu")
    send_keys: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    for (m, r) in sorted(decoded):
        if not 0 <= r < num_reduce:
            raise ValueError(f"reduce id {r} out of range")
        send_keys[reducer_rank(r, world)].append((m, r))
    # headers: (mapId, reduceId, nbytes)* per destination -- first the header sizes, then the headers
    flat = [v for ks in send_keys for (m, r) in ks for v in (m, r, decoded[(m, r)].numel())]
    hdr_out = torch.tensor(flat, dtype=torch.int64, device=device)
    n_send = torch.tensor([3 * len(ks) for ks in send_keys], dtype=torch.int64, device=device)
    n_recv = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(n_recv, n_send, group=group)
    n_recv_l = n_recv.tolist()
    hdr_in = torch.empty(int(sum(n_recv_l)), dtype=torch.int64, device=device)
    dist.all_to_all_single(hdr_in, hdr_out, output_split_sizes=n_recv_l, input_split_sizes=n_send.tolist(), group=group)
    triples = hdr_in.view(-1, 3).tolist()
    # payloads: receive slices in header order (= the sender's issue order per peer), sends straight from `decoded`
    keep_local = not local_through_collective
    remote_bytes = 0
    off = 0
    per_src: List[List[Tuple[int, int, int]]] = []
    for src in range(world):
        cnt = int(n_recv_l[src]) // 3
        per_src.append([tuple(t) for t in triples[off:off + cnt]])
        off += cnt
        if not (keep_local and src == rank):
            remote_bytes += sum(t[2] for t in per_src[-1])
    payload_in = torch.empty(remote_bytes, dtype=torch.uint8, device=device)
    out: Dict[Tuple[int, int], torch.Tensor] = {}
    ops = []
    pos = 0
    for src in range(world):
        for (m, r, n) in per_src[src]:
            assert reducer_rank(r, world) == rank
            if keep_local and src == rank:
                out[(m, r)] = decoded[(m, r)].reshape(-1)
                continue
            out[(m, r)] = payload_in[pos:pos + n]
            if n > 0:
                ops.append(dist.P2POp(dist.irecv, out[(m, r)], src, group))
            pos += n
    for dst in range(world):
        if keep_local and dst == rank:
            continue
        for k in send_keys[dst]:
            t = decoded[k].reshape(-1)
            if t.numel() > 0:
                ops.append(dist.P2POp(dist.isend, t, dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
