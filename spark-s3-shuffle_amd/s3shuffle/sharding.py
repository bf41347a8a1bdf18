"""Map-task -> GPU sharding and the on-store naming scheme of the reference.

The reference spreads map outputs over object-store prefixes by `mapId % folderPrefixes`
(S3ShuffleDispatcher.scala:142-143); the GPU path shards the same independent units — one map
task's `.data/.index/.checksum` triple — across the GPUs of a node by `mapId % nGPU`.  The
compress path has no exchange step, so ranks never communicate on the data path.
"""
from __future__ import annotations

from typing import List


def device_for_map(map_id: int, n_gpus: int) -> int:
    """GPU (rank) that owns map task `map_id`: mapId % nGPU."""
    if n_gpus <= 0:
        raise ValueError("n_gpus must be positive")
    if map_id < 0:
        raise ValueError("map_id must be non-negative")
    return map_id % n_gpus


def map_ids_for_rank(rank: int, n_gpus: int, maps_per_gpu: int) -> List[int]:
    """The first `maps_per_gpu` map ids owned by `rank` (weak scaling: per-GPU work is fixed)."""
    if not 0 <= rank < n_gpus:
        raise ValueError("rank out of range")
    return [rank + n_gpus * j for j in range(maps_per_gpu)]


def partition_maps(map_ids, n_gpus: int) -> List[List[int]]:
    """Splits an arbitrary list of map ids into per-GPU work lists (order preserved)."""
    out: List[List[int]] = [[] for _ in range(n_gpus)]
    for m in map_ids:
        out[device_for_map(int(m), n_gpus)].append(int(m))
    return out


def block_name(shuffle_id: int, map_id: int, kind: str) -> str:
    """ShuffleDataBlockId / ShuffleIndexBlockId / ShuffleChecksumBlockId names as the reference
    writes them: reduceId is always 0 (NOOP_REDUCE_ID) and the checksum block carries NO
    algorithm suffix (S3ShuffleHelper.scala:44-51)."""
    if kind not in ("data", "index", "checksum"):
        raise ValueError(kind)
    return f"shuffle_{shuffle_id}_{map_id}_0.{kind}"


def block_path(root_dir: str, app_id: str, shuffle_id: int, map_id: int, kind: str,
               folder_prefixes: int = 10) -> str:
    """`${rootDir}${mapId % folderPrefixes}/${appId}/${shuffleId}/${blockId.name}`
    (S3ShuffleDispatcher.getPath, S3ShuffleDispatcher.scala:120-144, default layout)."""
    root = root_dir if root_dir.endswith("/") else root_dir + "/"
    return f"{root}{map_id % folder_prefixes}/{app_id}/{shuffle_id}/{block_name(shuffle_id, map_id, kind)}"
