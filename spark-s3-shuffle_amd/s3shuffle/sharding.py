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


def java_non_negative_hash(name: str) -> int:
    """JavaUtils.nonNegativeHash(String): String.hashCode in 32-bit wrap-around arithmetic, absolute value, MIN_VALUE -> 0."""
    h = 0
    for c in name:
        h = (31 * h + ord(c)) & 0xFFFFFFFF
    v = h - (1 << 32) if h & 0x80000000 else h
    return 0 if v == -(1 << 31) else abs(v)


def block_path(root_dir: str, app_id: str, shuffle_id: int, map_id: int, kind: str,
               folder_prefixes: int = 10, use_spark_shuffle_fetch: bool = False) -> str:
    """`${rootDir}${mapId % folderPrefixes}/${appId}/${shuffleId}/${blockId.name}`
    (S3ShuffleDispatcher.getPath, S3ShuffleDispatcher.scala:120-144, default layout), or with
    spark.shuffle.s3.useSparkShuffleFetch Spark's fallback-storage layout
    `${rootDir}${appId}/${shuffleId}/${nonNegativeHash(name)}/${name}` (:132-141)."""
    root = root_dir if root_dir.endswith("/") else root_dir + "/"
    name = block_name(shuffle_id, map_id, kind)
    if use_spark_shuffle_fetch:
        return f"{root}{app_id}/{shuffle_id}/{java_non_negative_hash(name)}/{name}"
    return f"{root}{map_id % folder_prefixes}/{app_id}/{shuffle_id}/{name}"
