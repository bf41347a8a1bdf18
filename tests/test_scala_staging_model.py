"""The bounded-staging algorithm of scala/.../S3GpuMapOutput.scala, restated statement by statement in Python and run with
the oracle as the codec (no JDK here, so this checks the ALGORITHM, not the Scala text): partitions are appended in
ascending order into a staging buffer of a fixed size; whenever it fills, what is staged is compressed, checksummed and
appended to the data object — at a partition boundary the flushed partitions are final, in the middle of a partition the
piece becomes one complete codec stream and the partition's checksum keeps running over the compressed pieces.

What must hold for every staging size and every shape of map output: the object decodes, partition by partition, to what
was appended (concatenated streams); `partitionLengths` are the object's partition sizes; `checksums(p)` is the checksum
of the stored bytes of partition p — so a stock reader (S3ChecksumValidationStream + the JVM codec) accepts the object."""
import zlib

import numpy as np
import pytest

LZ4, SNAPPY = 1, 2
NONE, ADLER, CRC = 0, 1, 2


class StagedMapOutput:
    """S3GpuMapOutput.scala, same names; `flush` cites its lines"""

    def __init__(self, oracle, codec, algo, num_partitions, staging_bytes):
        self.o, self.codec, self.algo = oracle, codec, algo
        self.staging = bytearray()
        self.staging_bytes = staging_bytes
        self.partition_lengths = [0] * num_partitions
        self.checksums = [0] * num_partitions
        self.group_first, self.current = 0, -1
        self.group_offsets = [0]
        self.running = None  # zlib-style running value of the split partition (None = not split)
        self.obj = bytearray()
        self.num_partitions = num_partitions
        self.flushed_once = False

    def _open_partition(self, reduce_id):  # openPartition
        assert reduce_id >= self.current
        while self.current < reduce_id:
            self.current += 1
            self.group_offsets.append(self.group_offsets[-1])

    def append(self, reduce_id, b):
        if reduce_id != self.current:
            self._open_partition(reduce_id)
        o = 0
        while o < len(b):
            if len(self.staging) >= self.staging_bytes:
                self._flush(end_of_partition=False)
            k = min(len(b) - o, self.staging_bytes - len(self.staging))
            self.staging += b[o:o + k]
            o += k
            self.group_offsets[-1] += k

    def _sum_new(self):
        return 1 if self.algo == ADLER else 0

    def _sum_update(self, v, data):
        return zlib.adler32(data, v) if self.algo == ADLER else zlib.crc32(data, v)

    def _flush(self, end_of_partition):
        n = len(self.group_offsets) - 1
        if n <= 0:
            return
        self.flushed_once = True
        offs = np.array(self.group_offsets, np.int64)
        img, index, sums = self.o.compress_map_output(self.codec, self.algo, np.frombuffer(bytes(self.staging), np.uint8), offs)
        for i in range(n):
            p = self.group_first + i
            ln = int(index[i + 1] - index[i])
            continues = i == n - 1 and not end_of_partition
            piece = continues or (i == 0 and self.running is not None)
            if piece and self.running is None and self.algo != NONE:
                self.running = self._sum_new()
            if ln > 0:
                data = img[int(index[i]):int(index[i + 1])].tobytes()
                self.obj += data
                if piece and self.running is not None:
                    self.running = self._sum_update(self.running, data)
            self.partition_lengths[p] += ln
            if not piece:
                self.checksums[p] = int(sums[i]) if self.algo != NONE else 0
            elif not continues:
                if self.running is not None:
                    self.checksums[p] = self.running
                self.running = None
        self.group_first = self.current + 1 if end_of_partition else self.current
        self.group_offsets = [0] if end_of_partition else [0, 0]
        self.staging = bytearray()

    def _flush_on_jvm(self):  # flushOnJvm: the same partition streams from the JVM codec (here: the oracle's stream writer)
        n = len(self.group_offsets) - 1
        for i in range(n):
            p = self.group_first + i
            data = bytes(self.staging[self.group_offsets[i]:self.group_offsets[i + 1]])
            v = self._sum_new() if self.algo != NONE else None
            if data:
                stored = self.o.compress_stream(self.codec, np.frombuffer(data, np.uint8)).tobytes()
                self.obj += stored
                self.partition_lengths[p] = len(stored)
                if v is not None:
                    v = self._sum_update(v, stored)
            if v is not None:
                self.checksums[p] = v
        self.group_first = self.current + 1
        self.group_offsets = [0]
        self.staging = bytearray()

    def commit(self, min_bytes=0):
        if self.current < self.num_partitions - 1:
            self._open_partition(self.num_partitions - 1)
        if not self.flushed_once and len(self.staging) < min_bytes and sum(self.partition_lengths) == 0:
            self._flush_on_jvm()
        else:
            self._flush(end_of_partition=True)
        return self.partition_lengths, self.checksums, bytes(self.obj)


@pytest.mark.parametrize("codec,algo", [(LZ4, ADLER), (LZ4, CRC), (SNAPPY, ADLER), (LZ4, NONE)])
def test_bounded_staging_produces_an_object_every_reader_accepts(oracle, codec, algo):
    import corpus

    rng = np.random.default_rng(17 + codec * 3 + algo)
    for staging_bytes in (1 << 20, 70_000, 33_000, 4096):
        sizes = [int(rng.choice([0, 0, 1, 900, 20_000, 70_000, 140_000, 33_000])) for _ in range(9)]
        sizes[4] = 0
        parts = [corpus.chunk_corpus(int(rng.integers(0, corpus.N_KINDS)), n, rng).tobytes() if n else b"" for n in sizes]
        m = StagedMapOutput(oracle, codec, algo, len(parts), staging_bytes)
        for p, b in enumerate(parts):
            if not b and p % 2:
                continue  # a partition nothing was ever written to
            for o in range(0, len(b), 10_007):  # the writer's chunks do not line up with anything
                m.append(p, b[o:o + 10_007])
        lengths, sums, obj = m.commit()
        assert sum(lengths) == len(obj)
        index = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        for p, b in enumerate(parts):
            stored = obj[int(index[p]):int(index[p + 1])]
            if algo != NONE:
                assert sums[p] == oracle.checksum(algo, np.frombuffer(stored, np.uint8)), (staging_bytes, p)
            back = oracle.decompress_stream(codec, np.frombuffer(stored, np.uint8), len(b)) if stored else np.zeros(0, np.uint8)
            assert back.tobytes() == b, (staging_bytes, p)
        # the reader's call: the whole object as one batch range, verified per partition
        ref = np.array(sums, np.int64) if algo != NONE else None
        rc, back, bad = oracle.decompress_range(codec, algo, np.frombuffer(obj, np.uint8), index, ref, sum(len(b) for b in parts))
        assert rc == 0 and bad == -1 and back.tobytes() == b"".join(parts)
        if staging_bytes >= 1 << 20:  # everything fit: the object is the single-call image
            offs = np.concatenate([[0], np.cumsum([len(b) for b in parts])]).astype(np.int64)
            img, idx, s1 = oracle.compress_map_output(codec, algo, np.frombuffer(b"".join(parts), np.uint8), offs)
            assert img.tobytes() == obj and list(idx) == list(index)


def test_a_flush_exactly_at_a_partition_boundary(oracle):
    """partition 0 fills the staging buffer exactly; the flush happens when partition 1 (already open, nothing staged yet)
    wants to append: partition 0 is final, partition 1 'continues' with an empty first piece"""
    a, b = bytes(range(256)) * 16, b"tail" * 25
    m = StagedMapOutput(oracle, LZ4, ADLER, 3, 4096)
    m.append(0, a)
    m.append(1, b)
    lengths, sums, obj = m.commit()
    index = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    rc, back, bad = oracle.decompress_range(LZ4, ADLER, np.frombuffer(obj, np.uint8), index, np.array(sums, np.int64), len(a) + len(b))
    assert rc == 0 and back.tobytes() == a + b and lengths[2] == 0 and sums[2] == 1
    assert [sums[p] for p in (0, 1)] == [oracle.checksum(ADLER, np.frombuffer(obj[int(index[p]):int(index[p + 1])], np.uint8)) for p in (0, 1)]


@pytest.mark.parametrize("codec,algo", [(LZ4, ADLER), (SNAPPY, CRC), (LZ4, NONE)])
def test_small_map_outputs_take_the_jvm_codec_and_produce_the_same_object(oracle, codec, algo):
    """spark.shuffle.s3.gpu.minBytes: a map output below it is compressed by the JVM codec partition by partition (flushOnJvm) —
    the object, the lengths and the checksums are those of the library call (the oracle's single-call image)"""
    import corpus

    rng = np.random.default_rng(23)
    parts = [corpus.chunk_corpus(7, 9000, rng).tobytes(), b"", corpus.chunk_corpus(3, 40_000, rng).tobytes(), b"x", b""]
    m = StagedMapOutput(oracle, codec, algo, len(parts), 1 << 20)
    for p, b in enumerate(parts):
        if b or p == 1:
            m.append(p, b)
    lengths, sums, obj = m.commit(min_bytes=2 << 20)
    assert not m.flushed_once
    offs = np.concatenate([[0], np.cumsum([len(b) for b in parts])]).astype(np.int64)
    img, idx, s1 = oracle.compress_map_output(codec, algo, np.frombuffer(b"".join(parts), np.uint8), offs)
    assert obj == img.tobytes() and list(np.cumsum([0] + lengths)) == list(idx)
    if algo != NONE:
        assert sums == [int(x) for x in s1]
