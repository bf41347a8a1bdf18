//
// S3GpuBlockDecoder — what S3ShuffleReader.read (storage/S3ShuffleReader.scala:98-110) calls instead of
//   new S3ChecksumValidationStream(blockId, stream, algo)  +  serializerManager.wrapStream(blockId, …)
// when the reduce side of the GPU path is on (dispatcher.gpuReadEnabled, scala/patches/0001-gpu-codec.patch: with GPU writers,
// or on its own over the objects of JVM writers - lz4 of any block size, snappy, zstd): the prefetched block range (one
// ShuffleBlockId or one ShuffleBlockBatchId = several contiguous partitions of one map output,
// S3ShuffleBlockIterator.scala:37-42) is verified per partition against the `.checksum` object and decoded in ONE
// library call; the deserializer then reads plain bytes.
//
// A range that does not fit the GPU path — above 1 GiB compressed or decoded (one direct ByteBuffer has Int
// positions), below spark.shuffle.s3.gpu.minBytes, or written with a codec the library does not have — takes the
// reference's JVM stack (`jvmPath`): checksum validation stream + the codec's own input stream.  The objects a GPU
// writer produced are ordinary LZ4Block / SnappyOutputStream streams, so either side may be the JVM.
//
// NOT COMPILED IN THIS IMAGE (no JDK / scalac).
//
package org.apache.spark.shuffle.gpu

import java.io.InputStream
import java.nio.ByteBuffer
import java.util.concurrent.atomic.AtomicBoolean

import org.apache.spark.shuffle.helper.{S3ShuffleDispatcher, S3ShuffleHelper}
import org.apache.spark.storage.{BlockId, ShuffleBlockBatchId, ShuffleBlockId}

object S3GpuBlockDecoder {
  private def range(blockId: BlockId): (Int, Long, Int, Int) = blockId match {
    case ShuffleBlockId(s, m, r) => (s, m, r, r + 1)
    case ShuffleBlockBatchId(s, m, r0, r1) => (s, m, r0, r1)
    case other => throw new IllegalArgumentException(s"unexpected block $other")
  }

  /** Compressed length of the range from the cached `.index` (S3ShuffleHelper.scala:76-81). */
  def compressedLength(blockId: BlockId): Long = {
    val (shuffleId, mapId, r0, r1) = range(blockId)
    val lengths = S3ShuffleHelper.getPartitionLengths(shuffleId, mapId)
    lengths(r1) - lengths(r0)
  }

  /** True when `decode` should take this block (the reader keeps the JVM stack otherwise).  The codec is the one the stored
    * objects carry (dispatcher.gpuReadCodec: spark.shuffle.s3.gpu.codec for GPU writers, spark.io.compression.codec for JVM
    * writers).  Zstandard: one wavefront decodes one partition's frame at ~10 MB/s, so the library only pays when a range
    * holds MANY small frames - a batch range of at least spark.shuffle.s3.gpu.zstd.minPartitions partitions whose mean size
    * is at most spark.shuffle.s3.gpu.zstd.maxFrameBytes; everything else stays with zstd-jni on the task thread. */
  def accepts(blockId: BlockId): Boolean = {
    val d = S3ShuffleDispatcher.get
    d.gpuReadEnabled && S3SCodec.supportsDecode(d.gpuReadCodec) && {
      val n = compressedLength(blockId)
      val (_, _, r0, r1) = range(blockId)
      val zstdOk = d.gpuReadCodec != "zstd" ||
        (r1 - r0 >= d.gpuZstdMinPartitions && n / math.max(r1 - r0, 1) <= d.gpuZstdMaxFrameBytes)
      n >= d.gpuMinBytes && n <= S3GpuBuffers.MaxBuffer && zstdOk
    }
  }

  /** `jvmPath(in)` = the reference's stream stack for the same block over `in` (used when the decoded size turns out
    * to be above one buffer: the compressed bytes are already staged, nothing is fetched twice). */
  def decode(blockId: BlockId, stream: InputStream, jvmPath: InputStream => InputStream): InputStream = {
    val dispatcher = S3ShuffleDispatcher.get
    val (shuffleId, mapId, r0, r1) = range(blockId)
    val ctx = S3SCodec.forThread(S3SCodec.deviceFor(mapId, S3SCodec.devices()))
    val codec = S3SCodec.decodeCodecId(dispatcher.gpuReadCodec)
    val algo = S3SCodec.checksumId(dispatcher.checksumEnabled, dispatcher.checksumAlgorithm)
    // cumulative `.index` of the map output, relative to the range
    val lengths = S3ShuffleHelper.getPartitionLengths(shuffleId, mapId)
    val rel = Array.tabulate(r1 - r0 + 1)(i => lengths(r0 + i) - lengths(r0))
    val refs = if (algo == S3SCodec.CHECKSUM_NONE) null else S3ShuffleHelper.getChecksums(shuffleId, mapId).slice(r0, r1)
    val compLen = rel(r1 - r0)
    val comp = S3GpuBuffers.take(compLen)
    var compOwned = true
    try {
      S3GpuStreams.readFully(stream, comp, compLen) // the prefetcher's buffer -> page-locked staging
      val outLen = new Array[Long](1)
      S3SCodec.check(ctx, S3SCodec.decompressedSize(ctx, codec, comp, compLen, outLen), blockId.name)
      if (outLen(0) > S3GpuBuffers.MaxBuffer) { // decoded range above one buffer: the JVM codecs stream it
        compOwned = false
        return jvmPath(new S3GpuStreams.DirectBufferInputStream(comp, compLen))
      }
      val out = S3GpuBuffers.take(outLen(0))
      val bad = Array(-1)
      val rc =
        try S3SCodec.decompressRange(ctx, codec, algo, comp, compLen, rel, refs, r1 - r0, out, outLen(0), outLen, bad)
        catch { case t: Throwable => S3GpuBuffers.give(out); throw t }
      if (rc != S3SCodec.OK) S3GpuBuffers.give(out)
      S3SCodec.check(ctx, rc, blockId.name, if (bad(0) >= 0) r0 + bad(0) else -1)
      new S3GpuStreams.DirectBufferInputStream(out, outLen(0)) // gives `out` back to S3GpuBuffers on the first close()
    } finally {
      if (compOwned) S3GpuBuffers.give(comp)
      stream.close()
    }
  }
}

object S3GpuStreams {
  def readFully(in: InputStream, dst: ByteBuffer, n: Long): Unit = {
    val chunk = new Array[Byte](1 << 20)
    dst.clear()
    var left = n
    while (left > 0) {
      val k = in.read(chunk, 0, math.min(left, chunk.length.toLong).toInt)
      if (k < 0) throw new java.io.EOFException(s"block ended $left bytes early")
      dst.put(chunk, 0, k); left -= k
    }
  }

  /** Reads a pooled page-locked buffer; close() is idempotent (Spark closes shuffle streams more than once) and
    * returns the buffer to the pool exactly once. */
  final class DirectBufferInputStream(buf: ByteBuffer, n: Long) extends InputStream {
    require(n <= buf.capacity() && n <= Int.MaxValue)
    private val view = buf.duplicate()
    view.position(0); view.limit(n.toInt)
    private val closed = new AtomicBoolean(false)
    private def live: Boolean = !closed.get()
    override def read(): Int = if (live && view.hasRemaining) view.get() & 0xff else -1
    override def read(b: Array[Byte], off: Int, len: Int): Int =
      if (len == 0) 0
      else if (!live || !view.hasRemaining) -1
      else { val k = math.min(len, view.remaining()); view.get(b, off, k); k }
    override def available(): Int = if (live) view.remaining() else 0
    override def close(): Unit = if (closed.compareAndSet(false, true)) S3GpuBuffers.give(buf)
  }
}
