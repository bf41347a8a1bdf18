//
// S3GpuBlockDecoder — what S3ShuffleReader.read (storage/S3ShuffleReader.scala:98-110) calls instead of
//   new S3ChecksumValidationStream(blockId, stream, algo)  +  serializerManager.wrapStream(blockId, …)
// when spark.shuffle.s3.gpu.enabled=true: the prefetched block range (one ShuffleBlockId or one
// ShuffleBlockBatchId = several contiguous partitions of one map output, S3ShuffleBlockIterator.scala:37-42) is
// verified per partition against the `.checksum` object and decoded in ONE library call; the deserializer then
// reads plain bytes.
//
// Patch to S3ShuffleReader.read (the flatMap at :98-110):
//
//   .flatMap { case (blockId, stream) =>
//     val in = if (dispatcher.gpuEnabled && S3SCodec.supports(dispatcher.compressionCodecShortName) &&
//                  stream.maxBytes >= dispatcher.gpuMinBytes)
//                S3GpuBlockDecoder.decode(blockId, stream)                 // <- this file
//              else serializerManager.wrapStream(blockId, checked(stream))  // unchanged JVM path
//     serializerInstance.deserializeStream(in).asKeyValueIterator
//   }
//
// NOT COMPILED IN THIS IMAGE (no JDK / scalac).
//
package org.apache.spark.shuffle.gpu

import java.io.InputStream
import java.nio.ByteBuffer

import org.apache.spark.shuffle.helper.{S3ShuffleDispatcher, S3ShuffleHelper}
import org.apache.spark.storage.{BlockId, S3ShuffleBlockStream, ShuffleBlockBatchId, ShuffleBlockId}

object S3GpuBlockDecoder {
  private def range(blockId: BlockId): (Int, Long, Int, Int) = blockId match {
    case ShuffleBlockId(s, m, r) => (s, m, r, r + 1)
    case ShuffleBlockBatchId(s, m, r0, r1) => (s, m, r0, r1)
    case other => throw new IllegalArgumentException(s"unexpected block $other")
  }

  def decode(blockId: BlockId, stream: S3ShuffleBlockStream): InputStream = {
    val dispatcher = S3ShuffleDispatcher.get
    val (shuffleId, mapId, r0, r1) = range(blockId)
    val ctx = S3SCodec.forThread(S3SCodec.deviceFor(mapId, S3SCodec.deviceCount()))
    val codec = S3SCodec.codecId(dispatcher.compressionCodecShortName)
    val algo = S3SCodec.checksumId(dispatcher.checksumEnabled, dispatcher.checksumAlgorithm)
    // cumulative `.index` of the map output (cached by the helper, S3ShuffleHelper.scala:76-81), relative to the range
    val lengths = S3ShuffleHelper.getPartitionLengths(shuffleId, mapId)
    val rel = Array.tabulate(r1 - r0 + 1)(i => lengths(r0 + i) - lengths(r0))
    val refs = if (algo == S3SCodec.CHECKSUM_NONE) null else S3ShuffleHelper.getChecksums(shuffleId, mapId).slice(r0, r1)
    val compLen = stream.maxBytes
    val comp = S3GpuBuffers.take(compLen)
    try {
      S3GpuStreams.readFully(stream, comp, compLen) // the prefetcher's buffer -> page-locked staging
      val outLen = new Array[Long](1)
      S3SCodec.check(ctx, S3SCodec.decompressedSize(ctx, codec, comp, compLen, outLen), blockId.name)
      val out = S3GpuBuffers.take(outLen(0))
      val bad = Array(-1)
      val rc = S3SCodec.decompressRange(ctx, codec, algo, comp, compLen, rel, refs, r1 - r0, out, outLen(0), outLen, bad)
      if (rc != S3SCodec.OK) S3GpuBuffers.give(out)
      S3SCodec.check(ctx, rc, blockId.name, if (bad(0) >= 0) r0 + bad(0) else -1)
      new S3GpuStreams.DirectBufferInputStream(out, outLen(0)) // gives `out` back to S3GpuBuffers on close()
    } finally {
      S3GpuBuffers.give(comp)
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
      val k = in.read(chunk, 0, math.min(left, chunk.length).toInt)
      if (k < 0) throw new java.io.EOFException(s"block ended $left bytes early")
      dst.put(chunk, 0, k); left -= k
    }
  }

  final class DirectBufferInputStream(buf: ByteBuffer, n: Long) extends InputStream {
    buf.position(0); buf.limit(n.toInt)
    override def read(): Int = if (buf.hasRemaining) buf.get() & 0xff else -1
    override def read(b: Array[Byte], off: Int, len: Int): Int =
      if (!buf.hasRemaining) -1 else { val k = math.min(len, buf.remaining()); buf.get(b, off, k); k }
    override def available(): Int = buf.remaining()
    override def close(): Unit = S3GpuBuffers.give(buf)
  }
}
