//
// S3GpuMapOutput — what S3ShuffleMapOutputWriter (shuffle/S3ShuffleMapOutputWriter.scala) and
// S3SingleSpillShuffleMapOutputWriter (shuffle/S3SingleSpillShuffleMapOutputWriter.scala:24-64) call when
// spark.shuffle.s3.gpu.enabled=true.  The writer keeps its SPI surface (getPartitionWriter / openStream /
// commitAllPartitions / abort); two things change inside it:
//
//   1. S3ShuffleOutputStream.write (:168-202) appends the UNCOMPRESSED serialized bytes of the partition to
//      `staging` (page-locked, from S3SCodec.hostAlloc) instead of the BufferedOutputStream, and close() records
//      the partition boundary — Spark-level shuffle compression is off for these writers, so nothing upstream
//      compressed or checksummed the bytes;
//   2. commitAllPartitions (:91-118) calls `commit` below: ONE library call produces the exact `.data` byte image,
//      the partition lengths and the per-partition checksums; the object, the `.index` and the `.checksum` are then
//      written through the unchanged S3ShuffleHelper / dispatcher code, so the store layout is untouched.
//
// Patch to S3ShuffleMapOutputWriter (sketch of the three touched places, everything else unchanged):
//
//   private val gpu = if (dispatcher.gpuEnabled && S3SCodec.supports(dispatcher.compressionCodecShortName))
//                       new S3GpuMapOutput(shuffleId, mapId, numPartitions) else null   // zstd / lzf: JVM codecs as before
//   // S3ShuffleOutputStream.write(b, off, len):   if (gpu != null) gpu.append(reduceId, b, off, len) else bufferedStream.write(...)
//   // commitAllPartitions(checksums):             if (gpu != null) return gpu.commit(createBlock = () => dispatcher.createBlock(shuffleBlock))
//
// NOT COMPILED IN THIS IMAGE (no JDK / scalac).
//
package org.apache.spark.shuffle.gpu

import java.io.OutputStream
import java.nio.ByteBuffer

import org.apache.spark.shuffle.api.metadata.MapOutputCommitMessage
import org.apache.spark.shuffle.helper.{S3ShuffleDispatcher, S3ShuffleHelper}

class S3GpuMapOutput(shuffleId: Int, mapId: Long, numPartitions: Int) {
  private val dispatcher = S3ShuffleDispatcher.get
  private val device = S3SCodec.deviceFor(mapId, S3SCodec.deviceCount())
  private val ctx = S3SCodec.forThread(device)
  private val codec = S3SCodec.codecId(dispatcher.compressionCodecShortName)
  private val algo = S3SCodec.checksumId(dispatcher.checksumEnabled, dispatcher.checksumAlgorithm)

  private var staging: ByteBuffer = S3GpuBuffers.take(S3GpuBuffers.lastMapOutputSize)
  private val srcOffsets = new Array[Long](numPartitions + 1) // cumulative, srcOffsets(0) = 0
  private var lastPartition = -1

  /** Bytes of partition `reduceId` (ascending ids only — same precondition as getPartitionWriter, :67-73). */
  def append(reduceId: Int, b: Array[Byte], off: Int, len: Int): Unit = {
    if (reduceId < lastPartition)
      throw new RuntimeException("Precondition: Expect a monotonically increasing reducePartitionId.")
    while (lastPartition < reduceId) { lastPartition += 1; srcOffsets(lastPartition + 1) = srcOffsets(lastPartition) }
    if (staging.remaining() < len) staging = S3GpuBuffers.grow(staging, staging.position() + len)
    staging.put(b, off, len)
    srcOffsets(reduceId + 1) += len
  }

  /** commitAllPartitions: compress + checksum on the GPU, then the reference's own store writes. */
  def commit(createBlock: () => OutputStream): MapOutputCommitMessage = {
    while (lastPartition < numPartitions - 1) { lastPartition += 1; srcOffsets(lastPartition + 1) = srcOffsets(lastPartition) }
    val cap = S3SCodec.maxCompressedSize(ctx, codec, srcOffsets, numPartitions)
    val out = S3GpuBuffers.take(cap)
    val index = new Array[Long](numPartitions + 1)
    val sums = new Array[Long](math.max(numPartitions, 1))
    val total = new Array[Long](1)
    try {
      val rc = S3SCodec.compressMapOutput(ctx, codec, algo, staging, srcOffsets, numPartitions, out, cap, index,
        if (algo == S3SCodec.CHECKSUM_NONE) null else sums, total)
      S3SCodec.check(ctx, rc, s"shuffle_${shuffleId}_${mapId}_0.data")
      val partitionLengths = Array.tabulate(numPartitions)(p => index(p + 1) - index(p))
      if (total(0) > 0) { // the data block is opened lazily, like initStream() (:43-49)
        val stream = createBlock()
        try S3GpuBuffers.writeTo(stream, out, total(0)) finally stream.close()
      }
      // emission rule and order of commitAllPartitions (:111-115): index, then checksum, iff bytes or alwaysCreateIndex
      if (partitionLengths.sum > 0 || dispatcher.alwaysCreateIndex) {
        S3ShuffleHelper.writePartitionLengths(shuffleId, mapId, partitionLengths)
        if (dispatcher.checksumEnabled) S3ShuffleHelper.writeChecksum(shuffleId, mapId, sums.take(numPartitions))
      }
      S3GpuBuffers.lastMapOutputSize = staging.position()
      MapOutputCommitMessage.of(partitionLengths)
    } finally {
      S3GpuBuffers.give(out)
      S3GpuBuffers.give(staging)
    }
  }

  def abort(): Unit = S3GpuBuffers.give(staging)
}

/** Process-wide cache of page-locked direct buffers (pinning pages costs ~100 ms per GiB: never per task). */
object S3GpuBuffers {
  @volatile var lastMapOutputSize: Long = 8L << 20 // first guess = the reference's 8 MiB write buffer (S3ShuffleDispatcher.scala:55)
  private val free = new java.util.concurrent.ConcurrentLinkedDeque[ByteBuffer]()

  def take(atLeast: Long): ByteBuffer = {
    val it = free.iterator()
    while (it.hasNext) { val b = it.next(); if (b.capacity() >= atLeast && free.remove(b)) { b.clear(); return b } }
    val b = S3SCodec.hostAlloc(math.max(atLeast, 1L << 20))
    if (b == null) throw new OutOfMemoryError(s"s3s_host_alloc($atLeast)")
    b
  }
  def give(b: ByteBuffer): Unit = if (b != null) free.offerFirst(b)
  def grow(b: ByteBuffer, atLeast: Long): ByteBuffer = {
    val n = take(math.max(atLeast, 2L * b.capacity()))
    b.flip(); n.put(b); give(b); n
  }
  def writeTo(s: OutputStream, b: ByteBuffer, n: Long): Unit = {
    val chunk = new Array[Byte](1 << 20)
    b.position(0)
    var left = n
    while (left > 0) { val k = math.min(left, chunk.length).toInt; b.get(chunk, 0, k); s.write(chunk, 0, k); left -= k }
  }
}
