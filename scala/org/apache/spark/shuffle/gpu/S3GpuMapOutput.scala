//
// S3GpuMapOutput — what S3ShuffleMapOutputWriter (shuffle/S3ShuffleMapOutputWriter.scala) and
// S3SingleSpillShuffleMapOutputWriter (shuffle/S3SingleSpillShuffleMapOutputWriter.scala:24-64) call when
// spark.shuffle.s3.gpu.enabled=true (scala/patches/0001-gpu-codec.patch is the edit of those two files, of
// S3ShuffleReader and of S3ShuffleDispatcher).  The writer keeps its SPI surface (getPartitionWriter / openStream /
// commitAllPartitions / abort); two things change inside it:
//
//   1. S3ShuffleOutputStream.write (:168-202) appends the UNCOMPRESSED serialized bytes of the partition to
//      `staging` (page-locked, from S3SCodec.hostAlloc) instead of the BufferedOutputStream — the executor runs with
//      spark.shuffle.compress=false, so nothing upstream compressed the bytes;
//   2. commitAllPartitions (:91-118) calls `commit` below: the library produces the exact `.data` byte image, the
//      partition lengths and the per-partition checksums; the object, the `.index` and the `.checksum` are then
//      written through the unchanged S3ShuffleHelper / dispatcher code, so the store layout is untouched.
//
// Sizes: a direct ByteBuffer holds at most 2^31-1 bytes, a map output does not have to.  Staging is therefore a
// bounded buffer (spark.shuffle.s3.gpu.stagingBytes, at most 1 GiB) that is FLUSHED — compressed, checksummed and
// appended to the data object — whenever it fills: at a partition boundary the flushed partitions are final; in the
// middle of a partition the piece becomes one complete codec stream (all Spark codecs on this path support
// concatenated streams, which is what their fast spill merge relies on), and the partition's checksum is kept
// running on the JVM over the compressed pieces.  All positions are Long.
//
// NOT COMPILED IN THIS IMAGE (no JDK / scalac).
//
package org.apache.spark.shuffle.gpu

import java.io.{File, FileInputStream, OutputStream}
import java.nio.ByteBuffer
import java.util.zip.{Adler32, CRC32, Checksum}

import org.apache.spark.shuffle.api.metadata.MapOutputCommitMessage
import org.apache.spark.shuffle.helper.{S3ShuffleDispatcher, S3ShuffleHelper}

class S3GpuMapOutput(shuffleId: Int, mapId: Long, numPartitions: Int, createBlock: () => OutputStream) {
  private val dispatcher = S3ShuffleDispatcher.get
  private val device = S3SCodec.deviceFor(mapId, S3SCodec.devices())
  private val ctx = S3SCodec.forThread(device)
  private val codec = S3SCodec.codecId(dispatcher.gpuCodec)
  private val algo = S3SCodec.checksumId(dispatcher.checksumEnabled, dispatcher.checksumAlgorithm)

  private val stagingBytes: Long = math.min(dispatcher.gpuStagingBytes, S3GpuBuffers.MaxBuffer)
  private var staging: ByteBuffer = S3GpuBuffers.take(math.min(S3GpuBuffers.lastMapOutputSize, stagingBytes))

  // results, per partition of the map output
  private val partitionLengths = new Array[Long](numPartitions) // compressed bytes
  private val checksums = new Array[Long](numPartitions)
  // the staged group: partitions groupFirst .. current, cumulative offsets inside `staging`
  private var groupFirst = 0
  private var current = -1
  private val groupOffsets = new scala.collection.mutable.ArrayBuffer[Long]() += 0L
  // a partition that was flushed in pieces keeps its checksum running here (compressed bytes, in order)
  private var running: Checksum = null // non-null <=> partition groupFirst already has bytes in the object
  private var stream: OutputStream = null // the data block, opened lazily like initStream() (:43-49)
  private var uncompressedTotal = 0L
  private var flushedOnce = false // something of this map output already went through the library

  def bytesStaged: Long = uncompressedTotal

  /** Opens partitions up to `reduceId`; the ones in between stay empty (0 bytes in the object). */
  private def openPartition(reduceId: Int): Unit = {
    if (reduceId < current)
      throw new RuntimeException("Precondition: Expect a monotonically increasing reducePartitionId.")
    while (current < reduceId) {
      current += 1
      groupOffsets += groupOffsets.last // (groupOffsets.length == current - groupFirst + 2)
    }
  }

  /** Bytes of partition `reduceId` (ascending ids only — same precondition as getPartitionWriter, :67-73). */
  def append(reduceId: Int, b: Array[Byte], off: Int, len: Int): Unit = {
    if (reduceId != current) openPartition(reduceId)
    var o = off
    var left = len
    while (left > 0) {
      if (!staging.hasRemaining) {
        if (staging.capacity() < stagingBytes) staging = S3GpuBuffers.grow(staging, math.min(2L * staging.capacity(), stagingBytes))
        else flush(endOfPartition = false)
      }
      val k = math.min(left, staging.remaining())
      staging.put(b, o, k)
      o += k; left -= k
      groupOffsets(groupOffsets.length - 1) += k
      uncompressedTotal += k
    }
  }

  private val one = new Array[Byte](1)
  def append(reduceId: Int, b: Int): Unit = { one(0) = b.toByte; append(reduceId, one, 0, 1) }

  /** S3SingleSpillShuffleMapOutputWriter.transferMapSpillFile: the spill file holds the partitions back to back. */
  def appendFile(spill: File, uncompressedLengths: Array[Long]): Unit = {
    val in = new FileInputStream(spill)
    val chunk = new Array[Byte](1 << 20)
    try {
      var p = 0
      while (p < uncompressedLengths.length) {
        var left = uncompressedLengths(p)
        while (left > 0) {
          val k = in.read(chunk, 0, math.min(left, chunk.length.toLong).toInt)
          if (k < 0) throw new java.io.EOFException(s"${spill.getName} ended $left bytes early")
          append(p, chunk, 0, k)
          left -= k
        }
        p += 1
      }
    } finally in.close()
  }

  /** Compress + checksum what is staged and append it to the data block.  `endOfPartition = false`: the last staged
    * partition continues after the flush (its piece is a complete stream; the checksum keeps running). */
  private def flush(endOfPartition: Boolean): Unit = {
    val n = groupOffsets.length - 1 // staged partitions groupFirst .. current (the last one possibly partial)
    if (n <= 0) return
    flushedOnce = true
    val offs = groupOffsets.toArray
    val cap = S3SCodec.maxCompressedSize(ctx, codec, offs, n)
    val out = S3GpuBuffers.take(cap)
    val index = new Array[Long](n + 1)
    val sums = new Array[Long](n)
    val total = new Array[Long](1)
    try {
      val what = s"shuffle_${shuffleId}_${mapId}_0.data"
      if (dispatcher.gpuCommitBatch > 1) { // commits of concurrent tasks share one pipelined call (S3GpuCommitQueue)
        val r = S3GpuCommitQueue.compress(device, new S3GpuCommitQueue.Request(codec, algo, staging, offs, out, cap, index, sums))
        if (r.rc != S3SCodec.OK) S3SCodec.raise(r.rc, r.error, what)
        total(0) = r.total
      } else {
        val rc = S3SCodec.compressMapOutput(ctx, codec, algo, staging, offs, n, out, cap, index,
          if (algo == S3SCodec.CHECKSUM_NONE) null else sums, total)
        S3SCodec.check(ctx, rc, what)
      }
      var i = 0
      while (i < n) { // per partition: bytes, length, checksum (a split partition's checksum runs on the JVM)
        val p = groupFirst + i
        val len = index(i + 1) - index(i)
        val continues = i == n - 1 && !endOfPartition // more bytes of p follow this flush
        val piece = continues || (i == 0 && running != null) // p is written in more than one piece
        if (piece && running == null && algo != S3SCodec.CHECKSUM_NONE)
          running = S3GpuMapOutput.newChecksum(algo)
        if (len > 0) {
          if (stream == null) stream = createBlock()
          S3GpuBuffers.writeTo(stream, out, index(i), len, if (piece) running else null)
        }
        partitionLengths(p) += len
        if (!piece) checksums(p) = sums(i)
        else if (!continues) {
          if (running != null) checksums(p) = running.getValue
          running = null
        }
        i += 1
      }
    } finally S3GpuBuffers.give(out)
    // the next group starts with the partition that continues, or behind the last complete one
    groupFirst = if (endOfPartition) current + 1 else current
    groupOffsets.clear(); groupOffsets += 0L
    if (!endOfPartition) groupOffsets += 0L
    staging.clear()
  }

  /** A map output below spark.shuffle.s3.gpu.minBytes is not worth a library call (one 32 KiB block is a ~1 ms serial chain
    * on a wavefront: the call has that floor): the same partition streams come from the JVM codec — LZ4 objects are
    * byte-identical either way, Snappy objects decode the same.  Only for an output that was never flushed before. */
  private def flushOnJvm(): Unit = {
    val jvmCodec = org.apache.spark.io.CompressionCodec.createCodec(org.apache.spark.SparkEnv.get.conf, dispatcher.gpuCodec)
    val n = groupOffsets.length - 1
    var i = 0
    while (i < n) {
      val p = groupFirst + i
      val from = groupOffsets(i)
      val len = groupOffsets(i + 1) - from
      val sum: Checksum =
        S3GpuMapOutput.newChecksum(algo)
      if (len > 0) { // (an empty partition is 0 bytes in the object, like S3ShuffleMapOutputWriter's untouched partition writer)
        if (stream == null) stream = createBlock()
        val counted = new S3GpuBuffers.CountingStream(stream, sum) // (close() stops at it: the data block stays open)
        val out = jvmCodec.compressedOutputStream(counted)
        S3GpuBuffers.writeTo(out, staging, from, len, null)
        out.close()
        partitionLengths(p) = counted.count
      }
      if (sum != null) checksums(p) = sum.getValue // (of no bytes for an empty partition: Adler32 1, CRC32 0 — as the library answers)
      i += 1
    }
    groupFirst = current + 1
    groupOffsets.clear(); groupOffsets += 0L
    staging.clear()
  }

  /** commitAllPartitions: compress + checksum on the GPU, then the reference's own store writes. */
  def commit(): MapOutputCommitMessage = {
    try {
      if (current < numPartitions - 1) openPartition(numPartitions - 1)
      if (!flushedOnce && uncompressedTotal < dispatcher.gpuMinBytes) flushOnJvm()
      else flush(endOfPartition = true)
      if (stream != null) { stream.close(); stream = null }
      // emission rule and order of commitAllPartitions (:111-115): index, then checksum, iff bytes or alwaysCreateIndex
      if (partitionLengths.sum > 0 || dispatcher.alwaysCreateIndex) {
        S3ShuffleHelper.writePartitionLengths(shuffleId, mapId, partitionLengths)
        if (dispatcher.checksumEnabled) S3ShuffleHelper.writeChecksum(shuffleId, mapId, checksums)
      }
      S3GpuBuffers.lastMapOutputSize = math.max(uncompressedTotal, 1L << 20)
      MapOutputCommitMessage.of(partitionLengths)
    } finally release()
  }

  def abort(): Unit = {
    try if (stream != null) stream.close() finally release()
  }

  private def release(): Unit = { S3GpuBuffers.give(staging); staging = null }
}

/** Process-wide cache of page-locked direct buffers (pinning pages costs ~100 ms per GiB: never per task).  Bounded:
  * what does not fit under spark.shuffle.s3.gpu.pinnedPoolBytes goes back to the driver (s3s_host_free). */
object S3GpuMapOutput {
  /** The JVM twin of the library's checksum (a partition written in more than one flush; the JVM-codec fallback). */
  def newChecksum(algo: Int): Checksum = algo match {
    case S3SCodec.CHECKSUM_ADLER32 => new Adler32()
    case S3SCodec.CHECKSUM_CRC32 => new CRC32()
    case S3SCodec.CHECKSUM_CRC32C => new java.util.zip.CRC32C() // (Java 9+; Spark 4 requires 17)
    case _ => null
  }
}

object S3GpuBuffers {
  val MaxBuffer: Long = 1L << 30 // one direct ByteBuffer: Int positions
  @volatile var lastMapOutputSize: Long = 8L << 20 // first guess = the reference's 8 MiB write buffer (S3ShuffleDispatcher.scala:55)
  private val free = new java.util.ArrayDeque[ByteBuffer]()
  private var pooledBytes = 0L
  private def poolLimit: Long = S3ShuffleDispatcher.get.gpuPinnedPoolBytes

  def take(atLeast: Long): ByteBuffer = {
    if (atLeast > Int.MaxValue) throw new IllegalArgumentException(s"pinned buffer of $atLeast bytes: callers chunk above 2 GiB")
    free.synchronized {
      val it = free.iterator()
      while (it.hasNext) {
        val b = it.next()
        if (b.capacity() >= atLeast) { it.remove(); pooledBytes -= b.capacity(); b.clear(); return b }
      }
    }
    val b = S3SCodec.hostAlloc(math.max(atLeast, 1L << 20))
    if (b == null) throw new OutOfMemoryError(s"s3s_host_alloc($atLeast)")
    b
  }

  /** Idempotence is the caller's job (DirectBufferInputStream.close guards it); a buffer is pooled at most once. */
  def give(b: ByteBuffer): Unit = if (b != null) {
    val keep = free.synchronized {
      val dup = { val it = free.iterator(); var f = false; while (it.hasNext) f |= (it.next() eq b); f }
      if (dup) true
      else if (pooledBytes + b.capacity() <= poolLimit) { free.addFirst(b); pooledBytes += b.capacity(); true }
      else false
    }
    if (!keep) S3SCodec.hostFree(b)
  }

  def grow(b: ByteBuffer, atLeast: Long): ByteBuffer = {
    val n = take(atLeast)
    b.flip(); n.put(b); give(b); n
  }

  /** Counts (and optionally checksums) what passes through; close() flushes and stops here — the codec streams close
    * what they wrap, the data block must stay open for the next partition. */
  final class CountingStream(under: OutputStream, sum: Checksum) extends OutputStream {
    var count = 0L
    override def write(b: Int): Unit = { under.write(b); if (sum != null) sum.update(b); count += 1 }
    override def write(b: Array[Byte], off: Int, len: Int): Unit = {
      under.write(b, off, len); if (sum != null) sum.update(b, off, len); count += len
    }
    override def flush(): Unit = under.flush()
    override def close(): Unit = under.flush()
  }

  /** out[from, from+len) -> s in 1 MiB pieces; `sum` (optional) sees the same bytes in the same order. */
  def writeTo(s: OutputStream, out: ByteBuffer, from: Long, len: Long, sum: Checksum): Unit = {
    val chunk = new Array[Byte](1 << 20)
    val view = out.duplicate()
    view.limit((from + len).toInt); view.position(from.toInt) // (one buffer never exceeds MaxBuffer)
    while (view.hasRemaining) {
      val k = math.min(view.remaining(), chunk.length)
      view.get(chunk, 0, k)
      if (sum != null) sum.update(chunk, 0, k)
      s.write(chunk, 0, k)
    }
  }
}
