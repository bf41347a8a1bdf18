//
// S3SCodec — JVM side of the JNI binding (jni/s3s_jni.c) to include/s3shuffle_codec.h.
//
// One native context (HIP stream + device workspace + page-locked staging) per task thread, created lazily on the
// device `mapId % nGpu` (the rule the plugin already uses for folder prefixes, S3ShuffleDispatcher.scala:142-143).
// Return codes are turned into the exceptions the reference raises today, so the callers in S3GpuMapOutput.scala /
// S3GpuBlockDecoder.scala behave like the code they replace:
//   S3S_E_INVALID    RuntimeException            (S3ShuffleMapOutputWriter.scala:69-73, 96-100 preconditions)
//   S3S_E_CHECKSUM   SparkException("Invalid checksum detected for …")   (S3ChecksumValidationStream.scala:72-74)
//   S3S_E_BAD_FRAME  IOException("Stream is corrupted")                   ([EXT] LZ4BlockInputStream.refill)
//   anything else    IOException(lastError) -> Spark task retry
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no JDK / scalac): kept as source so that the binding is reviewable and
// `tests/test_jni_shim.py` can check that every @native method below has its C counterpart with the same arity.
//
package org.apache.spark.shuffle.gpu

import java.io.IOException
import java.nio.ByteBuffer

import org.apache.spark.SparkException

object S3SCodec {
  // ---- constants of include/s3shuffle_codec.h ----------------------------------------------------------------
  val CODEC_NONE = 0; val CODEC_LZ4 = 1; val CODEC_SNAPPY = 2
  val CODEC_ZSTD = 3 // reduce side only (decompressRange*, decompressedSize): S3GpuBlockDecoder takes ranges of many small frames, INTEGRATION.md
  val CODEC_LZF = 4 // reduce side only: LZFCompressionCodec streams (compress-lzf chunks around liblzf blocks)
  val CHECKSUM_NONE = 0; val CHECKSUM_ADLER32 = 1; val CHECKSUM_CRC32 = 2; val CHECKSUM_CRC32C = 3
  val OK = 0; val E_INVALID = -1; val E_CAPACITY = -2; val E_BAD_FRAME = -3; val E_CHECKSUM = -4; val E_HIP = -5
  val STATUS_NOT_RUN = -100 // per-entry status of a batch call that failed before this entry had a verdict: the call's return code is its error
  val OPT_LZ4_BLOCK_SIZE = 1; val OPT_SNAPPY_BLOCK_SIZE = 2
  val ABI_VERSION = 8

  // ---- native entry points (jni/s3s_jni.c, one line each) -------------------------------------------------------
  @native def abiVersion(): Int
  @native def deviceCount(): Int
  @native def create(device: Int, scratchBytes: Long): Long
  @native def destroy(handle: Long): Unit
  @native def setOption(handle: Long, key: Int, value: Long): Int
  @native def getOption(handle: Long, key: Int): Long
  @native def lastError(handle: Long): String
  @native def hostAlloc(bytes: Long): ByteBuffer
  @native def hostFree(buffer: ByteBuffer): Unit
  @native def maxCompressedSize(handle: Long, codec: Int, srcOffsets: Array[Long], n: Int): Long
  @native def decompressedSize(handle: Long, codec: Int, comp: ByteBuffer, compLen: Long, outLen: Array[Long]): Int
  @native def compressMapOutput(handle: Long, codec: Int, algo: Int, src: ByteBuffer, srcOffsets: Array[Long], n: Int,
                                dst: ByteBuffer, dstCap: Long, outIndex: Array[Long], outChecksums: Array[Long],
                                outTotal: Array[Long]): Int
  @native def compressMapOutputSegments(handle: Long, codec: Int, algo: Int, src: ByteBuffer, segOffsets: Array[Long],
                                        nSegs: Int, partFirstSeg: Array[Int], n: Int, dst: ByteBuffer, dstCap: Long,
                                        outIndex: Array[Long], outChecksums: Array[Long], outTotal: Array[Long]): Int
  @native def checksumRanges(handle: Long, algo: Int, data: ByteBuffer, offsets: Array[Long], n: Int,
                             out: Array[Long]): Int
  @native def decompressRange(handle: Long, codec: Int, algo: Int, comp: ByteBuffer, compLen: Long,
                              partOffsets: Array[Long], refChecksums: Array[Long], nparts: Int, dst: ByteBuffer,
                              dstCap: Long, outLen: Array[Long], outBadPartition: Array[Int]): Int

  // batched forms over host buffers (one entry per task / range; the library pipelines upload, codec and download)
  @native def compressMapOutputsBatch(handle: Long, codec: Int, algo: Int, src: Array[ByteBuffer],
                                      srcOffsets: Array[Array[Long]], dst: Array[ByteBuffer], dstCap: Array[Long],
                                      outIndex: Array[Array[Long]], outChecksums: Array[Array[Long]],
                                      outTotal: Array[Long], outStatus: Array[Int]): Int
  @native def decompressRangesBatch(handle: Long, codec: Int, algo: Int, comp: Array[ByteBuffer], compLen: Array[Long],
                                    partOffsets: Array[Array[Long]], refChecksums: Array[Array[Long]],
                                    dst: Array[ByteBuffer], dstCap: Array[Long], outLen: Array[Long],
                                    outBadPartition: Array[Int], outStatus: Array[Int]): Int

  // ---- loading + per-thread contexts -------------------------------------------------------------------------------
  @volatile private var loaded = false

  def load(libraryPath: String): Unit = synchronized {
    if (!loaded) {
      // spark.shuffle.s3.gpu.library: an absolute path, or a bare name (libs3shuffle_jni.so / s3shuffle_jni) that is looked up
      // on java.library.path — System.load refuses anything that is not absolute
      if (libraryPath.contains(java.io.File.separator)) System.load(new java.io.File(libraryPath).getAbsolutePath)
      else System.loadLibrary(libraryPath.stripPrefix("lib").stripSuffix(".so"))
      require(abiVersion() == ABI_VERSION, s"libs3shuffle_codec ABI ${abiVersion()} != $ABI_VERSION")
      loaded = true
    }
  }

  /** A context is not thread-safe: one per task thread and device.  Executor task threads come from a cached pool and die
    * after idling, so every context is registered with its owning thread and destroyed once that thread is gone (advisor r3:
    * a leaked context keeps device workspace, pinned staging and four 64-128 MiB pipeline buffers).  The reaping runs on the
    * thread that is about to create a context — never concurrently with the dead owner, so no native call is in flight. */
  private val contexts = new ThreadLocal[scala.collection.mutable.Map[Int, Long]] {
    override def initialValue() = scala.collection.mutable.Map.empty[Int, Long]
  }
  private val owners = new java.util.concurrent.ConcurrentHashMap[java.lang.Long, java.lang.ref.WeakReference[Thread]]()

  /** destroys the contexts of task threads that no longer exist; returns how many went */
  def reapDeadThreads(): Int = {
    var n = 0
    val it = owners.entrySet().iterator()
    while (it.hasNext) {
      val e = it.next()
      val t = e.getValue.get()
      if ((t == null || !t.isAlive) && owners.remove(e.getKey, e.getValue)) { destroy(e.getKey); n += 1 }
    }
    n
  }

  /** live contexts (all devices): what spark.shuffle.s3.gpu.maxContexts bounds */
  def liveContexts: Int = owners.size()

  def forThread(device: Int): Long = contexts.get().getOrElseUpdate(device, {
    val d = org.apache.spark.shuffle.helper.S3ShuffleDispatcher.get
    load(d.gpuLibrary)
    reapDeadThreads()
    if (owners.size() >= d.gpuMaxContexts)
      throw new IOException(s"more than ${d.gpuMaxContexts} GPU codec contexts alive (spark.shuffle.s3.gpu.maxContexts): " +
        "task threads x devices; raise the key or lower spark.executor.cores")
    val h = create(device, 0L)
    if (h == 0L) throw new IOException(s"s3s_create($device) failed: no HIP device (there is no CPU fallback)")
    // the JVM codecs' chunk sizes (spark.io.compression.{lz4,snappy}.blockSize): the objects must look like theirs
    if (setOption(h, OPT_LZ4_BLOCK_SIZE, d.gpuLz4BlockSize) != OK || setOption(h, OPT_SNAPPY_BLOCK_SIZE, d.gpuSnappyBlockSize) != OK) {
      val why = lastError(h)
      destroy(h)
      throw new IllegalArgumentException(s"codec block size not supported by the GPU path: $why")
    }
    owners.put(h, new java.lang.ref.WeakReference[Thread](Thread.currentThread()))
    h
  })

  /** Number of HIP devices.  `deviceCount` is a native: the library is loaded first (a task's first call into this
    * object is usually this one — before any context exists). */
  def devices(): Int = {
    load(org.apache.spark.shuffle.helper.S3ShuffleDispatcher.get.gpuLibrary)
    deviceCount()
  }

  /** mapId % nGpu — S3ShuffleDispatcher.getPath shards folder prefixes the same way. */
  def deviceFor(mapId: Long, devices: Int): Int = (mapId % math.max(devices, 1)).toInt

  def check(handle: Long, rc: Int, what: => String, badPartition: Int = -1): Unit =
    if (rc != OK) raise(rc, lastError(handle), what, badPartition)

  /** The exception the reference raises for this condition today (message of the context that ran the call). */
  def raise(rc: Int, message: String, what: String, badPartition: Int = -1): Nothing = rc match {
    case E_INVALID => throw new RuntimeException(s"Precondition: $message")
    case E_CHECKSUM => throw new SparkException(s"Invalid checksum detected for $what (partition $badPartition)")
    case E_BAD_FRAME => throw new IOException("Stream is corrupted")
    case _ => throw new IOException(s"$what: $message (code $rc)")
  }

  /** lz4 and snappy run on the GPU; zstd and lzf keep the reference's JVM stream stack (DESIGN.md §7.1): the patched
    * call sites test this next to spark.shuffle.s3.gpu.enabled, so an unsupported codec is a fallback, not an error. */
  def supports(sparkCodecShortName: String): Boolean = sparkCodecShortName.toLowerCase match {
    case "lz4" | "snappy" => true
    case _ => false
  }

  def codecId(sparkCodecShortName: String): Int = sparkCodecShortName.toLowerCase match {
    case "lz4" => CODEC_LZ4
    case "snappy" => CODEC_SNAPPY
    case other => throw new IllegalArgumentException(s"spark.io.compression.codec=$other stays on the JVM codecs")
  }

  /** The reduce side decodes one codec more than the map side compresses: Zstandard frames as zstd-jni writes them
    * (S3S_CODEC_ZSTD) and LZF streams (S3S_CODEC_LZF): all four codecs of CompressionCodec.createCodec decode on the GPU. */
  def supportsDecode(sparkCodecShortName: String): Boolean = sparkCodecShortName.toLowerCase match {
    case "lz4" | "snappy" | "zstd" | "lzf" => true
    case _ => false
  }

  def decodeCodecId(sparkCodecShortName: String): Int = sparkCodecShortName.toLowerCase match {
    case "zstd" => CODEC_ZSTD
    case "lzf" => CODEC_LZF
    case other => codecId(other)
  }

  def checksumId(enabled: Boolean, algorithm: String): Int =
    if (!enabled) CHECKSUM_NONE
    else algorithm.toUpperCase match {
      case "ADLER32" => CHECKSUM_ADLER32
      case "CRC32" => CHECKSUM_CRC32
      // Spark 4's third algorithm (java.util.zip.CRC32C).  The reference's own createChecksumAlgorithm knows the two above
      // only, so with this name its JVM-side validation stream still refuses; the GPU path validates it in the library.
      case "CRC32C" => CHECKSUM_CRC32C
      // S3ShuffleHelper.createChecksumAlgorithm (S3ShuffleHelper.scala:94-103) rejects everything else too
      case other => throw new UnsupportedOperationException(s"Unsupported shuffle checksum algorithm: $other")
    }
}
