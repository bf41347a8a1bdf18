//
// S3GpuCommitQueue — map tasks of one executor that reach commitAllPartitions (shuffle/S3ShuffleMapOutputWriter.scala:
// 91-118) at about the same time share ONE library call.  A task thread alone drives the batched host entry point with
// a single task (chunked upload under the codec kernels: ~28 GB/s host to host); several tasks in one call run as a
// pipeline — upload of the next task, codec of this one, download of the previous one — at ~45 GB/s, the PCIe rate
// (s3s_compress_map_outputs_batch, jni/s3s_jni.c compressMapOutputsBatch).  One daemon thread and one native context per
// device; requests are taken in arrival order, at most spark.shuffle.s3.gpu.commitBatch (default 8) per call, and a
// lone request is never held back waiting for company.
//
// NOT COMPILED IN THIS IMAGE (no JDK / scalac).
//
package org.apache.spark.shuffle.gpu

import java.nio.ByteBuffer
import java.util.concurrent.{ConcurrentHashMap, CountDownLatch, LinkedBlockingQueue}

import org.apache.spark.shuffle.helper.S3ShuffleDispatcher

object S3GpuCommitQueue {
  final class Request(val codec: Int, val algo: Int, val src: ByteBuffer, val srcOffsets: Array[Long], val dst: ByteBuffer,
                      val dstCap: Long, val index: Array[Long], val sums: Array[Long]) {
    @volatile var total = 0L
    @volatile var rc = S3SCodec.OK
    @volatile var error: String = ""
    private[gpu] val done = new CountDownLatch(1)
  }

  private final class Worker(device: Int) extends Thread(s"s3-gpu-commit-$device") {
    setDaemon(true)
    val queue = new LinkedBlockingQueue[Request]()
    private val maxBatch = math.max(S3ShuffleDispatcher.get.gpuCommitBatch, 1)

    override def run(): Unit = {
      var ctx = 0L // this thread's own context, created with the first request: a failure to create it (no device, no
      // library) must fail the waiting requests instead of ending the thread and leaving them blocked in `compress`
      val batch = new java.util.ArrayList[Request]()
      while (true) {
        batch.clear()
        batch.add(queue.take())
        queue.drainTo(batch, maxBatch - 1)
        if (ctx == 0L) {
          try ctx = S3SCodec.forThread(device)
          catch {
            case t: Throwable =>
              val it = batch.iterator()
              while (it.hasNext) { val r = it.next(); r.rc = S3SCodec.E_HIP; r.error = t.toString; r.done.countDown() }
              batch.clear()
          }
        }
        if (!batch.isEmpty) runBatch(ctx, batch)
      }
    }

    private def runBatch(ctx: Long, batch: java.util.ArrayList[Request]): Unit = {
      // one call per (codec, checksum) pair: in practice one pair per application
      val first = batch.get(0)
      val same = new scala.collection.mutable.ArrayBuffer[Request]()
      val it = batch.iterator()
      while (it.hasNext) { val r = it.next(); if (r.codec == first.codec && r.algo == first.algo) same += r else queue.put(r) }
      val n = same.length
      val totals = new Array[Long](n)
      val status = new Array[Int](n)
      // the JNI unit and the library have early returns (bad argument, allocation failure) that never reach the per-task
      // stamping: a fresh Array[Int] would read as n x OK with total 0 - an EMPTY map output committed instead of an
      // exception (advisor r4).  Every entry starts as "not run"; only the library's own verdict can make it OK.
      java.util.Arrays.fill(status, S3SCodec.STATUS_NOT_RUN)
      try {
        val rc = S3SCodec.compressMapOutputsBatch(ctx, first.codec, first.algo, same.map(_.src).toArray,
          same.map(_.srcOffsets).toArray, same.map(_.dst).toArray, same.map(_.dstCap).toArray, same.map(_.index).toArray,
          if (first.algo == S3SCodec.CHECKSUM_NONE) null else same.map(_.sums).toArray, totals, status)
        val why = if (rc != S3SCodec.OK) S3SCodec.lastError(ctx) else ""
        // backstop: a failed call in which every task claims OK cannot be trusted task by task
        if (rc != S3SCodec.OK && status.forall(_ == S3SCodec.OK)) java.util.Arrays.fill(status, S3SCodec.STATUS_NOT_RUN)
        var i = 0
        while (i < n) {
          same(i).total = totals(i)
          // every task has its own verdict; the call's return code belongs to exactly the tasks the library never finished
          // (S3S_STATUS_NOT_RUN: a call-level failure - HIP error, bad argument - before or between the tasks), so one task's
          // corrupt input or short buffer never fails its neighbours (advisor r3)
          same(i).rc = if (status(i) == S3SCodec.STATUS_NOT_RUN) (if (rc != S3SCodec.OK) rc else S3SCodec.E_HIP) else status(i)
          same(i).error = why
          i += 1
        }
      } catch {
        case t: Throwable => same.foreach { r => r.rc = S3SCodec.E_HIP; r.error = t.toString }
      } finally same.foreach(_.done.countDown())
    }
  }

  private val workers = new ConcurrentHashMap[Int, Worker]()

  private def worker(device: Int): Worker = {
    var w = workers.get(device)
    if (w == null) {
      val fresh = new Worker(device)
      w = workers.putIfAbsent(device, fresh)
      if (w == null) { fresh.start(); w = fresh }
    }
    w
  }

  /** Blocks the task thread until its request has been compressed (possibly together with other tasks' requests).
   *  The wait is UNINTERRUPTIBLE on purpose (advisor r3): Spark interrupts task threads on kill and on speculation, and an
   *  InterruptedException here would unwind S3GpuMapOutput.flush, whose `finally` hands `dst` back to the buffer pool (and
   *  commit / abort releases `src`) while the worker thread is still inside s3s_compress_map_outputs_batch - DMA-reading
   *  `src` and writing `dst`.  Another task could then be given those page-locked buffers and have them overwritten by the
   *  late download.  So the buffers stay owned by this call until the native call has returned; the interrupt is remembered
   *  and re-raised on the thread afterwards, where the task's own cancellation checks see it. */
  def compress(device: Int, r: Request): Request = {
    worker(device).queue.put(r)
    var interrupted = false
    var finished = false
    val w = worker(device)
    while (!finished) {
      // bounded waits: if the worker thread has died (an Error escaping run()) nothing will ever count the latch down -
      // the request fails with E_HIP instead of holding an executor slot for ever (advisor r4).  A live worker inside the
      // native call is still waited for without limit: the buffers belong to that call until it returns.
      try {
        finished = r.done.await(1, java.util.concurrent.TimeUnit.SECONDS)
        if (!finished && !w.isAlive && r.done.getCount > 0) {
          r.rc = S3SCodec.E_HIP; r.error = "GPU commit worker thread is gone"; finished = true
          // drop the dead worker so that the NEXT commit on this device starts a fresh one (instead of every later commit
          // waiting a second and failing for the rest of the executor's life), and take the request out of the dead queue:
          // it references src / dst buffers the caller releases now (advisor r5)
          workers.remove(device, w)
          w.queue.remove(r)
        }
      } catch { case _: InterruptedException => interrupted = true }
    }
    if (interrupted) Thread.currentThread().interrupt()
    r
  }
}
