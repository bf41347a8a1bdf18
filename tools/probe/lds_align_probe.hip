// lds_align_probe.hip — what an LDS access costs on gfx950 by width and alignment (ds_read / ds_write b32, b64, b128,
// u8, ds_bpermute), with scattered per-lane addresses as a match copy has them and with consecutive ones, at 16 waves
// per CU.  Also checks that unaligned b64 / b128 accesses return the right bytes.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_align_probe lds_align_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

constexpr int kLds = 8192;

template <int kOp>
__global__ __launch_bounds__(64) void probe(unsigned long long* out, uint32_t* chk, int iters, int stride, int mis) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLds + 64];
  for (int i = threadIdx.x; i < kLds + 64; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)lds;
  uint32_t a = base + ((threadIdx.x * stride) % (kLds - 64)) + mis;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, acc = 0;
  uint32_t lane4 = (threadIdx.x * 29 & 63) * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (kOp == 0) {
      asm volatile(REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r0) : "v"(a) : "memory");
    } else if constexpr (kOp == 1) {
      uint64_t r;
      asm volatile(REP16("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r) : "v"(a) : "memory");
      r0 = (uint32_t)r; r1 = (uint32_t)(r >> 32);
    } else if constexpr (kOp == 2) {
      __attribute__((ext_vector_type(4))) uint32_t r;
      asm volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r) : "v"(a) : "memory");
      r0 = r.x; r1 = r.y; r2 = r.z; r3 = r.w;
    } else if constexpr (kOp == 3) {
      asm volatile(REP16("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(acc) : "memory");
    } else if constexpr (kOp == 4) {
      uint64_t v = acc;
      asm volatile(REP16("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(v) : "memory");
    } else if constexpr (kOp == 5) {
      __attribute__((ext_vector_type(4))) uint32_t v = {acc, acc, acc, acc};
      asm volatile(REP16("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(v) : "memory");
    } else if constexpr (kOp == 6) {
      asm volatile(REP16("ds_read_u8 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r0) : "v"(a) : "memory");
    } else if constexpr (kOp == 7) {
      asm volatile(REP16("ds_write_b8 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(acc) : "memory");
    } else if constexpr (kOp == 8) {
      asm volatile(REP16("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r0) : "v"(lane4), "v"(acc) : "memory");
    } else if constexpr (kOp == 9) {
      asm volatile(REP16("ds_read_u16 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r0) : "v"(a) : "memory");
    } else if constexpr (kOp == 10) {
      asm volatile(REP16("ds_write_b16 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(acc) : "memory");
    }
    acc += r0 ^ r1 ^ r2 ^ r3;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
  if (blockIdx.x == 0 && chk) {  // lane's read result of the last iteration (reads only)
    chk[threadIdx.x * 4 + 0] = r0; chk[threadIdx.x * 4 + 1] = r1; chk[threadIdx.x * 4 + 2] = r2; chk[threadIdx.x * 4 + 3] = r3;
  }
  if (acc == 0x12345678u) out[1] = acc;
}

template <int kOp>
static void run(const char* name, int width, bool is_read, int stride, int mis) {
  unsigned long long* d; uint32_t* chk;
  hipMalloc(&d, 16); hipMalloc(&chk, 64 * 16);
  hipMemset(d, 0, 16); hipMemset(chk, 0, 64 * 16);
  const int blocks = 256 * 16, iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<kOp><<<blocks, 64>>>(d, nullptr, 10, stride, mis);
  hipEventRecord(e0);
  probe<kOp><<<blocks, 64>>>(d, chk, iters, stride, mis);
  hipEventRecord(e1);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  uint32_t h[256]; hipMemcpy(h, chk, sizeof h, hipMemcpyDeviceToHost);
  const char* ok = "";
  if (is_read && width >= 4 && kOp != 8) {
    int bad = 0;
    for (int l = 0; l < 64; l++) {
      const int ad = ((l * stride) % (kLds - 64)) + mis;
      for (int b = 0; b < width; b++) {
        const uint8_t want = (uint8_t)((ad + b) * 7 + 3);
        const uint8_t got = (uint8_t)(h[l * 4 + b / 4] >> (8 * (b % 4)));
        bad += want != got;
      }
    }
    ok = bad ? " WRONG DATA" : " data ok";
  }
  // LDS-pipe cycles per wave-instruction if the CU's LDS were the only limit: time * clock / (ops per CU)
  const double ops_per_cu = (double)blocks / 256.0 * iters * 16.0;
  printf("%-14s stride %4d mis %2d: %8.3f ms  %6.1f cycles/op/CU @2.4GHz  (%s)%s\n", name, stride, mis, ms,
         ms * 1e-3 * 2.4e9 / ops_per_cu, e == hipSuccess ? "ok" : hipGetErrorString(e), ok);
  hipFree(d); hipFree(chk);
}

int main() {
  const int strides[2] = {4, 148};   // consecutive dwords; scattered (37 dwords apart)
  for (int s : strides) {
    for (int mis : {0, 1, 2, 3}) run<0>("ds_read_b32", 4, true, s, mis);
    for (int mis : {0, 1, 2, 3}) run<3>("ds_write_b32", 4, false, s, mis);
    for (int mis : {0, 1}) run<6>("ds_read_u8", 1, true, s, mis);
    for (int mis : {0, 1}) run<7>("ds_write_b8", 1, false, s, mis);
    for (int mis : {0, 1, 2}) run<9>("ds_read_u16", 2, true, s, mis);
    for (int mis : {0, 1, 2}) run<10>("ds_write_b16", 2, false, s, mis);
  }
  for (int s : {8, 152}) {
    for (int mis : {0, 4, 1, 2}) run<1>("ds_read_b64", 8, true, s, mis);
    for (int mis : {0, 4, 1, 2}) run<4>("ds_write_b64", 8, false, s, mis);
  }
  for (int s : {16, 208}) {
    for (int mis : {0, 8, 4, 1}) run<2>("ds_read_b128", 16, true, s, mis);
    for (int mis : {0, 8, 4, 1}) run<5>("ds_write_b128", 16, false, s, mis);
  }
  run<8>("ds_bpermute", 4, true, 4, 0);
  return 0;
}
