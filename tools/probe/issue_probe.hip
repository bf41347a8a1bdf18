// issue_probe.hip — per-instruction wave-time costs on gfx950 (dependent SALU / VALU, taken and not-taken branches,
// readlane -> SALU, ballot -> s_ff1), at 1 and at 10 waves per CU.  Build: hipcc --offload-arch=gfx950 -O2 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int kTest>
__global__ __launch_bounds__(64) void probe(unsigned long long* out, int iters, int lds_pad) {
  extern __shared__ uint8_t pad[];
  if (lds_pad < 0) pad[threadIdx.x] = 0;
  uint32_t s = blockIdx.x, t = 1, v = threadIdx.x, w = 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (kTest == 0) {
      asm volatile(REP64("s_add_u32 %0, %0, %1\n") : "+s"(s) : "s"(t) : "scc");
    } else if constexpr (kTest == 1) {
      uint32_t a = s, b = t, c = s + 1, d = t + 1;
      asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                   : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
      s += a + b + c + d;
    } else if constexpr (kTest == 2) {  // taken conditional branches
      asm volatile(REP16("s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 7\n1:\n"
                         "s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 7\n2:\n"
                         "s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 7\n3:\n"
                         "s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 7\n4:\n")
                   : "+s"(s) : "s"(t) : "scc");
    } else if constexpr (kTest == 3) {  // not-taken conditional branches
      asm volatile(REP16("s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 7\n1:\n"
                         "s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 7\n2:\n"
                         "s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 7\n3:\n"
                         "s_cmp_lg_u32 %1, %1\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 7\n4:\n")
                   : "+s"(s) : "s"(t) : "scc");
    } else if constexpr (kTest == 4) {  // dependent VALU
      asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w));
    } else if constexpr (kTest == 5) {  // readlane -> SALU -> readlane (lane select from the SALU result)
      asm volatile(REP16("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(s) : "v"(v) : "scc");
    } else if constexpr (kTest == 6) {  // v_cmp (ballot) -> s_ff1 -> v_add using it (VALU<->SALU ping-pong)
      uint32_t lo;
      asm volatile(REP16("v_cmp_lt_u32 vcc, %2, %1\n s_ff1_i32_b64 %0, vcc\n v_add_u32 %1, %1, %0\n")
                   : "=&s"(lo), "+v"(v) : "v"(w) : "vcc", "scc");
      s += lo;
    } else if constexpr (kTest == 7) {  // independent VALU
      uint32_t a = v, b = v + 1, c = v + 2, d = v + 3;
      asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w));
      v += a + b + c + d;
    } else if constexpr (kTest == 8) {  // far taken branches (each jumps over 256 bytes of code)
#define FAR(n) "s_branch " #n "f\n" REP16("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n") #n ":\n"
      asm volatile(REP4(FAR(1) FAR(2) FAR(3) FAR(4)) ::: "memory", "scc");
    } else if constexpr (kTest == 9) {  // LDS read dependent chain
      uint32_t a = (v * 4) & 1023;
      asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 1020, %0\n") : "+v"(a)::"memory");
      v += a;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = s + v;
  }
}

template <int kTest>
void run(const char* name, int n_inst, unsigned long long* d_out, int blocks, size_t lds) {
  const int iters = 200;
  hipLaunchKernelGGL(probe<kTest>, dim3(blocks), dim3(64), lds, 0, d_out, iters, 0);
  hipLaunchKernelGGL(probe<kTest>, dim3(blocks), dim3(64), lds, 0, d_out, iters, 0);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
  printf("  %-44s %7.2f cycles per instruction-group (%d per iter)\n", name, (double)h[0] / iters / n_inst, n_inst);
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, 16 * 4096);
  for (int cfg = 0; cfg < 2; cfg++) {
    const int blocks = cfg == 0 ? 256 : 2560;
    const size_t lds = cfg == 0 ? 65536 : 16384;  // 1 wave per CU (big LDS) vs 10 per CU
    printf("== %d blocks, %zu B LDS per block (%s)\n", blocks, lds, cfg == 0 ? "~1-2 waves/CU" : "10 waves/CU");
    run<0>("dependent s_add_u32", 64, d_out, blocks, lds);
    run<1>("independent s_add_u32", 64, d_out, blocks, lds);
    run<2>("s_cmp + TAKEN s_cbranch (skip 1 instr)", 64, d_out, blocks, lds);
    run<3>("s_cmp + NOT-taken s_cbranch + s_add", 64, d_out, blocks, lds);
    run<4>("dependent v_add_u32", 64, d_out, blocks, lds);
    run<7>("independent v_add_u32", 64, d_out, blocks, lds);
    run<5>("v_readlane -> s_and pair", 16, d_out, blocks, lds);
    run<6>("v_cmp -> s_ff1 -> v_add triple", 16, d_out, blocks, lds);
    run<8>("s_branch over 256 B of code", 16, d_out, blocks, lds);
    run<9>("ds_read -> wait -> v_and chain", 16, d_out, blocks, lds);
  }
  return 0;
}
