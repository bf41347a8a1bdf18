// mem_latency_probe.hip — round-trip latency of the memory paths the LZ4 window block could use, as ONE wavefront
// sees them, alone on its CU and next to nine neighbours that keep the CU's vector L1 (TCP) busy the way the
// compressor does (64-lane x 16-byte gathers at random addresses of an L2-resident buffer).
//
//   gather     64 lanes x dwordx4 at random 16-byte elements (the candidate gather)          vector L1 -> L2
//   row        64 lanes x dword, 256 contiguous bytes at a random offset (the extension)      vector L1 -> L2
//   sload16    s_load_dwordx16 at a random 64-byte line (scalar data cache -> L2)
//   lds        ds_read_b32 dependent chain
//   bperm      ds_bpermute_b32 dependent chain (the LDS crossbar, no LDS memory)
// plus the rule the hardware applies when several lanes of ONE ds_write_b16 hit the same address (which lane's
// value is in memory afterwards), for a few lane sets.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o mem_latency_probe mem_latency_probe.hip        (GPU tool, tools/probe)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

enum { T_GATHER = 0, T_ROW, T_SLOAD, T_LDS, T_BPERM };

// role of a wavefront: (blockIdx.x % 10 == probe_slot) measures `test`; every other one runs `neighbour`
// (-1 = exits at once).  16 KiB of LDS per workgroup: ten per CU, like the compress kernel.
__global__ __launch_bounds__(64) void probe(const uint4* __restrict__ elems, const uint32_t* __restrict__ lines,
                                            uint32_t n_elems, uint32_t n_lines, int test, int neighbour, int probe_slot,
                                            int iters, unsigned long long* out) {
  __shared__ uint32_t lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (uint32_t)((i * 2654435761u) >> 20) & 4095u;
  __syncthreads();
  const bool measuring = (int)(blockIdx.x % 10) == probe_slot;
  const int what = measuring ? test : neighbour;
  if (what < 0) return;
  const int n = measuring ? iters : iters * 4;
  uint32_t idx = (uint32_t)((blockIdx.x * 64u + lane) * 2654435761u) % n_elems;
  uint32_t sidx = (uint32_t)(blockIdx.x * 40503u) % n_lines;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (what == T_GATHER) {
    for (int i = 0; i < n; i++) {
      uint4 v;
      const uint4* p = elems + idx;
      asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
      idx = v.x;
      acc += v.y;
    }
  } else if (what == T_ROW) {
    uint32_t row = sidx;
    for (int i = 0; i < n; i++) {
      uint32_t v;
      const uint32_t* p = lines + (size_t)row * 16 + 1 + lane;  // 256 bytes from a dword that is not line-aligned
      asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
      row = __builtin_amdgcn_readfirstlane(v) % (n_lines - 8);
      // (lane 0 reads dword 1 of the line: the generator put the next line index there as well)
      acc += v;
    }
  } else if (what == T_SLOAD) {
    uint32_t row = sidx;
    for (int i = 0; i < n; i++) {
      const uint32_t* p = lines + (size_t)row * 16;
      uint32_t nxt;
      asm volatile(
          "s_load_dwordx16 s[20:35], %1, 0x0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s20"
          : "=s"(nxt)
          : "s"(p)
          : "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34",
            "s35");
      row = nxt;
    }
    acc = row;
  } else if (what == T_LDS) {
    uint32_t a = (uint32_t)lane * 4u;
    for (int i = 0; i < n; i++) {
      asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0" : "+v"(a)::"memory");
    }
    acc = a;
  } else if (what == T_BPERM) {
    uint32_t a = (uint32_t)lane * 4u, d = lane;
    for (int i = 0; i < n; i++) {
      asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(d) : "v"(a) : "memory");
    }
    acc = d;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && measuring) {
    out[2 * (blockIdx.x / 10)] = t1 - t0;
    out[2 * (blockIdx.x / 10) + 1] = acc + idx;
  }
}

// which lane's value survives one ds_write_b16 whose active lanes all store to the same address?
__global__ __launch_bounds__(64) void write_rule(uint32_t* out) {
  __shared__ uint16_t t[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) t[i] = 0xffff;
  __syncthreads();
  // case c: active lanes and the slot each one stores to
  //  0: all lanes -> one slot            1: even lanes -> one slot     2: lanes 5..40 -> one slot
  //  3: pairs (lane, lane ^ 32) share a slot (32 slots in 32 different banks)
  //  4: pairs (lane, lane ^ 1) share a slot       5: groups of four consecutive lanes, slots 64 entries apart (same bank)
  //  6: pairs (lane, 63 - lane)
  for (int c = 0; c < 7; c++) {
    bool act = true;
    int slot = 100 * c;
    if (c == 1) act = (lane & 1) == 0;
    if (c == 2) act = lane >= 5 && lane <= 40;
    if (c == 3) slot = 1000 + (lane & 31);
    if (c == 4) slot = 1100 + (lane >> 1);
    if (c == 5) slot = 2048 + (lane >> 2) * 64;
    if (c == 6) slot = 1200 + (lane < 32 ? lane : 63 - lane);
    if (act) t[slot] = (uint16_t)lane;
    __syncthreads();
    uint32_t r = act ? t[slot] : 0xffffu;
    out[c * 64 + lane] = r;
    __syncthreads();
  }
}

static double run(const uint4* d_e, const uint32_t* d_l, uint32_t ne, uint32_t nl, int test, int neighbour, int blocks,
                  int iters, unsigned long long* d_out) {
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d_e, d_l, ne, nl, test, neighbour, 3, iters, d_out);
  hipDeviceSynchronize();
  const int nm = blocks / 10;
  std::vector<unsigned long long> h(2 * (size_t)nm);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nm; i++) s += (double)h[2 * i];
  return s / nm / iters;
}

int main() {
  const uint32_t ne = 1u << 19;  // 8 MiB of 16-byte elements (inside the L2 of the XCDs taken together, MALL at worst)
  const uint32_t nl = 1u << 17;  // 8 MiB of 64-byte lines
  std::vector<uint4> e(ne);
  std::vector<uint32_t> l((size_t)nl * 16);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (uint32_t)(s >> 16);
  };
  for (uint32_t i = 0; i < ne; i++) e[i] = make_uint4(rnd() % ne, rnd(), rnd(), rnd());
  for (uint32_t i = 0; i < nl; i++) {
    for (int k = 0; k < 16; k++) l[(size_t)i * 16 + k] = rnd();
    const uint32_t nxt = rnd() % (nl - 8);
    l[(size_t)i * 16] = nxt;
    l[(size_t)i * 16 + 1] = nxt;
  }
  uint4* d_e;
  uint32_t* d_l;
  unsigned long long* d_out;
  hipMalloc(&d_e, e.size() * sizeof(uint4));
  hipMalloc(&d_l, l.size() * 4);
  hipMalloc(&d_out, 16 * 4096);
  hipMemcpy(d_e, e.data(), e.size() * sizeof(uint4), hipMemcpyHostToDevice);
  hipMemcpy(d_l, l.data(), l.size() * 4, hipMemcpyHostToDevice);
  const char* names[] = {"gather 64 x 16 B (random)", "row 256 B (contiguous, unaligned line)", "s_load_dwordx16",
                         "ds_read_b32", "ds_bpermute_b32"};
  printf("round trip as one wavefront sees it, s_memtime ticks per dependent access (100 MHz counter x clock ratio: see the LDS line for scale)\n");
  printf("%-42s %12s %12s %12s\n", "", "alone", "10 x same", "9 x gather");
  for (int t = 0; t < 5; t++) {
    const double alone = run(d_e, d_l, ne, nl, t, -1, 2560, 400, d_out);
    const double same = run(d_e, d_l, ne, nl, t, t, 2560, 400, d_out);
    const double busy = run(d_e, d_l, ne, nl, t, T_GATHER, 2560, 400, d_out);
    printf("%-42s %12.1f %12.1f %12.1f\n", names[t], alone, same, busy);
  }
  uint32_t* d_w;
  hipMalloc(&d_w, 7 * 64 * 4);
  hipLaunchKernelGGL(write_rule, dim3(1), dim3(64), 0, 0, d_w);
  std::vector<uint32_t> w(7 * 64);
  hipMemcpy(w.data(), d_w, w.size() * 4, hipMemcpyDeviceToHost);
  const char* cases[] = {"all 64 lanes, one slot", "even lanes, one slot", "lanes 5..40, one slot", "pairs (l, l^32)",
                         "pairs (l, l^1)", "groups of 4 lanes, same bank", "pairs (l, 63-l)"};
  printf("same-address ds_write_b16 within one instruction: value read back by each active lane\n");
  for (int c = 0; c < 7; c++) {
    printf("  %-30s:", cases[c]);
    bool hi = true, lo = true;
    for (int lane = 0; lane < 64; lane++) {
      const uint32_t r = w[c * 64 + lane];
      if (r == 0xffffu) continue;
      // the group of `lane`: lanes that read the same value
      int gmin = 64, gmax = -1;
      for (int k = 0; k < 64; k++)
        if (w[c * 64 + k] == r) {
          gmin = k < gmin ? k : gmin;
          gmax = k > gmax ? k : gmax;
        }
      if ((int)r != gmax) hi = false;
      if ((int)r != gmin) lo = false;
    }
    printf(" %s   (lane 0..7 read:", hi ? "HIGHEST lane wins" : lo ? "LOWEST lane wins" : "neither highest nor lowest");
    for (int lane = 0; lane < 8; lane++) printf(" %u", w[c * 64 + lane]);
    printf(")\n");
  }
  return 0;
}
