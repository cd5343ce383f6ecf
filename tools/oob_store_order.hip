// oob_store_order.hip -- does a buffer STORE whose lanes are all (or partly) out of the descriptor's range still take its
// turn in the wave's vmcnt queue?  s_waitcnt vmcnt(N) promises "all but the N youngest vector-memory operations are done"
// only if operations complete in issue order.  If the memory pipeline answers a dropped (out-of-range) store at once,
// ahead of an older load that is still on its way to memory, then vmcnt(1) behind {load, dropped store} lets the wave
// go on before the load has landed -- and the destination register is read (or reused) too early.
// Background: lbm_regtile's round-2 form with stores masked by out-of-range offsets (instead of branches) produced wrong
// lattices once a SIMD held two waves of a tile; this test checks the hardware property that form relied on.
//
//   tools/oob_store_order            prints, per case, how many of the waited-for loads had NOT landed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// case 0: load, in-range store, vmcnt(1)        (reference: must always be 0 misses)
// case 1: load, ALL-lanes-out-of-range store, vmcnt(1)
// case 2: load, store with lanes 1..62 out of range (lanes 0 and 63 in range), vmcnt(1)
// case 3: load, all-out-of-range LOAD, vmcnt(1)  (what the shipped kernel does with its mail fetches)
// case 4: load, exec-masked-off store (branch form: no lane active), vmcnt(1) -- the compiler-free equivalent of `if (lane_is_edge) store`
// case 5: load, in-range store, vmcnt(2)        (POSITIVE CONTROL: nothing is waited for, the load must usually NOT have landed --
//                                                 if this case shows no misses either, the probe cannot see what it looks for)
template <int CASE>
__global__ void probe(const unsigned* src, unsigned* sink, unsigned nbytes_src, unsigned nbytes_sink, unsigned* miss, int iters, unsigned stride) {
  const int lane = threadIdx.x & 63;
  const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes_src, 0x00020000);
  const auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, (int)nbytes_sink, 0x00020000);
  unsigned misses = 0;
  const unsigned OOB = 0x80000000u;
  for (int i = 0; i < iters; ++i) {
    // a load that goes all the way to memory: sc1, a line nobody touched recently (stride walks a large buffer)
    const unsigned off = (unsigned)(((unsigned long long)(gw * 977u + (unsigned)i * stride) * 256ull) % (nbytes_src - 4096u)) & ~15u;
    unsigned want = off / 4u + (unsigned)lane * 4u;      // src[j] = j
    u4 got = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    u4 val = {(unsigned)i, gw, (unsigned)lane, 7u};
    u4 dummy = {0u, 0u, 0u, 0u};
    unsigned svoff;
    if (CASE == 0 || CASE == 5) svoff = (gw * 64u + (unsigned)lane) * 16u % (nbytes_sink - 16u) & ~15u;
    else if (CASE == 1 || CASE == 3) svoff = OOB;
    else if (CASE == 2) svoff = (lane == 0 || lane == 63) ? ((gw * 64u + (unsigned)lane) * 16u % (nbytes_sink - 16u) & ~15u) : OOB;
    else svoff = (gw * 64u + (unsigned)lane) * 16u % (nbytes_sink - 16u) & ~15u;
    const unsigned lvoff = off + (unsigned)lane * 16u;
    if (CASE == 3) {
      asm volatile(
          "buffer_load_dwordx4 %0, %2, %3, 0 offen sc1\n\t"
          "buffer_load_dwordx4 %1, %4, %3, 0 offen sc1\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          : "+v"(got), "+v"(dummy) : "v"(lvoff), "s"(rs), "v"(svoff) : "memory");
    } else if (CASE == 4) {
      asm volatile(
          "buffer_load_dwordx4 %0, %1, %2, 0 offen sc1\n\t"
          "s_mov_b64 s[20:21], exec\n\t"
          "s_mov_b64 exec, 0\n\t"
          "buffer_store_dwordx4 %3, %4, %5, 0 offen sc1\n\t"
          "s_mov_b64 exec, s[20:21]\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          : "+v"(got) : "v"(lvoff), "s"(rs), "v"(val), "v"(svoff), "s"(rd) : "memory", "s20", "s21");
    } else if (CASE == 5) {
      asm volatile(
          "buffer_load_dwordx4 %0, %1, %2, 0 offen sc1\n\t"
          "buffer_store_dwordx4 %3, %4, %5, 0 offen sc1\n\t"
          "s_waitcnt vmcnt(2)\n\t"
          : "+v"(got) : "v"(lvoff), "s"(rs), "v"(val), "v"(svoff), "s"(rd) : "memory");
    } else {
      asm volatile(
          "buffer_load_dwordx4 %0, %1, %2, 0 offen sc1\n\t"
          "buffer_store_dwordx4 %3, %4, %5, 0 offen sc1\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          : "+v"(got) : "v"(lvoff), "s"(rs), "v"(val), "v"(svoff), "s"(rd) : "memory");
    }
    // `got` is sampled right behind the wait (volatile asm statements keep their order): landed or not?
    unsigned g0, g1, g2, g3;
    asm volatile("v_mov_b32 %0, %1" : "=v"(g0) : "v"(got.x));
    asm volatile("v_mov_b32 %0, %1" : "=v"(g1) : "v"(got.y));
    asm volatile("v_mov_b32 %0, %1" : "=v"(g2) : "v"(got.z));
    asm volatile("v_mov_b32 %0, %1" : "=v"(g3) : "v"(got.w));
    const bool ok = g0 == want && g1 == want + 1u && g2 == want + 2u && g3 == want + 3u;
    misses += ok ? 0u : 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (misses) atomicAdd(miss, misses);
}

// Second question (the masked-store hang of lbm_regtile's round-2 loop, bisected in round 3 to "north / south store masked
// AND east / west store masked": in a wave that is neither the tile's first nor its last, an ALL-lanes-out-of-range store
// directly in front of a store with two lanes in range): is a buffer store that follows an all-out-of-range store
// back to back performed?  GAP = s_nop count between the two (-1: none).  The host reads the sink back.
template <int GAP, int ORDER>
__global__ void probe_pair(unsigned* sink, unsigned nbytes_sink, int iters) {
  const int lane = threadIdx.x & 63;
  const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, (int)nbytes_sink, 0x00020000);
  const unsigned OOB = 0x80000000u;
  for (int i = 0; i < iters; ++i) {
    const unsigned slot = (gw * (unsigned)iters + (unsigned)i) * 2u;            // two granules per (wave, iteration): lanes 0 and 63
    const unsigned in_off = (lane == 0) ? slot * 16u : (lane == 63) ? (slot + 1u) * 16u : OOB;
    u4 val = {0xabcd0000u + (unsigned)i, gw, (unsigned)lane, 0x600dbeefu};
    u4 junk = {1u, 2u, 3u, 4u};
    if (ORDER == 0) {          // all-out-of-range store FIRST, the real one right behind it
      if (GAP < 0) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\tbuffer_store_dwordx4 %3, %4, %2, 0 offen sc1\n\ts_nop 1"
                                :: "v"(junk), "v"(OOB), "s"(rd), "v"(val), "v"(in_off) : "memory");
      else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop %5\n\tbuffer_store_dwordx4 %3, %4, %2, 0 offen sc1\n\ts_nop 1"
                        :: "v"(junk), "v"(OOB), "s"(rd), "v"(val), "v"(in_off), "n"(GAP < 0 ? 0 : GAP) : "memory");
    } else {                   // the real one first, the all-out-of-range store right behind it
      asm volatile("buffer_store_dwordx4 %3, %4, %2, 0 offen sc1\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1"
                   :: "v"(junk), "v"(OOB), "s"(rd), "v"(val), "v"(in_off) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int GAP, int ORDER>
static long run_pair(unsigned* sink, unsigned nsink, const char* what) {
  const int blocks = 512, wpb = 8, iters = 64;
  hipMemset(sink, 0, nsink);
  hipLaunchKernelGGL((probe_pair<GAP, ORDER>), dim3(blocks), dim3(64 * wpb), 0, 0, sink, nsink, iters);
  hipDeviceSynchronize();
  const size_t n = (size_t)blocks * wpb * iters * 2;
  std::vector<unsigned> h(n * 4);
  hipMemcpy(h.data(), sink, n * 16, hipMemcpyDeviceToHost);
  long lost = 0;
  for (size_t g = 0; g < n; ++g) lost += (h[4 * g + 3] != 0x600dbeefu) ? 1 : 0;
  printf("pair  %-70s: %ld of %zu in-range granules were NOT written\n", what, lost, n);
  return lost;
}

int main() {
  const unsigned nsrc = 1u << 30, nsink = 1u << 24;   // 1 GiB of loads (well beyond the caches), 16 MiB sink
  unsigned *src, *sink, *miss;
  hipMalloc(&src, nsrc); hipMalloc(&sink, nsink); hipMalloc(&miss, 64);
  std::vector<unsigned> h(nsrc / 4);
  for (size_t j = 0; j < h.size(); ++j) h[j] = (unsigned)j;
  hipMemcpy(src, h.data(), nsrc, hipMemcpyHostToDevice);
  const char* names[6] = {"load, in-range store, vmcnt(1)", "load, ALL-lanes-out-of-range store, vmcnt(1)",
                          "load, store with 62 of 64 lanes out of range, vmcnt(1)", "load, all-out-of-range LOAD, vmcnt(1)",
                          "load, store with EXEC = 0 (branch form), vmcnt(1)", "CONTROL: load, in-range store, vmcnt(2): no wait at all"};
  int bad = 0;
  for (int c = 0; c < 6; ++c) {
    for (int wpb : {1, 4, 16}) {
      hipMemset(miss, 0, 64);
      const int blocks = 256 * (wpb == 1 ? 8 : 2), iters = 400;
      const unsigned stride = 104729u;
      switch (c) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
        default: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(64 * wpb), 0, 0, src, sink, nsrc, nsink, miss, iters, stride); break;
      }
      hipDeviceSynchronize();
      unsigned m = 0;
      hipMemcpy(&m, miss, 4, hipMemcpyDeviceToHost);
      const long total = (long)blocks * wpb * iters;
      printf("case %d  %-58s %2d waves/block: %u of %ld waited-for loads had not landed\n", c, names[c], wpb, m, total);
      if (c == 0 && m) bad = 1;
      if (c == 5 && m == 0) bad = 2;
    }
  }
  run_pair<-1, 0>(sink, nsink, "all-out-of-range store, then (back to back) a store with lanes 0, 63 in range");
  run_pair<0, 0>(sink, nsink, "the same with s_nop 0 between the two");
  run_pair<1, 0>(sink, nsink, "the same with s_nop 1 between the two");
  run_pair<4, 0>(sink, nsink, "the same with s_nop 4 between the two");
  run_pair<-1, 1>(sink, nsink, "the store with lanes 0, 63 in range FIRST, the all-out-of-range store behind it");
  printf(bad == 1 ? "FAIL: the reference case itself misses -- the probe is wrong\n"
         : bad == 2 ? "FAIL: the control case shows no misses -- the probe cannot see an early wait\n" : "done\n");
  return bad;
}
