// How many independent dependency chains does a SIMD of gfx950 need to issue fp32 vector instructions at its peak
// (one wave64 instruction per 2 cycles, MI355X_MICROARCH.md)?  Every lane runs C independent chains of dependent
// v_fma_f32; W waves per SIMD; the rate is reported in wave-instructions per cycle and SIMD at the clock the run sustains
// (s_memrealtime is 100 MHz; the shader clock comes from wall time and the known instruction count of a calibration run).
//   tools/valu_chain            (prints a table: chains per wave x waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C, int FORM>
__global__ __launch_bounds__(1024) void chain(float* out, int iters, float a, float b, unsigned long long* clocks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float x[C];
  float av = a + (float)threadIdx.x * 1e-9f, bv = b + (float)threadIdx.x * 1e-9f;
  asm volatile("" : "+v"(av), "+v"(bv));
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = (float)threadIdx.x * 1e-3f + (float)c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if constexpr (FORM == 0) x[c] = __builtin_fmaf(x[c], a, b);          // v_fma_f32 v, s, v, v  (VOP3, one scalar operand)
        else x[c] = __builtin_fmaf(av, bv, x[c]);                          // v_fmac_f32 v, v, v    (VOP2, registers only)
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  if (s == 123.456f) out[threadIdx.x] = s;     // (never true: keeps the chains alive)
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) clocks[0] = t1 - t0;      // shader clocks one wave took (s_memtime counts them)
}

static unsigned long long* d_clocks = nullptr;
static double g_cycles = 0.0;     // shader clocks per wave-instruction per SIMD of the last run, by the kernel's own clock
template <int C, int FORM>
double run(int waves_per_simd, int iters, float* d_out) {
  if (!d_clocks) hipMalloc(&d_clocks, 8);
  const int threads = 64 * 4 * waves_per_simd;          // one block per CU: waves_per_simd waves on each of its 4 SIMDs
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chain<C, FORM>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 0.999f, 0.001f, d_clocks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain<C, FORM>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 0.999f, 0.001f, d_clocks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)waves_per_simd * iters * 32.0 * C;      // wave-instructions per SIMD
  unsigned long long clk = 0;
  hipMemcpy(&clk, d_clocks, 8, hipMemcpyDeviceToHost);
  g_cycles = (double)clk / insts_per_simd;
  return insts_per_simd / (ms * 1e-3);                                           // per second per SIMD
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 4096);
  const int iters = 20000;
  printf("fp32 v_fma chains: wave-instructions per second per SIMD (x 1e9), and (shader clocks ONE wave spends per instruction of its own, by s_memtime)\n");
  printf("%-18s", "waves per SIMD:");
  for (int w : {1, 2, 3, 4}) printf("%16d", w);
  printf("\n");
  auto row = [&](const char* name, auto fn) {
    printf("%-18s", name);
    for (int w : {1, 2, 3, 4}) { double r = fn(w); printf("  %6.3f (%5.2f)", r * 1e-9, g_cycles * w); }
    printf("\n");
  };
  printf("v_fma_f32 with one scalar operand (VOP3):\n");
  row("1 chain per wave", [&](int w) { return run<1, 0>(w, iters, d_out); });
  row("2 chains per wave", [&](int w) { return run<2, 0>(w, iters, d_out); });
  row("4 chains per wave", [&](int w) { return run<4, 0>(w, iters, d_out); });
  row("8 chains per wave", [&](int w) { return run<8, 0>(w, iters, d_out); });
  printf("v_fmac_f32, vector registers only (VOP2):\n");
  row("1 chain per wave", [&](int w) { return run<1, 1>(w, iters, d_out); });
  row("2 chains per wave", [&](int w) { return run<2, 1>(w, iters, d_out); });
  row("4 chains per wave", [&](int w) { return run<4, 1>(w, iters, d_out); });
  row("8 chains per wave", [&](int w) { return run<8, 1>(w, iters, d_out); });
  return 0;
}
