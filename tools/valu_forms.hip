// Issue rate of fp32 vector instructions on gfx950 BY OPERAND FORM: four independent dependency chains per wave, 1 / 2 / 4
// waves per SIMD; wave-instructions per second per SIMD (x 1e9).  The peak is one wave64 instruction per two cycles
// (MI355X_MICROARCH.md); tools/valu_chain showed v_fma_f32 with a scalar operand at half of that whatever the
// occupancy -- which forms share that fate?        tools/valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(S) S S S S
#define REP32(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S)

template <int FORM>
__global__ __launch_bounds__(1024) void forms(float* out, int iters, float sa, float sb) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  float va = sa + threadIdx.x * 1e-9f, vb = sb + threadIdx.x * 1e-9f;
  unsigned long long m = (threadIdx.x & 1) ? ~0ull : 0x5555555555555555ull;
  m = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
  asm volatile("" : "+v"(va), "+v"(vb), "+s"(sa), "+s"(sb), "+s"(m));
  for (int i = 0; i < iters; ++i) {
#define STEP(ASM, ...) REP32(asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : __VA_ARGS__);)
    if constexpr (FORM == 0) { STEP("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5", "v"(va), "v"(vb)) }
    if constexpr (FORM == 1) { STEP("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5", "v"(va), "v"(vb)) }
    if constexpr (FORM == 2) { STEP("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5", "s"(sa), "v"(vb)) }
    if constexpr (FORM == 3) { STEP("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3", "s"(sa)) }
    if constexpr (FORM == 4) { STEP("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3", "v"(va)) }
    if constexpr (FORM == 5) { STEP("v_mul_f32 %0, 0x3f7fbe77, %0\n v_mul_f32 %1, 0x3f7fbe77, %1\n v_mul_f32 %2, 0x3f7fbe77, %2\n v_mul_f32 %3, 0x3f7fbe77, %3", "v"(va)) }
    if constexpr (FORM == 6) { STEP("v_fmamk_f32 %0, %0, 0x3f7fbe77, %4\n v_fmamk_f32 %1, %1, 0x3f7fbe77, %4\n v_fmamk_f32 %2, %2, 0x3f7fbe77, %4\n v_fmamk_f32 %3, %3, 0x3f7fbe77, %4", "v"(vb)) }
    if constexpr (FORM == 7) { STEP("v_fma_f32 %0, %0, %4, 1.0\n v_fma_f32 %1, %1, %4, 1.0\n v_fma_f32 %2, %2, %4, 1.0\n v_fma_f32 %3, %3, %4, 1.0", "v"(va)) }
    if constexpr (FORM == 8) { STEP("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5", "v"(va), "s"(m)) }
    if constexpr (FORM == 9) { STEP("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc", "v"(va)) }
    if constexpr (FORM == 10) { STEP("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3", "v"(va)) }
    if constexpr (FORM == 11) { STEP("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0", "v"(va)) }
    if constexpr (FORM == 12) { STEP("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v"(va)) }
    if constexpr (FORM == 13) { STEP("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3", "v"(va)) }
    if constexpr (FORM == 14) { STEP("v_fma_f32 %0, %0, %4, -%5\n v_fma_f32 %1, -%1, %4, %5\n v_fma_f32 %2, %2, %4, -%5\n v_fma_f32 %3, -%3, %4, %5", "v"(va), "v"(vb)) }
    if constexpr (FORM == 15) { STEP("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_subrev_f32 %2, %4, %2\n v_sub_f32 %3, %3, %4", "v"(va)) }
  }
  if constexpr (FORM >= 16) {           // packed forms: two floats per 64-bit register pair, four chains of pairs
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x0 + 4.f, x1 + 4.f}, y3 = {x2 + 4.f, x3 + 4.f}, pa = {va, va}, pb = {vb, vb};
    for (int i = 0; i < iters; ++i) {
#define PSTEP(ASM) REP32(asm volatile(ASM : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(pa), "v"(pb));)
      if constexpr (FORM == 16) { PSTEP("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5") }
      if constexpr (FORM == 17) { PSTEP("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4") }
      if constexpr (FORM == 18) { PSTEP("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4") }
    }
    x0 = y0.x + y1.x + y2.x + y3.x; x1 = y0.y + y1.y + y2.y + y3.y;
  }
  if (x0 + x1 + x2 + x3 == 123.456f) out[threadIdx.x] = x0;
}

template <int FORM>
double run(int w, int iters, float* d_out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(forms<FORM>, dim3(256), dim3(256 * w), 0, 0, d_out, iters, 0.999f, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(forms<FORM>, dim3(256), dim3(256 * w), 0, 0, d_out, iters, 0.999f, 0.001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)w * iters * 128.0 / (ms * 1e-3) * 1e-9;
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, 4096);
  const int iters = 4000;
  printf("%-58s %8s %8s %8s   (1e9 wave-instructions per second per SIMD)\n", "form (4 chains per wave); waves per SIMD:", "1", "2", "4");
#define ROW(F, NAME) printf("%-58s %8.3f %8.3f %8.3f\n", NAME, run<F>(1, iters, d_out), run<F>(2, iters, d_out), run<F>(4, iters, d_out));
  ROW(0, "v_fmac_f32 v, v, v            (VOP2, registers only)")
  ROW(4, "v_mul_f32 v, v, v             (VOP2, registers only)")
  ROW(10, "v_add_f32 v, v, v             (VOP2, registers only)")
  ROW(15, "v_sub_f32 / v_subrev_f32      (VOP2, registers only)")
  ROW(11, "v_mov_b32 v, v                (VOP1)")
  ROW(1, "v_fma_f32 v, v, v, v          (VOP3, registers only)")
  ROW(14, "v_fma_f32 v, -v, v, v         (VOP3, neg modifiers)")
  ROW(7, "v_fma_f32 v, v, v, 1.0        (VOP3, inline constant)")
  ROW(2, "v_fma_f32 v, v, s, v          (VOP3, one SGPR)")
  ROW(3, "v_mul_f32 v, s, v             (VOP2, SGPR src0)")
  ROW(5, "v_mul_f32 v, 0x3f7fbe77, v    (VOP2 + 32-bit literal)")
  ROW(6, "v_fmamk_f32 v, v, literal, v  (VOP2 + 32-bit literal)")
  ROW(9, "v_cndmask_b32 v, v, v, vcc    (VOP2)")
  ROW(8, "v_cndmask_b32 v, v, v, s[n:n+1] (VOP3, SGPR pair)")
  ROW(12, "v_mov_b32_dpp wave_shr:1      (DPP)")
  ROW(13, "v_rcp_f32 v, v                (transcendental)")
  ROW(16, "v_pk_fma_f32 v[2], v[2], v[2], v[2]  (two floats each)")
  ROW(17, "v_pk_add_f32 v[2], v[2], v[2]        (two floats each)")
  ROW(18, "v_pk_mul_f32 v[2], v[2], v[2]        (two floats each)")
  return 0;
}
