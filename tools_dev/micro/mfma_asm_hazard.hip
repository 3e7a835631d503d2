// Which orderings of an INLINE-ASSEMBLY vector instruction and a v_mfma_f32_16x16x32_f16 need wait states on gfx950?
// (tools_dev/asm_mfma_scan.py's rule and DESIGN.md Appendix A quote this probe.)  The compiler's hazard recogniser does not look
// inside asm statements, so whatever this prints as "WRONG" at gap g is a constellation inline assembly must never produce.
//   raw_b / raw_a : VALU writes the multiply's B / A registers, g wait states, then the multiply
//   war_a / war_b / war_c : the multiply, g wait states, then a VALU overwrites its A / B / C registers
// Every variant runs against the same sequence with 16 wait states; a difference in any lane of any wavefront is a hazard.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_asm_hazard mfma_asm_hazard.hip && ./mfma_asm_hazard      (profiles/round6_mfma_asm_hazard.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CLOB "s20", "s21", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
#define SETTLE "s_nop 7\n s_nop 7\n s_nop 7\n"
#define LOAD_AB                                                                                        \
  "v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n v_mov_b32 v12, %6\n v_mov_b32 v13, %7\n"                   \
  "v_mov_b32 v14, %8\n v_mov_b32 v15, %9\n v_mov_b32 v16, %10\n v_mov_b32 v17, %11\n"                 \
  "v_mov_b32 v24, %12\n v_mov_b32 v25, %12\n v_mov_b32 v26, %12\n v_mov_b32 v27, %12\n"
#define OUT4 "v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n v_mov_b32 %2, v22\n v_mov_b32 %3, v23\n"

struct Lane { unsigned a[4], b[4], n[4]; float c; };

// RAW on B: B first holds b[], the assembly-style writes put n[] there (v_cvt_pk_f16_f32 of two floats is what the split emits;
// a plain move has the same issue behaviour), NOPS, multiply
#define RAW_KERNEL(NAME, REGS, NOPS)                                                                   \
  __global__ void NAME(const Lane* in, float* out) {                                                   \
    const Lane L = in[blockIdx.x * 64 + threadIdx.x];                                                  \
    float r0, r1, r2, r3;                                                                              \
    asm volatile(LOAD_AB SETTLE REGS NOPS                                                              \
                 "v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[14:17], v[24:27]\n" SETTLE OUT4        \
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)                                              \
                 : "v"(L.a[0]), "v"(L.a[1]), "v"(L.a[2]), "v"(L.a[3]), "v"(L.b[0]), "v"(L.b[1]), "v"(L.b[2]), "v"(L.b[3]),  \
                   "v"(L.c), "v"(L.n[0]), "v"(L.n[1]), "v"(L.n[2]), "v"(L.n[3])                        \
                 : CLOB);                                                                              \
    float* o = out + (blockIdx.x * 64 + threadIdx.x) * 4;                                              \
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                        \
  }
#define WB "v_mov_b32 v14, %13\n v_mov_b32 v15, %14\n v_mov_b32 v16, %15\n v_mov_b32 v17, %16\n"
#define WA "v_mov_b32 v10, %13\n v_mov_b32 v11, %14\n v_mov_b32 v12, %15\n v_mov_b32 v13, %16\n"
// the split's own producers: a packed conversion (h0) and a mixed fma into the high half (h1) as the LAST write before the multiply
#define WB_CVT "v_mov_b32 v14, %13\n v_mov_b32 v15, %14\n v_mov_b32 v16, %15\n v_mov_b32 v17, %16\n v_cvt_pk_f16_f32 v17, %12, %12\n"
#define WB_MIX "v_mov_b32 v14, %13\n v_mov_b32 v15, %14\n v_mov_b32 v16, %15\n v_mov_b32 v17, %16\n v_fma_mixhi_f16 v17, %12, 1.0, 0 op_sel_hi:[0,0,0]\n"

// WAR: multiply, NOPS, then the registers REGS names are overwritten with n[]
#define WAR_KERNEL(NAME, REGS, NOPS)                                                                   \
  __global__ void NAME(const Lane* in, float* out) {                                                   \
    const Lane L = in[blockIdx.x * 64 + threadIdx.x];                                                  \
    float r0, r1, r2, r3;                                                                              \
    asm volatile(LOAD_AB SETTLE                                                                        \
                 "v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[14:17], v[24:27]\n" NOPS REGS SETTLE OUT4   \
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)                                              \
                 : "v"(L.a[0]), "v"(L.a[1]), "v"(L.a[2]), "v"(L.a[3]), "v"(L.b[0]), "v"(L.b[1]), "v"(L.b[2]), "v"(L.b[3]),  \
                   "v"(L.c), "v"(L.n[0]), "v"(L.n[1]), "v"(L.n[2]), "v"(L.n[3])                        \
                 : CLOB);                                                                              \
    float* o = out + (blockIdx.x * 64 + threadIdx.x) * 4;                                              \
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                        \
  }
#define WC "v_mov_b32 v24, %13\n v_mov_b32 v25, %14\n v_mov_b32 v26, %15\n v_mov_b32 v27, %16\n"
// reversed order: the FIRST overwritten register is the last of the operand
#define WA_R "v_mov_b32 v13, %16\n v_mov_b32 v12, %15\n v_mov_b32 v11, %14\n v_mov_b32 v10, %13\n"
#define WB_R "v_mov_b32 v17, %16\n v_mov_b32 v16, %15\n v_mov_b32 v15, %14\n v_mov_b32 v14, %13\n"

#define GAPS(M, NAME, REGS)                                                                            \
  M(NAME##_0, REGS, "") M(NAME##_1, REGS, "s_nop 0\n") M(NAME##_2, REGS, "s_nop 1\n") M(NAME##_3, REGS, "s_nop 2\n")   \
  M(NAME##_4, REGS, "s_nop 3\n") M(NAME##_6, REGS, "s_nop 5\n") M(NAME##_8, REGS, "s_nop 7\n") M(NAME##_ref, REGS, SETTLE)

GAPS(RAW_KERNEL, raw_b, WB)
GAPS(RAW_KERNEL, raw_a, WA)
GAPS(RAW_KERNEL, raw_b_cvt, WB_CVT)
GAPS(RAW_KERNEL, raw_b_mix, WB_MIX)
// what sits between instead of s_nop: scalar instructions / a satisfied s_waitcnt / an independent VALU instruction
#define FILL(M, NAME, REGS, F)                                                                         \
  M(NAME##_0, REGS, "") M(NAME##_1, REGS, F) M(NAME##_2, REGS, F F) M(NAME##_3, REGS, F F F)             \
  M(NAME##_4, REGS, F F F F) M(NAME##_6, REGS, F F F F F F) M(NAME##_8, REGS, F F F F F F F F) M(NAME##_ref, REGS, SETTLE)
FILL(RAW_KERNEL, raw_b_salu, WB, "s_mov_b32 s20, 0\n")
FILL(RAW_KERNEL, raw_b_wcnt, WB, "s_waitcnt vmcnt(0)\n")
FILL(RAW_KERNEL, raw_b_valu, WB, "v_mov_b32 v26, v26\n")
// the same RAW sequence while ANOTHER wavefront on the same SIMD issues matrix instructions back to back (512 threads: wavefronts 4-7
// share the SIMDs of wavefronts 0-3 and spin on multiplies; 0-3 run the sequence after a short sleep)
#define BUSY_KERNEL(NAME, REGS, NOPS)                                                                  \
  __global__ void __launch_bounds__(512) NAME(const Lane* in, float* out) {                            \
    const int wave = threadIdx.x >> 6;                                                                 \
    if (wave >= 4) {                                                                                   \
      asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"       \
                   "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"       \
                   "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"       \
                   "s_movk_i32 s20, 600\n"                                                             \
                   "1:\n v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[10:13], v[20:23]\n"             \
                   "v_mfma_f32_16x16x32_f16 v[24:27], v[10:13], v[10:13], v[24:27]\n"                  \
                   "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::: CLOB, "scc");  \
      return;                                                                                          \
    }                                                                                                  \
    const Lane L = in[blockIdx.x * 256 + threadIdx.x];                                                 \
    float r0, r1, r2, r3;                                                                              \
    asm volatile("s_sleep 8\n" LOAD_AB SETTLE REGS NOPS                                                \
                 "v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[14:17], v[24:27]\n" SETTLE OUT4        \
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)                                              \
                 : "v"(L.a[0]), "v"(L.a[1]), "v"(L.a[2]), "v"(L.a[3]), "v"(L.b[0]), "v"(L.b[1]), "v"(L.b[2]), "v"(L.b[3]),  \
                   "v"(L.c), "v"(L.n[0]), "v"(L.n[1]), "v"(L.n[2]), "v"(L.n[3])                        \
                 : CLOB);                                                                              \
    float* o = out + (blockIdx.x * 256 + threadIdx.x) * 4;                                             \
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                        \
  }
GAPS(BUSY_KERNEL, busy_raw_b, WB)
GAPS(BUSY_KERNEL, busy_raw_b_cvt, WB_CVT)
FILL(BUSY_KERNEL, busy_raw_b_salu, WB, "s_mov_b32 s21, 0\n")
FILL(BUSY_KERNEL, busy_raw_b_wcnt, WB, "s_waitcnt vmcnt(0)\n")
GAPS(WAR_KERNEL, war_a, WA)
GAPS(WAR_KERNEL, war_b, WB)
GAPS(WAR_KERNEL, war_c, WC)
GAPS(WAR_KERNEL, war_a_rev, WA_R)
GAPS(WAR_KERNEL, war_b_rev, WB_R)

typedef void (*Kern)(const Lane*, float*);
struct Case { const char* name; Kern k[8]; int threads; };
#define CASE(NAME) {#NAME, {NAME##_0, NAME##_1, NAME##_2, NAME##_3, NAME##_4, NAME##_6, NAME##_8, NAME##_ref}, 64}
#define BCASE(NAME) {#NAME, {NAME##_0, NAME##_1, NAME##_2, NAME##_3, NAME##_4, NAME##_6, NAME##_8, NAME##_ref}, 512}

static unsigned short f16bits(float f) {                 // small exactly-representable values only
  _Float16 h = (_Float16)f;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}

int main() {
  const int blocks = 2048, lanes = blocks * 64;
  std::vector<Lane> h(lanes);
  srand(7);
  auto pk = [&]() { return (unsigned)f16bits((rand() % 17) - 8) | ((unsigned)f16bits((rand() % 17) - 8) << 16); };
  for (auto& L : h) {
    for (int i = 0; i < 4; ++i) { L.a[i] = pk(); L.b[i] = pk(); L.n[i] = pk(); }
    L.c = (float)(rand() % 9);
  }
  Lane* d_in; float *d_out, *d_ref;
  hipMalloc(&d_in, lanes * sizeof(Lane));
  hipMalloc(&d_out, lanes * 16);
  hipMalloc(&d_ref, lanes * 16);
  hipMemcpy(d_in, h.data(), lanes * sizeof(Lane), hipMemcpyHostToDevice);
  const Case cases[] = {CASE(raw_b), CASE(raw_a), CASE(raw_b_cvt), CASE(raw_b_mix), CASE(raw_b_salu), CASE(raw_b_wcnt), CASE(raw_b_valu), CASE(war_a), CASE(war_b), CASE(war_c),
                        CASE(war_a_rev), CASE(war_b_rev), BCASE(busy_raw_b), BCASE(busy_raw_b_cvt),
                        BCASE(busy_raw_b_salu), BCASE(busy_raw_b_wcnt)};
  const int gaps[7] = {0, 1, 2, 3, 4, 6, 8};
  std::vector<float> ref(lanes * 4), got(lanes * 4);
  int bad_total = 0;
  for (const Case& c : cases) {
    const int nb = c.threads == 64 ? blocks : blocks / 4;   // 512-thread cases: 256 measured lanes per block
    hipLaunchKernelGGL(c.k[7], dim3(nb), dim3(c.threads), 0, 0, d_in, d_ref);
    hipMemcpy(ref.data(), d_ref, lanes * 16, hipMemcpyDeviceToHost);
    printf("%-16s", c.name);
    for (int g = 0; g < 7; ++g) {
      long bad = 0;
      for (int rep = 0; rep < 8; ++rep) {
        hipLaunchKernelGGL(c.k[g], dim3(nb), dim3(c.threads), 0, 0, d_in, d_out);
        hipMemcpy(got.data(), d_out, lanes * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < lanes * 4; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
      }
      printf("  gap %d: %s", gaps[g], bad ? "WRONG" : "ok");
      if (bad) printf("(%ld)", bad);
      bad_total += bad != 0;
    }
    printf("\n");
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
  printf("%d hazardous (kind, gap) cells\n", bad_total);
  return 0;
}
