// Round 4: the co-residency fault of DESIGN.md (appendix), as a matrix.  A streaming "victim" kernel on one stream, an
// MFMA-only "aggressor" kernel on another; the victim's output is compared with its solo output, 20 runs per cell.
//   aggressor axis: the MFMA instruction (bf16 16x16x32 | bf16 32x32x16 | f16 16x16x32 | fp32 16x16x4), the number of
//                   accumulator tiles (register footprint) and the declared occupancy (1 or 2 workgroups per CU:
//                   __launch_bounds__(256, 1 | 2) -- the second is the shape of conv2d_b / conv1x1_b)
//   victim axis:    what the victim does between its loads and its store (fp32 multiply-adds | integer adds | a plain copy |
//                   arithmetic on values that never came from memory)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize coresident_matrix.hip -o coresident_matrix && ./coresident_matrix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { V_FMA = 0, V_INT = 1, V_COPY = 2, V_NOLOAD = 3, V_PKMUL = 4, V_PKADD = 5, V_PKFMA = 6, V_PKNOLOAD = 7, V_PKMUL_B = 8, V_PKFMA_B = 9,
       V_PKMUL_BN = 10, V_OLD = 11, V_OLD_VG = 12, V_OLD_SC = 13, V_OLD_MOV = 14, V_OLD_DRAIN = 15,
       V_OLD_X4 = 16, NV = 17 };
static const char* VNAME[] = {"fp32 multiply-add", "integer add", "copy", "no loads", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
                              "v_pk_fma_f32, no loads", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                              "v_pk_mul_f32 op_sel_hi:[1,0], no loads", "round-3 victim (vector types, 16 KB weight table)",
                              "round-3 victim, weights from the lane id", "round-3 victim, scalar arithmetic",
                              "round-3 victim, loaded weights copied by v_mov_b32 first", "round-3 victim, s_waitcnt vmcnt(0) before the arithmetic",
                              "round-3 victim, weight row loaded as dwordx4"};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 instructions through inline asm: the file is built with -fno-slp-vectorize, so these are the ONLY packed ones
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 d; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// half-broadcast operand forms (what the compiler emits for {g.x, g.x} * v): src1's LOW half feeds both results
__device__ __forceinline__ f32x2 pk_mul_b(f32x2 a, f32x2 b) { f32x2 d; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fma_b(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <int KIND>
__global__ __launch_bounds__(256) void victim(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (KIND == V_OLD_DRAIN) {
    float4 xs[4], ys[4], gs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xs[k] = a[i + k * n]; ys[k] = b[i + k * n]; gs[k] = b[(i + k) & 1023]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 x = xs[k], y = ys[k], g = gs[k];
      const f32x2 wa = {g.x, g.x}, wb = {g.y, g.y};
      f32x2 lo = (wa * (f32x2){x.x, x.y} + wb * (f32x2){y.x, y.y}) + (f32x2){o.x, o.y};
      f32x2 hi = (wa * (f32x2){x.z, x.w} + wb * (f32x2){y.z, y.w}) + (f32x2){o.z, o.w};
      o = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
  } else if (KIND == V_OLD || KIND == V_OLD_VG || KIND == V_OLD_SC || KIND == V_OLD_MOV || KIND == V_OLD_X4) {
    // tools_dev/micro/pk_beside_mfma.hip's victim: per-lane weights g (from a 16 KB table | from the lane id) broadcast into
    // both halves of a packed operand -- the compiler emits v_pk_mul_f32 ... op_sel:[0,1], v_pk_fma_f32 ... op_sel_hi:[1,0,1]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 x = a[i + k * n], y = b[i + k * n];
      float4 g = b[(i + k) & 1023];
      if (KIND == V_OLD_VG) { g.x = 0.5f + (float)((i + k) & 1023) * 0.001f; g.y = 0.25f - (float)((i + k) & 1023) * 0.002f; }
      if (KIND == V_OLD_MOV) {
        float gx, gy;
        asm volatile("v_mov_b32 %0, %1" : "=v"(gx) : "v"(g.x));
        asm volatile("v_mov_b32 %0, %1" : "=v"(gy) : "v"(g.y));
        g.x = gx; g.y = gy;
      }
      if (KIND == V_OLD_X4) { g.x += g.z * 0.f; g.y += g.w * 0.f; }        // all four components needed: one dwordx4 load
      if (KIND == V_OLD_SC) {
        o.x = (g.x * x.x + g.y * y.x) + o.x; o.y = (g.x * x.y + g.y * y.y) + o.y;
        o.z = (g.x * x.z + g.y * y.z) + o.z; o.w = (g.x * x.w + g.y * y.w) + o.w;
      } else {
        const f32x2 wa = {g.x, g.x}, wb = {g.y, g.y};
        f32x2 lo = (wa * (f32x2){x.x, x.y} + wb * (f32x2){y.x, y.y}) + (f32x2){o.x, o.y};
        f32x2 hi = (wa * (f32x2){x.z, x.w} + wb * (f32x2){y.z, y.w}) + (f32x2){o.z, o.w};
        o = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    }
  } else if (KIND == V_PKMUL_BN) {
    const float f = (float)(i & 1023) * 0.25f + 1.f;
    f32x2 lo = {f, -f}, hi = {0.5f * f, f + 1.f};
    const f32x2 h = {0.75f, 123.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo = pk_mul_b(lo, h); hi = pk_mul_b(hi, lo); hi.x = hi.x * 1e-3f + 1.f; hi.y = hi.y * 1e-3f - 1.f; }
    o = make_float4(lo.x, lo.y, hi.x, hi.y);
  } else if (KIND == V_PKNOLOAD) {
    const float f = (float)(i & 1023) * 0.25f;
    f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
    const f32x2 ff = {f, -f}, h = {0.5f, 0.25f};
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo = pk_fma(lo, h, ff); hi = pk_fma(hi, h, lo); }
    o = make_float4(lo.x, lo.y, hi.x, hi.y);
  } else if (KIND == V_NOLOAD) {
    float f = (float)(i & 1023) * 0.25f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o.x = o.x * 0.5f + f; o.y = o.y * 0.25f + f; o.z = o.z + f; o.w = o.w * 0.125f - f; }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 x = a[i + k * n], y = b[i + k * n];
      if (KIND == V_FMA) {
        o.x = (0.5f * x.x + 0.25f * y.x) + o.x; o.y = (0.5f * x.y + 0.25f * y.y) + o.y;
        o.z = (0.5f * x.z + 0.25f * y.z) + o.z; o.w = (0.5f * x.w + 0.25f * y.w) + o.w;
      } else if (KIND == V_INT) {
        o.x = __int_as_float(__float_as_int(o.x) + __float_as_int(x.x) + (__float_as_int(y.x) >> 3));
        o.y = __int_as_float(__float_as_int(o.y) + __float_as_int(x.y) + (__float_as_int(y.y) >> 3));
        o.z = __int_as_float(__float_as_int(o.z) + __float_as_int(x.z) + (__float_as_int(y.z) >> 3));
        o.w = __int_as_float(__float_as_int(o.w) + __float_as_int(x.w) + (__float_as_int(y.w) >> 3));
      } else if (KIND == V_PKMUL_B || KIND == V_PKFMA_B) {
        f32x2 lo = {o.x, o.y}, hi = {o.z, o.w};
        const f32x2 xl = {x.x, x.y}, xh = {x.z, x.w}, yl = {y.x, y.y}, yh = {y.z, y.w};
        if (KIND == V_PKMUL_B) { f32x2 t = pk_mul_b(xl, yl), u = pk_mul_b(xh, yh); lo.x += t.x; lo.y += t.y; hi.x += u.x; hi.y += u.y; }
        else { lo = pk_fma_b(xl, yl, lo); hi = pk_fma_b(xh, yh, hi); }
        o = make_float4(lo.x, lo.y, hi.x, hi.y);
      } else if (KIND == V_PKMUL || KIND == V_PKADD || KIND == V_PKFMA) {
        f32x2 lo = {o.x, o.y}, hi = {o.z, o.w};
        const f32x2 xl = {x.x, x.y}, xh = {x.z, x.w}, yl = {y.x, y.y}, yh = {y.z, y.w}, h = {0.5f, 0.5f};
        if (KIND == V_PKMUL) { lo = pk_mul(xl, yl); hi = pk_mul(xh, yh); lo.x += o.x; lo.y += o.y; hi.x += o.z; hi.y += o.w; }
        else if (KIND == V_PKADD) { lo = pk_add(pk_add(xl, yl), lo); hi = pk_add(pk_add(xh, yh), hi); }
        else { lo = pk_fma(xl, h, pk_fma(yl, h, lo)); hi = pk_fma(xh, h, pk_fma(yh, h, hi)); }
        o = make_float4(lo.x, lo.y, hi.x, hi.y);
      } else {
        if (k == 3) o = x;
      }
    }
  }
  out[i] = o;
}

enum { A_BF16_16 = 0, A_BF16_32 = 1, A_F16_16 = 2, A_F32_16 = 3 };
static const char* ANAME[] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32"};

template <int KIND, int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void aggressor(const bf16x8* g, float* sink, int reps) {
  const int lane = threadIdx.x & 63;
  bf16x8 A = g[lane], B = g[lane + 64];
  float s = 0.f;
  if (KIND == A_BF16_32) {
    f32x16 acc[NACC / 4];
    for (int t = 0; t < NACC / 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int t = 0; t < NACC / 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < NACC / 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
  } else {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        if (KIND == A_BF16_16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[t], 0, 0, 0);
        else if (KIND == A_F16_16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, A)[0], __builtin_bit_cast(f32x4, B)[0], acc[t], 0, 0, 0);
      }
    }
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  }
  if (s == 123.456f) sink[blockIdx.x] = s;
}

struct Bufs { float4 *a, *b, *out; bf16x8* g; float* sink; hipStream_t sa, sb; long n; };

template <int VK>
static void launch_victim(const Bufs& B) {
  hipLaunchKernelGGL(victim<VK>, dim3(B.n / 256), dim3(256), 0, B.sb, B.a, B.b, B.out, B.n);
}

template <int VK, typename AGG>
static int run_cell(const Bufs& B, AGG agg, long* first_lane) {
  std::vector<float> ref(B.n * 4), got(B.n * 4);
  launch_victim<VK>(B);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), B.out, B.n * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  *first_lane = -1;
  for (int rep = 0; rep < 20; ++rep) {
    hipMemsetAsync(B.out, 0, B.n * 16, B.sb);
    hipDeviceSynchronize();
    agg();
    for (int k = 0; k < 6; ++k) launch_victim<VK>(B);
    hipDeviceSynchronize();
    hipMemcpy(got.data(), B.out, B.n * 16, hipMemcpyDeviceToHost);
    if (memcmp(got.data(), ref.data(), B.n * 16) != 0) {
      ++bad;
      if (*first_lane < 0)
        for (long i = 0; i < B.n * 4; ++i) if (memcmp(&got[i], &ref[i], 4)) { *first_lane = (i / 4) % 64; break; }
    }
  }
  return bad;
}

template <int AK, int NACC, int WPS>
static void row(const Bufs& B, int grid, int reps) {
  auto agg = [&]() { hipLaunchKernelGGL((aggressor<AK, NACC, WPS>), dim3(grid), dim3(256), 0, B.sa, B.g, B.sink, reps); };
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(aggressor<AK, NACC, WPS>));
  long l[NV];
  int b[NV];
  b[0] = run_cell<V_FMA>(B, agg, &l[0]); b[1] = run_cell<V_INT>(B, agg, &l[1]); b[2] = run_cell<V_COPY>(B, agg, &l[2]);
  b[3] = run_cell<V_NOLOAD>(B, agg, &l[3]); b[4] = run_cell<V_PKMUL>(B, agg, &l[4]); b[5] = run_cell<V_PKADD>(B, agg, &l[5]);
  b[6] = run_cell<V_PKFMA>(B, agg, &l[6]); b[7] = run_cell<V_PKNOLOAD>(B, agg, &l[7]); b[8] = run_cell<V_PKMUL_B>(B, agg, &l[8]);
  b[9] = run_cell<V_PKFMA_B>(B, agg, &l[9]); b[10] = run_cell<V_PKMUL_BN>(B, agg, &l[10]);
  b[11] = run_cell<V_OLD>(B, agg, &l[11]); b[12] = run_cell<V_OLD_VG>(B, agg, &l[12]); b[13] = run_cell<V_OLD_SC>(B, agg, &l[13]);
  b[14] = run_cell<V_OLD_MOV>(B, agg, &l[14]); b[15] = run_cell<V_OLD_DRAIN>(B, agg, &l[15]); b[16] = run_cell<V_OLD_X4>(B, agg, &l[16]);
  printf("| %s | %d | %d | %d |", ANAME[AK], NACC, WPS, fa.numRegs);
  for (int i = 0; i < NV; ++i) { if (b[i]) printf(" **%d** (lane %ld) |", b[i], l[i]); else printf(" 0 |"); }
  printf("\n");
  fflush(stdout);
}

int main() {
  Bufs B;
  B.n = 1 << 20;
  hipMalloc(&B.a, B.n * 4 * 16); hipMalloc(&B.b, B.n * 4 * 16); hipMalloc(&B.out, B.n * 16);
  std::vector<float> h(B.n * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(B.a, h.data(), B.n * 4 * 16, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 7) % 1000) / 400.f - 1.2f;
  hipMemcpy(B.b, h.data(), B.n * 4 * 16, hipMemcpyHostToDevice);
  hipMalloc(&B.g, 4096); hipMemset(B.g, 0x3c, 4096); hipMalloc(&B.sink, 1 << 16);
  hipStreamCreate(&B.sa); hipStreamCreate(&B.sb);
  printf("victim wrong in N of 20 runs (lane of the first wrong float4)\n");
  printf("| aggressor MFMA | acc tiles | declared WG/CU | registers |");
  for (int i = 0; i < NV; ++i) printf(" victim: %s |", VNAME[i]);
  printf("\n|---|---|---|---|");
  for (int i = 0; i < NV; ++i) printf("---|");
  printf("\n");
  {   // no aggressor
    auto none = []() {};
    long l;
    int b[NV] = {run_cell<V_FMA>(B, none, &l), run_cell<V_INT>(B, none, &l), run_cell<V_COPY>(B, none, &l), run_cell<V_NOLOAD>(B, none, &l),
                 run_cell<V_PKMUL>(B, none, &l), run_cell<V_PKADD>(B, none, &l), run_cell<V_PKFMA>(B, none, &l), run_cell<V_PKNOLOAD>(B, none, &l),
                 run_cell<V_PKMUL_B>(B, none, &l), run_cell<V_PKFMA_B>(B, none, &l), run_cell<V_PKMUL_BN>(B, none, &l),
                 run_cell<V_OLD>(B, none, &l), run_cell<V_OLD_VG>(B, none, &l), run_cell<V_OLD_SC>(B, none, &l),
                 run_cell<V_OLD_MOV>(B, none, &l), run_cell<V_OLD_DRAIN>(B, none, &l), run_cell<V_OLD_X4>(B, none, &l)};
    printf("| none | - | - | - |");
    for (int i = 0; i < NV; ++i) printf(" %d |", b[i]);
    printf("\n");
  }
  row<A_BF16_16, 16, 1>(B, 256, 4000);
  row<A_BF16_16, 64, 1>(B, 256, 4000);
  row<A_BF16_16, 100, 1>(B, 256, 4000);
  row<A_BF16_16, 48, 2>(B, 512, 4000);       // the conv2d_b / conv1x1_b shape: two workgroups per CU, <= 256 registers
  row<A_BF16_16, 16, 2>(B, 512, 4000);
  row<A_BF16_32, 64, 1>(B, 256, 2000);
  row<A_F16_16, 64, 1>(B, 256, 4000);
  row<A_F32_16, 64, 1>(B, 256, 2000);
  return 0;
}
