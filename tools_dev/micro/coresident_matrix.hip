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

enum { V_FMA = 0, V_INT = 1, V_COPY = 2, V_NOLOAD = 3 };
static const char* VNAME[] = {"fp32 multiply-add", "integer add", "copy", "no loads"};

template <int KIND>
__global__ __launch_bounds__(256) void victim(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (KIND == V_NOLOAD) {
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
  long l0, l1, l2, l3;
  int b0 = run_cell<V_FMA>(B, agg, &l0), b1 = run_cell<V_INT>(B, agg, &l1), b2 = run_cell<V_COPY>(B, agg, &l2), b3 = run_cell<V_NOLOAD>(B, agg, &l3);
  printf("| %s | %d | %d | %d | %d (lane %ld) | %d (lane %ld) | %d (lane %ld) | %d (lane %ld) |\n", ANAME[AK], NACC, WPS, fa.numRegs,
         b0, l0, b1, l1, b2, l2, b3, l3);
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
  printf("| aggressor MFMA | acc tiles | declared WG/CU | registers | victim: %s | %s | %s | %s |\n|---|---|---|---|---|---|---|---|\n",
         VNAME[0], VNAME[1], VNAME[2], VNAME[3]);
  {   // no aggressor
    auto none = []() {};
    long l;
    int b0 = run_cell<V_FMA>(B, none, &l), b1 = run_cell<V_INT>(B, none, &l), b2 = run_cell<V_COPY>(B, none, &l), b3 = run_cell<V_NOLOAD>(B, none, &l);
    printf("| none | - | - | - | %d | %d | %d | %d |\n", b0, b1, b2, b3);
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
