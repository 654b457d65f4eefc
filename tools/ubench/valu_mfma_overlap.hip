// Microbenchmark: do VALU / transcendental instructions of one wave overlap with the MFMAs of ANOTHER wave on the same SIMD?
// A workgroup is 8 waves = 2 per SIMD (wave w runs on SIMD w % 4): waves 0-3 run the "a" loop, waves 4-7 the "b" loop, so each
// SIMD holds one wave of each kind.  Three launches per pair: a alone (b idle), b alone, both.  both ~ max(a, b): overlap;
// both ~ a + b: the two share an issue resource.  The inline-asm loops (v_add_f32 ... v_permlane32_swap) are latency-bound alone
// (~7.5 cycles per instruction, 4 of them issue time): read their rows as issue-time accounting -- both ~ mfma + 4 cycles x count.
//   hipcc --offload-arch=gfx950 -O3 -o valu_mfma_overlap tools/ubench/valu_mfma_overlap.hip && ./valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { MFMA_BF16 = 0, MFMA_F32 = 1, VALU_FMA = 2, VALU_EXP = 3, VALU_PKFMA = 4, IDLE = 5, VALU_MULLO = 6, LDS_READ = 7,
       OP_ADD_F32 = 10, OP_MUL_F32, OP_MAX_F32, OP_MED3_F32, OP_ADD_U32, OP_AND_B32, OP_CNDMASK, OP_MOV, OP_CVT_PK_BF16, OP_PK_ADD_F32, OP_LSHL_ADD, OP_SWAP };
#define ASMLOOP(TXT)                                                                  \
  {                                                                                   \
    unsigned v[8];                                                                    \
    for (int i = 0; i < 8; ++i) v[i] = __float_as_uint(in[t][i & 3]) + i;             \
    unsigned c = __float_as_uint(in[t + 1][0]);                                       \
    for (int it = 0; it < iters; ++it)                                                \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(TXT : "+v"(v[i]) : "v"(c)); \
    unsigned s_ = 0;                                                                  \
    for (int i = 0; i < 8; ++i) s_ += v[i];                                           \
    return (float)s_;                                                                 \
  }

template <int KIND>
__device__ __forceinline__ float work(const f32x4* in, int iters) {
  const int t = threadIdx.x & 63;
  if constexpr (KIND == MFMA_BF16) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const bf16x8 a = __builtin_bit_cast(bf16x8, in[t]), b = __builtin_bit_cast(bf16x8, in[t + 64]);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    return s;
  } else if constexpr (KIND == MFMA_F32) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const float a = in[t][0], b = in[t][1];
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    return s;
  } else if constexpr (KIND == VALU_FMA) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[t][i & 3] + i;
    const float c = in[t + 1][0];
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c, 0.5f);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    return s;
  } else if constexpr (KIND == VALU_PKFMA) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){in[t][i & 3] + i, in[t][(i + 1) & 3]};
    const f32x2 c = {in[t + 1][0], in[t + 1][1]}, h = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], c, h);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    return s;
  } else if constexpr (KIND == VALU_MULLO) {
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = __float_as_uint(in[t][i & 3]) + i;
    const unsigned c = __float_as_uint(in[t + 1][0]) | 1u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    return (float)s;
  } else if constexpr (KIND == LDS_READ) {
    __shared__ __attribute__((aligned(16))) f32x4 buf[512];
    buf[threadIdx.x] = in[t];
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(t * 16)), "n"(i * 1024));
        acc += r;
      }
    return acc[0] + acc[1];
  } else if constexpr (KIND == OP_ADD_F32) ASMLOOP("v_add_f32 %0, %0, %1")
  else if constexpr (KIND == OP_MUL_F32) ASMLOOP("v_mul_f32 %0, %0, %1")
  else if constexpr (KIND == OP_MAX_F32) ASMLOOP("v_max_f32 %0, %0, %1")
  else if constexpr (KIND == OP_MED3_F32) ASMLOOP("v_med3_f32 %0, %0, %1, 0")
  else if constexpr (KIND == OP_ADD_U32) ASMLOOP("v_add_u32 %0, %0, %1")
  else if constexpr (KIND == OP_AND_B32) ASMLOOP("v_and_b32 %0, %0, %1")
  else if constexpr (KIND == OP_CNDMASK) ASMLOOP("v_cndmask_b32 %0, %0, %1, vcc")
  else if constexpr (KIND == OP_MOV) ASMLOOP("v_mov_b32 %0, %1")
  else if constexpr (KIND == OP_CVT_PK_BF16) ASMLOOP("v_cvt_pk_bf16_f32 %0, %0, %1")
  else if constexpr (KIND == OP_LSHL_ADD) ASMLOOP("v_lshl_add_u32 %0, %0, 1, %1")
  else if constexpr (KIND == OP_SWAP) ASMLOOP("v_permlane32_swap_b32 %0, %1")
  else if constexpr (KIND == OP_PK_ADD_F32) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){in[t][i & 3] + i, in[t][(i + 1) & 3]};
    const f32x2 c = {in[t + 1][0], in[t + 1][1]};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    return s;
  } else if constexpr (KIND == VALU_EXP) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[t][i & 3] * 0.01f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    return s;
  } else {
    return 0.f;
  }
}

template <int A, int B>
__global__ __launch_bounds__(512, 2) void k(const f32x4* in, float* out, int ia, int ib) {
  const int w = threadIdx.x >> 6;
  float r;
  if (w < 4) r = work<A>(in, ia);
  else r = work<B>(in, ib);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int A, int B>
float timeit(const f32x4* in, float* out, int ia, int ib) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, in, out, ia, ib);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int A, int B>
void pair(const char* na, const char* nb, const f32x4* in, float* out, int ia, int ib) {
  const float ta = timeit<A, IDLE>(in, out, ia, 0), tb = timeit<IDLE, B>(in, out, 0, ib), tab = timeit<A, B>(in, out, ia, ib);
  printf("%-26s %7.3f ms | %-22s %7.3f ms | both %7.3f ms  (max %.3f, sum %.3f) -> %s\n", na, ta, nb, tb, tab, ta > tb ? ta : tb,
         ta + tb, tab < 0.5f * ((ta > tb ? ta : tb) + ta + tb) ? "OVERLAP" : "SERIAL");
}

int main() {
  f32x4* in;
  float* out;
  hipMalloc(&in, 1024 * 16);
  hipMalloc(&out, 1 << 22);
  hipMemset(in, 0x3c, 1024 * 16);
  const int N = 20000;
  // iteration counts chosen so that each side alone takes about the same time
  pair<MFMA_BF16, VALU_FMA>("mfma 16x16x32 bf16 (16 clk)", "v_fma_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, VALU_PKFMA>("mfma 16x16x32 bf16", "v_pk_fma_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, VALU_EXP>("mfma 16x16x32 bf16", "v_exp_f32 (quarter rate)", in, out, N, N);
  pair<MFMA_F32, VALU_FMA>("mfma 16x16x4 f32 (32 clk)", "v_fma_f32", in, out, N, 8 * N);
  pair<MFMA_F32, VALU_EXP>("mfma 16x16x4 f32", "v_exp_f32", in, out, N, 2 * N);
  pair<VALU_FMA, VALU_EXP>("v_fma_f32", "v_exp_f32", in, out, 4 * N, N);
  pair<MFMA_BF16, MFMA_BF16>("mfma bf16", "mfma bf16", in, out, N, N);
  pair<MFMA_BF16, VALU_MULLO>("mfma 16x16x32 bf16", "v_mul_lo_u32", in, out, N, N);
  pair<MFMA_BF16, OP_ADD_F32>("mfma bf16", "v_add_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_MUL_F32>("mfma bf16", "v_mul_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_MAX_F32>("mfma bf16", "v_max_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_MED3_F32>("mfma bf16", "v_med3_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_ADD_U32>("mfma bf16", "v_add_u32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_AND_B32>("mfma bf16", "v_and_b32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_CNDMASK>("mfma bf16", "v_cndmask_b32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_MOV>("mfma bf16", "v_mov_b32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_CVT_PK_BF16>("mfma bf16", "v_cvt_pk_bf16_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_LSHL_ADD>("mfma bf16", "v_lshl_add_u32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_PK_ADD_F32>("mfma bf16", "v_pk_add_f32", in, out, N, 4 * N);
  pair<MFMA_BF16, OP_SWAP>("mfma bf16", "v_permlane32_swap", in, out, N, 2 * N);
  pair<MFMA_F32, OP_ADD_U32>("mfma f32", "v_add_u32", in, out, N, 8 * N);
  pair<MFMA_F32, OP_PK_ADD_F32>("mfma f32", "v_pk_add_f32", in, out, N, 8 * N);
  pair<MFMA_F32, VALU_MULLO>("mfma f32", "v_mul_lo_u32", in, out, N, 2 * N);
  pair<VALU_FMA, VALU_MULLO>("v_fma_f32", "v_mul_lo_u32", in, out, 4 * N, N);
  pair<VALU_EXP, VALU_MULLO>("v_exp_f32", "v_mul_lo_u32", in, out, N, N);
  return 0;
}
