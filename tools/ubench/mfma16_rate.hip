// Microbenchmark: issue rate of v_mfma_f32_16x16x32_bf16 in the register pattern of the bf16 conv kernels
// (NT x PT accumulator tiles, one A fragment per PT MFMAs), with 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_rate tools/ubench/mfma16_rate.hip && ./mfma16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NT, int PT, bool LDSA>
__global__ __launch_bounds__(512, 2) void k(const f32x4* in, f32x4* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  f32x4 acc[NT][PT];
  for (int i = 0; i < NT; ++i) for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  bf16x8 a[NT], b[PT];
  for (int i = 0; i < NT; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + i * 64) & 1023]);
  for (int j = 0; j < PT; ++j) b[j] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 7 * j) & 1023]);
  ((f32x4*)lds)[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      bf16x8 av = a[i];
      if (LDSA) av = *(const bf16x8*)(lds + ((threadIdx.x & 63) * 16 + i * 1024 + (it & 7) * 2048));
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[j], acc[i][j], 0, 0, 0);
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int i = 0; i < NT; ++i) for (int j = 0; j < PT; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int PT, bool LDSA>
void run(const char* name, int threads, f32x4* in, f32x4* out) {
  const int iters = 2000, blocks = 256 * (512 / threads) * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, PT, LDSA>), dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * 16 * 16 * 32 * NT * PT * (double)iters * blocks * (threads / 64);
  printf("%-28s threads %d: %.3f ms  %.0f TFLOP/s\n", name, threads, ms, fl / ms / 1e9);
}

int main() {
  f32x4 *in, *out;
  hipMalloc(&in, 1024 * 16); hipMalloc(&out, 1 << 24);
  hipMemset(in, 0x3c, 1024 * 16);
  run<12, 2, false>("NT12 PT2 regs", 512, in, out);
  run<12, 2, false>("NT12 PT2 regs", 256, in, out);
  run<12, 2, true>("NT12 PT2 A from LDS", 512, in, out);
  run<12, 2, true>("NT12 PT2 A from LDS", 256, in, out);
  run<6, 4, false>("NT6 PT4 regs", 512, in, out);
  run<6, 4, true>("NT6 PT4 A from LDS", 512, in, out);
  run<4, 4, true>("NT4 PT4 A from LDS", 512, in, out);
  run<3, 8, true>("NT3 PT8 A from LDS", 512, in, out);
  return 0;
}
