// Microbenchmark: how fast can a CU fill LDS from an L2-resident global buffer?
//   mode 0: LDS-DMA, global_load_lds_dwordx4 (1 KB per wave instruction), DEPTH instructions in flight per wave
//   mode 1: global_load_dwordx4 into VGPRs + ds_write_b128 (the classic path), DEPTH loads in flight per wave
//   mode 2: LDS-DMA of dwords (global_load_lds_dword, 256 B per wave instruction)
// One workgroup per CU slot (grid = 256 * WGS), WAVES waves each; the source is a 24 KB "weight chunk" shared by every
// workgroup (as in the conv kernels) or a private 64 KB tile per workgroup (activation-like).
//   hipcc --offload-arch=gfx950 -O3 -o lds_fill_rate tools/ubench/lds_fill_rate.hip && ./lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void fill(const char* src, float* out, int iters, int src_bytes, int private_src) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + (private_src ? (size_t)blockIdx.x * src_bytes : 0);
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;
  const int pieces = src_bytes / 1024;           // 1 KB pieces; wave w takes pieces w, w + nw, ...
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int p0 = w; p0 < pieces; p0 += nw * DEPTH) {
      if (MODE == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int p = p0 + d * nw;
          if (p < pieces) glds16(base, (unsigned)(p * 1024 + lane * 16), lds0 + p * 1024);
        }
      } else if (MODE == 2) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int p = p0 + d * nw;
          if (p < pieces)
#pragma unroll
            for (int q = 0; q < 4; ++q) glds4(base, (unsigned)(p * 1024 + q * 256 + lane * 4), lds0 + p * 1024 + q * 256);
        }
      } else {
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int p = min(p0 + d * nw, pieces - 1);
          v[d] = *(const f32x4*)(base + p * 1024 + lane * 16);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int p = min(p0 + d * nw, pieces - 1);
          *(f32x4*)(lds + p * 1024 + lane * 16) = v[d];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    sink += ((const float*)lds)[(threadIdx.x * 4 + it) & 1023];
    __syncthreads();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int MODE, int DEPTH>
static void run(const char* name, int waves, int wgs_per_cu, int src_bytes, int private_src, const char* src, float* out) {
  const int iters = 200, grid = 256 * wgs_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)fill<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  fill<MODE, DEPTH><<<grid, waves * 64, src_bytes, 0>>>(src, out, 5, src_bytes, private_src);
  hipEventRecord(a);
  fill<MODE, DEPTH><<<grid, waves * 64, src_bytes, 0>>>(src, out, iters, src_bytes, private_src);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes_per_cu = (double)iters * wgs_per_cu * src_bytes, clk = ms * 1e-3 * 2.4e9;
  printf("%-10s depth %d  waves %2d  wg/cu %d  src %3d KB %-7s : %6.1f B/clk/CU (at 2.4 GHz)  %6.2f TB/s chip   err=%d\n", name, DEPTH, waves,
         wgs_per_cu, src_bytes / 1024, private_src ? "private" : "shared", bytes_per_cu / clk, bytes_per_cu * 256 / (ms * 1e-3) / 1e12,
         (int)hipGetLastError());
}

int main() {
  char* src; float* out;
  hipMalloc(&src, (size_t)512 * 64 * 1024 + 65536);
  hipMemset(src, 1, (size_t)512 * 64 * 1024 + 65536);
  hipMalloc(&out, 4 * 1024 * 1024);
  for (int priv = 0; priv < 2; ++priv) {
    const int sb = priv ? 64 * 1024 : 24 * 1024;
    for (int waves : {4, 8, 16}) {
      run<0, 1>("dma x4", waves, 1, sb, priv, src, out);
      run<0, 3>("dma x4", waves, 1, sb, priv, src, out);
      run<0, 6>("dma x4", waves, 1, sb, priv, src, out);
      run<1, 3>("vgpr+dsw", waves, 1, sb, priv, src, out);
      run<1, 6>("vgpr+dsw", waves, 1, sb, priv, src, out);
      run<2, 3>("dma x1", waves, 1, sb, priv, src, out);
    }
    run<0, 6>("dma x4", 8, 2, sb, priv, src, out);
    run<1, 6>("vgpr+dsw", 8, 2, sb, priv, src, out);
  }
  return 0;
}
