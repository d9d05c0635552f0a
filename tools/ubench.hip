// tools/ubench.hip -- single-wave instruction timing on gfx950 (how many shader cycles a latency-bound wave pays per
// dependent VALU / SALU / LDS / mixed instruction), used to tune the MQ coder's straight-line decision code.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench tools/ubench.hip && build/ubench [waves_per_block] [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define N 4096
__global__ void k(int mode, unsigned* out, unsigned long long* clk) {
  __shared__ unsigned lds[1024];
  unsigned a = threadIdx.x + 1, b = threadIdx.x * 7 + 3, c = 5, d = 9;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (i * 13 + 4) & 1020;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (mode == 0) {  // dependent v_add chain
#pragma unroll 16
    for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  } else if (mode == 1) {  // 4 independent chains
#pragma unroll 4
    for (int i = 0; i < N / 4; i++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(b));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b) : "v"(d));
    }
  } else if (mode == 2) {  // dependent LDS read chain
    unsigned p = (threadIdx.x * 4) & 1020;
#pragma unroll 8
    for (int i = 0; i < N; i++) p = *(volatile unsigned*)((char*)lds + p);
    a = p;
  } else if (mode == 3) {  // v_cmp -> vcc -> v_cndmask chain (VALU -> VCC -> VALU)
#pragma unroll 8
    for (int i = 0; i < N / 2; i++) {
      asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    }
  } else if (mode == 4) {  // v_cmp -> sgpr -> s_and -> v_cndmask (VALU -> SALU -> VALU)
#pragma unroll 8
    for (int i = 0; i < N / 3; i++) {
      asm volatile("v_cmp_lt_u32 s[20:21], %1, %0\n s_and_b64 s[20:21], s[20:21], exec\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");
    }
  } else if (mode == 5) {  // dependent SALU chain
    unsigned s = 1;
#pragma unroll 16
    for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
    a = s;
  } else if (mode == 6) {  // not-taken exec branch per 4 VALU
#pragma unroll 8
    for (int i = 0; i < N / 4; i++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (a == 0x12345) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c)); lds[a & 1023] = a; }
    }
  } else if (mode == 7) {  // taken divergent branch (half the lanes) per 4 VALU
#pragma unroll 8
    for (int i = 0; i < N / 4; i++) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if ((threadIdx.x ^ i) & 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c)); }
    }
  } else if (mode == 8) {  // v_add + LDS write + LDS read (independent) per iteration, wait at the end of the iteration
    unsigned p = (threadIdx.x * 4) & 1020, q = 0;
#pragma unroll 8
    for (int i = 0; i < N / 4; i++) {
      q = *(volatile unsigned*)((char*)lds + p);
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(q));
    }
  } else if (mode == 9) {  // v_bfe / v_lshl_add / v_ffbh dependent mix
#pragma unroll 8
    for (int i = 0; i < N / 4; i++) {
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
      asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
      asm volatile("v_bfe_u32 %0, %0, 2, 20" : "+v"(a));
      asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  int threads = argc > 1 ? atoi(argv[1]) : 64, blocks = argc > 2 ? atoi(argv[2]) : 1;
  unsigned* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)threads * blocks * 4); hipMalloc(&clk, 16);
  const char* names[] = {"dependent v_add", "4 independent VALU chains", "dependent LDS read", "v_cmp->vcc->v_cndmask (2 instr)", "v_cmp->s_and->v_cndmask (3 instr)",
                         "dependent s_add", "3 v_add + not-taken branch (cmp+saveexec+cbranch)", "3 v_add + divergent taken branch", "LDS read + 3 v_add, use at end", "ffbh/lshl_add/bfe/min chain"};
  for (int rep = 0; rep < 2; rep++)
    for (int m = 0; m < 10; m++) {
      if (m == 5) continue;  // (the scalar chain compiles into a many-second loop with this toolchain: left out)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, m, out, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      if (rep) printf("%-52s cycles %8llu (%.2f per loop instr unit of %d), refclk ticks %llu -> sclk %.0f MHz if refclk 100 MHz, kernel %.3f ms\n", names[m], h[0], (double)h[0] / N, N, h[1],
                      h[1] ? 100.0 * h[0] / h[1] : 0.0, ms);
    }
  return 0;
}
