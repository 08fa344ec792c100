#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP> __global__ void k(int *out, long *cyc, int iters) {
  int a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b = out[threadIdx.x & 7];
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = b + 0.5;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 1) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 2) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 3) { REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27");) }
    if (OP == 4) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %5, %5, %6, vcc\n v_cmp_lt_f64 vcc, %1, %4\n s_nop 1\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_lt_f64 vcc, %2, %4\n s_nop 1\n v_cndmask_b32 %7, %7, %8, vcc\n v_cmp_lt_f64 vcc, %3, %4\n s_nop 1\n v_cndmask_b32 %8, %8, %5, vcc" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
    if (OP == 5) { REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" ::: "memory");) }
    if (OP == 6) { REP8(asm volatile("s_add_u32 s20, s20, s21\n s_add_u32 s22, s22, s21\n s_add_u32 s23, s23, s21\n s_add_u32 s24, s24, s21\n s_add_u32 s25, s25, s21\n s_add_u32 s26, s26, s21\n s_add_u32 s27, s27, s21\n s_add_u32 s28, s28, s21" ::: "s20","s21","s22","s23","s24","s25","s26","s27","s28","scc");) }
    if (OP == 7) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %4\n s_add_u32 s20, s20, s21\n v_fma_f64 %1, %1, %4, %4\n s_add_u32 s22, s22, s21\n v_fma_f64 %2, %2, %4, %4\n s_add_u32 s23, s23, s21\n v_fma_f64 %3, %3, %4, %4\n s_add_u32 s24, s24, s21" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db) : "s20","s21","s22","s23","s24","scc");) }
  }
  out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(d0 + d1 + d2 + d3);
}
template <int OP> void run(const char *name, int *out, long *cyc, int per) {
  for (int waves = 1; waves <= 2; waves++) {
    int threads = 256*waves;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 4000);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 4000); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ninstr = 4000.0*8*per;
    printf("%-22s waves/SIMD %d: %.3f ms => %.2f ns per instr per SIMD\n", name, waves, ms, ms*1e6/(ninstr*waves));
  }
}
int main() {
  int *out; long *cyc; hipMalloc(&out, 1<<22); hipMalloc(&cyc, 64); hipMemset(out, 0, 1<<22);
  run<0>("v_cndmask_b32 vcc", out, cyc, 8); run<1>("v_add_u32", out, cyc, 8); run<2>("v_mov_b32", out, cyc, 8); run<3>("v_readlane_b32", out, cyc, 8);
  run<4>("cmp_f64+nop1+cndmask", out, cyc, 4); run<5>("s_nop 0", out, cyc, 8); run<6>("s_add_u32", out, cyc, 8); run<7>("fma_f64 + s_add pair", out, cyc, 4);
  return 0;
}
