// instruction issue rates on gfx950: N dependent-free instructions per loop iteration, one wave per SIMD and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP> __global__ void k(double *out, long *cyc, int iters) {
  double a0 = threadIdx.x*1.0 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double b = out[threadIdx.x & 7];
  unsigned long long vc;
  long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 2) { REP8(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 3) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %8\n v_cmp_lt_f64 vcc, %1, %8\n v_cmp_lt_f64 vcc, %2, %8\n v_cmp_lt_f64 vcc, %3, %8\n v_cmp_lt_f64 vcc, %4, %8\n v_cmp_lt_f64 vcc, %5, %8\n v_cmp_lt_f64 vcc, %6, %8\n v_cmp_lt_f64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(*(int*)&a0), "+v"(*(int*)&a1), "+v"(*(int*)&a2), "+v"(*(int*)&a3), "+v"(*(int*)&a4), "+v"(*(int*)&a5), "+v"(*(int*)&a6), "+v"(*(int*)&a7) : "v"(*(int*)&b) : "vcc");) }
    if (OP == 5) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 6) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 7) { REP8(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 8) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 3, %8\n v_lshl_add_u64 %1, %1, 3, %8\n v_lshl_add_u64 %2, %2, 3, %8\n v_lshl_add_u64 %3, %3, 3, %8\n v_lshl_add_u64 %4, %4, 3, %8\n v_lshl_add_u64 %5, %5, 3, %8\n v_lshl_add_u64 %6, %6, 3, %8\n v_lshl_add_u64 %7, %7, 3, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 9) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(*(int*)&a0), "+v"(*(int*)&a1), "+v"(*(int*)&a2), "+v"(*(int*)&a3), "+v"(*(int*)&a4), "+v"(*(int*)&a5), "+v"(*(int*)&a6), "+v"(*(int*)&a7) : "v"(*(int*)&b));) }
    if (OP == 10) { REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 11) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %0, %0, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
  }
  long t1 = clock64();
  out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char *name, double *out, long *cyc) {
  for (int waves = 1; waves <= 2; waves++) {
    int threads = 256*waves;            // 4 SIMDs x waves
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 2000);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 2000); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double ninstr = 2000.0*64;
    printf("%-14s waves/SIMD %d: %.2f clock64-ticks per instr per wave (x%d waves); %.3f ms => %.2f ns per instr per SIMD\n", name, waves, c/ninstr, waves, ms, ms*1e6/(ninstr*waves));
  }
}
int main() {
  double *out; long *cyc; hipMalloc(&out, 1<<22); hipMalloc(&cyc, 64); hipMemset(out, 0, 1<<22);
  run<0>("v_add_f64", out, cyc); run<5>("v_mul_f64", out, cyc); run<1>("v_fma_f64", out, cyc); run<11>("v_fma_f64 dep", out, cyc); run<2>("v_min_f64", out, cyc);
  run<3>("v_cmp_lt_f64", out, cyc); run<4>("v_cndmask_b32", out, cyc); run<6>("v_rcp_f64", out, cyc); run<10>("v_rsq_f64", out, cyc);
  run<7>("v_mov_b64", out, cyc); run<8>("v_lshl_add_u64", out, cyc); run<9>("v_add_u32", out, cyc);
  return 0;
}
