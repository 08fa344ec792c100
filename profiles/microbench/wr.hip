// write-stream experiment: the store pattern of k_correct_all (tiles of 64 x 4 zones marching along k, NF field arrays)
// with (A) one array per field (SoA, fields 1.1 GB apart) and (B) fields interleaved per 64-zone group (AoSoA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int LAYOUT, int NF, bool RD>
__global__ void __launch_bounds__(256) k(double *a, long nc, int sJ, long sK, int ni, int nj, int nk, int kc, double *sink)
{
  const int lane = threadIdx.x, row = threadIdx.y;
  const int i = blockIdx.x*64 + lane, j = blockIdx.y*4 + row, k0 = blockIdx.z*kc;
  if (i >= ni || j >= nj) return;
  double acc = 0.0;
  for (int k = k0; k < k0 + kc && k < nk; k++) {
    const long m = (long)k*sK + (long)j*sJ + i;
#pragma unroll
    for (int f = 0; f < NF; f++) {
      long idx;
      if (LAYOUT == 0) idx = (long)f*nc + m;
      else idx = ((m >> 6)*NF + f)*64 + (m & 63);
      if (RD) acc += a[idx]; else a[idx] = (double)(f + lane);
    }
  }
  if (RD && acc == 1.2345e-300) sink[0] = acc;
}
template <int LAYOUT, int NF, bool RD> void run(const char *name, double *a, long nc, int sJ, long sK, int n, double *sink) {
  dim3 grid(n/64, n/4, n/32), blk(64, 4);
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LAYOUT, NF, RD>), grid, blk, 0, 0, a, nc, sJ, sK, n, n, n, 32, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-40s %7.2f ms  %6.0f GB/s\n", name, ms, (double)n*n*n*NF*8/ms*1e-6);
  }
}
int main() {
  const int n = 512, sJ = 528; const long sK = (long)sJ*520, nc = sK*520;
  double *a, *sink; 
  if (hipMalloc(&a, (size_t)36*nc*8 + (1<<20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 64);
  hipMemset(a, 0, (size_t)36*nc*8);
  run<0, 36, false>("write 36 fields SoA", a, nc, sJ, sK, n, sink);
  run<1, 36, false>("write 36 fields AoSoA-64", a, nc, sJ, sK, n, sink);
  run<0, 12, false>("write 12 fields SoA", a, nc, sJ, sK, n, sink);
  run<1, 12, false>("write 12 fields AoSoA-64", a, nc, sJ, sK, n, sink);
  run<0, 36, true>("read 36 fields SoA", a, nc, sJ, sK, n, sink);
  run<1, 36, true>("read 36 fields AoSoA-64", a, nc, sJ, sK, n, sink);
  run<0, 6, false>("write 6 fields SoA", a, nc, sJ, sK, n, sink);
  run<0, 6, true>("read 6 fields SoA", a, nc, sJ, sK, n, sink);
  return 0;
}
