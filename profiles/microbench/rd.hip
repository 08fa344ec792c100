// read-stream experiment: the load pattern of k_flux2_update (tiles marching along k, NF field arrays), rows of 64 lanes
// starting on a 128-byte line (stride 64) or not (stride 63, as the kernel's overlapping tiles)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int STRIDE, int NF, int TJ>
__global__ void __launch_bounds__(64*TJ) k(const double *a, long nc, int sJ, long sK, int ni, int nj, int nk, int kc, double *sink)
{
  const int lane = threadIdx.x, row = threadIdx.y;
  const int i = 16 + blockIdx.x*STRIDE + lane, j = blockIdx.y*(STRIDE == 64 ? TJ : TJ - 1) + row, k0 = blockIdx.z*kc;
  if (i >= ni + 16 || j >= nj) return;
  double acc = 0.0;
  for (int k = k0; k < k0 + kc && k < nk; k++) {
    const long m = (long)k*sK + (long)j*sJ + i;
#pragma unroll
    for (int f = 0; f < NF; f++) acc += a[(long)f*nc + m];
  }
  if (acc == 1.2345e-300) sink[0] = acc;
}
template <int STRIDE, int NF, int TJ> void run(const char *name, double *a, long nc, int sJ, long sK, int n, double *sink) {
  dim3 grid((n + STRIDE - 1)/STRIDE, (n + (STRIDE == 64 ? TJ : TJ - 1) - 1)/(STRIDE == 64 ? TJ : TJ - 1), n/32), blk(64, TJ);
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<STRIDE, NF, TJ>), grid, blk, 0, 0, a, nc, sJ, sK, n, n, n, 32, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-44s %7.2f ms  %6.0f GB/s of zone data\n", name, ms, (double)n*n*n*NF*8/ms*1e-6);
  }
}
int main() {
  const int n = 512, sJ = 544; const long sK = (long)sJ*520, nc = sK*520;
  double *a, *sink;
  if (hipMalloc(&a, (size_t)36*nc*8 + (1<<20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 64);
  hipMemset(a, 0, (size_t)36*nc*8);
  run<64, 36, 8>("read 36 fields, aligned 64 x 8 tiles", a, nc, sJ, sK, n, sink);
  run<63, 36, 8>("read 36 fields, 63 x 7 of 64 x 8 (overlap)", a, nc, sJ, sK, n, sink);
  run<64, 36, 4>("read 36 fields, aligned 64 x 4 tiles", a, nc, sJ, sK, n, sink);
  run<63, 36, 4>("read 36 fields, 63 x 3 of 64 x 4 (overlap)", a, nc, sJ, sK, n, sink);
  return 0;
}
