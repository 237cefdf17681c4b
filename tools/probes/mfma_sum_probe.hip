// Which lanes does v_mfma_f64_4x4x4_4b_f64 contract?  A = per-lane value, B = 1: D per lane after one and after two
// applications (the flat build kernel's wave sum wants: two applications = the sum over each block of 16 lanes).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_sum_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double *out1, double *out2, int mode)
{
    const int lane = threadIdx.x;
    const double v = mode == 0 ? (double)(1 << (lane & 15)) : (double)lane;   // mode 0: which lanes of a block were summed (bit set)
    const double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    const double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(d1, 1.0, 0.0, 0, 0, 0);
    out1[lane] = d1;
    out2[lane] = d2;
}
int main()
{
    double *o1, *o2, h1[64], h2[64];
    hipMalloc(&o1, 512); hipMalloc(&o2, 512);
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o1, o2, mode);
        hipMemcpy(h1, o1, 512, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; l++) printf("lane %2d: one %8.0f (0x%04x)  two %8.0f (0x%05x)\n", l, h1[l], (unsigned)h1[l], h2[l], (unsigned)h2[l]);
    }
    return 0;
}
