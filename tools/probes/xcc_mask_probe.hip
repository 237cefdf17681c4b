#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(unsigned *out)
{
    extern __shared__ char lds[];
    lds[threadIdx.x] = 0;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    // stay a while so that every workgroup needs a CU of its own
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000) {}
}
int main()
{
    unsigned *d; hipMalloc(&d, 2 * 256 * 4);
    for (int pat = 0; pat < 6; pat++) {
        unsigned mask[8] = {0};
        for (int i = 0; i < 256; i++) {
            bool on = pat == 0 ? (i < 128) : pat == 1 ? ((i % 8) < 4) : pat == 2 ? ((i / 32) < 4) : pat == 3 ? (i < 32) : pat == 4 ? (i % 8 == 0) : true;
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask stream failed\n"); return 1; }
        hipMemset(d, 0xff, 2 * 256 * 4);
        hipFuncSetAttribute((const void *)who, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(who, dim3(256), dim3(64), 150 * 1024, s, d);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(who, dim3(256), dim3(64), 150 * 1024, s, d);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> h(512);
        hipMemcpy(h.data(), d, 2 * 256 * 4, hipMemcpyDeviceToHost);
        int hist[8] = {0};
        for (int b = 0; b < 256; b++) hist[h[2 * b] & 7]++;
        printf("pattern %d: %.2f ms for 256 workgroups of 2 ms; per XCC:", pat, ms);
        for (int x = 0; x < 8; x++) printf(" %d", hist[x]);
        printf("   first hw ids: %08x %08x %08x\n", h[1], h[3], h[5]);
        hipStreamDestroy(s);
    }
    return 0;
}
