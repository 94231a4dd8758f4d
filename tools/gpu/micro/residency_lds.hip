// Microbenchmark: one-wave workgroups resident per CU as a function of the static LDS size -- i.e. the LDS allocation granule of gfx950
// (every frame kernel's occupancy is set by its arena: which byte counts are the steps?).  Each workgroup sleeps for a fixed number of
// cycles; resident = grid * wave_time / kernel_time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS_BYTES>
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long spin, unsigned *out) {
    __shared__ unsigned lds[LDS_BYTES / 4];
    unsigned long long t0 = __builtin_readcyclecounter();
    lds[threadIdx.x] = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(2);
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)(t1 - t0) + (lds[5] & 0);
}
static double ticks_per_us = 100.0;
template <int LDS_BYTES>
void run(unsigned *d_out) {
    const int grid = 65536;
    const unsigned long long S = 48000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((spin_kernel<LDS_BYTES>), dim3(grid), dim3(64), 0, 0, S, d_out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    std::vector<unsigned> h(grid); hipMemcpy(h.data(), d_out, grid * 4, hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < grid; i++) cyc += h[i];
    cyc /= grid;
    printf("lds %6d B: %8.1f us  mean wave %7.0f ticks -> %5.1f workgroups / CU\n", LDS_BYTES, ms * 1e3, cyc, grid * cyc / ticks_per_us / (ms * 1e3) / 256);
}
__global__ void clock_kernel(unsigned long long *o) {
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    while (wall_clock64() - r0 < 100000) {}
    o[0] = __builtin_readcyclecounter() - c0; o[1] = wall_clock64() - r0;
}
int main() {
    unsigned long long *o; hipMalloc(&o, 16); hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, o); unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    ticks_per_us = 100.0 * h[0] / h[1];
    printf("readcyclecounter: %.1f ticks per us\n", ticks_per_us);
    unsigned *d_out; hipMalloc(&d_out, 65536 * 4);
    run<2048>(d_out); run<2384>(d_out); run<2560>(d_out); run<2640>(d_out); run<3072>(d_out); run<3840>(d_out); run<4096>(d_out);
    run<4608>(d_out); run<5120>(d_out); run<5124>(d_out); run<5376>(d_out); run<5632>(d_out); run<5636>(d_out); run<5848>(d_out); run<5888>(d_out); run<5968>(d_out);
    run<6144>(d_out); run<6148>(d_out); run<6224>(d_out); run<6400>(d_out); run<6404>(d_out); run<6656>(d_out); run<7168>(d_out); run<7680>(d_out); run<7684>(d_out); run<8192>(d_out);
    run<8196>(d_out); run<8960>(d_out); run<8964>(d_out); run<9728>(d_out); run<10000>(d_out); run<10240>(d_out); run<10244>(d_out);
    return 0;
}
