// Microbenchmark: how many one-wave workgroups does the dispatcher keep resident?  Each workgroup spins for `spin` core
// cycles (s_memtime) and touches `LDS` bytes of shared memory; kernel time vs grid size gives the resident count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int LDS_BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void spin_kernel(unsigned long long spin, unsigned *out, int work) {
    __shared__ unsigned lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = threadIdx.x;
    lds[threadIdx.x] = acc;
    if (work == 0) {
        while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(2);
    } else {
        // dependent LDS round trips: latency-bound work like the rule code
        for (unsigned long long k = 0; k < spin; k++) { acc = lds[(acc + k) & 63] + 1; lds[threadIdx.x] = acc; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[65536 * 2] = (unsigned)(t1 - t0); }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (unsigned)(t1 - t0); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); out[blockIdx.x * 2 + 1] = hw + (acc & 0); }
}
template <int LDS_BYTES, int THREADS>
void run(const char *name, int grid, unsigned long long spin, int work, unsigned *d_out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((spin_kernel<LDS_BYTES, THREADS>), dim3(grid), dim3(THREADS), 0, 0, spin, d_out, work);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(grid * 2); hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < grid; i++) cyc += h[i * 2];
    cyc /= grid;
    // resident = grid * wave_time / kernel_time; wave_time in us needs the clock: report both cycles and the implied count at 2.4 GHz
    printf("%-28s grid %6d threads %3d lds %6d: %8.1f us, mean wave %8.0f cyc -> resident %7.0f (@2.4GHz) = %5.1f / CU\n", name, grid, THREADS, LDS_BYTES, ms * 1e3, cyc,
           grid * cyc / 2400.0 / (ms * 1e3), grid * cyc / 2400.0 / (ms * 1e3) / 256);
}
__global__ void clock_kernel(unsigned long long *o) {
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    while (wall_clock64() - r0 < 100000) {}  // 1 ms of the 100 MHz constant clock
    o[0] = __builtin_readcyclecounter() - c0; o[1] = wall_clock64() - r0;
}
int main() {
    { unsigned long long *o; hipMalloc(&o, 16); hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, o); unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
      printf("s_memtime ticks per 100 MHz tick: %.3f (-> %.0f MHz)\n", (double)h[0] / h[1], 100.0 * h[0] / h[1]); }
    unsigned *d_out; hipMalloc(&d_out, 65536 * 8 * 4);
    const unsigned long long S = 48000;
    run<0, 64>("spin no-lds", 65536, S, 0, d_out);
    run<5000, 64>("spin lds5k", 65536, S, 0, d_out);
    run<10000, 64>("spin lds10k", 65536, S, 0, d_out);
    run<10000, 64>("spin lds10k grid4096", 4096, S, 0, d_out);
    run<10000, 64>("spin lds10k grid8192", 8192, S, 0, d_out);
    run<20000, 64>("spin lds20k", 65536, S, 0, d_out);
    run<40000, 256>("spin lds40k 4 waves", 16384, S, 0, d_out);
    run<10000, 64>("ldschain lds10k", 65536, 400, 1, d_out);
    run<10000, 64>("short spin lds10k", 65536, 4800, 0, d_out);
    run<0, 64>("short spin no-lds", 65536, 4800, 0, d_out);
    return 0;
}
