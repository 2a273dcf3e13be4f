// How fast does the GPU start workgroups?  (dev aid: hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_probe tools/dispatch_probe.hip)
// Launches grids of trivial workgroups shaped like the headline kernel's (64 threads, 7008 bytes of LDS, 40960 of them) and
// variants, and prints the kernel time: the floor a launch of that many workgroups has whatever they compute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SPIN>
__global__ void k_probe(unsigned *out, int lds_words)
{
    extern __shared__ unsigned sm[];
    if (lds_words) sm[threadIdx.x] = threadIdx.x;
    if (SPIN) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < SPIN) {}
    }
    if (out && blockIdx.x == 0x7fffffff) out[0] = sm[0];
}

template <int SPIN>
static float run(int grid, int threads, int lds, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k_probe<SPIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_probe<SPIN>, dim3(grid), dim3(threads), lds, 0, nullptr, lds / 4);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_probe<SPIN>, dim3(grid), dim3(threads), lds, 0, nullptr, lds / 4);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best * 1000.0f;
}

int main()
{
    struct { int grid, threads, lds; } cfg[] = {
        {40960, 64, 7008}, {40960, 64, 0}, {40960, 64, 2048}, {40960, 64, 16384},
        {20480, 128, 14016}, {10240, 256, 28032}, {81920, 64, 7008}, {163840, 64, 0}, {40960, 256, 0}, {40960, 1024, 0},
    };
    for (auto &c : cfg) {
        const float t0 = run<0>(c.grid, c.threads, c.lds, 10);
        const float t1 = run<1000>(c.grid, c.threads, c.lds, 5);   // every wave lives 10 us (100 MHz clock)
        printf("grid %6d x %4d threads, %5d B LDS: empty %7.1f us (%.0f workgroups/us)   10-us waves %7.1f us\n", c.grid, c.threads, c.lds,
               t0, c.grid / t0, t1);
    }
    return 0;
}
