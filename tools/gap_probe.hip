// What separates consecutive dependent launches on one stream?  Every workgroup stamps the 100 MHz wall clock
// at its start and end (atomicMin / atomicMax per launch), so the gap between launch i's last workgroup and launch
// i + 1's first one is read off the device itself, for a matrix of launch forms:
//   plain      hipLaunchKernelGGL
//   ext        hipExtLaunchKernelGGL, no events
//   ext+stop   hipExtLaunchKernelGGL with a stop event per launch (what atsc_compress_plan_dev_pipelined did in round 2)
//   dirty      plain, every workgroup leaves 144 bytes dirty (5.9 MB per launch)
//   2 streams  plain, launches alternate between two streams (no dependency between neighbours)
// and two grid shapes: 40960 workgroups of 64 threads living ~13 us each (k_compress<1,5,false,256>'s shape), and
// 512 workgroups living ~100 us.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/gap_probe tools/gap_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void work(unsigned long long *rec, int launch, unsigned ticks, unsigned char *dirty, int lds_probe)
{
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&rec[2 * launch], t0);
    if (lds_probe) smem[threadIdx.x] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (dirty && threadIdx.x < 18) ((unsigned long long *)(dirty + (size_t)blockIdx.x * 144))[threadIdx.x] = t0;
    if (threadIdx.x == 0) atomicMax(&rec[2 * launch + 1], wall_clock64());
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const int NL = 24;
    unsigned long long *rec;
    unsigned char *dirty;
    CK(hipMalloc(&rec, 2 * NL * sizeof(unsigned long long)));
    CK(hipMalloc(&dirty, 40960 * 144));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(NL);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventReleaseToDevice));
    std::vector<unsigned long long> h(2 * NL);
    struct Shape { unsigned grid, ticks; unsigned lds; const char *name; } shapes[] = {
        {40960, 1300, 7008, "40960 x 13 us, 7 KB LDS"}, {40960, 1300, 0, "40960 x 13 us, no LDS"}, {512, 10000, 0, "512 x 100 us"}};
    const char *forms[] = {"plain", "ext", "ext+stop", "dirty", "2 streams"};
    for (const Shape &sh : shapes)
        for (int f = 0; f < 5; ++f) {
            for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
                for (int i = 0; i < NL; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
                CK(hipMemcpy(rec, h.data(), 2 * NL * sizeof(unsigned long long), hipMemcpyHostToDevice));
                CK(hipDeviceSynchronize());
                for (int i = 0; i < NL; ++i) {
                    hipStream_t st = s[f == 4 ? (i & 1) : 0];
                    unsigned char *d = f == 3 ? dirty : nullptr;
                    if (f == 1)
                        hipExtLaunchKernelGGL(work, dim3(sh.grid), dim3(64), sh.lds, st, nullptr, nullptr, 0, rec, i, sh.ticks, d, sh.lds ? 1 : 0);
                    else if (f == 2)
                        hipExtLaunchKernelGGL(work, dim3(sh.grid), dim3(64), sh.lds, st, nullptr, ev[i], 0, rec, i, sh.ticks, d, sh.lds ? 1 : 0);
                    else
                        hipLaunchKernelGGL(work, dim3(sh.grid), dim3(64), sh.lds, st, rec, i, sh.ticks, d, sh.lds ? 1 : 0);
                }
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), rec, 2 * NL * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> gaps, durs;
            for (int i = 4; i + 1 < NL; ++i) {
                gaps.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) * 0.01);
                durs.push_back(((double)h[2 * i + 1] - (double)h[2 * i]) * 0.01);
            }
            std::sort(gaps.begin(), gaps.end());
            std::sort(durs.begin(), durs.end());
            const double span = ((double)h[2 * (NL - 1) + 1] - (double)h[2 * 4]) * 0.01 / (NL - 4);
            printf("%-26s %-10s kernel %7.1f us (median)  gap end->next start: median %6.2f us  min %6.2f  max %6.2f   per launch %7.1f us\n",
                   sh.name, forms[f], durs[durs.size() / 2], gaps[gaps.size() / 2], gaps.front(), gaps.back(), span);
            fflush(stdout);
        }
    return 0;
}
