// Prints how many 64-thread workgroups fit a CU for a range of dynamic LDS sizes (hipOccupancy API):
// shows the LDS allocation steps that bound the frames-in-flight of k_compress<1,*>.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void probe(float *p)
{
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    p[threadIdx.x] = smem[63 - threadIdx.x];
}
int main()
{
    int last = -1;
    for (int lds = 2048; lds <= 12288; lds += 16) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 64, lds) != hipSuccess) return 1;
        if (nb != last) printf("lds %5d B -> %d workgroups/CU\n", lds, nb);
        last = nb;
    }
    return 0;
}
