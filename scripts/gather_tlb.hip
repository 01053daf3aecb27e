// Microbenchmark: dependent random 64-byte loads as a function of the footprint (TLB reach / HBM channel effects on gfx950).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_tlb scripts/gather_tlb.hip && /tmp/gather_tlb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void chase(const uint4 *__restrict__ mem, uint64_t n_blocks, int steps, uint32_t *out, int dependent)
{
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t x = t * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t b = (x >> 20) % n_blocks;
        uint4 v = mem[b * 4];
        acc += v.x;
        if (dependent) x += v.y;              // memory holds zeros: the value does not change the sequence, the dependency stays
    }
    out[t] = acc;
}
int main()
{
    const int lanes = 1 << 21, steps = 64;
    uint32_t *out; hipMalloc(&out, lanes * 4);
    for (int dep = 1; dep >= 0; dep--)
        for (int gib : {1, 2, 4, 8, 16, 32, 64, 128}) {
            uint4 *mem; size_t bytes = (size_t)gib << 30;
            if (hipMalloc(&mem, bytes) != hipSuccess) { printf("alloc %d GiB failed\n", gib); continue; }
            hipMemset(mem, 0, bytes);
            uint64_t n_blocks = bytes / 64;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(chase, dim3(lanes / 256), dim3(256), 0, 0, mem, n_blocks, steps, out, dep);
            hipEventRecord(a);
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(chase, dim3(lanes / 256), dim3(256), 0, 0, mem, n_blocks, steps, out, dep);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
            double acc = (double)lanes * steps;
            printf("%s footprint %3d GiB: %.2f ms -> %.1f G loads/s, %.0f GB/s\n", dep ? "dependent  " : "independent", gib, ms, acc / ms / 1e6, acc * 64 / ms / 1e6);
            fflush(stdout);
            hipFree(mem);
        }
    return 0;
}
