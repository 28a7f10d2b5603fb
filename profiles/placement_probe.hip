// Does the streaming rate of a large buffer depend on where the allocation landed?
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe profiles/placement_probe.hip
//   ./placement_probe [GB per buffer = 7] [buffers = 4] [hold = 1]
// Allocates `buffers` buffers one after another (hold = 1: all stay allocated, so each lands somewhere
// else; hold = 0: freed before the next), streams each with a read-only kernel (16 B per lane,
// non-temporal, 8 loads in flight per thread) five times and prints the best and median rate.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stream_kernel(const v2d *__restrict__ p, size_t n_chunks, double *out) {
    // chunk = 256 threads x 8 x 16 B = 32 KB, one per workgroup iteration
    double acc = 0.0;
    for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const v2d *q = p + c * 2048 + threadIdx.x;
        v2d t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(q + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u].x + t[u].y;
    }
    if (acc == 1.2345e300) out[0] = acc;      // never true: keeps the loads
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 7.0;
    const int nbuf = argc > 2 ? atoi(argv[2]) : 4;
    const int hold = argc > 3 ? atoi(argv[3]) : 1;
    const size_t bytes = (size_t)(gb * 1e9) / 32768 * 32768;
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<void *> keep;
    for (int b = 0; b < nbuf; ++b) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %d failed\n", b); break; }
        hipMemset(p, 0, bytes);
        std::vector<float> ms;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, (const v2d *)p, bytes / 32768, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (it > 0) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("buffer %d at %p  %.2f GB: best %.3f ms = %.0f GB/s, median %.3f ms = %.0f GB/s\n", b, p,
               bytes / 1e9, ms.front(), bytes / 1e6 / ms.front(), ms[ms.size() / 2], bytes / 1e6 / ms[ms.size() / 2]);
        if (hold) keep.push_back(p); else hipFree(p);
    }
    for (void *p : keep) hipFree(p);
    return 0;
}
