#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void k(double* p, int n) { if (threadIdx.x < n) p[threadIdx.x] += 1.0; }
__global__ void pub(const double* src, double* hostmapped, volatile unsigned long long* flag, unsigned long long seq, int n) {
    if (threadIdx.x < n) hostmapped[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *flag = seq; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    if (mode == 1) printf("setflags spin -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleSpin));
    if (mode == 2) printf("setflags yield -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleYield));
    if (mode == 3) printf("setflags blocking -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    double *d, *pinned, *mapped, *mapped_dev; unsigned long long *flag, *flag_dev;
    hipMalloc(&d, 1024 * 8); hipMemset(d, 0, 1024 * 8);
    hipHostMalloc(&pinned, 1024 * 8, hipHostMallocDefault);
    hipHostMalloc(&mapped, 1024 * 8, hipHostMallocMapped); hipHostGetDevicePointer((void**)&mapped_dev, mapped, 0);
    hipHostMalloc(&flag, 64, hipHostMallocMapped); hipHostGetDevicePointer((void**)&flag_dev, flag, 0); *flag = 0;
    hipStream_t s = 0;
    const int iters = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d, 64); hipStreamSynchronize(s); }
        double t0 = now();
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d, 64);
            if (variant == 0) { hipMemcpyAsync(pinned, d, 100 * 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            else if (variant == 1) { hipMemcpy(pinned, d, 100 * 8, hipMemcpyDeviceToHost); }
            else { unsigned long long seq = (unsigned long long)it + 1;
                   hipLaunchKernelGGL(pub, dim3(1), dim3(128), 0, s, d, mapped_dev, flag_dev, seq, 100);
                   while (*(volatile unsigned long long*)flag != seq) { } }
        }
        double t1 = now();
        const char* names[] = {"memcpyAsync+streamSync", "hipMemcpy", "publish kernel + host spin"};
        printf("mode %d  %-28s %.2f us per launch+fetch\n", mode, names[variant], (t1 - t0) / iters);
    }
    return 0;
}
