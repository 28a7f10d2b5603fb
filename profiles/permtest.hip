#include <hip/hip_runtime.h>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]);
    b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void swap16(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]);
    b = __hiloint2double(r1[1], r0[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0u, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0u, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double *x, double *y) {
    double a = x[threadIdx.x], b = x[64 + threadIdx.x];
    double a2 = a, b2 = b;
    swap32(a, b);
    swap16(a2, b2);
    y[threadIdx.x] = a; y[64 + threadIdx.x] = b;
    y[128 + threadIdx.x] = a2; y[192 + threadIdx.x] = b2;
    y[256 + threadIdx.x] = dpp<0xB1>(x[threadIdx.x]);     // quad_perm [1,0,3,2]
    y[320 + threadIdx.x] = dpp<0x4E>(x[threadIdx.x]);     // quad_perm [2,3,0,1]
    y[384 + threadIdx.x] = dpp<0x141>(x[threadIdx.x]);    // row_half_mirror
    y[448 + threadIdx.x] = dpp<0x128>(x[threadIdx.x]);    // row_ror:8
}
#include <cstdio>
#include <vector>
int main() {
    std::vector<double> x(128), y(512);
    for (int i = 0; i < 64; ++i) { x[i] = i; x[64 + i] = 100 + i; }
    double *dx, *dy;
    hipMalloc(&dx, 128 * 8); hipMalloc(&dy, 512 * 8);
    hipMemcpy(dx, x.data(), 128 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy);
    hipMemcpy(y.data(), dy, 512 * 8, hipMemcpyDeviceToHost);
    const char *names[] = {"swap32 a", "swap32 b", "swap16 a", "swap16 b", "xor1", "xor2", "half_mirror", "ror8"};
    for (int r = 0; r < 8; ++r) {
        printf("%-12s", names[r]);
        for (int i = 0; i < 64; ++i) printf(" %g", y[64 * r + i]);
        printf("\n");
    }
    return 0;
}
