// tools/sincos_probe.hip -- diagnostic: is device sincosf(x) bit-identical to (sinf(x), cosf(x)) on the ranges the
// hemisphere sampler uses (theta in [0, pi/2], phi in [0, 2 pi))?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, int n, unsigned int* diff)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincosf(x[i], &s, &c);
    if (__float_as_uint(s) != __float_as_uint(sinf(x[i]))) atomicAdd(&diff[0], 1u);
    if (__float_as_uint(c) != __float_as_uint(cosf(x[i]))) atomicAdd(&diff[1], 1u);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = 6.2831855f * (float)i / (float)n;
    float* dx; unsigned int* dd; unsigned int h[2] = {0, 0};
    hipMalloc(&dx, n * 4); hipMalloc(&dd, 8);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dd, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, dd);
    hipMemcpy(h, dd, 8, hipMemcpyDeviceToHost);
    printf("sincosf vs sinf: %u differ, vs cosf: %u differ, of %d\n", h[0], h[1], n);
    return 0;
}
