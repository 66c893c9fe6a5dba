// tools/libm_probe.hip -- diagnostic (not product, not test): how often do the device float transcendentals used by
// GetRayOnHemisphere (kernel.cu:101-122) differ bitwise from the host libm the oracle uses?
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/libm_probe.hip -o gpurun_out/libm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

enum { F_POW = 0, F_POW_D, F_ACOS, F_ACOS_D, F_SIN, F_SIN_D, F_COS, F_COS_D, F_N };

__global__ void k(const float* x, const float* y, float* out, int n, int fn)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], r = 0;
    switch (fn) {
    case F_POW: r = powf(a, b); break;
    case F_POW_D: r = (float)pow((double)a, (double)b); break;
    case F_ACOS: r = acosf(a); break;
    case F_ACOS_D: r = (float)acos((double)a); break;
    case F_SIN: r = sinf(a); break;
    case F_SIN_D: r = (float)sin((double)a); break;
    case F_COS: r = cosf(a); break;
    case F_COS_D: r = (float)cos((double)a); break;
    }
    out[i] = r;
}

static float host(int fn, float a, float b)
{
    switch (fn) {
    case F_POW: case F_POW_D: return powf(a, b);
    case F_ACOS: case F_ACOS_D: return acosf(a);
    case F_SIN: case F_SIN_D: return sinf(a);
    default: return cosf(a);
    }
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> x(n), y(n), o(n);
    float *dx, *dy, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, n * 4);
    const char* names[F_N] = {"powf", "pow(double)->float", "acosf", "acos(double)->float", "sinf", "sin(double)->float", "cosf", "cos(double)->float"};
    const float expos[4] = {1.f / 101.f, 1.f / 1001.f, 1.f / 10001.f, 1.f / 500001.f};
    for (int fn = 0; fn < F_N; ++fn) {
        int variants = (fn <= F_POW_D) ? 4 : (fn <= F_ACOS_D ? 2 : 1);
        for (int v = 0; v < variants; ++v) {
            srand(1234);
            for (int i = 0; i < n; ++i) {
                float u = (float)(rand() & 0xFFFFFF) / (float)0x1000000;
                if (fn <= F_POW_D) { x[i] = 1.f - u; y[i] = expos[v]; }
                else if (fn <= F_ACOS_D) { x[i] = v == 0 ? u : powf(1.f - u, 1.f / 1001.f); y[i] = 0; }
                else { x[i] = 2.f * 3.14159265358979323846f * u; y[i] = 0; }
            }
            hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
            hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n, fn);
            hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
            int diff = 0, maxulp = 0;
            for (int i = 0; i < n; ++i) {
                float h = host(fn, x[i], y[i]);
                int a, b; memcpy(&a, &h, 4); memcpy(&b, &o[i], 4);
                if (a != b) { ++diff; int d = abs(a - b); if (d > maxulp) maxulp = d; }
            }
            printf("%-22s variant %d: %.4f%% differ from host libm, max %d ulp\n", names[fn], v, 100.0 * diff / n, maxulp);
        }
    }
    return 0;
}
