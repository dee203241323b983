// pmc_probe -- the smallest HIP program (one streaming kernel, default stream) used to tell whether
// `rocprofv3 --pmc` works at all on a GPU box, independent of libsfx (profiles/README.md).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_axpy(int n, float a, const float* __restrict__ x, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + y[i];
}

int main() {
    const int n = 1 << 24;
    float *x = nullptr, *y = nullptr;
    if (hipMalloc(&x, n * sizeof(float)) != hipSuccess || hipMalloc(&y, n * sizeof(float)) != hipSuccess) return 1;
    hipMemset(x, 0, n * sizeof(float));
    hipMemset(y, 0, n * sizeof(float));
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_axpy, dim3(n / 256), dim3(256), 0, 0, n, 2.0f, x, y);
    const hipError_t e = hipDeviceSynchronize();
    printf("pmc_probe: %s\n", hipGetErrorString(e));
    hipFree(x); hipFree(y);
    return e == hipSuccess ? 0 : 1;
}
