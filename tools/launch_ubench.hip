// Back-to-back dependent launches on one stream: time per launch for an empty kernel, a kernel with one
// global round trip, and one with three dependent round trips (the shape of a Cholesky block step).
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_ubench.hip -o tools/launch_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(double*) {}
__global__ void k_one(double* p) { if (threadIdx.x == 0) p[0] += 1.0; }
__global__ void k_three(double* p) {
    if (threadIdx.x == 0) {
        double a = p[0];
        int i = (int)a & 7;
        double b = p[64 + i * 64];
        int j = (int)b & 7;
        double c = p[1024 + j * 64];
        p[0] = a + b + c + 1.0;
    }
}
int main() {
    double* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 2000;
    for (int which = 0; which < 3; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, st);
            for (int i = 0; i < N; ++i) {
                if (which == 0) k_empty<<<64, 256, 0, st>>>(d);
                else if (which == 1) k_one<<<64, 256, 0, st>>>(d);
                else k_three<<<64, 256, 0, st>>>(d);
            }
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("kernel %d: %.2f us per launch\n", which, ms * 1000.f / N);
        }
    }
    return 0;
}
