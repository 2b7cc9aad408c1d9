// Developer check: gram8_mfma (lmpc_kernels.hip.h) against a host sum, for CH = 1, 2, 6 terminal-block columns per lane and a wide dynamic range of M.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/gram_check.hip -o build_tmp/gram_check
#include "../racinglmpc_amd/csrc/lmpc_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
template <int CH> __global__ __launch_bounds__(64, 1) void gk(const double *M, double *W) {
    __shared__ double Mt[8 * 64 * CH], Wl[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8 * 64 * CH; i += 64) Mt[i] = M[i];
    __syncthreads();
    gram8_mfma<CH>(Mt, Wl, lane);
    __syncthreads();
    W[lane] = Wl[lane];
}
template <int CH> static double run() {
    const int n = 8 * 64 * CH;
    std::vector<double> M(n);
    unsigned s = 12345u + CH;
    for (int c = 0; c < 64 * CH; c++) {
        s = s * 1664525u + 1013904223u; const double sc = pow(10.0, -6.0 + 12.0 * ((s >> 8) % 1000) / 999.0);      // column scales 1e-6 .. 1e6
        for (int r = 0; r < 8; r++) { s = s * 1664525u + 1013904223u; M[c * 8 + r] = r == 7 ? 0.0 : sc * (((s >> 8) % 2001) / 1000.0 - 1.0); }
    }
    double *dM, *dW; hipMalloc(&dM, n * 8); hipMalloc(&dW, 64 * 8);
    hipMemcpy(dM, M.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gk<CH>, dim3(1), dim3(64), 0, 0, dM, dW); hipDeviceSynchronize();
    double W[64]; hipMemcpy(W, dW, sizeof(W), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
        long double acc = 0, mag = 0;
        for (int c = 0; c < 64 * CH; c++) { acc += (long double)M[c * 8 + i] * M[c * 8 + j]; mag += fabsl((long double)M[c * 8 + i] * M[c * 8 + j]); }
        const double err = (double)(fabsl(acc - W[i * 8 + j]) / (mag + 1e-300L));
        if (err > worst) worst = err;
    }
    hipFree(dM); hipFree(dW);
    return worst;
}
int main() {
    printf("gram8_mfma worst |W - M M'| / sum|products|: CH=1 %.2e  CH=2 %.2e  CH=6 %.2e\n", run<1>(), run<2>(), run<6>());
    return 0;
}
