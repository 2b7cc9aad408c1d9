// Developer microbenchmark: dependent-chain latency (cycles/op) of the cross-lane primitives used by lmpc_solve_kernel.
// hipcc --offload-arch=gfx950 -O3 -I. tools/microbench.hip -o /tmp/mb && /tmp/mb
#include "../racinglmpc_amd/csrc/lmpc_kernels.hip.h"
#include <cstdio>
template <int OP> __global__ void mb(double *out, long long *cyc, int iters) {
    const int lane = threadIdx.x, lg = lane >> 3, lc = lane & 7;
    double v = 1.0 + lane * 1e-3, acc = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { v = v * 1.0000001 + 1e-9; }
        if (OP == 1) { v = dpp_blk<0x118, 0x8>(dpp_blk<0x114, 0x6>(v, v), v) + 1e-9; }
        if (OP == 2) { v = frcp(v + 1.5); }
        if (OP == 3) { v = lane_gather(v, 4 * ((lane * 7 + 3) & 63)) + 1e-9; }
        if (OP == 4) { v = rdlane(v, 54) + lane * 1e-9; }
        if (OP == 5) { v = 1.0 / (v + 1.5); }
        if (OP == 6) { v = sum_over_c(v) * 0.125; }
        if (OP == 7) { v = sum_over_g(v) * 0.125; }
        if (OP == 8) { v = wsum(v) * (1.0 / 64); }
        if (OP == 9) { v = sqrt(v + 2.0); }
        if (OP == 10) { double a, b; swap16(v, a, b); v = a + b * 1e-9; }
        if (OP == 11) { double a, b; swap32(v, a, b); v = a + b * 1e-9; }
        if (OP == 12) { v = dpp_mov<DPP_QP_X1>(v) + 1e-9; }
        if (OP == 13) { __shared__ double sh[64]; sh[lane] = v; __syncthreads(); v = sh[(lane * 5 + 1) & 63] + 1e-9; __syncthreads(); }
    }
    long long t1 = __builtin_readcyclecounter();
    acc = v;
    out[lane] = acc;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    const char *names[] = {"fma_f64", "2x dpp_blk (A form copy)", "frcp (rcp + 2 Newton)", "lane_gather(bpermute)", "rdlane(readlane)", "div_f64", "sum_over_c", "sum_over_g", "wsum", "sqrt_f64",
                           "swap16+add", "swap32+add", "dpp_mov+add", "LDS write+sync+read+sync"};
    const int iters = 2000;
#define RUN(OP) { hipLaunchKernelGGL(mb<OP>, dim3(1), dim3(64), 0, 0, d, c, iters); hipDeviceSynchronize(); hipLaunchKernelGGL(mb<OP>, dim3(1), dim3(64), 0, 0, d, c, iters); long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-28s %8.1f cycles/op\n", names[OP], (double)h / iters); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    return 0;
}
