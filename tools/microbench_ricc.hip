// Developer microbenchmark: cycles per stage of the Riccati recursion (ricc_factor, lmpc_kernels.hip.h) run by ONE wave alone on a CU, with
// the parts of a stage switched off one at a time (-DRICC_VAR_NOSYM: no symmetrisation transpose; -DRICC_VAR_NOSTORE: factors not written to
// LDS).  Measured on MI355X: 934 cycles per stage (391 ns at 2.39 GHz), 854 without the transpose, 844 without the stores.  Build the variants in the container, run them on the GPU box:
//   for v in BASE NOSYM NOSTORE; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRICC_VAR_$v tools/microbench_ricc.hip -o build_tmp/mb_ricc_$v; done
#include "../racinglmpc_amd/csrc/lmpc_kernels.hip.h"
#include <cstdio>
template <int N> __global__ __launch_bounds__(64, 1) void mb(double *out, long long *cyc, int reps) {
    __shared__ double AB[48 * N], kap[2 * N], th[8 * N + 48], Qf2[36], PiT[36], Phi[64 * N], PiAll[64 * N], Mi[4 * N], Q2[36], Fx[12], R2[4], dR2[2], Fu[8], dump[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 48 * N; i += 64) { const int r = (i % 48) / 8, c = i % 8; AB[i] = (r == c ? 1.0 : 0.0) + 0.01 * ((i * 7) % 13 - 6) * (c < 6 ? 0.1 : 1.0); }
    for (int i = lane; i < 2 * N; i += 64) kap[i] = 0.5 + 0.01 * i;
    for (int i = lane; i < 8 * N + 48; i += 64) th[i] = 0.3 + 0.001 * i;
    if (lane < 36) { Qf2[lane] = 0.0; PiT[lane] = (lane % 7 == 0) ? 50.0 : 0.1; Q2[lane] = (lane % 7 == 0) ? 2.0 : 0.0; }
    if (lane < 12) Fx[lane] = (lane == 5) ? 1.0 : (lane == 11 ? -1.0 : 0.0);
    if (lane < 4) R2[lane] = (lane % 3 == 0) ? 2.0 : 0.0;
    if (lane < 2) dR2[lane] = 10.0 + 90.0 * lane;
    if (lane < 8) { const double fu[8] = {1, 0, -1, 0, 0, 1, 0, -1}; Fu[lane] = fu[lane]; }
    __syncthreads();
    const ricc_consts rc = ricc_setup(lane, Q2, Fx, R2, dR2, Fu);
    int bad = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) bad |= ricc_factor<N, true>(rc, AB, kap, th, Qf2, PiT, Phi, PiAll, Mi, (double *)nullptr, dump + lane);
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[lane] = Phi[lane] + Mi[lane & 3] + bad;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    const int reps = 400;
    hipLaunchKernelGGL(mb<12>, dim3(1), dim3(64), 0, 0, d, c, 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(mb<12>, dim3(1), dim3(64), 0, 0, d, c, reps); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); double o[64]; hipMemcpy(o, d, 64 * 8, hipMemcpyDeviceToHost);
    printf("ricc_factor<12>: %.0f counter ticks / stage, %.1f ns / stage (%.1f us per factorisation), check %g\n", (double)h / (reps * 12), ms * 1e6 / (reps * 12), ms * 1e3 / reps, o[0]);
    return 0;
}
