// Developer microbenchmark: dependent-chain latency of the FP64 matrix-core instructions on gfx950 (single wave).
// hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma.hip -o /tmp/mbm && /tmp/mbm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int OP> __global__ void mb(double *out, long long *cyc, int iters) {
    const int lane = threadIdx.x;
    double v = 1.0 + lane * 1e-3, w = 0.5 - lane * 1e-4;
    v4d c = {0.0, 0.0, 0.0, 0.0}, c2 = {0.0, 0.0, 0.0, 0.0};
    double s = 0.0;
    float fv = 1.0f + lane * 1e-3f; v4f fc = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { v = v * 1.0000001 + 1e-9; }
        if (OP == 1) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, w, c, 0, 0, 0); }
        if (OP == 2) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, w, c, 0, 0, 0); w = c[0] * 1e-9; }
        if (OP == 3) { s = __builtin_amdgcn_mfma_f64_4x4x4f64(v, w, s, 0, 0, 0); }
        if (OP == 4) { s = __builtin_amdgcn_mfma_f64_4x4x4f64(v, w, s, 0, 0, 0); w = s * 1e-9; }
        if (OP == 5) { fc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv, fv, fc, 0, 0, 0); fv = fc[0] * 1e-9f; }
        if (OP == 6) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, w, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, v, c2, 0, 0, 0); }
        if (OP == 7) { v4d z = {0.0, 0.0, 0.0, 0.0}; c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, w, z, 0, 0, 0); w = c[0] * 1e-9; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = v + w + c[0] + c[1] + c2[0] + s + fv + fc[0];
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    const char *names[] = {"fma_f64", "mfma_f64_16x16x4 acc-dep", "mfma_f64_16x16x4 operand-dep(+mul)", "mfma_f64_4x4x4 acc-dep", "mfma_f64_4x4x4 operand-dep(+mul)",
                           "mfma_f32_16x16x4 operand-dep", "2 indep mfma_f64_16x16x4", "mfma_f64_16x16x4 zero-acc operand-dep"};
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(OP) { hipLaunchKernelGGL(mb<OP>, dim3(1), dim3(64), 0, 0, d, c, iters); hipDeviceSynchronize(); hipEventRecord(e0); hipLaunchKernelGGL(mb<OP>, dim3(1), dim3(64), 0, 0, d, c, iters * 50); hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-40s %8.1f counter ticks/op  %8.1f ns/op\n", names[OP], (double)h / (iters * 50), ms * 1e6 / (iters * 50)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
