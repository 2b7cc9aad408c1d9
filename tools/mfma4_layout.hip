// Developer probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950.
// For every pair (la, lb) a one-hot A (lane la) and one-hot B (lane lb) are multiplied; the ballot of non-zero results shows
// which (A lane, B lane) pairs meet and where their product lands.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m;
        }
}
int main() {
    unsigned long long *d, h[4096]; hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d meets B lanes -> D lanes:", la);
        for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb]) {
            printf(" %d->", lb);
            for (int l = 0; l < 64; l++) if (h[la * 64 + lb] >> l & 1) printf("%d,", l);
        }
        printf("\n");
    }
    return 0;
}
