// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment:
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_f64_layout tools/exp/mfma_f64_layout.cpp && tools/bin/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}
int main() {
    double ha[64], hb[64], hd[256], *da, *db, *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    // hypothesis: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16].  A[i][k] = (i + 1) * 1000^k-ish codes
    for (int l = 0; l < 64; ++l) { ha[l] = (l % 16 + 1) * (l / 16 == 0 ? 1.0 : l / 16 == 1 ? 100.0 : l / 16 == 2 ? 1e4 : 1e6); hb[l] = 0.0; }
    // B = one-hot column selector per k: B[k][j] = 1 if j == k + 3 -> D[i][k+3] = A[i][k]
    for (int l = 0; l < 64; ++l) hb[l] = (l % 16 == l / 16 + 3) ? 1.0 : 0.0;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v)
            if (hd[l * 4 + v] != 0.0) printf("lane %2d v %d = %g\n", l, v, hd[l * 4 + v]);
    return 0;
}
