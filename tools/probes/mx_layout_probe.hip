// Layout discovery for v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales) on gfx950, with
// exact small-integer data. Prints which operand / scale hypothesis reproduces the integer matmul.
//   build: hipcc --offload-arch=gfx950 -O2 tools/probes/mx_layout_probe.hip -o gpurun_out/mx_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__global__ void k_mfma(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb) {
    int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    c[l] = acc;
}

static uint8_t fp8_of_int(int v) {  // e4m3fn encodings of 0,1,2,3,4 and negatives
    static const uint8_t tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    uint8_t s = v < 0 ? 0x80 : 0;
    return tab[abs(v)] | s;
}

int main() {
    const int M = 16, N = 16, K = 128;
    int A[M][K], B[K][N];
    srand(1);
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) A[i][k] = rand() % 5 - 2;
    for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) B[k][j] = rand() % 7 - 3;
    int ea[M][4], eb[N][4];  // block exponents (E8M0 minus 127)
    for (int i = 0; i < M; ++i) for (int q = 0; q < 4; ++q) ea[i][q] = rand() % 5 - 2;
    for (int j = 0; j < N; ++j) for (int q = 0; q < 4; ++q) eb[j][q] = rand() % 5 - 2;
    v8i *da, *db; v4f* dc; int *dsa, *dsb;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    uint8_t ha[64][32], hb[64][32]; int hsa[64], hsb[64]; float hc[64][4];
    for (int hyp = 0; hyp < 3; ++hyp) {
        for (int scaled = 0; scaled < 2; ++scaled) {
            for (int l = 0; l < 64; ++l) {
                const int r = l & 15, g = l >> 4;
                for (int t = 0; t < 32; ++t) {
                    int k;
                    if (hyp == 0) k = 32 * g + t;                               // H0: lane group g owns k-block g, bytes in order
                    else if (hyp == 1) k = (t < 16) ? 16 * g + t : 64 + 16 * g + (t - 16);  // H1: two K=64 halves
                    else k = 4 * ((t / 4) * 4 + g) + (t % 4);                   // H2: dword-interleaved across lane groups
                    ha[l][t] = fp8_of_int(A[r][k]);
                    hb[l][t] = fp8_of_int(B[k][r] > 4 ? 4 : (B[k][r] < -4 ? -4 : B[k][r]));
                }
                // scale VGPR: byte 0 = the lane's own (row, group) exponent, other bytes garbage
                hsa[l] = (scaled ? (127 + ea[r][g]) : 127) | 0x11223300;
                hsb[l] = (scaled ? (127 + eb[r][g]) : 127) | 0x55667700;
            }
            hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
            hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
            hipLaunchKernelGGL((k_mfma<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
            hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
            // expectation under "scale of lane (r,g) applies to k-block of that lane" (only meaningful for hyp 0)
            int bad_std = 0, bad_T = 0;
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < N; ++j) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) {
                        int q = k / 32;
                        double s = scaled ? ldexp(1.0, ea[i][q] + eb[j][q]) : 1.0;
                        int bv = B[k][j] > 4 ? 4 : (B[k][j] < -4 ? -4 : B[k][j]);
                        ref += s * A[i][k] * bv;
                    }
                    // C layout candidates: standard (col = lane&15, row = (lane>>4)*4 + reg) and its transpose
                    const float got_std = hc[(i / 4) * 16 + j][i % 4];
                    const float got_T = hc[(j / 4) * 16 + i][j % 4];
                    if (fabs(got_std - ref) > 1e-3) ++bad_std;
                    if (fabs(got_T - ref) > 1e-3) ++bad_T;
                }
            printf("hyp %d scaled %d: mismatches std-layout %d, transposed-layout %d (of 256)\n", hyp, scaled, bad_std, bad_T);
        }
    }
    // op_sel: which byte of the scale VGPR is used (hypothesis: opsel k -> byte k), layout H1 + "lane (r,g) scales block g"
    for (int op = 0; op < 4; ++op) {
        for (int l = 0; l < 64; ++l) {
            const int r = l & 15, g = l >> 4;
            for (int t = 0; t < 32; ++t) {
                const int k = (t < 16) ? 16 * g + t : 64 + 16 * g + (t - 16);
                ha[l][t] = fp8_of_int(A[r][k]);
                hb[l][t] = fp8_of_int(B[k][r] > 4 ? 4 : (B[k][r] < -4 ? -4 : B[k][r]));
            }
            hsa[l] = 0x7f7f7f7f; hsb[l] = 0x7f7f7f7f;
            ((uint8_t*)&hsa[l])[op] = (uint8_t)(127 + ea[r][g]);
            ((uint8_t*)&hsb[l])[op] = (uint8_t)(127 + eb[r][g]);
        }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
        if (op == 0) hipLaunchKernelGGL((k_mfma<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        if (op == 1) hipLaunchKernelGGL((k_mfma<1, 1>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        if (op == 2) hipLaunchKernelGGL((k_mfma<2, 2>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        if (op == 3) hipLaunchKernelGGL((k_mfma<3, 3>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
        hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k) {
                    int q = k / 32;
                    int bv = B[k][j] > 4 ? 4 : (B[k][j] < -4 ? -4 : B[k][j]);
                    ref += ldexp(1.0, ea[i][q] + eb[j][q]) * A[i][k] * bv;
                }
                if (fabs(hc[(i / 4) * 16 + j][i % 4] - ref) > 1e-3) ++bad;
            }
        printf("op_sel %d -> byte %d: mismatches %d (of 256)\n", op, op, bad);
    }
    // fp8 conversion check: v_cvt_pk_fp8_f32 on a value table
    return 0;
}
