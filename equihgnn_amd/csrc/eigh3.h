// Symmetric 3x3 eigen-decomposition as a device function (shared by csrc/eigh3.hip and csrc/faformer_geom.hip, so that a
// frame built inside a fused kernel is bitwise the one geo_eigh3 returns): cyclic Jacobi rotations in double precision on
// the fp32 matrix, eigenvalues ascending, eigenvectors in columns, each column's largest component made positive.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void jacobi_rotate(double a[3][3], double v[3][3], int p, int q) {
    if (fabs(a[p][q]) < 1e-300) return;
    const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    const double app = a[p][p], aqq = a[q][q], apq = a[p][q];
    a[p][p] = app - t * apq;
    a[q][q] = aqq + t * apq;
    a[p][q] = a[q][p] = 0.0;
    const int r = 3 - p - q;  // the third index
    const double arp = a[r][p], arq = a[r][q];
    a[r][p] = a[p][r] = c * arp - s * arq;
    a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double vip = v[i][p], viq = v[i][q];
        v[i][p] = c * vip - s * viq;
        v[i][q] = s * vip + c * viq;
    }
}

// m: the six entries of the upper triangle (00, 01, 02, 11, 12, 22); v_out[3 * i + c] = component i of eigenvector c;
// w_out (optional): the eigenvalues
__device__ __forceinline__ void eigh3_upper(const float m[6], float v_out[9], float* w_out) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    a[0][0] = m[0]; a[0][1] = a[1][0] = m[1]; a[0][2] = a[2][0] = m[2];
    a[1][1] = m[3]; a[1][2] = a[2][1] = m[4];
    a[2][2] = m[5];
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-30 * diag || off == 0.0) break;
        jacobi_rotate(a, v, 0, 1);
        jacobi_rotate(a, v, 0, 2);
        jacobi_rotate(a, v, 1, 2);
    }
    // ascending order of eigenvalues
    int idx[3] = {0, 1, 2};
    double w[3] = {a[0][0], a[1][1], a[2][2]};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2 - i; ++j)
            if (w[idx[j]] > w[idx[j + 1]]) { const int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int s = idx[c];
        double x = v[0][s], y = v[1][s], z = v[2][s];
        const double ax = fabs(x), ay = fabs(y), az = fabs(z);
        const double big = (ax >= ay && ax >= az) ? x : ((ay >= az) ? y : z);
        if (big < 0) { x = -x; y = -y; z = -z; }
        v_out[0 + c] = (float)x;
        v_out[3 + c] = (float)y;
        v_out[6 + c] = (float)z;
        if (w_out) w_out[c] = (float)w[s];
    }
}
