// Batched symmetric 3x3 eigen-decomposition — torch.linalg.eigh(C, UPLO="U") on the 3x3 covariance
// matrices of FAFormer's frame averaging (fa_former_layer.py:100: one matrix per atom for the local
// frames, one per cloud for the global frames).  One thread per matrix, cyclic Jacobi rotations in
// double precision (the input is the fp32 covariance, as in the reference; the decomposition itself
// is then accurate to fp32 rounding of the OUTPUT, which keeps eigenvectors of close eigenvalues as
// well-defined as they can be).  Eigenvalues ascending, eigenvectors in columns, each column's
// largest component made positive (the reference's LAPACK signs are arbitrary; every consumer
// averages over all 8 sign patterns).
#include "common.h"
#include "eigh3.h"

namespace {

__global__ void k_eigh3(const float* __restrict__ a_in, int64_t B, float* __restrict__ w_out,
                        float* __restrict__ v_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
        const float* m = a_in + b * 9;
        const float up[6] = {m[0], m[1], m[2], m[4], m[5], m[8]};   // upper triangle
        float v[9], w[3];
        eigh3_upper(up, v, w);
#pragma unroll
        for (int i = 0; i < 9; ++i) v_out[b * 9 + i] = v[i];
        if (w_out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w_out[b * 3 + c] = w[c];
        }
    }
}
}  // namespace

extern "C" int geo_eigh3(const float* a, int64_t B, float* w, float* v, void* stream_) {
    if (B < 0) return EQH_ERR_ARG;
    if (B == 0) return EQH_OK;
    if (!a || !v) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_eigh3, dim3(eqh_grid_for(B, 128, 4096)), dim3(128), 0, stream, a, B, w, v);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
