// 128x128 diagonal-block Cholesky + triangular inverse, one workgroup, all in LDS.
//
// Stands in for the unblocked dpotf2 / dtrtri steps inside LAPACK dpotrf that
// scipy.linalg.cho_factor reaches (fvgp/gp_lin_alg.py:245).  The blocked driver
// (api.hip: panel_factor) calls it once per 128 columns; its outputs are
//   * L11 in place (lower triangle of the block only -- the strict upper is never read or
//     written, matching cho_factor's "upper is unspecified"),
//   * inv(L11) as a dense 128x128 (upper = 0) so that the panel TRSM  A21 * L11^-T  and the
//     block substitutions of potrs become plain MFMA GEMMs,
//   * sum(log L_ii) of the block's valid rows (feeds calculate_Chol_logdet, gp_lin_alg.py:337-338),
//   * info: 1-based global index of the first non-positive pivot (dpotrf's info > 0).
//
// Resource shape matters as much as speed: under look-ahead this kernel must start while the
// trailing SYRK occupies every CU with two 72 KB workgroups, so it is built to fit into what ONE
// retiring SYRK workgroup frees: the lower triangle is held as 36 packed 16x16 tiles
// (36 x 2048 B = 72 KB + 1 KB, below the 76 KB a SYRK workgroup holds), 512 threads at <= 128 VGPRs.
//
// Algorithm (16-wide blocks):
//   for p = 0..7:  TRSM of the rows below by per-row forward substitution against L_pp;
//                  trailing tiles C_ij -= X_i X_j^T on fp64 MFMA 16x16x4; wave 0 takes tile
//                  (p+1,p+1) first and factors it while the other waves finish the update: four-column
//                  panels whose 4x4 diagonal block is factored as wave-uniform scalars (Newton reciprocal
//                  square roots), the rows under it solved per lane group, and ONE rank-4 MFMA update of the
//                  rest of the tile per panel (PanelBlock).
//   inverse:       wave j owns block column j of inv(L):  X_jj = inv(L_jj),
//                  X_ij = -inv(L_ii) * sum_{k=j..i-1} L_ik X_kj.  The f64 MFMA accumulator layout
//                  (row = q+4v) is exactly the next MFMA's B-operand layout, so the whole column
//                  stays in registers and goes straight to global memory.
#include "leaf_body.h"

namespace {

__global__ __launch_bounds__(512, 4) void leaf_kernel(LeafArgs g) {
    __shared__ double sT[NT * TSZ];      // 73,728 B
    __shared__ double srd[LEAF_SRD];     // 1 / L_aa, the W operands of the diagonal tile at hand, the solvers' counter (leaf_body.h)
    leaf_body<false>(g, g.A + (long)blockIdx.x * g.a_stride, g.linv + (long)blockIdx.x * g.linv_stride,
                     g.logdet_part ? g.logdet_part + (long)blockIdx.x * 128 : nullptr, g.info_base + (int)blockIdx.x * 128, sT, srd, (int)threadIdx.x);
}

}  // namespace


int launch_leaf(fvgp_handle *h, double *A, int64_t lda, double *linv, double *logdet_part, int info_base, int do_factor, int nvalid,
                int tiles_only) {
    LeafArgs g;
    g.tiles_only = tiles_only; g.preloaded = 0; g.col_flag = nullptr; g.col_base = 0;
    g.A = A; g.lda = lda; g.linv = linv; g.logdet_part = logdet_part; g.info = h->dinfo; g.info_base = info_base;
    g.do_factor = do_factor; g.a_stride = 0; g.linv_stride = 0; g.nvalid = nvalid; g.stamps = h->leaf_stamps;
    g.yield = (h->leaf_yield && do_factor) ? h->cu_yield : nullptr;
    hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(512), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_leaf_inverse_batched(fvgp_handle *h, const double *L, int64_t ldl, int64_t nblk, double *linv) {
    if (nblk <= 0) return 0;
    LeafArgs g;
    g.A = const_cast<double *>(L); g.lda = ldl; g.linv = linv; g.logdet_part = nullptr; g.info = h->dinfo; g.info_base = 0;
    g.do_factor = 0; g.a_stride = 128 * ldl + 128; g.linv_stride = LEAF_DOUBLES; g.nvalid = 128; g.stamps = nullptr; g.tiles_only = 0; g.yield = nullptr; g.preloaded = 0; g.col_flag = nullptr; g.col_base = 0;
    hipLaunchKernelGGL(leaf_kernel, dim3((unsigned)nblk), dim3(512), 0, h->stream, g);
    HIPCHK(hipGetLastError());
    return 0;
}

#ifdef FVGP_LEAF_FINE
// diagnostic build: every wave's event times in the factor loop (leaf_body.h, FVGP_WFINE)
extern "C" int fvgp_hip_debug_fine(unsigned long *out_host, int n) {
    if (n > 512) n = 512;
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wfine), (size_t)n * sizeof(unsigned long)) != hipSuccess) return -1;
    return n;
}
#endif
