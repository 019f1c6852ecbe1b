// mmq_args.h — argument block shared by the integer matrix-core mat-mul kernels (mmq_i8.hip: prompt batches; mmq_skinny.hip:
// continuous-batching decode steps of <= 32 columns).
#pragma once
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// Up to three weight matrices of one type that multiply the SAME activations (wq / wk / wv, ffn_gate / ffn_up) share a launch:
// their row panels form one list (matrix i owns panels [panel0_i, panel0_{i+1})), which fills the chip where each of them
// alone needed a K split — and a second kernel — to do so.
struct mmq8_mat {
    const uint8_t * W;
    int64_t w_nb1;
    int N, panel0;
    float * dst;
    int64_t dst_stride;
    float * part;        // ksplit > 1: partial [ksplit][M][N] results of this matrix (summed in a fixed order by k_splitk_reduce)
    const float * add;   // optional epilogue: + add[m * add_stride + n] (bias row: stride 0, residual: stride N)
    int64_t add_stride;
    int qt;              // two-format launches of the skinny kernel: 4 / 5 / 6 = how THIS matrix is stored (0 otherwise)
};
struct mmq8_args {
    mmq8_mat mat[3];
    int n_mat;
    int K, M;
    const q8k_dev * act;  // [M][K/256]
    int n_panels, m_tiles;
    int ksplit;          // > 1: blockIdx.y owns a contiguous range of super-blocks and writes its partial result
    int has_epi;         // skinny kernel: `epi` describes what happens to the results instead of the plain store
    mmq_epi epi;
};

void launch_mmq_skinny(hipStream_t s, int type, const mmq8_args & a);
// matrices of two K-quant formats in one launch (a.mat[].qt set); false: this pair of formats is not built
bool launch_mmq_skinny_mixed(hipStream_t s, const mmq8_args & a);
// prompt batches on the same unit (mmq_skinny.hip, wide form): token tiles per workgroup (4 / 2) when it applies, 0 when it does not
int mmq_wide_tiles(int type, int64_t K, const int64_t * N, int n_mat, int64_t M, int64_t w_nb1, int ksplit);
void launch_mmq_wide(hipStream_t s, int type, int tt, const mmq8_args & a);  // mmq_skinny.hip; a.mat[].panel0 / n_panels are set by the launcher

}  // namespace mi355x
