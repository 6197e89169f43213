// kgw_dense.hip -- the dense side of the KGWAS path: MFMA kernels (fp32-in / fp32-accumulate; bf16 x 3 exact-split products where the
// fp32 matrix pipe was the limit) for every Linear of the model, their weight gradients, loss, optimiser.  ONE translation unit,
// split by kernel family (round 6):
//   kgw_dense_tn.h          C[M,N] = A[rows,M]^T B[rows,N] over TALL inputs: the weight gradients of the feature MLPs
//                           (kgwas/model.py:13-21), of the per-relation lin_src maps (kgwas/conv.py:138,142), d u_r / d v_r
//   kgw_dense_linear.h      Linear forward / dX; the two hidden layers of SimpleMLP in one launch and their backward
//   kgw_dense_transform.h   the per-relation transform after aggregate-then-transform (conv.py:138-144,190; model.py:74-75) and
//                           its backward
//   kgw_dense_optim.h       Adam (kgwas/kgwas.py:116,151) incl. the fused optimiser launch of a captured step
//   kgw_dense_loss.h        read-out + LD-weighted MSE (model.py:86, kgwas.py:139-145)
//   kgw_dense_relvec.h      attention vectors u_r / v_r (conv.py:138-151) and the parameter-only end of the backward pass
//
// Mapping of the tall TN product (gfx950): v_mfma_f32_32x32x2_f32.  The MFMA operand layout IS the memory layout:
// lane l supplies A[row k = l>>5][column i = l&31] -- a wave-instruction reads two contiguous 128-float rows.
// Columns are interleaved (tile t owns columns MT*i + t) so one 16-byte load per lane feeds all MT tiles.
// One wavefront owns the whole (32 MT) x (32 NT) accumulator and streams its slice of rows; waves of a block are summed
// through LDS, blocks through a partial buffer and a second kernel in a fixed order: no atomics, deterministic.
#include "kgw_common.h"
#include "kgw_fold_common.h"
#include <stdlib.h>
#include <cstdlib>


#include "kgw_dense_tn.h"
#include "kgw_dense_linear.h"
#include "kgw_dense_transform.h"
#include "kgw_dense_optim.h"
#include "kgw_dense_loss.h"
#include "kgw_dense_relvec.h"
