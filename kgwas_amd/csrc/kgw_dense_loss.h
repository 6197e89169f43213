// kgw_dense_loss.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): LD-weighted MSE (kgw_wmse) and the fused read-out + loss node (kgw_readout_wmse*).
#pragma once

// ======================================================================================================
// kgw_wmse: LD-score weighted MSE of the seed predictions, loss = mean(w[n_id] * (pred - y[n_id])^2) in float64
// (kgwas/kgwas.py:139-145: float32 residual and square, float64 weight, float64 mean), and its gradient.
// One block; fixed-order reduction.
// ======================================================================================================
namespace {

__global__ void __launch_bounds__(256) k_wmse_fwd(const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                  const float* __restrict__ y, const double* __restrict__ w, int n,
                                                  double* __restrict__ loss) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int g = n_id[i];
        const float d = pred[i] - y[g];
        acc += w[g] * (double)(d * d);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sm[0] / (double)n;
}

__global__ void __launch_bounds__(256) k_wmse_bwd(const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                  const float* __restrict__ y, const double* __restrict__ w, int n,
                                                  const double* __restrict__ gloss, float* __restrict__ dpred) {
    const double g0 = gloss[0] / (double)n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int g = n_id[i];
        const float d = pred[i] - y[g];
        dpred[i] = (float)(g0 * w[g]) * (2.0f * d);        // the float64 product meets the float32 square here
    }
}

}  // namespace

extern "C" int kgw_wmse_fwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                            double* loss, kgw_stream_t stream_) {
    if (!pred || !n_id || !y || !w || !loss) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    k_wmse_fwd<<<1, 256, 0, (hipStream_t)stream_>>>(pred, n_id, y, w, n, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_wmse_bwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                            const double* grad_loss, float* dpred, kgw_stream_t stream_) {
    if (!pred || !n_id || !y || !w || !grad_loss || !dpred) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    k_wmse_bwd<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(pred, n_id, y, w, n, grad_loss, dpred);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ======================================================================================================
// kgw_readout_wmse: read-out Linear(128 -> 1) (+ ReLU) of the seed rows (kgwas/model.py:86) fused with the
// LD-score weighted MSE (kgwas/kgwas.py:139-145).  One block: wavefront w takes seeds w, w+4, ...; partial sums are
// combined in a fixed order.  _bwd also produces the gradients of the read-out weight / bias and dH (zero for the
// rows beyond the seeds).
// ======================================================================================================
namespace {

// One wavefront per seed, four per block; per-seed / per-block partial results go to a scratch buffer and a second,
// single-block launch folds them in index order -- parallel across the chip, yet a fixed summation order.  (A
// "last block folds" hand-off inside one launch was tried: its device-scope fence cost more than the second launch.)
__global__ void __launch_bounds__(256) k_readout_wmse_fwd(const float* __restrict__ H, const float* __restrict__ wl,
                                                          const float* __restrict__ bl, const int32_t* __restrict__ n_id,
                                                          const float* __restrict__ y, const double* __restrict__ w, int n,
                                                          int relu, float* __restrict__ pred, double* __restrict__ terms) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const float2 w2 = ((const float2*)wl)[lane];
    const float2 h2 = ((const float2*)(H + (int64_t)i * KGW_C))[lane];
    float p = kgw_wave_allsum(fmaf(h2.x, w2.x, h2.y * w2.y)) + bl[0];
    if (relu) p = fmaxf(p, 0.f);
    if (lane == 0) {
        const int g = n_id[i];
        const float d = p - y[g];
        pred[i] = p;
        terms[i] = w[g] * (double)(d * d);
    }
}

__global__ void __launch_bounds__(256) k_fold_f64(const double* __restrict__ terms, int n, double* __restrict__ out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int q = threadIdx.x; q < n; q += 256) acc += terms[q];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] / (double)n;
}

__global__ void __launch_bounds__(256) k_readout_wmse_bwd(const float* __restrict__ H, const float* __restrict__ wl,
                                                          const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                          const float* __restrict__ y, const double* __restrict__ w, int n,
                                                          int64_t rows, int relu, const double* __restrict__ gloss,
                                                          float* __restrict__ dH, float* __restrict__ part) {
    __shared__ float sw[4][KGW_C + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    float2 dw = make_float2(0.f, 0.f);
    float dp = 0.f;
    if (i < n) {
        const float2 w2 = ((const float2*)wl)[lane];
        const int g = n_id[i];
        const float p = pred[i];
        dp = (float)(gloss[0] / (double)n * w[g]) * (2.0f * (p - y[g]));
        if ((relu & 1) && !(p > 0.f)) dp = 0.f;
        const float2 h2 = ((const float2*)(H + i * KGW_C))[lane];
        // (bit 1 of `relu`: H itself is the output of a ReLU whose backward the caller folds in here: dH *= (H > 0))
        const bool mk = (relu & 2) != 0;
        ((float2*)(dH + i * KGW_C))[lane] = make_float2((!mk || h2.x > 0.f) ? dp * w2.x : 0.f,
                                                         (!mk || h2.y > 0.f) ? dp * w2.y : 0.f);
        dw = make_float2(dp * h2.x, dp * h2.y);
    } else if (i < rows) {
        ((float2*)(dH + i * KGW_C))[lane] = make_float2(0.f, 0.f);
    }
    if ((int64_t)blockIdx.x * 4 >= n) return;            // blocks without seeds hold no partial
    sw[wave][2 * lane] = dw.x; sw[wave][2 * lane + 1] = dw.y;
    if (lane == 0) sw[wave][KGW_C] = dp;
    __syncthreads();
    if (threadIdx.x <= KGW_C) {                    // block partial: 128 weight columns + the bias term
        const int c = threadIdx.x;
        part[(int64_t)blockIdx.x * (KGW_C + 1) + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    }
}

// d w_lin [128] and d b_lin from the per-block partials [nb][129]: 129 columns x 7 row groups of one block, fixed order
__global__ void __launch_bounds__(1024) k_readout_fold(const float* __restrict__ part, int nb, float* __restrict__ dwl,
                                                       float* __restrict__ dbl) {
    __shared__ float sm[7][KGW_C + 1];
    const int c = threadIdx.x % (KGW_C + 1), g = threadIdx.x / (KGW_C + 1);
    if (g < 7) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = g;
        for (; q + 21 < nb; q += 28) {
            a0 += part[(int64_t)q * (KGW_C + 1) + c];        a1 += part[(int64_t)(q + 7) * (KGW_C + 1) + c];
            a2 += part[(int64_t)(q + 14) * (KGW_C + 1) + c]; a3 += part[(int64_t)(q + 21) * (KGW_C + 1) + c];
        }
        for (; q < nb; q += 7) a0 += part[(int64_t)q * (KGW_C + 1) + c];
        sm[g][c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (g == 0) {
        const float t = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + sm[6][c]);
        if (c < KGW_C) dwl[c] = t; else dbl[0] = t;
    }
}

// Training step with a unit loss gradient (loss.backward()): forward and backward of the read-out in ONE launch per
// stage -- the per-seed stage computes prediction, loss term, d prediction, the dH row and the block's weight-gradient
// partial; the fold stage adds up the loss terms (float64, index order) and the partials.  Two launches instead of four.
__global__ void __launch_bounds__(256) k_readout_wmse_train(const float* __restrict__ H, const float* __restrict__ wl,
                                                            const float* __restrict__ bl, const int32_t* __restrict__ n_id,
                                                            const float* __restrict__ y, const double* __restrict__ w, int n,
                                                            int64_t rows, int relu, float* __restrict__ pred,
                                                            double* __restrict__ terms, float* __restrict__ dH,
                                                            float* __restrict__ part) {
    __shared__ float sw[4][KGW_C + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    float2 dw = make_float2(0.f, 0.f);
    float dp = 0.f;
    if (i < n) {
        const float2 w2 = ((const float2*)wl)[lane];
        const float2 h2 = ((const float2*)(H + i * KGW_C))[lane];
        float p = kgw_wave_allsum(fmaf(h2.x, w2.x, h2.y * w2.y)) + bl[0];
        if (relu & 1) p = fmaxf(p, 0.f);
        const int g = n_id[i];
        const float d = p - y[g];
        if (lane == 0) {
            pred[i] = p;
            terms[i] = w[g] * (double)(d * d);
        }
        dp = (float)(1.0 / (double)n * w[g]) * (2.0f * d);
        if ((relu & 1) && !(p > 0.f)) dp = 0.f;
        const bool mk = (relu & 2) != 0;
        ((float2*)(dH + i * KGW_C))[lane] = make_float2((!mk || h2.x > 0.f) ? dp * w2.x : 0.f,
                                                         (!mk || h2.y > 0.f) ? dp * w2.y : 0.f);
        dw = make_float2(dp * h2.x, dp * h2.y);
    } else if (i < rows) {
        ((float2*)(dH + i * KGW_C))[lane] = make_float2(0.f, 0.f);
    }
    if ((int64_t)blockIdx.x * 4 >= n) return;            // blocks without seeds hold no partial
    sw[wave][2 * lane] = dw.x; sw[wave][2 * lane + 1] = dw.y;
    if (lane == 0) sw[wave][KGW_C] = dp;
    __syncthreads();
    if (threadIdx.x <= KGW_C) {
        const int c = threadIdx.x;
        part[(int64_t)blockIdx.x * (KGW_C + 1) + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    }
}

__global__ void __launch_bounds__(1024) k_readout_train_fold(const float* __restrict__ part, int nb, const double* __restrict__ terms,
                                                             int n, float* __restrict__ dwl, float* __restrict__ dbl,
                                                             double* __restrict__ loss) {
    __shared__ float sm[7][KGW_C + 1];
    __shared__ double sd[256];
    const int c = threadIdx.x % (KGW_C + 1), g = threadIdx.x / (KGW_C + 1);
    if (g < 7) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = g;
        for (; q + 21 < nb; q += 28) {
            a0 += part[(int64_t)q * (KGW_C + 1) + c];        a1 += part[(int64_t)(q + 7) * (KGW_C + 1) + c];
            a2 += part[(int64_t)(q + 14) * (KGW_C + 1) + c]; a3 += part[(int64_t)(q + 21) * (KGW_C + 1) + c];
        }
        for (; q < nb; q += 7) a0 += part[(int64_t)q * (KGW_C + 1) + c];
        sm[g][c] = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x < 256) {                                  // the loss: same order as k_fold_f64
        double acc = 0.0;
        for (int q = threadIdx.x; q < n; q += 256) acc += terms[q];
        sd[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sd[0] / (double)n;
    if (g == 0) {
        const float t = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + sm[6][c]);
        if (c < KGW_C) dwl[c] = t; else dbl[0] = t;
    }
}

}  // namespace

extern "C" int kgw_readout_wmse_train_parts(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                            const float* y, const double* w, int32_t n, int64_t rows, int32_t relu, float* pred,
                                            double* loss, float* dH, float* dw_lin, float* db_lin, double* terms, float* scratch,
                                            KgwReadoutFold* fold_out, kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !dH || !dw_lin || !db_lin || !terms || !scratch || !fold_out)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    k_readout_wmse_train<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(H, w_lin, b_lin, n_id, y, w, n, rows, relu, pred,
                                                                                        terms, dH, scratch);
    KGW_LAUNCH_CHECK();
    *fold_out = KgwReadoutFold{scratch, terms, dw_lin, db_lin, loss, (n + 3) / 4, n};
    return KGW_OK;
}

extern "C" int kgw_readout_train_fold(const KgwReadoutFold* f, kgw_stream_t stream_) {
    if (!f || !f->scratch || !f->terms || !f->dw_lin || !f->db_lin || !f->loss) return KGW_E_NULL;
    if (f->n <= 0 || f->nb <= 0) return KGW_E_RANGE;
    k_readout_train_fold<<<1, 1024, 0, (hipStream_t)stream_>>>(f->scratch, f->nb, f->terms, f->n, f->dw_lin, f->db_lin, f->loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_fwd(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                    const float* y, const double* w, int32_t n, int32_t relu, float* pred,
                                    double* loss, double* scratch, kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !scratch) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    k_readout_wmse_fwd<<<(n + 3) / 4, 256, 0, st>>>(H, w_lin, b_lin, n_id, y, w, n, relu, pred, scratch);
    KGW_LAUNCH_CHECK();
    k_fold_f64<<<1, 256, 0, st>>>(scratch, n, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_bwd(const float* H, const float* w_lin, const float* pred, const int32_t* n_id,
                                    const float* y, const double* w, int32_t n, int64_t rows, int32_t relu,
                                    const double* grad_loss, float* dH, float* dw_lin, float* db_lin, float* scratch,
                                    kgw_stream_t stream_) {
    if (!H || !w_lin || !pred || !n_id || !y || !w || !grad_loss || !dH || !dw_lin || !db_lin || !scratch)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    k_readout_wmse_bwd<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(H, w_lin, pred, n_id, y, w, n, rows, relu, grad_loss, dH,
                                                                    scratch);
    KGW_LAUNCH_CHECK();
    k_readout_fold<<<1, 1024, 0, st>>>(scratch, (n + 3) / 4, dw_lin, db_lin);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_train(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                      const float* y, const double* w, int32_t n, int64_t rows, int32_t relu, float* pred,
                                      double* loss, float* dH, float* dw_lin, float* db_lin, double* terms, float* scratch,
                                      kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !dH || !dw_lin || !db_lin || !terms || !scratch)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    // (round 4, measured and dropped: the whole node as ONE block of 16 wavefronts walking the 512 rows -- no partial buffer, no
    //  fold launch -- ran the step 40 - 45 us SLOWER: 32 dependent row trips per wavefront instead of one)
    k_readout_wmse_train<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(H, w_lin, b_lin, n_id, y, w, n, rows, relu, pred, terms, dH,
                                                                      scratch);
    KGW_LAUNCH_CHECK();
    k_readout_train_fold<<<1, 1024, 0, st>>>(scratch, (n + 3) / 4, terms, n, dw_lin, db_lin, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
