// Weighted normal equations of qphb.solve_convex_opt (hybdrt/models/qphb.py:465-466) and the L2 assembly of
// qphb.calculate_qp_l2_matrix (qphb.py:53-120), batched over spectra that share one response matrix A:
//
//     P_b = (W_b A)' (W_b A) + L2_b ,   L2_b = sum_k S_bk^1/2 (M_k o scale_bk) S_bk^1/2
//     q_b = -(W_b A)' (W_b y_b) + l1
//
// FP64 only.  The contraction runs on v_mfma_f64_16x16x4_f64: a 256-thread workgroup
// (4 wavefronts as 2x2) owns a 64x64 tile of the lower triangle of P_b, stages 16-row slabs of A (already
// multiplied by w_b) through LDS and mirrors the tile into the upper triangle on store.  The reference forms
// L2 with two dense n^3 products per derivative order; here it is the O(n^2) elementwise epilogue of the tile.
#include "common.hpp"

namespace hipdrt {

typedef double v4d __attribute__((ext_vector_type(4)));

static constexpr int GT = 64;        // tile edge
static constexpr int GK = 16;        // K slab
static constexpr int GLD = 80;       // LDS row stride in doubles (k-rows land 32 banks apart: conflict-free b64 reads)


// ROWP: also write the row-major copy P (stand-alone entry points); the fit loop reads the packed tiles only
template <bool DOP, bool ROWP>
__global__ __launch_bounds__(256, 5) void gram_kernel(int m, int n, const double* __restrict__ A, int lda,
                                                   const double* __restrict__ w, GramL2 g, double* __restrict__ P,
                                                   int ldp, long long p_stride, const int* __restrict__ active,
                                                   int ntile, double* __restrict__ Ppk, long long ppk_stride, int nchp,
                                                   long long a_stride) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    A += (size_t)b * a_stride;
    // decode lower-triangular tile index -> (ti >= tj)
    int t = blockIdx.x, ti = 0;
    while (t >= ti + 1) { t -= ti + 1; ++ti; }
    const int tj = t;
    const int i0 = ti * GT, j0 = tj * GT;
    const bool diag = (ti == tj);

    __shared__ double sI[GK * GLD];
    __shared__ double sJ[GK * GLD];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wi = (wv >> 1) * 32, wj = (wv & 1) * 32;     // wave's 32x32 sub-tile
    const double* wb = w + (size_t)b * m;

    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (v4d){0.0, 0.0, 0.0, 0.0};

    // staging map: thread -> (k = tid/16, 4 consecutive columns), fetched as two 16-byte loads when the
    // leading dimension and n are even (always true for the plan's matrices)
    const int sk = tid >> 4, sc = (tid & 15) * 4;
    const bool vec2 = ((lda | n) & 1) == 0;
    // slab k0+GK is fetched (global -> registers) while slab k0 is multiplied out of LDS; the products with w are
    // formed only when the slab is written to LDS, so nothing waits on the loads inside the MFMA loop
    double vi[4], vj[4], wkr = 0.0;
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vi[e] = 0.0; vj[e] = 0.0; }
        wkr = 0.0;
        if (k < m) {
            wkr = wb[k];
            const double* row = A + (size_t)k * lda;
            if (vec2) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int ci = i0 + sc + e, cj = j0 + sc + e;
                    if (ci < n) { const double2 t = *reinterpret_cast<const double2*>(row + ci); vi[e] = t.x; vi[e + 1] = t.y; }
                    if (!diag && cj < n) { const double2 t = *reinterpret_cast<const double2*>(row + cj); vj[e] = t.x; vj[e + 1] = t.y; }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = i0 + sc + e, cj = j0 + sc + e;
                    if (ci < n) vi[e] = row[ci];
                    if (!diag && cj < n) vj[e] = row[cj];
                }
            }
        }
    };
    // 16x16 sub-tiles that are pure padding (beyond n) or lie above the diagonal (the epilogue takes those elements
    // from the mirror) are not multiplied at all
    const int nt16 = (n + 15) >> 4;
    const int tr0 = __builtin_amdgcn_readfirstlane((i0 + wi) >> 4), tc0 = __builtin_amdgcn_readfirstlane((j0 + wj) >> 4);
    const bool need00 = tr0 < nt16 && tc0 < nt16 && tr0 >= tc0;
    const bool need01 = tr0 < nt16 && tc0 + 1 < nt16 && tr0 >= tc0 + 1;
    const bool need10 = tr0 + 1 < nt16 && tc0 < nt16 && tr0 + 1 >= tc0;
    const bool need11 = tr0 + 1 < nt16 && tc0 + 1 < nt16 && tr0 + 1 >= tc0 + 1;
    fetch(0);
    for (int k0 = 0; k0 < m; k0 += GK) {
        __syncthreads();   // previous slab fully consumed
        *reinterpret_cast<double2*>(&sI[sk * GLD + sc]) = make_double2(wkr * vi[0], wkr * vi[1]);
        *reinterpret_cast<double2*>(&sI[sk * GLD + sc + 2]) = make_double2(wkr * vi[2], wkr * vi[3]);
        if (!diag) {
            *reinterpret_cast<double2*>(&sJ[sk * GLD + sc]) = make_double2(wkr * vj[0], wkr * vj[1]);
            *reinterpret_cast<double2*>(&sJ[sk * GLD + sc + 2]) = make_double2(wkr * vj[2], wkr * vj[3]);
        }
        __syncthreads();
        if (k0 + GK < m) fetch(k0 + GK);
        const double* sj = diag ? sI : sJ;
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a0 = sI[kr * GLD + wi + (lane & 15)];
            double a1 = sI[kr * GLD + wi + 16 + (lane & 15)];
            double b0 = sj[kr * GLD + wj + (lane & 15)];
            double b1 = sj[kr * GLD + wj + 16 + (lane & 15)];
            if (need00) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            if (need01) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[0][1], 0, 0, 0);
            if (need10) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[1][0], 0, 0, 0);
            if (need11) acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
    }

    // sqrt(s_k) of the tile's rows / columns once per workgroup (the L2 epilogue needs them per element); they take
    // over the slab buffers, so the kernel needs 20 kB of LDS and eight workgroups fit a CU
    double (*sqI)[GT] = reinterpret_cast<double (*)[GT]>(sI);
    double (*sqJ)[GT] = reinterpret_cast<double (*)[GT]>(sJ);
    // On a log-uniform tau grid the DRT block of every penalty matrix is symmetric Toeplitz, M_k[i][j] = t_k[|i - j|]
    // with t_k = its first row: the tile needs the 127 differences around i0 - j0 only, staged behind the sqrt tables,
    // and no dense matrix is read in the epilogue (three 2 MB matrices per spectrum and outer iteration otherwise).
    constexpr int TW = 2 * GT;                                      // window slots per order (127 used)
    double (*tw)[TW] = reinterpret_cast<double (*)[TW]>(sI + 3 * GT);
    const int dbase = i0 - j0 - (GT - 1);                           // difference of window slot 0
    double fac[3] = {0, 0, 0};
    // Tiles of the DRT block that lie wholly beyond the reach of the penalty matrices -- every |i - j| of the tile larger than the
    // last non-zero entry of the Toeplitz first rows -- would add (sqrt(s_i) * 0) * sqrt(s_j) = 0 to every element: no tables,
    // no epilogue arithmetic, the same bits (28 of the 45 tiles of a 514 x 514 matrix lie two or more tile diagonals out; the 7
    // of them that touch the special-parameter columns qualify when those columns of the penalty matrices are zero outside the
    // special block, which the plan checks once)
    const bool l2on = g.s && !(!DOP && g.toep && g.toep_maxd >= 0 && (j0 >= g.ns || g.spec_zero) && i0 >= g.ns && dbase > g.toep_maxd);
    if (l2on) {
#pragma unroll
        for (int k = 0; k < 3; ++k) fac[k] = g.dfac[k] * (g.use_rho ? g.rho[(size_t)b * 3 + k] : 1.0);
        __syncthreads();   // last slab consumed
        const double* sb_ = g.s + (size_t)b * 3 * n;
        for (int e = tid; e < 3 * GT; e += 256) {
            const int k = e / GT, c = e % GT;
            sqI[k][c] = (i0 + c < n) ? sqrt(sb_[k * n + i0 + c]) : 0.0;
            sqJ[k][c] = (j0 + c < n) ? sqrt(sb_[k * n + j0 + c]) : 0.0;
        }
        if (g.toep) {
            // (the window holds M_k[|i - j|] * fac[k]: the product the epilogue used to form per element)
            const int nd = n - g.ns;                                // size of the DRT block
            for (int e = tid; e < 3 * TW; e += 256) {
                const int k = e / TW, sl = e % TW;
                int dd = dbase + sl;
                dd = dd < 0 ? -dd : dd;
                tw[k][sl] = (g.dfac[k] > 0.0 && dd < nd) ? g.mk[k][(size_t)g.ns * g.ldm + g.ns + dd] * fac[k] : 0.0;
            }
        }
        __syncthreads();
    }
    // epilogue: + L2, store lower tile and its mirror.  With the swapped operands the accumulator of lane l, register
    // r is element (row = l&15, column = (l>>4) + 4r) of the sub-tile.
    double* Pb = (ROWP && P) ? P + (size_t)b * p_stride : nullptr;   // row-major copy is optional (the resident QP kernel reads Ppk)
    double dfac2[3] = {0, 0, 0};
    if (l2on) {
        if (DOP) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dfac2[k] = g.dop_dfac[k] * (g.use_rho ? g.dop_rho[(size_t)b * 3 + k] : 1.0);
        }
    }
    const int dop_lo = DOP ? g.dop_start : 0, dop_hi = DOP ? g.dop_start + g.dop_size : 0;
    double* Pk = Ppk ? Ppk + (size_t)b * ppk_stride : nullptr;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double vals[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + a * 16 + (lane & 15);
                const int j = j0 + wj + c * 16 + (lane >> 4) + 4 * r;
                double v = 0.0;
                if (i < n && j < n) {
                    v = acc[a][c][r];
                    if (l2on) {
                        double l2 = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            if (g.dfac[k] > 0.0) {
                                // (i, j) = (lane&15, lane>>4 + 4r): reading the mirror element keeps the wave's
                                // addresses contiguous when the matrices are bitwise symmetric
                                double mv;
                                if (g.toep && i >= g.ns && j >= g.ns) {
                                    mv = tw[k][(i - j) - dbase];
                                } else {
                                    mv = g.sym ? g.mk[k][(size_t)j * g.ldm + i] : g.mk[k][(size_t)i * g.ldm + j];
                                    if (i >= g.ns && j >= g.ns) mv *= fac[k];
                                    else if (DOP && i >= dop_lo && i < dop_hi && j >= dop_lo && j < dop_hi) mv *= dfac2[k];
                                }
                                l2 += (sqI[k][i - i0] * mv) * sqJ[k][j - j0];
                            }
                        }
                        v += l2;
                    } else if (!g.s && g.l2) {
                        v += g.l2[(size_t)b * g.l2_stride + (size_t)i * g.ldl2 + j];
                    }
                    if (ROWP && Pb && !(diag && j > i)) {      // upper part of a diagonal tile comes from the mirror
                        Pb[(size_t)i * ldp + j] = v;
                        if (i != j) Pb[(size_t)j * ldp + i] = v;
                    }
                }
                vals[r] = v;
            }
            // second copy for the Cholesky in the factor's tile layout (qp_resident.hpp): double2 h*64 + i*4 + q holds
            // columns q + 8h, q + 8h + 4 of row i -- two 16-byte stores per lane, 2 KB contiguous per tile
            if (Pk) {
                const int tr = ((i0 + wi) >> 4) + a, tc = ((j0 + wj) >> 4) + c;
                if (tr < nchp && tc < nchp && tr >= tc) {
                    double2* tile = reinterpret_cast<double2*>(Pk + ((size_t)tr * nchp + tc) * 256);
                    const int fo = (lane & 15) * 4 + (lane >> 4);
                    tile[fo] = make_double2(vals[0], vals[1]);
                    tile[64 + fo] = make_double2(vals[2], vals[3]);
                }
            }
        }
}

// row-major symmetric P -> lower tiles in the factor's tile layout; one wavefront per tile, grid (tiles, B)
__global__ __launch_bounds__(64) void pack_p_kernel(int n, const double* __restrict__ P, int ldp, long long p_stride,
                                                    double* __restrict__ Ppk, long long ppk_stride, int nchp) {
    int t = blockIdx.x, tr = 0;
    while (t >= tr + 1) { t -= tr + 1; ++tr; }
    const int tc = t, b = blockIdx.y, lane = threadIdx.x;
    const double* Pb = P + (size_t)b * p_stride;
    double vals[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tr * 16 + (lane & 15), j = tc * 16 + (lane >> 4) + 4 * r;
        double v = 0.0;
        if (i < n && j < n) v = (i >= j) ? Pb[(size_t)i * ldp + j] : Pb[(size_t)j * ldp + i];   // lower triangle only
        vals[r] = v;
    }
    double2* tile = reinterpret_cast<double2*>(Ppk + (size_t)b * ppk_stride + ((size_t)tr * nchp + tc) * 256);
    const int fo = (lane & 15) * 4 + (lane >> 4);
    tile[fo] = make_double2(vals[0], vals[1]);
    tile[64 + fo] = make_double2(vals[2], vals[3]);
}

// q_b[i] = -sum_k (w_k A_ki)(w_k y_k) + l1_i ; grid (ceil(n/256), B)
__global__ __launch_bounds__(256) void qvec_kernel(int m, int n, const double* __restrict__ A, int lda,
                                                   const double* __restrict__ w, const double* __restrict__ y,
                                                   const double* __restrict__ l1, double l1_scalar,
                                                   double* __restrict__ q, const int* __restrict__ active,
                                                   long long a_stride) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    A += (size_t)b * a_stride;
    __shared__ double sw[256], swy[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double* wb = w + (size_t)b * m;
    const double* yb = y + (size_t)b * m;
    double acc = 0.0;
    for (int k0 = 0; k0 < m; k0 += 256) {
        __syncthreads();
        const int kk = k0 + threadIdx.x;
        if (kk < m) { const double wk = wb[kk]; sw[threadIdx.x] = wk; swy[threadIdx.x] = wk * yb[kk]; }
        __syncthreads();
        const int lim = (m - k0) < 256 ? (m - k0) : 256;
        if (i < n) {
            for (int k = 0; k < lim; ++k) acc += (sw[k] * A[(size_t)(k0 + k) * lda + i]) * swy[k];
        }
    }
    if (i < n) q[(size_t)b * n + i] = -acc + (l1 ? l1[i] : l1_scalar);
}

// row-major M[nrow][ncol] -> tiles [ntile_rows][nchp][256] in the factor's tile layout, columns shifted right by
// col_offset (the special-parameter slots), zero padding elsewhere; one wavefront per tile
__global__ __launch_bounds__(64) void pack_rows_kernel(int nrow, int ncol, int col_offset, const double* __restrict__ M,
                                                       int ldm, double* __restrict__ tiles, int nchp) {
    const int tr = blockIdx.y, tc = blockIdx.x, lane = threadIdx.x;
    double vals[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tr * 16 + (lane & 15), j = tc * 16 + (lane >> 4) + 4 * r - col_offset;
        vals[r] = (i < nrow && j >= 0 && j < ncol) ? M[(size_t)i * ldm + j] : 0.0;
    }
    double2* tile = reinterpret_cast<double2*>(tiles + ((size_t)tr * nchp + tc) * 256);
    const int fo = (lane & 15) * 4 + (lane >> 4);
    tile[fo] = make_double2(vals[0], vals[1]);
    tile[64 + fo] = make_double2(vals[2], vals[3]);
}

void launch_pack_rows(hipStream_t st, int nrow, int ncol, int col_offset, const double* M, int ldm, int ntile_rows,
                      double* tiles, int nchp) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(nchp, ntile_rows), dim3(64), 0, st, nrow, ncol, col_offset, M, ldm, tiles,
                       nchp);
}

// out = scale * Y Y' for the rows Y of `nex` tile rows (tile-packed, `nch` tiles per tile row, the first `ncol` tile columns
// count): the full posterior covariance from the rows B L^-T that the variance kernel leaves behind the factor.  One 16 x 16
// tile of the (symmetric) result per workgroup, lower tiles computed and mirrored; chunks in ascending order (fixed sums).
__global__ __launch_bounds__(256) void rows_outer_kernel(const double* __restrict__ Y, int nch, int ncol, int nrow, double scale,
                                                         double* __restrict__ out, int ld) {
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    __shared__ double A[256], Bt[256];
    const int t = threadIdx.x, i = t >> 4, j = t & 15;
    auto at = [](int r, int k) { return 2 * ((k >> 3) * 64 + r * 4 + (k & 3)) + ((k >> 2) & 1); };   // (row, column) inside a tile
    double acc = 0.0;
    for (int c = 0; c < ncol; ++c) {
        A[t] = Y[((size_t)ti * nch + c) * 256 + t];
        Bt[t] = Y[((size_t)tj * nch + c) * 256 + t];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += A[at(i, k)] * Bt[at(j, k)];
        __syncthreads();
    }
    const int r = ti * 16 + i, cc = tj * 16 + j;
    if (r < nrow && cc < nrow) {
        out[(size_t)r * ld + cc] = acc * scale;
        out[(size_t)cc * ld + r] = acc * scale;
    }
}

void launch_rows_outer(hipStream_t st, const double* Y, int nch, int ncol, int nex, int nrow, double scale, double* out, int ld) {
    hipLaunchKernelGGL(rows_outer_kernel, dim3(nex, nex), dim3(256), 0, st, Y, nch, ncol, nrow, scale, out, ld);
}

void launch_pack_p(hipStream_t st, int B, int n, const double* P, int ldp, long long p_stride, double* Ppk,
                   long long ppk_stride, int nchp) {
    const int nt = (n + 15) / 16;
    hipLaunchKernelGGL(pack_p_kernel, dim3(nt * (nt + 1) / 2, B), dim3(64), 0, st, n, P, ldp, p_stride, Ppk, ppk_stride,
                       nchp);
}

void launch_gram_l2(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w, const GramL2& g,
                    double* P, int ldp, long long p_stride, const int* active, double* Ppk, long long ppk_stride,
                    int nchp, long long a_stride) {
    const int nt = (n + GT - 1) / GT;
    const int ntile = nt * (nt + 1) / 2;
    const bool dop = g.s && g.dop_size > 0;
#define HIPDRT_GRAM(D, R) hipLaunchKernelGGL((gram_kernel<D, R>), dim3(ntile, B), dim3(256), 0, st, m, n, A, lda, w, g, P, ldp, \
                                             p_stride, active, ntile, Ppk, ppk_stride, nchp, a_stride)
    if (P) { if (dop) HIPDRT_GRAM(true, true); else HIPDRT_GRAM(false, true); }
    else { if (dop) HIPDRT_GRAM(true, false); else HIPDRT_GRAM(false, false); }
#undef HIPDRT_GRAM
}

void launch_qvec(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w, const double* y,
                 const double* l1, double l1_scalar, double* q, const int* active, long long a_stride) {
    // one matrix for the whole batch: a matrix product with a shared operand (hyper.hip: batch_products_kernel), whatever the
    // batch size -- a fit's bits do not depend on how many are fitted beside it; few fits with very large matrices keep the
    // vector kernel (a handful of workgroups would stream all of A)
    if (a_stride == 0 && (size_t)m * n < ((size_t)1 << 20)) {
        launch_qvec_batched(st, B, m, n, A, lda, w, y, l1, l1_scalar, q, active);
        return;
    }
    hipLaunchKernelGGL(qvec_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, m, n, A, lda, w, y, l1, l1_scalar, q,
                       active, a_stride);
}

void launch_weighted_gram(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w,
                          const double* b, const double* l2, long long l2_stride, int ldl2, const double* l1,
                          double* P, int ldp, long long p_stride, double* q, const int* active) {
    GramL2 g{};
    g.l2 = l2; g.l2_stride = l2_stride; g.ldl2 = ldl2; g.s = nullptr;
    launch_gram_l2(st, B, m, n, A, lda, w, g, P, ldp, p_stride, active);
    launch_qvec(st, B, m, n, A, lda, w, b, l1, 0.0, q, active);
}

}  // namespace hipdrt
