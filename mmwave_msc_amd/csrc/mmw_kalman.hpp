// mmw_kalman.hpp -- device code shared by k_kalman.hip (k_predict) and k_dbscan.hip (k_post: the Kalman
// update and the BallTree DBSCAN of a frame in one launch).
#pragma once

#include <cstddef>

#include "mmw_device.hpp"
#include "mmw_math.hpp"

namespace mmw {

// A TrackRec starts with 152 doubles (x, P, centroid, min, max, spread, group dispersion, N_est, lifetime)
// followed by its integers: the kernels copy that prefix into LDS as raw 8-byte words, ten independent
// loads per lane issued back to back (one global round trip), and pick the fields out of the copy.
constexpr int rX = 0, rP = 9, rCen = 90, rSpr = 108, rGd = 114, rNest = 150, rLife = 151, rInts = 152, kRecRaw = 154;
static_assert(offsetof(TrackRec, P) == rP * 8 && offsetof(TrackRec, centroid) == rCen * 8 && offsetof(TrackRec, spread) == rSpr * 8 &&
              offsetof(TrackRec, gd) == rGd * 8 && offsetof(TrackRec, n_est) == rNest * 8 && offsetof(TrackRec, lifetime) == rLife * 8 &&
              offsetof(TrackRec, point_num) == rInts * 8, "TrackRec prefix layout");
constexpr int kRecStage = 160;  // words actually copied: ten per lane of a 16-lane group (see stage_record)
static_assert(kRecStage >= kRecRaw && kRecStage * 8 <= (int)sizeof(TrackRec), "staged prefix");

// per-track LDS scratch (doubles)
constexpr int kPredScratch = kRecStage + 81 + 9 + 2;           // staged record | predict_math's scratch (kPredW)
constexpr int kUpdScratch = kRecStage + 81 + 54 + 54 + 6 + 36;   // staged record | update_math's scratch (kUpdW)

// All 64 lanes load and store unconditionally (the caller points idle groups at a valid record; nothing of
// theirs is stored later): ten loads, ten LDS stores, no branches.  The copy is 160 words, i.e. it runs a little
// into the record's ring bookkeeping -- inside the 187-word record: ring_len / uid, ring_n[], ring_slot[] sit in words
// 153..157 (k_scene reads them from the copy).  LP = lanes that own one track (16: four tracks per wave, the batched
// Kalman kernels; 32 / 64: the per-scene kernel k_scene, whose step is one track's latency), c = lane % LP.
template <int LP>
struct StageRegs { double t[(kRecStage + LP - 1) / LP]; };
// (the two halves apart: k_scene puts the frame's point loads between them -- requested behind the records, not in front)
template <int LP>
__device__ __forceinline__ void stage_record_load(const TrackRec *rec, int c, StageRegs<LP> &S)
{
    const double *src = reinterpret_cast<const double *>(rec);
    constexpr int NU = (kRecStage + LP - 1) / LP;
    constexpr bool exact = NU * LP == kRecStage;
#pragma unroll
    for (int u = 0; u < NU; u++) { const int w = c + LP * u; S.t[u] = src[(exact || w < kRecStage) ? w : kRecStage - 1]; }
}
template <int LP>
__device__ __forceinline__ void stage_record_store(double *R, int c, const StageRegs<LP> &S)
{
    constexpr int NU = (kRecStage + LP - 1) / LP;
    constexpr bool exact = NU * LP == kRecStage;
#pragma unroll
    for (int u = 0; u < NU; u++) { const int w = c + LP * u; if (exact || w < kRecStage) R[w] = S.t[u]; }
}
template <int LP = 16>
__device__ __forceinline__ void stage_record(const TrackRec *rec, double *R, int c)
{
    StageRegs<LP> S;
    stage_record_load<LP>(rec, c, S);
    stage_record_store<LP>(R, c, S);
}

// One Kalman update (update_state, Tracking.py:387-398; _get_Rc 299-312; filterpy's Joseph-form update) for the track
// whose record prefix is staged at `R` (LDS, kRecStage words), by the LP lanes that own it (c = lane % LP): the products
// are laid out over elements, so more lanes only shorten the loops -- the arithmetic per element is the same.  `W` = the
// track's scratch (kUpdW doubles), `rec` = where x and P go.  Idle groups (`live` false) compute and store nothing that
// matters.  All 64 lanes of the wave must call (the 6x6 inverse runs in the first 16 lanes of every group).
constexpr int wA = 0, wK = wA + 81, wSI = wK + 54, wC1 = wSI /* C1 replaces S^-1 once K is formed */, wY = wSI + 54, wRc = wY + 6,
              kUpdW = wRc + 36;
template <int DX, int LP>
__device__ __forceinline__ void update_math(TrackRec *rec, bool live, const double *R, double *W, int lane, int c, int &err)
{
    const double *Pw = R + rP;
    // Rc = Rm/N + ((N_est-N)/((N_est-1)N)) gd ; S = H P H^T + Rc ; SI = S^-1
    {
        const bool valid = live && c < 6;
        const int c16 = lane & 15;
        double v[6], det;
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = (c16 == i) ? 1.0 : 0.0;
        // (every lane of the group: the lanes that report the LU's outcome below must know that a zero denominator came first --
        //  Python raises ZeroDivisionError in _get_Rc, Tracking.py:299-312, before S is ever inverted)
        const double N = (double)reinterpret_cast<const int32_t *>(R + rInts)[0], nest = R[rNest];
        const double den = (nest - 1) * N;
        if (valid) {
            if (den == 0.0) err |= ERR_DIVZERO;
            const double coef = (nest - N) / den;
            const double hh = R[rSpr + c] / 2;
            const double dg = (hh * hh) / N;   // Rm[c][c] / N; the off-diagonal 0 / N is +0 (N >= 1): one division, same bits
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double rc = ((i == c) ? dg : 0.0) + coef * R[rGd + i * 6 + c];
                W[wRc + i * 6 + c] = rc;
                v[i] = Pw[i * 9 + c] + rc;  // S = H P H^T + R
            }
            W[wY + c] = R[rCen + c] - R[rX + c];  // y = z - H x
        }
        const bool ok = lu6_inverse_cols(v, lane, det);
        if (live && c < 16) {
            if (!ok && den != 0.0) err |= ERR_SINGULAR;   // (den == 0: S is inf / NaN because of the division, not singular)
            if (c >= 6 && c < 12) {
#pragma unroll
                for (int r = 0; r < 6; r++) W[wSI + r * 6 + c - 6] = v[r];
            }
        }
    }
    wave_sync();
    if constexpr (LP == 16) {
        // The batched kernels (four tracks per wave; k_post is bound by the instructions it issues, not by latency): lane c < DX
        // owns ROW c of K and C1 and COLUMN c of A and P, with the other index unrolled -- the identity's entries and the
        // m < 6 tests are compile-time, the operands every lane shares (S^-1, Rc, rows of K / A / C1) are LDS broadcasts, and nothing
        // is spent on k / 9, k % 9 and selects (81 elements dealt over 16 lanes took six rounds, the last with one lane).
        // Per element: the same operations in the same order as below.
        const bool act = live && c < DX;
        if (act) {  // K = P H^T S^-1, row c
            double prow[6];
#pragma unroll
            for (int m = 0; m < 6; m++) prow[m] = Pw[c * 9 + m];
#pragma unroll
            for (int cc = 0; cc < 6; cc++) {
                double a = prow[0] * W[wSI + cc];
#pragma unroll
                for (int m = 1; m < 6; m++) a += prow[m] * W[wSI + m * 6 + cc];
                W[wK + c * 6 + cc] = a;
            }
        }
        wave_sync();
        if (act) {
            const double *Kw = W + wK, *yw = W + wY, *Rcw = W + wRc;
            double pcol[DX];
#pragma unroll
            for (int m = 0; m < DX; m++) pcol[m] = Pw[m * 9 + c];
#pragma unroll
            for (int i = 0; i < DX; i++) {  // A = (I - K H) P, column c
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < DX; m++) {
                    const double d = (i == m) ? 1.0 : 0.0;
                    const double ikh = m < 6 ? d - Kw[i * 6 + m] : d;
                    a = (m == 0) ? ikh * pcol[0] : a + ikh * pcol[m];
                }
                W[wA + i * 9 + c] = a;
            }
            double krow[6];
#pragma unroll
            for (int m = 0; m < 6; m++) krow[m] = Kw[c * 6 + m];
            {  // x = x + K y
                double a = krow[0] * yw[0];
#pragma unroll
                for (int m = 1; m < 6; m++) a += krow[m] * yw[m];
                double xnew = R[rX + c] + a;
                if (c == 0) {  // Tracking.py:396-398: abs(variance.any()) > 0.6 <=> z[0] != x[0]
                    const double var = R[rCen] - xnew;
                    if (!(var == 0.0) && R[rLife] == 0.0) xnew += var * 0.4;
                }
                rec->x[c] = xnew;
            }
#pragma unroll
            for (int cc = 0; cc < 6; cc++) {  // C1 = K R, row c, into the S^-1 area (last read before the barrier above)
                double a = krow[0] * Rcw[cc];
#pragma unroll
                for (int m = 1; m < 6; m++) a += krow[m] * Rcw[m * 6 + cc];
                W[wC1 + c * 6 + cc] = a;
            }
        }
        wave_sync();
        if (act) {  // P = A (I-KH)^T + C1 K^T, column c
            const double *Aw = W + wA, *Kw = W + wK, *C1w = W + wC1;
            double kcc[6], ikh[DX];
#pragma unroll
            for (int m = 0; m < 6; m++) kcc[m] = Kw[c * 6 + m];
#pragma unroll
            for (int m = 0; m < DX; m++) {
                const double d = (c == m) ? 1.0 : 0.0;
                ikh[m] = m < 6 ? d - kcc[m] : d;
            }
#pragma unroll
            for (int i = 0; i < DX; i++) {
                double b = Aw[i * 9] * ikh[0];
#pragma unroll
                for (int m = 1; m < DX; m++) b += Aw[i * 9 + m] * ikh[m];
                double c2 = C1w[i * 6] * kcc[0];
#pragma unroll
                for (int m = 1; m < 6; m++) c2 += C1w[i * 6 + m] * kcc[m];
                rec->P[i * 9 + c] = b + c2;
            }
        }
        wave_sync();
        return;
    }
    if (live) {
        for (int k = c; k < DX * 6; k += LP) {  // K = P H^T S^-1
            const int i = k / 6, cc = k - i * 6;
            double a = Pw[i * 9] * W[wSI + cc];
#pragma unroll
            for (int m = 1; m < 6; m++) a += Pw[i * 9 + m] * W[wSI + m * 6 + cc];
            W[wK + k] = a;
        }
    }
    wave_sync();
    if (live) {
        const double *Kw = W + wK, *yw = W + wY, *Rcw = W + wRc;
        for (int k = c; k < 81; k += LP) {  // A = (I - K H) P
            const int i = k / 9, cc = k - i * 9;
            if (i < DX && cc < DX) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < DX; m++) {
                    const double d = (i == m) ? 1.0 : 0.0;
                    const double ikh = m < 6 ? d - Kw[i * 6 + m] : d;
                    a = (m == 0) ? ikh * Pw[cc] : a + ikh * Pw[m * 9 + cc];
                }
                W[wA + k] = a;
            }
        }
        if (c < DX) {  // x = x + K y
            double a = Kw[c * 6] * yw[0];
#pragma unroll
            for (int m = 1; m < 6; m++) a += Kw[c * 6 + m] * yw[m];
            double xnew = R[rX + c] + a;
            if (c == 0) {  // Tracking.py:396-398: abs(variance.any()) > 0.6 <=> z[0] != x[0]
                const double var = R[rCen] - xnew;
                if (!(var == 0.0) && R[rLife] == 0.0) xnew += var * 0.4;
            }
            rec->x[c] = xnew;
        }
        constexpr int NC = (54 + LP - 1) / LP;
        double c1[NC];  // C1 = K R, into the S^-1 area: every lane forms its entries first, then they are stored
#pragma unroll
        for (int u = 0; u < NC; u++) {
            const int k = c + LP * u;
            c1[u] = 0.0;
            if (k < DX * 6) {
                const int i = k / 6, cc = k - i * 6;
                double a = Kw[i * 6] * Rcw[cc];
#pragma unroll
                for (int m = 1; m < 6; m++) a += Kw[i * 6 + m] * Rcw[m * 6 + cc];
                c1[u] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < NC; u++) { const int k = c + LP * u; if (k < DX * 6) W[wC1 + k] = c1[u]; }
    }
    wave_sync();
    if (live) {
        const double *Aw = W + wA, *Kw = W + wK, *C1w = W + wC1;
        for (int k = c; k < 81; k += LP) {  // P = A (I-KH)^T + C1 K^T
            const int i = k / 9, cc = k - i * 9;
            if (i < DX && cc < DX) {
                double b = 0.0;
#pragma unroll
                for (int m = 0; m < DX; m++) {
                    const double d = (cc == m) ? 1.0 : 0.0;
                    const double ikh = m < 6 ? d - Kw[cc * 6 + m] : d;
                    b = (m == 0) ? Aw[i * 9] * ikh : b + Aw[i * 9 + m] * ikh;
                }
                double c2 = C1w[i * 6] * Kw[cc * 6];
#pragma unroll
                for (int m = 1; m < 6; m++) c2 += C1w[i * 6 + m] * Kw[cc * 6 + m];
                rec->P[k] = b + c2;
            }
        }
    }
    wave_sync();
}

// Lane K of every 16-lane row, to all lanes of that row: the one DPP control the 64-bit moves take (row_newbcast), as ONE
// v_mov_b64_dpp through the compiler's own builtin -- it knows the instruction is a DPP move, keeps the wait states a VALU write
// -> DPP read needs by SCHEDULING other work in between, and interleaves independent chains of products.  (Rounds 4-5 issued it
// from inline asm with a leading `s_nop 1` of its own, every broadcast chained to the accumulator it feeds so that the scheduler
// would not line all of them up: 492 of the update's 2249 instructions were s_nop, a fifth of its issue slots.)  All 64 lanes
// must be active (a DPP read of a disabled lane returns nothing useful), hence only in wave-uniform code.
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, false);
}
template <int K>
__device__ __forceinline__ double row_bcast_after(double v, double)
{
    return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, false);
}

// update_math for the batched kernels (16 lanes per track) with the operands the lanes of a track share -- S^-1, Rc, rows of K,
// A and C1 -- BROADCAST FROM THE REGISTERS of the lane that owns them (row_bcast) instead of written to and read back from the
// LDS: a wave's pass over four tracks moved 213 KB through the LDS, most of it 16 lanes reading the same eight bytes, and k_post
// was bound by that pipe (profiles/NOTEBOOK.md, round 4).  Lane c owns row c of K and C1, column c of A, P and (c < 6) of S, Rc;
// lanes 6..11 hold the columns of S^-1 when the inverse returns.  Every lane runs every instruction (lanes past DX and idle
// groups on garbage that is never stored): no divergence between a value's definition and its broadcast.  Per element the same
// operations in the same order as update_math.
template <int DX>
__device__ __forceinline__ void update_math_bcast(TrackRec *rec, bool live, const double *R, int lane, int c, int &err)
{
    const double *Pw = R + rP;
    const bool valid = live && c < 6;
    double v[6], rcol[6], det;
    const double N = (double)reinterpret_cast<const int32_t *>(R + rInts)[0], nest = R[rNest];
    const double den = (nest - 1) * N;
    if (valid && den == 0.0) err |= ERR_DIVZERO;
    const double coef = (nest - N) / den;
    const double hh = R[rSpr + c] / 2;           // (c >= 6: words of the staged record that mean something else -- never used)
    const double dg = (hh * hh) / N;             // Rm[c][c] / N; the off-diagonal 0 / N is +0 (N >= 1): one division, same bits
#pragma unroll
    for (int i = 0; i < 6; i++) {
        rcol[i] = ((i == c) ? dg : 0.0) + coef * R[rGd + i * 6 + c];   // Rc[i][c]
        const double sv = Pw[i * 9 + c] + rcol[i];                      // S = H P H^T + R
        v[i] = valid ? sv : ((c == i) ? 1.0 : 0.0);
    }
    const double y_own = R[rCen + c] - R[rX + c];   // y = z - H x (c < 6)
    const bool ok = lu6_inverse_cols(v, lane, det);
    if (live && !ok && den != 0.0) err |= ERR_SINGULAR;   // (den == 0: S is inf / NaN because of the division, not singular)
    // K = P H^T S^-1, row c: S^-1[m][cc] = v[m] of lane 6 + cc
    double krow[6];
    {
        double prow[6];
#pragma unroll
        for (int m = 0; m < 6; m++) prow[m] = Pw[c * 9 + m];
#define MMW_K_COL(cc)                                                                      \
        {                                                                                  \
            double a = prow[0] * row_bcast<6 + cc>(v[0]);                                  \
            a += prow[1] * row_bcast_after<6 + cc>(v[1], a);                                        \
            a += prow[2] * row_bcast_after<6 + cc>(v[2], a);                                        \
            a += prow[3] * row_bcast_after<6 + cc>(v[3], a);                                        \
            a += prow[4] * row_bcast_after<6 + cc>(v[4], a);                                        \
            a += prow[5] * row_bcast_after<6 + cc>(v[5], a);                                        \
            krow[cc] = a;                                                                  \
        }
        MMW_K_COL(0) MMW_K_COL(1) MMW_K_COL(2) MMW_K_COL(3) MMW_K_COL(4) MMW_K_COL(5)
#undef MMW_K_COL
    }
    {  // x = x + K y: y[m] = y_own of lane m
        double a = krow[0] * row_bcast<0>(y_own);
        a += krow[1] * row_bcast_after<1>(y_own, a);
        a += krow[2] * row_bcast_after<2>(y_own, a);
        a += krow[3] * row_bcast_after<3>(y_own, a);
        a += krow[4] * row_bcast_after<4>(y_own, a);
        a += krow[5] * row_bcast_after<5>(y_own, a);
        double xnew = R[rX + c] + a;
        if (c == 0) {  // Tracking.py:396-398: abs(variance.any()) > 0.6 <=> z[0] != x[0]
            const double var = R[rCen] - xnew;
            if (!(var == 0.0) && R[rLife] == 0.0) xnew += var * 0.4;
        }
        if (live && c < DX) rec->x[c] = xnew;
    }
    // C1 = K R, row c: Rc[m][cc] = rcol[m] of lane cc
    double c1row[6];
#define MMW_C1_COL(cc)                                                                     \
    {                                                                                      \
        double a = krow[0] * row_bcast<cc>(rcol[0]);                                       \
        a += krow[1] * row_bcast_after<cc>(rcol[1], a);                                             \
        a += krow[2] * row_bcast_after<cc>(rcol[2], a);                                             \
        a += krow[3] * row_bcast_after<cc>(rcol[3], a);                                             \
        a += krow[4] * row_bcast_after<cc>(rcol[4], a);                                             \
        a += krow[5] * row_bcast_after<cc>(rcol[5], a);                                             \
        c1row[cc] = a;                                                                     \
    }
    MMW_C1_COL(0) MMW_C1_COL(1) MMW_C1_COL(2) MMW_C1_COL(3) MMW_C1_COL(4) MMW_C1_COL(5)
#undef MMW_C1_COL
    // A = (I - K H) P, column c: K[i][m] = krow[m] of lane i
    double pcol[DX], acol[DX];
#pragma unroll
    for (int m = 0; m < DX; m++) pcol[m] = Pw[m * 9 + c];
#define MMW_A_ROW(i)                                                                       \
    if (i < DX) {                                                                          \
        double a = 0.0;                                                                    \
        _Pragma("unroll") for (int m = 0; m < DX; m++) {                                   \
            const double d = (i == m) ? 1.0 : 0.0;                                         \
            double ikh = d;                                                                \
            if (m < 6) ikh = d - row_bcast_after<i>(krow[m < 6 ? m : 0], a);                        \
            a = (m == 0) ? ikh * pcol[0] : a + ikh * pcol[m];                              \
        }                                                                                  \
        acol[i < DX ? i : 0] = a;                                                          \
    }
    MMW_A_ROW(0) MMW_A_ROW(1) MMW_A_ROW(2) MMW_A_ROW(3) MMW_A_ROW(4) MMW_A_ROW(5) MMW_A_ROW(6) MMW_A_ROW(7) MMW_A_ROW(8)
#undef MMW_A_ROW
    // P = A (I-KH)^T + C1 K^T, column c: A[i][m] = acol[i] of lane m, C1[i][m] = c1row[m] of lane i
    double ikhc[DX];
#pragma unroll
    for (int m = 0; m < DX; m++) {
        const double d = (c == m) ? 1.0 : 0.0;
        ikhc[m] = m < 6 ? d - krow[m < 6 ? m : 0] : d;
    }
#define MMW_P_TERM(i, m) row_bcast_after<m>(acol[i], b)
#define MMW_P_TERM0(i) row_bcast<0>(acol[i])
#define MMW_P_ROW(i)                                                                       \
    if (i < DX) {                                                                          \
        double b = MMW_P_TERM0(i) * ikhc[0];                                             \
        b += MMW_P_TERM(i, 1) * ikhc[1];                                                   \
        b += MMW_P_TERM(i, 2) * ikhc[2];                                                   \
        b += MMW_P_TERM(i, 3) * ikhc[3];                                                   \
        b += MMW_P_TERM(i, 4) * ikhc[4];                                                   \
        b += MMW_P_TERM(i, 5) * ikhc[5];                                                   \
        if (DX > 6) {                                                                      \
            b += MMW_P_TERM(i, 6) * ikhc[DX > 6 ? 6 : 0];                                  \
            b += MMW_P_TERM(i, 7) * ikhc[DX > 7 ? 7 : 0];                                  \
            b += MMW_P_TERM(i, 8) * ikhc[DX > 8 ? 8 : 0];                                  \
        }                                                                                  \
        double c2 = row_bcast<i>(c1row[0]) * krow[0];                                      \
        c2 += row_bcast_after<i>(c1row[1], c2) * krow[1];                                            \
        c2 += row_bcast_after<i>(c1row[2], c2) * krow[2];                                            \
        c2 += row_bcast_after<i>(c1row[3], c2) * krow[3];                                            \
        c2 += row_bcast_after<i>(c1row[4], c2) * krow[4];                                            \
        c2 += row_bcast_after<i>(c1row[5], c2) * krow[5];                                            \
        if (live && c < DX) rec->P[i * 9 + c] = b + c2;                                    \
    }
    MMW_P_ROW(0) MMW_P_ROW(1) MMW_P_ROW(2) MMW_P_ROW(3) MMW_P_ROW(4) MMW_P_ROW(5) MMW_P_ROW(6) MMW_P_ROW(7) MMW_P_ROW(8)
#undef MMW_P_ROW
#undef MMW_P_TERM
#undef MMW_P_TERM0
    wave_sync();
}

// ... for a record that is still in global memory (the batched kernels): `rec` = the group's track record (idle groups:
// any valid record), `Wj` = the group's kUpdScratch doubles = [staged record | scratch].
template <int DX>
__device__ __forceinline__ void update_one_track(TrackRec *rec, bool live, double *Wj, int lane, int c, int &err)
{
    stage_record<16>(rec, Wj, c);
    wave_sync();
#ifdef MMW_UPD_LDS   // (diagnostic builds: the shared operands through the LDS, as before round 4's broadcast form)
    update_math<DX, 16>(rec, live, Wj, Wj + kRecStage, lane, c, err);
#else
    update_math_bcast<DX>(rec, live, Wj, lane, c, err);
#endif
}

// _predict_all for ONE track per LP-lane group (predict_state, Tracking.py:372-385: filterpy predict with the motion
// model of constants.py:195-215) + the gate matrix of _calc_dist_fun (Tracking.py:549-560) into `G` (the track's record in
// gate_buf), on the record prefix staged at `R` (LDS; x and P are replaced by the predicted ones there, and in `rec` when
// STORE), scratch `W` (kPredW doubles).  Idle groups (`live` false) store nothing.  All 64 lanes of the wave must call.
constexpr int wPA = 0, wXn = wPA + 81, kPredW = wXn + 9 + 2;
template <int DX, int LP, bool STORE>
__device__ __forceinline__ void predict_math(const DevCfg &cfg, TrackRec *rec, double *G, bool live, double dt, double *R, double *W, int lane,
                                             int c, int &err)
{
    const double dtm = R[rLife] + dt;
    const double h = 0.5 * (dtm * dtm);
    if (live) {
        for (int k = c; k < 81; k += LP) {
            // A = F P.  F has ones on the diagonal, dt at (i,i+3), h at (i,i+6): the k-ordered dense
            // dot product reduces to these terms (the others are exact zeros).
            const int i = k / 9, cc = k - i * 9;
            if (i < DX && cc < DX) {
                double a = R[rP + k];
                if (i + 3 < DX) a += dtm * R[rP + (i + 3) * 9 + cc];
                if (i + 6 < DX) a += h * R[rP + (i + 6) * 9 + cc];
                W[wPA + k] = a;
            }
        }
        if (c < DX) {
            double xn = R[rX + c];
            if (c + 3 < DX) xn += dtm * R[rX + c + 3];
            if (c + 6 < DX) xn += h * R[rX + c + 6];
            W[wXn + c] = xn;
        }
    }
    wave_sync();
    if (live) {
        const double dt2 = dtm * dtm, dt3 = dt2 * dtm, dt4 = dt2 * dt2;
        for (int k = c; k < 81; k += LP) {
            const int i = k / 9, cc = k - i * 9;
            if (i < DX && cc < DX) {
                double b = W[wPA + k];  // B = A F^T
                if (cc + 3 < DX) b += W[wPA + i * 9 + cc + 3] * dtm;
                if (cc + 6 < DX) b += W[wPA + i * 9 + cc + 6] * h;
                double qn = 0.0;
                if (i / 3 == cc / 3) {  // block_diag of Q_discrete_white_noise(dim=3) (constants.py:210-215)
                    const int qi = i % 3, qc = cc % 3, sdeg = qi + qc;
                    const double base = sdeg == 0 ? 0.25 * dt4 : sdeg == 1 ? 0.5 * dt3 : sdeg == 2 ? ((qi == 1) ? dt2 : 0.5 * dt2)
                                      : sdeg == 3 ? dtm : 1.0;
                    qn = base * cfg.kf_q_std;
                }
                const double pn = b + qn;
                if (STORE) rec->P[k] = pn;
                R[rP + k] = pn;
            }
        }
        if (c < DX) { const double xn = W[wXn + c]; if (STORE) rec->x[c] = xn; R[rX + c] = xn; }
    }
    wave_sync();
    // gate matrix: lane c < 6 of the group holds column c of C = P[:6,:6] + diag((spread/2)^2) + group_disp_est
    {
        const bool valid = live && c < 6;
        const int c16 = lane & 15;
        double v[6], det;
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = (c16 == i) ? 1.0 : 0.0;  // idle groups: identity
        if (valid) {
            const double hh = R[rSpr + c] / 2;
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = (R[rP + i * 9 + c] + ((i == c) ? hh * hh : 0.0)) + R[rGd + i * 6 + c];
        }
        const bool ok = lu6_inverse_cols(v, lane, det);
        if (live && c < 16) {
            if (!ok) err |= ERR_SINGULAR;
            if (c >= 6 && c < 12) {
#pragma unroll
                for (int r = 0; r < 6; r++) G[r * 6 + c - 6] = v[r];
            }
            if (c == 0) G[36] = dlog(fabs(det));
            if (c < 6) G[37 + c] = R[rX + c];
        }
    }
    wave_sync();
}

// lane i + N of the 16-lane row, for lane i (lanes past the row's end: 0): two 32-bit DPP moves (row_shl)
template <int N>
__device__ __forceinline__ double row_from_right(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// predict_math for the batched kernel (k_predict: 16 lanes per track, the record written back to global memory, nothing read from
// the LDS copy afterwards) with lane c owning COLUMN c of P: A = F P needs nothing but the lane's own column (F shifts ROWS), B = A F^T
// takes columns c + 3 and c + 6 from the lanes three and six to the right (row_from_right), and the gate matrix wants exactly the
// first six entries of the lane's column -- no scratch in the LDS, no k / 9, k % 9.  Per element the same operations in the same
// order as predict_math.  Every lane runs every instruction; stores are guarded.
template <int DX>
__device__ __forceinline__ void predict_math_cols(const DevCfg &cfg, TrackRec *rec, double *G, bool live, double dt, const double *R, int lane, int c,
                                                  int &err)
{
    const double dtm = R[rLife] + dt;
    const double h = 0.5 * (dtm * dtm);
    const double dt2 = dtm * dtm, dt3 = dt2 * dtm, dt4 = dt2 * dt2;
    const bool own = c < DX;
    double acol[DX], pn[DX];
    {   // A = F P, column c (F has ones on the diagonal, dt at (i,i+3), h at (i,i+6): the k-ordered dense dot product reduces to these terms)
        double pcol[DX];
#pragma unroll
        for (int i = 0; i < DX; i++) pcol[i] = R[rP + i * 9 + (own ? c : 0)];
#pragma unroll
        for (int i = 0; i < DX; i++) {
            double a = pcol[i];
            if (i + 3 < DX) a += dtm * pcol[i + 3 < DX ? i + 3 : 0];
            if (i + 6 < DX) a += h * pcol[i + 6 < DX ? i + 6 : 0];
            acol[i] = a;
        }
    }
    double xn = R[rX + (own ? c : 0)];
    {
        const double x3 = R[rX + (c + 3 < DX ? c + 3 : 0)], x6 = R[rX + (c + 6 < DX ? c + 6 : 0)];
        if (c + 3 < DX) xn += dtm * x3;
        if (c + 6 < DX) xn += h * x6;
    }
    const int qc = c % 3, cb = c / 3;
#pragma unroll
    for (int i = 0; i < DX; i++) {  // B = A F^T + Q, column c
        const double a3 = row_from_right<3>(acol[i]), a6 = row_from_right<6>(acol[i]);
        double b = acol[i];
        if (c + 3 < DX) b += a3 * dtm;
        if (c + 6 < DX) b += a6 * h;
        double qn = 0.0;
        if (i / 3 == cb) {  // block_diag of Q_discrete_white_noise(dim=3) (constants.py:210-215)
            const int qi = i % 3, sdeg = qi + qc;
            const double base = sdeg == 0 ? 0.25 * dt4 : sdeg == 1 ? 0.5 * dt3 : sdeg == 2 ? ((qi == 1) ? dt2 : 0.5 * dt2)
                              : sdeg == 3 ? dtm : 1.0;
            qn = base * cfg.kf_q_std;
        }
        pn[i] = b + qn;
        if (live && own) rec->P[i * 9 + c] = pn[i];
    }
    if (live && own) rec->x[c] = xn;
    // gate matrix: lane c < 6 of the group holds column c of C = P[:6,:6] + diag((spread/2)^2) + group_disp_est
    const bool valid = live && c < 6;
    double v[6], det;
    const double hh = R[rSpr + c] / 2;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double cv = (pn[i] + ((i == c) ? hh * hh : 0.0)) + R[rGd + i * 6 + c];
        v[i] = valid ? cv : ((c == i) ? 1.0 : 0.0);  // idle groups: identity
    }
    const bool ok = lu6_inverse_cols(v, lane, det);
    if (live) {
        if (!ok) err |= ERR_SINGULAR;
        if (c >= 6 && c < 12) {
#pragma unroll
            for (int r = 0; r < 6; r++) G[r * 6 + c - 6] = v[r];
        }
        if (c == 0) G[36] = dlog(fabs(det));
        if (c < 6) G[37 + c] = xn;
    }
    wave_sync();
}

// ... for a record that is still in global memory: `Wj` = the group's kPredScratch doubles = [staged record | scratch];
// the gate record goes to gate_buf[s][j] (by effective_tracks position).
// (the arithmetic alone, on a record already staged in Wj: the track-wise k_predict stages it before it has looked at the scene's words)
template <int DX>
__device__ __forceinline__ void predict_staged_track(const DevCfg &cfg, const DevState &st, TrackRec *rec, bool live, int s, int j, double dt,
                                                     double *Wj, int lane, int c, int &err)
{
#ifdef MMW_PRED_LDS   // (diagnostic builds: the element-wise form with its LDS scratch, as before round 4's column form)
    predict_math<DX, 16, true>(cfg, rec, st.gate_buf + ((size_t)s * cfg.t_cap + j) * kGateRec, live, dt, Wj, Wj + kRecStage, lane, c, err);
#else
    predict_math_cols<DX>(cfg, rec, st.gate_buf + ((size_t)s * cfg.t_cap + j) * kGateRec, live, dt, Wj, lane, c, err);
#endif
}
template <int DX>
__device__ __forceinline__ void predict_one_track(const DevCfg &cfg, const DevState &st, TrackRec *rec, bool live, int s, int j, double dt,
                                                  double *Wj, int lane, int c, int &err)
{
    stage_record<16>(rec, Wj, c);
    wave_sync();
    predict_staged_track<DX>(cfg, st, rec, live, s, j, dt, Wj, lane, c, err);
}

// When the track-wise layout is used: a list entry packs (scene, position, slot) into one word (t_cap <= 63, fewer than 2^19
// scenes), and it only pays when there are more
// waves than the chip runs at once -- a small context (256 scenes x 4 tracks) is bound by the latency of one wave, and
// the lists put a dependent load in front of it (k_predict 8.3 -> 9.2 us there with the two of rounds 2-4).  The threshold is
// 1024 waves for contexts of more than 512 scenes; mmw_config.kalman_dense_min_units overrides it (the parity tests run both
// layouts on small contexts: tests/_layouts.py).
__host__ __device__ inline bool tracks_dense(const DevCfg &cfg, int nq) { return cfg.t_cap <= 63 && cfg.n_scenes < kUpdMaxScenes && cfg.n_scenes * nq > cfg.dense_min_units; }
// Contexts whose step is launch latency (<= kSmallContextScenes scenes, per-scene layout -- mmw_create picks it for them unless
// told otherwise): _predict_all runs at the head of k_track (k_track.hip, PRED instantiations) and k_predict is not
// launched -- one kernel boundary less.
__host__ __device__ inline bool pred_in_track(const DevCfg &cfg)
{
    int nq = (cfg.tr_max_tracks + 3) / 4;
    if (nq < 1) nq = 1;
    return cfg.n_scenes <= kSmallContextScenes && !tracks_dense(cfg, nq) && !cfg.seek_inner && !cfg.fused;
}

// _update_all laid out over the TRACKS of the context: wave `unit` takes four consecutive entries of ITS SHARD of the update
// lists k_track built this frame (mmw_device.hpp: upd_shards; every scene in them was tracked this frame, entries = its tracks
// 0 .. hdr->n_upd - 1).  Four real tracks per wave whatever the scenes hold (per-scene waves ran at 75 % of their lanes with
// 1..8 tracks per scene).  This frame's new tracks (spawn_scene, in worker blocks of the same launch) are not in the lists: they
// are not updated (Tracking.py:598-603 runs before _add_tracks).  `lds` = this wave's 4 * kUpdScratch doubles.
struct UpdCursor {
    const int32_t *list;
    int tot, k, stride4, last;   // this 16-lane group's entry index, the step between its entries, the region's last index
};
// (`n_units` = the units the launch runs for the dense work, at least the shard count: a launch smaller than the lists walks them
//  with that stride)
__device__ __forceinline__ UpdCursor upd_cursor(const DevCfg &cfg, const DevState &st, int unit, int n_units, int parity, int g, int &entry)
{
    const int nsh = upd_shards(cfg.n_scenes * kalman_waves_per_scene(cfg.tr_max_tracks)), sh = unit % nsh, u = unit / nsh;
    const size_t region = upd_region(cfg.n_scenes, cfg.t_cap);
    UpdCursor C;
    C.list = st.upd_list + ((size_t)parity * kUpdShards + sh) * region;
    C.stride4 = ((n_units - sh + nsh - 1) / nsh) * 4;   // (the grid is sized for tr_max_tracks per scene; a scene may hold more right after a frame of many new clusters)
    C.last = (int)region - 1;
    C.k = u * 4 + g;
    C.tot = st.upd_count[parity * kUpdWords + sh];
    entry = C.list[C.k < C.last ? C.k : C.last];   // with the length, not behind it: past the length it is stale and ignored
    return C;
}
template <int DX>
__device__ __forceinline__ void update_tracks_dense(const DevCfg &cfg, const DevState &st, int unit, int n_units, int parity, double *lds)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    int e;
    UpdCursor C = upd_cursor(cfg, st, unit, n_units, parity, g, e);
    while (C.k - g < C.tot) {   // (uniform)
        const bool live = C.k < C.tot;
        const int my_s = live ? e >> 12 : 0;
        TrackRec *rec = st.trk + (size_t)my_s * cfg.t_cap + (live ? (e & 63) : 0);
        int err = 0;
        update_one_track<DX>(rec, live, lds + g * kUpdScratch, lane, c, err);
        if (err && live) atomicOr(&st.hdr[my_s].err, err);
        C.k += C.stride4;
        if (C.k - g < C.tot) e = C.list[C.k < C.last ? C.k : C.last];
    }
}

// The per-scene layout (kept for contexts with t_cap > 63): tracks 4q.., 4(q+nq).. of scene s by ONE wave.
template <int DX>
__device__ __forceinline__ void update_tracks_wave(const DevCfg &cfg, const DevState &st, const int32_t *__restrict__ n_pts, int s, int q,
                                                   int nq, double *lds)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int n = n_pts[s];
    if (!frame_reaches_track(n, cfg.max_pts)) return;
    const SceneHdr *hdr = st.hdr + s;
    const int T = hdr->n_upd;
    if (q * 4 >= T) return;
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    double *Wj = lds + g * kUpdScratch;
    int err = 0;
    for (int j0 = q * 4; j0 < T; j0 += nq * 4) {
        const int j = j0 + g;
        const bool live = j < T;
        update_one_track<DX>(trk + (live ? order[j] : 0), live, Wj, lane, c, err);
    }
    if (err) atomicOr(&st.hdr[s].err, err);
}

}  // namespace mmw
