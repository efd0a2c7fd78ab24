// k_dbscan.hip -- apply_DBscan (Utils.py:250-291) with scikit-learn's BallTree
// semantics, one workgroup per scene, followed by TrackBuffer._add_tracks
// (Tracking.py:576-589, 697-703).
//
// Why a BallTree on a GPU: the reference hands sklearn a Python callable metric,
// sklearn answers with BallTree(leaf_size=30) over all 8 columns, and because the
// "distance" violates the triangle inequality the tree's prune / take-all
// shortcuts change the neighbour sets (13 % of queries differ from brute force on
// the synthetic scenes).  Bit-matching cluster ids therefore means reproducing the
// tree: split dimensions over 8 features, (value,index) median partition, ball
// centroids/radii, and the per-node PRUNE / ALL / leaf-TEST decision of
// BinaryTree._query_radius_single (sklearn/neighbors/_binary_tree.pxi.tp:1903-1980).
//
// Layout: x,y,z of the candidate points sit in LDS as fp64 SoA (indexed by point
// index); `idx` maps tree position -> point index so every node is a contiguous
// position range.  A query leaves a 2-bit state per leaf (64-bit mask per point); for
// clouds of <= 512 points it also records the neighbourhood itself as a bit row, and
// the labelling runs on the (transposed) bit matrix alone; larger clouds re-use the
// leaf states.  query_radius is walked one WAVE at a time (uniform node / level).
//
// Who runs it: the cloud is the concatenation of <= ring frames of UNASSIGNED points, so its
// size U varies from a few dozen (steady state: clutter only) to ring*max_pts.
//   U <= 256          256 threads, thread i owns point i with all 8 columns in registers: the worker blocks of k_post
//   U <= 1920         512 threads (thread per point up to 512, strided build above): k_dbscan_big / k_dbscan_startup
//   both, early       the 512-thread workgroups of k_chain on a side stream, while k_track and k_post are still running
// k_track pushes every scene that must cluster into one of two queues (or, without side workers, a work list); LDS is
// carved per capacity so that small clouds do not pay for the largest one.
#include <cstdlib>
#include "mmw_device.hpp"
#include "mmw_math.hpp"
#include "mmw_cloud.hpp"
#include "mmw_kalman.hpp"
#include "mmw_launch.hpp"

namespace mmw {

// Diagnostic build only (make STAMPS=1): per-phase cycle sums of lane 0 into stats[20 + phase].
#ifdef MMW_STAMPS
#define DSTAMP(k)                                                                            \
    do {                                                                                     \
        if (threadIdx.x == 0 && dbg) {                                                       \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();                  \
            atomicAdd(&dbg[20 + (k)], t_now - t_prev);                                       \
            t_prev = t_now;                                                                  \
        }                                                                                    \
    } while (0)
#define DSTAMP_INIT unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#else
#define DSTAMP(k)
#define DSTAMP_INIT
#endif

struct DbLds {
    double *X, *Y, *Z;             // [UM] by point index during the build, by tree position afterwards
    double *key;                   // [UM] split value by position (build) ...
    unsigned long long *mask;      // ... aliased: per-position leaf-state mask (query/label)
    int *idx;                      // [UM] position -> point index
    int *idx2;                     // [UM] partition target (build); labels by point index (output)
    int *lab;                      // [UM] labels by position
    int *front;                    // [UM]
    int *next;                     // [UM]
    unsigned char *core;           // [UM]
    unsigned char *leafpos;        // [UM] leaf number of a position
    int *nstart, *nend;            // [nodes+1]
    double *nsum;                  // [nodes][3]
    double *ncen;                  // [nodes][3]
    unsigned long long *nrad;      // [nodes] radius as raw bits (>= 0 so bit order == value order)
    unsigned long long *mm;        // [leaves/2][8][2] sortable min/max keys of the nodes of one level
    int *sdim;                     // [leaves/2]
    int *lbase;                    // [leaves/2] left-count scan value at node start
    int *blk;                      // [UM/64 + 1] block counts / prefixes
    int *misc;                     // [16]
    int *cnt;                      // [NB][CL+1] cluster member counting (spawn)
    int *cl_n, *cl_off;            // [CL+2]
    double *ccen;                  // [CL+1][6]
    double *fst;                   // [kFrontChunk][4] frontier staging of the labelling: mask bits, x, y, z
    unsigned long long *adj;       // [min(UM, kAdjMax)][W] + [4][8]: the eps-neighbourhoods of clouds of <= kAdjMax points as bit rows
                                   // (tree positions), then the frontier / reached / labelled / core sets of the labelling
};

__host__ __device__ inline size_t db_align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline int db_pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__host__ __device__ inline int db_levels(int U)
{
    // BinaryTree.__init__: n_levels = int(log2(max(1,(n-1)/leaf_size)) + 1)  (_binary_tree.pxi.tp:876-878)
    int n_levels = 1;
    while ((U - 1) >= kLeafSize * (1 << n_levels)) n_levels++;
    return n_levels;
}

// WRITE=false only sizes the layout (see k_track.hip: no null test on the private struct).
// ALL8 build: private copies of the per-node min/max keys (picked by lane & 7), merged after the atomics -- all
// lanes of a wave hitting the same sixteen LDS words serialise 64-fold
constexpr int kMmCopies = 8;
// (the 512-thread build of k_dbscan_big takes four: with eight its carve-up would cost a workgroup per CU)
__host__ __device__ inline int db_mm_copies(int UM) { return UM > 256 ? 4 : kMmCopies; }
constexpr int kFrontChunk = 32;
// Clouds of up to kAdjMax points keep their neighbourhoods as bit rows (query_radius writes them, dbscan_inner then works on
// bits alone): 32 KiB at 512 points, 8 KiB at 256.
constexpr int kAdjMax = 512;
// (the carve-up of the largest capacities, 1537 .. 1920 points, has room for 256-point rows only: 160 KiB of LDS)
__host__ __device__ inline int db_adj_cap(int UM) { return UM > 1536 ? 256 : (UM < kAdjMax ? UM : kAdjMax); }
// (rows are ((U + 63) / 64) | 1 words apart: an odd stride keeps the lanes of a wave, one row each, on different banks)
__host__ __device__ inline int db_adj_words(int UM)
{
    const int cap = db_adj_cap(UM);
    return cap * (((cap + 63) / 64) | 1) + 32;
}

// Transpose a 64 x 64 bit tile held one row per lane (bit c of lane i's word <-> bit i of lane c's word): six rounds of
// swapping the off-diagonal blocks with the partner lane.
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, int lane)
{
    unsigned long long m = 0x00000000FFFFFFFFULL;
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)x, j), hi = __shfl_xor((unsigned)(x >> 32), j);
        const unsigned long long y = ((unsigned long long)hi << 32) | lo;
        x = (lane & j) == 0 ? (x & m) | ((y & m) << j) : (x & ~m) | ((y & ~m) >> j);
        m ^= m << (j >> 1);
    }
    return x;
}

// MW = 64-bit words of a position's leaf-state mask (two bits per leaf): 1 for the LDS-resident classes (<= 32 leaves, the mask
// lies over key[]), more for the clouds of more than 1920 points, whose carve-up lives in global memory (k_dbscan_huge).
__host__ __device__ inline int db_mask_words(int UM)
{
    const int leaves = 1 << (db_levels(UM) - 1);
    return (2 * leaves + 63) / 64 > 1 ? (2 * leaves + 63) / 64 : 1;
}
__host__ __device__ inline int db_front_stride(int MW) { return MW == 1 ? 4 : 4 + MW; }  // doubles per staged frontier entry: mask word 0, x, y, z, (mask words 1..)
template <bool WRITE>
__host__ __device__ __forceinline__ size_t db_lds_layout(int UM, int CL, bool all8, char *base, DbLds *L, int MW = 1)
{
    const int NB = (UM + 63) / 64;
    const int levels = db_levels(UM), nodes = (1 << levels) - 1, half = (1 << (levels - 1)) / 2 > 0 ? (1 << (levels - 1)) / 2 : 1;
    size_t off = 0;
#define CARVE(field, type, count)                       \
    if constexpr (WRITE) L->field = (type *)(base + off);  \
    off = db_align16(off + sizeof(type) * (size_t)(count));
    CARVE(X, double, UM)
    CARVE(Y, double, UM)
    CARVE(Z, double, UM)
    if constexpr (WRITE) L->mask = (unsigned long long *)(base + off);
    // (thread-per-point build: key[] / front[] are also the cross-wave exchange of the sort, one slot per THREAD -- 256
    // threads up to a capacity of 256 points, 512 wherever it is above, see db_mm_copies -- whatever the capacity: a context
    // whose ring holds fewer points than the kernel has threads still sorts over every thread slot.  With UM slots the
    // idle threads' keys ran over idx[] .. nend[] for rings of fewer than ~150 points: found by tests/test_gpu_fuzz.py)
    const int xslots = UM <= 256 ? 256 : (UM < 512 ? 512 : UM);
    CARVE(key, double, all8 ? xslots : db_pow2ceil(UM))   // (generic build: also the 64-bit half of the sort keys, one per slot)
    CARVE(idx, int, UM)
    CARVE(idx2, int, UM)
    CARVE(lab, int, UM)
    CARVE(front, int, all8 ? xslots : db_pow2ceil(UM))    // (generic build: the 32-bit half of the sort keys)
    CARVE(next, int, UM)
    CARVE(core, unsigned char, UM)
    CARVE(leafpos, unsigned char, UM)
    CARVE(nstart, int, nodes + 1)
    CARVE(nend, int, nodes + 1)
    CARVE(nsum, double, nodes * 3)
    CARVE(ncen, double, nodes * 3)
    CARVE(nrad, unsigned long long, nodes + 1)
    CARVE(mm, unsigned long long, half * 16 * (all8 ? db_mm_copies(UM) : 1))
    CARVE(sdim, int, half)
    CARVE(lbase, int, half)
    CARVE(blk, int, (NB + 1) > 64 ? (NB + 1) : 64)
    CARVE(misc, int, 16)
    CARVE(cnt, int, NB *(CL + 1))
    CARVE(cl_n, int, CL + 2)
    CARVE(cl_off, int, CL + 2)
    CARVE(ccen, double, (CL + 1) * 6)
    CARVE(fst, double, kFrontChunk * db_front_stride(MW))
    CARVE(adj, unsigned long long, db_adj_words(UM))
    if (MW > 1) { CARVE(mask, unsigned long long, (size_t)UM * MW) }
#undef CARVE
    return off;
}

// The carve-up of a cloud the LDS cannot hold (k_dbscan_huge): the arrays every phase hammers -- the three coordinate columns,
// the exchange slots of the level sort, the per-node min / max words of the build -- in the LDS (152 KB at 4096 points), all the
// others in a slab of global memory.  Returns the slab's bytes, *hot_bytes = the LDS bytes.  (generic build only: all8 = false)
template <bool WRITE>
__host__ __device__ __forceinline__ size_t db_hybrid_layout(int UM, int CL, char *hot, char *cold, DbLds *L, int MW, size_t *hot_bytes)
{
    const int NB = (UM + 63) / 64;
    const int levels = db_levels(UM), nodes = (1 << levels) - 1, half = (1 << (levels - 1)) / 2 > 0 ? (1 << (levels - 1)) / 2 : 1;
    size_t oh = 0, oc = 0;
#define HOT(field, type, count)                            \
    if constexpr (WRITE) L->field = (type *)(hot + oh);    \
    oh = db_align16(oh + sizeof(type) * (size_t)(count));
#define COLD(field, type, count)                           \
    if constexpr (WRITE) L->field = (type *)(cold + oc);   \
    oc = db_align16(oc + sizeof(type) * (size_t)(count));
    HOT(X, double, UM)
    HOT(Y, double, UM)
    HOT(Z, double, UM)
    HOT(key, double, db_pow2ceil(UM))
    HOT(front, int, db_pow2ceil(UM))
    HOT(mm, unsigned long long, half * 16)
    HOT(sdim, int, half)
    HOT(lbase, int, half)
    HOT(misc, int, 16)
    COLD(idx, int, UM)
    COLD(idx2, int, UM)
    COLD(lab, int, UM)
    COLD(next, int, UM)
    COLD(core, unsigned char, UM)
    COLD(leafpos, unsigned char, UM)
    COLD(nstart, int, nodes + 1)
    COLD(nend, int, nodes + 1)
    COLD(nsum, double, nodes * 3)
    COLD(ncen, double, nodes * 3)
    COLD(nrad, unsigned long long, nodes + 1)
    COLD(blk, int, (NB + 1) > 64 ? (NB + 1) : 64)
    COLD(cnt, int, NB *(CL + 1))
    COLD(cl_n, int, CL + 2)
    COLD(cl_off, int, CL + 2)
    COLD(ccen, double, (CL + 1) * 6)
    COLD(fst, double, kFrontChunk * db_front_stride(MW))
    COLD(adj, unsigned long long, db_adj_words(UM))
    COLD(mask, unsigned long long, (size_t)UM * (MW > 1 ? MW : 1))
#undef HOT
#undef COLD
    if (hot_bytes) *hot_bytes = oh;
    return oc;
}

__device__ __forceinline__ int node_of(const DbLds &L, int p, int level)
{
    int node = 0;
    for (int t = 0; t < level; t++) {
        const int s = L.nstart[node], e = L.nend[node];
        const int mid = s + (e - s) / 2;
        node = 2 * node + 1 + (p >= mid ? 1 : 0);
    }
    return node;
}

// ---- bitonic network over the NT thread slots of a workgroup, 96-bit keys (hi64, lo32) --------------------
// Partner exchange for distance J: DPP inside quads (J = 1, 2) and inside rows of 16 (J = 4, 8: the two row
// shifts, picked by the lane's bit J), ds_bpermute for 16 and 32, LDS + barriers across waves.
template <int J>
__device__ __forceinline__ unsigned xor_lane32(unsigned v, int lane)
{
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    else if constexpr (J == 4 || J == 8) {
        const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + J, 0xF, 0xF, true);  // row_shl: lane i <- i + J
        const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + J, 0xF, 0xF, true);  // row_shr: lane i <- i - J
        return (lane & J) ? dn : up;
    } else return (unsigned)__shfl_xor((int)v, J);
}

template <int K, int J, int NT>
__device__ __forceinline__ void bitonic_round(unsigned long long &hi64, unsigned &lo32, int tid, unsigned long long *xh, unsigned *xl)
{
    unsigned long long ph;
    unsigned pl;
    if constexpr (J >= 64) {
        xh[tid] = hi64; xl[tid] = lo32;
        __syncthreads();
        ph = xh[tid ^ J]; pl = xl[tid ^ J];
        __syncthreads();
    } else {
        const int lane = tid & 63;
        const unsigned a = xor_lane32<J>((unsigned)hi64, lane), b = xor_lane32<J>((unsigned)(hi64 >> 32), lane);
        ph = ((unsigned long long)b << 32) | a;
        pl = xor_lane32<J>(lo32, lane);
    }
    const bool up = (tid & K) == 0, lower = (tid & J) == 0;
    const bool pless = ph < hi64 || (ph == hi64 && pl < lo32);  // partner sorts before me
    if ((lower == up) ? pless : !pless) { hi64 = ph; lo32 = pl; }  // lower slot keeps the smaller one when ascending
}
template <int K, int J, int NT>
struct BitonicJ {
    static __device__ __forceinline__ void run(unsigned long long &h, unsigned &l, int tid, unsigned long long *xh, unsigned *xl)
    {
        bitonic_round<K, J, NT>(h, l, tid, xh, xl);
        if constexpr (J > 1) BitonicJ<K, J / 2, NT>::run(h, l, tid, xh, xl);
    }
};
template <int K, int NT>
struct BitonicK {
    static __device__ __forceinline__ void run(unsigned long long &h, unsigned &l, int tid, unsigned long long *xh, unsigned *xl)
    {
        BitonicJ<K, K / 2, NT>::run(h, l, tid, xh, xl);
        if constexpr (K < NT) BitonicK<K * 2, NT>::run(h, l, tid, xh, xl);
    }
};

// The fp32 screen of one leaf for this lane's query (see dbscan_core, query_radius): n <= 60 candidates whose fp32
// coordinates lie at xf / yf / zf (uniform addresses: LDS broadcasts), two per packed instruction.  Returns the bits of the
// candidates whose fp32 metric is <= lo ("inside" for certain); `amb` = those in (lo, hi] -- for the fp64 formula.  Each
// comparison lands in its word through the carry: v_cmp -> vcc, then word = 2 word + vcc in one v_addc.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long leaf_screen(const float *xf, const float *yf, const float *zf, int n, float px, float py, float pz,
                                                          float c, float zw, float lo, float hi, unsigned long long &amb)
{
    const f32x2 PX = {px, px}, PY = {py, py}, PZ = {pz, pz}, NC = {-c, -c}, ONE = {1.0f, 1.0f}, ZW = {zw, zw};
    unsigned in0 = 0u, no0 = 0u, in1 = 0u, no1 = 0u;
    auto metric2 = [&](const f32x2 bx, const f32x2 by, const f32x2 bz) {
        const f32x2 w = __builtin_elementwise_fma(PY + by, NC, ONE);
        const f32x2 dx = PX - bx, dy = PY - by, dz = PZ - bz;
        f32x2 D = dx * dx;
        D = __builtin_elementwise_fma(dy, dy, D);
        D = __builtin_elementwise_fma(dz * ZW, dz, D);
        return w * D;
    };
    // four candidates a round: their twelve coordinates are requested together (past the leaf's end: stray values, masked below)
    auto quad = [&](int k, unsigned &inw, unsigned &now) {
        const f32x2 bx0 = {xf[k], xf[k + 1]}, by0 = {yf[k], yf[k + 1]}, bz0 = {zf[k], zf[k + 1]};
        const f32x2 bx1 = {xf[k + 2], xf[k + 3]}, by1 = {yf[k + 2], yf[k + 3]}, bz1 = {zf[k + 2], zf[k + 3]};
        const f32x2 d0 = metric2(bx0, by0, bz0), d1 = metric2(bx1, by1, bz1);
        asm volatile("v_cmp_ge_f32 vcc, %6, %2\n\t"
                     "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %7, %2\n\t"
                     "v_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %6, %3\n\t"
                     "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %7, %3\n\t"
                     "v_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %6, %4\n\t"
                     "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %7, %4\n\t"
                     "v_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %6, %5\n\t"
                     "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                     "v_cmp_ge_f32 vcc, %7, %5\n\t"
                     "v_addc_co_u32 %1, vcc, %1, %1, vcc"
                     : "+v"(inw), "+v"(now)
                     : "v"(d0.x), "v"(d0.y), "v"(d1.x), "v"(d1.y), "v"(lo), "v"(hi)
                     : "vcc");
    };
    n = __builtin_amdgcn_readfirstlane(n);  // (uniform by construction: scalar loop counters)
    const int n4 = (n + 3) & ~3, nA = n4 < 32 ? n4 : 32, nB = n4 - nA;
    for (int k = 0; k < nA; k += 4) quad(k, in0, no0);
    for (int k = 32; k < n4; k += 4) quad(k, in1, no1);
    // candidate k of a word that took m of them sits at bit m - 1 - k
    unsigned long long inb = nA > 0 ? (unsigned long long)(__brev(in0) >> (32 - nA)) : 0ULL;
    unsigned long long nob = nA > 0 ? (unsigned long long)(__brev(no0) >> (32 - nA)) : 0ULL;
    if (nB > 0) {
        inb |= (unsigned long long)(__brev(in1) >> (32 - nB)) << 32;
        nob |= (unsigned long long)(__brev(no1) >> (32 - nB)) << 32;
    }
    const unsigned long long valid = n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL);
    inb &= valid;
    amb = nob & ~inb & valid;
    return inb;
}

// The whole of DBSCAN.fit_predict for one cloud.  On return L.idx2[i] = label of
// point i (-1 noise) and the number of clusters is returned (uniform).
template <int NT, bool ALL8, int MW = 1>
__device__ __forceinline__ int dbscan_core(const DevCfg &cfg, const DbLds &L, const RowSrc src, int U, int UMc, double eps, int min_samples,
                                           unsigned long long *dbg, bool screened = false)
{
    DSTAMP_INIT
    (void)dbg;
    const int tid = threadIdx.x, lane = tid & 63;
    const double rw = cfg.db_range_weight, zw = cfg.db_z_weight;

    // ---- stage x,y,z in LDS (by point index); identity order.  ALL8 (U <= NT): thread i owns
    //      point i and keeps all 8 of its columns in registers for the whole tree build ----
    double f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0;
    int mypos = tid;  // ALL8: tree position of point `tid`
    if (ALL8) {
        if (tid < U) {
            const double2 *r2 = reinterpret_cast<const double2 *>(src.row(tid));
            const double2 a = r2[0], b = r2[1], c = r2[2], d = r2[3];
            f0 = a.x; f1 = a.y; f2 = b.x; f3 = b.y; f4 = c.x; f5 = c.y; f6 = d.x; f7 = d.y;
            L.X[tid] = f0; L.Y[tid] = f1; L.Z[tid] = f2;
            L.idx[tid] = tid;
            L.lab[tid] = -1;
            L.front[tid] = 0;
        }
    } else {
        for (int i = tid; i < U; i += NT) {
            const double *r = src.row(i);
            L.X[i] = r[0]; L.Y[i] = r[1]; L.Z[i] = r[2];
            L.idx[i] = i;
            L.lab[i] = -1;
        }
    }
    const int n_levels = db_levels(U);
    const int n_nodes = (1 << n_levels) - 1;
    if (tid == 0) { L.nstart[0] = 0; L.nend[0] = U; L.mm[0] = ~0ULL; L.mm[1] = 0ULL; L.misc[3] = 0; }
    __syncthreads();

    // ---- exact early exit: can ANY point reach min_samples tree-neighbours? ----
    // metric(a,b) = w_ab * E(a,b)^2 with E the weighted Euclidean norm sqrt(dx^2+dy^2+z_w*dz^2) (a true
    // norm for z_w >= 0) and w_ab = 1 - range_w*(ay+by)/2 >= wmin > 0 over the data's y range (node
    // centroids are means, so their y lies in that range too).  A BallTree neighbour q of p is either
    // leaf-tested, metric(p,q) <= eps => E(p,q)^2 <= eps/wmin, or taken with a whole node of centroid
    // c: metric(p,c) + radius <= eps with radius >= metric(c,q), so by the triangle inequality of E
    // E(p,q)^2 <= (sqrt(m(p,c)/wmin) + sqrt(m(c,q)/wmin))^2 <= 2*(m(p,c)+radius)/wmin <= 2*eps/wmin.
    // If no point has min_samples points (itself included) within E^2 <= 2*eps/wmin there is no core
    // point and every label is -1 -- exactly what sklearn returns -- and the tree is never built.
    // Steady-state rings of clutter end here.
    if (!screened && min_samples > 1 && zw >= 0.0 && eps >= 0.0) {
        double ylo = 1.7976931348623157e308, yhi = -1.7976931348623157e308;
        for (int i = tid; i < U; i += NT) {
            const double y = L.Y[i];
            ylo = y < ylo ? y : ylo;
            yhi = y > yhi ? y : yhi;
        }
        ylo = wave_min_d(ylo);
        yhi = wave_max_d(yhi);
        if (lane == 0 && ylo <= yhi) { atomicMin(&L.mm[0], sortable(ylo)); atomicMax(&L.mm[1], sortable(yhi)); }
        __syncthreads();
        const double ymin = unsortable(L.mm[0]), ymax = unsortable(L.mm[1]);
        const double wa = 1 - ymax * rw, wb = 1 - ymin * rw;
        const double wmin = wa < wb ? wa : wb;
        if (wmin > 0.0) {  // wave-uniform (same LDS values for every thread)
            const double R2 = 2.0 * (eps / wmin) * (1.0 + 1e-9);
            const int bparts = ALL8 ? (NT / U > 0 ? NT / U : 1) : 1;
            bool dense = false;
            if (!ALL8 || bparts == 1) {
                // the larger clouds: one dense point settles the question ("a core point is possible"), and a cloud
                // that holds a cluster has one within a few dozen candidates -- 64 at a time, then a look at the
                // flag the other threads may have raised
                for (int p = tid; p < U && !dense; p += NT) {
                    const double px = L.X[p], py = L.Y[p], pz = L.Z[p];
                    int c = 0;
                    for (int q0 = 0; q0 < U && !dense; q0 += 64) {
                        const int q1 = q0 + 64 < U ? q0 + 64 : U;
#pragma unroll 4
                        for (int q = q0; q < q1; q++) {
                            const double dx = px - L.X[q], dy = py - L.Y[q], dz = pz - L.Z[q];
                            c += ((dx * dx + dy * dy) + zw * (dz * dz) <= R2) ? 1 : 0;
                        }
                        if (c >= min_samples) { dense = true; L.misc[3] = 1; }
                        else if (L.misc[3] != 0) dense = true;
                    }
                }
            } else
            for (int t = tid; t < U * bparts; t += NT) {
                const int part = t / U, p = t - part * U;
                const double px = L.X[p], py = L.Y[p], pz = L.Z[p];
                int c = 0;
#pragma unroll 4
                for (int q = part; q < U; q += bparts) {
                    const double dx = px - L.X[q], dy = py - L.Y[q], dz = pz - L.Z[q];
                    c += ((dx * dx + dy * dy) + zw * (dz * dz) <= R2) ? 1 : 0;
                }
                if (bparts > 1) atomicAdd(&L.front[p], c);  // slices of one point add up in LDS
                else if (c >= min_samples) dense = true;
            }
            if (ALL8 && bparts > 1) {
                __syncthreads();
                if (tid < U && L.front[tid] >= min_samples) dense = true;
            }
            if (dense) L.misc[3] = 1;
            __syncthreads();
#ifdef MMW_STAMPS
            if (tid == 0 && dbg && L.misc[3] == 0) atomicAdd(&dbg[31], 1ULL);
#endif
            if (L.misc[3] == 0) {
                for (int i = tid; i < U; i += NT) L.idx2[i] = -1;
                __syncthreads();
                return 0;
            }
        }
    }

    auto feature = [&](int i, int f) -> double {
        if (f == 0) return L.X[i];
        if (f == 1) return L.Y[i];
        if (f == 2) return L.Z[i];
        return src.row(i)[f];
    };

    DSTAMP(0);
    // ---- _recursive_build, level by level (_binary_tree.pxi.tp:1040-1084) ----
    int *idx = L.idx, *idx2 = L.idx2;
    if (ALL8) {
        // Thread-per-point build: min/max by fire-and-forget LDS atomics, the median split by ONE bitonic sort
        // of the whole level, stable partition by ballots in lane (= point index) order.
        const int wave = tid >> 6;
        constexpr int MC = NT > 256 ? 4 : kMmCopies;  // == db_mm_copies(UMc): NT = 256 serves UMc <= 256, NT = 512 the larger class
        unsigned long long *xh = reinterpret_cast<unsigned long long *>(L.key);  // cross-wave exchange of the sort
        unsigned *xl = reinterpret_cast<unsigned *>(L.front);                     // (key[] / front[] are free here)
        unsigned char *leftflag = L.core;                                          // (free until the queries)
        int *posarr = L.next;   // point index -> tree position
        if (tid < U) posarr[tid] = tid;
        for (int level = 0; level + 1 < n_levels; level++) {
            const int first = (1 << level) - 1, nn = 1 << level;
            for (int e = tid; e < MC * nn * 16; e += NT) L.mm[e] = (e & 1) ? 0ULL : ~0ULL;
            __syncthreads();
            const bool act = tid < U;
            const int node = act ? node_of(L, mypos, level) : -1;
            if (act) {  // find_node_split_dim over all 8 features (_binary_tree.pxi.tp:598-645)
                unsigned long long *m = &L.mm[((lane & (MC - 1)) * nn + (node - first)) * 16];
                atomicMin(&m[0], sortable(f0)); atomicMax(&m[1], sortable(f0));
                atomicMin(&m[2], sortable(f1)); atomicMax(&m[3], sortable(f1));
                atomicMin(&m[4], sortable(f2)); atomicMax(&m[5], sortable(f2));
                atomicMin(&m[6], sortable(f3)); atomicMax(&m[7], sortable(f3));
                atomicMin(&m[8], sortable(f4)); atomicMax(&m[9], sortable(f4));
                atomicMin(&m[10], sortable(f5)); atomicMax(&m[11], sortable(f5));
                atomicMin(&m[12], sortable(f6)); atomicMax(&m[13], sortable(f6));
                atomicMin(&m[14], sortable(f7)); atomicMax(&m[15], sortable(f7));
            }
            __syncthreads();
            if (tid < nn * 16) {  // merge the private copies into copy 0
                unsigned long long v[MC];
#pragma unroll
                for (int q = 0; q < MC; q++) v[q] = L.mm[q * nn * 16 + tid];
                unsigned long long r = v[0];
#pragma unroll
                for (int q = 1; q < MC; q++) r = (tid & 1) ? (v[q] > r ? v[q] : r) : (v[q] < r ? v[q] : r);
                L.mm[tid] = r;
            }
            __syncthreads();
            DSTAMP(6);  // (diagnostic) min/max
            if (tid < nn) {
                double lo[8], hi[8];
#pragma unroll
                for (int f = 0; f < 8; f++) { lo[f] = unsortable(L.mm[(tid * 8 + f) * 2]); hi[f] = unsortable(L.mm[(tid * 8 + f) * 2 + 1]); }
                int jmax = 0;
                double best = 0;
#pragma unroll
                for (int f = 0; f < 8; f++) {
                    const double spread = hi[f] - lo[f];
                    if (spread > best) { best = spread; jmax = f; }
                }
                L.sdim[tid] = jmax;
            }
            __syncthreads();
            double kp = 0.0;
            int s = 0, e = 0;
            if (act) {
                const int sd = L.sdim[node - first];
                kp = sd == 0 ? f0 : sd == 1 ? f1 : sd == 2 ? f2 : sd == 3 ? f3 : sd == 4 ? f4 : sd == 5 ? f5 : sd == 6 ? f6 : f7;
                s = L.nstart[node];
                e = L.nend[node];
            }
            DSTAMP(7);  // (diagnostic) split dim + keys
            // partition_node_indices: the n_mid smallest under (value, index) go left
            // (_partition_nodes.pyx:35-39); both halves keep ascending point-index order.
            // Rank under (value, index) inside the node = position after sorting the whole level by
            // (node, value, index), minus the node's start (the nodes of a level are consecutive position
            // ranges in node order).  96-bit sort key: node(16) | order-preserving value bits(64) | point(16);
            // a bitonic network over the NT thread slots, cross-lane inside a wave, through LDS across waves.
            unsigned long long hi64 = ~0ULL;  // idle slots sort to the end
            unsigned lo32 = ~0u;
            if (act) {
                const unsigned long long sk = sortable(kp);
                hi64 = ((unsigned long long)node << 48) | (sk >> 16);
                lo32 = ((unsigned)(sk & 0xffffULL) << 16) | (unsigned)tid;
            }
            BitonicK<2, NT>::run(hi64, lo32, tid, xh, xl);
            if (hi64 != ~0ULL) {  // slot `tid` now holds the tid-th element of the level
                const int snode = (int)(hi64 >> 48), owner = (int)(lo32 & 0xffffu);
                const int ss = L.nstart[snode], ee = L.nend[snode];
                leftflag[owner] = (tid - ss) < (ee - ss) / 2 ? 1 : 0;
            }
            __syncthreads();
            const bool left = act && leftflag[tid] != 0;
            DSTAMP(8);  // (diagnostic) rank
            unsigned long long mine = 0;
            for (int nd = 0; nd < nn; nd++) {
                const unsigned long long b = __ballot(act && left && node == first + nd);
                if (node == first + nd) mine = b;
                if (lane == 0) L.blk[wave * nn + nd] = __popcll(b);
            }
            __syncthreads();
            int np = mypos;
            if (act) {
                int lc = __popcll(mine & lanemask_lt());  // lefts of my node with a smaller point index
                for (int w = 0; w < wave; w++) lc += L.blk[w * nn + (node - first)];
                const int nmid = (e - s) / 2;
                np = left ? s + lc : s + nmid + ((mypos - s) - lc);
                idx2[np] = tid;
                posarr[tid] = np;
            }
            if (tid < nn) {
                const int nd = first + tid, ss = L.nstart[nd], ee = L.nend[nd], nmid = (ee - ss) / 2;
                L.nstart[2 * nd + 1] = ss; L.nend[2 * nd + 1] = ss + nmid;
                L.nstart[2 * nd + 2] = ss + nmid; L.nend[2 * nd + 2] = ee;
            }
            DSTAMP(9);  // (diagnostic) partition
            mypos = np;
            { int *t = idx; idx = idx2; idx2 = t; }
            __syncthreads();
        }
    } else
    for (int level = 0; level + 1 < n_levels; level++) {
        const int first = (1 << level) - 1, nn = 1 << level;
        for (int e = tid; e < nn * 16; e += NT) L.mm[e] = (e & 1) ? 0ULL : ~0ULL;  // [node][f][0]=min key, [1]=max key
        __syncthreads();
        // find_node_split_dim over all 8 features (_binary_tree.pxi.tp:598-645)
        for (int p0 = 0; p0 < U; p0 += NT) {
            const int p = p0 + tid;
            const bool act = p < U;
            const int node = act ? node_of(L, p, level) : -1;
            const int nfirst = __builtin_amdgcn_readfirstlane(node);
            const bool uniform = __all(node == nfirst) != 0;  // wave-uniform
            const int i = act ? idx[p] : 0;
            const double *r = (!ALL8 && act) ? src.row(i) : nullptr;
#pragma unroll
            for (int f = 0; f < 8; f++) {
                double v = 0.0;
                if (act) v = ALL8 ? feature(i, f) : r[f];
                if (uniform) {
                    if (nfirst >= 0) {
                        const double mn = wave_min_d(v), mx = wave_max_d(v);
                        if (lane == 0) {
                            atomicMin(&L.mm[((nfirst - first) * 8 + f) * 2], sortable(mn));
                            atomicMax(&L.mm[((nfirst - first) * 8 + f) * 2 + 1], sortable(mx));
                        }
                    }
                } else if (act) {
                    atomicMin(&L.mm[((node - first) * 8 + f) * 2], sortable(v));
                    atomicMax(&L.mm[((node - first) * 8 + f) * 2 + 1], sortable(v));
                }
            }
        }
        __syncthreads();
        if (tid < nn) {
            int jmax = 0;
            double best = 0;
            for (int f = 0; f < 8; f++) {
                const double spread = unsortable(L.mm[(tid * 8 + f) * 2 + 1]) - unsortable(L.mm[(tid * 8 + f) * 2]);
                if (spread > best) { best = spread; jmax = f; }
            }
            L.sdim[tid] = jmax;
        }
        __syncthreads();
        DSTAMP(6);  // (diagnostic) min/max + split dim
        // partition_node_indices: the n_mid smallest under (value, index) go left
        // (_partition_nodes.pyx:35-39); both halves keep ascending point-index order.
        // Rank inside the node = slot after sorting the whole level by (node, value, point index), minus the
        // node's start (see the ALL8 branch).  Here the 96-bit keys live in LDS, one slot per position, and the
        // bitonic network runs over them: every thread owns pairs (i, i + j), so one barrier per round.
        {
            unsigned long long *xh = reinterpret_cast<unsigned long long *>(L.key);
            unsigned *xl = reinterpret_cast<unsigned *>(L.front);
            unsigned char *leftflag = L.core;  // by point index (free until the queries)
            const int Upad = db_pow2ceil(U);
            if (Upad <= NT && db_pow2ceil(UMc) >= NT) {  // (the exchange arrays hold pow2ceil(UMc) slots)
                // one slot per thread: the register network of the ALL8 build (cross-lane inside a wave, LDS only
                // for partner distances >= 64) -- 6 exchange rounds through LDS instead of 45 for 512 slots
                unsigned long long h = ~0ULL;  // padding sorts to the end
                unsigned l = ~0u;
                if (tid < U) {
                    const int node = node_of(L, tid, level), i = idx[tid];
                    const unsigned long long sk = sortable(feature(i, L.sdim[node - first]));
                    h = ((unsigned long long)node << 48) | (sk >> 16);
                    l = ((unsigned)(sk & 0xffffULL) << 16) | (unsigned)i;
                }
                BitonicK<2, NT>::run(h, l, tid, xh, xl);
                if (tid < U) {  // slot tid holds the tid-th element of the level
                    const int snode = (int)(h >> 48), owner = (int)(l & 0xffffu);
                    const int ss = L.nstart[snode], ee = L.nend[snode];
                    leftflag[owner] = (tid - ss) < (ee - ss) / 2 ? 1 : 0;
                }
                __syncthreads();
            } else {
            for (int p = tid; p < Upad; p += NT) {
                unsigned long long h = ~0ULL;  // padding sorts to the end
                unsigned l = ~0u;
                if (p < U) {
                    const int node = node_of(L, p, level), i = idx[p];
                    const unsigned long long sk = sortable(feature(i, L.sdim[node - first]));
                    h = ((unsigned long long)node << 48) | (sk >> 16);
                    l = ((unsigned)(sk & 0xffffULL) << 16) | (unsigned)i;
                }
                xh[p] = h; xl[p] = l;
            }
            __syncthreads();
            for (int k = 2; k <= Upad; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < Upad / 2; t += NT) {
                        const int i = 2 * j * (t / j) + (t % j), q = i + j;
                        const unsigned long long ah = xh[i], bh = xh[q];
                        const unsigned al = xl[i], bl = xl[q];
                        const bool b_first = bh < ah || (bh == ah && bl < al);  // element q sorts before element i
                        if (b_first == ((i & k) == 0)) { xh[i] = bh; xl[i] = bl; xh[q] = ah; xl[q] = al; }
                    }
                    __syncthreads();
                }
            for (int t = tid; t < U; t += NT) {  // slot t holds the t-th element of the level
                const int snode = (int)(xh[t] >> 48), owner = (int)(xl[t] & 0xffffu);
                const int ss = L.nstart[snode], ee = L.nend[snode];
                leftflag[owner] = (t - ss) < (ee - ss) / 2 ? 1 : 0;
            }
            __syncthreads();
            }
        }
        DSTAMP(8);  // (diagnostic) keys + rank
        const int NBLK = (U + 63) / 64;
        for (int p0 = 0; p0 < U; p0 += NT) {
            const int p = p0 + tid;
            const bool left = p < U && L.core[idx[p]] != 0;
            const unsigned long long b = __ballot(left);
            if (p0 + (tid & ~63) < U && lane == 0) L.blk[(p0 + tid) >> 6] = __popcll(b);
            // stash (rank of p among the lefts of its 64-block) | left flag in lab[] (restored to -1 below)
            if (p < U) L.lab[p] = __popcll(b & lanemask_lt()) | (left ? 0x40000000 : 0);
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int b = 0; b < NBLK; b++) { const int t = L.blk[b]; L.blk[b] = run; run += t; }
        }
        __syncthreads();
        for (int p = tid; p < U; p += NT) {  // scan value at node starts
            const int node = node_of(L, p, level);
            if (p == L.nstart[node]) L.lbase[node - first] = L.blk[p >> 6] + (L.lab[p] & 0x3fffffff);
        }
        __syncthreads();
        for (int p = tid; p < U; p += NT) {
            const int node = node_of(L, p, level);
            const int s = L.nstart[node], e = L.nend[node], nmid = (e - s) / 2;
            const int lb = L.blk[p >> 6] + (L.lab[p] & 0x3fffffff) - L.lbase[node - first];  // lefts in [s, p)
            const bool left = (L.lab[p] & 0x40000000) != 0;
            const int np = left ? s + lb : s + nmid + ((p - s) - lb);
            idx2[np] = idx[p];
        }
        __syncthreads();
        for (int p = tid; p < U; p += NT) L.lab[p] = -1;
        if (tid < nn) {
            const int node = first + tid, s = L.nstart[node], e = L.nend[node], nmid = (e - s) / 2;
            L.nstart[2 * node + 1] = s; L.nend[2 * node + 1] = s + nmid;
            L.nstart[2 * node + 2] = s + nmid; L.nend[2 * node + 2] = e;
        }
        { int *t = idx; idx = idx2; idx2 = t; }
        __syncthreads();
        DSTAMP(9);  // (diagnostic) partition
    }

    DSTAMP(1);
    // ---- init_node: centroids (leaf sums in ascending index order, parents = left + right)
    //      and radii (_ball_tree.pyx.tp:84-144) ----
    const int leaf0 = (1 << (n_levels - 1)) - 1, n_leaves = 1 << (n_levels - 1);
    for (int t = tid; t < n_leaves * 3; t += NT) {
        const int node = leaf0 + t / 3, c = t % 3;
        const double *col = c == 0 ? L.X : (c == 1 ? L.Y : L.Z);
        double acc = 0.0;
        int p = L.nstart[node];
        const int pe = L.nend[node];
        for (; p + 8 <= pe; p += 8) {  // (index loads, then value loads, then the adds in index order)
            int ii[8];
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) ii[u] = idx[p + u];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = col[ii[u]];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; p < pe; p++) acc += col[idx[p]];
        L.nsum[node * 3 + c] = acc;
    }
    for (int p = tid; p < U; p += NT) L.leafpos[p] = (unsigned char)(node_of(L, p, n_levels - 1) - leaf0);
    for (int e = tid; e < n_nodes; e += NT) L.nrad[e] = 0ULL;
    __syncthreads();
    for (int level = n_levels - 2; level >= 0; level--) {
        const int first = (1 << level) - 1, nn = 1 << level;
        for (int t = tid; t < nn * 3; t += NT) {
            const int node = first + t / 3, c = t % 3;
            L.nsum[node * 3 + c] = L.nsum[(2 * node + 1) * 3 + c] + L.nsum[(2 * node + 2) * 3 + c];
        }
        __syncthreads();
    }
    for (int t = tid; t < n_nodes * 3; t += NT) {
        const int node = t / 3;
        L.ncen[t] = L.nsum[t] / (double)(L.nend[node] - L.nstart[node]);
    }
    // From here on x,y,z are addressed by TREE POSITION (leaf ranges become contiguous reads):
    // permute the three columns in place (every by-index read above is complete: barrier in the loop).
    if (ALL8) {
        __syncthreads();
        if (tid < U) { L.X[mypos] = f0; L.Y[mypos] = f1; L.Z[mypos] = f2; }
    } else {
        for (int c = 0; c < 3; c++) {
            double *col = c == 0 ? L.X : (c == 1 ? L.Y : L.Z);
            __syncthreads();
            for (int p = tid; p < U; p += NT) L.key[p] = col[idx[p]];
            __syncthreads();
            for (int p = tid; p < U; p += NT) col[p] = L.key[p];
        }
    }
    // fp32 copies of the coordinates by tree position, for the screen in front of the leaf tests of query_radius (below): in
    // the three arrays that are dead between the build and the labelling -- the spare one of idx / idx2, next, lab (the
    // labelling's -1 are written again behind the queries) -- and the extents the screen's error bound needs
    float *XF = reinterpret_cast<float *>(idx2), *YF = reinterpret_cast<float *>(L.next), *ZF = reinterpret_cast<float *>(L.lab);
    if constexpr (MW > 1) {
        // (the clouds of more than 1920 points, db_hybrid_layout: those three are in global memory there, while the exchange slots of
        //  the level sort -- key[] and front[], in the LDS, dead after the build; the masks have their own array -- hold exactly
        //  three fp32 columns)
        XF = reinterpret_cast<float *>(L.key);
        YF = XF + UMc;
        ZF = reinterpret_cast<float *>(L.front);
    }
    if (tid == 0) { L.mm[0] = ~0ULL; L.mm[1] = 0ULL; L.mm[2] = 0ULL; }
    __syncthreads();
    double ext_lo = 1.7976931348623157e308, ext_hi = -1.7976931348623157e308, ext_m = 0.0;
    for (int p0 = 0; p0 < U; p0 += NT) {
        const int p = p0 + tid;
        const bool act = p < U;
        const int i = act ? p : 0;
        const double px = L.X[i], py = L.Y[i], pz = L.Z[i];
        if (act) {
            XF[p] = (float)px; YF[p] = (float)py; ZF[p] = (float)pz;
            ext_lo = py < ext_lo ? py : ext_lo;
            ext_hi = py > ext_hi ? py : ext_hi;
            const double ax = fabs(px), ay = fabs(py), az = fabs(pz);
            double am = ax > ay ? ax : ay;
            am = am > az ? am : az;
            ext_m = am > ext_m ? am : ext_m;   // (a NaN coordinate never enters: its comparisons are false in fp32 as in fp64)
        }
        int node = 0;
        for (int level = 0; level < n_levels; level++) {
            double d = act ? alt_dist(L.ncen[node * 3], L.ncen[node * 3 + 1], L.ncen[node * 3 + 2], px, py, pz, rw, zw) : 0.0;
            if (!(d > 0.0)) d = 0.0;
            const int nfirst = __builtin_amdgcn_readfirstlane(act ? node : -1);
            const bool uniform = __all((act ? node : -1) == nfirst) != 0;
            if (uniform) {
                const double mx = wave_max_d(d);
                if (lane == 0 && nfirst >= 0) atomicMax(&L.nrad[nfirst], (unsigned long long)__double_as_longlong(mx));
            } else if (act) {
                atomicMax(&L.nrad[node], (unsigned long long)__double_as_longlong(d));
            }
            if (level + 1 < n_levels) {
                const int s = L.nstart[node], e = L.nend[node];
                node = 2 * node + 1 + (p >= s + (e - s) / 2 ? 1 : 0);
            }
        }
    }
    ext_lo = wave_min_d(ext_lo);
    ext_hi = wave_max_d(ext_hi);
    ext_m = wave_max_d(ext_m);
    if (lane == 0 && ext_lo <= ext_hi) {
        atomicMin(&L.mm[0], sortable(ext_lo));
        atomicMax(&L.mm[1], sortable(ext_hi));
        atomicMax(&L.mm[2], (unsigned long long)__double_as_longlong(ext_m));
    }
    __syncthreads();

    DSTAMP(2);
    // ---- BallTree.query_radius(X, eps) for every point (_binary_tree.pxi.tp:1903-1980) ----
    const int lbits = n_levels - 1;
    // NearestNeighbors._fit with algorithm="auto" (sklearn/neighbors/_base.py:622-633): DBSCAN leaves n_neighbors at its default
    // of 5, and `n_neighbors >= n_samples // 2` answers clouds of 1 .. 11 points by BRUTE FORCE -- the exact pairwise metric
    // `<= eps` (_base.py:1054-1081, 1221-1250), no tree: the root (the only node of so small a cloud) is a TEST leaf for every
    // query, never PRUNE, never taken whole.  (Reachable with DB_MIN_SAMPLES_MIN <= 11; the no-core-point screens bound a
    // superset of either neighbourhood.)
    const bool brute = (U >> 1) <= kSkNeighbors;  // uniform
    // The leaf tests -- "is metric(p, q) <= eps" for every point q of a leaf some query of the wave reached: nine tenths of
    // this phase, 14 fp64 operations each -- go through an fp32 SCREEN first: the same formula on the fp32 copies, two
    // candidates per packed instruction, decides every pair whose fp32 value is further than E from eps; the few in between
    // are computed in fp64 as before.  E bounds |metric_fp32 - metric_fp64| for all pairs whose coordinate differences are
    // within R, R^2 = 4 max(eps, 1) / (wmin min(1, z_w)) (beyond R the metric is >= 4 max(eps, 1) and its fp32 value within
    // 15 % of it: "out" either way), from |coordinates| <= M and y in [ymin, ymax] (u = 2^-23, twice the unit roundoff; a
    // difference is off by <= 2Mu + 2u|d|, its square by <= 4RMu + 6uR^2, the weight by <= 12cMu + 4u, c = |range_w| / 2),
    // doubled.  Clouds whose extents make E useless (or the far-pair argument void) skip the screen: the decisions -- and with
    // them counts, rows, labels -- are those of the fp64 formula in every case.
    float scr_lo = 0.f, scr_hi = 0.f;
    bool use_scr = false;
    {
        const double ymin = unsortable(L.mm[0]), ymax = unsortable(L.mm[1]), M = __longlong_as_double((long long)L.mm[2]);
        const double wa = 1 - ymax * rw, wb = 1 - ymin * rw, wmin = wa < wb ? wa : wb;
        const double u = 1.0 / 8388608.0, c = 0.5 * fabs(rw), mz = zw < 1.0 ? zw : 1.0, e1 = eps > 1.0 ? eps : 1.0;
        if (wmin > 0.0 && zw >= 1.0 / 1024.0 && eps > 0.0 && L.mm[0] != ~0ULL) {
            const double R2 = 4.0 * e1 / (wmin * mz), R = sqrt(R2), Wm = 1.0 + 2.0 * c * M, Dm = (2.0 + zw) * R2;
            const double dD = (2.0 + zw) * (4.0 * R * M * u + 6.0 * u * R2) + 4.0 * u * Dm, dw = 12.0 * c * M * u + 4.0 * u;
            const double E = 2.0 * (Wm * dD + Dm * dw + 2.0 * u * Wm * Dm) + eps * (1.0 / 4194304.0);
            if (M <= 65536.0 * R && dw <= 0.1 * wmin && E < 0.25 * (eps < 1.0 ? eps : 1.0)) {
                use_scr = true;
#ifdef MMW_MUTANT_NO_MARGIN   // (mutation check of tests/test_gpu_parity.py::test_dbscan_pairs_at_the_threshold_vs_oracle: never the product)
                scr_lo = scr_hi = (float)eps;
#else
                scr_lo = (float)(eps - E);
                scr_hi = (float)(eps + E);
#endif
            }
        }
    }
    // A WAVE walks the tree as one: its 64 queries are neighbours in the tree (one or two leaves), so the nodes any of them
    // needs are nearly the nodes each of them needs -- and with node and level uniform every branch below is taken by
    // the whole wave, the candidates of a leaf are read once (one LDS broadcast per coordinate) and the distance block runs
    // on full lanes.  (One thread walking alone per query left the SIMDs ~25 % busy: every lane at its own node.)  Per lane:
    // `alive` bit l = "the walk reached this level's node through DESCEND states of mine".  The per-lane visit order is the
    // order of the private walk, so masks, counts and rows are the same.
    // Spare waves (thread-per-point build, U <= NT / 2) share the queries: slice `qpart` of `qparts` takes every qparts-th
    // candidate of a TEST leaf; counts meet in an LDS counter.
    const int Wq = (U + 63) >> 6;  // waves that hold one query each per lane
    const int qparts = ALL8 ? ((NT >> 6) / Wq > 0 ? (NT >> 6) / Wq : 1) : 1;
    int *qcount = L.front;
    // Clouds of <= kAdjMax points also record WHICH points are within eps: row p of L.adj, one bit per tree position, set
    // exactly where the labelling below would find "q in query_radius(p)" (a node taken whole: its range; a tested leaf:
    // the points that passed).  dbscan_inner then never touches a coordinate again.
    const bool use_adj = U <= db_adj_cap(UMc);  // uniform
    const int W = (U + 63) >> 6, WS = W | 1;
    unsigned long long *adj = L.adj;
    if (use_adj)
        for (int e = tid; e < U * WS + 32; e += NT) adj[e] = 0ULL;
    if (ALL8 && qparts > 1)
        for (int p = tid; p < U; p += NT) qcount[p] = 0;
    if (use_adj || (ALL8 && qparts > 1)) __syncthreads();
    {
        const int wv = tid >> 6;
        const int qpart = __builtin_amdgcn_readfirstlane(ALL8 ? wv / Wq : 0);  // (uniform per wave)
        const int pbase = ALL8 ? (wv - qpart * Wq) * 64 : wv * 64;
        const int pstep = ALL8 ? U : NT;                           // (thread-per-point build: one batch)
        for (int pb = pbase; pb < U && qpart < qparts; pb += pstep) {
            const int p = pb + lane;
            const bool act = p < U;
            const double px = L.X[act ? p : 0], py = L.Y[act ? p : 0], pz = L.Z[act ? p : 0];
            const float pxf = (float)px, pyf = (float)py, pzf = (float)pz, cf = (float)(0.5 * rw), zwf = (float)zw;
            unsigned long long m[MW];
#pragma unroll
            for (int w = 0; w < MW; w++) m[w] = 0ULL;
            int count = 0, node = 0, level = 0;  // node, level: uniform
            unsigned alive = 1u;
            for (;;) {
                int state = 0;  // 0 prune (or not mine), 1 all, 2 leaf test, 3 descend
                if (act && ((alive >> level) & 1u)) {
                    const double d = alt_dist(px, py, pz, L.ncen[node * 3], L.ncen[node * 3 + 1], L.ncen[node * 3 + 2], rw, zw);
                    const double rad = __longlong_as_double((long long)L.nrad[node]);
                    const double t = d - rad;
                    const double lb = t > 0 ? t : 0, ub = d + rad;
                    if (brute) state = 2;   // (one node: level == lbits == 0)
                    else if (lb > eps) state = 0;
                    else if (ub <= eps) state = 1;
                    else if (level == lbits) state = 2;
                    else state = 3;
                }
                // (the same LDS words for every lane: scalar from here on, the loops below are uniform)
                const int s = __builtin_amdgcn_readfirstlane(L.nstart[node]), e = __builtin_amdgcn_readfirstlane(L.nend[node]);
                if (state == 1 || state == 2) {
                    const int span = 1 << (lbits - level);
                    const int fl = (node + 1 - (1 << level)) * span;
                    const unsigned long long pat = state == 1 ? 0x5555555555555555ULL : 0xAAAAAAAAAAAAAAAAULL;
                    if constexpr (MW == 1) {
                        const unsigned long long sel = span == 32 ? ~0ULL : ((1ULL << (2 * span)) - 1ULL);
                        m[0] |= (pat & sel) << (2 * fl);
                    } else {
#pragma unroll
                        for (int w = 0; w < MW; w++) {  // bits [2 fl, 2 fl + 2 span) of the mask, word by word
                            const int lo = 2 * fl - 64 * w, hi = lo + 2 * span;
                            if (hi > 0 && lo < 64) {
                                const int a = lo < 0 ? 0 : lo, b = hi > 64 ? 64 : hi;
                                const unsigned long long sel = b - a == 64 ? ~0ULL : (((1ULL << (b - a)) - 1ULL) << a);
                                m[w] |= pat & sel;
                            }
                        }
                    }
                }
                if (state == 1 && qpart == 0) {
                    count += e - s;
                    if (use_adj)
                        for (int w = s >> 6; w <= (e - 1) >> 6; w++) {
                            const int lo = (s > w * 64 ? s : w * 64) - w * 64, hi = (e < w * 64 + 64 ? e : w * 64 + 64) - w * 64;
                            const unsigned long long bits = (hi == 64 ? ~0ULL : ((1ULL << hi) - 1ULL)) & ~((1ULL << lo) - 1ULL);
                            atomicOr(&adj[p * WS + w], bits);
                        }
                }
                if (level == lbits && __any(state == 2)) {
                    // (a leaf holds at most 2 * leaf_size = 60 points: one word of bits relative to its start, two row words)
                    // (spare waves: slice `qpart` of the leaf; everything here is uniform but the lane's own state and bits)
                    const int n = e - s, per = (n + qparts - 1) / qparts;
                    const int a = __builtin_amdgcn_readfirstlane(qpart * per < n ? qpart * per : n), b = a + per < n ? a + per : n;
                    unsigned long long bits = 0ULL, amb_all = b - a >= 64 ? ~0ULL : ((1ULL << (b - a)) - 1ULL);
                    if (use_scr && b > a) bits = leaf_screen(XF + s + a, YF + s + a, ZF + s + a, b - a, pxf, pyf, pzf, cf, zwf, scr_lo, scr_hi, amb_all);
                    unsigned long long amb = state == 2 ? amb_all : 0ULL;  // (the lanes that do not test this leaf drop their bits below)
                    while (amb) {  // what the screen left open (without it: every candidate): the fp64 formula
                        const int k = __ffsll((long long)amb) - 1;
                        amb &= amb - 1ULL;
                        const int q = s + a + k;
                        if (alt_dist(px, py, pz, L.X[q], L.Y[q], L.Z[q], rw, zw) <= eps) bits |= 1ULL << k;
                    }
                    bits <<= a;
                    if (state == 2) {
                        count += __popcll(bits);
                        if (use_adj) {
                            const int w0 = s >> 6, lo = s & 63;
                            const unsigned long long b0 = bits << lo, b1 = lo ? bits >> (64 - lo) : 0ULL;
                            if (b0) atomicOr(&adj[p * WS + w0], b0);
                            if (b1) atomicOr(&adj[p * WS + w0 + 1], b1);
                        }
                    }
                }
                if (__any(state == 3)) {
                    alive = (alive & ~(2u << level)) | (state == 3 ? 2u << level : 0u);
                    node = 2 * node + 1;
                    level++;
                    continue;
                }
                while (node != 0 && (node & 1) == 0) { node = (node - 1) >> 1; level--; }  // climb while right child
                if (node == 0) break;
                node++;  // left child -> its sibling
            }
            if (act) {
                // the key[] buffer is dead after the build: it now holds the masks
                if (qpart == 0) {
#pragma unroll
                    for (int w = 0; w < MW; w++) L.mask[(size_t)p * MW + w] = m[w];
                }
                if (qparts > 1) atomicAdd(&qcount[p], count);
                else L.core[p] = count >= min_samples ? 1 : 0;
            }
        }
    }
    __syncthreads();
    for (int p = tid; p < U; p += NT) L.lab[p] = -1;  // (held the fp32 z column during the queries)
    if (qparts > 1)
        for (int p = tid; p < U; p += NT) L.core[p] = qcount[p] >= min_samples ? 1 : 0;
    __syncthreads();

    DSTAMP(3);
    // ---- dbscan_inner (sklearn/cluster/_dbscan_inner.pyx): clusters seeded in ascending
    //      point index; frontier expansion instead of the DFS stack (same labels) ----
    int n_clusters = 0;
    int *front = L.front, *next = L.next;
    if (use_adj) {
        // Bit-set form.  The rows are transposed first (row q then says WHO has q in its neighbourhood: the distance itself
        // is symmetric, bit for bit, but the rows are not -- the tree takes whole nodes by a bound that is no true triangle
        // inequality for this weighted distance -- so a cluster is what its seed reaches along rows of cores, in seed
        // order, exactly dbscan_inner's; a union-find would not do).  One round of the expansion is then,
        // per unlabelled point, a few ANDs of its row with the frontier set F and a ballot: no atomics, one barrier.
        unsigned long long *F0 = adj + U * WS, *F1 = F0 + 8;
        {
            const int wave = tid >> 6, nw = NT >> 6;
            int pair = 0;
            for (int a = 0; a < W; a++)
                for (int b = a; b < W; b++, pair++) {
                    if (pair % nw != wave) continue;  // (uniform per wave)
                    const int ra = a * 64 + lane, rb = b * 64 + lane;
                    unsigned long long x = ra < U ? adj[ra * WS + b] : 0ULL;             // tile (a, b)
                    unsigned long long y = (a != b && rb < U) ? adj[rb * WS + a] : 0ULL;  // tile (b, a)
                    x = transpose64(x, lane);
                    if (a != b) y = transpose64(y, lane);
                    if (a == b) { if (ra < U) adj[ra * WS + a] = x; }
                    else {
                        if (rb < U) adj[rb * WS + a] = x;
                        if (ra < U) adj[ra * WS + b] = y;
                    }
                }
        }
        __syncthreads();
        for (;;) {
            if (tid == 0) L.misc[0] = 0x7fffffff;
            __syncthreads();
            {
                int best = 0x7fffffff;
                for (int p = tid; p < U; p += NT)
                    if (L.core[p] && L.lab[p] < 0) { const int k = idx[p] * 4096 + p; best = k < best ? k : best; }
                for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(best, o); best = t < best ? t : best; }
                if (lane == 0 && best != 0x7fffffff) atomicMin(&L.misc[0], best);
            }
            __syncthreads();
            const int seedkey = L.misc[0];
            if (seedkey == 0x7fffffff) break;
            __syncthreads();
            const int sp = seedkey & 4095;
            if (tid < 8) F0[tid] = (sp >> 6) == tid ? 1ULL << (sp & 63) : 0ULL;
            if (tid == 0) L.lab[sp] = n_clusters;
            __syncthreads();
            unsigned long long *Fc = F0, *Fn = F1;
            for (;;) {
                for (int p0 = 0; p0 < U; p0 += NT) {
                    const int p = p0 + tid;
                    bool hit = false;
                    if (p < U && L.lab[p] < 0) {
                        unsigned long long acc = 0ULL;
                        for (int w = 0; w < W; w++) acc |= adj[p * WS + w] & Fc[w];
                        hit = acc != 0ULL;
                    }
                    if (hit) L.lab[p] = n_clusters;
                    const unsigned long long hb = __ballot(hit && L.core[p]);
                    if (lane == 0 && p < U) Fn[p >> 6] = hb;  // (whole words: a wave's points share one)
                }
                __syncthreads();
                unsigned long long any = 0ULL;
                for (int w = 0; w < W; w++) any |= Fn[w];  // (uniform: the same LDS words for every thread)
                if (!any) break;
                { unsigned long long *t = Fc; Fc = Fn; Fn = t; }
            }
            n_clusters++;
        }
    } else
    for (;;) {
        if (tid == 0) L.misc[0] = 0x7fffffff;
        __syncthreads();
        {
            int best = 0x7fffffff;
            for (int p = tid; p < U; p += NT)
                if (L.core[p] && L.lab[p] < 0) { const int k = idx[p] * 4096 + p; best = k < best ? k : best; }
            for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(best, o); best = t < best ? t : best; }
            if (lane == 0 && best != 0x7fffffff) atomicMin(&L.misc[0], best);
        }
        __syncthreads();
        const int seedkey = L.misc[0];
        if (seedkey == 0x7fffffff) break;
        __syncthreads();
        if (tid == 0) { const int sp = seedkey & 4095; L.lab[sp] = n_clusters; front[0] = sp; L.misc[2] = 0; }
        __syncthreads();
        int fcount = 1;
        while (fcount > 0) {
            // The frontier goes through a contiguous staging array, kFrontChunk points at a time: every thread
            // then reads the same addresses (LDS broadcast) that depend on nothing it loaded before, so one
            // round trip brings four entries instead of a position load followed by four dependent gathers.
            for (int f0 = 0; f0 < fcount; f0 += kFrontChunk) {
                const int fc = fcount - f0 < kFrontChunk ? fcount - f0 : kFrontChunk;
                if (f0 > 0) __syncthreads();  // the previous chunk has been consumed
                constexpr int FS = MW == 1 ? 4 : 4 + MW;  // (db_front_stride)
                if (tid < fc) {
                    const int pp = front[f0 + tid];
                    L.fst[tid * FS + 0] = __longlong_as_double((long long)L.mask[(size_t)pp * MW]);
                    L.fst[tid * FS + 1] = L.X[pp];
                    L.fst[tid * FS + 2] = L.Y[pp];
                    L.fst[tid * FS + 3] = L.Z[pp];
#pragma unroll
                    for (int w = 1; w < MW; w++) L.fst[tid * FS + 3 + w] = __longlong_as_double((long long)L.mask[(size_t)pp * MW + w]);
                }
                __syncthreads();
                for (int q = tid; q < U; q += NT) {
                    if (L.lab[q] >= 0) continue;
                    const int lq = L.leafpos[q];
                    const double qx = L.X[q], qy = L.Y[q], qz = L.Z[q];
                    bool hit = false;
                    for (int f = 0; f < fc && !hit; f += 4) {
                        double4 en[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const double2 *e2 = reinterpret_cast<const double2 *>(L.fst + (f + u < fc ? f + u : f) * FS);
                            const double2 a = e2[0], b = e2[1];
                            en[u] = make_double4(a.x, a.y, b.x, b.y);
                            if constexpr (MW > 1) {  // the mask word this point's leaf lies in
                                if ((lq >> 5) != 0) en[u].x = L.fst[(f + u < fc ? f + u : f) * FS + 3 + (lq >> 5)];
                            }
                        }
                        // leaf states first: most (point, frontier point) pairs are PRUNE, and a wave's 64 points
                        // sit in one or two leaves, so the distance block below is skipped by whole waves
                        int stt[4];
                        bool test = false;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            stt[u] = (int)(((unsigned long long)__double_as_longlong(en[u].x) >> (2 * (lq & 31))) & 3ULL);
                            hit = hit || stt[u] == 1;
                            test = test || stt[u] == 2;
                        }
                        if (test && !hit) {
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (stt[u] == 2 && alt_dist(en[u].y, en[u].z, en[u].w, qx, qy, qz, rw, zw) <= eps) hit = true;
                        }
                    }
                    if (hit) {
                        L.lab[q] = n_clusters;
                        if (L.core[q]) next[atomicAdd(&L.misc[2], 1)] = q;
                    }
                }
            }
            __syncthreads();
            fcount = L.misc[2];
            __syncthreads();
            if (tid == 0) L.misc[2] = 0;
            { int *t = front; front = next; next = t; }
            __syncthreads();
        }
        n_clusters++;
    }
    DSTAMP(4);
    // labels by point index: scatter into whichever of the two index buffers is free,
    // the caller always finds them in L.idx2
    int *labi = (idx == L.idx) ? L.idx2 : L.idx;
    for (int p = tid; p < U; p += NT) labi[idx[p]] = L.lab[p];
    __syncthreads();
    if (labi != L.idx2) {
        for (int i = tid; i < U; i += NT) L.idx2[i] = labi[i];
        __syncthreads();
    }
    return n_clusters;
}

// Tracking.py:697-703 for the scenes of one size class: apply_DBscan on the global ring,
// batch.clear(), _add_tracks.
__device__ __forceinline__ void finish_scene_stats(const DevState &st, int s, int U, int ncl)
{
    if (st.stats) {  // algorithmic bytes: ring rows in, labels out, new track records + ring rows out
        unsigned long long *sl = stats_slot(st, s);
        atomicAdd(&sl[1], (unsigned long long)(64 * U + 4 * U) + (unsigned long long)ncl * (sizeof(TrackRec) + 64ULL * 64ULL));
        atomicAdd(&sl[3], 1ULL);
        atomicAdd(&sl[4], (unsigned long long)U);
        atomicAdd(&sl[7], (unsigned long long)ncl);
    }
}

// _add_tracks (Tracking.py:576-589) for `nspawn` clusters of the cloud dbscan_core has just labelled: cluster c is
// the set of points with label lab0 + c (rows keep input order, Utils.py:285-287); its track is appended at list
// position T0 + c.  All threads of the workgroup call; L.misc[4] holds the first creation ordinal.
template <int NT>
__device__ __forceinline__ void add_clusters(const DevCfg &cfg, const DevState &st, const DbLds &L, const RowSrc src, int s, int U, int CL,
                                             int lab0, int nspawn, int T0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int *labi = L.idx2;
    // members of every cluster in ascending point index
    const int NB = (U + 63) / 64, CLS = CL + 1;
    for (int i0 = 0; i0 < U; i0 += NT) {
        const int i = i0 + tid;
        const int cls_i = (i < U) ? labi[i] - lab0 : -1;
        const int b = i >> 6;
        if (i0 + (tid & ~63) < U) {
            unsigned long long mine = 0;
            for (int c = 0; c < nspawn; c++) {
                const unsigned long long bal = __ballot(cls_i == c);
                if (cls_i == c) mine = bal;
                if (lane == 0) L.cnt[b * CLS + c] = __popcll(bal);
            }
            if (i < U) L.lab[i] = __popcll(mine & lanemask_lt());  // rank inside its 64-block
        }
    }
    __syncthreads();
    for (int c = tid; c < nspawn; c += NT) {
        int run = 0;
        for (int b = 0; b < NB; b++) { const int t = L.cnt[b * CLS + c]; L.cnt[b * CLS + c] = run; run += t; }
        L.cl_n[c] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < nspawn; c++) { L.cl_off[c] = run; run += L.cl_n[c]; }
        L.cl_off[nspawn] = run;
    }
    __syncthreads();
    int *memb = L.front;  // free after labelling
    for (int i = tid; i < U; i += NT) {
        const int c = labi[i] - lab0;
        if (c >= 0 && c < nspawn) memb[L.cl_off[c] + L.cnt[(i >> 6) * CLS + c] + L.lab[i]] = i;
    }
    __syncthreads();
    int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    // PointCluster stats (Tracking.py:120-136): sequential mean in row order
    for (int t = tid; t < nspawn * 6; t += NT) {
        const int c = t / 6, m = t % 6;
        const int n = L.cl_n[c], off = L.cl_off[c];
        TrackRec *rec = trk + order[T0 + c];
        // (rows come from the ring -- the LDS x,y,z are in tree-position order by now --: eight loads in
        //  flight per round trip, the sum itself stays sequential in row order)
        double sum = 0.0, mn = 0.0, mx = 0.0;
        int r = 0;
        for (; r + 8 <= n; r += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src.row(memb[off + r + u])[m];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                sum += v[u];
                mn = (r + u == 0 || v[u] < mn) ? v[u] : mn;
                mx = (r + u == 0 || v[u] > mx) ? v[u] : mx;
            }
        }
        for (; r < n; r++) {
            const double v = src.row(memb[off + r])[m];
            sum += v;
            mn = (r == 0 || v < mn) ? v : mn;
            mx = (r == 0 || v > mx) ? v : mx;
        }
        const double cen = sum / (double)n;
        L.ccen[c * 6 + m] = cen;
        rec->centroid[m] = cen;
        rec->minv[m] = mn;
        rec->maxv[m] = mx;
    }
    __syncthreads();
    // ClusterTrack.__init__ / KalmanState.__init__ (Tracking.py:87-97, 210-230)
    for (int c = 0; c < nspawn; c++) {
        const int slot = order[T0 + c];
        TrackRec *rec = trk + slot;
        const int n = L.cl_n[c], off = L.cl_off[c];
        for (int e = tid; e < 81; e += NT) {
            const int i = e / 9, k = e % 9;
            rec->P[e] = (i == k && i < cfg.dx) ? 1.0 * cfg.kf_p_init : 0.0;
        }
        for (int e = tid; e < 36; e += NT) rec->gd[e] = (e / 6 == e % 6) ? 1.0 * cfg.kf_group_disp_est_init : 0.0;
        for (int e = tid; e < 9; e += NT) rec->x[e] = e < 6 ? L.ccen[c * 6 + e] : 0.0;
        for (int e = tid; e < 6; e += NT) rec->spread[e] = 0.0;
        for (int e = tid; e < MMW_NKP; e += NT) rec->kp[e] = st.default_posture[e];
        if (tid == 0) {
            const double v3 = L.ccen[c * 6 + 3], v4 = L.ccen[c * 6 + 4], v5 = L.ccen[c * 6 + 5];
            rec->is_static = sqrt((v3 * v3 + v4 * v4) + v5 * v5) < cfg.tr_vel_thres ? 1 : 0;
            rec->point_num = n;
            rec->n_est = 0.0;
            rec->lifetime = 0.0;
            rec->ring_len = 1;
            rec->uid = L.misc[4] + c;
            rec->inner = cfg.ring;  // a fresh BatchedData: size FB_FRAMES_BATCH + 1, associate_pointcloud has not run on it
            for (int k = 0; k < MMW_RING_MAX; k++) { rec->ring_slot[k] = k; rec->ring_n[k] = 0; }
            rec->ring_n[0] = n;
        }
        const int keep = min(n, cfg.ring_rows);
        double *dst = st.trk_ring + (((size_t)s * cfg.t_cap + slot) * cfg.ring + 0) * (size_t)cfg.ring_rows * 8;
        for (int e = tid; e < keep * 8; e += NT) dst[e] = src.row(memb[off + (e >> 3)])[e & 7];
    }
}

template <int NT, bool ALL8, int MW = 1>
__device__ __forceinline__ void spawn_scene(const DevCfg &cfg, const DevState &st, const DbLds &L, int s, int UMc, int CL, int UM_out,
                                            bool screened, int parity, int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    const int tid = threadIdx.x;
    SceneHdr *hdr = st.hdr + s;
    const int U = hdr->db_u;
    const RowSrc src = ring_rows_of(cfg, st, hdr, s);
    __syncthreads();  // every thread has read the header before anyone rewrites it below
    const int ncl = dbscan_core<NT, ALL8, MW>(cfg, L, src, U, UMc, cfg.db_eps, cfg.db_min_samples, stats_slot(st, s), screened);
    const int *labi = L.idx2;
    if (labels_out)
        for (int i = tid; i < U; i += NT) labels_out[(size_t)s * UM_out + i] = labi[i];
    if (tid == 0) {
        if (db_n_out) db_n_out[s] = U;
        hdr->need_db = 0;
        finish_scene_stats(st, s, U, ncl);
    }
    if (ncl == 0) return;

    // ---- batch.clear() (Tracking.py:53-58) + _add_tracks (Tracking.py:576-589) ----
    const int T0 = hdr->n_tracks;
    int nspawn = ncl;
    int err = 0;
    if (T0 + ncl > cfg.t_cap || ncl > CL) { nspawn = min(max(cfg.t_cap - T0, 0), CL); err |= ERR_CAPACITY; }
    __syncthreads();
    if (tid == 0) {
        hdr->g_len = 0;
        for (int k = 0; k < MMW_RING_MAX; k++) hdr->g_n[k] = 0;
        hdr->n_tracks = T0 + nspawn;
        if (nspawn > 0) {  // the next k_predict takes the new tracks T0.. from here (the older ones from the update lists)
            const int pos = atomicAdd(&st.spc_count[parity], 1);
            st.spc_list[((size_t)parity * cfg.n_scenes + pos) * 2] = s;
            st.spc_list[((size_t)parity * cfg.n_scenes + pos) * 2 + 1] = T0;
        }
        L.misc[4] = hdr->next_uid;
        hdr->next_uid += nspawn;
        if (err) atomicOr(&hdr->err, err);
    }
    add_clusters<NT>(cfg, st, L, src, s, U, CL, 0, nspawn, T0);
}

// ClusterTrack.seek_inner_clusters (Tracking.py:409-448) with its call site (Tracking.py:656) active, cfg.seek_inner:
// one launch between k_track and k_post, one workgroup per scene.  For every track whose associate_pointcloud ran
// this frame (k_track marks them), in list order:
//   * `cluster.point_num > DB_POINTS_THRES and spread.any() > DB_SPREAD_THRES` -- the second term compares a BOOL
//     (x-spread != 0) with the threshold, as the reference does;
//   * change_buffer_size(FB_FRAMES_BATCH_STATIC | FB_FRAMES_BATCH) -- permanent -- and add_frame(cluster.pointcloud): the
//     cloud k_track has just appended is appended a second time;
//   * apply_DBscan(effective_data, eps = DB_INNER_EPS) with the default min_samples; more than one cluster ->
//     _add_tracks([clusters[1]]) (Tracking.py:658-660): the track is appended BEFORE _update_all (k_post) and before
//     the frame's own DBSCAN trigger, which is re-evaluated here (`len(effective_tracks) < TR_MAX_TRACKS`).
// The per-scene layout of the Kalman kernels is forced for such contexts (hdr->n_upd is all k_post needs).
constexpr int kInnerThreads = 512;
__global__ __launch_bounds__(kInnerThreads) void k_inner(DevCfg cfg, DevState st, const int32_t *__restrict__ n_pts, int UMc,
                                                         int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int n = n_pts[s];
    int32_t *ib = st.inner_buf + (size_t)s * (kInnerHdr + st.inner_cap);
    if (tid < kInnerHdr) ib[tid] = 0;
    if (!frame_reaches_track(n, cfg.max_pts)) return;  // the frame never reached track()
    SceneHdr *hdr = st.hdr + s;
    int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    const int Tin = hdr->n_tracks;  // after _maintain_tracks: tracks with points are never removed and keep their order
    int T = Tin, calls = 0, stored = 0, err = 0;
    DbLds L;
    __syncthreads();
    for (int j = 0; j < Tin; j++) {
        const int slot = order[j];
        TrackRec *rec = trk + slot;
        const int inner = rec->inner;
        if (!(inner & kInnerTouched)) continue;  // uniform
        const double xspread = rec->maxv[0] - rec->minv[0];
        const double any = xspread != 0.0 ? 1.0 : 0.0;
        const bool go = rec->point_num > cfg.db_points_thres && any > cfg.db_spread_thres;
        int len = rec->ring_len, rn[MMW_RING_MAX], rs[MMW_RING_MAX];
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { rn[k] = rec->ring_n[k]; rs[k] = rec->ring_slot[k]; }
        const bool is_static = rec->is_static != 0;
        __syncthreads();  // everyone has read the record
        if (!go) {
            if (tid == 0) rec->inner = inner & 255;
            continue;
        }
        // ---- change_buffer_size + add_frame(cluster.pointcloud) ----
        const int size = is_static ? cfg.fb_frames_batch_static : cfg.ring - 1;  // FB_FRAMES_BATCH_STATIC | FB_FRAMES_BATCH
        const int src_slot = rs[len - 1], rows = rn[len - 1];  // the cloud associate_pointcloud appended
        while (len >= size && len > 0) {
            const int first = rs[0];
#pragma unroll
            for (int k = 1; k < MMW_RING_MAX; k++) if (k < len) { rs[k - 1] = rs[k]; rn[k - 1] = rn[k]; }
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) if (k == len - 1) rs[k] = first;
            len--;
        }
        int dst_slot = rs[0];
#pragma unroll
        for (int k = 1; k < MMW_RING_MAX; k++) if (k == len) dst_slot = rs[k];
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k == len) rn[k] = rows;
        len++;
        double *ring = st.trk_ring + ((size_t)s * cfg.t_cap + slot) * cfg.ring * (size_t)cfg.ring_rows * 8;
        const int keep = min(rows, cfg.ring_rows);
        if (dst_slot != src_slot) {
            const double2 *a = reinterpret_cast<const double2 *>(ring + (size_t)src_slot * cfg.ring_rows * 8);
            double2 *b = reinterpret_cast<double2 *>(ring + (size_t)dst_slot * cfg.ring_rows * 8);
            for (int e = tid; e < keep * 4; e += kInnerThreads) b[e] = a[e];
        }
        if (tid == 0) {
            rec->ring_len = len;
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) { rec->ring_n[k] = k < len ? rn[k] : 0; rec->ring_slot[k] = rs[k]; }
            rec->inner = size;
        }
        // ---- apply_DBscan(self.batch.effective_data, eps=DB_INNER_EPS) ----
        RowSrc src;
        const int big = 0x7fffffff;
        src.gb = ring;
        src.stride = (size_t)cfg.ring_rows * 8;
        src.slots = (unsigned)rs[0] | ((unsigned)rs[1] << 8) | ((unsigned)rs[2] << 16) | ((unsigned)rs[3] << 24);
        src.c1 = len > 1 ? rn[0] : big;
        src.c2 = len > 2 ? rn[0] + rn[1] : big;
        src.c3 = len > 3 ? rn[0] + rn[1] + rn[2] : big;
        int U = 0;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k < len) U += rn[k];
        bool whole = true;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k < len && rn[k] > cfg.ring_rows) whole = false;
        if (U > UMc || !whole) { err |= ERR_CAPACITY; continue; }  // a frame that was not stored whole, or more points than the BallTree holds
        __syncthreads();  // the copied rows are visible to the whole workgroup (this barrier is also a global-memory fence)
        {   // sklearn's input validation in front of the inner apply_DBscan (Utils.py:272-278): all 8 columns of the track's
            // ring rows -- an assigned point can carry a NaN / infinite doppler or peakVal (the gate only sees columns 0..5).
            // The reference's ValueError leaves track() in the middle of _associate_points_to_tracks: the scene is flagged,
            // this track's inner clustering does not run.
            const int nfb = cloud_nonfinite_bits(src, U);
            if (nfb) { err |= nf_error_of(nfb); continue; }
        }
        const bool tpp = U <= kInnerThreads;  // uniform
        db_lds_layout<true>(UMc, 2, tpp, lds_raw, &L);
        const int ncl = tpp ? dbscan_core<kInnerThreads, true>(cfg, L, src, U, UMc, cfg.db_inner_eps, cfg.db_min_samples, nullptr)
                            : dbscan_core<kInnerThreads, false>(cfg, L, src, U, UMc, cfg.db_inner_eps, cfg.db_min_samples, nullptr);
        // the call, for mmw_get_inner: rows and labels in call order
        if (calls < 16 && tid == 0) ib[2 + calls] = U;
        if (stored + U <= st.inner_cap) {
            for (int i = tid; i < U; i += kInnerThreads) ib[kInnerHdr + stored + i] = L.idx2[i];
            stored += U;
        }
        calls++;
        if (ncl > 1) {  // new_track_clusters = [track_clusters[1]]
            if (T + 1 > cfg.t_cap) { err |= ERR_CAPACITY; }
            else {
                __syncthreads();
                if (tid == 0) { L.misc[4] = hdr->next_uid; hdr->next_uid += 1; }
                __syncthreads();
                add_clusters<kInnerThreads>(cfg, st, L, src, s, U, 2, 1, 1, T);
                T++;
            }
        }
        __syncthreads();  // LDS and the scene's records are reused by the next track
    }
    if (tid == 0) {
        ib[0] = calls;
        ib[1] = stored;
        const int nd = hdr->need_db;
        if (T != Tin) {
            hdr->n_tracks = T;
            hdr->n_upd = T;   // _update_all covers the inner tracks as well
            if (T >= cfg.tr_max_tracks) {  // Tracking.py:693-697: apply_DBscan is not called this frame after all
                hdr->need_db = 0;
                if (db_n_out) db_n_out[s] = -1;
            }
        }
        if (nd & 2) {   // k_track left the verdict on a ring that holds a non-finite row to this re-evaluated trigger
            if (T < cfg.tr_max_tracks) {
                err |= nd >> 2;
                if (db_n_out) db_n_out[s] = kDbRaised;
            }
            hdr->need_db = 0;
        }
        if (err) atomicOr(&hdr->err, err);
    }
}

// The two queues k_track fills WHILE IT RUNS (q[8p + ...] over list 0: the clouds of <= 256 points that can hold a
// cluster; q[kQBig + 8p + ...] over list 1: the clouds of more than 256 points) and their consumers.  An item is
// apply_DBscan + _add_tracks of one scene: 40-90 us of BallTree chain for a small cloud (after the exact pair count of the
// screen), 100-250 us for a large one.  The workgroups of k_chain take them on a second stream BESIDE k_track and k_post, so
// a chain sits in the shadow of the bulk kernels instead of behind them; the worker blocks of k_post (small) and
// k_dbscan_big (large) follow on the context's stream, take whatever is left and wait for the claimed items to finish --
// correctness never depends on k_chain having run.  `epoch` = this step's number: k_post raises q[kQStop] to it when it
// starts, i.e. when no more pushes can come.  Every wait is bounded.
#ifndef MMW_CHAIN_BLOCKS   // (scripts/chain_blocks.sh: diagnostic builds with another count)
#define MMW_CHAIN_BLOCKS 12
#endif
constexpr int kChainBlocks = MMW_CHAIN_BLOCKS;
constexpr int kSpinLimit = 1 << 18;  // polls of ~0.3 us: how long a side-stream worker keeps trying to claim from a non-empty queue
// Waits that MUST succeed (an entry behind its count: a few instructions in the pushing workgroup; the end of a claimed item: one
// BallTree chain) are bounded by TIME, not by iterations -- the 100 MHz s_memrealtime counter; a slow clock or a profiler that
// serialises kernels must not turn into a spurious give-up: 0.2 s for an entry, 2 s for the end of the claimed items
constexpr unsigned long long kMustWaitTicks = 20000000ULL, kDoneWaitTicks = 200000000ULL;
// ... and how long a side-stream worker polls EMPTY queues before it leaves (~3 ms; k_post / k_dbscan_big take whatever comes
// later).  Short on purpose: should the context's stream ever sit behind a polling worker in one hardware queue -- two
// contexts whose streams share queues crosswise can do that, the probe only sees its own pair -- the damage is these 3 ms.
constexpr int kIdleLimit = 1 << 12;
// Polls are RELAXED device-scope atomic loads (served by the L2, no side effects): an ACQUIRE load invalidates the caches of
// the polling CU -- and the non-coherent lines of its XCD's L2 -- every time, and 64 pollers doing that made k_track, which
// runs beside them, 50 % slower.  One acquire fence follows a successful claim instead.
__device__ __forceinline__ int q_load(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void q_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// The small queue from k_post (the step's pushes are complete): tickets are taken with one atomicAdd -- a ticket past the
// count ends the block, and the counters are reset before their parity is used again.  (k_chain claims with a
// compare-and-swap only when an entry is there, so a worker that gives up -- bounded wait -- never holds a ticket.)
template <int NT = 256>
__device__ __forceinline__ void chain_worker_loop(const DevCfg &cfg, const DevState &st, char *lds_raw, int UMc, int CL, int UM_out, int parity,
                                                  int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    DbLds L;
    db_lds_layout<true>(UMc, CL, true, lds_raw, &L);
    // screen scratch behind the BallTree carve-up (post_lds_bytes reserves it)
    char *scr = lds_raw + db_align16(db_lds_layout<false>(UMc, CL, true, nullptr, nullptr));
    float4 *P4 = reinterpret_cast<float4 *>(scr);
    int *cnt = reinterpret_cast<int *>(scr + 4096), *flag = cnt + 256;
    unsigned long long *mm = reinterpret_cast<unsigned long long *>(flag + 2);
    int *ticket = flag + 8;  // (behind mm[3]; post_lds_bytes reserves it)
    int32_t *q = st.q + parity * 8;
    int32_t *ring = st.db_list;  // list 0
    bool have = false;  // an item of this block is being finished (uniform)
    for (;;) {
        __syncthreads();  // every thread is done with the previous scene: its stores are issued, LDS is free again
        if (threadIdx.x == 0) {
            if (have) { __threadfence(); atomicAdd(&q[kQDone], 1); }
            int s = -1, h = -1;
            h = atomicAdd(&q[kQHead], 1) & kQIdxMask;   // (tickets; the tag of the step in the upper bits: mmw_device.hpp)
            if (h >= q_load(&q[kQCount])) h = -1;
            if (h >= 0) {
                int32_t *e = ring + h;
                int v = 0;
                for (const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); (v = q_load(e)) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < kMustWaitTicks;) __builtin_amdgcn_s_sleep(2);
                if (v == 0) { atomicAdd(&st.q[kQTimeout], 1); atomicAdd(&q[kQDone], 1); }
                else {
                    __hip_atomic_store(e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q_acquire();  // what the pushing workgroup stored for the scene is visible from here on
                    s = v - 1;
                }
            }
            *ticket = s;
        }
        __syncthreads();
        const int s = *ticket;  // (rewritten only behind the barrier at the top of the next round)
        if (s < 0) return;
        have = true;
        SceneHdr *hdr = st.hdr + s;
        const int U = hdr->db_u;
        // (seek_inner contexts run no k_chain; k_inner may have filled the track list after k_track queued the scene: then
        //  there is no apply_DBscan this frame -- uniform)
        if (!(cfg.seek_inner && !hdr->need_db)) {
            if (cloud_pairs_prove_no_core<NT>(cfg, ring_rows_of(cfg, st, hdr, s), U, P4, cnt, mm, flag))
                cloud_finish_empty(st, hdr, s, U, UM_out, labels_out, db_n_out);
            else
                spawn_scene<NT, true>(cfg, st, L, s, UMc, CL, UM_out, true, parity, labels_out, db_n_out);
        }
    }
}

// The clouds of more than 256 points (a scene without tracks clusters its whole ring: the start-up frames, and every scene
// whose tracks have all expired): 100-250 us of BallTree chain each on a 512-thread workgroup.  They sit in a queue k_track
// fills while it runs (q[kQBig + ...], ring = list 1).  Consumers:
//   k_chain           twelve workgroups on a side stream, BESIDE k_track and k_post, for this queue and the small clouds'
//                     (claims by compare-and-swap, leaves when k_post has begun and both queues are empty; not in the
//                     start-up frames, whose pushes carry no release: cfg.big_live);
//   k_dbscan_big      behind k_post on the context's stream: takes what is left (tickets by atomicAdd: the pushes are
//                     complete) and does not end before every claimed cloud is finished;
//   k_dbscan_startup  the same for the first frames after a reset, when every cloud fits one point per thread, under a
//                     register budget that lets two workgroups share a CU;
//   k_post            in contexts of <= kSmallContextScenes scenes, whose step is launch latency: its worker blocks take the
//                     large clouds too (256 threads, strided build -- rare there) and k_dbscan_big is not launched.
// Correctness never depends on k_chain having run.
constexpr int kBigThreads = 512;
template <int NT, bool AFTER_TRACK, bool TPP_ONLY>
__device__ __forceinline__ void big_worker_loop(const DevCfg &cfg, const DevState &st, char *lds_raw, int UMc, int CL, int UM_out, int parity,
                                                int epoch, int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    DbLds L;
    // the ticket word sits behind the BallTree carve-up (big_lds_bytes reserves it)
    const size_t a = db_lds_layout<false>(UMc, CL, true, nullptr, nullptr), b = TPP_ONLY ? 0 : db_lds_layout<false>(UMc, CL, false, nullptr, nullptr);
    int *ticket = reinterpret_cast<int *>(lds_raw + db_align16(a > b ? a : b));
    int32_t *q = st.q + kQBig + parity * 8;
    int32_t *ring = st.db_list + cfg.n_scenes;  // list 1
    bool have = false;
    for (;;) {
        __syncthreads();  // every thread is done with the previous cloud: its stores are issued, LDS is free again
        if (threadIdx.x == 0) {
            if (have) { __threadfence(); atomicAdd(&q[kQDone], 1); }
            int s = -1, h = -1;
            if (AFTER_TRACK) {
                if ((q_load(&q[kQHead]) & kQIdxMask) < q_load(&q[kQCount])) {  // (an empty queue costs two loads, no atomic)
                    h = atomicAdd(&q[kQHead], 1) & kQIdxMask;
                    if (h >= q_load(&q[kQCount])) h = -1;
                }
            } else {
                for (int spins = 0; spins < kSpinLimit; spins++) {
                    const int hh = q_load(&q[kQHead]), c = q_load(&q[kQCount]);
                    if ((hh & ~kQIdxMask) == q_tag(epoch) && (hh & kQIdxMask) < c) {
                        if (atomicCAS(&q[kQHead], hh, hh + 1) == hh) { h = hh & kQIdxMask; break; }
                        continue;
                    }
                    if (q_load(&st.q[kQStop]) - epoch >= 0) break;  // k_post of this step has begun and the queue is empty: done
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            if (h >= 0) {
                int32_t *e = ring + h;
                int v = 0;
                for (const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); (v = q_load(e)) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < kMustWaitTicks;) __builtin_amdgcn_s_sleep(2);
                if (v == 0) { atomicAdd(&st.q[kQTimeout], 1); atomicAdd(&q[kQDone], 1); }
                else {
                    __hip_atomic_store(e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q_acquire();  // what the pushing workgroup stored for the scene is visible from here on
                    s = v - 1;
                }
            }
            *ticket = s;
        }
        __syncthreads();
        const int s = *ticket;
        if (s < 0) return;
        have = true;
        if (cfg.seek_inner && !st.hdr[s].need_db) continue;  // cancelled by k_inner (uniform)
        // a cloud that fits one point per thread takes the thread-per-point build of the small class (registers
        // hold the 8 columns, one bitonic sort per level): 3-4x less tree-build time than the strided build
        const bool tpp = TPP_ONLY || st.hdr[s].db_u <= NT;  // uniform
        db_lds_layout<true>(UMc, CL, tpp, lds_raw, &L);
        if (tpp) spawn_scene<NT, true>(cfg, st, L, s, UMc, CL, UM_out, false, parity, labels_out, db_n_out);
        else if constexpr (!TPP_ONLY) spawn_scene<NT, false>(cfg, st, L, s, UMc, CL, UM_out, false, parity, labels_out, db_n_out);
    }
}

// the launch must not end before every claimed cloud is finished (a side-stream worker may still hold one): the next
// launches read what the spawn writes.  Bounded: ~2 s, then a sticky error.
__device__ __forceinline__ void big_wait_done(const DevState &st, int parity)
{
    const int32_t *qp = st.q + kQBig + parity * 8;
    const int want = q_load(&qp[kQCount]);
    for (const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); q_load(&qp[kQDone]) < want && __builtin_amdgcn_s_memrealtime() - t0 < kDoneWaitTicks;) __builtin_amdgcn_s_sleep(8);
    if (q_load(&qp[kQDone]) < want) atomicAdd(&st.q[kQTimeout], 1);
    q_acquire();
}

// Next frame's schedule for k_track: scenes by descending track count -- a counting sort over the scene headers by ONE 256-thread
// workgroup of k_post (its own block: as a chore of the last worker block, two passes of one dependent load per 256 scenes, it was
// the longest chain of the launch -- 18 us at 4096 scenes); the order inside a count is irrelevant.  (Scenes without tracks
// FIRST -- they are the ones that cluster their whole ring, 100-250 us on a chain worker -- was tried: no measurable gain in either
// window, and they are the filler k_track's tail wants.)  The key is n_upd, which nothing in this launch writes: n_tracks may be
// raised by a spawning worker between the two passes, and a scene counted in one bin but scattered into another would break the
// permutation.
// TWO classes, each by descending track count: first the scenes that CAN reach apply_DBscan next frame (fewer than TR_MAX_TRACKS
// tracks: Tracking.py:693-697), then the full ones.  A cloud that needs the BallTree is a 45-60 us chain that starts when its
// scene's workgroup of k_track ends; pushed from the launch's first round it is finished long before k_post, pushed from the last
// one it is what k_post's block 0 -- and with it the step -- waits for.  With "most tracks first" alone the scenes that can push
// were the LAST of the launch: at K = T block 0 left at 45-60 us in two frames of three while the update blocks were done at 35
// (scripts/wg_times_post_frames.py, NOTEBOOK round 5).  (A full scene can still trigger when a track expires in this frame's
// maintenance; rare, and correct either way -- the order is a schedule, not a decision.)
// hist: LDS, 2 (t_cap + 1) + 1 ints.  Loads in batches of eight per thread, all in flight at once.
// (192: just above a ring of clutter; same box, alternating: 256 equal within the noise, 128 -- too many scenes in front -- 3-6 % slower)
#ifndef MMW_SCHED_BIG_U   // (diagnostic builds: another ring size from which a scene leads the schedule)
#define MMW_SCHED_BIG_U 192
#endif
template <int NT = 256>
__device__ __forceinline__ void post_schedule_sort(const DevCfg &cfg, const DevState &st, int parity, int *hist)
{
    const int tid = threadIdx.x, nb = cfg.t_cap + 1, S = cfg.n_scenes;
    auto bin_of = [&](int key) {
        const int t = (key & 0xffff) > cfg.t_cap ? cfg.t_cap : (key & 0xffff);
        // (in front of everything: the scenes whose ring holds MORE THAN CLUTTER -- more than MMW_SCHED_BIG_U unassigned points: a cloud that
        //  was not clustered away this frame comes back next frame, a 45-250 us chain again --, then the scenes without tracks, then the
        //  first class by descending track count.
        //  Ascending -- "the fewer tracks a scene has kept, the more of its points are unassigned" -- was measured: the launch then
        //  ends on its heaviest workgroups, mixed population + 4 %)
        if (key >> 16) return 0;
        return t == 0 ? 1 : (t < cfg.tr_max_tracks ? 1 : 1 + nb) + (nb - t);
    };
    for (int i = tid; i <= 2 * nb + 1; i += NT) hist[i] = 0;
    __syncthreads();
    for (int base = 0; base < S; base += NT * 8) {
        int key[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sc = base + u * NT + tid;
            key[u] = sc < S ? ((st.hdr[sc].n_upd < 0 ? 0 : st.hdr[sc].n_upd) & 0xffff) | (st.hdr[sc].db_u > MMW_SCHED_BIG_U ? 0x10000 : 0) : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (base + u * NT + tid < S) atomicAdd(&hist[bin_of(key[u])], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b <= 2 * nb + 1; b++) { const int c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (int base = 0; base < S; base += NT * 8) {
        int key[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sc = base + u * NT + tid;
            key[u] = sc < S ? ((st.hdr[sc].n_upd < 0 ? 0 : st.hdr[sc].n_upd) & 0xffff) | (st.hdr[sc].db_u > MMW_SCHED_BIG_U ? 0x10000 : 0) : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sc = base + u * NT + tid;
            if (sc < S) st.perm[(size_t)(parity ^ 1) * S + atomicAdd(&hist[bin_of(key[u])], 1)] = sc;
        }
    }
}

// k_post: what follows the association of a frame, in ONE launch of 256-thread workgroups of two kinds:
//   blocks [0, G0)   apply_DBscan + _add_tracks (Tracking.py:697-703) for the clouds of <= 256 points k_track's / k_scene's
//                    screens could not rule out (work list 3, and what k_chain has left of queue 0): the exact pair
//                    count once more (a few microseconds, for the handful of scenes per step that arrive), then the
//                    BallTree, a latency chain of ~60 us that would otherwise leave the chip idle;
//   block G0         next frame's schedule for k_track (post_schedule_sort above; not in the fused step)
//   the others       _update_all (Tracking.py:598-603) of four (scene, quarter) units each, one wave per unit
//                    (update_tracks_wave, mmw_kalman.hpp) -- the bulk work the BallTree scenes hide under.
// The two touch disjoint state: the update covers the hdr->n_upd tracks that existed before this frame's
// clusters, the spawn appends records behind them.
// Two waves per SIMD (no register cap: the BallTree path of the worker blocks takes ~205 VGPRs, the update 156).  Rounds 2-4 ran
// the launch under a 168-VGPR cap (three waves per SIMD) for the update's sake, the workers spilling 54 VGPRs / 188 bytes of
// scratch per lane; since the update's broadcasts moved from the LDS to DPP moves (round 4) the third wave buys it nothing --
// same box, alternating (scripts/ab_libs.sh, profiles/NOTEBOOK.md round 5): 4096 scenes k_post 36-38 us either way, 512 scenes
// (whose DBSCAN is all in these worker blocks) 16.6 -> 14.5 us, the step 0.0648 -> 0.0627 ms -- and nothing spills.
#ifndef MMW_POST_OCC   // (diagnostic builds: another register budget for the launch)
#define MMW_POST_OCC 2
#endif
template <int DX, int NT>
__global__ __launch_bounds__(NT, MMW_POST_OCC) void k_post(DevCfg cfg, DevState st, const int32_t *__restrict__ n_pts, int nq, int G0, int UMc, int CL,
                                              int UMb, int CLb, int UM_out, int parity, int epoch, int32_t *__restrict__ labels_out,
                                              int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#if defined(MMW_STAMPS) && defined(MMW_STAMPS_POST)   // diagnostic build: start / end of every workgroup of this launch (scripts/wg_times_post.py)
    struct WgStamp {
        unsigned long long *w;
        __device__ WgStamp(const DevState &st) : w(nullptr) {
            if (threadIdx.x == 0 && blockIdx.x < 2048) {
                w = st.stats + kStatSlots * kStatWords + 256 + blockIdx.x * 4;
                w[0] = __builtin_amdgcn_s_memrealtime();
                w[1] = __builtin_amdgcn_s_memtime();
            }
        }
        __device__ ~WgStamp() { if (w) { w[2] = __builtin_amdgcn_s_memrealtime(); w[3] = __builtin_amdgcn_s_memtime(); } }
    } wg_stamp(st);
#endif
    // k_track has finished: no more pushes this step.  Block 0 says so before anything else, in EVERY step (side workers or not: the
    // stop epoch is what paces the side stream, k_chain claims only when it is exactly one step behind its own)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&st.q[kQStop], epoch);
    if ((int)blockIdx.x < G0) {
        __builtin_amdgcn_s_setprio(3);  // the latency chain goes first whenever it has an instruction ready
        // Side-stream workers (the large contexts): k_track is complete, so the queues' counts are final and their heads only grow --
        // one round trip tells a worker block that nothing is left to claim, and it is gone: its workgroup slot is one the Kalman
        // update behind it is waiting for (the four dependent atomics / loads of the queue protocol kept all 256 of them for 17 us).
        // Block 0 stays: it releases k_chain and holds the launch until every claimed cloud is finished.
        if (cfg.side_worker && blockIdx.x != 0) {
            // (k_chain moves the head WHILE this is read: one thread decides for the workgroup -- waves that read different
            //  heads would part ways in front of the worker loop's barriers)
            int *leave = reinterpret_cast<int *>(lds_raw);
            if (threadIdx.x == 0) {
                const int c3 = st.db_count[parity * 4 + 3], c0 = q_load(&st.q[parity * 8 + kQCount]), h0 = q_load(&st.q[parity * 8 + kQHead]) & kQIdxMask;
                const int cb = UMb > 0 ? q_load(&st.q[kQBig + parity * 8 + kQCount]) - (q_load(&st.q[kQBig + parity * 8 + kQHead]) & kQIdxMask) : 0;
                *leave = (c3 == 0 && h0 >= c0 && cb <= 0) ? 1 : 0;
            }
            __syncthreads();
            const int go = *leave;
            __syncthreads();  // (the word is the worker loop's LDS again from here)
            if (go) return;
        }
        // No side-stream workers this step (small contexts, the start-up frames, a profiler): what the work list and the two
        // queues hold now is all there is, and nobody else can have claimed any of it -- three counters in one round trip, and
        // a worker of a step without apply_DBscan (most steps) is gone; the atomics below are 4 dependent round trips more.
        if (!cfg.side_worker) {
            const int c3 = st.db_count[parity * 4 + 3], c0 = st.q[parity * 8 + kQCount], cb = st.q[kQBig + parity * 8 + kQCount];
            if ((c3 | c0 | cb) == 0) return;  // (uniform: the same words in every thread)
        }
        {   // list 3 (the clouds k_track did not queue early): a static share per block, as short as a pair count each
            DbLds L;
            db_lds_layout<true>(UMc, CL, true, lds_raw, &L);
            char *scr = lds_raw + db_align16(db_lds_layout<false>(UMc, CL, true, nullptr, nullptr));
            float4 *P4 = reinterpret_cast<float4 *>(scr);
            int *cnt = reinterpret_cast<int *>(scr + 4096), *flag = cnt + 256;
            unsigned long long *mm = reinterpret_cast<unsigned long long *>(flag + 2);
            const int count = st.db_count[parity * 4 + 3];
            for (int w = blockIdx.x; w < count; w += G0) {
                const int s = st.db_list[(size_t)3 * cfg.n_scenes + w];
                SceneHdr *hdr = st.hdr + s;
                const int U = hdr->db_u;
                if (cfg.seek_inner && !hdr->need_db) continue;  // k_inner filled the track list: no apply_DBscan this frame (uniform)
                if (cloud_pairs_prove_no_core<NT>(cfg, ring_rows_of(cfg, st, hdr, s), U, P4, cnt, mm, flag))
                    cloud_finish_empty(st, hdr, s, U, UM_out, labels_out, db_n_out);
                else
                    spawn_scene<NT, true>(cfg, st, L, s, UMc, CL, UM_out, true, parity, labels_out, db_n_out);
                __syncthreads();  // LDS is reused by the next scene
            }
        }
        chain_worker_loop<NT>(cfg, st, lds_raw, UMc, CL, UM_out, parity, labels_out, db_n_out);
        if (UMb > 0 && st.q[kQBig + parity * 8 + kQCount] != 0) {  // small context: the large clouds here as well (k_track is complete: plain load)
            big_worker_loop<NT, true, false>(cfg, st, lds_raw, UMb, CLb, UM_out, parity, 0, labels_out, db_n_out);
            if (blockIdx.x == 0 && threadIdx.x == 0) big_wait_done(st, parity);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // the launch must not end before every claimed scene is finished (k_chain may still hold one): the next
            // launches read what the spawn writes.  Bounded: ~2 s, then a sticky error.
            const int32_t *qp = st.q + parity * 8;
            const int want = q_load(&qp[kQCount]);
            for (const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); q_load(&qp[kQDone]) < want && __builtin_amdgcn_s_memrealtime() - t0 < kDoneWaitTicks;) __builtin_amdgcn_s_sleep(8);
            if (q_load(&qp[kQDone]) < want) atomicAdd(&st.q[kQTimeout], 1);
            q_acquire();
        }
        return;
    }
    if ((int)blockIdx.x == G0) {  // (not launched by the fused step: k_scene reads no schedule, every scene is resident)
        post_schedule_sort<NT>(cfg, st, parity, reinterpret_cast<int *>(lds_raw));
        return;
    }
    const int wave = threadIdx.x >> 6;
    const int unit = ((int)blockIdx.x - G0 - 1) * (NT / 64) + wave;
    if (unit >= cfg.n_scenes * nq) return;
    double *scratch = reinterpret_cast<double *>(lds_raw) + (size_t)wave * 4 * kUpdScratch;
    if (tracks_dense(cfg, nq)) {  // four real tracks per wave, from the lists k_track built this frame
        update_tracks_dense<DX>(cfg, st, unit, cfg.n_scenes * nq, parity, scratch);
        return;
    }
    const int us = unit / nq, q = unit - us * nq;
    const int s = st.perm[(size_t)parity * cfg.n_scenes + us];  // this step's schedule (the worker above writes the next one)
    update_tracks_wave<DX>(cfg, st, n_pts, s, q, nq, scratch);
}

// The chain workers of the side stream: 512-thread workgroups that serve BOTH queues while k_track and k_post run -- the
// large clouds first (the longer chains), then the small ones (pair-count screen, then the BallTree on the thread-per-point
// build).  One kernel, one stream: every further stream with a spinning kernel is one more hardware queue the context's
// stream must not share (see probe_side_stream in mmw_api.hip).
__global__ __launch_bounds__(kBigThreads) void k_chain(DevCfg cfg, DevState st, int UMc, int CL, int UM_out, int parity, int epoch,
                                               int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    DbLds L;
    const size_t la = db_lds_layout<false>(UMc, CL, true, nullptr, nullptr), lb = db_lds_layout<false>(UMc, CL, false, nullptr, nullptr);
    char *scr = lds_raw + db_align16(la > lb ? la : lb);  // screen scratch + ticket behind the BallTree carve-up (chain_lds_bytes)
    float4 *P4 = reinterpret_cast<float4 *>(scr);
    int *cnt = reinterpret_cast<int *>(scr + 4096), *flag = cnt + 256;
    unsigned long long *mm = reinterpret_cast<unsigned long long *>(flag + 2);
    int *ticket = flag + 8;  // [2]: scene, queue
    int32_t *qs = st.q + parity * 8, *qb = st.q + kQBig + parity * 8;
    int have = 0;  // queue of the item this workgroup is finishing (1 small, 2 large; uniform)
    for (;;) {
        __syncthreads();  // every thread is done with the previous cloud: its stores are issued, LDS is free again
        if (threadIdx.x == 0) {
            if (have) { __threadfence(); atomicAdd(&(have == 2 ? qb : qs)[kQDone], 1); }
            int s = -1, h = -1, kind = 0;
            for (int spins = 0; spins < kIdleLimit + cfg.n_scenes; spins++) {  // (k_track's first push comes later in a larger context)
                // Whose pushes are these?  The queues of a parity serve every second step and this launch knows only ITS step's
                // arguments (output buffers, big_live).  It is paced by the stop epochs alone (no event orders the two streams), and it
                // idles out after ~3 ms: with steps queued ahead of a stalled context stream (a long upload, a caller's kernel) the
                // launches of several steps can pass through here before the first k_track runs.  So: claim only while the step
                // before ours has reached its k_post (the counters of our parity were reset by its k_track, what is pushed now is
                // ours) and no later step has (ours is over: a launch that comes this late leaves).
                const int stop = q_load(&st.q[kQStop]);
#ifndef MMW_MUTANT_CHAIN_NOGATE   // (diagnostic build: tests/test_gpu_runahead.py must FAIL without the two lines below)
                if (stop - epoch >= 1) break;
                if (stop - (epoch - 1) < 0) { __builtin_amdgcn_s_sleep(8); continue; }
#endif
                // (a claim is a compare-and-swap on TAG + index: the head word was tagged with our step's number when the queue was reset
                //  for us; a worker that read `stop` above and was then held up for two steps fails here instead of taking a later
                //  step's cloud into ITS step's output buffers -- what the stop check alone let happen under six processes)
                const int hb = q_load(&qb[kQHead]), cb = cfg.big_live ? q_load(&qb[kQCount]) : 0;  // (start-up frames: pushed without a release, not ours)
                if ((hb & ~kQIdxMask) == q_tag(epoch) && (hb & kQIdxMask) < cb) {
                    if (atomicCAS(&qb[kQHead], hb, hb + 1) == hb) { h = hb & kQIdxMask; kind = 2; break; }
                    continue;
                }
                const int hs = q_load(&qs[kQHead]), cs = q_load(&qs[kQCount]);
                if ((hs & ~kQIdxMask) == q_tag(epoch) && (hs & kQIdxMask) < cs) {
                    if (atomicCAS(&qs[kQHead], hs, hs + 1) == hs) { h = hs & kQIdxMask; kind = 1; break; }
                    continue;
                }
                if (stop - epoch >= 0) break;  // k_post of this step had begun before the queues were looked at, and both are empty: done
                __builtin_amdgcn_s_sleep(8);
            }
            if (h >= 0) {
                // the entry follows its count by a few instructions in the pushing workgroup
                int32_t *e = st.db_list + (kind == 2 ? cfg.n_scenes : 0) + h;
                int v = 0;
                for (const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); (v = q_load(e)) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < kMustWaitTicks;) __builtin_amdgcn_s_sleep(2);
                if (v == 0) { atomicAdd(&st.q[kQTimeout], 1); atomicAdd(&(kind == 2 ? qb : qs)[kQDone], 1); }
                else {
                    __hip_atomic_store(e, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q_acquire();  // what the pushing workgroup stored for the scene is visible from here on
                    s = v - 1;
                }
            }
            ticket[0] = s;
            ticket[1] = kind;
        }
        __syncthreads();
        const int s = ticket[0];  // (rewritten only behind the barrier at the top of the next round)
        if (s < 0) return;
        have = ticket[1];
        SceneHdr *hdr = st.hdr + s;
        const int U = hdr->db_u;
        if (have == 2) {
            const bool tpp = U <= kBigThreads;  // uniform
            db_lds_layout<true>(UMc, CL, tpp, lds_raw, &L);
            if (tpp) spawn_scene<kBigThreads, true>(cfg, st, L, s, UMc, CL, UM_out, false, parity, labels_out, db_n_out);
            else spawn_scene<kBigThreads, false>(cfg, st, L, s, UMc, CL, UM_out, false, parity, labels_out, db_n_out);
        } else {
            db_lds_layout<true>(UMc, CL, true, lds_raw, &L);
            if (cloud_pairs_prove_no_core<kBigThreads>(cfg, ring_rows_of(cfg, st, hdr, s), U, P4, cnt, mm, flag))
                cloud_finish_empty(st, hdr, s, U, UM_out, labels_out, db_n_out);
            else
                spawn_scene<kBigThreads, true>(cfg, st, L, s, UMc, CL, UM_out, true, parity, labels_out, db_n_out);
        }
    }
}

__global__ __launch_bounds__(kBigThreads) void k_dbscan_big(DevCfg cfg, DevState st, int UMc, int CL, int UM_out, int parity,
                                                    int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // (k_track is complete: the count is final and a plain load.  No large cloud this step -- nearly every step of a
    // tracked scene set -- and the whole launch leaves on that one word)
    if (st.q[kQBig + parity * 8 + kQCount] == 0) return;
    big_worker_loop<kBigThreads, true, false>(cfg, st, lds_raw, UMc, CL, UM_out, parity, 0, labels_out, db_n_out);
    if (blockIdx.x == 0 && threadIdx.x == 0) big_wait_done(st, parity);
}

__global__ __launch_bounds__(kBigThreads, 4) void k_dbscan_startup(DevCfg cfg, DevState st, int UMc, int CL, int UM_out, int parity,
                                                                   int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    if (st.q[kQBig + parity * 8 + kQCount] == 0) return;
    big_worker_loop<kBigThreads, true, true>(cfg, st, lds_raw, UMc, CL, UM_out, parity, 0, labels_out, db_n_out);
    if (blockIdx.x == 0 && threadIdx.x == 0) big_wait_done(st, parity);
}

// Utils.apply_DBscan on caller-provided clouds: pts[S][max_n][8]
__global__ __launch_bounds__(256) void k_dbscan_only(DevCfg cfg, int UM, const double *__restrict__ pts,
                                                    const int32_t *__restrict__ n_all, int max_n, double eps,
                                                    int min_samples, int32_t *__restrict__ labels_out,
                                                    int32_t *__restrict__ ncl_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    DbLds L;
    db_lds_layout<true>(UM, 0, false, lds_raw, &L);
    const int s = blockIdx.x, tid = threadIdx.x;
    const int U = n_all[s];
    if (U <= 0 || U > UM) { if (tid == 0 && ncl_out) ncl_out[s] = 0; return; }
    RowSrc src;
    src.gb = pts + (size_t)s * max_n * 8;
    src.stride = 0;
    src.slots = 0;
    src.c1 = src.c2 = src.c3 = 0x7fffffff;
    if (const int nfb = cloud_nonfinite_bits(src, U)) {   // sklearn raises ValueError: no labels, n_clusters = -(error bit)
        if (tid == 0 && ncl_out) ncl_out[s] = -nf_error_of(nfb);
        return;
    }
    const int ncl = dbscan_core<256, false>(cfg, L, src, U, UM, eps, min_samples, nullptr);
    for (int i = tid; i < U; i += 256) labels_out[(size_t)s * max_n + i] = L.idx2[i];
    if (tid == 0 && ncl_out) ncl_out[s] = ncl;
}

// ---- Clouds of more than 1920 points (a context with ring * max_pts up to 4096: apply_DBscan itself has no limit,
//      Utils.py:250-291) ----------------------------------------------------------------------------------------------------
// 64 .. 128 leaves: the carve-up (up to ~0.5 MB) does not fit the LDS and lives in GLOBAL memory instead, one slab per worker
// (DevState::huge_scratch) -- the same dbscan_core / add_clusters, instantiated over pointers into that slab: the address
// space is the only difference (workgroup barriers order global memory inside a workgroup as they order the LDS: its waves
// share the CU's L1, stores write through), and the leaf-state mask of a position is MW = 4 words instead of one.  A chain
// of L2 round trips instead of LDS ones, several times slower per cloud: this is the path that makes such a context POSSIBLE
// (a scene that lost its tracks clusters its whole ring), not one the step is tuned around.  k_track puts these scenes on
// work list 2; one launch behind k_dbscan_big, only in contexts whose rings can hold such a cloud.
constexpr int kHugeThreads = 512;
constexpr int kHugeMW = 4;       // <= 128 leaves
static_assert(MMW_RING_MAX * MMW_MAX_PTS_LIMIT <= 4096 && 2 * (MMW_RING_MAX * MMW_MAX_PTS_LIMIT / 30 + 1) <= 64 * 2 * kHugeMW, "the largest cloud: 12 position bits in the labelling's seed key, <= 128 leaves");
constexpr int kHugeWorkers = 64;
__global__ __launch_bounds__(kHugeThreads) void k_dbscan_huge(DevCfg cfg, DevState st, int UMc, int CL, int UM_out, int parity,
                                                              int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    DbLds L;
    db_hybrid_layout<true>(UMc, CL, lds_raw, st.huge_scratch + (size_t)blockIdx.x * st.huge_stride, &L, kHugeMW, nullptr);
    const int count = st.db_count[parity * 4 + 2];
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int s = st.db_list[(size_t)2 * cfg.n_scenes + w];
        if (cfg.seek_inner && !st.hdr[s].need_db) continue;  // cancelled by k_inner (uniform)
        spawn_scene<kHugeThreads, false, kHugeMW>(cfg, st, L, s, UMc, CL, UM_out, false, parity, labels_out, db_n_out);
        __syncthreads();  // the slab is reused by the next scene
    }
}

// Utils.apply_DBscan on caller-provided clouds of more than 1920 points (mmw_dbscan): as k_dbscan_only, slab per workgroup
__global__ __launch_bounds__(kHugeThreads) void k_dbscan_only_huge(DevCfg cfg, DevState st, int UM, int n_clouds, const double *__restrict__ pts,
                                                                   const int32_t *__restrict__ n_all, int max_n, double eps, int min_samples,
                                                                   int32_t *__restrict__ labels_out, int32_t *__restrict__ ncl_out)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    DbLds L;
    db_hybrid_layout<true>(UM, 0, lds_raw, st.huge_scratch + (size_t)blockIdx.x * st.huge_stride, &L, kHugeMW, nullptr);
    const int tid = threadIdx.x;
    for (int s = blockIdx.x; s < n_clouds; s += gridDim.x) {
        const int U = n_all[s];
        if (U <= 0 || U > UM) { if (tid == 0 && ncl_out) ncl_out[s] = 0; continue; }
        RowSrc src;
        src.gb = pts + (size_t)s * max_n * 8;
        src.stride = 0;
        src.slots = 0;
        src.c1 = src.c2 = src.c3 = 0x7fffffff;
        if (const int nfb = cloud_nonfinite_bits(src, U)) {   // sklearn raises ValueError: no labels, n_clusters = -(error bit)
            if (tid == 0 && ncl_out) ncl_out[s] = -nf_error_of(nfb);
            continue;
        }
        const int ncl = dbscan_core<kHugeThreads, false, kHugeMW>(cfg, L, src, U, UM, eps, min_samples, nullptr);
        for (int i = tid; i < U; i += kHugeThreads) labels_out[(size_t)s * max_n + i] = L.idx2[i];
        if (tid == 0 && ncl_out) ncl_out[s] = ncl;
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------------------
static const int kClassUM[3] = {256, 768, 1920};
static size_t dbscan_huge_lds_bytes(int UM, int cl);

int dbscan_class_um(int cls, int UM) { return kClassUM[cls] < UM ? kClassUM[cls] : UM; }
int dbscan_class_cl(int cls, int UM, int t_cap, int min_samples)
{
    const int um = dbscan_class_um(cls, UM);
    int cl = um / (min_samples > 0 ? min_samples : 1) + 1;
    return cl < t_cap ? cl : t_cap;
}
size_t dbscan_lds_bytes(int cls, int UM, int t_cap, int min_samples)
{
    const int um = dbscan_class_um(cls, UM), cl = dbscan_class_cl(cls, UM, t_cap, min_samples);
    const size_t strided = db_lds_layout<false>(um, cl, false, nullptr, nullptr), tpp = db_lds_layout<false>(um, cl, true, nullptr, nullptr);
    if (cls == 0) return tpp;
    return strided > tpp ? strided : tpp;  // k_dbscan_big carves either way per cloud
}
size_t dbscan_only_lds_bytes(int UM) { return db_lds_layout<false>(UM < kClassUM[2] ? UM : kClassUM[2], 0, false, nullptr, nullptr); }

static size_t post_lds_bytes(int UM, int t_cap, int min_samples, int waves = 4)
{
    const size_t upd = (size_t)waves * 4 * kUpdScratch * sizeof(double);
    const size_t db = db_align16(dbscan_lds_bytes(0, UM, t_cap, min_samples)) + 4096 + (256 + 2) * 4 + 3 * 8 + 16 + 64;  // + pair-count scratch + ticket
    return upd > db ? upd : db;
}

// capacity of the large-cloud kernels this step: no cloud has more than `u_bound` points (the caller knows how many frames the
// rings can hold so soon after a reset), none more than UM, and the BallTree emulation holds 1920
static int big_um(int UM, int u_bound)
{
    int um = u_bound < UM ? u_bound : UM;
    return um < kClassUM[2] ? um : kClassUM[2];
}
static int big_cl(int um, int t_cap, int min_samples)
{
    const int cl = um / (min_samples > 0 ? min_samples : 1) + 1;
    return cl < t_cap ? cl : t_cap;
}
static size_t big_lds_bytes(int um, int cl, bool tpp_only)
{
    const size_t a = db_lds_layout<false>(um, cl, true, nullptr, nullptr), b = tpp_only ? 0 : db_lds_layout<false>(um, cl, false, nullptr, nullptr);
    return db_align16(a > b ? a : b) + 16;  // + the ticket word
}
// k_chain: the capacity of the large clouds (at least one slot per thread), + the pair-count scratch and the ticket
static int chain_um(int UM, int u_bound)
{
    const int um = big_um(UM, u_bound);
    return um < kBigThreads ? kBigThreads : um;
}
static size_t chain_lds_bytes(int um, int t_cap, int min_samples)
{
    return big_lds_bytes(um, big_cl(um, t_cap, min_samples), false) + 4096 + (256 + 2) * 4 + 3 * 8 + 16 + 64;
}

hipError_t prepare_dbscan(int UM, int t_cap, int min_samples)
{
    const int bum = big_um(UM, UM);
    const size_t big = big_lds_bytes(bum, big_cl(bum, t_cap, min_samples), false);
    const size_t post = post_lds_bytes(UM, t_cap, min_samples) > big ? post_lds_bytes(UM, t_cap, min_samples) : big;  // (small contexts: the large clouds in k_post)
    hipError_t e = hipFuncSetAttribute((const void *)k_post<9, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)post);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_post<6, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)post);
    if (e != hipSuccess) return e;
    const size_t c5 = chain_lds_bytes(kBigThreads, t_cap, min_samples), post512 = (c5 > big ? c5 : big) > post ? (c5 > big ? c5 : big) : post;   // (behind the fused step: 512-thread worker blocks)
    e = hipFuncSetAttribute((const void *)k_post<9, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)post512);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_post<6, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)post512);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_dbscan_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds_bytes(chain_um(UM, UM), t_cap, min_samples));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_dbscan_startup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
    if (e != hipSuccess) return e;
    if (UM > kClassUM[2]) {   // the clouds of more than 1920 points: hot arrays in the LDS, the rest on slabs in global memory
        const size_t hot = dbscan_huge_lds_bytes(UM, big_cl(UM, t_cap, min_samples));
        e = hipFuncSetAttribute((const void *)k_dbscan_huge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hot);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void *)k_dbscan_only_huge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hot);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute((const void *)k_dbscan_only, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dbscan_only_lds_bytes(UM));
}

int inner_um(const DevCfg &cfg)
{
    long long v = (long long)cfg.ring * cfg.ring_rows;
    if (v < 512) v = 512;  // (capacity only: the 512-thread build wants one exchange slot per thread)
    return (int)(v < 30 * 64 ? v : 30 * 64);  // the BallTree emulation holds <= 32 leaves
}
static size_t inner_lds_bytes(const DevCfg &cfg)
{
    const int um = inner_um(cfg);
    const size_t a = db_lds_layout<false>(um, 2, false, nullptr, nullptr), b = db_lds_layout<false>(um, 2, true, nullptr, nullptr);
    return a > b ? a : b;
}
hipError_t prepare_inner(const DevCfg &cfg)
{
    return hipFuncSetAttribute((const void *)k_inner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)inner_lds_bytes(cfg));
}
size_t inner_lds_demand(const DevCfg &cfg) { return inner_lds_bytes(cfg); }
void launch_inner(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, int32_t *db_n, hipStream_t stream)
{
    hipLaunchKernelGGL(k_inner, dim3(cfg.n_scenes), dim3(kInnerThreads), inner_lds_bytes(cfg), stream, cfg, st, n_pts, inner_um(cfg), db_n);
}

// _update_all + the BallTree DBSCAN of the small clouds (work list 3)
void launch_post(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, int UM, int u_bound, int parity, int epoch, int32_t *labels,
                 int32_t *db_n, hipStream_t stream)
{
    const int nq = kalman_waves_per_scene(cfg.tr_max_tracks);
    const int S = cfg.n_scenes, units = cfg.fused ? 0 : S * nq;  // (fused step: _update_all ran inside k_scene, only the DBSCAN workers are left)
    int G0 = S < 256 ? S : 256;
    const int umc = dbscan_class_um(0, UM), cl = dbscan_class_cl(0, UM, cfg.t_cap, cfg.db_min_samples);
    size_t lds = post_lds_bytes(UM, cfg.t_cap, cfg.db_min_samples);
    // a small context's step is launch latency: its large clouds are taken here too and k_dbscan_big is not launched
    int umb = 0, clb = 0;
    if (S <= kSmallContextScenes && big_um(UM, u_bound) > kClassUM[0]) {
        umb = big_um(UM, u_bound);
        clb = big_cl(umb, cfg.t_cap, cfg.db_min_samples);
        const size_t big = big_lds_bytes(umb, clb, false);
        if (big > lds) lds = big;
        // (with that much LDS a CU holds one workgroup: workers + update units must stay one wave of workgroups)
        constexpr int kSmallContextWorkers = 64;  // (32 .. 128 measured equal)
        if (G0 > kSmallContextWorkers) G0 = kSmallContextWorkers;
    }
    if (units == 0) {
        // Behind the fused step only the DBSCAN workers are left: 512-THREAD blocks, as k_chain's -- a BallTree chain is ~40 us on
        // 512 threads against 45-60 on 256 (queries and the register sort network split two ways), and with the update gone nothing
        // in the launch wants two blocks per CU.  (For the contexts whose update runs here the same was measured and lost:
        // eight update waves that start and end together, profiles/NOTEBOOK.md round 5.)
        // (the 512-thread build wants one exchange slot per thread: the small clouds' carve-up holds at least 512 points, as k_chain's)
        const int umc5 = umc < kBigThreads ? kBigThreads : umc, cl5 = big_cl(umc5, cfg.t_cap, cfg.db_min_samples);
        const size_t c5 = chain_lds_bytes(umc5, cfg.t_cap, cfg.db_min_samples);
        if (c5 > lds) lds = c5;
        if (umb > 0 && umb < kBigThreads) {   // ... and so does the large clouds' (a context whose rings hold 257 .. 511 points)
            umb = kBigThreads;
            clb = big_cl(umb, cfg.t_cap, cfg.db_min_samples);
            const size_t b5 = big_lds_bytes(umb, clb, false);
            if (b5 > lds) lds = b5;
        }
        if (cfg.dx == 9) mmw_launch(k_post<9, 512>, dim3(G0), dim3(512), lds, stream, cfg, st, n_pts, nq, G0, umc5, cl5, umb, clb, UM, parity, epoch, labels, db_n);
        else mmw_launch(k_post<6, 512>, dim3(G0), dim3(512), lds, stream, cfg, st, n_pts, nq, G0, umc5, cl5, umb, clb, UM, parity, epoch, labels, db_n);
        return;
    }
    const int upd_blocks = (units + 3) / 4;
    const dim3 grid(G0 + 1 + upd_blocks);  // workers | the schedule sort | _update_all
    if (cfg.dx == 9) mmw_launch(k_post<9, 256>, grid, dim3(256), lds, stream, cfg, st, n_pts, nq, G0, umc, cl, umb, clb, UM, parity, epoch, labels, db_n);
    else mmw_launch(k_post<6, 256>, grid, dim3(256), lds, stream, cfg, st, n_pts, nq, G0, umc, cl, umb, clb, UM, parity, epoch, labels, db_n);
}

// The chain workers beside k_track and k_post (a second stream; see k_chain)
void launch_chain(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int epoch, int32_t *labels, int32_t *db_n, hipStream_t side)
{
    // (2 / 4 / 8 / 12 / 16 workgroups, now that the pair counts run in k_track and only BallTree chains arrive: 0.2025 / 0.195 / 0.190 /
    //  0.189 / 0.190 ms per step in a 100-step window, 0.210 / 0.203 / 0.194 / 0.192 / 0.192 in a 150-step one: profiles/NOTEBOOK.md)
#ifdef MMW_STAMPS
    static const int want = getenv("MMW_CHAIN_BLOCKS") ? atoi(getenv("MMW_CHAIN_BLOCKS")) : kChainBlocks;  // diagnostic build only
#else
    const int want = kChainBlocks;
#endif
    const int um = chain_um(UM, u_bound), cl = big_cl(um, cfg.t_cap, cfg.db_min_samples);
    const int g = want < cfg.n_scenes ? want : cfg.n_scenes;
    hipLaunchKernelGGL(k_chain, dim3(g > 0 ? g : 1), dim3(kBigThreads), chain_lds_bytes(um, cfg.t_cap, cfg.db_min_samples), side, cfg, st, um, cl, UM,
                       parity, epoch, labels, db_n);
}

// The larger clouds, what the side-stream workers have not taken: k_dbscan_big (k_dbscan_startup in the first frames
// after a reset: 512-point clouds at most, two workgroups per CU -- 3.0 instead of 4.0 ms for frame 0 of 4096 scenes).
void launch_dbscan_big(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int32_t *labels, int32_t *db_n, hipStream_t stream)
{
    const int um = big_um(UM, u_bound);
    if (kClassUM[0] >= um) return;  // no cloud can exceed the small class
    if (cfg.n_scenes <= kSmallContextScenes) return;  // k_post has taken them (launch_post)
    const int S = cfg.n_scenes, cl = big_cl(um, cfg.t_cap, cfg.db_min_samples);
    if (um <= kBigThreads && um < UM) {
        const size_t lds = big_lds_bytes(um, cl, true);
        int g = 256 * ((160u * 1024u) / lds >= 2 ? 2 : 1);
        if (g > S) g = S;
        mmw_launch(k_dbscan_startup, dim3(g), dim3(kBigThreads), lds, stream, cfg, st, um, cl, UM, parity, labels, db_n);
        return;
    }
    const int g = S < 256 ? S : 256;  // 176 VGPRs: one 512-thread workgroup per CU
    mmw_launch(k_dbscan_big, dim3(g), dim3(kBigThreads), big_lds_bytes(um, cl, false), stream, cfg, st, um, cl, UM, parity, labels, db_n);
}

void launch_dbscan_only(const DevCfg &cfg, const DevState &st, int UM, const double *pts, const int32_t *n, int max_n, double eps, int min_samples,
                        int32_t *labels, int32_t *ncl, hipStream_t stream)
{
    if (max_n > kClassUM[2]) {  // clouds the LDS cannot hold: slabs in global memory
        const int g = cfg.n_scenes < kHugeWorkers ? cfg.n_scenes : kHugeWorkers;
        hipLaunchKernelGGL(k_dbscan_only_huge, dim3(g), dim3(kHugeThreads), dbscan_huge_lds_bytes(UM, 0), stream, cfg, st, UM, cfg.n_scenes, pts, n, max_n,
                           eps, min_samples, labels, ncl);
        return;
    }
    const int umk = UM < kClassUM[2] ? UM : kClassUM[2];  // (a context whose rings hold more: the LDS classes stop at 1920)
    hipLaunchKernelGGL(k_dbscan_only, dim3(cfg.n_scenes), dim3(256), dbscan_only_lds_bytes(umk), stream, cfg, umk, pts, n, max_n, eps, min_samples,
                       labels, ncl);
}

// the clouds of more than 1920 points (work list 2): contexts whose rings can hold one
int dbscan_huge_workers(int n_scenes) { return n_scenes < kHugeWorkers ? n_scenes : kHugeWorkers; }
size_t dbscan_huge_slab_bytes(int UM, int t_cap, int min_samples)
{
    if (UM <= kClassUM[2]) return 0;
    const size_t a = db_hybrid_layout<false>(UM, big_cl(UM, t_cap, min_samples), nullptr, nullptr, nullptr, kHugeMW, nullptr);
    return (a + 255) & ~(size_t)255;
}
static size_t dbscan_huge_lds_bytes(int UM, int cl)
{
    size_t hot = 0;
    db_hybrid_layout<false>(UM, cl, nullptr, nullptr, nullptr, kHugeMW, &hot);
    return hot;
}
void launch_dbscan_huge(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int32_t *labels, int32_t *db_n, hipStream_t stream)
{
    if (UM <= kClassUM[2] || u_bound <= kClassUM[2]) return;  // no ring of this context can hold such a cloud (yet)
    const int cl = big_cl(UM, cfg.t_cap, cfg.db_min_samples);
    mmw_launch(k_dbscan_huge, dim3(dbscan_huge_workers(cfg.n_scenes)), dim3(kHugeThreads), dbscan_huge_lds_bytes(UM, cl), stream, cfg, st, UM, cl, UM, parity,
               labels, db_n);
}

}  // namespace mmw
