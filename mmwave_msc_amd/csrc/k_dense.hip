// k_dense.hip -- Dense-1 of the MARS CNN (train.py:49,87: Dense(512 k, relu) on the flattened conv output; 75 % of the CNN's
// multiply-adds) for the split-fp16 arithmetic of k_mars_conv16, as ONE kernel:
//
//     H = relu(bias + hi . W_hi + 2^-11 (hi . W_lo' + lo' . W_hi))
//
// with the activation split as the conv kernel leaves it (fp16 halves of every fp32 value, a = hi + 2^-11 lo') and the
// BN-folded weights split the same way.  Both operands are stored with the halves INTERLEAVED in runs of 32 values,
// [hi 0..31 | lo' 0..31 | hi 32..63 | lo' 32..63 | ...]: a K-step of 32 is then one 128-byte line per row.  Every partial product of two 11-bit significands is exact in the fp32 accumulator;
// the dropped lo'.lo' term is 2^-22 relative (mars.py).
//
// Why a kernel of our own: as two library GEMMs (hi.W_hi and [hi | lo'].[W_lo' ; W_hi]) the `hi` activations are fetched
// twice, the two products meet in a separate elementwise pass, and each GEMM is 1.7 waves of 256 x 256 tiles on 256 CUs.
// Here a K-step stages the four operand tiles ONCE and issues THREE matrix instructions per pair of fragments (main += hi.W_hi,
// corr += hi.W_lo', corr += lo'.W_hi): 1.5 MFMAs per fragment read from LDS instead of 0.5, half the activation traffic,
// bias + merge + ReLU in the epilogue.
//
// Geometry: workgroup = 512 threads = 8 waves, tiles of 256 or 128 batch rows x 192 or 128 features (k_mars_dense1_t below), K-step 32; a
// wave owns 64 rows x 32 NT features = 2 x NT tiles of v_mfma_f32_32x32x16_f16 with TWO accumulator sets (main, corr).  Operand tiles go
// global -> LDS by the DMA path (global_load_lds, 16 B per lane, no staging registers): A (rows x 128 B = hi | lo') and W (features x
// 128 B).  The LDS image of a tile is lane-linear (what the DMA writes), so the bank swizzle is applied to the SOURCE address: 16-byte
// unit c of row r lives in slot r * 8 + (c ^ ((r >> 1) & 7)) -- the 16 lanes the LDS serves together read 16 different bank groups.
// The tiles form rings in LDS (three A slots; three W slots, two for the 192-wide tile): one raw barrier per K-step with a counted
// vmcnt, two tiles' requests in flight across it.
// Rows are padded to the tile by the caller (mars.py allocates whole tiles; a pad row only feeds its own output row).
// Whole 128-byte lines per row (this layout) and an activation row stride that is not a multiple of 4 KB (mars.py) are worth 6 % and
// 4 %.  (One accumulator set -- main scaled by 2^11 through a third weight array -- allows 256 x 256 tiles: 0.97 instead of 1.05 ms, but
// three roundings of the accumulator per K-step instead of one made the keypoints less accurate than the fp32 path's on the stress
// inputs: measured, not kept.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <type_traits>

namespace mmw {

typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
typedef float df16 __attribute__((ext_vector_type(16)));

namespace dense {
constexpr int BM = 256, BN = 128, BK = 32, kThreads = 512;
constexpr int kRowBytes = BK * 2 * 2;              // 128 B of a tile row: hi (64 B) | lo' (64 B)
constexpr float kInvSplit = 1.0f / 2048.0f;

// 16-byte unit (row, c) of a tile image -> byte offset (c = 0..7: the row's eight units, 0..3 = hi, 4..7 = lo')
__device__ __forceinline__ int unit_off(int row, int c) { return (row * 8 + (c ^ ((row >> 1) & 7))) * 16; }

// An LDS-DMA request (global -> LDS, 16 B per lane, no staging registers) as ONE asm statement: m0 = the wave's destination, uniform base +
// 32-bit lane offset.  Why not __builtin_amdgcn_global_load_lds: with an LDS-DMA instruction in flight the compiler's wait counting gives
// up on the LDS counter -- every wait for a fragment becomes `s_waitcnt lgkmcnt(0)`, i.e. for ALL requested fragments, also those asked
// for on purpose a phase ahead.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_asm(const void *ubase, unsigned voff, void *lds_wave_base)
{
    const unsigned m = (unsigned)(unsigned long long)((__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ubase), "s"(m) : "memory", "m0");
}
#pragma clang diagnostic pop
}  // namespace dense

using namespace dense;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ONE kernel template for the three tile shapes (TBM batch rows x BN = WN * 32 * NT features; 8 waves as (8 / WN) x WN, a wave owns 64 rows
// x 32 NT features = 2 x NT MFMA tiles, two accumulator sets):
//   <256, 2, 3, 2>  256 x 192, feature counts that are a multiple of 192 (define_CNN_3D: 1536): three A slots and TWO W slots in 144 KB
//                   (W requested one K-step ahead; a third slot does not fit);
//   <256, 2, 2, 3>  256 x 128, the other feature counts (define_CNN: 512); three slots each, 144 KB;
//   <128, 4, 1, 3>  128 x 128, the tail of the tile list (a last wave of workgroups that fills less than half the chip runs as twice as
//                   many half tiles); 96 KB.
// ---- 256 x 192 tiles, for feature counts that are a multiple of 192 (define_CNN_3D: 1536): fewer bytes from L2 per
//      multiply-add (56 KB per K-step for 1.5 x the work of the 48 KB of a 256 x 128 tile) and a third fewer workgroups.  Two
//      accumulator sets of a 64 x 96 wave tile are 192 of the 256 registers a lane has at two waves per SIMD; the LDS holds a
//      ring of three A tiles (two in flight) and of two W tiles (one in flight) = 144 KB.  Same K order per output element as
//      the kernel above: bit-identical results.
//      SOFTWARE-PIPELINED over the four (half step, row block) phases of a K-step (round 6; before: every phase asked for its
//      fragments and then waited for them with nothing queued on the matrix pipe -- both waves of a SIMD are in the same phase,
//      they drained it together four times per K-step and once more behind the barrier: 64 % MFMA-busy).  Now
//        * the activation fragments of phase p + 1 are requested BEFORE the nine MFMAs of phase p (two fragment buffers), the weight
//          fragments of the next half step replace this one's as each feature block's last MFMAs have been issued;
//        * the barrier of a K-step sits in front of its LAST phase, whose operands are in registers: barrier skew and the first
//          reads of the next tile hide behind those MFMAs (a build without the barrier is not faster);
//        * the LDS-DMA requests are asm (glds16_asm), so the compiler's fragment waits are counted (lgkmcnt(6) / (4) / (2) ...);
//        * a piece's source is a uniform base + one of two per-lane offsets (4 VGPRs instead of seven pointers: 250 VGPRs, no scratch
//          -- a spilled build's reloads are `s_waitcnt vmcnt(0)` in the middle of the counted waits).
//      Measured (31 744 x 6144 x 1536, same box, alternating; profiles/r06e_*): 1.476 -> 1.322 ms, 64 -> 80 % MFMA-busy.  What is
//      left, by timing-only builds (MMW_DIAG_DENSE_NODMA / _NOLDS / _NOBAR: garbage results): MFMAs alone 0.79 ms, + barrier and
//      DMA 0.92, + fragment reads instead 1.02, everything 1.33 -- and GRBM_GUI_ACTIVE says the chip runs this kernel at 1.6 GHz
//      (the MFMA-only build at ~2.3): with all three units busy the clock, not a unit, is what gives.  Every request an L2 hit
//      (operands from a 1 MB footprint) was worth 5 %, weights requested two K-steps ahead instead of one 2.5 %, a wave that only
//      touches lines six steps ahead -5 % (slower): the fabric's latency is covered. ----
template <int TBM, int WN, int NT, int WS>
__global__ __launch_bounds__(kThreads, 1) void k_mars_dense1_t(const _Float16 *__restrict__ a2, long long lda, const _Float16 *__restrict__ w2,
                                                               long long ldw, const float *__restrict__ bias, float *__restrict__ out, int K, int N,
                                                               long long row0, int tiles_n)
{
    constexpr int BNT = WN * 32 * NT;                  // features of a tile
    constexpr int CA = TBM / 64, CW = BNT / 64;       // this wave's 1 KB pieces of an A tile / a W tile
    constexpr int kA = TBM * kRowBytes, kW = BNT * kRowBytes;
    constexpr int kWLead = WS == 3 ? 3 : 2;           // W(j + kWLead) is requested in stretch j (A: always j + 3)
    static_assert(TBM == 64 * (8 / WN) && (WS == 2 || WS == 3) && CA + CW <= 4 * NT, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: ring slots, tile origins and DMA destinations stay out of the VGPRs)
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt to the eight XCDs round-robin; a group of consecutive LOGICAL tiles (one band of
    // batch rows against all feature tiles) goes to one XCD, whose L2 then serves the band's activations to every tile of it
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const long long m0 = row0 + (long long)tm * TBM;
    const int n0 = tn * BNT;
    // this wave's pieces of a tile (1 KB = 8 rows x 128 B each): A pieces wave * CA + i, W pieces wave * CW + i.  A piece's source is a
    // uniform base + i * 8 rows + one of TWO per-lane offsets (the swizzle of rows 8 p + r depends on p's parity only): four VGPRs
    // instead of a 64-bit pointer per piece -- the fragment double buffer needs the registers
    const int rr = lane >> 3, c0 = (lane & 7) ^ (rr >> 1);
    const int ldA = (int)lda, ldW = (int)ldw;
    const int pbW = (wave * CW) & 1;   // parity of this wave's first W piece (CA is even: its first A piece is an even one)
    const unsigned vA0 = (unsigned)((rr * ldA + c0 * 8) * 2), vA1 = (unsigned)((rr * ldA + (c0 ^ 4) * 8) * 2);
    const unsigned vW0 = (unsigned)((rr * ldW + (c0 ^ (pbW * 4)) * 8) * 2), vW1 = (unsigned)((rr * ldW + (c0 ^ ((pbW ^ 1) * 4)) * 8) * 2);
    const char *const uA = reinterpret_cast<const char *>(a2 + (m0 + wave * (8 * CA)) * lda),
               *const uW = reinterpret_cast<const char *>(w2 + (long long)(n0 + wave * (8 * CW)) * ldw);
    char *const dA = lds + wave * CA * 1024, *const dW = lds + 3 * kA + wave * CW * 1024;
#ifdef MMW_DIAG_DENSE_NODMA   // (timing-only builds, scripts/ab_dense.sh: no requests / MMW_DIAG_DENSE_NOLDS: no fragment reads / _NOBAR: no barrier)
#define glds16_asm(a, b, c) ((void)0)
#endif
    auto pieceA = [&](int i, int kt, int slot) { glds16_asm(uA + ((long long)i * 16 * ldA + (long long)kt * (4 * BK)), (i & 1) ? vA1 : vA0, dA + slot * kA + i * 1024); };
    auto pieceW = [&](int i, int kt, int slot) { glds16_asm(uW + ((long long)i * 16 * ldW + (long long)kt * (4 * BK)), (i & 1) ? vW1 : vW0, dW + slot * kW + i * 1024); };
    const int offA0 = unit_off(wm * 64 + (lane & 31), lane >> 5), offW0 = unit_off(wn * (32 * NT) + (lane & 31), lane >> 5);
    df16 am[2][NT], ac[2][NT];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < NT; y++)
#pragma unroll
            for (int r = 0; r < 16; r++) { am[x][y][r] = 0.f; ac[x][y][r] = 0.f; }
    dh8 wh[NT], wl[NT], ab[2][2];   // weight fragments of the half step in hand; activation fragments [buffer][hi, lo']
#ifdef MMW_DIAG_DENSE_NOLDS
    for (int y = 0; y < NT; y++) { wh[y] = dh8{1, 1, 1, 1, 1, 1, 1, 1}; wl[y] = wh[y]; }
    ab[0][0] = ab[0][1] = ab[1][0] = ab[1][1] = wh[0];
    auto LA = [&](int, const char *, int, int) {};
    auto LW = [&](const char *, int, int) {};
#else
    // lane (row = l & 31, k-group g = l >> 5) reads unit c = 2 kk + g (hi) and 4 + 2 kk + g (lo'): one register per operand -- tile x / y is
    // 32 rows = 4096 bytes further, the second half step (kk = 1) flips bit 5 of the offset (unit c ^ 2), lo' bit 6
    auto LA = [&](int buf, const char *ba, int kk, int x) {
        ab[buf][0] = *reinterpret_cast<const dh8 *>(ba + ((offA0 ^ (kk * 32)) + x * 4096));
        ab[buf][1] = *reinterpret_cast<const dh8 *>(ba + ((offA0 ^ (kk * 32) ^ 64) + x * 4096));
    };
    auto LW = [&](const char *bw, int kk, int y) {
        wh[y] = *reinterpret_cast<const dh8 *>(bw + ((offW0 ^ (kk * 32)) + y * 4096));
        wl[y] = *reinterpret_cast<const dh8 *>(bw + ((offW0 ^ (kk * 32) ^ 64) + y * 4096));
    };
#endif
    // (corr first and last, main between them: the second corr product does not follow the first back to back)
#define MMW_MF(buf, x, y)                                                                               \
    do {                                                                                                \
        ac[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[buf][0], wl[y], ac[x][y], 0, 0, 0);       \
        am[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[buf][0], wh[y], am[x][y], 0, 0, 0);       \
        ac[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[buf][1], wh[y], ac[x][y], 0, 0, 0);       \
    } while (0)
    const int KT = K / BK;
    // requests in the order the counted waits rely on.  Two W slots:  W(0) A(0) | A(1) W(1) | A(2), then per stretch W(j + 2) A(j + 3);
    // three: W(0) A(0) | W(1) A(1) | W(2) A(2), then per stretch W(j + 3) A(j + 3)
    auto tileW = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < CW; i++) pieceW(i, kt, slot);
    };
    auto tileA = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < CA; i++) pieceA(i, kt, slot);
    };
    tileW(0, 0); tileA(0, 0);
    if (KT > 1) {
        if (WS == 3) { tileW(1, 1); tileA(1, 1); } else { tileA(1, 1); tileW(1, 1); }
    }
    if (KT > 2) {
        if (WS == 3) tileW(2, 2);
        tileA(2, 2);
    }
    if (KT > 2) wait_vmcnt<WS == 3 ? 2 * (CA + CW) : 2 * CA + CW>();
    else if (KT > 1) wait_vmcnt<CA + CW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int sa = 0, sw = 0;   // ring slots of the tile in hand
    {   // tile 0: everything but its last phase, fragments requested one phase ahead
        const char *ba = lds, *bw = lds + 3 * kA;
        LA(0, ba, 0, 0);
#pragma unroll
        for (int y = 0; y < NT; y++) LW(bw, 0, y);
        __builtin_amdgcn_sched_barrier(0);
        LA(1, ba, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < NT; y++) { MMW_MF(0, 0, y); __builtin_amdgcn_sched_barrier(0); }
        LA(0, ba, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < NT; y++) { MMW_MF(1, 1, y); __builtin_amdgcn_sched_barrier(0); LW(bw, 1, y); __builtin_amdgcn_sched_barrier(0); }
        LA(1, ba, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < NT; y++) { MMW_MF(0, 0, y); __builtin_amdgcn_sched_barrier(0); }
    }
    // One barrier-to-barrier stretch: the last phase of tile j, then -- if there is a tile j + 1 -- its first three.  MORE1..3 = "tiles
    // j + 1, j + 2, j + 3 exist" as COMPILE-TIME flags (the steady state is straight-line code; the last three stretches are
    // instantiated separately): at a join of two paths the compiler's wait counting falls back to lgkmcnt(0).
    // The stretch's requests -- CW pieces of W, then CA of A -- go one per feature block into its 4 NT gaps between MFMA groups.
    auto stretch = [&](auto M1, auto M2, auto M3, int j) {
        constexpr bool more1 = decltype(M1)::value, more2 = decltype(M2)::value, more3 = decltype(M3)::value;
        constexpr bool moreW = kWLead == 3 ? more3 : more2;
        // here: every read of tile j has been requested (its last phase's operands are in, or on their way to, registers)
        const int sa1 = sa + 1 == 3 ? 0 : sa + 1, sw1 = sw + 1 == WS ? 0 : sw + 1;   // slots of tile j + 1; tile j's own become those of A(j + 3), W(j + kWLead)
        const char *ba = lds + sa1 * kA, *bw = lds + 3 * kA + sw1 * kW;
        auto request = [&](int q) {   // q-th gap of the stretch (compile time)
            if (q < CW) { if constexpr (moreW) pieceW(q, j + kWLead, sw); }
            else if (q < CW + CA) { if constexpr (more3) pieceA(q - CW, j + 3, sa); }
        };
        if constexpr (more1) {
            if constexpr (more2) wait_vmcnt<WS == 3 ? CA + CW : CA>();   // tile j + 1 and everything before it; the requests of the stretch before stay in flight
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's reads of tile j have arrived, its buffers may be refilled (the
                                                  // builtin, not asm: the compiler's own wait counting must see it)
#ifndef MMW_DIAG_DENSE_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(1);
        // ---- last phase of tile j (registers only) | first reads of tile j + 1 | the stretch's first requests ----
        if constexpr (more1) LA(0, ba, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < NT; y++) {
            MMW_MF(1, 1, y);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more1) LW(bw, 0, y);
            request(y);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (more1) {
            // ---- tile j + 1, phases (0, 0), (0, 1), (1, 0) ----
            LA(1, ba, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < NT; y++) {
                MMW_MF(0, 0, y);
                __builtin_amdgcn_sched_barrier(0);
                request(NT + y);
                __builtin_amdgcn_sched_barrier(0);
            }
            LA(0, ba, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < NT; y++) {
                MMW_MF(1, 1, y);
                __builtin_amdgcn_sched_barrier(0);
                LW(bw, 1, y);
                request(2 * NT + y);
                __builtin_amdgcn_sched_barrier(0);
            }
            LA(1, ba, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < NT; y++) {
                MMW_MF(0, 0, y);
                __builtin_amdgcn_sched_barrier(0);
                request(3 * NT + y);
                __builtin_amdgcn_sched_barrier(0);
            }
            sa = sa1; sw = sw1;
        } else {
#pragma unroll
            for (int q = NT; q < CW + CA; q++) request(q);   // (nothing: no tile j + 3 without a tile j + 1)
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using T = std::true_type;
    using F = std::false_type;
    int j = 0;
    for (; j + 3 < KT; j++) stretch(T{}, T{}, T{}, j);
    if (j + 2 < KT) { stretch(T{}, T{}, F{}, j); j++; }
    if (j + 1 < KT) { stretch(T{}, F{}, F{}, j); j++; }
    stretch(F{}, F{}, F{}, j);
#undef MMW_MF
#ifdef MMW_DIAG_DENSE_NODMA
#undef glds16_asm
#endif
    // ---- epilogue: D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]; a register of the 64 lanes = two rows x 128 B ----
#pragma unroll
    for (int y = 0; y < NT; y++) {
        const int col = n0 + wn * (32 * NT) + y * 32 + (lane & 31);
        const float bv = bias[col];
#pragma unroll
        for (int x = 0; x < 2; x++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long row = m0 + wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = (am[x][y][r] + ac[x][y][r] * kInvSplit) + bv;
                out[row * N + col] = v > 0.f ? v : (v != v ? v : 0.f);   // relu that keeps NaN (as torch's)
            }
        }
    }
}

// rows_padded: a multiple of 256; K a multiple of 32; N a multiple of 128.
// The tile list is cut where its last wave of workgroups would fill less than half the chip: bands of 256 rows that make whole
// waves go to the 256-row instantiation, the rest -- as twice as many 128-row tiles -- to the other (18 304 rows x 1536: 864
// tiles = 3 waves + 96 tiles on 256 CUs; the 96 become 192 half tiles: 3.5 tile-times instead of 4).
constexpr int kLds192 = 3 * 256 * kRowBytes + 2 * 192 * kRowBytes;   // 144 KB: three A slots, two W slots
constexpr int kLds256 = 3 * (256 + 128) * kRowBytes;                  // 144 KB
constexpr int kLds128 = 3 * (128 + 128) * kRowBytes;                  // 96 KB
int launch_mars_dense1(const void *a2, long long lda, const void *w2, long long ldw, const float *bias, float *out, int rows_padded, int K, int N,
                       hipStream_t stream)
{
    // per DEVICE (a process may drive several GPUs, each from its own thread): the dynamic-LDS attribute of the three kernels
    // and the CU count behind the tile split, taken once on whichever thread gets there first
    struct DevPrep { std::once_flag once; int n_cu = 256; bool ok = false; };
    static DevPrep g_prep[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    DevPrep &P = g_prep[dev];
    std::call_once(P.once, [&]() {
        if (hipFuncSetAttribute((const void *)k_mars_dense1_t<256, 2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds192) != hipSuccess) return;
        if (hipFuncSetAttribute((const void *)k_mars_dense1_t<256, 2, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds256) != hipSuccess) return;
        if (hipFuncSetAttribute((const void *)k_mars_dense1_t<128, 4, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds128) != hipSuccess) return;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) P.n_cu = prop.multiProcessorCount;
        P.ok = true;
    });
    if (!P.ok) return -1;
    const int n_cu = P.n_cu;
    const int bands = rows_padded / BM, tiles_n = N / BN;
    const long long total = (long long)bands * tiles_n;
    int bands_main = bands;
    const long long rem = total % n_cu;
    if (rem != 0 && 2 * rem <= n_cu) {
        // whole waves of workgroups as 256-row tiles (cut at a band boundary), the remainder as 128-row tiles
        bands_main = (int)(((total / n_cu) * n_cu) / tiles_n);
        if (2LL * (bands - bands_main) * tiles_n > n_cu) bands_main = bands;   // (the remainder would be more than one wave of half tiles)
    }
    const _Float16 *A = reinterpret_cast<const _Float16 *>(a2), *W = reinterpret_cast<const _Float16 *>(w2);
    if (N % 192 == 0) {   // 256 x 192 tiles for the whole waves of workgroups, the rest as below
        const int tn2 = N / 192;
        const long long total2 = (long long)bands * tn2, rem2 = total2 % n_cu;
        int bm2 = bands;
        if (rem2 != 0 && 2 * rem2 <= n_cu) {   // the last wave of workgroups would fill less than half the chip: cut it off at a band boundary
            bm2 = (int)(((total2 / n_cu) * n_cu) / tn2);
            if (2LL * (bands - bm2) * tiles_n > n_cu) bm2 = bands;   // (... unless the rest is more than one wave of 128 x 128 half tiles)
        }
        if (bm2 > 0)
            hipLaunchKernelGGL((k_mars_dense1_t<256, 2, 3, 2>), dim3(bm2 * tn2), dim3(kThreads), kLds192, stream, A, lda, W, ldw, bias, out, K, N, 0LL, tn2);
        if (bm2 < bands)
            hipLaunchKernelGGL((k_mars_dense1_t<128, 4, 1, 3>), dim3(2 * (bands - bm2) * tiles_n), dim3(kThreads), kLds128, stream, A, lda, W, ldw, bias, out, K,
                               N, (long long)bm2 * BM, tiles_n);
        return 0;
    }
    if (bands_main > 0)
        hipLaunchKernelGGL((k_mars_dense1_t<256, 2, 2, 3>), dim3(bands_main * tiles_n), dim3(kThreads), kLds256, stream, A, lda, W, ldw, bias, out, K, N, 0LL,
                           tiles_n);
    if (bands_main < bands)
        hipLaunchKernelGGL((k_mars_dense1_t<128, 4, 1, 3>), dim3(2 * (bands - bands_main) * tiles_n), dim3(kThreads), kLds128, stream, A, lda, W, ldw, bias,
                           out, K, N, (long long)bands_main * BM, tiles_n);
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// The head of the MARS CNN for a SMALL batch -- one scene's tracks (the drop-in's estimate_posture, Tracking.py:705-734:
// at most a handful of rows): Dense-1 + ReLU and Dense-2 in Keras' own fp32 arithmetic, as two thin kernels.
// A tile kernel is the wrong shape there: 8 rows make one band of eight 256 x 192 tiles, i.e. eight workgroups stream the whole
// 37.7 MB weight matrix while 248 CUs idle (~100 us).  Here the matrix is cut along the FEATURES over the whole chip: a
// workgroup owns kHeadCols = 8 output columns, walks K with 16-byte loads (its 8 weight rows: 196 KB, each byte of the matrix read
// once, by one workgroup), multiplies every chunk into up to kHeadRows = 8 batch rows held in registers and reduces over the
// lanes at the end: bound by how fast 256 CUs can pull the weights (~10 us), not by one band's latency.
// Plain fp32 fused multiply-adds (what Keras' fp32 Dense does; no fp16 split, so no range word to look at); the summation
// order differs from the tile kernels' -- 1e-7 relative, inside the 1e-4 tolerance of the keypoints.
namespace head {
constexpr int kHeadCols = 8, kHeadRows = 8, kHeadThreads = 256;
}
template <int ROWS>
__global__ __launch_bounds__(head::kHeadThreads) void k_mars_head_dense1(const float *__restrict__ act, long long lda, const float *__restrict__ w,
                                                                         long long ldw, const float *__restrict__ bias, float *__restrict__ hidden,
                                                                         int n_rows, int K, int N, const int32_t *__restrict__ dev_rows)
{
    using namespace head;
    if (dev_rows != nullptr) {   // a row count only the device knows (the range fix-up): row tiles past it leave at once
        n_rows = dev_rows[0] < n_rows ? dev_rows[0] : n_rows;
        if ((int)blockIdx.y * ROWS >= n_rows) return;
    }
    __shared__ float red[kHeadThreads / 64][ROWS][kHeadCols];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * kHeadCols, r0 = blockIdx.y * ROWS;
    float acc[ROWS][kHeadCols];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < kHeadCols; c++) acc[r][c] = 0.f;
    const float4 *W[kHeadCols];
#pragma unroll
    for (int c = 0; c < kHeadCols; c++) W[c] = reinterpret_cast<const float4 *>(w + (long long)(c0 + c < N ? c0 + c : N - 1) * ldw);
    for (int k4 = tid; k4 < K / 4; k4 += kHeadThreads) {
        float4 wv[kHeadCols], av[ROWS];
#pragma unroll
        for (int c = 0; c < kHeadCols; c++) wv[c] = W[c][k4];
#pragma unroll
        for (int r = 0; r < ROWS; r++) av[r] = reinterpret_cast<const float4 *>(act + (long long)(r0 + r < n_rows ? r0 + r : n_rows - 1) * lda)[k4];
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
            for (int c = 0; c < kHeadCols; c++) {
                float a = acc[r][c];
                a = __builtin_fmaf(av[r].x, wv[c].x, a);
                a = __builtin_fmaf(av[r].y, wv[c].y, a);
                a = __builtin_fmaf(av[r].z, wv[c].z, a);
                a = __builtin_fmaf(av[r].w, wv[c].w, a);
                acc[r][c] = a;
            }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int c = 0; c < kHeadCols; c++) {
            float v = acc[r][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[wave][r][c] = v;
        }
    __syncthreads();
    if (tid < ROWS * kHeadCols) {
        const int r = tid / kHeadCols, c = tid - r * kHeadCols;
        if (r0 + r < n_rows && c0 + c < N) {
            float v = bias[c0 + c];
#pragma unroll
            for (int wv = 0; wv < kHeadThreads / 64; wv++) v += red[wv][r][c];
            hidden[(long long)(r0 + r) * N + c0 + c] = v > 0.f ? v : (v != v ? v : 0.f);   // ReLU (NaN stays NaN, as Keras')
        }
    }
}

// Dense-2 (BatchNormalization folded in, as mars.py lays it out): kp[b][j] = bias2[j] + sum_k hidden[b][k] w2[j][k].  A WAVE per
// output (b, j), lanes along K with every load of the wave in flight at once (K = 1536: 24 per lane, three batches of eight): the
// 57 x n_rows dot products are independent, and as a loop of one workgroup per row they were 100 us of dependent round trips.
__global__ __launch_bounds__(256) void k_mars_head_dense2(const float *__restrict__ hidden, const float *__restrict__ w2, const float *__restrict__ bias2,
                                                          float *__restrict__ kp, int K, int NOUT, int n_rows, const int32_t *__restrict__ dev_rows)
{
    if (dev_rows != nullptr) n_rows = dev_rows[0] < n_rows ? dev_rows[0] : n_rows;
    const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_rows * NOUT) return;   // (wave-uniform)
    const int b = item / NOUT, j = item - b * NOUT;
    const float *h = hidden + (long long)b * K, *wr = w2 + (long long)j * K;
    float a = 0.f;
    int k = lane;
    for (; k + 7 * 64 < K; k += 8 * 64) {
        float hv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { hv[u] = h[k + u * 64]; wv[u] = wr[k + u * 64]; }
#pragma unroll
        for (int u = 0; u < 8; u++) a = __builtin_fmaf(hv[u], wv[u], a);
    }
    for (; k < K; k += 64) a = __builtin_fmaf(h[k], wr[k], a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) kp[(long long)b * NOUT + j] = a + bias2[j];
}

int launch_mars_head_small(const float *act, long long lda, const float *w1, long long ldw, const float *bias1, const float *w2, const float *bias2,
                           float *hidden, float *kp, int n_rows, int K, int N1, int NOUT, hipStream_t stream, const int32_t *dev_rows)
{
    using namespace head;
    const dim3 grid((N1 + kHeadCols - 1) / kHeadCols, (n_rows + kHeadRows - 1) / kHeadRows);
    if (n_rows <= 2)
        hipLaunchKernelGGL(k_mars_head_dense1<2>, dim3(grid.x, (n_rows + 1) / 2), dim3(kHeadThreads), 0, stream, act, lda, w1, ldw, bias1, hidden, n_rows, K, N1, dev_rows);
    else if (n_rows <= 4)
        hipLaunchKernelGGL(k_mars_head_dense1<4>, dim3(grid.x, (n_rows + 3) / 4), dim3(kHeadThreads), 0, stream, act, lda, w1, ldw, bias1, hidden, n_rows, K, N1, dev_rows);
    else
        hipLaunchKernelGGL(k_mars_head_dense1<kHeadRows>, grid, dim3(kHeadThreads), 0, stream, act, lda, w1, ldw, bias1, hidden, n_rows, K, N1, dev_rows);
    hipLaunchKernelGGL(k_mars_head_dense2, dim3((n_rows * NOUT + 3) / 4), dim3(256), 0, stream, hidden, w2, bias2, kp, N1, NOUT, n_rows, dev_rows);
    return 0;
}

}  // namespace mmw
