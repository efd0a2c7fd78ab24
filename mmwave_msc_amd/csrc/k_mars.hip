// k_mars.hip -- the two Conv3D layers of the MARS regressor (reference src/train.py:73-82,
// define_CNN_3D: Conv3D(16,3x3x3,same,relu) -> Conv3D(32,3x3x3,same,relu)) fused in one kernel on
// the fp32 matrix cores (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32: exact f32 fma chains).
//
// Per sample the work is two implicit GEMMs over a zero-padded 5x10x10 volume kept in LDS:
//   conv1: [192 positions] x [K = 27 taps x 5 ch = 135] x [16 out]      0.83 MFLOP
//   conv2: [192 positions] x [K = 27 taps x 16 ch = 432] x [32 out]     5.31 MFLOP
// A workgroup is two waves sharing one sample (each takes half of the position tiles) and loops
// over samples; ALL weights live in registers as MFMA B operands for the whole kernel (conv2:
// 216 VGPRs per lane), the A operands are single ds_read_b32 per MFMA from the padded volume
// (tap and channel offsets are immediates), the intermediate activation never leaves LDS.
// Input = mmw_features' channels-last tensor; output [B][192][32] is the Keras Flatten order
// (d,h,w,c), so Dense-1 takes Keras' weight rows as they are.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

namespace mmw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPV = 500;      // padded volume 5 x 10 x 10 per channel
constexpr int kK1 = 34;       // conv1 k-steps of 4 (135 -> 136)
constexpr int kK2 = 216;      // conv2 k-steps of 2 (432)

__device__ __forceinline__ int padded_origin(int pos)  // position (d,h,w) -> index of its (kd,kh,kw)=(0,0,0) tap
{
    const int d = pos >> 6, h = (pos >> 3) & 7, w = pos & 7;
    return d * 100 + h * 10 + w;
}

constexpr float kSplitScale = 2048.0f;  // 2^11 (see k_mars_conv16 below)
// SPLIT = workgroups per sample.  1: a workgroup takes whole samples, one after the other (the throughput form: thousands of
// samples).  3: the form for a handful of samples (one scene's tracks, MarsCNN.forward_small) -- there the launch is ONE
// sample's latency, 18 us of which are the 216 x 3 conv2 matrix instructions a wave issues for its three position tiles:
// three workgroups share a sample, each computes conv1 for the whole volume (0.83 of 6.1 MFLOP, redundantly) and ONE conv2
// tile per wave.
template <int SPLIT>
__global__ __launch_bounds__(128, 1) void k_mars_conv(const float *__restrict__ feat, const float *__restrict__ w1,
                                                       const float *__restrict__ b1, const float *__restrict__ w2,
                                                       const float *__restrict__ b2, float *__restrict__ out, int B, const int32_t *__restrict__ dev_rows)
{
    // (dev_rows: a row count that only the device knows -- the range fix-up's flagged samples: workgroups past it leave at once)
    if (dev_rows != nullptr && (SPLIT == 1 ? (int)blockIdx.x : (int)blockIdx.x / SPLIT) >= dev_rows[0]) return;
    __shared__ float Xp[5 * kPV];    // input, channel-major, zero border
    __shared__ float H1[16 * kPV];   // relu(conv1), channel-major, zero border
    __shared__ float W1s[136 * 16];  // conv1 kernel [k = tap*5+ic][oc], row 135 = zero padding of K
    __shared__ int A1s[136];         // LDS offset of tap/channel k relative to a position's origin
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < 5 * kPV; i += 128) Xp[i] = 0.f;
    for (int i = tid; i < 16 * kPV; i += 128) H1[i] = 0.f;

    // ---- weights as MFMA B operands, resident for the whole kernel ----
    // conv2 (32x32x2): lane l holds B[k = 2*ks + (l>>5)][oc = l&31], k = tap*16 + ic  (Keras kernel (kd,kh,kw,in,out))
    float w2r[kK2];
#pragma unroll
    for (int ks = 0; ks < kK2; ks++) w2r[ks] = w2[(2 * ks + (lane >> 5)) * 32 + (lane & 31)];
    // conv1 (16x16x4): lane l needs B[k = 4*ks + (l>>4)][oc = l&15], k = tap*5 + ic (k = 135 pads K);
    // its weights and offset table stay in LDS -- the registers belong to conv2
    for (int i = tid; i < 136 * 16; i += 128) W1s[i] = i < 135 * 16 ? w1[i] : 0.f;
    for (int k = tid; k < 136; k += 128) {
        const bool real = k < 135;
        const int tap = real ? k / 5 : 0, ic = real ? k - 5 * tap : 0;
        A1s[k] = ic * kPV + (tap / 9) * 100 + ((tap / 3) % 3) * 10 + (tap % 3);
    }
    const float bias1 = b1[lane & 15], bias2 = b2[lane & 31];
    // conv2 A operand: lane l reads channel (2*ks)%16 + (l>>5) at position (tile*32 + (l&31)) shifted by the tap
    constexpr int TPW = 3 / SPLIT;                        // conv2 tiles (32 positions) per wave
    const int part = SPLIT == 1 ? 0 : blockIdx.x % SPLIT;   // which third of the sample's conv2 tiles
    const int tile0 = SPLIT == 1 ? wave * 3 : part * 2 + wave;
    int a2base[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) a2base[t] = (lane >> 5) * kPV + padded_origin((tile0 + t) * 32 + (lane & 31));
    __syncthreads();

    for (int b = SPLIT == 1 ? blockIdx.x : blockIdx.x / SPLIT; b < B; b += SPLIT == 1 ? gridDim.x : B) {
        // ---- input sample (channels-last [192][5]) into the padded volume ----
        const float *x = feat + (size_t)b * 960;
        for (int e = tid; e < 960; e += 128) {
            const int pos = e / 5, c = e - pos * 5;
            Xp[c * kPV + padded_origin(pos) + 111] = x[e];  // +111: interior starts at (1,1,1)
        }
        __syncthreads();
        // ---- conv1 + bias + relu -> H1 : 12 tiles of 16 positions, 6 per wave ----
#pragma unroll 1
        for (int t = 0; t < 6; t++) {
            const int mt = wave * 6 + t;
            const int aorg = padded_origin(mt * 16 + (lane & 15));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < kK1; ks++) {
                const int k = 4 * ks + (lane >> 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Xp[A1s[k] + aorg], W1s[k * 16 + (lane & 15)], acc, 0, 0, 0);
            }
            // C/D: col = lane&15 (out channel), row = (lane>>4)*4 + reg (position in tile)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pos = mt * 16 + (lane >> 4) * 4 + r;
                const float v = acc[r] + bias1;
                H1[(lane & 15) * kPV + padded_origin(pos) + 111] = v > 0.f ? v : 0.f;
            }
        }
        __syncthreads();
        // ---- conv2 + bias + relu -> out[b][pos][oc] : 6 tiles of 32 positions, 3 per wave ----
        f32x16 cc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) cc[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < kK2; ks++) {
            const int tap = (2 * ks) / 16, ic0 = (2 * ks) % 16;
            const int koff = ic0 * kPV + (tap / 9) * 100 + ((tap / 3) % 3) * 10 + (tap % 3);  // compile-time constant
#pragma unroll
            for (int t = 0; t < TPW; t++) cc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(H1[a2base[t] + koff], w2r[ks], cc[t], 0, 0, 0);
        }
        // C/D: col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        float *o = out + (size_t)b * 192 * 32;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oc = lane & 31;
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                const float v = cc[t][r] + bias2;
                o[((tile0 + t) * 32 + row) * 32 + oc] = v > 0.f ? v : 0.f;
            }
        }
        __syncthreads();  // both waves are done reading Xp / H1 before the next sample overwrites them
    }
}

void launch_mars_conv(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, float *out, int B,
                      hipStream_t stream, const int32_t *dev_rows)
{
    if (B <= 0) return;
    if (B <= 64) {   // a handful of samples: three workgroups per sample (one conv2 tile per wave), the launch is one sample's latency
        hipLaunchKernelGGL(k_mars_conv<3>, dim3(B * 3), dim3(128), 0, stream, feat, w1, b1, w2, b2, out, B, dev_rows);
        return;
    }
    const int grid = B < 512 ? B : 512;  // 2 workgroups per CU, persistent over samples
    hipLaunchKernelGGL(k_mars_conv<1>, dim3(grid), dim3(128), 0, stream, feat, w1, b1, w2, b2, out, B, (const int32_t *)nullptr);
}
// ================================================================================================================
// k_mars_conv16 -- the same two layers on the fp16 matrix cores, fp32-exact by operand splitting (the scheme of
// Dense-1, mars.py): every fp32 value a is carried as hi = fp16(a) and lo' = fp16((a - hi) * 2^11), and a product
// a.w is accumulated in fp32 as hi.hi + 2^-11 (hi.lo' + lo'.hi) -- each partial product exact, the dropped lo'.lo'
// term 2^-22 relative.  Against the fp32-MFMA kernel above: v_mfma_f32_32x32x16_f16 does in 32 cycles eight times the
// K of v_mfma_f32_32x32x2_f32 in 64, so three partial products cost 3/16 of the fp32 matrix-core time.
//
// Layout of the work (NZ = 3: Conv3D pair of define_CNN_3D, train.py:73-82; NZ = 1: Conv2D pair of define_CNN,
// train.py:35-44):
//  * ONE WAVE = ONE SAMPLE, four independent waves per workgroup, no block barrier inside the sample loop; the next
//    sample's input is fetched into registers while the current one is convolved.
//  * Activations sit in LDS channels-LAST as fp16 pairs (hi, lo'): X8[pos][8] (5 channels + 3 zeros) and
//    H1[pos][16], positions padded in y and x only (10 x 10 per plane).  A K-step of the MFMA is then whole taps:
//    conv2 (32x32x16): one tap x 16 input channels, one ds_read_b128 per lane and operand half;
//    conv1 (16x16x32): four taps x 8 channels.  Taps that would read a z-plane outside the volume contribute nothing
//    and are skipped (conv2: the tile's plane is uniform) or fed from a zero slot (conv1: per lane).
//  * The matrix roles are swapped (weights = A, activations = B), so a lane's accumulator registers are consecutive
//    OUTPUT CHANNELS of one position: four of them pack into one 8-byte LDS write of the channels-last layout.
//  * conv2's weights stay in registers for the whole kernel as fp16 (hi, lo') fragments (216 VGPRs, as the fp32 kernel);
//    conv1's fragments are shared through LDS.
// Output: out16[B][2][NZ*2048] fp16 = [hi | lo'] of relu(conv2), in Keras' Flatten order (d,h,w,c).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int NZ>
struct Conv16 {
    static constexpr int kTaps = NZ == 3 ? 27 : 9;
    static constexpr int kPos = NZ * 64;             // output positions per sample
    static constexpr int kPad = NZ * 100;            // padded positions (10 x 10 per plane)
    static constexpr int kS1 = (kTaps + 3) / 4;      // conv1 K-steps: four taps x 8 channels each
    // X8 hi/lo, H1 hi/lo (+ zero slot each).  The 4 KiB staging tile of the conv2 epilogue overlays X8 when three planes
    // make X8 large enough (and LDS scarce); the cells it dirties are zeroed again before the next sample's input is placed
    static constexpr bool kStageOverX = NZ == 3;
    static constexpr int kStage = 2 * 32 * 32 * 2;
    static constexpr int kWaveLds = (kPad + 1) * 16 * 2 + (kPad + 1) * 32 * 2 + (kStageOverX ? 0 : kStage);
    static constexpr int kW1Lds = kS1 * 64 * 16 * 2;
    static constexpr int kW2Lds = kTaps * 64 * 16;   // the lo' halves of conv2's weight fragments (the hi halves stay in registers)
    static constexpr int kLds = 4 * kWaveLds + kW1Lds + kW2Lds;
    static_assert(!kStageOverX || (kPad + 1) * 16 * 2 >= kStage, "staging tile fits the X8 region");
    __device__ static __forceinline__ int padded(int p) { return (p >> 6) * 100 + (((p >> 3) & 7) + 1) * 10 + (p & 7) + 1; }
    __device__ static __forceinline__ int tap_off(int tap)  // relative to the centre tap
    {
        const int kd = NZ == 3 ? tap / 9 : 1, r = NZ == 3 ? tap % 9 : tap;
        return (kd - 1) * 100 + (r / 3 - 1) * 10 + (r % 3 - 1);
    }
    __device__ static __forceinline__ int tap_kd(int tap) { return NZ == 3 ? tap / 9 : 1; }
};

__device__ __forceinline__ void split16(float a, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)a;
    lo = (_Float16)((a - (float)hi) * kSplitScale);
}
// The split holds a value exactly only inside fp16's range (|a| < 65 504; beyond it hi is inf and the sample's keypoints come
// out inf / NaN or, through a ReLU, as wrong finite numbers): the kernel reports it (range_flag).  Inputs are tested one by
// one (NaN and inf included); activations -- finite fp32 values as long as the inputs were -- by a running maximum of their
// magnitudes, one instruction each in epilogues that are VALU-bound.
__device__ __forceinline__ void split16(float a, _Float16 &hi, _Float16 &lo, bool &over)
{
    over |= !(fabsf(a) < 65504.0f);
    split16(a, hi, lo);
}
__device__ __forceinline__ void split16(float a, _Float16 &hi, _Float16 &lo, float &amax)
{
    amax = fmaxf(amax, fabsf(a));
    split16(a, hi, lo);
}

#ifdef MMW_STAMPS
__device__ unsigned long long g_conv_stamps[8];
#define CSTAMP(k)                                                                  \
    do {                                                                           \
        if (blockIdx.x == 0 && tid == 0) {                                         \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();        \
            g_conv_stamps[k] += t_now - t_prev;                                    \
            t_prev = t_now;                                                        \
        }                                                                          \
    } while (0)
#else
#define CSTAMP(k)
#endif

constexpr int kRangeFixCap = 64;   // = MMW_RANGE_FIXUP_CAP (include/mmw.h)
template <int NZ>
__global__ __launch_bounds__(256, 1) void k_mars_conv16(const float *__restrict__ feat, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        const float *__restrict__ b2, _Float16 *__restrict__ out, long long ld_out, int B,
                                                        int32_t *__restrict__ range_flag, int32_t *__restrict__ sample_flags)
{
    using C = Conv16<NZ>;
    bool any_over = false;   // some sample of this wave left fp16's range (range_flag; per sample: sample_flags)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the wave's LDS region and sample index stay out of the VGPRs)
    // ---- LDS carve-up ----
    h8 *W1hi = reinterpret_cast<h8 *>(lds_raw);                 // [kS1][64] conv1 A fragments (shared)
    h8 *W1lo = W1hi + C::kS1 * 64;
    h8 *W2lo = W1lo + C::kS1 * 64;                              // [kTaps][64] lo' halves of conv2's A fragments (shared)
    char *mine = lds_raw + C::kW1Lds + C::kW2Lds + wave * C::kWaveLds;
    constexpr int CE = C::kPad + 1;                             // cells per array (entry kPad = zero slot)
    h8 *Xhi = reinterpret_cast<h8 *>(mine);                     // [CE]  8 channels (5 + 3 zeros) per cell
    h8 *Xlo = Xhi + CE;
    // H1: 16 channels per cell as TWO PLANES of 8 (channels 0..7 | 8..15), 16 bytes per cell and plane: the 16 lanes the
    // LDS serves together read 16 different cells of one plane, i.e. 16 different 16-byte bank groups (see the tiles below)
    h8 *Hhi = Xlo + CE;                                         // [2][CE]
    h8 *Hlo = Hhi + 2 * CE;                                     // [2][CE]
    // [2][32 pos][32 oc]; over X8 (dead once conv1 has run) or behind H1
    _Float16 *stage = C::kStageOverX ? reinterpret_cast<_Float16 *>(Xhi) : reinterpret_cast<_Float16 *>(Hlo + 2 * CE);
    {   // zero this wave's volumes once: borders and zero slots stay zero, interiors are rewritten per sample
        uint4 *z = reinterpret_cast<uint4 *>(mine);
        for (int i = lane; i < C::kWaveLds / 16; i += 64) z[i] = uint4{0, 0, 0, 0};
    }
    // ---- conv1 fragments (A = weights, 16x16x32): lane (oc = l & 15, g = l >> 4) of step s holds w1[tap 4s+g][ic j][oc] ----
    for (int i = tid; i < C::kS1 * 64; i += 256) {
        const int s = i >> 6, l = i & 63, oc = l & 15, tap = 4 * s + (l >> 4);
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = (tap < C::kTaps && j < 5) ? w1[(tap * 5 + j) * 16 + oc] : 0.f;
            _Float16 a, b;
            split16(v, a, b);
            hi[j] = a; lo[j] = b;
        }
        W1hi[i] = hi; W1lo[i] = lo;
    }
    // ---- conv2 fragments (A = weights, 32x32x16): lane (oc = l & 31, h = l >> 5) of tap t holds w2[t][ic 8h+j][oc] ----
    // (the hi halves stay in registers for the whole kernel; the lo' halves are shared through LDS: with both in registers
    //  the 256 architectural VGPRs are full and the compiler cannot keep a tap's operands in flight behind the MFMAs)
    h8 w2hi[C::kTaps];
#pragma unroll
    for (int t = 0; t < C::kTaps; t++) {
        h8 lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = w2[((t * 16) + 8 * (lane >> 5) + j) * 32 + (lane & 31)];
            _Float16 a, b;
            split16(v, a, b);
            w2hi[t][j] = a; lo[j] = b;
        }
        if (wave == 0) W2lo[t * 64 + lane] = lo;
    }
    // conv1 D: row = oc = (lane >> 4) * 4 + reg, col = position;  conv2 D: row = oc = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float bias1[4], bias2[16];
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = b1[(lane >> 4) * 4 + r];
#pragma unroll
    for (int r = 0; r < 16; r++) bias2[r] = b2[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    __syncthreads();  // W1 / W2lo fragments visible; the only block barrier of the kernel

    // ---- tiles.  The LDS serves a 16-byte-per-lane read 16 lanes at a time, one 16-byte bank group each: the 16 cells
    //      those lanes address must differ modulo 16.  Rows are 10 cells apart, so a tile pairs rows FOUR apart
    //      (40 = 8 mod 16): 8 + 8 cells that tile the residues exactly, wherever the tap shifts them.
    //      conv1 tile (16 positions) c of plane d: rows c and c + 4;  conv2 tile (32 positions) u of plane d: rows
    //      2u, 2u + 4, 2u + 1, 2u + 5.  Column r of the MFMA's B operand = position (row(r >> 3), x = r & 7). ----
    const int x8 = lane & 7;
    const int cell1 = ((lane >> 3) & 1) * 40 + 10 + x8 + 1;                       // conv1: + (d * 100 + c * 10)
    // conv2: the sixteen lanes the LDS serves together are NOT consecutive ones -- of each 32 they are {0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31} (measured: with consecutive sixteens every operand read was a 2-way bank conflict, 52 % of the
    // kernel's LDS cycles).  col2 = the tile column this lane's MFMA column stands for: those two lane sets in order.
    const int l32 = lane & 31;
    const int col2 = l32 < 4 ? l32 : l32 < 12 ? l32 + 12 : l32 < 16 ? l32 - 8 : l32 < 20 ? l32 + 8 : l32 < 28 ? l32 - 12 : l32;
    const int cell2 = (((col2 >> 3) & 1) * 4 + ((col2 >> 4) & 1)) * 10 + 10 + (col2 & 7) + 1;  // + (d * 100 + 2u * 10)
    const int h2 = lane >> 5;

    const int stride = gridDim.x * 4;
    int b = blockIdx.x * 4 + wave;
#ifdef MMW_STAMPS
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 0) g_conv_stamps[7] += 1;
#endif
    // input of the first sample: lane owns positions lane, lane + 64, (lane + 128): 5 floats each
    constexpr int PPL = C::kPos / 64;
    float xin[PPL][5];
    if (b < B) {
#pragma unroll
        for (int q = 0; q < PPL; q++)
#pragma unroll
            for (int c = 0; c < 5; c++) xin[q][c] = feat[(size_t)b * C::kPos * 5 + (q * 64 + lane) * 5 + c];
    }
    for (; b < B; b += stride) {
        bool over = false;   // an input of THIS sample outside fp16's range was split ...
        float amax = 0.f;    // ... largest activation magnitude of this sample split so far
        // ---- this sample's input into the padded channels-last volume, split ----
        if (C::kStageOverX) {  // the previous sample's staging tile lay over the first cells of X8: borders back to zero
            uint4 *z = reinterpret_cast<uint4 *>(Xhi);
#pragma unroll
            for (int i = 0; i < C::kStage / 16 / 64; i++) z[i * 64 + lane] = uint4{0, 0, 0, 0};
        }
#ifndef MMW_DIAG_CONV_NOINPUT
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 5; c++) { _Float16 a, l2; split16(xin[q][c], a, l2, over); hi[c] = a; lo[c] = l2; }
            const int pp = C::padded(q * 64 + lane);
            Xhi[pp] = hi; Xlo[pp] = lo;
        }
        // ---- prefetch the next sample (consumed at the top of the next iteration) ----
        if (b + stride < B) {
#pragma unroll
            for (int q = 0; q < PPL; q++)
#pragma unroll
                for (int c = 0; c < 5; c++) xin[q][c] = feat[(size_t)(b + stride) * C::kPos * 5 + (q * 64 + lane) * 5 + c];
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        CSTAMP(1);  // input staging
        // ---- conv1: the four 16-position tiles of a plane at a time (they share the weight fragments); D[oc][pos] ----
#ifndef MMW_DIAG_CONV_NOCONV1
#pragma unroll
        for (int d = 0; d < NZ; d++) {
            int pc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pc[u] = d * 100 + u * 10 + cell1;
            f32x4 am[4], ac[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { am[u] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[u] = am[u]; }
            // A K-step whose four taps all read a plane outside the volume is skipped (d and s are compile-time: 17 of the
            // 21 steps remain); the operands of the next step are requested before the twelve MFMAs of this one are issued --
            // the loop is bound by the LDS (10 KiB per wave and step), which must not also wait for the matrix pipe.
            auto step_valid = [&](int s2) {
                bool v = false;
#pragma unroll
                for (int g2 = 0; g2 < 4; g2++) {
                    const int t2 = 4 * s2 + g2;
                    v = v || (t2 < C::kTaps && (unsigned)(d + C::tap_kd(t2) - 1) < (unsigned)NZ);
                }
                return v;
            };
            h8 wh, wl, xh[4], xl[4];
            auto load_step = [&](int s2, h8 &wh2, h8 &wl2, h8 (&xh2)[4], h8 (&xl2)[4]) {
                const int g = lane >> 4, tap = 4 * s2 + g;
                const int kd = NZ == 3 ? tap / 9 : 1;
                const bool ok = tap < C::kTaps && (unsigned)(d + kd - 1) < (unsigned)NZ;
                // tap offset of this lane's tap (the four candidates of the step are compile-time constants)
                const int o0 = C::tap_off(4 * s2 + 0 < C::kTaps ? 4 * s2 + 0 : 0), o1 = C::tap_off(4 * s2 + 1 < C::kTaps ? 4 * s2 + 1 : 0),
                          o2 = C::tap_off(4 * s2 + 2 < C::kTaps ? 4 * s2 + 2 : 0), o3 = C::tap_off(4 * s2 + 3 < C::kTaps ? 4 * s2 + 3 : 0);
                const int off = g == 0 ? o0 : g == 1 ? o1 : g == 2 ? o2 : o3;
                wh2 = W1hi[s2 * 64 + lane]; wl2 = W1lo[s2 * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = ok ? pc[u] + off : C::kPad; xh2[u] = Xhi[i]; xl2[u] = Xlo[i]; }
            };
            {
                int s0 = 0;
#pragma unroll
                for (int ss = C::kS1 - 1; ss >= 0; ss--) if (step_valid(ss)) s0 = ss;
                load_step(s0, wh, wl, xh, xl);
            }
#pragma unroll
            for (int s = 0; s < C::kS1; s++) {
                if (!step_valid(s)) continue;   // compile-time
                int nxt = -1;
#pragma unroll
                for (int ss = C::kS1 - 1; ss > s; ss--) if (step_valid(ss)) nxt = ss;
                h8 nwh = wh, nwl = wl, nxh[4], nxl[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { nxh[u] = xh[u]; nxl[u] = xl[u]; }
                if (nxt >= 0) load_step(nxt, nwh, nwl, nxh, nxl);
                __builtin_amdgcn_sched_barrier(0);
#ifdef MMW_DIAG_CONV_NOMFMA1
                asm volatile("" ::"v"(wh), "v"(wl), "v"(xh[0]), "v"(xh[1]), "v"(xh[2]), "v"(xh[3]), "v"(xl[0]), "v"(xl[1]), "v"(xl[2]), "v"(xl[3]));
#else
#pragma unroll
                for (int u = 0; u < 4; u++) am[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[u], am[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++) ac[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[u], ac[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++) ac[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[u], ac[u], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                wh = nwh; wl = nwl;
#pragma unroll
                for (int u = 0; u < 4; u++) { xh[u] = nxh[u]; xl[u] = nxl[u]; }
            }
#ifdef MMW_DIAG_CONV_NOEPI1
            {
                float keep = 0.f;
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) keep += am[u][r] + ac[u][r];
                if (keep == 12345.678f) out[lane] = (_Float16)keep;
            }
#else
#pragma unroll
            for (int u = 0; u < 4; u++) {
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = (am[u][r] + ac[u][r] * (1.0f / kSplitScale)) + bias1[r];
                    v = v > 0.f ? v : 0.f;
                    _Float16 a, l2;
                    split16(v, a, l2, amax);
                    hi[r] = a; lo[r] = l2;
                }
                // channels 4g .. 4g+3 of the position: plane g >> 1, half (g & 1) of its 16-byte cell
                const int g = lane >> 4;
                reinterpret_cast<h4 *>(Hhi + (g >> 1) * CE + pc[u])[g & 1] = hi;
                reinterpret_cast<h4 *>(Hlo + (g >> 1) * CE + pc[u])[g & 1] = lo;
            }
#endif
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        CSTAMP(2);  // conv1
        // ---- conv2: the two 32-position tiles of a plane at a time (the plane is uniform: taps that read a plane outside
        //      the volume are skipped at compile time); D[oc][pos].
        //      The EPILOGUE of plane d - 1 (bias, relu, split, staging tile, stores: VALU work, a quarter of the kernel when it ran
        //      between the planes with the matrix pipe idle -- one wave per SIMD, nobody else to fill it) is cut into sixteen chunks
        //      and issued BETWEEN the taps of plane d, behind each tap's six MFMAs: two accumulator sets, the registers were
        //      there.  Only the last plane's epilogue runs alone (its staging tile lies over X8, which the next sample's
        //      input wants). ----
        _Float16 *o = out + (size_t)b * ld_out;
        const h8 *Hh = Hhi + h2 * CE, *Hl = Hlo + h2 * CE;   // this lane's channel plane
        f32x16 acc[2][4];                                     // [set = plane & 1][am0, ac0, am1, ac1]
        // chunk k of the epilogue of plane dp (accumulators S), sixteen of them: 0..3 = tile 0's four channel groups (bias, relu, split)
        // -> staging tile [column r][oc]; 4..7 = tile 0's four output stores, each fed by ONE 16-byte read of the staging tile that is
        // requested in FRONT of the tap's MFMAs (epi_pre) and stored behind them (epi_post): no wait for the LDS; 8..11, 12..15 = tile 1
        uint4 q4 = uint4{0, 0, 0, 0};
        const int st_half = (lane >> 2) & 1, st_chunk = lane & 3;   // a store instruction writes whole 128-byte lines: for each of its two rows, four positions x [hi 64 B | lo' 64 B]
        auto epi_pre = [&](int k) {
            if ((k & 7) < 4) return;
            if ((k & 7) == 4) {   // the tile's four groups are staged
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const int j = (k & 7) - 4, ps = j >> 1, i = j & 1;
            q4 = reinterpret_cast<const uint4 *>(stage)[st_half * 128 + ps * 64 + (lane >> 5) * 32 + (i * 4 + ((lane & 31) >> 3)) * 4 + st_chunk];
        };
        auto epi_post = [&](const f32x16 (&S)[4], int dp, int k) {
            const int u = k >> 3, kk = k & 7;
            if (kk < 4) {   // bias, relu, split; four consecutive channels per register group
                const int qg = kk;
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float m = S[2 * u][qg * 4 + r], c = S[2 * u + 1][qg * 4 + r];
                    float v = (m + c * (1.0f / kSplitScale)) + bias2[qg * 4 + r];
                    v = v > 0.f ? v : 0.f;
                    _Float16 a, l2;
                    split16(v, a, l2, amax);
                    hi[r] = a; lo[r] = l2;
                }
                const int oc0 = 8 * qg + 4 * (lane >> 5);
                *reinterpret_cast<h4 *>(stage + col2 * 32 + oc0) = hi;
                *reinterpret_cast<h4 *>(stage + 1024 + col2 * 32 + oc0) = lo;
            } else {
                // columns 8k .. 8k+7 of the tile are row (2u + (k >> 1) + 4 (k & 1)) of the plane: eight positions x 32 channels
                // of the output; a position is one run of the interleaved layout [hi 32 | lo' 32] (k_dense.hip), 16 bytes per
                // lane and pass, two rows per pass
                const int j = kk - 4, ps = j >> 1, i = j & 1;
                const int kq = ps * 2 + (lane >> 5);
                const int row = 2 * u + (kq >> 1) + 4 * (kq & 1);
                const int p = i * 4 + ((lane & 31) >> 3);
                const size_t e = (size_t)(dp * 64 + row * 8 + p) * 64 + st_half * 32 + st_chunk * 8;   // halves
#ifdef MMW_DIAG_CONV_NOSTORE
                if (q4.x == 0x12345678u) *reinterpret_cast<uint4 *>(o + e) = q4;
#else
                *reinterpret_cast<uint4 *>(o + e) = q4;
#endif
                if (kk == 7) {   // the staging tile may be written again
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
#pragma unroll
        for (int d = 0; d < NZ; d++) {
            const int pc0 = d * 100 + cell2, pc1 = pc0 + 20;
            f32x16(&A)[4] = acc[d & 1];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < 16; r++) A[q][r] = 0.f;
            // Software-pipelined over the plane's valid taps: the operands of tap k+1 are requested BEFORE the six MFMAs
            // of tap k are issued.
            h8 xh0, xl0, xh1, xl1, wlo;
            {
                int t0 = 0;
#pragma unroll
                for (int tt = 0; tt < C::kTaps; tt++) if ((unsigned)(d + C::tap_kd(tt) - 1) < (unsigned)NZ) { t0 = tt; break; }
                const int i0 = pc0 + C::tap_off(t0), i1 = pc1 + C::tap_off(t0);
                xh0 = Hh[i0]; xl0 = Hl[i0]; xh1 = Hh[i1]; xl1 = Hl[i1];
                wlo = W2lo[t0 * 64 + lane];
            }
            int kth = 0;   // (compile time: the loop is unrolled) how many valid taps of this plane have been issued
#pragma unroll
            for (int tap = 0; tap < C::kTaps; tap++) {
                if ((unsigned)(d + C::tap_kd(tap) - 1) >= (unsigned)NZ) continue;  // compile-time (d and tap are unrolled)
                int nxt = -1;
#pragma unroll
                for (int tt = C::kTaps - 1; tt > tap; tt--) if ((unsigned)(d + C::tap_kd(tt) - 1) < (unsigned)NZ) nxt = tt;
                h8 nh0 = xh0, nl0 = xl0, nh1 = xh1, nl1 = xl1, nw = wlo;
#ifndef MMW_DIAG_CONV_NOEPI2
                if (d > 0 && kth < 16) epi_pre(kth);   // (a store chunk's read of the staging tile: in front of the MFMAs)
#endif
                if (nxt >= 0) {
                    const int i0 = pc0 + C::tap_off(nxt), i1 = pc1 + C::tap_off(nxt);
                    nh0 = Hh[i0]; nl0 = Hl[i0]; nh1 = Hh[i1]; nl1 = Hl[i1];
                    nw = W2lo[nxt * 64 + lane];
                }
                // (the scheduler must not pull tap k + 1's MFMAs up to their operands' loads: it would wait for the LDS there)
                __builtin_amdgcn_sched_barrier(0);
#ifdef MMW_DIAG_CONV_NOMFMA2
                asm volatile("" ::"v"(xh0), "v"(xl0), "v"(xh1), "v"(xl1), "v"(wlo), "v"(w2hi[tap]));
#else
                A[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xh0, A[0], 0, 0, 0);
                A[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xh1, A[2], 0, 0, 0);
                A[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xl0, A[1], 0, 0, 0);
                A[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xl1, A[3], 0, 0, 0);
                A[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh0, A[1], 0, 0, 0);
                A[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xh1, A[3], 0, 0, 0);
#endif
#ifndef MMW_DIAG_CONV_NOEPI2
                if (d > 0 && kth < 16) {
                    // ... and a chunk of the previous plane's epilogue BETWEEN them: a wave issues in order, so VALU work placed behind the
                    // six MFMAs would start when the last of them does; the scheduler is told to deal it out, eight to an MFMA
                    if ((kth & 7) >= 4) __builtin_amdgcn_sched_barrier(0);   // (a store chunk -- one store, which waits for its LDS read -- stays behind the MFMAs)
                    epi_post(acc[(d - 1) & 1], d - 1, kth);
                    if ((kth & 7) < 4) {
#pragma unroll
                        for (int g = 0; g < 6; g++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        }
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                kth++;
                xh0 = nh0; xl0 = nl0; xh1 = nh1; xl1 = nl1; wlo = nw;
            }
            CSTAMP(3);  // conv2 MFMA loop (+ the previous plane's epilogue)
        }
#ifdef MMW_DIAG_CONV_NOEPI2
        {
            float keep = 0.f;
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 16; r++) keep += acc[st][q][r];
            if (keep == 12345.678f) o[lane] = (_Float16)keep;
        }
#else
#pragma unroll
        for (int k = 0; k < 16; k++) { epi_pre(k); epi_post(acc[(NZ - 1) & 1], NZ - 1, k); }   // the last plane's, alone
#endif
        CSTAMP(4);  // last epilogue + stores
        // this sample's verdict: its keypoints are meaningless under the split arithmetic (the caller recomputes exactly these
        // samples in fp32: mars.MarsCNN.forward, mmw_mars_range_fixup)
        const bool bad = __any(over || !(amax < 65504.0f));
        if (bad && sample_flags && lane == 0) {   // appended to the fix-up's list (any order; [0] = running count, [2 ..] = sample indices)
            const int pos = atomicAdd(&sample_flags[0], 1);
            if (pos < kRangeFixCap) sample_flags[2 + pos] = b;
        }
        any_over |= bad;
    }
    if (range_flag && any_over && lane == 0) atomicOr(range_flag, 1);
}
#ifdef MMW_STAMPS
extern "C" int mmw_diag_conv_stamps(unsigned long long *out /*[8]*/, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// per DEVICE (a process may drive several GPUs, each from its own thread: dist.LocalShardedTracker): the dynamic-LDS attribute
// (> 64 KB: without it the launch fails on that device) and the CU count behind the grid, taken once on whichever thread gets
// there first -- as launch_mars_dense1 does (k_dense.hip).  Returns 0, or -1 when the attribute could not be set.
template <int NZ>
static int launch_conv16_t(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, long long ld_out,
                           int B, int32_t *range_flag, int32_t *sample_flags, hipStream_t stream)
{
    struct DevPrep { std::once_flag once; int n_cu = 256; bool ok = false; };
    static DevPrep g_prep[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    DevPrep &P = g_prep[dev];
    std::call_once(P.once, [&]() {
        if (hipFuncSetAttribute((const void *)k_mars_conv16<NZ>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv16<NZ>::kLds) != hipSuccess) return;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) P.n_cu = prop.multiProcessorCount;
        P.ok = true;
    });
    if (!P.ok) return -1;
    int grid = (B + 3) / 4;
    if (grid > P.n_cu) grid = P.n_cu;  // one workgroup of four sample-waves per CU, persistent over samples
    hipLaunchKernelGGL(k_mars_conv16<NZ>, dim3(grid), dim3(256), Conv16<NZ>::kLds, stream, feat, w1, b1, w2, b2,
                       reinterpret_cast<_Float16 *>(out16), ld_out, B, range_flag, sample_flags);
    return 0;
}

int launch_mars_conv16(int nz, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, long long ld_out,
                       int B, int32_t *range_flag, int32_t *sample_flags, hipStream_t stream)
{
    if (B <= 0) return 0;
    if (nz == 3) return launch_conv16_t<3>(feat, w1, b1, w2, b2, out16, ld_out, B, range_flag, sample_flags, stream);
    return launch_conv16_t<1>(feat, w1, b1, w2, b2, out16, ld_out, B, range_flag, sample_flags, stream);
}

// ---- samples that left fp16's range under the split arithmetic, recomputed in Keras' own fp32 -- on the device, no host wait ----
// k_range_gather: the samples k_mars_conv16 appended to the fix-up list (list[0] = how many, list[2 ..] = which; the first
// `cap` of them count): list[1] = the number taken, the running count back to zero for the next call, bit 1 of the range word
// when there were more, and copies of their feature tensors.  One workgroup.
__global__ __launch_bounds__(256) void k_range_gather(const float *__restrict__ feat, int32_t *__restrict__ list, int n, int per,
                                                       int cap, float *__restrict__ small, int32_t *__restrict__ range_flag)
{
    __shared__ int used_s;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int total = list[0];
        const int used = total < cap ? total : cap;
        list[1] = used;
        list[0] = 0;
        if (total > cap && range_flag) atomicOr(range_flag, 2);   // more flagged samples than the fix-up holds: the rest stay meaningless
        used_s = used;
    }
    __syncthreads();
    const int m = used_s;
    for (int r = 0; r < m; r++) {
        const int i = list[2 + r];
        if (i < 0 || i >= n) continue;
        const float *src = feat + (size_t)i * per;
        for (int e = tid; e < per; e += 256) small[(size_t)r * per + e] = src[e];
    }
}
// k_range_scatter: the recomputed keypoints over the rows the split arithmetic left meaningless
__global__ __launch_bounds__(64) void k_range_scatter(const float *__restrict__ kp_small, const int32_t *__restrict__ list, float *__restrict__ kp,
                                                      int nout, int n)
{
    const int r = blockIdx.x;
    if (r >= list[1]) return;
    const int i = list[2 + r];
    if (i < 0 || i >= n) return;
    for (int j = threadIdx.x; j < nout; j += 64) kp[(size_t)i * nout + j] = kp_small[(size_t)r * nout + j];
}

void launch_range_gather(const float *feat, int32_t *list, int n, int per, int cap, float *small, int32_t *range_flag, hipStream_t stream)
{
    hipLaunchKernelGGL(k_range_gather, dim3(1), dim3(256), 0, stream, feat, list, n, per, cap, small, range_flag);
}
void launch_range_scatter(const float *kp_small, const int32_t *list, float *kp, int cap, int nout, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(k_range_scatter, dim3(cap), dim3(64), 0, stream, kp_small, list, kp, nout, n);
}

}  // namespace mmw
