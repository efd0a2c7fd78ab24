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
__global__ __launch_bounds__(128, 1) void k_mars_conv(const float *__restrict__ feat, const float *__restrict__ w1,
                                                       const float *__restrict__ b1, const float *__restrict__ w2,
                                                       const float *__restrict__ b2, float *__restrict__ out, int B)
{
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
    int a2base[3];
#pragma unroll
    for (int t = 0; t < 3; t++) a2base[t] = (lane >> 5) * kPV + padded_origin((wave * 3 + t) * 32 + (lane & 31));
    __syncthreads();

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
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
        f32x16 c0, c1, c2;
#pragma unroll
        for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < kK2; ks++) {
            const int tap = (2 * ks) / 16, ic0 = (2 * ks) % 16;
            const int koff = ic0 * kPV + (tap / 9) * 100 + ((tap / 3) % 3) * 10 + (tap % 3);  // compile-time constant
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(H1[a2base[0] + koff], w2r[ks], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(H1[a2base[1] + koff], w2r[ks], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(H1[a2base[2] + koff], w2r[ks], c2, 0, 0, 0);
        }
        // C/D: col = lane&31 (out channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        float *o = out + (size_t)b * 192 * 32;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oc = lane & 31;
            float v0 = c0[r] + bias2, v1 = c1[r] + bias2, v2 = c2[r] + bias2;
            o[((wave * 3 + 0) * 32 + row) * 32 + oc] = v0 > 0.f ? v0 : 0.f;
            o[((wave * 3 + 1) * 32 + row) * 32 + oc] = v1 > 0.f ? v1 : 0.f;
            o[((wave * 3 + 2) * 32 + row) * 32 + oc] = v2 > 0.f ? v2 : 0.f;
        }
        __syncthreads();  // both waves are done reading Xp / H1 before the next sample overwrites them
    }
}

void launch_mars_conv(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, float *out, int B,
                      hipStream_t stream)
{
    if (B <= 0) return;
    const int grid = B < 512 ? B : 512;  // 2 workgroups per CU, persistent over samples
    hipLaunchKernelGGL(k_mars_conv, dim3(grid), dim3(128), 0, stream, feat, w1, b1, w2, b2, out, B);
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
    static constexpr int kWaveLds = (kPad + 1) * 16 * 2 + (kPad + 1) * 32 * 2 + 2 * 32 * 32 * 2;  // X8 hi/lo, H1 hi/lo (+ zero slot), staging
    static constexpr int kW1Lds = kS1 * 64 * 16 * 2;
    static constexpr int kLds = 4 * kWaveLds + kW1Lds;
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

template <int NZ>
__global__ __launch_bounds__(256, 1) void k_mars_conv16(const float *__restrict__ feat, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        const float *__restrict__ b2, _Float16 *__restrict__ out, int B)
{
    using C = Conv16<NZ>;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- LDS carve-up ----
    h8 *W1hi = reinterpret_cast<h8 *>(lds_raw);                 // [kS1][64] conv1 A fragments (shared)
    h8 *W1lo = W1hi + C::kS1 * 64;
    char *mine = lds_raw + C::kW1Lds + wave * C::kWaveLds;
    h8 *Xhi = reinterpret_cast<h8 *>(mine);                     // [kPad + 1] (entry kPad = zero slot)
    h8 *Xlo = Xhi + (C::kPad + 1);
    h8 *Hhi = Xlo + (C::kPad + 1);                              // [kPad + 1][2] : 16 channels = two h8
    h8 *Hlo = Hhi + 2 * (C::kPad + 1);
    _Float16 *stage = reinterpret_cast<_Float16 *>(Hlo + 2 * (C::kPad + 1));  // [2][32 pos][32 oc]
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
    h8 w2hi[C::kTaps], w2lo[C::kTaps];
#pragma unroll
    for (int t = 0; t < C::kTaps; t++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = w2[((t * 16) + 8 * (lane >> 5) + j) * 32 + (lane & 31)];
            _Float16 a, b;
            split16(v, a, b);
            w2hi[t][j] = a; w2lo[t][j] = b;
        }
    }
    // conv1 D: row = oc = (lane >> 4) * 4 + reg, col = position;  conv2 D: row = oc = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float bias1[4], bias2[16];
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = b1[(lane >> 4) * 4 + r];
#pragma unroll
    for (int r = 0; r < 16; r++) bias2[r] = b2[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    __syncthreads();  // W1 fragments visible; the only block barrier of the kernel

    const int stride = gridDim.x * 4;
    int b = blockIdx.x * 4 + wave;
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
        // ---- this sample's input into the padded channels-last volume, split ----
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 5; c++) { _Float16 a, l2; split16(xin[q][c], a, l2); hi[c] = a; lo[c] = l2; }
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- conv1: tiles of 16 positions; D[oc][pos] ----
#pragma unroll 1
        for (int t = 0; t < C::kPos / 16; t++) {
            const int p = t * 16 + (lane & 15), pc = C::padded(p), d = p >> 6;
            f32x4 am = {0.f, 0.f, 0.f, 0.f}, ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < C::kS1; s++) {
                const int tap = 4 * s + (lane >> 4);
                const int kd = NZ == 3 ? tap / 9 : 1;
                const bool ok = tap < C::kTaps && (unsigned)(d + kd - 1) < (unsigned)NZ;
                int off = 0;
                {   // tap offset of this lane's tap (the four candidates of the step are compile-time constants)
                    const int g = lane >> 4;
                    const int o0 = C::tap_off(4 * s + 0 < C::kTaps ? 4 * s + 0 : 0), o1 = C::tap_off(4 * s + 1 < C::kTaps ? 4 * s + 1 : 0),
                              o2 = C::tap_off(4 * s + 2 < C::kTaps ? 4 * s + 2 : 0), o3 = C::tap_off(4 * s + 3 < C::kTaps ? 4 * s + 3 : 0);
                    off = g == 0 ? o0 : g == 1 ? o1 : g == 2 ? o2 : o3;
                }
                const int idx = ok ? pc + off : C::kPad;
                const h8 xh = Xhi[idx], xl = Xlo[idx];
                const h8 wh = W1hi[s * 64 + lane], wl = W1lo[s * 64 + lane];
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, am, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, ac, 0, 0, 0);
            }
            h4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = (am[r] + ac[r] * (1.0f / kSplitScale)) + bias1[r];
                v = v > 0.f ? v : 0.f;
                _Float16 a, l2;
                split16(v, a, l2);
                hi[r] = a; lo[r] = l2;
            }
            // channels 4g .. 4g+3 of position p
            reinterpret_cast<h4 *>(Hhi + 2 * pc)[lane >> 4] = hi;
            reinterpret_cast<h4 *>(Hlo + 2 * pc)[lane >> 4] = lo;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- conv2: tiles of 32 positions (half a plane: the plane d is uniform); D[oc][pos] ----
        _Float16 *o = out + (size_t)b * 2 * (C::kPos * 32);
#pragma unroll 1
        for (int t = 0; t < C::kPos / 32; t++) {
            const int p = t * 32 + (lane & 31), pc = C::padded(p), d = t >> 1;
            f32x16 am, ac;
#pragma unroll
            for (int r = 0; r < 16; r++) { am[r] = 0.f; ac[r] = 0.f; }
#pragma unroll
            for (int tap = 0; tap < C::kTaps; tap++) {
                if ((unsigned)(d + C::tap_kd(tap) - 1) >= (unsigned)NZ) continue;  // uniform: the tap reads a plane outside the volume
                const int idx = 2 * (pc + C::tap_off(tap)) + (lane >> 5);
                const h8 xh = Hhi[idx], xl = Hlo[idx];
                am = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xh, am, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2hi[tap], xl, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2lo[tap], xh, ac, 0, 0, 0);
            }
            // epilogue: bias, relu, split; four consecutive channels per register group -> staging tile [pos][oc]
#pragma unroll
            for (int qg = 0; qg < 4; qg++) {
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = (am[qg * 4 + r] + ac[qg * 4 + r] * (1.0f / kSplitScale)) + bias2[qg * 4 + r];
                    v = v > 0.f ? v : 0.f;
                    _Float16 a, l2;
                    split16(v, a, l2);
                    hi[r] = a; lo[r] = l2;
                }
                const int oc0 = 8 * qg + 4 * (lane >> 5);
                *reinterpret_cast<h4 *>(stage + (lane & 31) * 32 + oc0) = hi;
                *reinterpret_cast<h4 *>(stage + 1024 + (lane & 31) * 32 + oc0) = lo;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the tile's 32 x 32 halfs are contiguous in the output ([pos][oc]): 2 KiB per half, 16 bytes per lane and pass
            const uint4 *sv = reinterpret_cast<const uint4 *>(stage);
            uint4 *gh = reinterpret_cast<uint4 *>(o + t * 1024), *gl = reinterpret_cast<uint4 *>(o + C::kPos * 32 + t * 1024);
            gh[lane] = sv[lane]; gh[64 + lane] = sv[64 + lane];
            gl[lane] = sv[128 + lane]; gl[64 + lane] = sv[192 + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int NZ>
static void launch_conv16_t(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, int B,
                            hipStream_t stream)
{
    static bool prepared = false;
    if (!prepared) {
        hipFuncSetAttribute((const void *)k_mars_conv16<NZ>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv16<NZ>::kLds);
        prepared = true;
    }
    int grid = (B + 3) / 4;
    if (grid > 256) grid = 256;  // one workgroup of four sample-waves per CU, persistent over samples
    hipLaunchKernelGGL(k_mars_conv16<NZ>, dim3(grid), dim3(256), Conv16<NZ>::kLds, stream, feat, w1, b1, w2, b2,
                       reinterpret_cast<_Float16 *>(out16), B);
}

void launch_mars_conv16(int nz, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, int B,
                        hipStream_t stream)
{
    if (B <= 0) return;
    if (nz == 3) launch_conv16_t<3>(feat, w1, b1, w2, b2, out16, B, stream);
    else launch_conv16_t<1>(feat, w1, b1, w2, b2, out16, B, stream);
}

}  // namespace mmw
