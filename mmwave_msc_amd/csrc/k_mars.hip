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
#include <hip/hip_fp16.h>
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

// SPLIT = false: out[B][192][32] fp32.  SPLIT = true: the activation leaves the kernel already split for the fp16
// matrix cores, out16[B][2][6144] fp16 = [hi | lo'] with hi = fp16(a), lo' = fp16((a - hi) * 2^11): Dense-1 is then
// a @ W = hi.W_hi + 2^-11 (hi.W_lo' + lo'.W_hi) in two fp16 GEMMs with fp32 accumulation (mars.py), every product exact,
// the dropped lo'.lo' term 2^-22 relative.
constexpr float kSplitScale = 2048.0f;
template <bool SPLIT>
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
        __half *oh = reinterpret_cast<__half *>(out) + (size_t)b * 2 * 6144;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oc = lane & 31;
            float v[3] = {c0[r] + bias2, c1[r] + bias2, c2[r] + bias2};
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const float a = v[t] > 0.f ? v[t] : 0.f;
                const int e = ((wave * 3 + t) * 32 + row) * 32 + oc;
                if (SPLIT) {
                    const __half hi = __float2half_rn(a);
                    oh[e] = hi;
                    oh[6144 + e] = __float2half_rn((a - __half2float(hi)) * kSplitScale);
                } else {
                    o[e] = a;
                }
            }
        }
        __syncthreads();  // both waves are done reading Xp / H1 before the next sample overwrites them
    }
}

void launch_mars_conv(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, float *out, int B,
                      hipStream_t stream)
{
    if (B <= 0) return;
    const int grid = B < 512 ? B : 512;  // 2 workgroups per CU, persistent over samples
    hipLaunchKernelGGL(k_mars_conv<false>, dim3(grid), dim3(128), 0, stream, feat, w1, b1, w2, b2, out, B);
}
void launch_mars_conv_split(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, int B,
                            hipStream_t stream)
{
    if (B <= 0) return;
    const int grid = B < 512 ? B : 512;
    hipLaunchKernelGGL(k_mars_conv<true>, dim3(grid), dim3(128), 0, stream, feat, w1, b1, w2, b2, reinterpret_cast<float *>(out16), B);
}

}  // namespace mmw
