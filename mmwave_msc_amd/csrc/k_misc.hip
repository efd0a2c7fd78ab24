// k_misc.hip -- the streaming kernels either side of the tracker:
//   k_normalize   Utils.normalize_data (Utils.py:294-434)        HBM-bound, ordered compaction
//   k_feat_count / k_feat_scan   row offsets of the per-track feature tensors   (scene, track) order
//   k_features    relative_coordinates + format_single_frame     (Utils.py:437-520), one wave per 64-row frame
//   k_set_kp      track.keypoints = model output                 (Tracking.py:733-734)
//   k_pop_frame   BatchedData.pop_frame on the global ring         (Tracking.py:66-71)
//   k_export      flatten effective_tracks for read-back
//   k_table       fixed-size track summaries for the RCCL all-gather
#include <cstddef>
#include "mmw_device.hpp"
#include "mmw_math.hpp"
#include "mmw_launch.hpp"

namespace mmw {

// ---------------------------------------------------------------------------
// RT = the raw rows' type: double (mmw_normalize) or float (mmw_normalize_f32: 20 bytes per detected object, promoted
// exactly as it is loaded -- the IWR1443's objects are int16 counts scaled by a power of two, ReadDataIWR1443.py:150-170).
// R = rows per thread (max_pts / 256, rounded up): a scene is ONE round -- every load of the workgroup leaves at once, one
// barrier for the ordered compaction, whole-row 16-byte stores.  Measured at 4096 x 512 objects: 52-55 us = 3.2-3.4 TB/s of
// algorithmic bytes, three quarters of them WRITES (64 B per kept row against 20 B read): about half the 6.3 TB/s a plain copy
// reaches on this chip.  Staging both sides through LDS in whole lines (54.7 us) and one round instead of two per scene (55.5 us)
// changed nothing: neither coalescing nor the chain's length is what bounds it.
// The body both entries share: R raw rows (x, y, z, doppler, peakVal as doubles) of this thread -> the reference's arithmetic ->
// ordered compaction -> whole-row stores.  All 256 threads of the workgroup call (one barrier).
template <int R>
__device__ __forceinline__ void normalize_rows(const DevCfg &cfg, int s, int n, const double (&v)[R][5], double *__restrict__ out,
                                               int32_t *__restrict__ n_out, int *wcnt /* LDS [R * 4] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *dst = out + (size_t)s * cfg.max_pts * 8;
    double o[R][8];
    unsigned long long bal[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int i = q * 256 + tid;
        const double x = v[q][0], y = v[q][1], z = v[q][2], dop = v[q][3], pk = v[q][4];
        const double r = sqrt((x * x + y * y) + z * z);
        double vx, vy, vz;
        if (r == 0) { vx = 0; vy = dop; vz = 0; }           // Utils.py:387-390
        else { vx = dop * x / r; vy = dop * y / r; vz = dop * z / r; }
        o[q][0] = x;                                         // T . R_inv . [x,y,z,1]  (Utils.py:312-328)
        o[q][1] = cfg.tilt_cos * y + (-cfg.tilt_sin) * z;
        o[q][2] = (cfg.tilt_sin * y + cfg.tilt_cos * z) + cfg.s_height;
        o[q][3] = vx;
        o[q][4] = cfg.tilt_cos * vy + (-cfg.tilt_sin) * vz;
        o[q][5] = cfg.tilt_sin * vy + cfg.tilt_cos * vz;
        o[q][6] = dop;
        o[q][7] = pk;
        {   // Non-finite rows.  The reference multiplies full homogeneous 4-vectors by full 4 x 4 matrices, zeros included
            // (Utils.py:311-326): one NaN / infinite coordinate meets a zero (0 * inf = NaN) and makes all three transformed
            // coordinates NaN -- the filter below then drops the row (NaN compares false) --, one non-finite velocity component
            // (a NaN / infinite doppler) makes all three velocities NaN on a row that is kept.  Finite rows: nothing changes.
            constexpr int kNanInf = 0x3 | 0x4 | 0x200;
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            if (__builtin_amdgcn_class(x, kNanInf) || __builtin_amdgcn_class(y, kNanInf) || __builtin_amdgcn_class(z, kNanInf)) o[q][0] = o[q][1] = o[q][2] = qnan;
            if (__builtin_amdgcn_class(vx, kNanInf) || __builtin_amdgcn_class(vy, kNanInf) || __builtin_amdgcn_class(vz, kNanInf)) o[q][3] = o[q][4] = o[q][5] = qnan;
        }
        const bool keep = i < n && o[q][2] <= 2.5 && o[q][2] > 0 && o[q][1] > 0;   // Utils.py:423-427
        bal[q] = __ballot(keep);
        if (lane == 0) wcnt[q * 4 + wave] = __popcll(bal[q]);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int q = 0; q < R; q++) {
        int off = total;                                     // rows kept in the 64-row blocks before this one (blocks are in row order)
#pragma unroll
        for (int w = 0; w < 4; w++) { const int c = wcnt[q * 4 + w]; if (w < wave) off += c; total += c; }
        if ((bal[q] >> lane) & 1ULL) {
            double2 *d = reinterpret_cast<double2 *>(dst + (size_t)(off + __popcll(bal[q] & lanemask_lt())) * 8);   // (whole rows: 16-byte stores)
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = double2{o[q][2 * u], o[q][2 * u + 1]};
        }
    }
    if (tid == 0) n_out[s] = total;
}

template <typename RT, int R>
__global__ __launch_bounds__(256) void k_normalize(DevCfg cfg, const RT *__restrict__ raw, const int32_t *__restrict__ n_raw,
                                                   double *__restrict__ out, int32_t *__restrict__ n_out)
{
    __shared__ int wcnt[R * 4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int NP = cfg.max_pts;
    const int n = min(max(n_raw[s], 0), NP);
    const RT *in = raw + (size_t)s * NP * 5;
    double v[R][5];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int i = q * 256 + tid;
#pragma unroll
        for (int c = 0; c < 5; c++) v[q][c] = i < n ? (double)in[i * 5 + c] : 0.0;
    }
    normalize_rows<R>(cfg, s, n, v, out, n_out, wcnt);
}

// The radar's own wire format in, ring rows out: per scene the body of the detected-points TLV of an IWR1443 UART packet
// (MMWDEMO_UART_MSG_DETECTED_POINTS: u16 numObj, u16 xyzQFormat, then numObj x six little-endian int16 -- rangeIdx, dopplerIdx,
// peakVal, x, y, z; 12 bytes per object, ReadDataIWR1443.py:107-150), decoded as ReadIWR14xx.read decodes it (153-171: doppler
// indices above numDopplerBins / 2 - 1 get 65535 subtracted in int16, doppler = idx * dopplerResolutionMps, x, y, z / 2^Q) and
// normalised in the same registers.  packets: the bytes as they arrived (any 2-byte alignment of a body); tlv_offset[s] = byte
// offset of scene s's TLV BODY, < 0 = no detected-points TLV this frame (n_out = 0: the scene's frame is skipped).  The host only
// finds magic words (mmw_find_tlv).  Nothing outside packets[0 .. packets_bytes) is read: a body that does not lie inside it on a
// 2-byte boundary with all the objects it announces, or that announces more than max_pts objects (mmw_parse_uart: MMW_E_ARG),
// gives n_out = MMW_BAD_FRAME -- the scene's ERR_BADCOUNT in the mmw_step that follows.
template <int R>
__global__ __launch_bounds__(256) void k_normalize_tlv(DevCfg cfg, const uint8_t *__restrict__ packets, long long packets_bytes,
                                                       const long long *__restrict__ tlv_offset, double half_bins, double doppler_res,
                                                       double *__restrict__ out, int32_t *__restrict__ n_out)
{
    __shared__ int wcnt[R * 4];
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long off = tlv_offset[s];
    int n = 0;
    bool bad = false;
    double q = 1.0;
    const unsigned short *body = nullptr;
    if (off >= 0) {   // uniform
        bad = (off & 1) != 0 || off + 4 > packets_bytes;
        if (!bad) {
            body = reinterpret_cast<const unsigned short *>(packets + off);
            const int num = body[0], qfmt = body[1];
            bad = num > cfg.max_pts || off + 4 + 12LL * num > packets_bytes;
            n = bad ? 0 : num;
            q = ldexp(1.0, qfmt);
        }
    }
    double v[R][5];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = r * 256 + tid;
        unsigned short w[6] = {0, 0, 0, 0, 0, 0};
        if (i < n) {
            const unsigned short *o = body + 2 + (size_t)i * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) w[c] = o[c];
        }
        short dop = (short)w[1];
        if ((double)dop > half_bins) dop = (short)((int)dop - 65535);   // ReadDataIWR1443.py:150-157 (wraps in int16)
        v[r][0] = (double)(short)w[3] / q;
        v[r][1] = (double)(short)w[4] / q;
        v[r][2] = (double)(short)w[5] / q;
        v[r][3] = (double)dop * doppler_res;
        v[r][4] = (double)(short)w[2];
    }
    normalize_rows<R>(cfg, s, n, v, out, n_out, wcnt);
    if (bad && tid == 0) n_out[s] = -3;   // MMW_BAD_FRAME (the same thread wrote the 0 above)
}

// ---------------------------------------------------------------------------
// eligible tracks per scene (Tracking.py:721): a WAVE per scene, a lane per track of its list (t_cap <= 64) -- the records are scattered, so a
// thread per scene walked up to t_cap dependent round trips (19 us at 4096 x 8 tracks; as a loop inside the single scan workgroup below it
// had been 170 of the scan's 185 us); here every record of a scene is requested at once
__global__ __launch_bounds__(256) void k_feat_count(DevCfg cfg, DevState st, int32_t *__restrict__ row_off /*[S+1]*/)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= cfg.n_scenes) return;   // (wave-uniform)
    const SceneHdr *hdr = st.hdr + s;
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    const TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    bool elig = false;
    if (lane < hdr->n_tracks) {
        const TrackRec *rec = trk + order[lane];
        const int rl = rec->ring_len;
        int total = 0;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) total += k < rl ? rec->ring_n[k] : 0;
        elig = total > cfg.model_min_input;
    }
    const int c = __popcll(__ballot(elig));
    if (lane == 0) row_off[s] = c;
}

// single workgroup: in-place exclusive scan of the counts; row_off[S] = total
__global__ __launch_bounds__(1024) void k_feat_scan(DevCfg cfg, int32_t *__restrict__ row_off /*[S+1]*/)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x, S = cfg.n_scenes;
    const int per = (S + 1023) / 1024;
    const int s0 = tid * per, s1 = min(S, s0 + per);
    int sum = 0;
    for (int s = s0; s < s1; s++) sum += row_off[s];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int s = s0; s < s1; s++) { const int c = row_off[s]; row_off[s] = run; run += c; }
    if (tid == 1023) row_off[S] = part[1023];
}

// lane ^ J of a 32-bit value without the LDS crossbar where the hardware allows it: DPP inside quads (J = 1, 2) and inside rows of 16
// (J = 4, 8: the two row shifts, picked by the lane's bit J); ds_bpermute for 16 and 32 (mmw_cloud.hpp's xor_lane_d, for ints)
template <int J>
__device__ __forceinline__ int xor_lane_i(int v)
{
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4 || J == 8) {
        const int up = __builtin_amdgcn_update_dpp(0, v, 0x100 + J, 0xF, 0xF, true), dn = __builtin_amdgcn_update_dpp(0, v, 0x110 + J, 0xF, 0xF, true);   // row_shl: lane i <- i + J; row_shr: i - J
        return (__lane_id() & J) ? dn : up;
    } else return __shfl_xor(v, J);
}
// one compare-exchange step of the bitonic network on (key, row): block size SZ, partner lane ^ STRIDE
template <int SZ, int STRIDE>
__device__ __forceinline__ void bitonic_step(int lane, double &key, int &src)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(key);
    const int olo = xor_lane_i<STRIDE>((int)(unsigned)u), ohi = xor_lane_i<STRIDE>((int)(unsigned)(u >> 32)), os = xor_lane_i<STRIDE>(src);
    const double ok = __longlong_as_double((long long)(((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo));
    const bool up = (lane & SZ) == 0;          // ascending block
    const bool lower = (lane & STRIDE) == 0;   // this lane keeps the smaller of the pair
    const bool other_less = ok < key || (ok == key && os < src);
    const bool take = (up == lower) ? other_less : !other_less;
    if (take) { key = ok; src = os; }
}
template <int SZ, int STRIDE>
__device__ __forceinline__ void bitonic_merge(int lane, double &key, int &src)
{
    bitonic_step<SZ, STRIDE>(lane, key, src);
    if constexpr (STRIDE > 1) bitonic_merge<SZ, STRIDE / 2>(lane, key, src);
}

// np.argsort(padded[:, 0]) (Utils.py:513) + the gather it drives: a bitonic network over
// the wave on (x, row) -- ties ordered by row position -- then 5 fp32 stores per lane.  18 of its 21 steps exchange by DPP (partners inside a row of
// 16 lanes); with every step three trips through the LDS crossbar the kernel was a chain of ds_bpermute latencies (118 us at 98 k sorts).
__device__ inline void sort_rows_store(int lane, double v0, double v1, double v2, double v3, double v4, float *dst)
{
    double key = v0;
    int src = lane;
    bitonic_merge<2, 1>(lane, key, src);
    bitonic_merge<4, 2>(lane, key, src);
    bitonic_merge<8, 4>(lane, key, src);
    bitonic_merge<16, 8>(lane, key, src);
    bitonic_merge<32, 16>(lane, key, src);
    bitonic_merge<64, 32>(lane, key, src);
    const double s0 = __shfl(v0, src), s1 = __shfl(v1, src), s2 = __shfl(v2, src), s3 = __shfl(v3, src), s4 = __shfl(v4, src);
    // a lane's five values are 20 consecutive bytes: one 16-byte and one 4-byte store (4-byte aligned: global memory takes that)
    struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };
    *reinterpret_cast<F4 *>(dst + lane * 5) = F4{(float)s0, (float)s1, (float)s2, (float)s3};
    dst[lane * 5 + 4] = (float)s4;
}

// One wave per (eligible track, ring frame): lane r owns row r of the 64-row frame.  A workgroup = a scene: lane j of every wave
// first reads what track j's frames need (slot, ring sizes and slots, centroid: one round trip for the whole scene), a ballot
// gives every eligible track its output row, and the (track, frame) items are then dealt over the four waves -- each item one
// more round trip (its rows).  (Track after track with the waves over the frames, every step behind the previous one's loads,
// this kernel took 120 us for 18 k tracks: 0.27 of the HBM rate for 290 MB.)
__global__ __launch_bounds__(256) void k_features(DevCfg cfg, DevState st, const int32_t *__restrict__ row_off,
                                                  float *__restrict__ feat, int32_t *__restrict__ owner, int32_t *__restrict__ uid,
                                                  int cap_rows, const int32_t *__restrict__ n_in, int32_t *__restrict__ total_out)
{
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SceneHdr *hdr = st.hdr + s;
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    const TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    // (n_in: this frame's row counts -- the posture chain of mmw_frame_posture_host: a scene whose frame was skipped is not
    //  estimated, offline_main.py:56-60.  row_off NULL + total_out: the one-scene context, which needs no scan in front)
    const int T = (n_in && n_in[s] == 0) ? 0 : hdr->n_tracks, ring = cfg.ring;   // (T <= t_cap <= 64: one lane per track)
    const int row0 = row_off ? row_off[s] : 0;
    int slot = 0, rlen = 0, rn0 = 0, rn1 = 0, rn2 = 0, rn3 = 0, rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0, uidv = 0;
    double cx = 0.0, cy = 0.0;
    bool elig = false;
    if (lane < T) {
        slot = order[lane];
        const TrackRec *rec = trk + slot;
        rlen = rec->ring_len;
        rn0 = rec->ring_n[0]; rn1 = rec->ring_n[1]; rn2 = rec->ring_n[2]; rn3 = rec->ring_n[3];
        rs0 = rec->ring_slot[0]; rs1 = rec->ring_slot[1]; rs2 = rec->ring_slot[2]; rs3 = rec->ring_slot[3];
        cx = rec->centroid[0]; cy = rec->centroid[1];
        uidv = rec->uid;
        const int total = (rlen > 0 ? rn0 : 0) + (rlen > 1 ? rn1 : 0) + (rlen > 2 ? rn2 : 0) + (rlen > 3 ? rn3 : 0);
        elig = total > cfg.model_min_input;   // Tracking.py:721
    }
    // rows in track-list order; a track whose row does not fit the caller's buffer is dropped, and so is every later one
    const int my_row = row0 + __popcll(__ballot(elig) & lanemask_lt());
    const bool use = elig && my_row < cap_rows;
    const unsigned long long um = __ballot(use);
    if (wave == 0 && use) {
        owner[my_row * 2] = s;
        owner[my_row * 2 + 1] = lane;
        if (uid) uid[my_row] = uidv;
    }
    const int ne = __popcll(um);
    if (total_out && tid == 0) *total_out = ne;
    for (int item = wave; item < ne * ring; item += 4) {   // (uniform per wave; the next item's rows requested ahead of this item's sort: 135 us against 106 -- registers)
        const int i = item / ring, k = item - i * ring;
        unsigned long long m = um;
        for (int t = 0; t < i; t++) m &= m - 1ULL;
        const int j = __ffsll((long long)m) - 1;   // the i-th eligible track's list position
        const int sl = __shfl(slot, j), rl = __shfl(rlen, j), row = __shfl(my_row, j);
        const int n_k = __shfl(k == 0 ? rn0 : k == 1 ? rn1 : k == 2 ? rn2 : rn3, j), slot_k = __shfl(k == 0 ? rs0 : k == 1 ? rs1 : k == 2 ? rs2 : rs3, j);
        const double ccx = __shfl(cx, j), ccy = __shfl(cy, j);
        float *dst = feat + (((size_t)row * ring + k) * 64) * 5;
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        if (k < rl && lane < min(n_k, 64)) {
            const double *p = st.trk_ring + ((((size_t)s * cfg.t_cap + sl) * ring + slot_k) * cfg.ring_rows + lane) * 8;
            const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 6);
            v0 = a.x - ccx;                                     // relative_coordinates Utils.py:455-463
            v1 = a.y - ccy;
            v2 = p[2] - 0;
            v3 = b.x - 0;
            v4 = ((b.y - 0) - cfg.intensity_mu) / cfg.intensity_std;  // Utils.py:502
        }
        sort_rows_store(lane, v0, v1, v2, v3, v4, dst);
    }
    if (st.stats && wave == 0) {  // algorithmic bytes: ring rows read (<= 64 per frame, 64 B each), fp32 tensor written
        int rows_b = 0;
        if (use) rows_b = (rlen > 0 ? min(rn0, 64) : 0) + (rlen > 1 ? min(rn1, 64) : 0) + (rlen > 2 ? min(rn2, 64) : 0) + (rlen > 3 ? min(rn3, 64) : 0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rows_b += __shfl_xor(rows_b, o);
        if (lane == 0 && ne) {
            unsigned long long *sl = stats_slot(st, s);
            atomicAdd(&sl[kStatFeatBytes], 64ULL * (unsigned long long)rows_b + (unsigned long long)ne * (unsigned long long)(ring * 64 * 5 * 4));
            atomicAdd(&sl[kStatFeatRows], (unsigned long long)ne);
        }
    }
}

// Utils.format_single_frame (+ relative_coordinates) on caller-provided frames:
// frames[B][ring][64][8] fp64, counts[B][ring] (rows valid, <0 = frame absent), ref[B][2].
__global__ __launch_bounds__(256) void k_format_frames(DevCfg cfg, const double *__restrict__ frames, const int32_t *__restrict__ counts,
                                                       const double *__restrict__ ref, float *__restrict__ feat, int B)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ring = cfg.ring;
    const int item = blockIdx.x * 4 + wave;
    if (item >= B * ring) return;
    const int b = item / ring;
    const int m = min(counts[item], 64);
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (lane < m) {
        const double *p = frames + ((size_t)item * 64 + lane) * 8;
        v0 = p[0] - ref[b * 2];
        v1 = p[1] - ref[b * 2 + 1];
        v2 = p[2] - 0;
        v3 = p[6] - 0;
        v4 = ((p[7] - 0) - cfg.intensity_mu) / cfg.intensity_std;
    }
    sort_rows_store(lane, v0, v1, v2, v3, v4, feat + (size_t)item * 64 * 5);
}

__global__ void k_set_kp(DevCfg cfg, DevState st, const float *__restrict__ kp, const int32_t *__restrict__ owner, int n_rows,
                         const int32_t *__restrict__ dev_rows)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / 64, e = g % 64;
    if (dev_rows) n_rows = min(n_rows, *dev_rows);   // (a row count only the device knows: the launch covers the capacity)
    if (row >= n_rows || e >= MMW_NKP) return;
    const int s = owner[row * 2], j = owner[row * 2 + 1];
    if (s < 0 || s >= cfg.n_scenes) return;
    const SceneHdr *hdr = st.hdr + s;
    if (j < 0 || j >= hdr->n_tracks) return;
    TrackRec *rec = st.trk + (size_t)s * cfg.t_cap + st.order[(size_t)s * cfg.t_cap + j];
    rec->kp[e] = kp[(size_t)row * MMW_NKP + e];
}

// The same assignment one or more frames LATER (the CNN of frame f runs beside the tracker of frame f+1): the track
// list may have been re-ordered or shortened by _maintain_tracks since the features were taken, so a row is matched
// by the track's creation ordinal instead of its list position; a track that has expired meanwhile drops its row.
__global__ void k_set_kp_uid(DevCfg cfg, DevState st, const float *__restrict__ kp, const int32_t *__restrict__ owner,
                             const int32_t *__restrict__ uid, int n_rows)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / 64, e = g % 64;
    if (row >= n_rows) return;
    const int s = owner[row * 2], u = uid[row];
    if (s < 0 || s >= cfg.n_scenes) return;
    const int T = min(st.hdr[s].n_tracks, cfg.t_cap);
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    // list positions only move towards the front (ballot compaction), so start at the recorded one
    int hit = -1;
    for (int j0 = 0; j0 < T; j0 += 64) {
        const int j = j0 + e;
        const bool m = j < T && trk[order[j]].uid == u;
        const unsigned long long b = __ballot(m);
        if (b) { hit = order[j0 + __ffsll((long long)b) - 1]; break; }
    }
    if (hit < 0 || e >= MMW_NKP) return;
    trk[hit].kp[e] = kp[(size_t)row * MMW_NKP + e];
}

// ---------------------------------------------------------------------------
__global__ void k_export(DevCfg cfg, DevState st, mmw_track_record *__restrict__ out, int cap)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = g / cap, j = g % cap;
    if (s >= cfg.n_scenes) return;
    mmw_track_record *o = out + (size_t)s * cap + j;
    const SceneHdr *hdr = st.hdr + s;
    if (j >= hdr->n_tracks) {   // a list position beyond the scene's tracks: zeros, written HERE (no memset in front of the launch)
        unsigned long long *z = reinterpret_cast<unsigned long long *>(o);
        static_assert(sizeof(mmw_track_record) % 8 == 0, "zero fill by 8-byte words");
        for (size_t e = 0; e < sizeof(mmw_track_record) / 8; e++) z[e] = 0ULL;
        return;
    }
    const TrackRec *rec = st.trk + (size_t)s * cfg.t_cap + st.order[(size_t)s * cfg.t_cap + j];
    for (int e = 0; e < 9; e++) o->x[e] = e < cfg.dx ? rec->x[e] : 0.0;
    for (int e = 0; e < 81; e++) o->P[e] = (e / 9 < cfg.dx && e % 9 < cfg.dx) ? rec->P[e] : 0.0;
    for (int e = 0; e < 6; e++) {
        o->centroid[e] = rec->centroid[e];
        o->min_vals[e] = rec->minv[e];
        o->max_vals[e] = rec->maxv[e];
        o->spread_est[e] = rec->spread[e];
    }
    for (int e = 0; e < 36; e++) o->group_disp_est[e] = rec->gd[e];
    o->n_est = rec->n_est;
    o->lifetime = rec->lifetime;
    o->point_num = rec->point_num;
    o->is_static = rec->is_static;
    o->ring_len = rec->ring_len;
    o->uid = rec->uid;
    for (int k = 0; k < MMW_RING_MAX; k++) o->ring_n[k] = k < rec->ring_len ? rec->ring_n[k] : 0;
    for (int e = 0; e < MMW_NKP; e++) o->keypoints[e] = rec->kp[e];
    // (the record's trailing padding: the buffer is not cleared in front of the launch, and a caller may compare records as bytes)
    constexpr size_t kUsed = offsetof(mmw_track_record, keypoints) + sizeof(float) * MMW_NKP;
    static_assert(sizeof(mmw_track_record) - kUsed == 4 || sizeof(mmw_track_record) == kUsed, "trailing padding of mmw_track_record");
    if (sizeof(mmw_track_record) > kUsed) reinterpret_cast<int32_t *>(o)[kUsed / 4] = 0;
}

__global__ void k_table(DevCfg cfg, DevState st, mmw_track_summary *__restrict__ out, int slots, int scene_base)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = g / slots, j = g % slots;
    if (s >= cfg.n_scenes) return;
    mmw_track_summary *o = out + (size_t)s * slots + j;
    const SceneHdr *hdr = st.hdr + s;
    o->scene = scene_base + s;
    o->slot = j;
    const bool alive = j < hdr->n_tracks;
    o->alive = alive ? 1 : 0;
    const TrackRec *rec = alive ? st.trk + (size_t)s * cfg.t_cap + st.order[(size_t)s * cfg.t_cap + j] : nullptr;
    o->is_static = alive ? rec->is_static : 0;
    o->point_num = alive ? rec->point_num : 0;
    o->lifetime = alive ? (float)rec->lifetime : 0.f;
    for (int e = 0; e < 9; e++) o->x[e] = (alive && e < cfg.dx) ? (float)rec->x[e] : 0.f;
    for (int e = 0; e < 6; e++) o->centroid[e] = alive ? (float)rec->centroid[e] : 0.f;
    for (int e = 0; e < MMW_NKP; e++) o->keypoints[e] = alive ? rec->kp[e] : 0.f;
    // calc_fade_square (Visualizer.py:14-29) over calc_projection_points (Utils.py:180-219), in the reference's
    // operation order, fp64 with the float32 keypoints widened (numpy 1.26, the reference's pinned version)
    double px = 0, pz = 0, size = 0;
    if (alive) {
        const double xo = rec->x[0] + (double)rec->kp[3], yo = rec->x[1] + (double)rec->kp[41], zo = (double)rec->kp[22];
        const double xd = xo - cfg.m_x, yd = yo - cfg.m_y, zd = zo - cfg.m_z;
        px = xd == 0 ? xo : -cfg.m_y / (yd / xd) + cfg.m_x;
        pz = zd == 0 ? zo : -cfg.m_y / (yd / zd) + cfg.m_z;
        const double sz = cfg.fade_max - (rec->x[1] + (double)rec->kp[12]) * cfg.fade_weight;
        size = fmax(cfg.fade_min, fmin(cfg.fade_max, sz));
    }
    o->fade_x = (float)px;
    o->fade_z = (float)pz;
    o->fade_size = (float)size;
}

// flags == nullptr: every scene (mmw_reset); else only the scenes whose flag is non-zero (mmw_reset_scenes), and nothing of the
// context-wide schedules
__global__ void k_reset(DevCfg cfg, DevState st, const int32_t *__restrict__ flags)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < cfg.n_scenes && (!flags || flags[g])) {
        SceneHdr *h = st.hdr + g;
        h->n_tracks = 0;
        h->g_len = 1;  // BatchedData() starts with ONE empty frame (Utils.py:35-41, Tracking.py:38-41)
        for (int k = 0; k < MMW_RING_MAX; k++) { h->g_n[k] = 0; h->g_slot[k] = k; }
        h->need_db = 0;
        h->err = 0;
        h->db_u = 0;
        h->next_uid = 0;
        h->n_upd = 0;
        h->skipped = flags ? 1 : 0;  // (a scene reset on its own is in no update list: the next k_predict looks at its header)
        if (!flags) {
            st.perm[g] = g;
            st.perm[cfg.n_scenes + g] = g;
        } else {
            for (int k = 0; k < cfg.t_cap; k++) st.order[(size_t)g * cfg.t_cap + k] = k;
        }
    }
    if (flags) return;
    for (int e = g; e < 2 * kUpdWords; e += gridDim.x * blockDim.x) st.upd_count[e] = 0;
    if (g < 2) st.spc_count[g] = 0;
    if (g == 0) st.q[kQTimeout] = 0;   // (a bounded wait that gave up is reported by mmw_check until the context is reset)
    const size_t tot = (size_t)cfg.n_scenes * cfg.t_cap;
    for (size_t e = g; e < tot; e += (size_t)gridDim.x * blockDim.x) st.order[e] = (int32_t)(e % cfg.t_cap);
}

#ifdef MMW_DIAG_POISON
__global__ __launch_bounds__(1024) void k_poison_lds()
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    unsigned long long *w = reinterpret_cast<unsigned long long *>(lds_raw);
    for (int i = threadIdx.x; i < 160 * 1024 / 8; i += blockDim.x) w[i] = 0x7ff8dead7fffbeefULL;   // a NaN as fp64, huge as int32 / int64
    __syncthreads();
}
void launch_poison(hipStream_t stream)
{
    static bool prepared = false;
    if (!prepared) { (void)hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); prepared = true; }
    hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(1024), 160 * 1024, stream);   // (one workgroup per CU at a time: four rounds over 256 CUs)
}
#endif

void launch_normalize(const DevCfg &cfg, const void *raw, bool f32, const int32_t *n_raw, double *out, int32_t *n_out, hipStream_t st)
{
    static_assert(MMW_MAX_PTS_LIMIT <= 4 * 256, "k_normalize (and k_track / k_scene) take at most four rows per thread: a larger limit needs a round loop");
    const int r = (cfg.max_pts + 255) / 256;   // rows per thread: 1, 2 or 4 (max_pts <= 1024)
#define MMW_NORM(RT, R) mmw_launch(k_normalize<RT, R>, dim3(cfg.n_scenes), dim3(256), 0, st, cfg, reinterpret_cast<const RT *>(raw), n_raw, out, n_out)
    if (f32) { if (r <= 1) MMW_NORM(float, 1); else if (r == 2) MMW_NORM(float, 2); else MMW_NORM(float, 4); }
    else { if (r <= 1) MMW_NORM(double, 1); else if (r == 2) MMW_NORM(double, 2); else MMW_NORM(double, 4); }
#undef MMW_NORM
}
void launch_normalize_tlv(const DevCfg &cfg, const uint8_t *packets, long long packets_bytes, const long long *tlv_offset, double half_bins,
                          double doppler_res, double *out, int32_t *n_out, hipStream_t st)
{
    const int r = (cfg.max_pts + 255) / 256;
#define MMW_NORM_TLV(R) mmw_launch(k_normalize_tlv<R>, dim3(cfg.n_scenes), dim3(256), 0, st, cfg, packets, packets_bytes, tlv_offset, half_bins, doppler_res, out, n_out)
    if (r <= 1) MMW_NORM_TLV(1); else if (r == 2) MMW_NORM_TLV(2); else MMW_NORM_TLV(4);
#undef MMW_NORM_TLV
}
void launch_feat_scan(const DevCfg &cfg, const DevState &s, int32_t *row_off, hipStream_t st)
{
    hipLaunchKernelGGL(k_feat_count, dim3((cfg.n_scenes + 3) / 4), dim3(256), 0, st, cfg, s, row_off);
    hipLaunchKernelGGL(k_feat_scan, dim3(1), dim3(1024), 0, st, cfg, row_off);
}
void launch_features(const DevCfg &cfg, const DevState &s, const int32_t *row_off, float *feat, int32_t *owner, int32_t *uid, int cap,
                     hipStream_t st, const int32_t *n_in, int32_t *total_out)
{
    hipLaunchKernelGGL(k_features, dim3(cfg.n_scenes), dim3(256), 0, st, cfg, s, row_off, feat, owner, uid, cap, n_in, total_out);
}
void launch_format_frames(const DevCfg &cfg, const double *frames, const int32_t *counts, const double *ref, float *feat, int B, hipStream_t st)
{
    if (B <= 0) return;
    hipLaunchKernelGGL(k_format_frames, dim3((B * cfg.ring + 3) / 4), dim3(256), 0, st, cfg, frames, counts, ref, feat, B);
}
void launch_set_kp(const DevCfg &cfg, const DevState &s, const float *kp, const int32_t *owner, int n_rows, hipStream_t st, const int32_t *dev_rows)
{
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(k_set_kp, dim3((n_rows * 64 + 255) / 256), dim3(256), 0, st, cfg, s, kp, owner, n_rows, dev_rows);
}
void launch_set_kp_uid(const DevCfg &cfg, const DevState &s, const float *kp, const int32_t *owner, const int32_t *uid, int n_rows,
                       hipStream_t st)
{
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(k_set_kp_uid, dim3((n_rows * 64 + 255) / 256), dim3(256), 0, st, cfg, s, kp, owner, uid, n_rows);
}
void launch_export(const DevCfg &cfg, const DevState &s, mmw_track_record *out, int cap, hipStream_t st)
{
    const int tot = cfg.n_scenes * cap;
    hipLaunchKernelGGL(k_export, dim3((tot + 127) / 128), dim3(128), 0, st, cfg, s, out, cap);
}
void launch_table(const DevCfg &cfg, const DevState &s, mmw_track_summary *out, int slots, int base, hipStream_t st)
{
    const int tot = cfg.n_scenes * slots;
    hipLaunchKernelGGL(k_table, dim3((tot + 127) / 128), dim3(128), 0, st, cfg, s, out, slots, base);
}
// BatchedData.pop_frame() (Tracking.py:66-71) on the global ring of the scenes whose flag is set: the oldest frame
// goes, its physical slot moves behind the live ones (the same rotation add_frame does when the ring is full).
__global__ void k_pop_frame(DevCfg cfg, DevState st, const int32_t *__restrict__ flags)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cfg.n_scenes || (flags && !flags[s])) return;
    SceneHdr *hdr = st.hdr + s;
    const int len = hdr->g_len;
    if (len <= 0) return;
    const int first = hdr->g_slot[0];
    for (int k = 1; k < len; k++) { hdr->g_slot[k - 1] = hdr->g_slot[k]; hdr->g_n[k - 1] = hdr->g_n[k]; }
    hdr->g_slot[len - 1] = first;
    hdr->g_n[len - 1] = 0;
    hdr->g_len = len - 1;
}
// BatchedData.change_buffer_size (Tracking.py:60-64) on the global ring of the flagged scenes: from the next
// add_frame on, frames are popped while len >= new_size.
__global__ void k_set_batch_size(DevCfg cfg, DevState st, const int32_t *__restrict__ flags, int new_size)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cfg.n_scenes || (flags && !flags[s])) return;
    SceneHdr *hdr = st.hdr + s;
    hdr->skipped = (hdr->skipped & ~(kSkipRingMask << kSkipRingShift)) | (new_size << kSkipRingShift);
}
void launch_set_batch_size(const DevCfg &cfg, const DevState &s, const int32_t *flags, int new_size, hipStream_t st)
{
    hipLaunchKernelGGL(k_set_batch_size, dim3((cfg.n_scenes + 255) / 256), dim3(256), 0, st, cfg, s, flags, new_size);
}
// mmw_clear_errors: the sticky bits `bits` of the flagged scenes (nullptr: every scene) are cleared; nothing else changes
__global__ void k_clear_errors(DevCfg cfg, DevState st, const int32_t *__restrict__ flags, int bits)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cfg.n_scenes || (flags && !flags[s])) return;
    st.hdr[s].err &= ~bits;
}
void launch_clear_errors(const DevCfg &cfg, const DevState &s, const int32_t *flags, int bits, hipStream_t st)
{
    hipLaunchKernelGGL(k_clear_errors, dim3((cfg.n_scenes + 255) / 256), dim3(256), 0, st, cfg, s, flags, bits);
}
void launch_pop_frame(const DevCfg &cfg, const DevState &s, const int32_t *flags, hipStream_t st)
{
    hipLaunchKernelGGL(k_pop_frame, dim3((cfg.n_scenes + 255) / 256), dim3(256), 0, st, cfg, s, flags);
}

void launch_reset(const DevCfg &cfg, const DevState &s, const int32_t *flags, hipStream_t st)
{
    hipLaunchKernelGGL(k_reset, dim3((cfg.n_scenes + 255) / 256 > 0 ? (cfg.n_scenes + 255) / 256 : 1), dim3(256), 0, st, cfg, s, flags);
}

// Do kernels of stream B run while a kernel of stream A is spinning?  Streams are multiplexed onto a few hardware queues, and
// two streams that share one execute in order: a chain worker (side stream) that polls for what k_track (context's stream)
// publishes would then hold k_track back until its bounded wait runs out.  mmw_api.hip probes once per stream set-up:
// waiters on the side streams, one setter on the context's stream.
__global__ void k_probe_wait(int32_t *w, int slot, int polls)
{
    int seen = 0;
    for (int i = 0; i < polls && !seen; i++) {
        seen = __hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!seen) __builtin_amdgcn_s_sleep(8);
    }
    w[1 + slot] = seen;
}
__global__ void k_probe_set(int32_t *w) { __hip_atomic_store(&w[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
void launch_probe_wait(int32_t *w, int slot, int polls, hipStream_t st) { hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, st, w, slot, polls); }
void launch_probe_set(int32_t *w, hipStream_t st) { hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, st, w); }

}  // namespace mmw
