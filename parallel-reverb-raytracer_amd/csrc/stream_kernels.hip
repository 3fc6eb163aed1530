// stream_kernels.hip — the per-impulse streaming stages: microphone / HRTF attenuation
// (reference rayverb/kernel.cpp:505-625, kernels `attenuate` and `hrtf`), predelay
// (rayverb/rayverb.h:49-97) and time binning (rayverb/rayverb.cpp:48-77, flattenImpulses).
//
// These are the HBM-bound kernels of the path: 64 B in (+ 64 B out when materialised) per
// impulse.  Impulses are 64-byte records, so loads/stores are laid out so that one wave
// instruction always covers a contiguous span:
//   * materialised attenuate: 4 lanes per impulse, 16 B per lane  (1 KiB per wave instruction);
//   * fused attenuate+bin   : 16 lanes per impulse, 4 B per lane, so that the 8 band volumes sit
//     one per lane and one wave-wide float-atomic instruction adds 4 impulses x (2 channels x
//     8 bands) = 4 x 64 contiguous bytes into the [bin][channel][band] accumulation image
//     (memory-side atomics are paid per 64-byte request — MI355X_MICROARCH "Global float atomics").
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include "kernels.h"
#include "rvb_math.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#define WAVE 64

namespace {

typedef float nt_float4_t __attribute__((ext_vector_type(4)));

struct ModelDev {
    int hrtf;
    uint32_t nchannels;
    v3 mic;
    v3 sdir[8];             // speaker directions, already normalised (kernel.cpp:511)
    float coeff[8];
    const float * table;    // [2][360*180+1][8]
    v3 pointing, up;
    v3 bx, by, bz;          // the listener's basis of kernel.cpp:538-549 — it depends on (pointing, up) only, so it is computed once
                            // on the host with the same operations (rvb_math.h is shared) instead of once per impulse
    v3 ear[2];              // kernel.cpp:599-603
    bool exact_rows;        // measurement / test switch RVB_HRTF_EXACT_ROWS=1: every table row through the binary64 atan2 (angle_deg)
};

// reference kernel.cpp:537-549
__host__ __device__ __forceinline__ v3 transform3(v3 pointing, v3 up, v3 d)
{
    v3 x = normalize3(cross3(up, pointing));
    v3 y = cross3(pointing, x);
    v3 z = pointing;
    return mk3(dot3(x, d), dot3(y, d), dot3(z, d));
}

ModelDev make_model(const AttenuationModel & m)
{
    ModelDev d;
    d.hrtf = m.hrtf;
    d.nchannels = m.nchannels;
    d.mic = mk3(m.mic[0], m.mic[1], m.mic[2]);
    for (int i = 0; i < 8; ++i) {
        d.sdir[i] = normalize3(mk3(m.speaker_dir[i][0], m.speaker_dir[i][1], m.speaker_dir[i][2]));
        d.coeff[i] = m.speaker_coeff[i];
    }
    d.table = m.hrtf_table;
    d.pointing = mk3(m.facing[0], m.facing[1], m.facing[2]);
    d.up = mk3(m.up[0], m.up[1], m.up[2]);
    d.bx = normalize3(cross3(d.up, d.pointing));
    d.by = cross3(d.pointing, d.bx);
    d.bz = d.pointing;
    const float width = 0.1f;                                   // kernel.cpp:597
    d.ear[0] = transform3(d.pointing, d.up, mk3(-width, 0.0f, 0.0f)) + d.mic;
    d.ear[1] = transform3(d.pointing, d.up, mk3(width, 0.0f, 0.0f)) + d.mic;
    const char * e = getenv("RVB_HRTF_EXACT_ROWS");          // (read per launch: a test flips it inside one process)
    d.exact_rows = e && e[0] == '1';
    return d;
}

__device__ __forceinline__ float atan2_cr(float y, float x) { return (float) atan2((double) y, (double) x); }

// reference kernel.cpp:505-513: gain of one speaker for an impulse at `pos`
__device__ __forceinline__ float speaker_gain(const ModelDev & m, uint32_t ch, v3 pos)
{
    const v3 direction = normalize3(pos - m.mic);               // getDirection, kernel.cpp:528
    return (1 - m.coeff[ch]) + m.coeff[ch] * dot3(normalize3(direction), m.sdir[ch]);
}

// reference kernel.cpp:563-584: table row selected for an impulse at `pos` (same for both ears)
// transform (kernel.cpp:538-549) with the precomputed basis: the three dot products that remain per impulse
__device__ __forceinline__ v3 to_listener(const ModelDev & m, v3 d)
{
    return mk3(dot3(m.bx, d), dot3(m.by, d), dot3(m.bz, d));
}

// Table row = a * 180 + e with a = (long) (degrees(azimuth) + 180) % 360, e = 90 - (long) degrees(elevation) (kernel.cpp:569-584).
// Only the INTEGER parts of the two angles in degrees matter.  The oracle's atan2 is the correctly rounded binary32 value
// (evaluated in binary64: ~150 double-precision instructions per call); here the angle is first taken with the binary32 atan2f
// (~40 instructions) and the binary64 evaluation is kept for the cases where that could change the integer part:
//   deg_fast and deg_exact differ by at most 57.3 * |atan2f - atan2| + two roundings of a value <= 360
//   <= 57.3 * 1.5e-6 (atan2f: 6 ulp of pi at most, OpenCL / ocml accuracy) + 2 * 1.6e-5 < 1.2e-4 degrees,
// so an angle that is farther than kAngleMargin = 2e-3 degrees from every integer truncates to the same integer either way
// (about 0.4 % of the angles are nearer and take the binary64 path; NaN compares false and takes it too).
// tests/test_gpu_parity.py holds whole traces' rows against the always-exact evaluation (RVB_HRTF_EXACT_ROWS=1).
#define RVB_DEG_PER_RAD 57.295779513082320877f
__device__ __forceinline__ float angle_deg(float y, float x, float offset, bool always_exact)
{
    float deg = atan2f(y, x) * RVB_DEG_PER_RAD + offset;
    const float kAngleMargin = 2e-3f;
    const bool sure = fabsf(deg - rintf(deg)) > kAngleMargin;
    if (!sure || always_exact)
        deg = atan2_cr(y, x) * RVB_DEG_PER_RAD + offset;        // the reference's operations on the correctly rounded angle
    return deg;
}

__device__ __forceinline__ int64_t row_of(float az_deg_plus_180, float el_deg)
{
    int64_t a = (int64_t) az_deg_plus_180;
    a %= 360;
    int64_t e = (int64_t) el_deg;
    e = 90 - e;
    return a * 180 + e;     // e == 180 runs into the next azimuth row (quirk Q5); row 360*180 is zero padding
}

__device__ __forceinline__ int64_t hrtf_row(const ModelDev & m, v3 pos)
{
    const v3 t = to_listener(m, normalize3(pos - m.mic));
    const float az = angle_deg(t.x, t.z, 180.0f, m.exact_rows);
    const float el = angle_deg(t.y, sqrtf(t.x * t.x + t.z * t.z), 0.0f, m.exact_rows);
    return row_of(az, el);
}

// The same row for two neighbouring lanes that share one impulse (a quad of attenuate_kernel, the two lanes of a bin in
// ordered_sum_hrtf_kernel): azimuth and elevation are both atan2(y, x) of different arguments, so the even lane evaluates the
// azimuth and the odd lane the elevation with ONE call, then they swap by DPP.  Same operations on the same operands as hrtf_row.
template <int CTRL> __device__ __forceinline__ float qdpp_f(float v);
__device__ __forceinline__ int64_t hrtf_row_quad(const ModelDev & m, v3 pos, uint32_t q)
{
    const v3 t = to_listener(m, normalize3(pos - m.mic));
    const bool odd = q & 1u;
    const float y = odd ? t.y : t.x;
    const float x = odd ? sqrtf(t.x * t.x + t.z * t.z) : t.z;
    const float deg = angle_deg(y, x, odd ? 0.0f : 180.0f, m.exact_rows);
    const float other = qdpp_f<0xB1>(deg);                     // quad_perm [1,0,3,2]: the pair lane's angle
    return row_of(odd ? other : deg, odd ? deg : other);
}

// reference kernel.cpp:616-622: arrival-time shift of one ear
__device__ __forceinline__ float hrtf_time(const ModelDev & m, uint32_t ch, v3 pos, float time)
{
    const float dist0 = length3(pos - m.mic);
    const float dist1 = length3(pos - m.ear[ch]);
    const float diff = dist1 - dist0;
    return time + diff * seconds_per_meter();
}

__device__ __forceinline__ float attenuated_time(const ModelDev & m, uint32_t ch, v3 pos, float time)
{
    return m.hrtf ? hrtf_time(m, ch, pos, time) : time;
}

// rayverb.h:89 fixPredelay, then rayverb.cpp:69 SAMPLE = round(time * samplerate)
__device__ __forceinline__ uint32_t time_bin(float time, float predelay, float sample_rate)
{
    const float t = time > predelay ? time - predelay : 0.0f;
    return (uint32_t) roundf(t * sample_rate);
}

// ---- materialised attenuation: 4 lanes per impulse -------------------------------------------
// DPP moves inside a quad (lanes 4k..4k+3): quad_perm broadcast of lane K, pair swaps
template <int CTRL> __device__ __forceinline__ float qdpp_f(float v)
{
    return __uint_as_float((uint32_t) __builtin_amdgcn_mov_dpp((int) __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ uint32_t qdpp_u(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xF, 0xF, true);
}
#define QUAD_BCAST(k) ((k) * 0x55)
#define QUAD_SWAP1 0xB1
#define QUAD_SWAP2 0x4E
#define ATT_UNROLL 1      // 16-byte chunks per lane per pass; with one workgroup per 4 KiB the dispatcher provides the parallelism

// One 16-byte chunk of one impulse per lane: chunk 0/1 = volume, 2 = position, 3 = time.
// Speaker model (kernel.cpp:505-535).  The two normalisations of kernel.cpp:511/:528 are three divisions each
// by the same length: lane k of the quad divides component k, so a wave spends ONE correctly rounded division
// per normalisation instead of three (same operations on the same operands: bit-identical results).
__device__ __forceinline__ float4 attenuate_chunk_speaker(const ModelDev & m, uint32_t ch, uint32_t q, const float4 v)
{
    const float px = qdpp_f<QUAD_BCAST(2)>(v.x), py = qdpp_f<QUAD_BCAST(2)>(v.y), pz = qdpp_f<QUAD_BCAST(2)>(v.z);
    const float time = qdpp_f<QUAD_BCAST(3)>(v.x);
    uint32_t nz = (q < 2 && (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f)) ? 1u : 0u;
    nz |= qdpp_u<QUAD_SWAP1>(nz);
    nz |= qdpp_u<QUAD_SWAP2>(nz);                                 // kernel.cpp:524 any(volume != 0)
    float4 o = make_float4(0, 0, 0, 0);
    if (nz) {
        const v3 d = mk3(px, py, pz) - m.mic;                      // getDirection, kernel.cpp:528
        const float len = length3(d);
        const float own = q == 0 ? d.x : (q == 1 ? d.y : d.z);
        const float n_own = len == 0.0f ? own : own / len;         // normalize3: a zero vector stays zero
        const v3 n = mk3(qdpp_f<QUAD_BCAST(0)>(n_own), qdpp_f<QUAD_BCAST(1)>(n_own), qdpp_f<QUAD_BCAST(2)>(n_own));
        const float len2 = length3(n);                             // kernel.cpp:511 normalises the unit vector again
        const float u_own = len2 == 0.0f ? n_own : n_own / len2;
        const v3 u = mk3(qdpp_f<QUAD_BCAST(0)>(u_own), qdpp_f<QUAD_BCAST(1)>(u_own), qdpp_f<QUAD_BCAST(2)>(u_own));
        const float g = (1 - m.coeff[ch]) + m.coeff[ch] * dot3(u, m.sdir[ch]);
        if (q < 2) o = make_float4(v.x * g, v.y * g, v.z * g, v.w * g);
        else if (q == 2) o.x = time;
    }
    return o;
}

// HRTF model (kernel.cpp:586-625): table row by azimuth / elevation, per-ear arrival-time shift
__device__ __forceinline__ float4 attenuate_chunk_hrtf(const ModelDev & m, uint32_t ch, uint32_t q, const float4 v)
{
    const float px = qdpp_f<QUAD_BCAST(2)>(v.x), py = qdpp_f<QUAD_BCAST(2)>(v.y), pz = qdpp_f<QUAD_BCAST(2)>(v.z);
    const float time = qdpp_f<QUAD_BCAST(3)>(v.x);
    uint32_t nz = (q < 2 && (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f)) ? 1u : 0u;
    nz |= qdpp_u<QUAD_SWAP1>(nz);
    nz |= qdpp_u<QUAD_SWAP2>(nz);                                 // kernel.cpp:607
    float4 o = make_float4(0, 0, 0, 0);
    if (nz) {
        const v3 pos = mk3(px, py, pz);
        const int64_t row = hrtf_row_quad(m, pos, q);
        if (q < 2) {
            const float4 t = reinterpret_cast<const float4 *>(m.table + ((uint64_t) ch * (360 * 180 + 1) + (uint64_t) row) * 8)[q];
            o = make_float4(v.x * t.x, v.y * t.y, v.z * t.z, v.w * t.w);
        } else if (q == 2) {
            o.x = hrtf_time(m, ch, pos, time);
        }
    }
    return o;
}

template <bool HRTF>
__global__ __launch_bounds__(256) void attenuate_kernel(ModelDev m, uint32_t ch, const float4 * __restrict__ in,
                                                        float4 * __restrict__ out, uint64_t n)
{
    const uint32_t q = threadIdx.x & 3u;
    const uint64_t nchunks = n * 4;                                // whole quads: a quad's four chunks are live together
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t c0 = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; c0 < nchunks; c0 += stride * ATT_UNROLL) {
        float4 v[ATT_UNROLL];
#pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u) {                    // all loads leave before the first result is needed
            const uint64_t c = c0 + (uint64_t) u * stride;
            v[u] = make_float4(0, 0, 0, 0);
            if (c < nchunks) {
                const nt_float4_t t = __builtin_nontemporal_load(reinterpret_cast<const nt_float4_t *>(in + c));
                v[u] = make_float4(t.x, t.y, t.z, t.w);
            }
        }
#pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u) {
            const uint64_t c = c0 + (uint64_t) u * stride;
            if (c < nchunks) {
                const float4 o = HRTF ? attenuate_chunk_hrtf(m, ch, q, v[u]) : attenuate_chunk_speaker(m, ch, q, v[u]);
                const nt_float4_t t = {o.x, o.y, o.z, o.w};
                __builtin_nontemporal_store(t, reinterpret_cast<nt_float4_t *>(out + c));
            }
        }
    }
}

// min non-zero / max attenuated time (the inputs of findPredelay, rayverb.h:49-74, and of MAX_SAMPLE, rayverb.cpp:57).
// Four lanes per impulse like attenuate_kernel: one 16-byte chunk per lane (1 KiB per wave instruction), position and
// time broadcast by DPP, so the per-ear time shift (two square roots per ear) is evaluated once per 16 impulses and wave
// instruction — the former 16-lanes-per-impulse layout spent four times the instructions on it and ran at 0.9 TB/s.
__global__ __launch_bounds__(256) void time_range_kernel(ModelDev m, const float4 * __restrict__ in, uint64_t n, uint32_t * range)
{
    const uint32_t q = threadIdx.x & 3u;
    const uint64_t nchunks = n * 4;
    float tmin = __builtin_inff(), tmax = 0.0f;
    for (uint64_t c = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (uint64_t) gridDim.x * blockDim.x) {
        const nt_float4_t t4 = __builtin_nontemporal_load(reinterpret_cast<const nt_float4_t *>(in + c));
        const float4 v = make_float4(t4.x, t4.y, t4.z, t4.w);
        const float px = qdpp_f<QUAD_BCAST(2)>(v.x), py = qdpp_f<QUAD_BCAST(2)>(v.y), pz = qdpp_f<QUAD_BCAST(2)>(v.z);
        const float time = qdpp_f<QUAD_BCAST(3)>(v.x);
        uint32_t nz = (q < 2 && (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f)) ? 1u : 0u;
        nz |= qdpp_u<QUAD_SWAP1>(nz);
        nz |= qdpp_u<QUAD_SWAP2>(nz);
        if (!nz)
            continue;           // attenuated impulse is {0, 0}: no part in findPredelay / maxtime
        const v3 pos = mk3(px, py, pz);
        // speaker channels all keep the input time; of the two ears, the quad's even lanes take the left one and the odd lanes the
        // right one (the wave-wide reduction below joins them): one time shift — two square roots — per lane instead of two
        const float t = attenuated_time(m, q & 1u, pos, time);
        if (t != 0.0f) tmin = fminf(tmin, t);
        tmax = fmaxf(tmax, t);
    }
    for (int off = 32; off > 0; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off));
        tmax = fmaxf(tmax, __shfl_xor(tmax, off));
    }
    // One atomic per wave only when it can still move the result: with one workgroup per 4 KiB there are millions of waves,
    // and that many atomics on two addresses serialise (70 ms at 12.8 M impulses).  A stale read only costs a redundant atomic.
    if ((threadIdx.x & 63u) == 0) {
        const volatile uint32_t * seen = range;
        if (tmin != __builtin_inff() && __float_as_uint(tmin) < seen[0]) atomicMin(range + 0, __float_as_uint(tmin));
        if (__float_as_uint(tmax) > seen[1]) atomicMax(range + 1, __float_as_uint(tmax));
    }
}

// accumulation image: acc[bin][channel][band], float atomics.
// Two phases per 64 impulses of a wave, through LDS:
//   1. impulse per lane: gain of every channel and the bin — computed ONCE per impulse
//      (the 16-lanes-per-impulse layout would recompute them in 16 lanes: ~30 wave instructions per
//      impulse, issue-bound; this form needs ~5);
//   2. 16 lanes per impulse (8 bands x 2 channels): one float-atomic wave instruction adds
//      4 impulses x 64 contiguous bytes.
#define HIST_ROW 20     // LDS words per staged impulse: 16 record words + padding (b128-aligned, 4-way conflicts at most)
#define HIST_MAXCH 8

// HIST_WAVES waves per workgroup, each with a staging area of its own (no cross-wave traffic).  One-wave workgroups: when
// another impulse response's path_kernel holds 24 of a CU's 32 wave slots (IrPipeline), a four-wave workgroup finds room on a
// CU only now and then, single waves slip into the free slots.
#ifndef HIST_WAVES
#define HIST_WAVES 1
#endif
__global__ __launch_bounds__(64 * HIST_WAVES) void histogram_fast_kernel(ModelDev m, const float4 * __restrict__ in, uint64_t n,
                                                             float predelay, float sample_rate, uint64_t nbins,
                                                             float * __restrict__ acc)
{
    __shared__ __attribute__((aligned(16))) float stage[HIST_WAVES][64 * HIST_ROW];      // the wave's 64 records
    __shared__ float gains[HIST_WAVES][64 * HIST_MAXCH];
    __shared__ uint32_t bins[HIST_WAVES][64 * 2];                                        // speakers: one bin; hrtf: one per ear
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float * st = stage[wave];
    float * gn = gains[wave];
    uint32_t * bn = bins[wave];
    const uint64_t nwaves = (uint64_t) gridDim.x * HIST_WAVES;
    const uint64_t ngroups = (n + 63) / 64;
    const uint32_t f = lane & 15u, band = f & 7u, chsel = f >> 3;
    for (uint64_t grp = (uint64_t) blockIdx.x * HIST_WAVES + wave; grp < ngroups; grp += nwaves) {
        const uint64_t first = grp * 64;
        // stage 64 records (4 KiB) with four fully coalesced 1-KiB wave loads
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t chunk = first * 4 + (uint64_t) k * 64 + lane;        // 16-byte chunk index
            float4 v = make_float4(0, 0, 0, 0);
            if (chunk < n * 4) {
                const nt_float4_t t = __builtin_nontemporal_load(reinterpret_cast<const nt_float4_t *>(in) + chunk);
                v = make_float4(t.x, t.y, t.z, t.w);
            }
            const uint32_t imp = (uint32_t) ((k * 64 + lane) >> 2), part = lane & 3u;
            *reinterpret_cast<float4 *>(st + imp * HIST_ROW + part * 4) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): the wave's own LDS writes (one wave per staging area)
        __builtin_amdgcn_wave_barrier();
        // phase 1: lane = impulse
        {
            const float4 v0 = *reinterpret_cast<const float4 *>(st + lane * HIST_ROW);
            const float4 v1 = *reinterpret_cast<const float4 *>(st + lane * HIST_ROW + 4);
            const float4 p4 = *reinterpret_cast<const float4 *>(st + lane * HIST_ROW + 8);
            const float time = st[lane * HIST_ROW + 12];
            const bool nonzero = v0.x != 0.0f || v0.y != 0.0f || v0.z != 0.0f || v0.w != 0.0f
                              || v1.x != 0.0f || v1.y != 0.0f || v1.z != 0.0f || v1.w != 0.0f;
            const v3 pos = mk3(p4.x, p4.y, p4.z);
            uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;                       // 0xFFFFFFFF: contributes nothing
            if (nonzero && first + lane < n) {
                if (m.hrtf) {
                    const int64_t row = hrtf_row(m, pos);
                    // gains of an hrtf impulse are per band: keep the row, phase 2 reads the table
                    gn[lane * HIST_MAXCH] = __uint_as_float((uint32_t) row);
                    b0 = time_bin(hrtf_time(m, 0, pos, time), predelay, sample_rate);
                    b1 = time_bin(hrtf_time(m, 1, pos, time), predelay, sample_rate);
                } else {
                    for (uint32_t ch = 0; ch < m.nchannels; ++ch)
                        gn[lane * HIST_MAXCH + ch] = speaker_gain(m, ch, pos);
                    b0 = b1 = time_bin(time, predelay, sample_rate);
                }
            }
            bn[lane * 2] = b0;
            bn[lane * 2 + 1] = b1;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // phase 2: 16 lanes per impulse, 4 impulses per wave instruction
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t imp = j * 4 + (lane >> 4);
            const float vol = st[imp * HIST_ROW + band];
            for (uint32_t pair = 0; pair < m.nchannels; pair += 2) {
                const uint32_t ch = pair + chsel;
                if (ch >= m.nchannels)
                    continue;
                const uint64_t bin = bn[imp * 2 + (m.hrtf ? ch : 0)];
                if (bin >= nbins)
                    continue;                      // zero-volume impulse (or beyond the histogram)
                float gain;
                if (m.hrtf) gain = m.table[((uint64_t) ch * (360 * 180 + 1) + (uint64_t) __float_as_uint(gn[imp * HIST_MAXCH])) * 8 + band];
                else gain = gn[imp * HIST_MAXCH + ch];
                atomicAdd(acc + (bin * m.nchannels + ch) * 8 + band, vol * gain);
            }
        }
        __builtin_amdgcn_wave_barrier();          // staging area is reused by the next group
    }
}

// acc[bin][ch][band] -> out[ch][band][bin] (+=, so that several shards / image passes can add up).
// 64-bin tiles through LDS: the read is one contiguous span, the writes are 256-byte runs per (ch, band).
#define TR_BINS 64
__global__ __launch_bounds__(256) void histogram_transpose_kernel(const float * __restrict__ acc, float * __restrict__ out,
                                                                  uint32_t nchannels, uint64_t nbins)
{
    __shared__ float tile[64][TR_BINS + 1];                  // [ch*8+band][bin], up to 8 channels
    const uint32_t cb = nchannels * 8;
    const uint64_t ntiles = (nbins + TR_BINS - 1) / TR_BINS;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t bin0 = t * TR_BINS;
        const uint32_t count = (uint32_t) min((uint64_t) TR_BINS, nbins - bin0) * cb;
        for (uint32_t i = threadIdx.x; i < count; i += 256)
            tile[i % cb][i / cb] = acc[bin0 * cb + i];
        __syncthreads();
        const uint32_t width = (uint32_t) min((uint64_t) TR_BINS, nbins - bin0);
        for (uint32_t i = threadIdx.x; i < cb * TR_BINS; i += 256) {
            const uint32_t row = i / TR_BINS, col = i % TR_BINS;
            if (col < width)
                out[(uint64_t) row * nbins + bin0 + col] += tile[row][col];
        }
        __syncthreads();
    }
}

// ---- exact mode -------------------------------------------------------------------------------
// Key of an impulse = its bin; `sentinel` (= nbins, the first value past every bin) for an impulse that adds nothing.
// Keys therefore need key_bits_for(nbins) bits only (20 at workload C2 instead of 32: three radix passes instead of four).
__global__ __launch_bounds__(256) void bin_keys_kernel(ModelDev m, uint32_t ch, const rvb_impulse * __restrict__ in, uint64_t n,
                                                       uint64_t index_base, float predelay, float sample_rate, uint32_t sentinel,
                                                       uint32_t * __restrict__ keys, uint32_t * __restrict__ values)
{
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const float4 * r = reinterpret_cast<const float4 *>(in + i);
        const float4 v0 = r[0], v1 = r[1], p = r[2];
        const float time = r[3].x;
        const bool nonzero = v0.x != 0.0f || v0.y != 0.0f || v0.z != 0.0f || v0.w != 0.0f
                          || v1.x != 0.0f || v1.y != 0.0f || v1.z != 0.0f || v1.w != 0.0f;
        // A zero-volume impulse attenuates to {0, 0} (quirk Q2): the reference adds its zeros to bin 0, which changes nothing
        // (x + 0 = x, and a sum that starts at +0 never becomes -0).  It gets the sentinel key — sorted last, matched by no
        // bin — instead of bin 0: the blocked third of all shadow rays would otherwise make ONE lane of ordered_sum_kernel walk
        // millions of entries (2.2 s at workload C2).
        uint32_t key = sentinel;
        if (nonzero) key = min(time_bin(attenuated_time(m, ch, mk3(p.x, p.y, p.z), time), predelay, sample_rate), sentinel);
        keys[index_base + i] = key;
        values[index_base + i] = (uint32_t) (index_base + i);
    }
}

// HRTF model: the two ears shift the arrival time differently (kernel.cpp:616-622), so each ear has its own bin per impulse.  Both
// keys come from ONE pass over the impulses, into ONE list of 2 n entries that one radix sort orders: ear e's entry of impulse j
// sits at e * n + j and carries key e * (nbins + 1) + bin (its sentinel: e * (nbins + 1) + nbins), so the sorted list is ear 0's
// bins, ear 0's silent impulses, ear 1's bins, ear 1's silent impulses — each run in impulse order (the sort is stable).
__global__ __launch_bounds__(256) void bin_keys_hrtf_kernel(ModelDev m, const rvb_impulse * __restrict__ in, uint64_t count, uint64_t index_base,
                                                            uint64_t n, float predelay, float sample_rate, uint32_t nbins,
                                                            uint32_t * __restrict__ keys, uint32_t * __restrict__ values)
{
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t) gridDim.x * blockDim.x) {
        const float4 * r = reinterpret_cast<const float4 *>(in + i);
        const float4 v0 = r[0], v1 = r[1], p = r[2];
        const float time = r[3].x;
        const bool nonzero = v0.x != 0.0f || v0.y != 0.0f || v0.z != 0.0f || v0.w != 0.0f
                          || v1.x != 0.0f || v1.y != 0.0f || v1.z != 0.0f || v1.w != 0.0f;
        uint32_t k0 = nbins, k1 = nbins;                      // (a zero-volume impulse adds nothing: bin_keys_kernel)
        if (nonzero) {
            const v3 pos = mk3(p.x, p.y, p.z);
            k0 = min(time_bin(hrtf_time(m, 0, pos, time), predelay, sample_rate), nbins);
            k1 = min(time_bin(hrtf_time(m, 1, pos, time), predelay, sample_rate), nbins);
        }
        const uint64_t j = index_base + i;
        keys[j] = k0;
        keys[n + j] = nbins + 1u + k1;
        values[j] = (uint32_t) j;
        values[n + j] = (uint32_t) j;
    }
}

// Where each bin's run starts in the sorted key list: starts[key] = first position of that key (entries of absent keys keep
// the caller's 0xFFFFFFFF fill).  One coalesced pass over the keys replaces a 23-step binary search per bin — 19 M dependent
// random reads at workload C2, which made the summation kernel fetch 3 GB for 0.5 GB of impulses.
__global__ __launch_bounds__(256) void bin_starts_kernel(const uint32_t * __restrict__ keys, uint64_t n, uint64_t nbins, uint32_t * __restrict__ starts,
                                                         uint32_t * __restrict__ ends)
{
    for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t key = keys[k];
        if (key >= nbins)
            continue;
        if (k == 0 || keys[k - 1] != key)
            starts[key] = (uint32_t) k;
        if (k + 1 == n || keys[k + 1] != key)     // (with both ends known the summation loop has no data-dependent exit: its gathers overlap)
            ends[key] = (uint32_t) (k + 1);
    }
}

// One lane per bin: add the bin's impulses in impulse order (the order of rayverb.cpp:67-74) ON TOP of what the histogram
// holds (the caller zeroes it; a second context that continues the fold with the next ray shard starts from the first one's
// sums, so the chain reproduces the serial order over all shards).  Speaker channels keep the input time (kernel.cpp:530-533)
// and share the bin, so ONE sorted list serves NCH channels of the speaker model (first_channel .. first_channel + NCH - 1);
// the two ears of the HRTF model have their own bins (NCH = 1, one list per ear).
#ifndef SUM_UNROLL
#define SUM_UNROLL 4
#endif
// TWO lanes per bin — the even lane folds bands 0-3, the odd lane bands 4-7 (each reads its 16-byte half of the volume; both
// read the position) — so the gather of 8 M scattered 64-byte records has twice the loads in flight per bin.
template <bool HRTF, int NCH>
__global__ __launch_bounds__(64) void ordered_sum_kernel(ModelDev m, uint32_t first_channel, const rvb_impulse * __restrict__ diffuse,
                                                         uint64_t ndiffuse, const rvb_impulse * __restrict__ images,
                                                         const uint32_t * __restrict__ values,
                                                         const uint32_t * __restrict__ starts, const uint32_t * __restrict__ ends,
                                                         uint64_t nbins, uint64_t bin_begin, uint64_t bin_end, float * __restrict__ hist, uint32_t xcd_chunk)
{
    // (bins [bin_begin, bin_end) of this launch: the caller may fold the histogram bin range by bin range, each range leaving for
    // the host as soon as it is final — rvb_ir_accumulate_export)
    // xcd_chunk != 0: workgroups b and b + 8 share an XCD (round-robin dispatch, speed only), so a run of `xcd_chunk` consecutive
    // workgroups' bins is given to the eight-apart workgroups of ONE XCD: the two 64-byte records of a 128-byte line (bounces b and
    // b + 1 of a ray, some 1 500 bins apart at workload C2) are then gathered through the same L2, the second one from cache.
    uint32_t block = blockIdx.x;
    if (xcd_chunk) {
        const uint32_t span = 8u * xcd_chunk;
        if (block / span < gridDim.x / span) {
            const uint32_t within = block % span;
            block = block - within + (within & 7u) * xcd_chunk + (within >> 3);
        }
    }
    const uint64_t t = (uint64_t) block * blockDim.x + threadIdx.x;
    const uint64_t bin = bin_begin + (t >> 1);
    const uint32_t half = (uint32_t) t & 1u;
    if (bin >= bin_end)
        return;
    const uint64_t lo = starts[bin];
    if (lo == 0xFFFFFFFFull)
        return;                               // nothing lands in this bin: the histogram keeps what it holds
    float sum[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            sum[c][b] = hist[((uint64_t) (first_channel + c) * 8 + half * 4 + b) * nbins + bin];
    const uint64_t hi = ends[bin];
    // SUM_UNROLL impulses per round: their index loads, then their record gathers, leave together; the adds stay in impulse order
    for (uint64_t k = lo; k < hi; k += SUM_UNROLL) {
        float4 v[SUM_UNROLL], p[SUM_UNROLL];
#pragma unroll
        for (int u = 0; u < SUM_UNROLL; ++u) {
            const uint64_t kk = k + u < hi ? k + u : lo;
            const uint64_t idx = values[kk];
            const rvb_impulse * imp = idx < ndiffuse ? diffuse + idx : images + (idx - ndiffuse);
            const float4 * r = reinterpret_cast<const float4 *>(imp);
            v[u] = r[half];
            p[u] = r[2];
        }
#pragma unroll
        for (int u = 0; u < SUM_UNROLL; ++u) {
            if (k + u >= hi) break;
            const float vol[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            const v3 pos = mk3(p[u].x, p[u].y, p[u].z);      // (keyed into a bin: the volume is non-zero)
            if (HRTF) {
                const float * tb = m.table + ((uint64_t) first_channel * (360 * 180 + 1) + (uint64_t) hrtf_row(m, pos)) * 8 + half * 4;
#pragma unroll
                for (int b = 0; b < 4; ++b) sum[0][b] += vol[b] * tb[b];
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float g = speaker_gain(m, first_channel + c, pos);
#pragma unroll
                    for (int b = 0; b < 4; ++b) sum[c][b] += vol[b] * g;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            hist[((uint64_t) (first_channel + c) * 8 + half * 4 + b) * nbins + bin] = sum[c][b];
}

// The HRTF model's ordered sum, both ears in ONE launch over the combined list of bin_keys_hrtf_kernel: two lanes per (bin, ear) —
// the even lane folds bands 0-3 and evaluates the azimuth, the odd lane bands 4-7 and the elevation (hrtf_row_quad: one atan2 per
// lane and impulse instead of two, the binary32 one unless the integer part of an angle is in doubt).
#ifndef RVB_HRTF_SUM_THREADS
#define RVB_HRTF_SUM_THREADS 256
#endif
__global__ __launch_bounds__(RVB_HRTF_SUM_THREADS) void ordered_sum_hrtf_kernel(ModelDev m, const rvb_impulse * __restrict__ diffuse, uint64_t ndiffuse,
                                                              const rvb_impulse * __restrict__ images, const uint32_t * __restrict__ values,
                                                              const uint32_t * __restrict__ starts, const uint32_t * __restrict__ ends,
                                                              uint64_t nbins, uint64_t bin_begin, uint64_t bin_end, float * __restrict__ hist)
{
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    // (bin, ear) in bin-major order: a workgroup folds BOTH ears of a run of neighbouring bins.  The two ears' times differ by at
    // most 0.29 ms (13 bins at 44.1 kHz), so the impulses of ear 1's bin b are those of ear 0's bins b-13 .. b+13: gathered by the
    // same workgroup, or its neighbour, at about the same time, the second gather of a record finds it in cache (with all of ear 0's
    // bins first and ear 1's after them every record was fetched from HBM twice).
    const uint64_t slot = 2 * bin_begin + (t >> 1);          // bins [bin_begin, bin_end) of this launch
    const uint32_t half = (uint32_t) t & 1u;
    if (slot >= 2 * bin_end)
        return;
    const uint32_t ear = (uint32_t) slot & 1u;
    const uint64_t bin = slot >> 1;
    const uint64_t key = (uint64_t) ear * (nbins + 1) + bin;
    const uint64_t lo = starts[key];
    if (lo == 0xFFFFFFFFull)
        return;                               // nothing lands in this bin: the histogram keeps what it holds (both lanes of the pair leave)
    float sum[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        sum[b] = hist[((uint64_t) ear * 8 + half * 4 + b) * nbins + bin];
    const uint64_t hi = ends[key];
    const float * table = m.table + (uint64_t) ear * (360 * 180 + 1) * 8 + half * 4;
    for (uint64_t k = lo; k < hi; k += SUM_UNROLL) {
        float4 v[SUM_UNROLL], p[SUM_UNROLL];
#pragma unroll
        for (int u = 0; u < SUM_UNROLL; ++u) {
            const uint64_t kk = k + u < hi ? k + u : lo;
            const uint64_t idx = values[kk];
            const rvb_impulse * imp = idx < ndiffuse ? diffuse + idx : images + (idx - ndiffuse);
            const float4 * r = reinterpret_cast<const float4 *>(imp);
            v[u] = r[half];
            p[u] = r[2];
        }
#pragma unroll
        for (int u = 0; u < SUM_UNROLL; ++u) {
            if (k + u >= hi) break;           // (the two lanes of a bin agree: the DPP exchange below always finds its partner)
            const float4 tb = *reinterpret_cast<const float4 *>(table + (uint64_t) hrtf_row_quad(m, mk3(p[u].x, p[u].y, p[u].z), half) * 8);
            sum[0] += v[u].x * tb.x;
            sum[1] += v[u].y * tb.y;
            sum[2] += v[u].z * tb.z;
            sum[3] += v[u].w * tb.w;
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
        hist[((uint64_t) ear * 8 + half * 4 + b) * nbins + bin] = sum[b];
}

__global__ __launch_bounds__(256) void flat_keys_kernel(const rvb_attenuated_impulse * __restrict__ in, uint64_t n, float sample_rate,
                                                        uint32_t * __restrict__ keys, uint32_t * __restrict__ values,
                                                        uint32_t * max_time_bits)
{
    float tmax = 0.0f;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const float4 * r = reinterpret_cast<const float4 *>(in + i);
        const float4 v0 = r[0], v1 = r[1];
        const float t = r[2].x;
        tmax = fmaxf(tmax, t);                 // MAX_SAMPLE counts every impulse (rayverb.cpp:54-57)
        // an all-zero volume adds nothing to its bin (x + 0 = x): keyed past every bin, so that the many {0, 0} entries of an
        // attenuated array (quirk Q2) do not pile up on the one lane that owns bin 0
        const bool nonzero = v0.x != 0.0f || v0.y != 0.0f || v0.z != 0.0f || v0.w != 0.0f
                          || v1.x != 0.0f || v1.y != 0.0f || v1.z != 0.0f || v1.w != 0.0f;
        keys[i] = nonzero ? (uint32_t) roundf(t * sample_rate) : 0xFFFFFFFFu;
        values[i] = (uint32_t) i;
    }
    for (int off = 32; off > 0; off >>= 1)
        tmax = fmaxf(tmax, __shfl_xor(tmax, off));
    if ((threadIdx.x & 63u) == 0 && __float_as_uint(tmax) > *(const volatile uint32_t *) max_time_bits)
        atomicMax(max_time_bits, __float_as_uint(tmax));     // only a wave that can still raise the maximum pays for the atomic
}

// fixPredelay (rayverb.h:76-90) on a resident AttenuatedImpulse array: the time is the first float of the third 16-byte chunk
__global__ __launch_bounds__(256) void fix_predelay_kernel(rvb_attenuated_impulse * __restrict__ a, uint64_t n, float seconds)
{
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const float t = a[i].time;
        a[i].time = t > seconds ? t - seconds : 0.0f;
    }
}

__global__ __launch_bounds__(64) void flat_ordered_sum_kernel(const rvb_attenuated_impulse * __restrict__ in,
                                                              const uint32_t * __restrict__ keys, const uint32_t * __restrict__ values,
                                                              const uint32_t * __restrict__ starts,
                                                              uint64_t n, uint64_t nbins, float * __restrict__ out)
{
    const uint64_t bin = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= nbins)
        return;
    const uint64_t lo = starts[bin] == 0xFFFFFFFFu ? n : starts[bin];
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t k = lo; k < n && keys[k] == (uint32_t) bin; ++k) {
        const float4 * r = reinterpret_cast<const float4 *>(in + values[k]);
        const float4 v0 = r[0], v1 = r[1];
        sum[0] += v0.x; sum[1] += v0.y; sum[2] += v0.z; sum[3] += v0.w;
        sum[4] += v1.x; sum[5] += v1.y; sum[6] += v1.z; sum[7] += v1.w;
    }
    for (int b = 0; b < 8; ++b)
        out[(uint64_t) b * nbins + bin] = sum[b];
}

// Workgroups for a streaming kernel of `items` work-items.  NOT capped at a few workgroups per CU: measured on MI355X
// (tools/copy_probe.hip, 819 MB -> 819 MB, 16 B per lane) a grid-strided 2048-workgroup launch moves 4.5-5.3 TB/s, one
// workgroup per 4 KiB moves 6.1-6.5 TB/s — the dispatcher then sweeps HBM as one moving window instead of 2048 streams
// 8 MB apart.  The kernels keep their grid-stride loops for the (never reached in practice) 2^31-workgroup limit.
unsigned stream_blocks(uint64_t items, unsigned per_block)
{
    static const uint64_t cap = getenv("RVB_STREAM_BLOCK_CAP") ? strtoull(getenv("RVB_STREAM_BLOCK_CAP"), nullptr, 10) : 0x7FFFFFFFull;
    uint64_t blocks = (items + per_block - 1) / per_block;
    if (blocks > cap) blocks = cap;
    return (unsigned) (blocks ? blocks : 1);
}

}  // namespace

void rvb_launch_attenuate(const AttenuationModel & m, uint32_t channel, const rvb_impulse * in, uint64_t n,
                          rvb_attenuated_impulse * out, hipStream_t s)
{
    if (n == 0) return;
    const dim3 grid(stream_blocks((n * 4 + ATT_UNROLL - 1) / ATT_UNROLL, 256));
    if (m.hrtf)
        hipLaunchKernelGGL(attenuate_kernel<true>, grid, dim3(256), 0, s, make_model(m), channel,
                           reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), n);
    else
        hipLaunchKernelGGL(attenuate_kernel<false>, grid, dim3(256), 0, s, make_model(m), channel,
                           reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), n);
}

void rvb_launch_time_range(const AttenuationModel & m, const rvb_impulse * in, uint64_t n, uint32_t * range, hipStream_t s)
{
    if (n == 0) return;
    // Every wave ends with two reads of the same two result words (and an atomic when it can still move them): with one workgroup
    // per 4 KiB — the launch shape the other streaming kernels want — that is 1.6 M reads of one line, which, not HBM, then sets the
    // kernel's time (0.34 ms at 12.8 M impulses; 0.17 ms with 2 048 grid-strided workgroups, 0.18 with 8 192, 0.25 with 65 536).
    static const unsigned cap = getenv("RVB_TIME_RANGE_BLOCKS") ? (unsigned) atoi(getenv("RVB_TIME_RANGE_BLOCKS")) : 2048u;
    hipLaunchKernelGGL(time_range_kernel, dim3(std::min(stream_blocks(n * 4, 256), cap ? cap : 2048u)), dim3(256), 0, s, make_model(m),
                       reinterpret_cast<const float4 *>(in), n, range);
}

void rvb_launch_histogram_fast(const AttenuationModel & m, const rvb_impulse * in, uint64_t n, float predelay,
                               float sample_rate, uint64_t nbins, float * acc, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(histogram_fast_kernel, dim3(stream_blocks(n, 64 * HIST_WAVES)), dim3(64 * HIST_WAVES), 0, s, make_model(m),
                       reinterpret_cast<const float4 *>(in), n, predelay, sample_rate, nbins, acc);
}

void rvb_launch_histogram_transpose(const float * acc, float * out, uint32_t nchannels, uint64_t nbins, hipStream_t s)
{
    hipLaunchKernelGGL(histogram_transpose_kernel, dim3(stream_blocks((nbins + TR_BINS - 1) / TR_BINS * 256, 256)), dim3(256), 0, s,
                       acc, out, nchannels, nbins);
}

void rvb_launch_bin_keys(const AttenuationModel & m, uint32_t channel, const rvb_impulse * in, uint64_t n, uint64_t index_base,
                         float predelay, float sample_rate, uint32_t sentinel, uint32_t * keys, uint32_t * values, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(bin_keys_kernel, dim3(stream_blocks(n, 256)), dim3(256), 0, s, make_model(m), channel, in, n,
                       index_base, predelay, sample_rate, sentinel, keys, values);
}

void rvb_launch_bin_keys_hrtf(const AttenuationModel & m, const rvb_impulse * in, uint64_t count, uint64_t index_base, uint64_t n,
                              float predelay, float sample_rate, uint32_t nbins, uint32_t * keys, uint32_t * values, hipStream_t s)
{
    if (count == 0) return;
    hipLaunchKernelGGL(bin_keys_hrtf_kernel, dim3(stream_blocks(count, 256)), dim3(256), 0, s, make_model(m), in, count, index_base, n,
                       predelay, sample_rate, nbins, keys, values);
}

void rvb_launch_ordered_sum_hrtf(const AttenuationModel & m, const rvb_impulse * diffuse, uint64_t ndiffuse, const rvb_impulse * images,
                                 const uint32_t * sorted_values, const uint32_t * starts, const uint32_t * ends, uint64_t nbins, float * hist,
                                 hipStream_t s, uint64_t bin_begin, uint64_t bin_end)
{
    if (bin_end > nbins) bin_end = nbins;
    if (nbins == 0 || bin_begin >= bin_end) return;
    hipLaunchKernelGGL(ordered_sum_hrtf_kernel, dim3((unsigned) ((4 * (bin_end - bin_begin) + RVB_HRTF_SUM_THREADS - 1) / RVB_HRTF_SUM_THREADS)), dim3(RVB_HRTF_SUM_THREADS), 0, s, make_model(m), diffuse, ndiffuse,
                       images, sorted_values, starts, ends, nbins, bin_begin, bin_end, hist);
}

void rvb_launch_ordered_sum(const AttenuationModel & m, uint32_t first_channel, uint32_t nchannels, const rvb_impulse * diffuse,
                            uint64_t ndiffuse, const rvb_impulse * images, uint64_t nimages,
                            const uint32_t * sorted_values, const uint32_t * starts, const uint32_t * ends, uint64_t n,
                            uint64_t nbins, float * hist, hipStream_t s, uint64_t bin_begin, uint64_t bin_end)
{
    (void) nimages;
    if (bin_end > nbins) bin_end = nbins;
    if (nbins == 0 || n == 0 || bin_begin >= bin_end) return;
    const dim3 grid((unsigned) ((2 * (bin_end - bin_begin) + 63) / 64)), block(64);      // two lanes per bin
    const ModelDev md = make_model(m);
    static const uint32_t xcd_chunk = getenv("RVB_SUM_XCD_CHUNK") ? (uint32_t) atoi(getenv("RVB_SUM_XCD_CHUNK")) : 0u;      // workgroups per XCD run, 0 = off
#define RVB_SUM(HRTF, NCH) hipLaunchKernelGGL((ordered_sum_kernel<HRTF, NCH>), grid, block, 0, s, md, first_channel, diffuse, ndiffuse, \
                                              images, sorted_values, starts, ends, nbins, bin_begin, bin_end, hist, xcd_chunk)
    if (m.hrtf) { RVB_SUM(true, 1); return; }
    switch (nchannels) {                       // speaker channels of one sorted list
    case 1: RVB_SUM(false, 1); break;
    case 2: RVB_SUM(false, 2); break;
    case 3: RVB_SUM(false, 3); break;
    case 4: RVB_SUM(false, 4); break;
    default:                                   // more than four: in groups (64 accumulators per lane would spill)
        for (uint32_t c = 0; c < nchannels; c += 4) {
            const uint32_t k = nchannels - c < 4 ? nchannels - c : 4;
            rvb_launch_ordered_sum(m, first_channel + c, k, diffuse, ndiffuse, images, nimages, sorted_values, starts, ends, n, nbins, hist, s, bin_begin, bin_end);
        }
    }
#undef RVB_SUM
}

void rvb_launch_flat_keys(const rvb_attenuated_impulse * in, uint64_t n, float sample_rate, uint32_t * keys, uint32_t * values,
                          uint32_t * max_time_bits, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(flat_keys_kernel, dim3(stream_blocks(n, 256)), dim3(256), 0, s, in, n, sample_rate, keys, values, max_time_bits);
}

void rvb_launch_fix_predelay(rvb_attenuated_impulse * a, uint64_t n, float seconds, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(fix_predelay_kernel, dim3(stream_blocks(n, 256)), dim3(256), 0, s, a, n, seconds);
}

void rvb_launch_flat_ordered_sum(const rvb_attenuated_impulse * in, const uint32_t * sorted_keys, const uint32_t * sorted_values,
                                 const uint32_t * starts, uint64_t n, uint64_t nbins, float * out, hipStream_t s)
{
    if (nbins == 0) return;
    hipLaunchKernelGGL(flat_ordered_sum_kernel, dim3((unsigned) ((nbins + 63) / 64)), dim3(64), 0, s, in, sorted_keys,
                       sorted_values, starts, n, nbins, out);
}

void rvb_launch_bin_starts(const uint32_t * sorted_keys, uint64_t n, uint64_t nbins, uint32_t * starts, uint32_t * ends, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(bin_starts_kernel, dim3(stream_blocks(n, 256)), dim3(256), 0, s, sorted_keys, n, nbins, starts, ends);
}

size_t rvb_sort_temp_bytes(uint64_t n)
{
    size_t bytes = 0;
    (void) rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                              (const uint32_t *) nullptr, (uint32_t *) nullptr, (size_t) n, 0, 32, (hipStream_t) 0);
    return bytes;
}

void rvb_sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys_in, uint32_t * keys_out,
                    const uint32_t * values_in, uint32_t * values_out, uint64_t n, int key_bits, hipStream_t s)
{
    if (n == 0) return;
    (void) rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, values_in, values_out, (size_t) n, 0, (unsigned) key_bits, s);
}

// Grouping of the trace's work records by spatial bucket: sort (bucket key, record index) pairs on the
// key bits [begin_bit, end_bit) only — a one- or two-pass radix sort; values come from a counting iterator.
size_t rvb_group_records_temp_bytes(uint64_t n)
{
    size_t bytes16 = 0;
    (void) rocprim::radix_sort_pairs(nullptr, bytes16, (const uint16_t *) nullptr, (uint16_t *) nullptr,
                                     rocprim::counting_iterator<uint32_t>(0), (uint32_t *) nullptr, (size_t) n, 0, 16, (hipStream_t) 0);
    size_t bytes = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                                                   rocprim::counting_iterator<uint32_t>(0), (uint32_t *) nullptr, (size_t) n, 0, 32,
                                                   (hipStream_t) 0);
    (void) hipGetLastError();          // a size query launches nothing; drop whatever state it left behind
    return e == hipSuccess ? std::max(bytes, bytes16) : 0;
}

hipError_t rvb_group_records(void * temp, size_t temp_bytes, const uint32_t * keys, uint32_t * keys_scratch, uint32_t * order,
                             uint64_t n, uint32_t first_record, int begin_bit, int end_bit, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    // values = record numbers first_record .. first_record + n (a slice of the launch's records)
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_scratch, rocprim::counting_iterator<uint32_t>(first_record), order,
                                     (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, s);
}

hipError_t rvb_group_records16(void * temp, size_t temp_bytes, const uint16_t * keys, uint16_t * keys_scratch, uint32_t * order,
                               uint64_t n, uint32_t first_record, int begin_bit, int end_bit, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_scratch, rocprim::counting_iterator<uint32_t>(first_record), order,
                                     (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, s);
}
