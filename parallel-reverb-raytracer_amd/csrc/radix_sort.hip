// radix_sort.hip — stable LSD radix sort of (32-bit key, 32-bit value) pairs on a few key bits, written for ONE property the
// library sort does not have: its kernels can run in what a resident path kernel leaves free.
//
// NOT the default (RVB_SORT=own selects it; rocPRIM's radix sort otherwise).  Results are identical (both sorts are stable; the
// parity tests run with either, tests/test_gpu_parity.py::test_own_radix_sort_gives_the_same_bytes).  Measured at workload C2 on
// one MI355X: alone on the GPU the record grouping takes 0.49 ms against rocPRIM's 0.37 and the exact mode 0.95 against 0.83 ms; in
// the bench pipeline the sorts' elapsed times do shrink as intended (grouping 2.1 -> 1.5 ms, exact mode 4.3 -> 3.1 ms) but the
// path kernels they now run beside stretch by more (6.7 -> 7.6 ms) and an IR takes 4.83 instead of 4.65 ms.  Kept as the measured
// answer to "why is a library sort on the hot path": the sort is not what the step waits for.
//
// Where it is used: the grouping of the trace's work records by BVH leaf position between path_kernel and shadow_kernel (16 key
// bits, values = record numbers) and the exact histogram mode's (time bin, impulse index) sort (≈20 key bits).  Both run while
// OTHER impulse responses' path kernels are resident (several contexts per GPU take turns), and those occupy the
// register files almost completely: two path_pair_kernel launches hold 6 waves x 80 VGPRs of the 512 per SIMD lane.  rocPRIM's
// onesweep kernels are 1024-thread workgroups at 96 VGPRs — sixteen waves that need four free slots of 96 registers on every SIMD
// of one CU at the same time, which a CU that runs even ONE path kernel (3 waves x 80 per SIMD) cannot offer; measured in the
// bench pipeline, a 0.37-ms grouping took up to 5.6 ms.  Here every kernel is a single-wave workgroup with at most 32 VGPRs and
// 1 KiB of LDS: it fits beside six resident path waves per SIMD, and no workgroup ever waits for another one (no look-back chain:
// a pass is three launches — tile histograms, a scan per digit, the scatter).
//
// One pass over `bits` <= 8 key bits (RADIX digits):
//   tile_histogram_kernel   a wave counts the digits of its TILE consecutive items in LDS and writes counts[digit][tile]
//   digit_scan_kernel       one wave per digit turns counts[digit][*] into exclusive prefixes, totals[digit] = the digit's count
//   scatter_kernel          a wave re-reads its tile 64 items at a time, ranks each item among the items of the same digit
//                           (match-any by `bits` ballots: items keep their order = stable), and writes it to
//                           base[digit] + prefix[digit][tile] + items of that digit seen so far in the tile + rank
#include "kernels.h"

#include <algorithm>

namespace {

constexpr uint32_t SORT_WAVE = 64;
constexpr uint32_t SORT_TILE = 4096;           // items per wave
constexpr uint32_t SORT_MAX_DIGITS = 256;

__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t & total)
{
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if ((threadIdx.x & 63u) >= (uint32_t) off) incl += up;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

__global__ __launch_bounds__(SORT_WAVE) __attribute__((amdgpu_num_vgpr(32)))
void tile_histogram_kernel(const uint32_t * __restrict__ keys, uint64_t n, uint32_t shift, uint32_t ndigits, uint32_t ntiles,
                           uint32_t * __restrict__ counts)
{
    __shared__ uint32_t hist[SORT_MAX_DIGITS];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x, mask = ndigits - 1u;
    for (uint32_t d = lane; d < ndigits; d += SORT_WAVE) hist[d] = 0;
    __syncthreads();
    const uint64_t begin = (uint64_t) tile * SORT_TILE;
    const uint64_t end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    if (end - begin == SORT_TILE && (reinterpret_cast<uintptr_t>(keys + begin) & 15u) == 0) {
        const uint4 * k4 = reinterpret_cast<const uint4 *>(keys + begin);       // (tiles start at multiples of 16 KiB)
#pragma unroll 4
        for (uint32_t i = lane; i < SORT_TILE / 4; i += SORT_WAVE) {
            const uint4 k = k4[i];
            atomicAdd(&hist[(k.x >> shift) & mask], 1u);
            atomicAdd(&hist[(k.y >> shift) & mask], 1u);
            atomicAdd(&hist[(k.z >> shift) & mask], 1u);
            atomicAdd(&hist[(k.w >> shift) & mask], 1u);
        }
    } else {
        for (uint64_t i = begin + lane; i < end; i += SORT_WAVE)
            atomicAdd(&hist[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t d = lane; d < ndigits; d += SORT_WAVE) counts[(uint64_t) d * ntiles + tile] = hist[d];
}

__global__ __launch_bounds__(SORT_WAVE) __attribute__((amdgpu_num_vgpr(32)))
void digit_scan_kernel(uint32_t * __restrict__ counts, uint32_t ntiles, uint32_t * __restrict__ totals)
{
    // a lane scans 8 consecutive tiles per round (a round = 512 tiles: few dependent rounds even for thousands of tiles)
    constexpr uint32_t PER = 8;
    const uint32_t lane = threadIdx.x, d = blockIdx.x;
    uint32_t * row = counts + (uint64_t) d * ntiles;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += SORT_WAVE * PER) {
        const uint32_t first = t0 + lane * PER;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            v[j] = first + j < ntiles ? row[first + j] : 0u;
            sum += v[j];
        }
        uint32_t total;
        uint32_t run = carry + wave_exclusive_scan(sum, total);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (first + j < ntiles) row[first + j] = run;
            run += v[j];
        }
        carry += total;
    }
    if (lane == 0) totals[d] = carry;
}

// IMPLICIT: the input values are value_base + position (record numbers / impulse indices: no array to read)
template <bool IMPLICIT, bool WRITE_KEYS>
__global__ __launch_bounds__(SORT_WAVE) __attribute__((amdgpu_num_vgpr(32)))
void scatter_kernel(const uint32_t * __restrict__ keys_in, const uint32_t * __restrict__ values_in, uint32_t value_base, uint64_t n,
                    uint32_t shift, uint32_t bits, uint32_t ntiles, const uint32_t * __restrict__ counts,
                    const uint32_t * __restrict__ totals, uint32_t * __restrict__ keys_out, uint32_t * __restrict__ values_out)
{
    __shared__ uint32_t offset[SORT_MAX_DIGITS];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x, ndigits = 1u << bits, mask = ndigits - 1u;
    // where each digit's items of this tile go: (items of smaller digits) + (items of this digit in earlier tiles)
    {
        const uint32_t per_lane = SORT_MAX_DIGITS / SORT_WAVE;                  // 4 consecutive digits per lane
        uint32_t t[per_lane], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < per_lane; ++j) {
            const uint32_t d = lane * per_lane + j;
            t[j] = d < ndigits ? totals[d] : 0u;
            sum += t[j];
        }
        uint32_t all;
        uint32_t base = wave_exclusive_scan(sum, all);
#pragma unroll
        for (uint32_t j = 0; j < per_lane; ++j) {
            const uint32_t d = lane * per_lane + j;
            if (d < ndigits) offset[d] = base + counts[(uint64_t) d * ntiles + tile];
            base += t[j];
        }
    }
    __syncthreads();
    const uint64_t begin = (uint64_t) tile * SORT_TILE;
    const uint64_t end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint64_t i0 = begin; i0 < end; i0 += 4 * SORT_WAVE) {
        // four rounds of 64 items: their loads leave together, the ranking is in item order
        uint32_t key[4], val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t i = i0 + r * SORT_WAVE + lane;
            key[r] = i < end ? keys_in[i] : 0u;
            val[r] = IMPLICIT ? value_base + (uint32_t) i : (i < end ? values_in[i] : 0u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t i = i0 + r * SORT_WAVE + lane;
            const bool live = i < end;
            const uint32_t digit = (key[r] >> shift) & mask;
            unsigned long long peers = __builtin_amdgcn_ballot_w64(live);       // the lanes that hold an item with MY digit
            for (uint32_t b = 0; b < bits; ++b) {
                const bool set = (digit >> b) & 1u;
                const unsigned long long with = __builtin_amdgcn_ballot_w64(live && set);
                peers &= set ? with : ~with;
            }
            if (live) {
                const uint32_t rank = (uint32_t) __popcll(peers & lt);
                const uint32_t at = offset[digit] + rank;                       // every peer reads before the first one adds (in-order LDS)
                if (rank == 0) atomicAdd(&offset[digit], (uint32_t) __popcll(peers));
                if (WRITE_KEYS) keys_out[at] = key[r];
                values_out[at] = val[r];
            }
        }
    }
}

}  // namespace

static uint32_t sort_tiles(uint64_t n) { return (uint32_t) ((n + SORT_TILE - 1) / SORT_TILE); }

size_t rvb_radix_sort_temp_bytes(uint64_t n)
{
    return ((size_t) sort_tiles(n) * SORT_MAX_DIGITS + SORT_MAX_DIGITS) * sizeof(uint32_t);
}

// Sorts on key bits [begin_bit, end_bit), 8 bits per pass.  The input is (keys, values) — values == nullptr: value_base + position.
// Passes alternate between the two buffer pairs, starting with A; *keys_sorted / *values_sorted say where the result is
// (want_keys == false: the last pass does not write keys; *keys_sorted is then null).  n < 2^32.
hipError_t rvb_radix_sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys, const uint32_t * values, uint32_t value_base,
                                uint32_t * keys_a, uint32_t * values_a, uint32_t * keys_b, uint32_t * values_b, uint64_t n,
                                int begin_bit, int end_bit, bool want_keys, const uint32_t ** keys_sorted, const uint32_t ** values_sorted,
                                hipStream_t s)
{
    *keys_sorted = keys;
    *values_sorted = values;
    if (n == 0 || end_bit <= begin_bit) return hipSuccess;
    if (n >= (1ull << 32) || temp_bytes < rvb_radix_sort_temp_bytes(n)) return hipErrorInvalidValue;
    const uint32_t ntiles = sort_tiles(n);
    uint32_t * counts = static_cast<uint32_t *>(temp);
    uint32_t * totals = counts + (size_t) ntiles * SORT_MAX_DIGITS;
    const int total_bits = end_bit - begin_bit;
    const int passes = (total_bits + 7) / 8;
    const uint32_t * kin = keys;
    const uint32_t * vin = values;
    for (int p = 0; p < passes; ++p) {
        // bits of this pass: spread evenly (20 bits = 7 + 7 + 6) so that no pass scans more digit rows than it needs
        const int done = (total_bits * p) / passes, upto = (total_bits * (p + 1)) / passes;
        const uint32_t bits = (uint32_t) (upto - done), shift = (uint32_t) (begin_bit + done), ndigits = 1u << bits;
        const bool last = p + 1 == passes;
        uint32_t * kout = (p & 1) ? keys_b : keys_a;
        uint32_t * vout = (p & 1) ? values_b : values_a;
        const bool write_keys = !last || want_keys;
        hipLaunchKernelGGL(tile_histogram_kernel, dim3(ntiles), dim3(SORT_WAVE), 0, s, kin, n, shift, ndigits, ntiles, counts);
        hipLaunchKernelGGL(digit_scan_kernel, dim3(ndigits), dim3(SORT_WAVE), 0, s, counts, ntiles, totals);
        if (vin == nullptr) {
            if (write_keys) hipLaunchKernelGGL((scatter_kernel<true, true>), dim3(ntiles), dim3(SORT_WAVE), 0, s, kin, vin, value_base, n, shift, bits, ntiles, counts, totals, kout, vout);
            else hipLaunchKernelGGL((scatter_kernel<true, false>), dim3(ntiles), dim3(SORT_WAVE), 0, s, kin, vin, value_base, n, shift, bits, ntiles, counts, totals, kout, vout);
        } else {
            if (write_keys) hipLaunchKernelGGL((scatter_kernel<false, true>), dim3(ntiles), dim3(SORT_WAVE), 0, s, kin, vin, value_base, n, shift, bits, ntiles, counts, totals, kout, vout);
            else hipLaunchKernelGGL((scatter_kernel<false, false>), dim3(ntiles), dim3(SORT_WAVE), 0, s, kin, vin, value_base, n, shift, bits, ntiles, counts, totals, kout, vout);
        }
        kin = write_keys ? kout : nullptr;
        vin = vout;
    }
    *keys_sorted = kin;
    *values_sorted = vin;
    return hipGetLastError();
}
