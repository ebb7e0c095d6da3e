// kernels_dict.hpp -- harvesting the row dictionary of an explicit system (gfx950).
//
// Given explicit coefficient planes (A0, aW, aE, aS, aN, b: a host-assembled system passed
// through the reference's seam, or the 3-phase / ImpSolid assembly), find the DISTINCT rows and
// give every cell the code of its row, so the system can run on the matrix-free kernels.  Exact:
// rows are compared bit for bit (the 64-bit hash only routes; a third pass verifies every cell
// against the representative of its slot and any mismatch cancels the dictionary).
//
//   k_dict_insert   every cell hashes its row and claims / finds a slot of an open-addressing
//                   table (atomicCAS on the key), counts itself, and the first cell to arrive
//                   becomes the slot's representative
//   (host)          slots ordered by population -> row indices (most frequent rows first, so the
//                   32 commonest rows share one conflict-free LDS bank row), table of rows
//   k_dict_encode   every cell looks its slot up again, checks its row against the
//                   representative's bit for bit, writes its 16-bit code; also notes whether any
//                   cell outside the first / last column has a non-zero right-hand side
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_setup.hpp"

namespace deff {

constexpr unsigned DICT_SLOTS = 1u << 14;          // open-addressing table, power of two

struct DictTable {
    unsigned long long *key;     // [DICT_SLOTS] 0 = empty
    unsigned int *count;         // [DICT_SLOTS] cells per slot
    unsigned long long *rep;     // [DICT_SLOTS] representative cell index + 1 (0 = none yet)
    unsigned int *flags;         // [0] overflow (table full), [1] mismatch, [2] b outside the wall columns
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 29);
}

__device__ __forceinline__ unsigned long long row_hash(const CoefSoA &c, size_t p)
{
    unsigned long long h = 0x243F6A8885A308D3ull;
    h = mix64(h, (unsigned long long)__double_as_longlong(c.a0[p]));
    h = mix64(h, (unsigned long long)__double_as_longlong(c.aW[p]));
    h = mix64(h, (unsigned long long)__double_as_longlong(c.aE[p]));
    h = mix64(h, (unsigned long long)__double_as_longlong(c.aS[p]));
    h = mix64(h, (unsigned long long)__double_as_longlong(c.aN[p]));
    h = mix64(h, (unsigned long long)__double_as_longlong(c.b[p]));
    return h ? h : 1ull;
}

// returns the slot of key h, inserting it when absent; DICT_SLOTS when the table is full
__device__ __forceinline__ unsigned dict_slot(DictTable t, unsigned long long h, bool insert)
{
    unsigned s = (unsigned)(h >> 20) & (DICT_SLOTS - 1);
    for (unsigned probe = 0; probe < DICT_SLOTS; ++probe, s = (s + 1) & (DICT_SLOTS - 1)) {
        unsigned long long k = t.key[s];
        if (k == h) return s;
        if (k == 0) {
            if (!insert) return DICT_SLOTS;
            k = atomicCAS(&t.key[s], 0ull, h);
            if (k == 0 || k == h) return s;
        }
    }
    return DICT_SLOTS;
}

static __global__ void k_dict_insert(CoefSoA c, size_t n, DictTable t)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned s = dict_slot(t, row_hash(c, p), true);
        if (s == DICT_SLOTS) { t.flags[0] = 1; continue; }
        atomicAdd(&t.count[s], 1u);
        if (t.rep[s] == 0) atomicCAS(&t.rep[s], 0ull, (unsigned long long)p + 1);
    }
}

// slot2code[s] = code (row index x 8) of slot s, 0xFFFF when the slot got no row (too many rows)
static __global__ void k_dict_encode(CoefSoA c, size_t n, int nx, int nxt, DictTable t, const uint16_t *__restrict__ slot2code,
                              uint16_t *__restrict__ code)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned s = dict_slot(t, row_hash(c, p), false);
        if (s == DICT_SLOTS || slot2code[s] == 0xFFFFu) { t.flags[1] = 1; code[p] = 0; continue; }
        const size_t q = (size_t)(t.rep[s] - 1);
        const bool same = __double_as_longlong(c.a0[p]) == __double_as_longlong(c.a0[q]) &&
                          __double_as_longlong(c.aW[p]) == __double_as_longlong(c.aW[q]) &&
                          __double_as_longlong(c.aE[p]) == __double_as_longlong(c.aE[q]) &&
                          __double_as_longlong(c.aS[p]) == __double_as_longlong(c.aS[q]) &&
                          __double_as_longlong(c.aN[p]) == __double_as_longlong(c.aN[q]) &&
                          __double_as_longlong(c.b[p]) == __double_as_longlong(c.b[q]);
        if (!same) t.flags[1] = 1;
        const int j = (int)(p % (size_t)nx);
        if (j != 0 && j != nxt - 1 && c.b[p] != 0) t.flags[2] = 1;
        code[p] = slot2code[s];
    }
}

// rows[k*6 + plane] = plane[cell[k]] for the representatives
static __global__ void k_dict_gather(CoefSoA c, const unsigned long long *__restrict__ cell, int nrows, double *__restrict__ rows)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrows) return;
    const size_t p = (size_t)cell[k];
    rows[k * 6 + 0] = c.a0[p]; rows[k * 6 + 1] = c.aW[p]; rows[k * 6 + 2] = c.aE[p];
    rows[k * 6 + 3] = c.aS[p]; rows[k * 6 + 4] = c.aN[p]; rows[k * 6 + 5] = c.b[p];
}

}  // namespace deff
