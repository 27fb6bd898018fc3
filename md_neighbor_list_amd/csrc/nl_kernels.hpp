// nl_kernels.hpp -- hand-written CDNA4 (gfx950, wave64) kernels of the Verlet-list hot path.
//
// Pipeline of one build (reference lines in each kernel's comment):
//   k_hash      cell hash + rank of each particle inside its cell        (a3)
//   scan        exclusive scan of the cell histogram -> cell_start      (a4)
//   k_reorder   positions physically moved into cell order, id carried  (a5: the reorder the reference left dead)
//   k_sweep<COUNT>  27-cell pair search, counts only                    (a7)
//   scan        exclusive scan of the counts -> key_pointer (CSR)       (a9)
//   k_sweep<FILL>   27-cell pair search, wave-ballot compaction into the final CSR rows (a7+a8+a9)
//
// Arithmetic contract (bit-exact pair set vs the scalar CPU class): d = qj - qi per component,
// r2 = (dx*dx + dy*dy) + dz*dz with separately rounded multiplies and adds -- this TU is compiled with
// -ffp-contract=off and uses the explicit __f*_rn forms -- and a pair is kept unless r2 > rc2
// (neighlist_cpu.hpp:219-223).  No MFMA: nothing here is a contraction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace nl {

constexpr int WAVE = 64;

// bits of the device status word
enum : uint32_t { ST_OUT_OF_BOX = 1u, ST_CAPACITY = 2u, ST_DOMAIN = 4u, ST_INDEX_OVERFLOW = 8u };

// One cell-sorted particle. gid = the id written into neighbour rows and used for the i<j half-list rule.
template <typename T> struct Pos;
template <> struct alignas(16) Pos<float> {
  float x, y, z;
  int32_t gid;
};
template <> struct alignas(32) Pos<double> {
  double x, y, z;
  int32_t gid;
  int32_t row;
};

template <typename T> struct Grid {
  T ims[3];          // 1/ms rounded to T (neighlist_cpu.hpp:409-411)
  int32_t m[3];      // global mesh (neighlist_cpu.hpp:384-386)
  int32_t mzl;       // z layers held locally: m[2], or owned + 2 ghost layers for a slab
  int32_t z_origin;  // global z layer of local layer 0
  int32_t slab;      // 0: periodic wrap in z, every cell owned; 1: layers 0 and mzl-1 are ghosts
  int32_t n_rows;    // particles [0, n_rows) are owned (get rows), [n_rows, n) are ghosts
  // minimum-image mode (nl_set_periodic; not in the reference, which wraps cells but never distances):
  int32_t dbg;       // diagnostics (NL_DEBUG_FLAGS): 512 = binning kernels without the keep_in_flight of their loads
  int32_t pbc;       // 1: a particle whose cell index was wrapped (or that sits in a slab's wrapped ghost layer) is
                     //    stored at its periodic image next to that cell: coordinate -+ L
  int32_t z_first;   // slab: global layer that local layer 0 stands for, z_lo - 1 (may be -1)
  T L[3];            // box lengths rounded to T
};

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }

// GenHash(q) + ApplyPBC, neighlist_cpu.hpp:51-66: idx = (int32)(q * ims) truncated, one +-m wrap, then
// idx.x + (idx.y + idx.z*my)*mx -- here with the z index taken relative to the local slab.
// Returns -1 where the reference would index out of bounds, -2 for a particle outside this rank's layers.
template <typename T>
__device__ __forceinline__ int32_t local_cell(const Grid<T>& g, T x, T y, T z, int32_t* lz_out,
                                              int32_t* row_out = nullptr, T* shift_out = nullptr) {
  const T t[3] = {mul_rn(x, g.ims[0]), mul_rn(y, g.ims[1]), mul_rn(z, g.ims[2])};
  int32_t idx[3];
  T sh[3] = {0, 0, 0};  // minimum-image mode: what to add to the coordinate so that it lies in / next to its cell
  bool bad = false;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if (!(t[d] > (T)-2147483000.0 && t[d] < (T)2147483000.0)) bad = true;  // NaN / overflow: UB in the reference
    int32_t v = (int32_t)t[d];
    // (minimum-image mode takes the floor: the reference's truncation files a particle at -0.3 cells into cell 0,
    // harmless in its open box, wrong for images)
    if (g.pbc && t[d] < (T)0 && (T)v != t[d]) v -= 1;
    if (v < 0) v += g.m[d], sh[d] = g.L[d];
    if (v >= g.m[d]) v -= g.m[d], sh[d] = -g.L[d];
    if (v < 0 || v >= g.m[d]) bad = true;
    idx[d] = v;
  }
  if (bad) return -1;
  int32_t lz = idx[2] - g.z_origin;
  if (lz < 0) lz += g.m[2];
  if (lz >= g.mzl) return -2;
  if (g.slab) {  // the layer this local layer stands for may lie beyond the box end: the particle is its image there
    const int32_t acting = g.z_first + lz;
    if (acting < 0) sh[2] -= g.L[2];
    if (acting >= g.m[2]) sh[2] += g.L[2];
  }
  if (shift_out) shift_out[0] = sh[0], shift_out[1] = sh[1], shift_out[2] = sh[2];
  *lz_out = lz;
  if (row_out) *row_out = idx[1] + lz * g.m[1];  // the row of x-cells the particle lies in
  return idx[0] + (idx[1] + lz * g.m[1]) * g.m[0];
}

// Takes loaded values in an empty asm: whatever was loaded before it is in flight together.  The scheduler, left alone,
// keeps few registers live and waits for every load before it issues the next (one memory round trip per load).
__device__ __forceinline__ void keep_in_flight(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_in_flight(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_in_flight(int32_t& v) { asm volatile("" : "+v"(v)); }

template <typename T>
__device__ __forceinline__ void load_xyz(const T* __restrict__ q, int32_t stride, int32_t i, T& x, T& y, T& z) {
  const T* p = q + (size_t)i * stride;
  if (stride == 4) {
    if constexpr (sizeof(T) == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      x = v.x, y = v.y, z = v.z;
    } else {
      const double2 a = *reinterpret_cast<const double2*>(p);
      x = a.x, y = a.y, z = p[2];
    }
  } else {
    x = p[0], y = p[1], z = p[2];
  }
}

// a3: make_mesh (neighlist_gpu.hpp:26-41) / MakeMeshidOfPtcl (neighlist_cpu.hpp:134-144).
// One thread per particle; the returning atomic on the cell histogram is the particle's rank in its cell.
template <typename T>
__global__ void __launch_bounds__(256) k_hash(const T* __restrict__ q, int32_t stride, int32_t n, Grid<T> g,
                                               int32_t* __restrict__ cell_count, int32_t* __restrict__ rank,
                                               uint32_t* __restrict__ status) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T x, y, z;
  load_xyz(q, stride, i, x, y, z);
  int32_t lz = 0;
  const int32_t c = local_cell(g, x, y, z, &lz);
  if (c < 0) {
    atomicOr(status, c == -1 ? ST_OUT_OF_BOX : ST_DOMAIN);
    rank[i] = -1;
    return;
  }
  if (g.slab) {
    const bool in_owned_layer = lz >= 1 && lz < g.mzl - 1;
    if (in_owned_layer != (i < g.n_rows)) atomicOr(status, ST_DOMAIN);
  }
  rank[i] = atomicAdd(&cell_count[c], 1);
}

// a5: the physical reorder (dead code in the reference: CopyGather neighlist_gpu.hpp:144-151,
// SortPtclData neighlist_cpu.hpp:176-180).  sorted[cell_start[c] + rank] = {x, y, z, id}.
template <typename T>
__global__ void __launch_bounds__(256) k_reorder(const T* __restrict__ q, int32_t stride,
                                                  const int32_t* __restrict__ gid, int32_t n, Grid<T> g,
                                                  const int32_t* __restrict__ cell_start,
                                                  const int32_t* __restrict__ rank, Pos<T>* __restrict__ sorted,
                                                  int32_t* __restrict__ sorted_row, int32_t* __restrict__ sorted_gid) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i];
  if (r < 0) return;
  T x, y, z;
  load_xyz(q, stride, i, x, y, z);
  int32_t lz = 0;
  T sh[3];
  const int32_t c = local_cell(g, x, y, z, &lz, nullptr, sh);
  const int32_t dst = cell_start[c] + r;
  Pos<T> p;
  p.x = x, p.y = y, p.z = z;
  if (g.pbc) p.x = add_rn(x, sh[0]), p.y = add_rn(y, sh[1]), p.z = add_rn(z, sh[2]);
  if (gid == reinterpret_cast<const int32_t*>(1)) {  // NL_GID_IN_W: the id travels in the w component of the Vec
    if constexpr (sizeof(T) == 4) p.gid = __float_as_int(q[(size_t)i * 4 + 3]);
    else p.gid = (int32_t)__double_as_longlong(q[(size_t)i * 4 + 3]);
  } else {
    p.gid = gid ? gid[i] : i;
  }
  if constexpr (sizeof(T) == 8) p.row = i;
  sorted[dst] = p;
  sorted_row[dst] = i;
  sorted_gid[dst] = p.gid;
}

// ---------------------------------------------------------------------------------------------- scans
// Exclusive scan of int32 counts (a4: thrust reduce_by_key + inclusive_scan neighlist_gpu.hpp:153-175 /
// MakeNextDest neighlist_cpu.hpp:146-152; a9: MakeNeighListForEachPtcl neighlist_cpu.hpp:361-367).
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;  // per thread, loaded as 4 x int4
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int32_t wave_incl_scan(int32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int32_t u = __shfl_up(v, d, WAVE);
    if (lane >= d) v += u;
  }
  return v;
}
// Inclusive scan over lanes 0..31 with DPP row shifts (no LDS traffic, 5 VALU instructions): shr 1,2,4,8 inside
// each row of 16 lanes, then row_bcast:15 carries row 0's total into row 1.  Lanes >= 32 are not meaningful.
__device__ __forceinline__ int32_t scan32_dpp(int32_t v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  return v;
}
// Sum over the 64 lanes (6 DPP adds + one v_readlane): wave-uniform result.
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t x) {
  int32_t v = (int32_t)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int64_t wave_incl_scan64(int64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int64_t u = __shfl_up(v, d, WAVE);
    if (lane >= d) v += u;
  }
  return v;
}

__device__ __forceinline__ void load_items(const int32_t* __restrict__ in, int64_t n, int64_t base, int32_t* v) {
  if (base + SCAN_ITEMS <= n) {
    const int4* p = reinterpret_cast<const int4*>(in + base);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; k++) {
      const int4 a = p[k];
      v[4 * k] = a.x, v[4 * k + 1] = a.y, v[4 * k + 2] = a.z, v[4 * k + 3] = a.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) v[k] = base + k < n ? in[base + k] : 0;
  }
}

// out[i] = exclusive prefix; out[n] = total.  OFF = int32_t (the reference's key_pointer type, neighlist_cpu.hpp:15,29):
// flags ST_INDEX_OVERFLOW when the total exceeds INT32_MAX; OFF = int64_t (wide builds: lists beyond 2^31 entries,
// BASELINE config 4 on one device) never overflows.
//
// ONE launch (chained scan with look-back): a block takes its number from a ticket counter (so every block with a
// smaller number is running or done), publishes the sum of its 4096 items in look[b] (flag 1), adds up the entries of
// the blocks before it -- wave 0 reads 64 of them per step and stops at the nearest one that already holds an
// inclusive prefix (flag 2) -- and publishes its own prefix.  look[] = flag << 62 | value; look[nb_max], [nb_max + 1]
// hold the ticket and the count of blocks that are through with look[]: the last of those clears the entries and the
// two counters, so the array is all zero again when the kernel ends (no memset per build, graph replays included).
// Everything a block needs from another one is inside the 64-bit entry, so all accesses to look[] are RELAXED
// device-scope atomics (they bypass the per-XCD L2).  Acquire / release here would write back and invalidate the
// whole L2 of the XCD at every step: the first version did, and took 36 us for 256 blocks.
constexpr uint64_t SCAN_FLAG_SUM = 1ull << 62, SCAN_FLAG_PREFIX = 2ull << 62, SCAN_VALUE = (1ull << 62) - 1;
__device__ __forceinline__ uint64_t scan_look_load(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void scan_look_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename OFF>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_chained(const int32_t* __restrict__ in, int64_t n,
                                                                uint64_t* __restrict__ look, int32_t nb_max,
                                                                int64_t* __restrict__ total,
                                                                OFF* __restrict__ out,
                                                                uint32_t* __restrict__ status,
                                                                uint32_t* __restrict__ total_split) {
  __shared__ int32_t wsum[SCAN_THREADS / WAVE];
  __shared__ int64_t before_s;
  __shared__ int32_t block_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t* counters = reinterpret_cast<uint32_t*>(look + nb_max);
  if (tid == 0) block_s = (int32_t)__hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int32_t b = block_s, nb = (int32_t)gridDim.x;
  const int64_t base = (int64_t)b * SCAN_BLOCK + (int64_t)tid * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  load_items(in, n, base, v);
  int32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += v[k];
  const int32_t inc = wave_incl_scan(s, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t woff = 0, sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_THREADS / WAVE; k++) {
    if (k < w) woff += wsum[k];
    sum += wsum[k];
  }
  if (w == 0) {
    if (lane == 0 && b > 0) scan_look_store(&look[b], SCAN_FLAG_SUM | (uint64_t)sum);
    int64_t before = 0;
    for (int32_t hi = b - 1; hi >= 0; hi -= WAVE) {  // wave-uniform
      const int32_t idx = hi - lane;
      uint64_t e = SCAN_FLAG_PREFIX;  // (in front of block 0: prefix 0)
      for (;;) {
        if (idx >= 0) e = scan_look_load(&look[idx]);
        if (__ballot((e >> 62) == 0) == 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      const uint64_t has_prefix = __ballot((e >> 62) == 2);  // never zero in the last step (idx < 0 lanes, or block 0)
      const int first = has_prefix ? __builtin_ctzll(has_prefix) : WAVE;
      int64_t part = lane <= first ? (int64_t)(e & SCAN_VALUE) : 0;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, WAVE);
      before += part;
      if (has_prefix) break;
    }
    if (lane == 0) {
      scan_look_store(&look[b], SCAN_FLAG_PREFIX | (uint64_t)(before + sum));
      before_s = before;
    }
  }
  __syncthreads();
  const int64_t before = before_s;
  OFF run = (OFF)before + (OFF)(woff + inc - s);
  if (sizeof(OFF) == 4 && base + SCAN_ITEMS <= n) {
    int4* p = reinterpret_cast<int4*>(out + base);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; k++) {
      int4 a;
      a.x = (int32_t)run, run += v[4 * k];
      a.y = (int32_t)run, run += v[4 * k + 1];
      a.z = (int32_t)run, run += v[4 * k + 2];
      a.w = (int32_t)run, run += v[4 * k + 3];
      p[k] = a;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      if (base + k < n) out[base + k] = run;
      run += v[k];
    }
  }
  if (b == nb - 1 && tid == 0) {
    const int64_t t = before + sum;
    total[0] = t;
    out[n] = (OFF)t;
    if (sizeof(OFF) == 4 && t > 2147483647LL) atomicOr(status, ST_INDEX_OVERFLOW);
    if (total_split) {  // the grand total next to the status word: one small device->host copy per build
      total_split[0] = (uint32_t)t;
      total_split[1] = (uint32_t)((unsigned long long)t >> 32);
    }
  }
  // through with look[] (own prefix published before this, by the barrier above): the last block to get here clears it
  if (tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);  // this thread's store to look[b] has been acknowledged before the count goes up
    block_s = (int32_t)__hip_atomic_fetch_add(&counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (block_s == nb - 1) {
    for (int32_t k = tid; k < nb; k += SCAN_THREADS) scan_look_store(&look[k], 0);
    if (tid == 0) {
      __hip_atomic_store(&counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&counters[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Same scan for short arrays in ONE launch (small boxes: a few thousand cells): a
// single workgroup, every thread scans a contiguous run of K items, the 1024 run totals are scanned through LDS.
constexpr int SCAN_SMALL_MAX = 4096;  // beyond that the per-thread runs get long and the three-kernel scan is faster
template <typename OFF>
__global__ void __launch_bounds__(1024) k_scan_small(const int32_t* __restrict__ in, int32_t n,
                                                      int64_t* __restrict__ total, OFF* __restrict__ out,
                                                      uint32_t* __restrict__ total_split) {
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t K = (n + 1023) / 1024;
  const int32_t b = tid * K, e = min(b + K, n);
  int32_t s = 0;
  for (int32_t i = b; i < e; i++) s += in[i];
  const int32_t inc = wave_incl_scan(s, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t woff = 0, all = 0;
  for (int k = 0; k < 16; k++) {
    const int32_t v = wsum[k];
    woff += k < w ? v : 0;
    all += v;
  }
  int32_t run = woff + inc - s;
  for (int32_t i = b; i < e; i++) {
    const int32_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (tid == 0) {
    out[n] = all;
    total[0] = all;
    if (total_split) total_split[0] = (uint32_t)all, total_split[1] = 0;
  }
}

// ---------------------------------------------------------------------------------------------- binning
// a3 + a4 + a5 without one global atomic per particle (k_hash is bound by the ~20 G/s rate of scattered returning
// atomics: 50 us per million particles).  Two-level counting sort through LDS:
//   k_bin_rows     every block histograms its chunk of particles over the R = my*mzl ROWS of x-cells in LDS and
//                  reserves its share of each row with one returning global atomic per (block, row);
//   (scan)         exclusive scan of the R row totals -> row_start;
//   k_bin_scatter  every block re-reads its chunk and appends {x,y,z,id} + input index to its rows' reserved
//                  ranges (LDS cursors) in a temporary array: particles are now grouped by row;
//   k_bin_cells    one workgroup per row: LDS histogram over the row's mx cells -> cell_start, then the row's
//                  particles are placed in cell order into sorted[] / sorted_row[].
// The cell of a particle is the reference's (local_cell above); x-cell and row are its two factors.
constexpr int BIN_THREADS = 1024;
constexpr int BIN_UNROLL = 4;  // particles per thread whose loads are in flight together (k_bin_rows, k_bin_scatter)
constexpr int BIN_MAX_ROWS = 12288;  // 48 KiB of LDS counters
constexpr int BIN_MAX_MX = 4096;

// One pass of the binning kernels over a range of particles.  A whole build is one pass over [0, n).  A slab build that
// overlaps the halo exchange (nl_make_list_slab_begin / _finish) runs two: the owned particles [0, n_rows) first --
// their rows of x-cells are the layers 1 .. mzl-2, which start right behind the n_ghost_lo particles of ghost layer 0
// in the sorted array -- and the ghosts [n_rows, n) when they have arrived: layer 0 at the front, layer mzl-1 behind
// the owned particles.  The three regions of the sorted array never overlap, so the passes are independent.
struct BinPhase {
  int32_t i_beg, i_end;        // particles of this pass
  int32_t split_row;           // a row r starts at (r < split_row ? base_lo : base_hi) + (this pass's particles in rows < r)
  int32_t base_lo, base_hi;
  int32_t cells_row0, cells_n0, cells_row1;  // k_bin_cells: block b < cells_n0 -> row cells_row0 + b, else cells_row1 + b - cells_n0
  int32_t check_lo;            // >= 0: this pass must hold exactly that many particles in rows < split_row (ST_DOMAIN)
  const int32_t* dyn;          // != nullptr (nl_make_list_distributed, whole-build passes only): the pass ends at particle
                               // g.n_rows + dyn[0] + dyn[1] -- the ghost counts of this build, known on the device only (the
                               // host sizes the launch for i_end, an upper bound, and never waits for them)
};

template <typename T>
__global__ void __launch_bounds__(BIN_THREADS) k_bin_rows(const T* __restrict__ q, int32_t stride, int32_t n, int32_t chunk,
                                                          Grid<T> g, int32_t nrows, int32_t* __restrict__ row_count,
                                                          int32_t* __restrict__ blk_base, uint32_t* __restrict__ status,
                                                          BinPhase ph) {
  __shared__ int32_t hist[BIN_MAX_ROWS];
  const int tid = threadIdx.x;
  for (int32_t r = tid; r < nrows; r += BIN_THREADS) hist[r] = 0;
  __syncthreads();
  const int32_t i_end = ph.dyn ? min(ph.i_end, g.n_rows + ph.dyn[0] + ph.dyn[1]) : ph.i_end;
  const int32_t beg = ph.i_beg + blockIdx.x * chunk, end = min(beg + chunk, i_end);
  // BIN_UNROLL particles per thread and trip, all loads first: one memory round trip per trip instead of one per
  // particle (a chunk is 4 particles per thread).
  for (int32_t i0 = beg + tid; i0 < end; i0 += BIN_UNROLL * BIN_THREADS) {
    T x[BIN_UNROLL], y[BIN_UNROLL], z[BIN_UNROLL];
#pragma unroll
    for (int u = 0; u < BIN_UNROLL; u++) load_xyz(q, stride, min(i0 + u * BIN_THREADS, end - 1), x[u], y[u], z[u]);
    if (!(g.dbg & 512)) {
#pragma unroll
      for (int u = 0; u < BIN_UNROLL; u++) keep_in_flight(x[u]), keep_in_flight(y[u]), keep_in_flight(z[u]);
    }
#pragma unroll
    for (int u = 0; u < BIN_UNROLL; u++) {
      const int32_t i = i0 + u * BIN_THREADS;
      if (i >= end) break;
      int32_t lz = 0, row = 0;
      const int32_t c = local_cell(g, x[u], y[u], z[u], &lz, &row);
      if (c < 0) {
        atomicOr(status, c == -1 ? ST_OUT_OF_BOX : ST_DOMAIN);
        continue;
      }
      if (g.slab) {
        const bool in_owned_layer = lz >= 1 && lz < g.mzl - 1;
        if (in_owned_layer != (i < g.n_rows)) atomicOr(status, ST_DOMAIN);
      }
      atomicAdd(&hist[row], 1);
    }
  }
  __syncthreads();
  for (int32_t r = tid; r < nrows; r += BIN_THREADS) {
    const int32_t h = hist[r];
    blk_base[(size_t)blockIdx.x * nrows + r] = h ? atomicAdd(&row_count[r], h) : 0;
  }
}

// UNROLL: particles per thread whose loads are in flight together: 8 where a chunk is 8 particles per thread (one round
// trip per chunk: binning 2 + 3 at cfg 2 41.6 -> 38.1 us), 4 for the smaller chunks of small boxes.  (k_bin_rows is
// 1 us slower with 8.)
template <typename T, int UNROLL = BIN_UNROLL>
__global__ void __launch_bounds__(BIN_THREADS) k_bin_scatter(const T* __restrict__ q, int32_t stride,
                                                             const int32_t* __restrict__ gid, int32_t n, int32_t chunk,
                                                             Grid<T> g, int32_t nrows, const int32_t* __restrict__ row_count,
                                                             int32_t* __restrict__ row_start,
                                                             const int32_t* __restrict__ blk_base, Pos<T>* __restrict__ tmp,
                                                             int32_t* __restrict__ tmp_row, uint32_t* __restrict__ status,
                                                             BinPhase ph) {
  __shared__ int32_t cursor[BIN_MAX_ROWS];
  __shared__ int32_t wsum[BIN_THREADS / WAVE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // Every block scans the R row totals itself (a few thousand adds) instead of waiting for a scan kernel: one
  // dependent launch less per build.  Thread t owns a run of K consecutive rows; block 0 publishes row_start for
  // k_bin_cells.
  {
    const int32_t K = (nrows + BIN_THREADS - 1) / BIN_THREADS;
    const int32_t b = min(tid * K, nrows), e = min(b + K, nrows);
    int32_t ssum = 0;
    for (int32_t i = b; i < e; i++) ssum += row_count[i];
    const int32_t inc = wave_incl_scan(ssum, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t woff = 0, all = 0;
#pragma unroll
    for (int k = 0; k < BIN_THREADS / WAVE; k++) {
      const int32_t v = wsum[k];
      woff += k < w ? v : 0;
      all += v;
    }
    int32_t run = woff + inc - ssum;
    for (int32_t i = b; i < e; i++) {
      const int32_t start = run + (i < ph.split_row ? ph.base_lo : ph.base_hi);
      cursor[i] = start + blk_base[(size_t)blockIdx.x * nrows + i];
      if (blockIdx.x == 0) {
        row_start[i] = start;
        if (i == ph.split_row && ph.check_lo >= 0 && run != ph.check_lo) atomicOr(status, ST_DOMAIN);
      }
      run += row_count[i];
    }
    if (blockIdx.x == 0 && tid == 0) row_start[nrows] = all + ph.base_hi;
  }
  __syncthreads();
  const int32_t i_end = ph.dyn ? min(ph.i_end, g.n_rows + ph.dyn[0] + ph.dyn[1]) : ph.i_end;
  const int32_t beg = ph.i_beg + blockIdx.x * chunk, end = min(beg + chunk, i_end);
  const bool gid_in_w = gid == reinterpret_cast<const int32_t*>(1);  // NL_GID_IN_W: the id travels in the w component
  for (int32_t i0 = beg + tid; i0 < end; i0 += UNROLL * BIN_THREADS) {  // all loads of a trip first (see k_bin_rows)
    T x[UNROLL], y[UNROLL], z[UNROLL];
    int32_t id[UNROLL];
    // three loops, one per id source, each free of branches between its loads: a load under a branch is waited for at
    // the join (one memory round trip per particle instead of one per trip)
    if (gid_in_w) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const int32_t i = min(i0 + u * BIN_THREADS, end - 1);
        load_xyz(q, stride, i, x[u], y[u], z[u]);
        if constexpr (sizeof(T) == 4) id[u] = __float_as_int(q[(size_t)i * 4 + 3]);
        else id[u] = (int32_t)__double_as_longlong(q[(size_t)i * 4 + 3]);
      }
    } else if (gid) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const int32_t i = min(i0 + u * BIN_THREADS, end - 1);
        load_xyz(q, stride, i, x[u], y[u], z[u]);
        id[u] = gid[i];
      }
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const int32_t i = min(i0 + u * BIN_THREADS, end - 1);
        load_xyz(q, stride, i, x[u], y[u], z[u]);
        id[u] = i;
      }
    }
    if (!(g.dbg & 512)) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) keep_in_flight(x[u]), keep_in_flight(y[u]), keep_in_flight(z[u]), keep_in_flight(id[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int32_t i = i0 + u * BIN_THREADS;
      if (i >= end) break;
      int32_t lz = 0, row = 0;
      T sh[3];
      const int32_t c = local_cell(g, x[u], y[u], z[u], &lz, &row, sh);
      if (c < 0) continue;
      const int32_t dst = atomicAdd(&cursor[row], 1);
      Pos<T> p;
      p.x = x[u], p.y = y[u], p.z = z[u];
      // (minimum-image mode: x keeps its value until k_bin_cells has derived the x-cell from it)
      if (g.pbc) p.y = add_rn(y[u], sh[1]), p.z = add_rn(z[u], sh[2]);
      p.gid = id[u];
      if constexpr (sizeof(T) == 8) p.row = i;
      tmp[dst] = p;
      tmp_row[dst] = i;
    }
  }
}

// FINE (nl_rows.hpp, fp32 open box): the row's particles are sorted by (quarter of the cell along z, x-cell) instead of
// the x-cell alone: the row of x-cells becomes four FINE ROWS, each contiguous in x, and the table written is
// fine_start[(r * mx + cx) * 4 + qz] (4 M + 1 entries) in the place of cell_start.  The quarter a particle lies in
// comes from the fraction of the same rounded product t = z * ims the reference truncates for the cell index
// (neighlist_cpu.hpp:51-59): floor(4 (t - trunc(t))) -- a negative fraction (a particle just below the box: the
// reference's truncation files t = -0.3 into cell 0) counts as quarter 0.
constexpr int BIN_FINE_MAX_MX = 2048;
template <typename T, bool FINE = false>
__global__ void __launch_bounds__(256) k_bin_cells(Grid<T> g, int32_t nrows, const int32_t* __restrict__ row_start,
                                                   const Pos<T>* __restrict__ tmp, const int32_t* __restrict__ tmp_row,
                                                   int32_t* __restrict__ cell_start, Pos<T>* __restrict__ sorted,
                                                   int32_t* __restrict__ sorted_row, int32_t* __restrict__ sorted_gid,
                                                   BinPhase ph) {
  __shared__ int32_t cnt[FINE ? 4 * BIN_FINE_MAX_MX : BIN_MAX_MX];
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t bx = blockIdx.x, mx = g.m[0];
  const int32_t nb = FINE ? 4 * mx : mx;  // bins of the row
  const int32_t r = bx < ph.cells_n0 ? ph.cells_row0 + bx : ph.cells_row1 + (bx - ph.cells_n0);
  const int32_t beg = row_start[r], end = row_start[r + 1];
  for (int32_t c = tid; c < nb; c += 256) cnt[c] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  // x-cell exactly as local_cell computes it (the particle passed the range checks in k_bin_rows)
  auto xcell1 = [&](T x) {
    const T tx = mul_rn(x, g.ims[0]);
    int32_t v = (int32_t)tx;
    if (g.pbc && tx < (T)0 && (T)v != tx) v -= 1;
    if (v < 0) v += mx;
    if (v >= mx) v -= mx;
    return v;
  };
  auto quarter = [&](T q, T ims) {  // first two fractional bits of q * ims (the subtraction is exact)
    const T t = mul_rn(q, ims);
    const T f = sub_rn(t, (T)(int32_t)t);
    return (f >= (T)0.25 ? 1 : 0) + (f >= (T)0.5 ? 1 : 0) + (f >= (T)0.75 ? 1 : 0);
  };
  auto bin_of = [&](const Pos<T>& p) {
    const int32_t c = xcell1(p.x);
    if constexpr (FINE) return quarter(p.z, g.ims[2]) * mx + c;
    else return c;
  };
  // The row's particles stay in registers between the histogram and the placement (rows of up to BC_KEEP * 256: one
  // read of tmp instead of two, and all of a thread's loads in one round trip); longer rows are read twice.
  constexpr int BC_KEEP = 6;
  const bool keep = end - beg <= BC_KEEP * 256 && end > beg && !(g.dbg & 512);
  Pos<T> pk[BC_KEEP];
  int32_t rk[BC_KEEP];
  if (keep) {
#pragma unroll
    for (int u = 0; u < BC_KEEP; u++) {
      const int32_t k = min(beg + tid + u * 256, end - 1);
      pk[u] = tmp[k];
      rk[u] = tmp_row[k];
    }
#pragma unroll
    for (int u = 0; u < BC_KEEP; u++) {
      keep_in_flight(pk[u].x), keep_in_flight(pk[u].y), keep_in_flight(pk[u].z), keep_in_flight(pk[u].gid), keep_in_flight(rk[u]);
      if constexpr (sizeof(T) == 8) keep_in_flight(pk[u].row);
    }
#pragma unroll
    for (int u = 0; u < BC_KEEP; u++)
      if (beg + tid + u * 256 < end) atomicAdd(&cnt[bin_of(pk[u])], 1);
  } else {
    for (int32_t k = beg + tid; k < end; k += 256) atomicAdd(&cnt[bin_of(tmp[k])], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..nb) in place, 256 entries at a time, and the row's slice of cell_start
  for (int32_t base = 0; base < nb; base += 256) {
    const int32_t c = base + tid;
    const int32_t v = c < nb ? cnt[c] : 0;
    const int32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int k = 0; k < w; k++) woff += wsum[k];
    const int32_t carry = carry_s;
    const int32_t excl = carry + woff + inc - v;
    if (c < nb) {
      cnt[c] = excl;
      // (FINE: bin c = quarter * mx + x-cell is filed at (r * mx + x-cell) * 4 + quarter: rows_fine_index)
      if constexpr (FINE) cell_start[((size_t)r * mx + (c % mx)) * 4 + c / mx] = beg + excl;
      else cell_start[(size_t)r * nb + c] = beg + excl;
    }
    __syncthreads();
    if (tid == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (r == nrows - 1 && tid == 0) cell_start[(size_t)nrows * nb] = end;
  auto place = [&](Pos<T> p, int32_t row_of) {
    const int32_t dst = beg + atomicAdd(&cnt[bin_of(p)], 1);
    if (g.pbc) {  // minimum-image mode: a wrapped x index means the particle is stored at its image
      const T tx = mul_rn(p.x, g.ims[0]);
      int32_t v = (int32_t)tx;
      if (tx < (T)0 && (T)v != tx) v -= 1;
      if (v < 0) p.x = add_rn(p.x, g.L[0]);
      if (v >= mx) p.x = add_rn(p.x, -g.L[0]);
    }
    sorted[dst] = p;
    sorted_row[dst] = row_of;
    sorted_gid[dst] = p.gid;  // the ids alone, 4 bytes apart: what the expansion kernel stages
  };
  if (keep) {
#pragma unroll
    for (int u = 0; u < BC_KEEP; u++)
      if (beg + tid + u * 256 < end) place(pk[u], rk[u]);
  } else {
    for (int32_t k = beg + tid; k < end; k += 256) place(tmp[k], tmp_row[k]);
  }
}

// ---------------------------------------------------------------------------------------------- sweep
// a7 + a8 + a9: the pair search.  One workgroup per i-cell.  The particles of the 27 neighbour cells
// (9 contiguous x-runs of the cell-sorted array, periodic wrap as MakeNeighMeshId neighlist_gpu.hpp:125-142 /
// ApplyPBC neighlist_cpu.hpp:61-66) are staged back-to-back in LDS, so every 64-lane j-tile is dense.
// Each wave then takes groups of up to G i-particles of the cell into SGPRs and walks the tiles:
// lanes own j, i is wave-uniform, the cut-off mask comes out of v_cmp as a 64-bit SGPR pair (the wave64
// counterpart of the reference's __ballot/__popc append, kernel_impl.cuh:272-280), and
//   COUNT: the row count is advanced by s_bcnt1 of the mask,
//   FILL : accepted j ids are written at key_pointer[row] + count + mbcnt(mask), i.e. straight into the final
//          CSR row (no staging row + transpose as in kernel_impl.cuh:218-239).
// Half-list rule: the pair is kept in the row of the smaller id (RegistInteractPair neighlist_cpu.hpp:225-236),
// so lane j is accepted only if gid_j > gid_i; that also drops j == i.
enum { MODE_COUNT = 0, MODE_FILL = 1, MODE_COUNT_MASKS = 2 };  // COUNT_MASKS: COUNT that also keeps every hit mask

// n / d for 0 <= n < 2^31 and a divisor fixed at launch: q = (umulhi(n, m) + n) >> s  (Granlund-Montgomery
// round-up form), evaluated without the 33-bit overflow.  The host fills m and s (fastdiv_make in nl_api.hip).
struct FastDiv {
  uint32_t d, m, s;
};
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  const uint32_t t = __umulhi(n, f.m);
  return (t + ((n - t) >> 1)) >> (f.s - 1);
}

template <typename T> struct SweepArgs {
  const Pos<T>* __restrict__ sorted;
  const int32_t* __restrict__ sorted_row;
  const int32_t* __restrict__ sorted_gid;  // sorted[k].gid, compact (k_fill_masks reads ids only: 4 x fewer cache lines)
  const int32_t* __restrict__ cell_start;
  int32_t mx, my, mzl, slab;
  FastDiv div_mx, div_my;         // i-cell index -> (cx, cy, cz) without integer division
  T rc2;                          // largest T value <= rc*rc in double, so !(r2 > rc2) == !((double)r2 > rc2_double)
  int32_t* __restrict__ count;    // [n_rows] number_of_partners (COUNT writes, FILL reads nothing from it)
  int32_t* __restrict__ progress; // [n_rows] scratch, only touched when a stencil needs more than one LDS batch
  const void* __restrict__ key_pointer;  // [n_rows + 1] int32 offsets, or int64 when `wide`
  int32_t wide;                   // this build's list may exceed 2^31 entries: key_pointer / base_sorted hold int64
  int32_t n_rows;                 // owned particles (rows of the list); slots of ghosts carry row ids >= n_rows
  int32_t* __restrict__ list;
  const int64_t* __restrict__ total;
  int64_t capacity;
  uint32_t* __restrict__ status;
  T L[3];                         // box lengths rounded to T (minimum-image mode)
  int32_t pbc;                    // minimum-image mode: stencil cells reached through the periodic wrap are staged
                                  // at their image, coordinate -+ L (nl_set_periodic; not in the reference)
  T ms[3];                        // cell edge (screened fp64 search: origin of the relative coordinates)
  int32_t z_origin;               // global z layer of local layer 0
  int32_t isplit;                          // two-sweep path: workgroups per cell (each a part of the cell's i-particles)
  int32_t mask_nb;                         // mask rows per sorted slot: 1, or up to FD_NB LDS batches in a dense build
  int32_t* __restrict__ full27_list;       // hand-over list: local cell indices of the cells a kernel leaves to the batched search
  int32_t* __restrict__ full27_count;      // their number (COUNT pass): a meta word next to the status word, zeroed with it
  int32_t* __restrict__ fill_list_count;   // cells the expansion hands to k_fill_list (in full27_list; a meta word of its own)
  uint32_t* __restrict__ masks;  // [rows] low plane of the hit words: 64 x 16 bits per row (mask_store/mask_load): bit t of word l of a sorted slot = staged particle t*64+l accepted (COUNT_MASKS -> k_fill_masks)
  uint8_t* __restrict__ masks_hi;  // [rows] high plane: 64 x 8 bits per row (tiles 16..23; not touched for a stream of <= 16 tiles)
  unsigned long long* dbg_buf;  // diagnostics only: cycle accumulators (dbg & 4)
  int32_t dbg;  // diagnostics only (NL_DEBUG_FLAGS): 1 = skip the search, 2 = skip the staging copy; 0 in production
};

template <typename T> struct SweepCfg;
// Hit masks in memory: a lane keeps one bit per j-tile, at most 24 (a staged stream has at most CAP / 64 = 20 tiles).
// Two planes, each an array of rows of its own: the low 16 bits of the 64 lanes (128 bytes per row, one cache line) and
// their high 8 bits (64 bytes per row) -- every store and load naturally aligned and consecutive over the lanes.  A
// stream of at most 16 tiles (1024 particles: every cell of BASELINE config 3, a quarter of config 2's) has no high
// bits: its rows of the high plane are neither written nor read (round 3; cfg 3: expansion -8 %, 67 MB less traffic
// each way).  The condition is a property of the cell (hi_plane_used), the same in the COUNT sweep and the expansion.
// (Round 2 measurements at cfg 2 / cfg 3, both planes in one 192-byte row: 3-byte words back to back, lane l at byte
// 3 l, with unaligned accesses: build +0.5 % / +1 %; aligned 256-byte rows, one dword per lane: COUNT -0.5 % / -2 %,
// expansion +4 % / -5 % -- the expansion is bound by what it reads.)
constexpr int MASK_LO_BYTES = 2 * WAVE, MASK_HI_BYTES = WAVE, MASK_ROW_BYTES = MASK_LO_BYTES + MASK_HI_BYTES;  // 192 per row in all
__device__ __forceinline__ bool hi_plane_used(int32_t mask_nb, int32_t ntiles) { return mask_nb > 1 || ntiles > 16; }
// NT: non-temporal stores -- the rows are read once, by the expansion, after 200 MB more of them have been written: the
// 4-wave COUNT sweep keeps them out of the L2's way (cfg 2: build -1.2 %; the sparse-box instances are 1.7 % slower with
// them -- their 128 MB of rows are still in the last-level cache when the expansion comes -- and keep plain stores).
template <bool NT = false>
__device__ __forceinline__ void mask_store(uint32_t* masks, uint8_t* masks_hi, size_t row, int lane, uint32_t bits, bool hi) {
  uint16_t* const lo_p = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(masks) + row * MASK_LO_BYTES) + lane;
  uint8_t* const hi_p = masks_hi + row * MASK_HI_BYTES + lane;
  if constexpr (NT) {
    __builtin_nontemporal_store((uint16_t)bits, lo_p);
    if (hi) __builtin_nontemporal_store((uint8_t)(bits >> 16), hi_p);  // (uniform)
  } else {
    *lo_p = (uint16_t)bits;
    if (hi) *hi_p = (uint8_t)(bits >> 16);  // (uniform)
  }
}
// The expansion reads every mask row once: in the 2-wave instances as non-temporal loads, which leaves the L2 to the ids
// and list offsets the kernel reads 27 times / at random (cfg 2: expansion 0.181 -> 0.174 ms; the 1-wave instance of
// sparse boxes is 4 % slower with them and keeps plain loads).
template <bool NT, typename P> __device__ __forceinline__ P mask_ld(const P* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
__device__ __forceinline__ uint32_t mask_load(const uint32_t* masks, const uint8_t* masks_hi, size_t row, int lane) {  // both planes
  return (uint32_t)reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(masks) + row * MASK_LO_BYTES)[lane] |
         (uint32_t)(masks_hi + row * MASK_HI_BYTES)[lane] << 16;
}
template <> struct SweepCfg<float> { static constexpr int CAP = 1280; };   // 20 KB of LDS: 8 workgroups = 32 waves per CU
template <> struct SweepCfg<double> { static constexpr int CAP = 1280; };  // 40 KB of LDS (registers, not LDS, limit fp64 occupancy)

constexpr int SWEEP_WAVES = 4;
constexpr int SWEEP_G = 5;
constexpr int NSEG = 18;

// ---- screened fp64 search.  The fp64 vector rate of the chip is half its fp32 rate, and the fp64 distance test is what
// bounds the fp64 builds (BASELINE config 5).  The fp64 sweeps therefore stage the stream as FLOATS RELATIVE TO THE
// CENTRE OF THE i-CELL, decide every pair whose fp32 r2 lies outside a band around rc2 in fp32, and evaluate only the
// pairs inside the band (a few in a million) with the reference's exact fp64 expression on the original coordinates:
// the accepted set is the reference's, bit for bit.  Band: with u = x - centre, A_c = |u_j,c| + |u_i,c|, the fp32 value
// differs from the exact sum of squares by at most (4.02 + 3) 2^-24 sum_c A_c^2 <= 7.1 2^-24 (L1(u_j) + L1(u_i))^2
// (rounding of u to float, of the difference, of the three products / FMAs); the kernels use 16 2^-24 (R_i + R_j)^2 with
// R the largest L1 norm of the group's i-particles / of the staged batch, plus 4 2^-24 rc2 for the rounding of rc2.
struct alignas(16) PosS {
  float x, y, z;
  int32_t gid;
};
struct ScreenCtx {
  float lo, hi;               // r2f < lo: in range; r2f > hi: out of range; else exact test
  float lo_in, hi_out;        // the same two tests as sign bits: lo_in - r2f >= 0 <=> r2f < lo, r2f - hi_out >= 0 <=> r2f > hi
  const int32_t* sidx;        // [batch slots] sorted-array index of every staged particle (exact coordinates)
  const uint8_t* swrap;       // [batch slots] periodic-face code of its segment (minimum-image mode)
};
template <typename T, bool SCREEN> struct TileOf { typedef Pos<T> type; };
template <typename T> struct TileOf<T, true> { typedef PosS type; };

// Compile-time switches: timing instruments only (the kernel experiments of rounds 2 and 3 -- compare form against
// sign-bit recording, split tile parts, register staging, priorities, clamped row addresses -- are settled and gone;
// profiles/r02_count_sweep_investigation.txt has their A/B tables).
#ifndef NL_STAMP  // per-phase wave cycles of the COUNT_MASKS sweeps into dbg_buf[16..] (tools/count_phases.py, tools/rows_phases.py)
#define NL_STAMP 0
#endif
#ifndef NL_STAMP_FILL  // the same for the expansion kernels (tools/fill_phases.py); not together with NL_STAMP
#define NL_STAMP_FILL 0
#endif
#ifndef NL_DIAG  // wrong lists, timing only: 1 = no tile tests, 2 = no mask stores, 4 = no staging loads, 16 = gathered tiles
#define NL_DIAG 0
#endif
constexpr int NL_PRIO = 3;  // wave priority outside the tile loop of the COUNT_MASKS sweeps
// r2 of one staged fp32 particle against the GC i-particles: the reference's expression, every operation rounded on
// its own.  (The squares of two tests through one v_pk_mul_f32 -- each half an IEEE multiply, same bits -- measured
// 10 % slower: packed fp32 issues at half the rate of the plain forms on gfx950, profiles/r02_count_sweep_investigation.txt.)
template <int GC>
__device__ __forceinline__ void r2_group_f32(float xj, float yj, float zj, const float* xi, const float* yi, const float* zi, float* r2) {
#pragma unroll
  for (int k = 0; k < GC; k++) {
    const float dx = sub_rn(xj, xi[k]), dy = sub_rn(yj, yi[k]), dz = sub_rn(zj, zi[k]);
    r2[k] = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
  }
}

// One group of GC (compile-time, 1..5) i-particles against the nj staged j-particles.
// pi_l / base_l: lane k < GC holds i-particle k and the list offset of its row.  Returns, in lane k, the number
// of accepted partners of i-particle k.  All per-i state is wave-uniform (SGPRs): position, id, running count.
// COUNT_MASKS additionally keeps every hit: one VGPR per i-particle in which lane l sets bit t when staged
// particle t*64 + l is accepted (1 VALU per test, no scalar work); the 64 words of an i-particle go to
// a.masks[(slot0 + k) * 64 + l] at the end (store_masks: only for cells whose stencil fits one LDS batch, which
// also bounds t by CAP/64 <= 32).
// NOSELF (full list, COUNT_MASKS, the whole stream in one batch): "every j != i in range" is tested as "every j in
// range" -- no id compare and no s_and per test -- and the row's own particle (distance 0: always in range), staged at
// stream position self0 + k, is taken out of the word and of the count once at the end.
template <typename T, int MODE, int GC, bool FULL = false, bool NOSELF = false, bool SCREEN = false, bool PBC = false>
__device__ __forceinline__ int32_t search_group(const SweepArgs<T>& a, const typename TileOf<T, SCREEN>::type* tile, int32_t nj,
                                                int32_t ntiles, int lane, const Pos<T>& pi_l, int64_t base_l,
                                                int32_t slot0 = 0, bool store_masks = false, int32_t self0 = 0, int32_t batch = 0,
                                                const ScreenCtx* sc = nullptr, float uxi_l = 0.f, float uyi_l = 0.f, float uzi_l = 0.f,
                                                uint32_t* words_out = nullptr) {
  static_assert(!NOSELF || (FULL && MODE == MODE_COUNT_MASKS), "NOSELF is a form of the full-list COUNT_MASKS search");
  static_assert(SweepCfg<T>::CAP / WAVE <= 24, "one bit per j-tile in the 24-bit word a lane keeps (mask_store)");
  static_assert(!SCREEN || sizeof(T) == 8, "the screened search is the fp64 search");
  typedef typename TileOf<T, SCREEN>::type TileT;
  typedef typename std::conditional<SCREEN, float, T>::type R;  // type of the coordinates the hot loop works in
  R xi[GC], yi[GC], zi[GC];
  int32_t gi[GC];
  uint32_t cur[GC];  // hits so far
  int32_t* rowp[GC];  // FILL: where the row's next entry goes (wave-uniform 64-bit pointer: the list may exceed 2^31 entries)
  uint32_t bits[GC];  // COUNT_MASKS: bit t of lane l = staged particle t*64 + l accepted
  // VBITS (fp32 COUNT_MASKS, half list or NOSELF): the accept decision never leaves the vector unit.  Sign bit of
  // rc2 - r2 clear <=> !(r2 > rc2) (coordinates are finite: the binning rejects the rest); sign bit of
  // gid_j - (gid_i + 1) clear <=> gid_j > gid_i (ids are non-negative; the sentinels' INT32_MIN wraps to "greater" but
  // they are never in range).  OR the two words and v_alignbit the sign into the lane's word: four 2-operand VALU
  // instructions per test where two v_cmp (SGPR-pair results), an s_and, an add-with-carry, an s_bcnt1 and an s_add
  // were -- and no VALU -> SALU -> VALU dependency inside a tile.  The word collects NOT-accepted bits; inverted, and
  // the row counted from it (popcount + one DPP sum per i-particle), after the last tile.
  constexpr bool VBITS_F32 = MODE == MODE_COUNT_MASKS && !SCREEN && sizeof(T) == 4 && (!FULL || NOSELF);
  // The screened fp64 search in the same form: lo_in - r2f and r2f - hi_out carry "decided in range" / "decided out of
  // range" in their sign bits, the AND of the two over a tile's tests says whether any pair needs the exact expression
  // (ONE compare and branch per tile), and the exact answer then clears the bit the screen put into the word.
  constexpr bool VBITS_SCREEN = MODE == MODE_COUNT_MASKS && SCREEN && (!FULL || NOSELF);
  constexpr bool VBITS = VBITS_F32 || VBITS_SCREEN;
  uint32_t gi1[GC];
#pragma unroll
  for (int k = 0; k < GC; k++) bits[k] = 0;
#pragma unroll
  for (int k = 0; k < GC; k++) {
    if constexpr (SCREEN) {
      xi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, uxi_l), k));
      yi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, uyi_l), k));
      zi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, uzi_l), k));
    } else if constexpr (sizeof(T) == 4) {
      xi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.x), k));
      yi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.y), k));
      zi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.z), k));
    } else {
      xi[k] = __shfl(pi_l.x, k, WAVE);
      yi[k] = __shfl(pi_l.y, k, WAVE);
      zi[k] = __shfl(pi_l.z, k, WAVE);
    }
    gi[k] = __builtin_amdgcn_readlane(pi_l.gid, k);
    cur[k] = 0u;
    gi1[k] = (uint32_t)gi[k] + 1u;
    if (MODE == MODE_FILL) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)base_l, k);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)((uint64_t)base_l >> 32), k);
      rowp[k] = a.list + (int64_t)(((uint64_t)hi << 32) | lo);
    }
  }

  // One tile of 64 staged j-particles (lanes) against the GC i-particles (SGPRs).  All vector work of the GC
  // tests comes first (masks land in SGPR pairs), the scalar bookkeeping afterwards: a scalar instruction that
  // consumes a v_cmp result stalls the wave until the compare has left the VALU.
  auto test_tile = [&](const TileT& pj, int32_t tix) {
    if constexpr (VBITS_SCREEN) {
      constexpr int HALF = (GC + 1) / 2;
      uint32_t uacc = 0;  // sign bit: some test of this tile is inside the band for this lane
#pragma unroll
      for (int k = 0; k < GC; k++) {
        if (k == HALF) __builtin_amdgcn_sched_barrier(0);
        const float dx = pj.x - xi[k], dy = pj.y - yi[k], dz = pj.z - zi[k];
        const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const uint32_t t_in = __builtin_bit_cast(uint32_t, sc->lo_in - r2), t_out = __builtin_bit_cast(uint32_t, r2 - sc->hi_out);
        uint32_t w = t_in;  // sign clear: decided in range
        if (!NOSELF) w |= (uint32_t)pj.gid - gi1[k];
        bits[k] = __builtin_amdgcn_alignbit(bits[k], w, 31);  // an undecided pair goes in as "not accepted"
        uacc |= t_in & t_out;
      }
      if (__builtin_amdgcn_ballot_w64((int32_t)uacc < 0) != 0) {  // (uniform, rare) the reference's expression on the original coordinates
        const int32_t at = tix * WAVE + lane;
        Pos<double> pe;
        pe.x = 0, pe.y = 0, pe.z = 0;
        if ((int32_t)uacc < 0) {
          pe = a.sorted[sc->sidx[at]];
          if (PBC) {
            const int32_t wr = sc->swrap[at];
            pe.x = add_rn(pe.x, (double)((wr & 3) - 1) * a.L[0]);
            pe.y = add_rn(pe.y, (double)(((wr >> 2) & 3) - 1) * a.L[1]);
            pe.z = add_rn(pe.z, (double)(((wr >> 4) & 3) - 1) * a.L[2]);
          }
        }
#pragma unroll
        for (int k = 0; k < GC; k++) {
          const float dx = pj.x - xi[k], dy = pj.y - yi[k], dz = pj.z - zi[k];
          const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          const bool unc = (int32_t)(__builtin_bit_cast(uint32_t, sc->lo_in - r2) & __builtin_bit_cast(uint32_t, r2 - sc->hi_out)) < 0;
          if (__builtin_amdgcn_ballot_w64(unc) == 0) continue;  // uniform
          const double xe = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.x), k), __builtin_amdgcn_readlane(__double2loint(pi_l.x), k));
          const double ye = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.y), k), __builtin_amdgcn_readlane(__double2loint(pi_l.y), k));
          const double ze = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.z), k), __builtin_amdgcn_readlane(__double2loint(pi_l.z), k));
          const double ex = sub_rn(pe.x, xe), ey = sub_rn(pe.y, ye), ez = sub_rn(pe.z, ze);
          const double e2 = add_rn(add_rn(mul_rn(ex, ex), mul_rn(ey, ey)), mul_rn(ez, ez));
          const bool upper = NOSELF ? true : pj.gid > gi[k];
          if (unc && !(e2 > a.rc2) && upper) bits[k] &= ~1u;  // accepted after all
        }
      }
      return;
    }
    if constexpr (VBITS_F32) {
      // The tests go in two parts with a scheduling barrier between them: left alone the compiler hoists the
      // subtractions of all GC tests to the top of the tile, 4 live registers per test, and the kernel loses a wave per SIMD.
      constexpr int HALF = (GC + 1) / 2;
#pragma unroll
      for (int k = 0; k < GC; k++) {
        if (k == HALF) __builtin_amdgcn_sched_barrier(0);
        const float dx = sub_rn(pj.x, xi[k]), dy = sub_rn(pj.y, yi[k]), dz = sub_rn(pj.z, zi[k]);
        const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
        uint32_t w = __builtin_bit_cast(uint32_t, sub_rn((float)a.rc2, r2));
        if (!NOSELF) w |= (uint32_t)pj.gid - gi1[k];
        bits[k] = __builtin_amdgcn_alignbit(bits[k], w, 31);  // 2 bits + (w >> 31)
      }
      return;
    }
    uint64_t mask[GC];
    bool hit[GC];  // per-lane predicate: lives in an SGPR pair as a lane mask, costs no VALU
    if constexpr (SCREEN) {
      // fp32 screening of the GC tests of this tile, ONE branch for all of them (a branch per test would put a
      // VALU -> SALU -> branch dependency between the tests)
      uint64_t m_in[GC], m_unc[GC], any_unc = 0;
#pragma unroll
      for (int k = 0; k < GC; k++) {
        const float dx = pj.x - xi[k], dy = pj.y - yi[k], dz = pj.z - zi[k];
        const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        m_in[k] = __builtin_amdgcn_ballot_w64(r2 < sc->lo);
        m_unc[k] = ~(m_in[k] | __builtin_amdgcn_ballot_w64(r2 > sc->hi));  // two compares per test; a NaN is "uncertain"
        any_unc |= m_unc[k];
      }
      if (any_unc) {  // (uniform, rare) some pair is inside the band: the reference's expression on the original coordinates
        const int32_t at = tix * WAVE + lane;
        Pos<double> pe;
        pe.x = 0, pe.y = 0, pe.z = 0;
        if ((any_unc >> lane) & 1ull) {
          pe = a.sorted[sc->sidx[at]];
          if (PBC) {
            const int32_t wr = sc->swrap[at];
            pe.x = add_rn(pe.x, (double)((wr & 3) - 1) * a.L[0]);
            pe.y = add_rn(pe.y, (double)(((wr >> 2) & 3) - 1) * a.L[1]);
            pe.z = add_rn(pe.z, (double)(((wr >> 4) & 3) - 1) * a.L[2]);
          }
        }
#pragma unroll
        for (int k = 0; k < GC; k++) {
          if (m_unc[k] == 0) continue;  // uniform
          const double xe = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.x), k), __builtin_amdgcn_readlane(__double2loint(pi_l.x), k));
          const double ye = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.y), k), __builtin_amdgcn_readlane(__double2loint(pi_l.y), k));
          const double ze = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pi_l.z), k), __builtin_amdgcn_readlane(__double2loint(pi_l.z), k));
          const double ex = sub_rn(pe.x, xe), ey = sub_rn(pe.y, ye), ez = sub_rn(pe.z, ze);
          const double e2 = add_rn(add_rn(mul_rn(ex, ex), mul_rn(ey, ey)), mul_rn(ez, ez));
          m_in[k] |= __builtin_amdgcn_ballot_w64(!(e2 > a.rc2)) & m_unc[k];
        }
      }
#pragma unroll
      for (int k = 0; k < GC; k++) {
        const bool upper = NOSELF ? true : FULL ? pj.gid != gi[k] : pj.gid > gi[k];
        mask[k] = NOSELF ? m_in[k] : m_in[k] & __builtin_amdgcn_ballot_w64(upper);
        hit[k] = MODE == MODE_FILL ? ((mask[k] >> lane) & 1ull) != 0 : false;
      }
    } else {
      T r2g[GC];
      if constexpr (sizeof(T) == 4) {
        r2_group_f32<GC>(pj.x, pj.y, pj.z, xi, yi, zi, r2g);
      } else {
#pragma unroll
        for (int k = 0; k < GC; k++) {
          const T dx = sub_rn(pj.x, xi[k]), dy = sub_rn(pj.y, yi[k]), dz = sub_rn(pj.z, zi[k]);
          r2g[k] = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
        }
      }
#pragma unroll
      for (int k = 0; k < GC; k++) {
        const T r2 = r2g[k];
        // half list: j is kept by the particle with the smaller id; full list: everyone but i itself
        const bool in_range = !(r2 > a.rc2), upper = NOSELF ? true : FULL ? pj.gid != gi[k] : pj.gid > gi[k];
        hit[k] = in_range && upper;
        mask[k] = NOSELF ? __builtin_amdgcn_ballot_w64(in_range)
                         : __builtin_amdgcn_ballot_w64(in_range) & __builtin_amdgcn_ballot_w64(upper);
      }
    }
#pragma unroll
    for (int k = 0; k < GC; k++) {
      if (MODE == MODE_FILL) {
        if (hit[k]) {
          const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[k] >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mask[k], 0u));
          // uniform 64-bit row pointer (SGPR pair) + the lane's 32-bit place among this tile's hits: the
          // global_store saddr form, valid for any list size
          rowp[k][pre] = pj.gid;
        }
        rowp[k] += __popcll(mask[k]);
      }
      if (MODE == MODE_COUNT_MASKS) {
        // bits = 2 bits + hit in ONE vector instruction: add-with-carry of the word to itself, the carry-in being the
        // hit mask (an SGPR pair).  Tile t therefore ends at bit ntiles - 1 - t; reversed once before the store.
        uint64_t carry_out;
        asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits[k]), "=s"(carry_out) : "s"(mask[k]));
      }
      cur[k] += (uint32_t)__popcll(mask[k]);
    }
  };

  // The staged stream is padded to a whole tile with sentinels (gid = INT32_MIN can never satisfy
  // gid_j > gid_i, and their position 1e18 is never in range), so tiles need no per-lane tail handling.  Two register sets in ping-pong: the ds_read of the
  // next tile is in flight while the current one is tested, and no register copies are needed.
  const int32_t last = (ntiles - 1) * WAVE + lane;
#if NL_DIAG & 16
  // Timing experiment (wrong lists, right counts): the tile is GATHERED -- lane l of tile t reads slot P(64 t + l), P =
  // the stream's runs of 24 consecutive slots in reverse order (a bijection): what a search over a candidate list of
  // sub-cell runs would do to the LDS read (per-lane addresses, bank conflicts) and to the instruction count.
  const int32_t g_runs = ntiles * WAVE / 24;
  auto gat = [&](int32_t s) {
    const int32_t r = (int32_t)__umulhi((uint32_t)s, 178956971u);  // s / 24 for s < 2^16
    return r < g_runs ? (g_runs - 1 - r) * 24 + (s - r * 24) : s;
  };
  TileT pa = tile[gat(lane)], pb;
  int32_t t = 0;
  if (MODE == MODE_COUNT_MASKS) __builtin_amdgcn_s_setprio(0);
  for (; t + 1 < ntiles; t += 2) {
    pb = tile[gat((t + 1) * WAVE + lane)];
    test_tile(pa, t);
    pa = tile[gat(min((t + 2) * WAVE + lane, last))];
    test_tile(pb, t + 1);
  }
#else
  TileT pa = tile[lane], pb;
  int32_t t = (NL_DIAG & 1) ? ntiles : 0;
  if (MODE == MODE_COUNT_MASKS) __builtin_amdgcn_s_setprio(0);  // the tile loop yields to waves that are setting up, staging, storing
  for (; t + 1 < ntiles; t += 2) {  // both tests unconditional, so neither load can be sunk next to its use
    pb = tile[(t + 1) * WAVE + lane];
    test_tile(pa, t);
    pa = tile[min((t + 2) * WAVE + lane, last)];
    test_tile(pb, t + 1);
  }
#endif
  if (t < ntiles) test_tile(pa, t);
  if (MODE == MODE_COUNT_MASKS) __builtin_amdgcn_s_setprio(NL_PRIO);
  if constexpr (VBITS) {
    uint32_t w[GC], tot[GC];
    // the group's rows are mask_nb rows apart: one 64-bit row address per plane, the others at 32-bit multiples of the stride
    const size_t mrow0 = (size_t)slot0 * a.mask_nb + batch;
    char* const lo0 = reinterpret_cast<char*>(a.masks) + mrow0 * MASK_LO_BYTES;
    uint8_t* const hi0 = a.masks_hi + mrow0 * MASK_HI_BYTES;
    const bool hi = hi_plane_used(a.mask_nb, ntiles);
#pragma unroll
    for (int k = 0; k < GC; k++) {
      w[k] = __brev(~bits[k]) >> (32 - ntiles);  // tile t ended at bit ntiles - 1 - t; the bits above were never written
      if (NOSELF && lane == ((self0 + k) & (WAVE - 1))) w[k] &= ~(1u << ((self0 + k) >> 6));
      if (words_out) words_out[k] = w[k];  // (k_sweep_lean_f32: the caller stores the words after the search)
      else if (store_masks && !(NL_DIAG & 2)) {
        reinterpret_cast<uint16_t*>(lo0 + (uint32_t)(k * a.mask_nb) * MASK_LO_BYTES)[lane] = (uint16_t)w[k];
        if (hi) (hi0 + (uint32_t)(k * a.mask_nb) * MASK_HI_BYTES)[lane] = (uint8_t)(w[k] >> 16);
      }
    }
#pragma unroll
    for (int k = 0; k < GC; k += 2) {  // two rows per DPP sum: a row has at most CAP < 2^16 accepted partners per batch
      const uint32_t two = wave_sum_dpp((uint32_t)__popc(w[k]) | (k + 1 < GC ? (uint32_t)__popc(w[k + 1]) << 16 : 0u));
      tot[k] = two & 0xffffu;
      if (k + 1 < GC) tot[k + 1] = two >> 16;
    }
    uint32_t mine_v = 0;
#pragma unroll
    for (int k = 0; k < GC; k++) mine_v = lane == k ? tot[k] : mine_v;
    return (int32_t)mine_v;
  }
  if (MODE == MODE_COUNT_MASKS) {
    if (store_masks && !(NL_DIAG & 2)) {
#pragma unroll
      for (int k = 0; k < GC; k++) {
        uint32_t w = __brev(bits[k]) >> (32 - ntiles);
        if (NOSELF && lane == ((self0 + k) & (WAVE - 1))) w &= ~(1u << ((self0 + k) >> 6));
        mask_store(a.masks, a.masks_hi, (size_t)(slot0 + k) * a.mask_nb + batch, lane, w, hi_plane_used(a.mask_nb, ntiles));  // mask row of (slot, LDS batch)
      }
    }
    if (NOSELF) {
#pragma unroll
      for (int k = 0; k < GC; k++) cur[k] -= 1u;
    }
  }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < GC; k++) mine = lane == k ? cur[k] : mine;
  return (int32_t)mine;
}

// Everything a workgroup knows about its i-cell: the cell's own particles [ibeg, ibeg + ni) and, per lane
// s < NSEG, segment s of the stencil stream (start in the sorted array, length, offset in the stream).
struct CellCtx {
  int32_t ibeg, ni, seg_src, seg_len, seg_off, total_j;
  int32_t cx, cy, cz;  // the i-cell (cz: local layer)
  int32_t wrap;        // per segment: (wx + 1) | (wy + 1) << 2 | (wz + 1) << 4, w = -1 / 0 / +1: the segment's cells are
                       // reached through the low / no / high periodic face of that axis
};

// XCD-aware block -> cell order: blocks b, b+8, b+16.. share an XCD (and its L2), so give each XCD a contiguous range
// of cells (a z-slab of the box) instead of every 8th cell.  Returns the linear index w of this block's cell among
// the gridDim.x cells of the launch (x fastest, then y, then z).
__device__ __forceinline__ int32_t xcd_cell_index() {
  const int32_t nb = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <typename T> __device__ __forceinline__ bool cell_setup_at(const SweepArgs<T>& a, int lane, int32_t cx, int32_t cy, int32_t cz, CellCtx& c);

// Maps the workgroup to its i-cell (XCD-aware) and loads the segment table.  Returns false for an empty cell.
template <typename T> __device__ __forceinline__ bool cell_setup(const SweepArgs<T>& a, int lane, CellCtx& c) {
  const int32_t w = xcd_cell_index();
  const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), cx = w - wy * a.mx;
  const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
  return cell_setup_at(a, lane, cx, cy, cz, c);
}

// Segment `lane` (< NSEG) of the stencil stream of cell (cx, cy, cz; cz: local layer): the cells [i0, i1) of the
// cell-sorted array, as indices into cell_start, and through which periodic faces they are reached (CellCtx::wrap).
// One segment per lane: 9 (dz,dy) rows x 2 x-parts; slots 0..8 = first x-part of the nine rows (never empty in the
// interior), 9..17 = the part behind the periodic wrap in x (empty unless cx is 0 or mx - 1).
template <typename T>
__device__ __forceinline__ void segment_cells(const SweepArgs<T>& a, int lane, int32_t cx, int32_t cy, int32_t cz,
                                              int32_t& i0, int32_t& i1, int32_t& wrap) {
  const int32_t s = lane % 9, part = lane / 9, dz = s / 3 - 1, dy = s % 3 - 1;
  int32_t y = cy + dy, z = cz + dz;
  if (y < 0) y += a.my;
  if (y >= a.my) y -= a.my;
  if (!a.slab) {
    if (z < 0) z += a.mzl;
    if (z >= a.mzl) z -= a.mzl;
  }
  int32_t x0, x1;  // cells [x0, x1) of that row
  if (cx == 0) {
    x0 = part ? 0 : a.mx - 1, x1 = part ? 2 : a.mx;
  } else if (cx == a.mx - 1) {
    x0 = part ? 0 : a.mx - 2, x1 = part ? 1 : a.mx;
  } else {
    x0 = cx - 1, x1 = part ? cx - 1 : cx + 2;
  }
  const int32_t rowbase = (y + z * a.my) * a.mx;
  i0 = rowbase + x0, i1 = rowbase + x1;
  // through which periodic faces this segment is reached (used in minimum-image mode only)
  const int32_t wx = (cx == 0 && part == 0) ? -1 : (cx == a.mx - 1 && part == 1) ? 1 : 0;
  const int32_t wy = cy + dy < 0 ? -1 : cy + dy >= a.my ? 1 : 0;
  const int32_t wz = a.slab ? 0 : cz + dz < 0 ? -1 : cz + dz >= a.mzl ? 1 : 0;
  wrap = (wx + 1) | (wy + 1) << 2 | (wz + 1) << 4;
}

// The segment table of the i-cell (cx, cy, cz): cz is the local layer.
template <typename T> __device__ __forceinline__ bool cell_setup_at(const SweepArgs<T>& a, int lane, int32_t cx, int32_t cy, int32_t cz, CellCtx& c) {
  const int32_t cell = cx + (cy + cz * a.my) * a.mx;
  // A build whose binning has flagged the particles as inconsistent with the slab description (ST_DOMAIN: a particle
  // in the wrong layer, or -- split slab builds -- a ghost count that does not match the data) may have overlapping
  // regions in the sorted array and a cell table that is not monotonic: nobody walks it.  (Read with the cell table.)
  const uint32_t st_word = *a.status;
  c.ibeg = a.cell_start[cell];
  c.ni = a.cell_start[cell + 1] - c.ibeg;
  c.cx = cx, c.cy = cy, c.cz = cz;

  // (the empty-cell exit comes after the segment-table loads so that both round trips are in flight together)
  c.seg_src = 0, c.seg_len = 0, c.wrap = 0x15;
  if (lane < NSEG) {
    int32_t i0, i1;
    segment_cells(a, lane, cx, cy, cz, i0, i1, c.wrap);
    c.seg_src = a.cell_start[i0];
    c.seg_len = a.cell_start[i1] - c.seg_src;
  }
  if (c.ni <= 0 || (st_word & ST_DOMAIN)) return false;
  c.seg_off = scan32_dpp(c.seg_len) - c.seg_len;  // exclusive offsets in the staged stream
  c.total_j = __builtin_amdgcn_readlane(c.seg_off + c.seg_len, NSEG - 1);
  return true;
}

// The pair search of one cell: stage the stencil stream into `tile` (in batches of CAP), search it group by group.
// SCREEN (fp64 only): `tile_raw` is screen_lds_bytes(CAP) bytes laid out as PosS[CAP] | sidx[CAP] | swrap[CAP] | float[NW].
constexpr int screen_lds_bytes(int cap) { return cap * 21 + 32; }
template <typename T, int MODE, int CAP = SweepCfg<T>::CAP, int NW = SWEEP_WAVES, bool FULL = false, bool PBC = false, bool SCREEN = false>
__device__ __forceinline__ void cell_search(const SweepArgs<T>& a, const CellCtx& c, Pos<T>* tile_raw, int tid, int lane,
                                            int wave) {
  // (Groups of up to 7 in the fp32 COUNT_MASKS sweeps -- a cell of 41..56 particles in 8 passes over the stream instead
  // of 12 -- measured level with 5: 0.2779 against 0.2768 ms at cfg 2.)
  constexpr int G = SWEEP_G;
  static_assert(G == 5, "search_group dispatch covers group sizes 1..5");
  typedef typename TileOf<T, SCREEN>::type TileT;
  TileT* const tile = reinterpret_cast<TileT*>(tile_raw);
  int32_t* const sidx = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(tile_raw) + (size_t)CAP * 16);
  uint8_t* const swrap = reinterpret_cast<uint8_t*>(sidx + CAP);
  float* const rmax_w = reinterpret_cast<float*>(reinterpret_cast<char*>(tile_raw) + ((size_t)CAP * 21 + 15) / 16 * 16);
  // centre of the i-cell: origin of the relative coordinates of the screened search
  T ox = 0, oy = 0, oz = 0;
  if constexpr (SCREEN) {
    ox = ((T)c.cx + (T)0.5) * a.ms[0], oy = ((T)c.cy + (T)0.5) * a.ms[1], oz = ((T)(c.cz + a.z_origin) + (T)0.5) * a.ms[2];
  }
  const int32_t ibeg = c.ibeg, ni = c.ni, total_j = c.total_j;
  const int32_t nbatch = (total_j + CAP - 1) / CAP;

  // i-groups: `rounds` groups per wave, sized so that all waves get the same number of groups.
  const int32_t rounds = (ni + NW * G - 1) / (NW * G);
  const int32_t ngroups = rounds * NW;
  const int32_t gbase = ni / ngroups, grem = ni - gbase * ngroups;  // group g: gbase (+ 1 if g < grem) particles
  auto group_begin = [&](int32_t g) { return g * gbase + min(g, grem); };

  // where the cell's own particles sit in the stream (NOSELF): in the (dz,dy) = (0,0) row, first or wrapped x-part
  int32_t own = 0;
  constexpr bool CAN_NOSELF = FULL && MODE == MODE_COUNT_MASKS;
  if (CAN_NOSELF) {
    const int32_t s4 = __builtin_amdgcn_readlane(c.seg_src, 4), l4 = __builtin_amdgcn_readlane(c.seg_len, 4);
    const bool in4 = ibeg >= s4 && ibeg < s4 + l4;
    own = in4 ? __builtin_amdgcn_readlane(c.seg_off, 4) + ibeg - s4
              : __builtin_amdgcn_readlane(c.seg_off, 13) + ibeg - __builtin_amdgcn_readlane(c.seg_src, 13);
  }

#if NL_STAMP
  uint64_t t_prev = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0}, n_tests = 0;  // wave-uniform: SGPRs
  auto stamp = [&](int phase) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    t_acc[phase] += now - t_prev;
    t_prev = now;
  };
#endif
  for (int32_t batch = 0; batch < nbatch; batch++) {
    const int32_t win0 = batch * CAP;
    const int32_t nj = min(total_j - win0, CAP);
    if (batch) __syncthreads();  // everyone is done reading the previous batch
    // ---- stage: copy the stream window [win0, win0 + nj) into LDS.  The waves share the segments (slot s goes
    // to wave s mod 4: the nine never-empty slots spread 3/2/2/2); a wave copies a segment 128 particles at a
    // time with both 16-byte loads in flight before the LDS writes.
    // A stream of one batch (fp32, open box, COUNT_MASKS) goes by LDS-DMA instead, 16 bytes per lane (below).  History:
    // dword LDS-DMA in round 1 and "all of a wave's 16-byte loads in flight before the first LDS write" early in round 2
    // both measured SLOWER than this loop, in which the compiler waits for every load before it issues the next
    // (profiles/r02_count_staging_ab.txt) -- the staging phase was starved of issue slots, not waiting for memory
    // (DESIGN.md section 5); with s_setprio around the tile loop and 16-byte DMA it is 2 300 cycles of a wave's 36 000.
    float rmax = 0.f;  // SCREEN: largest L1 norm of the relative coordinates this thread stages
    constexpr bool DMA = sizeof(T) == 4 && !PBC && !SCREEN && MODE == MODE_COUNT_MASKS;
    for (int32_t sg = wave; sg < NSEG; sg += NW) {
      const int32_t len = __builtin_amdgcn_readlane(c.seg_len, sg);
      if (len == 0) continue;
      if constexpr (DMA) {
        if (nbatch == 1) {  // (uniform) the whole stream in one batch: LDS-DMA, 16 bytes per lane, no registers, every
          // piece in flight until the barrier below
          const int32_t src1 = __builtin_amdgcn_readlane(c.seg_src, sg), off1 = __builtin_amdgcn_readlane(c.seg_off, sg);
#pragma unroll 1
          for (int32_t kb = 0; kb < len; kb += WAVE) {
            if (kb + lane < len)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.sorted + src1 + kb + lane),
                                               (__attribute__((address_space(3))) void*)(tile + off1 + kb), 16, 0, 0);
          }
          continue;
        }
      }
      const int32_t src = __builtin_amdgcn_readlane(c.seg_src, sg);
      const int32_t off = __builtin_amdgcn_readlane(c.seg_off, sg) - win0;
      // minimum-image kernels (PBC): a segment reached through a periodic face is staged at its image.  Uniform per
      // segment; a template parameter because even a never-taken run-time branch here cost the open-box path 3-6 %.
      const int32_t wr = PBC ? __builtin_amdgcn_readlane(c.wrap, sg) : 0x15;  // (compile-time: the open-box kernels
                                                                              // carry none of this)
      if constexpr (SCREEN) {
        const T sx = (T)((wr & 3) - 1) * a.L[0], sy = (T)(((wr >> 2) & 3) - 1) * a.L[1], sz = (T)(((wr >> 4) & 3) - 1) * a.L[2];
        for (int32_t k = lane; k < len; k += WAVE) {
          Pos<T> v0 = a.sorted[src + k];
          if (PBC && wr != 0x15) v0.x = add_rn(v0.x, sx), v0.y = add_rn(v0.y, sy), v0.z = add_rn(v0.z, sz);
          if (nbatch == 1 || (uint32_t)(off + k) < (uint32_t)CAP) {
            PosS u;
            u.x = (float)(v0.x - ox), u.y = (float)(v0.y - oy), u.z = (float)(v0.z - oz), u.gid = v0.gid;
            tile[off + k] = u;
            sidx[off + k] = src + k;
            if (PBC) swrap[off + k] = (uint8_t)wr;
            rmax = fmaxf(rmax, fabsf(u.x) + fabsf(u.y) + fabsf(u.z));
          }
        }
      } else {
        if (PBC && wr != 0x15) {
          const T sx = (T)((wr & 3) - 1) * a.L[0], sy = (T)(((wr >> 2) & 3) - 1) * a.L[1], sz = (T)(((wr >> 4) & 3) - 1) * a.L[2];
          for (int32_t k = lane; k < len; k += WAVE) {
            Pos<T> v0 = a.sorted[src + k];
            v0.x = add_rn(v0.x, sx), v0.y = add_rn(v0.y, sy), v0.z = add_rn(v0.z, sz);
            if (nbatch == 1 || (uint32_t)(off + k) < (uint32_t)CAP) tile[off + k] = v0;
          }
          continue;
        }
        for (int32_t k = lane; k < len; k += 2 * WAVE) {
          const int32_t k1 = k + WAVE;
          const bool p1 = k1 < len;
#if NL_DIAG & 4
          Pos<T> v0, v1;
          v0.x = (T)k, v0.y = (T)src, v0.z = (T)off, v0.gid = k, v1 = v0;
#else
          const Pos<T> v0 = a.sorted[src + k];
          const Pos<T> v1 = a.sorted[src + (p1 ? k1 : k)];
#endif
          if (nbatch == 1 || (uint32_t)(off + k) < (uint32_t)CAP) tile[off + k] = v0;
          if (p1 && (nbatch == 1 || (uint32_t)(off + k1) < (uint32_t)CAP)) tile[off + k1] = v1;
        }
      }
    }
    {  // sentinel padding up to the next tile boundary
      const int32_t pad = nj + tid;
      if (pad < ((nj + WAVE - 1) & ~(WAVE - 1))) {
        // far outside any box (r2 ~ 1e36 / 1e300, finite: never in range) AND an id that is never the upper one
        if constexpr (SCREEN) {
          PosS sentinel;
          sentinel.x = 1.0e18f, sentinel.y = 0.f, sentinel.z = 0.f, sentinel.gid = INT32_MIN;
          tile[pad] = sentinel;
          sidx[pad] = 0;
          if (PBC) swrap[pad] = 0x15;
        } else {
          Pos<T> sentinel;
          sentinel.x = sizeof(T) == 4 ? (T)1.0e18f : (T)1.0e150, sentinel.y = 0, sentinel.z = 0, sentinel.gid = INT32_MIN;
          if constexpr (sizeof(T) == 8) sentinel.row = 0;
          tile[pad] = sentinel;
        }
      }
    }
    if constexpr (SCREEN) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, d, WAVE));
      if (lane == 0) rmax_w[wave] = rmax;
    }
#if NL_STAMP
    stamp(1);  // staging: loads issued, LDS written
#endif
    __syncthreads();
#if NL_STAMP
    stamp(2);  // barrier
#endif
    float rj = 0.f;  // SCREEN: largest L1 norm of a staged particle's relative coordinates
    if constexpr (SCREEN) {
#pragma unroll
      for (int w = 0; w < NW; w++) rj = fmaxf(rj, rmax_w[w]);
    }

    // ---- search: this wave's groups against every tile of the batch
    const int32_t ntiles = (nj + WAVE - 1) / WAVE;
    // A group's i-particles, rows and running counts are loaded one group ahead (dense cells: a wave has a dozen groups
    // per batch, and each group's own loads -- position -> row -> progress, dependent -- cost as much as its search).
    // Unconditional loads of valid slots (a load under a branch is waited for at the join).
    Pos<T> pre_p;
    int32_t pre_row = 0, pre_prog = 0;
    int64_t pre_base = 0;
    auto fetch_group = [&](int32_t g) {
      const int32_t idx = ibeg + min(group_begin(g) + lane, ni - 1);
      pre_p = a.sorted[idx];
      pre_row = a.sorted_row[idx];
      if (MODE == MODE_FILL)
        pre_base = a.wide ? static_cast<const int64_t*>(a.key_pointer)[pre_row] : (int64_t)static_cast<const int32_t*>(a.key_pointer)[pre_row];
      pre_prog = batch ? a.progress[pre_row] : 0;  // entries already produced by earlier batches
    };
    const bool ahead = !(a.dbg & 1024);  // diagnostics: 1024 = every group fetches its own data when it starts
    if (wave < ngroups && ahead) fetch_group(wave);
    for (int32_t g = wave; g < ngroups; g += NW) {
      const int32_t i0 = group_begin(g);
      const int32_t gcount = gbase + (g < grem ? 1 : 0);  // wave-uniform; 0 for the last groups of a small cell
      if (gcount <= 0) break;
      // lane k < gcount holds i-particle k of the group
      if (!ahead) fetch_group(g);
      Pos<T> pi_l = pre_p;
      const int32_t row_l = pre_row, before = pre_prog;
      int64_t base_l = pre_base + before;
      if (lane >= gcount) pi_l.x = 0, pi_l.y = 0, pi_l.z = 0, pi_l.gid = 0;
      if (ahead && g + NW < ngroups && group_begin(g + NW) < ni) fetch_group(g + NW);
      const int32_t slot0 = ibeg + i0;
      // SCREEN: the i-particles relative to the cell centre, as floats, and the band of this group against this batch
      float uxi = 0.f, uyi = 0.f, uzi = 0.f;
      ScreenCtx sc;
      sc.lo = sc.hi = 0.f, sc.sidx = sidx, sc.swrap = swrap;
      if constexpr (SCREEN) {
        float ri = 0.f;
        if (lane < gcount) {
          uxi = (float)(pi_l.x - ox), uyi = (float)(pi_l.y - oy), uzi = (float)(pi_l.z - oz);
          ri = fabsf(uxi) + fabsf(uyi) + fabsf(uzi);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ri = fmaxf(ri, __shfl_xor(ri, d, WAVE));
        const float rc2f = (float)a.rc2;
        const float band = 9.5367432e-7f * (ri + rj) * (ri + rj) + 2.3841858e-7f * rc2f;  // 16 2^-24 S + 4 2^-24 rc2
        sc.lo = rc2f - band, sc.hi = rc2f + band;
        sc.lo_in = nextafterf(sc.lo, -1.0f), sc.hi_out = nextafterf(sc.hi, 3.0e38f);  // (lo > 0: rc2 dwarfs the band)
      }
      // hit masks are kept for cells whose stencil fits the mask rows the build provides per slot: one LDS batch in the
      // usual regime, up to FD_NB in a dense build (k_fill_dense); the expansion kernels re-search the rest
      const bool keep = nbatch <= a.mask_nb && CAP == SweepCfg<T>::CAP;
      int32_t mine;  // lane k < gcount: hits of i-particle k in this batch
      if constexpr (CAN_NOSELF) {
        if (nbatch == 1) {  // (uniform) the row's own particle is in this, the only, batch
          switch (gcount) {
            case 1: mine = search_group<T, MODE, 1, FULL, true, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, own + i0, 0, &sc, uxi, uyi, uzi); break;
            case 2: mine = search_group<T, MODE, 2, FULL, true, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, own + i0, 0, &sc, uxi, uyi, uzi); break;
            case 3: mine = search_group<T, MODE, 3, FULL, true, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, own + i0, 0, &sc, uxi, uyi, uzi); break;
            case 4: mine = search_group<T, MODE, 4, FULL, true, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, own + i0, 0, &sc, uxi, uyi, uzi); break;
            default: mine = search_group<T, MODE, 5, FULL, true, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, own + i0, 0, &sc, uxi, uyi, uzi); break;
          }
          if (lane < gcount) a.count[row_l] = mine;
          continue;
        }
      }
#if NL_STAMP
      stamp(3);  // group prologue (look-ahead loads consumed, next issued)
#endif
      switch (gcount) {
        case 1: mine = search_group<T, MODE, 1, FULL, false, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, 0, batch, &sc, uxi, uyi, uzi); break;
        case 2: mine = search_group<T, MODE, 2, FULL, false, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, 0, batch, &sc, uxi, uyi, uzi); break;
        case 3: mine = search_group<T, MODE, 3, FULL, false, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, 0, batch, &sc, uxi, uyi, uzi); break;
        case 4: mine = search_group<T, MODE, 4, FULL, false, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, 0, batch, &sc, uxi, uyi, uzi); break;
        default: mine = search_group<T, MODE, 5, FULL, false, SCREEN, PBC>(a, tile, nj, ntiles, lane, pi_l, base_l, slot0, keep, 0, batch, &sc, uxi, uyi, uzi); break;
      }
#if NL_STAMP
      stamp(4);  // search_group: readlanes, tile loop, mask stores
      n_tests += (uint64_t)(gcount * ntiles);
#endif
      if (lane < gcount) {
        if (nbatch > 1) {
          mine += before;
          a.progress[row_l] = mine;
        }
        if (MODE != MODE_FILL && batch == nbatch - 1) a.count[row_l] = mine;
      }
#if NL_STAMP
      stamp(5);  // count store
#endif
    }
  }
#if NL_STAMP
  if (MODE == MODE_COUNT_MASKS && lane == 0) {  // one set of atomics per wave, spread over 1024 slots (the host sums them)
    unsigned long long* const slot = a.dbg_buf + 64 + (blockIdx.x & 1023) * 16;
    for (int ph = 1; ph < 6; ph++) atomicAdd(slot + ph, (unsigned long long)t_acc[ph]);
    atomicAdd(slot + 8, (unsigned long long)n_tests);
    atomicAdd(slot + 9, 1ull);
  }
#endif
}

// LDS batch of the sweeps.  (Measured at BASELINE config 5, fp64, 311 particles per cell: half the batch -- 20 KB of
// LDS per workgroup instead of 40, 5-6 resident waves per SIMD instead of 4 -- is 10 % SLOWER, 8.54 against 7.77 ms:
// the dense sweeps are bound by the fp64 vector rate, and every batch costs a staging step and two barriers.)
template <typename T, int MODE> constexpr int sweep_cap() { return SweepCfg<T>::CAP; }

template <typename T, int MODE, bool FULL = false, bool PBC = false>
__device__ __forceinline__ void sweep_cell(const SweepArgs<T>& a) {
  constexpr int CAP = sweep_cap<T, MODE>();
  constexpr bool SCREEN = sizeof(T) == 8;  // fp64 sweeps: fp32 screening + exact test inside the band (search_group)
  __shared__ __attribute__((aligned(32))) char tile_bytes[SCREEN ? screen_lds_bytes(CAP) : CAP * (int)sizeof(Pos<T>)];
  Pos<T>* const tile = reinterpret_cast<Pos<T>*>(tile_bytes);
  if (MODE == MODE_FILL) {
    if (a.total[0] > a.capacity) {  // uniform: every workgroup leaves, nothing is written out of bounds
      if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  CellCtx c;
  if (MODE == MODE_COUNT_MASKS) __builtin_amdgcn_s_setprio(NL_PRIO);
#if NL_STAMP
  const uint64_t t_entry = __builtin_amdgcn_s_memtime();
#endif
  if (MODE == MODE_COUNT_MASKS || a.isplit <= 1) {
    if (!cell_setup(a, lane, c)) return;
  } else {
    // NL_ISPLIT = S > 1 (diagnostics): a cell is shared by S workgroups, each taking a contiguous part of the cell's
    // i-particles against the whole stream.  Meant for boxes of a few thousand dense cells (BASELINE config 5: 3375
    // workgroups of ~2 ms on 1024 slots); measured: S = 3 -3 %, S = 5 +3 %, S = 8 +11 % -- every part stages the whole
    // stream again -- so the default stays one workgroup per cell.
    const int32_t wp = xcd_cell_index(), w = wp / a.isplit, part = wp - w * a.isplit;
    const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), cx = w - wy * a.mx;
    const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
    if (!cell_setup_at(a, lane, cx, cy, cz, c)) return;
    const int32_t lo = (int32_t)((int64_t)c.ni * part / a.isplit), hi = (int32_t)((int64_t)c.ni * (part + 1) / a.isplit);
    c.ibeg += lo, c.ni = hi - lo;
    if (c.ni <= 0) return;
  }
#if NL_STAMP
  if (MODE == MODE_COUNT_MASKS && lane == 0) atomicAdd(a.dbg_buf + 64 + (blockIdx.x & 1023) * 16, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_entry));
#endif
  cell_search<T, MODE, CAP, SWEEP_WAVES, FULL, PBC, SCREEN>(a, c, tile, tid, lane, wave);
}

template <typename T, int MODE, bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) k_sweep(SweepArgs<T> a) {
  sweep_cell<T, MODE, FULL, PBC>(a);
}
// The fp32 COUNT passes are held to 80 SGPRs: the SGPR file admits 8 waves per SIMD only up to 80 per wave (6 at
// the 102 the compiler takes by itself).  FILL needs the extra SGPRs (cursors + masks): capped, it spills into its
// inner loop and loses more than the occupancy gains.
template <bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) __attribute__((amdgpu_num_sgpr(80)))
k_sweep_count_f32(SweepArgs<float> a) {
  sweep_cell<float, MODE_COUNT, FULL, PBC>(a);
}
template <bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) __attribute__((amdgpu_num_sgpr(80)))
k_sweep_count_masks_f32(SweepArgs<float> a) {
  sweep_cell<float, MODE_COUNT_MASKS, FULL, PBC>(a);
}

// ------------------------------------------------------------------------------------------ list from masks
// a8 + a9 without a second distance sweep: COUNT_MASKS left, for every i-particle (sorted slot), 64 words whose
// bits say which particles of its cell's staged stencil stream were accepted.  This kernel re-stages only the ids
// of the stream (4 B per particle, LDS-DMA from the compact sorted_gid) and expands the bits into the final CSR rows,
// four rows per wave at a time: lane l owns word l, a DPP prefix sum of the popcounts gives its offset inside the
// row, and it takes its set bits one by one (bit t -> staged particle t*64 + l: LDS reads of one instruction always
// hit 64 different banks) into a per-wave LDS copy of the row, which then leaves as runs of 64 consecutive entries.
// Cells whose stencil needed several LDS batches have no masks: they are searched again here, exactly as
// k_sweep<FILL> does.
__device__ __forceinline__ int32_t scan64_dpp(int32_t v) {  // inclusive scan over all 64 lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// longest row k_fill_masks assembles in LDS (longer ones are written entry by entry); half list: 5 + 5 KiB of LDS =
// 16 workgroups per CU
template <bool FULL> constexpr int EXPAND_RMAX_OF = FULL ? 192 : 160;
constexpr int EXPAND_WAVES = 2;  // waves per workgroup of k_fill_masks: 16 workgroups (cells) in flight per CU

// EW, CAP: 2 waves and room for the ids of a full stream, or (sparse boxes, as k_sweep_lean_f32's small instance) one
// wave per cell and half of it: half as many cell tables and id stagings, no barrier between waves (cfg 3: build -2.5 %).
// (One wave per cell with 40 rows up front at cfg 2: 96 VGPRs, 5 waves per SIMD, expansion 0.207 against 0.175 ms.)
template <typename T, bool FULL = false, bool PBC = false, typename OFF = int32_t, int RB = 24, int EW = EXPAND_WAVES, int CAP = SweepCfg<T>::CAP>
__global__ void __launch_bounds__(EW* WAVE, (sizeof(OFF) == 8 ? 4 : sizeof(T) == 4 ? (FULL || RB > 12 ? 7 : 8) : 4)) __attribute__((amdgpu_num_sgpr(80)))
k_fill_masks(SweepArgs<T> a) {
  constexpr int EXPAND_RMAX = EXPAND_RMAX_OF<FULL>;
  // One LDS array (a second __shared__ object next to an LDS-DMA target makes hipcc drain the DMA before every
  // ds_read): the ids of the stencil stream (4.5 KiB) + per wave four rows being put together.
  __shared__ __attribute__((aligned(32))) int32_t lds[CAP + EW * 4 * EXPAND_RMAX];
  int32_t* const gids = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if NL_STAMP_FILL
  uint64_t ft_prev = __builtin_amdgcn_s_memtime(), ft_acc[6] = {0, 0, 0, 0, 0, 0};
  auto fstamp = [&](int phase) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    ft_acc[phase] += now - ft_prev;
    ft_prev = now;
  };
#endif
  const int64_t total = a.total[0];  // read together with the cell table: one round trip, not two
  CellCtx c;
  const bool ok = cell_setup(a, lane, c);
#if NL_STAMP_FILL
  fstamp(0);
#endif
  if (total > a.capacity) {  // the list is too small: the host grows it and runs this kernel again
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
    return;
  }
  if (!ok) return;
  if (c.total_j > CAP) {
    // No masks for this cell (its stencil needed several LDS batches in COUNT_MASKS): it goes on the list of
    // k_fill_list, which searches it again as k_sweep<FILL> does.  Rare (very dense cells only).
    // (k_fill_list, a kernel of its own: inlined here the re-search cost this kernel 74 spilled SGPRs and 6 % of its time)
    if (tid == 0) a.full27_list[atomicAdd(a.fill_list_count, 1)] = c.cx + (c.cy + c.cz * a.my) * a.mx;  // local cell index
    return;
  }

  // Rows of the cell are dealt to the waves in contiguous chunks.  A wave loads the masks and row offsets of up to
  // RB rows first and only then starts storing: vmcnt retires in order, so a load issued behind stores would wait
  // for the whole write latency of every store before it (that cost 1.7 us per row in the first version).
  // RB (template parameter) is 24, or 12 in a build whose cells hold ~20 particles or fewer (the host picks the kernel by
  // the mean): a wave with 10 rows -- BASELINE config 3 -- otherwise issues 14 x 3 loads of rows it does not have
  // (expansion 0.201 -> 0.181 ms there; with 12 config 2 needs a second pass per wave: +13 %; both in one kernel, chosen
  // per cell: spills under the 64-register cap, +17 %).
  const int32_t per_wave = (c.ni + EW - 1) / EW;
  const int32_t r_beg = min(wave * per_wave, c.ni), r_end = min(r_beg + per_wave, c.ni);
  uint32_t w[RB];
  OFF base_l;  // lane u: list offset of row r0 + u
  const bool hi_plane = hi_plane_used(1, (c.total_j + WAVE - 1) / WAVE);
  auto load_rows = [&](int32_t r0) {
    {
      // key_pointer[sorted_row[slot]] of the batch's rows, a lane per row: two dependent loads per batch in the shadow
      // of the mask loads (round 2 had a gather kernel in front, k_row_base: a launch and 8 MB of traffic per build).
      // Slots past the cell's last row belong to other cells, ghosts or the padding: any row id is good enough there.
      const int32_t srow = a.sorted_row[c.ibeg + r0 + min(lane, RB - 1)];
      base_l = static_cast<const OFF*>(a.key_pointer)[min((uint32_t)srow, (uint32_t)a.n_rows)];
    }
    // Consecutive slots are consecutive rows of each plane: ONE 64-bit base address per plane and batch, the rows at
    // immediate offsets from it.  Rows past the cell's last one are read as well (the next cells' rows, or the 64 rows of padding behind
    // the last slot) and ignored.  (With `slot = ibeg + min(r0 + u, ni - 1)` per row the compiler spent 17 scalar
    // instructions per row on 64-bit address arithmetic: 400 per wave.)
    const size_t slot0 = (size_t)(c.ibeg + r0);
    const char* const lo0 = reinterpret_cast<const char*>(a.masks) + slot0 * MASK_LO_BYTES;  // (one mask row per slot here)
    if (hi_plane) {  // (uniform over the workgroup: a property of the cell)
      const uint8_t* const hi0 = a.masks_hi + slot0 * MASK_HI_BYTES;
#pragma unroll
      for (int u = 0; u < RB; u++)
        w[u] = (uint32_t)mask_ld<EW == 2>(reinterpret_cast<const uint16_t*>(lo0 + u * MASK_LO_BYTES) + lane) | (uint32_t)mask_ld<EW == 2>(hi0 + u * MASK_HI_BYTES + lane) << 16;
    } else {
#pragma unroll
      for (int u = 0; u < RB; u++) w[u] = (uint32_t)mask_ld<EW == 2>(reinterpret_cast<const uint16_t*>(lo0 + u * MASK_LO_BYTES) + lane);
    }
  };
  auto base = [&](int u) -> OFF {  // u: compile-time constant after unrolling
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)base_l, u);
    if constexpr (sizeof(OFF) == 4) return (OFF)lo;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)((uint64_t)base_l >> 32), u);
    return (OFF)((uint64_t)hi << 32 | lo);
  };
  load_rows(r_beg);  // issued before the id staging: independent of the segment table
  if (a.dbg & 32) return;  // diagnostics: setup + loads only

  // stage the ids of the stencil stream, from the compact copy of the id field
  // LDS-DMA (global_load_lds_dword: lane l's dword lands at the uniform LDS address + 4 l): no VGPRs and no LDS
  // write instructions, and every segment's loads are in flight together.  (Loading and writing segment by segment
  // through registers cost one memory round trip per segment, 9 per wave.)
#pragma unroll
  for (int s = 0; s < (NSEG + EW - 1) / EW; s++) {
    const int sg = min(wave + s * EW, NSEG - 1);  // wave-uniform
    const int32_t len = wave + s * EW < NSEG ? __builtin_amdgcn_readlane(c.seg_len, sg) : 0;
    const int32_t src = __builtin_amdgcn_readlane(c.seg_src, sg);
    const int32_t off = __builtin_amdgcn_readlane(c.seg_off, sg);
    for (int32_t kb = 0; kb < len; kb += WAVE) {
      if (kb + lane < len)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.sorted_gid + src + kb + lane),
                                         (__attribute__((address_space(3))) void*)(gids + off + kb), 4, 0, 0);
    }
  }
#if NL_STAMP_FILL
  fstamp(1);  // row loads and id DMA issued
#endif
  __syncthreads();  // ids staged
#if NL_STAMP_FILL
  fstamp(2);  // barrier (DMA landed)
#endif

  const int32_t* const g = gids + lane;
  for (int32_t r0 = r_beg; r0 < r_end; r0 += RB) {
    if (r0 != r_beg) load_rows(r0);  // dense cells only
    // four rows are expanded together: their LDS reads and stores are independent, so one trip through the
    // bit loop pays the LDS latency once for four rows
#pragma unroll
    for (int u0 = 0; u0 < RB; u0 += 4) {
      if (r0 + u0 >= r_end) continue;  // wave-uniform
      uint32_t word[4], ptr[4];
      int32_t nrow[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        word[q] = r0 + u0 + q < r_end ? (w[u0 + q] & 0xFFFFFFu) : 0u;
        if (a.dbg & 16) word[q] = 0;  // diagnostics: no expansion
        const int32_t cnt = __popc(word[q]);
        const int32_t incl = scan64_dpp(cnt);
        nrow[q] = __builtin_amdgcn_readlane(incl, 63);
        ptr[q] = (uint32_t)(incl - cnt);  // place inside the row
      }
      const int32_t nmax = max(max(nrow[0], nrow[1]), max(nrow[2], nrow[3]));
#if NL_STAMP_FILL
      fstamp(3);  // words arrived, popcounts, scans
#endif
      if (nmax <= EXPAND_RMAX) {
        // The set bits of a lane are consecutive entries of the row, but one trip through the bit loop writes one
        // entry per lane: ~13 lanes spread over the whole 290-byte row, 5-6 trips per row (8 for a full list).  So
        // the row is put together in LDS first (same loop, ds_write instead of a global store) and leaves as
        // 64-entry runs: 2-3 store instructions per row, each to consecutive addresses.
        int32_t* const cw = lds + CAP + wave * 4 * EXPAND_RMAX;
#pragma unroll
        for (int q = 0; q < 4; q++) ptr[q] += q * EXPAND_RMAX;
        while (word[0] | word[1] | word[2] | word[3]) {
          int32_t val[4];
          bool on[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            on[q] = word[q] != 0;
            const int32_t t = on[q] ? __ffs(word[q]) - 1 : 0;
            val[q] = g[t * WAVE];  // unconditional read of a valid slot: the four reads go out back to back
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (on[q]) {
              cw[ptr[q]] = val[q];
              ptr[q]++;
              word[q] &= word[q] - 1;
            }
          }
        }
#if NL_STAMP_FILL
        fstamp(4);  // bit loop
#endif
        __builtin_amdgcn_wave_barrier();  // the buffer is private to the wave: LDS executes its accesses in order
        for (int32_t e = lane; e - lane < nmax; e += WAVE) {
          int32_t val[4];
#pragma unroll
          for (int q = 0; q < 4; q++) val[q] = cw[q * EXPAND_RMAX + min(e, EXPAND_RMAX - 1)];
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (e < nrow[q]) a.list[(size_t)base(u0 + q) + e] = val[q];
        }
        __builtin_amdgcn_wave_barrier();
#if NL_STAMP_FILL
        fstamp(5);  // rows read back and stored
#endif
        continue;
      }
      while (word[0] | word[1] | word[2] | word[3]) {  // a very long row: straight to memory
        int32_t val[4];
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          on[q] = word[q] != 0;
          const int32_t t = on[q] ? __ffs(word[q]) - 1 : 0;
          val[q] = g[t * WAVE];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (on[q]) {
            a.list[(size_t)base(u0 + q) + ptr[q]] = val[q];
            ptr[q]++;
            word[q] &= word[q] - 1;
          }
        }
      }
    }
  }
#if NL_STAMP_FILL
  if (lane == 0) {
    unsigned long long* const slot = a.dbg_buf + 64 + (blockIdx.x & 1023) * 16;
    for (int ph = 0; ph < 6; ph++) atomicAdd(slot + ph, (unsigned long long)ft_acc[ph]);
    atomicAdd(slot + 9, 1ull);
  }
#endif
}

// The cells k_fill_masks found without masks (local cell indices in full27_list): the rows by a second distance search,
// the batched FILL search of k_sweep<FILL>, a workgroup per listed cell.
template <typename T, bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) k_fill_list(SweepArgs<T> a) {
  constexpr int CAP = sweep_cap<T, MODE_FILL>();
  constexpr bool SCREEN = sizeof(T) == 8;
  __shared__ __attribute__((aligned(32))) char tile_bytes[SCREEN ? screen_lds_bytes(CAP) : CAP * (int)sizeof(Pos<T>)];
  Pos<T>* const tile = reinterpret_cast<Pos<T>*>(tile_bytes);
  if (a.total[0] > a.capacity) return;  // (k_fill_masks has raised ST_CAPACITY)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t count = *a.fill_list_count;
  for (int32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
    if (idx != (int32_t)blockIdx.x) __syncthreads();  // everyone is done with the previous cell's LDS
    const int32_t cell = a.full27_list[idx];
    const int32_t row = cell / a.mx, cx = cell - row * a.mx, cz = row / a.my, cy = row - cz * a.my;
    CellCtx c;
    if (!cell_setup_at(a, lane, cx, cy, cz, c)) continue;
    cell_search<T, MODE_FILL, CAP, SWEEP_WAVES, FULL, PBC, SCREEN>(a, c, tile, tid, lane, wave);
  }
}

// ------------------------------------------------------------------------------------- list from masks, dense cells
// The expansion for builds whose cells hold hundreds of particles (2 x cut-off, BASELINE config 5: 311 per cell, a
// stencil stream of 8400): COUNT_MASKS kept one mask row per (slot, LDS batch), up to FD_NB batches.  One workgroup per
// cell stages the ids of the WHOLE stream (40 KB of LDS), then every wave walks its rows two at a time: the popcounts of
// the row's words give every lane its place (one DPP scan), and the lanes put their set bits of batch b, tile t --
// staged particle b * CAP + t * 64 + lane -- straight into the row (the bit loops run with most lanes live here: ~9 set
// bits per lane and row).  Replaces the second distance sweep (k_sweep<FILL>: 4.4 of the 7.8 ms of a config-5 build).
// Cells whose stream needs more than a.mask_nb batches carry no masks and are searched again, as in k_fill_masks.
constexpr int FD_NB = 7;       // LDS batches of ids a workgroup holds: 35 KB
constexpr int FD_WAVES = 4;
constexpr int FD_RMAX = 1024;  // longest row assembled in LDS (4 KB per wave); longer rows go entry by entry

template <typename T, bool FULL = false, bool PBC = false, typename OFF = int32_t>
__global__ void __launch_bounds__(FD_WAVES* WAVE) k_fill_dense(SweepArgs<T> a, const OFF* __restrict__ base_sorted) {
  constexpr int CAP = SweepCfg<T>::CAP;
  __shared__ __attribute__((aligned(32))) int32_t lds[FD_NB * CAP + FD_WAVES * FD_RMAX];
  int32_t* const gids = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t total = a.total[0];
  CellCtx c;
  const bool ok = cell_setup(a, lane, c);
  if (total > a.capacity) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
    return;
  }
  if (!ok) return;
  const int32_t nbatch = (c.total_j + CAP - 1) / CAP;
  if (nbatch > a.mask_nb) {  // no masks for this cell: search it again through the same LDS
    constexpr int CAPS = (int)(FD_NB * CAP * sizeof(int32_t) / sizeof(Pos<T>)) / WAVE * WAVE;
    cell_search<T, MODE_FILL, CAPS, FD_WAVES, FULL, PBC>(a, c, reinterpret_cast<Pos<T>*>(gids), tid, lane, wave);
    return;
  }
  const int32_t nb = a.mask_nb;
  constexpr int FD_RB = 4;
  // Every load of a batch of rows is issued before the first of them is waited for: a load under a branch (the compiler
  // turns "b < nbatch ? load : 0" into one, the condition being uniform) is waited for at the join, one memory round
  // trip per word -- 28 per batch in the first cut, 74 % of the wave cycles in s_waitcnt.  Hence: unconditional loads
  // of valid rows, an empty asm that takes every loaded value (the loads cannot sink under a branch), then the selects.
  auto load_rows = [&](int32_t r0, uint32_t (&w)[FD_RB][FD_NB], OFF (&base)[FD_RB]) {
#pragma unroll
    for (int u = 0; u < FD_RB; u++) {
      const int32_t slot = c.ibeg + min(r0 + u * FD_WAVES, c.ni - 1);
      base[u] = base_sorted[slot];
#pragma unroll
      for (int b = 0; b < FD_NB; b++) w[u][b] = mask_load(a.masks, a.masks_hi, (size_t)slot * nb + min(b, nb - 1), lane);
    }
#pragma unroll
    for (int u = 0; u < FD_RB; u++)
#pragma unroll
      for (int b = 0; b < FD_NB; b++) asm volatile("" : "+v"(w[u][b]));
#pragma unroll
    for (int u = 0; u < FD_RB; u++)
#pragma unroll
      for (int b = 0; b < FD_NB; b++) w[u][b] = (b < nbatch && r0 + u * FD_WAVES < c.ni) ? (w[u][b] & 0xFFFFFFu) : 0u;
  };
  // A wave's rows come in register batches of FD_RB: all their words are loaded before the first store of the batch
  // (vmcnt retires in order, so a load issued behind stores waits for every one of them: with one row prefetched per
  // row the kernel spent 74 % of its wave cycles in s_waitcnt -- profiles/r02_v2_cfg5_dense_fill_first_cut_pmc.txt).
  uint32_t wb[FD_RB][FD_NB];
  OFF baseb[FD_RB];
  // ids of the whole stream, LDS-DMA (see k_fill_masks)
  for (int32_t sg = wave; sg < NSEG; sg += FD_WAVES) {
    const int32_t len = __builtin_amdgcn_readlane(c.seg_len, sg);
    const int32_t src = __builtin_amdgcn_readlane(c.seg_src, sg);
    const int32_t off = __builtin_amdgcn_readlane(c.seg_off, sg);
    for (int32_t kb = 0; kb < len; kb += WAVE) {
      if (kb + lane < len)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.sorted_gid + src + kb + lane),
                                         (__attribute__((address_space(3))) void*)(gids + off + kb), 4, 0, 0);
    }
  }
  __syncthreads();
  const int32_t* const g = gids + lane;
  int32_t* const cw = lds + FD_NB * CAP + wave * FD_RMAX;
  for (int32_t rb = wave; rb < c.ni; rb += FD_WAVES * FD_RB) {
    load_rows(rb, wb, baseb);
#pragma unroll
    for (int u = 0; u < FD_RB; u++) {
    const int32_t r = rb + u * FD_WAVES;
    if (r >= c.ni) break;  // uniform
    uint32_t (&wd)[FD_NB] = wb[u];
    const OFF base = baseb[u];
    int32_t cnt = 0;
#pragma unroll
    for (int b = 0; b < FD_NB; b++) cnt += __popc(wd[b]);
    const int32_t incl = scan64_dpp(cnt);
    const int32_t nrow = __builtin_amdgcn_readlane(incl, 63);
    uint32_t ptr = (uint32_t)(incl - cnt);
    const bool in_lds = nrow <= FD_RMAX;  // uniform
    // two batches' words at a time: two independent chains of LDS read -> write per trip
#pragma unroll
    for (int b = 0; b < FD_NB; b += 2) {
      if (b >= nbatch) break;  // uniform
      uint32_t w0 = wd[b], w1 = b + 1 < FD_NB ? wd[b + 1] : 0u;
      const int32_t* const g0 = g + b * CAP;
      const int32_t* const g1 = g + (b + 1 < FD_NB ? b + 1 : b) * CAP;
      // (entries of batch b come before those of b + 1 inside a lane's run: the run's order is free, its extent is not)
      uint32_t p0 = ptr, p1 = ptr + (uint32_t)__popc(w0);
      ptr = p1 + (uint32_t)__popc(w1);
      while (__builtin_amdgcn_ballot_w64((w0 | w1) != 0)) {
        const bool on0 = w0 != 0, on1 = w1 != 0;
        const int32_t v0 = g0[(on0 ? __ffs(w0) - 1 : 0) * WAVE];
        const int32_t v1 = g1[(on1 ? __ffs(w1) - 1 : 0) * WAVE];
        if (on0) {
          if (in_lds) cw[p0] = v0;
          else a.list[(size_t)base + p0] = v0;
          p0++, w0 &= w0 - 1;
        }
        if (on1) {
          if (in_lds) cw[p1] = v1;
          else a.list[(size_t)base + p1] = v1;
          p1++, w1 &= w1 - 1;
        }
      }
    }
    if (in_lds) {
      __builtin_amdgcn_wave_barrier();  // the buffer is private to the wave: LDS executes its accesses in order
      for (int32_t e = lane; e < nrow; e += WAVE) a.list[(size_t)base + e] = cw[e];
      __builtin_amdgcn_wave_barrier();
    }
    }
  }
}

}  // namespace nl

#include "nl_lean.hpp"
#include "nl_rows.hpp"

namespace nl {

// base_sorted[slot] = key_pointer[sorted_row[slot]]: the list offset of every row, in cell order, so that the placement
// pass needs one (prefetchable) load per i-particle instead of two dependent ones.
template <typename OFF>
__global__ void __launch_bounds__(256) k_row_base(const OFF* __restrict__ key_pointer,
                                                   const int32_t* __restrict__ sorted_row, int32_t n_rows, int32_t n,
                                                   OFF* __restrict__ base_sorted) {
  const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int32_t r = sorted_row[s];
  // ghosts (slab builds) have no row; in a build that failed its checks a slot may never have been written: unsigned
  // compare, so that whatever it holds is not used as an index
  base_sorted[s] = (uint32_t)r < (uint32_t)n_rows ? key_pointer[r] : (OFF)0;
}

}  // namespace nl
