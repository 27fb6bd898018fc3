// nl_half.hpp -- the half-shell pair search: every unordered PAIR OF CELLS is visited once, every distance tested once.
//
// The reference's scalar class walks, for every cell, the cell itself and 13 of its 26 neighbours
// (MakeNeighMeshId neighlist_cpu.hpp:107-132, MakePairListNaive :239-270) and files an accepted pair under
// min(i, j) (RegistInteractPair :225-236).  The 27-cell sweeps of nl_kernels.hpp test every pair from both sides and
// throw one answer away; here the workgroup of cell A tests its particles i against the staged stream
//     [ A itself | the 13 cells that FOLLOW A in (dz, dy, dx) order ]
// and keeps BOTH outcomes of one distance test:
//   forward  (id_j > id_i, or j in A):  bit (tile, lane) of the F word of row i -- 64 lanes x 16 bits per row, as before
//            but over 14 cells instead of 27;
//   reverse  (id_j < id_i, j in an upper cell B): bit i of the R word of (j, A) -- one 64-bit word per particle j and
//            lower neighbour cell, whose bits run over the particles of that cell.
// The row of a particle p of cell B is then: its F word (B's own sweep) + its 13 R words (written by the 13 cells
// below B).  Counts come from popcounts (F: SGPR counters of the sweep; R: one byte per (particle, lower cell)), the
// rows are expanded from the words by k_fill_half -- no second distance sweep and no atomics on global memory.
//
// Dense cells: an R word has 64 bits, so a cell with more than HS_MAXI particles ("irregular") takes no part: it is
// dropped from its neighbours' streams and does not sweep.  A cell whose 27-cell stencil contains an irregular cell
// (itself included) is built the old way, by a full 27-cell search of its own rows (count here, re-search in the fill
// kernel): rare in the regime this path is selected for (mean <= ~40 per cell: P(n > 64) ~ 1e-4 per cell).
#pragma once

namespace nl {

constexpr int HS_MAXI = 64;                 // most particles a regular cell holds (bits of an R word)
constexpr int HS_NUP = 13;                  // upper (= lower) neighbour cells
constexpr int HS_NSEG = HS_NUP + 1;         // segments of the staged stream: own cell + upper cells
constexpr int HS_CAP = HS_NSEG * HS_MAXI;   // 896 staged particles = 14 tiles of 64: bits 0..13 of an F word
constexpr uint32_t HS_FINAL = 0x80000000u;  // fcnt: the row's count was written by the 27-cell search (irregular stencil)

// Offset of upper neighbour k: the 13 cells that follow (0,0,0) in (dz, dy, dx) lexicographic order.  (The reference
// takes the 13 that precede it, neighlist_cpu.hpp:107-132: the same set of cell pairs, visited from the other side.)
__device__ __forceinline__ void hs_offset(int k, int32_t& dx, int32_t& dy, int32_t& dz) {
  if (k == 0) dx = 1, dy = 0, dz = 0;
  else if (k < 4) dx = k - 2, dy = 1, dz = 0;
  else dx = (k - 4) % 3 - 1, dy = (k - 4) / 3 - 1, dz = 1;
}

// What a workgroup knows about its cell.  Per lane: lane 0 = the cell itself, lanes 1..13 = upper cell k = lane - 1,
// lanes 14..26 = lower cell k = lane - 14 (the cell this one is upper neighbour k of).
struct HalfCtx {
  int32_t ibeg, ni;                     // the cell's own particles
  int32_t cell_src, cell_len;           // per lane: that stencil cell's particles in the sorted array (0 if it does not exist)
  int32_t seg_len, seg_off, total_j;    // per lane < 14: its share of the staged stream (0 when dropped); stream length
  int32_t wrap;                         // per lane: periodic faces crossed on the way to that cell, as CellCtx::wrap
  int32_t cx, cy, cz;
  bool own_regular;                     // ni <= HS_MAXI: this cell sweeps (and is part of its lower neighbours' streams)
  bool owned;                           // rows of this cell belong to this rank (slab builds: not a ghost layer)
  bool full27;                          // owned and some cell of the 27-cell stencil is irregular: rows by 27-cell search
};

// The table in two steps, so that its loads can be in flight while other work runs: half_issue computes, per lane, the
// stencil cell and loads its two cell_start entries; half_finish turns them into the HalfCtx.
struct HalfRaw {
  int32_t beg, end;   // per lane: cell_start[c], cell_start[c + 1] of that stencil cell (0, 0 if it does not exist)
  int32_t wrap;
  int32_t cx, cy, cz;
};

template <typename T> __device__ __forceinline__ void half_issue_at(const SweepArgs<T>& a, int lane, int32_t cx, int32_t cy, int32_t cz, HalfRaw& r) {
  r.cx = cx, r.cy = cy, r.cz = cz;
  int32_t dx = 0, dy = 0, dz = 0;
  if (lane >= 1 && lane < 14) {
    hs_offset(lane - 1, dx, dy, dz);
  } else if (lane >= 14 && lane < 27) {
    hs_offset(lane - 14, dx, dy, dz);
    dx = -dx, dy = -dy, dz = -dz;
  }
  int32_t x = cx + dx, y = cy + dy, z = cz + dz, wx = 0, wy = 0, wz = 0;
  if (x < 0) x += a.mx, wx = -1;
  if (x >= a.mx) x -= a.mx, wx = 1;
  if (y < 0) y += a.my, wy = -1;
  if (y >= a.my) y -= a.my, wy = 1;
  bool exists = lane < 27;
  if (a.slab) {  // no wrap in z: the ghost layers stand in for the periodic neighbours
    if (z < 0 || z >= a.mzl) exists = false;
  } else {
    if (z < 0) z += a.mzl, wz = -1;
    if (z >= a.mzl) z -= a.mzl, wz = 1;
  }
  r.beg = 0, r.end = 0;
  if (exists) {
    const int32_t cell = x + (y + z * a.my) * a.mx;
    r.beg = a.cell_start[cell];
    r.end = a.cell_start[cell + 1];
  }
  r.wrap = (wx + 1) | (wy + 1) << 2 | (wz + 1) << 4;
}

template <typename T> __device__ __forceinline__ void half_finish(const SweepArgs<T>& a, int lane, const HalfRaw& r, HalfCtx& c) {
  c.cx = r.cx, c.cy = r.cy, c.cz = r.cz;
  c.cell_src = r.beg, c.cell_len = r.end - r.beg, c.wrap = r.wrap;
  const uint64_t irregular = __builtin_amdgcn_ballot_w64(c.cell_len > HS_MAXI);
  c.ibeg = __builtin_amdgcn_readlane(c.cell_src, 0);
  c.ni = __builtin_amdgcn_readlane(c.cell_len, 0);
  c.own_regular = c.ni <= HS_MAXI;
  c.owned = !a.slab || (c.cz >= 1 && c.cz <= a.mzl - 2);
  c.full27 = c.owned && irregular != 0 && c.ni > 0;
  // The stream: own cell + the regular upper cells.  A cell of the lower ghost layer (slab builds, cz == 0) sweeps only
  // as the lower neighbour of layer 1: its dz = +1 cells.
  bool in_stream = lane < HS_NSEG && c.cell_len <= HS_MAXI;
  if (a.slab && c.cz == 0 && lane < 5) in_stream = false;  // the cell itself and upper cells 0..3 (dz = 0)
  c.seg_len = in_stream ? c.cell_len : 0;
  c.seg_off = scan32_dpp(c.seg_len) - c.seg_len;
  c.total_j = __builtin_amdgcn_readlane(c.seg_off + c.seg_len, HS_NSEG - 1);
}

template <typename T> __device__ __forceinline__ void half_setup_at(const SweepArgs<T>& a, int lane, int32_t cx, int32_t cy, int32_t cz, HalfCtx& c) {
  HalfRaw r;
  half_issue_at(a, lane, cx, cy, cz, r);
  half_finish(a, lane, r, c);
}

// Linear cell index w of the half-shell sweep -> cell: every local layer except, in a slab build, the upper ghost layer
// (nothing lies above it on this rank).  The sweep covers mx * my * (slab ? mzl - 1 : mzl) cells.
template <typename T> __device__ __forceinline__ void half_issue(const SweepArgs<T>& a, int lane, int32_t w, HalfRaw& r) {
  const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), cx = w - wy * a.mx;
  const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - wz * a.my;
  half_issue_at(a, lane, cx, cy, wz, r);
}

// One group of GC i-particles (SGPRs) against the staged stream (lanes own j), both outcomes of every test kept:
//   bits[k] : the F word of i-particle k being built, one bit per tile (add-with-carry, as search_group);
//   racc    : per lane, the reverse hits of the GC tests of the current tile, shifted in the same way; after each tile
//             its low GC bits are OR-ed into the lane's R word in LDS at bit i0 (ds_or_b64: the four waves of the
//             workgroup own different i, i.e. different bits of the same words).
// FULL (both directions of every pair are listed): forward and reverse are the same mask, no id test.
// Returns, in lane k, the forward count of i-particle k.
template <typename T, int GC, bool FULL>
__device__ __forceinline__ int32_t search_group_half(const SweepArgs<T>& a, const Pos<T>* tile, unsigned long long* rw,
                                                     int32_t ntiles, int lane, const Pos<T>& pi_l, int32_t i0, int32_t slot0,
                                                     bool write_f) {
  T xi[GC], yi[GC], zi[GC];
  int32_t gi[GC];
  uint32_t cur[GC], bits[GC];
  uint32_t racc = 0;
#pragma unroll
  for (int k = 0; k < GC; k++) {
    if constexpr (sizeof(T) == 4) {
      xi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.x), k));
      yi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.y), k));
      zi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.z), k));
    } else {
      xi[k] = __shfl(pi_l.x, k, WAVE);
      yi[k] = __shfl(pi_l.y, k, WAVE);
      zi[k] = __shfl(pi_l.z, k, WAVE);
    }
    gi[k] = __builtin_amdgcn_readlane(pi_l.gid, k);
    cur[k] = 0u, bits[k] = 0u;
  }
  auto test_tile = [&](const Pos<T>& pj, int32_t t) {
    uint64_t fw[GC], rv[GC];
#pragma unroll
    for (int k = 0; k < GC; k++) {
      const T dx = sub_rn(pj.x, xi[k]), dy = sub_rn(pj.y, yi[k]), dz = sub_rn(pj.z, zi[k]);
      const T r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
      const uint64_t in = __builtin_amdgcn_ballot_w64(!(r2 > a.rc2));
      if (FULL) {
        fw[k] = rv[k] = in;
      } else {  // the pair belongs to the row of the smaller id (neighlist_cpu.hpp:225-236)
        const uint64_t up = __builtin_amdgcn_ballot_w64(pj.gid > gi[k]);
        fw[k] = in & up, rv[k] = in & ~up;
      }
    }
    uint64_t carry_out;
#pragma unroll
    for (int k = GC - 1; k >= 0; k--)  // i-particle k ends at bit k of racc
      asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(racc), "=s"(carry_out) : "s"(rv[k]));
#pragma unroll
    for (int k = 0; k < GC; k++) {
      asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits[k]), "=s"(carry_out) : "s"(fw[k]));
      cur[k] += (uint32_t)__popcll(fw[k]);
    }
    const unsigned long long w = (unsigned long long)(racc & ((1u << GC) - 1u)) << i0;
    if (!(a.dbg & 64)) (void)__hip_atomic_fetch_or(&rw[t * WAVE + lane], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  const int32_t last = (ntiles - 1) * WAVE + lane;
  Pos<T> pa = tile[lane], pb;
  int32_t t = 0;
#if NL_PRIO
  __builtin_amdgcn_s_setprio(0);  // (as search_group: the tile loop yields to waves that are setting up, staging, storing)
#endif
  for (; t + 1 < ntiles; t += 2) {
    pb = tile[(t + 1) * WAVE + lane];
    test_tile(pa, t);
    pa = tile[min((t + 2) * WAVE + lane, last)];
    test_tile(pb, t + 1);
  }
  if (t < ntiles) test_tile(pa, t);
#if NL_PRIO
  __builtin_amdgcn_s_setprio(NL_PRIO);
#endif
  if (write_f) {
#pragma unroll
    for (int k = 0; k < GC; k++) {
      uint32_t w = __brev(bits[k]) >> (32 - ntiles);
      // FULL: the row's own particle (distance 0) sits at stream position i0 + k (the cell itself is staged first)
      if (FULL && lane == i0 + k) w &= ~1u;
      a.fmask[(size_t)(slot0 + k) * WAVE + lane] = (uint16_t)w;
    }
  }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < GC; k++) mine = lane == k ? cur[k] - (FULL ? 1u : 0u) : mine;
  return (int32_t)mine;
}

// The sweep of one cell whose stream is already staged (tile, rw zeroed): search, R words, 27-cell fall-back.
template <typename T, bool FULL, bool PBC>
__device__ __forceinline__ void half_search_cell(const SweepArgs<T>& a, const HalfCtx& c, Pos<T>* tile, unsigned long long* rw,
                                                 int tid, int lane, int wave) {
  if (c.own_regular) {
    const int32_t ni = c.ni, ntiles = (c.total_j + WAVE - 1) / WAVE;
    if (ni > 0 && ntiles > 0 && !(a.dbg & 1)) {
      constexpr int G = SWEEP_G, NW = SWEEP_WAVES;
      const int32_t rounds = (ni + NW * G - 1) / (NW * G), ngroups = rounds * NW, gsize = (ni + ngroups - 1) / ngroups;
      const bool write_f = c.owned && !c.full27;
      for (int32_t g = wave; g < ngroups; g += NW) {
        const int32_t i0 = g * gsize, gcount = min(gsize, ni - i0);
        if (gcount <= 0) break;
        Pos<T> pi_l;
        pi_l.x = 0, pi_l.y = 0, pi_l.z = 0, pi_l.gid = 0;
        // the cell's own particles open the staged stream (except in a slab's lower ghost layer, whose cells keep
        // only their dz = +1 part): no second global round trip for them
        if (lane < gcount) pi_l = (a.slab && c.cz == 0) ? a.sorted[c.ibeg + i0 + lane] : tile[i0 + lane];
        int32_t mine;
        switch (gcount) {
          case 1: mine = search_group_half<T, 1, FULL>(a, tile, rw, ntiles, lane, pi_l, i0, c.ibeg + i0, write_f); break;
          case 2: mine = search_group_half<T, 2, FULL>(a, tile, rw, ntiles, lane, pi_l, i0, c.ibeg + i0, write_f); break;
          case 3: mine = search_group_half<T, 3, FULL>(a, tile, rw, ntiles, lane, pi_l, i0, c.ibeg + i0, write_f); break;
          case 4: mine = search_group_half<T, 4, FULL>(a, tile, rw, ntiles, lane, pi_l, i0, c.ibeg + i0, write_f); break;
          default: mine = search_group_half<T, 5, FULL>(a, tile, rw, ntiles, lane, pi_l, i0, c.ibeg + i0, write_f); break;
        }
        if (write_f && lane < gcount) a.fcnt[c.ibeg + i0 + lane] = (uint32_t)mine;
      }
    }
    __syncthreads();
    // ---- the R words of the upper cells' particles (zeros from an empty cell: every (particle, lower cell) entry is
    // written by exactly one workgroup per build)
    for (int32_t sg = 1 + wave; sg < HS_NSEG && !(a.dbg & 2); sg += SWEEP_WAVES) {
      const int32_t len = __builtin_amdgcn_readlane(c.seg_len, sg);
      if (lane < len) {
        const unsigned long long w = rw[__builtin_amdgcn_readlane(c.seg_off, sg) + lane];
        const size_t at = (size_t)(sg - 1) * a.rstride + __builtin_amdgcn_readlane(c.cell_src, sg) + lane;
        a.rmask[at] = w;
        a.rcnt[at] = (uint8_t)__popcll(w);
      }
    }
  }
  // Some cell of this cell's stencil holds more than HS_MAXI particles: its rows are counted and filled by the 27-cell
  // search of k_full27 (a kernel of its own: inlined here, its registers would be this kernel's); the words written
  // above are not used for them.
  if (c.full27) {
    for (int32_t p = tid; p < c.ni; p += SWEEP_WAVES * WAVE) a.fcnt[c.ibeg + p] = HS_FINAL;
    if (tid == 0) a.full27_list[atomicAdd(a.full27_count, 1)] = c.cx + (c.cy + c.cz * a.my) * a.mx;  // local cell index
  }
}

// One workgroup per cell (a.cells_per_block > 1: several consecutive cells one after another -- measured slower, with and
// without prefetching the next cell's table and particles into registers; kept as a diagnostic knob, NL_HALF_CPB).
// All loads of the staging step are issued before the first LDS write: a load followed by its own LDS write inside
// the segment loop waited for memory once per segment -- 30 % of a wave's life (tools/half_phases.py).
template <typename T, bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE, (sizeof(T) == 4 ? 7 : 4)) k_sweep_half(SweepArgs<T> a) {
  __shared__ Pos<T> tile[HS_CAP];
  __shared__ unsigned long long rw[HS_CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NPRE = (HS_NSEG + SWEEP_WAVES - 1) / SWEEP_WAVES;  // segments per wave: wave, wave + 4, ...
#if NL_PRIO
  __builtin_amdgcn_s_setprio(NL_PRIO);
#endif
  const uint32_t st_word = *a.status;  // (read with the first cell table)
  const int32_t w0 = xcd_cell_index() * a.cells_per_block, w1 = min(w0 + a.cells_per_block, a.ncells_grid);
  // diagnostics (NL_DEBUG_FLAGS & 4, tools/half_phases.py): shader cycles of every wave per phase, summed into dbg_buf[8 + phase]
  uint64_t tprev = (a.dbg & 4) ? __builtin_amdgcn_s_memtime() : 0;
  auto stamp = [&](int phase, bool drain) {
    if (a.dbg & 4) {
      if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const uint64_t now = __builtin_amdgcn_s_memtime();
      if (lane == 0) atomicAdd(a.dbg_buf + 8 + phase, (unsigned long long)(now - tprev));
      tprev = now;
    }
  };
  for (int32_t w = w0; w < w1; w++) {
    if (w != w0) __syncthreads();  // everyone is done with the previous cell's LDS
    HalfRaw raw;
    HalfCtx c;
    half_issue(a, lane, w, raw);
    half_finish(a, lane, raw, c);
    stamp(0, true);
    if (st_word & ST_DOMAIN) return;  // an inconsistent cell table (see cell_setup_at): nobody walks it
    if (c.own_regular) {
      Pos<T> pre[NPRE];
      int32_t plen[NPRE], poff[NPRE];
#pragma unroll
      for (int u = 0; u < NPRE; u++) {  // (unconditional loads of a valid slot: the registers stay registers)
        const int sg = min(wave + SWEEP_WAVES * u, HS_NSEG - 1);
        plen[u] = wave + SWEEP_WAVES * u < HS_NSEG ? __builtin_amdgcn_readlane(c.seg_len, sg) : 0;
        poff[u] = __builtin_amdgcn_readlane(c.seg_off, sg);
        pre[u] = a.sorted[__builtin_amdgcn_readlane(c.cell_src, sg) + (lane < plen[u] ? lane : 0)];
      }
      const int32_t total_j = c.total_j, padded = (total_j + WAVE - 1) & ~(WAVE - 1);
      for (int32_t s = tid; s < padded; s += SWEEP_WAVES * WAVE) rw[s] = 0ull;
      if (total_j + tid < padded) {  // sentinels up to the tile boundary: never in range
        Pos<T> sentinel;
        sentinel.x = sizeof(T) == 4 ? (T)1.0e18f : (T)1.0e150, sentinel.y = 0, sentinel.z = 0, sentinel.gid = INT32_MIN;
        if constexpr (sizeof(T) == 8) sentinel.row = 0;
        tile[total_j + tid] = sentinel;
      }
#pragma unroll
      for (int u = 0; u < NPRE; u++) {
        if (lane < plen[u]) {
          Pos<T> v = pre[u];
          if (PBC) {  // a cell reached through a periodic face is staged at its image
            const int32_t wr = __builtin_amdgcn_readlane(c.wrap, min(wave + SWEEP_WAVES * u, HS_NSEG - 1));
            if (wr != 0x15) {
              v.x = add_rn(v.x, (T)((wr & 3) - 1) * a.L[0]);
              v.y = add_rn(v.y, (T)(((wr >> 2) & 3) - 1) * a.L[1]);
              v.z = add_rn(v.z, (T)(((wr >> 4) & 3) - 1) * a.L[2]);
            }
          }
          tile[poff[u] + lane] = v;
        }
      }
    }
    stamp(1, true);
    __syncthreads();
    stamp(2, false);
    half_search_cell<T, FULL, PBC>(a, c, tile, rw, tid, lane, wave);
    stamp(3, false);
    if ((a.dbg & 4) && lane == 0) atomicAdd(a.dbg_buf + 15, 1ull);
  }
}

// number_of_partners in original order from the per-slot pieces: forward count + the 13 reverse counts.
__global__ void __launch_bounds__(256) k_half_counts(const uint32_t* __restrict__ fcnt, const uint8_t* __restrict__ rcnt,
                                                      int64_t rstride, const int32_t* __restrict__ sorted_row, int32_t n_rows,
                                                      int32_t n, int32_t* __restrict__ count) {
  const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int32_t row = sorted_row[s];
  if ((uint32_t)row >= (uint32_t)n_rows) return;  // a ghost (slab builds)
  const uint32_t f = fcnt[s];
  if (f & HS_FINAL) return;  // counted by the 27-cell search
  uint32_t cnt = f;
#pragma unroll
  for (int k = 0; k < HS_NUP; k++) cnt += rcnt[(size_t)k * rstride + s];
  count[row] = (int32_t)cnt;
}

}  // namespace nl

namespace nl {

// ---------------------------------------------------------------------------------------- rows from F and R words
// The placement pass of the half-shell search: one workgroup per cell B.  Row p of B = the set bits of its F word
// (64 lanes x <= 14 bits over B's own staged stream: B + its 13 upper cells) + the set bits of its 13 R words (64 bits
// each, over the particles of one lower cell).  Only ids are staged (LDS-DMA from the compact sorted_gid): the upper
// stream exactly as the sweep laid it out, the lower cells at 64 ids apiece.  A wave loads the words and row offsets of
// all its rows first (one memory round trip), then expands four rows at a time into an LDS copy of the rows -- F bits
// with lane = stream lane (as k_fill_masks), R bits with lane = (row, lower cell, half word) -- which leaves as runs of
// 64 consecutive entries.  Cells with an irregular stencil are searched again with the 27-cell search, as in
// k_fill_masks.
constexpr int HF_WAVES = 4;      // waves per workgroup
constexpr int HF_RB = HS_MAXI / HF_WAVES;  // rows per wave (a regular cell has <= 64)
constexpr int HF_RMAX = 160;     // longest row assembled in LDS (longer ones are written entry by entry)
constexpr int HF_LDS_WORDS = HS_CAP + HS_NUP * HS_MAXI + HF_WAVES * 4 * HF_RMAX;

template <typename T, bool FULL, bool PBC, typename OFF>
__device__ __forceinline__ void fill_half_cell(const SweepArgs<T>& a, const OFF* __restrict__ base_sorted, int32_t* lds, int32_t w,
                                               int tid, int lane, int wave) {
  int32_t* const gid_u = lds;                        // ids of the upper stream (cell itself first)
  int32_t* const gid_l = lds + HS_CAP;               // ids of lower cell k at [k * 64, k * 64 + n_k)
  // w -> owned cell, as cell_setup
  const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), cx = w - wy * a.mx;
  const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
  const int64_t total = a.total[0];  // (read with the cell table: one round trip)
  const uint32_t st_word = *a.status;
  HalfCtx c;
  half_setup_at(a, lane, cx, cy, cz, c);
  if (total > a.capacity) {  // the list is too small: the host grows it and runs this kernel again
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
    return;
  }
  if (c.ni <= 0 || (st_word & ST_DOMAIN) || c.full27) return;  // (full27: rows by k_full27)

  // ---- this wave's rows (whole batches of four where possible): words and offsets, all loads before anything else
  const int32_t nbatch = (c.ni + 3) >> 2, bpw = nbatch / HF_WAVES, extra = nbatch % HF_WAVES;
  const int32_t r_beg = min(4 * (wave * bpw + min(wave, extra)), c.ni);
  const int32_t r_end = min(r_beg + 4 * (bpw + (wave < extra ? 1 : 0)), c.ni);
  uint32_t fw[HF_RB / 2];   // F words of rows (2v, 2v+1), this lane's 16 bits each: low / high half
  uint32_t rwd[HF_RB / 2];  // R half word of rows (2v, 2v+1): lane = (row & 1) * 32 + k * 2 + half
  const int32_t rk = (lane & 31) >> 1, rhalf = lane & 1, rq = lane >> 5;
#pragma unroll
  for (int v = 0; v < HF_RB / 2; v++) {
    fw[v] = 0;
    if (r_beg + 2 * v < r_end) {  // wave-uniform
      const int32_t slot = c.ibeg + r_beg + 2 * v, slot1 = min(slot + 1, c.ibeg + c.ni - 1);
      fw[v] = (uint32_t)a.fmask[(size_t)slot * WAVE + lane] | (uint32_t)a.fmask[(size_t)slot1 * WAVE + lane] << 16;
    }
  }
#pragma unroll
  for (int v = 0; v < HF_RB / 2; v++) {
    rwd[v] = 0;
    if (r_beg + 2 * v < r_end) {  // wave-uniform
      const int32_t slot = c.ibeg + min(r_beg + 2 * v + rq, c.ni - 1);
      rwd[v] = rk < HS_NUP ? reinterpret_cast<const uint32_t*>(a.rmask)[((size_t)rk * a.rstride + slot) * 2 + rhalf] : 0u;
    }
  }
  // a lower cell that does not exist (slab: none below the lower ghost layer -- not reached for owned cells) or is empty
  // wrote nothing: its words are not looked at (lengths from the table)
  const int32_t low_len = __shfl(c.cell_len, 14 + min(rk, HS_NUP - 1), WAVE);

  // ---- stage the ids: 14 upper segments + 13 lower cells, one LDS-DMA instruction each
  for (int32_t sg = wave; sg < 27; sg += HF_WAVES) {
    const int32_t len = sg < HS_NSEG ? __builtin_amdgcn_readlane(c.seg_len, sg) : __builtin_amdgcn_readlane(c.cell_len, sg);
    const int32_t src = __builtin_amdgcn_readlane(c.cell_src, sg);
    int32_t* const dst = sg < HS_NSEG ? gid_u + __builtin_amdgcn_readlane(c.seg_off, sg) : gid_l + (sg - HS_NSEG) * HS_MAXI;
    if (lane < len)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.sorted_gid + src + lane),
                                       (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
  }
  __syncthreads();

  int32_t* const cw = lds + HS_CAP + HS_NUP * HS_MAXI + wave * 4 * HF_RMAX;
  const int32_t* const gu = gid_u + lane;
  const int32_t* const gl = gid_l + rk * HS_MAXI + rhalf * 32;
#pragma unroll
  for (int u0 = 0; u0 < HF_RB; u0 += 4) {
    if (r_beg + u0 >= r_end) continue;  // wave-uniform
    uint32_t word[4], ptr[4], rword[2], rptr[2];
    int32_t nf[4], nrow[4];
    OFF base[4];  // the rows' list offsets: loaded now, used after the bit loops
#pragma unroll
    for (int q = 0; q < 4; q++) base[q] = base_sorted[c.ibeg + min(r_beg + u0 + q, c.ni - 1)];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      word[q] = r_beg + u0 + q < r_end ? ((fw[(u0 + q) / 2] >> (16 * (q & 1))) & 0xFFFFu) : 0u;
      const int32_t cnt = __popc(word[q]);
      const int32_t incl = scan64_dpp(cnt);
      nf[q] = __builtin_amdgcn_readlane(incl, 63);
      ptr[q] = (uint32_t)(incl - cnt);
    }
#pragma unroll
    for (int v = 0; v < 2; v++) {  // rows (u0 + 2v, u0 + 2v + 1): lanes 0..31 / 32..63
      const bool live = r_beg + u0 + 2 * v + rq < r_end && rk < HS_NUP && low_len > 0;
      rword[v] = live ? rwd[u0 / 2 + v] : 0u;
      // (bits beyond the lower cell's particle count are never set: the sweep only tests real particles)
      const int32_t cnt = __popc(rword[v]);
      const int32_t incl = scan32_dpp(cnt);  // independent scans of the two half waves
      const int32_t n_lo = __builtin_amdgcn_readlane(incl, 31), n_hi = __builtin_amdgcn_readlane(incl, 63);
      nrow[2 * v] = nf[2 * v] + n_lo, nrow[2 * v + 1] = nf[2 * v + 1] + n_hi;
      rptr[v] = (uint32_t)((rq ? nf[2 * v + 1] : nf[2 * v]) + incl - cnt);
    }
    const int32_t nmax = max(max(nrow[0], nrow[1]), max(nrow[2], nrow[3]));
    if (nmax <= HF_RMAX) {
#pragma unroll
      for (int q = 0; q < 4; q++) ptr[q] += q * HF_RMAX;
#pragma unroll
      for (int v = 0; v < 2; v++) rptr[v] += (2 * v + rq) * HF_RMAX;
      while (word[0] | word[1] | word[2] | word[3]) {  // F bits: bit t of lane l = upper stream particle t * 64 + l
        int32_t val[4];
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          on[q] = word[q] != 0;
          const int32_t t = on[q] ? __ffs(word[q]) - 1 : 0;
          val[q] = gu[t * WAVE];
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
      while (rword[0] | rword[1]) {  // R bits: bit b of (lower cell k, half h) = particle 32 h + b of that cell
        int32_t val[2];
        bool on[2];
#pragma unroll
        for (int v = 0; v < 2; v++) {
          on[v] = rword[v] != 0;
          const int32_t b = on[v] ? __ffs(rword[v]) - 1 : 0;
          val[v] = gl[b];
        }
#pragma unroll
        for (int v = 0; v < 2; v++) {
          if (on[v]) {
            cw[rptr[v]] = val[v];
            rptr[v]++;
            rword[v] &= rword[v] - 1;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();  // the buffer is private to the wave: LDS executes its accesses in order
      for (int32_t e = lane; e - lane < nmax; e += WAVE) {
        int32_t val[4];
#pragma unroll
        for (int q = 0; q < 4; q++) val[q] = cw[q * HF_RMAX + min(e, HF_RMAX - 1)];
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (e < nrow[q]) a.list[(size_t)base[q] + e] = val[q];
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // a very long row among the four: straight to memory
    while (word[0] | word[1] | word[2] | word[3]) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (word[q]) {
          const int32_t t = __ffs(word[q]) - 1;
          a.list[(size_t)base[q] + ptr[q]] = gu[t * WAVE];
          ptr[q]++;
          word[q] &= word[q] - 1;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const OFF b_lo = base[2 * v], b_hi = base[2 * v + 1];
      const OFF mybase = rq ? b_hi : b_lo;
      while (rword[v]) {
        const int32_t b = __ffs(rword[v]) - 1;
        a.list[(size_t)mybase + rptr[v]] = gl[b];
        rptr[v]++;
        rword[v] &= rword[v] - 1;
      }
    }
  }
}

template <typename T, bool FULL = false, bool PBC = false, typename OFF = int32_t>
__global__ void __launch_bounds__(HF_WAVES* WAVE, (sizeof(OFF) == 4 ? 7 : 6)) k_fill_half(SweepArgs<T> a, const OFF* __restrict__ base_sorted) {
  __shared__ __attribute__((aligned(32))) int32_t lds[HF_LDS_WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t w0 = xcd_cell_index() * a.cells_per_block, w1 = min(w0 + a.cells_per_block, a.ncells_grid);
  for (int32_t w = w0; w < w1; w++) {  // several cells per workgroup, as k_sweep_half
    if (w != w0) __syncthreads();
    fill_half_cell<T, FULL, PBC, OFF>(a, base_sorted, lds, w, tid, lane, wave);
  }
}

// The cells the half-shell kernels leave out (an irregular cell in their stencil; k_sweep_half lists them): the 27-cell
// search of nl_kernels.hpp, MODE_COUNT after the sweep (counts straight into number_of_partners) and MODE_FILL after
// k_fill_half.  A fixed grid walks the list; with no such cell every workgroup leaves after one load.
template <typename T, int MODE, bool FULL = false, bool PBC = false>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) k_full27(SweepArgs<T> a) {
  __shared__ Pos<T> tile[HS_CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t n = *a.full27_count;
  if (n <= 0) return;
  if (MODE == MODE_FILL && a.total[0] > a.capacity) return;  // (k_fill_half has raised ST_CAPACITY)
  if (*a.status & ST_DOMAIN) return;
  for (int32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const int32_t cell = a.full27_list[e];
    const int32_t wy = (int32_t)fastdiv((uint32_t)cell, a.div_mx), cx = cell - wy * a.mx;
    const int32_t cz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - cz * a.my;
    CellCtx c27;
    if (e != (int32_t)blockIdx.x) __syncthreads();  // the previous cell's LDS is free
    if (cell_setup_at(a, lane, cx, cy, cz, c27)) cell_search<T, MODE, HS_CAP, SWEEP_WAVES, FULL, PBC>(a, c27, tile, tid, lane, wave);
  }
}

}  // namespace nl
